"""Drop-in for the names the reference imports from mamba-ssm 1.2.2
(src/network/CleanUMamba.py:12,14).  See INTEGRATION.md."""
