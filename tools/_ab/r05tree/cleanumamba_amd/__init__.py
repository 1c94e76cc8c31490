"""cleanumamba_amd -- MI355X-native CleanUMamba forward+backward hot path.

Host side mirrors the reference's interfaces for this path only:
  cleanumamba_amd.network      <- src/network/{network,CleanUMamba,layers}.py
  cleanumamba_amd.mamba_ssm    <- the slice of mamba-ssm 1.2.2 the reference imports
  cleanumamba_amd.causal_conv1d<- the slice of causal-conv1d 1.1.0 it uses
  cleanumamba_amd.training     <- src/training/train_distributed.py (grad exchange) + train step
  cleanumamba_amd.util         <- src/util/{util,stft_loss}.py pieces on the measured step
The arithmetic is in csrc/ (HIP, gfx950) behind the C ABI of include/cleanumamba_hip.h.
"""
__version__ = "0.1.0"
