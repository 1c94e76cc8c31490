"""Net factory -- interface of src/network/network.py:5-11."""
from .CleanUMamba import CleanUMamba


def Net(network, net_config):
    if network == "CleanUMamba":
        return CleanUMamba(**net_config)
    raise NotImplementedError
