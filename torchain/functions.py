from torchain_amd.functions import ChainResults, _ChainLoss, chain_loss, to2d  # noqa: F401
