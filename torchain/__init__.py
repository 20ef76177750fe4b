"""Import shim: lets the reference's callers (``from torchain import io``,
``from torchain.functions import chain_loss, ChainResults``; example/chime5/train.py:11-12) run on
the MI355X-native implementation in ``torchain_amd`` unchanged."""
from torchain_amd import io  # noqa: F401
from torchain_amd import functions  # noqa: F401
