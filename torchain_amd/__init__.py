"""MI355X-native LF-MMI chain loss with torchain's ``chain_loss`` / ``ChainResults`` API.

Drop-in for one path of nttcslab-sp/torchain: ``torchain/functions.py`` (autograd wrapper),
``torchain/io.py`` (the two handles) and ``src/my_lib_chain.cpp`` (the Kaldi bridge), re-built as
hand-written HIP kernels for gfx950 behind a C ABI (include/torchain_hip.h).
"""
from . import io  # noqa: F401
from .functions import ChainResults, chain_loss  # noqa: F401
