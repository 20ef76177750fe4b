from torchain_amd.io import *  # noqa: F401,F403
from torchain_amd.io import DenominatorGraph, Supervision, set_kaldi_device  # noqa: F401
