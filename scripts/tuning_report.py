"""Development aid (GPU box): what tc_den_graph_tuning decides for the named workloads' graphs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from torchain_amd import io, synth  # noqa: E402

for name in sys.argv[1:] or ["C3", "C5", "R1", "R2"]:
    fst = synth.config_den_fst(name)
    g = io.DenominatorGraph(fst, synth.CONFIGS[name]["P"]).prepare(0)
    print(name, g.tuning(0))
