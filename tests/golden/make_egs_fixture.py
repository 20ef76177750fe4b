"""Writes tests/golden/chain_egs.ark / .scp / .json (run once from the repo root: python tests/golden/make_egs_fixture.py)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np  # noqa: E402

import kaldi_egs_writer as kw  # noqa: E402
from oracle import pyoracle  # noqa: E402
from test_egs import make_example  # noqa: E402
from torchain_amd import egs, synth  # noqa: E402

pyoracle.build()
fst = synth.random_den_fst(60, 4, 32, seed=11)
keyed = [("fix%02d" % i, make_example(fst, L, seed=100 + i, final_weights=(i % 2 == 0))) for i, L in enumerate([6, 6, 9, 6])]
ark = os.path.join(HERE, "chain_egs.ark")
kw.write_ark(ark, keyed, scp_path=None, matrix_kind="CM", dw="DW2")
with open(os.path.join(HERE, "chain_egs.scp"), "w") as f:
    off = 0
    blob = open(ark, "rb").read()
    for key, _ in keyed:
        pos = blob.index(key.encode() + b" ", off) + len(key) + 1
        f.write("%s tests/golden/chain_egs.ark:%d\n" % (key, pos))
        off = pos
got = list(egs.iter_archive(ark))
merge_len, y_seed = 6, 77
merged = egs.merge_chain_examples([eg for _, eg in got if eg["outputs"][0]["supervision"].frames_per_sequence == merge_len])
sup = merged["outputs"][0]["supervision"]
y = synth.random_nnet_output(sup.num_sequences, sup.frames_per_sequence, sup.label_dim, seed=y_seed)
meta = dict(keys=[k for k, _ in got], merge_len=merge_len, y_seed=y_seed,
            merged_num_logprob=float(pyoracle.num_forward_backward(sup, y)["logprob_weighted"]),
            examples=[dict(sup=[int(s.num_sequences), int(s.frames_per_sequence), int(s.label_dim), int(s.num_states), int(s.arc_begin[-1])],
                           feat_sum=float(eg["inputs"][0]["features"].sum()), arc_weight_sum=float(s.arc_weight.sum()))
                      for _, eg in got for s in [eg["outputs"][0]["supervision"]]])
json.dump(meta, open(os.path.join(HERE, "chain_egs.json"), "w"), indent=1)
print("wrote", ark, os.path.getsize(ark), "bytes")
