"""Step time of chain_loss + backward when the minibatches come from RandExample (development aid, GPU box).

Writes a synthetic chain-egs archive (C2's graph, one-sequence examples of 150 frames), then times the training-side
loop three ways: the same supervision every step (no reader), RandExample with its background look-ahead, and
RandExample reading synchronously.  What is timed is everything between two steps: reading + merging (reference
src/my_lib_example_rand.cpp:35-177), Supervision.from_synth, the upload and the loss."""
import os
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import kaldi_egs_writer as kw  # noqa: E402
from torchain_amd import io, synth  # noqa: E402
from torchain_amd.functions import chain_loss  # noqa: E402

name, n_egs, batch = "C2", int(os.environ.get("N_EGS", 256)), 64
cfg = synth.CONFIGS[name]
T, P = cfg["T"], cfg["P"]
fst = synth.config_den_fst(name)
graph = io.DenominatorGraph(fst, P)
pi = graph.initial_probs()
rng = np.random.default_rng(0)


def example(seed):
    sup = synth.random_supervision(fst, 1, T, 3, seed=seed, initial_probs=pi)
    n_in = 3 * T + 8
    feats = rng.standard_normal((n_in, 40)).astype(np.float32)
    in_idx = np.array([(0, t, 0) for t in range(-4, 3 * T + 4)], np.int32)
    out_idx = np.array([(0, 3 * t, 0) for t in range(T)], np.int32)
    return dict(inputs=[dict(name="input", indexes=in_idx, features=feats)],
                outputs=[dict(name="output", indexes=out_idx, supervision=sup, deriv_weights=np.ones(T, np.float32))])


tmp = tempfile.mkdtemp()
ark, scp = os.path.join(tmp, "egs.ark"), os.path.join(tmp, "egs.scp")
t0 = time.perf_counter()
kw.write_ark(ark, [("utt%04d" % i, example(100 + i)) for i in range(n_egs)], scp_path=scp)
io.print_key_length("scp:" + scp, scp + ".len")
print("wrote %d examples (%.1f MB) in %.1f s" % (n_egs, os.path.getsize(ark) / 1e6, time.perf_counter() - t0))

dev = torch.device("cuda", 0)
x = torch.randn(batch * T, P, device=dev, requires_grad=True)


def step(sup):
    loss, res = chain_loss(x, graph, sup, l2_regularize=cfg.get("l2", 0.0), leaky_hmm_coefficient=cfg["leaky"])
    # (the gradient as the model's backward would receive it: loss.backward() into a leaf adds a 629 MB copy or
    # accumulate pass per step that is not part of the path)
    torch.autograd.grad(loss, x)
    return res


def epochs(reader, n):
    """-> ms per step, steps, and of that: host ms waiting for the batch (reader.next) / enqueueing the step."""
    steps, wait, enq, val = 0, 0.0, 0.0, 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        while True:
            a = time.perf_counter()
            if not reader.next():
                break
            a2 = time.perf_counter()
            _inputs, sup = reader.value()
            b = time.perf_counter()
            val += b - a2
            step(sup)
            c = time.perf_counter()
            wait += b - a
            enq += c - b
            steps += 1
        reader.reset()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, steps, (wait - val) / steps * 1e3, enq / steps * 1e3, val / steps * 1e3


fixed = io.Supervision.from_synth(synth.random_supervision(fst, batch, T, 3, seed=7, initial_probs=pi))
for _ in range(10):
    step(fixed)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40):
    step(fixed)
torch.cuda.synchronize()
print("same supervision every step:        %.3f ms/step" % ((time.perf_counter() - t0) / 40 * 1e3))
for label, prefetch in (("RandExample, 8 look-ahead threads: ", 8), ("RandExample, 4 look-ahead threads: ", 4),
                        ("RandExample, 3 (the default):      ", True), ("RandExample, 2 look-ahead threads: ", 2),
                        ("RandExample, 1 look-ahead thread:  ", 1), ("RandExample, synchronous:          ", False)):
    rd = io.RandExample(scp, seed=1, batchsize=batch, prefetch=prefetch)
    epochs(rd, 1)
    ms, steps, wait, enq, val = epochs(rd, 5)
    print("%s %.3f ms/step over %d steps  (host: %.3f ms in next(), %.3f ms in value(), %.3f ms enqueueing the step)" % (
        label, ms, steps, wait, val, enq))
rd = io.RandExample(scp, seed=1, batchsize=batch, prefetch=False)
t0 = time.perf_counter()
n = 0
while rd.next():
    n += 1
print("reading + merging alone (no loss):  %.3f ms/batch" % ((time.perf_counter() - t0) / n * 1e3))
