"""Times the fused layout kernels (tc_to2d / tc_from2d) against the torch ops the reference uses either side
of the path (functions.py:118-125 permute+contiguous; functions.py:112 negate + autograd's inverse permute)
at the C3 activation shape (development aid)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from torchain_amd.functions import from2d_hip, to2d, to2d_hip  # noqa: E402

B, C, T = 256, 4096, 150
x = torch.randn(B, C, T, device="cuda")
g2d = torch.randn(T * B, C, device="cuda")
nbytes = 2 * x.numel() * 4


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


rows = [
    ("to2d: torch permute(2,0,1).contiguous()", lambda: to2d(x)),
    ("to2d: tc_to2d", lambda: to2d_hip(x)),
    ("back: torch (-g).view(T,B,C).permute(1,2,0).contiguous()", lambda: (-g2d).view(T, B, C).permute(1, 2, 0).contiguous()),
    ("back: tc_from2d(scale=-1)", lambda: from2d_hip(g2d, (B, C, T), -1.0)),
]
for name, fn in rows:
    ms = timeit(fn)
    print("%-62s %.3f ms  %.2f TB/s of its 2 x 629 MB" % (name, ms, nbytes / ms / 1e9))
