"""One-off extension of tests/test_gpu_planes.py's seeded sweeps (development aid; the log of a run is kept under profiles/):
   python tests/tools/fuzz_split.py <first seed> <number of seeds> [planes]     (default: the split-source class, 8-10 planes)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from oracle import pyoracle  # noqa: E402  (test infrastructure: this script is a test)
import test_gpu_planes as TP  # noqa: E402

pyoracle.build()
pyoracle.lib()
first, count = int(sys.argv[1]), int(sys.argv[2])
split = not (len(sys.argv) > 3 and sys.argv[3] == "planes")
cases = fails = 0
worst = {"objf": 0.0, "deriv": 0.0, "xent": 0.0}
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    for _ in range(4):
        cases += 1
        try:
            desc = TP.planes_fuzz_case(pyoracle, rng, split)
            print(desc, flush=True)
            for k in worst:
                worst[k] = max(worst[k], float(desc.split(k + " ")[1].split()[0]))
        except AssertionError as e:
            fails += 1
            print("FAIL seed %d: %s" % (seed, e), flush=True)
print("%d cases, %d failures, worst relative errors: %s" % (cases, fails, worst))
