"""One-off extension of tests/test_gpu_fuzz.py: the same case generator with other seeds (development aid; the log
of a run is kept under profiles/).   python tests/tools/fuzz_more.py <first seed> <number of seeds> [big]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from oracle import pyoracle  # noqa: E402  (test infrastructure: this script is a test)
import test_gpu_fuzz as F  # noqa: E402

pyoracle.build()
pyoracle.lib()
first, count = int(sys.argv[1]), int(sys.argv[2])
big = len(sys.argv) > 3
worst = {"objf": 0.0, "deriv": 0.0, "xent": 0.0}
cases = fails = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    for _ in range(8 if big else 24):
        cases += 1
        try:
            desc = F._case(pyoracle, rng, big)
            for k in worst:
                worst[k] = max(worst[k], float(desc.split(k + " ")[1].split()[0]))
        except AssertionError as e:
            fails += 1
            print("FAIL seed %d: %s" % (seed, e), flush=True)
print("%d cases, %d failures, worst relative errors: %s" % (cases, fails, worst))
