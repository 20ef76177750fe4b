"""Development aid: per-kernel averages of the counter passes scripts/pmc_den.sh wrote (gpurun_out/pmc_NAME)."""
import collections
import csv
import glob
import re
import sys

out = sys.argv[1]
want = sys.argv[2:]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(slab_\w+|big_\w+|den_\w+)", r["Kernel_Name"])
        k = m.group(1) if m else "other"
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
for k in sorted(tot):
    if want and k not in want:
        continue
    print(k)
    for c in sorted(tot[k]):
        print("   %-32s per launch %14.1f  (%d launches)" % (c, tot[k][c] / cnt[k][c], cnt[k][c]))
