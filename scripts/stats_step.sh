#!/bin/bash
root=$(pwd); out=$root/gpurun_out/stats_step; mkdir -p $out; export TMPDIR=/tmp; cd /tmp
XENT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o st -- python3 $root/scripts/time_chain_loss.py C2 3d > $out/run.log 2>&1
cd $root
python3 - $(find $out -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-70s calls %6s avg %9.1f us total %9.2f ms  %5.1f %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
PY
