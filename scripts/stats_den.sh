#!/bin/bash
# Development aid (GPU box): rocprofv3 --kernel-trace --stats over scripts/den_few.py -> per-kernel calls / average
#   scripts/stats_den.sh NAME X2 [S] [calls]   (TC_DEBUG honoured)
name=$1; shift
root=$(pwd)
out=$root/gpurun_out/stats_$name
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o st -- python3 $root/scripts/den_few.py "$@" > "$out/run.log" 2>&1
cd "$root"
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-70s calls %6s avg %9.1f us total %9.2f ms  %5.1f %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
PY
