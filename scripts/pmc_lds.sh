#!/bin/bash
# Development aid (GPU box): LDS counters of one workload's denominator launches, for one setting of TC_DEBUG.
#   TC_DEBUG=no_tune,old_arrange scripts/pmc_lds.sh NAME C3
name=$1; shift
root=$(pwd)
out=$root/gpurun_out/pmc_$name
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
i=0
for group in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  i=$((i + 1))
  rocprofv3 --output-format csv --pmc $group -d "$out/g$i" -o pmc -- python3 $root/scripts/den_few.py "$@" > "$out/g$i.log" 2>&1
done
cd "$root"
python3 scripts/pmc_summary.py "$out" | tee "$out/summary.txt"
