#!/bin/bash
# Development aid (GPU box): counter passes (one --pmc group per run, no tracing) over scripts/den_few.py.
#   scripts/pmc_den.sh NAME X2   -> gpurun_out/pmc_NAME/<group>/..._counter_collection.csv and a per-kernel summary
name=$1; shift
root=$(pwd)
out=$root/gpurun_out/pmc_$name
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
i=0
for group in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  i=$((i + 1))
  rocprofv3 --output-format csv --pmc $group -d "$out/g$i" -o pmc -- python3 $root/scripts/den_few.py "$@" > "$out/g$i.log" 2>&1
done
cd "$root"
python3 scripts/pmc_summary.py "$out"
