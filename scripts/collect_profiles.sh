#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats, HBM traffic counters and SQ counters for the
# default bench.py workload.  Each PMC group is its own pass (never combined with tracing).
# Output: gpurun_out/prof_$1/{kt,fetch,write,sq1,sq2}/...  then scripts/summarize_profiles.py.
#   scripts/collect_profiles.sh TAG [CONFIG [KERNEL]]     (default: C3, den_tied_kernel; R4: den_tied_planes_kernel)
set -u
tag=${1:-r02}
config=${2:-C3}
kernel=${3:-den_tied_kernel}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
# the one-off timing launches of tc_den_graph_prepare (48-frame batches of both kernels) would be averaged into the
# per-launch figures below: switched off for these passes (C3 keeps the fused kernel either way)
export TORCHAIN_HIP_DEBUG=no_tune
cmd="python3 $root/bench.py --config $config --steps 5 --warmup 2 --no-cpu-baseline --no-extras"
cd /tmp
if [ -z "${CLOCK_ONLY:-}" ]; then
# the kernel-trace pass runs the default bench (30 steps + 5 warm-up + 20 event-timed launches), so its
# average is taken over the same mix of launches as bench.py's HIP-event figure
rocprofv3 --output-format csv --kernel-trace --stats -d "$out/kt" -o kt -- python3 $root/bench.py --config $config --no-cpu-baseline --no-extras > "$out/kt.log" 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$out/fetch" -o pmc -- $cmd > "$out/fetch.log" 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$out/write" -o pmc -- $cmd > "$out/write.log" 2>&1
rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM -d "$out/sq1" -o pmc -- $cmd > "$out/sq1.log" 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d "$out/sq2" -o pmc -- $cmd > "$out/sq2.log" 2>&1
rocprofv3 --output-format csv --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES -d "$out/sq3" -o pmc -- $cmd > "$out/sq3.log" 2>&1
fi
cd "$root"
# the shader clock the kernel really runs at (the chip is power-capped: 2.1 - 2.4 GHz by batch): sampled while the same
# workload runs un-profiled, for the secondary ceilings (LDS-active and VALU-issue time need a clock)
(python3 bench.py --config $config --steps 30000 --warmup 2 --no-cpu-baseline --no-extras > "$out/clock_run.log" 2>&1) &
pid=$!
sleep 20   # (imports, the graph's schedules, the upload: the timed loop is running by now and for tens of seconds more)
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power" | head -6 > "$out/clock.txt"
kill $pid 2>/dev/null; wait $pid 2>/dev/null
cat "$out/clock.txt"
python3 scripts/summarize_profiles.py "$out" "$tag" "$kernel" "$config"
