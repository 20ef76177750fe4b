#!/bin/bash
# Development aid (runs on the GPU box): rocprofv3 kernel-trace statistics of one script.
#   scripts/prof_kernels.sh NAME scripts/time_den.py X2        -> gpurun_out/prof_NAME/k_kernel_stats.csv
#   TC_DEBUG=force_streamed scripts/prof_kernels.sh c3s scripts/time_den.py C3
name=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$name -o k --output-format csv -- python3 "$@" > gpurun_out/prof_$name.log 2>&1
head -14 gpurun_out/prof_$name/k_kernel_stats.csv | cut -c1-180
