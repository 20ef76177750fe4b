cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in C2 C5; do
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$c -o k --output-format csv -- python3 scripts/time_den.py $c > gpurun_out/prof_$c.log 2>&1
head -6 gpurun_out/prof_$c/k_kernel_stats.csv | cut -c1-200
done
