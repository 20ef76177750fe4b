cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_x2 -o x2 --output-format csv -- python3 scripts/time_den.py X2 > gpurun_out/prof_x2.log 2>&1
TC_DEBUG=force_streamed rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c3s -o c3s --output-format csv -- python3 scripts/time_den.py C3 > gpurun_out/prof_c3s.log 2>&1
find gpurun_out/prof_x2 gpurun_out/prof_c3s -name "*kernel_stats*" | head
for f in $(find gpurun_out/prof_x2 gpurun_out/prof_c3s -name "*kernel_stats.csv"); do echo == $f; head -14 $f | cut -c1-200; done
