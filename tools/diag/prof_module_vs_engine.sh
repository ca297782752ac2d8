cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pm /tmp/pe
rocprofv3 --kernel-trace --stats -d /tmp/pm -- python $R/tools/diag/module_loop_only.py > $R/gpurun_out/modloop.log 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/pm -name "*_results.db" | head -1) $R/gpurun_out/r03o_module_loop_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d /tmp/pe -- python $R/bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-roofline > $R/gpurun_out/engloop.log 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/pe -name "*_results.db" | head -1) $R/gpurun_out/r03o_engine_loop_kernel_stats.csv
tail -1 $R/gpurun_out/modloop.log
