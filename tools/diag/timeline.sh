cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/p_tl
rocprofv3 --kernel-trace -d /tmp/p_tl -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python $R/tools/diag/timeline.py $(find /tmp/p_tl -name "*_results.db" | head -1) > $R/gpurun_out/timeline_r06a.txt 2>&1
head -60 $R/gpurun_out/timeline_r06a.txt
