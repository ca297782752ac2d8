cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tl
rocprofv3 --kernel-trace -d /tmp/tl -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python $R/tools/diag/timeline.py $(find /tmp/tl -name "*_results.db" | head -1)
