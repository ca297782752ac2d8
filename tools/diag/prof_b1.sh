cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pb1
cat > /tmp/b1.py <<PY
import sys
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/puzzlefusion-plusplus_amd")
import torch, bench
print(bench.aggl_puzzles_per_s(torch.device("cuda:0"), n_puzzles=4))
PY
rocprofv3 --kernel-trace --stats -d /tmp/pb1 -- python /tmp/b1.py > /dev/null 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/pb1 -name "*_results.db" | head -1) $R/gpurun_out/b1_prof.csv
