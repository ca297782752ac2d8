# the one-puzzle-in-flight loop (bench.aggl_puzzles_per_s) with one environment switch alternating: bash tools/diag/b1_ab.sh NAME v0 v1
cd $GRAFT_REPO_ROOT
N=$1; shift
for r in 1 2; do for v in "$@"; do
  env $N=$v python -c "
import sys; sys.path[:0]=['.','puzzlefusion-plusplus_amd']
import torch, bench
r=bench.aggl_puzzles_per_s(torch.device('cuda:0'), n_puzzles=3)
print('$N=$v', r['value'], r.get('ddpm_steps'))" 2>/dev/null | tail -1
done; done
