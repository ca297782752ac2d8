cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in 0 1; do echo -n "evaltab=$v: "; PFPP_SA_EVAL_UTAB=$v python -c "
import torch, bench
dev = torch.device('cuda:0')
a = bench.aggl_puzzles_per_s(dev)
b = bench.aggl_puzzles_per_s(dev, n_puzzles=64, in_flight=32)
print(a['value'], b['value'])
" 2>/dev/null | tail -1; done; done
