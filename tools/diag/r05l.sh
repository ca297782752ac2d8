#!/bin/bash
# round 5, session 2, call 1: new skinny-end kernels (tests), W-direct probe with prefetched fragments, surface race diag
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train.py -x -q -m gpu -k "ada_linear or token_embedding_backward or fused_embedding_and_adaln or loss_and_grads_vs_reference or two_optimizer_steps or blocks_sequenced" > $O/pytest_new.txt 2>&1
tail -n 5 $O/pytest_new.txt
: > $O/wdirect_pf.txt
for shp in "3850 512 512" "3850 1536 512" "3850 512 1536" "3850 512 2048" "3850 2048 512" "16000 512 512" "16000 1536 512"; do
  echo "== $shp" >> $O/wdirect_pf.txt
  for b in tools/lab/_bin/wd_*; do timeout 60 $b $shp >> $O/wdirect_pf.txt 2>&1; done
done
grep -c "us per launch" $O/wdirect_pf.txt
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; done > $O/bench3.txt 2>&1
cat $O/bench3.txt
timeout 1500 python tools/diag/surface_race.py --trials 30 > $O/surface_race_a.txt 2>&1
tail -n 12 $O/surface_race_a.txt
