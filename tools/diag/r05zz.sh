#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05zz; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py -x -q -m gpu -k "gemm_wd_bit_identical or blocks_sequenced or benchmarked_size or eval" > $O/pytest_sel.txt 2>&1; tail -n 3 $O/pytest_sel.txt
for i in 1 2 3 4 5 6 7 8; do timeout 300 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "blocks_sequenced" 2>&1 | tail -n 1; done > $O/blocks_loop.txt; cat $O/blocks_loop.txt
B="--no-cpu-baseline --no-roofline"
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; }
{
for rep in 1 2 3; do
  echo "wd pf on  train: $(python bench.py --steps 20 --warmup 5 $B 2>/dev/null | line)"
  echo "wd pf off train: $(PFPP_WD_PF=0 python bench.py --steps 20 --warmup 5 $B 2>/dev/null | line)"
done
for rep in 1 2; do
  echo "wd pf on  sampler compact: $(python bench.py --mode sample --compact --steps 30 --warmup 5 $B 2>/dev/null | line)"
  echo "wd pf off sampler compact: $(PFPP_WD_PF=0 python bench.py --mode sample --compact --steps 30 --warmup 5 $B 2>/dev/null | line)"
done
} > $O/ab_wd_pf.txt 2>&1
cat $O/ab_wd_pf.txt
