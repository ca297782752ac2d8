#!/bin/bash
# packed fp32 off (default build) vs on (tools/lab/_bin/lib_pk_on.so): same-box A/B of the bench lines; new tests; fps race with the default
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05u; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train.py tests/test_gpu_parity.py -x -q -m gpu -k "ada_linear or token_embedding_backward or fused_embedding_and_adaln or loss_and_grads_vs_reference or two_optimizer_steps or blocks_sequenced or sampling_chain or fps or encoder" > $O/pytest_sel.txt 2>&1
tail -n 6 $O/pytest_sel.txt
B="--no-cpu-baseline --no-roofline"
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; }
{
for rep in 1 2 3; do
  echo "pk off train: $(python bench.py --steps 20 --warmup 5 $B 2>/dev/null | line)"
  echo "pk on  train: $(PFPP_LIB_PATH=tools/lab/_bin/lib_pk_on.so python bench.py --steps 20 --warmup 5 $B 2>/dev/null | line)"
done
for rep in 1 2; do
  echo "pk off sampler compact: $(python bench.py --mode sample --compact --steps 30 --warmup 5 $B 2>/dev/null | line)"
  echo "pk on  sampler compact: $(PFPP_LIB_PATH=tools/lab/_bin/lib_pk_on.so python bench.py --mode sample --compact --steps 30 --warmup 5 $B 2>/dev/null | line)"
done
echo "fused ends off (pk off): $(PFPP_TRAIN_ADA_BWD_FUSED=0 PFPP_TRAIN_EMBED_BWD_FUSED=0 PFPP_TRAIN_EMBED_FWD_FUSED=0 python bench.py --steps 20 --warmup 5 $B 2>/dev/null | line)"
echo "fused ends on  (pk off): $(python bench.py --steps 20 --warmup 5 $B 2>/dev/null | line)"
echo "fused ends off (pk off): $(PFPP_TRAIN_ADA_BWD_FUSED=0 PFPP_TRAIN_EMBED_BWD_FUSED=0 PFPP_TRAIN_EMBED_FWD_FUSED=0 python bench.py --steps 20 --warmup 5 $B 2>/dev/null | line)"
echo "fused ends on  (pk off): $(python bench.py --steps 20 --warmup 5 $B 2>/dev/null | line)"
} > $O/ab_pk_and_ends.txt 2>&1
cat $O/ab_pk_and_ends.txt
for rep in 1 2; do timeout 600 python tools/diag/fps_race.py --iters 8000 --other gemm --N 1024 --S 256 --F 16 2>&1 | grep -v amdgpu.ids | tail -n 1; timeout 600 python tools/diag/fps_race.py --iters 8000 --other gemm 2>&1 | grep -v amdgpu.ids | tail -n 1; done > $O/fps_race_default_nopk.txt 2>&1
cat $O/fps_race_default_nopk.txt
