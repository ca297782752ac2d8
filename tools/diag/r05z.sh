#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05z; mkdir -p $O
B="--no-cpu-baseline --no-roofline --latents-given"
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; }
{
for rep in 1 2; do
echo "default: $(python bench.py --steps 20 --warmup 5 $B 2>/dev/null | line)"
echo "ends unfused: $(PFPP_TRAIN_ADA_BWD_FUSED=0 PFPP_TRAIN_EMBED_BWD_FUSED=0 PFPP_TRAIN_EMBED_FWD_FUSED=0 python bench.py --steps 20 --warmup 5 $B 2>/dev/null | line)"
echo "pk on: $(PFPP_LIB_PATH=tools/lab/_bin/lib_pk_on.so python bench.py --steps 20 --warmup 5 $B 2>/dev/null | line)"
echo "embed fwd unfused only: $(PFPP_TRAIN_EMBED_FWD_FUSED=0 python bench.py --steps 20 --warmup 5 $B 2>/dev/null | line)"
echo "embed bwd unfused only: $(PFPP_TRAIN_EMBED_BWD_FUSED=0 python bench.py --steps 20 --warmup 5 $B 2>/dev/null | line)"
echo "ada unfused only: $(PFPP_TRAIN_ADA_BWD_FUSED=0 python bench.py --steps 20 --warmup 5 $B 2>/dev/null | line)"
done
} > $O/ab_latents_given.txt 2>&1
cat $O/ab_latents_given.txt
