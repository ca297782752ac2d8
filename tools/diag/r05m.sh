#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05m; mkdir -p $O
for f in 1 0; do for i in 1 2 3; do
  PFPP_TRAIN_ADA_BWD_FUSED=$f PFPP_TRAIN_EMBED_BWD_FUSED=$f PFPP_TRAIN_EMBED_FWD_FUSED=$f timeout 300 python -m pytest tests/test_gpu_train.py -x -q -s -m gpu -k "blocks_sequenced and True-False-1" 2>&1 | grep -E "differ by|passed|failed" | tr '\n' ' '; echo " fused=$f"
done; done > $O/moved_fraction.txt 2>&1
cat $O/moved_fraction.txt
timeout 600 python tools/diag/enc_determinism.py --iters 300 > $O/enc_det_alone.txt 2>&1; tail -n 3 $O/enc_det_alone.txt
timeout 600 python tools/diag/enc_determinism.py --iters 300 --load > $O/enc_det_load.txt 2>&1; tail -n 6 $O/enc_det_load.txt
timeout 600 python tools/diag/enc_determinism.py --iters 300 --second-stream > $O/enc_det_2s.txt 2>&1; tail -n 6 $O/enc_det_2s.txt
timeout 600 python tools/diag/enc_determinism.py --iters 300 --second-stream --load > $O/enc_det_2s_load.txt 2>&1; tail -n 6 $O/enc_det_2s_load.txt
