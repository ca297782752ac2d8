#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05w; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train.py -x -q -m gpu -k "ada_linear or token_embedding_backward or fused_embedding_and_adaln" > $O/pytest_sel.txt 2>&1; tail -n 3 $O/pytest_sel.txt
: > $O/wdirect_il2.txt
for shp in "3850 512 512" "3850 1536 512" "3850 512 1536" "3850 512 2048" "3850 2048 512" "16000 512 512"; do
  echo "== $shp" >> $O/wdirect_il2.txt
  for b in tools/lab/_bin/wd_*; do timeout 60 $b $shp >> $O/wdirect_il2.txt 2>&1; done
done
grep -v "^GPU core\|^Memory access" $O/wdirect_il2.txt | sed 's/(fp32-grade), //' | cut -c1-150
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_b; rocprofv3 --kernel-trace --stats -d /tmp/p_b -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --serial > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/p_b -name "*_results.db" | head -1) $GRAFT_REPO_ROOT/$O/train_serial_kernel_stats.csv
grep -i "ada_\|embed_" $GRAFT_REPO_ROOT/$O/train_serial_kernel_stats.csv
