# L2 hit / miss counters of the training bench with the tile-major and the chunk-major split-K assignment (own --pmc passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r01h}
for K in 0 1; do
  rm -rf /tmp/l2_$K
  PFPP_GRAD_KXCD=$K rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d /tmp/l2_$K -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --serial > /dev/null
  python $R/tools/rocprof_summary.py $(find /tmp/l2_$K -name "*_results.db" | head -1) $R/gpurun_out/${TAG}_bench_train_pmc_L2_kxcd$K.csv --pmc
done
