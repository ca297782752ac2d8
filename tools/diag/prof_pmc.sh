# HBM traffic counters of the training bench (separate passes, --kernel-trace only next to --pmc)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r01h}
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_$C -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --serial > /dev/null
  python $R/tools/rocprof_summary.py $(find /tmp/pmc_$C -name "*_results.db" | head -1) $R/gpurun_out/${TAG}_bench_train_pmc_$C.csv --pmc
done
