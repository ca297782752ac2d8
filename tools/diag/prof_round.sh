# one round's rocprofv3 evidence: kernel-trace stats of the training bench (overlapped + serial), the sampler (all slots / compact),
# the stress mode, and the HBM counter passes (separate --pmc runs next to --kernel-trace only).  CSV summaries -> gpurun_out/
#   usage: bash tools/diag/prof_round.sh r02a [train|sampler|stress|pmc ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
shift
WHAT=${@:-train sampler stress pmc}
B="--no-cpu-baseline --no-roofline"
stats() { python $R/tools/rocprof_summary.py $(find $1 -name "*_results.db" | head -1) $R/gpurun_out/$2; }
for w in $WHAT; do
  case $w in
  train)
    rm -rf /tmp/p_a /tmp/p_b
    rocprofv3 --kernel-trace --stats -d /tmp/p_a -- python $R/bench.py --steps 20 --warmup 3 $B > $R/gpurun_out/${TAG}_train_overlap.json
    rocprofv3 --kernel-trace --stats -d /tmp/p_b -- python $R/bench.py --steps 20 --warmup 3 $B --serial > $R/gpurun_out/${TAG}_train_serial.json
    stats /tmp/p_a ${TAG}_bench_train_kernel_stats.csv; stats /tmp/p_b ${TAG}_bench_train_serial_kernel_stats.csv ;;
  sampler)
    rm -rf /tmp/p_c /tmp/p_d
    rocprofv3 --kernel-trace --stats -d /tmp/p_c -- python $R/bench.py --mode sample --steps 20 --warmup 3 $B > $R/gpurun_out/${TAG}_sampler_full.json
    rocprofv3 --kernel-trace --stats -d /tmp/p_d -- python $R/bench.py --mode sample --compact --steps 20 --warmup 3 $B > $R/gpurun_out/${TAG}_sampler_compact.json
    stats /tmp/p_c ${TAG}_bench_sampler_full_kernel_stats.csv; stats /tmp/p_d ${TAG}_bench_sampler_compact_kernel_stats.csv ;;
  stress)
    rm -rf /tmp/p_e
    rocprofv3 --kernel-trace --stats -d /tmp/p_e -- python $R/bench.py --mode stress --steps 6 --warmup 2 $B > $R/gpurun_out/${TAG}_stress.json
    stats /tmp/p_e ${TAG}_bench_stress_kernel_stats.csv ;;
  pmc)
    for C in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/pmc_$C
      rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_$C -- python $R/bench.py --steps 3 --warmup 1 $B --serial > /dev/null
      python $R/tools/rocprof_summary.py $(find /tmp/pmc_$C -name "*_results.db" | head -1) $R/gpurun_out/${TAG}_bench_train_pmc_$C.csv --pmc
    done ;;
  esac
done
