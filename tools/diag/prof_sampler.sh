# rocprofv3 kernel traces of the sampler step (all slots / compact); CSV summaries into gpurun_out/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r01h}
rm -rf /tmp/prof_c /tmp/prof_d
rocprofv3 --kernel-trace --stats -d /tmp/prof_c -- python $R/bench.py --mode sample --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $R/gpurun_out/${TAG}_sampler_full.json
rocprofv3 --kernel-trace --stats -d /tmp/prof_d -- python $R/bench.py --mode sample --compact --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $R/gpurun_out/${TAG}_sampler_compact.json
python $R/tools/rocprof_summary.py $(find /tmp/prof_c -name "*_results.db" | head -1) $R/gpurun_out/${TAG}_bench_sampler_full_kernel_stats.csv
python $R/tools/rocprof_summary.py $(find /tmp/prof_d -name "*_results.db" | head -1) $R/gpurun_out/${TAG}_bench_sampler_compact_kernel_stats.csv
