# rocprofv3 kernel traces of the training bench: overlapped (default) and serialised; CSV summaries into gpurun_out/
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r01h}
rm -rf /tmp/prof_a /tmp/prof_b
rocprofv3 --kernel-trace --stats -d /tmp/prof_a -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $R/gpurun_out/${TAG}_overlap.json
rocprofv3 --kernel-trace --stats -d /tmp/prof_b -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --serial > $R/gpurun_out/${TAG}_serial.json
python $R/tools/rocprof_summary.py $(find /tmp/prof_a -name "*_results.db" | head -1) $R/gpurun_out/${TAG}_bench_train_kernel_stats.csv
python $R/tools/rocprof_summary.py $(find /tmp/prof_b -name "*_results.db" | head -1) $R/gpurun_out/${TAG}_bench_train_serial_kernel_stats.csv
