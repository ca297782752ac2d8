cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
PFPP_SA_TRAIN_WIDE=1 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_wide -o w -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/prof_wide.log 2>&1
f=$(find gpurun_out/prof_wide -name "*kernel_stats.csv" | head -1)
head -30 "$f" | cut -c1-200
