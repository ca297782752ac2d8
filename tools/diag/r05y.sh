#!/bin/bash
# end-of-round evidence: kernel stats (train overlap / serial, sampler, stress), HBM counters, the default bench line, module-surface ratio
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O/r05y
bash tools/diag/prof_round.sh r05y train sampler stress pmc > $O/r05y/prof.log 2>&1
bash tools/diag/prof_b1.sh > /dev/null 2>&1; mv $O/b1_prof.csv $O/r05y_b1_loop_kernel_stats.csv
cd $GRAFT_REPO_ROOT
python bench.py > $O/r05y_bench_default_line.json 2> $O/r05y_bench_default.err
tail -c 600 $O/r05y_bench_default_line.json
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_train.py -q -s -m gpu -k "module_surface_runs_the_benchmarked" 2>&1 | grep -E "engine-level|passed|failed"; done > $O/r05y/module_ratio.txt 2>&1
cat $O/r05y/module_ratio.txt
