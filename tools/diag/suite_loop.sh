#!/bin/bash
# N consecutive runs of the whole -m gpu suite on one box (no retry wrapper, every run must be green)
cd $GRAFT_REPO_ROOT
N=${1:-25}; TAG=${2:-A}
mkdir -p gpurun_out/suite
OUT=gpurun_out/suite/suite_loop_$TAG.txt
: > $OUT
for i in $(seq 1 $N); do
  t0=$(date +%s)
  timeout 900 python -m pytest tests -x -q -m gpu > /tmp/suite_run.txt 2>&1
  rc=$?
  echo "run $i rc=$rc $(( $(date +%s) - t0 ))s: $(tail -n 1 /tmp/suite_run.txt)" >> $OUT
  if [ $rc -ne 0 ]; then tail -n 60 /tmp/suite_run.txt >> $OUT; fi
done
grep -c "rc=0" $OUT
