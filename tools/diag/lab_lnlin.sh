cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in same rotate rotate_evict; do
rm -rf /tmp/pb1
rocprofv3 --kernel-trace --stats -d /tmp/pb1 -- python $R/tools/diag/lnlin_cases.py $c > /dev/null 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/pb1 -name "*_results.db" | head -1) /tmp/b1_$c.csv > /dev/null
echo "$c: $(grep lnlin_small /tmp/b1_$c.csv | cut -d, -f1,2,4 | tr '\n' ' ')"
done
