cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pb1
cat > /tmp/b1.py <<PY
import sys
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/puzzlefusion-plusplus_amd")
import torch, bench
print(bench.aggl_puzzles_per_s(torch.device("cuda:0"), n_puzzles=2))
PY
rocprofv3 --kernel-trace --stats -d /tmp/pb1 -- python /tmp/b1.py > /dev/null 2>&1
python - <<PY
import sqlite3, glob
db = glob.glob("/tmp/pb1/**/*_results.db", recursive=True)[0]
c = sqlite3.connect(db)
cur = c.execute("select * from kernels limit 1")
cols = [d[0] for d in cur.description]
print(cols)
q = "select name, grid_x, grid_y, count(*), avg(end - start) / 1000.0 from kernels where name like '%gemm_pl_kernel<1, 1, 2, 1%' or name like '%lnlin%' or name like '%gemm_small%' or name like '%attn_dense%' group by name, grid_x, grid_y order by name, grid_x"
try:
    for r in c.execute(q): print(r[0][:60], r[1:])
except Exception as e:
    print("ERR", e)
PY
