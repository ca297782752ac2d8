"""the torch-native / runtime-copy launches of one overlapped training iteration with their stream and neighbours (from a rocprofv3
--kernel-trace database): what is left of the glue between our kernels.   python tools/diag/glue_list.py <results.db>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select k.queue_id, k.stream_id, k.start, k.end, s.kernel_name from rocpd_kernel_dispatch k "
                  "join rocpd_info_kernel_symbol s on k.kernel_id = s.id order by k.start").fetchall()
marks = [i for i, r in enumerate(rows) if "mse_loss_kernel" in r[4]]
a, b = marks[6], marks[7]
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")[:80]
per_stream_prev = {}
for i in range(a, b):
    q, s, t0, t1, n = rows[i]
    key = (q, s)
    if "rocclr" in n or "at::native" in n or "at6native" in n:
        print(f"stream {key}  +{(t0 - rows[a][2]) / 1e3:8.1f} us  {(t1 - t0) / 1e3:5.1f} us  {short(n)}   <- after {short(per_stream_prev.get(key, '-'))[:50]}")
    per_stream_prev[key] = n
