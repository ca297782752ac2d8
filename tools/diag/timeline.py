"""per-stream timeline of the overlapped training step from a rocprofv3 --kernel-trace database
    python tools/diag/timeline.py <results.db>"""
import sqlite3, sys, collections

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select k.queue_id, k.stream_id, k.start, k.end, s.kernel_name from rocpd_kernel_dispatch k "
                  "join rocpd_info_kernel_symbol s on k.kernel_id = s.id order by k.start").fetchall()
print(len(rows), "dispatches")
# step boundaries: the AdamW kernel ends an iteration
adam = [r for r in rows if "adamw" in r[4]]
print("adamw launches", len(adam))
if len(adam) < 8:
    sys.exit()
# analyse iterations between the 5th and 9th AdamW end
for it in range(5, 9):
    t0, t1 = adam[it][3], adam[it + 1][3]
    win = [r for r in rows if r[2] >= t0 and r[3] <= t1 + 1]
    by = collections.defaultdict(list)
    for q, s, a, b, n in win:
        by[(q, s)].append((a, b, n))
    print(f"\\n== iteration {it}: {(t1 - t0) / 1e6:.3f} ms, {len(win)} kernels")
    allint = sorted((a, b) for q, s, a, b, n in win)
    # union busy
    busy, cur_a, cur_b = 0, None, None
    for a, b in allint:
        if cur_b is None or a > cur_b:
            if cur_b is not None: busy += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    busy += cur_b - cur_a
    print(f"   any-stream busy {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms")
    for key, ks in sorted(by.items(), key=lambda kv: -len(kv[1])):
        ks.sort()
        tot = sum(b - a for a, b, n in ks)
        gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
        big = sorted(((gp, ks[i][2][:40], ks[i + 1][2][:40]) for i, gp in enumerate(gaps) if gp > 20000), reverse=True)[:4]
        print(f"   stream {key}: {len(ks)} kernels, busy {tot / 1e6:.3f} ms, first start +{(ks[0][0] - t0) / 1e6:.3f}, last end +{(ks[-1][1] - t0) / 1e6:.3f} ms, "
              f"gaps: sum {sum(g for g in gaps if g > 0) / 1e6:.3f} ms, >20us: {len([g for g in gaps if g > 20000])}")
        for gp, a, b in big:
            print(f"        gap {gp / 1e3:.0f} us between {a} -> {b}")
