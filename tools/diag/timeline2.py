"""per-stream timeline of the overlapped training iteration from a rocprofv3 --kernel-trace database (iteration = from one mse_loss_kernel
to the next): per stream busy / gap time, first start / last end, and for the chain stream the kernels with the largest slow-down against
their serial-profile average (python tools/diag/timeline2.py <overlap results.db> [<serial results.db>])"""
import sqlite3, sys, collections

def load(path):
    db = sqlite3.connect(path)
    return db.execute("select k.queue_id, k.stream_id, k.start, k.end, s.kernel_name from rocpd_kernel_dispatch k "
                      "join rocpd_info_kernel_symbol s on k.kernel_id = s.id order by k.start").fetchall()

rows = load(sys.argv[1])
serial_avg = {}
if len(sys.argv) > 2:
    acc = collections.defaultdict(list)
    for q, s, a, b, n in load(sys.argv[2]):
        acc[n].append(b - a)
    serial_avg = {n: sum(v) / len(v) for n, v in acc.items()}
marks = [r[2] for r in rows if "mse_loss_kernel" in r[4]]
print(len(rows), "dispatches,", len(marks), "iterations")
short = lambda n: n.replace("(anonymous namespace)::", "").replace("pfpp_gemm_detail::pl::", "").replace("void ", "")[:70]
for it in range(6, min(9, len(marks) - 1)):
    t0, t1 = marks[it], marks[it + 1]
    win = [r for r in rows if t0 <= r[2] < t1]
    by = collections.defaultdict(list)
    for q, s, a, b, n in win:
        by[(q, s)].append((a, b, n))
    print(f"\n== iteration {it}: {(t1 - t0) / 1e6:.3f} ms, {len(win)} kernels")
    for key, ks in sorted(by.items(), key=lambda kv: -len(kv[1])):
        ks.sort()
        tot = sum(b - a for a, b, n in ks)
        gaps = [max(0, ks[i + 1][0] - ks[i][1]) for i in range(len(ks) - 1)]
        print(f"  stream {key}: {len(ks)} kernels, busy {tot / 1e6:.3f} ms, first +{(ks[0][0] - t0) / 1e6:.3f}, last end +{(ks[-1][1] - t0) / 1e6:.3f}, "
              f"gaps sum {sum(gaps) / 1e6:.3f} ms (>10us: {sum(1 for g in gaps if g > 10000)}, >50us: {sum(1 for g in gaps if g > 50000)})")
        big = sorted(((gp, short(ks[i][2]), short(ks[i + 1][2])) for i, gp in enumerate(gaps) if gp > 30000), reverse=True)[:5]
        for gp, a, b in big:
            print(f"        gap {gp / 1e3:.0f} us: {a} -> {b}")
        if serial_avg and len(ks) > 100:
            infl = collections.defaultdict(lambda: [0, 0, 0])
            for a, b, n in ks:
                if n in serial_avg:
                    e = infl[n]; e[0] += b - a; e[1] += serial_avg[n]; e[2] += 1
            tot_o = sum(e[0] for e in infl.values()); tot_s = sum(e[1] for e in infl.values())
            print(f"        kernel time here {tot_o / 1e6:.3f} ms vs {tot_s / 1e6:.3f} ms at serial averages (x{tot_o / tot_s:.2f})")
            for n, e in sorted(infl.items(), key=lambda kv: -(kv[1][0] - kv[1][1]))[:8]:
                print(f"          +{(e[0] - e[1]) / 1e3:7.1f} us  {e[2]:3d} x {e[0] / e[2] / 1e3:6.1f} us (serial {e[1] / e[2] / 1e3:6.1f})  {short(n)}")
