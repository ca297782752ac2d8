"""where does the training iteration end?  events at: end of the main chain's backward, end of the weight-gradient
stream, end of AdamW.     python tools/diag/tail_events.py"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "puzzlefusion-plusplus_amd"))
import torch
import bench

dev = torch.device("cuda:0")
wl = bench.TrainWorkload(32, 1024, None, 0, dev)
eng = wl.engine
rec = []
orig_all_done = eng._all_done
orig_opt = eng.optimizer_step
orig_fwd = eng.forward


def all_done():

    em, es = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    em.record(torch.cuda.current_stream())
    if eng._side is not None:
        es.record(eng._side)
    else:
        es.record(torch.cuda.current_stream())
    rec[-1].update(main_bwd_end=em, side_end=es)
    orig_all_done()


def fwd(*a, **k):
    e = torch.cuda.Event(enable_timing=True); e.record(torch.cuda.current_stream())
    rec.append(dict(start=e))
    out = orig_fwd(*a, **k)
    e2 = torch.cuda.Event(enable_timing=True); e2.record(torch.cuda.current_stream())
    rec[-1]["fwd_end"] = e2
    return out


def opt(**k):
    e0 = torch.cuda.Event(enable_timing=True); e0.record(torch.cuda.current_stream())
    orig_opt(**k)
    e = torch.cuda.Event(enable_timing=True); e.record(torch.cuda.current_stream())
    rec[-1].update(opt_start=e0, opt_end=e)


eng._all_done, eng.optimizer_step, eng.forward = all_done, opt, fwd
for _ in range(25):
    wl.step()
torch.cuda.synchronize()
rows = rec[5:]
def avg(a, b): return sum(r[a].elapsed_time(r[b]) for r in rows) / len(rows)
print(f"forward               {avg('start', 'fwd_end'):7.3f} ms")
print(f"fwd_end -> bwd end    {avg('fwd_end', 'main_bwd_end'):7.3f} ms")
print(f"main bwd end -> side  {avg('main_bwd_end', 'side_end'):7.3f} ms   (> 0: the main stream waits for the weight gradients)")
print(f"main bwd end -> opt0  {avg('main_bwd_end', 'opt_start'):7.3f} ms")
print(f"AdamW                 {avg('opt_start', 'opt_end'):7.3f} ms")
nxt = sum(rows[i]['opt_end'].elapsed_time(rows[i + 1]['start']) for i in range(len(rows) - 1)) / (len(rows) - 1)
print(f"opt end -> next fwd   {nxt:7.3f} ms")
print(f"iteration             {sum(rows[i]['start'].elapsed_time(rows[i + 1]['start']) for i in range(len(rows) - 1)) / (len(rows) - 1):7.3f} ms")
