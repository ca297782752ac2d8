"""Round 5, VERDICT r4 item 1: where does the two-rank gradient discrepancy come from?

Two ranks on ONE GPU over gloo (the configuration of tests/test_gpu_train.py's two-rank tests), in a loop, instrumented so that the
two halves of the path can be told apart:

  local  = the rank's own gradient from a backward under no_sync() (the six-layer C call, no exchange): the reference
  post   = the flat gradient buffer after loss_and_grads + finish_grad_exchange (per-layer calls, per-layer all-reduce)
  want   = local_0 + local_1, exchanged with a QUIET all-reduce (device synchronised, nothing else in flight)

  post != want  ->  something in the multi-rank step is wrong; with PFPP_RACE_SNAP=1 the slices are also cloned right in front of
                    their all-reduce (pre): pre != local = our schedule handed an unfinished slice over, pre == local but
                    post != sum(pre) = the exchange itself (gloo's copy streams) lost something.

usage: python tools/diag/rank_race.py [--trials T] [--iters N] [--armed] [--parent-engine]
(each trial = two fresh rank processes; the calling process optionally builds an engine first and keeps its GPU context, like a
pytest parent)."""
import argparse
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
for p_ in (str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)


class NS:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def make_engine(dev):
    import torch

    from pfpp_hip.train import DenoiserTrainEngine
    from puzzlefusion_plusplus.denoiser.model.modules.denoiser_transformer import DenoiserTransformer

    torch.manual_seed(1234)
    m = DenoiserTransformer(NS(model=NS(embed_dim=512, out_channels=7, num_layers=6, num_heads=8, num_dim=64, num_point=25)))
    return DenoiserTrainEngine(m.to(dev))


def inputs(rank, dev, big):
    import numpy as np
    import torch

    if big:
        from pfpp_hip import synthetic

        torch.manual_seed(77 + rank)
        B, P, L = 8, 20, 25
        nparts = [2 + (3 * i + rank) % 12 for i in range(B)]
        valid = torch.zeros(B, P)
        for b, n in enumerate(nparts):
            valid[b, :n] = 1
        ref = torch.zeros(B, P, dtype=torch.bool)
        ref[:, 0] = True
        x = torch.randn(B, P, 7)
        inp = [x, torch.randint(0, 1000, (B,)), torch.randn(B, P, L, 64) * 0.3, torch.rand(B, P, L, 3) - 0.5, valid, torch.rand(B, P, 1) + 0.5, ref]
        noise = torch.randn(B, P, 7)
        return [v.to(dev) for v in inp], noise.to(dev)
    g = np.load(ROOT / "tests" / "golden" / "denoiser.npz")
    t = np.load(ROOT / "tests" / "golden" / "train.npz")
    keys = ("x", "timesteps", "latent", "xyz", "part_valids", "scale", "ref_part")
    inp = [torch.from_numpy(g[k])[rank:rank + 1].to(dev) for k in keys]
    noise = torch.from_numpy(t["noise"])[rank:rank + 1].to(dev)
    return inp, noise


def where(eng, idx):
    f = eng.flat
    best = None
    for n in f.order:
        if f.offset[n] <= idx:
            best = n
    return best


def pattern(eng, d, post, want, scale_ref, name_hint=None):
    """where the off elements sit inside their parameter: row / column ranges and a few (got, want) pairs"""
    import torch

    f = eng.flat
    out = []
    for n in f.order:
        o = f.offset[n]
        k = f.named[n].numel()
        sel = d[o:o + k] > 2e-5 * scale_ref
        c = int(sel.sum())
        if c == 0:
            continue
        shp = tuple(f.named[n].shape)
        idx = torch.nonzero(sel).flatten()
        if len(shp) == 2:
            r, cc = idx // shp[1], idx % shp[1]
            rows = sorted(set(r.tolist()))
            cols = sorted(set(cc.tolist()))
            desc = f"rows {rows[:12]}{'...' if len(rows) > 12 else ''} (n={len(rows)}) cols {cols[0]}..{cols[-1]} (n={len(cols)})"
        else:
            desc = f"idx {idx[:8].tolist()}"
        ex = [(float(post[o + int(i)]), float(want[o + int(i)])) for i in idx[:4]]
        out.append(f"{n} {shp}: {c} off; {desc}; got/want {[(round(g_, 6), round(w_, 6)) for g_, w_ in ex]}")
    return " || ".join(out[:4])


def loopback_main(a):
    """ONE process, no torch.distributed: the exchange is forced active and replaced by a loop-back (copy to pinned host memory and
    back on a pool stream, like gloo's CUDA path) — the per-layer pfpp_tlayers_bwd(i, i + 1) calls, the comm stream and the AdamW behind
    it run exactly as with N > 1.  Bit pattern of the result against the six-layer call."""
    import torch

    from pfpp_hip import parallel

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    eng = make_engine(dev)
    f = eng.flat
    inp, noise = inputs(0, dev, a.big)
    train = bool(a.train)
    seed = 4242
    mode = a.loopback

    class Handle:
        def __init__(self, ev):
            self.ev = ev

        def wait(self):
            if self.ev is not None:
                torch.cuda.current_stream().wait_event(self.ev)

    ex = eng._exchange
    parallel.GradExchange.active = staticmethod(lambda: True)
    parallel.GradExchange.mean_factor = staticmethod(lambda: 1.0)
    ex.gather_rows = lambda rows, index, dim=1: (rows.contiguous(), index.contiguous())
    pool = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(4)]
    state = {"k": 0}

    def reduce_lb(x, y):
        if y <= x:
            return
        if mode == 1:
            ex._handles.append(Handle(None))
            return
        ev = torch.cuda.Event()
        ev.record()
        st = pool[state["k"] % len(pool)]
        state["k"] += 1
        st.wait_event(ev)
        with torch.cuda.stream(st):
            host = torch.empty(y - x, dtype=torch.float32, pin_memory=True)
            host.copy_(f.grads[x:y], non_blocking=True)
            st.synchronize()
            f.grads[x:y].copy_(host, non_blocking=True)
            ev2 = torch.cuda.Event()
            ev2.record()
        ex._handles.append(Handle(ev2))

    ex._reduce = reduce_lb
    ex.finish = lambda: ([h.wait() for h in ex._handles], ex._handles.clear(), 1.0)[2]

    def close_step():
        eng.optimizer_step(lr=0.0, weight_decay=0.0)
        f.zero_grad()
        torch.cuda.synchronize()

    with eng.no_sync():
        eng.loss_and_grads(*inp, noise, train=train, seed=seed)
    torch.cuda.synchronize()
    want = f.grads.clone()
    close_step()
    scale_ref = float(want.abs().max())
    bad = 0
    t0 = time.time()
    for it in range(a.iters):
        if a.armed:
            eng.arm_optimizer(lr=0.0, weight_decay=0.0, zero_grad=False)
        eng.loss_and_grads(*inp, noise, train=train, seed=seed)
        eng.finish_grad_exchange()
        torch.cuda.synchronize()
        d = (f.grads - want).abs()
        err = float(d.max()) / scale_ref
        if err > 2e-5:
            bad += 1
            if bad <= 12:
                print(f"[loopback {mode}] it {it}: BAD {err:.3e}: {pattern(eng, d, f.grads, want, scale_ref)}", flush=True)
        close_step()
    print(f"LOOPBACK mode={mode} armed={a.armed} big={a.big} iters={a.iters}: {bad} bad iterations, {time.time() - t0:.0f}s, env="
          f"{ {k: v for k, v in os.environ.items() if k.startswith('PFPP_') or k.startswith('AMD_') or k.startswith('GPU_')} }", flush=True)


def load_main(a):
    """background load: another process keeping the GPU busy with its own training iterations"""
    import torch

    dev = torch.device("cuda:0")
    eng = make_engine(dev)
    inp, noise = inputs(1, dev, True)
    t_end = time.time() + a.load
    while time.time() < t_end:
        for _ in range(20):
            eng.loss_and_grads(*inp, noise, train=True, seed=1)
            eng.optimizer_step(lr=0.0, weight_decay=0.0)
            eng.flat.zero_grad()
        torch.cuda.synchronize()


def rank_main(a):
    import torch
    import torch.distributed as dist

    rank, world = a.rank, 2
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(a.port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    eng = make_engine(dev)
    f = eng.flat
    inp, noise = inputs(rank, dev, a.big)
    n_tab = f.offset["transformer_layers.0.norm1.linear.weight"]
    train = bool(a.train)
    seed = 4242

    def close_step():
        eng.optimizer_step(lr=0.0, weight_decay=0.0)
        f.zero_grad()
        torch.cuda.synchronize()

    # ---- local reference: six-layer call, no exchange
    with eng.no_sync():
        eng.loss_and_grads(*inp, noise, train=train, seed=seed)
    torch.cuda.synchronize()
    local = f.grads.clone()
    close_step()
    with eng.no_sync():
        eng.loss_and_grads(*inp, noise, train=train, seed=seed)
    torch.cuda.synchronize()
    rep = float((f.grads - local).abs().max() / local.abs().max())
    close_step()
    # the tables travel as rows in the exchanged step: their dense reference is the same scatter, summed over the ranks below
    want = local.clone()
    dist.all_reduce(want)                       # quiet exchange
    torch.cuda.synchronize()
    scale_ref = float(want.abs().max())

    snap = os.environ.get("PFPP_RACE_SNAP", "0") == "1"
    pre = torch.zeros_like(local) if snap else None
    if snap:
        orig = eng._exchange._reduce

        def reduce_snap(x, y):
            if y > x:
                pre[x:y].copy_(f.grads[x:y])          # on the stream the all-reduce is issued from
            orig(x, y)

        eng._exchange._reduce = reduce_snap

    bad = 0
    for it in range(a.iters):
        if a.armed:
            eng.arm_optimizer(lr=0.0, weight_decay=0.0, zero_grad=False)
        eng.loss_and_grads(*inp, noise, train=train, seed=seed)
        eng.finish_grad_exchange()
        torch.cuda.synchronize()
        post = f.grads.clone()
        d = (post - want).abs()
        err = float(d.max()) / scale_ref
        msg = ""
        if err > 2e-5:
            bad += 1
            idx = int(d.argmax())
            # how many elements are off, and in which parameters
            offn = int((d > 2e-5 * scale_ref).sum())
            msg = f" BAD post-vs-want {err:.3e} at {idx} ({where(eng, idx)}), {offn} elements off :: {pattern(eng, d, post, want, scale_ref)}"
            if snap:
                dl = (pre - local).abs()
                dl[:n_tab] = 0
                e_loc = float(dl.max()) / scale_ref
                spre = pre.clone()
                dist.all_reduce(spre)
                torch.cuda.synchronize()
                dx = (post - spre).abs()
                dx[:n_tab] = 0
                e_x = float(dx.max()) / scale_ref
                msg += f" | pre-vs-local {e_loc:.3e} ({where(eng, int(dl.argmax()))}) post-vs-sum(pre) {e_x:.3e} ({where(eng, int(dx.argmax()))})"
            elif True:
                # per-parameter report of the worst few
                rows = []
                for n in f.order:
                    o = f.offset[n]
                    k = f.named[n].numel()
                    e = float(d[o:o + k].max()) / scale_ref
                    if e > 2e-5:
                        rows.append((e, n, int((d[o:o + k] > 2e-5 * scale_ref).sum()), k))
                rows.sort(reverse=True)
                msg += " | " + "; ".join(f"{n}: {e:.2e} ({c}/{k})" for e, n, c, k in rows[:6])
        elif snap:
            spre = pre.clone()
            dist.all_reduce(spre)
            torch.cuda.synchronize()
        print(f"[rank {rank}] it {it}: err {err:.2e} (repeat {rep:.1e}){msg}", flush=True)
        close_step()
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(3 if bad else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=6)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--armed", action="store_true")
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--train", type=int, default=0)
    ap.add_argument("--parent-engine", action="store_true")
    ap.add_argument("--rank", type=int, default=-1)
    ap.add_argument("--port", type=int, default=0)
    ap.add_argument("--loopback", type=int, default=0, help="1: exchange = no-op, 2: copy through pinned host memory on pool streams")
    ap.add_argument("--load", type=float, default=0.0, help="seconds of background load (run as its own process)")
    ap.add_argument("--with-load", action="store_true", help="loopback: start a background-load process next to the loop")
    a = ap.parse_args()
    if a.load > 0:
        return load_main(a)
    if a.loopback:
        bg = None
        if a.with_load:
            bg = subprocess.Popen([sys.executable, __file__, "--load", "100000"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            time.sleep(20)
        try:
            return loopback_main(a)
        finally:
            if bg is not None:
                bg.kill()
                bg.wait()
    if a.rank >= 0:
        return rank_main(a)
    import socket

    if a.parent_engine:
        import torch

        dev = torch.device("cuda:0")
        engs = []
        for _ in range(3):
            e = make_engine(dev)
            i_, n_ = inputs(0, dev, False)
            e.loss_and_grads(*i_, n_, train=False)
            engs.append(e)
        torch.cuda.synchronize()
    fails = 0
    for t in range(a.trials):
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, __file__, "--iters", str(a.iters), "--port", str(port), "--train", str(a.train)] + (["--armed"] if a.armed else []) + (["--big"] if a.big else [])
        t0 = time.time()
        ps = [subprocess.Popen(cmd + ["--rank", str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
        outs = []
        for p in ps:
            try:
                outs.append(p.communicate(timeout=600)[0])
            except subprocess.TimeoutExpired:
                p.kill()
                outs.append("TIMEOUT")
        rc = [p.returncode for p in ps]
        badlines = [ln for o in outs for ln in o.splitlines() if "BAD" in ln or "Error" in ln or "Traceback" in ln]
        fails += 1 if any(rc) else 0
        print(f"trial {t}: rc {rc} {time.time() - t0:.0f}s " + ("OK" if not any(rc) else "FAIL"), flush=True)
        for ln in badlines[:12]:
            print("   " + ln, flush=True)
        if any(r not in (0, 3) for r in rc):
            print(outs[0][-1500:], flush=True)
    print(f"SUMMARY armed={a.armed} big={a.big} train={a.train} snap={os.environ.get('PFPP_RACE_SNAP', '0')} env={ {k: v for k, v in os.environ.items() if k.startswith('PFPP_') or k.startswith('AMD_') or k.startswith('HIP_') or k.startswith('GPU_')} }: "
          f"{fails} failing trials of {a.trials}", flush=True)


if __name__ == "__main__":
    main()
