"""Round 5: is the frozen (eval-mode) encoder of the module-surface tests a function of its inputs?  tools/diag/surface_race.py showed
the two-rank test's rare failure on BOTH sides of its comparison (a single process alone, no exchange), every parameter's gradient
moved by ~1 % — what a changed INPUT of the step looks like (a VQ code of one token), not a corrupted slice.

The same batch through model._extract_features N times — alone, next to a co-running process, and with a second stream of this process
issuing the encoder of another batch at the same time (the FeaturePipeline situation) — compared bit for bit with the first result; on
a mismatch the level outputs captured by pfpp_hip.encoder.extract_features say where it starts.

usage: python tools/diag/enc_determinism.py [--iters N] [--load] [--second-stream]"""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
for p_ in (str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd"), str(ROOT / "tests")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--load", action="store_true")
    ap.add_argument("--second-stream", action="store_true")
    ap.add_argument("--train-encoder", action="store_true")
    a = ap.parse_args()
    import test_gpu_train as tg
    from pfpp_hip import encoder as E
    from pfpp_hip.denoiser import layout_of
    from torch.utils.data import DataLoader

    dev = torch.device("cuda:0")
    model = tg._surface_model(dev, 100)
    model.train()
    if a.train_encoder:
        model.encoder.train()
    loader = DataLoader(tg._PuzzleList(range(0, 8, 2)), batch_size=2, shuffle=False, drop_last=True)
    batches = []
    for b in loader:
        b = model.on_before_batch_transfer(dict(b))
        b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
        b = model.on_after_batch_transfer(b)
        gt = torch.cat([b["part_trans"], b["part_rots"]], dim=-1).float().contiguous()
        torch.manual_seed(5)
        noise = torch.randn(gt.shape, device=dev)
        t = torch.randint(0, 1000, (gt.shape[0],), device=dev).long()
        noisy = model.noise_scheduler.add_noise(gt, noise, t)
        batches.append((b, noisy))
    torch.cuda.synchronize()
    pk = model.encoder.packed_train() if a.train_encoder else model.encoder.packed()
    L = model.encoder.cfg.ae.num_point

    def run(i, capture=None):
        b, noisy = batches[i]
        slot = layout_of(b["part_valids"], L).slot32
        return E.extract_features(pk, b["part_pcs"].contiguous(), noisy.contiguous(), slot, L, capture=capture)

    load = tg._start_gpu_load(3600) if a.load else None
    side = torch.cuda.Stream(device=dev) if a.second_stream else None
    try:
        cap0 = {}
        with torch.no_grad():
            ref = [t_.clone() for t_ in run(0, cap0)]
            cap0 = {k: v.clone() for k, v in cap0.items()}
            torch.cuda.synchronize()
            bad = 0
            for it in range(a.iters):
                if side is not None:
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        other = run(1)
                cap = {}
                got = run(0, cap)
                torch.cuda.synchronize()
                if not all(torch.equal(x, y) for x, y in zip(got, ref)):
                    bad += 1
                    d = (got[0] - ref[0]).abs()
                    print(f"iter {it}: latent differs in {int((d > 0).sum())} elements / {int((d.flatten(2).amax(2) > 0).sum() if d.dim() > 2 else 0)} "
                          f"tokens, max {float(d.max()):.3e}; xyz equal {torch.equal(got[1], ref[1])}", flush=True)
                    for k in cap0:
                        if k in cap and cap[k].shape == cap0[k].shape and not torch.equal(cap[k], cap0[k]):
                            dd = (cap[k].double() - cap0[k].double()).abs()
                            print(f"    {k}: {int((dd > 0).sum())} of {dd.numel()} elements differ, max {float(dd.max()):.3e} (scale {float(cap0[k].double().abs().max()):.3e})", flush=True)
                del cap
        print(f"mismatching iterations: {bad} of {a.iters} (load {a.load}, second stream {a.second_stream}, train-mode encoder {a.train_encoder})")
    finally:
        if load is not None:
            load.kill()
            load.wait()


if __name__ == "__main__":
    main()
