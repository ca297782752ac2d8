"""run-to-run stability of the training gradients inside ONE long-lived process (allocator reuse, engines created and dropped):
loss_and_grads on the golden inputs and a one-step module-surface fit, repeated; prints the relative deviation from the first run.
Atomics reorder sums (<= 2e-6); anything above that is an ordering bug.   usage: python tools/diag/grad_repeat.py [n] [churn]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
for p in (str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd"), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np
import torch

import test_gpu_train as TG
from oracle import weights

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
churn = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
gold = lambda name: dict(np.load(ROOT / "tests" / "golden" / f"{name}.npz"))
wsd = lambda which: weights.denoiser_state_dict() if which == "denoiser" else weights.vqvae_state_dict()
from pfpp_hip.train import DenoiserTrainEngine

inp, noise, _ = TG.golden_inputs(gold, dev)
base = [None, None]
worst = [0.0, 0.0]
import tempfile
for it in range(n):
    if churn:      # allocator churn: blocks of many sizes allocated and freed between the runs
        junk = [torch.empty(int(s), device=dev) for s in torch.randint(1 << 10, 1 << 24, (40,)).tolist()]
        del junk
    for r in range(2):
        eng = DenoiserTrainEngine(TG.make_module(wsd, dev))
        eng.loss_and_grads(*[v[r:r + 1] for v in inp], noise[r:r + 1], train=False)
        g = eng.flat.grads.cpu()
        del eng
        if base[r] is None:
            base[r] = g
        else:
            d = float((g.double() - base[r].double()).abs().max() / base[r].double().abs().max())
            worst[0] = max(worst[0], d)
            if d > 2e-6:
                print(f"iter {it} rank-slice {r}: loss_and_grads deviates {d:.3e}", flush=True)
print(f"loss_and_grads: worst deviation {worst[0]:.3e} over {n} repeats", flush=True)
base = None
for it in range(max(2, n // 4)):
    with tempfile.TemporaryDirectory() as td:
        model = TG._surface_model(dev, 100)
        TG._surface_fit(model, list(range(0, 8, 2)), 1, Path(td) / "alone", max_steps=1, accumulate_grad_batches=1)
        torch.cuda.synchronize()
        m = model.denoiser.train_engine().flat.exp_avg.cpu()
        del model
    if base is None:
        base = m
    else:
        d = float((m.double() - base.double()).abs().max() / base.double().abs().max())
        worst[1] = max(worst[1], d)
        if d > 2e-6:
            print(f"iter {it}: module-surface fit deviates {d:.3e}", flush=True)
print(f"module-surface fit: worst deviation {worst[1]:.3e}", flush=True)
