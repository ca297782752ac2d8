"""section times of the batched auto_aggl loop (synchronised around each section)  python tools/diag/aggl_sections.py"""
import sys, time, collections
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "puzzlefusion-plusplus_amd"))
import torch
from pfpp_hip import config, synthetic
import puzzlefusion_plusplus.auto_aggl as A

dev = torch.device("cuda:0")
torch.manual_seed(4321)
model = A.AutoAgglomerative(config.auto_aggl_config()).to(dev).eval()
with torch.no_grad():
    model.encoder.vector_quantization.embedding.weight.uniform_(-1.0, 1.0)
puzzles = []
for i in range(33):
    b = {k: v.to(dev) for k, v in synthetic.make_batch(500 + i, 1, num_points=1000).items()}
    b.update(synthetic.make_matching(b, seed=i))
    puzzles.append(b)
model.test_step(puzzles[0])
model.test_batch(puzzles[1:])
acc = collections.defaultdict(float)
cnt = collections.defaultdict(int)

def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); acc[label] += time.perf_counter() - t0; cnt[label] += 1
        return r
    setattr(obj, name, g)

wrap(A._PuzzleState, "__init__", "state init")
wrap(A._PuzzleState, "edge_features", "edge features")
wrap(A._PuzzleState, "after_verify", "after_verify (threshold, promotion, merge)")
wrap(A._PuzzleState, "result", "result (compose, metrics)")
wrap(model, "_make_step", "make_step")
wrap(model.noise_scheduler, "step", "scheduler.step")
orig_ms = model._make_step
def ms(*a, **k):
    fn = orig_ms(*a, **k)
    def g(x, t):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(x, t)
        torch.cuda.synchronize(); acc["denoise step (rotate+encode+transformer)"] += time.perf_counter() - t0; cnt["denoise step (rotate+encode+transformer)"] += 1
        return r
    return g
model._make_step = ms
vf = model.verifier.forward
def vfw(*a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = vf(*a, **k)
    torch.cuda.synchronize(); acc["verifier"] += time.perf_counter() - t0; cnt["verifier"] += 1
    return r
model.verifier.forward = vfw
torch.cuda.synchronize(); t0 = time.perf_counter()
outs = model.test_batch(puzzles[1:])
torch.cuda.synchronize(); total = time.perf_counter() - t0
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"{v * 1e3:9.1f} ms  x{cnt[k]:5d}  {k}")
print(f"{total * 1e3:9.1f} ms  total (with the extra synchronisations), accounted {sum(acc.values()) * 1e3:.1f}")
