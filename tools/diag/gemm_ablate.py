"""What bounds the split-f16 GEMM's K loop?  Builds libpfpp_hip.so variants with one ingredient removed
(-DPFPP_ABLATE=n, see csrc/gemm.hip) and times the big shapes.  Results are WRONG by construction; timing only.

    python tools/diag/gemm_ablate.py build      # here (hipcc cross-compiles), writes tools/diag/ablate_build/
    python tools/diag/gemm_ablate.py run        # on the GPU box
"""
import ctypes
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
CSRC = ROOT / "puzzlefusion-plusplus_amd" / "csrc"
OUT = ROOT / "tools" / "diag" / "ablate_build"
VARIANTS = {0: "baseline", 1: "global loads of the first K-tile only", 4: "no split / LDS stores", 5: "no loads, no stores", 6: "no loads, no stores, no barrier"}


def build():
    OUT.mkdir(parents=True, exist_ok=True)
    objs = [str(CSRC / "build" / f"{n}.o") for n in ("lib", "pointops", "vq", "transformer_ops", "attention", "edgefeat", "gemm_ring", "gemm_ws", "sa_fused",
                                                       "gemm_grad", "train_ops", "attention_bwd", "bn_train", "metrics", "merge", "augment")]
    for v in VARIANTS:
        o = OUT / f"gemm_{v}.o"
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT / 'include'}", f"-I{CSRC}",
                        *( [f"-DPFPP_ABLATE={v}"] if v < 100 else ["-DPFPP_HOIST=1"]), "-c", str(CSRC / "gemm.hip"), "-o", str(o)], check=True)
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, str(o), "-o", str(OUT / f"libpfpp_ablate_{v}.so")], check=True)
        print("built variant", v)


def run():
    sys.path.insert(0, str(ROOT / "puzzlefusion-plusplus_amd"))
    for v, name in VARIANTS.items():
        code = f"""
import sys; sys.path.insert(0, {str(ROOT / 'puzzlefusion-plusplus_amd')!r})
from pathlib import Path
from pfpp_hip import _lib
_lib.LIB_PATH = Path({str(OUT / f'libpfpp_ablate_{v}.so')!r})
import torch
from pfpp_hip import ops
from pfpp_hip.packing import PW
dev = torch.device('cuda:0')
for M, N, K in ((3850, 512, 512), (3850, 1536, 512), (3850, 512, 2048), (3850, 4096, 512), (16000, 512, 2048)):
    A = torch.randn(M, K, device=dev); pw = PW(torch.randn(N, K, device=dev) * 0.05)
    for _ in range(3): ops.linear(A, pw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.linear(A, pw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"  {{M}}x{{N}}x{{K}}: {{us:7.1f}} us  {{2.0*M*N*K/us/1e6:6.1f}} TF/s")
"""
        print(f"== {v}: {name}")
        subprocess.run([sys.executable, "-c", code], check=False)


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
