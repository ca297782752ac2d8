"""Build libpfpp_hip.so (gfx950) in-tree with hipcc.

The shared object lands next to this file so it travels with the source tree
(it is git-ignored, not gpurun-ignored).  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE.parent / "csrc"
INCLUDE = HERE.parent.parent / "include"
LIB_PATH = HERE / "libpfpp_hip.so"

# file -> extra flags.  The point-cloud / VQ / scheduler kernels must reproduce the
# CPU's rounding sequence, so FMA contraction is off there and fused operations are
# written explicitly; the GEMM's MFMA path does not care.
SOURCES = {
    "lib.hip": [],
    "pointops.hip": ["-ffp-contract=off"],
    "vq.hip": ["-ffp-contract=off"],
    "transformer_ops.hip": ["-ffp-contract=off"],
    "attention.hip": [],
    "edgefeat.hip": ["-ffp-contract=off"],
    "gemm.hip": [],
    "gemm_pl.hip": ["-munsafe-fp-atomics"] + (["-DPFPP_PL_LAB"] if os.environ.get("PFPP_PL_LAB") else []),
    "sa_fused.hip": [],
    "sa_train.hip": ["-munsafe-fp-atomics"],
    "tblock_small.hip": [],
    "gemm_grad.hip": ["-munsafe-fp-atomics"],
    "train_ops.hip": ["-munsafe-fp-atomics"],
    "attention_bwd.hip": [],
    "bn_train.hip": [],
    "metrics.hip": ["-ffp-contract=off"],
    "merge.hip": ["-ffp-contract=off"],
    "augment.hip": ["-ffp-contract=off"],
}
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{INCLUDE}", f"-I{CSRC}"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libpfpp_hip.so cannot be built")
    return exe


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile every HIP source for gfx950 and link libpfpp_hip.so; returns its path."""
    hipcc = _hipcc()
    objdir = CSRC / "build"
    objdir.mkdir(exist_ok=True)
    headers = [INCLUDE / "pfpp.h", CSRC / "pfpp_common.h", CSRC / "gemm_common.h", CSRC / "sa_common.h"]
    objs = []
    for src, extra in SOURCES.items():
        s = CSRC / src
        o = objdir / (s.stem + ".o")
        objs.append(o)
        if force or _stale(o, [s, *headers]):
            cmd = [hipcc, *COMMON, *extra, "-c", str(s), "-o", str(o)]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
    if force or _stale(LIB_PATH, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-o", str(LIB_PATH)]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in os.sys.argv, verbose=True))
