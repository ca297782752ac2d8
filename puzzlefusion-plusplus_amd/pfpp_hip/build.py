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
    "tlayer.hip": [],
    "heads.hip": ["-munsafe-fp-atomics"],
    "lnlin_small.hip": [],
    "gemm_small.hip": [],
    "gemm_wd.hip": [],
    "embed_small.hip": ["-ffp-contract=off"],
    "gemm_grad.hip": ["-munsafe-fp-atomics"],
    "train_ops.hip": ["-munsafe-fp-atomics"],
    "ada_bwd.hip": [],
    "embed_train.hip": ["-ffp-contract=off"],
    "attention_bwd.hip": [],
    "bn_train.hip": [],
    "metrics.hip": ["-ffp-contract=off"],
    "merge.hip": ["-ffp-contract=off"],
    "augment.hip": ["-ffp-contract=off"],
}
# -fma-mix-insts (target feature off): without it the compiler may fold `(_Float16)(a * b + c)` into v_fma_mixlo_f16 — ONE rounding of
# the exact value — while the `x - (float)hi` of the same hi / lo split goes through `v_cvt_f16_f32(RN32(x))`: in the rare
# double-rounding case the two hi's differ by an fp16 ulp and hi + lo is off by 5e-4 relative.  With the mix
# instructions unavailable every fp32 -> fp16 conversion is a plain v_cvt of the rounded fp32 value, so a split can only ever
# see one hi.  tests/test_abi_and_host.py disassembles the library and checks that none is left.
NO_MIX = ["-Xclang", "-target-feature", "-Xclang", "-fma-mix-insts", "-DPFPP_ATTEST_NO_MIX=1"]     # the switch and its attestation travel together (pfpp_build_info)
# -packed-fp32-ops (target feature off): no v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 anywhere in the library.  Round 5: with them, fps_kernel
# picked a wrong farthest point once in 10^2 .. 10^4 launches whenever a GEMM of another stream shared its CUs — the running minimum of a
# lane's first point kept a stale value in lanes 52-61 of a wave (tools/diag/fps_race.py: 490 wrong chains in 48,000 launches
# with the packed instructions in five different builds of the kernel, 0 in 48,000 + 8,000 in the two builds without them; exchange slots,
# barrier flavour, LDS contents and point loads all ruled out).  The same instructions sat in the hi / lo splits of the GEMM operand
# staging (x - hi feeding v_cvt_pk_f16_f32).  tests/test_abi_and_host.py checks the disassembly; PFPP_PACKED_FP32=1 (lab) leaves them on.
NO_PK = [] if os.environ.get("PFPP_PACKED_FP32") == "1" else ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-DPFPP_ATTEST_NO_PK=1"]
# PFPP_CHAIN_PRIO=n (lab, build time): s_setprio n in the kernels of the training step's dependency chain (csrc/pfpp_common.h)
CHAIN_PRIO = [f"-DPFPP_CHAIN_PRIO={int(os.environ['PFPP_CHAIN_PRIO'])}"] if os.environ.get("PFPP_CHAIN_PRIO") else []
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{INCLUDE}", f"-I{CSRC}", *NO_MIX, *NO_PK, *CHAIN_PRIO]


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
    # the linked library newer than every source and header: nothing to do — also where the object files did not travel (the GPU box
    # gets libpfpp_hip.so but not csrc/build/, .gpurunignore): without this a test fixture there would recompile all 25 units
    if not force and not _stale(LIB_PATH, [CSRC / src for src in SOURCES] + headers):
        return LIB_PATH
    objs, jobs = [], []
    for src, extra in SOURCES.items():
        s = CSRC / src
        o = objdir / (s.stem + ".o")
        objs.append(o)
        if force or _stale(o, [s, *headers]):
            jobs.append([hipcc, *COMMON, *extra, "-c", str(s), "-o", str(o)])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        # the x86 half of the compilation does not know the AMDGPU feature name NO_MIX switches off: its notice is noise
        err = "\n".join(ln for ln in r.stderr.splitlines() if ln.strip() and "'-fma-mix-insts' is not a recognized feature" not in ln
                        and "'-packed-fp32-ops' is not a recognized feature" not in ln)
        if err:
            print(err, file=os.sys.stderr, flush=True)
        if r.returncode != 0:
            raise subprocess.CalledProcessError(r.returncode, cmd)

    if jobs:          # translation units are independent: one hipcc per core (PFPP_BUILD_JOBS overrides)
        from concurrent.futures import ThreadPoolExecutor

        workers = max(1, min(len(jobs), int(os.environ.get("PFPP_BUILD_JOBS", os.cpu_count() or 1))))
        with ThreadPoolExecutor(workers) as pool:
            list(pool.map(run, jobs))
    if force or _stale(LIB_PATH, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-o", str(LIB_PATH)]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in os.sys.argv, verbose=True))
