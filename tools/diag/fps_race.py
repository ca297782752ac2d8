"""Round 5: pfpp_fps / pfpp_ball_query / pfpp_sample_levels of one fixed input in a loop on the current stream while a second stream of
the same process runs the same kernels on other inputs — every result compared bit for bit with the first.  (tools/diag/enc_determinism.py
found the eval-mode encoder's farthest-point indices changing in ~7 % of the passes once another stream encodes at the same time.)

usage: [PFPP_LIB=path/to/variant.so] python tools/diag/fps_race.py [--iters N] [--other fps|gemm|none] [--F 8] [--N 512]"""
import argparse
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
for p_ in (str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--other", default="fps")
    ap.add_argument("--F", type=int, default=8)
    ap.add_argument("--N", type=int, default=512)
    ap.add_argument("--S", type=int, default=256)
    a = ap.parse_args()
    from pfpp_hip import _lib

    if os.environ.get("PFPP_LIB"):
        _lib.LIB_PATH = Path(os.environ["PFPP_LIB"]).resolve()
    from pfpp_hip import ops

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    x0 = (torch.rand(a.F, a.N, 3, generator=g) * 2 - 1).to(dev)
    x1 = (torch.rand(a.F, a.N, 3, generator=g) * 2 - 1).to(dev)
    big = torch.randn(4096, 512, generator=g).to(dev)
    from pfpp_hip.packing import PW

    wbig = PW(torch.randn(512, 512, generator=g).to(dev))
    side = torch.cuda.Stream(device=dev)
    if a.other in ("planes", "wd", "ln", "ew"):      # (only then: the first record of this script had none of these allocations)
        bigp = ops.SplitAct.empty(big.shape[0], big.shape[1], dev)
        ln_g, ln_b = torch.ones(512, device=dev), torch.zeros(512, device=dev)
        ops.layernorm(big, gamma=ln_g, beta=ln_b, out=bigp)
        ew = torch.randn(64 * 1024 * 1024 // 4, generator=g).to(dev)

    def work(x):
        idx, nx = ops.fps(x, a.S)
        ball = ops.ball_query(x, nx, 0.2, 32)
        return idx, nx, ball

    ref = [t.clone() for t in work(x0)]
    torch.cuda.synchronize()
    bad = [0, 0, 0]
    for it in range(a.iters):
        if a.other != "none":
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                if a.other == "fps":
                    o = work(x1)
                elif a.other == "gemm":          # fp32 activations: register-staged split-f16 GEMM (csrc/gemm.hip)
                    o = ops.linear(big, wbig)
                elif a.other == "planes":        # pre-split activations: LDS-DMA plane GEMM (csrc/gemm_pl.hip)
                    o = ops.linear(bigp, wbig)
                elif a.other == "wd":            # weight-direct GEMM (csrc/gemm_wd.hip)
                    o = ops.gemm_wd(bigp, wbig)
                elif a.other == "ew":            # a torch elementwise kernel: no LDS at all
                    o = ew + 1.0
                elif a.other == "ln":
                    o = ops.layernorm(big, gamma=ln_g, beta=ln_b)
                else:
                    raise SystemExit("unknown --other")
        got = work(x0)
        torch.cuda.synchronize()
        for k in range(3):
            if not torch.equal(got[k], ref[k]):
                bad[k] += 1
                if sum(bad) <= 12:
                    d = (got[k] != ref[k])
                    rows = d.flatten(1).any(1).nonzero().flatten().tolist()
                    first = [int(d[r].flatten().nonzero()[0]) for r in rows]
                    print(f"iter {it}: output {('fps_idx', 'new_xyz', 'ball_idx')[k]} differs in fragments {rows}, first differing element {first}", flush=True)
                    if k == 0:
                        r, c = rows[0], first[0]
                        print(f"    fragment {r} steps {max(c - 4, 0)}..{c + 2}: expected {ref[0][r, max(c - 4, 0):c + 3].tolist()} got {got[0][r, max(c - 4, 0):c + 3].tolist()}"
                              f"  (differing steps in the chain: {int(d[r].sum())}; got index seen earlier in the expected chain at step "
                              f"{(ref[0][r, :c] == got[0][r, c]).nonzero().flatten().tolist()})", flush=True)
    lib = _lib.load()
    if hasattr(lib, "pfpp_lab_fps_violations"):
        import ctypes

        v = (ctypes.c_ulonglong * 4)()
        lib.pfpp_lab_fps_violations(v, 1)
        print(f"exchange entries read: older step {v[0]}, newer step {v[1]}, wrong wave tag {v[2]} (reads checked by wave 0: {v[3]})")
    print(f"lib {_lib.LIB_PATH.name} other={a.other} F={a.F} N={a.N} S={a.S}: mismatching iterations fps_idx {bad[0]} new_xyz {bad[1]} ball_idx {bad[2]} of {a.iters}")


if __name__ == "__main__":
    main()
