"""count the branches inside every innermost loop of every kernel in a gfx950 assembly listing
    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only x.hip -o x.s ; python tools/diag/loop_branches.py x.s
A steady-state loop with data-dependent branches is several scheduling regions: the compiler cannot interleave loads / LDS
stores under the MFMAs across them."""
import re, sys

txt = open(sys.argv[1]).read().split("\n")
kernel = None
i = 0
while i < len(txt):
    ln = txt[i]
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        kernel = m.group(1)
    if "Inner Loop Header" in ln:
        label = ln.split(":")[0].strip()
        j = i + 1
        branches, mfma, lines = 0, 0, 0
        while j < len(txt):
            t = txt[j]
            if re.match(r"^\.LBB\d+_\d+:", t) and "Inner Loop Header" in t:
                break
            if "s_endpgm" in t:
                break
            if "s_cbranch" in t:
                if label in t:           # back edge: end of this loop
                    break
                branches += 1
            if "v_mfma" in t:
                mfma += 1
            lines += 1
            j += 1
        if mfma >= 8:
            print(f"{kernel[:70]:70s} loop {label:10s} {lines:5d} lines, {mfma:4d} MFMAs, {branches:2d} inner branches")
    i += 1
