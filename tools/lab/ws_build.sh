#!/bin/bash
# builds the wave-specialised GEMM probe variants ("ABL PRIO") into tools/lab/_run/ws_a<ABL>_p<PRIO>, and the W-direct baseline
cd "$(dirname "$0")/../.." && mkdir -p tools/lab/_run
FL="--offload-arch=gfx950 -O3 -std=c++17 -Itools/lab -Xclang -target-feature -Xclang -fma-mix-insts -Xclang -target-feature -Xclang -packed-fp32-ops"
build() { local a=$1 pr=${2:-0} pf=${3:-0} nl=${4:-4} lp=${5:-0}; /opt/rocm/bin/hipcc $FL -DABL=$a -DPRIO=$pr -DPFD=$pf -DNLD=$nl -DLPRIO=$lp -DWS_SPIN_LIMIT=2000000 tools/lab/gemm_ws_probe.hip -o tools/lab/_run/ws_a${a}_p${pr}_f${pf}_n${nl}_l${lp} 2>&1 | grep -E "error|spill"; }
for v in "$@"; do build $v & done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -DMT_=2 -DNT_=1 -DDEPTH=3 tools/lab/gemm_wdirect_probe.hip -o tools/lab/_run/wdbase 2>&1 | grep -E "error|spill" &
wait
