#!/bin/bash
# builds the W-direct probe variants ("MT NT DEPTH [XCD] [ABL]") into tools/lab/_bin/wd_<MT>_<NT>_<D>_x<XCD>_a<ABL>
cd "$(dirname "$0")/../.." && mkdir -p tools/lab/_bin
build() { local mt=$1 nt=$2 d=$3 x=${4:-1} a=${5:-0} pf=${6:-0} kt=${7:-1} il=${8:-0}; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -DMT_=$mt -DNT_=$nt -DDEPTH=$d -DXCD=$x -DABL=$a -DPF=$pf -DKT_=$kt -DIL=$il \
  tools/lab/gemm_wdirect_probe.hip -o tools/lab/_bin/wd_${mt}_${nt}_${d}_x${x}_a${a}_p${pf}_k${kt}_i${il} 2>&1 | grep -E "error|spill" ; }
for v in "$@"; do build $v & done; wait
