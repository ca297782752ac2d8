#!/bin/bash
# builds libpfpp_hip variants whose sa_train.hip is compiled with -DSA_ABL=n (run here, before gpurun): bash tools/lab/sa_ablate.sh 1 2 4 8 ...
# then on the GPU box: PFPP_LIB_PATH=tools/lab/_run/libpfpp_abl<n>.so python tools/diag/enc_time.py
R=$(cd $(dirname $0)/../.. && pwd)
C=$R/puzzlefusion-plusplus_amd/csrc
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$C -Xclang -target-feature -Xclang -fma-mix-insts -DPFPP_ATTEST_NO_MIX=1 \
    -Xclang -target-feature -Xclang -packed-fp32-ops -DPFPP_ATTEST_NO_PK=1 -munsafe-fp-atomics -DSA_ABL=$n -c $C/sa_train.hip -o /tmp/sa_train_abl$n.o 2>/dev/null &
done
wait
for n in "$@"; do
  objs=$(ls $C/build/*.o | grep -v "/sa_train.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/sa_train_abl$n.o -o $R/tools/lab/_run/libpfpp_abl$n.so && echo built abl$n
done
