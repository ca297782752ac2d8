# encoder alone (tools/diag/enc_time.py) and the overlapped iteration, alternating one environment switch: bash tools/diag/ab_enc.sh NAME
cd $GRAFT_REPO_ROOT
N=$1
for v in 0 1 0 1; do echo "== $N=$v"; env $N=$v timeout 300 python tools/diag/enc_time.py 2>&1 | grep -v "amdgpu.ids"; done
bash tools/diag/ab_env3.sh $N 3
