cd $GRAFT_REPO_ROOT
run() { local tag=$1; shift; for i in 1 2 3 4; do r=$(timeout 400 python -m pytest tests/test_gpu_train.py -q -m gpu -W always -k "$1" 2>&1 | grep -cE "re-running once"); echo "$tag run $i: retries=$r"; done; }
run A "loss_and_grads_vs_reference or dropout or two_rank_training_through"
run B "blocks_sequenced or two_rank_training_through"
run C "module_surface_runs or host_side or two_rank_training_through"
