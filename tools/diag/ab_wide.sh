timeout 400 python -m pytest tests/test_gpu_train.py -m gpu -q -x -k "encoder_train" 2>&1 | tail -15
for v in 1 0 1 0; do echo "wide=$v"; PFPP_SA_TRAIN_WIDE=$v timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>&1 | tail -3 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d.get('extra', {}).get('final_loss'))
    elif 'rror' in l: print(l.strip()[:300])
"; done
