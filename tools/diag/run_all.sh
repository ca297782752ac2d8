timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for i in 1 2; do python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['extra']['final_loss'])
"; done
