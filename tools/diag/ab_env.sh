# A/B of environment settings on ONE box: bash tools/diag/ab_env.sh "A=1" "A=2 B=3" ...   (each argument = one configuration; 2 rounds)
for r in 1 2; do for cfg in "$@"; do echo -n "$cfg : "; env $cfg python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['extra']['final_loss'])
"; done; done
