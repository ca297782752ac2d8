# A/B of two builds of the library on ONE box: tools/lab/_bin/lib_old.so vs lib_new.so (copied over the in-tree library in turn)
L=puzzlefusion-plusplus_amd/pfpp_hip/libpfpp_hip.so
for r in 1 2 3; do for v in old new; do cp tools/lab/_bin/lib_$v.so $L; echo -n "$v "; python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['extra']['final_loss'])
"; done; done
