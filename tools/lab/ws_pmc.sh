#!/bin/bash
# effective clock and matrix-pipe occupancy of the probes (counters in their own passes)
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out/pmc
for b in ws_a0_p0_f0 ws_a2_p0_f0 ws_a5_p0_f0 wdbase; do
  for c in "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
    tag=$(echo $c | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc/${b}_$tag -o out --output-format csv -- tools/lab/_run/$b 3850 512 2048 > /dev/null 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc/*')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
        print(d.split('/')[-1], {k: round(sum(v[5:]) / max(1, len(v[5:])), 1) for k, v in acc.items()})
    for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
        rows = list(csv.DictReader(open(f)))
        ds = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows][5:]
        print('   kernel avg ns', sum(ds) / max(1, len(ds)), 'n', len(ds))
PY
