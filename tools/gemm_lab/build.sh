#!/bin/bash
# builds the lab executables next to this file (they link the in-tree libpfpp_hip.so)
set -e
cd "$(dirname "$0")"
ROOT=$(cd ../.. && pwd)
python3 -c "import sys; sys.path.insert(0, '$ROOT/puzzlefusion-plusplus_amd'); from pfpp_hip import build; build.build()"
LIBDIR=$ROOT/puzzlefusion-plusplus_amd/pfpp_hip
for f in lab lab2 trprobe; do
  [ -f $f.hip ] || continue
  hipcc --offload-arch=gfx950 -O2 -std=c++17 -I$ROOT/include $f.hip -o $f -L$LIBDIR -lpfpp_hip -Wl,-rpath,'$ORIGIN/../../puzzlefusion-plusplus_amd/pfpp_hip' 2>&1 | grep -E "error" || true
done
ls -la lab lab2 trprobe
