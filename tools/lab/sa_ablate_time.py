"""tools/diag/enc_time.py on a library variant (PFPP_LAB_LIB = a libpfpp_hip.so built by tools/lab/sa_ablate.sh): results are wrong by
construction, only the per-launch times mean anything"""
import os, sys, runpy
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")]
from pfpp_hip import _lib
if os.environ.get("PFPP_LAB_LIB"):
    _lib.LIB_PATH = (ROOT / os.environ["PFPP_LAB_LIB"]).resolve()
runpy.run_path(str(ROOT / "tools/diag/enc_time.py"), run_name="__main__")
