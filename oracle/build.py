"""TEST INFRASTRUCTURE: build oracle/_ref-free C restatement -> oracle/libpfpp_oracle.so (gcc)."""
from __future__ import annotations

import subprocess
from pathlib import Path

HERE = Path(__file__).resolve().parent
SRC = HERE / "pfpp_oracle.c"
LIB = HERE / "libpfpp_oracle.so"


def build(force: bool = False) -> Path:
    if force or not LIB.exists() or LIB.stat().st_mtime < SRC.stat().st_mtime:
        subprocess.run(
            ["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-shared", "-fPIC",
             str(SRC), "-o", str(LIB), "-lm"],
            check=True,
        )
    return LIB


if __name__ == "__main__":
    print(build(force=True))
