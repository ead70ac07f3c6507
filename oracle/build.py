"""Build recipe for the C part of the oracle (TEST INFRASTRUCTURE ONLY).

The reference is pure Python (nothing to compile), so there is no ``oracle/_ref``; this compiles our own
C restatement ``gram_oracle.c`` into ``oracle/_build/libgram_oracle.so`` (git-ignored, travels to the GPU box).
"""
from __future__ import annotations

import subprocess
from pathlib import Path

HERE = Path(__file__).resolve().parent
OUT = HERE / "_build" / "libgram_oracle.so"


def build(force: bool = False) -> Path:
    src = HERE / "gram_oracle.c"
    if OUT.exists() and not force and OUT.stat().st_mtime >= src.stat().st_mtime:
        return OUT
    OUT.parent.mkdir(exist_ok=True)
    subprocess.run(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", str(src), "-o", str(OUT), "-lm"], check=True)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
