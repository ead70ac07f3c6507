"""Build recipe for the C part of the oracle (TEST INFRASTRUCTURE ONLY).

The reference is pure Python (nothing to compile), so there is no ``oracle/_ref``; this compiles our own
C restatement ``gram_oracle.c`` into ``oracle/_build/libgram_oracle.so`` (git-ignored, travels to the GPU box).
"""
from __future__ import annotations

import subprocess
from pathlib import Path

HERE = Path(__file__).resolve().parent
OUT = HERE / "_build" / "libgram_oracle.so"


def _cpu_signature() -> str:
    """Model name + ISA flags of the host CPU: a library built with -march=native is only reused on the same kind of CPU."""
    import hashlib
    try:
        txt = Path("/proc/cpuinfo").read_text()
        keep = [ln for ln in txt.splitlines() if ln.startswith(("model name", "flags"))][:2]
    except OSError:
        keep = []
    return hashlib.sha1("\n".join(keep).encode()).hexdigest()


def build(force: bool = False) -> Path:
    src = HERE / "gram_oracle.c"
    sig_file = OUT.with_suffix(".cpu")
    sig = _cpu_signature()
    if (OUT.exists() and not force and OUT.stat().st_mtime >= src.stat().st_mtime and sig_file.exists()
            and sig_file.read_text() == sig):
        return OUT
    OUT.parent.mkdir(exist_ok=True)
    # -O3 -march=native (vectorised where the op order allows), no -ffast-math: the CPU figure is a fair many-core number and
    # the arithmetic stays the reference's.  Built on the machine it runs on (build() is called again on the GPU box when the
    # shipped .so was built for another CPU: see c_oracle._load).
    subprocess.run(["gcc", "-O3", "-march=native", "-fopenmp", "-shared", "-fPIC", str(src), "-o", str(OUT), "-lmvec", "-lm"], check=True)
    sig_file.write_text(sig)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
