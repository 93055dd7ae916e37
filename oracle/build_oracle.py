"""Build the C part of the oracle (gcc).  Output: oracle/_build/librn_oracle.so.

Test infrastructure only — see the header of rn_oracle.c.  There is no oracle/_ref build:
the reference is pure Python on top of TensorFlow (absent here), so nothing of it compiles.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "librn_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "rn_oracle.c")
    hdr = os.path.join(HERE, "..", "include", "rn_math.h")
    os.makedirs(OUT_DIR, exist_ok=True)
    if (not force and os.path.exists(OUT)
            and os.path.getmtime(OUT) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return OUT
    cmd = ["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
           "-o", OUT, src, "-lm"]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
