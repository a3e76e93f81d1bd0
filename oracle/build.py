"""Builds oracle/libkg_oracle.so (gcc, plain C).  Test infrastructure only."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "kg_oracle.c")
LIB = os.path.join(HERE, "libkg_oracle.so")


def build(force=False):
    if (not force and os.path.exists(LIB)
            and os.path.getmtime(LIB) >= os.path.getmtime(SRC)):
        return LIB
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", SRC, "-o", LIB, "-lm"]
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
