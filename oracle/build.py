"""Builds the oracle's C pieces (test / baseline infrastructure) into oracle/_build/.

  _build/libroaring_port.so   gcc -O3 -fopenmp: the reference's Mutations algorithm over roaring-format
                              containers (oracle/roaring_port.c), used as bench.py's cpu_baseline ("port").

The reference itself cannot be compiled here (oracle/_ref is therefore never produced): every
translation unit on the path needs CRoaring, oneTBB, Boost, spdlog and nlohmann_json >= 3.6, none of
which is in the image.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build")


def build_all(force=False):
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(HERE, "roaring_port.c")
    target = os.path.join(OUT, "libroaring_port.so")
    if force or not os.path.exists(target) or os.path.getmtime(src) > os.path.getmtime(target):
        cmd = ["gcc", "-O3", "-std=c11", "-fPIC", "-shared", "-fopenmp", "-mpopcnt", "-mavx2", "-mbmi2", "-Wall", "-Wextra",
               src, "-o", target]
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            sys.stderr.write(proc.stdout + proc.stderr)
            raise RuntimeError("building the oracle port failed")
    return [target]


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv))
