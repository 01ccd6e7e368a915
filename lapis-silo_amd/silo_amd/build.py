"""Compile the native pieces in-tree.

  lib/libsilo_gpu.so     HIP kernels + C ABI (include/silo_gpu.h), hipcc --offload-arch=gfx950
  lib/libsilo_engine.so  C++ host mirror of silo::query_engine (include/silo_engine.h), links libsilo_gpu
  lib/silo_query         CLI: load a data set directory (reference input formats), answer /query bodies from stdin
  lib/libbitprog_host.so g++ build of the bit-program interpreter for host-logic unit tests

hipcc cross-compiles gfx950 without a GPU, so this runs in the build container and the resulting
.so files travel to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # .../lapis-silo_amd
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(PKG, "host")
LIB = os.path.join(PKG, "lib")
INCLUDE = os.path.join(ROOT, "include")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        sys.stderr.write(proc.stdout + proc.stderr)
        raise RuntimeError("build step failed: " + " ".join(cmd))
    if "warning:" in proc.stderr:  # the builds are warning-free (-Wall -Wextra): a new warning should be seen
        sys.stderr.write(proc.stderr)
    return proc


def _glob(directory, suffixes):
    out = []
    if os.path.isdir(directory):
        for name in sorted(os.listdir(directory)):
            if name.endswith(suffixes):
                out.append(os.path.join(directory, name))
    return out


def hipcc_path():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def build_gpu(force=False):
    os.makedirs(LIB, exist_ok=True)
    target = os.path.join(LIB, "libsilo_gpu.so")
    sources = _glob(CSRC, (".hip",))
    headers = _glob(CSRC, (".h",)) + _glob(INCLUDE, (".h",))
    # one object per translation unit (lib/obj/, git-ignored), so that touching one kernel file recompiles only it
    obj_dir = os.path.join(LIB, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    objects = []
    for source in sources:
        obj = os.path.join(obj_dir, os.path.basename(source) + ".o")
        if force or _newer(obj, [source] + headers):
            _run([
                hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wextra", "-I", INCLUDE,
                "-c", source, "-o", obj,
            ])
        objects.append(obj)
    if force or _newer(target, objects):
        _run([hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", *objects, "-o", target, "-Wl,-rpath,/opt/rocm/lib", "-ldl"])
    return target


def build_engine(force=False):
    os.makedirs(LIB, exist_ok=True)
    target = os.path.join(LIB, "libsilo_engine.so")
    sources = _glob(HOST, (".cpp",))
    if not sources:
        return None
    deps = sources + _glob(HOST, (".h",)) + _glob(INCLUDE, (".h",)) + [os.path.join(LIB, "libsilo_gpu.so")]
    if force or _newer(target, deps):
        _run([
            "g++", "-O2", "-g", "-std=c++20", "-fPIC", "-shared", "-Wall", "-Wextra", "-I", INCLUDE, "-I", HOST,
            *sources, "-o", target, "-L", LIB, "-lsilo_gpu", "-Wl,-rpath,$ORIGIN", "-pthread", "-ldl",
        ])
    return target


def build_cli(force=False):
    """lib/silo_query: load a data set directory, answer /query bodies from stdin."""
    src = os.path.join(PKG, "tools", "silo_query.cpp")
    target = os.path.join(LIB, "silo_query")
    deps = [src, os.path.join(LIB, "libsilo_engine.so"), os.path.join(INCLUDE, "silo_engine.h")]
    if force or _newer(target, deps):
        _run([
            "g++", "-O2", "-std=c++20", "-Wall", "-Wextra", "-I", INCLUDE, src, "-o", target, "-L", LIB, "-lsilo_engine",
            "-lsilo_gpu", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + LIB, "-pthread",
        ])
    return target


def build_host_tools(force=False):
    os.makedirs(LIB, exist_ok=True)
    src = os.path.join(ROOT, "tests", "host_tools", "bitprog_host.cpp")
    if not os.path.exists(src):
        return None
    target = os.path.join(LIB, "libbitprog_host.so")
    if force or _newer(target, [src, os.path.join(CSRC, "bitprog.h")]):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I", INCLUDE, src, "-o", target])
    # hooks into the pure host functions of libsilo_engine.so (dates, lineage aliases, insertion parsing)
    logic_src = os.path.join(ROOT, "tests", "host_tools", "host_logic.cpp")
    logic_target = os.path.join(LIB, "libhost_logic.so")
    engine_lib = os.path.join(LIB, "libsilo_engine.so")
    if os.path.exists(logic_src) and os.path.exists(engine_lib) and (force or _newer(logic_target, [logic_src, engine_lib, os.path.join(CSRC, "layout_choice.h")])):
        _run(["g++", "-O2", "-std=c++20", "-fPIC", "-shared", "-I", INCLUDE, "-I", HOST, logic_src, "-o", logic_target, "-L", LIB,
              "-lsilo_engine", "-lsilo_gpu", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + LIB])
    return target


def build_all(force=False):
    return [build_gpu(force), build_engine(force), build_cli(force), build_host_tools(force)]


if __name__ == "__main__":
    for path in build_all(force="--force" in sys.argv):
        print(path)
