"""In-tree build of the native pieces (hipcc cross-compiles gfx950 without a GPU).

  aim_amd/libaim_hip.so   C-ABI + kernels      (hipcc --offload-arch=gfx950)
  aim_amd/host/host       the C host program   (gcc, links libaim_hip.so)

`python -m aim_amd.build` builds both; nothing is JIT-compiled at import time.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libaim_hip.so")
HOST_SRC = os.path.join(HERE, "host", "host.c")
HOST_BIN = os.path.join(HERE, "host", "host")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_lib(force=False):
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "aim_hip.h")]
    if force or _newer(LIB, srcs):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        _run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
              "-I" + CSRC, "-o", LIB, os.path.join(CSRC, "aim_capi.hip")])
    return LIB


def build_host(force=False):
    if not os.path.exists(HOST_SRC):
        return None
    if force or _newer(HOST_BIN, [HOST_SRC, LIB]):
        _run(["gcc", "-O2", "-std=gnu11", "-Wall", "-I" + os.path.join(ROOT, "include"), "-o", HOST_BIN, HOST_SRC,
              "-L" + HERE, "-laim_hip", "-Wl,-rpath,$ORIGIN/..", "-lm", "-lpthread"])
    return HOST_BIN


def build_all(force=False):
    build_lib(force)
    build_host(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
