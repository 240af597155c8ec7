"""In-tree build of the native pieces (hipcc cross-compiles gfx950 without a GPU).

  aim_amd/libaim_hip.so   C-ABI + kernels      (hipcc --offload-arch=gfx950)
  aim_amd/host/host       the C host program   (gcc, links libaim_hip.so)

`python -m aim_amd.build` builds both; nothing is JIT-compiled at import time.

The library is several translation units -- aim_capi.hip (the C-ABI, planning, the small batch-I/O kernels) and one tu_*.hip
per kernel family, each of which instantiates the kernels of ONE header -- compiled in parallel into build/obj/ and linked
once: a full rebuild takes as long as the slowest family instead of the sum, and editing one kernel header recompiles only
the units that include it.

  python -m aim_amd.build [--force] [--variant NAME --flags "-DAIM_..."]
      --variant: an A/B build of the library into build_ab/lib_NAME.so (objects in build/obj_NAME/), loaded with AIM_LIB=...
"""
import concurrent.futures
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(HERE, "libaim_hip.so")
HOST_SRC = os.path.join(HERE, "host", "host.c")
HOST_BIN = os.path.join(HERE, "host", "host")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + INCLUDE, "-I" + CSRC]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _deps(path, seen=None):
    """`path` and every project header it includes, transitively."""
    seen = seen if seen is not None else set()
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(path).read(), flags=re.M):
        for d in (CSRC, INCLUDE):
            _deps(os.path.join(d, inc), seen)
    return seen


def _toolchain_id():
    """Resolved compiler path + its --version text: part of the staleness key of the object files."""
    path = os.path.realpath(HIPCC)
    try:
        ver = subprocess.run([HIPCC, "--version"], capture_output=True, text=True, timeout=60).stdout.strip()
    except Exception as e:            # (no compiler: the build itself will say so)
        ver = "unavailable: %r" % (e,)
    return path + "\n" + ver


def units():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def build_lib(force=False, out=LIB, objdir=None, extra_flags=()):
    objdir = objdir or os.path.join(ROOT, "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    flag_stamp = os.path.join(objdir, "flags.txt")
    flags = HIP_FLAGS + list(extra_flags)
    stamp = " ".join(flags) + "\n" + _toolchain_id()      # (objects of another compiler or ROCm version are stale too -- ADVICE r04)
    if not os.path.exists(flag_stamp) or open(flag_stamp).read() != stamp:
        force = True
    jobs, objs = [], []
    for src in units():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _newer(obj, _deps(src)):
            jobs.append([HIPCC] + flags + ["-c", src, "-o", obj])
    if jobs:
        workers = max(1, min(len(jobs), os.cpu_count() or 1))
        with concurrent.futures.ThreadPoolExecutor(workers) as ex:
            list(ex.map(_run, jobs))
    if jobs or not os.path.exists(flag_stamp):
        open(flag_stamp, "w").write(stamp)
    if jobs or _newer(out, objs):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


def build_host(force=False):
    if not os.path.exists(HOST_SRC):
        return None
    if force or _newer(HOST_BIN, [HOST_SRC, LIB, os.path.join(INCLUDE, "aim_hip.h")]):
        _run(["gcc", "-O2", "-std=gnu11", "-Wall", "-I" + INCLUDE, "-o", HOST_BIN, HOST_SRC,
              "-L" + HERE, "-laim_hip", "-Wl,-rpath,$ORIGIN/..", "-lm", "-lpthread"])
    return HOST_BIN


def build_all(force=False):
    build_lib(force)
    build_host(force)


if __name__ == "__main__":
    args = sys.argv[1:]
    if "--variant" in args:
        name = args[args.index("--variant") + 1]
        extra = args[args.index("--flags") + 1].split() if "--flags" in args else []
        build_lib("--force" in args, out=os.path.join(ROOT, "build_ab", "lib_%s.so" % name),
                  objdir=os.path.join(ROOT, "build", "obj_" + name), extra_flags=extra)
    else:
        build_all(force="--force" in args)
