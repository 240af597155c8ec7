#!/usr/bin/env python3
"""Register allocation of the kernels in libaim_hip.so, read from the CODE OBJECT's metadata notes (.vgpr_count / .agpr_count / .sgpr_count /
.private_segment_fixed_size) -- rocprofv3's dispatch record reports the granule-rounded ARCH VGPR count only (128 for a kernel that allocates 252 with its
accumulation registers: VERDICT r05 item 9). `python tools/codeobj_regs.py [substring]` prints them; pmc_summary.py imports kernel_regs()."""
import os, re, shutil, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_regs(lib=None):
    """{demangled kernel name: {vgpr, agpr, sgpr, scratch_bytes, lds_static_bytes}} for every gfx950 kernel bundled in `lib`."""
    lib = lib or os.environ.get("AIM_LIB") or os.path.join(ROOT, "aim_amd", "libaim_hip.so")
    out = {}
    with tempfile.TemporaryDirectory() as td:
        tmp = os.path.join(td, "lib.so")
        shutil.copy(lib, tmp)                                    # (llvm-objdump --offloading writes the bundles next to its input)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", tmp], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=td)
        for f in sorted(os.listdir(td)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(td, f)], capture_output=True, text=True).stdout
            for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
                blk = ".agpr_count:" + blk
                get = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, None])[1]
                sym = get("name")
                if not sym:
                    continue
                out[sym] = {"vgpr": int(get("vgpr_count") or 0), "agpr": int(get("agpr_count") or 0), "sgpr": int(get("sgpr_count") or 0),
                            "scratch_bytes": int(get("private_segment_fixed_size") or 0), "lds_static_bytes": int(get("group_segment_fixed_size") or 0)}
    names = list(out)
    if names:
        dem = subprocess.run([shutil.which("c++filt") or "c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
        out = {d.strip() or n: dict(out[n], symbol=n) for n, d in zip(names, dem)}
    return out


def lookup(regs, kernel_name):
    """The entry whose demangled name matches rocprofv3's Kernel_Name (which may or may not carry `void` / the argument list)."""
    norm = lambda s: re.sub(r"\s+", "", re.sub(r"^void\s+", "", s)).split("(")[0]
    k = norm(kernel_name)
    for n, v in regs.items():
        if norm(n) == k:
            return v
    return None


if __name__ == "__main__":
    sub = sys.argv[1] if len(sys.argv) > 1 else ""
    for n, v in sorted(kernel_regs().items()):
        if sub in n:
            print("%4d vgpr %4d agpr %4d sgpr %6d B scratch  %s" % (v["vgpr"], v["agpr"], v["sgpr"], v["scratch_bytes"], n))
