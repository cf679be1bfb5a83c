"""In-tree build of libjsg.so (hand-written HIP for gfx950 + the C-ABI).  hipcc cross-compiles without a GPU."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libjsg.so")

SOURCES = ["jsg_kernels.hip", "jsg_stft_a.hip", "jsg_stft_b.hip", "jsg_engine.cpp", "jsg_host_math.cpp"]
HEADERS = ["jsg_internal.h", "jsg_block_queue.h", "jsg_exact_math.h", "jsg_stft_kernel.h", "jsg_colormap_tables.inc", os.path.join(ROOT, "include", "jsg.h")]
# device-compile flags of every .hip unit:
#   -fno-slp-vectorize: the kernel packs its complex arithmetic into v_pk_*_f32 by hand (re, im in one register pair); the automatic
#       SLP pass pairs unrelated scalars and pays ~140 register moves per FFT
#   -amdgpu-kernarg-preload-count: the leading scalar kernel arguments are delivered in SGPRs
HIP_FLAGS = ["-fno-slp-vectorize", "-mllvm", "-amdgpu-kernarg-preload-count=16"]
# every unit: only the C-ABI of include/jsg.h (JSG_API) leaves the library; the launchers, plan helpers and module probes stay inside
VISIBILITY = ["-fvisibility=hidden", "-fvisibility-inlines-hidden"]
# ... and per unit: the ILP-first machine scheduler for the 512 / 1024 / 2048 / 8192-point kernels (jsg_stft_a.hip explains and
# gives the measurements); the 4096-point kernels (jsg_stft_b.hip) keep the default one
UNIT_FLAGS = {"jsg_stft_a.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libjsg.so cannot be built (there is no CPU fallback)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(name: str, hip_flags, verbose: bool = False) -> str:
    """DEV: an experimental build of the library with extra device-compile flags (-mllvm options, -D switches of a patched tree ...) as
    tools/variants/libjsg_<name>.so, for tools/abbench and tools/strided_probe.py (A/B of several builds).  The product build below
    never uses these."""
    hipcc = _hipcc()
    vdir = os.path.join(ROOT, "tools", "variants")
    objdir = os.path.join(PKG, "build", "variants", name)
    os.makedirs(vdir, exist_ok=True)
    os.makedirs(objdir, exist_ok=True)
    out = os.path.join(vdir, f"libjsg_{name}.so")
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        if not src.endswith(".hip"):      # host objects are the product's
            o = os.path.join(PKG, "build", os.path.splitext(src)[0] + ".o")
            objs.append(o)
            continue
        objs.append(o)
        cmd = [hipcc, "-std=c++17", "-O3", "-fPIC", "-c", s, "-o", o, "-I", os.path.join(ROOT, "include"),
               f"--offload-arch={ARCH}", "-DJSG_DEV_KNOBS"] + VISIBILITY + HIP_FLAGS + ([] if os.environ.get("JSG_NO_UNIT_FLAGS") else UNIT_FLAGS.get(src, [])) + list(hip_flags)
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    subprocess.check_call([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", out] + objs)
    return out


def build_lib(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/* into jadespectrogram_amd/libjsg.so (only what is out of date)."""
    hipcc = _hipcc()
    objdir = os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    flavour_file = os.path.join(objdir, "flavour")
    flavour = "product"
    if not os.path.exists(flavour_file) or open(flavour_file).read().strip() != flavour:
        force = True
    objs = []
    recompiled, relinked = [], False
    cmds = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            recompiled.append(src)
            cmd = [hipcc, "-std=c++17", "-O3", "-fPIC", "-c", s, "-o", o, "-I", os.path.join(ROOT, "include")] + VISIBILITY
            if src.endswith(".hip"):
                cmd += [f"--offload-arch={ARCH}"] + HIP_FLAGS + UNIT_FLAGS.get(src, [])
            else:
                # host translation units: keep float arithmetic exactly as written (bit parity with the reference)
                cmd += ["-ffp-contract=off", "-D__HIP_PLATFORM_AMD__"]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            cmds.append(cmd)
    # the translation units are independent: compile them side by side (the two kernel units take about a minute each)
    procs = [subprocess.Popen(c) for c in cmds]
    failed = [c for c, p in zip(cmds, procs) if p.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    if force or _stale(LIB, objs):
        relinked = True
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    with open(flavour_file, "w") as f:
        f.write(flavour + "\n")
    # build record: what was compiled in this call and what was reused, for whoever reads the driver's log
    import json, time
    info = {"flavour": flavour, "recompiled": recompiled, "relinked": relinked, "lib_mtime": time.strftime("%Y-%m-%d %H:%M:%S", time.localtime(os.path.getmtime(LIB))),
            "objects": {os.path.basename(o): time.strftime("%Y-%m-%d %H:%M:%S", time.localtime(os.path.getmtime(o))) for o in objs}}
    try:
        info["commit"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
        info["dirty"] = bool(subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "jadespectrogram_amd", "include"],
                                                     stderr=subprocess.DEVNULL).decode().strip())
        with open(os.path.join(PKG, "_build_info.json"), "w") as f:   # travels with the snapshot (the GPU box has no .git)
            json.dump(info, f)
    except Exception:
        pass
    build_lib.last = info
    return LIB


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":      # python -m jadespectrogram_amd._build --variant NAME flags...
        print(build_variant(sys.argv[2], sys.argv[3:], verbose=True))
    else:
        print(build_lib(force="--force" in sys.argv, verbose=True))
