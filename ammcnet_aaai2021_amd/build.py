"""Build libammc_hip.so (hipcc, gfx950 only) in-tree, next to this file.

    python -m ammcnet_aaai2021_amd.build [--force]
    AMMC_HIPCC_FLAGS="-DAMMC_TAP_DEBUG" python -m ammcnet_aaai2021_amd.build --force     # profiling build (AMMC_S16_DBG)

hipcc cross-compiles without a GPU, so this runs in the authoring container; the
built .so travels to the GPU box with the repo snapshot (it is git-ignored).
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libammc_hip.so")
STAMP = os.path.join(HERE, ".libammc_hip.stamp")
ARCH = "gfx950"


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest() -> str:
    h = hashlib.sha256()
    files = sources() + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    files.append(os.path.join(os.path.dirname(HERE), "include", "ammc_hip.h"))
    h.update(os.environ.get("AMMC_HIPCC_FLAGS", "").encode())
    for f in files:
        with open(f, "rb") as fp:
            h.update(f.encode())
            h.update(fp.read())
    return h.hexdigest()


def file_digests() -> dict:
    """sha256[:12] of every source file of the library: what a committed profile is tied to (a PMC figure of a kernel is
    quoted by bench.py only while the file that defines the kernel, the shared header and the C ABI are what they were
    when the profile was taken) - compiled into the library (`ammc_source_digests()`)"""
    files = sources() + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    files.append(os.path.join(os.path.dirname(HERE), "include", "ammc_hip.h"))
    out = {}
    for f in files:
        if os.path.basename(f) == "capi_misc.hip":           # (holds the digests themselves)
            continue
        with open(f, "rb") as fp:
            out[os.path.basename(f)] = hashlib.sha256(fp.read()).hexdigest()[:12]
    return out


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def build(force: bool = False, verbose: bool = False, variant: str = "") -> str:
    """variant: an A/B build next to the shipped one (`libammc_hip_<variant>.so`, flags from AMMC_HIPCC_FLAGS; tools
    select it with AMMC_LIB=<path>).  The shipped library and its stamp are not touched."""
    dig = _digest()
    LIB = globals()["LIB"] if not variant else os.path.join(HERE, f"libammc_hip_{variant}.so")
    STAMP = globals()["STAMP"] if not variant else os.path.join(HERE, f".libammc_hip_{variant}.stamp")
    if not force and os.path.exists(LIB) and os.path.exists(STAMP):
        with open(STAMP) as fp:
            if fp.read().strip() == dig:
                return LIB
    objs = []
    obj_dir = os.path.join(HERE, "build" + ("_" + variant if variant else ""))
    os.makedirs(obj_dir, exist_ok=True)
    procs = []
    for src in sources():
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        cmd = [hipcc_path(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math"]
        cmd += os.environ.get("AMMC_HIPCC_FLAGS", "").split()
        if os.path.basename(src) == "capi_misc.hip":
            cmd.append('-DAMMC_SRC_DIGESTS="' + ",".join(f"{k}={v}" for k, v in sorted(file_digests().items())) + '"')
        cmd += ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
    cmd = [hipcc_path(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout.decode()}")
    with open(STAMP, "w") as fp:
        fp.write(dig)
    return LIB


if __name__ == "__main__":
    if "--digests" in sys.argv:
        import json
        print(json.dumps(file_digests()))
        sys.exit(0)
    var = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else ""
    print(build(force="--force" in sys.argv, verbose=True, variant=var))
