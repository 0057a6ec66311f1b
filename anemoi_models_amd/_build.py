"""Build ``libanemoi_amd.so`` (the C-ABI kernel library) for gfx950 with hipcc, in-tree.

``hipcc`` cross-compiles without a GPU, so this runs in the build container; the resulting
``anemoi_models_amd/lib/libanemoi_amd.so`` travels with the tree to the GPU box.
"""

from __future__ import annotations

import glob
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libanemoi_amd.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]
# Sources with inline-asm MFMAs or hand-padded wide stores: their device listing is kept next to the object (lib/obj/<name>-hip-amdgcn-amd-amdhsa-gfx950.s,
# not shipped, not committed) so that tools/isa_hazard_audit.py -- and tests/test_host_logic.py -- can check that nothing
# touches an MFMA result before its wait states have passed (hipcc pads nothing around such a statement).
ASM_SOURCES = ("attention", "gemm", "weight_grad", "edge_attention")


def device_listing(name: str):
    """Path of the gfx950 listing of ``csrc/<name>.hip`` written by the last build, or None if absent / older than the source."""
    path = os.path.join(LIB_DIR, "obj", f"{name}-hip-amdgcn-amd-amdhsa-{ARCH}.s")
    src = os.path.join(CSRC, name + ".hip")
    if os.path.exists(path) and os.path.getmtime(path) >= os.path.getmtime(src):
        return path
    return None


def _hipcc_version(hipcc: str) -> str:
    """One line naming the compiler (the hand-counted wait states of the inline-asm kernels were validated against it)."""
    try:
        out = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout
        hip = next((ln.strip() for ln in out.splitlines() if ln.startswith("HIP version")), "")
        clang = next((ln.strip() for ln in out.splitlines() if "clang version" in ln), "")
        return "; ".join(v for v in (hip, clang) if v)[:200].replace('"', "'").replace("\\", "/") or "unknown"
    except OSError:
        return "unknown"


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build the gfx950 kernel library")
    return exe


def sources() -> list:
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps() -> list:
    return sources() + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(
        os.path.join(os.path.dirname(PKG_DIR), "include", "*.h")
    )


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    if any(os.path.getmtime(f) > t for f in _deps()):
        return True
    try:  # built by another hipcc than the one on this box?
        with open(os.path.join(LIB_DIR, "obj", ".hipcc_version")) as f:
            return f.read() != _hipcc_version(_hipcc())
    except (OSError, RuntimeError):
        return False  # (no stamp / no compiler here: a shipped library is used as it is)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every ``csrc/*.hip`` for gfx950 and link the shared library.  Returns its path.

    One builder at a time: an exclusive ``flock`` on ``lib/.build.lock`` is held around the whole build and staleness is
    re-checked under it, so the N ranks of ``bench.py --gpus N`` (or parallel test workers) that find the library stale at
    once compile it once -- the others wait and then use it.  The link goes to a temporary name and is moved into place
    atomically: no rank ever dlopens a half-written library."""
    if not force and not is_stale():
        return LIB_PATH
    import fcntl

    os.makedirs(os.path.join(LIB_DIR, "obj"), exist_ok=True)
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not is_stale():  # another process built it while this one waited
                return LIB_PATH
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force: bool, verbose: bool) -> str:
    hipcc = _hipcc()
    version = _hipcc_version(hipcc)
    # objects of another compiler are never reused: anemoi_build_info() names ONE hipcc (the one the hand-counted wait
    # states of the inline-asm kernels were validated against), so a compiler upgrade rebuilds everything
    stamp = os.path.join(LIB_DIR, "obj", ".hipcc_version")
    try:
        with open(stamp) as f:
            same_compiler = f.read() == version
    except OSError:
        same_compiler = False
    if not same_compiler:
        force = True
    hdr_time = max(os.path.getmtime(f) for f in _deps() if not f.endswith(".hip"))

    def compile_one(src: str) -> str:
        obj = os.path.join(LIB_DIR, "obj", os.path.basename(src)[:-4] + ".o")
        listed = os.path.basename(src)[:-4] not in ASM_SOURCES or device_listing(os.path.basename(src)[:-4]) is not None
        if not force and listed and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_time):
            return obj
        cmd = [hipcc, *FLAGS, f'-DANEMOI_HIPCC_VERSION="{version}"', "-c", src, "-o", obj]
        if os.path.basename(src)[:-4] in ASM_SOURCES:
            cmd.insert(1, "-save-temps=obj")
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{res.stderr}")
        if verbose and res.stderr.strip():
            print(res.stderr)
        base = os.path.basename(src)[:-4]
        if base in ASM_SOURCES:  # keep the device listing only (the other intermediates are ~100 MB)
            keep = {base + ".o", f"{base}-hip-amdgcn-amd-amdhsa-{ARCH}.s"}
            for f in glob.glob(os.path.join(LIB_DIR, "obj", base + "[-.]*")):
                if os.path.basename(f) not in keep:
                    os.remove(f)
        return obj

    with ThreadPoolExecutor(max_workers=min(4, len(sources()) or 1)) as pool:
        objs = list(pool.map(compile_one, sources()))
    tmp = f"{LIB_PATH}.{os.getpid()}.tmp"
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", tmp, *objs]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError(f"hipcc link failed:\n{res.stderr}")
    os.replace(tmp, LIB_PATH)
    with open(stamp, "w") as f:
        f.write(version)
    return LIB_PATH


if __name__ == "__main__":
    import sys

    print(build(force="--force" in sys.argv, verbose=True))
