"""Build the gfx950 shared library (C ABI of include/relpose_gnn_hip.h) in-tree with hipcc.

    python relpose-gnn_amd/build.py            # or: from relpose_gnn_amd.build import build; build()

hipcc cross-compiles for gfx950 without a GPU; the resulting .so is git-ignored but travels with the
working tree.  Rebuilds only when a source is newer than the library.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "librelpose_gnn_hip.so")
SOURCES = ("gemm_f32.hip", "winograd.hip", "stem.hip", "stem_bf16.hip", "conv_bf16.hip", "encoder_ops.hip", "gnn_ops.hip", "forward.hip", "timing.hip", "host_ops.hip")
ARCH = "gfx950"
# What a PMC profile of the dominant (Winograd) kernel is stamped with: its translation unit + the one header whose DEVICE code it
# inlines (rpg_coherent.h: agent-scope slab accesses, arrival ticket; split out of rpg_common.h in round 5).  rpg_common.h
# itself stays out: it holds launch helpers and declarations of the OTHER units' entry points, and hashing it orphaned the committed
# profile on every unrelated declaration (round 3).
WINOGRAD_SOURCES = ("winograd.hip", "rpg_coherent.h")


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def source_digest(names=None) -> str:
    """12 hex digits identifying the kernel sources (csrc/* + the C header): profiles/ artefacts are stamped with it so
    that bench.py can tell whether a committed PMC profile was measured on the kernels it is running.  ``names``: only
    those files of csrc/ (a profile of ONE kernel is stamped with the digest of the translation unit that holds it and the
    shared header -- WINOGRAD_SOURCES for the dominant kernel -- so that work on other kernels does not orphan it)."""
    import hashlib
    h = hashlib.sha256()
    if names is not None:
        files = [os.path.join(CSRC, f) for f in sorted(names)]
    else:
        files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [
            os.path.join(os.path.dirname(HERE), "include", "relpose_gnn_hip.h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [
        os.path.join(os.path.dirname(HERE), "include", "relpose_gnn_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    defines = os.environ.get("RPG_BUILD_DEFINES", "").split()
    if defines:
        # probe builds (tools/probes/README.md) never replace the product library (a later plain build() kept a
        # -DRPG_PROBE_* library as the product because only source mtimes were compared): they are written next to it and
        # loaded through RPG_HIP_LIB
        return _build_to(os.path.join(LIB_DIR, "librelpose_gnn_hip_probe.so"), defines, verbose, obj_suffix=".probe.o")
    if not force and not needs_build():
        return LIB_PATH
    return _build_to(LIB_PATH, [], verbose)


def _build_to(lib_path: str, defines, verbose: bool, obj_suffix: str = ".o") -> str:
    os.makedirs(LIB_DIR, exist_ok=True)
    from concurrent.futures import ThreadPoolExecutor

    def compile_one(src):
        obj = os.path.join(LIB_DIR, src.replace(".hip", obj_suffix))
        # defines: extra -D flags of a probe build (RPG_BUILD_DEFINES, e.g. -DRPG_PROBE_WS64); empty for the product
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", *defines,
               "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj
    with ThreadPoolExecutor(max_workers=min(4, os.cpu_count() or 1)) as pool:      # the translation units are independent
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib_path + ".tmp", *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(lib_path + ".tmp", lib_path)
    return lib_path


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
