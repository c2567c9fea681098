"""Builds libdsic_hip.so (gfx950) in-tree with hipcc.

    python domain-specific-image-compression_amd/build.py [--force]

Objects go to csrc/_obj/, the shared library next to this file.  hipcc
cross-compiles without a GPU, so this also runs in the GPU-less build
container; the built .so travels to the GPU box with the repository snapshot.
"""
from __future__ import annotations

import concurrent.futures
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libdsic_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
# -ffp-contract=off: the fp64 table math and the GDN epilogue must not be fused
# differently from their CPU restatements.
FLAGS = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}"]
FLAGS += os.environ.get("DSIC_EXTRA_FLAGS", "").split()   # e.g. -DWINO_RING=1 for A/B builds
# per-file flags.  metrics.hip: the SLP vectorizer packs the 11-tap filter chains of ssim_level_kernel into
# v_pk_fma_f32 with the weights duplicated into SGPR pairs - 200 spilled SGPRs (v_readlane in the loop) and a
# v_mov per packed operand; plain v_fma_f32 chains are shorter and spill nothing.
FILE_FLAGS = {"metrics.hip": ["-fno-slp-vectorize"]}


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def _deps_mtime():
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    return max([os.path.getmtime(h) for h in hdrs] + [os.path.getmtime(__file__)])


def _compile(src, obj):
    cmd = [HIPCC, *FLAGS, *FILE_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return r.stderr


def build_library(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    # objects built with other flags (A/B and diagnostic builds set DSIC_EXTRA_FLAGS) are stale
    flag_file = os.path.join(OBJ, ".flags")
    flag_str = " ".join(FLAGS) + " | " + repr(sorted(FILE_FLAGS.items()))
    if not os.path.exists(flag_file) or open(flag_file).read() != flag_str:
        force = True
    hdr_m = _deps_mtime()
    # objects whose source has left the tree are removed, never linked
    live = {os.path.basename(src) + ".o" for src in _sources()}
    for o in glob.glob(os.path.join(OBJ, "*.o")):
        if os.path.basename(o) not in live:
            os.remove(o)
    jobs, objs = [], []
    for src in _sources():
        obj = os.path.join(OBJ, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_m):
            jobs.append((src, obj))
    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for (src, _), warn in zip(jobs, ex.map(lambda j: _compile(*j), jobs)):
                if verbose or warn.strip():
                    print(f"[dsic build] {os.path.basename(src)}\n{warn}", file=sys.stderr)
    if jobs or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(flag_file, "w") as f:
        f.write(flag_str)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
