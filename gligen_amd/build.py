"""Build libgligen_amd.so (HIP kernels + engine + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs on the build container as well as on the
MI355X box. Objects are rebuilt only when a source/header is newer than the library.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
INCLUDE = HERE.parent / "include"
LIB = HERE / "libgligen_amd.so"
SOURCES = ["gemm.hip", "ffn.hip", "attention.hip", "norm.hip", "misc.hip", "convnext.hip", "train.hip", "engine.hip", "capi.hip"]
# attention: keep MFMA accumulators in VGPRs (gfx950 has one unified register file); the default AGPR
# form costs a v_accvgpr_read/write pair per accumulator per KV tile around the softmax rescale
# ffn: the GEGLU micro-steps of the row-local feed-forward are plain fp32 on purpose (packed fp32 is dearer beside MFMAs): no SLP packing
EXTRA_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "ffn.hip": ["-fno-slp-vectorize", "-fno-honor-nans"]}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the gfx950 kernels cannot be built")
    return exe


def _stale() -> bool:
    if not LIB.exists():
        return True
    lib_m = LIB.stat().st_mtime
    deps = list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h")) + list(CSRC.glob("*.inc")) + list(INCLUDE.glob("*.h"))
    return any(p.stat().st_mtime > lib_m for p in deps)


def build_native(force: bool = False, verbose: bool = False) -> Path:
    if not force and not _stale():
        return LIB
    hipcc = _hipcc()
    objdir = HERE / "build"
    objdir.mkdir(exist_ok=True)

    def compile_one(src: str) -> Path:
        obj = objdir / (src + ".o")
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-I", str(INCLUDE), "-c", str(CSRC / src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    build_kbench()
    return LIB


def build_kbench() -> Path:
    """Developer tool: per-shape kernel micro-benchmark (csrc/kbench.hip), linked against the library."""
    exe = HERE / "build" / "kbench"
    cmd = [_hipcc(), *FLAGS, "-I", str(INCLUDE), str(CSRC / "kbench.hip"), "-o", str(exe),
           "-L", str(HERE), "-lgligen_amd", "-Wl,-rpath,$ORIGIN/.."]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"kbench build failed:\n{r.stderr}")
    return exe


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
