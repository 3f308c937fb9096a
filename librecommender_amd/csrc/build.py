"""Build ``liblibreco_hip.so`` (the C-ABI of include/libreco_hip.h) for gfx950 with hipcc.

Usage: ``python -m librecommender_amd.csrc.build [--force]``.  Cross-compiles without a GPU.
Objects go to ``build/`` (git-ignored), the library to ``librecommender_amd/lib/`` (in-tree so
it travels with the source snapshot to the GPU box).
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
OBJ_DIR = ROOT / "build" / "hip"
LIB_DIR = HERE.parent / "lib"
LIB_PATH = LIB_DIR / "liblibreco_hip.so"
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]
# per-file extras.  deepfm_l1_sb.hip: keep the MFMA accumulators in VGPRs — with the default (AGPR) form the compiler copied both
# accumulators into AGPRs and back around every field's MFMA chain (64 v_accvgpr_* per 48 MFMAs in the hot loop)
# softmax_ce.hip: same — the running-maximum rescale touches the accumulators with VALU instructions, so with the AGPR form they
# lived in VGPRs between stages and were copied in and out around every stage's MFMAs (128 v_accvgpr_* per stage)
EXTRA_FLAGS = {"deepfm_l1_sb.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "softmax_ce.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the gfx950 kernels cannot be built")
    return exe


def sources() -> list[Path]:
    return sorted(HERE.glob("*.hip"))


def _stale(out: Path, deps: list[Path]) -> bool:
    if not out.exists():
        return True
    t = out.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> Path:
    hipcc = _hipcc()
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    headers = sorted(HERE.glob("*.hpp")) + [ROOT / "include" / "libreco_hip.h"]
    jobs = []
    objs = []
    for src in sources():
        obj = OBJ_DIR / (src.stem + ".o")
        objs.append(obj)
        if force or _stale(obj, [src, *headers]):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src.name, []), "-c", str(src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, r

    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for src, r in ex.map(compile_one, jobs):
                if verbose:
                    print(f"[hipcc] {src.name}: {'ok' if r.returncode == 0 else 'FAILED'}")
                if r.returncode != 0:
                    sys.stderr.write(r.stdout + r.stderr)
                    raise RuntimeError(f"hipcc failed on {src}")
    if force or jobs or _stale(LIB_PATH, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *map(str, objs), "-o",
               str(LIB_PATH)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link of liblibreco_hip.so failed")
        if verbose:
            print(f"[hipcc] linked {LIB_PATH}")
    build_host(force, verbose)
    return LIB_PATH


HOST_SRC = HERE.parent / "hostsrc" / "host_loops.c"
HOST_LIB = LIB_DIR / "liblibreco_host.so"


def build_host(force: bool = False, verbose: bool = True) -> Path:
    """gcc build of the host-loop helper (plain C, no GPU code)."""
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    if force or _stale(HOST_LIB, [HOST_SRC]):
        gcc = shutil.which("gcc") or shutil.which("cc")
        if gcc is None:
            raise RuntimeError("gcc not found: liblibreco_host.so cannot be built")
        r = subprocess.run([gcc, "-O2", "-Wall", "-shared", "-fPIC", str(HOST_SRC), "-o", str(HOST_LIB)],
                           capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("gcc failed on host_loops.c")
        if verbose:
            print(f"[gcc] built {HOST_LIB}")
    return HOST_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
