"""Build libsd_hip.so (gfx950) in-tree with hipcc.

    python speech-diarization_amd/build_native.py [--force] [--verbose]

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting .so is git-ignored but travels with the tree to the GPU box.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
INCLUDE = PKG_DIR.parent / "include"
LIB_PATH = PKG_DIR / "libsd_hip.so"
STAMP = PKG_DIR / "csrc" / ".build_stamp"
OFFLOAD_ARCH = "gfx950"

SOURCES = ["sd_api.hip", "sd_conv_gemm.hip", "sd_conv_gemm_f16.hip", "sd_res2net_f16.hip", "sd_fbank.hip", "sd_fbank_utt16.hip", "sd_fbank_generic.hip", "sd_pool.hip", "sd_scores.hip", "sd_asp_fused.hip", "sd_affinity.hip", "sd_ecapa.hip"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X path cannot be built on this machine")
    return exe


def _fingerprint() -> str:
    h = hashlib.sha256()
    files = sorted(CSRC.glob("*.hip")) + sorted(CSRC.glob("*.h")) + sorted(INCLUDE.glob("*.h"))
    for f in files:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()


def needs_build() -> bool:
    if not LIB_PATH.exists() or not STAMP.exists():
        return True
    return STAMP.read_text().strip() != _fingerprint()


def build(force: bool = False, verbose: bool = False, stamp: bool = False, variant: str | None = None, cflags: str = "") -> Path:
    """stamp=True: diagnostic build with in-kernel timeline stamps (-DSD_STAMP); never ship or time it.
    variant="name": an A/B build with extra `cflags` into variants/libsd_hip_<name>.so (selected at run time with
    SD_HIP_LIB=<path>; experiments only: the product library is always libsd_hip.so)."""
    if variant:
        return _build_variant(variant, cflags, verbose)
    if not force and not stamp and not needs_build():
        return LIB_PATH
    obj_dir = CSRC / "obj"
    obj_dir.mkdir(exist_ok=True)
    common = [
        _hipcc(), f"--offload-arch={OFFLOAD_ARCH}", "-O3", "-std=c++17", "-fPIC",
        "-fno-gpu-rdc", f"-I{INCLUDE}", f"-I{CSRC}", "-Wall", "-Wno-unused-function",
    ]
    if stamp:
        common.append("-DSD_STAMP")
    procs = []
    objs = []
    for src in SOURCES:
        obj = obj_dir / (Path(src).stem + ".o")
        objs.append(str(obj))
        cmd = common + ["-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0 or verbose:
            sys.stderr.write(f"--- {src}\n{out}\n")
        failed |= p.returncode != 0
    if failed:
        raise RuntimeError("hipcc failed; see output above")
    link = [_hipcc(), f"--offload-arch={OFFLOAD_ARCH}", "-shared", "-fPIC", "-o", str(LIB_PATH)] + objs
    res = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout)
        raise RuntimeError("link of libsd_hip.so failed")
    STAMP.write_text("stamp-build" if stamp else _fingerprint())
    return LIB_PATH


def _build_variant(name: str, cflags: str, verbose: bool) -> Path:
    out_dir = PKG_DIR / "variants"
    obj_dir = CSRC / "obj" / f"v_{name}"
    out_dir.mkdir(exist_ok=True)
    obj_dir.mkdir(parents=True, exist_ok=True)
    common = [_hipcc(), f"--offload-arch={OFFLOAD_ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", f"-I{INCLUDE}", f"-I{CSRC}",
              "-Wall", "-Wno-unused-function"] + cflags.split()
    procs, objs = [], []
    for src in SOURCES:
        obj = obj_dir / (Path(src).stem + ".o")
        objs.append(str(obj))
        cmd = common + ["-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0 or verbose:
            sys.stderr.write(f"--- {src}\n{out}\n")
        if p.returncode != 0:
            raise RuntimeError("hipcc failed; see output above")
    lib = out_dir / f"libsd_hip_{name}.so"
    res = subprocess.run([_hipcc(), f"--offload-arch={OFFLOAD_ARCH}", "-shared", "-fPIC", "-o", str(lib)] + objs,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout)
        raise RuntimeError("link failed")
    return lib


if __name__ == "__main__":
    variant, cflags = None, ""
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        variant, cflags = sys.argv[i + 1], (sys.argv[i + 2] if len(sys.argv) > i + 2 else "")
    path = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv, stamp="--stamp" in sys.argv, variant=variant, cflags=cflags)
    print(path)
