#!/usr/bin/env python3
"""Build ``libnequip_amd.so`` (the C-ABI of ``include/nequip_amd.h``) for gfx950 with hipcc.

Cross-compiles without a GPU.  The shared object is written next to this file (in-tree, git-ignored)
so it travels to the GPU box with the repository snapshot.

    python -m nequip_amd.csrc.build [--force] [--jobs N] [--no-torch-ops]

Second target (host C++ only): ``libnequip_amd_torch.so`` + ``nequip_amd_aoti_run`` -- the dispatcher ops registered
from C++ and a package runner for deployments without a Python interpreter (``build_torch_ops``).
"""

from __future__ import annotations

import argparse
import concurrent.futures
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
LIB_NAME = "libnequip_amd.so"
LIB_PATH = os.path.join(HERE, LIB_NAME)
OBJ_DIR = os.path.join(HERE, "build")
GEN_DIR = os.path.join(HERE, "generated")
SPEC_DIR = os.path.join(HERE, "generated_spec")

SOURCES = ["plan.cpp", "csr.hip", "tp_generic.hip", "edge_embed.hip", "edge_vectors.hip", "radial_mlp.hip", "node_ops.hip",
           "neighbor_list.hip", "wgrad.hip", "edge_pairs.hip", "energy_head.hip"]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libnequip_amd.so (set HIPCC or install ROCm)")


def _extra_flags(src: str) -> list:
    """Per-source flags.  generated_spec/*: no SLP vectorisation -- in the forward tensor-product kernels the packed
    fp32 ops it forms (v_pk_mul/fma_f32) cost more register shuffles (294 v_mov of 775 VALU instructions in the cfg-3
    middle layer) than they save, and 146 instead of 107 VGPRs, i.e. three instead of four wavefronts per SIMD."""
    if os.path.basename(os.path.dirname(src)) == "generated_spec" and os.environ.get("NQA_SPEC_SLP", "") in ("", "0"):
        return ["-fno-slp-vectorize"]
    return []


def _flags() -> list:
    return [
        f"--offload-arch={ARCH}",
        "-O3",
        "-std=c++17",
        "-fPIC",
        "-x",
        "hip",
        "-Wall",
        "-Wno-unused-function",
        "-Wno-unused-variable",
        "-Wno-unused-but-set-variable",
        f"-I{os.path.join(REPO, 'include')}",
        f"-I{HERE}",
    ]


def _digest(paths) -> str:
    h = hashlib.sha256()
    for p in sorted(paths):
        h.update(p.encode())
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(_flags()).encode())
    return h.hexdigest()


def _headers() -> list:
    hs = [os.path.join(REPO, "include", "nequip_amd.h")]
    for d in (HERE, GEN_DIR):  # (generated_spec/*.hip are sources, each hashed on its own)
        if os.path.isdir(d):
            hs += [os.path.join(d, f) for f in os.listdir(d) if f.endswith(".h")]
    return hs


def generate_tables(force: bool = False) -> None:
    gen_script = os.path.join(HERE, "gen_tables.py")
    wigner = os.path.join(REPO, "nequip_amd", "o3", "wigner.py")
    outs = [os.path.join(GEN_DIR, "cg_generated.h"), os.path.join(GEN_DIR, "sh_generated.h")]
    stamp = os.path.join(GEN_DIR, ".stamp")
    want = _digest([gen_script, wigner])
    if not force and all(os.path.exists(o) for o in outs) and os.path.exists(stamp):
        if open(stamp).read().strip() == want:
            return
    subprocess.run([sys.executable, gen_script], check=True, cwd=REPO)
    with open(stamp, "w") as f:
        f.write(want)


def generate_specs() -> list:
    sys.path.insert(0, HERE)
    import gen_spec  # noqa: E402

    return [os.path.relpath(p, HERE) for p in gen_spec.generate(SPEC_DIR)]


def _compile_one(src: str, force: bool) -> str:
    src_path = os.path.join(HERE, src)
    obj = os.path.join(OBJ_DIR, os.path.splitext(os.path.basename(src))[0] + ".o")
    stamp = obj + ".stamp"
    want = _digest([src_path] + _headers()) + "|" + " ".join(_extra_flags(src_path))
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return obj
    cmd = [_hipcc()] + _flags() + _extra_flags(src_path) + ["-c", src_path, "-o", obj]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{' '.join(cmd)}\n{res.stdout}\n{res.stderr}")
    if res.stderr.strip():
        sys.stderr.write(res.stderr)
    with open(stamp, "w") as f:
        f.write(want)
    return obj


def build(force: bool = False, jobs: int = 0, verbose: bool = True) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    generate_tables(force)
    sources = [s for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    sources += generate_specs()
    jobs = jobs or min(len(sources), os.cpu_count() or 1)
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile_one(s, force), sources))
    link_stamp = LIB_PATH + ".stamp"
    want = _digest(objs)
    if force or not os.path.exists(LIB_PATH) or not os.path.exists(link_stamp) or open(link_stamp).read().strip() != want:
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed:\n{' '.join(cmd)}\n{res.stdout}\n{res.stderr}")
        with open(link_stamp, "w") as f:
            f.write(want)
    if verbose:
        print(f"[nequip_amd] built {LIB_PATH} ({os.path.getsize(LIB_PATH) / 1e6:.1f} MB) from {len(objs)} objects")
    return LIB_PATH


TORCH_LIB_NAME = "libnequip_amd_torch.so"
TORCH_LIB_PATH = os.path.join(HERE, TORCH_LIB_NAME)
RUNNER_PATH = os.path.join(HERE, "nequip_amd_aoti_run")


def _cxx() -> str:
    for cand in (os.environ.get("CXX"), shutil.which("g++"), shutil.which("c++")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("no C++ compiler found for libnequip_amd_torch.so (set CXX)")


def build_torch_ops(force: bool = False, verbose: bool = True):
    """``libnequip_amd_torch.so`` (the dispatcher ops registered from C++, ``torch_ops/nequip_amd_torch.cpp``) and the
    stand-alone package runner ``nequip_amd_aoti_run``: host C++ only (g++ against this interpreter's libtorch), linked to
    ``libnequip_amd.so`` next to them.  Returns (library path, runner path)."""
    import torch
    from torch.utils import cpp_extension as ce

    tdir = os.path.join(HERE, "torch_ops")
    lib_src, run_src = os.path.join(tdir, "nequip_amd_torch.cpp"), os.path.join(tdir, "aoti_run.cpp")
    tlib = ce.library_paths()[0]
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    common = ["-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM",
              f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", f"-I{os.path.join(REPO, 'include')}"]
    common += [f"-I{p}" for p in ce.include_paths()] + [f"-I{os.path.join(rocm, 'include')}"]
    link = [f"-L{tlib}", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10", "-lc10_hip", f"-L{HERE}"]
    rpath = ["-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}"]
    steps = [
        (TORCH_LIB_PATH, [lib_src], ["-fPIC", "-shared"], ["-lnequip_amd"]),
        (RUNNER_PATH, [run_src], [], ["-lnequip_amd_torch", "-lnequip_amd"]),
    ]
    for out, srcs, extra, libs in steps:
        stamp = out + ".stamp"
        want = _digest(srcs + [os.path.join(REPO, "include", h) for h in ("nequip_amd.h", "nequip_amd_torch.h")]) + "|" + torch.__version__ + "|" + " ".join(common)
        if not force and os.path.exists(out) and os.path.exists(stamp) and open(stamp).read().strip() == want:
            continue
        cmd = [_cxx()] + common + extra + srcs + ["-o", out] + link + libs + rpath
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"building {os.path.basename(out)} failed:\n{' '.join(cmd)}\n{res.stdout}\n{res.stderr}")
        with open(stamp, "w") as f:
            f.write(want)
    if verbose:
        print(f"[nequip_amd] built {TORCH_LIB_PATH} and {RUNNER_PATH}")
    return TORCH_LIB_PATH, RUNNER_PATH


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=0)
    ap.add_argument("--no-torch-ops", action="store_true", help="skip libnequip_amd_torch.so / nequip_amd_aoti_run")
    a = ap.parse_args()
    build(force=a.force, jobs=a.jobs)
    if not a.no_torch_ops:
        build_torch_ops(force=a.force)
