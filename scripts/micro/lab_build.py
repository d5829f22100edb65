"""Build one pair_lab binary per generator variant (cross-compiles here, runs on the GPU box).

    python scripts/micro/lab_build.py name[:struct]:ENV=VAL,ENV=VAL[:hipcc-flag,...] ...

e.g.  base  abl1::NQA_GEN_PAIR_ABL=1  occ3::NQA_GEN_PAIR_OCC=3  cu:l3n_mid:
Generated sources go to nequip_amd/csrc/generated_lab/<name>.hip (next to generated_spec/, so the relative includes hold),
binaries to scripts/micro/lab/<name>.out (both git-ignored; the binaries travel with the gpurun snapshot)."""
import concurrent.futures
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "nequip_amd", "csrc")
LAB_SRC = os.path.join(CSRC, "generated_lab")
LAB_BIN = os.path.join(ROOT, "scripts", "micro", "lab")
sys.path.insert(0, CSRC)


def emit(name, struct, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        import gen_spec
        importlib.reload(gen_spec)
        st = [s for s in gen_spec.baseline_structures() if s.name == struct][0]
        src = gen_spec._emit(st)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    path = os.path.join(LAB_SRC, name + ".hip")
    if not os.path.exists(path) or open(path).read() != src:
        open(path, "w").write(src)
    return path, "bwd_pair_split_kernel" in src


def compile_one(job):
    name, path, split, flags = job
    out = os.path.join(LAB_BIN, name + ".out")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", f"-I{CSRC}",
           f"-I{os.path.join(ROOT, 'include')}", f'-DSPEC_FILE="{path}"', "-Wno-unused-value", "-Wno-unused-function"]
    cmd += (["-DLAB_SPLIT"] if split else []) + flags + ["-save-temps=obj"] * 0
    cmd += [os.path.join(ROOT, "scripts", "micro", "pair_lab.hip"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    return name, r.returncode, (r.stderr or "")[-3000:]


if __name__ == "__main__":
    os.makedirs(LAB_SRC, exist_ok=True)
    os.makedirs(LAB_BIN, exist_ok=True)
    jobs = []
    for spec in sys.argv[1:]:
        parts = spec.split(":")
        name = parts[0]
        struct = parts[1] if len(parts) > 1 and parts[1] else "l2n_mid"
        env = dict(kv.split("=", 1) for kv in parts[2].split(",") if kv) if len(parts) > 2 else {}
        flags = [f_ for f_ in parts[3].split(",") if f_] if len(parts) > 3 else []
        path, split = emit(name, struct, env)
        jobs.append((name, path, split, flags))
    with concurrent.futures.ThreadPoolExecutor(max_workers=6) as ex:
        for name, rc, err in ex.map(compile_one, jobs):
            print(name, "ok" if rc == 0 else f"FAILED\n{err}")
