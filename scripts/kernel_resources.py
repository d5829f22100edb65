"""Register / LDS / spill figures of the built kernels, read from the code objects' metadata notes (no GPU, no recompile).

    python scripts/kernel_resources.py [object-name-substring ...]      # e.g. radial_mlp node_ops l2n_mid

Every `nequip_amd/csrc/build/*.o` holds a clang offload bundle; `llvm-objdump --offloading` unbundles the gfx950 code
object and `llvm-readelf --notes` prints the `amdhsa.kernels` records (.vgpr_count, .vgpr_spill_count,
.group_segment_fixed_size, ...).  `tests/test_kernel_resources.py` pins the figures DESIGN.md quotes for the kernels of the
benchmarked path."""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "nequip_amd", "csrc", "build")
LLVM = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def demangle(names):
    filt = shutil.which("c++filt") or os.path.join(LLVM, "llvm-cxxfilt")
    out = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True, check=True).stdout
    return out.strip().splitlines()


def kernels_of(obj_path):
    """{demangled kernel name: dict(vgpr, sgpr, vgpr_spill, sgpr_spill, lds, scratch)} of one build object."""
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(obj_path))
        shutil.copy(obj_path, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], capture_output=True, check=True, cwd=tmp)
        cos = [f for f in glob.glob(local + ".*") if "amdgcn" in f]
        if not cos:
            return {}
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", cos[0]], capture_output=True, text=True,
                               check=True).stdout
    recs, cur = [], None
    for ln in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", ln)
        if not m:
            continue
        key, val = m.group(1), m.group(2).strip()
        if key == "agpr_count" or (key == "args" and cur is None):
            pass
        if re.match(r"\s*- \.", ln) and key in ("agpr_count", "args"):  # first key of a kernel record
            cur = {}
            recs.append(cur)
        if cur is not None and key in ("name", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                                       "group_segment_fixed_size", "private_segment_fixed_size"):
            cur[key] = val
    recs = [r for r in recs if "name" in r and "vgpr_count" in r]
    names = demangle([r["name"] for r in recs])
    return {n: dict(vgpr=int(r["vgpr_count"]), sgpr=int(r.get("sgpr_count", 0)), vgpr_spill=int(r.get("vgpr_spill_count", 0)),
                    sgpr_spill=int(r.get("sgpr_spill_count", 0)), lds=int(r.get("group_segment_fixed_size", 0)),
                    scratch=int(r.get("private_segment_fixed_size", 0))) for n, r in zip(names, recs)}


def waves_per_simd(vgpr):
    """gfx950: 512 VGPRs per SIMD lane, allocation granularity 8."""
    return min(8, 512 // max(8, ((vgpr + 7) // 8) * 8))


if __name__ == "__main__":
    pats = sys.argv[1:]
    for obj in sorted(glob.glob(os.path.join(BUILD, "*.o"))):
        if pats and not any(p in os.path.basename(obj) for p in pats):
            continue
        ks = kernels_of(obj)
        print(f"== {os.path.basename(obj)} ({len(ks)} kernels)")
        for n, r in sorted(ks.items()):
            short = re.sub(r"\(.*$", "", n.replace("(anonymous namespace)::", "")).replace("void ", "").replace("nqa::", "")
            print(f"  {short[:78]:78s} vgpr {r['vgpr']:3d} ({waves_per_simd(r['vgpr'])} w/SIMD) spill {r['vgpr_spill']:3d} "
                  f"lds {r['lds']:6d} scratch {r['scratch']}")
