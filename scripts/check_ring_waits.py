#!/usr/bin/env python3
"""Re-count, in the ISA of the built library, what the LDS-ring pair kernels' counted waits assume (gen_spec.py, "LDS ring").

`bwd_pair_ring_kernel` orders its LDS reads behind the LDS-DMA copies with `s_waitcnt vmcnt(N)`: N = the vector-memory
operations (copies, result stores, atomics) the wavefront issues between a chunk's copies and its evaluation -- one pair's
worth of each in the steady state.  hipcc does not see the copies (inline asm), so nothing checks N but this script:

  * the loop body must contain AT LEAST N counted operations (fewer: the wait would let a chunk be read before it landed);
  * hipcc must not have put a wait of its own (`vmcnt(k)`, k < 16, outside the rare index-block reload) into the loop: it
    would drain the ring every iteration (correct, but the kernel would be the register kernel again);
  * M0 is written only by the copy statements.

Checked: every `bwd_pair_ring_kernel` instantiation of every structure.  NOT checked: `bwd_pair_split_ring_kernel` (l_max 3: one
loop per part inside a switch; its two heaviest parts spill 4-6 registers, whose scratch reloads are compiler waits inside
the loop -- known, measured with them: cu100k 34.5 -> 30.7 ms).

    python scripts/check_ring_waits.py [csrc/build]      # the objects of the last build; exit code 1 on a violation
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(obj):
    with tempfile.TemporaryDirectory() as td:
        subprocess.run(["cp", obj, os.path.join(td, "x.o")], check=True)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "x.o"], cwd=td, capture_output=True)
        out = ""
        for co in [f for f in os.listdir(td) if "amdgcn" in f]:
            out += subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], cwd=td, capture_output=True, text=True).stdout
        return out


def kernels(text):
    cur, body = None, []
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            if cur:
                yield cur, body
            cur, body = m.group(1), []
        elif cur and line.startswith("\t"):
            body.append(line.strip().split("//")[0].strip())
    if cur:
        yield cur, body


def wait_table(src):
    """The constants of the generated `spec_wait_vm<GX ? (ATOM ? a : b) : c>` lines: per (GX, ATOM) the allowed counts."""
    tab = {(True, True): {0}, (True, False): {0}, (False, False): {0}}
    for m in re.finditer(r"spec_wait_vm<GX \? \(ATOM \? (\d+) : (\d+)\) : (\d+)>", open(src).read()):
        a, b, c = (int(x) for x in m.groups())
        tab[(True, True)].add(a)
        tab[(True, False)].add(b)
        tab[(False, False)].add(c)
    return tab


def check(name, ins, allowed):
    problems = []
    waits = [(i, int(m.group(1))) for i, s in enumerate(ins) for m in [re.match(r"s_waitcnt vmcnt\((\d+)\)", s)] if m]
    # the loop starts behind the prologue: owner-row loads, the first copies, then hipcc's own waits for the owner rows
    # (forced to completion before the loop, the last of them a vmcnt(0))
    first_copy = min(i for i, s in enumerate(ins) if s.startswith("global_load_lds"))
    drained = min([i for i, n in waits if n == 0 and i > first_copy], default=first_copy)
    mine = [i for i, n in waits if n in allowed and n > 0 and i > drained]
    if not mine:
        return ["no counted wait of the generated table found"]
    steady = max(allowed)
    start = min(mine)
    last = max(i for i, s in enumerate(ins) if s.startswith("global_load_lds"))
    body = ins[start:last + 1]
    counted = sum(1 for s in body if s.startswith(("global_store", "global_atomic", "global_load_lds")))
    if counted < steady:
        problems.append(f"loop body holds {counted} counted operations, the steady wait assumes {steady}")
    reload_ = [i for i, s in enumerate(body) if s.startswith("global_load_dword")]
    for i, s in enumerate(body):
        m = re.match(r"s_waitcnt vmcnt\((\d+)\)", s)
        if m and int(m.group(1)) not in allowed and not any(0 < i - r <= 12 for r in reload_):
            problems.append(f"wait `{s}` inside the loop is not in the generated table {sorted(allowed)}")
    for s in ins:
        if re.search(r"\bm0\b", s) and not s.startswith("s_mov_b32 m0,"):
            problems.append(f"M0 touched outside the copy statements: `{s}`")
    return problems


def main():
    build = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "nequip_amd", "csrc", "build")
    spec = os.path.join(ROOT, "nequip_amd", "csrc", "generated_spec")
    n, bad = 0, 0
    for src in sorted(os.listdir(spec)):
        obj = os.path.join(build, src.replace(".hip", ".o"))
        if not src.endswith(".hip") or not os.path.exists(obj) or "bwd_pair_ring_kernel" not in open(os.path.join(spec, src)).read():
            continue
        tab = wait_table(os.path.join(spec, src))
        for name, ins in kernels(disassemble(obj)):
            m = re.search(r"bwd_pair_ring_kernelILi(\d)ELb(\d)ELb(\d)E", name)
            if not m:
                continue
            n += 1
            for p in check(name, ins, tab[(m.group(2) == "1", m.group(3) == "1")]):
                bad += 1
                print(f"{src} <{m.group(1)}, GX={m.group(2)}, ATOM={m.group(3)}>: {p}")
    print(f"{n} ring kernel instantiations checked, {bad} problem(s)")
    return 1 if bad or not n else 0


if __name__ == "__main__":
    sys.exit(main())
