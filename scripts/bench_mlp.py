"""Micro-benchmark of the fused radial MLP kernels alone (cfg-3 sizes): TFLOP/s per layer shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nequip_amd.nn.mlp import ScalarMLPFunction
dev = torch.device("cuda:0")
E = int(os.environ.get("E", 400558))
SHAPES = [int(x) for x in os.environ.get("SHAPES", "704,192").split(",")]
for H, W in [(128, w) for w in SHAPES]:
    mlp = ScalarMLPFunction(8, W, 1, H).to(dev).eval()
    emb = (torch.randn(E, 8, device=dev) * 0.5).requires_grad_(True)
    g = torch.randn(E, W, device=dev)
    FWD_ONLY = os.environ.get("FWD_ONLY", "") == "1"  # (ablation runs: the backward kernels read the same NQA_MLP_DBG bits)
    out = mlp(emb)
    if not FWD_ONLY:
        torch.autograd.grad(out, emb, g)
    torch.cuda.synchronize()
    def timeit(fn, n=20):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        fn(); torch.cuda.synchronize(); s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
    with torch.no_grad():
        tf = timeit(lambda: mlp(emb))
    if FWD_ONLY:
        print(f"H={H} W={W}: fwd {tf*1e3:.0f} us", flush=True)
        continue
    out = mlp(emb)
    tb = timeit(lambda: torch.autograd.grad(out, emb, g, retain_graph=True))
    fl = 2.0 * E * H * W
    # rocBLAS reference for the same GEMM shapes
    h = torch.randn(E, H, device=dev); w1 = torch.randn(H, W, device=dev)
    tr = timeit(lambda: torch.mm(h, w1)); tr2 = timeit(lambda: torch.mm(g, w1.t()))
    print(f"H={H} W={W}: fwd {tf*1e3:.0f} us {fl/tf/1e9:.1f} TF | bwd {tb*1e3:.0f} us {fl/tb/1e9:.1f} TF | rocBLAS mm fwd {tr*1e3:.0f} us {fl/tr/1e9:.1f} TF, bwd {tr2*1e3:.0f} us {fl/tr2/1e9:.1f} TF", flush=True)
