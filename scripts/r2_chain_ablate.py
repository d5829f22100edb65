"""Timing of the fused node-chain launches of the cfg-3 model alone (forward and backward of both layer boundaries)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nequip_amd.nn import ConvNetLayer
from nequip_amd.o3._node_chain import NodeStage
dev = torch.device("cuda:0")
w = bench.WORKLOADS["water10k"]
cfg = bench.model_cfg(w, 39.6)
model = bench.build_model(cfg, ["H", "O"], dev)
layers = [m for m in model.modules() if isinstance(m, ConvNetLayer)]
N = int(os.environ.get("N", 10125))
types = torch.randint(0, 2, (N,), device=dev)
table = [m for m in model.modules() if type(m).__name__ == "NodeTypeEmbed"][0].embed_module.weight.detach()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
for L, (cur, nxt) in enumerate(zip(layers, layers[1:])):
    st = NodeStage(cur.conv.linear_2, cur.equivariant_nonlin, nxt.conv.linear_1, nxt.conv.sc, 0.16)
    a = torch.randn(N, st.dim_a, device=dev); add = torch.randn(N, st.dim_h, device=dev) if cur.conv.sc is not None else None
    h, y, s = st.forward(a, add, types, table)
    gy = torch.randn_like(y); gs = torch.randn_like(s)
    tf = timeit(lambda: st.forward(a, add, types, table))
    tb = timeit(lambda: st.backward(gy, gs, h, types, table, add is not None))
    print(f"boundary {L}: fwd {tf:.0f} us  bwd {tb:.0f} us  (dbg={os.environ.get('NQA_CHAIN_DBG','0')} G={os.environ.get('NQA_CHAIN_G','auto')})", flush=True)
