"""Dump the pair-centric topology of a bench workload for scripts/micro/pair_lab.hip (run on the GPU box).

    python scripts/micro/dump_topo.py water /tmp/topo_water.bin      # cfg-3: 10 125 atoms
    python scripts/micro/dump_topo.py cu20k /tmp/topo_cu20k.bin      # cfg-5 model on a fifth of the box

File: int32 header [N, E, P, 0], then int32 arrays owner_rowptr[N+1], pair_other[P], pair_row[P], edge_in[P], edge_out[P],
other_rowptr[N+1], other_slot[P], then float64 pos[N, 3], float64 cell[3, 3] -- what ``EdgePairing.owner_csr`` hands to
``nqa_tp_scatter_bwd_pairs`` (the lists are built by the package's own device kernels)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from nequip_amd.nn._topology import EdgeTopology
from nequip_amd.utils import synthetic as syn

which, out = sys.argv[1], sys.argv[2]
dev = torch.device("cuda:0")
if which == "water":
    pos, types, cell, _ = syn.water_box(15, seed=0)
    r_max = 4.5
elif which == "cu20k":
    pos, types, cell, _ = syn.copper_box((25, 25, 8))
    r_max = 4.5
elif which == "cu100k":
    pos, types, cell, _ = syn.copper_box((25, 25, 40))
    r_max = 4.5
else:
    raise SystemExit("water | cu20k | cu100k")
data = syn.make_data(pos, types, r_max, cell)
ei = data["edge_index"].to(dev)
shifts = data.get("edge_cell_shift")
N, E = len(pos), ei.shape[1]
topo = EdgeTopology(ei[0], ei[1], N)
pairing = topo.pairing(shifts.to(dev) if shifts is not None else None)
assert pairing is not None, "list does not pair up"
lists = [t.cpu().numpy().astype(np.int32) for t in pairing.owner_csr]
P = pairing.num_pairs
with open(out, "wb") as f:
    np.array([N, E, P, 0], dtype=np.int32).tofile(f)
    for a in lists:
        a.tofile(f)
    np.asarray(pos, dtype=np.float64).tofile(f)
    np.asarray(cell, dtype=np.float64).reshape(3, 3).tofile(f)
print(f"{which}: N={N} E={E} P={P} -> {out}", [a.shape for a in lists])
