"""Message launch (first layer hoisted, fused aggregation) over the row count: tile kernel (ring of two), tile kernel with the
small-launch deep ring, weight-stationary kernel — 16 launches over 4 different MLPs captured in a hipGraph (no host launch cost),
median of 9 replays.  Decides ws_launch's minimum row count and g4c_mlp_small_launch_tiles' limit.
Usage: [LAYERS=2] python scripts/sweep_message_launch.py [rows ...]"""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import _lib, ops, plan
from graphs4cfd_amd.nn import blocks as B
torch.set_grad_enabled(False)
lib = _lib.load(); dev = torch.device("cuda", 0); H = 128
LAYERS = int(os.environ.get("LAYERS", "3"))
sizes = [int(a) for a in sys.argv[1:]] or [6000, 12000, 16000, 20000, 26000, 33000, 50000, 75000, 120000, 200000]
blks = [B.GNBlock((3 * H, (H,) * LAYERS, True), (2 * H, (H, H, H), True)).to(dev) for _ in range(4)]
for rows in sizes:
    n = rows // 6; rows = n * 6
    e, pr, pc = torch.randn(rows, H, device=dev), torch.randn(n, H, device=dev), torch.randn(n, H, device=dev)
    colh = torch.arange(n).repeat_interleave(6)
    ei = torch.stack([torch.randint(0, n, (rows,)), colh]).to(dev)
    ep, csr = plan.edge_csr(ei, n)
    out, agg = torch.empty(rows, H, device=dev), torch.empty(n, H, device=dev)
    pks = [b.edge_mlp._packed_cols("hoist", 0, H, [H], [False], False) for b in blks]
    src = [ops.Source(e), ops.Source(pr, index=ep.row, additive=True), ops.Source(pc, index=ep.col, additive=True)]
    res = {}
    for name, ws, lim in (("tile ring2", 0, 0), ("tile deep", 0, 1 << 30), ("ws", 2, 0)):
        lib.g4c_mlp_ws_enable(ws); lib.g4c_mlp_bx6i_enable(0); lib.g4c_mlp_small_launch_tiles(lim)
        def body():
            for _ in range(4):
                for pk in pks: ops.mlp_forward(pk, src, rows, 0, out=out, agg=(csr, agg, True))
        body(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g): body()
        g.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(9):
            s_, t_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record(); g.replay(); t_.record(); torch.cuda.synchronize()
            ts.append(s_.elapsed_time(t_) * 1e3 / 16)
        res[name] = statistics.median(ts)
    print(f"{rows:7d} rows ({(rows + 31) // 32:5d} tiles): " + "   ".join(f"{k} {v:7.1f} us" for k, v in res.items()))
lib.g4c_mlp_ws_enable(1); lib.g4c_mlp_small_launch_tiles(512)
