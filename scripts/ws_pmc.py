"""A few launches of the level-1 message MLP (600k rows, first layer hoisted, fused aggregation) on one kernel, for rocprofv3 --pmc
passes (scripts/pmc_ws.sh).  Usage: python scripts/ws_pmc.py ws|bx6i|tile [--node]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import _lib, ops, plan
from graphs4cfd_amd.nn import blocks as B
kernel = sys.argv[1] if len(sys.argv) > 1 else "ws"
torch.set_grad_enabled(False)
lib = _lib.load()
lib.g4c_mlp_ws_enable(2 if kernel == "ws" else 0); lib.g4c_mlp_bx6i_enable(2 if kernel == "bx6i" else 0)
dev = torch.device("cuda", 0); H = 128; rows = 600000; n = rows // 6
torch.manual_seed(0)
blk = B.GNBlock((3 * H, (H, H, H), True), (2 * H, (H, H, H), True)).to(dev)
e, pr, pc = torch.randn(rows, H, device=dev), torch.randn(n, H, device=dev), torch.randn(n, H, device=dev)
colh = torch.arange(n).repeat_interleave(6)
ei = torch.stack([torch.randint(0, n, (rows,)), colh]).to(dev)
ep, csr = plan.edge_csr(ei, n)
pk = blk.edge_mlp._packed_cols("hoist", 0, H, [H], [False], False)
src = [ops.Source(e, pre_act=_lib.ACT_SELU), ops.Source(pr, index=ep.row, additive=True), ops.Source(pc, index=ep.col, additive=True)]
out, agg = torch.empty(rows, H, device=dev), torch.empty(n, H, device=dev)
if "--fused" in sys.argv:        # one launch per MP layer (g4c_mp_layer_forward_bx6) at config 2's level-1 size: 10k nodes, 60k edges, heads
    rows = 60000; n = rows // 6
    e, v = torch.randn(rows, H, device=dev), torch.randn(n, H, device=dev)
    ei = torch.stack([torch.randint(0, n, (rows,)), torch.arange(n).repeat_interleave(6)]).to(dev)
    for _ in range(6):
        out = blk.step(v, e, ei, _lib.ACT_SELU, e_pre_act=_lib.ACT_SELU, next_msg=blk.edge_mlp)
    assert int(lib.g4c_mlp_last_kernel()) == 4
elif "--node" in sys.argv:          # the level-1 node launch: [aggregate | v] -> MLP -> LayerNorm -> SELU, + the next layer's two first-layer products (heads)
    v = torch.randn(n, H, device=dev)
    res = None
    for _ in range(6):
        res = blk.node_mlp.run_with_heads([ops.Source(agg), ops.Source(v)], n, _lib.ACT_SELU, blk.edge_mlp, H, [H, H])
    assert res is not None
else:
    for _ in range(6): ops.mlp_forward(pk, src, rows, 0, out=out, agg=(csr, agg, True))
torch.cuda.synchronize()
