"""Experiment (round 6): the level-1 node update on the weight-stationary kernel — one weighted block (the aggregate), the v-side
first-layer product as ONE additive block (direct rows), three layers, LayerNorm, SELU — against today's tile-kernel launch with heads;
what a heads-only launch may cost is the difference to beat."""
import os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import _lib, ops
from graphs4cfd_amd.nn import blocks as B
torch.set_grad_enabled(False)
lib = _lib.load(); dev = torch.device("cuda", 0); H = 128
ops.set_mlp_precision("f16x3")
torch.manual_seed(0)
node = B.MLP(2 * H, (H, H, H), True).to(dev); nxt = B.MLP(3 * H, (H, H, H), True).to(dev)


def timeit(fn, reps=8, inner=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for r in range(reps):
        s_, t_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        for _ in range(inner): fn()
        t_.record(); torch.cuda.synchronize()
        ts.append(s_.elapsed_time(t_) / inner * 1e3)
    return statistics.median(ts), min(ts)


for rows in (100000, 25000, 12500):
    agg, v = torch.randn(rows, H, device=dev), torch.randn(rows, H, device=dev)
    W0 = node._linears()[0].weight
    pn = (v @ W0[:, H:].T).contiguous()
    pk = node._packed_cols("hoist_node", 0, H, [H], [False], False)
    out = torch.empty(rows, H, device=dev)
    ref = node.run_coded([ops.Source(agg), ops.Source(v)], rows, _lib.ACT_SELU)
    lib.g4c_mlp_ws_enable(2)
    y = ops.mlp_forward(pk, [ops.Source(agg), ops.Source(pn, additive=True)], rows, _lib.ACT_SELU, out=out)
    k = int(lib.g4c_mlp_last_kernel())
    err = (y - ref).abs().max().item()
    t_ws = timeit(lambda: ops.mlp_forward(pk, [ops.Source(agg), ops.Source(pn, additive=True)], rows, _lib.ACT_SELU, out=out))
    lib.g4c_mlp_ws_enable(1)
    t_plain = timeit(lambda: node.run_coded([ops.Source(agg), ops.Source(v)], rows, _lib.ACT_SELU))
    t_heads = timeit(lambda: node.run_with_heads([ops.Source(agg), ops.Source(v)], rows, _lib.ACT_SELU, nxt, H, [H, H]))
    print(f"rows {rows:6d}: node MLP on ws (kernel {k}) {t_ws[0]:6.1f} us (min {t_ws[1]:6.1f}), max|ws - tile| {err:.1e};  tile kernel without heads {t_plain[0]:6.1f} us, "
          f"with 2 heads {t_heads[0]:6.1f} us")
