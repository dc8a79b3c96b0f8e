"""ws (weight-stationary persistent kernel, mlp_ws.hip) against the 32-row-tile kernel: the MP layers' message launch (first
layer hoisted: one weighted block + two gathered additive blocks; or without additive blocks), with and without the fused
aggregation, odd sizes, no stored rows, scattered output rows; then same-process A/B timing against the tile kernel and the dual-tile
kernel (bx6i).  Usage: python scripts/ws_check.py [--time] [--rows N]"""
import argparse, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import _lib, ops, plan
from graphs4cfd_amd.nn import blocks as B
ap = argparse.ArgumentParser(); ap.add_argument("--time", action="store_true"); ap.add_argument("--rows", type=int, default=600000); ap.add_argument("--kernel", default="ws", choices=["ws"])
a = ap.parse_args()
torch.set_grad_enabled(False)
lib = _lib.load()
enable = getattr(lib, f"g4c_mlp_{a.kernel}_enable")
lib.g4c_mlp_bx6i_enable(0)
lib.g4c_mlp_ws_enable(0)
dev = torch.device("cuda", 0); H = 128
torch.manual_seed(0)
blk = B.GNBlock((3 * H, (H, H, H), True), (2 * H, (H, H, H), True)).to(dev)
bad = []


def both(fn):
    enable(0); ref = fn()
    enable(2)
    try:
        got = fn()
    finally:
        enable(0)
    return ref, got


def cmp(name, ref, got, tol):
    d = (ref - got).abs().max().item() if ref.numel() else 0.0
    ok = d <= tol and bool(torch.isfinite(got).all())
    print(f"{'ok  ' if ok else 'FAIL'} {name:58s} max|{a.kernel} - tile| = {d:.2e} (tol {tol:g})")
    if not ok: bad.append(name)


for rows in (600000, 100000, 6001, 999, 65, 64, 33, 32, 7, 1):
    n = max(rows // 6, 2)
    e, v = torch.randn(rows, H, device=dev), torch.randn(n, H, device=dev)
    row = torch.randint(0, n, (rows,), device=dev, dtype=torch.int32)
    col = (torch.arange(rows, device=dev) // 6).clamp(max=n - 1).to(torch.int32)
    W1 = blk.edge_mlp.state_dict()["MLP.linear_1.weight"]
    pr, pc = (v @ W1[:, H:2 * H].T).contiguous(), (v @ W1[:, 2 * H:].T).contiguous()
    pk = blk.edge_mlp._packed_cols("hoist", 0, H, [H], [False], False)
    src = [ops.Source(e, pre_act=_lib.ACT_SELU), ops.Source(pr, index=row, additive=True), ops.Source(pc, index=col, additive=True)]
    cmp(f"edge hoisted rows={rows}", *both(lambda: ops.mlp_forward(pk, src, rows)), 2e-5)
    idx = torch.randint(0, rows, (rows,), device=dev, dtype=torch.int32)
    cmp(f"one block through an index, SELU out rows={rows}", *both(lambda: ops.mlp_forward(pk, [ops.Source(e, index=idx)], rows, _lib.ACT_SELU)), 2e-5)
    oidx = torch.randperm(rows, device=dev).to(torch.int32)
    cmp(f"edge hoisted, scattered output rows={rows}", *both(lambda: ops.mlp_forward(pk, src, rows, out=torch.zeros(rows, H, device=dev), out_idx32=oidx)), 2e-5)
for rows, ragged in ((600000, False), (19972, True), (116, True)):
    n = rows // 6
    if ragged:
        deg = torch.randint(0, 10, (n,)); deg[0] = 0; deg[-1] = 0
        colh = torch.arange(n).repeat_interleave(deg)
    else:
        colh = torch.arange(n).repeat_interleave(6)
    E = int(colh.numel())
    ei = torch.stack([torch.randint(0, n, (E,)), colh]).to(dev)
    ep, csr = plan.edge_csr(ei, n)
    e, v = torch.randn(E, H, device=dev), torch.randn(n, H, device=dev)
    W1 = blk.edge_mlp.state_dict()["MLP.linear_1.weight"]
    pr, pc = (v @ W1[:, H:2 * H].T).contiguous(), (v @ W1[:, 2 * H:].T).contiguous()
    pk = blk.edge_mlp._packed_cols("hoist", 0, H, [H], [False], False)
    src = [ops.Source(e, pre_act=_lib.ACT_SELU), ops.Source(pr, index=ep.row, additive=True), ops.Source(pc, index=ep.col, additive=True)]
    for mean in (True, False):
        def run():
            agg = torch.full((n, H), float("nan"), device=dev)
            y = ops.mlp_forward(pk, src, E, agg=(csr, agg, mean))
            return y, agg
        (y0, a0), (y1, a1) = both(run)
        cmp(f"fused agg rows={E} ragged={ragged} mean={mean}: e'", y0, y1, 2e-5)
        cmp(f"fused agg rows={E} ragged={ragged} mean={mean}: agg == reduce(e') bit-exact", ops.segment_reduce(y1, csr, mean), a1, 0.0)
        def run_noe():
            agg = torch.full((n, H), float("nan"), device=dev)
            ops.mlp_forward(pk, src, E, agg=(csr, agg, mean), store_rows=False)
            return agg
        enable(2); a2 = run_noe(); enable(0)
        cmp(f"fused agg rows={E} ragged={ragged} mean={mean}: rows not stored, same aggregate", a1, a2, 0.0)
print(f"all {a.kernel} checks passed" if not bad else "FAILED: " + ", ".join(bad))
if a.time:
    rows = a.rows; n = rows // 6
    e, v = torch.randn(rows, H, device=dev), torch.randn(n, H, device=dev)
    colh = torch.arange(n).repeat_interleave(6)
    ei = torch.stack([torch.randint(0, n, (rows,)), colh]).to(dev)
    ep, csr = plan.edge_csr(ei, n)
    pr, pc = torch.randn(n, H, device=dev), torch.randn(n, H, device=dev)
    pk = blk.edge_mlp._packed_cols("hoist", 0, H, [H], [False], False)
    src = [ops.Source(e, pre_act=_lib.ACT_SELU), ops.Source(pr, index=ep.row, additive=True), ops.Source(pc, index=ep.col, additive=True)]
    out, agg = torch.empty(rows, H, device=dev), torch.empty(n, H, device=dev)
    cases = {"edge(hoisted)": lambda: ops.mlp_forward(pk, src, rows, 0, out=out),
             "edge(hoisted)+agg": lambda: ops.mlp_forward(pk, src, rows, 0, out=out, agg=(csr, agg, True))}
    def setk(k):
        lib.g4c_mlp_ws_enable(2 if k == "ws" else 0); lib.g4c_mlp_bx6i_enable(2 if k == "bx6i" else 0)
    for cname, fn in cases.items():
        times = {"tile": [], "bx6i": [], "ws": []}
        for k in times:
            setk(k); fn(); fn()
        torch.cuda.synchronize()
        for r in range(15):
            for k in times:
                setk(k)
                s_, t_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s_.record(); fn(); fn(); fn(); t_.record(); torch.cuda.synchronize()
                times[k].append(s_.elapsed_time(t_) / 3 * 1e3)
        setk("tile")
        print(f"{cname:20s} " + "   ".join(f"{k} median {statistics.median(v):8.1f} us (min {min(v):8.1f})" for k, v in times.items()))
