"""GPU check of the persistent ping-pong MLP kernel (mlp_px6.hip) against the 32-row-tile kernel on identical inputs:
every launch shape the models use (hoisted edge MLP, node MLP with heads, narrow / gathered sources, single layer, decoder
with residual, fused aggregation), odd row counts, then A/B timing.  Usage: python scripts/px6_check.py [--time]"""
import argparse, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import _lib, ops, plan
from graphs4cfd_amd.nn import blocks as B

ap = argparse.ArgumentParser(); ap.add_argument("--time", action="store_true"); ap.add_argument("--rows", type=int, default=600000)
ap.add_argument("--rounds", type=int, default=10); ap.add_argument("--precision", default="bf16x6")
a = ap.parse_args()
ops.set_mlp_precision(a.precision)
lib = _lib.load()
dev = torch.device("cuda", 0); H = 128
torch.manual_seed(0)
torch.set_grad_enabled(False)
fails = []


def both(fn):
    lib.g4c_mlp_px6_enable(0); ref = fn(); torch.cuda.synchronize()
    lib.g4c_mlp_px6_enable(1); new = fn(); torch.cuda.synchronize()
    return ref, new


def cmp(name, ref, new, tol):
    ref = ref if isinstance(ref, (list, tuple)) else [ref]; new = new if isinstance(new, (list, tuple)) else [new]
    worst = 0.0
    for r, n in zip(ref, new):
        if not torch.isfinite(n).all():
            worst = float("inf"); break
        worst = max(worst, (r - n).abs().max().item())
    ok = worst <= tol
    print(f"{'ok  ' if ok else 'FAIL'} {name:58s} max|px6 - tile| = {worst:.2e} (tol {tol:g})")
    if not ok: fails.append(name)


blk = B.GNBlock((3 * H, (H, H, H), True), (2 * H, (H, H, H), True)).to(dev)
blk2 = B.GNBlock((3 * H, (H, H, H), True), (2 * H, (H, H, H), True)).to(dev)
for rows in (600000, 100000, 6000, 1000, 65, 64, 33, 32, 7):
    n = max(rows // 6, 1)
    e = torch.randn(rows, H, device=dev); v = torch.randn(n, H, device=dev); agg = torch.randn(n, H, device=dev)
    row = torch.randint(0, n, (rows,), device=dev, dtype=torch.int32)
    col = (torch.arange(rows, device=dev) // 6).clamp(max=n - 1).to(torch.int32)
    pr, pc = torch.randn(n, H, device=dev), torch.randn(n, H, device=dev)
    pk_e = blk.edge_mlp._packed_cols("hoist", 0, H, [H], [False], False)
    src_e = [ops.Source(e, pre_act=_lib.ACT_SELU), ops.Source(pr, index=row, additive=True), ops.Source(pc, index=col, additive=True)]
    cmp(f"edge hoisted rows={rows}", *both(lambda: ops.mlp_forward(pk_e, src_e, rows, 0)), 2e-5)
    pk_v = blk.node_mlp.packed([H, H], [False, False])
    src_v = [ops.Source(agg), ops.Source(v)]
    cmp(f"node rows={n}", *both(lambda: ops.mlp_forward(pk_v, src_v, n, _lib.ACT_SELU)), 2e-5)

    def heads():
        r = blk.node_mlp.run_with_heads(src_v, n, _lib.ACT_SELU, blk2.edge_mlp, H, [H, H])
        return [r[0]] + list(r[1])
    if a.precision != "bf16": cmp(f"node + 2 heads rows={n}", *both(heads), 2e-5)
    # plain 3-block edge MLP through gather indices (small launches are not hoisted)
    pk_e3 = blk.edge_mlp.packed([H, H, H], [False] * 3)
    src_e3 = [ops.Source(e), ops.Source(v, index=row), ops.Source(v, index=col)]
    cmp(f"edge 3 blocks (gathered) rows={rows}", *both(lambda: ops.mlp_forward(pk_e3, src_e3, rows, 0)), 2e-5)

# narrow sources + gathered source + negate (UpMP), DownMP, encoders, decoder with residual, single layer
for n in (100000, 999):
    nl = max(n // 4, 1)
    rel = torch.randn(n, 2, device=dev); f_l = torch.randn(nl, H, device=dev); f_h = torch.randn(n, H, device=dev)
    parent = torch.randint(0, nl, (n,), device=dev, dtype=torch.int32)
    up = B.MLP(2 + 2 * H, (H, H, H), True).to(dev)
    cmp(f"UpMP [-rel | f_l[parent] | f_h] rows={n}", *both(lambda: up.run([ops.Source(rel, negate=True), ops.Source(f_l, parent), ops.Source(f_h)], n, activation=torch.tanh)), 2e-5)
    down = B.MLP(2 + H, (H, H, H), True).to(dev)
    cmp(f"DownMP [rel | field] rows={n}", *both(lambda: down.run([ops.Source(rel), ops.Source(f_h)], n)), 2e-5)
    dec = B.MLP(H, (H, H, 3), False).to(dev)
    field = torch.randn(n, 6, device=dev)
    cmp(f"decoder + residual rows={n}", *both(lambda: dec.run([ops.Source(f_h)], n, resid=field, resid_col0=3)), 2e-5)
    pk1 = blk.edge_mlp._packed_cols("hoist1", H, 2 * H, [H], [False], True)
    cmp(f"single layer product rows={n}", *both(lambda: ops.mlp_forward(pk1, [ops.Source(f_h)], n)), 2e-5)
    w64 = torch.randn(n, 64, device=dev)
    m64 = B.MLP(64 + H, (H, H, H), True).to(dev)
    cmp(f"[64-wide | 128] rows={n}", *both(lambda: m64.run([ops.Source(w64), ops.Source(f_h)], n)), 2e-5)

# fused aggregation: e' and agg vs the separate reduction (bit-exact), uniform and ragged degrees
for n, ragged in ((100000, False), (5000, True), (37, True)):
    if ragged:
        deg = torch.randint(0, 9, (n,)); deg[0] = 0
    else:
        deg = torch.full((n,), 6)
    colh = torch.repeat_interleave(torch.arange(n), deg)
    rows = int(colh.numel())
    csr = plan.segments_of_sorted(colh.to(dev), n)
    e = torch.randn(rows, H, device=dev)
    row = torch.randint(0, n, (rows,), device=dev, dtype=torch.int32); col = colh.to(dev).to(torch.int32)
    pr, pc = torch.randn(n, H, device=dev), torch.randn(n, H, device=dev)
    pk_e = blk.edge_mlp._packed_cols("hoist", 0, H, [H], [False], False)
    src_e = [ops.Source(e, pre_act=_lib.ACT_SELU), ops.Source(pr, index=row, additive=True), ops.Source(pc, index=col, additive=True)]
    old_fuse = ops.FUSE_AGG
    for mean in (True, False):
        ops.FUSE_AGG = True
        lib.g4c_mlp_px6_enable(1)
        ag = torch.full((n, H), float("nan"), device=dev)
        y = ops.mlp_forward(pk_e, src_e, rows, 0, agg=(csr, ag, mean))
        torch.cuda.synchronize()
        ops.FUSE_AGG = False
        lib.g4c_mlp_px6_enable(0)
        y0 = ops.mlp_forward(pk_e, src_e, rows, 0)
        ag_sep = ops.segment_reduce(y, csr, mean)          # reduction of px6's own rows: must be bit-identical
        torch.cuda.synchronize()
        cmp(f"fused agg rows={rows} ragged={ragged} mean={mean}: e'", y0, y, 2e-5)
        cmp(f"fused agg rows={rows} ragged={ragged} mean={mean}: agg (bit-exact)", ag_sep, ag, 0.0)
    ops.FUSE_AGG = old_fuse
lib.g4c_mlp_px6_enable(1)
print("FAILED: " + ", ".join(fails) if fails else "all px6 checks passed")

if a.time:
    rows = a.rows; n = rows // 6
    e = torch.randn(rows, H, device=dev); v = torch.randn(n, H, device=dev); agg = torch.randn(n, H, device=dev)
    row = torch.randint(0, n, (rows,), device=dev, dtype=torch.int32)
    col = (torch.arange(rows, device=dev) // 6).clamp(max=n - 1).to(torch.int32)
    pr, pc = torch.randn(n, H, device=dev), torch.randn(n, H, device=dev)
    out_e, out_v = torch.empty(rows, H, device=dev), torch.empty(n, H, device=dev)
    ho = [torch.empty(n, H, device=dev), torch.empty(n, H, device=dev)]
    pk_e = blk.edge_mlp._packed_cols("hoist", 0, H, [H], [False], False)
    src_e = [ops.Source(e, pre_act=_lib.ACT_SELU), ops.Source(pr, index=row, additive=True), ops.Source(pc, index=col, additive=True)]
    src_v = [ops.Source(agg), ops.Source(v)]
    csr = plan.segments_of_sorted(col.long(), n)
    ag = torch.empty(n, H, device=dev)

    def agg_case():
        ops.FUSE_AGG = True
        ops.mlp_forward(pk_e, src_e, rows, 0, out=out_e, agg=(csr, ag, True))
        ops.FUSE_AGG = False
    cases = {"edge(hoisted)": lambda: ops.mlp_forward(pk_e, src_e, rows, 0, out=out_e),
             "edge(hoisted)+agg": agg_case,
             "node+heads": lambda: blk.node_mlp.run_with_heads(src_v, n, _lib.ACT_SELU, blk2.edge_mlp, H, [H, H], out=out_v, head_outs=ho)}
    for cname, fn in cases.items():
        times = {0: [], 1: []}
        for on in (0, 1):
            lib.g4c_mlp_px6_enable(on); fn(); fn(); torch.cuda.synchronize()
        for r in range(a.rounds):
            for on in (0, 1):
                lib.g4c_mlp_px6_enable(on)
                s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); [fn() for _ in range(3)]; t.record(); torch.cuda.synchronize()
                times[on].append(s.elapsed_time(t) / 3 * 1e3)
        print(f"{cname:20s} tile kernel median {statistics.median(times[0]):8.1f} us (min {min(times[0]):8.1f})   px6 median {statistics.median(times[1]):8.1f} us (min {min(times[1]):8.1f})")
    lib.g4c_mlp_px6_enable(1)
sys.exit(1 if fails else 0)
