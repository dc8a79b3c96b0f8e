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
ap.add_argument("--stress", type=int, default=0, help="repeat the large launches this many times against one tile-kernel result (races show up as rare mismatches)")
a = ap.parse_args()
torch.set_grad_enabled(False)
lib = _lib.load()
enable = getattr(lib, f"g4c_mlp_{a.kernel}_enable")
lib.g4c_mlp_bx6i_enable(0)
lib.g4c_mlp_ws_enable(0)
dev = torch.device("cuda", 0); H = 128
bad = []


def both(fn):
    enable(0); ref = fn()
    enable(2)
    try:
        got = fn()
    finally:
        enable(0)
    return ref, got


def cmp(name, ref, got, tol, mean_tol=None):
    ref, got = ref.float(), got.float()
    d = (ref - got).abs()
    dmax = d.max().item() if ref.numel() else 0.0
    dmean = d.mean().item() if ref.numel() else 0.0
    ok = dmax <= tol and bool(torch.isfinite(got).all()) and (mean_tol is None or dmean <= mean_tol)
    print(f"{'ok  ' if ok else 'FAIL'} {name:70s} max|{a.kernel} - tile| = {dmax:.2e} (tol {tol:g})" + (f" mean {dmean:.1e}" if mean_tol is not None else ""))
    if not ok: bad.append(name)


def check_variant(prec, layers):
    """One arithmetic (f16x3 stream / rounded-bf16 mode) and depth (3 = MuS-GNN's MLPs, 2 = REMuS-GNN's).  In the rounded-bf16 mode the
    two kernels add the products of a row in different orders (16x16x32 against 32x32x16 MFMAs), and a last-bit difference of a hidden
    pre-activation can flip its rounding to bf16: a few elements differ by ~1e-3, the mean difference stays at round-off level."""
    ops.set_mlp_precision(prec)
    torch.manual_seed(0)
    hid = (H,) * layers
    blk = B.GNBlock((3 * H, hid, True), (2 * H, hid, True)).to(dev)
    tol, mtol = (2e-5, None) if prec != "bf16" else (3e-2, 2e-5)
    tag = f"[{prec}, {layers} layers] "
    W1 = blk.edge_mlp.state_dict()["MLP.linear_1.weight"]
    pk = blk.edge_mlp._packed_cols("hoist", 0, H, [H], [False], False)
    for rows in (600000, 100000, 6001, 999, 65, 64, 33, 32, 7, 1):
        n = max(rows // 6, 2)
        e, v = torch.randn(rows, H, device=dev), torch.randn(n, H, device=dev)
        row = torch.randint(0, n, (rows,), device=dev, dtype=torch.int32)
        col = (torch.arange(rows, device=dev) // 6).clamp(max=n - 1).to(torch.int32)
        pr, pc = (v @ W1[:, H:2 * H].T).contiguous(), (v @ W1[:, 2 * H:].T).contiguous()
        src = [ops.Source(e, pre_act=_lib.ACT_SELU), ops.Source(pr, index=row, additive=True), ops.Source(pc, index=col, additive=True)]
        cmp(tag + f"edge hoisted rows={rows}", *both(lambda: ops.mlp_forward(pk, src, rows)), tol, mtol)
        idx = torch.randint(0, rows, (rows,), device=dev, dtype=torch.int32)
        cmp(tag + f"one block through an index, SELU out rows={rows}", *both(lambda: ops.mlp_forward(pk, [ops.Source(e, index=idx)], rows, _lib.ACT_SELU)), tol, mtol)
        srci = [ops.Source(e, index=idx)] + src[1:]
        cmp(tag + f"edge hoisted, weighted block through an index rows={rows}", *both(lambda: ops.mlp_forward(pk, srci, rows)), tol, mtol)
        oidx = torch.randperm(rows, device=dev).to(torch.int32)
        cmp(tag + f"edge hoisted, scattered output rows={rows}", *both(lambda: ops.mlp_forward(pk, src, rows, out=torch.zeros(rows, H, device=dev), out_idx32=oidx)), tol, mtol)
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
        pr, pc = (v @ W1[:, H:2 * H].T).contiguous(), (v @ W1[:, 2 * H:].T).contiguous()
        src = [ops.Source(e, pre_act=_lib.ACT_SELU), ops.Source(pr, index=ep.row, additive=True), ops.Source(pc, index=ep.col, additive=True)]
        for mean in (True, False):
            def run():
                agg = torch.full((n, H), float("nan"), device=dev)
                y = ops.mlp_forward(pk, src, E, agg=(csr, agg, mean))
                return y, agg
            (y0, a0), (y1, a1) = both(run)
            cmp(tag + f"fused agg rows={E} ragged={ragged} mean={mean}: e'", y0, y1, tol, mtol)
            cmp(tag + f"fused agg rows={E} ragged={ragged} mean={mean}: agg == reduce(e') bit-exact", ops.segment_reduce(y1, csr, mean), a1, 0.0)
            def run_noe():
                agg = torch.full((n, H), float("nan"), device=dev)
                ops.mlp_forward(pk, src, E, agg=(csr, agg, mean), store_rows=False)
                return agg
            enable(2); a2 = run_noe(); enable(0)
            cmp(tag + f"fused agg rows={E} ragged={ragged} mean={mean}: rows not stored, same aggregate", a1, a2, 0.0)
        if prec == "bf16":
            # REMuS-GNN's angle launch in this mode: bf16 rows in (stored activated by the previous launch), bf16(SELU(row)) rows out
            e16 = torch.nn.functional.selu(e).to(torch.bfloat16)
            src16 = [ops.Source(e16)] + src[1:]
            for dt, act in ((torch.bfloat16, _lib.ACT_SELU), (torch.bfloat16, _lib.ACT_NONE), (None, _lib.ACT_NONE)):
                def run16():
                    agg = torch.full((n, H), float("nan"), device=dev)
                    y = ops.mlp_forward(pk, src16, E, agg=(csr, agg, True), rows_dtype=dt, rows_act=act)
                    return y, agg
                (y0, a0), (y1, a1) = both(run16)
                assert y1.dtype == (dt or torch.float32)
                cmp(tag + f"bf16 rows in, out {dt} act {act} rows={E} ragged={ragged}: e'", y0, y1, 6e-2 if dt else tol, 1e-4 if dt else mtol)
                cmp(tag + f"bf16 rows in, out {dt} act {act} rows={E} ragged={ragged}: aggregate", a0, a1, tol, mtol)
    ops.set_mlp_precision("f16x3")


VARIANTS = (("f16x3", 3), ("f16x3", 2), ("bf16", 2), ("bf16", 3))
for prec, layers in VARIANTS:
    check_variant(prec, layers)
if a.stress:
    for prec, layers in VARIANTS[:3]:
        ops.set_mlp_precision(prec)
        torch.manual_seed(1)
        hid = (H,) * layers
        blk = B.GNBlock((3 * H, hid, True), (2 * H, hid, True)).to(dev)
        rows = 600000; n = rows // 6
        e, v = torch.randn(rows, H, device=dev), torch.randn(n, H, device=dev)
        W1 = blk.edge_mlp.state_dict()["MLP.linear_1.weight"]
        pr, pc = (v @ W1[:, H:2 * H].T).contiguous(), (v @ W1[:, 2 * H:].T).contiguous()
        pk = blk.edge_mlp._packed_cols("hoist", 0, H, [H], [False], False)
        idx = torch.randint(0, rows, (rows,), device=dev, dtype=torch.int32)
        col = (torch.arange(rows, device=dev) // 6).to(torch.int32)
        ei = torch.stack([torch.randint(0, n, (rows,), device=dev), col.long()])
        ep, csr = plan.edge_csr(ei, n)
        adds = [ops.Source(pr, index=ep.row, additive=True), ops.Source(pc, index=ep.col, additive=True)]
        agg = torch.empty(n, H, device=dev)
        cases = {"indexed + adds, no aggregation": lambda: ops.mlp_forward(pk, [ops.Source(e, index=idx)] + adds, rows, _lib.ACT_SELU),
                 "direct + adds, no aggregation": lambda: ops.mlp_forward(pk, [ops.Source(e, pre_act=_lib.ACT_SELU)] + adds, rows),
                 "direct + adds + aggregation": lambda: ops.mlp_forward(pk, [ops.Source(e, pre_act=_lib.ACT_SELU)] + adds, rows, agg=(csr, agg, True))}
        if prec != "bf16":
            cases["indexed, no adds, no aggregation"] = lambda: ops.mlp_forward(pk, [ops.Source(e, index=idx)], rows, _lib.ACT_SELU)
        for cname, fn in cases.items():
            enable(2); first = fn().clone(); worst = 0.0
            for r in range(a.stress):
                worst = max(worst, (fn() - first).abs().max().item())
            enable(0)
            ok = worst == 0.0
            print(f"{'ok  ' if ok else 'FAIL'} [{prec}, {layers} layers] stress x{a.stress}: {cname:40s} max difference between repeated launches {worst:.2e}")
            if not ok: bad.append(f"stress {prec} {layers} {cname}")
    ops.set_mlp_precision("f16x3")
torch.manual_seed(0)
blk = B.GNBlock((3 * H, (H, H, H), True), (2 * H, (H, H, H), True)).to(dev)
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
    # REMuS-GNN's level-1 angle launch (config 3): rounded-bf16 mode, 2 layers, 5 angles per edge, bf16 rows in and out, fused aggregation
    ops.set_mlp_precision("bf16")
    rows = 2_500_000; n = rows // 5
    blk2 = B.GNBlock((3 * H, (H, H), True), (2 * H, (H, H), True)).to(dev)
    e16 = torch.randn(rows, H, device=dev).to(torch.bfloat16)
    colh = torch.arange(n).repeat_interleave(5)
    # (senders with the locality of a mesh: a neighbour within +-64 edges)
    rowh = (colh + torch.randint(-64, 65, (rows,))).clamp(0, n - 1)
    ei = torch.stack([rowh, colh]).to(dev)
    ep, csr = plan.edge_csr(ei, n)
    pr, pc = torch.randn(n, H, device=dev), torch.randn(n, H, device=dev)
    pk2 = blk2.edge_mlp._packed_cols("hoist", 0, H, [H], [False], False)
    src2 = [ops.Source(e16), ops.Source(pr, index=ep.row, additive=True), ops.Source(pc, index=ep.col, additive=True)]
    agg2 = torch.empty(n, H, device=dev)
    src2f = [ops.Source(e16.float())] + src2[1:]
    cases2 = {"angle(bf16 mode, 2 layers, bf16 rows in / out)+agg": lambda: ops.mlp_forward(pk2, src2, rows, agg=(csr, agg2, True), rows_dtype=torch.bfloat16, rows_act=_lib.ACT_SELU),
              "angle(bf16 mode, 2 layers, fp32 rows in, bf16 out)+agg": lambda: ops.mlp_forward(pk2, src2f, rows, agg=(csr, agg2, True), rows_dtype=torch.bfloat16, rows_act=_lib.ACT_SELU),
              "angle(bf16 mode, 2 layers, fp32 rows in / out)+agg": lambda: ops.mlp_forward(pk2, src2f, rows, agg=(csr, agg2, True)),
              "angle(bf16 mode, 2 layers, bf16 rows in, none out)+agg": lambda: ops.mlp_forward(pk2, src2, rows, agg=(csr, agg2, True), store_rows=False)}
    for cname, fn in cases2.items():
        times = {"tile": [], "ws": []}
        for k in times:
            setk(k); fn(); fn()
        torch.cuda.synchronize()
        for r in range(10):
            for k in times:
                setk(k)
                s_, t_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s_.record(); fn(); fn(); t_.record(); torch.cuda.synchronize()
                times[k].append(s_.elapsed_time(t_) / 2 * 1e3)
        setk("tile")
        print(f"{cname:56s} " + "   ".join(f"{k} median {statistics.median(v):8.1f} us (min {min(v):8.1f})" for k, v in times.items()))
    ops.set_mlp_precision("f16x3")
