"""The fused MP layer (ops.mp_layer_forward / g4c_mp_layer_forward_bx6: message MLP + aggregation + node MLP + heads in one launch)
against the separate launches of the same GNBlock (blocks.FUSE_LAYER = False): v', e', the next layer's products, over sizes from one
tile to 200k edges, constant and ragged in-degrees (empty segments, the last targets without edges), with / without stored e', with /
without heads; then the same step timed both ways.  Usage: python scripts/mp_layer_check.py [--time]"""
import argparse, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import _lib, ops, plan
from graphs4cfd_amd.nn import blocks as B
ap = argparse.ArgumentParser(); ap.add_argument("--time", action="store_true"); ap.add_argument("--no-check", action="store_true")
a = ap.parse_args()
torch.set_grad_enabled(False)
lib = _lib.load()
dev = torch.device("cuda", 0); H = 128
ops.set_mlp_precision("f16x3")
bad = []


def cmp(name, ref, got, tol):
    if ref is None or got is None:
        ok = ref is None and got is None
        print(f"{'ok  ' if ok else 'FAIL'} {name:80s} both absent" if ok else f"FAIL {name}: one absent")
        if not ok: bad.append(name)
        return
    d = (ref - got).abs().max().item() if ref.numel() else 0.0
    ok = d <= tol and bool(torch.isfinite(got).all())
    print(f"{'ok  ' if ok else 'FAIL'} {name:80s} max|fused - separate| = {d:.2e} (tol {tol:g})")
    if not ok: bad.append(name)


def graph(n, deg, ragged, seed):
    g = torch.Generator().manual_seed(seed)
    if ragged:
        d = torch.randint(0, 2 * deg + 1, (n,), generator=g); d[0] = 0; d[-3:] = 0
    else:
        d = torch.full((n,), deg)
    col = torch.arange(n).repeat_interleave(d)
    row = torch.randint(0, n, (int(col.numel()),), generator=g)
    return torch.stack([row, col]).to(dev)


for layers in (() if a.no_check else (3, 2)):
    torch.manual_seed(layers)
    hid = (H,) * layers
    blk = B.GNBlock((3 * H, hid, True), (2 * H, hid, True)).to(dev)
    nxt = B.GNBlock((3 * H, hid, True), (2 * H, hid, True)).to(dev)
    for n, deg, ragged in ((400, 6, False), (1500, 6, True), (12500, 6, False), (12500, 9, True), (33000, 6, False), (37, 6, True), (6, 6, False)):
        ei = graph(n, deg, ragged, n)
        E = int(ei.size(1))
        v, e = torch.randn(n, H, device=dev), torch.randn(E, H, device=dev)
        for heads in (True, False):
            for keep in (True, False):
                res = []
                for fuse in (False, True):
                    B.FUSE_LAYER = fuse
                    was_min = B.FUSE_LAYER_MIN_ROWS; B.FUSE_LAYER_MIN_ROWS = 1
                    out = blk.step(v, e, ei, _lib.ACT_SELU, e_pre_act=_lib.ACT_SELU, next_msg=nxt.edge_mlp if heads else None, keep_e=keep)
                    B.FUSE_LAYER_MIN_ROWS = was_min
                    res.append(out)
                tag = f"[{layers} layers] n={n} E={E} ragged={ragged} heads={heads} keep_e={keep}: "
                cmp(tag + "v'", res[0][0], res[1][0], 2e-5)
                if keep:      # (keep_e=False: the separate form still stores e' at sizes where its message launch cannot reduce its own rows)
                    cmp(tag + "e'", res[0][1], res[1][1], 2e-5)
                if heads:
                    # (the separate form only makes products from HOIST_MIN_ROWS edges on: compare with the products of its v')
                    W1 = nxt.edge_mlp._linears()[0].weight
                    pr = res[0][2] if res[0][2] is not None else [res[0][0] @ W1[:, H:2 * H].T, res[0][0] @ W1[:, 2 * H:].T]
                    assert res[1][2] is not None
                    for j in range(2):
                        cmp(tag + f"product {j}", pr[j], res[1][2][j], 1e-4 if res[0][2] is None else 2e-5)
print("all fused-layer checks passed" if not bad else "FAILED: " + ", ".join(bad))
if a.time:
    torch.manual_seed(0)
    blk = B.GNBlock((3 * H, (H, H, H), True), (2 * H, (H, H, H), True)).to(dev)
    for n in (1500, 10000, 12500, 25000):
        ei = graph(n, 6, False, n)
        E = int(ei.size(1))
        v, e = torch.randn(n, H, device=dev), torch.randn(E, H, device=dev)
        ts = {}
        for fuse in (False, True):
            B.FUSE_LAYER = fuse
            def chain():
                vv, ee, pr = v, e, None
                for _ in range(4):
                    vv, ee, pr = blk.step(vv, ee, ei, _lib.ACT_SELU, e_pre_act=_lib.ACT_SELU, products=pr, next_msg=blk.edge_mlp)
                return vv
            chain(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                chain()
            t = []
            for _ in range(30):
                s_, t_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s_.record(); g.replay(); t_.record(); torch.cuda.synchronize()
                t.append(s_.elapsed_time(t_) * 1e3 / 4)
            ts[fuse] = statistics.median(t)
        print(f"n={n:6d} E={E:7d}: one MP layer (4-layer chain in a hipGraph)  separate launches {ts[False]:7.1f} us   fused {ts[True]:7.1f} us")
