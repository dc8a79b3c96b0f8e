import os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import _lib, ops, plan
from graphs4cfd_amd.nn import blocks as B
torch.set_grad_enabled(False)
lib = _lib.load(); dev = torch.device("cuda", 0); H = 128
ops.set_mlp_precision("bf16")
def rs(t):          # the same bf16 rows in the row-split kernel's column order
    return ops.RsOrderedRows.tag(t[:, ops._rs_k_order(t.device)].contiguous())
def bench(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for r in range(8):
        s_, t_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        for _ in range(reps): fn()
        t_.record(); torch.cuda.synchronize()
        ts.append(s_.elapsed_time(t_) / reps * 1e3)
    return statistics.median(ts)
n, K, layers = 500000, 5, 2
E = K * n
blk = B.GNBlock((3 * H, (H,) * layers, True), (2 * H, (H,) * layers, True)).to(dev)
e32, pr, pc = torch.randn(E, H, device=dev), torch.randn(n, H, device=dev), torch.randn(n, H, device=dev)
e16 = torch.nn.functional.selu(e32).to(torch.bfloat16); pr16, pc16 = pr.to(torch.bfloat16), pc.to(torch.bfloat16)
pk = blk.edge_mlp._packed_cols("hoist", 0, H, [H], [False], False)
pk_rs = blk.edge_mlp._packed_cols("hoist_rs", 0, H, [H], [False], False, rs_order=True)
tgt = torch.arange(n).repeat_interleave(K)
for name, snd in (("random senders", torch.randint(0, n, (E,))), ("senders within +-64 rows", (tgt + torch.randint(-64, 65, (E,))).clamp(0, n - 1)),
                  ("senders within +-4096 rows", (tgt + torch.randint(-4096, 4097, (E,))).clamp(0, n - 1)), ("senders = receivers", tgt.clone())):
    ei = torch.stack([snd, tgt]).to(dev)
    ep, csr = plan.edge_csr(ei, n)
    srcs = {"ws": [ops.Source(e16), ops.Source(pr16, index=ep.row, additive=True), ops.Source(pc16, index=ep.col, additive=True)],
            "rs1": [ops.Source(rs(e16)), ops.Source(rs(pr16), index=ep.row, additive=True), ops.Source(rs(pc16), index=ep.col, additive=True)]}
    for tag, pack in (("ws", pk), ("rs1", pk_rs)):
        src = srcs[tag]
        agg = torch.empty((n, H), device=dev)
        t = bench(lambda: ops.mlp_forward(pack, src, E, agg=(csr, agg, True), rows_dtype=torch.bfloat16, rows_act=_lib.ACT_SELU))
        print(f"{name:28s} {tag:4s} kernel {int(lib.g4c_mlp_last_kernel())}: {t:7.1f} us   {1536e6 / t / 1e6:.2f} TB/s algorithmic")
    # no stores of rows (aggregate only)
    for tag, pack in (("ws", pk), ("rs1", pk_rs)):
        src = srcs[tag]
        agg = torch.empty((n, H), device=dev)
        t = bench(lambda: ops.mlp_forward(pack, src, E, agg=(csr, agg, True), store_rows=False))
        print(f"{name:28s} {tag:4s} rows not stored: {t:7.1f} us")
