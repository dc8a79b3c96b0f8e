import os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import _lib, ops, plan
from graphs4cfd_amd.nn import blocks as B
torch.set_grad_enabled(False)
lib = _lib.load(); dev = torch.device("cuda", 0); H = 128
ops.set_mlp_precision("bf16")
def rs(t):          # the same bf16 rows in the row-split kernel's column order
    return ops.RsOrderedRows.tag(t[:, ops._rs_k_order(t.device)].contiguous())
n, K, layers = 500000, 5, 2
E = K * n
torch.manual_seed(0)
blk = B.GNBlock((3 * H, (H,) * layers, True), (2 * H, (H,) * layers, True)).to(dev)
e16 = torch.nn.functional.selu(torch.randn(E, H, device=dev)).to(torch.bfloat16)
pr16, pc16 = torch.randn(n, H, device=dev).to(torch.bfloat16), torch.randn(n, H, device=dev).to(torch.bfloat16)
tgt = torch.arange(n).repeat_interleave(K)
ei = torch.stack([(tgt + torch.randint(-4096, 4097, (E,))).clamp(0, n - 1), tgt]).to(dev)
ep, csr = plan.edge_csr(ei, n)
src = [ops.Source(rs(e16)), ops.Source(rs(pr16), index=ep.row, additive=True), ops.Source(rs(pc16), index=ep.col, additive=True)]
pk_rs = blk.edge_mlp._packed_cols("hoist_rs", 0, H, [H], [False], False, rs_order=True)
agg = torch.empty((n, H), device=dev, dtype=torch.bfloat16 if os.environ.get('AGG16') == '1' else torch.float32)
fn = lambda: ops.mlp_forward(pk_rs, src, E, agg=(csr, agg, True), rows_dtype=torch.bfloat16, rows_act=_lib.ACT_SELU)
for _ in range(3): fn()
torch.cuda.synchronize()
ts = []
for r in range(8):
    s_, t_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s_.record()
    for _ in range(10): fn()
    t_.record(); torch.cuda.synchronize()
    ts.append(s_.elapsed_time(t_) / 10 * 1e3)
print(f"{os.path.basename(os.environ.get('G4C_LIB_PATH', 'shipped')):32s} rs1 2.5M rows K 5 bf16: median {statistics.median(ts):7.1f} us  min {min(ts):7.1f}")
