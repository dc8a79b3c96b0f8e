"""Tuning: hoisted edge-MLP / node-MLP time vs row count for each tile mode (env G4C_MLP_FORCE_MODE is read once per
process, so this script re-executes itself per mode)."""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from graphs4cfd_amd import ops
    from graphs4cfd_amd.nn import blocks as B
    dev = torch.device("cuda", 0); H = 128
    torch.manual_seed(0)
    blk = B.GNBlock((3 * H, (H, H, H), True), (2 * H, (H, H, H), True)).to(dev)
    for rows in [int(x) for x in sys.argv[2].split(",")]:
        n = max(rows // 6, 64)
        v, e = torch.randn(n, H, device=dev), torch.randn(rows, H, device=dev)
        row = torch.randint(0, n, (rows,), device=dev, dtype=torch.int32)
        col = (torch.arange(rows, device=dev) // 6).clamp(max=n - 1).to(torch.int32)
        pr, pc = torch.randn(n, H, device=dev), torch.randn(n, H, device=dev)
        out = torch.empty(rows, H, device=dev)
        pk = blk.edge_mlp._packed_cols("hoist", 0, H, [H], [False], False)
        srcs = [ops.Source(e), ops.Source(pr, index=row, additive=True), ops.Source(pc, index=col, additive=True)]
        f = lambda: ops.mlp_forward(pk, srcs, rows, 0, out=out)
        for _ in range(3): f()
        torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); [f() for _ in range(20)]; b.record(); torch.cuda.synchronize()
        print(f"{rows} {a.elapsed_time(b) / 20 * 1e3:.1f}", flush=True)
else:
    rows = sys.argv[1] if len(sys.argv) > 1 else "8000,16000,32000,50000,65536,75000,100000,131072,160000,200000,300000"
    res = {}
    for mode in ("0", "64", "32", "322", "324"):
        env = dict(os.environ, G4C_MLP_FORCE_MODE=mode)
        out = subprocess.run([sys.executable, __file__, "child", rows], env=env, capture_output=True, text=True).stdout
        res[mode] = {int(l.split()[0]): float(l.split()[1]) for l in out.strip().splitlines() if l[:1].isdigit()}
    print("rows      policy     64-row     32-row   split<2>   split<4>   (us per hoisted edge-MLP launch)")
    for r in [int(x) for x in rows.split(",")]:
        print(f"{r:8d} " + " ".join(f"{res[m].get(r, float('nan')):10.1f}" for m in ("0", "64", "32", "322", "324")))
