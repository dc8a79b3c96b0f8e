"""bench.py — rollout timesteps/s of the graphs4cfd message-passing hot path on MI355X.

    python bench.py [--workload headline|c2|c3|c5-1gpu] [--gpus 1] [--steps 200] [--warmup 5]
    python bench.py --gpus N ...          (N > 1 outside a torch.distributed.run job: starts its own N ranks, same line as below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Workloads (BASELINE.json `configs`):
  headline  NsThreeScaleGNN (published arch, H = 128, 4/2/4/2/4 MP layers), synthetic 100 000-node 2-D mesh (uniform random
            points, kNN k = 6, grid-clustered levels 2 and 3), fp32 in / out.  The metric's configuration; with --gpus N the same
            mesh is node-partitioned over the ranks (config 4: strong scaling, one halo exchange per MP layer over RCCL).
  c2        NsTwoScaleGNN, 10 000 nodes, fp32 (config 2).
  c3        REMuS-GNN (NsRotEquiTreeScaleGNN, 3 scales, k = 5), 100 000 nodes, MLP operands rounded to bf16 (config 3).
  c5-1gpu   NsFourScaleGNN, 1 000 000-node 3-D mesh on ONE GPU (config 5's mesh and model without the partition).
A step = one forward of the whole model + the rollout bookkeeping kernel, replayed from a hipGraph.

One JSON line on stdout (rank 0).  Extra objects:
  roofline         — dominant kernel (fused MLP): algorithmic FLOP / measured duration, from HIP-event pairs around every launch
                     of an eager pass of the same step, on the launch stream.
  roofline_scatter — the CSR segment-reduce ("scatter-sum") against the HBM roofline: level-1 messages in ROTATING buffers (no
                     reuse out of the 256 MiB Infinity Cache) and, for comparison, in one buffer.
  cpu_baseline     — the oracle (pure-torch restatement of the reference CPU path) timed on this host, bounded sample.
  partition_check  — (N > 1) two partitioned steps against a single-rank rollout of the same mesh, per-rank compute time,
                     per-exchange time, halo bytes, whether the step was captured.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md (v_mfma_f32_32x32x2_f32)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 (v_mfma_f32_32x32x16_bf16), same guide
BX6_PRODUCTS = 6                # bf16 partial products the exact-split kernel executes per fp32 multiply-add
F16X3_PRODUCTS = 3              # fp16 partial products of the two-way split ("f16x3")
PEAK_HBM_GBS = 8000.0           # HBM3E spec

WORKLOADS = {
    "headline": {"model": "NsThreeScaleGNN", "nodes": 100_000, "dim": 2, "precision": "f16x3",
                 "metric": "rollout timesteps/s (100k-node 2D mesh)"},
    "c2": {"model": "NsTwoScaleGNN", "nodes": 10_000, "dim": 2, "precision": "f16x3",
           "metric": "rollout timesteps/s (MuS-GNN 2-scale, 10k-node mesh, fp32)"},
    "c3": {"model": "NsRotEquiTreeScaleGNN", "nodes": 100_000, "dim": 2, "precision": "bf16",
           "metric": "rollout timesteps/s (REMuS-GNN 3-scale, 100k-node mesh, bf16 MLP operands)"},
    "c5-1gpu": {"model": "NsFourScaleGNN", "nodes": 1_000_000, "dim": 3, "precision": "f16x3",
                "metric": "rollout timesteps/s (MuS-GNN 4-scale, 1M-node 3D mesh, 1 GPU)"},
}
MUS_LEVELS = {"NsOneScaleGNN": 1, "NsTwoScaleGNN": 2, "NsThreeScaleGNN": 3, "NsFourScaleGNN": 4}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="headline", choices=sorted(WORKLOADS))
    ap.add_argument("--nodes", type=int, default=None, help="override the workload's mesh size (then not a BASELINE configuration)")
    ap.add_argument("--model", default=None, help="override the workload's model class")
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--precision", default=None, choices=["bf16x6", "f16x3", "fp32", "bf16"],
                    help="arithmetic of the fused MLPs: bf16x6 (fp32-accurate split products on the bf16 matrix pipe), fp32 "
                         "(fp32 MFMA kernels) or bf16 (operands rounded to bf16, ~1e-2 deviation: config 3 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strict-range", action="store_true", help="skip the bf16x6 (fp32 exponent range) rollout beside an f16x3 headline")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-side-configs", action="store_true", help="skip the short legs of BASELINE configs 2 and 3 beside the default headline run")
    ap.add_argument("--no-partition-check", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=75.0)
    a = ap.parse_args()
    w = WORKLOADS[a.workload]
    a.custom = a.nodes is not None or a.model is not None or (a.precision is not None and a.precision != w["precision"])
    a.nodes = a.nodes if a.nodes is not None else w["nodes"]
    a.model = a.model or w["model"]
    a.precision = a.precision or w["precision"]
    a.dim = w["dim"]
    a.metric = w["metric"]
    return a


def launcher_needed(gpus, env):
    """`python bench.py --gpus N` (N > 1) outside a torch.distributed.run job: this process is not a rank, it starts them."""
    return gpus > 1 and "WORLD_SIZE" not in env and "RANK" not in env


def launcher_command(gpus, argv, port):
    """The driver's own launch line (one process per GPU of ONE node, rendezvous on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def launch_ranks(gpus, argv):
    """Re-runs this script as `gpus` ranks; rank 0's JSON line goes straight to stdout, the ranks' stderr is passed on (its tail
    again after a failure, beside the return code of the job)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on these hosts (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = launcher_command(gpus, argv, port)
    proc = subprocess.Popen(cmd, env=env, stdout=None, stderr=subprocess.PIPE, text=True, errors="replace")
    tail = []
    for line in proc.stderr:
        sys.stderr.write(line)
        tail.append(line)
        del tail[:-60]
    rc = proc.wait()
    if rc != 0:
        sys.stderr.write(f"bench.py: the {gpus}-rank job ({' '.join(cmd)}) exited with rc {rc}; last lines of the ranks' stderr:\n")
        sys.stderr.write("".join(tail[-30:]))
    return rc


def build_workload(args, gfd, S, dev):
    """(graph on the host, model on the device, number of predicted fields)."""
    remus = args.model == "NsRotEquiTreeScaleGNN"
    if remus:
        graph = S.remus_graph(args.nodes, k=5, seed=0)
        arch = S.remus_arch(args.hidden)
    else:
        graph = S.mus_graph(args.nodes, levels=MUS_LEVELS[args.model], dim=args.dim, seed=0)
        arch = S.mus_arch(args.model, args.hidden, dim=args.dim)
    torch.manual_seed(0)
    model = getattr(gfd.nn, args.model)(arch=arch, device=dev)
    model.eval()
    return graph, model, int(model.num_fields)


def cpu_baseline(args, S, weights, nf, budget_s):
    """Oracle rollout steps on the host cores: bounded sample (>= 1 step; a smaller mesh of the same kind when one step of
    the workload's mesh would not fit the budget)."""
    from oracle import g4c_oracle as O
    # torch CPU ops stop scaling long before the host's core count on this path (measured on the GPU box,
    # 2 x EPYC 9575F = 256 hw threads: 8/16/32/64/128/256 threads -> 1.29/1.10/1.26/2.13/6.11/103 s per
    # 20k-node step, profiles/r01_cpu_thread_sweep.log), so the baseline uses the fastest setting
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    remus = args.model == "NsRotEquiTreeScaleGNN"
    # ~9 s per step at 100k nodes (MuS 3-scale), ~3x that for REMuS: cap the sample mesh so that one step fits the budget
    cap = 100_000 if not remus else 40_000
    n_sample = min(args.nodes, cap)
    if n_sample == args.nodes and not remus:
        graph = S.mus_graph(args.nodes, levels=MUS_LEVELS[args.model], dim=args.dim, seed=0)
    elif remus:
        graph = S.remus_graph(n_sample, k=5, seed=0)
    else:
        graph = S.mus_graph(n_sample, levels=MUS_LEVELS[args.model], dim=args.dim, seed=0)
    g = graph.to_dict()

    def one(g):
        pred = O.remus_forward(g, weights) if remus else O.mus_forward(args.model, g, weights, nf)
        g = dict(g)
        g["field"] = O.shift_and_replace(g["field"], pred, nf)
        return g
    with torch.no_grad():
        tw = time.perf_counter()
        g = one(g)                          # untimed warm-up step (allocator, thread pool, first-touch of the plan tensors)
        warm_s = time.perf_counter() - tw
        t0 = time.perf_counter()
        steps = 0
        while True:
            g = one(g)
            steps += 1
            el = time.perf_counter() - t0
            if steps >= 5 or el + el / steps > budget_s:
                break
    cpu_model = "unknown CPU"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    scale = n_sample / args.nodes          # cost per step is linear in the mesh size (every op is per node / edge / angle)
    sample = (f"{steps} rollout step(s) of " + (f"the same {args.nodes}-node mesh" if n_sample == args.nodes else
              f"a {n_sample}-node mesh of the same kind (value = measured steps/s x {scale:.3g}: the path is linear in the mesh size)")
              + f" and weights, oracle (op-for-op torch restatement of the reference CPU path, per-step pool_edge rebuild), "
              f"{torch.get_num_threads()} torch threads, {el:.1f} s after one untimed warm-up step ({warm_s:.1f} s)"
              + ("" if steps >= 5 else f"; fewer than 5 steps: the next one would have exceeded the {budget_s:.0f} s budget (--cpu-budget-s)"))
    return {"value": steps / el * scale, "unit": "rollout timesteps/s", "cores": cores, "kind": "port", "sample": sample,
            "cpu": cpu_model, "hw_threads": os.cpu_count(), "steps_timed": steps}


def pmc_traffic(workload="headline"):
    """HBM bytes per launch from the committed PMC collection (scripts/collect_pmc_traffic.sh: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same bench, gfx950 FETCH_SIZE correction).
    rocprofv3 cannot run inside the timed process, so the latest committed collection is quoted with its source."""
    import glob
    sfx = "" if workload == "headline" else f"_{workload}"
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_traffic{sfx}.json")))
    if not files:
        return None, None
    k = json.load(open(files[-1]))["kernels"]

    def avg(prefix):
        sel = [v for name, v in k.items() if name.split("<")[0].endswith(prefix) or name.startswith(prefix)]
        n = sum(v["dispatches"] for v in sel)
        return sum(v["hbm_bytes_per_launch"] * v["dispatches"] for v in sel) / n if n else None
    def largest(prefix):
        sel = [v.get("hbm_bytes_largest_launch") for name, v in k.items() if name.startswith(prefix) and v.get("hbm_bytes_largest_launch")]
        return max(sel) if sel else None
    return {"avg": avg, "scatter": largest("segment_reduce_kernel") or avg("segment_reduce_kernel")}, os.path.relpath(files[-1], ROOT)


def reference_flop_per_step(model, g):
    """2*MAC of every nn.Linear at the row count the reference applies it to (163 840 FLOP per edge and
    131 072 per node for an MP layer at H = 128, etc.), for a MuS-GNN V-cycle on this mesh."""
    def mlp_flop(mlp):
        return 2.0 * sum(l.weight.numel() for l in mlp._linears())
    from graphs4cfd_amd import partition
    levels = 1 + sum(1 for n in model._PROGRAM if n.startswith("down_mp"))
    edges = [e.shape[1] for e in partition.coarse_topology(g, levels)]
    nodes = [int(g.pos.size(0))] + [int(getattr(g, f"pos_{l}").size(0)) for l in range(2, levels + 1)]
    total = mlp_flop(model.edge_encoder) * edges[0] + mlp_flop(model.node_encoder) * nodes[0] + mlp_flop(model.node_decoder) * nodes[0]
    lvl = 0
    for name in model._PROGRAM:
        blk = getattr(model, name)
        if name.startswith("down_mp"):
            total += mlp_flop(blk.down_mlp) * nodes[lvl]
            lvl += 1
        elif name.startswith("up_mp"):
            lvl -= 1
            total += mlp_flop(blk.up_mlp) * nodes[lvl]
        else:
            total += mlp_flop(blk.edge_mlp) * edges[lvl] + mlp_flop(blk.node_mlp) * nodes[lvl]
    return total


KERNEL_SOURCES = {"mlp_ws_kernel": ("mlp_ws.hip", "mlp_common.h"), "mlp_bx6_kernel": ("mlp_fused.hip", "mlp_common.h"),
                  "mlp_bx6i_kernel": ("mlp_bx6i.hip", "mlp_common.h"), "mlp_split_kernel": ("mlp_fused.hip", "mlp_common.h"),
                  "mlp_rs1_kernel": ("mlp_rs.hip", "mlp_common.h"), "mlp_rs2_kernel": ("mlp_rs.hip", "mlp_common.h")}


def source_sha16(kernel):
    """First 16 hex digits of sha256 over the kernel's source files (what a committed PMC summary was collected from)."""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES.get(kernel, ()):
        with open(os.path.join(ROOT, "graphs4cfd_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_pipe_util():
    """MfmaUtil / VALUBusy of the shipped kernels in the default arithmetic, IMPORTED from the newest committed rocprofv3 --pmc
    summaries (scripts/pmc_ws.sh via scripts/refresh_artifacts.sh: profiles/r*_pmc_mlp_ws.txt = the level-1 message launch,
    r*_pmc_mlp_bx6_node.txt = the level-1 node launch) — rocprofv3 cannot run inside the timed process.  Every entry says which
    file it came from and whether that file was collected from the kernel sources this run was built from
    (`collected_from_current_kernel_source`; None for summaries older than the stamp).  None when no file is committed."""
    import glob
    import re
    out = {}
    for key, pat, kern in (("level1_message_launch", "r*_pmc_mlp_ws.txt", "mlp_ws_kernel"),
                           ("level1_node_launch", "r*_pmc_mlp_bx6_node.txt", "mlp_bx6_kernel")):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", pat)))
        if not files:
            continue
        txt = open(files[-1]).read()
        ent = {"source": os.path.relpath(files[-1], ROOT), "imported": True}
        m = re.search(r"source_sha16\s*[:=]\s*([0-9a-f]{16})", txt)
        ent["collected_from_current_kernel_source"] = (m.group(1) == source_sha16(kern)) if m else None
        for name in ("MfmaUtil", "VALUBusy"):
            m = re.search(name + r"\s+\d+\s+per-dispatch\s+(\d+)", txt)
            if m:
                ent[name + "_percent"] = int(m.group(1))
        iv, im = re.search(r"SQ_INSTS_VALU\s+\d+\s+per-dispatch\s+(\d+)", txt), re.search(r"SQ_INSTS_MFMA\s+\d+\s+per-dispatch\s+(\d+)", txt)
        if iv and im and int(im.group(1)):
            ent["valu_per_mfma"] = int(iv.group(1)) / int(im.group(1))
        m1, m2 = re.search(r"SQ_LDS_BANK_CONFLICT\s+\d+\s+per-dispatch\s+(\d+)", txt), re.search(r"SQ_LDS_IDX_ACTIVE\s+\d+\s+per-dispatch\s+(\d+)", txt)
        if m1 and m2 and int(m2.group(1)):
            ent["lds_bank_conflict_share_of_lds_cycles"] = int(m1.group(1)) / int(m2.group(1))
        out[key] = ent
    return out or None


def roofline_blocks(args, result, model, graph_cpu, dev, ops, Rollout):
    """Eager instrumented pass of the same step: HIP-event pair around every launch, on the launch stream."""
    eager = Rollout(model, graph_cpu.clone().to(dev), 12, capture=False)
    eager.run(2)
    torch.cuda.synchronize(dev)
    # three passes of three steps, the least disturbed one is kept (a single stalled launch — seen once: 8 ms — would otherwise
    # halve the averages of a 43-launch step)
    kt, best = None, None
    for _ in range(3):
        with ops.KernelTimer() as kt_try:
            eager.run(3)
        torch.cuda.synchronize(dev)
        total = sum(a.elapsed_time(e) for _k, _f, _b, a, e in kt_try.records)
        if best is None or total < best:
            kt, best = kt_try, total
    summ = kt.summary()
    remus = args.model == "NsRotEquiTreeScaleGNN"

    SPLIT_FAMILY = ("mlp_bx6_kernel", "mlp_bx6i_kernel", "mlp_ws_kernel", "mlp_rs1_kernel", "mlp_rs2_kernel")      # kernels on the bf16 / f16 matrix pipe
    WHAT = {"mlp_ws_kernel": "weight-stationary persistent kernel (mlp_ws.hip): message launches of >= 20k rows in the f16x3 stream / rounded-bf16 mode",
            "mlp_bx6_kernel": "split-operand 32-row tile kernel (mlp_fused.hip): node / encoder / pool / unpool / decoder launches, heads, small launches",
            "mlp_bx6i_kernel": "dual-tile kernel (mlp_bx6i.hip): message launches of >= 400k rows in the bf16x6 stream",
            "mlp_rs1_kernel": "row-split persistent kernel (mlp_rs.hip): rounded-bf16 message launches of >= 20k rows over receivers of one in-degree 4..8 "
                              "(REMuS-GNN's angle launches); a wave owns 16 rows through all layers, weights resident in LDS, aggregation a segmented scan",
            "mlp_rs2_kernel": "row-split update kernel (mlp_rs.hip): the update MLP behind such a message launch — [bf16 aggregate | bf16 e] -> two layers -> "
                              "LayerNorm -> SELU -> e' + two product heads, the five weight blocks = all 160 KB of LDS",
            "mlp_split_kernel": "fp32-MFMA kernel (mlp_fused.hip: g4c_mlp_forward)"}

    def price(kind, flops, seconds):
        """roofline pricing of `flops` ALGORITHMIC FLOP (2*K*N per row and layer) done in `seconds` by kernel `kind`: the fp32
        kernel executes exactly those on the fp32 MFMA pipe; a split-operand kernel executes six bf16 (bf16x6) / three f16 (f16x3)
        MFMA products per fp32 multiply-add, so it is priced in executed FLOP against the dense bf16 / f16 peak (algorithmic rate
        alongside); in rounded-bf16 mode (config 3) one product per multiply-add."""
        alg = flops / seconds / 1e12
        if kind in SPLIT_FAMILY:
            prod = {"bf16x6": BX6_PRODUCTS, "f16x3": F16X3_PRODUCTS}.get(args.precision, 1)
            out = {"achieved": prod * alg, "peak": PEAK_BF16_MFMA_TFLOPS, "frac": prod * alg / PEAK_BF16_MFMA_TFLOPS,
                   "mfma_dtype": {6: "bf16 (6 exact partial products per fp32 MAC, fp32 accumulate)",
                                  3: "f16 (two-way fp16 operand split, 3 partial products per fp32 MAC, fp32 accumulate; same dense peak as bf16)",
                                  1: "bf16 (operands rounded to bf16, fp32 accumulate)"}[prod],
                   "algorithmic_tflops": alg}
            if prod > 1:
                out["algorithmic_vs_fp32_mfma_peak"] = alg / PEAK_FP32_MFMA_TFLOPS
            return out
        return {"achieved": alg, "peak": PEAK_FP32_MFMA_TFLOPS, "frac": alg / PEAK_FP32_MFMA_TFLOPS, "mfma_dtype": "f32"}

    def kernel_entry(kind):
        """One kernel family's own launches: count, average duration, both roofline views."""
        m = summ[kind]
        gbps = m["bytes"] / m["seconds"] / 1e9
        return {"what": WHAT.get(kind, kind), "launches_per_step": m["launches"] // 3, "avg_launch_us": 1e6 * m["seconds"] / m["launches"],
                "flop_per_launch": m["flops"] / m["launches"], "ms_per_step": 1e3 * m["seconds"] / 3, **price(kind, m["flops"], m["seconds"]),
                "hbm_view": {"achieved": gbps, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBS,
                             "algorithmic_bytes_per_launch": m["bytes"] / m["launches"]}}

    traffic, traffic_src = pmc_traffic(args.workload)
    if args.custom:
        traffic = None
    mlp_kinds = [k for k in summ if k.startswith("mlp_")]
    dom = max(mlp_kinds, key=lambda k: summ[k]["seconds"])      # the kernel family with the most GPU time, by what actually ran
    big = kernel_entry(dom)
    tot_f = sum(summ[k]["flops"] for k in mlp_kinds)
    tot_t = sum(summ[k]["seconds"] for k in mlp_kinds)
    fallback = sum(summ[k]["launches"] // 3 for k in mlp_kinds if k not in SPLIT_FAMILY)
    pmc = pmc_pipe_util() if (args.workload == "headline" and args.precision == "f16x3" and not args.custom) else None
    result["roofline"] = {
        "bound": "mfma", "kernel": f"{dom} — {WHAT.get(dom, dom)}",
        "achieved": big["achieved"], "peak": big["peak"], "unit": "TFLOP/s", "frac": big["frac"], "mfma_dtype": big["mfma_dtype"],
        "algorithmic_tflops": big.get("algorithmic_tflops", big["achieved"]),
        "traffic": traffic["avg"](dom) if traffic else None, "traffic_source": traffic_src if traffic else None,
        "launches_per_step": big["launches_per_step"], "avg_launch_us": big["avg_launch_us"], "flop_per_launch": big["flop_per_launch"],
        "ms_per_step_in_kernel": big["ms_per_step"],
        "share_of_fused_mlp_time": summ[dom]["seconds"] / tot_t,
        # (precision "bf16x6" / "bf16": MLPs with an input block wider than 128 columns run on the fp32-MFMA kernels)
        "mlp_launches_per_step": {"total": sum(summ[k]["launches"] // 3 for k in mlp_kinds), "fp32_mfma_fallback": fallback
                                  if args.precision != "fp32" else 0},
        "flop_definition": "flop_per_launch = algorithmic FLOP (2*K*N per row and layer) of the launches as executed.  The node-side "
                           "products of every edge MLP's first layer are hoisted to one product per node (exact re-association), "
                           "so a step executes fewer FLOP than the reference formulation's count",
        "all_mlp_kernels": {"algorithmic_tflops": tot_f / tot_t / 1e12, "flop_per_step": tot_f / 3, "ms_per_step": 1e3 * tot_t / 3},
        "other_mlp_kernels": {k: kernel_entry(k) for k in mlp_kinds if k != dom}}
    if "algorithmic_vs_fp32_mfma_peak" in big:
        result["roofline"]["algorithmic_vs_fp32_mfma_peak"] = big["algorithmic_vs_fp32_mfma_peak"]
    if pmc:
        result["roofline"]["mfma_util_pmc_imported"] = pmc
    # the same kernel against the HBM roofline: algorithmic bytes (every input block row read once, every output row written once:
    # 4 * (sum of input widths + output width [+ heads]) per row) / launch time.  The larger fraction names the nearer roof; when
    # the kernel is far from both (< 0.4 of either) and its counters show the vector ALUs busier than the matrix pipe, it is bound
    # by instruction issue, not by a roof: bound = "issue" (achieved / peak / frac stay those of the nearer roof)
    hbm = dict(big["hbm_view"])
    if result["roofline"]["traffic"]:
        hbm["measured_traffic_GBps"] = result["roofline"]["traffic"] / (big["avg_launch_us"] * 1e-6) / 1e9
        hbm["measured_traffic_frac"] = hbm["measured_traffic_GBps"] / PEAK_HBM_GBS
    mfma_frac = result["roofline"]["frac"]
    if hbm["frac"] > mfma_frac:
        mf = {k: result["roofline"][k] for k in ("achieved", "peak", "unit", "frac", "mfma_dtype")}
        result["roofline"].update({"bound": "hbm", "achieved": hbm["achieved"], "peak": hbm["peak"], "unit": "GB/s", "frac": hbm["frac"],
                                   "mfma_view": mf})
    result["roofline"]["hbm_view"] = hbm
    pmc_dom = (pmc or {}).get("level1_message_launch" if dom == "mlp_ws_kernel" else "level1_node_launch" if dom == "mlp_bx6_kernel" else "")
    if max(hbm["frac"], mfma_frac) < 0.4:
        if pmc_dom and pmc_dom.get("VALUBusy_percent", 0) > pmc_dom.get("MfmaUtil_percent", 0):
            # (only with counters of THIS kernel in hand — ADVICE r04: the side workloads have none and keep the nearer roof)
            result["roofline"]["nearest_roof"] = result["roofline"]["bound"]
            result["roofline"]["bound"] = "issue"
            result["roofline"]["bound_note"] = (f"below 0.4 of both roofs; PMC (imported, {pmc_dom['source']}): VALUBusy {pmc_dom.get('VALUBusy_percent')} % > "
                                                f"MfmaUtil {pmc_dom.get('MfmaUtil_percent')} %: vector-instruction issue, not a roof, bounds it")
        else:
            result["roofline"]["bound_note"] = "below 0.4 of both roofs; no counters of this kernel imported: `bound` names the nearer roof"
    if not remus:
        ref_flop = reference_flop_per_step(model, graph_cpu)
        result["roofline"]["all_mlp_kernels"].update({"reference_formulation_flop_per_step": ref_flop,
                                                     "reference_formulation_tflops": ref_flop / (tot_t / 3) / 1e12})

    # ---- scatter-sum: the level-1 aggregation as a standalone g4c_segment_reduce launch (in the rollout it is fused into the
    # edge launch or folded into the node launch's gather, bit-identical); messages in ROTATING buffers so that no launch finds
    # its input in the 256 MiB Infinity Cache from the previous one, and — for comparison — in one buffer
    from graphs4cfd_amd import plan as _plan
    g_dev = eager.graph
    if remus:
        ep1, csr1 = _plan.edge_csr(g_dev.angle_index, int(g_dev.edge_index.size(1)))
    else:
        ep1, csr1 = _plan.edge_csr(g_dev.edge_index, int(g_dev.field.size(0)))
    n_buf = 3 if csr1.n * args.hidden * 4 < (8 << 30) else 2
    bufs = [torch.randn((csr1.n, args.hidden), dtype=torch.float32, device=dev) for _ in range(n_buf)]
    agg1 = torch.empty((csr1.n_seg, args.hidden), dtype=torch.float32, device=dev)

    def time_scatter(rotate):
        for it in range(3):
            ops.segment_reduce(bufs[it % n_buf if rotate else 0], csr1, True, out=agg1)
        with ops.KernelTimer() as kt1:
            for it in range(12):
                ops.segment_reduce(bufs[it % n_buf if rotate else 0], csr1, True, out=agg1)
        torch.cuda.synchronize(dev)
        sel = [(b, a.elapsed_time(e) * 1e-3) for k, f, b, a, e in kt1.records if k == "segment_reduce"]
        nbytes, tsum = sel[0][0], sum(t for _, t in sel)
        return {"bytes": nbytes, "avg_launch_us": 1e6 * tsum / len(sel), "achieved": nbytes * len(sel) / tsum / 1e9,
                "frac": nbytes * len(sel) / tsum / 1e9 / PEAK_HBM_GBS}
    rot, same = time_scatter(True), time_scatter(False)
    s = summ.get("segment_reduce")
    result["roofline_scatter"] = {
        "bound": "hbm", "kernel": "segment_reduce_kernel (g4c_segment_reduce)", "peak": PEAK_HBM_GBS, "unit": "GB/s",
        "achieved": rot["achieved"], "frac": rot["frac"], "algorithmic_bytes_per_launch": rot["bytes"], "avg_launch_us": rot["avg_launch_us"],
        "traffic": traffic["scatter"] if traffic else None,
        "residency": f"{n_buf} message buffers of {rot['bytes'] / 1e6:.0f} MB used in rotation ({n_buf * rot['bytes'] / 2**20:.0f} MiB "
                     "in flight > the 256 MiB Infinity Cache): every launch streams its messages from HBM",
        "same_buffer": {**same, "residency": "one message buffer read again by every launch (partly resident in the Infinity Cache)"},
        "how": "12 standalone launches on level-1 message tensors; in the rollout this aggregation runs inside the edge launch "
               "(whole CSR segments per row tile) or the node launch's gather"}
    if s:
        result["roofline_scatter"]["step_launches"] = {
            "what": "the segment reductions one rollout step still launches (DownMP cluster means, pool_edge, small coarse levels)",
            "achieved": s["bytes"] / s["seconds"] / 1e9, "algorithmic_bytes_per_launch": s["bytes"] / s["launches"],
            "launches_per_step": s["launches"] // 3, "avg_launch_us": 1e6 * s["seconds"] / s["launches"]}
    # one level-1 MP layer = its two largest fused-MLP launches (message MLP incl. the aggregation; update MLP)
    recs = [(k, f, b, a.elapsed_time(e) * 1e-3) for k, f, b, a, e in kt.records if k.startswith("mlp_")]
    fmax = max(f for k, f, b, t in recs)
    top = [t for k, f, b, t in recs if f == fmax]
    import collections as _c
    cnt = _c.Counter(f for k, f, b, t in recs if f != fmax)
    if cnt:
        f_node = max(cnt, key=lambda f: (cnt[f], f))
        node_t = [t for k, f, b, t in recs if f == f_node]
        result["ms_per_mp_layer_level1"] = 1e3 * (sum(top) / len(top) + sum(node_t) / len(node_t))
    top_kind = next(k for k, f, b, t in recs if f == fmax)
    top_bytes = next(b for k, f, b, t in recs if f == fmax)
    result["roofline"]["largest_launch"] = {"what": "level-1 message MLP (first layer hoisted, aggregation fused)", "kernel": top_kind, "flop": fmax,
                                            "launches_per_step": len(top) // 3, "avg_launch_us": 1e6 * sum(top) / len(top),
                                            **price(top_kind, fmax * len(top), sum(top)),
                                            "hbm_view": {"algorithmic_bytes_per_launch": top_bytes,
                                                         "achieved": top_bytes * len(top) / sum(top) / 1e9, "unit": "GB/s",
                                                         "frac": top_bytes * len(top) / sum(top) / 1e9 / PEAK_HBM_GBS}}
    eager.close()


def clock_under_load(runner, dev, seconds=2.0):
    """The shader clock the package sustains while THIS workload replays (rocm-smi polled from a side thread during `seconds` of
    hipGraph replays, outside every timed region): the dense peaks the roofline fractions are quoted against assume 2.4 GHz, and
    under its power cap the part clocks lower."""
    import re, subprocess, threading
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            try:
                out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            except Exception:
                return
            m = re.search(r"GPU\[%d\]\s*:\s*sclk clock level:[^(]*\((\d+)Mhz\)" % (dev.index or 0), out)
            w = re.search(r"GPU\[%d\]\s*:\s*[^\n]*Power \(W\):\s*([0-9.]+)" % (dev.index or 0), out)
            if m:
                samples.append((time.perf_counter(), int(m.group(1)), float(w.group(1)) if w else None))
    th = threading.Thread(target=poll, daemon=True)
    t0 = time.perf_counter()
    th.start()
    chunk = max(1, min(runner.max_steps - 2, 50))
    while time.perf_counter() - t0 < seconds:
        runner.rewind()
        runner.run(chunk)
        torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    stop.set()
    th.join(timeout=15)
    busy = [(c, w) for t, c, w in samples if t0 + 0.5 <= t <= t1]          # (samples taken while the replays were running)
    if not busy:
        return None
    clocks = sorted(c for c, _ in busy)
    powers = [w for _, w in busy if w is not None]
    return {"clock_mhz_under_load": clocks[len(clocks) // 2], "samples": len(busy), "nominal_mhz": 2400,
            "package_power_w": (sorted(powers)[len(powers) // 2] if powers else None),
            "how": f"rocm-smi --showclocks polled during {seconds:g} s of hipGraph replays of this workload (median of the samples under load)"}


def config1_leg(gfd, S, ops, dev, cpu_steps=1):
    """BASELINE config 1: single-scale MuS-GNN, ONE MP layer, 2k-node synthetic 2-D mesh, one rollout step on the CPU path — here the
    reference's unit (one GNBlock, nn/blocks.py:175-186) on the HIP kernels and on the CPU oracle, and one rollout step of the
    published single-scale model (NsOneScaleGNN: encoders + its MP layers + decoder) both ways."""
    from graphs4cfd_amd.nn import blocks as B
    from graphs4cfd_amd import _lib
    from oracle import g4c_oracle as O
    H = 128
    g = S.mus_graph(2000, levels=1, seed=0)
    torch.manual_seed(0)
    model = gfd.nn.NsOneScaleGNN(arch=S.mus_arch("NsOneScaleGNN", H), device=dev)
    model.eval()
    blk = B.GNBlock((3 * H, (H, H, H), True), (2 * H, (H, H, H), True)).to(dev)
    n, E = int(g.pos.size(0)), int(g.edge_index.size(1))
    v, e, ei = torch.randn(n, H, device=dev), torch.randn(E, H, device=dev), g.edge_index.to(dev)
    out = {"mesh": {"nodes": n, "edges": E}, "precision": ops.mlp_precision()}
    with torch.no_grad():
        for _ in range(3):
            blk(v, e, ei)
        torch.cuda.synchronize(dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            blk(v, e, ei)
        b.record()
        torch.cuda.synchronize(dev)
        out["mp_layer_hip_us"] = a.elapsed_time(b) * 1e3 / 50
        w = {"b." + k: t.detach().cpu() for k, t in blk.state_dict().items()}
        vc, ec, eic = v.cpu(), e.cpu(), g.edge_index
        torch.set_num_threads(16)
        O.gn_block(vc, ec, eic, w, "b")
        t0 = time.perf_counter()
        ref_v, _ = O.gn_block(vc, ec, eic, w, "b")
        out["mp_layer_cpu_oracle_us"] = (time.perf_counter() - t0) * 1e6
        out["mp_layer_max_abs_diff"] = (blk(v, e, ei)[0].cpu() - ref_v).abs().max().item()
        # one rollout step of the published single-scale model
        y = model.solve(g.clone(), 2, capture=False)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        y = model.solve(g.clone(), 1, capture=False)
        torch.cuda.synchronize(dev)
        out["rollout_step_hip_ms"] = (time.perf_counter() - t0) * 1e3
        wm = {k: t.detach().cpu() for k, t in model.state_dict().items()}
        O.mus_solve("NsOneScaleGNN", g.to_dict(), wm, 1, model.num_fields)
        t0 = time.perf_counter()
        ref = O.mus_solve("NsOneScaleGNN", g.to_dict(), wm, 1, model.num_fields)
        out["rollout_step_cpu_oracle_ms"] = (time.perf_counter() - t0) * 1e3
        out["rollout_step_max_abs_diff"] = (y.cpu() - ref).abs().max().item()
    out["cpu_threads"] = 16
    out["what"] = ("one GNBlock launch chain (message MLP + aggregation + node MLP, eager, incl. launch overhead) and one eager solve() step of "
                   "NsOneScaleGNN on a 2 000-node mesh, against the CPU oracle on 16 threads")
    return out


def partition_check(args, runner, model, graph_cpu, dev, rank, world, Rollout):
    """Two partitioned steps against a single-rank rollout of the same mesh (rank 0 computes it), then one instrumented eager
    step: per-rank compute time, per-exchange time, halo bytes."""
    import ctypes as C
    import torch.distributed as dist
    from graphs4cfd_amd import _lib, ops
    # ---- every rank first says where its launches go (ranks 1..N-1 run code paths a one-GPU box never executes —
    # the DeviceGuard of every C entry point, the per-device range-flag buffers, the CU count the persistent grids are sized with) and
    # the job stops, with the offending rank's own message, before anything is timed if a rank's buffers are not on ITS device
    def where():
        d_lib, d_cu = C.c_int32(-1), C.c_int32(-1)
        _lib.check(_lib.load().g4c_device_info(runner.field.data_ptr(), C.byref(d_lib), C.byref(d_cu)))
        flags = ops._range_bufs.get(ops._indexed(dev))
        other = [str(d) for d in ops._range_bufs if d != ops._indexed(dev)]          # flag buffers this process made on ANOTHER device
        m = {"rank": rank, "device": str(dev), "torch_current_device": int(torch.cuda.current_device()), "library_device": int(d_lib.value),
             "cu_count": int(d_cu.value), "field_device": str(runner.field.device),
             "range_flags_device": None if flags is None else str(flags.device), "range_flags_elsewhere": other,
             "weights_device": str(next(model.parameters()).device)}
        wrong = [k for k in ("field_device", "range_flags_device", "weights_device") if m[k] not in (None, str(dev))]
        if m["library_device"] != dev.index or m["torch_current_device"] != dev.index or other:
            wrong.append("library_device / torch_current_device / range_flags_elsewhere")
        m["ok"] = not wrong
        everyone = [None] * world
        dist.all_gather_object(everyone, m)
        if any(not x["ok"] for x in everyone):
            raise SystemExit("partition_check: a rank's buffers / launches are not on its own device: " +
                             "; ".join(json.dumps(x) for x in everyone if not x["ok"]))
        return everyone
    where()                                     # (before the first launch: model, field, library)
    runner.run(2)                               # step 1 eager, step 2 captured (or eager when capture is off / failed)
    everyone = where()                          # (after it: the range-flag buffers exist now)
    full = runner.gather_outputs()              # [N, nf * max_steps] on every rank
    nf = runner.nf
    out = {"steps": 2, "tol": 5e-4, "capture": bool(runner.capture), "capture_note": runner.capture_error, "devices": everyone}
    if rank == 0:
        single = Rollout(model, graph_cpu.clone().to(dev), 2, capture=False)
        single.run(2)
        torch.cuda.synchronize(dev)
        dev_abs = (full[:, : 2 * nf] - single.result()[:, : 2 * nf]).abs()
        diff = dev_abs.max().item()
        out["max_abs_diff_vs_single_rank"] = diff
        if args.precision == "bf16":
            # rounded-bf16 mode (config 3): which kernel takes a launch depends on its row count (the row-split kernels from 20k rows),
            # so a sub-mesh and the whole mesh sum in different orders and round different hidden activations to the neighbouring bf16 —
            # the two results agree to the mode's noise floor (the one its oracle test bounds: tests/test_gpu_parity.py
            # test_remus_20k_vs_oracle), not to fp32 round-off
            p999 = dev_abs.flatten().kthvalue(max(1, int(0.999 * dev_abs.numel()))).values.item()
            out.update({"tol": None, "criterion": "rounded-bf16 mode: mean < 1e-2, 99.9th percentile < 3e-2, max < 1e-1",
                        "mean_abs_diff_vs_single_rank": dev_abs.mean().item(), "p999_abs_diff_vs_single_rank": p999})
            out["ok"] = bool(dev_abs.mean().item() < 1e-2 and p999 < 3e-2 and diff < 1e-1)
        else:
            out["ok"] = bool(diff <= out["tol"])
        single.close()
        del single
    ok = torch.tensor([1 if out.get("ok", True) else 0], dtype=torch.int32, device=dev)
    dist.broadcast(ok, 0)
    out["ok"] = bool(int(ok.item()))
    # instrumented eager step: events around the whole step and around every exchange
    ex = runner.fwd.xch
    ex.reset_stats()
    ex.timing = True
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dist.barrier()
    torch.cuda.synchronize(dev)
    s.record()
    with torch.no_grad():
        runner._one()
    e.record()
    torch.cuda.synchronize(dev)
    ex.timing = False
    step_ms = s.elapsed_time(e)
    ex_us = [a.elapsed_time(b) * 1e3 for a, b in ex.events]
    mine = torch.tensor([step_ms, sum(ex_us) * 1e-3, float(len(ex_us)), float(ex.bytes_sent), float(ex.bytes_recv),
                         float(runner.mesh.owned_global[0].numel())], dtype=torch.float64, device=dev)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    runner.steps_done += 1
    out["per_rank"] = [{"rank": q, "owned_nodes": int(v[5].item()), "eager_step_ms": round(v[0].item(), 3),
                        "in_exchanges_ms": round(v[1].item(), 3), "compute_ms": round(v[0].item() - v[1].item(), 3),
                        "exchanges_per_step": int(v[2].item()),
                        "avg_exchange_us": round(1e3 * v[1].item() / max(v[2].item(), 1.0), 1),
                        "halo_bytes_sent_per_step": int(v[3].item()), "halo_bytes_received_per_step": int(v[4].item())}
                       for q, v in enumerate(allr)]
    return out


def side_config(name, gfd, S, ops, Rollout, dev, steps, warmup=3):
    """A short in-process leg of another BASELINE configuration (beside the default headline run, so that the driver's line carries
    it): its own mesh — built on the device —, model and arithmetic; hipGraph-replayed rollout timed like the headline; one eager
    instrumented step for the dominant kernel's roofline fractions."""
    w = WORKLOADS[name]
    remus = w["model"] == "NsRotEquiTreeScaleGNN"
    old = ops.set_mlp_precision(w["precision"])
    try:
        if remus:
            graph = S.remus_graph(w["nodes"], k=5, seed=0, device=dev)
            arch = S.remus_arch(128)
        else:
            graph = S.mus_graph(w["nodes"], levels=MUS_LEVELS[w["model"]], dim=w["dim"], seed=0, device=dev)
            arch = S.mus_arch(w["model"], 128, dim=w["dim"])
        torch.manual_seed(0)
        model = getattr(gfd.nn, w["model"])(arch=arch, device=dev)
        model.eval()
        ro = Rollout(model, graph, steps + warmup + 8, capture=True)
        ro.run(2 + warmup)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        ro.run(steps)
        redo = ro.validate()              # (f16x3: range flags read inside the timed region, as in the headline leg)
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        out = {"recomputed_in_bf16x6": bool(redo), "metric": w["metric"], "value": steps / el, "unit": "rollout timesteps/s", "ms_per_step": 1e3 * el / steps, "steps": steps,
               "warmup": warmup, "precision": w["precision"], "nodes": w["nodes"], "model": w["model"], "data": "synthetic (device-built mesh)",
               "outputs_finite": bool(torch.isfinite(ro.outputs).all().item()), "cached_static_encoders": ro.static.misses > 0}
        ro.close()
        eager = Rollout(model, graph, 6, capture=False)
        eager.run(2)
        torch.cuda.synchronize(dev)
        with ops.KernelTimer() as kt:
            eager.run(2)
        torch.cuda.synchronize(dev)
        summ = kt.summary()
        eager.close()
        kinds = [k for k in summ if k.startswith("mlp_")]
        dom = max(kinds, key=lambda k: summ[k]["seconds"])
        m = summ[dom]
        prod = {"bf16x6": BX6_PRODUCTS, "f16x3": F16X3_PRODUCTS}.get(w["precision"], 1)
        out["dominant_kernel"] = {"kernel": dom, "launches_per_step": m["launches"] // 2, "avg_launch_us": 1e6 * m["seconds"] / m["launches"],
                                  "hbm_frac": m["bytes"] / m["seconds"] / 1e9 / PEAK_HBM_GBS,
                                  "mfma_frac": prod * m["flops"] / m["seconds"] / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                                  "share_of_fused_mlp_time": m["seconds"] / sum(summ[k]["seconds"] for k in kinds)}
        return out
    finally:
        ops.set_mlp_precision(old)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if launcher_needed(args.gpus, os.environ):
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    import graphs4cfd_amd as gfd
    from graphs4cfd_amd import ops, synthetic as S
    from graphs4cfd_amd.nn.model import Rollout
    ops.set_mlp_precision(args.precision)
    remus = args.model == "NsRotEquiTreeScaleGNN"

    # G4C_BENCH_SAME_GPU=1 (functional check on a single-GPU box only): every rank uses cuda:0 and the gloo transport
    same_gpu = os.environ.get("G4C_BENCH_SAME_GPU", "0") == "1"
    if same_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        if same_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    graph_cpu, model, nf = build_workload(args, gfd, S, dev)
    total_steps = args.warmup + args.steps
    check = None
    if world > 1:
        from graphs4cfd_amd import partition
        runner = partition.DistributedRollout(model, graph_cpu, total_steps + 6, rank, world, dev, capture=not same_gpu)
        if not args.no_partition_check:
            check = partition_check(args, runner, model, graph_cpu, dev, rank, world, Rollout)
    else:
        runner = Rollout(model, graph_cpu.clone().to(dev), total_steps + 2, capture=True)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    # untimed: the eager first step (plans, packing), the capture step, then W warm-up replays
    if check is None:
        runner.run(2)
    runner.run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    runner.run(args.steps)
    # the default arithmetic runs optimistically (fp16 exponent range, every conversion range-checked on the device): reading the
    # flags — and recomputing in bf16x6 had anything been clipped — belongs to the work, so it is inside the timed region
    recomputed = bool(runner.validate())
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    finite = bool(torch.isfinite(runner.outputs).all().item())

    if remus:
        n_mp = 16                          # 4/2/4/2/4 EdgeMP layers (+ 2 DownEdgeMP, 2 UpEdgeMP)
        what = (f"REMuS-GNN (NsRotEquiTreeScaleGNN, published arch, H={args.hidden}) rollout on a {args.nodes}-node synthetic 2D mesh, "
                f"kNN k=5, 3 scales, {int(graph_cpu.angle_index.size(1))} angles, hipGraph-replayed step")
    else:
        n_mp = sum(1 for n in S.MUS_LAYERS[args.model].split() if n.startswith("mp"))
        what = (f"{args.model} (published arch, H={args.hidden}) rollout on a {args.nodes}-node synthetic {args.dim}D mesh, "
                f"kNN k=6, {MUS_LEVELS[args.model]} grid-clustered scale(s), hipGraph-replayed step")
    if world > 1 and args.workload == "headline" and not args.custom:
        what = (f"BASELINE config {'4' if world == 4 else '4-style'} (100k-node mesh node-partitioned {world}-way, halo exchange via RCCL/xGMI, "
                f"{world} MI355X): " + what)
    result = {
        "metric": args.metric if not args.custom else f"rollout timesteps/s ({args.model}, {args.nodes} nodes)",
        "value": args.steps / elapsed, "unit": "rollout timesteps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": {"fp32": "f32", "bf16x6": "f32 (MLP products: exact 3-way bf16 split of both operands, 6 partial products on the bf16 MFMA "
                  "pipe, fp32 accumulate; error vs fp64 <= the fp32-MFMA kernels')",
                  "f16x3": "f32 (MLP products: two-way fp16 split of both operands, 22 significand bits each, 3 partial products on the f16 "
                  "MFMA pipe, fp32 accumulate; error vs fp64 within that of an fp32 GEMM)",
                  "bf16": "bf16 MLP operands (fp32 accumulate, bias, SELU, LayerNorm, aggregation)"}[args.precision], "data": "synthetic",
        "config": {"workload": what, "name": args.workload if not args.custom else "custom", "nodes": args.nodes,
                   "edges": int(graph_cpu.edge_index.size(1)), "mp_layers_per_step": n_mp,
                   "partition": "none" if world == 1 else (f"{world}-way node partition (recursive coordinate bisection), " + (
                       "edge-latent halo exchange per EdgeMP layer (RCCL)" if remus else "halo exchange per MP layer (RCCL)"))},
        "outputs_finite": finite,
    }
    if getattr(runner, "_perm", None) is not None:
        result["config"]["node_numbering"] = ("level-1 nodes renumbered along a Morton curve for the rollout (graphs4cfd_amd/reorder.py: same mesh, "
                                              "output rows mapped back; G4C_REORDER=0 runs the mesh as numbered, -0.6 % steps/s)")
    # BASELINE.json's second figure: average over the step's MP layers of all levels (pool / unpool / encoders included)
    result["ms_per_mp_layer"] = result["ms_per_step"] / n_mp
    if check is not None:
        result["partition_check"] = check

    if rank == 0 and args.precision == "f16x3":
        result["range_safety"] = {
            "scheme": "optimistic: f16x3 products (fp16 exponent range); every value converted to fp16 is range-checked on the device; the "
                      "flags are read INSIDE the timed region (Rollout.validate); a rollout that reached +-65504 anywhere is recomputed "
                      "from its input window in bf16x6 (fp32 exponent range) and stays there, so a delivered result never holds a clipped value",
            "recomputed_in_bf16x6": recomputed}
    if rank == 0 and world == 1 and args.precision == "f16x3" and not args.no_strict_range:
        # the same workload, same process, in the arithmetic a clipped rollout falls back to (three-way bf16 split, six products):
        # the rate of a model whose activations DO leave the fp16 range (the reference computes in fp32: nn/model.py:303-321)
        ops.set_mlp_precision("bf16x6")
        model.invalidate_packed()
        n_strict = max(10, args.steps // 4)
        r2 = Rollout(model, graph_cpu.clone().to(dev), n_strict + args.warmup + 4, capture=True)
        r2.run(2 + args.warmup)
        torch.cuda.synchronize(dev)
        ts = time.perf_counter()
        r2.run(n_strict)
        torch.cuda.synchronize(dev)
        es = time.perf_counter() - ts
        result["range_safety"]["fallback_arithmetic"] = {"value": n_strict / es, "unit": "rollout timesteps/s", "ms_per_step": 1e3 * es / n_strict,
                                                         "steps": n_strict, "precision": "bf16x6",
                                                         "outputs_finite": bool(torch.isfinite(r2.outputs).all().item())}
        r2.close()
        ops.set_mlp_precision(args.precision)
        model.invalidate_packed()

    if rank == 0:
        st = getattr(runner, "static", None)
        result["cached_static_encoders"] = bool(st is not None and st.misses > 0 and st.hits > 0)
        result["config"]["static_cache"] = ("the launches whose inputs solve() never changes (selu(edge_encoder(edge_attr)); REMuS-GNN: the five "
                                            "angle encoders) run once per rollout, in the first eager step; a bare forward() recomputes them")

    clk = None
    if rank == 0 and world == 1 and not args.no_roofline:
        try:
            clk = clock_under_load(runner, dev)
        except Exception as exc:          # (no rocm-smi on the box: the line simply has no clock)
            clk = {"error": f"{type(exc).__name__}: {exc}"}
        roofline_blocks(args, result, model, graph_cpu, dev, ops, Rollout)
        if clk and "clock_mhz_under_load" in clk:
            r = result["roofline"]
            scale = clk["clock_mhz_under_load"] / clk["nominal_mhz"]
            r["clock_mhz_under_load"] = clk["clock_mhz_under_load"]
            r["peak_at_that_clock"] = r["peak"] * scale if r.get("unit") == "TFLOP/s" else r["peak"]
            r["frac_at_that_clock"] = r["achieved"] / r["peak_at_that_clock"]
            r["clock_note"] = (f"every MFMA `frac` is quoted against the dense peak at {clk['nominal_mhz']} MHz; under this load the package sustains "
                               f"{clk['clock_mhz_under_load']} MHz ({clk.get('package_power_w')} W), i.e. a ceiling of {scale:.3f} for such a frac "
                               "(HBM peaks do not scale with the shader clock)")
            r["clock_sampling"] = clk

    if rank == 0 and world == 1 and args.workload == "headline" and not args.custom and not args.no_side_configs:
        # BASELINE configs 2 and 3 as short legs of the same process (their own meshes, models and arithmetic), so that the
        # driver-timed line carries them; `python bench.py --workload c2|c3` gives the full lines
        del runner
        torch.cuda.empty_cache()
        result["configs"] = {}
        try:
            result["configs"]["c1"] = config1_leg(gfd, S, ops, dev)
        except Exception as exc:
            result["configs"]["c1"] = {"error": f"{type(exc).__name__}: {exc}"}
        for name, n_steps in (("c2", 400), ("c3", 40)):
            try:
                result["configs"][name] = side_config(name, gfd, S, ops, Rollout, dev, n_steps)
            except Exception as exc:       # a failing side leg must not take the headline line with it
                result["configs"][name] = {"error": f"{type(exc).__name__}: {exc}"}
            torch.cuda.empty_cache()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        weights = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        result["cpu_baseline"] = cpu_baseline(args, S, weights, nf, args.cpu_budget_s)

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        torch.distributed.destroy_process_group()
    if check is not None and not check["ok"]:
        raise SystemExit("partitioned rollout differs from the single-rank rollout beyond 5e-4")


if __name__ == "__main__":
    main()
