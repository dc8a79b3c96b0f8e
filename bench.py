"""bench.py — rollout timesteps/s of the MuS-GNN hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Workload (BASELINE.json metric "rollout timesteps/s ... on 100k-node 2D mesh"; SURVEY.md §8(d) C4 mesh):
NsThreeScaleGNN with the published arch (H = 128, 4/2/4/2/4 MP layers), synthetic 100 000-node 2D
mesh (uniform random points, kNN k = 6, grid-clustered levels 2 and 3), fp32, random-init weights.
A step = one forward of the whole V-cycle + the rollout bookkeeping kernel, replayed from a hipGraph.
With N > 1 the same mesh is node-partitioned over the ranks (strong scaling, one halo exchange per MP
layer over RCCL).

One JSON line on stdout (rank 0).  Extra objects:
  roofline     — dominant kernel (g4c fused MLP, fp32 MFMA): algorithmic FLOP / measured duration,
                 from HIP-event pairs around every launch of an eager pass of the same step.
  roofline_scatter — the CSR segment-reduce ("scatter-sum") against the HBM roofline, same method.
  cpu_baseline — the oracle (pure-torch restatement of the reference CPU path) timed on this host.
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
BX6_PRODUCTS = 6                # bf16 partial products the default kernel executes per fp32 multiply-add
PEAK_HBM_GBS = 8000.0           # HBM3E spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nodes", type=int, default=100_000)
    ap.add_argument("--model", default="NsThreeScaleGNN")
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--precision", default="bf16x6", choices=["bf16x6", "fp32", "bf16"],
                    help="arithmetic of the fused MLPs: bf16x6 (default: fp32-accurate split products on the bf16 matrix pipe), fp32 "
                         "(fp32 MFMA kernels) or bf16 (operands rounded to bf16, ~1e-2 deviation; never a headline number)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=30.0)
    return ap.parse_args()


def cpu_baseline(model_name, graph, weights, nf, budget_s):
    """Oracle rollout steps on the host cores: bounded sample (>= 1 step, <= budget)."""
    from oracle import g4c_oracle as O
    # torch CPU ops stop scaling long before the host's core count on this path (measured on the GPU box,
    # 2 x EPYC 9575F = 256 hw threads: 8/16/32/64/128/256 threads -> 1.29/1.10/1.26/2.13/6.11/103 s per
    # 20k-node step, profiles/r01_cpu_thread_sweep.log), so the baseline uses the fastest setting
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    g = graph.to_dict()
    t0 = time.perf_counter()
    steps = 0
    with torch.no_grad():
        while True:
            pred = O.mus_forward(model_name, g, weights, nf)
            g = dict(g)
            g["field"] = O.shift_and_replace(g["field"], pred, nf)
            steps += 1
            el = time.perf_counter() - t0
            if steps >= 3 or el + el / steps > budget_s:
                break
    return {"value": steps / el, "unit": "rollout timesteps/s", "cores": cores, "kind": "port",
            "sample": f"{steps} rollout step(s) of the same {graph.num_nodes}-node mesh and weights, "
                      f"oracle (op-for-op torch restatement of the reference CPU path, per-step pool_edge rebuild), "
                      f"{torch.get_num_threads()} torch threads, {el:.1f} s"}


def pmc_traffic():
    """HBM bytes per launch from the committed PMC collection (scripts/collect_pmc_traffic.sh: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same bench, gfx950 FETCH_SIZE correction).
    rocprofv3 cannot run inside the timed process, so the latest committed collection is quoted with its source."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, None
    k = json.load(open(files[-1]))["kernels"]

    def avg(prefix):
        sel = [v for name, v in k.items() if name.startswith(prefix)]
        n = sum(v["dispatches"] for v in sel)
        return sum(v["hbm_bytes_per_launch"] * v["dispatches"] for v in sel) / n if n else None
    def largest(prefix):
        sel = [v.get("hbm_bytes_largest_launch") for name, v in k.items() if name.startswith(prefix) and v.get("hbm_bytes_largest_launch")]
        return max(sel) if sel else None
    # (the level-1 aggregation = the largest segment-reduce dispatches of the profiled bench run)
    return {"avg": avg, "scatter": largest("segment_reduce_kernel") or avg("segment_reduce_kernel")}, os.path.relpath(files[-1], ROOT)


def reference_flop_per_step(model, g, S):
    """2*MAC of every nn.Linear at the row count the reference applies it to (163 840 FLOP per edge and
    131 072 per node for an MP layer at H = 128, etc.), for the MuS-GNN V-cycle on this mesh."""
    def mlp_flop(mlp):
        return 2.0 * sum(l.weight.numel() for l in mlp._linears())
    import numpy as np
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


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    import graphs4cfd_amd as gfd
    from graphs4cfd_amd import ops, synthetic as S
    from graphs4cfd_amd.nn.model import Rollout
    ops.set_mlp_precision(args.precision)
    if args.precision == "bf16":       # reduced precision: timing only
        args.no_roofline = True

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

    levels = {"NsOneScaleGNN": 1, "NsTwoScaleGNN": 2, "NsThreeScaleGNN": 3, "NsFourScaleGNN": 4}[args.model]
    graph_cpu = S.mus_graph(args.nodes, levels=levels, seed=0)
    arch = S.mus_arch(args.model, args.hidden)
    torch.manual_seed(0)
    model = getattr(gfd.nn, args.model)(arch=arch, device=dev)
    model.eval()
    nf = model.num_fields
    total_steps = args.warmup + args.steps

    if world > 1:
        from graphs4cfd_amd import partition
        runner = partition.DistributedRollout(model, graph_cpu, total_steps + 2, rank, world, dev, capture=not same_gpu)
    else:
        runner = Rollout(model, graph_cpu.clone().to(dev), total_steps + 2, capture=True)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    # untimed: the eager first step (plans, packing), the capture step, then W warm-up replays
    runner.run(2)
    runner.run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    runner.run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    finite = bool(torch.isfinite(runner.outputs).all().item())

    result = {
        "metric": "rollout timesteps/s (100k-node 2D mesh)", "value": args.steps / elapsed, "unit": "rollout timesteps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": {"fp32": "f32", "bf16x6": "f32 (MLP products: exact 3-way bf16 split of both operands, 6 partial products on the bf16 MFMA "
                  "pipe, fp32 accumulate; error vs fp64 <= the fp32-MFMA kernels')",
                  "bf16": "bf16 MLP operands (fp32 accumulate, bias, SELU, LayerNorm, aggregation)"}[args.precision], "data": "synthetic",
        "config": {"workload": f"{args.model} (published arch, H={args.hidden}) rollout on a {args.nodes}-node synthetic 2D mesh, "
                               f"kNN k=6, {levels} grid-clustered scale(s), hipGraph-replayed step",
                   "nodes": args.nodes, "edges": int(graph_cpu.edge_index.size(1)), "mp_layers_per_step": sum(
                       1 for n in S.MUS_LAYERS[args.model].split() if n.startswith("mp")),
                   "partition": "none" if world == 1 else f"{world}-way node partition, halo exchange per MP layer (RCCL)"},
        "outputs_finite": finite,
    }
    # BASELINE.json's second figure: average over the step's MP layers of all levels (pool / unpool / encoders included)
    result["ms_per_mp_layer"] = result["ms_per_step"] / result["config"]["mp_layers_per_step"]

    if rank == 0 and world == 1 and not args.no_roofline:
        # eager instrumented pass of the same step: HIP-event pair around every launch, on the launch stream
        eager = Rollout(model, graph_cpu.clone().to(dev), 8, capture=False)
        eager.run(2)
        torch.cuda.synchronize(dev)
        with ops.KernelTimer() as kt:
            eager.run(3)
        torch.cuda.synchronize(dev)
        summ = kt.summary()
        s = summ["segment_reduce"]

        def price(kind, flops, seconds):
            """roofline pricing of `flops` ALGORITHMIC fp32 FLOP (2*K*N per row and layer) done in `seconds` by kernel `kind`:
            the fp32 kernels execute exactly those on the fp32 MFMA pipe; the default bf16x6 kernel executes six bf16 MFMA
            products per fp32 multiply-add, so it is priced in executed bf16 FLOP against the dense bf16 peak, with the
            algorithmic rate (and what it would be against the fp32-MFMA peak) alongside."""
            alg = flops / seconds / 1e12
            if kind.startswith("mlp_bx6"):
                return {"achieved": BX6_PRODUCTS * alg, "peak": PEAK_BF16_MFMA_TFLOPS, "frac": BX6_PRODUCTS * alg / PEAK_BF16_MFMA_TFLOPS,
                        "mfma_dtype": "bf16 (6 exact partial products per fp32 MAC, fp32 accumulate)",
                        "algorithmic_fp32_tflops": alg, "algorithmic_vs_fp32_mfma_peak": alg / PEAK_FP32_MFMA_TFLOPS}
            return {"achieved": alg, "peak": PEAK_FP32_MFMA_TFLOPS, "frac": alg / PEAK_FP32_MFMA_TFLOPS, "mfma_dtype": "f32"}

        def mfma_entry(kind):
            m = summ[kind]
            return {"launches_per_step": m["launches"] // 3, "avg_launch_us": 1e6 * m["seconds"] / m["launches"],
                    "flop_per_launch": m["flops"] / m["launches"], "ms_per_step": 1e3 * m["seconds"] / 3,
                    **price(kind, m["flops"], m["seconds"])}

        traffic, traffic_src = pmc_traffic()
        default_workload = (args.nodes == 100_000 and args.model == "NsThreeScaleGNN" and args.hidden == 128)
        if not default_workload:
            traffic = None
        mlp_kinds = [k for k in summ if k.startswith("mlp_")]
        # FLOP of the reference's formulation (every Linear applied per row of its concatenated input)
        ref_flop = reference_flop_per_step(model, graph_cpu, S)
        dom = max(mlp_kinds, key=lambda k: summ[k]["seconds"])      # dominant kernel instantiation by GPU time
        big = mfma_entry(dom)
        tot_f = sum(summ[k]["flops"] for k in mlp_kinds)
        tot_t = sum(summ[k]["seconds"] for k in mlp_kinds)
        # (rocprofv3 reports it as <name><N, true|false>: all-vectorisable sources or not)
        result["roofline"] = {"bound": "mfma", "kernel": (dom + "<1, *, *> (g4c_mlp_forward_bx6 / _heads_bx6)") if dom.startswith("mlp_bx6")
                              else dom.replace(">", ", *>") + " (g4c_mlp_forward)",
                              "achieved": big["achieved"], "peak": big["peak"], "unit": "TFLOP/s", "frac": big["frac"],
                              "mfma_dtype": big["mfma_dtype"], "algorithmic_fp32_tflops": big.get("algorithmic_fp32_tflops", big["achieved"]),
                              "algorithmic_vs_fp32_mfma_peak": big.get("algorithmic_vs_fp32_mfma_peak", big["frac"]),
                              "traffic": traffic["avg"](dom.rstrip(">")) if traffic else None, "traffic_source": traffic_src if traffic else None,
                              "launches_per_step": big["launches_per_step"],
                              "avg_launch_us": big["avg_launch_us"], "flop_per_launch": big["flop_per_launch"],
                              "ms_per_step_in_kernel": big["ms_per_step"],
                              "flop_definition": "flop_per_launch = algorithmic fp32 FLOP (2*K*N per row and layer) of the launches as "
                                                 "executed.  The node-side products of every edge MLP's first layer are hoisted to one "
                                                 "product per node (exact re-association), so a step executes fewer FLOP than the "
                                                 "reference formulation's count below",
                              "all_mlp_kernels": {"algorithmic_fp32_tflops": tot_f / tot_t / 1e12,
                                                  "algorithmic_vs_fp32_mfma_peak": tot_f / tot_t / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                                                  "flop_per_step": tot_f / 3, "ms_per_step": 1e3 * tot_t / 3,
                                                  "reference_formulation_flop_per_step": ref_flop,
                                                  "reference_formulation_tflops": ref_flop / (tot_t / 3) / 1e12},
                              "other_mlp_kernels": {k: mfma_entry(k) for k in mlp_kinds if k != dom}}
        result["roofline_scatter"] = {"bound": "hbm", "kernel": "segment_reduce_kernel (g4c_segment_reduce)", "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                      "step_launches": {"what": "the segment reductions one rollout step still launches (DownMP cluster means, "
                                                                "pool_edge, coarse levels without aggregation on load)",
                                                        "achieved": s["bytes"] / s["seconds"] / 1e9,
                                                        "algorithmic_bytes_per_launch": s["bytes"] / s["launches"],
                                                        "launches_per_step": s["launches"] // 3,
                                                        "avg_launch_us": 1e6 * s["seconds"] / s["launches"]}}
        # the level-1 aggregation alone (the 358.8 MB case of BASELINE.md §4), as a standalone g4c_segment_reduce launch on
        # the level-1 edge latents: in the default rollout this reduction is folded into the node-MLP launch (aggregation
        # on load, g4c_src_t.seg_off), so the kernel is timed here explicitly — same kernel, same plan, same bytes
        from graphs4cfd_amd import plan as _plan
        g_dev = eager.graph
        ep1, csr1 = _plan.edge_csr(g_dev.edge_index, int(g_dev.field.size(0)))
        msgs = torch.randn((csr1.n, args.hidden), dtype=torch.float32, device=dev)
        agg1 = torch.empty((csr1.n_seg, args.hidden), dtype=torch.float32, device=dev)
        for _ in range(3):
            ops.segment_reduce(msgs, csr1, True, out=agg1)
        with ops.KernelTimer() as kt1:
            for _ in range(10):
                ops.segment_reduce(msgs, csr1, True, out=agg1)
        torch.cuda.synchronize(dev)
        sel = [(b, a.elapsed_time(e) * 1e-3) for k, f, b, a, e in kt1.records if k == "segment_reduce"]
        bmax = sel[0][0]
        lvl1 = {"bytes": bmax, "avg_launch_us": 1e6 * sum(t for _, t in sel) / len(sel),
                "achieved": bmax * len(sel) / sum(t for _, t in sel) / 1e9,
                "frac": bmax * len(sel) / sum(t for _, t in sel) / 1e9 / PEAK_HBM_GBS,
                "how": "10 standalone launches on the level-1 messages; in the rollout this aggregation runs inside the "
                       "node-MLP launch's gather (bit-identical)"}
        result["roofline_scatter"].update({"achieved": lvl1["achieved"], "frac": lvl1["frac"], "algorithmic_bytes_per_launch": bmax,
                                           "avg_launch_us": lvl1["avg_launch_us"], "traffic": traffic["scatter"] if traffic else None,
                                           "level1": lvl1})
        # one level-1 MP layer = the two largest fused-MLP launches of the layer (edge MLP; node MLP incl. the aggregation on load)
        recs = [(k, f, b, a.elapsed_time(e) * 1e-3) for k, f, b, a, e in kt.records]
        n1, e1 = int(g_dev.field.size(0)), int(csr1.n)
        edge_t = [t for k, f, b, t in recs if k.startswith("mlp_") and f == max(f2 for k2, f2, b2, t2 in recs if k2.startswith("mlp_"))]
        mlp_bpr = lambda rows: 4.0 * rows           # (bytes recorded per launch are proportional to its row count)
        node_f = sorted({f for k, f, b, t in recs if k.startswith("mlp_")}, reverse=True)
        seg_t = [t for k, f, b, t in recs if k == "segment_reduce" and b == bmax]
        # node launches: the most frequent large fused-MLP shape after the edge MLP
        import collections as _c
        cnt = _c.Counter(f for k, f, b, t in recs if k.startswith("mlp_") and f != max(node_f))
        if cnt:
            f_node = max(cnt, key=lambda f: (cnt[f], f))
            node_t = [t for k, f, b, t in recs if k.startswith("mlp_") and f == f_node]
            result["ms_per_mp_layer_level1"] = 1e3 * (sum(edge_t) / len(edge_t) + sum(node_t) / len(node_t)
                                                      + (sum(seg_t) / len(seg_t) if seg_t else 0.0))
        fmax = max(f for k, f, b, t in recs if k.startswith("mlp_"))
        top = [t for k, f, b, t in recs if k.startswith("mlp_") and f == fmax]
        result["roofline"]["largest_launch"] = {"what": "level-1 edge MLP (first layer hoisted)", "flop": fmax, "launches_per_step": len(top) // 3,
                                                "avg_launch_us": 1e6 * sum(top) / len(top), **price(dom, fmax * len(top), sum(top))}
        eager.close()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        weights = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        result["cpu_baseline"] = cpu_baseline(args.model, graph_cpu, weights, nf, args.cpu_budget_s)

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
