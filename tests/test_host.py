"""Host-side logic that needs no GPU: the C-ABI library loads and exports every symbol the header declares,
the static-plan builders (host C++) agree with the oracle's topology, the gfd-compatible module surface
(state_dict keys, checkpoint format, Graph container, error behaviour)."""
import os
import re

import numpy as np
import pytest
import torch

import graphs4cfd_amd as gfd
from graphs4cfd_amd import _lib, plan, synthetic as S
from oracle import g4c_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "g4c.h")).read()
    declared = set(re.findall(r"\b(g4c_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"libg4c.so does not export {name}"
    assert declared == set(_lib.EXPORTED_SYMBOLS), (declared ^ set(_lib.EXPORTED_SYMBOLS))
    assert lib.g4c_version() >= 1


def test_plan_csr_matches_stable_argsort():
    rng = np.random.default_rng(0)
    keys = torch.from_numpy(rng.integers(0, 37, size=500))
    p = plan.build_csr(keys, 40, torch.device("cpu"))
    order = torch.sort(keys, stable=True)[1].to(torch.int32)
    assert torch.equal(p.perm, order)
    assert torch.equal(p.off.long(), torch.cat([torch.zeros(1, dtype=torch.long), torch.bincount(keys, minlength=40).cumsum(0)]))
    assert p.max_deg == int(torch.bincount(keys).max())
    # already grouped in order -> no permutation at all
    sorted_keys = torch.arange(50).repeat_interleave(6)
    assert plan.build_csr(sorted_keys, 50, torch.device("cpu")).perm is None
    with pytest.raises(ValueError):
        plan.build_csr(torch.tensor([0, 7]), 5, torch.device("cpu"))


def test_pool_edge_plan_topology_matches_oracle(golden):
    import ctypes as C
    lib = _lib.load()
    for tag in ("mean", "empty"):
        c = golden("blocks.pt")[f"pool_edge_{tag}"]
        idx = np.ascontiguousarray(c["idx"].numpy())
        ei = np.ascontiguousarray(c["edge_index"].numpy())
        n_edges = ei.shape[1]
        coarse = np.empty((2, n_edges), dtype=np.int64)
        perm = np.empty(n_edges, dtype=np.int32)
        off = np.empty(n_edges + 1, dtype=np.int32)
        kept = C.c_int64(0)
        nc = lib.g4c_plan_pool_edge(idx.ctypes.data, idx.shape[0], ei.ctypes.data, n_edges, coarse.ctypes.data,
                                    perm.ctypes.data, off.ctypes.data, C.byref(kept))
        got = torch.from_numpy(coarse.reshape(-1)[: 2 * nc].reshape(2, nc).copy())
        assert torch.equal(got, c["edge_index_out"])
        # the segmented permutation reproduces the reference's mean of duplicate edges
        ea = c["edge_attr"]
        if nc:
            seg = torch.repeat_interleave(torch.arange(nc), torch.from_numpy(np.diff(off[: nc + 1])).long())
            pooled = O.scatter(ea[torch.from_numpy(perm[: kept.value].copy()).long()], seg, nc, "mean")
            torch.testing.assert_close(pooled, c["edge_attr_out"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("cls", sorted(S.MUS_LAYERS))
def test_state_dict_keys_and_shapes_match_reference(golden, cls):
    c = golden("models_mus.pt")[cls]
    model = getattr(gfd.nn, cls)(arch=c["arch"])
    sd = model.state_dict()
    assert list(sd.keys()) == list(c["weights"].keys())
    assert all(sd[k].shape == v.shape for k, v in c["weights"].items())
    model.load_state_dict(c["weights"])
    assert model.num_params == c["num_params"] and model.num_fields == c["arch"]["decoder"][1][-1]


def test_remus_state_dict_and_seeded_init_match_reference(golden):
    c = golden("model_remus.pt")
    torch.manual_seed(400)           # the seed make_golden.py used: same construction order -> same init
    model = gfd.nn.NsRotEquiTreeScaleGNN(arch=c["arch"])
    assert list(model.state_dict().keys()) == list(c["weights"].keys())
    for k, v in c["weights"].items():
        assert torch.equal(model.state_dict()[k], v), k


def test_checkpoint_round_trip_in_reference_format(golden, tmp_path):
    path = os.path.join(os.path.dirname(__file__), "golden", "reference_saved.chk")
    model = gfd.nn.NsOneScaleGNN(checkpoint=path)      # a file written by the reference's save_checkpoint
    assert list(model.state_dict().keys()) == golden("checkpoint_io.pt")["keys"]
    out = str(tmp_path / "mine.chk")
    model.save_checkpoint(out, n_out=2, epoch=5)
    chk = torch.load(out, weights_only=False)
    assert set(chk) >= {"arch", "weights", "n_out", "epoch"} and chk["arch"] == model.arch
    again = gfd.nn.NsOneScaleGNN(checkpoint=out)
    assert all(torch.equal(a, b) for a, b in zip(again.state_dict().values(), model.state_dict().values()))
    w = str(tmp_path / "weights.pt")
    torch.save(model.state_dict(), w)
    third = gfd.nn.NsOneScaleGNN(arch=model.arch, weights=w)
    assert all(torch.equal(a, b) for a, b in zip(third.state_dict().values(), model.state_dict().values()))


def test_graph_container_and_collate():
    g = gfd.Graph(pos=torch.zeros(5, 2), field=torch.ones(5, 3), edge_index=torch.tensor([[0, 1], [1, 2]]))
    assert g.num_nodes == 5 and g.num_edges == 2 and hasattr(g, "field") and not hasattr(g, "loc")
    g.idx1_to_idx2 = torch.arange(5)
    assert getattr(g, "idx1_to_idx2").numel() == 5 and "idx1_to_idx2" in g.keys()
    assert g.to("cpu") is g
    b = gfd.nn.collate([g.clone(), g.clone()])
    assert b.num_nodes == 10 and b.edge_index.tolist() == [[0, 1, 5, 6], [1, 2, 6, 7]]
    assert b.batch.tolist() == [0] * 5 + [1] * 5


def test_product_path_has_no_cpu_fallback_and_validates():
    mlp = gfd.nn.blocks.MLP(8, (16, 16), True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mlp(torch.randn(4, 8))
    model = gfd.nn.NsOneScaleGNN(arch=S.mus_arch("NsOneScaleGNN", 32))
    with pytest.raises(AssertionError):
        model.solve(S.mus_graph(50), 0)
    with pytest.raises(ValueError, match="not recognized"):
        gfd.nn.NsTwoScaleGNN(model="nope")
    with pytest.raises(ValueError):
        gfd.nn.blocks.MLP(4, (8,))
    src = open(os.path.join(ROOT, "graphs4cfd_amd", "ops.py")).read() + open(os.path.join(ROOT, "graphs4cfd_amd", "nn", "blocks.py")).read()
    assert "oracle" not in src, "the product path must not import the oracle"


def test_pool_edge_plan_rejects_unassigned_fine_nodes():
    """A -1 entry (the 'empty' value of the reference's mask2idx tables) must be an argument error, not a heap overrun."""
    import ctypes as C
    lib = _lib.load()
    idx = np.array([0, 1, -1, 1], dtype=np.int64)
    ei = np.array([[0, 1, 2], [1, 2, 3]], dtype=np.int64)
    coarse, perm, off = np.empty((2, 3), dtype=np.int64), np.empty(3, dtype=np.int32), np.empty(4, dtype=np.int32)
    kept = C.c_int64(0)
    nc = lib.g4c_plan_pool_edge(idx.ctypes.data, 4, ei.ctypes.data, 3, coarse.ctypes.data, perm.ctypes.data, off.ctypes.data, C.byref(kept))
    assert nc < 0 and b"negative" in lib.g4c_last_error()


def test_plan_caches_are_bounded_by_bytes():
    c = plan._Cache(capacity=100, max_bytes=10_000)
    keep = []
    for i in range(20):
        k = torch.zeros(250, dtype=torch.int32)      # 1000 bytes pinned by the key + 1000 by the value
        keep.append(k)
        c.put(plan._Cache.key(k), (k,), (torch.zeros(250, dtype=torch.int32), 7))
        assert c.bytes <= 10_000 and c.bytes == sum(e[2] for e in c.data.values())
    assert len(c.data) == 5 and c.get(plan._Cache.key(keep[-1])) is not None and c.get(plan._Cache.key(keep[0])) is None
    big = torch.zeros(100_000, dtype=torch.int32)    # larger than the bound: still cached (alone) for the rollout that needs it
    c.put(plan._Cache.key(big), (big,), None)
    assert len(c.data) == 1 and c.get(plan._Cache.key(big)) is None and plan._Cache.key(big) in c.data
    c.clear()
    assert c.bytes == 0 and not c.data
    assert isinstance(plan.snapshot(), list)


def test_connect_knn_keeps_in_degree_k_with_coincident_points():
    torch.manual_seed(0)
    pos = torch.rand(60, 2)
    pos[10:15] = pos[10]            # five coincident points: a k + 1 query need not return the centre itself
    ei, ea = S.connect_knn(pos, 3)
    assert (np.bincount(ei[1].numpy(), minlength=60) == 3).all() and bool((ei[0] != ei[1]).all())
    torch.testing.assert_close(ea, pos[ei[1]] - pos[ei[0]])


def test_reference_package_name_is_an_alias():
    """`import graphs4cfd as gfd` (the examples' import) binds to the same module objects as graphs4cfd_amd."""
    import graphs4cfd as ref_name
    import graphs4cfd.nn.mus_gnn as mus
    from graphs4cfd.transforms import Compose, ConnectKNN, GridClustering, ScaleEdgeAttr
    assert ref_name.nn is gfd.nn and ref_name.Graph is gfd.Graph and ref_name.DataLoader is gfd.DataLoader
    assert mus.NsThreeScaleGNN is gfd.nn.NsThreeScaleGNN and ConnectKNN is gfd.transforms.ConnectKNN
    g = Compose([ConnectKNN(4), ScaleEdgeAttr(0.1), GridClustering([0.2])])(gfd.Graph(pos=torch.rand(200, 2)))
    assert g.edge_index.shape == (2, 800) and hasattr(g, "cluster_2")
    with pytest.raises(ImportError, match="plot"):
        ref_name.plot


def test_invalidate_packed_bumps_the_weights_epoch():
    from graphs4cfd_amd import ops
    model = gfd.nn.NsOneScaleGNN(arch=S.mus_arch("NsOneScaleGNN", 32))
    e0 = ops.weights_epoch()
    model.invalidate_packed()
    e1 = ops.weights_epoch()
    model.load_state_dict(model.state_dict())
    e2 = ops.weights_epoch()
    model.float()
    assert e0 < e1 < e2 < ops.weights_epoch()
    assert not hasattr(model, "_require_inference")


def test_locality_renumbering_keeps_the_mesh():
    """reorder.reorder_nodes: a permutation of the level-1 nodes along a Morton curve; same edges, grouped by the new target with the
    per-target order kept; level-2 maps carried along; layouts it does not know are left alone."""
    from graphs4cfd_amd.reorder import reorder_nodes
    g = S.mus_graph(3000, levels=3, seed=5)
    g2, perm = reorder_nodes(g)
    n = 3000
    assert sorted(perm.tolist()) == list(range(n))
    assert torch.equal(g2.pos, g.pos[perm]) and torch.equal(g2.field, g.field[perm]) and torch.equal(g2.idx1_to_idx2, g.idx1_to_idx2[perm])
    assert torch.equal(g2.e_12, g.e_12[perm]) and g2.pos_2 is g.pos_2 and g2.idx2_to_idx3 is g.idx2_to_idx3
    col = g2.edge_index[1]
    assert bool((col[1:] >= col[:-1]).all()), "edges grouped by the new target"
    old_edges = set(map(tuple, g.edge_index.t().tolist()))
    new_edges = set((int(perm[r]), int(perm[c])) for r, c in g2.edge_index.t().tolist())
    assert old_edges == new_edges
    # per target: the same senders in the same order, with the same attributes
    for tgt_new in (0, 17, n - 1):
        tgt_old = int(perm[tgt_new])
        a = g.edge_index[0][g.edge_index[1] == tgt_old]
        b = perm[g2.edge_index[0][g2.edge_index[1] == tgt_new]]
        assert torch.equal(a, b)
        torch.testing.assert_close(g2.edge_attr[g2.edge_index[1] == tgt_new], g.edge_attr[g.edge_index[1] == tgt_old])
    # neighbours get nearby numbers: mean |row - col| far below a random numbering's n / 3
    assert float((g2.edge_index[0] - g2.edge_index[1]).abs().float().mean()) < 0.1 * float((g.edge_index[0] - g.edge_index[1]).abs().float().mean())
    assert reorder_nodes(S.remus_graph(300, k=5, seed=1)) is None
    g.extra = torch.zeros(7)
    assert reorder_nodes(g) is None


def test_mlp_precision_names_and_f16_range_warning():
    """The arithmetic modes the host accepts, and the input-magnitude hint of the default one: solve() warns when an
    input tensor is large enough for a hidden activation to reach fp16's range (the f16x3 kernels clip there), and only then."""
    import warnings
    from graphs4cfd_amd import ops
    from graphs4cfd_amd.nn import model as M
    assert ops.PRECISIONS == ("fp32", "bf16", "bf16x6", "f16x3") and ops.mlp_precision() in ops.PRECISIONS
    with pytest.raises(ValueError):
        ops.set_mlp_precision("fp16")
    g = S.mus_graph(300, levels=1, seed=3)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        M._warn_f16_range(g)
    g.field = g.field * (2.0 * M.F16_INPUT_WARN / float(g.field.abs().max()))
    with pytest.warns(RuntimeWarning, match="f16x3"):
        M._warn_f16_range(g)


def test_bench_starts_its_own_ranks_when_not_under_torchrun(monkeypatch):
    """`python bench.py --gpus N` (the driver's command shape) must not die in argument checking: outside a
    torch.distributed.run job it launches the N ranks itself, inside one it is a rank."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("g4c_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.launcher_needed(2, {}) and bench.launcher_needed(8, {"PATH": "/bin"})
    assert not bench.launcher_needed(1, {})
    assert not bench.launcher_needed(2, {"WORLD_SIZE": "2", "RANK": "0"})
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "3"], 29511)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3"]
    # main() takes the launcher branch (and returns the job's return code) before it touches a GPU
    seen = {}

    def fake_launch(gpus, argv):
        seen["gpus"], seen["argv"] = gpus, list(argv)
        return 7
    monkeypatch.setattr(bench, "launch_ranks", fake_launch)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert exc.value.code == 7 and seen == {"gpus": 2, "argv": ["--gpus", "2", "--steps", "2", "--warmup", "1"]}


def test_range_watch_hands_every_hit_to_all_live_consumers_before_clearing():
    """ops.RangeWatch / f16_range_poll (host logic; the flag array stands on the CPU here): whoever reads the shared per-site flags
    distributes each hit to every live consumer that named the site, keeps what nobody asked for for f16_range_report, and only
    then clears — a second consumer of the same model, or a report in between, cannot erase the first one's evidence."""
    from graphs4cfd_amd import ops
    dev = torch.device("cpu")
    saved = (dict(ops._range_bufs), set(ops._range_unclaimed))
    try:
        ops._range_bufs.clear(); ops._range_unclaimed.clear()
        buf = ops._range_buffer(dev)
        sa, sb, so = ops._range_slot("T.model_a.mlp"), ops._range_slot("T.model_b.mlp"), ops._range_slot("T.orphan.mlp")
        wa1 = ops.RangeWatch(dev, ["T.model_a.mlp"])
        buf[sa] = 1                                   # a launch of model A clipped
        wa2 = ops.RangeWatch(dev, ["T.model_a.mlp"])  # a second rollout of A is built: its entry drain must not eat wa1's hit
        assert wa2.hits == set() and wa1.hits == {"T.model_a.mlp"} and int(buf.sum()) == 0
        buf[sb] = 1; buf[so] = 1
        assert ops.f16_range_report(dev, sites=["T.model_b.mlp"]) == ["T.model_b.mlp"]          # nobody watches B: reported, cleared
        assert ops.f16_range_report(dev, clear=False) == ["T.orphan.mlp"]
        assert wa1.take() == ["T.model_a.mlp"] and wa1.take() == [] and wa2.take() == []
        buf[sa] = 1                                   # indistinguishable: both live consumers of the site get it
        assert wa2.take() == ["T.model_a.mlp"] and wa1.take() == ["T.model_a.mlp"]
        wa1.close(); wa2.close()
        buf[sa] = 1
        assert sorted(ops.f16_range_report(dev)) == ["T.model_a.mlp", "T.orphan.mlp"]          # no live watch: unclaimed again
        assert ops.f16_range_report(dev) == []
    finally:
        ops._range_bufs.clear(); ops._range_bufs.update(saved[0])
        ops._range_unclaimed.clear(); ops._range_unclaimed.update(saved[1])


def test_csr_plan_knows_a_uniform_in_degree():
    """CsrPlan.uniform_deg (what G4C_AGG_UNIFORM is fed with): k when EVERY segment has exactly k rows, else 0 (one shorter or
    one empty segment is enough)."""
    dev = torch.device("cpu")
    assert plan.build_csr(torch.arange(50).repeat_interleave(6), 50, dev).uniform_deg == 6
    assert plan.build_csr(torch.arange(7).repeat_interleave(5), 7, dev).uniform_deg == 5
    ragged = torch.cat([torch.arange(49).repeat_interleave(6), torch.full((5,), 49)])
    assert plan.build_csr(ragged, 50, dev).uniform_deg == 0
    assert plan.build_csr(torch.arange(49).repeat_interleave(6), 50, dev).uniform_deg == 0          # the last target is empty
    assert plan.build_csr(torch.arange(3).repeat_interleave(40), 3, dev).uniform_deg == 0           # beyond 32 rows: not offered


def test_row_split_column_order_is_the_mfma_lane_order():
    """ops._rs_k_order (include/g4c.h "row-split order"): position 32 j + 8 g + 4 h + e holds feature 32 j + 16 h + 4 g + e — the eight
    values lane (n, g) of a 16x16x32 MFMA holds of 32-feature step j are 16 contiguous bytes; rs_rows_to_natural undoes it; the tag
    survives row slices and is refused for anything but bf16 [n, 128] rows."""
    import torch
    from graphs4cfd_amd import ops
    order = ops._rs_k_order(torch.device("cpu")).tolist()
    assert sorted(order) == list(range(128))
    for j in range(4):
        for g in range(4):
            got = order[32 * j + 8 * g: 32 * j + 8 * g + 8]
            # a C-layout lane holds features 16 b + 4 g + e of feature blocks b = 2 j and 2 j + 1
            assert got == [16 * (2 * j) + 4 * g + e for e in range(4)] + [16 * (2 * j + 1) + 4 * g + e for e in range(4)]
    rows = torch.arange(3 * 128, dtype=torch.float32).reshape(3, 128).to(torch.bfloat16)
    tagged = ops.RsOrderedRows.tag(rows[:, order].contiguous())
    assert isinstance(tagged[1:], ops.RsOrderedRows)
    back = ops.rs_rows_to_natural(tagged)
    assert type(back) is torch.Tensor and torch.equal(back, rows)
    with pytest.raises(ValueError):
        ops.RsOrderedRows.tag(rows.float())


def test_remus_program_knows_which_run_outputs_only_mlps_read():
    """remus_gnn._mlp_readers_only: the edge latents a run of EdgeMPs leaves behind may be stored as the bf16 rows their readers round
    them to (rounded-bf16 mode) exactly where no UpEdgeMP projects them with fp32 arithmetic (edgeScalarToNodeVector, nn/blocks.py:420-430)
    before the level's next run replaces them."""
    from graphs4cfd_amd.nn.remus_gnn import NsRotEquiTreeScaleGNN as M
    prog = M._PROGRAM
    at = {name: k for k, (_, name, _) in enumerate(prog)}
    lvl = {name: l for _, name, l in prog}
    got = {name: M._mlp_readers_only(prog, at[name], lvl[name]) for name in ("mp114", "mp212", "mp34", "mp222", "mp124")}
    # mp114 -> down_mp12 (products) and up_mp21 (skip input); mp212 -> down_mp23, up_mp32 (skip); mp124 -> decoder: MLP operands only.
    # mp34 -> up_mp32 and mp222 -> up_mp21 project the latents of the coarse side: fp32 readers.
    assert got == {"mp114": True, "mp212": True, "mp34": False, "mp222": False, "mp124": True}, got
