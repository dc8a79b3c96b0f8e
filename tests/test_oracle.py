"""The oracle (oracle/g4c_oracle.py) against the golden vectors produced by the reference's own
source (tests/golden/make_golden.py) and against independent dense formulations."""
import pytest
import torch

from oracle import g4c_oracle as O

TOL = dict(rtol=1e-5, atol=2e-5)


def test_scatter_against_dense_and_hand_case(golden):
    c = golden("blocks.pt")["scatter"]
    src, idx = c["src"], c["index"]
    onehot = torch.zeros(7, src.size(0))
    onehot[idx, torch.arange(src.size(0))] = 1.0
    dense_sum = onehot @ src
    cnt = onehot.sum(1, keepdim=True).clamp(min=1)
    torch.testing.assert_close(O.scatter(src, idx, 7, "sum"), dense_sum, **TOL)
    torch.testing.assert_close(O.scatter(src, idx, 7, "mean"), dense_sum / cnt, **TOL)
    torch.testing.assert_close(O.scatter(src, idx, 7, "sum"), c["sum_7"], **TOL)
    torch.testing.assert_close(O.scatter(src, idx, 7, "mean"), c["mean_7"], **TOL)
    torch.testing.assert_close(O.scatter(src, idx, None, "mean"), c["mean_none"], **TOL)
    # hand-computed: targets 2,4,6 empty -> exactly 0 for mean
    m = O.scatter(src, idx, 7, "mean")
    assert torch.all(m[[2, 4, 6]] == 0)
    torch.testing.assert_close(m[0], (src[0] + src[1] + src[7]) / 3, **TOL)


def test_coalesce_hand_case():
    ei = torch.tensor([[2, 0, 2, 1, 0], [1, 1, 1, 0, 1]])
    ea = torch.tensor([[1.0], [2.0], [3.0], [4.0], [6.0]])
    oi, oa = O.coalesce(ei, ea, 3, "mean")
    assert oi.tolist() == [[0, 1, 2], [1, 0, 1]]
    assert oa.view(-1).tolist() == [4.0, 4.0, 2.0]
    oi, oa = O.coalesce(ei, ea, 3, "sum")
    assert oa.view(-1).tolist() == [8.0, 4.0, 4.0]


@pytest.mark.parametrize("i", range(8))
def test_mlp(golden, i):
    c = golden("blocks.pt")[f"mlp_{i}"]
    w = {f"m.{k}": v for k, v in c["weights"].items()}
    torch.testing.assert_close(O.mlp(c["x"], w, "m"), c["y"], **TOL)


@pytest.mark.parametrize("tag", ["h128_mean", "h32_sum", "h32_mean", "irregular"])
def test_gnblock(golden, tag):
    c = golden("blocks.pt")[f"gnblock_{tag}"]
    w = {f"b.{k}": v for k, v in c["weights"].items()}
    v, e = O.gn_block(c["v"], c["e"], c["edge_index"], w, "b", c["aggr"])
    torch.testing.assert_close(v, c["v_out"], **TOL)
    torch.testing.assert_close(e, c["e_out"], **TOL)


@pytest.mark.parametrize("tag", ["mean", "sum", "empty"])
def test_pool_edge(golden, tag):
    c = golden("blocks.pt")[f"pool_edge_{tag}"]
    ei, ea = O.pool_edge(c["idx"], c["edge_index"], c["edge_attr"], "sum" if tag == "sum" else "mean")
    assert torch.equal(ei, c["edge_index_out"])
    torch.testing.assert_close(ea, c["edge_attr_out"], **TOL)


@pytest.mark.parametrize("H", [32, 128])
def test_down_up(golden, H):
    c = golden("blocks.pt")[f"downup_h{H}"]
    g = c["graph"]
    wd = {f"d.{k}": v for k, v in c["down_weights"].items()}
    wu = {f"u.{k}": v for k, v in c["up_weights"].items()}
    f2, ei2, ea2 = O.down_mp(g, c["field1"], g["edge_index"], c["edge_attr1"], wd, "d", 1, torch.tanh)
    torch.testing.assert_close(f2, c["down_field"], **TOL)
    assert torch.equal(ei2, c["down_edge_index"])
    torch.testing.assert_close(ea2, c["down_edge_attr"], **TOL)
    f1 = O.up_mp(g, f2, c["field1"], wu, "u", 2, torch.tanh)
    torch.testing.assert_close(f1, c["up_field"], **TOL)


def test_remus_blocks(golden):
    b = golden("blocks.pt")
    g = b["remus_graph"]
    c = b["edgemp"]
    w = {f"b.{k}": v for k, v in c["weights"].items()}
    e, a = O.edge_mp(c["e"], c["a"], c["angle_index"], w, "b")
    torch.testing.assert_close(e, c["e_out"], **TOL)
    torch.testing.assert_close(a, c["a_out"], **TOL)
    c = b["downedgemp"]
    w = {f"b.{k}": v for k, v in c["weights"].items()}
    torch.testing.assert_close(O.down_edge_mp(c["e1"], c["e2"], c["a12"], c["angle_index12"], w, "b"), c["e2_out"], **TOL)
    c = b["upedgemp_21"]
    w = {f"b.{k}": v for k, v in c["weights"].items()}
    out = O.up_edge_mp(g["pos"], g["y_idx_21"], g["x_idx_21"], g["weights_21"], c["edge_attr2"], g["edge_index2"],
                       g["edgeUnitVectorInverse2"], g["coarse_mask2"], c["edge_attr1"], g["edge_index"],
                       g["edgeUnitVector"], w, "b")
    torch.testing.assert_close(out, c["e1_out"], **TOL)
    c2 = b["upedgemp_32"]
    out = O.up_edge_mp(g["pos"], g["y_idx_32"], g["x_idx_32"], g["weights_32"], c2["edge_attr3"], g["edge_index3"],
                       g["edgeUnitVectorInverse3"], g["coarse_mask3"], c2["edge_attr2"], g["edge_index2"],
                       g["edgeUnitVector2"], w, "b", g["coarse_mask2"])
    torch.testing.assert_close(out, c2["e2_out"], **TOL)


def test_remus_helpers(golden):
    b = golden("blocks.pt")
    g = b["remus_graph"]
    c = b["es2nv"]
    torch.testing.assert_close(O.edge_scalar_to_node_vector(c["s1"], g["edge_index"], unit_inv=g["edgeUnitVectorInverse"]),
                               c["v1"], **TOL)
    torch.testing.assert_close(O.edge_scalar_to_node_vector(c["sH"], g["edge_index2"], unit_inv=g["edgeUnitVectorInverse2"],
                                                            coarse_mask=g["coarse_mask2"]), c["vH"], **TOL)
    torch.testing.assert_close(O.edge_scalar_to_node_vector(c["s1"], g["edge_index"], unit=g["edgeUnitVector"]),
                               c["v1_lstsq"], rtol=1e-4, atol=1e-4)
    with pytest.raises(AssertionError):
        O.edge_scalar_to_node_vector(c["s1"], g["edge_index"])
    c = b["knn_interpolate"]
    torch.testing.assert_close(O.knn_interpolate(c["x"], g["y_idx_21"], g["x_idx_21"], g["weights_21"]), c["y"], **TOL)
    c = b["restriction"]
    assert torch.equal(O.restriction(int(g["coarse_mask2"].sum()), g["coarse_mask2"], g["edge_index2"], g["pos"].size(0)),
                       c["edge_index_out"])


@pytest.mark.parametrize("cls", sorted(O.MUS_PROGRAMS))
def test_mus_models(golden, cls):
    c = golden("models_mus.pt")[cls]
    nf = c["arch"]["decoder"][1][-1]
    torch.testing.assert_close(O.mus_forward(cls, c["graph"], c["weights"], nf), c["forward"], rtol=1e-4, atol=5e-5)
    torch.testing.assert_close(O.mus_solve(cls, c["graph"], c["weights"], 3, nf), c["solve3"], rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("cls", ["NsFourScaleGNN", "NsTwoScaleGNN"])
def test_mus_models_3d(golden, cls):
    """3-D meshes (BASELINE config 5's data path): 3-wide edge attributes through the same programs."""
    c = golden("models_mus_3d.pt")[cls]
    assert c["graph"]["pos"].shape[1] == 3 and c["graph"]["edge_attr"].shape[1] == 3
    torch.testing.assert_close(O.mus_forward(cls, c["graph"], c["weights"], 3), c["forward"], rtol=1e-4, atol=5e-5)
    torch.testing.assert_close(O.mus_solve(cls, c["graph"], c["weights"], 3, 3), c["solve3"], rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("cls", sorted(O.MUGS_PROGRAMS))
def test_mugs_models(golden, cls):
    """gMuS-GNN family against the reference's own forward / solve on graphs from its own transforms."""
    c = golden("models_mugs.pt")[cls]
    torch.testing.assert_close(O.mugs_forward(cls, c["graph"], c["weights"], 3), c["forward"], rtol=1e-4, atol=5e-5)
    torch.testing.assert_close(O.mugs_solve(cls, c["graph"], c["weights"], 3, 3), c["solve3"], rtol=1e-4, atol=2e-4)


def test_remus_model(golden):
    c = golden("model_remus.pt")
    torch.testing.assert_close(O.remus_forward(c["graph"], c["weights"]), c["forward"], rtol=1e-4, atol=5e-5)
    torch.testing.assert_close(O.remus_solve(c["graph"], c["weights"], 3), c["solve3"], rtol=1e-4, atol=2e-4)


def test_training_steps_against_reference_autograd(golden):
    """the oracle's loss and parameter gradients for two rollout steps of GNN.fit's inner loop against the
    reference's own (train-mode forward + GraphLoss + backward, recorded by make_golden.py gen_training)."""
    c = golden("training.pt")
    steps = O.training_steps("NsThreeScaleGNN", c["graph"], c["weights"], 2, 3, c["lambda_d"])
    for (loss, pred, grads), ref in zip(steps, c["steps"]):
        torch.testing.assert_close(loss, ref["loss"], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(pred, ref["pred"], rtol=1e-4, atol=5e-5)
        assert set(grads) == set(ref["grads"])
        for k in grads:
            scale = float(ref["grads"][k].abs().max())
            assert float((grads[k] - ref["grads"][k]).abs().max()) <= 1e-4 * scale + 1e-9, k
        norm = float(torch.sqrt(sum((v ** 2).sum() for v in grads.values())))
        assert abs(norm - ref["grad_norm2"]) <= 1e-4 * ref["grad_norm2"]


def test_rollouts(golden):
    r = golden("rollout.pt")
    c = r["two_scale"]
    torch.testing.assert_close(O.mus_solve("NsTwoScaleGNN", c["graph"], c["weights"], 1, 3), c["solve1"], rtol=1e-4, atol=5e-5)
    torch.testing.assert_close(O.mus_solve("NsTwoScaleGNN", c["graph"], c["weights"], 5, 3), c["solve5"], rtol=1e-4, atol=2e-4)
    s50 = O.mus_solve("NsTwoScaleGNN", c["graph"], c["weights"], 50, 3)
    # fp32 differences compound over an autoregressive rollout: tolerance stated per length
    torch.testing.assert_close(s50[:, :30], c["solve50"][:, :30], rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(s50, c["solve50"], rtol=1e-2, atol=1e-2)
    c = r["one_scale_nin2"]
    torch.testing.assert_close(O.mus_solve("NsOneScaleGNN", c["graph"], c["weights"], 4, 3), c["solve4"], rtol=1e-4, atol=2e-4)
    with pytest.raises(AssertionError):
        O.mus_solve("NsOneScaleGNN", c["graph"], c["weights"], 0, 3)


def test_collate_of_a_list_of_graphs_matches_the_reference_batch(golden):
    """nn.model.collate (what solve([g1, g2]) runs on; reference nn/model.py:308-309 -> Batch.from_data_list) on the fixture's two
    graphs: the oracle's rollout of OUR batch graph reproduces the reference's solve() of the list — index tensors offset by the
    running node count, everything else concatenated along dim 0, `batch` = graph id per node."""
    import graphs4cfd_amd as gfd
    from graphs4cfd_amd.nn.model import collate
    c = golden("solve_list.pt")
    graphs = [gfd.Graph(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in d.items()}) for d in c["graphs"]]
    b = collate(graphs)
    n1, n2 = graphs[0].num_nodes, graphs[1].num_nodes
    e1 = graphs[0].edge_index.size(1)
    assert b.num_nodes == n1 + n2 and torch.equal(b.batch, torch.cat([torch.zeros(n1), torch.ones(n2)]).long())
    assert torch.equal(b.edge_index[:, :e1], graphs[0].edge_index) and torch.equal(b.edge_index[:, e1:], graphs[1].edge_index + n1)
    assert torch.equal(b.field, torch.cat([graphs[0].field, graphs[1].field]))
    w = {k: v for k, v in c["weights"].items()}
    ref = O.mus_solve("NsOneScaleGNN", b.to_dict(), w, 3, 3)
    torch.testing.assert_close(ref, c["solve3"], rtol=1e-5, atol=1e-5)
