"""gfd.DataLoader / Collater (reference: graphs4cfd/loader.py:7-75): batches of Graphs, REMuS angle-index correction."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphs4cfd_amd as gfd                       # noqa: E402
from graphs4cfd_amd import synthetic as S          # noqa: E402


def test_collater_offsets_node_and_edge_indexed_attributes():
    """loader.py:17-55: `edge_index*` address nodes (offset by the running node count); `angle_index{l}` address the edges of level l
    and `angle_index{l}{l+1}` the edges of level l (row) / l+1 (col): offset by the running EDGE counts of those levels."""
    graphs = [S.remus_graph(220 + 30 * i, k=5, seed=70 + i) for i in range(3)]
    keep = [g.clone() for g in graphs]
    batch = gfd.Collater()(graphs)
    for g, k in zip(graphs, keep):                      # the inputs are left alone (a dataset held in memory is collated every epoch)
        assert all(torch.equal(getattr(g, a), getattr(k, a)) for a in ("angle_index", "angle_index2", "angle_index12", "angle_index23"))
    assert torch.equal(gfd.Collater()(graphs).angle_index12, batch.angle_index12)
    n_off = e1 = e2 = e3 = 0
    pieces = {k: [] for k in ("edge_index", "edge_index2", "angle_index", "angle_index2", "angle_index3", "angle_index12", "angle_index23")}
    for g in keep:
        pieces["edge_index"].append(g.edge_index + n_off)
        pieces["edge_index2"].append(g.edge_index2 + n_off)
        pieces["angle_index"].append(g.angle_index + e1)
        pieces["angle_index2"].append(g.angle_index2 + e2)
        pieces["angle_index3"].append(g.angle_index3 + e3)
        pieces["angle_index12"].append(g.angle_index12 + torch.tensor([[e1], [e2]]))
        pieces["angle_index23"].append(g.angle_index23 + torch.tensor([[e2], [e3]]))
        n_off += g.num_nodes
        e1 += g.edge_index.size(1); e2 += g.edge_index2.size(1); e3 += g.edge_index3.size(1)
    for k, v in pieces.items():
        assert torch.equal(getattr(batch, k), torch.cat(v, 1)), k
    assert batch.num_nodes == n_off and torch.equal(batch.batch, torch.repeat_interleave(torch.arange(3), torch.tensor([g.num_nodes for g in keep])))
    assert torch.equal(batch.field, torch.cat([g.field for g in keep]))


def test_dataloader_applies_batch_transform_and_keeps_targets():
    data = []
    for i in range(5):
        g = S.mus_graph(150, levels=1, seed=i)
        g.target = torch.randn(150, 6)
        data.append(g)
    seen = []
    loader = gfd.DataLoader(data, batch_size=2, shuffle=False, transform=lambda b: (seen.append(b.num_nodes), b)[1])
    sizes = [b.num_nodes for b in loader]
    assert sizes == [300, 300, 150] and seen == sizes
    # worker processes (the reference's scripts use num_workers=4): graphs and batch transforms cross the process boundary
    workers = gfd.DataLoader(data, batch_size=2, shuffle=False, transform=gfd.transforms.GridClustering([0.3]), num_workers=2)
    assert [(b.num_nodes, hasattr(b, "cluster_2")) for b in workers] == [(300, True), (300, True), (150, True)]
    b = next(iter(loader))
    assert b.target.shape == (300, 6) and int(b.edge_index.max()) == 299 and int(b.edge_index[:, :900].max()) == 149


def test_knn_interp_weights_as_a_batch_transform():
    """BuildKnnInterpWeights on a collated batch (examples/training/NsREMuSGNN/NsRotEquiTreeScaleGNN.py:39-41): neighbours are
    searched per graph; indices address the batch's compact level lists, i.e. each graph's own indices shifted by the number
    of level-l / level-(l-1) nodes before it."""
    graphs = [S.remus_graph(200 + 40 * i, k=5, seed=90 + i) for i in range(3)]
    batch = gfd.Collater(gfd.transforms.BuildKnnInterpWeights(5))(graphs)
    for hi, lo in ((2, 1), (3, 2)):
        off_c = off_f = 0
        ys, xs, ws = [], [], []
        for g in graphs:
            ys.append(getattr(g, f"y_idx_{hi}{lo}") + off_f); xs.append(getattr(g, f"x_idx_{hi}{lo}") + off_c); ws.append(getattr(g, f"weights_{hi}{lo}"))
            off_c += int(getattr(g, f"coarse_mask{hi}").sum())
            off_f += g.num_nodes if lo == 1 else int(getattr(g, f"coarse_mask{lo}").sum())
        assert torch.equal(getattr(batch, f"y_idx_{hi}{lo}"), torch.cat(ys))
        assert torch.equal(getattr(batch, f"x_idx_{hi}{lo}"), torch.cat(xs))
        torch.testing.assert_close(getattr(batch, f"weights_{hi}{lo}"), torch.cat(ws))


# ------------------------------------------------------------------ scaling / augmentation transforms against the reference's
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "augment.pt")


def _same(got: gfd.Graph, ref: dict, what: str):
    for k, v in ref.items():
        if torch.is_tensor(v):
            g = getattr(got, k)
            assert g.shape == v.shape, (what, k)
            if v.is_floating_point():
                torch.testing.assert_close(g, v, rtol=1e-5, atol=1e-6, msg=lambda m: f"{what}: {k}: {m}")
            else:
                assert torch.equal(g, v), (what, k)


def test_scale_rotate_flip_subset_match_the_reference():
    """tests/golden/augment.pt (make_golden.py gen_augment): the reference's ScaleNs -> GraphRotation -> flip_graph_dim ->
    NodeSubset chain on a MuS graph with a two-step history and a three-step target; rotation of an advection graph (`loc`),
    of a REMuS graph (edge unit vectors and their pseudo-inverses, `edge_attr` untouched) and a 3-D Tait-Bryan rotation."""
    c = torch.load(GOLD, weights_only=False)
    T = gfd.transforms
    stages = c["mus_uvp"]["stages"]
    g = gfd.Graph(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in stages[0][1].items()})
    ops = {"ScaleNs": T.ScaleNs({'u': (-2.1, 2.6), 'v': (-2.25, 2.1), 'p': (-3.7, 2.35), 'Re': (500, 1000)}, format='uvp'),
           "GraphRotation": T.GraphRotation(37.0, eq='ns', format='uvp'),
           "flip_y": lambda gr: T.flip_graph_dim(gr, 1, eq='ns', format='uvp'),
           "NodeSubset": T.NodeSubset(list(range(0, 120, 3)))}
    for name, ref in stages[1:]:
        g = ops[name](g)
        keys = ("pos", "field", "target", "glob", "omega") if name == "NodeSubset" else ref.keys()    # (NodeSubset: node attributes only)
        _same(g, {k: ref[k] for k in keys}, name)
    for case, kw in (("adv", dict(eq='adv')), ("remus_uv", dict(eq='ns', format='uv')), ("rot3d", {})):
        e = c[case]
        g = gfd.Graph(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in e["input"].items()})
        _same(T.GraphRotation(e["theta"], **kw)(g), e["output"], case)
    with pytest.raises(ValueError):
        T.flip_graph_dim(gfd.Graph(**c["remus_uv"]["input"]), 0, eq='ns', format='uv')


def test_random_transforms_keep_shapes_and_ranges():
    torch.manual_seed(0)
    g = S.mus_graph(200, levels=1, seed=1)
    g.target = torch.randn(200, 3)
    f0 = g.field.clone()
    g = gfd.transforms.AddUniformNoise(0.01)(g)
    assert float((g.field - f0).abs().max()) <= 0.01 and float((g.field - f0).abs().max()) > 0
    speed = g.field[:, :2].norm(dim=1).clone()
    g = gfd.transforms.RandomGraphRotation(eq='ns', format='uvp')(g)
    g = gfd.transforms.RandomGraphFlip(eq='ns', format='uvp')(g)
    torch.testing.assert_close(g.field[:, :2].norm(dim=1), speed, rtol=1e-5, atol=1e-6)      # rotations / mirrors keep |u|
    g2 = gfd.Graph(pos=torch.rand(100, 2), field=torch.randn(100, 3), glob=torch.rand(100, 1))
    assert gfd.transforms.RandomNodeSubset(0.8)(g2).num_nodes == 80
    assert gfd.transforms.RandomNodeSubset(30)(g2).pos.shape == (30, 2)


def test_periodic_knn_connect_matches_the_reference():
    """ConnectKNN on periodic domains (a given period per axis, or "auto") and the Guillard transform with a periodic axis,
    against the reference's outputs (neighbour search on the (cos, sin) embedding, wrapped edge vectors)."""
    c = torch.load(GOLD, weights_only=False)
    for name in ("per_y_auto", "per_xy"):
        e = c[name]
        g = gfd.transforms.ConnectKNN(6, period=e["period"])(gfd.Graph(pos=e["pos"].clone()))
        assert torch.equal(g.edge_index, e["edge_index"]), name
        torch.testing.assert_close(g.edge_attr, e["edge_attr"], rtol=1e-5, atol=1e-6)
    e = c["guillard_periodic"]
    g = gfd.transforms.GuillardCoarseningAndConnectKNN(k=(6, 6, 6), period=(None, "auto"), scale_edge_attr=(0.1, 0.25, 0.5))(gfd.Graph(pos=e["pos"].clone()))
    _same(g, e["graph"], "guillard_periodic")
    e = c["remus_periodic"]
    g = gfd.transforms.BuildRemusGraph(num_levels=3, k=5, period=(None, "auto"), scale_edge_length=(0.1, 0.2, 0.4))(gfd.Graph(pos=e["pos"].clone()))
    _same(g, e["graph"], "remus_periodic")


def test_r2_metric():
    torch.manual_seed(3)
    t = torch.randn(50, 6)
    assert gfd.metrics.r2(t, t) == 1.0
    p = t + 0.1 * torch.randn_like(t)
    ref = 1 - float(((t - p) ** 2).sum() / ((t - t.mean()) ** 2).sum())
    assert abs(gfd.metrics.r2(p, t) - ref) < 1e-6
    with pytest.raises(RuntimeError):
        gfd.metrics.r2(t[None], t[None])


# ------------------------------------------------------------------ datasets
def test_dataset_record_layouts_match_the_reference(tmp_path):
    """Adv / NsCircle / NsEllipse `data2graph` (datasets.py:158-337) on NaN-padded records, against the reference's own output;
    then the storage side: the same records through a .npy file (memory-mapped and preloaded), `get_sequence`, `__getitem__`,
    `idx=`, the transform hook and the loud failure for HDF5 without h5py."""
    import numpy as np
    c = torch.load(GOLD, weights_only=False)["datasets"]["cases"]
    D = gfd.datasets
    classes = {"Adv": (D.Adv, {}), "NsCircle_uvp": (D.NsCircle, {"format": "uvp"}), "NsCircle_uv": (D.NsCircle, {"format": "uv"}),
               "NsEllipse_uv": (D.NsEllipse, {"format": "uv"}), "NsEllipse_uvp": (D.NsEllipse, {"format": "uvp"})}
    info = {"n_in": 2, "n_out": 3, "step": 2, "T": 12}
    for name, e in c.items():
        cls, kw = classes[name]
        stack = torch.stack([e["record"], e["record"].flip(1)])                  # two "simulations"
        ds = cls(data=stack, training_info=info, **kw)
        assert len(ds) == 2 and ds.training_sequences_length == e["length"]
        g = ds.data2graph(e["record"], *e["args"])
        _same(g, e["graph"], name)
        _same(ds.get_sequence(0, 1, n_in=2, n_out=3, step=2), e["graph"], name + " get_sequence")
    # storage: .npy file, lazily and preloaded; one simulation only; transforms
    e = c["NsCircle_uvp"]
    path = os.path.join(tmp_path, "ns.npy")
    np.save(path, torch.stack([e["record"], e["record"]]).numpy())
    seen = []
    for preload in (False, True):
        ds = D.NsCircle("uvp", path=path, training_info=info, preload=preload, transform=lambda g: seen.append(g.num_nodes))
        assert len(ds) == 2
        _same(ds.get_sequence(1, 1, n_in=2, n_out=3, step=2), e["graph"], f"npy preload={preload}")
        s = ds[0]
        assert s.field.shape == (40, 6) and s.target.shape == (40, 9) and s.omega.shape == (40, 1)
    assert seen == [40] * 4
    one = D.NsCircle("uvp", path=path, training_info=info, idx=1, preload=True)
    assert len(one) == 1
    with pytest.raises(ValueError):
        D.NsCircle("uvp", path=path, training_info=info, idx=1)
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError):
            len(D.NsCircle("uvp", path=os.path.join(tmp_path, "x.h5"), training_info=info))


def test_interpolate_nodes_matches_the_reference(tmp_path):
    """InterpolateNodes (scipy griddata: cubic for the fields, linear + threshold / rounding for omega / bound) against the
    reference's output, and InterpolateNodesToXml reading the vertices of a NekMesh-style xml file."""
    e = torch.load(GOLD, weights_only=False)["interpolate_nodes"]
    mk = lambda: gfd.Graph(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in e["input"].items()})
    _same(gfd.transforms.InterpolateNodes(e["new_pos"])(mk()), e["output"], "interpolate_nodes")
    xml = os.path.join(tmp_path, "mesh.xml")
    with open(xml, "w") as f:
        f.write("<NEKTAR><GEOMETRY><VERTEX>" + "".join(f'<V ID="{i}">{float(p[0])!r} {float(p[1])!r} 0.0</V>' for i, p in enumerate(e["new_pos"]))
                + "</VERTEX></GEOMETRY></NEKTAR>")
    _same(gfd.transforms.InterpolateNodesToXml(xml)(mk()), e["output"], "interpolate_nodes via xml")
    g = mk()
    g.edge_index = torch.zeros(2, 1, dtype=torch.long)
    with pytest.raises(ValueError):
        gfd.transforms.InterpolateNodes(e["new_pos"])(g)
