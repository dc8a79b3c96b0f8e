"""`GridClustering` (transforms/mus.py:9-65) with the node positions resident on the GPU — voxel
ids and their sorted unique set as device tensors, cluster centres through `g4c_segment_reduce` — against the reference
transform's own outputs (tests/golden/transforms.pt) and, at the headline mesh size, against the host path that the CPU
suite pins to those fixtures.  Integer outputs bit-equal; positions / relative positions bit-equal with the host path
(same summation order, true divisions) and within the fixture tolerance of tests/test_synthetic.py against the golden."""
import pytest
import torch

from graphs4cfd_amd import synthetic as S
from graphs4cfd_amd.graph import Graph

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _levels(g, n_levels):
    names = []
    for lvl in range(2, n_levels + 2):
        names += [f"pos_{lvl}", f"cluster_{lvl}", f"mask_{lvl}", f"idx{lvl - 1}_to_idx{lvl}", f"e_{lvl - 1}{lvl}"]
    return names


@pytest.mark.parametrize("tag", ["mus_2d", "mus_3d"])
def test_grid_clustering_on_device_vs_reference_outputs(golden, tag):
    c = golden("transforms.pt")[tag]
    ref = c["graph"]
    g = S.add_grid_levels(Graph(pos=ref["pos"].to(DEV)), c["cells"])
    for name in _levels(g, len(c["cells"])):
        got, want = getattr(g, name), ref[name]
        assert got.device.type == "cuda", name
        if want.dtype == torch.int64:
            assert torch.equal(got.cpu(), want), f"{tag}.{name}"
        else:
            torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=1e-6, msg=lambda m: f"{tag}.{name}: {m}")


@pytest.mark.parametrize("n,dim,levels", [(100_000, 2, 3), (60_000, 3, 3), (1, 2, 1), (7, 3, 2)])
def test_grid_clustering_on_device_equals_host_path(n, dim, levels):
    pos = torch.rand(n, dim, generator=torch.Generator().manual_seed(n + dim))
    cells = S.default_cells(max(n, 8), dim, levels + 1)
    host = S.add_grid_levels(Graph(pos=pos.clone()), cells)
    dev = S.add_grid_levels(Graph(pos=pos.to(DEV)), cells)
    for name in _levels(host, levels):
        a, b = getattr(dev, name).cpu(), getattr(host, name)
        assert a.dtype == b.dtype and a.shape == b.shape, name
        assert torch.equal(a, b), f"{name}: {(a.double() - b.double()).abs().max().item()}"


# ------------------------------------------------------------------ ConnectKNN (transforms/connect.py:9-92) on the device
@pytest.mark.parametrize("tag", ["mus_2d", "mus_3d"])
def test_connect_knn_on_device_vs_reference_outputs(golden, tag):
    """The reference transform's own edges (k-d tree / torch_cluster.knn on the host) from the cell-grid search
    (`g4c_knn_grid`): edge_index bit-equal, edge_attr equal."""
    c = golden("transforms.pt")[tag]
    ref = c["graph"]
    edge_index, edge_attr = S.connect_knn(ref["pos"].to(DEV), c["k"])
    assert edge_index.device.type == "cuda" and edge_index.dtype == torch.int64
    assert torch.equal(edge_index.cpu(), ref["edge_index"]), tag
    torch.testing.assert_close((edge_attr / (2 * c["r"])).cpu(), ref["edge_attr"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("n,dim,k", [(100_000, 2, 6), (200_000, 3, 6), (5_000, 2, 5), (9, 2, 6), (3_000, 3, 12), (40_000, 2, 16)])
def test_connect_knn_on_device_equals_host_path(n, dim, k):
    """Bit-equal edges with the host path (pinned to the reference's outputs by tests/test_synthetic.py) at the
    headline mesh size, in 3-D, on a thin strip (elongated cell grid), with barely more points than neighbours
    and at the largest k of the kernel."""
    pos = torch.rand(n, dim, generator=torch.Generator().manual_seed(7 * n + k))
    if n == 5_000:
        pos[:, 1] *= 0.02
    ei_h, ea_h = S.connect_knn(pos.clone(), k)
    ei_d, ea_d = S.connect_knn(pos.to(DEV), k)
    assert ei_d.shape == (2, n * k)
    assert torch.equal(ei_d.cpu(), ei_h)
    assert torch.equal(ea_d.cpu(), ea_h)


@pytest.mark.parametrize("n,k,period", [(50_000, 6, (None, "auto")), (50_000, 6, ("auto", None)), (8_000, 5, (2.5, None)), (300, 6, (None, 1.0))])
def test_connect_knn_periodic_axis_on_device_equals_host_path(n, k, period):
    """One periodic axis of a 2-D cloud (transforms/connect.py:38-71: the axis enters the search as a point on a circle; its edge
    components are wrapped): the device path (3-D cell grid on the embedding + float64 re-ranking of 2k + 2 candidates) gives the host
    path's edges — which tests/test_synthetic.py pins to the reference's outputs — and the same wrapped edge attributes."""
    pos = torch.rand(n, 2, generator=torch.Generator().manual_seed(n + k)) * torch.tensor([2.5, 1.0])
    ei_h, ea_h = S.connect_knn(pos.clone(), k, period=period)
    ei_d, ea_d = S.connect_knn(pos.to(DEV), k, period=period)
    assert ei_d.device.type == "cuda" and ei_d.shape == (2, n * k)
    assert torch.equal(ei_d.cpu(), ei_h)
    torch.testing.assert_close(ea_d.cpu(), ea_h, rtol=0, atol=1e-6)
    ax = 0 if period[0] is not None else 1
    d = float(pos[:, ax].max() - pos[:, ax].min()) if period[ax] == "auto" else float(period[ax])
    assert float(ea_d[:, ax].abs().max()) <= d / 2 + 1e-6             # wrapped


@pytest.mark.parametrize("n,dim,k,period", [(40_000, 2, 6, ("auto", "auto")), (9_000, 2, 5, (2.5, 1.0)), (30_000, 3, 6, (None, "auto", None)),
                                            (30_000, 3, 6, ("auto", None, "auto")), (20_000, 3, 4, ("auto", "auto", "auto")), (60, 2, 6, ("auto", "auto"))])
def test_connect_knn_any_periodic_axes_on_device_equals_host_path(n, dim, k, period):
    """Two periodic axes of a 2-D cloud, one to three periodic axes of a 3-D cloud (transforms/connect.py:38-71 embeds them in 4 to 6
    dimensions): the device path searches the raw coordinates over the cloud + ghost copies near the periodic faces and orders
    up to 16 candidates by their float64 embedded distances — the host path's edges and wrapped attributes.  (60 points: too few for
    ghosts, the host path takes the device tensor.)  A graded cloud makes the margin check grow the ghost band."""
    g = torch.Generator().manual_seed(n + k + dim)
    pos = torch.rand(n, dim, generator=g)
    pos[:, 0] = pos[:, 0] ** 2                                # (graded: dense near x = 0, coarse near x = 1)
    pos = pos * torch.tensor([2.5, 1.0, 1.5][:dim])
    ei_h, ea_h = S.connect_knn(pos.clone(), k, period=period)
    ei_d, ea_d = S.connect_knn(pos.to(DEV), k, period=period)
    assert ei_d.shape == (2, n * k) and (ei_d.is_cuda or n < 1000)
    assert torch.equal(ei_d.cpu(), ei_h)
    torch.testing.assert_close(ea_d.cpu(), ea_h, rtol=0, atol=1e-6)
    for ax in range(dim):
        if period[ax] is not None:
            d = float(pos[:, ax].max() - pos[:, ax].min()) if period[ax] == "auto" else float(period[ax])
            assert float(ea_d[:, ax].abs().max()) <= d / 2 + 1e-6


def test_connect_knn_on_device_clustered_cloud():
    """A strongly non-uniform cloud (most cells empty, a few crowded: the ring search has to widen): still exact."""
    g = torch.Generator().manual_seed(3)
    centres = torch.rand(20, 2, generator=g)
    pos = (centres[torch.randint(0, 20, (30_000,), generator=g)] + 0.002 * torch.randn(30_000, 2, generator=g)).float()
    ei_h, _ = S.connect_knn(pos.clone(), 6)
    ei_d, _ = S.connect_knn(pos.to(DEV), 6)
    assert torch.equal(ei_d.cpu(), ei_h)


def test_knn_grid_rejects_bad_arguments():
    with pytest.raises(ValueError, match="g4c_knn_grid"):
        S.knn_neighbours_device(torch.rand(5, 2, device=DEV), 6)      # not more points than neighbours
    with pytest.raises(ValueError, match="g4c_knn_grid"):
        S.knn_neighbours_device(torch.rand(100, 2, device=DEV), 17)   # k beyond the kernel's register budget


# ------------------------------------------------- get_knn_interpolate_weights (transforms/interpolate.py:110-131)
@pytest.mark.parametrize("n_x,n_y,dim,k", [(25_000, 100_000, 2, 3), (100_000, 30_000, 3, 4), (4, 50, 2, 4), (2_000, 0, 2, 3)])
def test_knn_interp_weights_on_device_equals_host_path(n_x, n_y, dim, k):
    """Queries from a second cloud (coarse -> fine interpolation of gMuS / REMuS), some of them outside the bounding
    box of the searched cloud, fewer points than one cell ring, and no queries at all: indices bit-equal, weights
    (torch ops on the same pairs) to the last bits."""
    g = torch.Generator().manual_seed(n_x + n_y)
    pos_x = torch.rand(n_x, dim, generator=g)
    pos_y = torch.rand(n_y, dim, generator=g) * 1.2 - 0.1
    y_h, x_h, w_h = S.knn_interp_weights(pos_x, pos_y, k)
    y_d, x_d, w_d = S.knn_interp_weights(pos_x.to(DEV), pos_y.to(DEV), k)
    assert x_d.device.type == "cuda"
    assert torch.equal(y_d.cpu(), y_h) and torch.equal(x_d.cpu(), x_h)
    torch.testing.assert_close(w_d.cpu(), w_h, rtol=1e-6, atol=0.0)     # torch's own sum / reciprocal on either side


# ------------------------------------------------- REMuS tables (transforms/remus.py:9-45, 151-176) from device tensors
def test_remus_angle_tables_on_device_equal_host_path():
    """`extend_graph` and `angleIndexDownMP` are index arithmetic + elementwise ops: with the edges on the GPU the angle
    indices are bit-equal to the host path's (itself pinned to the reference transform by tests/test_synthetic.py), the
    angle attributes (norms, cos, sin through torch ops on either side) equal to 1e-6."""
    k = 5
    pos = torch.rand(20_000, 2, generator=torch.Generator().manual_seed(11))
    ei_h, ea_h = S.connect_knn(pos, k)
    ei_d, ea_d = S.connect_knn(pos.to(DEV), k)
    assert torch.equal(ei_d.cpu(), ei_h)
    u_h, ai_h, aa_h = S.extend_graph(ei_h, ea_h, k)
    u_d, ai_d, aa_d = S.extend_graph(ei_d, ea_d, k)
    assert ai_d.device.type == "cuda" and torch.equal(ai_d.cpu(), ai_h)
    torch.testing.assert_close(u_d.cpu(), u_h, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(aa_d.cpu(), aa_h, rtol=1e-5, atol=1e-6)
    # a coarse level: every third node, kNN among them, the down-MP angle table between the two levels
    coarse = torch.arange(0, pos.size(0), 3)
    ei2_h, ea2_h = S.connect_knn(pos[coarse], k)
    ei2_h = coarse[ei2_h]                                   # coarse edges in level-1 numbering (transforms/remus.py:120-128)
    di_h, da_h = S.angle_index_down(ei_h, ea_h, ei2_h, ea2_h, coarse, k)
    di_d, da_d = S.angle_index_down(ei_d, ea_d, ei2_h.to(DEV), ea2_h.to(DEV), coarse.to(DEV), k)
    assert torch.equal(di_d.cpu(), di_h)
    torch.testing.assert_close(da_d.cpu(), da_h, rtol=1e-5, atol=1e-6)


# ------------------------------------------------- the transform classes and the model on a device-built graph
@pytest.mark.parametrize("tag", ["mus_2d", "mus_3d"])
def test_transform_classes_on_device_reproduce_the_reference_pipeline(golden, tag):
    """`Compose([ConnectKNN, ScaleEdgeAttr, GridClustering])` (examples/training/NsMuSGNN/*.py) applied to a Graph whose
    positions are on the GPU == the attributes the reference's own pipeline produced."""
    import graphs4cfd_amd as gfd
    T = gfd.transforms
    c = golden("transforms.pt")[tag]
    g = T.Compose([T.ConnectKNN(c["k"]), T.ScaleEdgeAttr(c["r"]), T.GridClustering(c["cells"])])(Graph(pos=c["graph"]["pos"].to(DEV)))
    for name, want in c["graph"].items():
        got = getattr(g, name)
        assert got.device.type == "cuda", name
        if want.dtype == torch.int64:
            assert torch.equal(got.cpu(), want), f"{tag}.{name}"
        else:
            torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=1e-6, msg=lambda m: f"{tag}.{name}: {m}")


def test_device_built_mesh_is_the_host_built_mesh_and_feeds_the_model():
    """`mus_graph(device=...)`: kNN edges, scaling and three grid levels built on the GPU are bit-equal to the host-built
    mesh, and the model's rollout from it is the rollout from the uploaded host mesh, bit for bit."""
    import graphs4cfd_amd as gfd
    host = S.mus_graph(30_000, levels=3, seed=5)
    dev = S.mus_graph(30_000, levels=3, seed=5, device=DEV)
    for name, want in host.to_dict().items():
        got = getattr(dev, name)
        assert got.device.type == "cuda" and torch.equal(got.cpu(), want), name
    torch.manual_seed(2)
    model = gfd.nn.NsThreeScaleGNN(arch=S.mus_arch("NsThreeScaleGNN", 128), device=DEV)
    with torch.no_grad():
        a = model.solve(dev, 3)
        b = model.solve(host.clone().to(DEV), 3)
    assert torch.isfinite(a).all() and torch.equal(a, b)


def test_device_built_remus_graph_is_the_host_built_one_and_feeds_the_model():
    """`remus_graph(device=...)` / `BuildRemusGraph` on a device Graph: index tensors bit-equal to the host-built graph (which
    tests/test_synthetic.py pins to the reference transform's output), float tables (norms, cos / sin, weights: torch ops on
    either side) to 1e-5, and REMuS-GNN's forward from it equal to the forward from the uploaded host graph at the FWD
    tolerance of test_gpu_parity.py."""
    import graphs4cfd_amd as gfd
    host = S.remus_graph(20_000, k=5, seed=21)
    dev = S.remus_graph(20_000, k=5, seed=21, device=DEV)
    for name, want in host.to_dict().items():
        got = getattr(dev, name)
        assert got.device.type == "cuda", name
        if want.dtype in (torch.int64, torch.bool):
            assert torch.equal(got.cpu(), want), name
        elif "Inverse" in name:
            torch.testing.assert_close(got.cpu(), want, rtol=1e-4, atol=1e-5, msg=lambda m: f"{name}: {m}")
        else:
            torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=1e-6, msg=lambda m: f"{name}: {m}")
    built = gfd.transforms.BuildRemusGraph(3, 5, scale_edge_length=tuple(2.0 * 20_000 ** -0.5 * f for f in (1, 2, 4)))(
        Graph(pos=host.pos.to(DEV)))
    assert built.angle_index12.device.type == "cuda" and torch.equal(built.angle_index12.cpu(), host.angle_index12)
    torch.manual_seed(22)
    model = gfd.nn.NsRotEquiTreeScaleGNN(arch=S.remus_arch(128), device=DEV)
    with torch.no_grad():
        a = model.forward(dev)
        b = model.forward(host.clone().to(DEV))
    torch.testing.assert_close(a, b, rtol=5e-4, atol=5e-4)


@pytest.mark.parametrize("cls,levels", [("NsTwoGuillardScaleGNN", 2), ("NsThreeGuillardScaleGNN", 3), ("NsFourGuillardScaleGNN", 4)])
def test_gmus_transforms_on_device_reproduce_the_reference_pipeline(golden, cls, levels):
    """`Compose([GuillardCoarseningAndConnectKNN, BuildKnnInterpWeights])` (examples/training/NsMuGSGNN/*.py) on a Graph whose
    positions are on the GPU == the graph the reference's own transforms produced (tests/golden/models_mugs.pt)."""
    import graphs4cfd_amd as gfd
    T = gfd.transforms
    ref = golden("models_mugs.pt")[cls]["graph"]
    g = T.Compose([T.GuillardCoarseningAndConnectKNN(k=(6,) * levels, period=None, scale_edge_attr=(0.1, 0.2, 0.4, 0.8)[:levels]),
                   T.BuildKnnInterpWeights(6)])(Graph(pos=ref["pos"].to(DEV)))
    for name, want in ref.items():
        if name in ("field", "glob", "omega", "batch"):
            continue
        got = getattr(g, name)
        assert got.device.type == "cuda", name
        if want.dtype in (torch.int64, torch.bool):
            assert torch.equal(got.cpu(), want), f"{cls}.{name}"
        else:
            torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=1e-6, msg=lambda m: f"{cls}.{name}: {m}")


def test_device_built_gmus_graph_is_the_host_built_one():
    host = S.mugs_graph(40_000, levels=3, seed=4)
    dev = S.mugs_graph(40_000, levels=3, seed=4, device=DEV)
    for name, want in host.to_dict().items():
        got = getattr(dev, name)
        assert got.device.type == "cuda", name
        if want.dtype in (torch.int64, torch.bool):
            assert torch.equal(got.cpu(), want), name
        else:
            torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=0.0, msg=lambda m: f"{name}: {m}")
