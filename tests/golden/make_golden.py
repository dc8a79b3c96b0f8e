"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own code.

Run in the authoring container only (needs /root/reference):

    python tests/golden/make_golden.py

It imports `/root/reference/graphs4cfd` unmodified through the stand-ins in `refenv.py`,
builds seeded synthetic Graphs + seeded weights, runs the reference blocks / models /
rollouts / transforms, and stores inputs, weights and outputs as plain tensors
(`torch.save` of dicts).  The fixtures carry inputs *and* topology, so neither the tests
nor the GPU box ever need the reference, PyG or torch_cluster.

Every case records which reference function produced it (file:line) so the parity tests
read like tests of the reference itself.
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refenv  # noqa: E402

gfd = refenv.import_reference()
from graphs4cfd.nn import blocks as rb  # noqa: E402  (the reference's blocks.py)
from graphs4cfd.transforms.remus import extend_graph  # noqa: E402

torch.set_num_threads(4)


def sd(module):
    return {k: v.detach().clone() for k, v in module.state_dict().items()}


def graph_dict(g):
    return {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in g.__dict__.items()}


def save(name, obj):
    path = os.path.join(HERE, name)
    torch.save(obj, path)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.0f} KiB")


# ------------------------------------------------------------------------------ meshes
def mus_graph(n, k, cells, seed, nf=3, loc=False, n_in=1, dim=2):
    torch.manual_seed(seed)
    g = gfd.Graph(pos=torch.rand(n, dim))
    g = gfd.transforms.ConnectKNN(k, period=None if dim == 3 else (None, None))(g)
    g = gfd.transforms.ScaleEdgeAttr(0.1)(g)
    if cells:
        g = gfd.transforms.GridClustering(cells)(g)
    g.field = torch.randn(n, nf * n_in)
    if loc:
        g.loc = torch.randn(n, 2)
    g.glob = torch.rand(n, 1)
    g.omega = (torch.rand(n, 1) > 0.9).float()
    return g


def remus_graph(n, k, seed):
    torch.manual_seed(seed)
    g = gfd.Graph(pos=torch.rand(n, 2))
    g = gfd.transforms.BuildRemusGraph(num_levels=3, k=k, scale_edge_length=(0.1, 0.2, 0.4))(g)
    g = gfd.transforms.BuildKnnInterpWeights(k)(g)
    g.field = torch.randn(n, 2)
    g.glob = torch.rand(n, 1)
    g.omega = (torch.rand(n, 1) > 0.9).float()
    return g


def mus_arch(cls_name, H, nf, node_in, d=2):
    mp = ((H + 2 * H, (H, H, H), True), (H + H, (H, H, H), True))
    down = (d + H, (H, H, H), True)
    up = (d + H + H, (H, H, H), True)
    layers = {
        "NsOneScaleGNN": ["mp11", "mp12", "mp13", "mp14", "mp15", "mp16", "mp17", "mp18"],
        "NsTwoScaleGNN": ["mp111", "mp112", "mp113", "mp114", "down_mp12", "mp21", "mp22", "mp23", "mp24",
                          "up_mp21", "mp121", "mp122", "mp123", "mp124"],
        "NsThreeScaleGNN": ["mp111", "mp112", "mp113", "mp114", "down_mp12", "mp211", "mp212", "down_mp23",
                            "mp31", "mp32", "mp33", "mp34", "up_mp32", "mp221", "mp222", "up_mp21",
                            "mp121", "mp122", "mp123", "mp124"],
        "NsFourScaleGNN": ["mp111", "mp112", "mp113", "mp114", "down_mp12", "mp211", "mp212", "down_mp23",
                           "mp311", "mp312", "down_mp34", "mp41", "mp42", "mp43", "mp44", "up_mp43",
                           "mp321", "mp322", "up_mp32", "mp221", "mp222", "up_mp21",
                           "mp121", "mp122", "mp123", "mp124"],
        "AdvOneScaleGNN": ["mp111", "mp112", "mp121", "mp122"],
        "AdvTwoScaleGNN": ["mp111", "mp112", "down_mp12", "mp21", "mp22", "mp23", "mp24", "up_mp21",
                           "mp121", "mp122"],
        "AdvThreeScaleGNN": ["mp111", "mp112", "down_mp12", "mp211", "mp212", "down_mp23",
                             "mp31", "mp32", "mp33", "mp34", "up_mp32", "mp221", "mp222", "up_mp21",
                             "mp121", "mp122"],
        "AdvFourScaleGNN": ["mp111", "mp112", "down_mp12", "mp211", "mp212", "down_mp23", "mp311", "mp312",
                            "down_mp34", "mp41", "mp42", "mp43", "mp44", "up_mp43", "mp321", "mp322",
                            "up_mp32", "mp221", "mp222", "up_mp21", "mp121", "mp122"],
    }[cls_name]
    arch = {"edge_encoder": (d, (H, H, H), False), "node_encoder": (node_in, (H, H, H), False)}
    for name in layers:
        arch[name] = down if name.startswith("down") else up if name.startswith("up") else mp
    arch["decoder"] = (H, (H, H, nf), False)
    return arch


def remus_arch(H):
    mp = ((H + 2 * H, (H, H), True), (H + H, (H, H), True))
    arch = {}
    for n in ["angle_encoder", "angle_encoder12", "angle_encoder2", "angle_encoder23", "angle_encoder3"]:
        arch[n] = (4, (H, H), True)
    for n in ["edge_encoder", "edge_encoder2", "edge_encoder3"]:
        arch[n] = (3, (H, H), True)
    for n in ["mp111", "mp112", "mp113", "mp114", "down_mp12", "mp211", "mp212", "down_mp23",
              "mp31", "mp32", "mp33", "mp34", "mp221", "mp222", "mp121", "mp122", "mp123", "mp124"]:
        arch[n] = mp
    arch["up_mp32"] = (H + H, (H, H, H), True)
    arch["up_mp21"] = (H + H, (H, H, H), True)
    arch["decoder"] = (H, (H, 1), False)
    return arch


# ------------------------------------------------------------------------------ blocks
def gen_blocks():
    out = {}
    torch.manual_seed(100)
    # a1: MLP  (reference graphs4cfd/nn/blocks.py:117-144)
    for i, (fin, widths, ln, m) in enumerate([
            (5, (32, 32, 32), False, 37),
            (384, (128, 128, 128), True, 70),
            (130, (128, 128, 128), True, 33),
            (128, (128, 3), False, 50),
            (4, (128, 128), True, 129),
            (7, (48, 16, 32), True, 21),
            (96, (32, 32, 32, 32), True, 65),
            (3, (64, 1), False, 5)]):
        mlp = rb.MLP(fin, widths, ln)
        if ln:  # non-trivial affine
            mlp.MLP.layer_norm.weight.data.uniform_(0.5, 1.5)
            mlp.MLP.layer_norm.bias.data.uniform_(-0.5, 0.5)
        x = torch.randn(m, fin) * 1.5
        with torch.no_grad():
            y = mlp(x)
        out[f"mlp_{i}"] = dict(ref="nn/blocks.py:117-144", args=(fin, widths, ln), weights=sd(mlp), x=x, y=y)

    # a3: scatter as used  (blocks.py:46-47,183,231; PyG semantics restated in refenv)
    idx = torch.tensor([0, 0, 3, 3, 3, 5, 1, 0], dtype=torch.long)
    src = torch.randn(8, 6)
    out["scatter"] = dict(ref="torch_geometric.utils.scatter via nn/blocks.py:183", src=src, index=idx,
                          sum_7=refenv.scatter(src, idx, 0, 7, "sum"), mean_7=refenv.scatter(src, idx, 0, 7, "mean"),
                          mean_none=refenv.scatter(src, idx, 0, None, "mean"))

    # a2: GNBlock on a kNN graph, H=128 and H=32, mean and sum  (blocks.py:147-190)
    for tag, H, n, aggr in [("h128_mean", 128, 60, "mean"), ("h32_sum", 32, 90, "sum"), ("h32_mean", 32, 90, "mean")]:
        g = mus_graph(n, 6, None, seed=7)
        blk = rb.GNBlock((3 * H, (H, H, H), True), (2 * H, (H, H, H), True), aggr=aggr)
        v, e = torch.randn(n, H), torch.randn(g.edge_index.size(1), H)
        with torch.no_grad():
            v2, e2 = blk(v, e, g.edge_index)
        out[f"gnblock_{tag}"] = dict(ref="nn/blocks.py:175-186", args=((3 * H, (H, H, H), True), (2 * H, (H, H, H), True)),
                                     aggr=aggr, weights=sd(blk), v=v, e=e, edge_index=g.edge_index, v_out=v2, e_out=e2)
    # GNBlock on an irregular graph: unsorted targets, isolated nodes (mean -> 0), duplicate edges, a hub node
    torch.manual_seed(101)
    H, n, m = 32, 40, 300
    ei = torch.randint(0, n - 5, (2, m))  # nodes n-5..n-1 receive nothing
    ei[1, :80] = 3                        # hub with in-degree >= 80
    ei[:, 100] = ei[:, 101]               # duplicate edge
    blk = rb.GNBlock((3 * H, (H, H, H), True), (2 * H, (H, H, H), True))
    v, e = torch.randn(n, H), torch.randn(m, H)
    with torch.no_grad():
        v2, e2 = blk(v, e, ei)
    out["gnblock_irregular"] = dict(ref="nn/blocks.py:175-186", args=((3 * H, (H, H, H), True), (2 * H, (H, H, H), True)),
                                    aggr="mean", weights=sd(blk), v=v, e=e, edge_index=ei, v_out=v2, e_out=e2)

    # a5: pool_edge  (blocks.py:51-68)
    g = mus_graph(200, 6, [0.15], seed=8)
    ea = torch.randn(g.edge_index.size(1), 16)
    for aggr in ("mean", "sum"):
        ei_l, ea_l = rb.pool_edge(g.idx1_to_idx2, g.edge_index, ea, aggr=aggr)
        out[f"pool_edge_{aggr}"] = dict(ref="nn/blocks.py:51-68", idx=g.idx1_to_idx2, edge_index=g.edge_index, edge_attr=ea,
                                        edge_index_out=ei_l, edge_attr_out=ea_l)
    # degenerate: every edge is intra-cluster -> empty result
    ei_l, ea_l = rb.pool_edge(torch.zeros(200, dtype=torch.long), g.edge_index, ea)
    out["pool_edge_empty"] = dict(ref="nn/blocks.py:51-68", idx=torch.zeros(200, dtype=torch.long), edge_index=g.edge_index,
                                  edge_attr=ea, edge_index_out=ei_l, edge_attr_out=ea_l)

    # a4/a6: DownMP then UpMP, H=32 and H=128  (blocks.py:193-290)
    for H, n in [(32, 200), (128, 120)]:
        g = mus_graph(n, 6, [0.15], seed=9)
        down = rb.DownMP((2 + H, (H, H, H), True), 1)
        up = rb.UpMP((2 + H + H, (H, H, H), True), 2)
        field1 = torch.randn(n, H)
        eattr1 = torch.randn(g.edge_index.size(1), H)
        gi = graph_dict(g)
        g.field, g.edge_attr = field1, eattr1
        pos1, ei1 = g.pos, g.edge_index
        with torch.no_grad():
            g = down(g, activation=torch.tanh)
            d_field, d_ei, d_ea, d_pos = g.field.clone(), g.edge_index.clone(), g.edge_attr.clone(), g.pos.clone()
            g = up(g, field1, pos1, activation=torch.tanh)
            u_field = g.field.clone()
        out[f"downup_h{H}"] = dict(ref="nn/blocks.py:219-237,265-290", graph=gi, field1=field1, edge_attr1=eattr1,
                                   down_args=((2 + H, (H, H, H), True), 1), up_args=((2 + H + H, (H, H, H), True), 2),
                                   down_weights=sd(down), up_weights=sd(up),
                                   down_field=d_field, down_edge_index=d_ei, down_edge_attr=d_ea, down_pos=d_pos,
                                   up_field=u_field)

    # REMuS blocks on a reference-built REMuS graph  (blocks.py:293-456)
    g = remus_graph(220, 5, seed=10)
    gi = graph_dict(g)
    H = 32
    E1, E2 = g.edge_index.size(1), g.edge_index2.size(1)
    A1, A12 = g.angle_index.size(1), g.angle_index12.size(1)
    emp = rb.EdgeMP((3 * H, (H, H), True), (2 * H, (H, H), True))
    e, a = torch.randn(E1, H), torch.randn(A1, H)
    with torch.no_grad():
        e2, a2 = emp(e, a, g.angle_index)
    out["edgemp"] = dict(ref="nn/blocks.py:322-333", args=((3 * H, (H, H), True), (2 * H, (H, H), True)), weights=sd(emp),
                         e=e, a=a, angle_index=g.angle_index, e_out=e2, a_out=a2)
    dmp = rb.DownEdgeMP((3 * H, (H, H), True), (2 * H, (H, H), True))
    e1, e2_, a12 = torch.randn(E1, H), torch.randn(E2, H), torch.randn(A12, H)
    with torch.no_grad():
        e2o = dmp(e1, e2_, a12, g.angle_index12)
    out["downedgemp"] = dict(ref="nn/blocks.py:360-381", args=((3 * H, (H, H), True), (2 * H, (H, H), True)), weights=sd(dmp),
                             e1=e1, e2=e2_, a12=a12, angle_index12=g.angle_index12, e2_out=e2o)
    ump = rb.UpEdgeMP((2 * H, (H, H, H), True))
    ea2, ea1 = torch.randn(E2, H), torch.randn(E1, H)
    with torch.no_grad():
        e1o = ump(g.pos, g.y_idx_21, g.x_idx_21, g.weights_21, ea2, g.edge_index2, g.edgeUnitVectorInverse2,
                  g.coarse_mask2, ea1, g.edge_index, g.edgeUnitVector)
    out["upedgemp_21"] = dict(ref="nn/blocks.py:408-456", args=((2 * H, (H, H, H), True),), weights=sd(ump),
                              edge_attr2=ea2, edge_attr1=ea1, e1_out=e1o)
    E3 = g.edge_index3.size(1)
    ea3, ea2b = torch.randn(E3, H), torch.randn(E2, H)
    with torch.no_grad():
        e2o = ump(g.pos, g.y_idx_32, g.x_idx_32, g.weights_32, ea3, g.edge_index3, g.edgeUnitVectorInverse3,
                  g.coarse_mask3, ea2b, g.edge_index2, g.edgeUnitVector2, g.coarse_mask2)
    out["upedgemp_32"] = dict(ref="nn/blocks.py:408-456", edge_attr3=ea3, edge_attr2=ea2b, e2_out=e2o)
    out["remus_graph"] = gi

    # a12: edgeScalarToNodeVector (blocks.py:88-114), inverse path, F=1 and F=H, plus the lstsq path
    s1 = torch.randn(E1, 1)
    sH = torch.randn(E2, 8)
    with torch.no_grad():
        out["es2nv"] = dict(
            ref="nn/blocks.py:88-114", s1=s1, sH=sH,
            v1=rb.edgeScalarToNodeVector(s1, g.edge_index, edgeUnitVectorInverse=g.edgeUnitVectorInverse),
            vH=rb.edgeScalarToNodeVector(sH, g.edge_index2, edgeUnitVectorInverse=g.edgeUnitVectorInverse2,
                                         coarse_mask=g.coarse_mask2),
            v1_lstsq=rb.edgeScalarToNodeVector(s1, g.edge_index, edgeUnitVector=g.edgeUnitVector))
    # a13: knn_interpolate (blocks.py:34-48)
    x = torch.randn(int(g.coarse_mask2.sum()), 12)
    with torch.no_grad():
        out["knn_interpolate"] = dict(ref="nn/blocks.py:34-48", x=x,
                                      y=rb.knn_interpolate(x, g.y_idx_21, g.x_idx_21, g.weights_21))
    # a15: restriction (blocks.py:9-32)
    n = g.pos.size(0)
    rg = gfd.Graph(field=torch.randn(int(g.coarse_mask2.sum()), 4))
    rb.restriction(rg, g.coarse_mask2, torch.randn(E2, 3), g.edge_index2, n, torch.device("cpu"))
    out["restriction"] = dict(ref="nn/blocks.py:9-32", edge_index_out=rg.edge_index)
    save("blocks.pt", out)


# ------------------------------------------------------------------------------ models
def gen_mus_models():
    out = {}
    H = 32
    cases = [
        ("NsOneScaleGNN", None, 3, False, 5),
        ("NsTwoScaleGNN", [0.12], 3, False, 5),
        ("NsThreeScaleGNN", [0.10, 0.20], 3, False, 5),
        ("NsFourScaleGNN", [0.08, 0.16, 0.32], 3, False, 5),
        ("AdvOneScaleGNN", None, 1, True, 5),
        ("AdvTwoScaleGNN", [0.12], 1, True, 5),
        ("AdvThreeScaleGNN", [0.10, 0.20], 1, True, 5),
        ("AdvFourScaleGNN", [0.08, 0.16, 0.32], 1, True, 5),
    ]
    for i, (cls, cells, nf, loc, node_in) in enumerate(cases):
        g = mus_graph(400, 6, cells, seed=20 + i, nf=nf, loc=loc)
        node_in = nf + (2 if loc else 0) + 2
        arch = mus_arch(cls, H, nf, node_in)
        torch.manual_seed(200 + i)
        model = getattr(gfd.nn, cls)(arch=arch)
        gi = graph_dict(g)
        with torch.no_grad():
            model.eval()
            y = model.forward(g)
        y3 = model.solve(g, 3)
        out[cls] = dict(ref="nn/mus_gnn.py forward + nn/model.py:303-327 solve", arch=arch, weights=sd(model),
                        graph=gi, forward=y, solve3=y3, num_params=model.num_params)
        assert torch.equal(g.field, gi["field"]) and torch.equal(g.edge_attr, gi["edge_attr"])
    save("models_mus.pt", out)


def gen_mus_3d():
    # BASELINE config 5's data path: the reference's NsFourScaleGNN (and, smaller, NsTwoScaleGNN) on a 3-D mesh —
    # edge_encoder in = 3, down in = 3 + H, up in = 3 + 2H (nn/mus_gnn.py:485-562 on transforms/connect.py, mus.py with dim 3)
    out = {}
    H = 32
    for i, (cls, cells, n) in enumerate((("NsFourScaleGNN", [0.20, 0.40, 0.80], 700), ("NsTwoScaleGNN", [0.25], 300))):
        g = mus_graph(n, 6, cells, seed=90 + i, nf=3, dim=3)
        arch = mus_arch(cls, H, 3, 5, d=3)
        torch.manual_seed(290 + i)
        model = getattr(gfd.nn, cls)(arch=arch)
        gi = graph_dict(g)
        with torch.no_grad():
            model.eval()
            y = model.forward(g)
        y3 = model.solve(g, 3)
        out[cls] = dict(ref="nn/mus_gnn.py forward + nn/model.py:303-327 solve, 3-D mesh", arch=arch, weights=sd(model),
                        graph=gi, forward=y, solve3=y3, num_params=model.num_params)
    save("models_mus_3d.pt", out)


MUGS_LAYERS = {
    "NsTwoGuillardScaleGNN": (["mp111", "mp112", "mp113", "mp114", "mp21", "mp22", "mp23", "mp24", "mp121", "mp122", "mp123", "mp124"],
                              ["mp121"]),
    "NsThreeGuillardScaleGNN": (["mp111", "mp112", "mp113", "mp114", "mp211", "mp212", "mp31", "mp32", "mp33", "mp34", "mp221", "mp222",
                                 "mp121", "mp122", "mp123", "mp124"], ["mp221", "mp121"]),
    "NsFourGuillardScaleGNN": (["mp111", "mp112", "mp113", "mp114", "mp211", "mp212", "mp311", "mp312", "mp41", "mp42", "mp43", "mp44",
                                "mp321", "mp322", "mp221", "mp222", "mp121", "mp122", "mp123", "mp124"], ["mp321", "mp221", "mp121"]),
}


def mugs_arch(cls_name, H, nf=3, node_in=5, d=2):
    """arch dict of the gMuS-GNN classes (nn/mugs_gnn.py docstrings): the first MP after every up-sampling takes node
    features [interpolated | stashed] = 2H wide."""
    layers, wide = MUGS_LAYERS[cls_name]
    levels = {"NsTwoGuillardScaleGNN": 2, "NsThreeGuillardScaleGNN": 3, "NsFourGuillardScaleGNN": 4}[cls_name]
    arch = {"edge_encoder": (d, (H, H, H), False), "node_encoder": (node_in, (H, H, H), False)}
    for l in range(2, levels + 1):
        arch[f"edge_encoder{l}"] = (d, (H, H, H), False)
    for name in layers:
        vw = 2 * H if name in wide else H
        arch[name] = ((H + 2 * vw, (H, H, H), True), (H + vw, (H, H, H), True))
    arch["decoder"] = (H, (H, H, nf), False)
    return arch


def gen_mugs_models():
    """gMuS-GNN family (nn/mugs_gnn.py:11-490) on graphs built by the reference's own
    GuillardCoarseningAndConnectKNN + BuildKnnInterpWeights (transforms/mugs.py:32-89, interpolate.py:133-155)."""
    out = {}
    H = 32
    for i, (cls, levels) in enumerate([("NsTwoGuillardScaleGNN", 2), ("NsThreeGuillardScaleGNN", 3), ("NsFourGuillardScaleGNN", 4)]):
        torch.manual_seed(500 + i)
        n = (500, 900, 3000)[i]          # the 4-level case needs > k nodes on its coarsest level
        g = gfd.Graph(pos=torch.rand(n, 2))
        g = gfd.transforms.GuillardCoarseningAndConnectKNN(k=(6,) * levels, period=None,
                                                           scale_edge_attr=(0.1, 0.2, 0.4, 0.8)[:levels])(g)
        g = gfd.transforms.BuildKnnInterpWeights(6)(g)
        g.batch = torch.zeros(n, dtype=torch.long)       # forward() slices graph.batch (nn/mugs_gnn.py:103); solve() sets the same
        g.field = torch.randn(n, 3)
        g.glob = torch.rand(n, 1)
        g.omega = (torch.rand(n, 1) > 0.9).float()
        arch = mugs_arch(cls, H)
        torch.manual_seed(520 + i)
        model = getattr(gfd.nn, cls)(arch=arch)
        gi = graph_dict(g)
        with torch.no_grad():
            model.eval()
            y = model.forward(g, 0)
        y3 = model.solve(g, 3)
        out[cls] = dict(ref="nn/mugs_gnn.py forward + nn/model.py:303-327 solve", arch=arch, weights=sd(model), graph=gi,
                        forward=y, solve3=y3, num_params=model.num_params)
        assert torch.equal(g.field, gi["field"])
    save("models_mugs.pt", out)


def gen_rollout():
    # a8: solve / shift_and_replace incl. n_in = 2 history window (model.py:303-327)
    out = {}
    H = 32
    g = mus_graph(300, 6, [0.12], seed=40)
    arch = mus_arch("NsTwoScaleGNN", H, 3, 5)
    torch.manual_seed(300)
    model = gfd.nn.NsTwoScaleGNN(arch=arch)
    gi = graph_dict(g)
    out["two_scale"] = dict(ref="nn/model.py:303-327", arch=arch, weights=sd(model), graph=gi,
                            solve1=model.solve(g, 1), solve5=model.solve(g, 5), solve50=model.solve(g, 50))
    g2 = mus_graph(300, 6, None, seed=41, n_in=2)
    arch2 = mus_arch("NsOneScaleGNN", H, 3, 3 * 2 + 2)
    torch.manual_seed(301)
    model2 = gfd.nn.NsOneScaleGNN(arch=arch2)
    gi2 = graph_dict(g2)
    out["one_scale_nin2"] = dict(ref="nn/model.py:303-327", arch=arch2, weights=sd(model2), graph=gi2,
                                 solve4=model2.solve(g2, 4))
    save("rollout.pt", out)


def gen_solve_list():
    # nn/model.py:308-309: solve() on a LIST of graphs = one rollout of their PyG batch (two meshes of different sizes, n_in = 1)
    H = 32
    g1, g2 = mus_graph(220, 6, None, seed=45), mus_graph(140, 6, None, seed=46)
    arch = mus_arch("NsOneScaleGNN", H, 3, 5)
    torch.manual_seed(302)
    model = gfd.nn.NsOneScaleGNN(arch=arch)
    d1, d2 = graph_dict(g1), graph_dict(g2)
    save("solve_list.pt", dict(ref="nn/model.py:303-321 (308-309: Batch.from_data_list)", arch=arch, weights=sd(model), graphs=[d1, d2],
                               solve3=model.solve([g1, g2], 3)))


def gen_remus_model():
    H = 32
    g = remus_graph(260, 5, seed=50)
    arch = remus_arch(H)
    torch.manual_seed(400)
    model = gfd.nn.NsRotEquiTreeScaleGNN(arch=arch)
    gi = graph_dict(g)
    with torch.no_grad():
        model.eval()
        y = model.forward(g)
    y3 = model.solve(g, 3)
    save("model_remus.pt", dict(ref="nn/remus_gnn.py:119-199", arch=arch, weights=sd(model), graph=gi,
                                forward=y, solve3=y3, num_params=model.num_params))


def gen_checkpoint():
    # checkpoint contract: file written by the reference's own save_checkpoint (model.py:329-349)
    H = 32
    g = mus_graph(150, 6, None, seed=60)
    arch = mus_arch("NsOneScaleGNN", H, 3, 5)
    torch.manual_seed(500)
    model = gfd.nn.NsOneScaleGNN(arch=arch)
    optimiser = torch.optim.Adam(model.parameters(), lr=1e-3)
    model.save_checkpoint(os.path.join(HERE, "reference_saved.chk"), n_out=1, epoch=3, optimiser=optimiser)
    with torch.no_grad():
        model.eval()
        y = model.forward(g)
    save("checkpoint_io.pt", dict(ref="nn/model.py:329-349,112-130", graph=graph_dict(g), forward=y,
                                  keys=list(model.state_dict().keys())))
    print(f"wrote reference_saved.chk: {os.path.getsize(os.path.join(HERE, 'reference_saved.chk')) / 1024:.0f} KiB")


def gen_transforms():
    # outputs of the pre-processing the synthetic-mesh generators must reproduce
    out = {}
    torch.manual_seed(70)
    pos = torch.rand(500, 2)
    g = gfd.Graph(pos=pos.clone())
    g = gfd.transforms.ConnectKNN(6)(g)
    g = gfd.transforms.ScaleEdgeAttr(0.1)(g)
    g = gfd.transforms.GridClustering([0.1, 0.2, 0.4])(g)
    out["mus_2d"] = dict(ref="transforms/connect.py:9-92, scale.py:29, mus.py:9-65", k=6, r=0.1, cells=[0.1, 0.2, 0.4],
                         graph=graph_dict(g))
    pos3 = torch.rand(400, 3)
    g3 = gfd.Graph(pos=pos3.clone())
    g3 = gfd.transforms.ConnectKNN(6, period=None)(g3)
    g3 = gfd.transforms.ScaleEdgeAttr(0.1)(g3)
    g3 = gfd.transforms.GridClustering([0.25, 0.5])(g3)
    out["mus_3d"] = dict(ref="transforms/connect.py:9-92, mus.py:9-65", k=6, r=0.1, cells=[0.25, 0.5], graph=graph_dict(g3))
    gr = remus_graph(300, 5, seed=71)
    out["remus_2d"] = dict(ref="transforms/remus.py:9-175, mugs.py:8-29, interpolate.py:110-155", k=5,
                           scale=(0.1, 0.2, 0.4), graph=graph_dict(gr))
    save("transforms.pt", out)


def gen_training():
    # GNN.fit's inner loop (nn/model.py:226-236): train-mode forward, GraphLoss (nn/losses.py:10-16), backward; then the
    # second rollout step on the fed-back detached prediction (:229-230).  Gradients of every parameter, both steps.
    H = 32
    g = mus_graph(400, 6, [0.10, 0.20], seed=60)
    g.omega[::7] = 1.0                                   # make sure the Dirichlet term of the loss is active
    arch = mus_arch("NsThreeScaleGNN", H, 3, 5)
    torch.manual_seed(600)
    model = gfd.nn.NsThreeScaleGNN(arch=arch)
    g.target = torch.randn(400, 6)
    gi = graph_dict(g)
    criterion = gfd.nn.GraphLoss(lambda_d=0.5)
    model.train()
    steps = []
    pred = None
    for t in range(2):
        if t > 0:
            g.field = model.shift_and_replace(g.field, pred.detach())
        model.zero_grad()
        pred = model.forward(g, t)
        loss = criterion(g, pred, g.target[:, 3 * t:3 * (t + 1)])
        loss.backward()
        steps.append(dict(loss=loss.detach().clone(), pred=pred.detach().clone(), grad_norm2=model.grad_norm2(),
                          grads={k: p.grad.detach().clone() for k, p in model.named_parameters()}))
    save("training.pt", dict(ref="nn/model.py:226-236 + nn/losses.py:10-16", arch=arch, weights=sd(model), graph=gi,
                             lambda_d=0.5, steps=steps))


def gen_augment():
    # scale.py:33-80, geometric.py:33-252, subset.py:7-30: the deterministic data transforms of the training pipelines
    out = {}
    g = mus_graph(120, 6, [0.2], seed=70, n_in=2)
    g.target = torch.randn(120, 9)
    stages = [("input", None),
              ("ScaleNs", gfd.transforms.ScaleNs({'u': (-2.1, 2.6), 'v': (-2.25, 2.1), 'p': (-3.7, 2.35), 'Re': (500, 1000)}, format='uvp')),
              ("GraphRotation", gfd.transforms.GraphRotation(37.0, eq='ns', format='uvp')),
              ("flip_y", lambda gr: gfd.transforms.geometric.flip_graph_dim(gr, 1, eq='ns', format='uvp')),
              ("NodeSubset", gfd.transforms.NodeSubset(list(range(0, 120, 3))))]
    seq = []
    for name, t in stages:
        if t is not None:
            g = t(g)
        seq.append((name, graph_dict(g)))
    out["mus_uvp"] = dict(ref="transforms/scale.py:33-80, geometric.py:33-114,170-216, subset.py:7-30", stages=seq)
    ga = mus_graph(90, 6, None, seed=71, nf=1, loc=True)
    gi = graph_dict(ga)
    out["adv"] = dict(ref="geometric.py:33-114 (eq='adv')", input=gi, theta=201.5,
                      output=graph_dict(gfd.transforms.GraphRotation(201.5, eq='adv')(ga)))
    gr = remus_graph(150, 5, seed=72)
    gr.target = torch.randn(150, 4)
    gi = graph_dict(gr)
    out["remus_uv"] = dict(ref="geometric.py:69-86 (angle_index: unit vectors + pseudo-inverses)", input=gi, theta=123.0,
                           output=graph_dict(gfd.transforms.GraphRotation(123.0, eq='ns', format='uv')(gr)))
    g3 = mus_graph(60, 6, None, seed=73, dim=3)
    gi = graph_dict(g3)
    out["rot3d"] = dict(ref="geometric.py:63-66 (Tait-Bryan)", input=gi, theta=[30.0, 75.0, 210.0],
                        output=graph_dict(gfd.transforms.GraphRotation([30.0, 75.0, 210.0])(g3)))
    # periodic kNN connect (transforms/connect.py:36-71) and the Guillard transform on a periodic domain (mugs.py:57-88)
    torch.manual_seed(74)
    pos = torch.rand(300, 2) * torch.tensor([4.0, 1.0])
    for name, period in (("per_y_auto", (None, "auto")), ("per_xy", (4.0, 1.0))):
        gp = gfd.transforms.ConnectKNN(6, period=period)(gfd.Graph(pos=pos.clone()))
        out[name] = dict(ref="transforms/connect.py:9-72", pos=pos.clone(), period=period, edge_index=gp.edge_index.clone(), edge_attr=gp.edge_attr.clone())
    gg = gfd.transforms.GuillardCoarseningAndConnectKNN(k=(6, 6, 6), period=(None, "auto"), scale_edge_attr=(0.1, 0.25, 0.5))(gfd.Graph(pos=pos.clone()))
    gb = gfd.transforms.BuildRemusGraph(num_levels=3, k=5, period=(None, "auto"), scale_edge_length=(0.1, 0.2, 0.4))(gfd.Graph(pos=pos[:200].clone()))
    out["remus_periodic"] = dict(ref="transforms/remus.py:63-148 with period", pos=pos[:200].clone(), graph=graph_dict(gb))
    out["guillard_periodic"] = dict(ref="transforms/mugs.py:32-89", pos=pos.clone(), graph=graph_dict(gg))
    # interpolate.py:13-49: griddata interpolation of a point cloud to new nodes
    torch.manual_seed(76)
    gp = gfd.Graph(pos=torch.rand(400, 2))
    gp.edge_index = None
    gp.loc, gp.field, gp.target = torch.randn(400, 2), torch.sin(3 * gp.pos[:, :1]) * torch.ones(1, 2), torch.cos(2 * gp.pos[:, 1:]) * torch.ones(1, 3)
    gp.omega = (gp.pos[:, :1] < 0.1).float()
    gp.bound = (2 * (gp.pos[:, 0] < 0.1)).type(torch.uint8)
    new_pos = 0.2 + 0.6 * torch.rand(150, 2)
    gi = graph_dict(gp)
    out["interpolate_nodes"] = dict(ref="transforms/interpolate.py:13-49", input=gi, new_pos=new_pos.clone(),
                                    output=graph_dict(gfd.transforms.InterpolateNodes(new_pos)(gp)))
    # datasets.py:120-337: record layout -> Graph (data2graph) of the three dataset classes, on synthetic NaN-padded records
    from graphs4cfd import datasets as rds
    torch.manual_seed(75)
    T, n_real, n_pad = 12, 40, 7
    cases = {}
    for name, cls, kw, cols in (("Adv", rds.Adv, {}, 5 + T), ("NsCircle_uvp", rds.NsCircle, {"format": "uvp"}, 4 + 3 * T),
                                ("NsCircle_uv", rds.NsCircle, {"format": "uv"}, 4 + 3 * T), ("NsEllipse_uv", rds.NsEllipse, {"format": "uv"}, 4 + 6 * T),
                                ("NsEllipse_uvp", rds.NsEllipse, {"format": "uvp"}, 4 + 6 * T)):
        rec = torch.randn(n_real + n_pad, cols)
        rec[:, 4 if name == "Adv" else 3] = torch.randint(0, 5 if name != "Adv" else 4, (n_real + n_pad,)).float()
        rec[n_real:] = float("nan")
        ds = cls(path="unused.h5", training_info={"n_in": 2, "n_out": 3, "step": 2, "T": T}, **kw)
        g = ds.data2graph(rec, 1, 1 + 2 * 2, 1 + (2 + 3) * 2 - 1, 2)
        cases[name] = dict(record=rec.clone(), args=(1, 5, 10, 2), graph=graph_dict(g), length=ds.training_sequences_length)
    out["datasets"] = dict(ref="datasets.py:26-44 (window length), :158-197, :222-266, :291-337", cases=cases)
    save("augment.pt", out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "augment":
        gen_augment()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "mus3d":
        gen_mus_3d()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "solve_list":
        gen_solve_list()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "training":      # (adds one fixture without rewriting the others)
        gen_training()
        sys.exit(0)
    gen_blocks()
    gen_mus_models()
    gen_mus_3d()
    gen_mugs_models()
    gen_rollout()
    gen_solve_list()
    gen_remus_model()
    gen_checkpoint()
    gen_transforms()
    gen_training()
    gen_augment()
