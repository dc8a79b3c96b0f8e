"""Node partition + halo exchange (graphs4cfd_amd/partition.py) on CPU: structural invariants of the
partition, and the partitioned V-cycle run on 2 gloo ranks with the ORACLE as the arithmetic back-end,
compared with the single-process oracle forward (the HIP back-end is covered by the -m gpu tests)."""
import os
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from graphs4cfd_amd import partition as P, synthetic as S
from oracle import g4c_oracle as O


@pytest.mark.parametrize("method", ["rcb", "strips"])
def test_partition_invariants(method, monkeypatch):
    monkeypatch.setenv("G4C_PARTITION", method)
    g = S.mus_graph(3000, levels=3, seed=2)
    for world in (2, 4, 3):
        parts = P.build_partition(g, 3, world)
        edges = P.coarse_topology(g, 3)
        owners = P.assign_owners(g, 3, world)
        sizes = [int(g.pos.size(0)), int(g.pos_2.size(0)), int(g.pos_3.size(0))]
        # the count rank-uniform decisions are taken on (what a halo exchange carries): the smallest rank's, the same for all
        uni = P.uniform_edge_counts(parts)
        assert uni == [min(parts[r][l].edge_index.shape[1] for r in range(world)) for l in range(3)]
        assert all(uni[l] <= parts[r][l].edge_index.shape[1] for r in range(world) for l in range(3))
        for l in range(3):
            own_all = np.concatenate([parts[r][l].owned for r in range(world)])
            assert np.array_equal(np.sort(own_all), np.arange(sizes[l])), "every node owned exactly once"
            e_all = np.concatenate([parts[r][l].edge_ids for r in range(world)])
            assert np.array_equal(np.sort(e_all), np.arange(edges[l].shape[1])), "every edge owned exactly once"
            for r in range(world):
                p = parts[r][l]
                loc2glob = np.concatenate([p.owned, p.halo])
                assert np.array_equal(loc2glob[p.edge_index[0]], edges[l][0, p.edge_ids])
                assert np.array_equal(loc2glob[p.edge_index[1]], edges[l][1, p.edge_ids])
                assert (p.edge_index[1] < p.n_own).all(), "edges are owned by the rank of their target"
                assert (owners[l][p.halo] != r).all() and np.array_equal(owners[l][p.halo], p.halo_owner)
                for q in range(world):   # what r sends to q is exactly q's halo owned by r, in q's order
                    need = parts[q][l].halo[parts[q][l].halo_owner == r]
                    assert np.array_equal(p.owned[p.send_idx[q]], need)
                    assert parts[q][l].recv_counts[r] == len(need)
        # clusters are wholly owned: a fine node and its parent have the same owner
        assert np.array_equal(owners[0], owners[1][g.idx1_to_idx2.numpy()])
        assert np.array_equal(owners[1], owners[2][g.idx2_to_idx3.numpy()])
        # balance: within 25 % of the mean
        counts = np.bincount(owners[0], minlength=world)
        assert counts.max() < 1.25 * counts.mean()


def test_rcb_owners_are_compact_and_deterministic():
    """Recursive coordinate bisection (the default): same table on every call, every rank used, fewer halo rows than slabs along x
    once the slabs get thin (8 ranks), in 2-D and 3-D."""
    for dim, n in ((2, 20000), (3, 20000)):
        g = S.mus_graph(n, levels=2, dim=dim, seed=4)
        a = P.assign_owners(g, 2, 8, "rcb")
        b = P.assign_owners(g, 2, 8, "rcb")
        assert all(np.array_equal(x, y) for x, y in zip(a, b)) and set(a[0].tolist()) == set(range(8))
        halo = {}
        for m in ("rcb", "strips"):
            os.environ["G4C_PARTITION"] = m
            try:
                parts = P.build_partition(g, 2, 8)
            finally:
                del os.environ["G4C_PARTITION"]
            halo[m] = max(p[0].n_halo for p in parts)
        assert halo["rcb"] < 0.8 * halo["strips"], halo
    with pytest.raises(ValueError):
        P.assign_owners(g, 2, 8, "metis")


class OracleImpl:
    """Test-only arithmetic back-end for MusPartitionedForward (CPU, oracle ops)."""

    def __init__(self, w, use_products=False):
        self.w, self.use_products = w, use_products
        self.overlaps, self.overlapped = use_products, 0     # product exchange -> also the interior/boundary split

    def new(self, rows, width, device):
        return torch.zeros(rows, width)

    def _products(self, v_out, n_own, next_name, pr_out):
        if next_name is None or pr_out is None:
            return None
        H = v_out.size(1)
        Wn = self.w[f"{next_name}.edge_mlp.MLP.linear_1.weight"]
        pr_out[:n_own] = v_out @ Wn[:, -2 * H:-H].T
        return pr_out, v_out @ Wn[:, -H:].T

    def encode(self, mesh, v_out, next_name=None, pr_out=None):
        x = torch.cat([mesh.inputs[k] for k in ("field", "loc", "glob", "omega") if k in mesh.inputs], 1)
        v_out.copy_(F.selu(O.mlp(x, self.w, "node_encoder")))
        return F.selu(O.mlp(mesh.edge_attr, self.w, "edge_encoder")), self._products(v_out, v_out.size(0), next_name, pr_out)

    @staticmethod
    def _act(e, pending):
        return F.selu(e) if pending else e

    def hoists(self, n_edges):
        """Product exchange (MusPartitionedForward): on for this back-end iff `use_products`."""
        return self.use_products

    def _mlp_tail(self, h, prefix):
        """O.mlp from after its first Linear: (SELU, Linear)*, LayerNorm."""
        n = 1
        while f"{prefix}.MLP.linear_{n + 1}.weight" in self.w:
            n += 1
        x = h
        for i in range(2, n + 1):
            x = F.linear(F.selu(x), self.w[f"{prefix}.MLP.linear_{i}.weight"], self.w[f"{prefix}.MLP.linear_{i}.bias"])
        g = self.w.get(f"{prefix}.MLP.layer_norm.weight")
        return x if g is None else F.layer_norm(x, (x.size(-1),), g, self.w[f"{prefix}.MLP.layer_norm.bias"], 1e-5)

    def mp(self, name, v, e, e_pending, edge_index, n_own, v_out, products=None, next_name=None, pr_out=None, overlap=None):
        row, col = edge_index
        H = v.size(1)
        if overlap is not None:     # interior edges before the exchange has been waited for, boundary edges after
            start, wait, sub = overlap
            W1, b1 = self.w[f"{name}.edge_mlp.MLP.linear_1.weight"], self.w[f"{name}.edge_mlp.MLP.linear_1.bias"]
            ea = self._act(e, e_pending)
            e_new = torch.full((row.numel(), self.w[f"{name}.edge_mlp.MLP.linear_1.bias"].numel()), float("nan"))
            handle = start()
            for tag in ("int", "bnd"):
                ids, r, c = (t.long() for t in sub[tag])
                if tag == "bnd":
                    wait(handle)
                assert torch.equal(r, row[ids]) and torch.equal(c, col[ids])
                assert bool((r < n_own).all()) if tag == "int" else bool((r >= n_own).all())
                h = ea[ids] @ W1[:, :-2 * H].T + products[0][r] + products[1][c] + b1
                e_new[ids] = self._mlp_tail(h, f"{name}.edge_mlp")
            assert not torch.isnan(e_new).any()                    # the two subsets cover every owned edge
            self.overlapped += 1
        elif products is None:
            e_new = O.mlp(torch.cat((self._act(e, e_pending), v[row], v[col]), 1), self.w, f"{name}.edge_mlp")
        else:       # W1 [e | v_row | v_col] = W1e e + (W1r v)[row] + (W1c v)[col]; the halo rows of W1r v were exchanged
            W1, b1 = self.w[f"{name}.edge_mlp.MLP.linear_1.weight"], self.w[f"{name}.edge_mlp.MLP.linear_1.bias"]
            h = self._act(e, e_pending) @ W1[:, :-2 * H].T + products[0][row] + products[1][col] + b1
            e_new = self._mlp_tail(h, f"{name}.edge_mlp")
        agg = O.scatter(e_new, col, n_own, "mean")
        v_out.copy_(F.selu(O.mlp(torch.cat((agg, v[:n_own]), 1), self.w, f"{name}.node_mlp")))
        return e_new, self._products(v_out, n_own, next_name, pr_out)

    def down(self, name, v_own, rel, parent, n_coarse, e, e_pending, pool_csr, v_out):
        msg = O.mlp(torch.cat((rel, v_own), 1), self.w, f"{name}.down_mlp")
        v_out.copy_(torch.tanh(O.scatter(msg, parent, n_coarse, "mean")))
        seg = torch.repeat_interleave(torch.arange(pool_csr.n_seg), (pool_csr.off[1:] - pool_csr.off[:-1]).long())
        return O.scatter(self._act(e, e_pending)[pool_csr.perm.long()], seg, pool_csr.n_seg, "mean")

    def up(self, name, v_coarse, v_old_own, rel, parent, v_out, next_name=None, pr_out=None):
        v_out.copy_(torch.tanh(O.mlp(torch.cat((-rel, v_coarse[parent], v_old_own), 1), self.w, f"{name}.up_mlp")))
        return self._products(v_out, v_out.size(0), next_name, pr_out)

    def decode(self, v_own, field, nf):
        return field[:, -nf:] + O.mlp(v_own, self.w, "node_decoder")


def _worker(rank, world, port, model_name, levels, out_dir, use_products, n_nodes=1500):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2 if world <= 4 else 1)
        g = S.mus_graph(n_nodes, levels=levels, seed=5)
        arch = S.mus_arch(model_name, 32)
        torch.manual_seed(11)
        import graphs4cfd_amd as gfd
        model = getattr(gfd.nn, model_name)(arch=arch)     # CPU: parameters only, never run
        w = {k: v.detach() for k, v in model.state_dict().items()}
        parts = P.build_partition(g, levels, world)
        mesh = P.LocalMesh(g, levels, parts[rank], torch.device("cpu"), rank, world)
        impl = OracleImpl(w, use_products)
        fwd = P.MusPartitionedForward(model._PROGRAM, mesh, impl, P.HaloExchanger(mesh), 32, 3)
        with torch.no_grad():
            pred = fwd.forward()
        assert not use_products or impl.overlapped > 0      # the interior/boundary path ran
        full = torch.zeros(g.pos.size(0), 3)
        full[mesh.owned_global[0]] = pred
        dist.all_reduce(full)
        if rank == 0:
            with torch.no_grad():
                ref = O.mus_forward(model_name, g.to_dict(), w, 3)
            torch.save({"full": full, "ref": ref, "halo": [mesh.n_halo, mesh.n_own]}, os.path.join(out_dir, "result.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("model_name,levels,use_products", [("NsThreeScaleGNN", 3, False), ("NsOneScaleGNN", 1, False),
                                                            ("NsThreeScaleGNN", 3, True)])
def test_partitioned_forward_matches_global_on_two_gloo_ranks(tmp_path, model_name, levels, use_products):
    """`use_products`: consecutive MP layers exchange the halo rows of W1r v (made by the previous layer) instead of v."""
    import torch.multiprocessing as mp
    port = 29600 + (os.getpid() % 300) + levels + 7 * int(use_products)
    mp.spawn(_worker, args=(2, port, model_name, levels, str(tmp_path), use_products), nprocs=2, join=True)
    r = torch.load(os.path.join(str(tmp_path), "result.pt"))
    assert all(h > 0 for h in r["halo"][0]), "the test mesh must actually have halos on every level"
    torch.testing.assert_close(r["full"], r["ref"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("world", [4, 8])
def test_partitioned_forward_matches_global_on_four_and_eight_gloo_ranks(tmp_path, world):
    """the partition at the world sizes of BASELINE configs 4 and 5 (every rank its own process, gloo transport,
    products carried by the halo exchange): a 3-scale mesh cut 4 and 8 ways — ranks with several neighbours, coarse levels on which
    some ranks own next to nothing — reassembles to the global forward."""
    import torch.multiprocessing as mp
    port = 29300 + (os.getpid() % 250) + world
    mp.spawn(_worker, args=(world, port, "NsThreeScaleGNN", 3, str(tmp_path), True, 3000), nprocs=world, join=True)
    r = torch.load(os.path.join(str(tmp_path), "result.pt"))
    assert r["halo"][0][0] > 0
    torch.testing.assert_close(r["full"], r["ref"], rtol=1e-4, atol=1e-4)


def _lonely_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # ranks 0 and 1 exchange two rows each way; rank 2 neither sends nor receives at this level
        counts = {0: ([0, 2, 0], [0, 2, 0]), 1: ([2, 0, 0], [2, 0, 0]), 2: ([0, 0, 0], [0, 0, 0])}[rank]
        send_idx = [torch.tensor([1, 3], dtype=torch.int32) if c else torch.zeros(0, dtype=torch.int32) for c in counts[0]]
        mesh = types.SimpleNamespace(world=world, rank=rank, n_own=[4], send_counts=[counts[0]], recv_counts=[counts[1]], send_idx32=[send_idx])
        v = torch.zeros(4 + sum(counts[1]), 3)
        v[:4] = torch.arange(12.).reshape(4, 3) + 100 * rank
        P.HaloExchanger(mesh).exchange(v, 1)
        torch.save(v, os.path.join(out_dir, f"v{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_halo_exchange_is_entered_by_ranks_with_an_empty_halo(tmp_path):
    """The exchange is a collective: a rank with nothing to send or receive at a level still has to enter it (zero-length
    splits), or the ranks that do exchange wait for it forever."""
    import torch.multiprocessing as mp
    port = 29900 + os.getpid() % 90
    mp.spawn(_lonely_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    v0, v1, v2 = (torch.load(os.path.join(tmp_path, f"v{r}.pt")) for r in range(3))
    own = lambda r: torch.arange(12.).reshape(4, 3) + 100 * r
    assert torch.equal(v0[4:], own(1)[[1, 3]]) and torch.equal(v1[4:], own(0)[[1, 3]]) and torch.equal(v2, own(2))


# ------------------------------------------------------------------------------------- REMuS-GNN: edge-latent halo
def test_remus_partition_invariants():
    from graphs4cfd_amd import partition_remus as PR
    g = S.remus_graph(1500, k=5, seed=3)
    for world in (2, 3):
        parts = PR.build_remus_partition(g, world)
        owner = PR.remus_owners(g, world)
        assert np.array_equal(np.sort(np.concatenate([p.nodes for p in parts])), np.arange(1500))
        for l, s in ((1, ""), (2, "2"), (3, "3")):
            ei = getattr(g, f"edge_index{s}").numpy()
            ai = getattr(g, f"angle_index{s}").numpy()
            assert np.array_equal(np.sort(np.concatenate([p.levels[l - 1].edge_ids for p in parts])), np.arange(ei.shape[1]))
            assert np.array_equal(np.sort(np.concatenate([p.levels[l - 1].angle_ids for p in parts])), np.arange(ai.shape[1]))
            for r, p in enumerate(parts):
                lv = p.levels[l - 1]
                assert (owner[ei[1][lv.edge_ids]] == r).all() and (owner[ei[1][lv.halo_edges]] != r).all()
                loc2glob = np.concatenate([lv.edge_ids, lv.halo_edges])
                assert np.array_equal(loc2glob[lv.angle_index[0]], ai[0][lv.angle_ids])
                assert np.array_equal(loc2glob[lv.angle_index[1]], ai[1][lv.angle_ids])
                for q in range(world):       # what r sends to q is exactly q's halo owned by r, in q's order
                    lq = parts[q].levels[l - 1]
                    assert np.array_equal(lv.edge_ids[p.send_idx[l][q]], lq.halo_edges[lq.halo_owner == r])
                    assert parts[q].recv_counts[l][r] == int((lq.halo_owner == r).sum())
                if l < 3:
                    ad = getattr(g, f"angle_index{l}{l + 1}").numpy()
                    nxt = p.levels[l]
                    assert np.array_equal(loc2glob[lv.down_angle_index[0]], ad[0][lv.down_angle_ids])
                    assert np.array_equal(nxt.edge_ids[lv.down_angle_index[1]], ad[1][lv.down_angle_ids])
        for lo in (2, 3):
            mask_lo = getattr(g, f"coarse_mask{lo}").numpy()
            for r, p in enumerate(parts):
                it = p.interp[lo]
                assert (owner[it.halo_nodes] != r).all() and mask_lo[it.halo_nodes].all()
                for q in range(world):
                    iq = parts[q].interp[lo]
                    assert np.array_equal(p.levels[lo - 1].nodes[p.send_idx[PR.CH_NODE[lo]][q]], iq.halo_nodes[iq.halo_owner == r])


class RemusOracleImpl:
    """Test-only arithmetic back-end for RemusPartitionedForward (CPU, oracle ops) with the contract of RemusHipImpl."""

    def __init__(self, w, mesh):
        self.w, self.mesh = w, mesh
        self.H = w["edge_encoder.MLP.linear_1.weight"].shape[0] if "edge_encoder.MLP.linear_1.weight" in w else 32

    def _ebuf(self, l, own_rows):
        m = self.mesh
        out = torch.zeros(m.n_edges[l] + m.n_halo_edges[l], own_rows.size(1))
        out[: m.n_edges[l]] = own_rows
        return out

    def _project(self, v, l):
        m = self.mesh
        col = m.col32[l].long()
        return (v[col].reshape(col.size(0), -1, 2) * m.unit[l].unsqueeze(1)).sum(-1)

    def encode(self):
        m, w = self.mesh, self.w
        sfx = {1: "", 2: "2", 3: "3"}
        e, a = {}, {}
        for l in (1, 2, 3):
            col = m.col32[l].long()
            x = torch.cat([self._project(m.inputs["field"], l), m.inputs["glob"][col], m.inputs["omega"][col]], 1)
            e[l] = self._ebuf(l, F.selu(O.mlp(x, w, f"edge_encoder{sfx[l]}")))
            a[l] = F.selu(O.mlp(m.angle_attr[l], w, f"angle_encoder{sfx[l]}"))
        ax = {1: F.selu(O.mlp(m.down_attr[1], w, "angle_encoder12")), 2: F.selu(O.mlp(m.down_attr[2], w, "angle_encoder23"))}
        return e, a, ax

    def mp(self, name, e, a, a_pending, lvl):
        e2, a2 = O.edge_mp(e, a, self.mesh.angle_index[lvl], self.w, name)
        return self._ebuf(lvl, F.selu(e2[: self.mesh.n_edges[lvl]])), F.selu(a2)

    def down(self, name, e_lo, e_hi, a_x, lvl):
        r = O.down_edge_mp(e_lo, e_hi, a_x, self.mesh.down_index[lvl], self.w, name)
        return self._ebuf(lvl + 1, F.selu(r[: self.mesh.n_edges[lvl + 1]]))

    def _node_vec(self, s, l, n):
        m = self.mesh
        return (m.unit_inv[l] @ s.reshape(n, m.k, -1)).transpose(1, 2).reshape(n, -1)

    def node_vectors(self, e_lo, lo):
        m = self.mesh
        n = m.n_level_nodes[lo]
        nb = torch.zeros(n + m.n_halo_nodes[lo], 2 * e_lo.size(1))
        nb[:n] = self._node_vec(e_lo[: m.n_edges[lo]], lo, n)
        return nb

    def up(self, name, nb, e_hi, lo):
        m, hi = self.mesh, lo - 1
        y, x, wt = m.interp_y[lo], m.interp_x32[lo].long(), m.interp_w[lo].reshape(-1, 1)
        n_hi = m.n_level_nodes[hi]
        interp = O.scatter(nb[x] * wt, y, n_hi, "sum") / O.scatter(wt, y, n_hi, "sum")
        v1 = torch.zeros(m.n_nodes, nb.size(1))
        v1[m.level_node32[hi].long()] = interp
        x_in = torch.cat([self._project(v1, hi), e_hi[: m.n_edges[hi]]], 1)
        return self._ebuf(hi, F.selu(O.mlp(x_in, self.w, f"{name}.up_mlp")))

    def decode(self, e1):
        m = self.mesh
        s = O.mlp(e1[: m.n_edges[1]], self.w, "edge_decoder")
        return m.inputs["field"][:, -2:] + self._node_vec(s, 1, m.n_nodes)


def _remus_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    import graphs4cfd_amd as gfd
    from graphs4cfd_amd import partition_remus as PR
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        g = S.remus_graph(1200, k=5, seed=7)
        model = gfd.nn.NsRotEquiTreeScaleGNN(arch=S.remus_arch(32))
        w = {k: v.detach() for k, v in model.state_dict().items()}
        parts = PR.build_remus_partition(g, world)
        mesh = PR.RemusLocalMesh(g, parts[rank], torch.device("cpu"), rank, world)
        xch = P.HaloExchanger(mesh)
        fwd = PR.RemusPartitionedForward(model._PROGRAM, mesh, RemusOracleImpl(w, mesh), xch)
        with torch.no_grad():
            pred = fwd.forward()
        full = torch.zeros(g.pos.size(0), 2)
        full[mesh.owned_global[0]] = pred
        dist.all_reduce(full)
        if rank == 0:
            with torch.no_grad():
                ref = O.remus_forward(g.to_dict(), w)
            torch.save({"full": full, "ref": ref, "halo": mesh.n_halo, "exchanges": xch.n_exchanges}, os.path.join(out_dir, "result.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_remus_partitioned_forward_matches_global_on_gloo_ranks(tmp_path, world):
    """REMuS-GNN on `world` gloo ranks (edge-latent halo before every EdgeMP / DownEdgeMP, node-vector halo in every UpEdgeMP)
    == the single-process oracle forward."""
    import torch.multiprocessing as mp
    port = 29400 + (os.getpid() % 300) + world
    mp.spawn(_remus_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = torch.load(os.path.join(str(tmp_path), "result.pt"))
    assert all(h > 0 for h in r["halo"]), "the test mesh must actually have halos on every channel"
    assert r["exchanges"] == 16 + 2 + 2      # one per EdgeMP, one per DownEdgeMP, one per UpEdgeMP
    torch.testing.assert_close(r["full"], r["ref"], rtol=1e-4, atol=1e-4)
