"""Pre-processing on the device against the host path: ConnectKNN (`g4c_knn_grid`) and
GridClustering (device voxel ids + `g4c_segment_reduce`) on the headline mesh.  Usage: python scripts/bench_preprocess.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import ops, synthetic as S          # noqa: E402
from graphs4cfd_amd.graph import Graph                   # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    dev = torch.device("cuda", 0)
    for n, dim in ((100_000, 2), (1_000_000, 3)):
        gq = torch.Generator().manual_seed(1)
        px = torch.rand(n // 4, dim, generator=gq)
        py = torch.rand(n, dim, generator=gq)
        pxd, pyd = px.to(dev), py.to(dev)
        print(f"n={n} dim={dim}: knn_interp_weights ({n // 4} -> {n} nodes, k=3) host {timed(lambda: S.knn_interp_weights(px, py, 3), 1):.1f} ms, device {timed(lambda: S.knn_interp_weights(pxd, pyd, 3), 5):.2f} ms")
        pos = torch.rand(n, dim, generator=torch.Generator().manual_seed(0))
        pos_d = pos.to(dev)
        cells = S.default_cells(n, dim, 3)
        host_knn = timed(lambda: S.connect_knn(pos, 6), 1)
        dev_knn = timed(lambda: S.connect_knn(pos_d, 6), 5)
        host_grid = timed(lambda: S.add_grid_levels(Graph(pos=pos), cells), 1)
        dev_grid = timed(lambda: S.add_grid_levels(Graph(pos=pos_d), cells), 5)
        # the search kernel alone (binning excluded), HIP events on the launch stream
        nbr = S.knn_neighbours_device(pos_d, 6)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            S.knn_neighbours_device(pos_d, 6)
        e1.record()
        torch.cuda.synchronize()
        if dim == 2:
            eh, _ = S.connect_knn(pos, 5)
            ed = eh.to(dev)
            assert torch.equal(S.guillard_coarsening(ed, n).cpu(), S.guillard_coarsening(eh, n))
            print(f"n={n}: guillard coarsening host (sequential) {timed(lambda: S.guillard_coarsening(eh, n), 1):.1f} ms, device (rounds) {timed(lambda: S.guillard_coarsening(ed, n), 3):.2f} ms; remus_graph host {timed(lambda: S.remus_graph(n, seed=1), 1):.0f} ms, device {timed(lambda: S.remus_graph(n, seed=1, device=dev), 2):.1f} ms")
        print(f"n={n} dim={dim}: connect_knn host {host_knn:.1f} ms, device {dev_knn:.2f} ms "
              f"(search + binning, stream time {e0.elapsed_time(e1) / 5:.2f} ms); "
              f"grid clustering (2 levels) host {host_grid:.1f} ms, device {dev_grid:.2f} ms; nbr {tuple(nbr.shape)}")


if __name__ == "__main__":
    main()
