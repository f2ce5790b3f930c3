"""soak: 300 ladder calls of mixed sizes, scale counts and fallback settings in one process; memory stays flat
(GPU box only):  python tools/soak.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from nimrud_amd import synth
from nimrud_amd.minimal import multiscale
rs = np.random.RandomState(5)
base = torch.cuda.memory_allocated()
ref = None
t0 = time.time()
for it in range(300):
    n = int(rs.randint(1000, 400000))
    pts, _ = synth.scene_cloud(n, extent=float(rs.uniform(3, 40)), n_poles=5, n_spheres=2, seed=int(rs.randint(1 << 30)))
    dev = torch.from_numpy(pts).cuda()
    k = int(rs.randint(1, 6))
    edges = [0.05 * 2 ** i for i in range(k)]
    out = multiscale.process_gpu(dev, dev, edges, [3 * e for e in edges], knn_min=int(rs.choice([0, 0, 6])))
    if it % 50 == 0:
        torch.cuda.synchronize()
        print(it, n, k, "allocated %.1f MB reserved %.1f MB" % (torch.cuda.memory_allocated() / 1e6, torch.cuda.memory_reserved() / 1e6), flush=True)
    assert torch.isfinite(out).all()
torch.cuda.synchronize()
print("300 mixed calls in %.1f s; finite everywhere" % (time.time() - t0))
