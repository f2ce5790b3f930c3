"""config 4 at full size on one GPU: 50M-point power-law LiDAR cloud, 5 scales, kNN fallback k_min = 8."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nimrud_amd import synth
from nimrud_amd.minimal import multiscale

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
t0 = time.time()
pts, _, edges, radii = synth.make_config("c4_lidar_50m", n=n)
print("generated %d points in %.0f s" % (n, time.time() - t0), flush=True)
dev = torch.from_numpy(pts).cuda()
del pts
out = torch.empty((n, 4 * len(edges)), dtype=torch.float64, device="cuda")
for knn in (0, 8):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, info = multiscale.process_gpu(dev, dev, edges, radii, out=out, knn_min=knn, return_info=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("knn_min=%d: %.1f ms per step, %.3g point-scales/s" % (knn, dt * 1e3, n * len(edges) / dt), flush=True)
print([(i.voxels, i.degenerate, i.extra_passes, i.leaves) for i in info])
pop = out[:, ::4]
print("population min/mean/max per scale:", [(int(pop[:, s].min()), float(pop[:, s].mean()), int(pop[:, s].max())) for s in range(len(edges))])
print("rows with population < 8 per scale:", [int((pop[:, s] < 8).sum()) for s in range(len(edges))])
print("peak HBM allocated: %.1f GB" % (torch.cuda.max_memory_allocated() / 1e9))
