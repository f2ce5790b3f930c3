"""vector-field operator timing (GPU box only): config-2 cloud, 4 attribute columns, 3 scales."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nimrud_amd import synth
from nimrud_amd.minimal import fields
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
pts, _, edges, radii = synth.make_config("c2_scene_1m", n=n)
dev = torch.from_numpy(pts).cuda()
attr = torch.rand((n, 4), dtype=torch.float64, device="cuda")
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = fields.vector_field_mean_gpu(dev, dev, attr, edges, radii)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("%d points x %d scales x 4 columns: %.2f ms = %.3g point-scales/s" % (n, len(edges), dt * 1e3, n * len(edges) / dt))
