"""do the (vector-ALU bound) search ladder and the (latency bound) forest walk overlap when they run as separate
kernels on two streams?  features of the c5 cloud on stream A, nm_forest_eval on a finished matrix on stream B."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nimrud_amd import synth
from nimrud_amd.minimal import multiscale, classification

pts, _, edges, radii = synth.make_config("c5_scene_10m_rf")
dev = torch.device("cuda", 0)
cloud = torch.from_numpy(pts).to(dev)
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g6_forest_c5.npz"))
model = classification.ForestModel.from_arrays({k: g[k] for k in g.files}, device=dev)
out_a = torch.empty((cloud.shape[0], 20), dtype=torch.float64, device=dev)
out_b = torch.empty_like(out_a)
multiscale.process_gpu(cloud, cloud, edges, radii, out=out_b)
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

def feats():
    with torch.cuda.stream(s1):
        multiscale.process_gpu(cloud, cloud, edges, radii, out=out_a)

def forest():
    with torch.cuda.stream(s2):
        model._eval(out_b, False, True)

def both():
    feats()
    forest()

print("features alone   %.3f ms" % timed(feats))
print("forest alone     %.3f ms" % timed(forest))
print("both, 2 streams  %.3f ms" % timed(both))
