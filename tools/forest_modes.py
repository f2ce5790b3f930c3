"""config 5's forest stage in its three placements, same process, same data (run on the GPU box):
   nm_set_forest_mode 1 = the search kernel's epilogue, 0 = a row walk of its own from memory (k_forest_ordered),
   2 = a launch of its own with the trees staged through LDS (k_forest_tiles).  labels and probabilities must be
   identical; prints ms per step of classify_cloud for each, and of the features alone.
   python tools/forest_modes.py [n_points]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nimrud_amd import synth, device as nm_device
from nimrud_amd.minimal import multiscale, classification

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
pts, _, edges, radii = synth.make_config("c5_scene_10m_rf", n=n)
arrays = dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                   "g6_forest_c5.npz")))
dev = torch.device("cuda", 0)
cloud = torch.from_numpy(np.ascontiguousarray(pts)).to(dev)
rt = nm_device.get_runtime(dev)
model = classification.ForestModel.from_arrays(arrays, device=dev)
out = torch.empty((n, 4 * len(edges)), dtype=torch.float64, device=dev)

def timed(fn, steps=8, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3

ms_feat = timed(lambda: multiscale.process_gpu(cloud, cloud, edges, radii, out=out))
print("features alone: %.3f ms per step" % ms_feat)
ref = None
for mode in (1, 0, 2):
    rt.check(rt.lib.nm_set_forest_mode(rt.ctx, mode))
    label, proba, feats = classification.classify_cloud(cloud, edges, radii, model, want_proba=True, out=out)
    torch.cuda.synchronize()
    if ref is None:
        ref = (label.clone(), proba.clone())
    else:
        same_l = bool(torch.equal(label, ref[0]))
        same_p = bool(torch.equal(proba, ref[1]))
        print("mode %d: labels identical to the epilogue's: %s, probabilities: %s (max |diff| %.3g)"
              % (mode, same_l, same_p, float((proba - ref[1]).abs().max())))
    ms = timed(lambda: classification.classify_cloud(cloud, edges, radii, model, out=out))
    print("mode %d: %.3f ms per step, forest stage %.3f ms" % (mode, ms, ms - ms_feat))
# the stand-alone evaluator on the finished matrix
for mode in (1, 2):
    rt.check(rt.lib.nm_set_forest_mode(rt.ctx, mode))
    ms = timed(lambda: model._eval(out, False, True))
    p, l, _ = model._eval(out, True, True)
    print("nm_forest_eval with mode %d: %.3f ms; labels identical: %s" % (mode, ms, bool(torch.equal(l, ref[0]))))
rt.check(rt.lib.nm_set_forest_mode(rt.ctx, 1))
