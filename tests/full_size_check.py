"""one-off parity check of a whole named configuration at FULL size against the plain-C oracle (every row of
every scale).  not collected by pytest (minutes, tens of GB of host memory); run on the GPU box:
    python tests/full_size_check.py c4_lidar_50m [n_points]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from nimrud_amd import synth
from nimrud_amd.minimal import multiscale
from oracle import nimrud_oracle as oracle

name = sys.argv[1] if len(sys.argv) > 1 else "c4_lidar_50m"
n = int(sys.argv[2]) if len(sys.argv) > 2 else None
pts, _, edges, radii = synth.make_config(name, n=n)
print("%s: %d points, %d scales" % (name, len(pts), len(edges)), flush=True)
dev = torch.from_numpy(pts).cuda()
t0 = time.time()
got = multiscale.process_gpu(dev, dev, edges, radii).cpu().numpy()
print("gpu %.1f s (incl. download)" % (time.time() - t0), flush=True)
lo, hi = pts.min(0), pts.max(0)
threads = min(64, os.cpu_count() or 1)
worst_eig, worst_cen, bad_pop = 0.0, 0.0, 0
for s, (e, r) in enumerate(zip(edges, radii)):
    t0 = time.time()
    want = oracle.one_scale_c(pts, pts, e, r, threads=threads, bounds=(lo, hi))
    g = got[:, 4 * s:4 * s + 4]
    bad_pop += int((g[:, 0] != want[:, 0]).sum())
    eig = np.abs(g[:, 2:] - want[:, 2:]) - 1e-5 * np.abs(want[:, 2:])
    cen = np.abs(g[:, 1] - want[:, 1]) - 1e-9 * np.abs(want[:, 1])
    worst_eig = max(worst_eig, float(eig.max()))
    worst_cen = max(worst_cen, float(cen.max()))
    print("scale %d (e=%g): populations differing %d, worst eigen excess %.2e, worst centroid excess %.2e, "
          "oracle %.0f s" % (s, e, int((g[:, 0] != want[:, 0]).sum()), eig.max(), cen.max(), time.time() - t0),
          flush=True)
ulp = 64 * np.spacing(np.abs(pts).max())
ok = bad_pop == 0 and worst_eig <= 1e-9 and worst_cen <= ulp + 1e-12
print("RESULT %s: %d x %d point-scales, populations exact: %s, eigen within 1e-5 rel + 1e-9: %s, "
      "centroid within 1e-9 rel + 64 ulp: %s" % ("PASS" if ok else "FAIL", len(pts), len(edges), bad_pop == 0,
                                                  worst_eig <= 1e-9, worst_cen <= ulp + 1e-12))
sys.exit(0 if ok else 1)
