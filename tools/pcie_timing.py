"""PCIe-inclusive rate of the drop-in call on host arrays."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nimrud_amd import synth
from nimrud_amd.minimal import multiscale
pts, _, edges, radii = synth.make_config("c3_scene_10m")
for rep in range(3):
    t0 = time.perf_counter()
    out = multiscale.process_single_core(pts, pts, edges, radii)
    dt = time.perf_counter() - t0
    print("process_single_core on host arrays: %.1f ms -> %.3g point-scales/s (in %.0f MB, out %.0f MB)"
          % (dt * 1e3, len(pts) * len(edges) / dt, pts.nbytes / 1e6, out.nbytes / 1e6))
