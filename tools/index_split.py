"""stage times of single-scale ladders on the config 3 cloud: how the fused index build's time splits over the
scales (GPU box only)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nimrud_amd import synth, device
from nimrud_amd.minimal import multiscale
pts, _, edges, radii = synth.make_config("c3_scene_10m")
cloud = torch.from_numpy(pts).cuda()
rt = device.get_runtime()
ladders = [[e] for e in edges] + [edges[:2], edges[2:], edges]
for ed in ladders:
    ra = [3 * e for e in ed]
    out = torch.empty((len(pts), 4 * len(ed)), dtype=torch.float64, device="cuda")
    for _ in range(3):
        multiscale.process_gpu(cloud, cloud, ed, ra, out=out)
    torch.cuda.synchronize()
    rt.lib.nm_profile_begin(rt.ctx)
    for _ in range(10):
        _, info = multiscale.process_gpu(cloud, cloud, ed, ra, out=out, return_info=True)
    ms = (ctypes.c_double * 4)(); n = ctypes.c_int64(0)
    rt.check(rt.lib.nm_profile_end(rt.ctx, ctypes.byref(ms), ctypes.byref(n)))
    print("edges %s: order %.3f index %.3f search %.3f ms; leaves %s voxels %s" % (
        ed, ms[0] / 10, ms[1] / 10, ms[2] / 10, [i.leaves for i in info], [i.voxels for i in info]), flush=True)
