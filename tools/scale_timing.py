"""per-scale timing and pass counts of the fused kernel, both ladder modes (GPU box only)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nimrud_amd import synth, device as nm_device
from nimrud_amd.minimal import multiscale

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
pts, _, edges, radii = synth.make_config("c3_scene_10m", n=n)
dev = torch.from_numpy(pts).cuda()
rt = nm_device.get_runtime()
for per_scale in (True, False):
    for rep in range(2):
        out, info = multiscale.process_gpu(dev, dev, edges, radii, return_info=True, per_scale=per_scale)
    torch.cuda.synchronize()
    print("per_scale" if per_scale else "ladder", [(i.voxels, i.extra_passes, i.leaves) for i in info])
    for s in range(len(edges)):
        rt.lib.nm_profile_begin(rt.ctx)
        for rep in range(3):
            multiscale.process_gpu(dev, dev, edges[s:s+1] if per_scale else edges, radii[s:s+1] if per_scale else radii, per_scale=per_scale)
        ms = (ctypes.c_double * 4)(); l = ctypes.c_int64(0)
        rt.lib.nm_profile_end(rt.ctx, ctypes.byref(ms), ctypes.byref(l))
        print("  scale", s, "launches", l.value, "keys %.3f index %.3f kernel %.3f ms (sum over launches)" % (ms[0], ms[1], ms[2]))
        if not per_scale:
            break
