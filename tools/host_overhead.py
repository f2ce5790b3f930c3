import time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from nimrud_amd import synth, device as _device
from nimrud_amd.minimal import multiscale
pts,_,edges,radii = synth.make_config("c3_scene_10m", n=2_000_000)
dev = torch.from_numpy(pts).cuda()
out = torch.empty((dev.shape[0], 20), dtype=torch.float64, device="cuda")
rt,_ = _device.as_cloud(dev)
for _ in range(5): multiscale.process_gpu(dev, dev, edges, radii, out=out)
torch.cuda.synchronize()
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    multiscale.process_gpu(dev, dev, edges, radii, out=out)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(22)
