"""is a ladder call capturable in a hipGraph?  (GPU box only)  a step has no host synchronisation since the
lattices are built on the device; this captures one process_gpu call with torch.cuda.graph, replays it - also on
fresh data in the same buffers - and times eager against replay.  progress goes to gpurun_out/graph_probe.log."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
from nimrud_amd import synth, device
from nimrud_amd.minimal import multiscale

os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
LOG = open(os.path.join(REPO, "gpurun_out", "graph_probe.log"), "a")


def say(*a):
    msg = " ".join(str(x) for x in a)
    print(msg, flush=True)
    LOG.write(msg + "\n")
    LOG.flush()
    os.fsync(LOG.fileno())


names = sys.argv[1:] or ["c1_uniform_100k", "c2_scene_1m", "c3_scene_10m"]
for name in names:
    steps = {"c1_uniform_100k": 200, "c2_scene_1m": 50}.get(name, 10)
    pts, _, edges, radii = synth.make_config(name)
    cloud = torch.from_numpy(pts).cuda()
    out = torch.empty((len(pts), 4 * len(edges)), dtype=torch.float64, device="cuda")
    want = multiscale.process_gpu(cloud, cloud, edges, radii).clone()
    rt = device.get_runtime()
    torch.cuda.synchronize()
    say(name, "eager reference done")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        multiscale.process_gpu(cloud, cloud, edges, radii, out=out)       # warm the workspace on this stream
    torch.cuda.synchronize()
    say(name, "side-stream run done")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        multiscale.process_gpu(cloud, cloud, edges, radii, out=out)
    torch.cuda.synchronize()
    say(name, "captured")
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    say(name, "replay bit-identical:", torch.equal(out, want))
    # a different cloud in the same buffer: the replay measures ITS extrema and builds ITS lattices
    pts2 = np.ascontiguousarray(pts * 0.5 + 1.0)
    cloud2 = torch.from_numpy(pts2).cuda()
    want2 = multiscale.process_gpu(cloud2, cloud2, edges, radii).clone()
    torch.cuda.synchronize()
    say(name, "eager on the second cloud done")
    cloud.copy_(cloud2)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    say(name, "replay on new data bit-identical:", torch.equal(out, want2))
    cloud.copy_(torch.from_numpy(pts).cuda())

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3
    eager = timed(lambda: multiscale.process_gpu(cloud, cloud, edges, radii, out=out))
    say(name, "eager %.3f ms/step" % eager)
    replay = timed(g.replay)
    rt.check_async(wait=True)
    say(name, "graph replay %.3f ms/step" % replay)
    del g
