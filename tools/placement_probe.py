"""does the placement of the output matrix / workspace in HBM change the kernel time? (GPU box only)
runs the benchmark ladder several times in one process with the feature matrix carved out of a big buffer at
different byte offsets, and reports the fused-kernel time of each."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nimrud_amd import synth, device as nm_device
from nimrud_amd.minimal import multiscale

pts, _, edges, radii = synth.make_config("c3_scene_10m")
dev = torch.from_numpy(pts).cuda()
n = dev.shape[0]
rt = nm_device.get_runtime()
big = torch.empty(n * 20 + (64 << 20), dtype=torch.float64, device="cuda")


def kernel_ms(out, reps=10):
    for _ in range(3):
        multiscale.process_gpu(dev, dev, edges, radii, out=out)
    torch.cuda.synchronize()
    rt.lib.nm_profile_begin(rt.ctx)
    for _ in range(reps):
        multiscale.process_gpu(dev, dev, edges, radii, out=out)
    ms = (ctypes.c_double * 4)(); l = ctypes.c_int64(0)
    rt.lib.nm_profile_end(rt.ctx, ctypes.byref(ms), ctypes.byref(l))
    return ms[2] / reps, ms[1] / reps, ms[0] / reps

for off in (0, 16, 512, 4096, 1 << 16, 1 << 20, (1 << 21) + 256, 3 << 20, 17 << 20, 0):
    out = big[off // 8: off // 8 + n * 20].view(n, 20)
    k, i, o = kernel_ms(out)
    print("out offset %9d B (addr %% 2MiB = %7d): kernels %.3f ms, index %.3f, order %.3f"
          % (off, out.data_ptr() % (2 << 20), k, i, o), flush=True)

# the workspace (sorted copy, indexes) and the cloud itself at different places
import random
random.seed(1)
keep = []
for trial in range(8):
    rt.release_workspace()
    torch.cuda.empty_cache()
    keep.append(torch.empty(random.randrange(1, 64) * 1000003, dtype=torch.uint8, device="cuda"))
    out = torch.empty((n, 20), dtype=torch.float64, device="cuda")
    k, i, o = kernel_ms(out)
    w = rt.workspace(1)
    print("trial %d: workspace at %% 2MiB = %7d, out at %% 2MiB = %7d: kernels %.3f ms, index %.3f, order %.3f"
          % (trial, w.data_ptr() % (2 << 20), out.data_ptr() % (2 << 20), k, i, o), flush=True)
    del out


# the workspace carved out of one big allocation at different 2 MiB-aligned offsets (same physical pool):
# which address bit, if any, decides?
rt.release_workspace()
del keep
torch.cuda.empty_cache()
need = rt.lib.nm_multiscale_workspace_bytes  # sized through a dry run of process_gpu below
out = torch.empty((n, 20), dtype=torch.float64, device="cuda")
multiscale.process_gpu(dev, dev, edges, radii, out=out)
wbytes = rt._work.numel()
rt.release_workspace()
torch.cuda.empty_cache()
pool = torch.empty(wbytes + (2 << 30), dtype=torch.uint8, device="cuda")
base = pool.data_ptr()
for off in (0, 2 << 20, 4 << 20, 8 << 20, 16 << 20, 32 << 20, 64 << 20, 128 << 20, 256 << 20, 512 << 20,
            1 << 30, (1 << 30) + (2 << 20), 3 << 29):
    rt._work = pool[off:off + wbytes]
    k, i, o = kernel_ms(out)
    print("workspace at pool + %10d (address bits 21..31 = %s): kernels %.3f ms, index %.3f"
          % (off, format(((base + off) >> 21) & 0x7FF, "011b"), k, i), flush=True)
