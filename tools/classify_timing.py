"""config 5: features + random-forest evaluation end to end on the GPU (GPU box only).
trains sklearn's RandomForestClassifier (32 trees, depth <= 12) on 2e5 labelled rows of GPU features,
flattens it, and times features + nm_forest_eval on the full cloud."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sklearn.ensemble import RandomForestClassifier
from nimrud_amd import synth
from nimrud_amd.minimal import multiscale, classification

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
pts, labels, edges, radii = synth.make_config("c3_scene_10m", n=n)
dev = torch.from_numpy(pts).cuda()
feats = multiscale.process_gpu(dev, dev, edges, radii)
rows = np.random.RandomState(0).choice(n, 200000, replace=False)
clf = RandomForestClassifier(n_estimators=32, max_depth=12, random_state=0, n_jobs=16)
clf.fit(feats[torch.from_numpy(rows).cuda()].cpu().numpy(), labels[rows])
model = classification.ForestModel.from_sklearn(clf)
print("nodes", model.left.shape[0], "classes", model.classes)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    feats = multiscale.process_gpu(dev, dev, edges, radii)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    lab = model.predict(feats)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("features %.2f ms, forest %.2f ms, end-to-end %.3g classified points/s"
          % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, n / (t2 - t0)))
check = np.random.RandomState(1).choice(n, 20000, replace=False)
want = clf.predict(feats[torch.from_numpy(check).cuda()].cpu().numpy())
got = lab[torch.from_numpy(check).cuda()].cpu().numpy()
print("labels equal to sklearn on 20k rows:", bool(np.array_equal(got, want)), "accuracy", float((got == labels[check]).mean()))
