"""
generate tests/golden/g6_forest_c5.npz: the classifier of BASELINE config 5 (SURVEY.md section 8d:
"C5 = C3 cloud + RF: 32 trees, max_depth 12, 5 classes, trained in-container with sklearn on 2e5 labelled
rows (label = generating primitive), random_state=0; exported arrays are the fixture").

runs in the authoring container (needs scikit-learn; ~2 minutes, ~6 GB):

    python tests/golden/make_forest_c5.py

what it does
  1. the config 5 cloud: synth.make_config("c5_scene_10m_rf") - the 10 M-point config 3 scene with
     five-class labels (open ground / pole / sphere / ground at a pole / ground under a sphere).
  2. the features of 2e5 training rows and 4 096 evaluation rows AGAINST THE FULL 10 M-POINT SEARCH CLOUD,
     from the oracle's C restatement (oracle/lattice_oracle.c, pinned against the reference's golden
     vectors in tests/test_oracle.py) - the reference's own Python would need days for this search cloud.
  3. sklearn's RandomForestClassifier - the classifier the reference applies to these feature matrices
     (prototypes/apc.py:1463: n_estimators, criterion, bootstrap) - fitted on the training rows.
  4. the fitted trees flattened to arrays (the layout of nimrud_amd.minimal.classification.ForestModel),
     plus, for the evaluation rows: their row numbers, features (fp64, as the oracle gave them),
     clf.predict_proba and clf.predict on exactly those features.
the fixture is data only: model arrays, inputs and sklearn's outputs.
"""

import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from nimrud_amd import synth                                  # noqa: E402
from oracle import nimrud_oracle as oracle                    # noqa: E402

N_TRAIN = 200_000
N_EVAL = 4096


def main():
    from sklearn.ensemble import RandomForestClassifier
    t0 = time.time()
    points, labels, edges, radii = synth.make_config("c5_scene_10m_rf")
    print("cloud", points.shape, "class counts", np.bincount(labels), "%.0f s" % (time.time() - t0))
    rs = np.random.RandomState(0)
    rows = rs.choice(len(points), N_TRAIN + N_EVAL, replace=False)
    train_rows, eval_rows = rows[:N_TRAIN], np.sort(rows[N_TRAIN:])
    bounds = (points.min(0), points.max(0))
    feats = oracle.process_c(points[rows], points, edges, radii, bounds=bounds)
    f_train = feats[:N_TRAIN]
    f_eval = oracle.process_c(points[eval_rows], points, edges, radii, bounds=bounds)
    print("features", feats.shape, "%.0f s" % (time.time() - t0))
    clf = RandomForestClassifier(n_estimators=32, max_depth=12, criterion="gini", bootstrap=True,
                                 random_state=0, n_jobs=8)
    clf.fit(f_train, labels[train_rows])
    left, right, feature, threshold, value, roots = [], [], [], [], [], []
    base = 0
    for est in clf.estimators_:
        t = est.tree_
        roots.append(base)
        left.append(np.where(t.children_left >= 0, t.children_left + base, -1))
        right.append(np.where(t.children_right >= 0, t.children_right + base, -1))
        feature.append(np.maximum(t.feature, 0))
        threshold.append(t.threshold)
        v = t.value[:, 0, :]
        value.append(v / v.sum(1)[:, None])
        base += t.node_count
    proba = clf.predict_proba(f_eval)
    label = clf.predict(f_eval).astype(np.int32)
    np.savez_compressed(
        os.path.join(HERE, "g6_forest_c5.npz"),
        left=np.concatenate(left).astype(np.int32), right=np.concatenate(right).astype(np.int32),
        feature=np.concatenate(feature).astype(np.int32),
        threshold=np.concatenate(threshold).astype(np.float64),
        value=np.concatenate(value).astype(np.float64), roots=np.asarray(roots, dtype=np.int32),
        classes=clf.classes_.astype(np.int32), n_features=np.int32(f_train.shape[1]),
        eval_rows=eval_rows.astype(np.int64), x=f_eval, proba=proba, label=label,
        truth=labels[eval_rows].astype(np.int32),
        edges=np.asarray(edges), radii=np.asarray(radii))
    depth = max(est.tree_.max_depth for est in clf.estimators_)
    print("g6_forest_c5: %d nodes, max depth %d, classes %s, accuracy on the evaluation rows %.4f, %.0f s"
          % (base, depth, clf.classes_, float((label == labels[eval_rows]).mean()), time.time() - t0))


if __name__ == "__main__":
    main()
