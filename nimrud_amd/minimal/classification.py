"""
classification of the multiscale features on the GPU.

nimrud/minimal/classification.py is a stub ("stub for classification applications for the multiscale
features", :1-3).  the classifier the reference actually applies to these feature matrices is
sklearn's RandomForestClassifier (prototypes/apc.py:1463) through predict / predict_proba
(apc.py:1008,1022,1034,1745,1752).  `ForestModel` is a fitted forest flattened to arrays in HBM and
evaluated per point by nm_forest_eval: features cast to fp32, `x[feature] <= threshold` goes left,
class distribution of the reached leaf averaged over trees - sklearn's arithmetic.
"""

import ctypes

import numpy as np
import torch

from nimrud_amd import _ffi
from nimrud_amd import device as _device


class ForestModel(object):
    """a random forest as flat node arrays.
        left, right   int32 (nodes,)  absolute child node, -1 at a leaf
        feature       int32 (nodes,)  split feature (0 at leaves)
        threshold     f64   (nodes,)
        value         f64   (nodes, C) class distribution of the node, rows sum to 1
        roots         int32 (trees,)  root node of each tree
        classes       (C,) labels returned by predict
    """

    FIELDS = ("left", "right", "feature", "threshold", "value", "roots", "classes")

    def __init__(self, left, right, feature, threshold, value, roots, classes, n_features=None,
                 device=None):
        self.rt = _device.get_runtime(device)
        dev = self.rt.device
        self.left = torch.as_tensor(np.asarray(left, dtype=np.int32)).to(dev)
        self.right = torch.as_tensor(np.asarray(right, dtype=np.int32)).to(dev)
        self.feature = torch.as_tensor(np.asarray(feature, dtype=np.int32)).to(dev)
        self.threshold = torch.as_tensor(np.asarray(threshold, dtype=np.float64)).to(dev)
        self.value = torch.as_tensor(np.ascontiguousarray(value, dtype=np.float64)).to(dev)
        self.roots = torch.as_tensor(np.asarray(roots, dtype=np.int32)).to(dev)
        self.classes = np.asarray(classes)
        self.n_features = int(n_features) if n_features is not None else \
            int(np.asarray(feature).max()) + 1
        if self.value.ndim != 2 or self.value.shape[0] != self.left.shape[0]:
            raise ValueError("value must be (nodes, classes)")
        packed, leaf_value, packed_roots = self.pack_nodes(
            np.asarray(left, dtype=np.int32), np.asarray(right, dtype=np.int32),
            np.asarray(feature, dtype=np.int32), np.asarray(threshold, dtype=np.float64),
            np.ascontiguousarray(value, dtype=np.float64), np.asarray(roots, dtype=np.int32))
        self.packed = torch.from_numpy(packed.view(np.uint8)).to(dev)
        # rows of 8 doubles (64 bytes, zero-padded) for up to 8 classes: a leaf's distribution is one aligned cache
        # line on the device (nm_forest::leaf_stride); packed rows otherwise
        self.leaf_stride = 8 if leaf_value.shape[1] <= 8 else leaf_value.shape[1]
        padded = np.zeros((leaf_value.shape[0], self.leaf_stride), dtype=np.float64)
        padded[:, :leaf_value.shape[1]] = leaf_value
        self.leaf_value = torch.from_numpy(padded).to(dev)
        self.packed_roots = torch.from_numpy(packed_roots).to(dev)
        packed8 = self.pack_nodes8(packed, self.n_features)
        self.packed8 = torch.from_numpy(packed8.view(np.uint8)).to(dev) if packed8 is not None else None
        self._c = _ffi.NmForest(
            d_left=self.left.data_ptr(), d_right=self.right.data_ptr(),
            d_feature=self.feature.data_ptr(), d_threshold=self.threshold.data_ptr(),
            d_value=self.value.data_ptr(), d_roots=self.roots.data_ptr(),
            n_nodes=self.left.shape[0], n_trees=self.roots.shape[0],
            n_classes=self.value.shape[1], n_features=self.n_features,
            d_packed=self.packed.data_ptr(), d_leaf_value=self.leaf_value.data_ptr(),
            d_packed_roots=self.packed_roots.data_ptr(), n_leaves=self.leaf_value.shape[0],
            leaf_stride=self.leaf_stride, d_packed8=self.packed8.data_ptr() if self.packed8 is not None else None)

    @staticmethod
    def pack_nodes(left, right, feature, threshold, value, roots):
        """model preprocessing on the host (once per model, not on the data path): renumber the nodes
        breadth-first so that siblings are adjacent (right = left + 1) and emit one 16-byte record per
        node {f64 threshold, i32 left, i32 feature}; leaves get left = -1 and feature = row of their
        class distribution in the returned leaf table."""
        rec = np.zeros(len(left), dtype=np.dtype([("threshold", "<f8"), ("left", "<i4"),
                                                  ("feature", "<i4")]))
        leaf_rows = []
        new_roots = np.zeros(len(roots), dtype=np.int32)
        nxt = 0
        for t, root in enumerate(roots):
            new_roots[t] = nxt
            queue = [(int(root), nxt)]
            nxt += 1
            while queue:
                old, new = queue.pop(0)
                if left[old] < 0:
                    rec[new] = (0.0, -1, len(leaf_rows))
                    leaf_rows.append(value[old])
                else:
                    rec[new] = (threshold[old], nxt, feature[old])
                    queue.append((int(left[old]), nxt))
                    queue.append((int(right[old]), nxt + 1))
                    nxt += 2
        return rec, np.ascontiguousarray(np.stack(leaf_rows)), new_roots

    @staticmethod
    def pack_nodes8(rec, n_features):
        """the renumbered nodes of pack_nodes as 8-byte records {fp32 threshold, uint32 packed} (struct
        nm_forest::d_packed8).  sklearn sends a row left when (double)(float)x <= threshold; for an fp32
        x that is the same as x <= the largest fp32 not above the threshold, so the threshold is stored
        rounded DOWN to fp32 and the comparison done in fp32: bit-identical decisions, half the bytes per
        node visit.  packed: left child << 13 | feature << 8 (the kernel ORs its lane's byte offset into the
        low bits and has the LDS address of the feature value); a leaf has bit 31 set and its row << 13.
        None when the forest does not fit the fields (more than 32 features or 2^18 nodes)."""
        if n_features > 32 or len(rec) >= (1 << 18):
            return None
        thr = rec["threshold"].astype(np.float64)
        t32 = thr.astype(np.float32)
        above = t32.astype(np.float64) > thr
        t32[above] = np.nextafter(t32[above], np.float32(-np.inf))
        assert np.all(t32.astype(np.float64) <= thr)
        leaf = rec["left"] < 0
        packed = np.where(leaf, np.uint32(1 << 31) | (rec["feature"].astype(np.uint32) << np.uint32(13)),
                          (rec["left"].astype(np.uint32) << np.uint32(13)) |
                          (rec["feature"].astype(np.uint32) << np.uint32(8))).astype(np.uint32)
        out = np.zeros(len(rec), dtype=np.dtype([("threshold", "<f4"), ("packed", "<u4")]))
        out["threshold"] = t32
        out["packed"] = packed
        return out

    @staticmethod
    def flatten_sklearn(clf):
        """arrays of a fitted sklearn RandomForestClassifier (host side, no GPU needed)."""
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
        return dict(left=np.concatenate(left).astype(np.int32),
                    right=np.concatenate(right).astype(np.int32),
                    feature=np.concatenate(feature).astype(np.int32),
                    threshold=np.concatenate(threshold).astype(np.float64),
                    value=np.concatenate(value).astype(np.float64),
                    roots=np.asarray(roots, dtype=np.int32), classes=np.asarray(clf.classes_),
                    n_features=int(clf.n_features_in_))

    @classmethod
    def from_sklearn(cls, clf, device=None):
        return cls(device=device, **cls.flatten_sklearn(clf))

    @classmethod
    def from_arrays(cls, arrays, device=None):
        kw = {k: arrays[k] for k in cls.FIELDS}
        if "n_features" in arrays:
            kw["n_features"] = int(arrays["n_features"])
        return cls(device=device, **kw)

    def _eval(self, features, want_proba, want_label):
        as_torch = isinstance(features, torch.Tensor)
        rt, x = _device.as_cloud(features, self.rt.device)
        if x.shape[1] < self.n_features:
            raise ValueError("feature matrix has %d columns, the forest needs %d"
                             % (x.shape[1], self.n_features))
        n = x.shape[0]
        proba = torch.empty((n, self._c.n_classes), dtype=torch.float64, device=rt.device) \
            if want_proba else None
        label = torch.empty(n, dtype=torch.int32, device=rt.device) if want_label else None
        rt.check(rt.lib.nm_forest_eval(rt.ctx, ctypes.byref(self._c), _device.ptr(x), n,
                                       _device.row_stride(x), _device.ptr(proba),
                                       _device.ptr(label), rt.stream()))
        return proba, label, as_torch

    def predict_proba(self, features):
        """(N, C) class probabilities: mean over trees of the leaf distributions."""
        proba, _, as_torch = self._eval(features, True, False)
        return proba if as_torch else proba.cpu().numpy()

    def predict(self, features):
        """(N,) class labels (argmax of predict_proba, first maximum wins)."""
        _, label, as_torch = self._eval(features, False, True)
        if as_torch:
            classes = torch.as_tensor(self.classes, device=label.device)
            return classes[label.to(torch.int64)]
        return self.classes[label.cpu().numpy()]


# ---- training-side glue --------------------------------------------------
# host logic around the GPU evaluator: how the reference's workbench gets from a labelled feature matrix
# to a fitted forest and a score.  training itself stays in scikit-learn, as in the reference
# (prototypes/apc.py:1463, clf.fit at :969); only evaluation is on the hot path.

def balanced_split(labels, seed=0):
    """balanced training / validation index sets (prototypes/apc.py:895-917): with m = floor(half the
    population of the rarest class), every class contributes m rows to each set, drawn without
    replacement.  labels are integers 0..C-1; returns (train_idx, valid_idx)."""
    labels = np.asarray(labels)
    rs = np.random.RandomState(seed)
    classes = np.unique(labels)
    m = int(min((labels == c).sum() for c in classes) // 2)
    train, valid = [], []
    for c in classes:
        idx = rs.permutation(np.nonzero(labels == c)[0])
        train.append(idx[:m])
        valid.append(idx[m:2 * m])
    return np.concatenate(train), np.concatenate(valid)


def train_forest(features, labels, n_estimators=32, max_depth=12, criterion="gini", bootstrap=True,
                 n_jobs=6, random_state=0, device=None):
    """fit sklearn's RandomForestClassifier (the reference's 'rf' choice, apc.py:1463: n_estimators,
    criterion, bootstrap, n_jobs=6) on host arrays and return (ForestModel on the GPU, fitted clf)."""
    from sklearn.ensemble import RandomForestClassifier
    x = features.cpu().numpy() if isinstance(features, torch.Tensor) else np.asarray(features)
    y = labels.cpu().numpy() if isinstance(labels, torch.Tensor) else np.asarray(labels)
    clf = RandomForestClassifier(n_estimators=n_estimators, max_depth=max_depth, criterion=criterion,
                                 bootstrap=bootstrap, n_jobs=n_jobs, random_state=random_state)
    clf.fit(x, y)
    return ForestModel.from_sklearn(clf, device=device), clf


def confusion_matrix(predicted, truth, n_classes=None):
    """full multiclass confusion matrix (prototypes/ml.py:521-552 mc_confusion): entry [row, col] =
    number of points of known class `col` that received label `row`.  integer labels 0..n-1; computed on
    the device when given GPU tensors."""
    if isinstance(predicted, torch.Tensor) or isinstance(truth, torch.Tensor):
        p = torch.as_tensor(predicted).to(torch.int64).reshape(-1)
        t = torch.as_tensor(truth).to(p.device).to(torch.int64).reshape(-1)
        n = int(n_classes) if n_classes is not None else int(max(p.max().item(), t.max().item())) + 1
        flat = torch.bincount(p * n + t, minlength=n * n)
        return flat.reshape(n, n).cpu().numpy().astype(np.float64)
    p = np.asarray(predicted).astype(np.int64).ravel()
    t = np.asarray(truth).astype(np.int64).ravel()
    n = int(n_classes) if n_classes is not None else int(max(p.max(), t.max())) + 1
    return np.bincount(p * n + t, minlength=n * n).reshape(n, n).astype(np.float64)


FUSED_MAX_FEATURES, FUSED_MAX_CLASSES = 20, 8       # limits of nm_set_forest_output


def classify_cloud(cloud, edge_lengths, radii, model, fused=True, want_proba=False, out=None, **kwargs):
    """features + forest in one go for a cloud resident on the GPU (BASELINE config 5: "random-forest
    classifier evaluation fused after feature assembly"): returns (labels int32 GPU tensor of class
    positions, (N, 4*S) feature tensor) - and the (N, C) probabilities in between when want_proba.
    fused=True evaluates the forest inside the search kernel, right behind each row's last scale
    (nm_set_forest_output): the wave that has just finished a row reads it back out of L2 and its 64
    spatially neighbouring lanes walk the trees together.  forests outside the fused path's limits (more
    than 20 features or 8 classes) and fused=False evaluate the finished matrix with nm_forest_eval; the
    numbers are the same.  other keyword arguments go to process_gpu (strict, knn_min, ...); with
    return_info=True the per-scale ScaleInfo list is appended to the returned tuple."""
    from nimrud_amd.minimal import multiscale
    n_features = 4 * len(edge_lengths)
    # return_info=True (process_gpu's per-scale ScaleInfo list) is passed on and comes back as the last element
    want_info = bool(kwargs.pop("return_info", False))
    can_fuse = (fused and model.packed8 is not None and model.n_features == n_features and
                n_features <= FUSED_MAX_FEATURES and model._c.n_classes <= FUSED_MAX_CLASSES and
                not kwargs.get("verbose") and not kwargs.get("per_scale"))

    def features_of(q):
        res = multiscale.process_gpu(q, q, edge_lengths, radii, out=out, return_info=want_info, **kwargs)
        return res if want_info else (res, None)

    def result(label, proba, feats, info):
        parts = (label, proba, feats) if want_proba else (label, feats)
        return parts + (info,) if want_info else parts

    if not can_fuse:
        feats, info = features_of(cloud)
        proba, label, _ = model._eval(feats, want_proba, True)
        return result(label, proba, feats, info)
    rt, dev_cloud = _device.as_cloud(cloud)
    n = dev_cloud.shape[0]
    label = torch.empty(n, dtype=torch.int32, device=rt.device)
    proba = torch.empty((n, model._c.n_classes), dtype=torch.float64, device=rt.device) \
        if want_proba else None
    rt.check(rt.lib.nm_set_forest_output(rt.ctx, ctypes.byref(model._c), _device.ptr(proba),
                                         model._c.n_classes, _device.ptr(label)))
    try:
        feats, info = features_of(dev_cloud)
    finally:
        rt.check(rt.lib.nm_set_forest_output(rt.ctx, None, None, 0, None))
    return result(label, proba, feats, info)
