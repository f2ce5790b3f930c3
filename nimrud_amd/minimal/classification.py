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
        self.leaf_value = torch.from_numpy(leaf_value).to(dev)
        self.packed_roots = torch.from_numpy(packed_roots).to(dev)
        self._c = _ffi.NmForest(
            d_left=self.left.data_ptr(), d_right=self.right.data_ptr(),
            d_feature=self.feature.data_ptr(), d_threshold=self.threshold.data_ptr(),
            d_value=self.value.data_ptr(), d_roots=self.roots.data_ptr(),
            n_nodes=self.left.shape[0], n_trees=self.roots.shape[0],
            n_classes=self.value.shape[1], n_features=self.n_features,
            d_packed=self.packed.data_ptr(), d_leaf_value=self.leaf_value.data_ptr(),
            d_packed_roots=self.packed_roots.data_ptr(), n_leaves=self.leaf_value.shape[0],
            reserved=0)

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
            return label
        return self.classes[label.cpu().numpy()]
