"""
FlexCloud: points plus named per-point "assets" (features, labels, ...) keyed by index sets, with a
set-intersection join.  mirror of nimrud/utils/point_clouds.py (:15-159) - same constructor, attributes
(`corner`, `points`, `num_points`, `id_index`, `assets`) and methods (`add_asset`, `intersection`, `take`),
same ValueErrors - with one addition: `FlexCloud(points, device="cuda")` keeps the points and every asset
as torch tensors in HBM, so the step either side of the hot path (load -> features -> labels -> export)
never leaves the device.  without `device` it is a host container of numpy arrays exactly like the
reference's.  this is bookkeeping, not arithmetic: no kernels of the library are involved.

additions for the pipeline: `add_multiscale_features`, `add_labels_from`,
and the file formats the reference's workbench reads and writes (.npy arrays, delimited ASCII:
prototypes/apc.py:221-229, 1802).
"""

import numpy as np
import torch


def _is_torch(x):
    return isinstance(x, torch.Tensor)


class FlexCloud(object):
    """given a 3d point cloud as a 2d array, shift its points close to the origin and track its
    features; supplemental information is stored as "assets": 1d or 2d arrays aligned with an index
    array into the cloud (point_clouds.py:16-48)."""

    def __init__(self, input_cloud, device=None):
        if input_cloud.ndim != 2:
            raise ValueError("input point cloud must be a 2D array")
        if input_cloud.shape[1] != 3:
            raise ValueError("must be initialized with a 3D point cloud")
        self.device = torch.device(device) if device is not None else None
        if self.device is not None:
            input_cloud = torch.as_tensor(input_cloud).to(self.device)
        elif _is_torch(input_cloud):
            self.device = input_cloud.device
        # bring the point cloud in to the origin (point_clouds.py:58-60)
        self.corner = input_cloud[0].clone() if _is_torch(input_cloud) else input_cloud[0]
        self.points = input_cloud - self.corner
        self.num_points = input_cloud.shape[0]
        self.id_index = torch.arange(self.num_points, device=self.device) if _is_torch(input_cloud) \
            else np.arange(self.num_points)
        self.assets = {}

    # ------------------------------------------------------------------------------------------------

    def _coerce(self, array, integer=False):
        if self.device is None:
            return np.asarray(array)
        t = torch.as_tensor(array).to(self.device)
        return t.to(torch.int64) if integer else t

    def add_asset(self, asset_array, index_array, asset_name, meta=None):
        """add a new asset array to the cloud's asset index.  the index array need not be sorted or
        unique; it is stored sorted and unique (first occurrence wins) with the asset aligned to it
        (point_clouds.py:69-111)."""
        if asset_name in self.assets:
            raise ValueError("asset {} already exists in asset dictionary".format(asset_name))
        asset_array = self._coerce(asset_array)
        index_array = self._coerce(index_array, integer=True)
        asset_array, index_array = self._validate_asset(asset_array, index_array)
        self.assets[asset_name] = {"asset": asset_array, "index": index_array, "meta": meta}

    def _validate_asset(self, asset_array, index_array):
        if asset_array.ndim > 2:
            raise ValueError("asset array has too many dimensions")
        n_index = index_array.numel() if _is_torch(index_array) else index_array.size
        if asset_array.shape[0] != n_index:
            raise ValueError("asset and index arrays misaligned")
        if n_index and (int(index_array.min()) < 0 or int(index_array.max()) >= self.num_points):
            raise ValueError("index array addresses outside the extant cloud")
        if _is_torch(index_array):
            flat = index_array.reshape(-1)
            ordered, perm = torch.sort(flat, stable=True)
            first = torch.ones_like(ordered, dtype=torch.bool)
            first[1:] = ordered[1:] != ordered[:-1]
            return asset_array.index_select(0, perm[first]), ordered[first]
        unique_indices, index_to_unique = np.unique(index_array, return_index=True)
        return asset_array.take(index_to_unique, axis=0), unique_indices

    def intersection(self, asset_names):
        """intersection of the index sets of the named assets, and the horizontal concatenation of the
        corresponding asset rows (point_clouds.py:115-143)."""
        acc = self.id_index
        for name in asset_names:
            this_index = self.assets[name]["index"]
            if _is_torch(acc):
                acc = acc[torch.isin(acc, this_index, assume_unique=True)]
            else:
                acc = np.intersect1d(acc, this_index, assume_unique=True)
        n = acc.numel() if _is_torch(acc) else acc.size
        blocks = []
        for name in asset_names:
            this_index = self.assets[name]["index"]
            this_asset = self.assets[name]["asset"]
            if _is_torch(acc):
                mask = torch.isin(this_index, acc, assume_unique=True)
                blocks.append(this_asset[mask].reshape(n, -1))
            else:
                mask = np.isin(this_index, acc, assume_unique=True)
                blocks.append(np.compress(mask, this_asset, axis=0).reshape(n, -1))
        if _is_torch(acc):
            common = torch.promote_types(blocks[0].dtype, blocks[-1].dtype) if blocks else None
            for b in blocks:
                common = torch.promote_types(common, b.dtype)
            return acc, torch.cat([b.to(common) for b in blocks], dim=1)
        return acc, np.concatenate(blocks, axis=1)

    def take(self, index_array=None, original_coordinates=True):
        """like ndarray.take(): a subset of the points, in the original coordinates if desired
        (point_clouds.py:147-159)."""
        pts = self.points + self.corner if original_coordinates else self.points
        if index_array is None:
            return pts
        if _is_torch(pts):
            return pts.index_select(0, self._coerce(index_array, integer=True).reshape(-1))
        return pts.take(index_array, axis=0)

    # ---- pipeline conveniences (not in the reference) -----------------------------------------------

    def add_multiscale_features(self, edge_lengths, radii, asset_name="geometry_mso", **kwargs):
        """run the multiscale operator on the whole cloud (query = search = these points, in the shifted
        coordinates, which is kinder to fp64) and store the (N, 4*S) matrix as an asset whose meta
        records the ladder, like the docstring example of the reference (point_clouds.py:29-35)."""
        from nimrud_amd.minimal import multiscale
        if self.device is not None and self.device.type == "cuda":
            pts = self.points.to(torch.float64).contiguous()
            feats = multiscale.process_gpu(pts, pts, edge_lengths, radii, **kwargs)
        else:
            host = self.points.cpu().numpy() if _is_torch(self.points) else np.asarray(self.points)
            feats = multiscale.process_single_core(host, host, edge_lengths, radii, **kwargs)
        self.add_asset(feats, self.id_index, asset_name,
                       meta={"voxel": list(edge_lengths), "scales": list(radii)})
        return self.assets[asset_name]["asset"]

    def add_labels_from(self, model, feature_asset, asset_name="predicted_label"):
        """classify every point that has `feature_asset` with a ForestModel and store the labels."""
        entry = self.assets[feature_asset]
        labels = model.predict(entry["asset"])
        self.add_asset(labels, entry["index"], asset_name, meta={"features": feature_asset})
        return self.assets[asset_name]["asset"]

    # ---- files ----------------------------------------------------------------------------------------

    @classmethod
    def from_file(cls, path, delimiter=None, device=None):
        """a cloud from a numpy binary (.npy) or a delimited ASCII table; the first three columns are
        the geometry, any remaining columns become the asset "columns" (apc.py:221-229)."""
        table = np.load(path, allow_pickle=False) if str(path).endswith(".npy") \
            else np.loadtxt(path, delimiter=delimiter, ndmin=2)
        cloud = cls(np.ascontiguousarray(table[:, :3]), device=device)
        if table.shape[1] > 3:
            cloud.add_asset(np.ascontiguousarray(table[:, 3:]), np.arange(len(table)), "columns")
        return cloud

    def export(self, path, asset_names=(), delimiter=" ", fmt="%.6f", original_coordinates=True):
        """write [x y z | assets...] for the points that carry every named asset: .npy, or delimited
        ASCII the way the reference's workbench writes its coloured clouds (apc.py:1802)."""
        if asset_names:
            index, block = self.intersection(list(asset_names))
            pts = self.take(index, original_coordinates=original_coordinates)
            cols = [pts, block]
        else:
            cols = [self.take(original_coordinates=original_coordinates)]
        host = [c.cpu().numpy() if _is_torch(c) else np.asarray(c) for c in cols]
        table = np.concatenate([h.astype(np.float64).reshape(len(host[0]), -1) for h in host], axis=1)
        if str(path).endswith(".npy"):
            np.save(path, table)
        else:
            np.savetxt(path, table, delimiter=delimiter, fmt=fmt)
        return table.shape
