"""
FlexCloud (nimrud_amd/utils/point_clouds.py): the reference's own tests
(nimrud/utils/tests/point_cloud_tests.py:16-156) restated, run for the host (numpy) container and for the
torch container on the CPU device; the cuda variant and the pipeline conveniences are in the gpu suite.
"""

import numpy as np
import pytest
import torch

from nimrud_amd.utils import point_clouds


def _np(x):
    return x.cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


DEVICES = [None, "cpu"]


@pytest.mark.parametrize("device", DEVICES)
def test_instantiation(device):
    # point_cloud_tests.py:16-55
    rs = np.random.RandomState(10)
    good = rs.rand(1000, 3)
    cloud = point_clouds.FlexCloud(good, device=device)
    assert np.array_equal(_np(cloud.corner), good[0])
    assert np.array_equal(_np(cloud.points + cloud.corner), good)
    assert hasattr(cloud, "assets") and cloud.num_points == 1000
    assert np.array_equal(_np(cloud.id_index), np.arange(1000))
    for bad in (rs.rand(1000, 2), rs.rand(1000, 4), rs.rand(3)):
        with pytest.raises(ValueError):
            point_clouds.FlexCloud(bad, device=device)


@pytest.mark.parametrize("device", DEVICES)
def test_add_asset(device):
    # point_cloud_tests.py:59-105: stored sorted and unique
    rs = np.random.RandomState(11)
    cloud = point_clouds.FlexCloud(rs.rand(1000, 3), device=device)
    asset_1 = rs.rand(100, 2)
    idx_1 = rs.permutation(1000)[:100]
    cloud.add_asset(asset_1, idx_1, "asset_1")
    order = np.argsort(idx_1)
    assert np.array_equal(_np(cloud.assets["asset_1"]["asset"]), asset_1.take(order, axis=0))
    assert np.array_equal(_np(cloud.assets["asset_1"]["index"]), idx_1.take(order))
    asset_2, idx_2 = np.vstack((asset_1, asset_1)), np.hstack((idx_1, idx_1))
    cloud.add_asset(asset_2, idx_2, "asset_2")
    shuffle = rs.permutation(200)
    cloud.add_asset(asset_2.take(shuffle, axis=0), idx_2.take(shuffle), "asset_3")
    for name in ("asset_2", "asset_3"):
        assert np.array_equal(_np(cloud.assets[name]["asset"]), asset_1.take(order, axis=0))
        assert np.array_equal(_np(cloud.assets[name]["index"]), idx_1.take(order))
    cloud.add_asset(idx_2.take(shuffle), idx_2.take(shuffle), "asset_4")     # scalar asset
    assert np.array_equal(_np(cloud.assets["asset_4"]["asset"]), idx_1.take(order))
    with pytest.raises(ValueError):
        cloud.add_asset(asset_1, idx_1, "asset_1")                           # name exists
    with pytest.raises(ValueError):
        cloud.add_asset(asset_1, idx_1[:50], "misaligned")
    with pytest.raises(ValueError):
        cloud.add_asset(asset_1, idx_1 + 950, "outside")
    with pytest.raises(ValueError):
        cloud.add_asset(rs.rand(100, 2, 2), idx_1, "too_many_dims")


@pytest.mark.parametrize("device", DEVICES)
def test_intersection_and_take(device):
    # point_cloud_tests.py:109-156
    rs = np.random.RandomState(12)
    points = rs.rand(1000, 3)
    cloud = point_clouds.FlexCloud(points, device=device)
    asset_1, asset_2 = rs.rand(100, 2), rs.rand(100)
    cloud.add_asset(asset_1, np.arange(100), "asset_1")
    cloud.add_asset(asset_2, np.arange(100) + 50, "asset_2")
    idx, block = cloud.intersection(["asset_1", "asset_2"])
    assert np.array_equal(_np(idx), np.arange(50, 100))
    assert np.array_equal(_np(block), np.hstack((asset_1[50:], asset_2[:50].reshape(-1, 1))))
    pick = rs.permutation(1000)[:100]
    assert np.allclose(_np(cloud.take(pick)), points.take(pick, axis=0), rtol=0, atol=1e-15)
    assert np.allclose(_np(cloud.take()), points, rtol=0, atol=1e-15)
    centred = points - points[0]
    assert np.array_equal(_np(cloud.take(pick, original_coordinates=False)), centred.take(pick, axis=0))
    assert np.array_equal(_np(cloud.take(original_coordinates=False)), centred)


def test_files_round_trip(tmp_path):
    rs = np.random.RandomState(13)
    table = np.hstack((rs.rand(50, 3) * 10, rs.rand(50, 2)))
    np.save(tmp_path / "cloud.npy", table)
    np.savetxt(tmp_path / "cloud.txt", table, delimiter=",", fmt="%.9f")
    for path, delim in ((tmp_path / "cloud.npy", None), (tmp_path / "cloud.txt", ",")):
        cloud = point_clouds.FlexCloud.from_file(str(path), delimiter=delim)
        assert cloud.num_points == 50
        assert np.allclose(_np(cloud.take()), table[:, :3], atol=1e-8)
        assert np.allclose(_np(cloud.assets["columns"]["asset"]), table[:, 3:], atol=1e-8)
        shape = cloud.export(str(tmp_path / "out.txt"), ["columns"], delimiter=";")
        assert shape == (50, 5)
        back = np.loadtxt(tmp_path / "out.txt", delimiter=";")
        assert np.allclose(back, table, atol=1e-5)
