"""
synthetic LiDAR-style clouds for the BASELINE.json configurations.

all coordinates are metres, generated in fp64 from `numpy.random.RandomState(seed)`, then rounded to
fp32-representable values and stored as fp64, so fp32 and fp64 storage of the same cloud are
bit-equivalent.  used by bench.py, the tests and tests/golden/make_golden.py.
"""

import numpy as np


def _round32(points):
    return np.ascontiguousarray(points.astype(np.float32).astype(np.float64))


def uniform_cloud(n, extent=10.0, seed=0):
    """config 1: uniform-random cloud in [0, extent)^3."""
    rs = np.random.RandomState(seed)
    return _round32(rs.rand(n, 3) * extent)


def scene_cloud(n, extent=60.0, n_poles=200, n_spheres=40, seed=1, offset=(0.0, 0.0, 0.0),
                five_class=False):
    """configs 2/3/5: 60 % ground plane on [0,extent]^2 (z ~ N(0, 0.01 m)), 10 % vertical poles
    (radius 0.05 m, height 6 m, xy jitter N(0, 0.01)), 30 % sphere shells (radius 1.5 m, radial noise
    N(0, 0.01)).  points are shuffled so row order carries no spatial coherence.
    labels: 0 ground, 1 pole, 2 sphere.  five_class=True (config 5) splits the ground by what stands on
    it - 3 = ground within 0.6 m of a pole axis, 4 = ground under a sphere (within 1.5 m of a sphere
    centre in plan), 0 = open ground - so the label still is the generating primitive plus its context,
    and the five classes differ in their multiscale neighborhoods.  the points are the same either way."""
    rs = np.random.RandomState(seed)
    n_ground = int(round(0.6 * n))
    n_pole = int(round(0.1 * n))
    n_sphere = n - n_ground - n_pole

    ground = np.empty((n_ground, 3))
    ground[:, :2] = rs.rand(n_ground, 2) * extent
    ground[:, 2] = rs.randn(n_ground) * 0.01

    pole_xy = rs.rand(n_poles, 2) * extent
    which = rs.randint(0, n_poles, n_pole)
    ang = rs.rand(n_pole) * 2 * np.pi
    pole = np.empty((n_pole, 3))
    pole[:, 0] = pole_xy[which, 0] + 0.05 * np.cos(ang) + rs.randn(n_pole) * 0.01
    pole[:, 1] = pole_xy[which, 1] + 0.05 * np.sin(ang) + rs.randn(n_pole) * 0.01
    pole[:, 2] = rs.rand(n_pole) * 6.0

    centre = np.empty((n_spheres, 3))
    centre[:, :2] = rs.rand(n_spheres, 2) * extent
    centre[:, 2] = 1.5 + rs.rand(n_spheres) * 2.0
    which = rs.randint(0, n_spheres, n_sphere)
    direction = rs.randn(n_sphere, 3)
    direction /= np.linalg.norm(direction, axis=1)[:, None]
    radius = 1.5 + rs.randn(n_sphere) * 0.01
    sphere = centre[which] + direction * radius[:, None]

    points = np.concatenate((ground, pole, sphere), axis=0)
    ground_label = np.zeros(n_ground, dtype=np.int32)
    if five_class:
        from scipy.spatial import cKDTree
        d_sphere, _ = cKDTree(centre[:, :2]).query(ground[:, :2])
        ground_label[d_sphere <= 1.5] = 4
        d_pole, _ = cKDTree(pole_xy).query(ground[:, :2])
        ground_label[d_pole <= 0.6] = 3
    labels = np.concatenate((
        ground_label,
        np.ones(n_pole, dtype=np.int32),
        np.full(n_sphere, 2, dtype=np.int32)))
    order = rs.permutation(n)
    points = points[order] + np.asarray(offset, dtype=np.float64)
    return _round32(points), labels[order]


def lidar_cloud(n, seed=3, r_min=1.0, r_max=150.0, n_boxes=500):
    """config 4: terrestrial-LiDAR-style cloud with power-law density.  a scanner stands 1.8 m above a
    ground plane at the origin; ground returns have azimuth uniform and range R with pdf ~ R^-2 on
    [r_min, r_max] (so areal density falls off as R^-3); 40 % of the returns lie on the vertical faces of
    `n_boxes` box facades, each box weighted by its inverse squared distance.  labels: 0 ground, 1 facade."""
    rs = np.random.RandomState(seed)
    n_ground = int(round(0.6 * n))
    n_facade = n - n_ground
    u = rs.rand(n_ground)
    rng = 1.0 / (1.0 / r_min - u * (1.0 / r_min - 1.0 / r_max))
    az = rs.rand(n_ground) * 2 * np.pi
    ground = np.stack((rng * np.cos(az), rng * np.sin(az), rs.randn(n_ground) * 0.01), axis=1)

    baz = rs.rand(n_boxes) * 2 * np.pi
    bdist = r_min + 4.0 + rs.rand(n_boxes) ** 0.5 * (r_max - r_min - 4.0)
    centre = np.stack((bdist * np.cos(baz), bdist * np.sin(baz)), axis=1)
    size = 4.0 + rs.rand(n_boxes, 2) * 12.0
    height = 3.0 + rs.rand(n_boxes) * 12.0
    weight = 1.0 / bdist ** 2
    which = rs.choice(n_boxes, size=n_facade, p=weight / weight.sum())
    face = rs.randint(0, 4, n_facade)
    t = rs.rand(n_facade) - 0.5
    sx, sy = size[which, 0], size[which, 1]
    fx = np.where(face == 0, -0.5 * sx, np.where(face == 1, 0.5 * sx, t * sx))
    fy = np.where(face == 2, -0.5 * sy, np.where(face == 3, 0.5 * sy, t * sy))
    facade = np.stack((centre[which, 0] + fx + rs.randn(n_facade) * 0.01,
                       centre[which, 1] + fy + rs.randn(n_facade) * 0.01,
                       rs.rand(n_facade) * height[which]), axis=1)
    points = np.concatenate((ground, facade), axis=0)
    labels = np.concatenate((np.zeros(n_ground, dtype=np.int32), np.ones(n_facade, dtype=np.int32)))
    order = rs.permutation(n)
    return _round32(points[order]), labels[order]


def morton_sort(points, edge_length):
    """order rows by the 63-bit Morton code of their `edge_length` cell (config 3 is stored this way:
    the layout a tiled LiDAR archive would hand over)."""
    cells = np.floor((points - points.min(0)) / edge_length).astype(np.uint64)

    def spread(v):
        v = v & np.uint64(0x1FFFFF)
        v = (v | (v << np.uint64(32))) & np.uint64(0x1F00000000FFFF)
        v = (v | (v << np.uint64(16))) & np.uint64(0x1F0000FF0000FF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x100F00F00F00F00F)
        v = (v | (v << np.uint64(4))) & np.uint64(0x10C30C30C30C30C3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
        return v

    code = spread(cells[:, 0]) | (spread(cells[:, 1]) << np.uint64(1)) | \
        (spread(cells[:, 2]) << np.uint64(2))
    return np.argsort(code, kind="stable")


CONFIGS = {
    # name: (generator kwargs, edge lengths, radii)
    "c1_uniform_100k": dict(kind="uniform", n=100_000, extent=10.0, seed=0,
                            edges=[0.25], radii=[0.75]),
    "c2_scene_1m": dict(kind="scene", n=1_000_000, extent=60.0, n_poles=200, n_spheres=40, seed=1,
                        edges=[0.10, 0.20, 0.40], radii=[0.30, 0.60, 1.20]),
    "c3_scene_10m": dict(kind="scene", n=10_000_000, extent=190.0, n_poles=2000, n_spheres=400,
                         seed=2, edges=[0.05, 0.10, 0.20, 0.40, 0.80],
                         radii=[0.15, 0.30, 0.60, 1.20, 2.40], morton=0.80),
    "c4_lidar_50m": dict(kind="lidar", n=50_000_000, seed=3, edges=[0.05, 0.10, 0.20, 0.40, 0.80],
                         radii=[0.15, 0.30, 0.60, 1.20, 2.40], knn_min=8),
    # the reference's own ladder shape on the config 3 cloud: ONE voxel edge, several radii (voxel 0.05, scales
    # 0.15 / 0.20 / 0.25: nimrud/utils/point_clouds.py:29-35; lists of (voxel, [scales...]):
    # nimrud/prototypes/apc.py:514-518).  the three scales share one lattice and one occupancy index; their
    # candidate windows are 7, 9 and 11 cells wide
    "ref_ladder_10m": dict(kind="scene", n=10_000_000, extent=190.0, n_poles=2000, n_spheres=400,
                           seed=2, edges=[0.05, 0.05, 0.05], radii=[0.15, 0.20, 0.25], morton=0.80),
    # config 5 = the config 3 cloud (same seed, same points) with five-class labels, classified by the
    # random forest of tests/golden/g6_forest_c5.npz (32 trees, depth <= 12) behind the last scale
    "c5_scene_10m_rf": dict(kind="scene", n=10_000_000, extent=190.0, n_poles=2000, n_spheres=400,
                            seed=2, edges=[0.05, 0.10, 0.20, 0.40, 0.80],
                            radii=[0.15, 0.30, 0.60, 1.20, 2.40], morton=0.80, five_class=True,
                            forest="g6_forest_c5.npz"),
}


def make_config(name, n=None, seed_offset=0):
    """returns (points, labels_or_None, edges, radii) for a named configuration; `n` overrides the
    point count (same generator, same density when the extent is scaled by the caller)."""
    cfg = dict(CONFIGS[name])
    cfg["seed"] = cfg["seed"] + 1000 * seed_offset
    if cfg["kind"] == "lidar":
        points, labels = lidar_cloud(n if n is not None else cfg["n"], seed=cfg["seed"])
        return points, labels, list(cfg["edges"]), list(cfg["radii"])
    if n is not None and n != cfg["n"]:
        # keep the areal density: scale the extent with sqrt(n)
        cfg["extent"] = cfg["extent"] * np.sqrt(n / cfg["n"]) if cfg["kind"] == "scene" \
            else cfg["extent"] * (n / cfg["n"]) ** (1.0 / 3.0)
        if cfg["kind"] == "scene":
            cfg["n_poles"] = max(1, int(round(cfg["n_poles"] * n / cfg["n"])))
            cfg["n_spheres"] = max(1, int(round(cfg["n_spheres"] * n / cfg["n"])))
        cfg["n"] = n
    if cfg["kind"] == "uniform":
        points, labels = uniform_cloud(cfg["n"], cfg["extent"], cfg["seed"]), None
    else:
        points, labels = scene_cloud(cfg["n"], cfg["extent"], cfg["n_poles"], cfg["n_spheres"],
                                     cfg["seed"], five_class=bool(cfg.get("five_class")))
    if cfg.get("morton"):
        order = morton_sort(points, cfg["morton"])
        points = np.ascontiguousarray(points[order])
        labels = labels[order] if labels is not None else None
    return points, labels, list(cfg["edges"]), list(cfg["radii"])
