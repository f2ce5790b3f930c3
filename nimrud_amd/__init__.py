"""
nimrud_amd - the multiscale neighborhood-feature hot path of grayhem/nimrud (nimrud/minimal), built
from scratch for the AMD Instinct MI355X (gfx950).

    from nimrud_amd.minimal import multiscale
    features = multiscale.process_single_core(cloud, cloud, [0.1, 0.2, 0.4], [0.3, 0.6, 1.2])

module layout mirrors the reference so that `import nimrud_amd as nimrud` is a drop-in for that path:
    nimrud_amd.minimal.multiscale / features / classification
    nimrud_amd.utils.geometry (VoxelFilter) / generic (batcher)
"""

__all__ = ["minimal", "utils"]
