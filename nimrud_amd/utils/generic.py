"""
host-side helpers of the pipeline.

`batcher` has the call signature and the results of the reference's helper of the same name
(nimrud/utils/generic.py:8-26), which the reference uses to walk a query cloud in chunks of
QUERY_CHUNK_SIZE (multiscale.py:94).  the GPU path has no use for query chunks - a kernel launch takes
the whole cloud - so here it is a convenience for callers that stream clouds from disk, and it also
understands torch tensors (device-resident clouds are cut into views, nothing is copied).
"""

import itertools

import numpy as np
import torch

_SLICEABLE = (np.ndarray, list, torch.Tensor)


def _lazy_groups(iterable, size):
    source = iter(iterable)
    group = list(itertools.islice(source, size))
    while group:
        yield group
        group = list(itertools.islice(source, size))


def batcher(collection, chunk_size):
    """generator over consecutive pieces of `collection`, `chunk_size` items each (the last one may be
    shorter).  anything that can be sliced comes back as slices of itself; any other iterable is read
    only as far as needed and comes back as lists."""
    chunk_size = int(chunk_size)
    if chunk_size < 1:
        raise ValueError("chunk_size must be at least 1")
    if not isinstance(collection, _SLICEABLE):
        return _lazy_groups(collection, chunk_size)
    total = len(collection)
    return (collection[lo:min(lo + chunk_size, total)] for lo in range(0, total, chunk_size))
