"""
small helpers shared by the pipeline.  mirrors nimrud/utils/generic.py (batcher, :8-26).
"""

import itertools

import numpy as np


def batcher(collection, chunk_size):
    """yield consecutive chunks of `chunk_size` items.  arrays and lists are sliced; any other
    iterable is consumed lazily and yielded as lists (last chunk may be short)."""
    if isinstance(collection, (np.ndarray, list)):
        for start in range(0, len(collection), chunk_size):
            yield collection[start:start + chunk_size]
        return
    iterator = iter(collection)
    while True:
        chunk = list(itertools.islice(iterator, chunk_size))
        if not chunk:
            return
        yield chunk
