"""Helper: build the (child, parent) edge table of a skeleton from a parent map."""
import numpy as np


def edges_from_parents(parents: dict) -> np.ndarray:
    """``parents[child] = parent`` -> int array (E, 2) of (child, parent) rows, oriented towards the centre."""
    return np.array(sorted(parents.items()), dtype=np.int64)
