"""Plugin contract of the kNN backends.

Interface-compatible with the reference's `NearestNeighborSearchBase` (hbird/nn/search_base.py:3-31): a backend is
built once from the bank tensor `[M, D]` (hbird/hbird_eval.py:272-279) and then asked, once per validation batch,
for the neighbours of `[nq, D]` un-normalised query tokens (hbird_eval.py:628).  The answer is a pair
`(indices [nq, k], distances [nq, k])`, rows best-first; only the indices are consumed by the evaluator.
"""
import abc


class NearestNeighborSearchBase(abc.ABC):
    def __init__(self, feature_memory, n_neighbors=30, distance_measure="dot_product", **kwargs):
        self.n_neighbors = n_neighbors
        self.distance_measure = str(distance_measure).lower()
        self.feature_memory = feature_memory
        self.device = feature_memory.device
        # two-step construction, in this order, like the reference: create the index, then fill it
        self.index = self._initialize_index()
        self._add_features_to_index()

    @abc.abstractmethod
    def _initialize_index(self):
        """Create and return the (empty) index object for `self.distance_measure`."""

    @abc.abstractmethod
    def _add_features_to_index(self):
        """Insert the rows of `self.feature_memory` (row i gets id i)."""

    @abc.abstractmethod
    def find_nearest_neighbors(self, q, k=None):
        """`q`: [nq, D] queries; `k`: neighbours per query (default `self.n_neighbors`) -> (indices, distances)."""
