"""Backend plugin surface, same shape as the reference's ABC (hbird/nn/search_base.py:3-31)."""
from abc import ABC, abstractmethod


class NearestNeighborSearchBase(ABC):
    """Constructed once with the feature memory, then `find_nearest_neighbors(q, k=None)` is called
    once per validation batch and returns `(indices, distances)` (hbird/hbird_eval.py:272-279, 628)."""

    def __init__(self, feature_memory, n_neighbors=30, distance_measure="dot_product", **kwargs):
        self.feature_memory = feature_memory
        self.n_neighbors = n_neighbors
        self.distance_measure = distance_measure.lower()
        self.device = feature_memory.device
        self.index = self._initialize_index()
        self._add_features_to_index()

    @abstractmethod
    def _initialize_index(self):
        """Initializes the nearest neighbor search index."""

    @abstractmethod
    def _add_features_to_index(self):
        """Adds feature vectors to the index."""

    @abstractmethod
    def find_nearest_neighbors(self, q, k=None):
        """Finds the nearest neighbors for a given query tensor."""
