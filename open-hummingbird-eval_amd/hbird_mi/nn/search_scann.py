"""`hbird.nn.search_scann` under its reference name (reference hbird/nn/search_scann.py:6-42).

ScaNN's approximate tree + asymmetric-hash search is NOT re-implemented (no Faiss/ScaNN at inference); the class
keeps the reference's constructor keywords and its quirk of ignoring the per-call `k` (search_scann.py:39-42),
and answers with the exact HIP search -- a superset of what ScaNN approximates."""
from hbird_mi.nn.search_hip import NearestNeighborSearchHIP


class NearestNeighborSearchScaNN(NearestNeighborSearchHIP):
    def __init__(self, feature_memory, n_neighbors=30, distance_measure="dot_product", num_leaves=512,
                 num_leaves_to_search=32, anisotropic_quantization_threshold=0.2, num_reordering_candidates=120,
                 dimensions_per_block=4, **kwargs):
        if distance_measure.lower() not in ("dot_product", "euclidean"):
            raise ValueError(f"Unsupported distance measure: {distance_measure.lower()}")   # search_scann.py:19-20
        self.num_leaves = num_leaves
        self.num_leaves_to_search = num_leaves_to_search
        self.anisotropic_quantization_threshold = anisotropic_quantization_threshold
        self.num_reordering_candidates = num_reordering_candidates
        self.dimensions_per_block = dimensions_per_block
        super().__init__(feature_memory, n_neighbors, distance_measure, **kwargs)

    def find_nearest_neighbors(self, q, k=None):
        return super().find_nearest_neighbors(q, None)      # always the constructor's n_neighbors
