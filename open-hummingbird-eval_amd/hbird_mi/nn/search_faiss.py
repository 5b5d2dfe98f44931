"""`hbird.nn.search_faiss` under its reference name: on MI355X the exact flat search of
faiss.GpuIndexFlatIP / GpuIndexFlatL2 (reference hbird/nn/search_faiss.py:6-90) is served by the HIP engine.
An unmodified reference `hbird_eval.py` imports this module lazily by name (hbird_eval.py:276), so putting
this package's `nn/` in place of `hbird/nn/` switches the backend without touching the evaluator."""
from hbird_mi.nn.search_hip import NearestNeighborSearchHIP


class NearestNeighborSearchFaiss(NearestNeighborSearchHIP):
    pass
