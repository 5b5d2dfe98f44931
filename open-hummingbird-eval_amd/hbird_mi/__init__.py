"""hbird_mi -- MI355X-native hot path of the Hummingbird dense-retrieval evaluation.

Mirrors the reference package layout for the path it replaces:
  hbird_mi.nn.search_base / search_hip   <->  hbird/nn/search_base.py, search_faiss.py
  hbird_mi.hbird_eval                    <->  hbird/hbird_eval.py (HbirdEvaluation, hbird_evaluation)
  hbird_mi.utils.eval_metrics            <->  hbird/utils/eval_metrics.py (PredsmIoU)
"""
__version__ = "0.1.0"
