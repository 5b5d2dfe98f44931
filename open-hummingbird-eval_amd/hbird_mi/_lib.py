"""ctypes binding of lib/libhbird_hip.so (C ABI: include/hbird_hip.h).

This is the only way the Python side reaches the HIP kernels.  There is no CPU fallback: if the
shared library is missing, or no GPU is visible when an index is created, the call fails loudly.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_uint64, c_void_p

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("HBIRD_HIP_LIB") or os.path.join(_PKG_ROOT, "lib", "libhbird_hip.so")   # override: A/B builds
CSRC_DIR = os.path.join(_PKG_ROOT, "csrc")

_lib = None

# name -> (restype, argtypes); mirrors include/hbird_hip.h one to one
SIGNATURES = {
    "hb_last_error": (c_char_p, []),
    "hb_device_count": (c_int, [POINTER(c_int)]),
    "hb_index_create": (c_int, [c_int, c_int, c_int, POINTER(c_void_p)]),
    "hb_index_free": (c_int, [c_void_p]),
    "hb_index_set_stream": (c_int, [c_void_p, c_void_p]),
    "hb_index_reserve": (c_int, [c_void_p, c_int64]),
    "hb_index_add": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int]),
    "hb_index_add_labels": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int]),
    "hb_index_ntotal": (c_int64, [c_void_p]),
    "hb_index_nlabels": (c_int64, [c_void_p]),
    "hb_index_reset": (c_int, [c_void_p]),
    "hb_index_search": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p, c_int]),
    "hb_index_search_aggregate": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int64, c_float, c_void_p, c_void_p,
                                          c_void_p, c_int]),
    "hb_index_aggregate": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int64, c_float,
                                   c_void_p, c_int]),
    "hb_index_aggregate_partial": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int64, c_float, c_void_p, c_int64,
                                           c_void_p]),
    "hb_index_reconstruct": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int]),
    "hb_index_gather_labels": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int]),
    "hb_index_set_label_table": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64]),
    "hb_index_copy_norms": (c_int, [c_void_p, c_void_p, c_int]),
    "hb_index_set_score_output": (c_int, [c_void_p, c_int]),
    "hb_index_distances_from_scores": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "hb_merge_topk": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "hb_packed_list_bytes": (c_int64, [c_int64, c_int]),
    "hb_merge_topk_packed": (c_int, [c_void_p, c_int64, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "hb_multi_create": (c_int, [c_int, c_int, POINTER(c_int), c_int, c_int, POINTER(c_void_p)]),
    "hb_multi_free": (c_int, [c_void_p]),
    "hb_multi_reserve": (c_int, [c_void_p, c_int64]),
    "hb_multi_add": (c_int, [c_void_p, c_void_p, c_int64, c_int]),
    "hb_multi_ntotal": (c_int64, [c_void_p]),
    "hb_multi_shard_rows": (c_int, [c_void_p, POINTER(c_int64), c_int]),
    "hb_multi_set_fp16": (c_int, [c_void_p, c_int]),
    "hb_multi_search": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "hb_normalize_rows": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "hb_patch_label_hist": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "hb_patch_scores": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "hb_patch_select": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p,
                                c_void_p]),
    "hb_gather_rows": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_void_p]),
    "hb_upsample_argmax": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "hb_upsample_argmax_confusion": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int64, c_int,
                                             c_void_p, c_void_p, c_void_p]),
    "hb_upsample_accumulate": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                       c_void_p]),
    "hb_argmax_channels": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "hb_confusion_update": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int64, c_int, c_void_p, c_void_p]),
    "hb_index_set_timing": (c_int, [c_void_p, c_int]),
    "hb_index_last_knn_ms": (c_int, [c_void_p, POINTER(c_double)]),
    "hb_index_set_tuning": (c_int, [c_void_p, c_int, c_int]),
    "hb_index_set_fp16": (c_int, [c_void_p, c_int]),
    "hb_index_last_fp16_fallbacks": (c_int, [c_void_p, POINTER(c_int64)]),
    "hb_index_set_fp16_escalation": (c_int, [c_void_p, c_int]),
    "hb_index_last_fp16_escalated": (c_int, [c_void_p, POINTER(c_int64)]),
    "hb_schedule_plan": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int64, POINTER(c_int64)]),
    "hb_schedule_plan_phased": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int64, POINTER(c_int64),
                                        c_void_p, c_int, POINTER(c_int), c_void_p]),
    "hb_schedule_plan_shared": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int64, POINTER(c_int64)]),
    "hb_index_set_cluster": (c_int, [c_void_p, c_int, c_int, c_int]),
    "hb_index_set_cluster_sharing": (c_int, [c_void_p, c_int]),
    "hb_index_set_label_denominator": (c_int, [c_void_p, c_int]),
    "hb_index_labels_to_fp32": (c_int, [c_void_p]),
    "hb_index_label_denominator": (c_int, [c_void_p, POINTER(c_int)]),
    "hb_index_copy_label_counts": (c_int, [c_void_p, c_void_p, c_int]),
    "hb_index_set_label_count_table": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int64]),
    "hb_index_cluster_stats": (c_int, [c_void_p, POINTER(c_int64)]),
    "hb_index_set_variant": (c_int, [c_void_p, c_int]),
    "hb_index_set_search_options": (c_int, [c_void_p, c_int, c_int64]),
    "hb_index_wg_stamps": (c_int, [c_void_p, c_void_p, c_int, POINTER(c_int)]),
    "hb_index_set_xcd_weights": (c_int, [c_void_p, c_int, POINTER(c_double)]),
    "hb_index_xcd_weights": (c_int, [c_void_p, c_int, POINTER(c_double), POINTER(c_int)]),
    "hb_schedule_plan_weighted": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_double), c_void_p, c_int64, POINTER(c_int64)]),
    "hb_index_kernel_clock": (c_int, [c_void_p, POINTER(c_double)]),
    "hb_index_xcd_stats": (c_int, [c_void_p, c_int, POINTER(c_double)]),
    "hb_calibration_new": (c_void_p, [c_int]),
    "hb_calibration_free": (None, [c_void_p]),
    "hb_calibration_state": (c_int, [c_void_p, POINTER(c_double), POINTER(c_int64)]),
    "hb_calibration_feed": (c_int, [c_void_p, c_void_p, c_int, POINTER(c_double), POINTER(c_int), c_int, c_double]),
    "hb_f16_adapt_replay": (c_int, [c_int, POINTER(c_double), POINTER(c_double), c_int64, POINTER(c_int)]),
    "hb_set_layout_form": (c_int, [c_int]),
    "hb_index_set_rerank_copy": (c_int, [c_void_p, c_int]),
    "hb_index_rerank_copy_bytes": (c_int, [c_void_p, POINTER(c_int64)]),
    "hb_index_schedule_info": (c_int, [c_void_p, POINTER(c_int64)]),
}


class HbirdHipError(RuntimeError):
    pass


class HbirdClassRangeError(HbirdHipError, ValueError):
    """A mask holds a class id outside [0, num_classes): F.one_hot of the reference raises for it (hbird_eval.py:319)."""


def lib() -> ctypes.CDLL:
    """Load libhbird_hip.so (built by __graft_entry__.build() / `make -C csrc`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HbirdHipError(
                f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; "
                f"g.build()' or make -C {CSRC_DIR}). There is no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        # HBIRD_PLAN_ONLY=1 (tests/test_sanitizers_cpu.py): LIB_PATH names the host-only sanitizer build of the work-list planner,
        # which exports the hb_schedule_plan* / hb_calibration_* entry points and hb_last_error only
        plan_only = os.environ.get("HBIRD_PLAN_ONLY") == "1"
        for name, (res, args) in SIGNATURES.items():
            if plan_only and not (name.startswith("hb_schedule_plan") or name.startswith("hb_calibration_") or name == "hb_f16_adapt_replay" or name == "hb_last_error"):
                continue
            fn = getattr(L, name)  # AttributeError here = header and library disagree
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error() -> str:
    msg = lib().hb_last_error()
    return msg.decode() if msg else ""


def check(rc: int, exc=HbirdHipError):
    if rc != 0:
        raise exc(last_error() or f"libhbird_hip call failed with status {rc}")


def device_count() -> int:
    n = c_int(0)
    lib().hb_device_count(ctypes.byref(n))
    return int(n.value)
