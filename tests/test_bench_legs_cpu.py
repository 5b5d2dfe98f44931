"""bench.py's host logic that needs no GPU: the pre-registered scaling model, the flat scalars the driver's record keeps,
the host budgets."""
import importlib.util
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test2", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def test_scaling_model_is_the_table_in_design_md():
    """DESIGN.md section 5 pre-registers the 1 / 2 / 4 / 8-GPU numbers; bench.py --gpus N prints the same model beside its measurement."""
    import bench_legs as legs
    rows = {n: legs.scaling_model(n) for n in (1, 2, 4, 8)}
    assert rows[1]["value"] == pytest.approx(21904 / 2.2568, rel=1e-9) and rows[1]["efficiency_vs_n_times_one_gpu"] == pytest.approx(1.0)
    for n in (2, 4, 8):
        r = rows[n]
        assert r["ms_per_step"] == pytest.approx(r["knn_ms"] + r["all_gather_ms"] + r["merge_ms"] + r["aggregate_ms"])
        assert 0.95 < r["efficiency_vs_n_times_one_gpu"] < 1.0 and rows[n]["value"] > rows[n // 2]["value"] * 1.9
    # the exchange grows with the ring, the kNN share shrinks with the shard
    assert rows[8]["all_gather_ms"] > rows[4]["all_gather_ms"] > rows[2]["all_gather_ms"] > 0
    # a node that holds its GPUs at a lower clock is predicted slower in proportion (this kernel's speed is its clock)
    slow = legs.scaling_model(8, clock_ghz=2.0)
    assert slow["knn_ms"] == pytest.approx(rows[8]["knn_ms"] * 2.38 / 2.0)
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    for n in (1, 2, 4, 8):
        q = f"{round(rows[n]['value']):,}"
        assert q in design, f"DESIGN.md section 5 does not quote the model's {n}-GPU prediction {q}"


def test_flat_scalars_reach_roofline():
    bench = _bench()
    res = {"roofline": {}, "xcd_shares_ab": {"equal_shares_kernel_ms": 2300.0, "calibrated_shares_kernel_ms": 2280.0, "calibrated_over_equal": 0.9913},
           "clusters_ab": {"clustered_kernel_ms": 2290.0, "unclustered_kernel_ms": 2280.0, "clusters_kept": 0, "cluster_in_timed_steps": [1, 1]},
           "use_fp16_mode": {"value": 8e4, "ms_per_step": 280.0, "candidate_kernel_ms": 270.0, "candidate_kernel_frac_of_fp16_mfma_peak": 0.47,
                             "clock_ghz_unprofiled": 1.6, "fallback_queries": 0, "calibration": {"locked": 0}},
           "e2e": {"failed": "x"}, "miou_parity": {"max_abs_miou_delta_vs_reference": 0.0}}
    bench.flatten_into_roofline(res)
    r = res["roofline"]
    assert r["equal_shares_kernel_ms"] == 2300.0 and r["fp16_value"] == 8e4 and r["fp16_fallback_queries"] == 0 and r["fp16_guard_locked"] == 0
    assert r["e2e_fp32_images_per_s"] is None and r["miou_max_abs_delta_vs_reference"] == 0.0 and r["unclustered_kernel_ms"] == 2280.0 and r["cluster_shape_in_timed_steps"] == "1x1"
    assert all(not isinstance(v, (dict, list)) for v in r.values())      # flat: the driver's record keeps scalars only


def test_host_budgets_and_spread():
    import bench_legs as legs
    b = legs.host_cpu_budget(); m = legs.host_mem_budget()
    assert b["cores"] >= 1 and (m["available_bytes"] is None or m["available_bytes"] > 0)
    assert legs.spread([3.0, 1.0, 2.0]) == {"min": 1.0, "median": 2.0, "max": 3.0}
    assert legs.safe("boom", lambda: 1 / 0)["failed"].startswith("boom")
