"""Host-side pieces of bench.py / ba_bench.py that can be exercised without a GPU: the JSON objects the driver
parses must come out well-formed whatever the stage times are."""
import json
import types

import bench
from visual_underwater_slam_amd import ba_bench


def test_usable_cores_is_positive_and_bounded_by_affinity():
    import os
    n = bench.usable_cores()
    assert 1 <= n <= (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count())


def test_ba_roofline_objects_for_configs2_shape():
    prob = types.SimpleNamespace(n_obs=1926616, n_points=48299, n_poses=2000, n_nodes=2000, band=224,
                                 st={"n_blocks": 308483, "n_pairs": 57452664})
    ms = {"linearize": 0.26, "schur": 1.8, "band_solve": 4.27, "backsub": 0.1, "eval_step": 0.12}
    top, stages = ba_bench.roofline(prob, ms)
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in top
    assert top["stage"] == "band_solve" and top["kernel"] == "chol_syrk_kernel" and 0 < top["frac"] < 1
    assert top["launches_per_solve"] == 2 * 111 + 28            # two-sided: 111 (TRSM, SYRK) pairs + 28 middle panels
    assert stages["linearize"]["algorithmic_bytes"] == 1926616 * 176 + 2000 * 432 + 48299 * 120   # SURVEY 8d formula
    json.dumps({"roofline": top, "stages": stages})
    fl, n = ba_bench.band_factor_flops(2000, 224)
    assert n == 250 and 1.5e10 < fl < 3e10
    # a short trajectory keeps the one-sided solve
    top2, _ = ba_bench.roofline(types.SimpleNamespace(n_obs=5000, n_points=500, n_poses=50, n_nodes=50, band=40,
                                                      st={"n_blocks": 900, "n_pairs": 40000}), ms)
    assert top2["kernel"] == "chol_trsm_update_kernel"


def test_measured_counters_reads_the_committed_pmc_summary():
    traffic, valu, ns = bench.measured_counters("fast_detect", 1000)
    assert traffic and traffic > 2 * 720 * 1280 * 1000 and valu and valu > 1e9 and 1.0 < ns < 2.5
    assert bench.measured_counters("hamming_track", 10)[0] > 0
