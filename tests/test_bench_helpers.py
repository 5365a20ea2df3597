"""Host-side pieces of bench.py / ba_bench.py that can be exercised without a GPU: the JSON objects the driver
parses must come out well-formed whatever the stage times are."""
import json
import os
import types

import pytest

import bench
from visual_underwater_slam_amd import ba_bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_usable_cores_is_positive_and_bounded_by_affinity():
    import os
    n = bench.usable_cores()
    assert 1 <= n <= (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count())


def test_ba_roofline_objects_for_configs2_shape():
    prob = types.SimpleNamespace(n_obs=1926616, n_points=48299, n_poses=2000, n_nodes=2000, band=224,
                                 tiles={"n_entries": 1754967}, n_pairs=57452664)
    ms = {"linearize": 0.26, "schur": 1.8, "band_solve": 4.27, "backsub": 0.1, "eval_step": 0.12}
    top, stages = ba_bench.roofline(prob, ms)
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in top
    assert top["stage"] == "band_solve" and top["kernel"] == "chol_syrk_kernel" and 0 < top["frac"] < 1
    assert top["launches_per_solve"] == 2 * 111 + 28            # two-sided: 111 (TRSM, SYRK) pairs + 28 middle panels
    assert stages["linearize"]["algorithmic_bytes"] == 1926616 * 176 + 2000 * 432 + 48299 * 120   # SURVEY 8d formula
    sch = stages["schur"]                                        # the tile-pair GEMM: MFMA-bound, flops as issued
    assert sch["bound"] == "mfma" and sch["flops"] == 2.0 * 48 * 48 * 3 * 1754967 and 0 < sch["frac"] < 1
    assert sch["useful_flops"] < sch["flops"]
    json.dumps({"roofline": top, "stages": stages})
    fl, n = ba_bench.band_factor_flops(2000, 224)
    assert n == 250 and 1.5e10 < fl < 3e10
    # a short trajectory keeps the one-sided solve
    top2, _ = ba_bench.roofline(types.SimpleNamespace(n_obs=5000, n_points=500, n_poses=50, n_nodes=50, band=40,
                                                      tiles={"n_entries": 1200}, n_pairs=40000), ms)
    assert top2["kernel"] == "chol_trsm_update_kernel"


def test_measured_counters_reads_the_committed_pmc_summary():
    traffic, valu, ns = bench.measured_counters("fast_detect", 1000)
    assert traffic and traffic > 2 * 720 * 1280 * 1000 and valu and valu > 1e9 and 1.0 < ns < 2.5
    assert bench.measured_counters("hamming_track", 10)[0] > 0


@pytest.mark.gpu
def test_bench_command_line_contract_on_a_small_stream(gpu):
    """`python bench.py` end to end (the driver's command, with a small stream so that it takes seconds): ONE JSON line
    with the contract's keys, the roofline object of the dominant kernel and the BA legs."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--frames", "24", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["value"] > 0 and j["higher_is_better"] is True
    assert j["roofline"]["bound"] in ("hbm", "mfma") and 0 < j["roofline"]["frac"] < 1
    ba = j["ba"]
    assert ba["lm"]["status"] == 0 and ba["value"] > 0 and ba["dropin"]["same_optimum_as_array_path"] is True
    assert ba["roofline"]["bound"] in ("hbm", "mfma")


@pytest.mark.gpu
def test_bench_two_rank_rehearsal_over_gloo_on_the_one_gpu(gpu):
    """The driver's multi-GPU command line -- torch.distributed.run, one rank per GPU, `bench.py --gpus N` -- rehearsed as
    two ranks on the ONE visible GPU with VUS_BENCH_BACKEND=gloo (RCCL needs one device per rank): the frame-sharded
    front-end with its gather, and the landmark-sharded LM with its reduce / broadcast, produce ONE JSON line from rank 0
    that carries what an 8-GPU result needs to be interpreted (bytes per trial, per-stage times)."""
    import os
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, VUS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--frames", "24", "--ba-sharded-kf", "300"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0 and out["config"]["frames_per_gpu"] == 24
    sh = out["ba_sharded"]
    assert sh["lm"]["status"] == 0 and sh["value"] > 0 and sh["config"]["ranks"] == 2
    assert sh["config"]["reduce_to_rank0_bytes_per_trial"] == 288 * 300 * (sh["config"]["band_blocks"] + 1) + 48 * 300
    assert set(sh["trial_stage_ms_this_rank"]) == {"linearize+allreduce_err", "schur+reduce_to_rank0", "band_solve_rank0+broadcast_step",
                                                   "backsub", "eval_step+allreduce_err"}
