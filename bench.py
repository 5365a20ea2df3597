#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X (contract in the task statement).

Primary line (BASELINE.json configs[1]): ORB detect+match stereo frames/s on a synthetic
1280x720 stereo stream, 2000 keypoints per image, 1000 frames resident in HBM per GPU.
One "step" = one call of StereoOrbFrontend.process() -- the product's own call sequence: FAST+NMS+smoothing,
top-2000 selection, orientation + rBRIEF, left->right and left(t)->left(t+1) brute-force Hamming -- over the rank's
shard of the stream, timed per stage with HIP events recorded through the front-end's stage hook.
With N ranks (BASELINE.json configs[3]) the stream has N*1000 frames, every rank owns a contiguous 1000-frame
shard plus a one-frame halo (dist.shard_frames), and the step ends with the all_gather of the per-frame feature-track
records (dist.gather_tracks, RCCL): "scaling": "weak".

The same JSON line carries `roofline` (dominant kernel: HBM reading + the VALU-issue reading that actually binds
it), `cpu_baseline` (OpenCV if this machine has it, else the C port, on a bounded sample of the same stream, 1 thread
and all usable cores) and `ba`: the second half of the metric, full-batch LM wall time at 2000 keyframes / 50k
landmarks (configs[2]) with its own roofline, drop-in boundary timing and CPU baseline; with N > 1 ranks `ba_sharded`:
the landmark-sharded LM of configs[4].
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

H, W, KP = 720, 1280, 2000
INT8_PEAK_TOPS = 5000.0   # dense int8 MFMA: 2x the ~2.5 PFLOP/s dense bf16 peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
N_SIMD = 256 * 4       # 256 CUs x 4 SIMDs

# Algorithmic bytes (single pyramid level), per stereo frame:
#   fast_detect : 2 images read once (2*H*W) + 2*2000 keypoint records of 16 B
#   select_topk : candidates read once is an implementation artefact -> counted as keypoint records above
#   orient_rbrief: 2*2000 descriptors of 32 B written
#   hamming     : descriptors re-read once (2*2000*32) + 2000 matches * 8 B
ALGO_BYTES = {
    "fast_detect": 2 * H * W + 2 * KP * 16,
    "select_topk": 2 * KP * 4,
    "orient_rbrief": 2 * KP * 32,
    "hamming_stereo": 2 * KP * 32 + KP * 8,
    "hamming_track": 2 * KP * 32 + KP * 8,
}
ALGO_BYTES_FRAME = 2 * H * W + 2 * KP * 16 + 2 * KP * 32 + 2 * KP * 32 + KP * 8  # = 2,179,200
STAGES = ["fast_detect", "select_topk", "orient_rbrief", "hamming_stereo", "hamming_track"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=1000, help="frames per GPU (configs[1]: 1000)")
    ap.add_argument("--cpu-frames", type=int, default=0, help="bounded cpu_baseline sample (0: sized from the core count)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ba", action="store_true")
    ap.add_argument("--no-first-call", action="store_true", help="skip the fresh-process measurement of the first optimize()")
    ap.add_argument("--no-object-graph", action="store_true", help="skip the per-object graph leg (~20 s of Python graph build)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the images -> trajectory leg")
    ap.add_argument("--ba-sharded-kf", type=int, default=10000, help="keyframes of the configs[4] leg run with N > 1 ranks")
    ap.add_argument("--pyramid", action="store_true", help="also measure the same stream through the 8-level x1.2 pyramid "
                    "(extra `pyramid8` object; off by default so that a profile of the default command sees one launch "
                    "shape per kernel)")
    ap.add_argument("--no-pyramid", action="store_true", help="accepted for older scripts; the default already")
    return ap.parse_args()


def usable_cores():
    """Cores this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU box hands a
    one-GPU job a share of the host, not all of it)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return n


def make_stream(n_frames, t0, device):
    from visual_underwater_slam_amd import synth
    cv = synth.canvas(torch, device)
    out = torch.empty((n_frames, 2, H, W), dtype=torch.uint8, device=device)
    chunk = 8
    for s in range(0, n_frames, chunk):
        n = min(chunk, n_frames - s)
        out[s:s + n] = synth.stereo_frames(t0 + s, n, H, W, xp=torch, device=device, canvas_arr=cv)
    return out


class StageTimer:
    """HIP events on the launch stream, recorded through StereoOrbFrontend.stage_hook (torch's current stream is the
    stream the C ABI launches on)."""

    def __init__(self):
        self.steps = []

    def __call__(self, name):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        if name == "begin":
            self.steps.append([])
        self.steps[-1].append((name, ev))

    def stage_ms(self):
        acc = {}
        for st in self.steps:
            for (_, e0), (n1, e1) in zip(st, st[1:]):
                acc.setdefault(n1, []).append(e0.elapsed_time(e1))
        return {k: sum(v) / len(v) for k, v in acc.items() if k != "end"}


def measured_counters(kernel, frames):
    """Per-launch HBM bytes and VALU wave-instructions of `kernel` from the committed PMC summary
    (profiles/traffic.json: per-image figures measured by tools/collect_profiles.sh with rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE / SQ_INSTS_VALU in separate passes).  (None, None) when the file is absent."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(path))
        key = {"fast_detect": "fast_tile_kernel", "orient_rbrief": "orient_rbrief_kernel",
               "select_topk": "select_topk_kernel", "hamming_stereo": "hamming_match_rows_kernel",
               "hamming_track": "hamming_match_mfma_kernel"}[kernel]
        per_img = t["bytes_per_image"][key]
        traffic = int((per_img["fetch"] + per_img["write"]) * 2 * frames)
        valu = t.get("valu_wave_insts_per_image", {}).get(key)
        return traffic, (None if valu is None else valu * 2 * frames), t.get("valu_ns_per_wave_inst_per_simd", {}).get(key)
    except Exception:
        return None, None, None


def cpu_baseline_frontend(t0, cores, requested_frames):
    """M1 on the host CPU.  OpenCV (the engine the reference's nodelet runs) if this machine has it, else the C port
    (-O3 -march=native build made here, OpenMP over images).  1 thread and all usable cores, median of 5 after a
    warm-up; the sample is sized so that every thread has an image to work on."""
    import numpy as np
    from visual_underwater_slam_amd import synth
    from oracle import oracle as O, engines
    probe = engines.probe()
    out = {"engines": engines.describe(probe)}
    n_all = requested_frames or max(32, min(128, 4 * cores))      # 2 images per frame: 8 images per thread
    img_all = synth.stereo_frames(t0, n_all)
    if probe["cv2"] is not None:
        try:
            import cv2
            def run_cv(n):
                t = time.perf_counter()
                for f in range(n):
                    engines.cv2_orb_detect_match(img_all[f, 0], img_all[f, 1], KP, 10)
                return time.perf_counter() - t
            cv2.setNumThreads(cores)
            run_cv(1)
            ts = [run_cv(min(n_all, 8)) for _ in range(5)]
            dt = statistics.median(ts)
            out.update({"value": round(min(n_all, 8) / dt, 3), "unit": "frames/s", "cores": cores, "kind": "reference",
                        "sample": f"cv2 {probe['cv2']} ORB_create(nfeatures=2000, fastThreshold=10) + BFMatcher(NORM_HAMMING) "
                                  f"left->right on the first {min(n_all, 8)} stereo frames, median of 5, cv2.setNumThreads({cores})"})
            return out
        except Exception as e:
            out["cv2_error"] = str(e)
    lib = O.lib(native=True)

    def run(n, threads):
        lib.vus_oracle_set_threads(threads)
        img = img_all[:n].reshape(2 * n, H, W)
        t = time.perf_counter()
        keys, cnt, blur = O.fast_detect(img, _lib=lib)
        kp, kc = O.select_topk(keys, cnt, KP, _lib=lib)
        desc, _ = O.orient_rbrief(img, blur, kp, kc, _lib=lib)
        f = np.arange(n, dtype=np.int32)
        O.hamming_match(desc, kp, kc, W, 2 * f, 2 * f + 1, 5, 0, 128, 64, _lib=lib)
        if n > 1:
            O.hamming_match(desc, kp, kc, W, 2 * f[:-1], 2 * f[:-1] + 2, -1, 0, 0, 64, _lib=lib)
        return time.perf_counter() - t
    run(min(n_all, 4), cores)                                       # warm-up
    t_all = statistics.median(run(n_all, cores) for _ in range(5))
    n_one = 2
    t_one = statistics.median(run(n_one, 1) for _ in range(5))
    out.update({"value": round(n_all / t_all, 3), "unit": "frames/s", "cores": cores, "kind": "port",
                "value_1thread": round(n_one / t_one, 3),
                "sample": f"C port (gcc -O3 -march=native, OpenMP over the {2 * n_all} images): first {n_all} stereo frames of "
                          f"the same stream, {cores} threads, median of 5 = {t_all:.2f} s; 1 thread: first {n_one} frames, "
                          f"median of 5 = {t_one:.2f} s.  Not OpenCV: cv2 is absent on this machine"})
    return out


def cpu_baseline_ba(seq, cores, gpu_seconds):
    """M2 on the host CPU at the SAME configs[2] problem.  gtsam (the engine batch.py:337 calls) if present, else the
    CPU port: OpenMP kernels (-O3 -march=native) + LAPACK banded Cholesky (oracle/ba_port.py)."""
    import numpy as np
    from visual_underwater_slam_amd import ba_pack
    from oracle import oracle as O, engines, ba_port
    probe = engines.probe()
    out = {"engines": engines.describe(probe)}
    n_kf, nL = len(seq["poses_init"]), len(seq["points_gt"])
    if probe["gtsam"] is not None:
        try:
            t = time.perf_counter()
            _, _, err, its = engines.gtsam_stereo_lm(seq, prior_on_gt=True)
            dt = time.perf_counter() - t
            out.update({"value": round(dt, 3), "unit": "s", "cores": cores, "kind": "reference",
                        "sample": f"gtsam {probe['gtsam']} LevenbergMarquardtOptimizer default params on the same graph "
                                  f"(includes the per-factor graph build), {its} iterations, final error {err:.1f}; one run",
                        "gpu_speedup": round(dt / gpu_seconds, 1)})
            return out
        except Exception as e:
            out["gtsam_error"] = str(e)
    pk = ba_pack.pack_observations(torch.from_numpy(seq["obs_pose"]), torch.from_numpy(seq["obs_point"]),
                                   torch.from_numpy(seq["meas"]), n_kf, nL)
    st = ba_pack.build_structure(pk)
    P = O.BAProblem(pk, seq["K"], seq["sigma"], (np.array([0], np.int32), seq["poses_gt"][:1], seq["prior_sigmas"][None]))
    runs = []
    with ba_port.set_threads(cores):
        # OpenBLAS's banded Cholesky does not always gain from threads: calibrate the BLAS pool on the first damped
        # system and give the baseline its best setting (the OpenMP kernels always use every core)
        probe_port = ba_port.BAPort(P, st, native=True)
        probe_port.linearize(seq["poses_init"], seq["points_init"])
        probe_port.schur(1e-5)
        probe_port.band_solve()                                    # warm-up: LAPACK import, first touch of the band
        calib = {}
        for bt in sorted({1, min(4, cores), min(8, cores), cores}):
            probe_port.blas_threads = bt
            probe_port.schur(1e-5)
            t = time.perf_counter()
            probe_port.band_solve()
            calib[bt] = time.perf_counter() - t
        best_bt = min(calib, key=calib.get)
        del probe_port
        for _ in range(5):
            _, _, rep = ba_port.BAPort(P, st, native=True, blas_threads=best_bt).optimize(seq["poses_init"], seq["points_init"], max_seconds=60)
            runs.append(rep)
    rep = sorted(runs, key=lambda r: r["seconds"])[2]
    with ba_port.set_threads(1):
        _, _, rep1 = ba_port.BAPort(P, st, native=True).optimize(seq["poses_init"], seq["points_init"], max_seconds=30)
    out.update({
        "value": round(rep["seconds"], 3), "unit": "s", "cores": cores, "kind": "port",
        "sample": f"full LM at configs[2] ({n_kf} keyframes / {nL} landmarks / {len(seq['obs_pose'])} stereo factors): CPU port = "
                  f"OpenMP kernels (gcc -O3 -march=native, {cores} threads) + LAPACK dpbtrf/dpbtrs (scipy OpenBLAS, {best_bt} threads: "
                  f"the fastest of {{{', '.join(f'{k}: {v:.2f} s' for k, v in calib.items())}}} per factorisation), median of 5, "
                  f"{rep['tries']} linear solves" + (" (stopped at the 60 s bound)" if rep["truncated"] else "") +
                  ".  Not GTSAM: gtsam is absent on this machine",
        "s_per_linear_solve": round(rep["seconds"] / max(rep["tries"], 1), 3),
        "stage_s": {k: round(v, 3) for k, v in rep["stage_s"].items()},
        "final_error": rep["final_error"],
        "one_thread": {"seconds": round(rep1["seconds"], 3), "linear_solves": rep1["tries"], "truncated_at_30s": rep1["truncated"],
                       "s_per_linear_solve": round(rep1["seconds"] / max(rep1["tries"], 1), 3),
                       "stage_s": {k: round(v, 3) for k, v in rep1["stage_s"].items()}},
        "gpu_speedup": round(rep["seconds"] / gpu_seconds, 1) if not rep["truncated"] else None,
    })
    return out


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # VUS_BENCH_BACKEND=gloo + several ranks on ONE GPU is a rehearsal of the multi-rank code path only
    backend = os.environ.get("VUS_BENCH_BACKEND", "nccl")
    dev_index = local % max(torch.cuda.device_count(), 1) if backend != "nccl" else local
    device = torch.device("cuda", dev_index)
    torch.cuda.set_device(device)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    assert a.gpus == world, f"--gpus {a.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    first_call = None
    under_profiler = "rocprofiler" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCP") for k in os.environ)
    if world == 1 and not a.no_ba and not a.no_first_call and under_profiler:
        # rocprofv3's preloaded tool may already have initialised the GPU in this process: starting a child from it is
        # what the GPU pool forbids.  The figure comes from the plain run (profiles/bench_rNN.json).
        first_call = {"skipped": "running under rocprofv3: no child process is started from a profiled process"}
    elif world == 1 and not a.no_ba and not a.no_first_call:
        # batch.py:337 calls optimize() ONCE per process: measure that first call in a fresh child process, before this
        # process has made any GPU call of its own (the two never share the GPU)
        from visual_underwater_slam_amd import ba_bench as _bb
        # twice: the first child may be the first process on this machine to read libvus_hip.so (cold file cache: the
        # library and its code object come from disk); the second is what every later fresh process sees
        first_cold_cache = _bb.first_call_probe()
        first_call = _bb.first_call_probe()
        if isinstance(first_call, dict) and isinstance(first_cold_cache, dict):
            first_call["first_process_on_this_machine"] = {k: first_cold_cache.get(k) for k in ("first_call_s", "first_call_phase_ms", "error")
                                                           if k in first_cold_cache}

    from visual_underwater_slam_amd import dist as vdist
    from visual_underwater_slam_amd.frontend import StereoOrbFrontend, ImageProcessorParams
    F = a.frames
    n_stream = world * F
    first, n_owned, n_halo = vdist.shard_frames(n_stream, world, rank)     # weak scaling: F owned frames per rank
    images = make_stream(n_halo, first, device)      # this rank's shard (+ halo frame) of the stream, resident in HBM
    fe = StereoOrbFrontend(H, W, max_frames=n_halo, params=ImageProcessorParams(max_features=KP), device=device)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    pending = []

    def step():
        res = fe.process(images, check=False)
        if world > 1:          # configs[3]: the step ends with the gather of the stream's feature-track records.  It is
            # STARTED here (RCCL's own stream, records copied to a staging buffer) and awaited one step later, so that it
            # overlaps the next step's kernels; drain() before every barrier: all K gathers end inside the timed region
            pending.append(vdist.gather_tracks_start(*vdist.owned_track_records(res, n_owned), n_stream, world, rank))
            while len(pending) > 1:
                vdist.gather_tracks_finish(pending.pop(0))

    def drain():
        while pending:
            vdist.gather_tracks_finish(pending.pop(0))

    # set-up pass (not a step): loads the code objects, validates that no image overflowed the candidate
    # buffer, and lets the clocks settle -- the first ~10 launches after an idle period run ~10 % slow
    for _ in range(8):
        step()
    drain()
    barrier()
    fe.check_overflow()
    for _ in range(a.warmup):
        step()
    drain()
    barrier()
    timer = StageTimer()
    fe.stage_hook = timer
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    drain()
    barrier()
    dt = time.perf_counter() - t0
    fe.stage_hook = None
    fe.check_overflow()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    stage_ms = timer.stage_ms()
    fast_parts = None
    if "fast_threshold" in stage_ms:      # the adaptive detector: threshold estimate (tile sample) + detection + the check
        fast_parts = {"threshold_estimate": round(stage_ms["fast_threshold"], 4), "detect_and_check": round(stage_ms["fast_detect"], 4)}
        stage_ms["fast_detect"] += stage_ms.pop("fast_threshold")
    stage_ms = {n: stage_ms[n] for n in STAGES}
    dom = max(stage_ms, key=stage_ms.get)
    n_proc = n_halo                                          # frames this rank's launches really covered
    achieved = ALGO_BYTES[dom] * n_proc / (stage_ms[dom] * 1e-3) / 1e9
    traffic, valu_insts, valu_ns = measured_counters(dom, n_proc)
    out = {
        "metric": "ORB detect+match frames/sec", "value": round(world * F * a.steps / dt, 2),
        "unit": "stereo frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "configs[1]: stereo ORB detect+match, 1280x720, 2000 kpts/image, "
                               f"{F}-frame stream per GPU resident in HBM, single pyramid level",
                   "frames_per_gpu": F, "keypoints_per_image": KP, "fast_threshold": 10,
                   "parallelism": (f"frames sharded x{world} (+1 halo frame per shard); all_gather of the feature-track records "
                                   f"(8 B per keypoint slot) started at the end of every step on RCCL's stream and awaited one "
                                   f"step later, all of them inside the timed region" if world > 1 else "1 GPU, no collective")},
        "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
        "fast_detect_parts_ms": fast_parts,
        "pipeline_GBps": round(ALGO_BYTES_FRAME * F * a.steps / dt / 1e9 * world, 2),
        "roofline": {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "algorithmic_bytes_per_launch": ALGO_BYTES[dom] * n_proc,
                     "note": "HBM reading of a kernel whose time goes into vector instructions and barrier waits, not into memory (see `valu`): traffic = FETCH_SIZE+WRITE_SIZE of "
                             "a separate rocprofv3 --pmc run (profiles/traffic.json), scaled to this launch"},
    }
    # every stage with its two readings: algorithmic bytes over the stage time against HBM, and -- where the committed PMC
    # summary has SQ_INSTS_VALU for its kernel -- vector wave-instructions over the stage time against the VALU issue peak
    # (1.04 ns per wave-instruction per SIMD).  orient_rbrief: eight keypoints per wave, 117 vector instructions per keypoint
    # (round 3: 232); what it waits for is the patch gather -- 92 L1 <- L2 line requests of 128 bytes per keypoint
    # (profiles/pmc_orient_r04.txt), 0.77 ms of its 2.17 ms without them -- neither HBM (3.7 GB compulsory) nor issue.
    rs = {}
    for name in STAGES:
        b = ALGO_BYTES[name] * n_proc
        t_ms = stage_ms[name]
        tr, vi, _ = measured_counters(name, n_proc)
        e = {"bound": "hbm", "algorithmic_bytes_per_launch": b, "achieved": round(b / (t_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS,
             "unit": "GB/s", "frac": round(b / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": tr, "ms": round(t_ms, 4)}
        if vi:
            e["valu_wave_insts_per_launch"] = int(vi)
            e["valu_issue_frac"] = round(vi * 1.04e-9 / N_SIMD / (t_ms * 1e-3), 3)
        rs[name] = e
    out["roofline_stages"] = rs
    if valu_insts and valu_ns:
        # The roof that binds: vector-ALU instruction issue.  SQ_INSTS_VALU of the launch spread over the chip's SIMDs
        # at (a) the full issue rate (v_add/v_xor class, 1.04 ns per wave-instruction per SIMD, measured by
        # tools/ubench/valu_rate.hip) = the VALU-issue PEAK, and (b) the rate of the op classes that make up most of this
        # kernel (v_min3/v_max3_i32, v_alignbyte, v_bfe, SDWA, v_dot4: 1.7 ns = 0.57x).  The measured time lies between
        # the two floors: the kernel issues vector instructions back to back.
        full_ns = 1.04
        t_full = valu_insts * full_ns * 1e-9 / N_SIMD * 1e3
        t_mix = valu_insts * valu_ns * 1e-9 / N_SIMD * 1e3
        out["roofline"]["valu"] = {
            "bound": "valu-issue", "wave_insts_per_launch": int(valu_insts), "simds": N_SIMD,
            "achieved": round(valu_insts / (stage_ms[dom] * 1e-3) / 1e9, 1), "peak": round(N_SIMD / full_ns, 1),
            "unit": "G wave-instructions/s", "frac": round(t_full / stage_ms[dom], 3),
            "floor_ms_at_full_rate": round(t_full, 3), "floor_ms_if_every_op_were_0.57x_class": round(t_mix, 3),
            "measured_ms": round(stage_ms[dom], 3),
            "note": "SQ_INSTS_VALU (PMC, profiles/traffic.json) vs the VALU issue peak; the integer min/max/byte-select ops "
                    "of FAST issue at 0.57x the full rate on gfx950 (profiles/valu_issue_rates_*.txt), so frac ~0.65 of the "
                    "full-rate peak is this instruction mix's ceiling.  Rounds 1-3 and the first adaptive detector sat at it "
                    "(0.65-0.9); since the strip pre-test and the matrix-core smoothing of round 4 the kernel issues 0.82 M "
                    "instead of 1.99 M vector instructions per image and a part of its time is waves parked at its five "
                    "barriers (SQ_WAIT_ANY 52 % of the wave-cycles, profiles/pmc_frontend_r04.txt)"}
    # second bound, informational: the track matcher is an int8 GEMM on the matrix cores (2000 x 2000 x 256 multiply-
    # accumulates per image pair = the algorithmic work of brute-force Hamming matching in its dot-product form)
    mm_ops = 2.0 * (n_proc - 1) * KP * KP * 256
    out["roofline_matcher"] = {"kernel": "hamming_track", "bound": "mfma", "achieved": round(mm_ops / (stage_ms["hamming_track"] * 1e-3) / 1e12, 1),
                               "peak": INT8_PEAK_TOPS, "unit": "TOP/s", "frac": round(mm_ops / (stage_ms["hamming_track"] * 1e-3) / 1e12 / INT8_PEAK_TOPS, 4),
                               "note": "dense int8 MFMA peak = 2x the ~2.5 PFLOP/s bf16 peak (MI355X_MICROARCH.md, matrix cores)"}
    if world > 1:
        out["stage_ms"]["gather_tracks_and_rest"] = round(dt / a.steps * 1e3 - sum(stage_ms.values()), 4)
    cores = usable_cores()
    if rank == 0 and not a.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline_frontend(0, cores, a.cpu_frames)
    if rank == 0 and a.pyramid and not a.no_pyramid and world == 1:
        # extra figure, not the headline: the same stream through the 8-level x1.2 ORB pyramid
        # (2,853,088 px per image, per-level quotas summing to 2000 keypoints)
        del fe
        torch.cuda.empty_cache()
        fp = StereoOrbFrontend(H, W, max_frames=F, device=device,
                               params=ImageProcessorParams(max_features=KP, n_levels=8, scale_factor=1.2))
        fp.process(images, check=True)
        torch.cuda.synchronize()
        tp = time.perf_counter()
        for _ in range(a.steps):
            fp.process(images, check=False)
        torch.cuda.synchronize()
        dtp = (time.perf_counter() - tp) / a.steps
        out["pyramid8"] = {"value": round(F / dtp, 2), "unit": "stereo frames/s", "ms_per_step": round(dtp * 1e3, 4),
                           "config": "8 levels x1.2 (1280x720 ... 357x201), detector/top-K/descriptor per level, "
                                     "level-major merge to 2000 keypoints per image, same matchers",
                           "algorithmic_GBps": round(17224352 * F / dtp / 1e9, 2)}
        del fp
    fe = None
    del images
    torch.cuda.empty_cache()
    if not a.no_ba and world == 1:
        from visual_underwater_slam_amd import ba_bench
        ba = ba_bench.run(device, with_object_graph=not a.no_object_graph)
        seq = ba.pop("_seq")
        if first_call is not None and "dropin" in ba:
            ba["dropin"]["first_call_s"] = first_call.get("first_call_s")
            ba["dropin"]["first_call"] = first_call
        if not a.no_end_to_end:
            ba["end_to_end"] = ba_bench.run_end_to_end(device)
        # the reference's complete graph (stereo + IMU + DVL + priors) at its own plumbing size, configs[0] ...
        ba["full_graph_configs0"] = ba_bench.run_full_graph(device, 50, 500, 100)
        # ... and at 2000 keyframes: the sparse track of round 1 (about 120 factors per keyframe) and a dense one at the
        # configs[2] factor count (2000 keyframes / ~50k landmarks / 2.0 M stereo factors + 1999 IMU + 1999 DVL)
        ba["full_graph_2000kf_sparse"] = ba_bench.run_full_graph(device, 2000, 50000, 1000)
        ba["full_graph_2000kf_2Mfactors"] = ba_bench.run_full_graph(device, 2000, 180000, 1000, kf_period=0.04)
        if not a.no_cpu_baseline:
            ba["cpu_baseline"] = cpu_baseline_ba(seq, cores, ba["value"])
        out["ba"] = ba
    if not a.no_ba and world > 1:
        from visual_underwater_slam_amd import ba_bench
        r = ba_bench.run_sharded(device, world, rank, n_kf=a.ba_sharded_kf, n_lm=50 * a.ba_sharded_kf)
        t = torch.tensor([r["seconds"]], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        r["value"], r["unit"], r["metric"] = round(float(t.item()), 4), "s", "landmark-sharded full-batch LM wall time (max over ranks)"
        r.pop("seconds")
        out["ba_sharded"] = r
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
