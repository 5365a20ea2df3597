"""Full-batch LM wall time at BASELINE.json configs[2]: 2000 keyframes x 50k landmarks (secondary
metric of bench.py).  Synthetic lawn-mower sequence (synth.ba_sequence), default gtsam LM parameters,
graph emitted through the gtsam-shaped bulk path.  Reports optimize() wall time, the one-off
structure set-up, and a per-stage breakdown measured with HIP events on the launch stream."""
import time

import torch


def stage_breakdown(sv, poses, points, lam=1e-5, reps=3):
    names = ["linearize", "schur", "band_solve", "backsub", "eval_step"]
    acc = {n: 0.0 for n in names}
    for _ in range(reps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        ev[0].record(); sv.linearize(poses, points)
        ev[1].record(); sv.schur(lam)
        ev[2].record(); sv.band_solve()
        ev[3].record(); sv.backsub()
        ev[4].record(); sv.eval_step(poses, points)
        ev[5].record()
        torch.cuda.synchronize()
        for i, n in enumerate(names):
            acc[n] += ev[i].elapsed_time(ev[i + 1]) / reps
    return {k: round(v, 4) for k, v in acc.items()}


def run(device, n_kf=2000, n_lm=50000, obs_per_kf=1000, with_breakdown=True):
    from . import synth
    from .ba import StereoBAProblem, StereoBASolver, LMParams
    t0 = time.perf_counter()
    s = synth.ba_sequence(n_kf, n_lm, obs_per_kf)
    gen_s = time.perf_counter() - t0
    nL = len(s["points_gt"])
    def build():
        return StereoBAProblem(s["obs_pose"], s["obs_point"], s["meas"], n_kf, nL, s["K"], s["sigma"],
                               prior_pose=[0], prior_T=s["poses_gt"][:1], prior_sigmas=s["prior_sigmas"][None],
                               device=device)
    cold = build().setup_seconds          # first call in the process: includes one-off code-object loading
    torch.cuda.empty_cache()
    prob = build()
    sv = StereoBASolver(prob)
    poses0 = torch.from_numpy(s["poses_init"]).to(device)
    points0 = torch.from_numpy(s["points_init"]).to(device)
    sv.optimize(poses0, points0, LMParams(maxIterations=1))          # warm-up (code objects, clocks)
    poses, points, rep = sv.optimize(poses0, points0, LMParams())
    out = {
        "metric": "full-batch LM wall time", "value": round(rep.seconds, 4), "unit": "s",
        "higher_is_better": False, "dtype": "f64",
        "config": {"workload": "configs[2]: stereo BA, synthetic lawn-mower sweep", "keyframes": n_kf,
                   "landmarks": nL, "stereo_factors": prob.n_obs, "band_blocks": prob.band,
                   "schur_blocks": prob.st["n_blocks"], "schur_pairs": prob.st["n_pairs"]},
        "structure_setup_s": round(prob.setup_seconds, 4), "structure_setup_first_call_s": round(cold, 4),
        "data_generation_s": round(gen_s, 2),
        "lm": {"iterations": rep.iterations, "linearizations": rep.outer, "linear_solves": rep.tries,
               "status": rep.status, "initial_error": rep.initial_error, "final_error": rep.final_error},
        "ms_per_linear_solve": round(1e3 * rep.seconds / max(rep.tries, 1), 3),
        "max_pose_error_m": float((poses[:, 9:].cpu() - torch.from_numpy(s["poses_gt"][:, 9:])).abs().max()),
    }
    if with_breakdown:
        out["stage_ms"] = stage_breakdown(sv, poses0, points0)
    return out


if __name__ == "__main__":
    import json
    print(json.dumps(run(torch.device("cuda:0"))))


def run_full_graph(device, n_kf=2000, n_lm=50000, obs_per_kf=1000):
    """The reference's complete graph (batch.py:270-305): stereo + ImuFactor + DVL + priors, with velocities and
    the shared bias as variables, at the configs[2] size."""
    import numpy as np
    from . import synth
    from .ba import StereoBAProblem, NavBASolver, NavFactors, LMParams
    from .gtsam.imu import Preintegrator
    t0 = time.perf_counter()
    s = synth.nav_sequence(n_kf, n_lm, obs_per_kf)
    pims, Ws = [], []
    I3 = np.eye(3)
    for i in range(1, n_kf):
        pre = Preintegrator(np.zeros(6), I3 * synth.IMU_ACC_COV, I3 * synth.IMU_GYRO_COV, I3 * synth.IMU_INT_COV)
        for smp in s["imu"][i - 1]:
            pre.integrate(smp[:3], smp[3:6], smp[6])
        pims.append(pre.packed()); Ws.append(pre.whitening().reshape(-1))
    gen_s = time.perf_counter() - t0
    nL = len(s["points_gt"])
    prob = StereoBAProblem(s["obs_pose"], s["obs_point"], s["meas"], n_kf, nL, s["K"], s["sigma"], prior_pose=[0],
                           prior_T=s["poses_gt"][:1], prior_sigmas=s["prior_sigmas"][None], device=device, pose_stride=2)
    nav = NavFactors(s["gravity"], imu=(np.arange(n_kf - 1), np.arange(1, n_kf), np.array(pims), np.array(Ws)),
                     dvl=(np.arange(1, n_kf), s["dvl"][1:], np.full(n_kf - 1, 0.1)),
                     vprior=(np.array([0]), s["vels_gt"][:1], np.full((1, 3), 0.1)), device=device)
    sv = NavBASolver(prob, nav)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    args = (d(s["poses_init"]), d(np.zeros_like(s["vels_gt"])), d(np.zeros(6)), d(s["points_init"]))
    sv.optimize(*args, LMParams(maxIterations=1))
    poses, vels, bias, points, rep = sv.optimize(*args, LMParams())
    return {
        "metric": "full-graph LM wall time (stereo + IMU + DVL + priors)", "value": round(rep.seconds, 4), "unit": "s",
        "config": {"keyframes": n_kf, "landmarks": nL, "stereo_factors": prob.n_obs, "imu_factors": n_kf - 1,
                   "dvl_factors": n_kf - 1, "camera_nodes": prob.n_nodes, "band_blocks": prob.band},
        "lm": {"iterations": rep.iterations, "linear_solves": rep.tries, "status": rep.status,
               "initial_error": rep.initial_error, "final_error": rep.final_error},
        "ms_per_linear_solve": round(1e3 * rep.seconds / max(rep.tries, 1), 3),
        "max_pose_error_m": float((poses[:, 9:].cpu() - torch.from_numpy(s["poses_gt"][:, 9:])).abs().max()),
        "max_velocity_error_mps": float((vels.cpu() - torch.from_numpy(s["vels_gt"])).abs().max()),
        "data_generation_s": round(gen_s, 2),
    }
