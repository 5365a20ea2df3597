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
