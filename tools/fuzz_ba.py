"""Differential fuzzing of the bundle-adjustment path: random small graphs (sizes, observation densities, start
perturbations) through the full LM on the GPU and on the CPU oracle; the accept/reject sequence, the error and lambda
histories and the optimum must agree (final error 1e-7, optimum 1e-5 relative; north_star allows 1e-4).
usage: python tools/fuzz_ba.py [n_cases] [seed]"""
import sys
import numpy as np
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from visual_underwater_slam_amd import synth, ba_pack
from visual_underwater_slam_amd.ba import StereoBAProblem, StereoBASolver
from oracle import oracle as O

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
for case in range(n_cases):
    n_kf = int(rng.integers(3, 70))
    n_lm = int(rng.integers(40, 900))
    obs = int(rng.integers(12, 250))
    s = synth.ba_sequence(n_kf, n_lm, obs)
    nL = len(s["points_gt"])
    scale = float(rng.choice([1.0, 1.0, 3.0]))          # some cases start three times further from the optimum
    p0 = s["poses_init"].copy(); x0 = s["points_init"].copy()
    p0[:, 9:] = s["poses_gt"][:, 9:] + scale * (p0[:, 9:] - s["poses_gt"][:, 9:])
    x0 = s["points_gt"] + scale * (x0 - s["points_gt"])
    prob = StereoBAProblem(s["obs_pose"], s["obs_point"], s["meas"], n_kf, nL, s["K"], s["sigma"], prior_pose=[0],
                           prior_T=s["poses_gt"][:1], prior_sigmas=s["prior_sigmas"][None])
    sv = StereoBASolver(prob)
    poses, points, rep = sv.optimize(torch.from_numpy(p0).cuda(), torch.from_numpy(x0).cuda())
    pk = ba_pack.pack_observations(torch.from_numpy(s["obs_pose"]), torch.from_numpy(s["obs_point"]),
                                   torch.from_numpy(s["meas"]), n_kf, nL)
    P = O.BAProblem(pk, s["K"], s["sigma"], (np.array([0], np.int32), s["poses_gt"][:1], s["prior_sigmas"][None]))
    op, ox, orep = O.ba_lm_optimize(P, prob.band, p0, x0)
    tag = f"case {case}: {n_kf} KF / {nL} L / {len(s['obs_pose'])} factors, band {prob.band}, start x{scale:g}"
    rel = max(np.abs(poses.cpu().numpy() - op).max() / np.abs(op).max(), np.abs(points.cpu().numpy() - ox).max() / np.abs(ox).max())
    same_path = (rep.iterations, rep.outer, rep.tries, rep.status) == (orep["iterations"], orep["outer"], orep["tries"], orep["status"])
    # intermediate errors may differ at cond(S) * eps (a weakly constrained 59-keyframe graph with cond 4e10 differed
    # by 5e-6 after the first step, tools/solve_accuracy_probe.py); the final error and the optimum must agree tightly
    ok = same_path and np.allclose(rep.err_hist, orep["err_hist"], rtol=1e-4) and \
        np.isclose(rep.err_hist[-1], orep["err_hist"][-1], rtol=1e-7) and \
        np.allclose(rep.lambda_hist, orep["lambda_hist"], rtol=1e-12) and rel < 1e-5
    if not ok:
        print("MISMATCH", tag)
        print("  gpu   :", rep.iterations, rep.outer, rep.tries, rep.status, [f"{e:.10g}" for e in rep.err_hist], rep.lambda_hist)
        print("  oracle:", orep["iterations"], orep["outer"], orep["tries"], orep["status"], [f"{e:.10g}" for e in orep["err_hist"]], orep["lambda_hist"])
        print("  rel", rel)
        raise AssertionError(tag)
    print(f"ok  {tag}: {rep.iterations} iterations, {rep.tries} trials, error {rep.initial_error:.3g} -> {rep.final_error:.3g}, rel {rel:.1e}", flush=True)
print(f"fuzz: {n_cases} graphs agree with the oracle")
