"""Differential fuzzing of the full graph (stereo + IMU + DVL + priors): random small sequences through the GPU LM
(band solve with the bias border) and the oracle's dense-solve LM.  usage: python tools/fuzz_nav.py [n_cases] [seed]"""
import os
import sys
import numpy as np
import torch
ROOT = __file__.rsplit("/", 2)[0]
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O
import test_nav_gpu as T      # reuses the test module's graph construction (setup)

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for case in range(n_cases):
    n_kf = int(rng.integers(3, 36)); n_lm = int(rng.integers(60, 700)); obs = int(rng.integers(15, 120))
    zp = bool(rng.integers(0, 2))
    s, P, N, prob, sv = T.setup(O, n_kf, n_lm, obs, zero_velocity_prior=zp)
    v0, b0 = np.zeros_like(s["vels_gt"]), np.zeros(6)
    poses, vels, bias, points, rep = sv.optimize(d(s["poses_init"]), d(v0), d(b0), d(s["points_init"]))
    op, ov, ob, opt, orep = O.nav_lm_optimize(P, N, s["poses_init"], v0, b0, s["points_init"])
    tag = f"case {case}: {n_kf} KF / {len(s['points_gt'])} L / {len(s['obs_pose'])} stereo factors, zero-velocity prior {zp}"
    # same trajectory; the ACCEPTANCE of the very last trial may differ when both solvers sit at the optimum and the
    # error changes in the 15th digit (a tie decided by round-off): linearisations, trials and status must agree
    assert (rep.outer, rep.tries, rep.status) == (orep["outer"], orep["tries"], orep["status"]), tag
    assert abs(rep.iterations - orep["iterations"]) <= 1, tag
    assert np.allclose(rep.err_hist, orep["err_hist"], rtol=1e-6), tag
    rel = max(T.relerr(poses.cpu().numpy(), op), T.relerr(points.cpu().numpy(), opt))
    dv = max(np.abs(vels.cpu().numpy() - ov).max(), np.abs(bias.cpu().numpy() - ob).max())
    assert rel < 1e-5 and dv < 1e-5, (tag, rel, dv)
    print(f"ok  {tag}: {rep.iterations} iterations, error {rep.initial_error:.3g} -> {rep.final_error:.3g}, rel {rel:.1e}, |dv,db| {dv:.1e}", flush=True)
print(f"fuzz: {n_cases} full graphs agree with the oracle")
