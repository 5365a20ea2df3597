"""How accurate is the reduced-system solve on a weakly constrained graph?  Rebuilds the fuzz case that differed from the
oracle at the 1e-6 level (59 keyframes, 885 factors) and compares the GPU's and the oracle's dp with an extended-
precision dense solve of the same band system."""
import sys
import numpy as np
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from visual_underwater_slam_amd import synth, ba_pack
from visual_underwater_slam_amd.ba import StereoBAProblem, StereoBASolver
from oracle import oracle as O

rng = np.random.default_rng(777)
for case in range(34):
    n_kf = int(rng.integers(3, 70)); n_lm = int(rng.integers(40, 900)); obs = int(rng.integers(12, 250))
    scale = float(rng.choice([1.0, 1.0, 3.0]))
s = synth.ba_sequence(n_kf, n_lm, obs)
nL = len(s["points_gt"])
prob = StereoBAProblem(s["obs_pose"], s["obs_point"], s["meas"], n_kf, nL, s["K"], s["sigma"], prior_pose=[0],
                       prior_T=s["poses_gt"][:1], prior_sigmas=s["prior_sigmas"][None])
sv = StereoBASolver(prob)
sv.linearize(torch.from_numpy(s["poses_init"]).cuda(), torch.from_numpy(s["points_init"]).cuda())
sv.schur(1e-5)
Sb = sv.Sband.cpu().numpy().copy(); gs = sv.gs.cpu().numpy().copy()
sv.band_solve()
dp_gpu = sv.dp.cpu().numpy().reshape(-1)
dp_cpu, st, _ = O.ba_band_solve(Sb, gs)
dp_cpu = dp_cpu.reshape(-1)
nP, B1 = Sb.shape[0], Sb.shape[1]
A = np.zeros((6 * nP, 6 * nP), np.longdouble)
for i in range(nP):
    for sl in range(min(i, B1 - 1) + 1):
        k = i - sl
        blk = Sb[i, sl].reshape(6, 6)
        if sl == 0:
            blk = np.tril(blk) + np.tril(blk, -1).T
        A[6 * i:6 * i + 6, 6 * k:6 * k + 6] = blk
        if sl:
            A[6 * k:6 * k + 6, 6 * i:6 * i + 6] = blk.T
Ad = A.astype(np.float64)
print("n_kf", n_kf, "factors", len(s["obs_pose"]), "cond(S) ~ %.3g" % np.linalg.cond(Ad))
x = np.linalg.solve(Ad, -gs.reshape(-1))
# two steps of iterative refinement in extended precision
for _ in range(3):
    r = (-gs.reshape(-1).astype(np.longdouble) - A @ x.astype(np.longdouble)).astype(np.float64)
    x = x + np.linalg.solve(Ad, r)
ref = x
rel = lambda a: float(np.abs(a - ref).max() / np.abs(ref).max())
print("rel error of dp vs refined dense solve:  GPU %.3e   oracle %.3e   (GPU vs oracle %.3e)" % (rel(dp_gpu), rel(dp_cpu), float(np.abs(dp_gpu - dp_cpu).max() / np.abs(dp_cpu).max())))
