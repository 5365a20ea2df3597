#!/usr/bin/env python3
"""Run one linearise + Schur + band solve + backsub + eval on the C3 problem (for rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visual_underwater_slam_amd import synth
from visual_underwater_slam_amd.ba import StereoBAProblem, StereoBASolver
n_kf, n_lm, obs = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else synth.CONFIGS2_BA
s = synth.ba_sequence(n_kf, n_lm, obs)
nL = len(s["points_gt"])
prob = StereoBAProblem(s["obs_pose"], s["obs_point"], s["meas"], n_kf, nL, s["K"], s["sigma"], prior_pose=[0],
                       prior_T=s["poses_gt"][:1], prior_sigmas=s["prior_sigmas"][None])
sv = StereoBASolver(prob)
poses = torch.from_numpy(s["poses_init"]).cuda(); points = torch.from_numpy(s["points_init"]).cuda()
for _ in range(3):
    sv.linearize(poses, points); sv.schur(1e-5); sv.band_solve(); sv.backsub(); sv.eval_step(poses, points)
torch.cuda.synchronize()
print("band", prob.band, "status", int(sv.status.item()))
