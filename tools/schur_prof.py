#!/usr/bin/env python3
"""Ten landmark eliminations at configs[2] (for rocprofv3 --kernel-trace / --pmc on the Schur kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visual_underwater_slam_amd import synth
from visual_underwater_slam_amd.ba import StereoBAProblem, StereoBASolver
size = synth.CONFIGS2_BA
s = synth.ba_sequence(*size)
nL = len(s["points_gt"])
prob = StereoBAProblem(s["obs_pose"], s["obs_point"], s["meas"], size[0], nL, s["K"], s["sigma"], prior_pose=[0],
                       prior_T=s["poses_gt"][:1], prior_sigmas=s["prior_sigmas"][None])
sv = StereoBASolver(prob)
poses = torch.from_numpy(s["poses_init"]).cuda(); points = torch.from_numpy(s["points_init"]).cuda()
sv.linearize(poses, points)
for _ in range(10):
    sv.schur(1e-5)
torch.cuda.synchronize()
