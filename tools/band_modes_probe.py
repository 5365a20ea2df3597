"""A/B of the band solve's panel-step modes at configs[2] (VUS_TUNE_BAND_MODE): stage times and the LM loop.
usage: python tools/band_modes_probe.py [modes...]   (default: 2 3)"""
import json, sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import numpy as np
import torch
from visual_underwater_slam_amd import synth, _lib, ba_bench
from visual_underwater_slam_amd.ba import StereoBAProblem, StereoBASolver

modes = [int(a) for a in sys.argv[1:]] or [2, 3]
n_kf = synth.CONFIGS2_BA[0]
s = synth.ba_sequence(*synth.CONFIGS2_BA)
nL = len(s["points_gt"])
prob = StereoBAProblem(s["obs_pose"], s["obs_point"], s["meas"], n_kf, nL, s["K"], s["sigma"], prior_pose=[0],
                       prior_T=s["poses_gt"][:1], prior_sigmas=s["prior_sigmas"][None])
sv = StereoBASolver(prob)
p0, x0 = torch.from_numpy(s["poses_init"]).cuda(), torch.from_numpy(s["points_init"]).cuda()
ref = None
for m in modes:
    _lib.call("vus_ba_set_tuning", _lib.TUNE_BAND_MODE, m)
    sv.optimize(p0, x0)
    st = ba_bench.stage_breakdown(sv, p0, x0)
    ts = []
    for _ in range(3):
        poses, pts, rep = sv.optimize(p0, x0)
        ts.append(rep.seconds)
    got = poses.cpu().numpy()
    if ref is None:
        ref = got
    print(json.dumps({"mode": m, "stage_ms": st, "lm_s": round(float(np.median(ts)), 5), "tries": rep.tries, "status": rep.status,
                      "final_error": rep.final_error, "max_diff_vs_first_mode": float(np.abs(got - ref).max())}), flush=True)
