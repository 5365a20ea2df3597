"""The landmark elimination at a BASELINE size: time of one vus_ba_schur call (vinv + the tile-pair kernel on the matrix
cores, right-hand side included) and agreement with a second run.  python tools/schur_ab.py [c2|c4|c2old]
(VUS_HIP_LIB selects another build of the library: tools/ab/build_variant.sh)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_underwater_slam_amd import synth, ba_bench                      # noqa: E402
from visual_underwater_slam_amd.ba import StereoBAProblem, StereoBASolver   # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
size = {"c2": synth.CONFIGS2_BA, "c2old": (2000, 50000, 1000), "c4": (10000, 500000, 1000)}[which]
s = synth.ba_sequence(*size)
nL = len(s["points_gt"])
prob = StereoBAProblem(s["obs_pose"], s["obs_point"], s["meas"], size[0], nL, s["K"], s["sigma"], prior_pose=[0],
                       prior_T=s["poses_gt"][:1], prior_sigmas=s["prior_sigmas"][None])
sv = StereoBASolver(prob)
poses, points = torch.from_numpy(s["poses_init"]).cuda(), torch.from_numpy(s["points_init"]).cuda()
sv.linearize(poses, points)
out = {"size": list(size), "landmarks": nL, "factors": prob.n_obs, "band": prob.band, "pairs": ba_bench.schur_pairs(prob),
       "tile_entries": prob.tiles["n_entries"], "tile_units": prob.tiles["n_units"]}
for _ in range(3):
    sv.schur(1e-5)
torch.cuda.synchronize()
ref = (sv.Sband.clone(), sv.gs.clone())
ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
for i in range(20):
    ev[i].record()
    sv.schur(1e-5)
ev[20].record()
torch.cuda.synchronize()
out["schur_ms"] = round(sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(20))[10], 4)
out["bit_identical_rerun"] = bool(torch.equal(sv.Sband, ref[0]) and torch.equal(sv.gs, ref[1]))
out["stage_ms"] = ba_bench.stage_breakdown(sv, poses, points)
print(json.dumps(out))
