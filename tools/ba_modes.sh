#!/bin/bash
# band-solve issue modes side by side (stage times of tools/ba_profile.py's three solves, HIP events)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for m in 2 1 0; do
  echo "== VUS_BAND_MODE=$m"
  VUS_BAND_MODE=$m python3 - <<PY
import torch, sys
sys.path.insert(0, "$ROOT")
from visual_underwater_slam_amd import ba_bench, synth
from visual_underwater_slam_amd.ba import StereoBAProblem, StereoBASolver
s = synth.ba_sequence(2000, 50000, 1000); nL = len(s["points_gt"])
prob = StereoBAProblem(s["obs_pose"], s["obs_point"], s["meas"], 2000, nL, s["K"], s["sigma"], prior_pose=[0], prior_T=s["poses_gt"][:1], prior_sigmas=s["prior_sigmas"][None])
sv = StereoBASolver(prob)
p0 = torch.from_numpy(s["poses_init"]).cuda(); x0 = torch.from_numpy(s["points_init"]).cuda()
for _ in range(2): sv.linearize(p0, x0); sv.schur(1e-5); sv.band_solve()
print(ba_bench.stage_breakdown(sv, p0, x0), "status", int(sv.status.item()))
PY
done
