"""Cycle marks of one step (s = 40) of chol_backsolve_kernel's solver workgroup (a -DVUS_TIMING build, VUS_HIP_LIB):
solve -> drain -> [look-ahead wave done] -> barrier -> products for the panel above + next loads -> barrier."""
import ctypes, json, sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import numpy as np
import torch
from visual_underwater_slam_amd import synth, _lib
from visual_underwater_slam_amd.ba import StereoBAProblem, StereoBASolver

n_kf = 2000
s = synth.ba_sequence(n_kf, 25 * n_kf, 1000)
nL = len(s["points_gt"])
prob = StereoBAProblem(s["obs_pose"], s["obs_point"], s["meas"], n_kf, nL, s["K"], s["sigma"], prior_pose=[0],
                       prior_T=s["poses_gt"][:1], prior_sigmas=s["prior_sigmas"][None])
sv = StereoBASolver(prob)
p0, x0 = torch.from_numpy(s["poses_init"]).cuda(), torch.from_numpy(s["points_init"]).cuda()
lib = _lib.load()
rows = []
for it in range(5):
    sv.linearize(p0, x0); sv.schur(1e-5); sv.band_solve(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 32)()
    assert lib.vus_debug_read_wtm(buf) == 0
    t = np.array(list(buf)[16:28], dtype=np.int64)
    rows.append(t - t[0])
med = np.median(np.array(rows), axis=0).astype(int).tolist()
print(json.dumps({"marks_from_step_start": dict(zip(["start", "solved", "drained", "after_barrier_1", "products_done", "end_of_step", "lookahead_flags_seen", "lookahead_y_loaded", "flag_stored", "parked", "rows_requested", "inverse_requested"], med))}))
