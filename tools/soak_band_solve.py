"""Soak test of the cooperative block-band solve: N repeated solves of the configs[2] reduced system (and a 7-rhs
navigation-sized one); every run must report status 0 and reproduce the first result to 1e-9."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from visual_underwater_slam_amd import synth, _lib
from visual_underwater_slam_amd.ba import StereoBAProblem, StereoBASolver

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
s = synth.ba_sequence(2000, 50000, 1000)
nL = len(s["points_gt"])
prob = StereoBAProblem(s["obs_pose"], s["obs_point"], s["meas"], 2000, nL, s["K"], s["sigma"], prior_pose=[0],
                       prior_T=s["poses_gt"][:1], prior_sigmas=s["prior_sigmas"][None])
sv = StereoBASolver(prob)
poses = torch.from_numpy(s["poses_init"]).cuda(); points = torch.from_numpy(s["points_init"]).cuda()
sv.linearize(poses, points)
ref = None
worst = 0.0
t0 = time.perf_counter()
for it in range(N):
    sv.schur(1e-5)
    sv.band_solve()
    st = int(sv.status.item())
    assert st == 0, f"run {it}: status {st}"
    dp = sv.dp.clone()
    if ref is None:
        ref = dp
    else:
        worst = max(worst, float((dp - ref).abs().max() / ref.abs().max()))
    if it % 50 == 49:
        print(f"{it + 1} solves, worst relative deviation {worst:.2e}", flush=True)
print(f"OK: {N} solves in {time.perf_counter() - t0:.1f} s, worst relative deviation from the first solve {worst:.2e}")
# multi-rhs sweep (navigation border): 4000 nodes, band 60, 7 right-hand sides
rng = np.random.default_rng(0)
nP, B, nr = 4000, 60, 7
Sb = np.zeros((nP, B + 1, 36))
Sb[:, 0] = (np.eye(6) * 50.0).reshape(-1)
Sb[:, 1:] = rng.normal(size=(nP, B, 36)) * 0.05
for i in range(min(B, nP)):
    Sb[i, i + 1:] = 0.0
d_S0 = torch.from_numpy(Sb).cuda(); rhs0 = torch.from_numpy(rng.normal(size=(nr, 6 * nP))).cuda()
d_st = torch.zeros(1, dtype=torch.int32, device="cuda")
ref = None
for it in range(N // 4):
    d_S, d_r = d_S0.clone(), rhs0.clone()
    _lib.call("vus_ba_band_solve_multi", d_S.data_ptr(), nP, B, d_r.data_ptr(), nr, d_st.data_ptr(), _lib.current_stream_ptr())
    assert int(d_st.item()) == 0
    if ref is None:
        ref = d_r.clone()
    else:
        worst = max(worst, float((d_r - ref).abs().max() / ref.abs().max()))
print(f"OK: {N // 4} multi-rhs solves, worst relative deviation {worst:.2e}")
