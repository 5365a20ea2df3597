"""Probe (not a test): how much do schur and band_solve slow each other down when they run at the same time?"""
import sys, time, numpy as np, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from visual_underwater_slam_amd import synth
from visual_underwater_slam_amd.ba import StereoBAProblem, StereoBASolver
s = synth.ba_sequence(*synth.CONFIGS2_BA)
nL = len(s["points_gt"])
prob = StereoBAProblem(s["obs_pose"], s["obs_point"], s["meas"], 2000, nL, s["K"], s["sigma"],
                       prior_pose=[0], prior_T=s["poses_gt"][:1], prior_sigmas=s["prior_sigmas"][None])
sv = StereoBASolver(prob)
poses, points = torch.from_numpy(s["poses_init"]).cuda(), torch.from_numpy(s["points_init"]).cuda()
sv.linearize(poses, points); sv.schur(1e-5); torch.cuda.synchronize()
S0 = sv.Sband.clone(); Sb2 = sv.Sband.clone()
def ev(): return torch.cuda.Event(enable_timing=True)
def timed(fn, n=5):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); a, b = ev(), ev(); a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
def solve_only(): sv.Sband.copy_(S0); sv.band_solve()
def copy_only(): sv.Sband.copy_(S0)
side = torch.cuda.Stream()
main_S = sv.Sband
def schur_side():
    sv.Sband = Sb2
    sv.schur(1e-5)
    sv.Sband = main_S
def both():
    sv.Sband.copy_(S0)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        schur_side()
    sv.band_solve()
    torch.cuda.current_stream().wait_stream(side)
t_copy = timed(copy_only); t_solve = timed(solve_only) - t_copy; t_schur = timed(schur_side); t_both = timed(both) - t_copy
print(f"copy {t_copy:.3f}  band_solve {t_solve:.3f}  schur {t_schur:.3f}  both concurrently {t_both:.3f}  (sum {t_solve + t_schur:.3f})")
print("status", int(sv.status.item()))
