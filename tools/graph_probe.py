"""Probe: the two-sided band solve (~290 launches on two streams) replayed from a captured hipGraph vs launched directly."""
import sys, numpy as np, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from visual_underwater_slam_amd import synth
from visual_underwater_slam_amd.ba import StereoBAProblem, StereoBASolver
s = synth.ba_sequence(2000, 50000, 1000)
nL = len(s["points_gt"])
prob = StereoBAProblem(s["obs_pose"], s["obs_point"], s["meas"], 2000, nL, s["K"], s["sigma"],
                       prior_pose=[0], prior_T=s["poses_gt"][:1], prior_sigmas=s["prior_sigmas"][None])
sv = StereoBASolver(prob)
sv.linearize(torch.from_numpy(s["poses_init"]).cuda(), torch.from_numpy(s["points_init"]).cuda()); sv.schur(1e-5); torch.cuda.synchronize()
S0 = sv.Sband.clone()
def timed(fn, n=7):
    ts = []
    for _ in range(n):
        sv.Sband.copy_(S0); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
t_direct = timed(sv.band_solve); dp_direct = sv.dp.clone()
side = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
sv.Sband.copy_(S0); torch.cuda.synchronize()
with torch.cuda.graph(g, stream=side):
    sv.band_solve()
torch.cuda.synchronize()
t_graph = timed(g.replay)
print(f"direct {t_direct:.3f} ms   graph replay {t_graph:.3f} ms   status {int(sv.status.item())}   max |d dp| {float((sv.dp - dp_direct).abs().max()):.2e}")
