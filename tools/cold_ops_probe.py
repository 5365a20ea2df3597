"""Probe: first-use cost of the pieces of a cold drop-in call (fresh process)."""
import sys, time
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import numpy as np, torch
def T(name, fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print(f"{name:42s} {1e3 * (time.perf_counter() - t):8.2f} ms", flush=True); return r
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
k = np.random.default_rng(0).integers(0, 50000, 1_900_000)
kt = T("H2D 15 MB pageable", lambda: torch.from_numpy(k).cuda())
for rep in range(2):
    T(f"torch.unique(return_inverse) #{rep}", lambda: torch.unique(kt, return_inverse=True))
    T(f"torch.searchsorted #{rep}", lambda: torch.searchsorted(torch.arange(50000, device='cuda'), kt))
    T(f"torch.sort stable #{rep}", lambda: torch.sort(kt, stable=True))
    T(f"torch.bincount+cumsum #{rep}", lambda: torch.cumsum(torch.bincount(kt, minlength=50000), 0))
    T(f"scatter (index_put) #{rep}", lambda: torch.empty_like(kt).index_put_((kt % 1000,), kt))
    T(f"torch.empty 277 MB + fill #{rep}", lambda: torch.empty(277_000_000 // 8, dtype=torch.float64, device='cuda').zero_())
from visual_underwater_slam_amd import _lib
T("_lib.load()", _lib.load)
from visual_underwater_slam_amd import synth
from visual_underwater_slam_amd.ba import StereoBAProblem, StereoBASolver
s = synth.ba_sequence(200, 5000, 300)
def mk():
    return StereoBAProblem(s["obs_pose"], s["obs_point"], s["meas"], 200, len(s["points_gt"]), s["K"], s["sigma"], prior_pose=[0], prior_T=s["poses_gt"][:1], prior_sigmas=s["prior_sigmas"][None])
p = T("small StereoBAProblem #0 (first HIP kernels)", mk)
p = T("small StereoBAProblem #1", mk)
sv = StereoBASolver(p)
x, y = torch.from_numpy(s["poses_init"]).cuda(), torch.from_numpy(s["points_init"]).cuda()
T("small optimize #0", lambda: sv.optimize(x, y))
T("small optimize #1", lambda: sv.optimize(x, y))
