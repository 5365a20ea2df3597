"""Where the FIRST optimize() of a process spends its set-up time (batch.py:337 is called once per process): every phase
of StereoBAProblem / StereoBASolver construction timed separately in a fresh process, synchronised after each."""
import json, sys, time
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import numpy as np
import torch
from visual_underwater_slam_amd import synth, ba_pack, _lib
from visual_underwater_slam_amd import ba as B

n_kf = synth.CONFIGS2_BA[0]
s = synth.ba_sequence(*synth.CONFIGS2_BA)
nL = len(s["points_gt"])
torch.cuda.init(); torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()
out = {}
def phase(name, fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    out[name] = round(1e3 * (time.perf_counter() - t), 2); return r
for rnd in ("first", "second"):
    out = {}
    dev = torch.device("cuda:0")
    op = phase("upload_obs_pose", lambda: torch.as_tensor(s["obs_pose"]).to(dev, torch.int32))
    ol = phase("upload_obs_point", lambda: torch.as_tensor(s["obs_point"]).to(dev, torch.int32))
    me = phase("upload_meas", lambda: torch.as_tensor(s["meas"]).to(dev, torch.float64))
    pk = phase("pack_observations_device", lambda: ba_pack.pack_observations_device(op, ol, me, n_kf, nL))
    st = {"band": B.band_of(pk)}
    tl = phase("build_tiles_device", lambda: B.build_tiles_device(pk, st["band"]))
    del tl
    def alloc():
        f64 = dict(dtype=torch.float64, device=dev)
        return [torch.empty((pk["n_obs"], 18), **f64), torch.empty((n_kf, st["band"] + 1, 36), **f64),
                torch.empty((int(_lib.load().vus_ba_band_solve_work_doubles(n_kf, st["band"], 1)),), **f64)]
    bufs = phase("alloc_W_Sband_work", alloc)
    del bufs, st, pk, op, ol, me
    prob = phase("StereoBAProblem", lambda: B.StereoBAProblem(s["obs_pose"], s["obs_point"], s["meas"], n_kf, nL, s["K"], s["sigma"],
                                                             prior_pose=[0], prior_T=s["poses_gt"][:1], prior_sigmas=s["prior_sigmas"][None]))
    sv = phase("StereoBASolver", lambda: B.StereoBASolver(prob))
    p0, x0 = torch.from_numpy(s["poses_init"]).cuda(), torch.from_numpy(s["points_init"]).cuda()
    phase("lm", lambda: sv.optimize(p0, x0))
    print(json.dumps({"round": rnd, "ms": out}), flush=True)
    del sv, prob
    if rnd == "first":
        torch.cuda.empty_cache()
