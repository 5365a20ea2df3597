"""Set-up cost of a stereo BA problem at configs[2]: pack, tile-pair structure (csrc/pack.hip), workspace."""
import sys, time; sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torch
from visual_underwater_slam_amd import synth, ba_pack
from visual_underwater_slam_amd.ba import build_tiles_device, band_of, StereoBAProblem, StereoBASolver
size = synth.CONFIGS2_BA
s = synth.ba_sequence(*size)
nL = len(s["points_gt"])
dev = "cuda:0"
op = torch.from_numpy(s["obs_pose"]).to(dev); ol = torch.from_numpy(s["obs_point"]).to(dev); me = torch.from_numpy(s["meas"]).to(dev)
for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    pk = ba_pack.pack_observations_device(op, ol, me, size[0], nL); torch.cuda.synchronize(); t1 = time.perf_counter()
    tl = build_tiles_device(pk, band_of(pk)); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"rep{rep}: pack {1e3*(t1-t):.2f} ms, tile structure {1e3*(t2-t1):.2f} ms ({tl['n_entries']} entries, {tl['n_units']} units)")
    del tl, pk
for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    prob = StereoBAProblem(s["obs_pose"], s["obs_point"], s["meas"], size[0], nL, s["K"], s["sigma"], prior_pose=[0],
                           prior_T=s["poses_gt"][:1], prior_sigmas=s["prior_sigmas"][None])
    torch.cuda.synchronize(); t1 = time.perf_counter()
    sv = StereoBASolver(prob); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"rep{rep}: StereoBAProblem (upload + pack + tiles) {1e3*(t1-t):.2f} ms, StereoBASolver (workspace) {1e3*(t2-t1):.2f} ms")
    del sv, prob
