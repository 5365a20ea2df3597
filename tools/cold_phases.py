"""Where the FIRST optimize() of a process spends its set-up time (batch.py:337 is called once per process): every phase
of StereoBAProblem / StereoBASolver construction timed separately in a fresh process, synchronised after each."""
import json, sys, time
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import numpy as np
import torch
from visual_underwater_slam_amd import synth, ba_pack, _lib
from visual_underwater_slam_amd import ba as B

n_kf = 2000
s = synth.ba_sequence(n_kf, 25 * n_kf, 1000)
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
    # the pieces of build_structure_device, one by one (first call: which of them carries the 50 ms?)
    import ctypes
    cp = B._CProblem(n_kf, nL, pk["n_obs"], 0, None, 1.0, None, _lib.ptr(pk["obs_pose"]), _lib.ptr(pk["obs_point"]), _lib.ptr(pk["point_ptr"]),
                     _lib.ptr(pk["obs_ppos"]), _lib.ptr(pk["pose_ptr"]), _lib.ptr(pk["pobs_lidx"]), None, None, None, 1)
    band0 = 224
    rows = phase("st_alloc_rows", lambda: torch.empty((2, n_kf), dtype=torch.int32, device=dev))
    phase("st_count_kernel", lambda: _lib.call("vus_ba_structure_count", ctypes.addressof(cp), band0, _lib.ptr(rows[0]), _lib.ptr(rows[1]), _lib.current_stream_ptr()))
    base = torch.empty((2, n_kf + 1), dtype=torch.int32, device=dev); tot = torch.empty((2,), dtype=torch.int64, device=dev)
    phase("st_scan", lambda: [_lib.call("vus_exclusive_scan_i32", _lib.ptr(rows[q]), n_kf, _lib.ptr(base[q]), _lib.ptr(tot[q:]), _lib.current_stream_ptr()) for q in range(2)])
    nb_, np_ = phase("st_tolist", lambda: [int(v) for v in tot.tolist()])
    lists = phase("st_alloc_lists", lambda: [torch.empty(np_, dtype=torch.int32, device=dev), torch.empty(np_, dtype=torch.int32, device=dev),
                                              torch.empty(nb_ + 1, dtype=torch.int32, device=dev), torch.empty(nb_, dtype=torch.int32, device=dev),
                                              torch.empty(nb_, dtype=torch.int32, device=dev)])
    phase("st_fill_kernel", lambda: _lib.call("vus_ba_structure_fill", ctypes.addressof(cp), band0, _lib.ptr(base[0]), _lib.ptr(base[1]), _lib.ptr(lists[2]),
                                              _lib.ptr(lists[3]), _lib.ptr(lists[4]), _lib.ptr(lists[0]), _lib.ptr(lists[1]), _lib.current_stream_ptr()))
    del lists, rows, base, tot
    st = phase("build_structure_device", lambda: B.build_structure_device(pk))
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
