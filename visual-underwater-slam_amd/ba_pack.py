"""Pack stereo observations into the SoA layout of include/vus.h (vus_ba_problem / vus_ba_structure).

This is the vectorised form of the reference's per-observation factor emission loop
(/root/reference/batch.py:295-305): one `GenericStereoFactor3D(StereoPoint2(uL,uR,v), noise, X(i),
L(id), K)` per observation becomes one row of `meas` / `obs_pose` / `obs_point`.

Index plumbing only (sorts, prefix sums, gathers) -- written with torch ops so the same code runs on
the GPU for the 2M-observation problems and on the CPU for tests.  No BA arithmetic happens here.
"""
import torch


def pack_observations(obs_pose, obs_point, meas, n_poses, n_points):
    """obs_* : 1-D integer tensors, meas [n_obs,3] float64, any order.  Returns a dict of tensors:
    L-order (sorted by point, then pose) meas/obs_pose/obs_point/point_ptr/obs_ppos and the P-order
    pose_ptr/pobs_lidx, plus `perm` (L-order row -> input row)."""
    dev = obs_pose.device
    obs_pose = obs_pose.to(torch.int64)
    obs_point = obs_point.to(torch.int64)
    n_obs = obs_pose.numel()
    key = obs_point * n_poses + obs_pose
    if int(n_points) * int(n_poses) < 2 ** 31:      # 32-bit keys: half the radix passes of the sort
        key = key.to(torch.int32)
    key_sorted, perm = torch.sort(key, stable=True)
    if n_obs > 1 and bool((key_sorted[1:] == key_sorted[:-1]).any()):
        raise NotImplementedError("two stereo factors between the same pose and landmark are not supported")
    op = obs_pose[perm]
    ol = obs_point[perm]
    m = meas[perm].contiguous()
    point_ptr = torch.zeros(n_points + 1, dtype=torch.int64, device=dev)
    point_ptr[1:] = torch.cumsum(torch.bincount(ol, minlength=n_points), 0)
    # P-order: stable sort by pose keeps points ascending inside each pose
    _, pobs_lidx = torch.sort(op.to(torch.int32) if n_poses < 2 ** 31 else op, stable=True)
    obs_ppos = torch.empty_like(pobs_lidx)
    obs_ppos[pobs_lidx] = torch.arange(n_obs, device=dev)
    pose_ptr = torch.zeros(n_poses + 1, dtype=torch.int64, device=dev)
    pose_ptr[1:] = torch.cumsum(torch.bincount(op, minlength=n_poses), 0)
    i32 = torch.int32
    return {
        "n_poses": int(n_poses), "n_points": int(n_points), "n_obs": int(n_obs),
        "meas": m, "obs_pose": op.to(i32), "obs_point": ol.to(i32), "point_ptr": point_ptr.to(i32),
        "obs_ppos": obs_ppos.to(i32), "pose_ptr": pose_ptr.to(i32), "pobs_lidx": pobs_lidx.to(i32),
        "perm": perm,
    }


def pack_observations_device(obs_pose, obs_point, meas, n_poses, n_points):
    """pack_observations() by the kernels of csrc/pack.hip (vus_ba_pack_observations): same dict, same contents, for CUDA
    tensors -- obs_pose / obs_point int32, meas [n_obs,3] float64, any row order.  No torch index operator is involved
    (their lazily loaded code objects cost the first call of a process ~90 ms); one small device read at the end."""
    from . import _lib
    dev = obs_pose.device
    n_obs = int(obs_pose.numel())
    i32 = dict(dtype=torch.int32, device=dev)
    obs_pose = obs_pose.to(torch.int32).contiguous()
    obs_point = obs_point.to(torch.int32).contiguous()
    meas = meas.to(torch.float64).contiguous()
    out = {"n_poses": int(n_poses), "n_points": int(n_points), "n_obs": n_obs,
           "meas": torch.empty((n_obs, 3), dtype=torch.float64, device=dev), "obs_pose": torch.empty(n_obs, **i32),
           "obs_point": torch.empty(n_obs, **i32), "point_ptr": torch.empty(int(n_points) + 1, **i32),
           "obs_ppos": torch.empty(n_obs, **i32), "pose_ptr": torch.empty(int(n_poses) + 1, **i32),
           "pobs_lidx": torch.empty(n_obs, **i32), "perm": torch.empty(n_obs, **i32)}
    flags = torch.empty(2, **i32)                  # [flags, band]
    nbytes = int(_lib.load().vus_pack_work_bytes(n_obs))
    work = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    p = _lib.ptr
    _lib.call("vus_ba_pack_observations", p(obs_pose), p(obs_point), p(meas), n_obs, int(n_poses), int(n_points),
              p(out["meas"]), p(out["obs_pose"]), p(out["obs_point"]), p(out["point_ptr"]), p(out["obs_ppos"]),
              p(out["pose_ptr"]), p(out["pobs_lidx"]), p(out["perm"]), p(flags), p(flags[1:]), p(work), nbytes,
              _lib.current_stream_ptr())
    f, out["band"] = (int(v) for v in flags.tolist())
    if f & 2:
        raise IndexError("a stereo factor refers to a pose or landmark index outside the problem")
    if f & 1:
        raise NotImplementedError("two stereo factors between the same pose and landmark are not supported")
    return out


def build_structure(pk):
    """Non-zero blocks of the reduced camera system and, per block, the pairs of P-order slots that
    see a common point (vus_ba_structure)."""
    dev = pk["obs_pose"].device
    n_obs, nP = pk["n_obs"], pk["n_poses"]
    op = pk["obs_pose"].to(torch.int64)
    ol = pk["obs_point"].to(torch.int64)
    pptr = pk["point_ptr"].to(torch.int64)
    ppos = pk["obs_ppos"].to(torch.int64)
    if n_obs == 0:
        z = torch.zeros(0, dtype=torch.int32, device=dev)
        return {"band": 0, "n_blocks": 0, "n_pairs": 0, "blk_ptr": torch.zeros(1, dtype=torch.int32, device=dev),
                "blk_i": z, "blk_k": z, "pair_a": z, "pair_b": z}
    # Everything below is sized by the number of co-observation pairs (57 M at configs[2]): 32-bit indices and a
    # 32-bit sort key wherever they fit halve the bytes every pass moves and the radix passes of the sort.
    i32 = torch.int32
    small = n_obs < 2 ** 31 and nP * nP < 2 ** 31
    idt = i32 if small else torch.int64
    ar = torch.arange(n_obs, device=dev)
    seg_start = pptr[ol]
    counts = ar - seg_start + 1                       # observation a pairs with seg_start..a
    total = int(counts.sum().item())
    if total >= 2 ** 31:
        raise NotImplementedError(f"{total} co-observation pairs exceed the int32 pair index")
    a_idx = torch.repeat_interleave(ar.to(idt), counts)
    excl = (torch.cumsum(counts, 0) - counts).to(idt)
    b_idx = seg_start.to(idt)[a_idx] + (torch.arange(total, device=dev, dtype=idt) - excl[a_idx])
    op_s = op.to(idt)
    pi, pk_ = op_s[a_idx], op_s[b_idx]                # pi >= pk_ (poses ascend inside a point)
    band = int((pi - pk_).max().item())
    key = pi * nP + pk_
    del pi, pk_
    # the P-order slots of both observations travel through the sort as ONE 64-bit payload: gathering them before
    # the sort walks ppos almost sequentially (a_idx ascends, b_idx ascends inside every landmark), after it there
    # is a single random 8-byte gather instead of four 4-byte ones
    payload = (ppos[a_idx].to(torch.int64) << 32) | ppos[b_idx].to(torch.int64)
    del a_idx, b_idx
    key_sorted, order = torch.sort(key, stable=True)
    del key
    payload = payload[order]
    del order
    ukey, cnt = torch.unique_consecutive(key_sorted, return_counts=True)
    del key_sorted
    blk_ptr = torch.zeros(ukey.numel() + 1, dtype=torch.int64, device=dev)
    blk_ptr[1:] = torch.cumsum(cnt, 0)
    return {
        "band": band, "n_blocks": int(ukey.numel()), "n_pairs": total,
        "blk_ptr": blk_ptr.to(i32), "blk_i": torch.div(ukey, nP, rounding_mode="floor").to(i32),
        "blk_k": (ukey % nP).to(i32),
        "pair_a": (payload >> 32).to(i32).contiguous(), "pair_b": (payload & 0xFFFFFFFF).to(i32).contiguous(),
    }
