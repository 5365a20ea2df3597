"""The end-to-end chain of the north star on the CPU oracle: stereo frames -> FAST / rBRIEF / Hamming -> feature ids ->
get_landmarks + batch_create (batch.py:144-176, 253-305) -> LM (batch.py:337).  TEST INFRASTRUCTURE ONLY (imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg): the checker of visual_underwater_slam_amd/sequence.py's
HIP chain, stage by stage.  Every stage is one of oracle.py's `_cpu` twins; nothing here calls the product."""
import numpy as np
import torch

from . import oracle as O


def frontend(frames, max_features, track_max_distance, cross_check, fast_threshold=10, border=31, stereo_threshold=5,
             min_disparity=0, max_disparity=128, stereo_max_distance=64):
    """frames uint8 [F,2,H,W] -> dict of the front-end's outputs (the defaults are ImageProcessorParams')."""
    F, _, H, W = frames.shape
    flat = np.ascontiguousarray(frames).reshape(2 * F, H, W)
    keys, cnt, blur = O.fast_detect(flat, thr=fast_threshold, border=border, cand_cap=max(32768, H * W // 16))
    kp, kc = O.select_topk(keys, cnt, max_features)
    desc, angle = O.orient_rbrief(flat, blur, kp, kc)
    f = np.arange(F, dtype=np.int32)
    sidx, sdist = O.hamming_match(desc, kp, kc, W, 2 * f, 2 * f + 1, stereo_threshold, min_disparity, max_disparity,
                                  stereo_max_distance)
    tidx = tdist = None
    if F > 1:
        tidx, tdist = O.hamming_match(desc, kp, kc, W, 2 * f[:-1], 2 * f[:-1] + 2, -1, 0, 0, track_max_distance)
    if cross_check:
        ridx, _ = O.hamming_match(desc, kp, kc, W, 2 * f + 1, 2 * f, stereo_threshold, -max_disparity, -min_disparity,
                                  stereo_max_distance)
        sidx = O.cross_check(sidx, ridx)
        if F > 1:
            ridx, _ = O.hamming_match(desc, kp, kc, W, 2 * f[:-1] + 2, 2 * f[:-1], -1, 0, 0, track_max_distance)
            tidx = O.cross_check(tidx, ridx)
    ids, feat, n_ids = O.track_ids(sidx, tidx, kp, kc, H, W)
    return dict(kp_keys=kp, kp_count=kc, desc=desc, stereo_idx=sidx, track_idx=tidx, ids=ids, feats=feat, n_ids=n_ids)


def factors(fe, odom_poses, cam, K, gate_px):
    """get_landmarks + the landmark loop of batch_create for all keyframes, then the initial-residual gate."""
    of, oi, om, first, pt = O.emit_stereo_factors(fe["ids"], fe["feats"], odom_poses, cam, fe["n_ids"])
    out = dict(obs_frame=of, obs_id=oi, obs_meas=om, lm_first=first, lm_point=pt)
    if gate_px > 0:
        r0 = O.stereo_initial_residuals(odom_poses, K, pt, of, oi, om)
        keep = np.abs(r0).max(1) <= gate_px
        still = np.zeros(len(first), bool)
        still[oi[keep]] = True
        out.update(obs_frame=of[keep], obs_id=oi[keep], obs_meas=om[keep], lm_first=np.where(still, first, -1),
                   initial_residuals=r0, gate_keep=keep)
    return out


def optimise(fac, seq, n_kf, K, sigma, prior_sigmas):
    """The reference's full graph (stereo factors of keyframes >= 1, ImuFactors, DVL factors, priors on X(0) and V(0))
    through the oracle's LM.  seq: poses_init, imu, dvl, gravity of synth.scene_sequence.  Returns
    (poses, vels, bias, landmark ids, points, report)."""
    from visual_underwater_slam_amd import ba_pack, synth
    from visual_underwater_slam_amd.gtsam.imu import Preintegrator
    seen = np.nonzero(fac["lm_first"] >= 0)[0]
    remap = -np.ones(len(fac["lm_first"]), np.int64)
    remap[seen] = np.arange(len(seen))
    pk = ba_pack.pack_observations(torch.from_numpy(fac["obs_frame"].astype(np.int64)), torch.from_numpy(remap[fac["obs_id"]]),
                                   torch.from_numpy(np.ascontiguousarray(fac["obs_meas"])), n_kf, len(seen))
    P = O.BAProblem(pk, K, sigma, (np.array([0], np.int32), seq["poses_init"][:1], np.asarray(prior_sigmas)[None]))
    I3 = np.eye(3)
    pims, Ws = [], []
    for i in range(1, n_kf):
        pre = Preintegrator(np.zeros(6), I3 * synth.IMU_ACC_COV, I3 * synth.IMU_GYRO_COV, I3 * synth.IMU_INT_COV)
        for smp in seq["imu"][i - 1]:
            pre.integrate(smp[:3], smp[3:6], 0.005)                                  # batch.py:290
        pims.append(pre.packed()); Ws.append(pre.whitening().reshape(-1))
    imu = (np.arange(n_kf - 1), np.arange(1, n_kf), np.array(pims), np.array(Ws))
    dvl = (np.arange(1, n_kf), seq["dvl"][1:], np.full(n_kf - 1, 0.1))
    vpr = (np.array([0]), np.zeros((1, 3)), np.full((1, 3), 0.1))
    N = O.NavFactors(seq["gravity"], imu=imu, dvl=dvl, vprior=vpr)
    poses, vels, bias, points, rep = O.nav_lm_optimize(P, N, seq["poses_init"], np.zeros((n_kf, 3)), np.zeros(6),
                                                       fac["lm_point"][seen])
    return poses, vels, bias, seen, points, rep
