"""Tiny BA run for __graft_entry__.smoke(): 12 keyframes / ~60 landmarks through the gtsam-shaped
API on the GPU, checked against the CPU oracle (test infrastructure, imported only here)."""
import numpy as np
import torch


def run():
    from . import synth, ba_pack
    from . import gtsam
    from .gtsam.symbol_shorthand import X, L
    from oracle import oracle as O      # checker only: smoke() is one of the three allowed importers

    seq = synth.ba_sequence(12, 60, 30)
    nL = len(seq["points_gt"])
    graph, initial = gtsam.NonlinearFactorGraph(), gtsam.Values()
    graph.add(gtsam.PriorFactorPose3(X(0), gtsam.Pose3.from_flat12(seq["poses_init"][0]),
                                     gtsam.noiseModel.Diagonal.Sigmas(seq["prior_sigmas"])))
    K = gtsam.Cal3_S2Stereo(*seq["K"])
    noise = gtsam.noiseModel.Isotropic.Sigma(3, seq["sigma"])
    for i in range(12):
        initial.insert(X(i), gtsam.Pose3.from_flat12(seq["poses_init"][i]))
    for j in range(nL):
        initial.insert(L(j), seq["points_init"][j])
    for a in range(len(seq["obs_pose"])):
        graph.push_back(gtsam.GenericStereoFactor3D(gtsam.StereoPoint2(*seq["meas"][a]), noise,
                                                    X(int(seq["obs_pose"][a])), L(int(seq["obs_point"][a])), K))
    res = gtsam.LevenbergMarquardtOptimizer(graph, initial, gtsam.LevenbergMarquardtParams()).optimize()
    pk = ba_pack.pack_observations(torch.from_numpy(seq["obs_pose"]), torch.from_numpy(seq["obs_point"]),
                                   torch.from_numpy(seq["meas"]), 12, nL)
    st = ba_pack.build_structure(pk)
    P = O.BAProblem(pk, seq["K"], seq["sigma"], (np.array([0], np.int32), seq["poses_init"][:1], seq["prior_sigmas"][None]))
    oposes, opoints, orep = O.ba_lm_optimize(P, st["band"], seq["poses_init"], seq["points_init"])
    got = np.stack([res.atPose3(X(i)).flat12() for i in range(12)])
    rel = np.abs(got - oposes).max() / np.abs(oposes).max()
    assert rel < 1e-6, f"BA poses differ from the oracle: rel {rel:g}"
    print(f"smoke ok: BA 12 keyframes / {nL} landmarks / {len(seq['obs_pose'])} stereo factors, "
          f"LM error {orep['initial_error']:.1f} -> {orep['final_error']:.3f}, poses within {rel:.1e} of the oracle")
