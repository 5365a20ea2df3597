"""BASELINE.json configs[4]'s problem size -- 10 000 keyframes x 500 000 landmarks (452,102 observed), 10 M stereo
factors, band 208 pose blocks -- through the HIP path on the ONE GPU of the test box.  (The 8-GPU landmark-sharded run
of configs[4] needs a node this pool does not give; its collectives are covered at world size 2/3 in tests/test_dist.py.
What is established here is that every kernel, the structure builder, the two-sided band solve and the LM loop are
correct at that size.)

As in tests/test_ba_c3_gpu.py the scalar oracle cannot run the whole problem, so parity is established on SUB-PROBLEMS
whose oracle results equal the corresponding slices of the full problem exactly -- a random 1 per mille of the landmarks
with all their observations (V, gl, W rows), three poses with every landmark they see and all observations of those
landmarks (Hpp, gp, gs, complete block rows of the reduced camera system) -- and through size-independent properties:
|S dp + gs| <= 1e-9 |gs|, status 0, monotone LM with every trial accepted, ground truth recovered.
Reference call site: gtsam.LevenbergMarquardtOptimizer(...).optimize(), /root/reference/batch.py:337."""
import numpy as np
import pytest
import torch

from visual_underwater_slam_amd import synth, ba_pack
from test_ba_c3_gpu import band_matvec, relerr

pytestmark = pytest.mark.gpu

N_KF, N_LM, OBS = 10000, 500000, 1000


@pytest.fixture(scope="module")
def c4(gpu):
    from visual_underwater_slam_amd.ba import StereoBAProblem, StereoBASolver
    s = synth.ba_sequence(N_KF, N_LM, OBS)
    nL = len(s["points_gt"])
    prob = StereoBAProblem(s["obs_pose"], s["obs_point"], s["meas"], N_KF, nL, s["K"], s["sigma"],
                           prior_pose=[0], prior_T=s["poses_gt"][:1], prior_sigmas=s["prior_sigmas"][None])
    sv = StereoBASolver(prob)
    yield s, prob, sv
    del sv, prob
    torch.cuda.empty_cache()


def sub_problem(oracle, s, mask):
    used = np.unique(s["obs_point"][mask])
    remap = -np.ones(len(s["points_gt"]), np.int64)
    remap[used] = np.arange(len(used))
    pk = ba_pack.pack_observations(torch.from_numpy(s["obs_pose"][mask]), torch.from_numpy(remap[s["obs_point"][mask]]),
                                   torch.from_numpy(s["meas"][mask]), N_KF, len(used))
    P = oracle.BAProblem(pk, s["K"], s["sigma"], (np.array([0], np.int32), s["poses_gt"][:1], s["prior_sigmas"][None]))
    return P, pk, used


def test_c4_has_the_baseline_size(c4):
    s, prob, sv = c4
    assert prob.n_poses == 10000 and 400000 < prob.n_points <= 500000 and 9.0e6 < prob.n_obs <= 1.0e7
    assert prob.band >= 150 and prob.tiles["n_entries"] > 5.0e6 and sv.use_split


def test_c4_linearisation_matches_oracle_on_a_landmark_sample(c4, oracle):
    s, prob, sv = c4
    sv.linearize(torch.from_numpy(s["poses_init"]).cuda(), torch.from_numpy(s["points_init"]).cuda())
    torch.cuda.synchronize()
    rng = np.random.default_rng(9)
    sel = np.sort(rng.choice(prob.n_points, prob.n_points // 1000, replace=False))
    mask = np.isin(s["obs_point"], sel)
    P, pk, used = sub_problem(oracle, s, mask)
    assert np.array_equal(used, sel)
    lin = oracle.ba_linearize(P, s["poses_init"], s["points_init"][sel])
    assert relerr(sv.V.cpu().numpy()[sel], lin["V"]) < 1e-10
    assert relerr(sv.gl.cpu().numpy()[sel], lin["gl"]) < 1e-10
    assert bool((prob.pk["perm"].cpu() == torch.arange(prob.n_obs)).all())
    assert relerr(sv.W.cpu().numpy()[np.nonzero(mask)[0]], lin["W"]) < 1e-10        # W in L-order on both sides


def test_c4_reduced_camera_rows_match_oracle_on_three_poses(c4, oracle):
    s, prob, sv = c4
    lam = 1e-5
    sv.linearize(torch.from_numpy(s["poses_init"]).cuda(), torch.from_numpy(s["points_init"]).cuda())
    sv.schur(lam)
    torch.cuda.synchronize()
    rows = np.array([0, 4567, 9999])
    seen = np.unique(s["obs_point"][np.isin(s["obs_pose"], rows)])
    mask = np.isin(s["obs_point"], seen)
    P, pk, used = sub_problem(oracle, s, mask)
    lin = oracle.ba_linearize(P, s["poses_init"], s["points_init"][used])
    sch = oracle.ba_schur(P, prob.band, lam, lin)
    assert relerr(sv.Hpp.cpu().numpy()[rows], lin["Hpp"][rows]) < 1e-10
    assert relerr(sv.gp.cpu().numpy()[rows], lin["gp"][rows]) < 1e-10
    assert relerr(sv.gs.cpu().numpy()[rows], sch["gs"][rows]) < 1e-10
    Sg = sv.Sband[torch.from_numpy(rows).cuda()].cpu().numpy()
    assert relerr(Sg, sch["Sband"][rows]) < 1e-10
    assert np.abs(Sg[1, 1:60]).max() > 0


def test_c4_band_solve_residual_and_status(c4):
    s, prob, sv = c4
    sv.linearize(torch.from_numpy(s["poses_init"]).cuda(), torch.from_numpy(s["points_init"]).cuda())
    sv.schur(1e-5)
    S0 = sv.Sband.clone()
    sv.band_solve()
    torch.cuda.synchronize()
    assert int(sv.status.item()) == 0
    res = band_matvec(S0, sv.dp) + sv.gs
    assert float(res.abs().max() / sv.gs.abs().max()) < 1e-9
    del S0


def test_c4_full_lm_converges_to_ground_truth(c4):
    s, prob, sv = c4
    poses, points, rep = sv.optimize(torch.from_numpy(s["poses_init"]).cuda(), torch.from_numpy(s["points_init"]).cuda())
    assert rep.status == 0 and rep.tries == rep.outer == rep.iterations
    hist = [rep.initial_error] + rep.err_hist
    assert all(b <= a for a, b in zip(hist, hist[1:]))
    assert rep.final_error < 1e-3 * rep.initial_error
    # Only X(0) carries a prior (batch.py:281): 10 000 keyframes away the estimate has drifted by what the 1-px
    # measurement noise lets the gauge rotate -- an error that grows with the distance from X(0), not a solver error
    # (measured: 0.24 m at 159 m).  Ground truth is recovered to 0.3 % of that distance, and locally to millimetres.
    t, t_gt = poses.cpu().numpy()[:, 9:], s["poses_gt"][:, 9:]
    dist0 = np.linalg.norm(t_gt - t_gt[0], axis=1)
    assert (np.linalg.norm(t - t_gt, axis=1) <= 0.003 * (1.0 + dist0)).all()
    step_err = np.linalg.norm((t[1:] - t[:-1]) - (t_gt[1:] - t_gt[:-1]), axis=1)
    assert np.median(step_err) < 2e-3 and step_err.max() < 0.02
    lm_err = np.linalg.norm(points.cpu().numpy() - s["points_gt"], axis=1)
    assert np.median(lm_err / (1.0 + np.linalg.norm(s["points_gt"] - t_gt[0], axis=1))) < 0.003
