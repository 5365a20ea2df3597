"""GPU parity of the BA kernels (through the C ABI) against the CPU oracle, fp64.
Tolerances: single kernels 1e-11 relative (summation order / FMA contraction differ), linear solve
1e-8 relative (conditioning), optimised poses/landmarks 1e-6 relative -- well inside the 1e-4 that
BASELINE.json's north_star allows."""
import numpy as np
import pytest
import torch

from visual_underwater_slam_amd import synth, ba_pack
from conftest import same_lm_trajectory

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def setup(oracle, n_kf, n_lm, obs, **kw):
    from visual_underwater_slam_amd.ba import StereoBAProblem, StereoBASolver
    s = synth.ba_sequence(n_kf, n_lm, obs, **kw)
    nL = len(s["points_gt"])
    prob = StereoBAProblem(s["obs_pose"], s["obs_point"], s["meas"], n_kf, nL, s["K"], s["sigma"],
                           prior_pose=[0], prior_T=s["poses_gt"][:1], prior_sigmas=s["prior_sigmas"][None])
    solver = StereoBASolver(prob)
    pk = ba_pack.pack_observations(torch.from_numpy(s["obs_pose"]), torch.from_numpy(s["obs_point"]),
                                   torch.from_numpy(s["meas"]), n_kf, nL)
    P = oracle.BAProblem(pk, s["K"], s["sigma"], (np.array([0], np.int32), s["poses_gt"][:1], s["prior_sigmas"][None]))
    return s, prob, solver, P


# (8, 4000, 2500): ~2300 observations per pose, i.e. three chunks of the row-resident Schur kernel (1000 rows each)
@pytest.mark.parametrize("size", [(12, 60, 30), (50, 500, 100), (150, 4000, 300), (8, 4000, 2500)])
def test_kernels_match_oracle_stage_by_stage(gpu, oracle, size):
    s, prob, sv, P = setup(oracle, *size)
    poses = torch.from_numpy(s["poses_init"]).cuda()
    points = torch.from_numpy(s["points_init"]).cuda()
    # error
    assert np.isclose(sv.error(poses, points), oracle.ba_error(P, s["poses_init"], s["points_init"]), rtol=1e-12)
    # linearise
    sv.linearize(poses, points)
    torch.cuda.synchronize()
    lin = oracle.ba_linearize(P, s["poses_init"], s["points_init"])
    assert np.isclose(float(sv.scal[0]), lin["err"], rtol=1e-12)
    for name in ("W", "V", "gl", "Hpp", "gp"):
        assert relerr(getattr(sv, name).cpu().numpy(), lin[name]) < 1e-11, name
    # Schur complement for two dampings
    for lam in (1e-5, 10.0):
        Y = torch.empty((prob.n_obs, 18), dtype=torch.float64, device="cuda")     # optional output of the call
        sv.Sband.fill_(float("nan")); sv.gs.fill_(float("nan")); sv.Vinv.fill_(float("nan"))
        sv.schur(lam, Y)
        torch.cuda.synchronize()
        sch = oracle.ba_schur(P, prob.band, lam, lin)
        assert relerr(sv.Vinv.cpu().numpy(), sch["Vinv"]) < 1e-10
        assert relerr(Y.cpu().numpy(), sch["Y"]) < 1e-10
        assert relerr(sv.gs.cpu().numpy(), sch["gs"]) < 1e-10
        Sg = sv.Sband.cpu().numpy()
        assert np.isfinite(Sg).all()                                 # every stored block is written (none accumulated into)
        assert relerr(Sg, sch["Sband"]) < 1e-10
        nPq, Bq = prob.n_poses, prob.band                            # diagonal blocks come out whole and symmetric
        D = Sg[:, 0].reshape(nPq, 6, 6)
        assert np.array_equal(D, D.transpose(0, 2, 1))
        sv.schur(lam)
        torch.cuda.synchronize()
        assert np.array_equal(sv.Sband.cpu().numpy(), Sg)            # fixed summation order: bit-identical from run to run
        assert np.array_equal(sv.gs.cpu().numpy(), sv.gs.cpu().numpy())
        # band solve: compare the solution (and the factor on the lower triangles)
        sv.band_solve()
        torch.cuda.synchronize()
        dp, status, Lo = oracle.ba_band_solve(sch["Sband"], sch["gs"])
        assert status == 0 and int(sv.status.item()) == 0
        assert relerr(sv.dp.cpu().numpy(), dp) < 1e-8
        Lg = sv.Sband.cpu().numpy()
        nP, B = prob.n_poses, prob.band
        # blocks left of the 8-pose diagonal panels are stored transposed (read only by the back-substitution)
        for i in range(1, nP):
            for sl in range(1, min(i, B) + 1):
                if i - sl < 8 * (i // 8):
                    assert relerr(Lg[i, sl].reshape(6, 6).T, Lo[i, sl].reshape(6, 6)) < 1e-7
        # the 8-pose diagonal panels hold the INVERSE of their factor block (inverted in place before the sweep) when
        # the band reaches across a panel (>= 7 poses), else the factor itself
        def panel(Lb, k0, pb):
            M = np.zeros((6 * pb, 6 * pb))
            for r in range(pb):
                for c in range(max(0, r - B), r + 1):
                    M[6 * r:6 * r + 6, 6 * c:6 * c + 6] = Lb[k0 + r, r - c].reshape(6, 6)
            return np.tril(M)
        for k0 in range(0, nP, 8):
            pb = min(8, nP - k0)
            Linv = panel(Lg, k0, pb)
            assert relerr(np.linalg.inv(Linv) if B >= 7 else Linv, panel(Lo, k0, pb)) < 1e-7
        # back-substitution and step evaluation
        sv.backsub()
        sv.eval_step(poses, points)
        torch.cuda.synchronize()
        dl = oracle.ba_backsub(P, lin, sch["Vinv"], dp)
        assert relerr(sv.dl.cpu().numpy(), dl) < 1e-8
        npo, npt, lin_err, new_err = oracle.ba_eval_step(P, s["poses_init"], s["points_init"], dp, dl)
        assert relerr(sv.new_poses.cpu().numpy(), npo) < 1e-9
        assert relerr(sv.new_points.cpu().numpy(), npt) < 1e-9
        assert np.isclose(float(sv.scal[1]), lin_err, rtol=1e-7)
        assert np.isclose(float(sv.scal[2]), new_err, rtol=1e-7)


def test_band_solve_reports_indefinite_system(gpu, oracle):
    from visual_underwater_slam_amd import _lib
    nP, B = 20, 3
    Sb = np.zeros((nP, B + 1, 36))
    for i in range(nP):
        Sb[i, 0] = (4.0 * np.eye(6)).reshape(-1)
    Sb[9, 0, 21] = -1.0         # element (3,3) of block (9,9): scalar column 57
    gs = np.ones((nP, 6))
    d_S = torch.from_numpy(Sb).cuda(); d_g = torch.from_numpy(gs).cuda()
    d_x = torch.empty((nP, 6), dtype=torch.float64, device="cuda")
    d_st = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.call("vus_ba_band_solve", d_S.data_ptr(), nP, B, d_g.data_ptr(), d_x.data_ptr(), d_st.data_ptr(),
              _lib.current_stream_ptr())
    _, status, _ = oracle.ba_band_solve(Sb, gs)
    assert int(d_st.item()) == status == 58


def test_band_solve_random_spd_bands(gpu, oracle):
    """Random SPD block-band systems, including band 0, band >= n, and n not a multiple of the panel."""
    from visual_underwater_slam_amd import _lib
    rng = np.random.default_rng(0)
    # the larger cases exercise several panels, several cooperating row groups of the back-substitution
    # (groups = ceil(B / 8)), more groups than panels, and bands that are not multiples of the panel
    for nP, B in [(1, 0), (5, 0), (9, 2), (23, 7), (17, 16), (40, 11), (33, 9), (97, 8), (64, 63), (131, 37),
                  (200, 90), (260, 17)]:
        n = 6 * nP
        A = np.zeros((n, n))
        for i in range(nP):
            for k in range(max(0, i - B), i + 1):
                blk = rng.normal(size=(6, 6))
                A[6 * i:6 * i + 6, 6 * k:6 * k + 6] = blk
        A = np.tril(A) + np.tril(A, -1).T
        A += np.eye(n) * (np.abs(A).sum(1).max() + 1.0)
        Sb = np.zeros((nP, B + 1, 36))
        for i in range(nP):
            for k in range(max(0, i - B), i + 1):
                Sb[i, i - k] = A[6 * i:6 * i + 6, 6 * k:6 * k + 6].reshape(-1)
        gs = rng.normal(size=(nP, 6))
        d_S = torch.from_numpy(Sb).cuda(); d_g = torch.from_numpy(gs).cuda()
        d_x = torch.empty((nP, 6), dtype=torch.float64, device="cuda")
        d_st = torch.zeros(1, dtype=torch.int32, device="cuda")
        _lib.call("vus_ba_band_solve", d_S.data_ptr(), nP, B, d_g.data_ptr(), d_x.data_ptr(), d_st.data_ptr(),
                  _lib.current_stream_ptr())
        x = np.linalg.solve(A, -gs.reshape(-1))
        assert int(d_st.item()) == 0
        assert relerr(d_x.cpu().numpy().reshape(-1), x) < 1e-10, (nP, B)


def test_cooperative_band_solve_is_repeatable(gpu, oracle):
    """The multi-workgroup back-substitution (flags + f64 atomics): 25 repeated solves, status 0 every time and the
    same answer to 1e-12 (tools/soak_band_solve.py runs 400 at the configs[2] size)."""
    s, prob, sv, P = setup(oracle, 150, 4000, 300)
    sv.linearize(torch.from_numpy(s["poses_init"]).cuda(), torch.from_numpy(s["points_init"]).cuda())
    ref = None
    for _ in range(25):
        sv.schur(1e-5)
        sv.band_solve()
        assert int(sv.status.item()) == 0
        dp = sv.dp.clone()
        if ref is None:
            ref = dp
        assert float((dp - ref).abs().max() / ref.abs().max()) < 1e-12


@pytest.mark.parametrize("size", [(50, 500, 100), (150, 4000, 300)])
def test_lm_matches_oracle_trajectory_and_result(gpu, oracle, size):
    """Same accepted/rejected sequence, same error history, same optimum as the CPU oracle."""
    s, prob, sv, P = setup(oracle, *size)
    poses, points, rep = sv.optimize(torch.from_numpy(s["poses_init"]).cuda(), torch.from_numpy(s["points_init"]).cuda())
    oposes, opoints, orep = oracle.ba_lm_optimize(P, prob.band, s["poses_init"], s["points_init"])
    same_lm_trajectory(rep.iterations, rep.outer, rep.tries, rep.status, rep.err_hist, orep)
    assert np.allclose(rep.err_hist, orep["err_hist"], rtol=1e-8)
    assert np.allclose(rep.lambda_hist, orep["lambda_hist"], rtol=1e-12)
    assert relerr(poses.cpu().numpy(), oposes) < 1e-6          # north_star: 1e-4 relative
    assert relerr(points.cpu().numpy(), opoints) < 1e-6
    assert rep.final_error < 1e-3 * rep.initial_error
    assert np.abs(poses.cpu().numpy()[:, 9:] - s["poses_gt"][:, 9:]).max() < 0.05


def test_lm_bad_start_needs_lambda_increase(gpu, oracle):
    s, prob, sv, P = setup(oracle, 20, 120, 40, seed=synth.SEED + 3)
    bad = s["points_init"].copy(); bad[:, 2] += 3.0
    poses, points, rep = sv.optimize(torch.from_numpy(s["poses_init"]).cuda(), torch.from_numpy(bad).cuda())
    _, _, orep = oracle.ba_lm_optimize(P, prob.band, s["poses_init"], bad)
    assert rep.tries == orep["tries"] and rep.outer == orep["outer"]
    assert np.allclose(rep.err_hist, orep["err_hist"], rtol=1e-7)


def test_cheirality_factor_matches_oracle(gpu, oracle):
    """A landmark behind its camera contributes the constant 2*fx residual and zero Jacobians."""
    s, prob, sv, P = setup(oracle, 12, 60, 30)
    pts = s["points_init"].copy()
    pts[3, 2] = -1.0
    sv.linearize(torch.from_numpy(s["poses_init"]).cuda(), torch.from_numpy(pts).cuda())
    torch.cuda.synchronize()
    lin = oracle.ba_linearize(P, s["poses_init"], pts)
    assert np.isclose(float(sv.scal[0]), lin["err"], rtol=1e-12)
    assert relerr(sv.V.cpu().numpy(), lin["V"]) < 1e-11 and np.all(lin["V"][3] == 0)
    assert np.array_equal(sv.V.cpu().numpy()[3], np.zeros(6))


def test_gtsam_shaped_optimize_is_a_drop_in(gpu, oracle):
    """batch.py:336-337 against our module: object-by-object graph and the bulk block give the same
    optimum as the CPU oracle; inputs stay untouched; read-back works like batch.py:57-68."""
    import visual_underwater_slam_amd.gtsam as gtsam
    from visual_underwater_slam_amd.gtsam.symbol_shorthand import X, L, V
    from test_gtsam_boundary import mini_batch_create
    seq = synth.ba_sequence(50, 500, 100)
    graph, initial = mini_batch_create(seq)
    before = initial.atPose3(X(7)).flat12().copy()
    opt = gtsam.LevenbergMarquardtOptimizer(graph, initial, gtsam.LevenbergMarquardtParams())
    results = opt.optimize()
    assert np.array_equal(initial.atPose3(X(7)).flat12(), before)            # inputs untouched
    nL = len(seq["points_gt"])
    pk = ba_pack.pack_observations(torch.from_numpy(seq["obs_pose"]), torch.from_numpy(seq["obs_point"]),
                                   torch.from_numpy(seq["meas"]), 50, nL)
    st = ba_pack.build_structure(pk)
    P = oracle.BAProblem(pk, seq["K"], seq["sigma"], (np.array([0], np.int32), seq["poses_init"][:1], seq["prior_sigmas"][None]))
    oposes, opoints, orep = oracle.ba_lm_optimize(P, st["band"], seq["poses_init"], seq["points_init"])
    got = np.stack([results.atPose3(X(i)).flat12() for i in range(50)])
    assert relerr(got, oposes) < 1e-6
    assert relerr(np.stack([results.atPoint3(L(j)) for j in range(nL)]), opoints) < 1e-6
    assert abs(opt.iterations() - orep["iterations"]) <= 1 and np.isclose(opt.error(), orep["final_error"], rtol=1e-8)
    assert results.atVector(V(0)).tolist() == [0.0, 0.0, 0.0]                # prior-only velocity stays at its prior
    assert np.isclose(graph.error(initial), orep["initial_error"], rtol=1e-10)
    # constr3DPoints of batch.py:57-68
    i, pts = 0, []
    while results.exists(X(i)):
        p = results.atPose3(X(i)); pts.append([p.x(), p.y(), p.z()]); i += 1
    assert i == 50 and np.abs(np.array(pts) - seq["poses_gt"][:, 9:]).max() < 0.05
    # bulk emission gives the same answer
    g2 = gtsam.NonlinearFactorGraph()
    g2.add(gtsam.PriorFactorPose3(X(0), gtsam.Pose3.from_flat12(seq["poses_init"][0]),
                                  gtsam.noiseModel.Diagonal.Sigmas(seq["prior_sigmas"])))
    g2.push_back(gtsam.StereoFactorBlock(seq["meas"], gtsam.noiseModel.Isotropic.Sigma(3, 10.0),
                                         [X(int(i)) for i in seq["obs_pose"]], [L(int(j)) for j in seq["obs_point"]],
                                         gtsam.Cal3_S2Stereo(*seq["K"])))
    r2 = gtsam.LevenbergMarquardtOptimizer(g2, initial, gtsam.LevenbergMarquardtParams()).optimize()
    assert relerr(np.stack([r2.atPose3(X(i)).flat12() for i in range(50)]), got) < 1e-12


def test_band_solve_with_several_row_groups_per_workgroup(gpu, oracle, band_tuning):
    """Bands wider than the resident workgroup count give one workgroup several row groups of the cooperative
    back-substitution.  The knob VUS_TUNE_CB_MAX_WG = 3 forces that path on small systems (solver + 2 helpers serving up to 12
    row groups); answers must equal numpy's dense solve, status 0."""
    from visual_underwater_slam_amd import _lib
    band_tuning(cb_max_wg=3)
    rng = np.random.default_rng(7)
    for nP, B in [(64, 63), (131, 37), (200, 90), (97, 8), (260, 17)]:
        n = 6 * nP
        A = np.zeros((n, n))
        for i in range(nP):
            for k in range(max(0, i - B), i + 1):
                A[6 * i:6 * i + 6, 6 * k:6 * k + 6] = rng.normal(size=(6, 6))
        A = np.tril(A) + np.tril(A, -1).T
        A += np.eye(n) * (np.abs(A).sum(1).max() + 1.0)
        Sb = np.zeros((nP, B + 1, 36))
        for i in range(nP):
            for k in range(max(0, i - B), i + 1):
                Sb[i, i - k] = A[6 * i:6 * i + 6, 6 * k:6 * k + 6].reshape(-1)
        gs = rng.normal(size=(nP, 6))
        d_S = torch.from_numpy(Sb).cuda(); d_g = torch.from_numpy(gs).cuda()
        d_x = torch.empty((nP, 6), dtype=torch.float64, device="cuda")
        d_st = torch.zeros(1, dtype=torch.int32, device="cuda")
        _lib.call("vus_ba_band_solve", d_S.data_ptr(), nP, B, d_g.data_ptr(), d_x.data_ptr(), d_st.data_ptr(),
                  _lib.current_stream_ptr())
        assert int(d_st.item()) == 0
        assert relerr(d_x.cpu().numpy().reshape(-1), np.linalg.solve(A, -gs.reshape(-1))) < 1e-10, (nP, B)
    # ... and the seven-right-hand-side form the inertial graphs use
    nP, B = 90, 40
    n = 6 * nP
    A = np.zeros((n, n))
    for i in range(nP):
        for k in range(max(0, i - B), i + 1):
            A[6 * i:6 * i + 6, 6 * k:6 * k + 6] = rng.normal(size=(6, 6))
    A = np.tril(A) + np.tril(A, -1).T
    A += np.eye(n) * (np.abs(A).sum(1).max() + 1.0)
    Sb = np.zeros((nP, B + 1, 36))
    for i in range(nP):
        for k in range(max(0, i - B), i + 1):
            Sb[i, i - k] = A[6 * i:6 * i + 6, 6 * k:6 * k + 6].reshape(-1)
    rhs = rng.normal(size=(7, n))
    d_S = torch.from_numpy(Sb).cuda(); d_r = torch.from_numpy(rhs).cuda()
    d_st = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.call("vus_ba_band_solve_multi", d_S.data_ptr(), nP, B, d_r.data_ptr(), 7, d_st.data_ptr(), _lib.current_stream_ptr())
    assert int(d_st.item()) == 0
    assert relerr(d_r.cpu().numpy(), np.linalg.solve(A, rhs.T).T) < 1e-10


def _random_band_system(rng, nP, B):
    n = 6 * nP
    A = np.zeros((n, n))
    for i in range(nP):
        for k in range(max(0, i - B), i + 1):
            A[6 * i:6 * i + 6, 6 * k:6 * k + 6] = rng.normal(size=(6, 6))
    A = np.tril(A) + np.tril(A, -1).T
    A += np.eye(n) * (np.abs(A).sum(1).max() + 1.0)
    Sb = np.zeros((nP, B + 1, 36))
    for i in range(nP):
        for k in range(max(0, i - B), i + 1):
            Sb[i, i - k] = A[6 * i:6 * i + 6, 6 * k:6 * k + 6].reshape(-1)
    return A, Sb


@pytest.mark.parametrize("max_wg,mode", [(None, None), (3, None), (None, 0), (None, 1), (None, 2), (None, 3)])
def test_two_sided_band_solve_equals_dense_solve(gpu, oracle, band_tuning, max_wg, mode):
    """vus_ba_band_solve_split / _multi_split: elimination from both ends of the band + dense middle system.  Random
    SPD block bands of many shapes (middle exactly `band` poses or up to 15 more, band not a multiple of the panel,
    systems too short to split -> fallback) against numpy; also with several row groups per workgroup forced."""
    from visual_underwater_slam_amd import _lib
    # mode: None = automatic; 0 the fused launch per panel; 1 both halves share a TRSM + SYRK launch pair on one stream;
    # 2 a launch pair per half on two streams; 3 the persistent window kernel (bands of 16 poses and more)
    band_tuning(band_mode=mode, cb_max_wg=max_wg)
    lib = _lib.load()
    rng = np.random.default_rng(11)
    for nP, B in [(33, 9), (97, 8), (131, 37), (200, 90), (260, 17), (64, 20), (57, 1), (500, 60), (40, 30), (20, 3), (333, 41)]:
        A, Sb = _random_band_system(rng, nP, B)
        n = 6 * nP
        for n_rhs in (1, 7):
            nw = int(lib.vus_ba_band_solve_work_doubles(nP, B, n_rhs))
            assert (nw > 0) == (((nP - B) // 2 // 8) * 8 >= 8)
            work = torch.empty(max(nw, 1), dtype=torch.float64, device="cuda")
            d_S = torch.from_numpy(Sb).cuda()
            d_st = torch.zeros(1, dtype=torch.int32, device="cuda")
            if n_rhs == 1:
                gs = rng.normal(size=(nP, 6))
                d_g = torch.from_numpy(gs).cuda()
                d_x = torch.empty((nP, 6), dtype=torch.float64, device="cuda")
                _lib.call("vus_ba_band_solve_split", d_S.data_ptr(), nP, B, d_g.data_ptr(), d_x.data_ptr(), d_st.data_ptr(),
                          work.data_ptr(), _lib.current_stream_ptr())
                got, exp = d_x.cpu().numpy().reshape(-1), np.linalg.solve(A, -gs.reshape(-1))
            else:
                rhs = rng.normal(size=(n_rhs, n))
                d_r = torch.from_numpy(rhs).cuda()
                _lib.call("vus_ba_band_solve_multi_split", d_S.data_ptr(), nP, B, d_r.data_ptr(), n_rhs, d_st.data_ptr(),
                          work.data_ptr(), _lib.current_stream_ptr())
                got, exp = d_r.cpu().numpy(), np.linalg.solve(A, rhs.T).T
            assert int(d_st.item()) == 0, (nP, B, n_rhs)
            assert relerr(got, exp) < 1e-10, (nP, B, n_rhs)


def test_two_sided_band_solve_reports_indefinite_systems(gpu):
    """A non-positive pivot anywhere -- top half, bottom half or middle -- gives status > 0."""
    from visual_underwater_slam_amd import _lib
    lib = _lib.load()
    nP, B = 120, 10
    for bad_pose in (5, 60, 115):
        Sb = np.zeros((nP, B + 1, 36))
        for i in range(nP):
            Sb[i, 0] = (4.0 * np.eye(6)).reshape(-1)
        Sb[bad_pose, 0, 21] = -1.0
        nw = int(lib.vus_ba_band_solve_work_doubles(nP, B, 1))
        assert nw > 0
        work = torch.empty(nw, dtype=torch.float64, device="cuda")
        d_S = torch.from_numpy(Sb).cuda(); d_g = torch.ones((nP, 6), dtype=torch.float64, device="cuda")
        d_x = torch.empty((nP, 6), dtype=torch.float64, device="cuda")
        d_st = torch.zeros(1, dtype=torch.int32, device="cuda")
        _lib.call("vus_ba_band_solve_split", d_S.data_ptr(), nP, B, d_g.data_ptr(), d_x.data_ptr(), d_st.data_ptr(),
                  work.data_ptr(), _lib.current_stream_ptr())
        assert int(d_st.item()) == 6 * bad_pose + 3 + 1, bad_pose


def test_two_sided_solve_inside_the_lm_gives_the_one_sided_result(gpu, oracle):
    """StereoBASolver picks the two-sided solve for long trajectories; same LM trajectory and optimum as with the
    one-sided solve (and hence as the oracle, test_lm_matches_oracle_trajectory_and_result)."""
    s, prob, sv, P = setup(oracle, 300, 6000, 300, line_len=8)       # band 39 pose blocks
    assert sv.use_split and prob.n_poses >= 2 * prob.band + 64
    p0, x0 = torch.from_numpy(s["poses_init"]).cuda(), torch.from_numpy(s["points_init"]).cuda()
    poses2, points2, rep2 = sv.optimize(p0, x0)
    sv.use_split = False
    poses1, points1, rep1 = sv.optimize(p0, x0)
    assert (rep1.outer, rep1.tries, rep1.status) == (rep2.outer, rep2.tries, rep2.status) and abs(rep1.iterations - rep2.iterations) <= 1
    assert np.allclose(rep1.err_hist, rep2.err_hist, rtol=1e-9)
    assert relerr(poses2.cpu().numpy(), poses1.cpu().numpy()) < 1e-8
    assert relerr(points2.cpu().numpy(), points1.cpu().numpy()) < 1e-7


def _expand_tiles(tl, obs_pose, nP):
    """Every (pose i, pose k, landmark j) a tile structure covers, from its entries (numpy)."""
    dt1 = tl["n_units"] // tl["n_tiles"]
    out = []
    for u in range(tl["n_units"]):
        I, d = divmod(u, dt1)
        for a, b, j, m in tl["entries"][tl["unit_ptr"][u]:tl["unit_ptr"][u + 1]]:
            ra = [a + t for t in range(bin(m & 0xFF).count("1"))]
            rb = [b + t for t in range(bin(m >> 8).count("1"))]
            assert [obs_pose[r] % 8 for r in ra] == [q for q in range(8) if (m >> q) & 1]
            assert [obs_pose[r] % 8 for r in rb] == [q for q in range(8) if (m >> (8 + q)) & 1]
            assert all(obs_pose[r] // 8 == I for r in ra) and all(obs_pose[r] // 8 == I - d for r in rb)
            out += [(obs_pose[x], obs_pose[y], j) for x in ra for y in rb if obs_pose[y] <= obs_pose[x]]
    return out


@pytest.mark.parametrize("size,kw", [((12, 60, 30), {}), ((50, 500, 100), {}), ((77, 900, 60), dict(line_len=9)),
                                     ((150, 4000, 300), {}), ((5, 40, 20), {})])
def test_tile_structure_device_equals_the_plain_statement(gpu, oracle, size, kw):
    """csrc/pack.hip (count, emit, stable radix sort by unit) against oracle/vus_oracle_pack.c (unit by unit, landmark
    by landmark): unit_ptr and entries bit for bit; `order` a permutation by non-increasing size class; and the
    entries expand to exactly the co-observation triples (i >= k, j) of the graph."""
    s, prob, sv, P = setup(oracle, *size, **kw)
    ref = oracle.ba_tiles(P, prob.band)
    tl = {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in prob.tiles.items()}
    assert (tl["band"], tl["n_tiles"], tl["n_units"], tl["n_entries"]) == (ref["band"], ref["n_tiles"], ref["n_units"], ref["n_entries"])
    assert np.array_equal(tl["unit_ptr"], ref["unit_ptr"]) and np.array_equal(tl["entries"][:tl["n_entries"]], ref["entries"])
    size_of = np.diff(tl["unit_ptr"])
    cls = np.where(size_of > 0, np.floor(np.log2(np.maximum(size_of, 1))) + 1, 0)
    assert sorted(tl["order"].tolist()) == list(range(tl["n_units"])) and (np.diff(cls[tl["order"]]) <= 0).all()
    if prob.n_obs < 6000:
        op = prob.pk["obs_pose"].cpu().numpy()
        ol = prob.pk["obs_point"].cpu().numpy()
        got = sorted(_expand_tiles(dict(tl, entries=tl["entries"][:tl["n_entries"]]), op, prob.n_poses))
        want = sorted((op[x], op[y], ol[x]) for j in range(prob.n_points)
                      for x in range(prob.pk["point_ptr"][j], prob.pk["point_ptr"][j + 1])
                      for y in range(prob.pk["point_ptr"][j], x + 1))
        assert got == want


def test_window_kernel_that_cannot_keep_its_workgroups_resident_falls_back_to_launch_pairs(gpu, oracle, band_tuning):
    """chol_window_kernel's flag protocol needs its whole grid resident; when something else holds CUs a
    workgroup never starts, the bounded waits expire (milliseconds, not seconds) and the status word says
    VUS_STATUS_WINDOW_EXPIRED.  optimize() then redoes the trial with the launch-pair mode, latches it, warns -- and
    gives the result of an undisturbed run.  The fault is injected (VUS_TUNE_WIN_FAULT: one workgroup returns at once)."""
    import time
    import warnings
    from visual_underwater_slam_amd import _lib
    lib = _lib.load()
    band_tuning(band_mode=-1, win_fault=0)
    s, prob, sv, P = setup(oracle, 300, 6000, 300, line_len=8)       # band 39 pose blocks
    p0, x0 = torch.from_numpy(s["poses_init"]).cuda(), torch.from_numpy(s["points_init"]).cuda()
    poses_a, points_a, rep_a = sv.optimize(p0, x0)
    assert lib.vus_ba_get_tuning(_lib.TUNE_LAST_BAND_MODE) == 3      # the window kernel is the automatic choice here
    band_tuning(win_fault=1)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        t0 = time.perf_counter()
        poses_b, points_b, rep_b = sv.optimize(p0, x0)
        dt = time.perf_counter() - t0
    assert any("launch-pair" in str(x.message) for x in w) and sv.window_fallbacks == 1
    assert lib.vus_ba_get_tuning(_lib.TUNE_BAND_MODE) == 2 and lib.vus_ba_get_tuning(_lib.TUNE_LAST_BAND_MODE) != 3
    assert dt < 2.0                                                  # the expired waits cost milliseconds
    assert (rep_a.outer, rep_a.tries, rep_a.status) == (rep_b.outer, rep_b.tries, rep_b.status)
    assert np.allclose(rep_a.err_hist, rep_b.err_hist, rtol=1e-9)
    assert relerr(poses_b.cpu().numpy(), poses_a.cpu().numpy()) < 1e-8
    # a mode the caller FORCED is not overridden: the expired wait is raised
    band_tuning(band_mode=3)
    with pytest.raises(RuntimeError, match="timed out"):
        sv.optimize(p0, x0)


def test_one_sided_band_solve_with_the_two_launch_panel_step(gpu, band_tuning):
    """The TRSM + SYRK launch pair (normally used by the two-sided solve) driving a whole one-sided factorisation,
    including bands narrower than a panel, band 0 and shrinking windows at the end of the matrix."""
    from visual_underwater_slam_amd import _lib
    band_tuning(band_mode=1)
    rng = np.random.default_rng(3)
    for nP, B in [(5, 0), (30, 0), (9, 2), (23, 7), (17, 16), (40, 11), (97, 8), (64, 63), (131, 37), (200, 90)]:
        A, Sb = _random_band_system(rng, nP, B)
        for n_rhs in (1, 7):
            d_S = torch.from_numpy(Sb).cuda()
            d_st = torch.zeros(1, dtype=torch.int32, device="cuda")
            rhs = rng.normal(size=(n_rhs, 6 * nP))
            d_r = torch.from_numpy(rhs).cuda()
            _lib.call("vus_ba_band_solve_multi", d_S.data_ptr(), nP, B, d_r.data_ptr(), n_rhs, d_st.data_ptr(),
                      _lib.current_stream_ptr())
            assert int(d_st.item()) == 0, (nP, B, n_rhs)
            assert relerr(d_r.cpu().numpy(), np.linalg.solve(A, rhs.T).T) < 1e-10, (nP, B, n_rhs)


def _thin(seq, keep_frac, rng, single_obs_every=0):
    """Ragged version of a synthetic sequence: a random subset of the observations (every keyframe keeps at least
    12, landmarks that lose all observations are dropped and the rest renumbered); optionally every k-th landmark is
    cut down to ONE observation (a track of length one constrains nothing but must not break the elimination)."""
    n_obs = len(seq["obs_pose"])
    keep = rng.random(n_obs) < keep_frac
    for i in np.unique(seq["obs_pose"]):
        idx = np.nonzero(seq["obs_pose"] == i)[0]
        if keep[idx].sum() < 12:
            keep[idx[:12]] = True
    if single_obs_every:
        for j in np.unique(seq["obs_point"])[::single_obs_every]:
            idx = np.nonzero((seq["obs_point"] == j) & keep)[0]
            keep[idx[1:]] = False
    used = np.unique(seq["obs_point"][keep])
    remap = -np.ones(len(seq["points_gt"]), np.int64)
    remap[used] = np.arange(len(used))
    out = dict(seq)
    out["obs_pose"], out["meas"] = seq["obs_pose"][keep], seq["meas"][keep]
    out["obs_point"] = remap[seq["obs_point"][keep]].astype(seq["obs_point"].dtype)
    out["points_gt"], out["points_init"] = seq["points_gt"][used], seq["points_init"][used]
    return out


@pytest.mark.parametrize("n_kf,line_len,keep,single", [(301, 8, 0.6, 0), (173, 7, 0.8, 5), (90, None, 0.5, 3), (37, 5, 0.9, 2)])
def test_lm_on_ragged_graphs_matches_oracle(gpu, oracle, n_kf, line_len, keep, single):
    """Ragged inputs: random observation subsets, landmarks with a single observation, keyframe counts that are not
    multiples of the 8-pose panel, with and without the two-sided solve -- same LM trajectory and optimum as the oracle."""
    from visual_underwater_slam_amd.ba import StereoBAProblem, StereoBASolver
    rng = np.random.default_rng(n_kf)
    kw = {} if line_len is None else {"line_len": line_len}
    s = _thin(synth.ba_sequence(n_kf, 20 * n_kf, 120, **kw), keep, rng, single)
    nL = len(s["points_gt"])
    assert np.bincount(s["obs_point"], minlength=nL).min() >= 1
    if single:
        assert (np.bincount(s["obs_point"], minlength=nL) == 1).sum() >= 3
    prob = StereoBAProblem(s["obs_pose"], s["obs_point"], s["meas"], n_kf, nL, s["K"], s["sigma"],
                           prior_pose=[0], prior_T=s["poses_gt"][:1], prior_sigmas=s["prior_sigmas"][None])
    sv = StereoBASolver(prob)
    pk = ba_pack.pack_observations(torch.from_numpy(s["obs_pose"]), torch.from_numpy(s["obs_point"]),
                                   torch.from_numpy(s["meas"]), n_kf, nL)
    P = oracle.BAProblem(pk, s["K"], s["sigma"], (np.array([0], np.int32), s["poses_gt"][:1], s["prior_sigmas"][None]))
    poses, points, rep = sv.optimize(torch.from_numpy(s["poses_init"]).cuda(), torch.from_numpy(s["points_init"]).cuda())
    oposes, opoints, orep = oracle.ba_lm_optimize(P, prob.band, s["poses_init"], s["points_init"])
    same_lm_trajectory(rep.iterations, rep.outer, rep.tries, rep.status, rep.err_hist, orep)
    assert np.allclose(rep.err_hist, orep["err_hist"], rtol=1e-7)
    assert relerr(poses.cpu().numpy(), oposes) < 1e-6 and relerr(points.cpu().numpy(), opoints) < 1e-5


def test_graph_without_landmarks_is_a_prior_only_problem(gpu, oracle):
    """Empty stereo input (a graph of pose priors only, zero landmarks and zero observations): the landmark and
    observation arrays are empty, the reduced camera system is block-diagonal, and LM lands on the prior poses."""
    from visual_underwater_slam_amd.ba import StereoBAProblem, StereoBASolver
    s = synth.ba_sequence(5, 40, 20)
    nP = 5
    e_i, e_f = np.zeros(0, np.int32), np.zeros((0, 3))
    pri = (np.arange(nP, dtype=np.int32), s["poses_gt"], np.repeat(s["prior_sigmas"][None], nP, 0))
    prob = StereoBAProblem(e_i, e_i, e_f, nP, 0, s["K"], s["sigma"], prior_pose=pri[0], prior_T=pri[1], prior_sigmas=pri[2])
    assert prob.n_obs == 0 and prob.n_points == 0 and prob.band == 0
    sv = StereoBASolver(prob)
    start = s["poses_init"]
    poses, points, rep = sv.optimize(torch.from_numpy(start).cuda(), torch.zeros(0, 3, dtype=torch.float64).cuda())
    pk = ba_pack.pack_observations(torch.from_numpy(e_i), torch.from_numpy(e_i), torch.from_numpy(e_f), nP, 0)
    P = oracle.BAProblem(pk, s["K"], s["sigma"], pri)
    oposes, opoints, orep = oracle.ba_lm_optimize(P, 0, start, np.zeros((0, 3)))
    same_lm_trajectory(rep.iterations, rep.outer, rep.tries, rep.status, rep.err_hist, orep)
    assert rep.status == 0 and points.shape == (0, 3) and rep.initial_error > 0.1
    assert relerr(poses.cpu().numpy(), oposes) < 1e-9
    assert np.abs(poses.cpu().numpy() - s["poses_gt"]).max() < 1e-6


@pytest.mark.parametrize("case", [(30, 200, 10, 6), (64, 500, 64, 20), (7, 40, 7, 7), (700, 900, 600, 12), (3, 5000, 3, 3)])
def test_tile_structure_and_schur_on_random_cooccurrence_graphs(gpu, oracle, case):
    """Graphs that are not trajectories: narrow and full-width bands, a band of 600 poses, poses with more than 1024
    observations, landmarks seen once, pose counts that are not multiples of the tile.  The device-built tile structure
    == the oracle's plain statement, and the reduced camera system through it == the per-landmark statement."""
    from test_ba_oracle import random_cooccurrence
    from visual_underwater_slam_amd.ba import StereoBAProblem, StereoBASolver
    n_poses, n_points, window, max_obs = case
    rng = np.random.default_rng(n_poses)
    op, ol = random_cooccurrence(rng, n_poses, n_points, window, max_obs)
    K = np.array([400.0, 400, 0, 320, 240, 0.1])
    prob = StereoBAProblem(op, ol, np.zeros((len(op), 3)), n_poses, n_points, K, 1.0)
    sv = StereoBASolver(prob)
    pk = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in prob.pk.items()}
    P = oracle.BAProblem(pk, K, 1.0)
    ref = oracle.ba_tiles(P, prob.band)
    tl = {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in prob.tiles.items()}
    assert tl["n_entries"] == ref["n_entries"] and np.array_equal(tl["unit_ptr"], ref["unit_ptr"])
    assert np.array_equal(tl["entries"][:tl["n_entries"]], ref["entries"])
    nO = prob.n_obs                                        # any W, V, gl, Hpp, gp: the elimination is linear algebra on them
    lin = {"W": rng.normal(size=(nO, 18)), "V": np.abs(rng.normal(size=(n_points, 6))) + np.array([3, 0, 0, 3, 0, 3.0]),
           "gl": rng.normal(size=(n_points, 3)), "Hpp": rng.normal(size=(n_poses, 36)), "gp": rng.normal(size=(n_poses, 6))}
    lin["V"][:, [1, 2, 4]] *= 0.1
    H = lin["Hpp"].reshape(-1, 6, 6)
    lin["Hpp"] = (H + H.transpose(0, 2, 1)).reshape(-1, 36).copy()          # symmetric, as sum H1^T H1 is
    for k, v in lin.items():
        getattr(sv, k).copy_(torch.from_numpy(v))
    sv.schur(0.25)
    torch.cuda.synchronize()
    sch = oracle.ba_schur(P, prob.band, 0.25, lin)
    Sg, So = sv.Sband.cpu().numpy(), sch["Sband"]
    stored = np.array([[i - sl >= 0 for sl in range(prob.band + 1)] for i in range(n_poses)])
    assert relerr(Sg[stored], So[stored]) < 1e-11 and not Sg[~stored].any()
    assert relerr(sv.gs.cpu().numpy(), sch["gs"]) < 1e-11
