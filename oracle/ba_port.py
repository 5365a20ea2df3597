"""CPU "port" of the full-batch LM for bench.py's cpu_baseline leg (TEST INFRASTRUCTURE, not the product).

Same Levenberg-Marquardt control flow as the oracle's vus_ba_lm_optimize_cpu and the product's ba.py (gtsam defaults,
/root/reference/batch.py:337), over the multi-threaded kernels of vus_oracle_ba_mt.c, with the reduced camera system
solved by LAPACK's banded Cholesky (dpbtrf/dpbtrs through scipy, OpenBLAS threads).  tests/test_ba_oracle.py checks
it against the scalar oracle.  Threads: `threads` sets both the OpenMP team and the BLAS pool.
"""
import ctypes
import time
from ctypes import c_double

import numpy as np

from . import oracle as O


def _structure(st):
    """ctypes vus_ba_structure from ba_pack.build_structure()'s dict (numpy or CPU torch tensors)."""
    keep = {k: np.ascontiguousarray(st[k].numpy() if hasattr(st[k], "numpy") else st[k], dtype=np.int32)
            for k in ("blk_ptr", "blk_i", "blk_k", "pair_a", "pair_b")}
    c = O._BAStructure(int(st["band"]), int(st["n_blocks"]), int(st["n_pairs"]), *[O._p(keep[k]).value for k in
                                                                                     ("blk_ptr", "blk_i", "blk_k", "pair_a", "pair_b")])
    return c, keep


class BAPort:
    def __init__(self, P, st, native=False, blas_threads=None):
        """blas_threads: size of the BLAS pool during the LAPACK band solve only (None = leave it as set_threads()
        made it).  OpenBLAS's dpbtrf does not always gain from more threads; bench.py calibrates this."""
        self.P, self.lib = P, O.lib(native=native)
        self.blas_threads = blas_threads
        self.S, self._keep = _structure(st)
        self.band = int(st["band"])
        nP, nL, nO = P.n_poses, P.n_points, P.n_obs
        z = np.zeros
        self.W, self.Y = z((nO, 18)), z((nO, 18))
        self.V, self.Vinv, self.gl, self.dl = z((nL, 6)), z((nL, 6)), z((nL, 3)), z((nL, 3))
        self.Hpp, self.gp, self.gs = z((nP, 36)), z((nP, 6)), z((nP, 6))
        self.ab = z((6 * self.band + 6, 6 * nP))
        self.t = {"linearize": 0.0, "schur": 0.0, "band_solve": 0.0, "backsub": 0.0, "eval_step": 0.0}

    def _timed(self, name, fn):
        t = time.perf_counter()
        r = fn()
        self.t[name] += time.perf_counter() - t
        return r

    def error(self, poses, points):
        e = np.zeros(1)
        O._check(self.lib.vus_ba_error_mt_cpu(self.P.ref(), O._p(poses), O._p(points), O._p(e)), "ba_error_mt")
        return float(e[0])

    def linearize(self, poses, points):
        e = np.zeros(1)
        self._timed("linearize", lambda: O._check(self.lib.vus_ba_linearize_mt_cpu(
            self.P.ref(), O._p(poses), O._p(points), O._p(self.W), O._p(self.V), O._p(self.gl), O._p(self.Hpp),
            O._p(self.gp), O._p(e)), "ba_linearize_mt"))
        return float(e[0])

    def schur(self, lam):
        self._timed("schur", lambda: O._check(self.lib.vus_ba_schur_mt_cpu(
            self.P.ref(), ctypes.byref(self.S), c_double(lam), O._p(self.W), O._p(self.V), O._p(self.gl), O._p(self.Hpp),
            O._p(self.gp), O._p(self.Vinv), O._p(self.Y), O._p(self.ab), O._p(self.gs)), "ba_schur_mt"))

    def band_solve(self):
        """dp = -S^-1 gs by LAPACK dpbtrf + dpbtrs; returns (dp, ok)."""
        import scipy.linalg as sl

        def run():
            try:
                c = sl.cholesky_banded(self.ab, lower=True, overwrite_ab=True, check_finite=False)
            except np.linalg.LinAlgError:
                return None
            return sl.cho_solve_banded((c, True), -self.gs.reshape(-1), check_finite=False).reshape(-1, 6)
        if self.blas_threads is not None:
            import threadpoolctl
            with threadpoolctl.threadpool_limits(limits=int(self.blas_threads), user_api="blas"):
                dp = self._timed("band_solve", run)
        else:
            dp = self._timed("band_solve", run)
        return dp, dp is not None

    def backsub(self, dp):
        self._timed("backsub", lambda: O._check(self.lib.vus_ba_backsub_mt_cpu(
            self.P.ref(), O._p(self.W), O._p(self.Vinv), O._p(self.gl), O._p(dp), O._p(self.dl)), "ba_backsub_mt"))

    def eval_step(self, poses, points, dp):
        npo, npt, out = np.zeros_like(poses), np.zeros_like(points), np.zeros(2)
        self._timed("eval_step", lambda: O._check(self.lib.vus_ba_eval_step_mt_cpu(
            self.P.ref(), O._p(poses), O._p(points), O._p(dp), O._p(self.dl), O._p(npo), O._p(npt), O._p(out)), "ba_eval_mt"))
        return npo, npt, float(out[0]), float(out[1])

    def optimize(self, poses, points, max_seconds=None, **params):
        """Returns (poses, points, report).  `max_seconds`: stop after the trial that crosses this wall time
        (report['truncated'] = True) -- bench.py bounds the baseline's run time with it."""
        prm = dict(O.LM_DEFAULTS); prm.update(params)
        poses = np.array(poses, dtype=np.float64, order="C", copy=True)
        points = np.array(points, dtype=np.float64, order="C", copy=True)
        t0 = time.perf_counter()
        lam = prm["lambda_initial"]
        current = self.error(poses, points)
        rep = {"iterations": 0, "outer": 0, "tries": 0, "status": 1, "initial_error": current, "err_hist": [],
               "lambda_hist": [], "truncated": False}
        while rep["iterations"] < prm["max_iterations"]:
            lin0 = self.linearize(poses, points)
            new_error, stop, accepted = current, False, False
            while True:
                self.schur(lam)
                dp, ok = self.band_solve()
                rep["tries"] += 1
                success = False
                if ok:
                    self.backsub(dp)
                    npo, npt, lin1, new1 = self.eval_step(poses, points, dp)
                    if np.isfinite(lin1) and np.isfinite(new1) and lin0 - lin1 >= 0.0:
                        cost = current - new1
                        if lin0 - lin1 > 2.220446049250313e-16 * lin0:
                            success = cost / (lin0 - lin1) > prm["min_model_fidelity"]
                        stop = abs(cost) < prm["rel_tol"] * current
                        if success:
                            poses, points, new_error = npo, npt, new1
                if max_seconds is not None and time.perf_counter() - t0 > max_seconds:
                    rep["truncated"] = True
                if success:
                    lam = max(prm["lambda_lower"], lam / prm["lambda_factor"]); accepted = True
                    break
                if stop or rep["truncated"]:
                    break
                lam *= prm["lambda_factor"]
                if lam >= prm["lambda_upper"]:
                    rep["status"] = 2
                    break
            rep["err_hist"].append(new_error); rep["lambda_hist"].append(lam)
            rep["outer"] += 1; rep["iterations"] += int(accepted)
            dec = current - new_error
            conv = new_error <= prm["error_tol"] or dec / current <= prm["rel_tol"] or dec <= prm["abs_tol"]
            current = new_error
            if rep["status"] == 2 or rep["truncated"]:
                break
            if conv:
                rep["status"] = 0
                break
        rep["final_error"], rep["final_lambda"] = current, lam
        rep["seconds"] = time.perf_counter() - t0
        rep["stage_s"] = dict(self.t)
        return poses, points, rep


def set_threads(n):
    """OpenMP team of the port's kernels + BLAS pool of scipy; returns a context manager."""
    import threadpoolctl
    O.lib().vus_oracle_set_threads(int(n))
    try:
        O.lib(native=True).vus_oracle_set_threads(int(n))
    except Exception:
        pass
    return threadpoolctl.threadpool_limits(limits=int(n))
