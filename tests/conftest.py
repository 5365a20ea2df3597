import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


def same_lm_trajectory(iterations, outer, tries, status, err_hist, orep):
    """The GPU LM walked the oracle's trajectory: same linearisations, trials and status.  The number of ACCEPTED steps
    may differ by one only through a round-off tie at the optimum: the last trial changes the error in the 15th digit
    and `cost_change > 0` is decided by summation order (the cooperative back-substitution sums with f64 atomics)."""
    assert (outer, tries, status) == (orep["outer"], orep["tries"], orep["status"])
    if iterations != orep["iterations"]:
        h = list(err_hist)
        assert abs(iterations - orep["iterations"]) == 1 and len(h) >= 2 and abs(h[-1] - h[-2]) <= 1e-12 * abs(h[-1]), \
            (iterations, orep["iterations"], h[-3:])


@pytest.fixture
def band_tuning():
    """Set the band solve's tuning knobs for one test (vus_ba_set_tuning) and restore them afterwards."""
    from visual_underwater_slam_amd import _lib
    lib = _lib.load()
    before = {k: lib.vus_ba_get_tuning(k) for k in (_lib.TUNE_BAND_MODE, _lib.TUNE_CB_MAX_WG, _lib.TUNE_WIN_FAULT)}

    def set_(band_mode=None, cb_max_wg=None, win_fault=None):
        if win_fault is not None:
            _lib.call("vus_ba_set_tuning", _lib.TUNE_WIN_FAULT, int(win_fault))
        if band_mode is not None:
            _lib.call("vus_ba_set_tuning", _lib.TUNE_BAND_MODE, int(band_mode))
        if cb_max_wg is not None:
            _lib.call("vus_ba_set_tuning", _lib.TUNE_CB_MAX_WG, int(cb_max_wg))
    yield set_
    for k, v in before.items():
        _lib.call("vus_ba_set_tuning", k, v)
