"""The C-ABI library loads and exports every symbol include/vus.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# entry points that return something else than a status code (bound by hand in _lib.load) or have no device work
HOST_ONLY = ("vus_abi_version", "vus_last_error", "vus_build_target", "vus_ba_work_doubles", "vus_nav_work_doubles",
             "vus_ba_band_solve_work_doubles", "vus_ba_get_tuning", "vus_pack_work_bytes", "vus_imu_preintegrate",
             "vus_ba_tiles_work_bytes")


def _declared():
    txt = open(os.path.join(ROOT, "include", "vus.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vus_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_entry_points():
    names = _declared()
    assert "vus_fast_detect" in names and "vus_hamming_match" in names and len(names) >= 9


def test_hip_library_exports_every_declared_symbol():
    import visual_underwater_slam_amd._lib as L
    assert os.path.exists(L.LIB_PATH), "libvus_hip.so not built: run __graft_entry__.build()"
    lib = ctypes.CDLL(L.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), f"{name} declared in vus.h but not exported"
    lib.vus_abi_version.restype = ctypes.c_int
    assert lib.vus_abi_version() == 1
    # the python binding covers every declared compute entry point
    L.load()
    for name in _declared():
        if name not in HOST_ONLY:
            assert name in L.SIGNATURES, f"{name} missing from _lib.SIGNATURES"


def test_build_target_is_reported_and_checked_against_the_device_name():
    import visual_underwater_slam_amd._lib as L
    lib = L.load()
    assert lib.vus_build_target().decode().startswith("gfx950")
    assert L.target_mismatch("gfx950:xnack-", "gfx950:sramecc+:xnack-") is None
    assert L.target_mismatch("gfx950", "gfx950:sramecc+:xnack+") is None               # a generic object runs anywhere
    assert "OFFLOAD=--offload-arch=gfx950" in L.target_mismatch("gfx950:xnack-", "gfx950:sramecc+:xnack+")
    assert "gfx942" in L.target_mismatch("gfx950:xnack-", "gfx942:sramecc+:xnack-")
    # the device is only asked for its target ID when a launch has failed for want of a code object
    assert L._explain_missing_code_object("null buffer") == "null buffer"


def test_oracle_exports_cpu_twins(oracle):
    lib = oracle.lib()
    for name in _declared():
        if name in HOST_ONLY or name == "vus_ba_set_tuning":
            continue
        assert hasattr(lib, name + "_cpu"), f"oracle lacks {name}_cpu"


def test_invalid_arguments_are_rejected_without_a_gpu():
    """Argument validation happens on the host before any launch."""
    import visual_underwater_slam_amd._lib as L
    lib = L.load()
    rc = lib.vus_fast_score(None, 1, 720, 1280, 1280, 10, None, None)
    assert rc == -1 and b"null" in lib.vus_last_error()
    rc = lib.vus_select_topk(ctypes.c_void_p(8), ctypes.c_void_p(8), 1, 10, 100000, ctypes.c_void_p(8),
                             ctypes.c_void_p(8), None)
    assert rc == -1 and b"max_kp" in lib.vus_last_error()
    # the cell-grouped schedule keeps its histogram in LDS: an image of more than 1024 cells of 64 x 64 pixels is refused
    # (callers use vus_orient_rbrief for those), a missing order is refused by the scheduled launch
    rc = lib.vus_orient_order(ctypes.c_void_p(8), ctypes.c_void_p(8), 1, 100, 4096, 4096, ctypes.c_void_p(8), None)
    assert rc == -1 and b"cells" in lib.vus_last_error()
    rc = lib.vus_orient_rbrief_ordered(ctypes.c_void_p(8), ctypes.c_void_p(8), 1, 64, 64, 64, ctypes.c_void_p(8), ctypes.c_void_p(8), 10,
                                       None, ctypes.c_void_p(8), ctypes.c_void_p(8), None)
    assert rc == -1 and b"null" in lib.vus_last_error()


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under the product package may reach it."""
    pkg = os.path.join(ROOT, "visual-underwater-slam_amd")
    banned = ("import oracle", "from oracle", "libvus_oracle", "oracle.oracle", "_cpu(")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                for b in banned:
                    assert b not in txt, f"{f} mentions {b!r}"


def test_tuning_knobs_round_trip_without_a_gpu():
    """vus_ba_set_tuning / vus_ba_get_tuning: host-only, validated, process-wide."""
    import visual_underwater_slam_amd._lib as L
    lib = L.load()
    before = lib.vus_ba_get_tuning(L.TUNE_BAND_MODE), lib.vus_ba_get_tuning(L.TUNE_CB_MAX_WG)
    try:
        L.call("vus_ba_set_tuning", L.TUNE_BAND_MODE, 1)
        L.call("vus_ba_set_tuning", L.TUNE_CB_MAX_WG, 3)
        assert (lib.vus_ba_get_tuning(L.TUNE_BAND_MODE), lib.vus_ba_get_tuning(L.TUNE_CB_MAX_WG)) == (1, 3)
        assert lib.vus_ba_set_tuning(L.TUNE_BAND_MODE, 9) == -1 and b"band mode" in lib.vus_last_error()
        assert lib.vus_ba_set_tuning(99, 0) == -1
    finally:
        L.call("vus_ba_set_tuning", L.TUNE_BAND_MODE, before[0])
        L.call("vus_ba_set_tuning", L.TUNE_CB_MAX_WG, before[1])
