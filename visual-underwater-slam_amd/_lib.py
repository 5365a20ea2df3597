"""ctypes binding of libvus_hip.so (C ABI: include/vus.h).  Fails loudly when the library is absent."""
import ctypes
import os
from ctypes import c_int, c_void_p, c_char_p, c_double

_HERE = os.path.dirname(os.path.abspath(__file__))
# VUS_HIP_LIB: another build of the same library (A/B timing of kernel variants); never a different implementation
LIB_PATH = os.environ.get("VUS_HIP_LIB") or os.path.join(_HERE, "csrc", "libvus_hip.so")

_P = c_void_p

# name -> argtypes (all return int except where noted); mirrors include/vus.h line by line
SIGNATURES = {
    "vus_fast_score": [_P, c_int, c_int, c_int, c_int, c_int, _P, _P],
    "vus_blur7": [_P, c_int, c_int, c_int, c_int, _P, _P],
    "vus_fast_detect": [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, _P, _P],
    "vus_fast_threshold_estimate": [_P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P],
    "vus_fast_detect_adaptive": [_P, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, c_int, _P, _P],
    "vus_fast_detect_retry": [_P, c_int, c_int, c_int, c_int, c_int, _P, c_int, c_int, _P, c_int, _P, _P, _P, _P],
    "vus_select_topk": [_P, _P, c_int, c_int, c_int, _P, _P, _P],
    "vus_orient_rbrief": [_P, _P, c_int, c_int, c_int, c_int, _P, _P, c_int, _P, _P, _P],
    "vus_orient_order": [_P, _P, c_int, c_int, c_int, c_int, _P, _P],
    "vus_orient_rbrief_ordered": [_P, _P, c_int, c_int, c_int, c_int, _P, _P, c_int, _P, _P, _P, _P],
    "vus_hamming_match": [_P, _P, _P, c_int, c_int, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P],
    "vus_triangulate": [_P, c_int, _P, _P, _P, _P],
    "vus_select_grid": [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P],
    "vus_cross_check": [_P, _P, c_int, c_int, _P, _P],
    "vus_resize_bilinear": [_P, c_int, c_int, c_int, c_int, _P, c_int, c_int, c_int, _P],
    "vus_pyramid_append": [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P,
                           _P],
    "vus_track_ids": [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P],
    "vus_stereo_initial_residuals": [_P, _P, _P, _P, _P, _P, c_int, _P, _P],
    "vus_emit_stereo_factors": [_P, _P, _P, _P, c_int, c_int, c_int, ctypes.c_longlong, _P, _P, _P, _P, _P, _P, _P, _P],
    # bundle adjustment (struct arguments are passed by address)
    "vus_ba_linearize": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "vus_ba_schur": [_P, _P, c_double, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P],
    "vus_ba_tiles_count": [_P, _P, _P],
    "vus_ba_tiles_fill": [_P, c_int, _P, c_int, _P, _P, _P, _P, ctypes.c_longlong, _P],
    "vus_ba_add_diag": [_P, c_int, c_int, c_double, _P],
    "vus_ba_band_solve": [_P, c_int, c_int, _P, _P, _P, _P],
    "vus_ba_backsub": [_P, _P, _P, _P, _P, _P, _P],
    "vus_ba_eval_step": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "vus_ba_error": [_P, _P, _P, _P, _P, _P],
    "vus_ba_band_solve_multi": [_P, c_int, c_int, _P, c_int, _P, _P],
    "vus_ba_band_solve_split": [_P, c_int, c_int, _P, _P, _P, _P, _P],
    "vus_ba_band_solve_multi_split": [_P, c_int, c_int, _P, c_int, _P, _P, _P],
    # graph packing (csrc/pack.hip)
    "vus_imu_preintegrate": [_P, _P, c_int, _P, _P, _P, _P],      # host pointers
    "vus_keys_to_indices": [_P, c_int, _P, _P, _P, _P, ctypes.c_longlong, _P],
    "vus_lookup_keys": [_P, c_int, _P, c_int, _P, _P, _P],
    "vus_ba_pack_observations": [_P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, ctypes.c_longlong, _P],
    "vus_exclusive_scan_i32": [_P, c_int, _P, _P, _P],
    # navigation factors
    "vus_nav_linearize": [_P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "vus_nav_assemble": [c_int, c_int, c_double, _P, _P, _P, _P, _P, _P, _P],
    "vus_nav_border_solve": [c_int, _P, _P, _P, _P, c_double, _P, _P, _P],
    "vus_nav_eval_step": [_P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "vus_nav_error": [_P, c_int, _P, _P, _P, _P, _P, _P],
    # host-only tuning knobs of the band solve (tests, A/B timing)
    "vus_ba_set_tuning": [c_int, c_int],
}

TUNE_BAND_MODE, TUNE_CB_MAX_WG, TUNE_LAST_BAND_MODE, TUNE_WIN_FAULT = 0, 1, 2, 3       # VUS_TUNE_* of include/vus.h
STATUS_WAIT_EXPIRED, STATUS_WINDOW_EXPIRED = -1, -3


class VusError(RuntimeError):
    """A libvus_hip.so entry point returned a negative code (gtsam raises RuntimeError likewise)."""


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64.so (soname libamdhip64.so.7).  Import it FIRST so that this
    # library binds to the same HIP runtime instance; loaded the other way round the process ends
    # up with two runtimes and every launch fails with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C visual-underwater-slam_amd/csrc`. There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.vus_abi_version.restype = c_int
    lib.vus_last_error.restype = c_char_p
    lib.vus_build_target.restype = c_char_p
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library disagree
        fn.argtypes = argtypes
        fn.restype = c_int
    lib.vus_ba_work_doubles.argtypes = [_P]
    lib.vus_ba_work_doubles.restype = ctypes.c_longlong
    lib.vus_ba_band_solve_work_doubles.argtypes = [c_int, c_int, c_int]
    lib.vus_ba_band_solve_work_doubles.restype = ctypes.c_longlong
    lib.vus_pack_work_bytes.argtypes = [c_int]
    lib.vus_pack_work_bytes.restype = ctypes.c_longlong
    lib.vus_ba_tiles_work_bytes.argtypes = [c_int]
    lib.vus_ba_tiles_work_bytes.restype = ctypes.c_longlong
    lib.vus_ba_get_tuning.argtypes = [c_int]
    lib.vus_ba_get_tuning.restype = c_int
    lib.vus_nav_work_doubles.argtypes = [_P]
    lib.vus_nav_work_doubles.restype = ctypes.c_longlong
    _lib = lib
    return lib


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise VusError(f"{name} failed ({rc}): {_explain_missing_code_object(lib.vus_last_error().decode())}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream


def target_mismatch(built: str, device_arch: str):
    """None if code objects built for target ID `built` (e.g. "gfx950:xnack-") load on a device whose gcnArchName is
    `device_arch` (e.g. "gfx950:sramecc+:xnack-"); else the sentence to raise.  A feature the build pins (":xnack-")
    must be the device's setting; a feature it leaves open matches any."""
    b, d = built.split(":"), device_arch.split(":")
    if b[0] != d[0]:
        return (f"libvus_hip.so was built for {built} but the device is {device_arch}: this library targets MI355X "
                f"(gfx950) only")
    for feat in b[1:]:
        if feat not in d[1:]:
            return (f"libvus_hip.so was built for target ID {built} but the device runs {device_arch}: no code object "
                    f"matches (every launch would fail with 'no kernel image is available'). Rebuild with "
                    f"`make -C visual-underwater-slam_amd/csrc clean all OFFLOAD=--offload-arch={b[0]}`")
    return None


def _explain_missing_code_object(text: str) -> str:
    """A launch that fails because no code object of the library matches the device says "no kernel image is available" /
    "invalid device function" and nothing else.  Only THEN is the device's target ID looked up (hipGetDeviceProperties
    costs ~30 ms the first time in a process: measured as 42 -> 75 ms on the first optimize() of a process when it sat in
    require_gpu()) and compared with the one the library was built for."""
    if "no kernel image" not in text and "invalid device function" not in text:
        return text
    try:
        import torch
        arch = getattr(torch.cuda.get_device_properties(torch.cuda.current_device()), "gcnArchName", "")
        why = target_mismatch(load().vus_build_target().decode(), arch) if arch else None
    except Exception:
        why = None
    return f"{text} -- {why}" if why else text


def check_target():
    """Explicit form of the same check (diagnostics, tests): raises if the library's code objects cannot load here."""
    import torch
    arch = getattr(torch.cuda.get_device_properties(torch.cuda.current_device()), "gcnArchName", "")
    why = target_mismatch(load().vus_build_target().decode(), arch) if arch else None
    if why:
        raise RuntimeError(why)


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("visual_underwater_slam_amd needs an MI355X (HIP device); no GPU is visible "
                           "and there is deliberately no CPU fallback.")
