"""Deterministic synthetic inputs for the hot path.

Everything is generated from counter-based 32-bit integer hashes evaluated with int64 array
arithmetic, so the SAME code runs under numpy (CPU tests, oracle inputs) and torch (on the GPU, for
the resident 1000-frame bench stream) and produces identical bytes.  No RNG stream of numpy or torch
is involved.

Stereo image stream (BASELINE.json configs[1]): a base canvas uint8[1024, 2048] of i.i.d. uniform
8x8-pixel blocks (corner-rich), seed 20261004; the left image of frame t is the 1280x720 canvas
window at offset (2t mod 700, t mod 280); the right image is the same scene shifted by an integer
disparity in 8..40 that is constant per 64-pixel column band; every pixel of every image gets its
own uniform noise in -4..4.
"""
import numpy as np

SEED = 20261004
CANVAS_H, CANVAS_W, BLOCK = 1024, 2048, 8
_M = 0xFFFFFFFF


def _mix32(x):
    """lowbias32 integer hash on int64 arrays holding values < 2**32 (numpy or torch)."""
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & _M
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & _M
    x = x ^ (x >> 16)
    return x


def _mix32_scalar(v):
    v &= _M
    v ^= v >> 16
    v = (v * 0x7FEB352D) & _M
    v ^= v >> 15
    v = (v * 0x846CA68B) & _M
    v ^= v >> 16
    return v


def _arange(xp, n, like=None):
    if xp is np:
        return np.arange(n, dtype=np.int64)
    return xp.arange(n, dtype=xp.int64, device=like)


def disparity_table(W, seed=SEED):
    """Integer disparity (8..40) of each 64-pixel column band of the right image."""
    n = (W + 63) // 64
    return [8 + _mix32_scalar(_mix32_scalar(seed) ^ (0xD15A0000 + b)) % 33 for b in range(n)]


def canvas(xp=np, device=None, seed=SEED):
    """uint8 canvas [1024, 2048] as an int64 array (values 0..255)."""
    by = _arange(xp, CANVAS_H, device) // BLOCK
    bx = _arange(xp, CANVAS_W, device) // BLOCK
    c = by[:, None] * (CANVAS_W // BLOCK) + bx[None, :]
    return _mix32(c ^ _mix32_scalar(seed)) & 255


def stereo_frames(t0, n_frames, H=720, W=1280, xp=np, device=None, seed=SEED, canvas_arr=None):
    """Frames t0 .. t0+n_frames-1 as uint8 [n_frames, 2, H, W] (index 1: 0 = left, 1 = right)."""
    assert H <= 720 + 24 and W <= 1280, "window must stay inside the canvas"
    cv = canvas(xp, device, seed) if canvas_arr is None else canvas_arr
    yy = _arange(xp, H, device)
    xx = _arange(xp, W, device)
    disp = disparity_table(W, seed)
    if xp is np:
        dcol = np.asarray([disp[x // 64] for x in range(W)], dtype=np.int64)
    else:
        dcol = xp.tensor([disp[x // 64] for x in range(W)], dtype=xp.int64, device=device)
    pix = yy[:, None] * W + xx[None, :]
    out = []
    for t in range(t0, t0 + n_frames):
        ox, oy = (2 * t) % 700, t % 280
        pair = []
        for cam in (0, 1):
            cols = ox + xx + (dcol if cam else 0)
            base = cv[oy:oy + H][:, cols]
            k = _mix32_scalar(_mix32_scalar((seed + t) & _M) ^ (0x9E3779B9 * (cam + 1)))
            noise = _mix32(pix ^ k) % 9 - 4
            img = base + noise
            img = img.clip(0, 255) if xp is np else img.clamp(0, 255)
            pair.append(img.astype(np.uint8) if xp is np else img.to(xp.uint8))
        out.append(np.stack(pair) if xp is np else xp.stack(pair))
    return np.stack(out) if xp is np else xp.stack(out)
