"""Host side of the stereo ORB front-end: the MI355X replacement of the reference's external
image-processor nodelet.

Interface mirrored: the nodelet is configured by the ROS parameters of
/root/reference/launch/stereo.launch:37-47 (`fast_threshold`, `stereo_threshold`, ...) and publishes
`gtsam_vio/CameraMeasurement` messages whose `features[].id, u0, v0, u1, v1` fields are what
/root/reference/batch.py:149-154 reads.  `ImageProcessorParams` carries those parameter names,
`StereoOrbFrontend.process()` runs the kernels on a whole resident batch of stereo frames, and
`camera_measurements()` emits records of exactly that message shape.

Python here only owns buffers and launches: every number is produced by libvus_hip.so.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import torch

from . import _lib


@dataclass
class ImageProcessorParams:
    """Parameter names follow launch/stereo.launch:37-47 where the reference has one."""
    fast_threshold: int = 10       # stereo.launch:43
    stereo_threshold: int = 5      # stereo.launch:47 -- |v_left - v_right| gate in pixels
    max_features: int = 2000       # BASELINE.json configs[1]: 2000 keypoints per image
    border: int = 31               # ORB edge threshold (31x31 patch)
    min_disparity: int = 0         # x_left - x_right gate of the stereo matcher
    max_disparity: int = 128
    stereo_max_distance: int = 64  # Hamming acceptance thresholds
    track_max_distance: int = 64
    cand_cap: int = 0              # candidate slots per image before top-K selection; 0 = sized from the image:
                                   # max(32768, H*W/16) -- strict 3x3 NMS leaves at most H*W/4 candidates, the
                                   # corner-dense bench texture ~H*W/37; an overflow is never silent (check_overflow)
    grid_row: int = 0              # stereo.launch:36-39 -- with grid_max_feature_num > 0 every cell of the
    grid_col: int = 0              # grid_row x grid_col grid keeps its grid_max_feature_num best corners
    grid_max_feature_num: int = 0  # (the nodelet's values: 3 x 4 cells, 4 per cell); 0 = global top max_features
    cross_check: bool = False      # keep a match only if it is mutual (query <-> train swapped gives it back)
    n_levels: int = 1              # ORB scale pyramid: 1 = single level; 8 with scale_factor 1.2 = Rublee et al.
    scale_factor: float = 1.2
    adaptive_fast: bool = True     # detect at a per-image threshold estimated from a sample of tiles and verified on the
                                   # device (vus_fast_threshold_estimate / _detect_adaptive / _detect_retry): the SAME
                                   # max_features keypoints as detection at fast_threshold, bit for bit, at a fraction
                                   # of the exact-score work.  Applies wherever a global top-K follows (single level, or
                                   # every pyramid level with its quota); grid bucketing needs every candidate: off there
    fast_sample_stride: int = 32   # every 32nd 128 x 24 tile is sampled (3 % of the image: ~90 survivors decide, 3.9 sigma
                                   # from a wrong answer at the 1.75x margin -- and a wrong answer only costs that image a retry)


def pyramid_layout(H: int, W: int, max_features: int, n_levels: int, scale_factor: float):
    """Level sizes and per-level keypoint quotas of the ORB pyramid (ORB paper sec. 6.1; the quota rule is the
    usual geometric split: level l gets a share proportional to scale_factor**-l, the last level the remainder)."""
    sizes = [(int(round(H / scale_factor ** l)), int(round(W / scale_factor ** l))) for l in range(n_levels)]
    if n_levels == 1:
        return sizes, [max_features]
    f = 1.0 / scale_factor
    nd = max_features * (1.0 - f) / (1.0 - f ** n_levels)
    quotas, total = [], 0
    for _ in range(n_levels - 1):
        q = int(round(nd))
        quotas.append(q)
        total += q
        nd *= f
    quotas.append(max(max_features - total, 0))
    return sizes, quotas


@dataclass
class Feature:
    """One entry of CameraMeasurement.features (batch.py:149-154)."""
    id: int
    u0: float
    v0: float
    u1: float
    v1: float


@dataclass
class CameraMeasurement:
    features: List[Feature] = field(default_factory=list)


class FrontendResult:
    """Device tensors produced by one process() call (views into the front-end's workspace)."""

    def __init__(self, fe, n_frames):
        F = n_frames
        self.n_frames = F
        self.W = fe.W
        self.H = fe.H
        self.kp_keys = fe.kp_keys[:2 * F]
        self.kp_count = fe.kp_count[:2 * F]
        self.desc = fe.desc[:2 * F]
        self.angle = fe.angle[:2 * F]
        self.cand_count = fe.cand_count[:2 * F]
        self.stereo_idx = fe.match_idx[:F]
        self.stereo_dist = fe.match_dist[:F]
        self.kp_level = fe.kp_level[:2 * F] if fe.kp_level is not None else None       # pyramid mode only
        self.kp_xy_q4 = fe.kp_xy_q4[:2 * F] if fe.kp_xy_q4 is not None else None       # level-0 position, 1/16 px
        self.track_idx = fe.match_idx[fe.max_frames:fe.max_frames + max(F - 1, 0)]
        self.track_dist = fe.match_dist[fe.max_frames:fe.max_frames + max(F - 1, 0)]

    def keypoints_xy_score(self):
        """Decode keys -> (x, y, score) int32 tensors [2F, K]; unused slots give score -1."""
        k = self.kp_keys.to(torch.int64) & 0xFFFFFFFF
        pos = k & 0xFFFFFF
        score = 255 - (k >> 24)
        valid = k != 0xFFFFFFFF
        y = torch.div(pos, self.W, rounding_mode="floor")
        x = pos - y * self.W
        score = torch.where(valid, score, torch.full_like(score, -1))
        return x.to(torch.int32), y.to(torch.int32), score.to(torch.int32)


class StereoOrbFrontend:
    """Batched stereo ORB: detect + describe both images of every frame, match left->right under
    the epipolar gate and left(t)->left(t+1) for tracks.  All buffers are allocated once here."""

    def __init__(self, H: int, W: int, max_frames: int, params: Optional[ImageProcessorParams] = None,
                 device: str = "cuda:0"):
        _lib.require_gpu()
        _lib.load()
        self.p = params or ImageProcessorParams()
        if self.p.cand_cap <= 0:
            import dataclasses
            self.p = dataclasses.replace(self.p, cand_cap=max(32768, (int(H) * int(W)) // 16))
        self.stage_hook = None      # optional callable(name), called on the launch stream after every stage of
                                    # process() (bench.py records HIP events there); None = no overhead
        if self.p.grid_max_feature_num > 0:
            assert self.p.grid_row >= 1 and self.p.grid_col >= 1 and self.p.n_levels == 1, \
                "grid bucketing needs grid_row, grid_col >= 1 and a single pyramid level"
        self.H, self.W, self.max_frames = int(H), int(W), int(max_frames)
        self.device = torch.device(device)
        n_img, K, cap = 2 * self.max_frames, self.p.max_features, self.p.cand_cap
        dev = self.device
        self.blur = torch.empty((n_img, H, W), dtype=torch.uint8, device=dev)
        self.cand_keys = torch.empty((n_img, cap), dtype=torch.int32, device=dev)
        self.cand_count = torch.zeros((n_img,), dtype=torch.int32, device=dev)
        self.kp_keys = torch.empty((n_img, K), dtype=torch.int32, device=dev)
        self.kp_count = torch.zeros((n_img,), dtype=torch.int32, device=dev)
        self.desc = torch.empty((n_img, K, 4), dtype=torch.int64, device=dev)
        self.angle = torch.empty((n_img, K), dtype=torch.uint8, device=dev)
        # the schedule of the orientation + rBRIEF launch: every image's keypoint slots grouped by 64 x 64-pixel cell
        # (vus_orient_order; a schedule only -- outputs are indexed by keypoint); images of more than 1024 cells go unordered
        self.kp_order = torch.empty((n_img, K), dtype=torch.int32, device=dev)
        # rows [0, F): stereo pairs; rows [max_frames, max_frames + F - 1): temporal pairs
        self.match_idx = torch.empty((2 * self.max_frames, K), dtype=torch.int32, device=dev)
        self.match_dist = torch.empty((2 * self.max_frames, K), dtype=torch.int32, device=dev)
        f = torch.arange(self.max_frames, dtype=torch.int32, device=dev)
        self.stereo_q, self.stereo_t = (2 * f).contiguous(), (2 * f + 1).contiguous()
        self.track_q, self.track_t = (2 * f).contiguous(), (2 * f + 2).contiguous()
        self.kp_level = self.kp_xy_q4 = None
        self.levels = None
        # sticky device-side maximum of the per-image candidate counts since the last check_overflow(): a
        # process(check=False) call that overflowed cand_cap is still reported by the next check
        self.cand_max_seen = torch.zeros((1,), dtype=torch.int32, device=dev)
        # per pyramid level the top-K is the level's quota: the same argument holds level by level
        self.adaptive = bool(self.p.adaptive_fast and self.p.grid_max_feature_num <= 0)
        if self.adaptive:
            self.fast_hist = torch.zeros((n_img, 256), dtype=torch.int32, device=dev)
            self.fast_thr = torch.zeros((n_img,), dtype=torch.int32, device=dev)      # the thresholds of the last process()
            self.fast_retry_list = torch.zeros((n_img,), dtype=torch.int32, device=dev)
            self.fast_retry_count = torch.zeros((1,), dtype=torch.int32, device=dev)  # images the check sent back to fast_threshold
        if self.p.cross_check:   # backward pairings (same row layout as match_idx)
            self.rev_idx = torch.empty((2 * self.max_frames, K), dtype=torch.int32, device=dev)
            self.rev_dist = torch.empty((2 * self.max_frames, K), dtype=torch.int32, device=dev)
        if self.p.n_levels > 1:
            # per-level workspaces: image (levels >= 1), smoothed image, keys, descriptors
            sizes, quotas = pyramid_layout(H, W, K, self.p.n_levels, self.p.scale_factor)
            assert all(h >= 2 * self.p.border + 8 and w >= 2 * self.p.border + 8 for h, w in sizes), \
                "pyramid level smaller than the detector border"
            self.kp_level = torch.zeros((n_img, K), dtype=torch.uint8, device=dev)
            self.kp_xy_q4 = torch.zeros((n_img, K, 2), dtype=torch.int32, device=dev)
            self.levels = []
            for l, ((h, w), q) in enumerate(zip(sizes, quotas)):
                q = max(q, 1)
                self.levels.append(dict(
                    H=h, W=w, quota=q,
                    img=None if l == 0 else torch.empty((n_img, h, w), dtype=torch.uint8, device=dev),
                    blur=self.blur if l == 0 else torch.empty((n_img, h, w), dtype=torch.uint8, device=dev),
                    keys=torch.empty((n_img, q), dtype=torch.int32, device=dev),
                    count=torch.zeros((n_img,), dtype=torch.int32, device=dev),
                    desc=torch.empty((n_img, q, 4), dtype=torch.int64, device=dev),
                    angle=torch.empty((n_img, q), dtype=torch.uint8, device=dev)))

    def process(self, images: torch.Tensor, check: bool = True) -> FrontendResult:
        """images: uint8 [F, 2, H, W] on the GPU (index 1: 0 = left / cam0, 1 = right / cam1).
        Asynchronous on the current stream unless `check` (which reads the overflow counters)."""
        assert images.dtype == torch.uint8 and images.is_cuda and images.is_contiguous()
        F = images.shape[0]
        assert images.shape[1:] == (2, self.H, self.W) and 1 <= F <= self.max_frames
        p, H, W, K = self.p, self.H, self.W, self.p.max_features
        n_img = 2 * F
        st = _lib.current_stream_ptr()
        ptr = _lib.ptr
        mark = self.stage_hook or (lambda name: None)
        mark("begin")
        if self.levels is None:
            self.cand_count[:n_img].zero_()
            if self.adaptive:
                _lib.call("vus_fast_threshold_estimate", ptr(images), n_img, H, W, W, p.fast_threshold, p.border, K,
                          p.fast_sample_stride, ptr(self.fast_hist), ptr(self.fast_thr), st)
                mark("fast_threshold")
                _lib.call("vus_fast_detect_adaptive", ptr(images), n_img, H, W, W, ptr(self.fast_thr), p.border,
                          ptr(self.blur), ptr(self.cand_keys), p.cand_cap, ptr(self.cand_count), st)
                _lib.call("vus_fast_detect_retry", ptr(images), n_img, H, W, W, p.fast_threshold, ptr(self.fast_thr), K,
                          p.border, ptr(self.cand_keys), p.cand_cap, ptr(self.cand_count), ptr(self.fast_retry_list),
                          ptr(self.fast_retry_count), st)
            else:
                _lib.call("vus_fast_detect", ptr(images), n_img, H, W, W, p.fast_threshold, p.border,
                          ptr(self.blur), ptr(self.cand_keys), p.cand_cap, ptr(self.cand_count), st)
            mark("fast_detect")
            if p.grid_max_feature_num > 0:
                _lib.call("vus_select_grid", ptr(self.cand_keys), ptr(self.cand_count), n_img, p.cand_cap, H, W,
                          p.grid_row, p.grid_col, p.grid_max_feature_num, K, ptr(self.kp_keys), ptr(self.kp_count), st)
            else:
                _lib.call("vus_select_topk", ptr(self.cand_keys), ptr(self.cand_count), n_img, p.cand_cap, K,
                          ptr(self.kp_keys), ptr(self.kp_count), st)
            mark("select_topk")
            self._orient(ptr(images), ptr(self.blur), n_img, H, W, ptr(self.kp_keys), ptr(self.kp_count), K, ptr(self.desc),
                         ptr(self.angle), st)
            mark("orient_rbrief")
        else:
            self._process_pyramid(images, n_img, st)
            mark("pyramid_detect_describe")
        _lib.call("vus_hamming_match", ptr(self.desc), ptr(self.kp_keys), ptr(self.kp_count), K, H, W,
                  ptr(self.stereo_q), ptr(self.stereo_t), F, p.stereo_threshold, p.min_disparity,
                  p.max_disparity, p.stereo_max_distance, ptr(self.match_idx), ptr(self.match_dist), st)
        mark("hamming_stereo")
        if F > 1:
            _lib.call("vus_hamming_match", ptr(self.desc), ptr(self.kp_keys), ptr(self.kp_count), K, H, W,
                      ptr(self.track_q), ptr(self.track_t), F - 1, -1, 0, 0, p.track_max_distance,
                      ptr(self.match_idx[self.max_frames:]), ptr(self.match_dist[self.max_frames:]), st)
        mark("hamming_track")
        if p.cross_check:
            # backward pairings: right -> left under the mirrored disparity gate, left(t+1) -> left(t)
            _lib.call("vus_hamming_match", ptr(self.desc), ptr(self.kp_keys), ptr(self.kp_count), K, H, W,
                      ptr(self.stereo_t), ptr(self.stereo_q), F, p.stereo_threshold, -p.max_disparity,
                      -p.min_disparity, p.stereo_max_distance, ptr(self.rev_idx), ptr(self.rev_dist), st)
            _lib.call("vus_cross_check", ptr(self.match_idx), ptr(self.rev_idx), F, K, ptr(self.match_idx), st)
            if F > 1:
                _lib.call("vus_hamming_match", ptr(self.desc), ptr(self.kp_keys), ptr(self.kp_count), K, H, W,
                          ptr(self.track_t), ptr(self.track_q), F - 1, -1, 0, 0, p.track_max_distance,
                          ptr(self.rev_idx[self.max_frames:]), ptr(self.rev_dist[self.max_frames:]), st)
                _lib.call("vus_cross_check", ptr(self.match_idx[self.max_frames:]), ptr(self.rev_idx[self.max_frames:]),
                          F - 1, K, ptr(self.match_idx[self.max_frames:]), st)
            mark("cross_check")
        if self.levels is None:   # two tiny device ops, asynchronous: keeps an overflow visible to a later check
            torch.maximum(self.cand_max_seen, self.cand_count[:n_img].max().reshape(1), out=self.cand_max_seen)
        mark("end")
        if check:
            self.check_overflow(n_img)
        return FrontendResult(self, F)

    def _process_pyramid(self, images, n_img, st):
        """Detect / select / describe on every level, then merge level-major into the per-image lists."""
        p, K, ptr = self.p, self.p.max_features, _lib.ptr
        self.kp_count[:n_img].zero_()
        self.kp_keys[:n_img].fill_(-1)          # VUS_KEY_INVALID
        self.pyr_cand_max = torch.zeros((1,), dtype=torch.int32, device=self.device)
        prev = images
        for l, L in enumerate(self.levels):
            h, w = L["H"], L["W"]
            if l > 0:
                P = self.levels[l - 1]
                _lib.call("vus_resize_bilinear", ptr(prev), n_img, P["H"], P["W"], P["W"], ptr(L["img"]), h, w, w, st)
                prev = L["img"]
            self.cand_count[:n_img].zero_()
            if self.adaptive:
                _lib.call("vus_fast_threshold_estimate", ptr(prev), n_img, h, w, w, p.fast_threshold, p.border, L["quota"],
                          p.fast_sample_stride, ptr(self.fast_hist), ptr(self.fast_thr), st)
                _lib.call("vus_fast_detect_adaptive", ptr(prev), n_img, h, w, w, ptr(self.fast_thr), p.border, ptr(L["blur"]),
                          ptr(self.cand_keys), p.cand_cap, ptr(self.cand_count), st)
                _lib.call("vus_fast_detect_retry", ptr(prev), n_img, h, w, w, p.fast_threshold, ptr(self.fast_thr), L["quota"],
                          p.border, ptr(self.cand_keys), p.cand_cap, ptr(self.cand_count), ptr(self.fast_retry_list),
                          ptr(self.fast_retry_count), st)
            else:
                _lib.call("vus_fast_detect", ptr(prev), n_img, h, w, w, p.fast_threshold, p.border, ptr(L["blur"]),
                          ptr(self.cand_keys), p.cand_cap, ptr(self.cand_count), st)
            torch.maximum(self.pyr_cand_max, self.cand_count[:n_img].max().reshape(1), out=self.pyr_cand_max)
            _lib.call("vus_select_topk", ptr(self.cand_keys), ptr(self.cand_count), n_img, p.cand_cap, L["quota"],
                      ptr(L["keys"]), ptr(L["count"]), st)
            self._orient(ptr(prev), ptr(L["blur"]), n_img, h, w, ptr(L["keys"]), ptr(L["count"]), L["quota"], ptr(L["desc"]),
                         ptr(L["angle"]), st)
            _lib.call("vus_pyramid_append", ptr(L["keys"]), ptr(L["count"]), ptr(L["desc"]), ptr(L["angle"]), n_img,
                      L["quota"], h, w, l, self.H, self.W, K, ptr(self.kp_keys), ptr(self.kp_count), ptr(self.desc),
                      ptr(self.angle), ptr(self.kp_level), ptr(self.kp_xy_q4), st)

    def _orient(self, img_p, blur_p, n_img, h, w, keys_p, count_p, k, desc_p, angle_p, st):
        """Orientation + rBRIEF of n_img images of h x w pixels with k keypoint slots each, in the cell-grouped order."""
        if ((h + 63) // 64) * ((w + 63) // 64) <= 1024:
            _lib.call("vus_orient_order", keys_p, count_p, n_img, k, h, w, _lib.ptr(self.kp_order), st)
            _lib.call("vus_orient_rbrief_ordered", img_p, blur_p, n_img, h, w, w, keys_p, count_p, k, _lib.ptr(self.kp_order),
                      desc_p, angle_p, st)
        else:
            _lib.call("vus_orient_rbrief", img_p, blur_p, n_img, h, w, w, keys_p, count_p, k, desc_p, angle_p, st)

    def check_overflow(self, n_img=None):
        """Raises if any image since the last check produced more FAST candidates than cand_cap (one device
        read: synchronises the stream)."""
        if self.levels is not None:
            worst = int(self.pyr_cand_max.item())
            if worst > self.p.cand_cap:
                raise _lib.VusError(f"FAST produced {worst} candidates in one pyramid level, more than cand_cap="
                                    f"{self.p.cand_cap}: raise ImageProcessorParams.cand_cap")
            return
        worst = int(self.cand_max_seen.item())
        self.cand_max_seen.zero_()
        if worst > self.p.cand_cap:
            raise _lib.VusError(f"FAST produced {worst} candidates in one image, more than cand_cap="
                                f"{self.p.cand_cap}: raise ImageProcessorParams.cand_cap")

    # ------------------------------------------------------------------------------------------
    def feature_tracks(self, res: FrontendResult):
        """GPU CameraMeasurement emitter (vus_track_ids): returns (ids int64 [F,K], feats f64 [F,K,4],
        n_ids).  ids[f,i] >= 0 marks a published feature of frame f with persistent id; feats holds its
        (u0, v0, u1, v1) in the normalised convention that batch.py:152-154 maps back to pixels."""
        F, K = res.n_frames, self.p.max_features
        dev = self.device
        ids = torch.empty((F, K), dtype=torch.int64, device=dev)
        feats = torch.empty((F, K, 4), dtype=torch.float64, device=dev)
        n_ids = torch.zeros((1,), dtype=torch.int64, device=dev)
        ptr = _lib.ptr
        _lib.call("vus_track_ids", ptr(res.stereo_idx), ptr(res.track_idx) if F > 1 else None, ptr(res.kp_keys),
                  ptr(res.kp_count), F, K, self.H, self.W, ptr(ids), ptr(feats), ptr(n_ids),
                  _lib.current_stream_ptr())
        return ids, feats, int(n_ids.item())

    def stereo_factors(self, ids: torch.Tensor, feats: torch.Tensor, n_ids: int, Rt: torch.Tensor, cam: torch.Tensor,
                       first_frame: int = 1):
        """What the reference's Python loops make of the CameraMeasurement stream, for all keyframes in two launches
        (vus_emit_stereo_factors): get_landmarks for every feature of every keyframe (batch.py:264-265, 144-176) and
        the landmark loop of batch_create (batch.py:295-305; keyframe 0 has none, hence first_frame = 1).
        ids / feats / n_ids: feature_tracks()'s output; Rt f64 [F,12]: zed_world_transform per keyframe; cam f64 [8]
        as for triangulate().  Returns a dict of device tensors: obs_frame int32 [n], obs_id int64 [n],
        obs_meas f64 [n,3] (uL, uR, v) in batch_create's factor order; lm_first int64 [n_ids] (-1: id not seen in
        keyframes >= first_frame), lm_point f64 [n_ids,3] (first-sighting world point = initial value of L(id))."""
        F, K = ids.shape
        dev = ids.device
        assert Rt.shape == (F, 12) and Rt.dtype == torch.float64 and Rt.is_contiguous() and cam.numel() == 8
        cap = F * K
        base = torch.empty((F + 1,), dtype=torch.int32, device=dev)
        count = torch.zeros((1,), dtype=torch.int32, device=dev)
        of = torch.empty((cap,), dtype=torch.int32, device=dev)
        oi = torch.empty((cap,), dtype=torch.int64, device=dev)
        om = torch.empty((cap, 3), dtype=torch.float64, device=dev)
        first = torch.empty((max(n_ids, 1),), dtype=torch.int64, device=dev)
        pt = torch.zeros((max(n_ids, 1), 3), dtype=torch.float64, device=dev)
        ptr = _lib.ptr
        _lib.call("vus_emit_stereo_factors", ptr(ids), ptr(feats), ptr(Rt), ptr(cam), F, K, int(first_frame), int(n_ids),
                  ptr(base), ptr(count), ptr(of), ptr(oi), ptr(om), ptr(first), ptr(pt), _lib.current_stream_ptr())
        n = int(count.item())
        return {"obs_frame": of[:n], "obs_id": oi[:n], "obs_meas": om[:n], "lm_first": first[:n_ids],
                "lm_point": pt[:n_ids]}

    def camera_measurements(self, res: FrontendResult) -> List[CameraMeasurement]:
        """Per frame, the published features as CameraMeasurement records (message shape of
        batch.py:149-154), built from feature_tracks()."""
        ids, feats, _ = self.feature_tracks(res)
        ids, feats = ids.cpu().numpy(), feats.cpu().numpy()
        out = []
        for f in range(res.n_frames):
            msg = CameraMeasurement()
            for i in (ids[f] >= 0).nonzero()[0].tolist():
                u0, v0, u1, v1 = feats[f, i]
                msg.features.append(Feature(id=int(ids[f, i]), u0=float(u0), v0=float(v0), u1=float(u1), v1=float(v1)))
            out.append(msg)
        return out


def triangulate(feat: torch.Tensor, cam: torch.Tensor, Rt: torch.Tensor) -> torch.Tensor:
    """get_landmarks (batch.py:144-176) on the GPU: feat [n,4] f64 (u0,v0,u1,v1), cam [8] f64
    (fx,fy,cx,cy,baseline,res_x,res_y,0), Rt [12] f64 -> [n,6] (X,Y,Z,uL,uR,v)."""
    _lib.require_gpu()
    assert feat.dtype == torch.float64 and feat.is_cuda and feat.is_contiguous() and feat.shape[1] == 4
    out = torch.empty((feat.shape[0], 6), dtype=torch.float64, device=feat.device)
    _lib.call("vus_triangulate", _lib.ptr(feat), feat.shape[0], _lib.ptr(cam), _lib.ptr(Rt), _lib.ptr(out),
              _lib.current_stream_ptr())
    return out
