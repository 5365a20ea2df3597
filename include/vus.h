/* vus.h -- C ABI of the MI355X-native hot path of hvak/visual-underwater-slam.
 *
 * One shared library, libvus_hip.so (visual-underwater-slam_amd/csrc), exports every function
 * declared here.  All pointers are DEVICE pointers (HBM) unless the name ends in `_host`.
 * Every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream),
 * allocates nothing, and returns 0 on success or a negative VUS_E_* code (no exceptions cross the
 * boundary; vus_last_error() gives the text for the calling thread).  The caller owns all buffers.
 *
 * The CPU oracle (oracle/vus_oracle.c, test infrastructure only) exports the same signatures with
 * a `_cpu` suffix, host pointers and no stream argument.
 *
 * What each entry point replaces in the reference (paths relative to /root/reference):
 *   front-end  : the external `gtsam_vio/ImageProcessorNodelet` wired in launch/stereo.launch:33-55
 *                (parameters :37-47; fast_threshold=10 at :43, stereo_threshold=5 at :47) whose
 *                output batch.py consumes at batch.py:29,149-154,323.
 *   triangulate: AUV_ISAM.get_landmarks, batch.py:144-176.
 *   BA kernels : the arithmetic behind gtsam.GenericStereoFactor3D (batch.py:300-305) and
 *                gtsam.LevenbergMarquardtOptimizer(...).optimize() (batch.py:337).
 */
#ifndef VUS_H
#define VUS_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VUS_OK 0
#define VUS_E_INVALID (-1)   /* bad argument (null pointer, size out of range, ...) */
#define VUS_E_HIP (-2)       /* a HIP runtime call failed */
#define VUS_E_UNSUPPORTED (-3)

#define VUS_ABI_VERSION 1

/* keypoint key: ((255 - score) << 24) | (y * W + x).  Ascending key order == descending FAST
 * score, ties in raster order.  Requires H * W <= 2^24. */
#define VUS_KEY_POS_BITS 24
#define VUS_KEY_POS_MASK 0x00FFFFFFu
#define VUS_KEY_INVALID 0xFFFFFFFFu

int vus_abi_version(void);
const char* vus_last_error(void);
/* The target ID the library's code objects were built for, e.g. "gfx950:xnack-" (csrc/Makefile, OFFLOAD).  A device
 * whose target ID does not match runs none of them ("no kernel image is available"): the Python binding compares this
 * string with the device's gcnArchName before the first launch and says what to rebuild. */
const char* vus_build_target(void);

/* ------------------------------------------------------------------------------------------
 * Stereo ORB front-end.  Images are uint8, row-major [n_img, H, pitch] (pitch >= W bytes).
 * ---------------------------------------------------------------------------------------- */

/* FAST-9/16 score map (Rosten & Drummond): score(y,x) = largest t such that the pixel is still a
 * corner with strict comparisons  (= max over the 16 arcs of 9 of min(|d|) - 1), written only where
 * score >= thr, else 0; the 3-pixel frame is 0.  score_out: uint8 [n_img, H, W]. */
int vus_fast_score(const uint8_t* img, int n_img, int H, int W, int pitch, int thr,
                   uint8_t* score_out, void* stream);

/* 7x7 separable integer smoothing (weights VUS_BLUR_W, sum 256 per axis, replicate border,
 * round-half-up on the 16-bit product).  out: uint8 [n_img, H, W]. */
int vus_blur7(const uint8_t* img, int n_img, int H, int W, int pitch, uint8_t* out, void* stream);

/* Fused detector: FAST score + strict 3x3 non-max suppression + border filter, candidates appended
 * as keys (any order) to cand_keys[n_img, cand_cap]; cand_count[n_img] must be zero on entry and
 * receives the TRUE number of candidates (may exceed cand_cap: the surplus is dropped and the caller
 * must treat count > cand_cap as an error).  Also writes the smoothed image (as vus_blur7) to
 * blur_out if it is non-null. */
int vus_fast_detect(const uint8_t* img, int n_img, int H, int W, int pitch, int thr, int border,
                    uint8_t* blur_out, uint32_t* cand_keys, int cand_cap, int* cand_count,
                    void* stream);

/* ---- the adaptive detector (round 4; the nodelet's fast_threshold, launch/stereo.launch:43, stays the contract) ----
 * The max_kp best keypoints of an image (vus_select_topk) depend only on pixels whose score reaches s*, the max_kp-th
 * best score among the non-max-suppression survivors: a pixel below s* can neither be selected nor suppress one at or
 * above it.  Detection at any per-image threshold thr_img[n] <= s* therefore selects the SAME keypoints as detection at
 * thr, while far fewer pixels reach the exact score.  Three calls, all asynchronous, no host decision in between:
 *   vus_fast_threshold_estimate  hist [n_img,256] (scratch, zeroed here): the survivors of a SAMPLE of the 128 x 24-pixel
 *       tiles (raster order, every sample_stride-th starting at tile sample_stride / 2), counted by score; with
 *       f = max(thr, VUS_FAST_SAMPLE_FLOOR):  thr_img[n] = the largest t in (f, 254] with
 *       count(score >= t) * tiles * VUS_FAST_MARGIN_DEN  >=  max_kp * sampled tiles * VUS_FAST_MARGIN_NUM,  else thr.
 *       The sample itself is detected at f: survivors with a score >= f are the same as at thr (a neighbour below f
 *       cannot suppress them), so the bins from f on are what a detection at thr yields; the bins below f stay zero.
 *       (An image whose estimate would lie in (thr, f] is detected at thr: slower, same keypoints.)
 *   vus_fast_detect_adaptive     vus_fast_detect with the per-image thresholds (device array).  With blur_out and
 *       cand_cap >= 512 the list of an image is filled as eight sub-lists (tile t into sub-list t mod 8, counters in the
 *       row itself) and compacted by a second launch before the call returns its work to the stream: cand_count[n] must
 *       be ZERO on entry (it is overwritten with the total), the result is the usual compact list in unspecified order;
 *       a sub-list that outgrows cand_cap / 8 - 1 entries is reported as an overflow of the list (cand_count[n] >
 *       cand_cap, the unwritten tail = VUS_KEY_INVALID).
 *   vus_fast_detect_retry        the check: every image with thr_img[n] > thr and cand_count[n] < max_kp (the estimate
 *       was too high: s* may lie below thr_img[n]) is detected again at thr -- cand_count[n] reset, candidates
 *       rewritten; blur_out of the adaptive pass stays valid.  retry_list [n_img] scratch, retry_count[0] = how many.
 * After the three calls vus_select_topk gives exactly what it gives after vus_fast_detect(thr): bit-identical keys. */
#ifndef VUS_FAST_MARGIN_NUM      /* overridable for A/B builds only: library and oracle must be built with the same pair */
#define VUS_FAST_MARGIN_NUM 7    /* 1.75: on the configs[1] stream 0 of 1000 images fail the check (1.5: 18, 1.25: 130) and the */
#define VUS_FAST_MARGIN_DEN 4    /* detection is at its fastest (2.96 ms per 1000 stereo frames; 1.5: 2.98, 2.0: 3.03, 1.25: 3.50) */
#endif
#define VUS_FAST_SAMPLE_FLOOR 40
#define VUS_FAST_TILE_W 128
#define VUS_FAST_TILE_H 24
int vus_fast_threshold_estimate(const uint8_t* img, int n_img, int H, int W, int pitch, int thr, int border, int max_kp,
                                int sample_stride, int* hist, int* thr_img, void* stream);
int vus_fast_detect_adaptive(const uint8_t* img, int n_img, int H, int W, int pitch, const int* thr_img, int border,
                             uint8_t* blur_out, uint32_t* cand_keys, int cand_cap, int* cand_count, void* stream);
int vus_fast_detect_retry(const uint8_t* img, int n_img, int H, int W, int pitch, int thr, const int* thr_img, int max_kp,
                          int border, uint32_t* cand_keys, int cand_cap, int* cand_count, int* retry_list, int* retry_count,
                          void* stream);

/* Keep the max_kp smallest keys of each image, sorted ascending.  kp_keys: [n_img, max_kp]
 * (unused tail = VUS_KEY_INVALID); kp_count[n_img] = min(max_kp, min(cand_count, cand_cap)). */
int vus_select_topk(const uint32_t* cand_keys, const int* cand_count, int n_img, int cand_cap,
                    int max_kp, uint32_t* kp_keys, int* kp_count, void* stream);

/* Grid-bucketed selection, the nodelet's grid_row / grid_col / grid_max_feature_num parameters
 * (launch/stereo.launch:36-39): the image is cut into grid_row x grid_col cells (cell of a pixel:
 * cy = y * grid_row / H, cx = x * grid_col / W, integer division) and every cell keeps its per_cell smallest
 * keys.  Output: cells in row-major order, each cell's keys ascending, packed from slot 0;
 * kp_count = number kept (<= min(max_kp, grid_row * grid_col * per_cell)); unused tail = VUS_KEY_INVALID. */
int vus_select_grid(const uint32_t* cand_keys, const int* cand_count, int n_img, int cand_cap, int H, int W,
                    int grid_row, int grid_col, int per_cell, int max_kp, uint32_t* kp_keys, int* kp_count,
                    void* stream);

/* Orientation (intensity centroid over the radius-15 disc of `img`, quantised to 30 bins with
 * integer arithmetic) and 256-bit rotated-BRIEF descriptor sampled from `blur`.
 * desc_out: uint64 [n_img, max_kp, 4] (bit b of word w = test 64*w + b, set when I(p0) < I(p1));
 * angle_out: uint8 [n_img, max_kp] (bin index). */
int vus_orient_rbrief(const uint8_t* img, const uint8_t* blur, int n_img, int H, int W, int pitch,
                      const uint32_t* kp_keys, const int* kp_count, int max_kp,
                      uint64_t* desc_out, uint8_t* angle_out, void* stream);

/* The same outputs with the keypoints of an image SCHEDULED in a spatially coherent order (the patch gathers of
 * neighbouring keypoints share cache lines while they are in flight: ~9 % off the kernel's time at 2000 keypoints per
 * 1280 x 720 image).  vus_orient_order: order int32 [n_img, max_kp] = a permutation of every image's slots that groups the
 * keypoints [0, kp_count) by 64 x 64-pixel cell, cells in raster order (WHICH permutation inside a cell is unspecified;
 * slots from kp_count on map to themselves); images of more than 1024 cells are rejected (call vus_orient_rbrief).
 * vus_orient_rbrief_ordered: slot s is served with keypoint order[s]; desc_out / angle_out are indexed by keypoint as in
 * vus_orient_rbrief -- bit-identical outputs for ANY permutation that fixes the unused slots. */
int vus_orient_order(const uint32_t* kp_keys, const int* kp_count, int n_img, int max_kp, int H, int W, int* order,
                     void* stream);
int vus_orient_rbrief_ordered(const uint8_t* img, const uint8_t* blur, int n_img, int H, int W, int pitch,
                              const uint32_t* kp_keys, const int* kp_count, int max_kp, const int* order,
                              uint64_t* desc_out, uint8_t* angle_out, void* stream);

/* ---- optional ORB scale pyramid (ImageProcessorParams.n_levels > 1; Rublee et al. 2011, sec. 6.1:
 * 8 levels, scale 1.2).  Level l is resized from level l-1; detection, top-K (per-level quota) and
 * description run on every level with the calls above (H, W, pitch of that level); the per-level
 * keypoints are then appended, level-major, into one list per image that the matchers consume.
 *
 * vus_resize_bilinear: bilinear down/up-sampling with pixel-centre alignment and 11-bit weights, all
 * integer:  for destination column dx:  num = (2 dx + 1) Ws - Wd, den = 2 Wd, ix = floor(num / den),
 * wx = round-half-up(2048 (num - ix den) / den); source columns clamp(ix), clamp(ix + 1) (rows alike);
 * out = (sum of the four products w * p + 2^21) >> 22.  dst: uint8 [n_img, Hd, pitch_d]. */
int vus_resize_bilinear(const uint8_t* src, int n_img, int Hs, int Ws, int pitch_s, uint8_t* dst,
                        int Hd, int Wd, int pitch_d, void* stream);

/* Append level `level`'s keypoints (first min(lvl_count, lvl_max_kp) entries of every image, in
 * order) to the merged per-image lists at slot kp_count[img], up to max_kp; kp_count is advanced.
 * Before the first level the caller zeroes kp_count and fills kp_keys with VUS_KEY_INVALID.
 * Positions are mapped to level 0 with pixel-centre alignment, in 1/16 pixel:
 *   xq = floor(((2 x + 1) * 8 * W0 + Wl / 2) / Wl) - 8   (yq alike; level 0: 16 x),
 * merged key = score byte of the level key | (y0 * W0 + x0) with x0 = clamp((xq + 8) >> 4, 0, W0-1),
 * so matchers, vus_track_ids and vus_triangulate see level-0 pixel positions.
 *   kp_level uint8 [n_img, max_kp] (KeyPoint.octave), kp_xy_q4 int32 [n_img, max_kp, 2] = (xq, yq);
 *   either may be NULL. */
int vus_pyramid_append(const uint32_t* lvl_keys, const int* lvl_count, const uint64_t* lvl_desc,
                       const uint8_t* lvl_angle, int n_img, int lvl_max_kp, int Hl, int Wl, int level,
                       int H0, int W0, int max_kp, uint32_t* kp_keys, int* kp_count, uint64_t* desc,
                       uint8_t* angle, uint8_t* kp_level, int32_t* kp_xy_q4, void* stream);

/* Brute-force Hamming matcher over n_pairs (query set, train set) pairs; the query set of pair p
 * is image q_index[p], the train set is image t_index[p] (indices into the [n_img, max_kp] arrays).  For every query keypoint: the train keypoint of smallest Hamming distance
 * among those passing the gates  |yq - yt| <= max_dy  (max_dy < 0: no gate)  and
 * min_disp <= xq - xt <= max_disp  (only when max_dy >= 0); ties -> lowest train index.
 * No gated candidate -> idx -1, dist 512; best distance > max_dist -> idx -1, dist = that distance.
 * idx_out / dist_out: int32 [n_pairs, max_kp]. */
int vus_hamming_match(const uint64_t* desc, const uint32_t* kp_keys, const int* kp_count,
                      int max_kp, int H, int W, const int* q_index, const int* t_index, int n_pairs,
                      int max_dy, int min_disp, int max_disp, int max_dist,
                      int32_t* idx_out, int32_t* dist_out, void* stream);

/* Optional mutual-nearest-neighbour filter ("cross-check"): a match i -> j of the forward pairing survives
 * only if the backward pairing (query and train sets swapped) matches j -> i.
 * idx_fwd, idx_bwd, idx_out: int32 [n_pairs, max_kp]; idx_out may alias idx_fwd. */
int vus_cross_check(const int32_t* idx_fwd, const int32_t* idx_bwd, int n_pairs, int max_kp, int32_t* idx_out,
                    void* stream);

/* CameraMeasurement emitter (what batch.py:149-154 reads from the nodelet's message): persistent
 * feature ids propagated along the left(t)->left(t+1) matches, and the features' normalised image
 * coordinates.  Frame f's left image is image 2f, its right image 2f+1 (as in vus_hamming_match).
 *   stereo_idx [n_frames, max_kp]   left->right match of frame f (-1: none)
 *   track_idx  [n_frames-1, max_kp] left(f)->left(f+1) match (-1: none); may be NULL when n_frames == 1
 * A left keypoint j of frame f inherits the id of the lowest-index keypoint of frame f-1 whose track
 * match is j (if that keypoint carries an id); stereo-matched keypoints without an id get fresh ids
 * in index order.  Keypoints without a stereo match keep an inherited id for later frames but are
 * not published.
 *   ids_out  int64 [n_frames, max_kp]   id of every PUBLISHED feature, -1 otherwise
 *   feat_out f64   [n_frames, max_kp, 4] (u0, v0, u1, v1) = 2*x/W - 1, 2*y/H - 1 for both cameras
 *                  (zeros where not published)
 *   n_ids_out int64 [1]                 number of ids issued */
int vus_track_ids(const int32_t* stereo_idx, const int32_t* track_idx, const uint32_t* kp_keys,
                  const int* kp_count, int n_frames, int max_kp, int H, int W, int64_t* ids_out,
                  double* feat_out, int64_t* n_ids_out, void* stream);

/* batch_update's get_landmarks call per keyframe + batch_create's landmark loop (batch.py:264-265, 295-305) for a whole
 * sequence of keyframes at once: what the Python loops of the reference turn the CameraMeasurement stream into.
 *   ids  int64 [n_frames, max_kp]    published feature ids of every keyframe (-1: none), as vus_track_ids emits them
 *   feat f64   [n_frames, max_kp, 4] their (u0, v0, u1, v1)
 *   Rt   f64   [n_frames, 12]        zed_world_transform of every keyframe (row-major R then t; batch.py:45-48,166)
 *   cam  f64   [8]                   as for vus_triangulate
 * Keyframes < first_frame emit nothing (batch.py:280-305 gives keyframe 0 no landmark loop: first_frame = 1).  For
 * every feature (f >= first_frame, slot i, id >= 0), in frame-major / slot-ascending order -- the order in which
 * batch_create pushes its GenericStereoFactor3D factors --
 *   obs_frame int32 [n], obs_id int64 [n], obs_meas f64 [n,3] = StereoPoint2(uL, uR, v)   (batch.py:300-304)
 * with n returned in count[0] (n <= n_frames * max_kp: the capacity the caller allocates), and per id < n_ids
 *   lm_first int64 [n_ids] = f * max_kp + i of its FIRST sighting in those keyframes (-1: never seen there)
 *   lm_point f64   [n_ids,3] = that sighting's world point, i.e. what batch.py:297-298 inserts as L(id)'s initial value
 *                              (untouched for ids never seen).
 * frame_base int32 [n_frames + 1] is scratch (receives the exclusive prefix sum of the per-keyframe feature counts). */
int vus_emit_stereo_factors(const int64_t* ids, const double* feat, const double* Rt, const double* cam, int n_frames,
                            int max_kp, int first_frame, long long n_ids, int* frame_base, int* count,
                            int* obs_frame, int64_t* obs_id, double* obs_meas, int64_t* lm_first, double* lm_point,
                            void* stream);

/* Unwhitened residual of every emitted stereo factor at the INITIAL estimate -- GenericStereoFactor3D's
 * h(X(f), L(id)) - measurement with X(f) = Rt[obs_frame], L(id) = lm_point[obs_id], K = (fx, fy, skew, cx, cy, b):
 *   resid f64 [n,3];  a point at or behind the camera (z <= 0, gtsam's cheirality case) gives +infinity in all three.
 * EXTENSION (not in the reference, which has no outlier handling of its own: the nodelet's RANSAC precedes it): lets the
 * caller gate gross mismatches of the brute-force matcher before they enter the graph (sequence.py, gate_px). */
int vus_stereo_initial_residuals(const double* Rt, const double* K, const double* lm_point, const int* obs_frame,
                                 const int64_t* obs_id, const double* obs_meas, int n, double* resid, void* stream);

/* get_landmarks of batch.py:144-176, elementwise over n features (fp64):
 *   feat [n,4] = (u0, v0, u1, v1) normalised image coordinates of the CameraMeasurement message,
 *   cam  [8]   = (fx, fy, cx, cy, baseline, resolution_x, resolution_y, unused),
 *   Rt   [12]  = row-major 3x3 rotation followed by translation (the TF of batch.py:45-48),
 *   out  [n,6] = (X, Y, Z world point, uL, uR, v). */
int vus_triangulate(const double* feat, int n, const double* cam, const double* Rt, double* out,
                    void* stream);

/* ------------------------------------------------------------------------------------------
 * Stereo bundle adjustment: what gtsam.LevenbergMarquardtOptimizer(graph, values, params)
 * .optimize() (batch.py:337) spends its time on, for a graph of GenericStereoFactor3D factors
 * (batch.py:300-305) plus PriorFactorPose3 gauge priors (batch.py:281).  All fp64.
 *
 * Variables: poses [n_poses,12] = row-major 3x3 R then t (camera-to-world, gtsam.Pose3);
 *            points [n_points,3].
 * Pose tangent: xi = (omega, v), right perturbation T * Exp(xi)  (gtsam Pose3 convention).
 * Observations are stored twice-indexed:
 *   "L-order": sorted by (point, pose); point_ptr is the CSR row pointer over it;
 *   "P-order": sorted by (pose, point); pose_ptr is the CSR row pointer over it.
 * Per-observation Jacobian products W, Y live in L-order (round 4; rounds 1-3 kept them in P-order): a landmark's rows
 * are consecutive, poses ascending -- its rows of an 8-pose tile are one run.
 * Empty inputs: n_points == 0 and n_obs == 0 are valid (a graph of pose priors only); the arrays and the
 * buffers of an empty dimension may then be NULL.  pose_ptr is always required (n_poses + 1 entries).
 * ---------------------------------------------------------------------------------------- */
typedef struct vus_ba_problem {
  int n_poses, n_points, n_obs, n_priors;
  const double* K;         /* [6] fx, fy, skew (ignored, as gtsam's StereoCamera does), cx, cy, baseline */
  double inv_sigma;        /* 1/sigma of the isotropic stereo noise model (batch.py:118: sigma = 10) */
  const double* meas;      /* [n_obs,3] (uL, uR, v), L-order */
  const int* obs_pose;     /* [n_obs] L-order */
  const int* obs_point;    /* [n_obs] L-order */
  const int* point_ptr;    /* [n_points+1] */
  const int* obs_ppos;     /* [n_obs] L-order index -> P-order slot */
  const int* pose_ptr;     /* [n_poses+1] */
  const int* pobs_lidx;    /* [n_obs] P-order slot -> L-order index */
  const int* prior_pose;   /* [n_priors] pose index of each PriorFactorPose3 */
  const double* prior_T;   /* [n_priors,12] prior mean */
  const double* prior_w;   /* [n_priors,6] 1/sigma per tangent coordinate (rot xyz, trans xyz) */
  int pose_stride;         /* camera-side node layout: pose i is node pose_stride*i (0 or 1: poses only;
                              2: node 2i = X(i), node 2i+1 = V(i) padded to 6, for graphs with vus_nav_factors).
                              Everything indexed "by node" (Sband rows, gs, dp) uses 6 doubles per node. */
} vus_ba_problem;

/* Camera-side "navigation" factors of the reference graph: velocity
 * variables V(i) [n_poses,3] next to every pose, ONE shared bias B(0) [6] = (acc, gyro) (batch.py:274),
 * and the factors that tie them:
 *   ImuFactor(X(i), V(i), X(j), V(j), B(0), pim)       batch.py:237-239,289-293
 *   DVL velocity factor on (V(i), X(i)): e = R_i m - v_i   batch.py:196-250 (correct Jacobians)
 *   PriorFactorVector on V(i)                            batch.py:282
 * pim rows are the 148-double records of the preintegration (layout: oracle/vus_oracle_nav.c PIM_*,
 * mirrored in visual-underwater-slam_amd/gtsam/imu.py); imu_W is the 9x9 whitening matrix L^-1 of
 * cov = L L^T, tangent order (theta, p, v). */
typedef struct vus_nav_factors {
  int n_imu;
  const int* imu_i;          /* [n_imu] pose index of the earlier state */
  const int* imu_j;          /* [n_imu] pose index of the later state */
  const double* imu_pim;     /* [n_imu,148] */
  const double* imu_W;       /* [n_imu,81] */
  double gravity[3];         /* n_gravity of PreintegrationParams (MakeSharedU(g): 0,0,-g) */
  int n_dvl;
  const int* dvl_pose;       /* [n_dvl] */
  const double* dvl_meas;    /* [n_dvl,3] body-frame velocity */
  const double* dvl_w;       /* [n_dvl] 1/sigma (isotropic, batch.py:98) */
  int n_vprior;
  const int* vprior_idx;     /* [n_vprior] */
  const double* vprior_v;    /* [n_vprior,3] */
  const double* vprior_w;    /* [n_vprior,3] 1/sigma */
} vus_nav_factors;

/* Linearise every factor at (poses, points):
 *   W   [n_obs,18]  H1^T H2 (6x3 row-major) per observation, L-order, whitened
 *   V   [n_points,6] sum H2^T H2, upper triangle (xx,xy,xz,yy,yz,zz)
 *   gl  [n_points,3] sum H2^T r
 *   Hpp [n_poses,36] sum H1^T H1 + prior information;  gp [n_poses,6] sum H1^T r + prior part
 *   err [1]          0.5 * sum |whitened residual|^2 over stereo factors and priors
 *   work             scratch of at least vus_ba_work_doubles(P) doubles (deterministic reductions)
 * Stereo residual/Jacobians follow gtsam::GenericStereoFactor / StereoCamera::project2
 * (cheirality z <= 0: residual 2*fx on all three rows, zero Jacobians). */
int vus_ba_linearize(const vus_ba_problem* P, const double* poses, const double* points,
                     double* W, double* V, double* gl, double* Hpp, double* gp, double* err,
                     double* work, void* stream);

/* ---- the landmark elimination on the matrix cores: block-sparse S -= Y W^T by 8 x 8-pose TILE PAIRS (round 4) ----
 * The sum over landmarks IS the K dimension of a GEMM: for the tile pair (I, K) and every landmark j seen from both
 * tiles, A_j = [Y_ij] (48 x 3, zero rows for poses of I that do not see j) and B_j = [W_kj] (48 x 3) are laid side by
 * side along K and contracted with v_mfma_f64_16x16x4 into the 48 x 48 tile of S.  A W row is then fetched once per
 * pose TILE that co-observes its landmark instead of once per co-observing pose (the per-pair kernel of rounds 1-3 fetched
 * 8.3 GB of W rows per launch at configs[2]; this one 2.8 GB, in runs of up to 8 consecutive rows).
 *
 * vus_ba_tiles: unit u = (tile row I = u / (Dt + 1), tile distance d = u % (Dt + 1)), Dt = ceil(band / 8), covers the
 * pose blocks (i, k) with i / 8 = I, k / 8 = I - d.  entries[unit_ptr[u] .. unit_ptr[u + 1]) are its landmarks in
 * ascending order, four ints each: a = first L-order row of the landmark in tile I, b = the same for tile I - d,
 * j = the landmark, mask = (8-bit mask of the poses of tile I that see it) | (the same for tile I - d) << 8 -- the rows of
 * a tile are consecutive in L-order, row t belonging to the pose of the t-th set bit.  order[]: the units, largest first
 * (a schedule, not a result).  Built once per graph, two launches + a sort (csrc/pack.hip):
 *   vus_ba_tiles_count   lm_entries[j] = entries landmark j contributes = m (m + 1) / 2 for m pose tiles that see it
 *   (host)               lm_base = exclusive prefix sums (vus_exclusive_scan_i32); the total sizes `entries`
 *   vus_ba_tiles_fill    unit_ptr [n_tiles * (Dt + 1) + 1], entries [n_entries, 4], order [n_tiles * (Dt + 1)];
 *                        work: vus_ba_tiles_work_bytes(n_entries) bytes
 * The reference has no counterpart: GTSAM finds its elimination structure inside optimize() (batch.py:337). */
typedef struct vus_ba_tiles {
  int band;              /* in poses; >= the widest keyframe span of a landmark */
  int n_tiles;           /* ceil(n_poses / 8) */
  int n_units;           /* n_tiles * (ceil(band / 8) + 1) */
  int n_entries;
  const int* unit_ptr;   /* [n_units + 1] */
  const int* entries;    /* [n_entries, 4] (16-byte aligned) */
  const int* order;      /* [n_units] */
} vus_ba_tiles;
int vus_ba_tiles_count(const vus_ba_problem* P, int* lm_entries, void* stream);
long long vus_ba_tiles_work_bytes(int n_entries);
int vus_ba_tiles_fill(const vus_ba_problem* P, int band, const int* lm_base, int n_entries, int* unit_ptr, int* entries,
                      int* order, void* work, long long work_bytes, void* stream);
/* Damped landmark elimination for one lambda (lambda*I damping, gtsam diagonalDamping=false):
 *   Vinv [n_points,6] = (V + lambda I)^-1 (upper triangle);  Y [n_obs,18] = W Vinv (L-order), OPTIONAL output
 *   (NULL: not written; the kernel forms the rows it needs on the fly);
 *   S band [n_nodes, band_nodes + 1, 36], entry (i, s) = the 6 x 6 block (i, i - s), = Hpp + lambda I - sum_j Y W^T: every
 *   stored block is written (none accumulated into), diagonal blocks whole;  gs [n_nodes,6] = gp - sum Y gl.
 * band_nodes: the half-bandwidth of Sband's storage in NODES (>= pose_stride * T->band).  counter: one int of device
 * scratch (the unit queue).  Replaces what GTSAM's elimination does inside optimize(), batch.py:337. */
int vus_ba_schur(const vus_ba_problem* P, const vus_ba_tiles* T, double lambda, const double* W, const double* V,
                 const double* gl, const double* Hpp, const double* gp, double* Vinv, double* Y, double* Sband,
                 int band_nodes, double* gs, int* counter, void* stream);

/* Sband(i,i) += value * I for every pose.  Used by the landmark-sharded multi-GPU solve: after the
 * all-reduce of the per-rank bands the pose damping lambda*I has been added once per rank, and
 * value = -(n_ranks-1)*lambda restores a single copy. */
int vus_ba_add_diag(double* Sband, int n_poses, int band, double value, void* stream);

/* ---- graph packing on the device (csrc/pack.hip): what batch_create leaves as lists of factor objects
 * (batch.py:295-305) becomes the arrays of vus_ba_problem without a host pass ----
 * `work`: vus_pack_work_bytes(n) bytes of device scratch for n keys / observations.
 *
 * vus_keys_to_indices: idx_out[i] = rank of keys[i] among the DISTINCT keys (ascending), uniq_out[0 .. n_unique) = those
 *   keys, n_unique[0] their number (device).  Keys are gtsam keys (non-negative: chr << 56 | index).
 * vus_lookup_keys: idx_out[i] = position of queries[i] in sorted_keys [m] (ascending, distinct), -1 if absent;
 *   first_miss[0] = the smallest i that missed, 0x7F7F7F7F if none (device).
 * vus_ba_pack_observations: rows (obs_pose[i], obs_point[i], meas[i]) in any order -> the L-order arrays (sorted by
 *   point, then pose) meas_L / obs_pose_L / obs_point_L / point_ptr [n_points+1] / obs_ppos, the P-order arrays
 *   pose_ptr [n_poses+1] / pobs_lidx, and perm (L-order row -> input row).  flags[0] (device): bit 0 = two factors
 *   between the same pose and landmark, bit 1 = an index out of range.  band[0] (device, may be NULL): the widest
 *   keyframe span of a landmark = the half-bandwidth, in pose blocks, of the reduced camera system.
 * vus_exclusive_scan_i32: out[i] = in[0] + .. + in[i-1] for i <= n as 32-bit offsets, total[0] = the whole sum in
 *   64 bits (so that the caller can refuse a sum the offsets cannot hold); n up to a few hundred thousand. */
long long vus_pack_work_bytes(int n);
int vus_keys_to_indices(const int64_t* keys, int n, int* idx_out, int64_t* uniq_out, int* n_unique, void* work,
                        long long work_bytes, void* stream);
int vus_lookup_keys(const int64_t* sorted_keys, int m, const int64_t* queries, int n, int* idx_out, int* first_miss,
                    void* stream);
int vus_ba_pack_observations(const int* obs_pose, const int* obs_point, const double* meas, int n_obs, int n_poses,
                             int n_points, double* meas_L, int* obs_pose_L, int* obs_point_L, int* point_ptr,
                             int* obs_ppos, int* pose_ptr, int* pobs_lidx, int* perm, int* flags, int* band, void* work,
                             long long work_bytes, void* stream);
int vus_exclusive_scan_i32(const int* in, int n, int* out, long long* total, void* stream);

/* Solve S dp = -gs by block-band Cholesky; n_poses counts NODES.  Sband is overwritten by the factor L in
 * the solver's own layout: the 6x6 blocks left of the 8-node diagonal panels hold their transposes, and (bands of
 * 7 nodes and more) the diagonal panels themselves hold the INVERSE of their 48x48 factor block.
 * status[0] = 0 ok, k+1 = non-positive pivot met in scalar column k (dp is then undefined), -1 = the
 * cooperative back-substitution gave up waiting (never observed; the waits are bounded so that a scheduling
 * anomaly cannot hang the GPU).  The sweep's flags live in the unused slots of block row 0 of Sband.
 * The back-substitution adds its partial products with f64 atomics: two solves of the same system agree to ~1e-13
 * relative, not bitwise. */
int vus_ba_band_solve(double* Sband, int n_poses, int band, const double* gs, double* dp,
                      int* status, void* stream);

/* Same factorisation with n_rhs (<= 8) right-hand sides solved in place: rhs [n_rhs, 6*n_nodes]
 * (no sign change).  Used for the bias border of graphs with inertial factors. */
int vus_ba_band_solve_multi(double* Sband, int n_nodes, int band, double* rhs, int n_rhs, int* status,
                            void* stream);

/* Two-sided variants of the two solves above: poses are eliminated from both ends of the band in the same launches
 * (top-down in place, bottom-up on a pose-reversed copy held in `work`), the dense system of the >= band middle
 * poses is solved in between, which halves the chain of dependent panel steps that bounds the solve.  Same result to
 * round-off; Sband is overwritten (contents unspecified).  `work`: vus_ba_band_solve_work_doubles() doubles; that
 * function returns 0 when the system is too short to split (n_nodes < band + 16): the calls then fall back to the
 * one-sided solve and ignore `work` (which must still be non-NULL).  status: as above; a non-positive pivot is
 * reported as SOME scalar column + 1 > 0 (the elimination order differs from the one-sided solve's). */
long long vus_ba_band_solve_work_doubles(int n_nodes, int band, int n_rhs);
int vus_ba_band_solve_split(double* Sband, int n_poses, int band, const double* gs, double* dp, int* status,
                            double* work, void* stream);
int vus_ba_band_solve_multi_split(double* Sband, int n_nodes, int band, double* rhs, int n_rhs, int* status,
                                  double* work, void* stream);

/* Thread safety of the two-sided solves: re-entrant.  Each call forks onto an auxiliary stream and a fork/join event
 * pair that belong to (device, caller's stream); two host threads solving on different streams of one device share
 * nothing, two calls on the SAME stream are ordered by that stream as usual.  Whatever the call returns, the caller's
 * stream has been made to wait for every launch the call issued on the auxiliary stream.
 *
 * Tuning knobs of the band solve (tests and A/B timing; the defaults are right for production).  Their initial values
 * are read from the environment ONCE, when the library is loaded (VUS_BAND_MODE, VUS_CB_MAX_WG); afterwards only
 * vus_ba_set_tuning() changes them, process-wide.  Host-only calls, no device work.
 *   VUS_TUNE_BAND_MODE   how a panel step is issued: -1 automatic; 0 one fused launch per panel; 1 a TRSM + SYRK launch
 *                        pair per panel (two systems share every launch); 2 the same pair per system on two streams;
 *                        3 the persistent window kernel (one launch for the whole chain; where it does not apply --
 *                        bands under 16 poses or over ~240, the one-sided entry points without `work` -- the
 *                        automatic choice among 0..2 is taken).
 *   VUS_TUNE_CB_MAX_WG   cap on the cooperating workgroups of the back-substitution (0 = from the occupancy query). */
#define VUS_TUNE_BAND_MODE 0
#define VUS_TUNE_CB_MAX_WG 1
#define VUS_TUNE_LAST_BAND_MODE 2   /* read-only (vus_ba_get_tuning): how the most recent factorisation was issued, 0..3 */
#define VUS_TUNE_WIN_FAULT 3        /* tests only: 1 makes one workgroup of the window kernel exit at once, as a workgroup
                                       that never became resident would; the launch then ends in VUS_STATUS_WINDOW_EXPIRED */
/* Negative values of a band solve's status word (every inter-workgroup wait is bounded; none of them is a hang):
 *   VUS_STATUS_WAIT_EXPIRED    a wait of the cooperative back-substitution expired;
 *   VUS_STATUS_WINDOW_EXPIRED  a wait of the persistent window kernel (mode 3) expired: its flag protocol needs every
 *                              workgroup of the launch resident at once, which other work on the device (another
 *                              process or stream holding CUs) can prevent.  The band is spoilt; a caller redoes the
 *                              Schur step and solves with VUS_TUNE_BAND_MODE 2 (launch pairs: no residency demand) --
 *                              ba.py does exactly that, once, and keeps the window kernel off afterwards. */
#define VUS_STATUS_WAIT_EXPIRED (-1)
#define VUS_STATUS_WINDOW_EXPIRED (-3)
int vus_ba_set_tuning(int knob, int value);
int vus_ba_get_tuning(int knob);

/* ---- navigation factors on the camera side (graphs with vus_nav_factors, pose_stride = 2) ----
 * vus_nav_linearize: residuals/Jacobians of every ImuFactor / DVL factor / velocity prior at
 * (poses, vels, bias), accumulated (camera-side blocks with f64 atomics -- at most four addends per block, so two runs
 * agree to ~1e-16 relative, not bitwise; the bias block and gradient in a fixed order) into
 *   Snav [n_nodes,4,36]  blocks (node, node-s), s = 0..3   (undamped)
 *   Scb  [n_nodes,36]    coupling of every node with the shared bias (node rows x bias columns)
 *   Sbb  [36], gnav [6*n_nodes], gb [6]
 *   err [1] = 0.5 * sum |whitened residual|^2 of these factors
 * work: at least vus_nav_work_doubles(N) doubles. */
int vus_nav_linearize(const vus_nav_factors* N, int n_poses, const double* poses, const double* vels,
                      const double* bias, double* Snav, double* Scb, double* Sbb, double* gnav, double* gb,
                      double* err, double* work, void* stream);
long long vus_nav_work_doubles(const vus_nav_factors* N);

/* Per lambda: Sband += Snav on the 4 innermost block diagonals, velocity nodes get lambda on their 3 real
 * coordinates and 1 on the 3 padding coordinates, gs += gnav; rhs [7, 6*n_nodes] is filled with
 * column 0 = -gs and columns 1..6 = the six columns of Scb. */
int vus_nav_assemble(int n_nodes, int band, double lambda, const double* Snav, const double* Scb,
                     const double* gnav, double* Sband, double* gs, double* rhs, void* stream);

/* After vus_ba_band_solve_multi(rhs, 7): eliminate the bias border,
 *   (Sbb + lambda I - Scb^T Z) db = -gb - Scb^T z0,   dc = z0 - Z db,
 * dc [6*n_nodes] (node layout), db [6]. */
int vus_nav_border_solve(int n_nodes, const double* rhs, const double* Scb, const double* Sbb, const double* gb,
                         double lambda, double* dc, double* db, void* stream);

/* new_vels = vels + dv (from dc), new_bias = bias + db;  out[0] = linearised error of the navigation factors
 * at the step (Jacobians at the OLD values), out[1] = their error at the new values (new_poses from vus_ba_eval_step). */
int vus_nav_eval_step(const vus_nav_factors* N, int n_poses, const double* poses, const double* vels,
                      const double* bias, const double* dc, const double* db, const double* new_poses,
                      double* new_vels, double* new_bias, double* out, double* work, void* stream);

/* HOST function (no GPU work, host pointers): on-manifold IMU preintegration, what gtsam::PreintegratedImuMeasurements does
 * inside integrateMeasurement (reference call sites batch.py:91, 290, 293; Forster et al., TRO 2017).  `pim` is the
 * 148-double record above, in and out: the n samples {acc[3], gyro[3], dt} are integrated ON TOP of the state it holds
 * (all zero except dR = identity and the bias estimate = a fresh interval).  acc_cov / gyro_cov / int_cov: the 3x3
 * continuous-time covariances of PreintegrationParams.  whiten (may be null): W = L^-1, cov = L L^T, row-major 9x9, the
 * factor's whitening matrix (imu_W); VUS_E_INVALID if the covariance is not positive definite. */
int vus_imu_preintegrate(double* pim, const double* samples, int n, const double* acc_cov, const double* gyro_cov,
                         const double* int_cov, double* whiten);

/* err[0] = error of the navigation factors at (poses, vels, bias). */
int vus_nav_error(const vus_nav_factors* N, int n_poses, const double* poses, const double* vels,
                  const double* bias, double* err, double* work, void* stream);

/* dl [n_points,3] = -Vinv (gl + sum_a W_a^T dp[pose_a]). */
int vus_ba_backsub(const vus_ba_problem* P, const double* W, const double* Vinv, const double* gl,
                   const double* dp, double* dl, void* stream);

/* Evaluate a step: new_poses = poses (+) dp (Pose3 retract = T * Expmap(xi)), new_points = points + dl,
 * out[0] = linearised error at the step 0.5*sum|r + J d|^2 (J at the OLD values, undamped),
 * out[1] = nonlinear error at the new values (stereo factors + priors). */
int vus_ba_eval_step(const vus_ba_problem* P, const double* poses, const double* points,
                     const double* dp, const double* dl, double* new_poses, double* new_points,
                     double* out, double* work, void* stream);

/* err[0] = 0.5 * sum |whitened residual|^2 at (poses, points): NonlinearFactorGraph.error(). */
int vus_ba_error(const vus_ba_problem* P, const double* poses, const double* points, double* err,
                 double* work, void* stream);

/* Number of doubles the `work` scratch of the calls above must hold for problem P (host-side, no launch). */
long long vus_ba_work_doubles(const vus_ba_problem* P);

#ifdef __cplusplus
}
#endif
#endif /* VUS_H */
