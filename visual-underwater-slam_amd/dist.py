"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed (backend "nccl" =
RCCL over xGMI on MI355X; "gloo" for the CPU rehearsals in tests/).

The reference has no distributed code at all (single Python process); these are
the two places where the path shards:

  front-end  frames are independent -> contiguous chunks of frames per rank with a one-frame halo so
             that every left(t)->left(t+1) pair is matched by exactly one rank; no collective on the
             data path, one all_gather of the fixed-size per-frame match records at the end (8 B per
             keypoint slot, started asynchronously: it overlaps the next batch's kernels);
  BA         landmarks are independent given the poses -> contiguous landmark ranges ("landmark
             block-rows") per rank, poses replicated.  Each rank linearises and eliminates its own
             landmarks; the exchange step is ONE reduce per lambda trial of the reduced camera system
             (block band + right-hand side) to rank 0, which solves it and broadcasts the step dp and
             the solve's status word (half the bytes per link of an all-reduce, and
             every rank acts on bit-identical values); the back-substitution is local.  Error scalars
             are all-reduced for the accept test.
"""
from typing import List, Tuple

import torch
import torch.distributed as dist

from . import _lib


# ---------------------------------------------------------------------------------------------
# front-end: frame sharding
def shard_frames(n_frames: int, world: int, rank: int, halo: int = 1) -> Tuple[int, int, int]:
    """Contiguous frame range of `rank`: returns (first, n_owned, n_with_halo).  The owner of frame t
    also owns the temporal pair (t, t+1); it therefore reads `halo` extra frames past its range
    (none for the last non-empty shard)."""
    base, rem = divmod(n_frames, world)
    n_owned = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    n_halo = min(halo, max(0, n_frames - (first + n_owned))) if n_owned > 0 else 0
    return first, n_owned, n_owned + n_halo


class _TrackGather:
    """An all_gather of feature-track records in flight (gather_tracks_start) and what finish() needs to unpack it."""
    __slots__ = ("work", "bufs", "counts", "staged", "device", "packed")


def gather_tracks_start(stereo_idx: torch.Tensor, track_idx: torch.Tensor, kp_keys_left: torch.Tensor,
                        n_frames: int, world: int, rank: int) -> _TrackGather:
    """Start the all_gather of the per-frame feature-track records (BASELINE.json configs[3]) and return at once: the
    collective runs on RCCL's own stream, so the caller may launch the next step's kernels before gather_tracks_finish().
    Inputs are this rank's owned rows: stereo_idx [n_owned, K], track_idx [n_owned, K] (row of the last frame of the
    stream is all -1), kp_keys_left [n_owned, K]; they are copied into a staging buffer here and may be overwritten
    afterwards.  One collective of 8 B per keypoint slot: the two match indices (-1 .. K - 1, K <= 32767) share one int32
    word, the key is the second (16 KB per frame at K = 2000; with K > 32767 the three int32 records travel unpacked)."""
    K = stereo_idx.shape[1]
    h = _TrackGather()
    h.counts = [shard_frames(n_frames, world, r)[1] for r in range(world)]
    mx = max(h.counts)
    n_own = stereo_idx.shape[0]
    assert track_idx.shape[0] == n_own and kp_keys_left.shape[0] == n_own == h.counts[rank]
    h.packed = K <= 32767
    h.device = stereo_idx.device
    if h.packed:
        pad = torch.full((2, mx, K), -1, dtype=torch.int32, device=h.device)
        pad[0, :n_own] = (stereo_idx.to(torch.int32) & 0xFFFF) | (track_idx.to(torch.int32) << 16)
        pad[1, :n_own] = kp_keys_left.to(torch.int32)
    else:
        pad = torch.full((3, mx, K), -1, dtype=torch.int32, device=h.device)
        pad[0, :n_own], pad[1, :n_own], pad[2, :n_own] = stereo_idx, track_idx, kp_keys_left.to(torch.int32)
    h.work = None
    if world > 1:
        h.staged = pad.is_cuda and dist.get_backend() == "gloo"      # CPU rehearsal of the RCCL path
        src = pad.cpu() if h.staged else pad
        h.bufs = [torch.empty_like(src) for _ in range(world)]
        h.work = dist.all_gather(h.bufs, src, async_op=True)
    else:
        h.staged = False
        h.bufs = [pad]
    return h


def gather_tracks_finish(h: _TrackGather):
    """Wait for the collective of gather_tracks_start() and return the three full [n_frames, K] int32 tensors."""
    if h.work is not None:
        h.work.wait()
    bufs = [b.to(h.device) for b in h.bufs] if h.staged else h.bufs
    if h.packed:
        w0 = torch.cat([b[0, :c] for b, c in zip(bufs, h.counts)], 0)
        keys = torch.cat([b[1, :c] for b, c in zip(bufs, h.counts)], 0)
        stereo = (w0 << 16) >> 16            # sign-extending the low half restores -1
        track = w0 >> 16                     # arithmetic shift: the high half keeps its sign
        return stereo, track, keys
    return tuple(torch.cat([b[t, :c] for b, c in zip(bufs, h.counts)], 0) for t in range(3))


def gather_tracks(stereo_idx: torch.Tensor, track_idx: torch.Tensor, kp_keys_left: torch.Tensor,
                  n_frames: int, world: int, rank: int):
    """gather_tracks_start() + gather_tracks_finish(): the blocking form."""
    return gather_tracks_finish(gather_tracks_start(stereo_idx, track_idx, kp_keys_left, n_frames, world, rank))


def owned_track_records(res, n_owned: int):
    """The rows of a FrontendResult (computed on a shard WITH its halo frame) that this rank owns, in the shape
    gather_tracks() takes: stereo matches of the owned frames, the owned temporal pairs (the last frame of the
    whole stream has none: a row of -1), and the left keypoint keys."""
    K = res.stereo_idx.shape[1]
    track = torch.full((n_owned, K), -1, dtype=torch.int32, device=res.stereo_idx.device)
    n_pairs = min(n_owned, res.track_idx.shape[0])
    track[:n_pairs] = res.track_idx[:n_pairs]
    return res.stereo_idx[:n_owned], track, res.kp_keys[0:2 * n_owned:2]


# ---------------------------------------------------------------------------------------------
# BA: landmark block-row sharding
def shard_landmarks(obs_point: torch.Tensor, n_points: int, world: int) -> List[int]:
    """Split points 0..n_points-1 into `world` contiguous ranges of roughly equal OBSERVATION count.
    Returns the world+1 range boundaries."""
    counts = torch.bincount(obs_point.to(torch.int64), minlength=n_points)
    cum = torch.cumsum(counts, 0)
    total = int(cum[-1].item()) if n_points > 0 else 0
    bounds = [0]
    for r in range(1, world):
        target = total * r // world
        b = int(torch.searchsorted(cum, torch.tensor(target, device=cum.device), right=False).item())
        bounds.append(max(bounds[-1], min(b, n_points)))
    bounds.append(n_points)
    return bounds


def shard_observations(obs_pose, obs_point, meas, n_points: int, world: int, rank: int):
    """This rank's observations with landmark indices renumbered to its local range.
    Returns (obs_pose, obs_point_local, meas, lo, hi)."""
    bounds = shard_landmarks(obs_point, n_points, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    sel = (obs_point >= lo) & (obs_point < hi)
    return obs_pose[sel], obs_point[sel] - lo, meas[sel], lo, hi


def global_band(obs_pose, obs_point, n_points: int) -> int:
    """Half-bandwidth (in pose blocks) of the full reduced camera system: every rank must allocate
    the same band for the all-reduce."""
    if obs_pose.numel() == 0:
        return 0
    op, ol = obs_pose.to(torch.int64), obs_point.to(torch.int64)
    big = int(op.max().item()) + 1
    mn = torch.full((n_points,), big, dtype=torch.int64, device=op.device).scatter_reduce(0, ol, op, "amin")
    mx = torch.zeros((n_points,), dtype=torch.int64, device=op.device).scatter_reduce(0, ol, op, "amax")
    seen = mn < big
    return int((mx[seen] - mn[seen]).max().item()) if bool(seen.any()) else 0


def allreduce_sum(t: torch.Tensor):
    """In-place sum over ranks.  RCCL reduces device tensors directly; the gloo rehearsal stages
    through the host when the tensor lives on a GPU."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return t
    if t.is_cuda and dist.get_backend() == "gloo":
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)
    else:
        dist.all_reduce(t)
    return t


def _collective(t: torch.Tensor, fn):
    """Run an in-place collective on `t`; the gloo rehearsal stages GPU tensors through the host."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return t
    if t.is_cuda and dist.get_backend() == "gloo":
        h = t.cpu()
        fn(h)
        t.copy_(h)
    else:
        fn(t)
    return t


def broadcast_from_rank0(t: torch.Tensor):
    return _collective(t, lambda x: dist.broadcast(x, src=0))


def reduce_sum_to_rank0(t: torch.Tensor):
    """Sum over ranks delivered to rank 0 only (in place there; the other ranks' buffers are left in an unspecified
    state): half the bytes per link of an all-reduce (no all-gather phase)."""
    return _collective(t, lambda x: dist.reduce(x, dst=0))


class ShardedStereoBASolver:
    """Landmark-sharded LM: wraps a StereoBASolver built on this rank's observations (all poses, local
    landmarks, band forced to the global band) and inserts the collectives."""

    def __init__(self, obs_pose, obs_point, meas, n_poses, n_points, K, sigma, prior_pose=None, prior_T=None,
                 prior_sigmas=None, device="cuda:0"):
        from .ba import StereoBAProblem, StereoBASolver
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        obs_pose = torch.as_tensor(obs_pose).to(device)
        obs_point = torch.as_tensor(obs_point).to(device)
        meas = torch.as_tensor(meas).to(device)
        band = global_band(obs_pose, obs_point, n_points)
        op, ol, m, self.lo, self.hi = shard_observations(obs_pose, obs_point, meas, n_points, self.world, self.rank)
        if self.rank != 0:                      # priors are counted once, on rank 0
            prior_pose, prior_T, prior_sigmas = None, None, None
        self.problem = StereoBAProblem(op, ol, m, n_poses, self.hi - self.lo, K, sigma, prior_pose, prior_T,
                                       prior_sigmas, device=device, band=band)
        self.solver = _ShardSolver(self.problem, self.world)

    def optimize(self, poses, points, params=None):
        """poses: full [n_poses,12]; points: full [n_points,3].  Returns (poses, local points slice,
        report); use gather_points() for the full landmark array."""
        return self.solver.optimize(poses, points[self.lo:self.hi], params)

    def gather_points(self, local_points: torch.Tensor, n_points: int) -> torch.Tensor:
        full = torch.zeros((n_points, 3), dtype=torch.float64, device=local_points.device)
        full[self.lo:self.hi] = local_points
        return allreduce_sum(full)


def _make_shard_solver():
    from .ba import StereoBASolver

    class _Impl(StereoBASolver):
        def __init__(self, problem, world):
            super().__init__(problem)
            self.world = world
            self.rank = dist.get_rank() if dist.is_initialized() else 0

        def error(self, poses, points):
            super().error(poses, points)
            allreduce_sum(self.scal[0:1])
            return float(self.scal[0].item())

        def linearize(self, poses, points):
            super().linearize(poses, points)
            allreduce_sum(self.scal[0:1])          # linearised-at-zero error of the whole graph

        def schur(self, lam):
            super().schur(lam)
            # THE exchange step: the reduced camera system (block band + rhs) summed over the landmark shards.
            # Only rank 0 solves it, so a reduce to rank 0 is enough: half the bytes per link of an all-reduce.
            if self.world > 1:
                reduce_sum_to_rank0(self.Sband)
                reduce_sum_to_rank0(self.gs)
                if self.rank == 0:            # every rank added its own lambda I
                    _lib.call("vus_ba_add_diag", _lib.ptr(self.Sband), self.P.n_poses, self.P.band,
                              -(self.world - 1) * float(lam), _lib.current_stream_ptr())

        def band_solve(self):
            if self.world == 1:
                return super().band_solve()
            # Rank 0 solves; its step dp and its status word (< 0: bounded wait expired; > 0: non-positive pivot)
            # are THE step and THE status on every rank, so that all ranks accept / reject / raise together.
            # (A replicated solve would cost the same wall time, the all-gather half of an all-reduce on top, and
            # its cooperative back-substitution -- f64 atomics -- would let the ranks drift apart in the last bits.)
            if self.rank == 0:
                super().band_solve()
            broadcast_from_rank0(self.dp)
            broadcast_from_rank0(self.status)

        def eval_step(self, poses, points):
            super().eval_step(poses, points)
            allreduce_sum(self.scal[1:3])

    return _Impl


def _ShardSolver(problem, world):
    return _make_shard_solver()(problem, world)
