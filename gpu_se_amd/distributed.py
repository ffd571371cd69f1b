"""Particle ensembles sharded over the GPUs of one node: one process per GPU (torch.distributed; backend "nccl" is
RCCL on ROCm, over xGMI), each rank owns a contiguous block of the particles.

The reference is single-GPU; this is the MI355X-native scale-out of the same filter:
predict and update are independent per particle; every shard resamples LOCALLY (an "island" particle filter)
and carries its accumulated log-weight; the only exchange is the posterior merge -- one small all-reduce:

    merge buffer [P][22] fp64, rank p fills row p with  { log A_p, S0, S1[5], S2[15] } / zeros elsewhere,
    all-reduce(SUM) makes every row visible on every rank (an all-gather expressed as the all-reduce the
    north-star names), then  estimate = sum_p a_p (S1_p / S0_p) / sum_p a_p,  a_p = exp(log A_p - max_q log A_q).

The collective payload is 22 P doubles (1.4 KB at P = 8): latency-bound, xGMI bandwidth is irrelevant; it is
issued asynchronously so that it overlaps the next predict.
"""
import numpy
import torch
import torch.distributed as dist

from gpu_se_amd import _lib
from gpu_se_amd.filter.particle import ParallelParticleFilter, moments_to_mean_cov

MERGE_WIDTH = 22


def shard_bounds(n_total, world, rank):
    """Contiguous split of n_total particles: the first (n_total % world) ranks hold one more."""
    base, extra = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, base + (1 if rank < extra else 0)


def pack_merge_row(log_island, s0, s1, s2, kernel_pivot, common_pivot):
    """Row of the merge buffer from one shard's device moments (taken about the shard's own pivot), shifted to
    the pivot all ranks share so that rows can be added:
        [log A, S0, sum w (x - p0)(5), upper(sum w (x - p0)(x - p0)^T)(15)]."""
    delta = numpy.asarray(kernel_pivot, dtype=numpy.float64) - common_pivot
    first = s1 + delta * s0
    second = s2 + numpy.outer(s1, delta) + numpy.outer(delta, s1) + s0 * numpy.outer(delta, delta)
    return numpy.concatenate([[log_island, s0], first, second[numpy.triu_indices(5)]])


def merge_rows(rows, pivot):
    """Combine the shards' rows (numpy [P][22]) into the global (mean, covariance).  Shard p enters with weight
    a_p = exp(log A_p - max log A); within a shard moments are normalised by its S0."""
    rows = numpy.asarray(rows, dtype=numpy.float64)
    live = (rows[:, 1] > 0) & numpy.isfinite(rows[:, 0])
    if not live.any():
        raise FloatingPointError("every shard has zero weight (all likelihoods underflowed)")
    la = numpy.where(live, rows[:, 0], -numpy.inf)
    a = numpy.exp(la - la[live].max())
    a = a / a.sum()
    s0 = numpy.where(live, rows[:, 1], 1.0)
    first = (a[:, None] * rows[:, 2:7] / s0[:, None]).sum(axis=0)           # E[x - pivot]
    iu = numpy.triu_indices(5)
    second = numpy.zeros((5, 5))
    second[iu] = (a[:, None] * rows[:, 7:22] / s0[:, None]).sum(axis=0)
    second = second + second.T - numpy.diag(numpy.diag(second))
    mean = pivot + first
    cov = second - numpy.outer(first, first)
    return mean, cov


class ShardedParticleFilter:
    """One big particle filter whose particles live on all ranks of ``group``.

    With group=None (or a single-rank group) this is exactly a ParallelParticleFilter.  Estimates are
    NORMALISED (mean and covariance of the posterior), unlike the reference's raw ``weights @ particles``,
    because shards cannot share one un-normalised weight scale without the exchange this design avoids.
    """

    def __init__(self, f, g, N_particles, x0, state_pdf, measurement_pdf, group=None, merge_device=None,
                 reduce_every=32):
        self.group = group
        self.world = dist.get_world_size(group) if group is not None else 1
        self.rank = dist.get_rank(group) if group is not None else 0
        self.N_particles = int(N_particles)
        lo, n_local = shard_bounds(self.N_particles, self.world, self.rank)
        self.local = ParallelParticleFilter(f, g, n_local, x0, state_pdf, measurement_pdf, element_offset=lo)
        dev = self.local._x.device
        self.local._island = torch.zeros((2, 1), dtype=torch.float64, device=dev)
        # a shard's prior weight is its share of the particles
        self.local._island[0, 0] = float(numpy.log(n_local / self.N_particles))   # [2][B = 1]
        self._pivot = numpy.asarray(x0.weights, dtype=numpy.float64) @ numpy.asarray(x0.means, dtype=numpy.float64) \
            / float(numpy.sum(x0.weights))
        # the collective runs where the backend wants its tensors: device for nccl/RCCL, host for gloo (CPU tests)
        backend = dist.get_backend(group) if group is not None else None
        self._merge_device = merge_device or (dev if backend in (None, "nccl") else torch.device("cpu"))
        self._wsum = torch.zeros(1, dtype=torch.float64, device=self._merge_device)
        self._pending = None
        # The weight-sum all-reduce is bookkeeping (island weights travel as log-weights and are only combined when
        # an estimate is asked for), so it is issued every reduce_every-th resample: a torch.distributed call costs
        # ~25 us of host time, more than a whole cycle of device work at N = 2^20.
        self.reduce_every = max(1, int(reduce_every))
        self._resamples = 0

    @classmethod
    def from_get_parts(cls, n_local, group=None):
        """The sim_base.get_parts wiring (prior = state-noise mixture shifted to X_ss) with n_local particles on
        every rank."""
        from gpu_se_amd import sim_base
        from gpu_se_amd.model import Bioreactor
        world = dist.get_world_size(group) if group is not None else 1
        xss = Bioreactor.find_SS(numpy.array([0.06, 0.2]), numpy.array([260 / 180, 640 / 24.6, 1000 / 116, 0, 0]))
        state_pdf, measurement_pdf = sim_base.get_noise()
        x0, _ = sim_base.get_noise()
        x0.means += numpy.asarray(xss[None, :], dtype=numpy.float32)
        return cls(Bioreactor.homeostatic_DEs, Bioreactor.static_outputs, n_local * world, x0, state_pdf,
                   measurement_pdf, group=group)

    # ------------------------------------------------------------------ the cycle (all local)
    def predict(self, u, dt):
        self.local.predict(u, dt)

    def update(self, u, z):
        """Local update.  The launch that carries the update also executes a pending resample-gather, which is where
        the shard's log-weight absorbs the weight sum; with more than one rank that sum is then all-reduced
        asynchronously (RCCL) -- nothing waits for it until wait_collectives() / an estimate."""
        resampled = self.local._pend_gather
        self.local.update(u, z)
        if resampled:
            self._resamples += 1
            if self._resamples % self.reduce_every == 0:
                self._reduce_weight_sum()

    def resample(self, r=None):
        """Local systematic resample (island filter); see update() for the bookkeeping."""
        self.local.resample(r=r)

    def _reduce_weight_sum(self):
        if self.group is not None:
            self.wait_collectives()
            # a buffer of its own: the next launch rewrites island[1] while the collective may still be in flight
            if self._merge_device.type == "cuda":
                self._wsum.copy_(self.local._island[1])
            else:
                self._wsum = self.local._island[1].cpu()
            self._pending = dist.all_reduce(self._wsum, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def wait_collectives(self):
        if self._pending is not None:
            self._pending.wait()
            self._pending = None

    # ------------------------------------------------------------------ posterior merge
    def _local_row(self):
        s0, s1, s2, kernel_pivot = self.local._raw_moments()      # synchronises this rank's stream
        log_island = float(self.local._island[0, 0].item())
        return pack_merge_row(log_island, s0, s1, s2, kernel_pivot, self._pivot)

    def _merged(self):
        row = self._local_row()
        if self.group is None:
            return merge_rows(row[None, :], self._pivot)
        self.wait_collectives()
        buf = torch.zeros((self.world, MERGE_WIDTH), dtype=torch.float64, device=self._merge_device)
        buf[self.rank] = torch.as_tensor(row, dtype=torch.float64, device=self._merge_device)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        return merge_rows(buf.cpu().numpy(), self._pivot)

    def point_estimate(self):
        """Posterior mean over ALL shards, host array (5,)."""
        return self._merged()[0]

    def point_covariance(self):
        """Largest singular value of the posterior covariance over all shards."""
        return numpy.linalg.svd(self._merged()[1], compute_uv=False)[0]
