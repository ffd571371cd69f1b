"""Particle ensembles sharded over the GPUs of one node: one process per GPU (torch.distributed; backend "nccl" is
RCCL on ROCm, over xGMI), each rank owns a contiguous block of the particles.

The reference is single-GPU; this is the MI355X-native scale-out of the same filter:
predict and update are independent per particle; every shard resamples LOCALLY (an "island" particle filter)
and carries its accumulated log-weight; the only exchange is the posterior merge -- one small all-reduce:

    merge buffer [P][22] fp64, rank p fills row p with  { log A_p, S0, S1[5], S2[15] } / zeros elsewhere,
    all-reduce(SUM) makes every row visible on every rank (an all-gather expressed as the all-reduce the
    north-star names), then  estimate = sum_p A_p S1_p / sum_p A_p S0_p:  particle i of shard p carries the global
    weight A_p w_i, where A_p absorbs the weight sums of the shard's past resamples and w_i is its live local weight.

The collective payload is 22 P doubles (1.4 KB at P = 8): latency-bound, xGMI bandwidth is irrelevant; it is
issued asynchronously so that it overlaps the next predict.

``resample(exact=True)`` removes the island approximation: one systematic resample of the WHOLE
ensemble, bit-identical to a single GPU holding the concatenation of the shards.  It costs an all-gather of the tile sums
(8 B per 1024 particles) and one all-to-all of the offspring whose ancestor lives on another rank (20 B each, at most
n per rank, over xGMI); see ``exact_resample_plan`` for the bookkeeping every rank derives on its own.
"""
import numpy
import torch
import torch.distributed as dist

from gpu_se_amd import _lib
from gpu_se_amd.filter.particle import ParallelParticleFilter, moments_to_mean_cov

MERGE_WIDTH = 22


def shard_bounds(n_total, world, rank, align=1):
    """Contiguous split of n_total particles: the first (n_total % world) ranks hold one more.  ``align`` > 1 (and at
    least ``align`` particles per rank): whole blocks of ``align`` particles are dealt out instead and the last rank takes
    the ragged remainder -- every shard but the last then holds a multiple of ``align``, which is what exact resampling
    needs of the tile boundaries (align = GSE_TILE)."""
    n_total, world, align = int(n_total), int(world), int(align)
    if align > 1 and n_total >= world * align:
        units = -(-n_total // align)
        base, extra = divmod(units, world)
        lo_u = rank * base + min(rank, extra)
        hi_u = lo_u + base + (1 if rank < extra else 0)
        return lo_u * align, min(hi_u * align, n_total) - lo_u * align
    base, extra = divmod(n_total, world)
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
    """Combine the shards' rows (numpy [P][22]) into the global (mean, covariance).  Shard p enters with its total
    weight A_p S0_p (its accumulated island weight times the sum of its live local weights -- S0_p = 1 right after
    a local resample, anything between update() and resample()); within a shard moments are normalised by S0_p."""
    rows = numpy.asarray(rows, dtype=numpy.float64)
    a, live = shard_shares(rows[:, 0], rows[:, 1])
    s0 = numpy.where(live, rows[:, 1], 1.0)
    first = (a[:, None] * rows[:, 2:7] / s0[:, None]).sum(axis=0)           # E[x - pivot]
    iu = numpy.triu_indices(5)
    second = numpy.zeros((5, 5))
    second[iu] = (a[:, None] * rows[:, 7:22] / s0[:, None]).sum(axis=0)
    second = second + second.T - numpy.diag(numpy.diag(second))
    mean = pivot + first
    cov = second - numpy.outer(first, first)
    return mean, cov


def shard_shares(log_island, s0):
    """Normalised total weight of every shard, a_p = A_p S0_p / sum_q A_q S0_q, formed in the log domain; shards with
    no weight (S0 = 0 or a log-weight that is not finite) get 0.  Returns (a, live mask)."""
    log_island = numpy.asarray(log_island, dtype=numpy.float64)
    s0 = numpy.asarray(s0, dtype=numpy.float64)
    live = (s0 > 0) & numpy.isfinite(log_island) & numpy.isfinite(s0)
    if not live.any():
        raise FloatingPointError("every shard has zero weight (all likelihoods underflowed)")
    la = numpy.where(live, log_island + numpy.log(numpy.where(live, s0, 1.0)), -numpy.inf)
    a = numpy.exp(la - la[live].max())
    return a / a.sum(), live


def island_ess(log_island):
    """Effective number of islands (sum a)^2 / sum a^2 of the shards' accumulated weights: P when the shards weigh the
    same, 1 when one shard carries the posterior."""
    a, _ = shard_shares(log_island, numpy.ones(len(log_island)))
    return 1.0 / float(numpy.sum(a * a))


def output_boundaries(end_keys, n_total, r):
    """J[q] = number of outputs j in [0, n_total) whose threshold u_j = (j + r) / n_total does not exceed the normalised
    cumulative weight at the end of shard q - 1; J[0] = 0.  Outputs [J[q], J[q + 1]) are the offspring of shard q's
    particles (ancestor of j = first k with c[k] >= u_j, particle.py:96-100, and c ends shard q at end_keys[q]).
    The comparison uses the very expression the kernels use for u_j, so the split is exact, not approximate."""
    end_keys = numpy.asarray(end_keys, dtype=numpy.float64)
    n = int(n_total)
    nd = float(n)
    u = lambda j: (numpy.asarray(j, dtype=numpy.float64) + r) / nd
    out = [0]
    for e in end_keys:
        j = int(min(max(numpy.floor(e * nd - r), -1), n - 1))         # candidate for the last output with u_j <= e
        while j >= 0 and u(j) > e:
            j -= 1
        while j + 1 < n and u(j + 1) <= e:
            j += 1
        out.append(max(j + 1, out[-1]))
    out[-1] = n                                                       # c ends at 1 >= every u_j
    return out


def exact_resample_plan(J, spans, rank):
    """Split sizes of the all-to-all for ``rank``: what it sends to every rank (offspring of ITS particles, outputs
    [J[rank], J[rank + 1]), cut by the ranks' output ranges ``spans`` = [(lo, count)]), and what it receives from every
    rank (the pieces of its own output range)."""
    overlap = lambda a0, a1, b0, b1: max(0, min(a1, b1) - max(a0, b0))
    send = [overlap(J[rank], J[rank + 1], lo, lo + cnt) for lo, cnt in spans]
    lo, cnt = spans[rank]
    recv = [overlap(J[q], J[q + 1], lo, lo + cnt) for q in range(len(spans))]
    return send, recv


class TorchComm:
    """The collectives ShardedParticleFilter needs, on a torch.distributed group (backend "nccl" is RCCL on ROCm).  Anything
    with these five methods and three attributes can stand in for it -- the tests drive eight shards from eight THREADS of one
    process through such a stand-in (tests/gpu_util.ThreadComm: a GPU box admits six processes on its card, BASELINE config 4
    has eight ranks)."""

    def __init__(self, group):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)

    def all_reduce(self, tensor, async_op=False):
        """In-place sum over the ranks; with async_op a work object whose wait() orders the current stream behind it."""
        return dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def all_gather(self, parts, tensor):
        dist.all_gather(parts, tensor, group=self.group)

    def all_to_all_single(self, dst, src, recv, send):
        dist.all_to_all_single(dst, src, output_split_sizes=recv, input_split_sizes=send, group=self.group)


class ShardedParticleFilter:
    """One big particle filter whose particles live on all ranks of ``group``.

    With group=None (or a single-rank group) this is exactly a ParallelParticleFilter.  Estimates are
    NORMALISED (mean and covariance of the posterior), unlike the reference's raw ``weights @ particles``,
    because shards cannot share one un-normalised weight scale without the exchange this design avoids.
    """

    collectives_issued = 0                                  # all-reduces / all-gathers / all-to-alls issued by this object
    merge_cov_size = False                                  # merges also leave the covariance size in estimate_device[5] (a serial
                                                            # Jacobi chain on the device: off unless a device consumer wants it)
    _shares_host = _side_stream = _shares_event = None      # (device groups only: see __init__)
    _shares_recorded = False

    def __init__(self, f, g, N_particles, x0, state_pdf, measurement_pdf, group=None, merge_device=None,
                 reduce_every=32, even_split=False):
        # (``group``: a torch.distributed group, or any object with TorchComm's interface)
        self._comm = None if group is None else (group if hasattr(group, "all_to_all_single") else TorchComm(group))
        self.group = group
        self.world = self._comm.world if group is not None else 1
        self.rank = self._comm.rank if group is not None else 0
        self.N_particles = int(N_particles)
        # shards on tile boundaries whenever every rank can have a tile of its own: exact resampling (and with it the
        # island policy below) needs that of every shard but the last
        # (even_split: the plain contiguous split, equal shards up to one particle -- what from_get_parts(n_local) promises;
        #  exact resampling then needs n_local to be a multiple of the tile)
        self._align = 1 if even_split else _lib.GSE_TILE
        self._spans = [shard_bounds(self.N_particles, self.world, q, self._align) for q in range(self.world)]
        self._exact_capable = not any(cnt % _lib.GSE_TILE for _, cnt in self._spans[:-1])
        lo, n_local = self._spans[self.rank]
        self.local = ParallelParticleFilter(f, g, n_local, x0, state_pdf, measurement_pdf, element_offset=lo)
        dev = self.local._x.device
        self.local._island = torch.zeros((2, 1), dtype=torch.float64, device=dev)
        # a shard's prior weight is its share of the particles
        self.local._island[0, 0] = float(numpy.log(n_local / self.N_particles))   # [2][B = 1]
        self._pivot = numpy.asarray(x0.weights, dtype=numpy.float64) @ numpy.asarray(x0.means, dtype=numpy.float64) \
            / float(numpy.sum(x0.weights))
        # the collective runs where the backend wants its tensors: device for nccl/RCCL, host for gloo (CPU tests)
        backend = self._comm.backend if group is not None else None
        self._merge_device = merge_device or (dev if backend in (None, "nccl") else torch.device("cpu"))
        self._pending = None
        # The weight-sum all-reduce is bookkeeping (island weights travel as log-weights and are only combined when
        # an estimate is asked for), so it is issued every reduce_every-th resample: a torch.distributed call costs
        # ~25 us of host time, more than a whole cycle of device work at N = 2^20.
        self.reduce_every = max(1, int(reduce_every))
        self._resamples = 0
        self._exchange_due = False                          # the next update() issues the exchange of the shards' log-weights
        # Island policy: the shards' accumulated log-weights are exchanged with that same all-reduce (P doubles, slot p
        # written by rank p); when the effective number of islands falls below ess_fraction * P the NEXT resample is an
        # exact one over the whole ensemble, which levels the shards (every rank sees the same numbers one cycle later,
        # so every rank takes the same decision).  ess_fraction = 0 switches the policy off.
        self.ess_fraction = 0.5
        if not self._exact_capable:
            # the shards do not sit on tile boundaries (fewer than one tile per rank, or an even split of a count that is no
            # multiple of the tile): no exact resample and no policy that would call for one in the middle of a run
            # (resample(exact=True) still raises)
            import warnings
            warnings.warn("ShardedParticleFilter: the shards of %d particles on %d ranks do not end on tile boundaries (%d "
                          "particles); the island policy is off (ess_fraction = 0)" % (self.N_particles, self.world, _lib.GSE_TILE))
            self.ess_fraction = 0.0
        self._shares = torch.zeros(self.world, dtype=torch.float64, device=self._merge_device)
        self._shares_cycle = None                           # resample count at which the pending exchange was issued
        # On the device the exchanged log-weights come back to the host BESIDE the cycles (round 4): a side stream waits for the
        # collective and copies them to pinned memory; the policy reads them `shares_lag` resamples after the exchange, when
        # the copy has long landed -- every rank at the same resample count, so every rank decides alike -- and the stream
        # the cycles run on never waits for the collective nor the host for the stream.  (Reading them one cycle later through
        # .cpu() drained the launch queue at every exchange: ~60 us without work per rank, four times in the driver's K = 20.)
        self._shares_host = self._side_stream = self._shares_event = None
        if self._merge_device.type == "cuda" and self.group is not None:
            self._shares_host = torch.zeros(self.world, dtype=torch.float64).pin_memory()
            self._side_stream = torch.cuda.Stream(device=self._merge_device)
            self._shares_event = torch.cuda.Event()
        self._force_exact = False
        self._islands_apart = False                         # an island resample has run since the last exact one
        self.exact_resamples = 0
        self.last_island_ess = float(self.world)
        import gpu_se_amd
        # host generator that is the SAME on every rank (exact resampling draws one r for the whole ensemble)
        self._common_rng = numpy.random.default_rng(gpu_se_amd._seed_state["seed"] ^ 0xE4AC7)
        self.last_exact = None
        self._ends_index = None                             # device indices of the shard ends in the tile prefix (built once)
        self._merge_buf = torch.zeros((self.world, MERGE_WIDTH), dtype=torch.float64, device=self._merge_device)
        self._dev_rows = self._dev_moments = self._dev_status = self._pivot_c = self._dev_est = None   # device merge (merge_begin)
        self._merge_work, self._merge_pending = None, False
        self.collectives_issued = 0                         # all-reduces / all-gathers / all-to-alls this object has issued

    @classmethod
    def from_get_parts(cls, n_local, group=None):
        """The sim_base.get_parts wiring (prior = state-noise mixture shifted to X_ss) with n_local particles on
        every rank."""
        from gpu_se_amd import sim_base
        from gpu_se_amd.model import Bioreactor
        world = (group.world if hasattr(group, "all_to_all_single") else dist.get_world_size(group)) if group is not None else 1
        xss = Bioreactor.find_SS(numpy.array([0.06, 0.2]), numpy.array([260 / 180, 640 / 24.6, 1000 / 116, 0, 0]))
        state_pdf, measurement_pdf = sim_base.get_noise()
        x0, _ = sim_base.get_noise()
        x0.means += numpy.asarray(xss[None, :], dtype=numpy.float32)
        return cls(Bioreactor.homeostatic_DEs, Bioreactor.static_outputs, n_local * world, x0, state_pdf,
                   measurement_pdf, group=group, even_split=True)

    # ------------------------------------------------------------------ the cycle (all local)
    def predict(self, u, dt):
        self.local.predict(u, dt)

    def update(self, u, z):
        """Local update.  The launch that carries the update also executes a pending resample-gather, which is where
        the shard's log-weight absorbs the weight sum; with more than one rank that sum is then all-reduced
        asynchronously (RCCL) -- nothing waits for it until wait_collectives() / an estimate."""
        self.local.update(u, z)
        if self._exchange_due:
            # (the resample was counted when it was called -- round 5: whoever ran its copy, this launch or a flush before it,
            #  the shard's log-weight has absorbed the weight sum by now)
            self._exchange_due = False
            if self.group is not None and self.world > 1:
                self._exchange_shares()

    def resample(self, r=None, exact=False):
        """Local systematic resample (island filter); see update() for the bookkeeping.  ``exact=True``: one systematic
        resample of the whole ensemble across the ranks (every rank must call it, with the same ``r`` if one is given)."""
        if self.group is not None and self.world > 1:
            self._consider_shares()
            if exact or self._force_exact:
                self._force_exact = False
                return self._exact_resample(r)
        self.local.resample(r=r)
        self._resamples += 1
        self._islands_apart = True
        self._exchange_due = self._resamples % self.reduce_every == 0

    def run_cycles(self, us, zs, dt, rs=None):
        """K island cycles predict -> update -> resample of this shard enqueued by ONE call (ParallelParticleFilter.run_cycles ->
        gse_pf_cycles, which hands the shard's bookkeeping to every launch): the launches of K rounds of the three methods.
        The island policy looks at the exchanged log-weights before the batch (an exact resample of the whole ensemble first,
        if the shards have drifted apart) and an exchange that falls due inside the batch is issued by the next update()."""
        K = len(us)
        if K == 0:
            return
        multi = self.group is not None and self.world > 1
        if multi:
            self._consider_shares()
            if self._force_exact:
                self._force_exact = False
                self._exact_resample(None)
        self.local.run_cycles(us, zs, dt, rs)
        before = self._resamples
        self._resamples += K                                   # (counted where resample() counts them: at the call)
        self._islands_apart = True
        # an exchange that falls due inside the batch is issued by the next update(), like one that falls due on its last resample
        # (the copy of that resample -- where the shard's log-weight absorbs the weight sum -- rides that update's launch)
        self._exchange_due = self._exchange_due or (self._resamples // self.reduce_every != before // self.reduce_every)

    def _consider_shares(self):
        """One cycle after the shards' log-weights were exchanged: read them (the only synchronisation of the island
        mode, once per reduce_every cycles) and decide, identically on every rank, whether the islands have drifted
        apart far enough for an exact resample."""
        if self._shares_cycle is None or self._resamples < self._shares_cycle + self.shares_lag:
            return
        self._shares_cycle = None
        if self._shares_event is not None:
            self._shares_event.synchronize()                # (the copy landed cycles ago: nothing to wait for)
            shares = self._shares_host.numpy().copy()
        else:
            self.wait_collectives()
            shares = self._shares.cpu().numpy()
        self.last_island_ess = island_ess(shares)
        if self.ess_fraction > 0 and self.last_island_ess < self.ess_fraction * self.world:
            self._force_exact = True

    def _exact_resample(self, r):
        import ctypes
        f = self.local
        lib = _lib.load()
        f._flush()
        dev = f._x.device
        n, P = f.N_particles, self.world
        spans = self._spans
        if not self._exact_capable:
            raise ValueError("exact resampling needs every shard but the last to hold a multiple of %d particles" % _lib.GSE_TILE)
        tiles = [-(-cnt // _lib.GSE_TILE) for _, cnt in spans]
        T_total, tile_offset = sum(tiles), sum(tiles[:self.rank])
        stream = torch.cuda.current_stream().cuda_stream
        if r is None:                                       # one number for the whole ensemble, the same on every rank
            r = float(self._common_rng.random())
        # 0. A particle's weight in the whole ensemble is A_p w_i (A_p = exp(island[0]) of its shard).  After island
        #    resamples the A_p differ, so the local weights are brought to one scale first: w_i * exp(log A_p - max_q
        #    log A_q), fp32 like the weights themselves.  (Exact-only sequences on equal shards: every A_p is the same
        #    and the scale is exactly 1 -- nothing is touched and the result is bit for bit the single-GPU one.)
        #    The scale is formed and applied on the device; the log-weights come to the host with the ONE copy of this
        #    method (step 3), where they are checked.
        w_src = f._w
        uneven = any(cnt != spans[0][1] for _, cnt in spans)
        mine_log = None
        if self._islands_apart or uneven or f.weight_scale:
            # (with scaled weights -- GSE_OP_WEIGHT_SCALE -- a shard's live weights also carry a binary exponent of their own)
            mine_log = torch.zeros(P, dtype=torch.float64, device=self._merge_device)
            mine_log[self.rank:self.rank + 1].copy_(f._island[0] + (f._wscale * 0.6931471805599453 if f.weight_scale else 0.0))
            self._comm.all_reduce(mine_log)
            self.collectives_issued += 1
            mine_log = mine_log.to(dev)
            scale = torch.exp(mine_log[self.rank] - mine_log.max()).to(torch.float32)      # (exactly 1 on the heaviest shard)
            w_src = f._w * scale
        # 1. tile sums of every shard
        mine = torch.empty((tiles[self.rank],), dtype=torch.float64, device=dev)
        _lib.check(lib.gse_pf_tile_sums(w_src.data_ptr(), f._ld, 1, n, mine.data_ptr(), stream))
        sums = self._all_gather_uneven(mine, tiles).to(dev)
        # 2. + 3. the prefix over ALL tiles, then this shard's normalised cumulative weights
        prefix = torch.empty((T_total + 1,), dtype=torch.float64, device=dev)
        _lib.check(lib.gse_tile_prefix(sums.data_ptr(), T_total, prefix.data_ptr(), stream))
        c = torch.empty((n,), dtype=torch.float64, device=dev)
        _lib.check(lib.gse_pf_shard_cumsum(w_src.data_ptr(), n, prefix.data_ptr(), tile_offset, T_total, c.data_ptr(), stream))
        ends = numpy.cumsum(tiles)
        if self._ends_index is None:
            self._ends_index = torch.as_tensor(numpy.concatenate([ends, [T_total]]), device=dev)
        picked = prefix[self._ends_index]
        # the one device -> host copy of the method: the cumulative weights at the shard ends (they size the all-to-all, whose
        # split sizes torch.distributed takes from the host) and, if the islands were apart, their log-weights
        both = (torch.cat([picked, mine_log]) if mine_log is not None else picked).cpu().numpy()
        host = both[:P + 1]
        if mine_log is not None and not numpy.all(numpy.isfinite(both[P + 1:])):
            raise FloatingPointError("a shard's accumulated log-weight is not finite")
        total = host[-1]
        if not (total > 0 and numpy.isfinite(total)):
            raise FloatingPointError("the ensemble's weights do not sum to a positive finite number")
        J = output_boundaries(host[:-1] / total, self.N_particles, r)
        send, recv = exact_resample_plan(J, spans, self.rank)
        # 4. the offspring of this shard's particles, rows [count][5], then the exchange
        count = J[self.rank + 1] - J[self.rank]
        rows = torch.empty((max(count, 1), _lib.GSE_NX), dtype=torch.float32, device=dev)
        _lib.check(lib.gse_pf_shard_gather(f._x.data_ptr(), f._ld, n, c.data_ptr(), J[self.rank], count, self.N_particles,
                                           ctypes.c_double(r), rows.data_ptr(), None, stream))
        got = self._all_to_all_rows(rows[:count], send, recv)
        f._x[0, :, :n] = got.to(dev).t()
        f._w[0, :n] = 1.0 / n                               # every particle of the ensemble weighs 1 / N again:
        if f.weight_scale:
            f._wscale.zero_()
        f._island[0, 0] = float(numpy.log(n / self.N_particles))     # the shard's share is its share of the particles
        f._tiles_valid, f._scan_slot, f._moments_host, f._samples_ok = False, None, None, False
        f._w_uniform = True
        self._islands_apart = False
        self.exact_resamples += 1
        self.last_exact = dict(r=r, J=J, send=send, recv=recv)

    def _all_gather_uneven(self, mine, counts):
        """All ranks' 1-d tensors (lengths ``counts``) concatenated; on the collective's device."""
        width = max(counts)
        padded = torch.zeros((width,), dtype=mine.dtype, device=self._merge_device)
        padded[:mine.numel()] = mine.to(self._merge_device)
        parts = [torch.empty_like(padded) for _ in counts]
        self._comm.all_gather(parts, padded)
        self.collectives_issued += 1
        return torch.cat([p[:k] for p, k in zip(parts, counts)])

    def _all_to_all_rows(self, rows, send, recv):
        src = rows.to(self._merge_device).contiguous()
        dst = torch.empty((sum(recv), rows.shape[1]), dtype=rows.dtype, device=self._merge_device)
        self._comm.all_to_all_single(dst, src, recv, send)
        self.collectives_issued += 1
        return dst

    def _exchange_shares(self):
        """Asynchronous all-reduce of the shards' accumulated log-weights (slot p = rank p's island[0], zeros elsewhere):
        the global weight bookkeeping of SURVEY 8e.  Nothing waits for it until the next resample() / an estimate."""
        self.wait_collectives()
        if self._shares_event is not None and self._shares_recorded:
            torch.cuda.current_stream().wait_event(self._shares_event)     # (the side stream's copy of the previous exchange)
        # a buffer of its own: later launches keep updating island[0] while the collective may still be in flight
        self._shares.zero_()
        if self._merge_device.type == "cuda":
            self._shares[self.rank:self.rank + 1].copy_(self.local._island[0])
        else:
            self._shares[self.rank] = float(self.local._island[0, 0].item())
        self._pending = self._comm.all_reduce(self._shares, async_op=True)
        self.collectives_issued += 1
        self._shares_cycle = self._resamples
        if self._shares_event is not None:
            with torch.cuda.stream(self._side_stream):
                self._pending.wait()                        # the SIDE stream waits for the collective ...
                self._shares_host.copy_(self._shares, non_blocking=True)
                self._shares_event.record(self._side_stream)
            self._shares_recorded = True

    @property
    def shares_lag(self):
        """Resamples between an exchange of the shards' log-weights and the policy's look at them (the same on every rank)."""
        return 1 if self._shares_event is None else max(1, min(4, self.reduce_every - 1))

    def wait_collectives(self):
        if self._pending is not None:
            self._pending.wait()
            self._pending = None

    # ------------------------------------------------------------------ posterior merge
    def _local_row(self):
        """This shard's row of the merge buffer, formed on the HOST (the round-1..3 path; kept as the reference the device
        merge is tested against): synchronises this rank's stream."""
        s0, s1, s2, kernel_pivot = self.local._raw_moments()
        log_island = float(self.local._island[0, 0].item())
        return pack_merge_row(log_island, s0, s1, s2, kernel_pivot, self._pivot)

    def _merged_host(self):
        row = self._local_row()
        if self.group is None:
            return merge_rows(row[None, :], self._pivot)
        self.wait_collectives()
        buf = self._merge_buf
        buf.zero_()
        buf[self.rank] = torch.as_tensor(row, dtype=torch.float64, device=self._merge_device)
        self._comm.all_reduce(buf)
        return merge_rows(buf.cpu().numpy(), self._pivot)

    def _merge_buffers(self):
        if self._dev_rows is None:
            import ctypes
            dev = self.local._x.device
            self._dev_rows = torch.zeros((self.world, MERGE_WIDTH), dtype=torch.float64, device=dev)
            self._dev_moments = torch.zeros((2, 26), dtype=torch.float64, device=dev)      # [0] the shard's, [1] the ensemble's
            self._dev_est = torch.zeros(6, dtype=torch.float64, device=dev)                # mean (5), covariance size
            self._dev_status = torch.zeros(1, dtype=torch.int32, device=dev)
            self._pivot_c = _lib.doubles(self._pivot, 5)
            spec = self._merge_spec = _lib.MergeSpec()
            spec.P, spec.rank = self.world, self.rank
            for k in range(5):
                spec.common_pivot[k] = float(self._pivot[k])
            spec.rows, spec.moments26, spec.merged26 = (self._dev_rows.data_ptr(), self._dev_moments[0].data_ptr(),
                                                        self._dev_moments[1].data_ptr())
            spec.estimate6, spec.status = self._dev_est.data_ptr(), self._dev_status.data_ptr()
            self._merge_spec_ref = ctypes.byref(spec)

    def merge_begin(self, flushed=False, into=None):
        """Enqueue the posterior merge and return at once (gse_pf_merge_begin, two launches).  When a resample is pending on the
        scans the last update left -- the state between resample() and the next predict(), where the reference's loop asks for
        its estimate (sim_base.py:258-262) -- NOTHING IS FLUSHED (round 5): the moments are those of the ensemble the resample
        will produce, weighted by offspring counts (the resident loop's route: gse_pf_moments_resampled), the row's log-weight is
        what the copy's launch will make of the shard's, and the copy itself keeps riding the next cycle's single launch.  One
        kernel finishes the partial sums, writes this shard's row of the merge buffer ON THE DEVICE (pivot shift, accumulated
        log-weight, binary exponent of the stored weights) and zeroes the other rows; then the all-reduce of the [P][22] buffer
        -- asynchronous on RCCL, so that it overlaps whatever the caller enqueues next.  Without a group (one shard) that same
        kernel also combines and estimates: two launches behind the cycle's one.  Nothing comes to the host (with the gloo
        backend of the CPU tests the buffer has to: gloo reduces host tensors).
        ``flushed=True``: round 4's route (the pending copy runs first, gse_pf_moments of the copied ensemble) -- kept as the
        reference the new route is tested against.  ``into``: a device tensor of 26 doubles that receives the merged moments
        instead of the object's own buffer (a loop that keeps a record per tick, as sim_base.py:287-296 does, hands in row t
        and gets all its covariance sizes from ONE gse_moments_estimates call at the end)."""
        f = self.local
        pending = (not flushed and f._pend_gather and f._gather_slot is not None and f._pend_predict is None and
                   f.N_particles <= (1 << 21))
        if not pending:
            f._flush()
        self._merge_buffers()
        spec = self._merge_spec
        spec.flags = (_lib.GSE_MERGE_PENDING if pending else 0) | (_lib.GSE_MERGE_COV_SIZE if self.merge_cov_size else 0)
        spec.slot = f._slot_arg(f._gather_slot) if pending else 0
        spec.r = float(f._c_r[0])
        spec.island = f._island.data_ptr()
        spec.wscale = f._wscale.data_ptr() if f.weight_scale else None
        if into is not None and not (into.is_cuda and into.dtype == torch.float64 and into.is_contiguous() and into.numel() == 26):
            raise ValueError("into must be a contiguous CUDA tensor of 26 doubles")
        self._merged_out = self._dev_moments[1] if into is None else into
        spec.merged26 = self._merged_out.data_ptr()
        _lib.call("gse_pf_merge_begin", f._x.data_ptr(), f._w.data_ptr(), f._ld, f.N_particles, self._merge_spec_ref,
                  f._ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
        rows = self._dev_rows
        self._merge_work = None
        if self.group is not None:
            self.wait_collectives()
            if self._merge_device.type == "cuda":
                self._merge_work = self._comm.all_reduce(rows, async_op=True)
            else:
                host_rows = rows.cpu()
                self._comm.all_reduce(host_rows)
                rows.copy_(host_rows)
            self.collectives_issued += 1
        self._merge_pending = True

    def merge_end(self):
        """The moments of the whole ensemble as a DEVICE tensor of 26 doubles (gse_pf_moments layout: what gse_loop_control and
        gse_moments_estimates take), after merge_begin: the current stream waits for the all-reduce (no host wait), then ONE
        small kernel combines the shards' rows and forms the estimate (gse_merge_end; ``estimate_device`` holds mean and
        covariance size).  A single shard without a group was combined by merge_begin's kernel already."""
        if not self._merge_pending:
            self.merge_begin()
        if self._merge_work is not None:
            self._merge_work.wait()                       # (stream-ordered: the host does not block)
            self._merge_work = None
        if self.group is not None:
            _lib.call("gse_merge_end", self._merge_spec_ref, torch.cuda.current_stream().cuda_stream)
        self._merge_pending = False
        return self._merged_out

    @property
    def estimate_device(self):
        """[mean (5), largest singular value of the covariance] of the last merge, a device tensor of 6 doubles (no copy)."""
        return self._dev_est

    def _merged(self):
        f = self.local
        m = torch.cat([self.merge_end(), self._dev_status.to(torch.float64), f._wstatus.to(torch.float64)]).cpu().numpy()
        f._check_status(m[27:])                             # (a resample of this shard met a weight sum that is not finite)
        status = int(m[26])
        if status & _lib.GSE_MERGE_STATUS_OVERFLOW:
            raise FloatingPointError("a shard's weight sum is not finite: the stored fp32 weights overflowed the launch's "
                                     "power-of-two scale (GSE_OP_WEIGHT_SCALE)")
        if status & _lib.GSE_MERGE_STATUS_NO_WEIGHT:
            raise FloatingPointError("every shard has zero weight (all likelihoods underflowed)")
        _, mean, cov = moments_to_mean_cov(m[0], m[1:6], m[6:21], m[21:26])
        return mean, cov

    def point_estimate(self):
        """Posterior mean over ALL shards, host array (5,)."""
        return self._merged()[0]

    def point_covariance(self):
        """Largest singular value of the posterior covariance over all shards."""
        return numpy.linalg.svd(self._merged()[1], compute_uv=False)[0]


class ShardedGaussianSumFilter:
    """One Gaussian-sum unscented Kalman filter whose bank of Gaussians lives on all ranks of ``group`` (round 5; VERDICT r04
    missing #4 -- the north star shards "independent ensembles" with an all-reduce for "the global weight sum / posterior
    merge", and the reference's protocol runs banks up to 2^18.5 Gaussians, results/gsf_openloop/gsf_run_seq.py:483-484).

    Rank p owns a contiguous block of the Gaussians.  Its prior rows and its sigma-point noise are those the WHOLE bank would
    draw for these Gaussians (Philox elements are numbered globally: ``element_offset``), so predict + update of the shards
    are, bit for bit, the slices of one filter holding the whole bank.  Shards resample LOCALLY (islands, as the sharded
    particle filter's); right before, the shard's accumulated log-weight absorbs the sum of its live weights
    (gse_island_absorb).  The posterior merge is the particle filter's: gse_gsukf_moments26 gives the mergeable row
    (S0, S1, S2 + sum w P about a pivot), gse_merge_row shifts it to the common pivot, ONE all-reduce of [P][22] doubles
    (RCCL), gse_merge_end combines: mean = sum_p a_p S1_p / S0_p, covariance likewise (filter/gs_ukf.py:438-449 semantics,
    normalised: the estimate of the mixture  sum_p A_p sum_i w_i N(mu_i, P_i)).

    With group=None (or one rank) this is a ParallelGaussianSumUnscentedKalmanFilter with normalised estimates."""

    def __init__(self, f, g, N_particles, x0, state_pdf, measurement_pdf, group=None):
        from gpu_se_amd.filter.gs_ukf import ParallelGaussianSumUnscentedKalmanFilter
        self._comm = None if group is None else (group if hasattr(group, "all_to_all_single") else TorchComm(group))
        self.group = group
        self.world = self._comm.world if group is not None else 1
        self.rank = self._comm.rank if group is not None else 0
        self.N_particles = int(N_particles)
        self._spans = [shard_bounds(self.N_particles, self.world, q) for q in range(self.world)]
        lo, n_local = self._spans[self.rank]
        if n_local < 1:
            raise ValueError("%d Gaussians cannot be dealt to %d ranks" % (self.N_particles, self.world))
        self.local = ParallelGaussianSumUnscentedKalmanFilter(f, g, n_local, x0, state_pdf, measurement_pdf, element_offset=lo)
        if self.local._model is not None:
            raise NotImplementedError("the merge rows are laid out for the 5-state built-in model")
        dev = self.local._w.device
        # a shard's prior weight is its share of the Gaussians
        self._log_island = torch.full((1,), float(numpy.log(n_local / self.N_particles)), dtype=torch.float64, device=dev)
        self._pivot = numpy.asarray(x0.weights, dtype=numpy.float64) @ numpy.asarray(x0.means, dtype=numpy.float64) \
            / float(numpy.sum(x0.weights))
        backend = self._comm.backend if group is not None else None
        self._merge_device = dev if backend in (None, "nccl") else torch.device("cpu")
        self._rows = torch.zeros((self.world, MERGE_WIDTH), dtype=torch.float64, device=dev)
        self._moments = torch.zeros((2, 26), dtype=torch.float64, device=dev)
        self._est = torch.zeros(6, dtype=torch.float64, device=dev)
        self._status = torch.zeros(1, dtype=torch.int32, device=dev)
        self._pivot_c = _lib.doubles(self._pivot, 5)
        import ctypes
        spec = self._spec = _lib.MergeSpec()
        spec.P, spec.rank, spec.flags = self.world, self.rank, _lib.GSE_MERGE_COV_SIZE
        for k in range(5):
            spec.common_pivot[k] = float(self._pivot[k])
        spec.rows, spec.moments26, spec.merged26 = self._rows.data_ptr(), self._moments[0].data_ptr(), self._moments[1].data_ptr()
        spec.estimate6, spec.status = self._est.data_ptr(), self._status.data_ptr()
        self._spec_ref = ctypes.byref(spec)
        self.collectives_issued = 0

    @classmethod
    def from_get_parts(cls, n_local, group=None):
        """The sim_base.get_parts wiring (prior = state-noise mixture shifted to X_ss) with n_local Gaussians on every rank."""
        from gpu_se_amd import sim_base
        from gpu_se_amd.model import Bioreactor
        world = (group.world if hasattr(group, "all_to_all_single") else dist.get_world_size(group)) if group is not None else 1
        xss = Bioreactor.find_SS(numpy.array([0.06, 0.2]), numpy.array([260 / 180, 640 / 24.6, 1000 / 116, 0, 0]))
        state_pdf, measurement_pdf = sim_base.get_noise()
        x0, _ = sim_base.get_noise()
        x0.means += numpy.asarray(xss[None, :], dtype=numpy.float32)
        return cls(Bioreactor.homeostatic_DEs, Bioreactor.static_outputs, n_local * world, x0, state_pdf, measurement_pdf,
                   group=group)

    # ------------------------------------------------------------------ the cycle (all local)
    def predict(self, u, dt):
        self.local.predict(u, dt)

    def update(self, u, z):
        self.local.update(u, z)

    def resample(self, r=None):
        """Local systematic resample of this shard's Gaussians; its accumulated log-weight absorbs the sum of the weights the
        resample is about to replace by 1 / n."""
        f = self.local
        if f._pend_predict is not None or f._pend_update is not None:
            f._flush()                                     # (small banks defer predict and update into resample's launch: the
                                                           #  weight sum is wanted BETWEEN update and resample)
        _lib.call("gse_island_absorb", f._w.data_ptr(), f.N_particles, self._log_island.data_ptr(), None,
                  torch.cuda.current_stream().cuda_stream)
        f.resample(r=r)

    # ------------------------------------------------------------------ posterior merge
    def merge(self):
        """Enqueue the posterior merge and return the merged moments (device tensor, 26 doubles, the particle filter's layout;
        ``estimate_device`` = mean and covariance size): the shard's moments, its row, the all-reduce, the combination --
        stream-ordered, nothing comes to the host (gloo reduces host tensors: the CPU tests' backend copies the buffer)."""
        f = self.local
        stream = torch.cuda.current_stream().cuda_stream
        f._moments26_into(self._moments[0:1])
        self._rows.zero_()
        _lib.call("gse_merge_row", self._moments[0].data_ptr(), self._log_island.data_ptr(), None, self._pivot_c,
                  self._rows[self.rank].data_ptr(), stream)
        if self.group is not None:
            if self._merge_device.type == "cuda":
                self._comm.all_reduce(self._rows)
            else:
                host_rows = self._rows.cpu()
                self._comm.all_reduce(host_rows)
                self._rows.copy_(host_rows)
            self.collectives_issued += 1
        _lib.call("gse_merge_end", self._spec_ref, stream)
        return self._moments[1]

    @property
    def estimate_device(self):
        return self._est

    def _merged(self):
        m = torch.cat([self.merge(), self._status.to(torch.float64)]).cpu().numpy()
        self.local._check_status()                          # (a covariance that could not be factorised: gs_ukf.py:336-339)
        if int(m[26]) & _lib.GSE_MERGE_STATUS_OVERFLOW:
            raise FloatingPointError("a shard's weight sum is not finite")
        if int(m[26]) & _lib.GSE_MERGE_STATUS_NO_WEIGHT:
            raise FloatingPointError("every shard has zero weight (all likelihoods underflowed)")
        _, mean, cov = moments_to_mean_cov(m[0], m[1:6], m[6:21], m[21:26])
        return mean, cov

    def point_estimate(self):
        """Posterior mean over ALL shards, host array (5,) (filter/gs_ukf.py:438-440, normalised)."""
        return self._merged()[0]

    def point_covariance(self):
        """Largest singular value of the posterior covariance over all shards (filter/gs_ukf.py:442-449)."""
        return numpy.linalg.svd(self._merged()[1], compute_uv=False)[0]
