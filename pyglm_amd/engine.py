"""Device-side state and sweep orchestration for one shard of postsynaptic neurons on one MI355X.

The reference runs `for n in range(N): regressions[n].resample(...)` in Python (pyglm/models.py:169-171); here the
shard's neurons go through the HIP kernels in batches: one activation contraction + one PG/log-likelihood pass for
the whole shard, then per batch the omega-weighted Gram, posterior assembly, tableau flips and the Gaussian draw.
PyTorch is used for device memory and streams only; all arithmetic is in libpyglm_hip.so (include/pyglm_hip.h).
"""
import ctypes
import functools
import time

import numpy as np
import torch

from . import _lib
from ._lib import FlipState, CholState, call, ptr

F64 = torch.float64
I32 = torch.int32


def _r(x, m):
    return (x + m - 1) // m * m


def _on_device(fn):
    """run a GibbsEngine method with the engine's GPU as the current device and put the caller's device back afterwards: the library's
    launches act on the current HIP device, and an engine must not move its process onto its own device behind the caller's back (a
    temporary engine on another GPU, several engines in one process)"""
    @functools.wraps(fn)
    def wrapped(self, *args, **kwargs):
        with torch.cuda.device(self.dev):
            return fn(self, *args, **kwargs)
    return wrapped


def make_draws(seed, sweep, neuron_ids, N, D):
    """Host random inputs of the non-PG steps, keyed by (seed, sweep, GLOBAL neuron) so that results do not depend on
    how neurons are sharded: perm = permutation(N) (regression.py:286), u = N uniforms (:315), z = D+1 normals (:334,
    the first sum(a)*B+1 are used)."""
    nloc = len(neuron_ids)
    perm = np.empty((nloc, N), dtype=np.int32)
    u = np.empty((nloc, N))
    z = np.empty((nloc, D + 1))
    for i, n in enumerate(neuron_ids):
        rng = np.random.Generator(np.random.Philox(key=int(seed) & (2 ** 64 - 1), counter=[int(sweep), int(n), 0, 0]))
        perm[i] = rng.permutation(N)
        u[i] = rng.random(N)
        z[i] = rng.standard_normal(D + 1)
    return perm, u, z


def make_gamma_draws(seed, sweep, neuron_ids, alpha):
    """standard Gamma(alpha, 1) variates for the noise-variance update of the Gaussian model (regression.py:433-445), keyed like
    make_draws by (seed, sweep, GLOBAL neuron) on a separate counter lane"""
    g = np.empty(len(neuron_ids))
    for i, n in enumerate(neuron_ids):
        rng = np.random.Generator(np.random.Philox(key=int(seed) & (2 ** 64 - 1), counter=[int(sweep), int(n), 1, 0]))
        g[i] = rng.standard_gamma(alpha)
    return g


def prior_terms(S_w, mu_w, S_b, mu_b):
    """natural parameters (regression.py:138-151) + the per-block prior constant of the collapsed flips.
    S_w (n,N,B,B), mu_w (n,N,B), S_b (n,), mu_b (n,)  ->  Jw, hw, Jb, hb, c0 (n,N)."""
    Jw = np.linalg.inv(S_w)
    hw = np.einsum("nmij,nmj->nmi", Jw, mu_w)
    Jb = 1.0 / S_b
    hb = Jb * mu_b
    _, logdet = np.linalg.slogdet(Jw)
    c0 = 0.5 * logdet - 0.5 * np.einsum("nmi,nmi->nm", mu_w, hw)
    return Jw, hw, Jb, hb, c0


class BlockPrior(object):
    """block-diagonal weight prior of a shard in table form: the K distinct (J_w, h_w, c0) blocks and, per (local neuron, presynaptic
    neuron), the index of its block.  A network prior pushes two or three distinct blocks for N^2 pairs; the tables travel to the
    GPU instead of the expanded (nloc, N, B, B) array.  Accepted by GibbsEngine.sweep in place of the dense Jw (hw and c0 = None)."""

    def __init__(self, Jw_u, hw_u, c0_u, label):
        self.Jw_u = np.ascontiguousarray(Jw_u, dtype=np.float64)
        self.hw_u = np.ascontiguousarray(hw_u, dtype=np.float64)
        self.c0_u = np.ascontiguousarray(c0_u, dtype=np.float64)
        self.label = np.ascontiguousarray(label, dtype=np.int32)

    def dense(self):
        return self.Jw_u[self.label], self.hw_u[self.label], self.c0_u[self.label]


class _Dataset(object):
    pass


class GibbsEngine(object):
    OBS = {"bernoulli": 0, "negbin": 1, "gaussian": 2}

    def __init__(self, N, B, n0=0, n1=None, device=None, obs="bernoulli", xi=1.0, batch=None, mem_budget_bytes=None,
                 design_only=False, visit_order=True, gram=None, likelihood_only=False, planes=None, i8_group=None, i8_slice=None,
                 i8_resident=None, device_share=1):
        """device: torch device of this shard (default: the process's current GPU).  design_only: only the design matrix is kept (basis
        convolution).  likelihood_only: activation / log-likelihood / means only -- no sweep buffers, no residue planes (a held-out data
        set costs X', Y and Psi, nothing else).  i8_group / i8_slice / i8_resident: override the integer Gram's plan (_i8_plan) -- neurons
        per product launch, time bins per slice, X's planes kept (True) or converted per slice (False); tests and probes.  device_share: number of
        engines (ranks) that share this GPU -- each then stays inside an equal share of its memory (bench.py's one-GPU dry run of N ranks)."""
        if not torch.cuda.is_available():
            raise _lib.PglError("pyglm_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        _lib.load()
        self.N, self.B, self.D = int(N), int(B), int(N) * int(B)
        self.n0, self.n1 = int(n0), int(N if n1 is None else n1)
        self.nloc = self.n1 - self.n0
        assert 0 <= self.n0 < self.n1 <= self.N
        self.dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.dev.index is None:
            self.dev = torch.device("cuda", torch.cuda.current_device())
        with torch.cuda.device(self.dev):
            self._i8_over = dict(group=i8_group, slice=i8_slice, resident=i8_resident)
            self._device_share = max(1, int(device_share))
            self._init(N, B, obs, xi, batch, mem_budget_bytes, design_only, visit_order, gram, likelihood_only, planes)

    def _init(self, N, B, obs, xi, batch, mem_budget_bytes, design_only, visit_order, gram, likelihood_only, planes):
        self.obs, self.xi = self.OBS[obs], float(xi)
        self.Dp = _r(self.D + 1, 16)
        self.ldn = _r(self.nloc, 2)
        self.ldj = _r(self.D + 2, 16)
        self.kmax = _lib.load().pgl_flip_kmax()
        self.R = _lib.load().pgl_flip_window_blocks(self.B)
        if self.R < 1:
            raise ValueError("B=%d too large for the proposal window" % self.B)
        self.datasets = []
        # the likelihood Gram X'OX: "fp64" = the fp64-MFMA kernel; "int8" = exact integer arithmetic on the int8 MFMA (residue planes +
        # CRT, pgl_i8_*; operands rounded to integers scaled from their column norms -- measured error several times below the fp64
        # kernel's own); "auto" (default) takes the integer path per data set where it is the faster one and its
        # planes -- whole or in time slices -- fit in memory (_i8_plan)
        self.gram = gram or "auto"
        assert self.gram in ("auto", "fp64", "int8")
        # number of residue planes (moduli) of the integer path: an int, or None = pgl_i8_min_planes (13: integer column norms of 2^50,
        # measured error several times below the fp64 kernel's own); every further plane buys 4 more bits
        self.planes = int(planes) if planes else None
        assert self.planes is None or 1 <= self.planes <= _lib.load().pgl_i8_max_planes()
        self._i8_scratch = None
        # sweep tableau kept in proposal order (updates after a window touch only the rows not yet proposed); False keeps J's order and
        # full-tableau updates -- same decisions, and the final tableau is then the complete sweep(A, S) (used by a full-size test)
        self.visit_order = bool(visit_order)
        # the compacted active block of the weight draw overlays the tableau, which is dead once a batch's flips are done (kept apart
        # with visit_order=False, where tests read the final tableau): two (D+2)^2 buffers per neuron instead of three
        self.share_tableau = self.visit_order
        # neurons per batch: given, or chosen when the first data set is in place (_ensure_batch) -- what is left of the GPU then has to hold
        # the batch's posterior systems AND, on the integer Gram path, the residue planes
        self._batch_arg, self._mem_budget = batch, mem_budget_bytes
        self.nb = None
        self.design_only = design_only
        self.likelihood_only = bool(likelihood_only)
        if likelihood_only:
            self._alloc_shard()
        elif not design_only:
            self._alloc_shard()
            if batch is not None:
                self._ensure_batch()
        self.flip_single_pass = False   # True: one pass over the trailing tableau per proposal window instead of per pair (same bits; tests)
        self.keep_logodds = False       # True: sweep() leaves the flip log-odds in self.logodds (parity tests)
        self.logodds = None
        self.profile = False            # True: every stage of pgl_sweep is timed with HIP events; a collection of stage names: only those
        self._times = None              # pgl_stage_times_t filled by pgl_sweep while `profile` is on
        self._i8_norm = None            # integer Gram: per-batch sums of squares of the columns of omega_n X + the largest omega per neuron
        self._ev = []
        # host-side accounting of sweep(): seconds blocked waiting for the GPU (pgl_get_state) and seconds of host work done while the GPU was
        # busy (host_overlap) -- what a rank's wall time has to be cleared of to see its exposed host share (bench.py: host_busy_ms_per_step)
        self.wait_seconds = 0.0
        self.overlap_seconds = 0.0
        self.launch_seconds = 0.0      # inside the pgl_sweep call itself: ~3 000 launches at the headline size (and, when the queue is full, waiting)

    # ------------------------------------------------------------------ stage timing (HIP events on the launch stream)
    def _tic(self, name, work=0.0):
        if not self.profile:
            return None
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(self.dev))
        return (name, e0, e1, work)

    def _toc(self, h):
        if h is not None:
            h[2].record(torch.cuda.current_stream(self.dev))
            self._ev.append(h)

    @_on_device
    def collect_timings(self):
        """-> {stage: dict(ms=total, calls=n, work=sum)} since the last call"""
        torch.cuda.synchronize(self.dev)
        out = {}
        for name, e0, e1, work in self._ev:
            d = out.setdefault(name, dict(ms=0.0, calls=0, work=0.0))
            d["ms"] += e0.elapsed_time(e1)
            d["calls"] += 1
            d["work"] += work
        self._ev = []
        if self._times is not None:            # the stages of pgl_sweep (HIP events recorded on the launch stream by the library)
            call("pgl_stage_times_collect", ctypes.byref(self._times))
            lib = _lib.load()
            for i in range(_lib.NSTAGES):
                if self._times.calls[i]:
                    d = out.setdefault(lib.pgl_stage_name(i).decode(), dict(ms=0.0, calls=0, work=0.0))
                    d["ms"] += self._times.ms[i]
                    d["calls"] += self._times.calls[i]
                    d["work"] += self._times.work[i]
            self._times = None
        return out

    # ------------------------------------------------------------------ buffers
    def _z(self, *shape, dtype=F64):
        return torch.zeros(*shape, dtype=dtype, device=self.dev)

    def _free_bytes(self):
        """memory this engine may still take on its GPU.  Ranks that share one device (device_share = their number) each stay inside an
        equal share of it, whatever the others have allocated so far."""
        free, total = torch.cuda.mem_get_info(self.dev)
        share = self._device_share
        if share > 1:
            free = min(free, int(total * 0.94) // share - torch.cuda.memory_reserved(self.dev))
        return max(0, free)

    def _ensure_batch(self):
        if self.nb is not None or self.design_only or self.likelihood_only:
            return
        batch = self._batch_arg
        per_neuron = (2 if self.share_tableau else 3) * self.ldj * self.ldj * 8 + 2 * self.kmax * self.ldj * 8 + 2 * (self.kmax + 1) ** 2 * 8
        if batch is None:
            budget = self._mem_budget if self._mem_budget is not None else int(self._free_bytes() * 0.45)
            batch = max(2, min(self.nloc, budget // per_neuron))
            # equal batches: every batch pays the same latency-bound steps (one workgroup per neuron in the proposal and solve kernels,
            # ~100 panel launches of the Cholesky), so 1024 neurons go as 4 x 256, not as 3 x 295 + 139 -- or, before the overlay, 5 x 204 + 4
            batch = -(-self.nloc // -(-self.nloc // batch))
        self.nb = int(min(batch, self.nloc))
        self._alloc_batch()

    def _alloc_batch(self):
        nb, ldj, N, D, kmax = self.nb, self.ldj, self.N, self.D, self.kmax
        self.Jslots = self._z(1, nb, ldj, ldj)
        self.Jbuf = self.Jslots[0]
        self.Mtab = self._z(nb, ldj, ldj)
        self.Ac = self.Mtab if self.share_tableau else self._z(nb, ldj, ldj)
        self.hc = self._z(2, nb, ldj)
        self.Tinv = self._z(nb, 64, 64)
        self.G = self._z(nb, kmax, kmax)
        self.Lws = self._z(nb, (kmax + 1) * (kmax + 1))
        self.Ut = self._z(nb, kmax, ldj)
        self.Wt_ws = self._z(nb, kmax, ldj)
        self.d_idx = self._z(nb, kmax, dtype=I32)
        self.d_sign = self._z(nb, kmax)
        self.d_cnt = self._z(nb, dtype=I32)
        self.batch_k = self._z(nb, dtype=I32)
        self.act = self._z(nb, D + 1, dtype=I32)
        self.na = self._z(nb, dtype=I32)

    def _alloc_shard(self):
        """whole-shard state (all a likelihood-only engine needs besides its data)"""
        N, D, nl = self.N, self.D, self.nloc
        # chain state and per-neuron results live at the head of ONE device buffer, the sweep's inputs behind them (_io_reserve): a sweep
        # is one host-to-device copy from a pinned staging buffer instead of a dozen small ones
        self._din = self._hin = self._hout = None
        self._io_reserve(0)
        self.skip = self._z(nl, dtype=I32)
        self._c0_dense = None
        self.Wt = self._z(self.Dp, self.ldn)          # k-major weights for the activation contraction
        self.bias = self._z(nl)
        self.border = self._z(2 * self.ldn, self.Dp)
        if self.obs == 2:
            # Gaussian observations: omega = 1/eta is constant in t, so X'X is formed once per dataset (add_data) and scaled per sweep
            self.G0 = self._z(self.ldj, self.ldj)
            self.inv_eta = torch.ones(nl, dtype=F64, device=self.dev)
            self.eta = np.ones(nl)

    def _st(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    # ------------------------------------------------------------------ the sweep's I/O block
    def _io_state_layout(self):
        """byte offsets of the state block at the head of the I/O buffer: a (nloc x N int32) | W (nloc x D f64) | b | ll (nloc f64) | status
        (nloc int32) -> (off_a, off_W, off_b, off_ll, off_status, state_bytes), every piece 16-byte aligned"""
        nl, N, D = self.nloc, self.N, self.D
        al = lambda x: (x + 15) // 16 * 16
        off_a = 0
        off_W = al(off_a + 4 * nl * N)
        off_b = al(off_W + 8 * nl * D)
        off_ll = al(off_b + 8 * nl)
        off_st = al(off_ll + 8 * nl)
        return off_a, off_W, off_b, off_ll, off_st, al(off_st + 4 * nl)

    def _io_reserve(self, in_bytes):
        """make the I/O buffer (device) and its pinned staging twin hold the state block + in_bytes of inputs; a_dev / W_dev / b_dev / ll / status
        are views of the device buffer's head (kept across a growth: the state is copied over)"""
        off_a, off_W, off_b, off_ll, off_st, sb = self._io_state_layout()
        need = sb + int(in_bytes)
        if self._din is not None and self._din.numel() >= need:
            return
        cap = max(need, sb + 4096)
        cap = cap + cap // 8                                    # head-room: a prior that changes form between sweeps does not reallocate
        new = torch.zeros(cap, dtype=torch.uint8, device=self.dev)
        if self._din is not None:
            new[:sb].copy_(self._din[:sb])
        self._din = new
        self._stats_dev = None
        self._hin = torch.zeros(cap, dtype=torch.uint8).pin_memory()
        self._hin_np = self._hin.numpy()
        if self._hout is None:
            self._hout = torch.zeros(sb, dtype=torch.uint8).pin_memory()
            self._hout_np = self._hout.numpy()
        nl, N, D = self.nloc, self.N, self.D
        self.a_dev = new[off_a:off_a + 4 * nl * N].view(I32).view(nl, N)
        self.W_dev = new[off_W:off_W + 8 * nl * D].view(F64).view(nl, D)
        self.b_dev = new[off_b:off_b + 8 * nl].view(F64)
        self.ll = new[off_ll:off_ll + 8 * nl].view(F64)
        self.status = new[off_st:off_st + 4 * nl].view(I32)
        self._sweep_cache = None

    # ------------------------------------------------------------------ data
    @_on_device
    def add_data(self, Y, X=None, basis=None):
        """models.py:66-80: Y is (T, N) counts; X (T, N, B) optional, else built on the device from `basis` (L, B)
        by pgl_design_matrix (utils/basis.py:5-34)."""
        Y = np.ascontiguousarray(Y, dtype=np.float64)
        T = Y.shape[0]
        assert Y.shape == (T, self.N)
        ds = _Dataset()
        ds.T, ds.Tp = T, _r(T, 16)
        ds.X = self._z(ds.Tp, self.Dp)
        ds.Xt = self._z(self.Dp, ds.Tp)
        st = self._st()
        if X is None:
            assert basis is not None
            basis = np.ascontiguousarray(basis, dtype=np.float64)
            R, B = basis.shape
            assert B == self.B
            clip = int(basis.min() >= 0 and Y.min() >= 0)
            S_dev = torch.from_numpy(Y).to(self.dev)
            b_dev = torch.from_numpy(basis).to(self.dev)
            call("pgl_design_matrix", ptr(S_dev), self.N, ptr(b_dev), ptr(ds.X), self.Dp, ptr(ds.Xt), ds.Tp, T, self.N, self.B, R, clip, st)
            torch.cuda.synchronize(self.dev)
        else:
            X = np.ascontiguousarray(X, dtype=np.float64).reshape(T, self.D)
            ds.X[:T, :self.D] = torch.from_numpy(X).to(self.dev)
            ds.X[:T, self.D] = 1.0
            call("pgl_transpose", ptr(ds.X), self.Dp, ptr(ds.Xt), ds.Tp, T, self.D + 1, st)
        ds.elem0 = sum(d.T for d in self.datasets)
        self.datasets.append(ds)
        if self.design_only:
            return ds
        ds.Y = self._z(T, self.ldn)
        ds.Y[:, :self.nloc] = torch.from_numpy(np.ascontiguousarray(Y[:, self.n0:self.n1])).to(self.dev)
        self._ensure_batch()
        ds.Psi = self._z(T, self.ldn)
        if not self.likelihood_only:
            ds.OK = self._z(ds.Tp, 2 * self.ldn)      # [Omega | Kappa], rows >= T stay zero
        ds.llpart = self._z(_lib.load().pgl_pg_loglik_partials(T), self.nloc)
        if self.likelihood_only:
            torch.cuda.synchronize(self.dev)
            ds.X = ds.OK = None                   # the activation contraction reads X' only
            ds.int8 = False
            return ds
        plan = self._i8_plan(T)
        ds.int8 = plan is not None
        if ds.int8:
            # residue planes of X, scaled column by column from the columns' norms and maxima: kept for the life of the data set where
            # they fit (plan["resident"]), else converted slice by slice inside the sweep
            lib = _lib.load()
            ds.planes = plan["planes"]
            stat = self._z(2, self.D)
            ds.sA = self._z(self.D)
            call("pgl_i8_colstats", ptr(ds.X), self.Dp, None, 0, T, self.D, 1, ptr(stat[0]), ptr(stat[1]), st)
            call("pgl_i8_scales", ptr(stat[0]), ptr(stat[1]), self.D, T, ds.planes, ptr(ds.sA), st)
            ds.xmax = stat[0]               # column maxima of X: with _i8_norm, the sweep takes the norms of omega_n X per batch (pgl_sweep_t.i8_norm)
            if self._i8_norm is None:
                self._i8_norm = self._z((self.nb + 2) * self.Dp + self.nloc + 2)
            ds.PA = None
            if plan["resident"]:
                ds.PA = torch.empty(lib.pgl_i8_plane_bytes(self.D, T) // lib.pgl_i8_max_planes() * ds.planes, dtype=torch.int8, device=self.dev)
                call("pgl_i8_planes_t", ptr(ds.Xt), ds.Tp, None, 0, ptr(ds.sA), ptr(ds.PA), T, self.D, 1, ds.planes, 0, st)
            self._i8_reserve(T, plan)
            torch.cuda.synchronize(self.dev)
        if self.obs == 2:
            ones = self._z(ds.Tp, 2)
            ones[:T, 0] = 1.0
            call("pgl_weighted_gram", ptr(ds.X), self.Dp, self.Dp, ptr(ones), 2, ds.Tp, self.D, 1, ptr(self.G0), self.ldj, self.ldj * self.ldj,
                 int(len(self.datasets) > 1), st)
            torch.cuda.synchronize(self.dev)
        return ds

    # ------------------------------------------------------------------ integer-MFMA Gram: when, and its scratch
    I8_GROUPS = (64, 32, 16, 8, 4, 2, 1)     # neurons converted and multiplied per launch (their planes are `planes` T D bytes each)
    I8_MIN_D, I8_MIN_T = 1024, 2048
    I8_SMALL_T = 16384                       # below I8_MIN_D columns: only for long data sets, and only where a cost model says so (_i8_pays)

    def _i8_pays(self, T, planes=13):
        """gram='auto' below I8_MIN_D columns: the integer path only if a cost model says it wins by 10 %.  Per neuron and time bin, from rates
        measured on MI355X at T = 50 000 (profiles/archive/r04_small_D_crossover.md): the fp64 kernel takes 2.9e-14 s per multiply-add slot of its
        lower 128-tiles; an item of the integer product (one 320-tile of one plane) 1.9e-8 s on its CU, a launch _i8_rounds item-times for
        its group of neurons; the plane conversion 1.95e-13 s per byte.  Measured, ms per sweep int8 / fp64: D = 320 12.8 / 17.7, D = 500
        (padded to 640) 33.6 / 33.7, D = 640 41.5 / 56, D = 650 (padded to 960) 64.3 / 79.6, D = 900 95.2 / 165.6.
        The group size the model is priced at comes from the WHOLE model (min(64, N) neurons per launch), not from how many neurons this
        shard or batch holds: the two Gram paths differ in the last bits, so the choice must be the same on 1 GPU and on 8 (a shard of 2
        neurons of a 128-neuron model takes the integer path like the whole model does, at a launch it does not fill)."""
        if T < self.I8_SMALL_T:
            return False
        nt, nq = -(-self.D // 128), -(-self.D // 320)
        t64 = nt * (nt + 1) // 2 * 128 * 128 * 2.9e-14
        G = max(g for g in self.I8_GROUPS if g <= max(1, min(self.I8_GROUPS[0], self.N)))
        t8 = self._i8_rounds(G, nq * (nq + 1) // 2, planes) / G * 1.9e-8 + planes * nq * 320 * 1.95e-13
        return t8 < 0.9 * t64

    @staticmethod
    def _i8_rounds(G, ntiles, planes, cus=256):
        """item-times one product launch over G neurons takes (pgl_i8gram.hip): G a multiple of 8 -> per-XCD lists (cus / 8 workgroups each)
        of (planes - 1) x G/8 x ntiles whole items, then the last plane's items in K quarters; otherwise one flat list.  What the choice of
        the group size goes by: 8 neurons of a small model (BASELINE configs[1]: 3 tiles per plane, 312 items for 256 CUs) leave the chip
        half idle in their second round, 64 run 9.75 full rounds."""
        if G % 8:
            return float(-(-ntiles * planes * G // cus))
        per = cus // 8
        npx = G // 8
        full, quarters = ntiles * npx * (planes - 1), 4 * ntiles * npx
        t, rf = divmod(full, per)
        if rf:
            t += 1
            quarters = max(0, quarters - 4 * (per - rf))
        return t + -(-quarters // per) / 4.0

    def _i8_plan(self, T):
        """-> None (this data set's Gram runs on the fp64 kernel) or how it goes through the int8 MFMA: dict(planes, resident, G, slice).
        The integer path is taken if asked for, or (auto) at shapes where it is the faster one (320 x 320 tiles, 13 planes: not for small D
        or short T).  Memory decides the rest: X's planes stay resident if they are a small part of what is free; the planes of omega_g X
        for a group of G neurons and the group's residues must fit -- G = 8 (one neuron per XCD) at large D, more where a plane has only
        a few tiles (_i8_rounds); if a whole data set's planes do not fit (BASELINE configs[4]: 86 GB of planes per neuron), the product
        runs in time slices that add up in the residues (pgl_sweep_t.i8_slice), and only if not even a short slice fits does the data set
        fall back to the fp64 kernel (with a warning)."""
        if self.obs == 2 or self.design_only or self.likelihood_only or self.gram == "fp64":
            return None
        if self.gram != "int8" and (T < self.I8_MIN_T or (self.D < self.I8_MIN_D and not self._i8_pays(T, self.planes or 13))):
            return None
        lib = _lib.load()
        planes = self.planes or lib.pgl_i8_min_planes(T)
        if lib.pgl_i8_norm_bits(planes, T) < 8:
            raise ValueError("%d residue planes cannot hold T = %d time bins" % (planes, T))
        over = self._i8_over
        mp = lib.pgl_i8_max_planes()
        Dq = lib.pgl_i8_padded_rows(self.D)
        pa_full = lib.pgl_i8_plane_bytes(self.D, T) // mp * planes
        kp_full = max(256, -(-T // 64) * 64)                     # bytes per plane row: the bins, padded to the 64-byte K tile
        per_bin = planes * Dq                                    # bytes of one time bin in one set of planes
        r1 = lib.pgl_i8_residue_bytes(self.D) // mp * planes
        free = self._free_bytes()
        have = self._i8_scratch[0] if self._i8_scratch else 0
        budget = int(0.85 * free) + have
        gmax = int(max(1, min(over["group"] or self.I8_GROUPS[0], self.nb or self.nloc)))
        groups = [G for G in self.I8_GROUPS if G <= gmax] or [1]
        if over["group"]:
            groups = [G for G in groups if G == min(over["group"], gmax)] or [min(over["group"], gmax)]
        keep = pa_full <= 0.3 * free if over["resident"] is None else bool(over["resident"])
        ntiles = (Dq // 320) * (Dq // 320 + 1) // 2
        cus = torch.cuda.get_device_properties(self.dev).multi_processor_count

        def fit(resident, G):      # time bins per slice that fit with this choice
            fixed = (pa_full if resident else 0) + G * r1 + (3 * G * Dq * Dq if G % 8 == 0 else 0)
            return int((budget - fixed) // (per_bin * (G + (0 if resident else 1))))
        if over["slice"]:
            S = max(64, int(over["slice"]) // 64 * 64)
            return dict(planes=planes, resident=keep, G=groups[0] if over["group"] else min(gmax, 8), slice=S if S < T else 0)
        for resident in ((True, False) if keep else (False,)):
            ok = [G for G in groups if fit(resident, G) >= kp_full]
            if ok:
                # the largest group up to 8 that fits (one neuron per XCD), more only where the per-neuron cost of a launch (item-times) falls
                # by 1.5 % or more with it (memory is better spent elsewhere)
                cost = {G: self._i8_rounds(G, ntiles, planes, cus) / G for G in ok}
                small = [G for G in ok if G <= 8]
                pick = max(small) if small else min(ok)
                for G in sorted(G for G in ok if G > 8 and G > pick):
                    if cost[G] < 0.985 * cost[pick]:
                        pick = G
                return dict(planes=planes, resident=resident, G=pick, slice=0)
        cands = [(resident, G, fit(resident, G)) for resident in ((True, False) if keep else (False,)) for G in groups if G <= 8]
        cands = [c for c in cands if c[2] >= 16384]
        if not cands:
            if self.gram == "int8":
                raise _lib.PglError("gram='int8': the residue planes do not fit")
            import warnings
            warnings.warn("gram='auto': not even a 16384-bin slice of the residue planes of this data set (D = %d, T = %d) fits in the %.0f GB free "
                          "on %s; its likelihood Gram runs on the fp64 MFMA kernel" % (self.D, T, free / 1e9, self.dev), RuntimeWarning, stacklevel=3)
            return None
        resident, G, S = max(cands, key=lambda c: (min(c[2], kp_full // 2), c[1]))       # long slices first (at least two are needed anyway), then large groups
        nsl = -(-T // (S // 1024 * 1024))
        S = -(-(-(-T // nsl)) // 1024) * 1024                                             # equal slices, multiples of 1024 bins
        return dict(planes=planes, resident=resident, G=G, slice=int(S))

    def _i8_reserve(self, T, plan):
        """scratch for the planes of omega_g X (one time slice of them) and the residues of a group of G neurons -- and for a slice of X's own
        planes where those are not resident -- sized for the largest need seen so far.  One slice length per engine (pgl_sweep_t.i8_slice): the
        shortest any of its data sets needs; a data set no longer than that runs unsliced, and slices of any length add up to the same bits,
        so data sets planned with different slicing simply share the shorter one."""
        lib = _lib.load()
        mp, planes = lib.pgl_i8_max_planes(), plan["planes"]
        S, G, T0 = plan["slice"], plan["G"], 0
        if self._i8_scratch:
            S0 = self._i8_scratch[6]
            if S0 and (not S or S0 < S):
                S = S0
        Ts = min(S, T) if S else T
        pb1, r1 = lib.pgl_i8_plane_bytes(self.D, Ts) // mp * planes, lib.pgl_i8_residue_bytes(self.D) // mp * planes
        pas = 0 if plan["resident"] else pb1
        if self._i8_scratch:
            _, T0, G0, PB, R, _, S0, PAs = self._i8_scratch[:8]
            if G0 == G and S0 == S and T0 >= T and PB.numel() >= G * pb1 and R.numel() >= G * r1 and (PAs.numel() if PAs is not None else 0) >= pas:
                return
            pb1, r1 = max(pb1, PB.numel() // G0), max(r1, R.numel() // G0)      # keep what earlier data sets need
            pas = max(pas, PAs.numel() if PAs is not None else 0)
            G = min(G, G0)
        self._i8_scratch = None
        torch.cuda.empty_cache()
        Dq = lib.pgl_i8_padded_rows(self.D)
        self._i8_scratch = (G * (pb1 + r1) + pas, max(T, T0), G, torch.empty(G * pb1, dtype=torch.int8, device=self.dev),
                            torch.empty(G * r1, dtype=torch.int8, device=self.dev),
                            self._z(3, G, self.D),          # per group: column maxima, sums of squares, scales of omega_g X
                            S, torch.empty(pas, dtype=torch.int8, device=self.dev) if pas else None,
                            # the three extra residue slots per neuron of the K-quarter split of the last plane (groups that fill the per-XCD lists)
                            torch.empty(3 * G * Dq * Dq, dtype=torch.int8, device=self.dev) if G % 8 == 0 else None)

    @_on_device
    def drop_int8(self, i):
        """put data set i on the fp64 Gram kernel and release its residue planes (the population model does this on every rank when ANY
        rank could not take the integer path, so that the choice never depends on the rank)"""
        ds = self.datasets[i]
        if ds.int8:
            ds.int8 = False
            ds.PA = ds.sA = ds.xmax = None
            if not any(getattr(d, "int8", False) for d in self.datasets):
                self._i8_norm = None
                self._i8_scratch = None             # nobody multiplies planes any more: the group buffers (56 GB at cfg3) go back too
            torch.cuda.empty_cache()

    @_on_device
    def set_noise(self, eta):
        """noise variances eta (nloc,) of the Gaussian observation model (regression.py:380-398)"""
        assert self.obs == 2
        self.eta = np.asarray(eta, dtype=np.float64).reshape(self.nloc).copy()
        self.inv_eta.copy_(torch.from_numpy(1.0 / self.eta))

    @_on_device
    def design_matrix(self, i=0):
        ds = self.datasets[i]
        if ds.X is None:                          # likelihood-only engines keep the transposed copy only
            return ds.Xt[:self.D, :ds.T].t().contiguous().cpu().numpy().reshape(ds.T, self.N, self.B)
        return ds.X[:ds.T, :self.D].cpu().numpy().reshape(ds.T, self.N, self.B)

    # ------------------------------------------------------------------ activation / PG / log-likelihood
    def _upload_weights(self, a, W, b):
        aw = (np.asarray(a, dtype=np.float64)[:, :, None] * np.asarray(W, dtype=np.float64)).reshape(self.nloc, self.D)
        Wt = np.zeros((self.Dp, self.ldn))
        Wt[:self.D, :self.nloc] = aw.T
        self.Wt.copy_(torch.from_numpy(Wt))
        self.bias.copy_(torch.from_numpy(np.asarray(b, dtype=np.float64).reshape(self.nloc)))

    def _psi_pass(self, draw, seed, sweep):
        """activation (regression.py:195-201) for the whole shard + PG/kappa/log-lik (:491-511). Returns ll (nloc,) device."""
        st = self._st()
        for i, ds in enumerate(self.datasets):
            h = self._tic("activation", 2.0 * ds.T * self.D * self.nloc)
            call("pgl_activation", ptr(ds.Xt), ds.Tp, ptr(self.Wt), self.ldn, ptr(ds.Psi), self.ldn, ds.T, self.Dp, self.nloc, st)
            self._toc(h)
            h = self._tic("pg_loglik", float(ds.T) * self.nloc)
            om = ds.OK if draw else None
            kp = ctypes.c_void_p(ds.OK.data_ptr() + 8 * self.ldn) if draw else None
            if self.obs == 2:      # self.ll then holds the sums of squared residuals
                call("pgl_gaussian_stats", ptr(ds.Psi), self.ldn, ptr(self.bias), ptr(ds.Y), self.ldn, ptr(self.inv_eta), ptr(om), 2 * self.ldn,
                     kp, 2 * self.ldn, ptr(ds.llpart), ptr(self.ll), int(i > 0), ds.T, self.nloc, st)
                self._toc(h)
                continue
            call("pgl_pg_loglik", ptr(ds.Psi), self.ldn, ptr(self.bias), ptr(ds.Y), self.ldn, ptr(om), 2 * self.ldn, kp, 2 * self.ldn,
                 ptr(ds.llpart), ptr(self.ll), int(i > 0), ds.T, self.nloc, self.obs, self.xi, int(seed), int(sweep), self.n0, ds.elem0, st)
            self._toc(h)
        return self.ll

    @_on_device
    def log_likelihood(self, a, W, b):
        """per-neuron sum_t log p(y_t | psi_t) (regression.py:491-494 summed as at models.py:93-94)."""
        self._upload_weights(a, W, b)
        return self._ll_host(self._psi_pass(False, 0, 0))

    def _ll_host(self, ll_dev):
        return self._ll_host_np(ll_dev.cpu().numpy().copy())

    def _ll_host_np(self, ll):
        if self.obs == 2:          # regression.py:399-403 summed over t: -T/2 log(2 pi eta) - sse / (2 eta)
            T = sum(ds.T for ds in self.datasets)
            ll = -0.5 * T * np.log(2 * np.pi * self.eta) - 0.5 * ll / self.eta
        return ll

    @_on_device
    def sse(self, a, W, b):
        """sum_t (y - mean)^2 per local neuron, the statistic of _resample_eta (regression.py:433-445)"""
        assert self.obs == 2
        self._upload_weights(a, W, b)
        return self._psi_pass(False, 0, 0).cpu().numpy().copy()

    @_on_device
    def psi(self, a, W, b, i=0):
        self._upload_weights(a, W, b)
        self._psi_pass(False, 0, 0)
        ds = self.datasets[i]
        return ds.Psi[:, :self.nloc].cpu().numpy()

    # ------------------------------------------------------------------ one Gibbs sweep of the shard's regressions
    @_on_device
    def sweep(self, a, W, b, rho, Jw, hw, Jb, hb, c0, perm, u, z, seed, sweep, omega_override=None, host_overlap=None, nrun=0,
              after_queue=None, readback=True, nfirst=0, copy=True, want_stats=False):
        """regression.py:265-280 for every local neuron, as ONE call of pgl_sweep (include/pyglm_hip.h): the whole sweep is queued on the
        stream without a host synchronisation.  a (nloc,N) bool, W (nloc,N,B), b (nloc,), hyper-parameters in natural form
        (prior_terms), random inputs from make_draws.  Returns (a, W, b, ll_before) as host arrays.
        omega_override: list of (T, nloc) arrays replacing the PG draws (test hook: the reference fixtures inject omega).
        host_overlap: optional callable run on the host while the GPU works through the queue (seconds at full size), e.g. to draw the
        next sweep's permutations.
        nrun, nfirst: sweep only local neurons [nfirst, nfirst + nrun) (what one rank of a larger job would do, timed on this GPU -- bench.py's
        scaling_proxy times every shard of a G-rank job this way); the returned rows of the others are their input.
        after_queue: optional callable(engine) run right after the sweep has been queued, with the engine's device current -- the
        population model packs the new rows (packed_state) and starts its all_gather there, behind the sweep on the same stream.
        readback=False: the new (a, W, b) stay on the device (a_dev / W_dev / b_dev, packed_state); only ll and the status flags come
        back: returns (None, None, None, ll); flags of a non-positive-definite system do not raise here (self.last_status; they travel in
        the packed rows and the population model raises on every rank).
        want_stats: the rows' sufficient statistics for the network prior (row_stats) are taken behind the sweep and come back with the
        state in the same wait: self.last_row_stats (nloc, 1 + B + B^2), a view of a pinned buffer valid until the next sweep.
        copy=False: the returned a (int32), W, b are VIEWS of the engine's pinned read-back buffer, valid until its next sweep -- for a
        caller that stores them into arrays of its own right away (the population model: one pass over 42 MB at N = 1024 instead of three)."""
        nloc, N, B, D = self.nloc, self.N, self.B, self.D
        if self.likelihood_only or self.design_only:
            raise _lib.PglError("this engine was built without sweep buffers (likelihood_only / design_only)")
        self._ensure_batch()
        st = self._st()
        a = np.asarray(a).astype(bool)
        rho = np.asarray(rho, dtype=np.float64)
        det = np.all((rho < 1e-6) | (rho > 1 - 1e-6), axis=1)           # regression.py:153-155 (decided again, per row, on the device)
        # chain state + hyper-parameters + random inputs -> ONE pinned staging buffer -> one copy to the device
        label = None
        if isinstance(Jw, BlockPrior):
            label, c0, hw, Jw = Jw.label, Jw.c0_u, Jw.hw_u, Jw.Jw_u        # tables + labels travel; the device gathers c0
        off_a, off_W, off_b, off_ll, off_st, sb = self._io_state_layout()
        items = [("rho", rho, np.float64), ("Jw", Jw, np.float64), ("hw", hw, np.float64), ("Jb", Jb, np.float64), ("hb", hb, np.float64),
                 ("c0", c0, np.float64), ("perm", perm, np.int32), ("u", u, np.float64), ("z", z, np.float64)]
        if label is not None:
            items.append(("label", label, np.int32))
        arrs, offs, off = {}, {}, sb
        for k, v, dt in items:
            arr = np.ascontiguousarray(v, dtype=dt)
            arrs[k], offs[k] = arr, off
            off = (off + arr.nbytes + 15) // 16 * 16
        self._io_reserve(off - sb)
        hin = self._hin_np
        hin[off_a:off_a + 4 * nloc * N].view(np.int32)[:] = a.reshape(-1)
        hin[off_W:off_W + 8 * nloc * D].view(np.float64)[:] = np.asarray(W, dtype=np.float64).reshape(-1)
        hin[off_b:off_b + 8 * nloc].view(np.float64)[:] = np.asarray(b, dtype=np.float64).reshape(-1)
        for k, arr in arrs.items():
            hin[offs[k]:offs[k] + arr.nbytes] = arr.reshape(-1).view(np.uint8)
        self._din[:off].copy_(self._hin[:off], non_blocking=True)
        base = self._din.data_ptr()
        dp = {k: ctypes.c_void_p(base + o) for k, o in offs.items()}
        dp.setdefault("label", None)
        self.logodds = torch.empty((nloc, N), dtype=F64, device=self.dev) if self.keep_logodds else None
        if label is not None and (self._c0_dense is None):
            self._c0_dense = self._z(nloc, N)
        keep = []
        ovs = []
        if omega_override is not None:
            for i, ds in enumerate(self.datasets):
                ov = torch.from_numpy(np.ascontiguousarray(omega_override[i], dtype=np.float64).reshape(ds.T, nloc)).to(self.dev)
                keep.append(ov)
                ovs.append(ov)
        if self.profile and self._times is None:
            self._times = _lib.StageTimes()
            if self.profile is not True:
                lib = _lib.load()
                names = [lib.pgl_stage_name(i).decode() for i in range(_lib.NSTAGES)]
                self._times.mask = sum(1 << names.index(n) for n in self.profile)
        i8 = self._i8_scratch
        # the argument block of pgl_sweep: every buffer is persistent, so the struct is built once and only what changes from sweep to sweep
        # is assigned (the I/O pointers, the hints, the optional outputs)
        sig = (base, len(self.datasets), id(i8), self.nb, bool(ovs), tuple(int(ds.int8) for ds in self.datasets))
        if self._sweep_cache is None or self._sweep_cache[0] != sig or ovs:
            dsets = (_lib.Dataset * len(self.datasets))()
            for i, ds in enumerate(self.datasets):
                dsets[i] = _lib.Dataset(ds.T, ds.Tp, ptr(ds.X), ptr(ds.Xt), ptr(ds.Y), ptr(ds.Psi), ptr(ds.OK), ptr(ds.llpart), ds.elem0, int(ds.int8),
                                        int(getattr(ds, "planes", 0) or 0), ptr(getattr(ds, "sA", None)), ptr(getattr(ds, "PA", None)),
                                        ptr(ovs[i]) if ovs else None, ptr(getattr(ds, "xmax", None)))
            sw = _lib.Sweep(N, B, self.n0, nloc, self.nb, self.obs, self.xi, int(self.visit_order), self.planes or 0, i8[2] if i8 else 0,
                            dsets, len(self.datasets), ptr(self.a_dev), ptr(self.W_dev), ptr(self.b_dev),
                            None, None, None, None, None, None, None, None, None, None,
                            ptr(getattr(self, "inv_eta", None)), ptr(getattr(self, "G0", None)),
                            ptr(self.ll), ptr(self.status), None,
                            ptr(self.Wt), ptr(self.bias), ptr(self.border), ptr(self.skip), None,
                            ptr(self.Jbuf), ptr(self.Mtab), ptr(self.Ac), ptr(self.hc), ptr(self.Tinv), ptr(self.G), ptr(self.Lws), ptr(self.Ut),
                            ptr(self.Wt_ws), ptr(self.d_idx), ptr(self.d_sign), ptr(self.d_cnt), ptr(self.batch_k), ptr(self.act), ptr(self.na),
                            ptr(i8[3]) if i8 else None, ptr(i8[4]) if i8 else None, ptr(i8[5]) if i8 else None,
                            int(i8[6]) if i8 else 0, ptr(i8[7]) if i8 else None, ptr(i8[8]) if i8 else None, None, 0, 0, 0, 0, 0, 0, None)
            self._sweep_cache = (sig, sw, dsets)
        sw = self._sweep_cache[1]
        if ovs:
            self._sweep_cache = None             # (the override tensors die with this call)
        sw.rho, sw.Jw, sw.hw, sw.label, sw.Jb, sw.hb, sw.c0 = dp["rho"], dp["Jw"], dp["hw"], dp["label"], dp["Jb"], dp["hb"], dp["c0"]
        sw.perm, sw.u, sw.z = dp["perm"], dp["u"], dp["z"]
        sw.logodds, sw.c0_dense = ptr(self.logodds), ptr(self._c0_dense)
        sw.i8_norm = ptr(self._i8_norm) if i8 else None
        n_act = int(a.sum(axis=1).max()) if a.size else 0
        sw.nrun, sw.nfirst, sw.all_deterministic = int(nrun), int(nfirst), int(det.all())
        sw.init_rows_bound = 1 + B * n_act
        sw.active_rows_bound = (1 + B * int(np.round(rho).sum(axis=1).max())) if det.all() else 0
        sw.flip_single_pass = int(self.flip_single_pass)
        sw.times = ctypes.pointer(self._times) if self.profile else None
        t_launch = time.perf_counter()
        call("pgl_sweep", ctypes.byref(sw), int(seed), int(sweep), st)
        self.launch_seconds += time.perf_counter() - t_launch
        self.last_row_stats = None
        if want_stats:
            if getattr(self, "_stats_dev", None) is None:
                self._stats_dev = torch.empty((nloc, 1 + B + B * B), dtype=F64, device=self.dev)
                self._stats_host = torch.empty((nloc, 1 + B + B * B), dtype=F64).pin_memory()
            call("pgl_row_stats", ptr(self.a_dev), ptr(self.W_dev), ptr(self._stats_dev), N, B, nloc, self.n0, st)
            self._stats_host.copy_(self._stats_dev, non_blocking=True)        # (pgl_get_state below waits for the stream)
        if after_queue is not None:
            after_queue(self)
        if host_overlap is not None:
            t_ov = time.perf_counter()
            host_overlap()
            self.overlap_seconds += time.perf_counter() - t_ov
        t_wait = time.perf_counter()
        # state, log-likelihood and flags back into the pinned twin of the state block (pgl_get_state waits for the stream)
        hout = self._hout_np
        hp = self._hout.data_ptr()
        call("pgl_get_state", ctypes.byref(sw), ctypes.c_void_p(hp + off_a) if readback else None, ctypes.c_void_p(hp + off_W) if readback else None,
             ctypes.c_void_p(hp + off_b) if readback else None, ctypes.c_void_p(hp + off_ll), ctypes.c_void_p(hp + off_st), st)
        self.wait_seconds += time.perf_counter() - t_wait
        a_i = hout[off_a:off_a + 4 * nloc * N].view(np.int32).reshape(nloc, N) if readback else None
        W_new = hout[off_W:off_W + 8 * nloc * D].view(np.float64).reshape(nloc, N, B) if readback else None
        b_new = hout[off_b:off_b + 8 * nloc].view(np.float64) if readback else None
        if readback and copy:
            W_new, b_new = W_new.copy(), b_new.copy()
        ll = hout[off_ll:off_ll + 8 * nloc].view(np.float64).copy()
        status = hout[off_st:off_st + 4 * nloc].view(np.int32).copy()
        if want_stats:
            self.last_row_stats = self._stats_host.numpy()
        del keep
        self.last_status = status
        if status.any() and readback:
            # the reference's np.linalg.cholesky raises at the first such neuron (regression.py:369-370), with the neurons before it already
            # updated.  Here the whole shard has been computed: the exception names every neuron concerned (global indices) and carries the
            # shard's results, of which the rows of the other neurons are good; the caller's state is left as it was before the sweep
            bad = np.nonzero(status)[0]
            err = np.linalg.LinAlgError("posterior system not positive definite for neurons %s (local %s, flags %s)"
                                        % ((bad[:8] + self.n0).tolist(), bad[:8].tolist(), status[bad[:8]].tolist()))
            err.neurons = (bad + self.n0).tolist()
            err.flags = status[bad].tolist()
            err.state = (a_i.astype(bool), W_new.copy(), b_new.copy()) if readback else None
            raise err
        if readback and not copy:
            return a_i, W_new, b_new, self._ll_host_np(ll)
        return (a_i.astype(bool) if readback else None), W_new, b_new, self._ll_host_np(ll)       # (astype copies out of the staging buffer)

    @_on_device
    def row_stats(self):
        """[count, sum w, sum w w'] over the active off-diagonal weight vectors of every local neuron's row, (nloc, 1 + B + B^2) on the device,
        from the state the last sweep left there (pgl_row_stats): what the network prior needs of this shard (networks.py:132-149).
        Queued on the current stream (behind the sweep)."""
        out = torch.empty((self.nloc, 1 + self.B + self.B * self.B), dtype=F64, device=self.dev)
        call("pgl_row_stats", ptr(self.a_dev), ptr(self.W_dev), ptr(out), self.N, self.B, self.nloc, self.n0, self._st())
        return out

    @_on_device
    def packed_state(self):
        """the shard's chain state as the sweep left it on the device, one row of bytes per neuron: W | b | eta (0) | row statistics | a
        (models.state_row_layout) -- what a rank contributes to the per-sweep all_gather.  Queued on the current stream (behind the sweep)."""
        from .models import state_row_layout
        ob, oe, os_, oa, rb = state_row_layout(self.N, self.B)
        nl = self.nloc
        p = torch.zeros((nl, rb), dtype=torch.uint8, device=self.dev)
        p[:, :ob] = self.W_dev.view(torch.uint8).view(nl, ob)
        p[:, ob:oe] = self.b_dev.view(torch.uint8).view(nl, 8)
        if self.obs != 2:
            # the slot of the Gaussian model's noise variance carries the sweep's status flags here (0 = fine): after the gather EVERY rank sees
            # every neuron's flag and raises the same LinAlgError in the same sweep, instead of one rank leaving the others at a collective
            p[:, oe:os_] = self.status.to(F64).view(torch.uint8).view(nl, 8)
        p[:, os_:oa] = self.row_stats().view(torch.uint8).view(nl, oa - os_)
        p[:, oa:oa + self.N] = self.a_dev.to(torch.uint8)
        return p

    def _gram(self, s, nbb, slot):
        """omega-weighted Gram of local neurons [s, s+nbb) into the batch's J (regression.py:251-252): the stage on its own (probes, tests;
        a sweep runs the same kernels from pgl_sweep)"""
        D, ldn, Dp, ldj = self.D, self.ldn, self.Dp, self.ldj
        st = self._st()
        J = self.Jslots[slot]
        if self.obs == 2:
            call("pgl_scaled_gram", ptr(self.G0), ldj, ctypes.c_void_p(self.inv_eta.data_ptr() + 8 * s), ptr(J), ldj, ldj * ldj, D, nbb, st)
            return
        for i, ds in enumerate(self.datasets):
            if ds.int8:
                G = self._i8_scratch[2]
                for g0 in range(0, nbb, G):
                    gz = min(G, nbb - g0)
                    self._i8_group(ds, ctypes.c_void_p(ds.OK.data_ptr() + 8 * (s + g0)), 2 * self.ldn, gz,
                                   ctypes.c_void_p(J.data_ptr() + 8 * g0 * self.ldj * self.ldj), int(i > 0))
                continue
            call("pgl_weighted_gram", ptr(ds.X), Dp, Dp, ctypes.c_void_p(ds.OK.data_ptr() + 8 * s), 2 * ldn, ds.Tp, D, nbb, ptr(J), ldj,
                 ldj * ldj, int(i > 0), st)

    def _i8_group(self, ds, om, ldo, gz, Jp, accumulate):
        """J[g] (+)= X' diag(om[:, g]) X for gz <= group size weight columns at `om` (device pointer, leading dimension ldo): column
        statistics -> scales -> per time slice: residue planes -> int8 products mod p (added up in the residues) -> CRT"""
        D, Dp, ldj = self.D, self.Dp, self.ldj
        st = self._st()
        _, _, G, PB, R, stat, S, PAs = self._i8_scratch[:8]
        assert gz <= G
        npl = ds.planes
        for c0 in range(0, gz, 8):          # (the statistics pass takes at most 8 weight columns)
            cz = min(8, gz - c0)
            call("pgl_i8_colstats", ptr(ds.X), Dp, ctypes.c_void_p(om.value + 8 * c0), ldo, ds.T, D, cz, ptr(stat[0][c0:]), ptr(stat[1][c0:]), st)
        call("pgl_i8_scales", ptr(stat[0]), ptr(stat[1]), gz * D, ds.T, npl, ptr(stat[2]), st)
        if not S and ds.PA is not None:
            call("pgl_i8_planes_t", ptr(ds.Xt), ds.Tp, om, ldo, ptr(stat[2]), ptr(PB), ds.T, D, gz, npl, 0, st)
            call("pgl_i8_gram", ptr(ds.PA), ptr(PB), ptr(R), ds.T, D, gz, npl, st)
        else:
            S = S or ds.T
            for t0 in range(0, ds.T, S):
                ts = min(S, ds.T - t0)
                xt = ctypes.c_void_p(ds.Xt.data_ptr() + 8 * t0)
                omt = ctypes.c_void_p(om.value + 8 * t0 * ldo)
                if ds.PA is None:
                    call("pgl_i8_planes_t", xt, ds.Tp, None, 0, ptr(ds.sA), ptr(PAs), ts, D, 1, npl, t0, st)
                call("pgl_i8_planes_t", xt, ds.Tp, omt, ldo, ptr(stat[2]), ptr(PB), ts, D, gz, npl, t0, st)
                call("pgl_i8_gram_slice", ptr(ds.PA) if ds.PA is not None else ptr(PAs), ds.T if ds.PA is not None else 0, t0, ptr(PB), ptr(R), ts, ds.T,
                     D, gz, npl, int(t0 > 0), st)
        call("pgl_i8_crt", ptr(R), ptr(ds.sA), ptr(stat[2]), Jp, ldj, ldj * ldj, ds.T, D, gz, npl, accumulate, st)

    # test hooks --------------------------------------------------------------------------------------------------
    @_on_device
    def posterior(self, i):
        """assembled (J_post (D+1,D+1), h_post (D+1,)) of batch slot i as dense symmetric host arrays"""
        D = self.D
        M = self.Jbuf[i, :D + 2, :D + 2].cpu().numpy()
        L = np.tril(M)
        full = L + np.tril(L, -1).T
        return full[:D + 1, :D + 1], full[D + 1, :D + 1]
