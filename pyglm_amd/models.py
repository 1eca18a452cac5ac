"""Population models with the reference's interface (pyglm/models.py): SparseBernoulliGLM & friends.

`add_data() / resample_model() / log_likelihood()` are drop-ins; underneath, the N per-neuron regressions are not
looped over in Python (models.py:169-171) but sent through the HIP kernels in batches by a `GibbsEngine`, and they
shard by postsynaptic neuron over the ranks of a torch.distributed process group (one process per GPU):

    rank r owns neurons [r*N/G, (r+1)*N/G)   -- its Y columns, its rows of (A, W, b); X is replicated.
    per sweep:  ONE all_gather_into_tensor of the shard's packed rows (W | b | eta or status flags | row statistics | a as bytes), taken from
                the device buffers the sweep updated and launched behind it on the stream (the network prior needs the full (A, W),
                models.py:230 -- and of W only the rows' sufficient statistics, which travel with them);
                all_reduce of the N per-neuron fp64 log-likelihoods (own entries, zeros elsewhere) in log_likelihood().
Random inputs are keyed by (seed, sweep, global neuron), so results do not depend on the number of ranks.
"""
import numpy as np
import numpy.random as npr

from . import networks as _networks
from . import regression as _regression
from .utils.utils import logistic


def _dist():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist
    except ImportError:
        pass
    return None


def state_row_layout(N, B):
    """byte layout of one neuron's row of the packed state the ranks exchange: W (N*B doubles) | b (1 double) | eta (1 double: noise variance,
    Gaussian model; 0 otherwise) | row statistics (1 + B + B^2 doubles: count, sum and sum of outer products of the row's active off-diagonal
    weight vectors -- the network prior's sufficient statistics, networks.py:132-149) | a (N bytes, padded to a multiple of 8)
    ->  (offset of b, of eta, of the statistics, of a, bytes per row)"""
    D = N * B
    os_ = 8 * D + 16
    oa = os_ + 8 * (1 + B + B * B)
    return 8 * D, 8 * D + 8, os_, oa, oa + -(-N // 8) * 8


def host_row_stats(a, W, n0):
    """the statistics pgl_row_stats forms on the device, on the host: [count, sum w, sum w w'] over the active presynaptic m != n0 + n of every row
    n of a (n, N) / W (n, N, B) -- for state that never was on a GPU (a network resampled from user-supplied (A, W), the CPU test engine)"""
    a = np.array(a, dtype=bool)
    n, N = a.shape
    B = W.shape[-1]
    r = np.arange(n)
    ok = n0 + r < N
    a[r[ok], n0 + r[ok]] = False
    out = np.empty((n, 1 + B + B * B))
    for i in range(n):                       # row by row: a row's numbers must not depend on how many rows are handed in together
        Wm = np.asarray(W[i], dtype=np.float64)[a[i]]
        out[i, 0] = Wm.shape[0]
        out[i, 1:1 + B] = Wm.sum(axis=0)
        out[i, 1 + B:] = Wm.T.dot(Wm).reshape(-1)
    return out


def shard_bounds(N, world, rank):
    """contiguous, balanced neuron ranges"""
    base, rem = divmod(N, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class NonlinearAutoregressiveModel(object):
    """(models.py:8-201) y_n[t] ~ p(f(w_n . x[t])), x = basis-filtered history of all neurons."""
    DRAW_AHEAD_MIN_SIZE = 0            # N*N*B from which the next sweep's host draws are made while the GPU is busy (every size: a small model's sweep leaves the host idle for most of a millisecond)

    def __init__(self, N, regressions, basis=None, B=10, device=None, engine_factory=None, seed=None, engine_kwargs=None, shard=None):
        self.N = N
        assert len(regressions) == N
        self.regressions = regressions
        if basis is None:
            basis = np.eye(B)
        else:
            assert basis.ndim == 2
        self.basis = basis
        self.B = self.basis.shape[1]
        self.data_list = []
        # ---- device / distribution
        dist = _dist()
        self.world = dist.get_world_size() if dist else 1
        self.rank = dist.get_rank() if dist else 0
        self.n0, self.n1 = shard_bounds(N, self.world, self.rank)
        # shard=(n0, n1): this process sweeps neurons [n0, n1) only, whatever the process group says, and exchanges nothing -- the other
        # rows of (A, W, b) keep their values.  What ONE rank of a job too large for the GPUs at hand does (bench.py --neurons).
        self._shard_override = shard is not None
        if shard is not None:
            self.n0, self.n1 = int(shard[0]), int(shard[1])
            assert 0 <= self.n0 < self.n1 <= N
        self._device = device
        self._engine_factory = engine_factory
        self._engine_kwargs = engine_kwargs or {}
        self._engine = None
        # like the reference's PG samplers (regression.py:476) the stream seed comes from NumPy's global RNG
        self.seed = int(npr.randint(2 ** 31)) if seed is None else int(seed)
        if dist and seed is None:
            import torch
            t = torch.tensor([self.seed], dtype=torch.int64)
            if dist.get_backend() == "nccl":
                t = t.to(self._comm_dev())
            dist.broadcast(t, 0)
            self.seed = int(t.item())
        self._adopt_state()
        self.sweeps_done = 0
        self._fresh_stats = self._stats_kept = None      # per-row statistics for the network prior (resample_regressions -> resample_network)
        self.comm_seconds = 0.0        # wall time this rank has spent inside collectives (all_gather of rows, scalar all_reduce)
        self.collectives = 0           # collectives issued by this rank (one per sweep + one per log_likelihood())

    def _comm_dev(self):
        """device the RCCL collectives of this rank stage through: the shard's GPU"""
        import torch
        if self._engine is not None and hasattr(self._engine, "dev"):
            return self._engine.dev
        return torch.device(self._device) if self._device else torch.device("cuda", torch.cuda.current_device())

    # ---- engine
    @property
    def engine(self):
        if self._engine is None:
            reg = self.regressions[0]
            kw = dict(obs=getattr(reg, "_obs", "bernoulli") or "bernoulli", xi=getattr(reg, "xi", 1.0))
            kw.update(self._engine_kwargs)
            if self._engine_factory is not None:
                self._engine = self._engine_factory(self.N, self.B, self.n0, self.n1, **kw)
            else:
                from .engine import GibbsEngine
                import torch
                dev = self._device or ("cuda:%d" % torch.cuda.current_device())
                self._engine = GibbsEngine(self.N, self.B, self.n0, self.n1, device=dev, **kw)
        return self._engine

    # ---- chain state: three arrays of the model, of which every regression's (a, W, b) are row views (regression._adopt)
    def _adopt_state(self):
        """(re)build the model's state arrays from its regressions and make their a / W / b views of them.  Cheap when nothing changed hands:
        N identity checks.  A regression that was swapped in from outside (`model.regressions[n] = other`) is adopted here."""
        N = self.N
        st = getattr(self, "_st", None)
        regs = self.regressions
        if st is not None and all(r._store is not None and r._store[0] is st[0] and r._store[3] == n for n, r in enumerate(regs)):
            return st
        st = (np.zeros((N, N), dtype=bool), np.zeros((N, N, self.B)), np.zeros((N, 1)))
        for n, r in enumerate(regs):
            r._adopt(st[0], st[1], st[2], n)
        self._st = st
        return st

    # ---- state read-backs (models.py:54-64): row = postsynaptic.  Copies, like the reference's np.array([...])
    @property
    def weights(self):
        return self._adopt_state()[1].copy()

    @property
    def adjacency(self):
        return self._adopt_state()[0].copy()

    @property
    def biases(self):
        return self._adopt_state()[2].ravel().copy()

    def add_data(self, data, X=None):
        """(models.py:66-80)"""
        N, B = self.N, self.B
        assert isinstance(data, np.ndarray) and data.ndim == 2 and data.shape[1] == self.N
        T = data.shape[0]
        if X is not None:
            assert X.shape == (T, N, B)
        ds = self.engine.add_data(data, X=X, basis=self.basis)
        dist = _dist()
        if dist is not None and hasattr(self.engine, "drop_int8"):
            # gram="auto" looks at the free memory of ITS GPU: make the choice collective (integer path only if every rank can take it), so
            # that results never depend on which rank a neuron lives on
            import torch
            t = torch.tensor([int(bool(getattr(ds, "int8", False)))], dtype=torch.int32)
            if dist.get_backend() == "nccl":
                t = t.to(self._comm_dev())
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if not int(t.item()):
                self.engine.drop_int8(len(self.engine.datasets) - 1)
        self.data_list.append((_LazyX(self.engine, len(self.engine.datasets) - 1) if X is None else X, data))

    # ---- local <-> global state
    def _local_state(self):
        # views, not copies: engine.sweep uploads them before it returns, and the model's arrays are only written after it has
        A, W, b = self._adopt_state()
        return A[self.n0:self.n1], W[self.n0:self.n1], b[self.n0:self.n1, 0]

    def _store_rows(self, lo, hi, a, W, b):
        """rows [lo, hi) of the chain state <- (a, W, b): three array assignments (the regressions' a / W / b are views of these arrays).
        a may be boolean or 0/1 integers / bytes, W (rows, N, B) or (rows, N*B), contiguous or strided: one pass over each"""
        A_, W_, b_ = self._adopt_state()
        A_[lo:hi] = a
        W_[lo:hi].reshape(hi - lo, -1)[...] = np.asarray(W).reshape(hi - lo, -1) if getattr(W, "ndim", 2) == 3 else W
        b_[lo:hi, 0] = np.asarray(b).reshape(-1)

    def _gather_rows(self, arr):
        """all_gather of per-neuron rows over the shard axis (ranks may own different counts)"""
        dist = _dist()
        if dist is None or self._shard_override:
            return arr
        import time
        import torch
        t0 = time.perf_counter()
        nccl = dist.get_backend() == "nccl"
        counts = [shard_bounds(self.N, self.world, r) for r in range(self.world)]
        maxc = max(hi - lo for lo, hi in counts)
        was_bool = arr.dtype == np.bool_
        if was_bool:
            arr = arr.astype(np.uint8)           # (bytes travel; not every collective backend takes bool tensors)
        pad = np.zeros((maxc,) + arr.shape[1:], dtype=arr.dtype)
        pad[:arr.shape[0]] = arr
        t = torch.from_numpy(np.ascontiguousarray(pad))
        if nccl:
            t = t.to(self._comm_dev())
        outs = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(outs, t)
        out = np.concatenate([o.cpu().numpy()[: hi - lo] for o, (lo, hi) in zip(outs, counts)], axis=0)
        if was_bool:
            out = out.astype(bool)
        self.comm_seconds += time.perf_counter() - t0
        self.collectives += 1
        return out

    # ---- the per-sweep exchange: one packed all_gather
    def _pack_rows_host(self, a, W, b, eta=None, stats=None):
        """host (a, W, b[, eta]) of the local rows and their statistics -> packed uint8 rows (state_row_layout)"""
        ob, oe, os_, oa, rb = state_row_layout(self.N, self.B)
        nl = a.shape[0]
        buf = np.zeros((nl, rb), dtype=np.uint8)
        buf[:, :ob] = np.ascontiguousarray(W, dtype=np.float64).reshape(nl, -1).view(np.uint8)
        buf[:, ob:oe] = np.ascontiguousarray(b, dtype=np.float64).reshape(nl, 1).view(np.uint8)
        if eta is not None:
            buf[:, oe:os_] = np.ascontiguousarray(eta, dtype=np.float64).reshape(nl, 1).view(np.uint8)
        if stats is not None:
            buf[:, os_:oa] = np.ascontiguousarray(stats, dtype=np.float64).reshape(nl, -1).view(np.uint8)
        buf[:, oa:oa + self.N] = np.asarray(a).astype(np.uint8)
        return buf

    def _unpack_rows(self, buf):
        """-> (a, W, b, eta, row statistics) of the rows in buf (n, row bytes): a (n, N) as bytes and W (n, N*B) are strided VIEWS of buf -- the
        caller stores them into the model's arrays in one pass (_store_rows) --, b, eta and the statistics small copies"""
        ob, oe, os_, oa, rb = state_row_layout(self.N, self.B)
        n = buf.shape[0]
        W = buf[:, :ob].view(np.float64)
        b = buf[:, ob:oe].view(np.float64).reshape(n).copy()
        eta = buf[:, oe:os_].view(np.float64).reshape(n).copy()
        stats = buf[:, os_:oa].view(np.float64).copy()
        a = buf[:, oa:oa + self.N]
        return a, W, b, eta, stats

    def _gather_start(self, packed):
        """packed: torch uint8 (local rows, row bytes), on the shard's GPU or on the host.  Starts ONE all_gather_into_tensor of it (ranks
        may own different counts: rows are padded to the largest shard) and returns a handle for _gather_finish.  RCCL takes the device
        tensor as it is -- the collective is queued behind whatever the current stream still has to do --, gloo a host copy."""
        import time
        import torch
        dist = _dist()
        nccl = dist.get_backend() == "nccl"
        if not nccl and packed.is_cuda:
            # gloo works from host memory, and the copy to the host waits for the whole sweep: it is made in _gather_finish -- after the
            # host has drawn the next sweep's random inputs (host_overlap) -- and is not counted as time inside the collective
            return ("deferred", packed)
        t0 = time.perf_counter()
        counts = [shard_bounds(self.N, self.world, r) for r in range(self.world)]
        # rows per rank in the gathered buffer: the largest shard (N not a multiple of the world size: smaller shards are padded with zero
        # rows); _gather_min_rows (test hook) forces padding where every shard is equal -- e.g. on the one rank a one-GPU box has
        maxc = max(max(hi - lo for lo, hi in counts), int(getattr(self, "_gather_min_rows", 0)))
        if nccl and not packed.is_cuda:
            packed = packed.to(self._comm_dev())
        if packed.shape[0] < maxc:
            pad = torch.zeros((maxc, packed.shape[1]), dtype=torch.uint8, device=packed.device)
            pad[:packed.shape[0]] = packed
            packed = pad
        out = torch.empty((self.world * maxc, packed.shape[1]), dtype=torch.uint8, device=packed.device)
        work = dist.all_gather_into_tensor(out, packed.contiguous(), async_op=True)
        self.comm_seconds += time.perf_counter() - t0
        self.collectives += 1
        return work, out, counts, maxc, packed

    def _gather_finish(self, handle):
        """-> (A, W, b, eta, row statistics) of ALL neurons as host arrays"""
        import time
        if handle[0] == "deferred":
            handle = self._gather_start(handle[1].cpu())         # (the wait for the sweep is the sweep's time, not the collective's)
        t0 = time.perf_counter()
        work, out, counts, maxc, _keep = handle
        work.wait()
        if out.is_cuda:
            # into a pinned buffer kept across sweeps (a fresh pageable 43 MB array per sweep costs several times the transfer)
            import torch
            hb = getattr(self, "_gather_host", None)
            if hb is None or hb.shape != out.shape:
                hb = self._gather_host = torch.empty(out.shape, dtype=torch.uint8).pin_memory()
            hb.copy_(out)
            host = hb.numpy()
        else:
            host = out.numpy()
        if all(hi - lo == maxc for lo, hi in counts):
            rows = host                                    # equal shards, no padding: the gathered buffer IS the N rows
        else:
            rows = np.concatenate([host[r * maxc: r * maxc + (hi - lo)] for r, (lo, hi) in enumerate(counts)], axis=0)
        self.comm_seconds += time.perf_counter() - t0
        return self._unpack_rows(rows)

    def log_likelihood(self, datas=None):
        """(models.py:82-96) sum over datasets and neurons; `datas` other than the stored data is evaluated through a
        temporary engine."""
        if datas is None:
            eng = self.engine
        else:
            eng = self._heldout_engine(datas)
        a, W, b = self._local_state()
        if self.engine_obs() == "gaussian":
            eng.set_noise([r.eta for r in self.regressions[self.n0:self.n1]])
        ll_loc = np.asarray(eng.log_likelihood(a, W, b), dtype=np.float64).reshape(-1)       # one value per local neuron
        dist = _dist()
        if dist is None:
            return float(np.sum(ll_loc))
        # the only collective on the likelihood path: ONE all_reduce -- of the N per-neuron values, each rank contributing its own entries and
        # zeros elsewhere (x + 0 is exact, so the reduction's order cannot matter), then summed in neuron order on every rank: the total is
        # the same to the last bit whatever the number of ranks (a scalar all_reduce of per-rank sums regroups the additions; 8 N bytes
        # instead of 8 cost nothing on xGMI)
        import time
        import torch
        t0 = time.perf_counter()
        vec = np.zeros(self.N)
        vec[self.n0:self.n1] = ll_loc
        t = torch.from_numpy(vec)
        if dist.get_backend() == "nccl":
            t = t.to(self._comm_dev())
        dist.all_reduce(t)
        ll = float(np.sum(t.cpu().numpy()))
        self.comm_seconds += time.perf_counter() - t0
        self.collectives += 1
        return ll

    def _heldout_engine(self, datas):
        """likelihood-only engine (X', Y, Psi: no sweep buffers, no residue planes) for data other than the stored data, on THIS shard's
        device; cached by content, so evaluating the same held-out set every iteration uploads it once"""
        from .utils.utils import fingerprint
        from .engine import GibbsEngine
        items = []
        for d in datas:
            X, Y = d if isinstance(d, tuple) else (None, d)
            if isinstance(X, _LazyX):
                X = None
            items.append((X, np.asarray(Y)))
        key = tuple((None if X is None else fingerprint(X), fingerprint(Y)) for X, Y in items)
        cache = getattr(self, "_heldout_cache", None)
        if cache is None or cache[0] != key:
            self._heldout_cache = None
            kw = dict(obs=self.engine_obs() or "bernoulli", xi=getattr(self.regressions[0], "xi", 1.0), likelihood_only=True)
            if self._engine_factory is not None:
                eng = self._engine_factory(self.N, self.B, self.n0, self.n1, **kw)
            else:
                eng = GibbsEngine(self.N, self.B, self.n0, self.n1, device=self.engine.dev, **kw)
            for X, Y in items:
                eng.add_data(Y, X=X, basis=self.basis)
            cache = self._heldout_cache = (key, eng)
        return cache[1]

    def engine_obs(self):
        return getattr(self.regressions[0], "_obs", "bernoulli")

    @property
    def means(self):
        """(models.py:153-163) E[y | X] per dataset, (T, N)"""
        a, W, b = self._local_state()
        mus = []
        for i in range(len(self.data_list)):
            psi = self.engine.psi(a, W, b, i)
            obs = self.engine_obs()
            mu = logistic(psi) if obs == "bernoulli" else psi if obs == "gaussian" else self.regressions[0].xi * np.exp(psi)
            mus.append(self._gather_rows(np.ascontiguousarray(mu.T)).T)
        return mus

    def generate(self, keep=True, T=100, verbose=False, intvl=10):
        """(models.py:98-151) forward simulation, serial in t (host; not on the Gibbs hot path)."""
        if T == 0:
            return np.zeros((0, self.N))
        assert isinstance(T, int), "Size must be an integer number of time bins"
        N, B = self.N, self.B
        L = self.basis.shape[0]
        flipped = self.basis[::-1]                      # row 0 of the basis = previous bin
        assert not np.allclose(flipped, self.basis)
        Wm = self.weights.reshape(N, N * B)               # as the reference: the stored W, not a*W
        bias = self.biases
        Y = np.zeros((T + L, N))
        X = np.zeros((T + L, N, B))
        for t in range(L, T + L):
            if verbose and t % intvl == 0:
                print("Generate t={}".format(t))
            X[t] = Y[t - L:t].T.dot(flipped)
            psi = Wm.dot(X[t].reshape(N * B)) + bias
            Y[t] = self.regressions[0].rvs(psi=psi)
        if keep:
            self.add_data(Y[L:], X=X[L:])
        return X[L:], Y[L:]

    # ---- Gibbs
    def resample_model(self):
        self.resample_regressions()

    def _sweep_inputs(self):
        """everything engine.sweep needs for the shard at the current chain state: (a, W, b, rho, Jw, hw, Jb, hb, c0, perm, u, z)"""
        from .engine import make_draws, prior_terms
        regs = self.regressions[self.n0:self.n1]
        a, W, b = self._local_state()
        versions = tuple(r._hyp_version for r in regs)
        cache = getattr(self, "_hyper_cache", None)
        # the cached terms are those of a network push (or of the last sweep) and are only trusted while nobody outside holds one of the
        # live hyper-parameter arrays (regression._handed_out): such a holder may edit in place at any time, as users of the reference do
        if cache is not None and cache[0] == versions and not any(r._handed_out for r in regs):
            rho, Jw, hw, Jb, hb, c0 = cache[1]          # pushed by resample_network and untouched since
        else:
            hyp = [r._hyper() for r in regs]            # internal read: does not mark the terms stale (the public properties do)
            rho = np.array([h[0] for h in hyp])
            S_w = np.array([h[1] for h in hyp])
            mu_w = np.array([h[2] for h in hyp])
            S_b = np.array([h[3][0, 0] for h in hyp])
            mu_b = np.array([h[4][0] for h in hyp])
            Jw, hw, Jb, hb, c0 = prior_terms(S_w, mu_w, S_b, mu_b)
            self._hyper_cache = (versions, (rho, Jw, hw, Jb, hb, c0))
        # the non-PG random inputs depend on (seed, sweep, neuron) only: those of the NEXT sweep are drawn while the GPU works on
        # this one (engine.sweep's host_overlap hook) and picked up here
        key = (self.seed, self.sweeps_done, self.n0, self.n1)
        pre = getattr(self, "_draws_ahead", None)
        if pre is not None and pre[0] == key:
            perm, u, z = pre[1]
        else:
            perm, u, z = make_draws(self.seed, self.sweeps_done, range(self.n0, self.n1), self.N, self.N * self.B)
        return a, W, b, rho, Jw, hw, Jb, hb, c0, perm, u, z

    def resample_regressions(self):
        """(models.py:169-171) all local neurons through the GPU engine, then ONE all_gather of the new rows (packed on the device and
        launched behind the sweep on its stream; the host draws the next sweep's random inputs meanwhile)."""
        from .engine import make_draws
        regs = self.regressions[self.n0:self.n1]
        a, W, b, rho, Jw, hw, Jb, hb, c0, perm, u, z = self._sweep_inputs()
        self._draws_ahead = None

        def draw_ahead():
            nxt = (self.seed, self.sweeps_done + 1, self.n0, self.n1)
            self._draws_ahead = (nxt, make_draws(self.seed, self.sweeps_done + 1, range(self.n0, self.n1), self.N, self.N * self.B))
        gaussian = self.engine_obs() == "gaussian"
        if gaussian:
            self.engine.set_noise([r.eta for r in regs])
        kw = dict(host_overlap=draw_ahead) if self.N * self.N * self.B >= self.DRAW_AHEAD_MIN_SIZE else {}
        exchange = _dist() is not None and not self._shard_override
        handle = []
        if exchange and not gaussian and hasattr(self.engine, "packed_state"):
            # the shard's new rows never visit the host on their own: packed from the sweep's device buffers, gathered, read back once
            kw.update(after_queue=lambda eng: handle.append(self._gather_start(eng.packed_state())), readback=False)
        need_stats = hasattr(getattr(self, "network", None), "weight_blocks")
        if hasattr(self.engine, "_hout_np"):
            kw["copy"] = False             # (the rows are stored into the model's arrays below, before the engine is used again)
            if need_stats and "readback" not in kw:
                kw["want_stats"] = True    # (taken behind the sweep, back in the same wait as the state)
        a, W, b, self.last_loglik_local = self.engine.sweep(a, W, b, rho, Jw, hw, Jb, hb, c0, perm, u, z, self.seed, self.sweeps_done, **kw)
        if handle:
            A_all, W_all, b_all, flags, stats_all = self._gather_finish(handle[0])
            if np.any(flags != 0):
                # (the eta slot of a non-Gaussian row carries the sweep's status flags: every rank sees them all and raises the same error,
                # with its state as it was before the sweep -- regression.py:369-370 raises LinAlgError from np.linalg.cholesky)
                bad = np.nonzero(flags)[0]
                err = np.linalg.LinAlgError("posterior system not positive definite for neurons %s (flags %s)"
                                            % (bad[:8].tolist(), flags[bad[:8]].astype(int).tolist()))
                err.neurons, err.flags, err.state = bad.tolist(), flags[bad].astype(int).tolist(), None
                raise err
            self.sweeps_done += 1
            self._store_rows(0, self.N, A_all, W_all, b_all)
            self._fresh_stats = stats_all
            return
        if gaussian:
            # noise variances (regression.py:433-445): residual sums of squares under the NEW weights from the device, gamma draws
            # keyed by the global neuron index
            from .engine import make_gamma_draws
            T_total = sum(d[1].shape[0] for d in self.data_list)
            sse = self.engine.sse(a, W, b)
            eta = np.empty(len(regs))
            for i, r in enumerate(regs):
                alpha, beta = r.eta_posterior(T_total, sse[i])
                eta[i] = 1.0 / (make_gamma_draws(self.seed, self.sweeps_done, [self.n0 + i], alpha)[0] * (1.0 / beta))
        self.sweeps_done += 1
        # the rows' sufficient statistics for the network prior (networks.py:132-149), from the state the sweep left on the device
        stats = None
        if need_stats:
            stats = getattr(self.engine, "last_row_stats", None)
            if stats is not None:
                stats = stats.copy()
            else:
                stats = self.engine.row_stats().cpu().numpy() if hasattr(self.engine, "row_stats") else host_row_stats(a, W, self.n0)
        if self._shard_override:
            # (only this shard's rows move: the statistics of the others are formed once and kept -- bench.py --neurons, tools)
            keep = getattr(self, "_stats_kept", None)
            if stats is not None:
                if keep is None:
                    A_, W_, _ = self._adopt_state()
                    keep = self._stats_kept = host_row_stats(A_, W_, 0)
                keep[self.n0:self.n1] = stats
            self._store_rows(self.n0, self.n1, a, W, b)
            self._fresh_stats = keep if stats is not None else None
            if gaussian:
                for i, r in enumerate(regs):
                    r.eta = float(eta[i])
            return
        if not exchange:
            A_all, W_all, b_all, eta_all, stats_all = a, W, b, (eta if gaussian else None), stats
        else:
            import torch
            packed = torch.from_numpy(self._pack_rows_host(a, W, b, eta if gaussian else None, stats))
            A_all, W_all, b_all, eta_all, stats_all = self._gather_finish(self._gather_start(packed))
        self._store_rows(0, self.N, A_all, W_all, b_all)
        self._fresh_stats = stats_all
        if gaussian:
            for n, r in enumerate(self.regressions):
                r.eta = float(eta_all[n])

    # ---- chain state (checkpoint / resume; also lets two samplers be run from one state)
    _STATE_ATTRS = ("regressions", "sweeps_done", "_hyper_cache", "network")

    def get_state(self):
        """deep copy of everything a sweep reads or writes on the host: the regressions (a, W, b, hyper-parameters, noise variances), the
        network prior, the sweep counter (the random inputs are keyed by (seed, sweep, neuron)) and the cached natural-parameter terms.
        Device buffers hold no chain state between sweeps.  `set_state(get_state())` resumes the chain exactly."""
        import copy
        keep = {}
        for r in self.regressions:               # stand-alone engines hold device memory and are not chain state
            keep[id(r)] = (r.__dict__.pop("_engine_cache", None), r.__dict__.pop("_lik_engine_cache", None))
        try:
            st = {k: copy.deepcopy(getattr(self, k)) for k in self._STATE_ATTRS if hasattr(self, k)}
        finally:
            for r in self.regressions:
                r._engine_cache, r._lik_engine_cache = keep[id(r)]
        st["seed"] = self.seed
        return st

    def set_state(self, state):
        import copy
        assert state["seed"] == self.seed, "state of a chain with another seed"
        for k in self._STATE_ATTRS:
            if k in state:
                setattr(self, k, copy.deepcopy(state[k]))
        for r in self.regressions:
            r._engine_cache = r._lik_engine_cache = None
        self._st = None                          # the restored regressions share copies of the state arrays: link the model to those
        r0 = self.regressions[0]
        if r0._store is not None and all(r._store is not None and r._store[0] is r0._store[0] for r in self.regressions):
            self._st = r0._store[:3]
        self._adopt_state()
        self._draws_ahead = None
        self._fresh_stats = self._stats_kept = None

    def plot(self, *args, **kwargs):
        raise NotImplementedError("plotting is outside the scope of the MI355X hot path (SURVEY.md section 2, row 8)")


class _LazyX(object):
    """stands for the device-resident design matrix in data_list; materialises on the host only when indexed"""

    def __init__(self, engine, i):
        self.engine, self.i = engine, i

    def __array__(self, dtype=None, copy=None):
        return self.engine.design_matrix(self.i)

    @property
    def shape(self):
        ds = self.engine.datasets[self.i]
        return (ds.T, self.engine.N, self.engine.B)


class HierarchicalNonlinearAutoregressiveModel(NonlinearAutoregressiveModel):
    """(models.py:204-236) adds the network prior over the regressions' hyper-parameters."""

    def __init__(self, N, network, regressions, basis=None, B=10, **kw):
        super(HierarchicalNonlinearAutoregressiveModel, self).__init__(N, regressions, basis=basis, B=B, **kw)
        self.network = network

    def resample_model(self):
        self._fresh_stats = None
        super(HierarchicalNonlinearAutoregressiveModel, self).resample_model()
        stats, self._fresh_stats = self._fresh_stats, None     # the rows' statistics of exactly the state the regressions' sweep just stored
        self.resample_network(_row_stats=stats)

    def _network_stats(self, row_stats):
        """((n, sum w, sum w w') off the diagonal, the same on it) from the per-row statistics every rank holds after the gather: N rows of
        1 + B + B^2 doubles added up in neuron order (the same bits whatever the sharding), and the N self-connections read off (A, W)"""
        A, W, _ = self._adopt_state()
        B = self.B
        tot = np.sum(np.asarray(row_stats, dtype=np.float64), axis=0)
        r = np.arange(self.N)
        d = A[r, r]
        Wd = W[r, r, :] * d[:, None]
        return (tot[0], tot[1:1 + B], tot[1 + B:].reshape(B, B)), (float(d.sum()), Wd.sum(axis=0), Wd.T.dot(Wd))

    def resample_network(self, _row_stats=None):
        """(models.py:228-236).  Every rank holds the full (A, W) after the all_gather and draws the same network
        parameters from an identically seeded host generator; the push evaluates mu_W / sigma_W / rho once
        instead of once per row.  Inside resample_model() the network's sufficient statistics come from the per-row statistics the ranks
        exchanged with their rows (_row_stats; O(N B^2) on every rank); called on its own -- the state may have been edited since the last
        sweep -- the network walks (A, W) as the reference does."""
        net = self.network
        if hasattr(net, "_set_rng"):
            # the package's own NIW networks draw from a generator keyed by (seed, sweep) -- identical on every rank, and 20 us to make where
            # re-seeding NumPy's global Mersenne twister and putting it back costs 110 (a sixth of a sweep at BASELINE configs[0])
            net._set_rng(np.random.RandomState(np.random.Philox(key=self.seed & (2 ** 64 - 1), counter=[self.sweeps_done, 0, 2, 0])))
            try:
                if _row_stats is not None:
                    net.resample(self._adopt_state()[:2], stats=self._network_stats(_row_stats))
                else:
                    net.resample(self._adopt_state()[:2])         # (the network only reads them)
            finally:
                net._set_rng(None)
        else:
            state = npr.get_state()
            npr.seed((self.seed * 1000003 + self.sweeps_done) % (2 ** 32))      # a user's network class draws from NumPy's global generator: identical on every rank
            try:
                net.resample(self._adopt_state()[:2])
            finally:
                npr.set_state(state)
        if hasattr(net, "weight_blocks"):
            # one shared block (+ one for self-connections): pushed as such -- the (N, N, B, B) expansion of sigma_W is 210 MB at N = 1024,
            # a fifth of a second of host time per sweep on every rank, and nothing on the hot path reads it
            mu_off, S_off, mu_self, S_self = net.weight_blocks()
            rho = net.rho
            for n, reg in enumerate(self.regressions):
                reg._push_block_prior(mu_off, S_off, mu_self, S_self, n, rho[n])
            self._cache_block_hypers(mu_off, S_off, mu_self, S_self, rho)
            return
        sigma, mu, rho = net.sigma_W, net.mu_W, net.rho
        for n, reg in enumerate(self.regressions):
            reg._set("_S_w", np.array(sigma[n]))      # (copies: the regression owns its hyper-parameters, as after the reference's setters)
            reg._set("_mu_w", np.array(mu[n]))
            reg._set("_rho", np.array(rho[n]))
        self._cache_pushed_hypers(sigma, mu, rho)

    def _cache_block_hypers(self, mu_off, S_off, mu_self, S_self, rho):
        """natural-parameter terms of the shard's rows for a (shared block, self block) push: two table entries and the labels"""
        from .engine import prior_terms, BlockPrior
        n0, n1, N, B = self.n0, self.n1, self.N, self.B
        regs = self.regressions[n0:n1]
        nl = n1 - n0
        # (which block a pair takes never changes: the label table is built once per shard, not once per sweep)
        lkey = (n0, n1, N, S_self is not None)
        lc = getattr(self, "_label_cache", None)
        if lc is None or lc[0] != lkey:
            label = np.zeros((nl, N), dtype=np.int32)
            if S_self is not None:
                label[np.arange(nl), np.arange(n0, n1)] = 1
            lc = self._label_cache = (lkey, label)
        label = lc[1]
        S_u, mu_u = [S_off], [mu_off]
        if S_self is not None:
            S_u.append(S_self)
            mu_u.append(mu_self)
        Jw_u, hw_u, _, _, c0_u = prior_terms(np.array(S_u)[None], np.array(mu_u)[None], np.ones(1), np.zeros(1))
        prior = BlockPrior(Jw_u[0], hw_u[0], c0_u[0], label)
        S_b = np.array([r._S_b[0, 0] for r in regs])
        mu_b = np.array([r._mu_b[0] for r in regs])
        Jb = 1.0 / S_b
        self._hyper_cache = (tuple(r._hyp_version for r in regs), (np.array(rho[n0:n1], dtype=float), prior, None, Jb, Jb * mu_b, None))

    def _cache_pushed_hypers(self, sigma, mu, rho):
        """natural-parameter terms of the shard's rows for the hyper-parameters just pushed.  Network priors produce a
        handful of distinct (mu, Sigma) blocks (off-diagonal / self-connection), so the B x B inversions are done once per
        distinct block and scattered, instead of once per (n, m) pair."""
        from .engine import prior_terms
        n0, n1, N, B = self.n0, self.n1, self.N, self.B
        regs = self.regressions[n0:n1]
        sig = sigma[n0:n1].reshape(-1, B * B)
        mus = mu[n0:n1].reshape(-1, B)
        key = np.concatenate((sig, mus), axis=1)
        # label every row with the index of its distinct block: two frequent candidates first, np.unique for the rest
        label = np.full(key.shape[0], -1, dtype=np.int64)
        blocks = []
        for _ in range(2):
            todo = np.nonzero(label < 0)[0]
            if todo.size == 0:
                break
            cand = key[todo[0]]
            label[todo[np.all(key[todo] == cand, axis=1)]] = len(blocks)
            blocks.append(cand)
        todo = np.nonzero(label < 0)[0]
        if todo.size > 4 * (n1 - n0) + 8:                 # not a structured prior: let resample_regressions do the dense path
            self._hyper_cache = None
            return
        if todo.size:
            u_rest, inv = np.unique(key[todo], axis=0, return_inverse=True)
            label[todo] = len(blocks) + inv.reshape(-1)
            blocks.extend(list(u_rest))
        u = np.array(blocks)
        idx = label
        Jw_u, hw_u, _, _, c0_u = prior_terms(u[:, :B * B].reshape(1, -1, B, B), u[:, B * B:].reshape(1, -1, B), np.ones(1), np.zeros(1))
        nl = n1 - n0
        from .engine import BlockPrior
        prior = BlockPrior(Jw_u[0], hw_u[0], c0_u[0], idx.reshape(nl, N))      # tables + labels: never expanded to (nl, N, B, B)
        S_b = np.array([r._S_b[0, 0] for r in regs])
        mu_b = np.array([r._mu_b[0] for r in regs])
        Jb = 1.0 / S_b
        self._hyper_cache = (tuple(r._hyp_version for r in regs), (np.array(rho[n0:n1], dtype=float), prior, None, Jb, Jb * mu_b, None))


GLM = NonlinearAutoregressiveModel
NetworkGLM = HierarchicalNonlinearAutoregressiveModel


class _DefaultMixin(object):
    _network_class = None
    _regression_class = None

    def __init__(self, N, B=10, basis=None, network=None, network_kwargs=None, regressions=None, regression_kwargs=None, **kw):
        """(models.py:246-267)"""
        B = B if basis is None else basis.shape[1]
        if network is None:
            network = self._network_class(N, B, **(network_kwargs or {}))
        if regressions is None:
            regressions = [self._regression_class(N, B, **(regression_kwargs or {})) for _ in range(N)]
        super(_DefaultMixin, self).__init__(N, network, regressions, B=B, basis=basis, **kw)


class GaussianGLM(_DefaultMixin, NetworkGLM):
    """(models.py:270-272)"""
    _network_class = _networks.NIWDenseNetwork
    _regression_class = _regression.GaussianRegression


class SparseGaussianGLM(_DefaultMixin, NetworkGLM):
    """(models.py:274-276)"""
    _network_class = _networks.NIWSparseNetwork
    _regression_class = _regression.SparseGaussianRegression


class BernoulliGLM(_DefaultMixin, NetworkGLM):
    _network_class = _networks.NIWDenseNetwork
    _regression_class = _regression.BernoulliRegression


class SparseBernoulliGLM(_DefaultMixin, NetworkGLM):
    _network_class = _networks.NIWSparseNetwork
    _regression_class = _regression.SparseBernoulliRegression


class NegativeBinomialGLM(_DefaultMixin, NetworkGLM):
    _network_class = _networks.NIWDenseNetwork
    _regression_class = _regression.NegativeBinomialRegression


class SparseNegativeBinomialGLM(_DefaultMixin, NetworkGLM):
    _network_class = _networks.NIWSparseNetwork
    _regression_class = _regression.SparseNegativeBinomialRegression
