#!/usr/bin/env python
"""Headline benchmark: full Gibbs sweeps/sec (SparseBernoulliGLM.resample_model) at N=1024, T=100k, B=5 on synthetic spike trains,
neurons sharded over --gpus MI355X, one process per GPU over RCCL.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no launcher around it (WORLD_SIZE unset) the script starts its N ranks itself, as fresh processes, before anything has
touched a GPU; under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` it is one of the ranks.

One JSON line on rank 0 (contract in the task statement): `value` = sweeps/s of the WHOLE model (max over ranks of the timed region),
`roofline` for the dominant kernel (the integer-MFMA Gram) from HIP events recorded on the launch stream, `hbm_stage` for the streaming
kernel next to it (residue-plane conversion), `per_rank` timings incl. the time inside collectives and the host-only share, `collective` =
backend, world size and every rank's device, `fixed_state` = the stage table of the sweep after the timed region, run twice from the same chain
state (`value` follows a chain that is still thinning out; this is one point of it),
`int8_vs_fp64` = one sweep from the same state on both Gram paths, `fp64_gram_path` = the same sampler with the Gram on the fp64 kernel,
`cpu_baseline` = the oracle timed on this box's host cores on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {   # BASELINE.json configs
    "cfg1": dict(N=4, B=1, T=10000, L=100),
    "cfg2": dict(N=128, B=5, T=50000, L=100),
    "cfg3": dict(N=1024, B=5, T=100000, L=100),
    "cfg4": dict(N=512, B=5, T=100000, L=100, obs="negbin"),      # NegativeBinomialGLM (dense prior): PG shape b = y + xi
    "cfg3g": dict(N=1024, B=5, T=100000, L=100, obs="gaussian"),  # not in BASELINE.json: SparseGaussianGLM at the cfg3 shape
    # configs[4]: dense-network prior (rho = 1: one 32 769-dim Cholesky per neuron), 8 x MI355X.  One neuron's residue planes are 86 GB, so the
    # integer Gram runs in time slices; a rank's shard is 512 neurons at ~1.3 s each, so the bench times `--neurons k` of them
    "cfg5": dict(N=4096, B=8, T=200000, L=100, dense=True, neurons=8),
}
PEAK_HBM_GBS = 8000.0
PEAK_I8_MFMA_TOPS = 5000.0    # dense i8 MFMA, 2x the bf16 figure of MI355X_MICROARCH.md (4.92 POP/s measured at 2.39 GHz on constant operands)
PEAK_F64_MFMA_TFLOPS = 78.6   # 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz; v_mfma_f64_16x16x4_f64 = 64 cyc (tools/ubench2_f64.hip, measured)
UBENCH_I8_MFMA_TOPS = 3944.0  # what MI355X_MICROARCH.md's i8 MFMA micro-benchmark sustains (the nominal peak is not reachable under the power cap)
TOP_STAGES = ("activation", "pg_loglik", "border", "gram", "gram.stats", "gram.planes", "gram.int8", "gram.crt", "gram_scale", "flips", "weights")
# stages that keep their HIP events inside the timed region: the top level (a few hundred event pairs per sweep).  The pieces of the flip
# stage (one pair per chunk / window and batch) are timed in one further sweep instead
TIMED_STAGES = TOP_STAGES + ("assemble", "pack")
TIMED_STAGES_SMALL = ("gram", "gram.int8", "gram.planes", "gram_scale")   # launch-bound sizes (a sweep of a few ms): the dominant kernel only


def synth(N, B, T, L, seed=0):
    """i.i.d. Bernoulli(0.08) spikes, cosine basis / L, priors of examples/synthetic.py:40-42."""
    from pyglm_amd.utils.basis import cosine_basis
    rng = np.random.default_rng(seed)
    basis = cosine_basis(B, L=L) / L
    Y = (rng.random((T, N)) < 0.08).astype(np.float64)
    return basis, Y


def cpu_baseline(model, cfg, neurons=2, proposals=32, budget_s=150.0):
    """The oracle (NumPy/BLAS restatement of the reference, oracle/pyglm_oracle.py + the C PG sampler) on this host's cores for `neurons`
    WHOLE neuron regressions of the same workload: activation, PG draw, the full-T likelihood statistics (the T x D
    temporary and the full dgemm of regression.py:251-252) and the weight draw are run in full; of the N collapsed-flip proposals (each costs
    the same two dense Choleskys, regression.py:343-378) the first `proposals` are run and scaled to N.  The sweep is N identical-cost
    neurons (models.py:169-171), so the per-neuron time is scaled by N / neurons.  Labelled as extrapolated."""
    from oracle import pyglm_oracle as orc
    N, B, T = cfg["N"], cfg["B"], cfg["T"]
    D = N * B
    eng = model.engine
    ds = eng.datasets[0]
    X = ds.X[:T, :D].cpu().numpy()
    rng = np.random.default_rng(1)
    t_wall = time.perf_counter()
    per = []
    for i in range(neurons):
        n = model.n0 + i
        r0 = model.regressions[n]
        rho, S_w, mu_w, S_b, mu_b = r0._hyper()
        r = orc.Regression(N, B, rho=rho.copy(), mu_w=mu_w.copy(), S_w=S_w.copy(), mu_b=mu_b.copy(), S_b=S_b.copy())
        r.a, r.W, r.b = r0.a.copy(), r0.W.copy(), r0.b.copy()
        y = model.data_list[0][1][:, n].astype(float)
        t0 = time.perf_counter()
        psi = r.activation(X)
        om = orc.pg_draw(None, psi, 1, n)
        t_act = time.perf_counter() - t0
        t0 = time.perf_counter()
        Jp, hp = r.prior_stats()
        Jl, hl = r.lkhd_stats([(X, y)], [om])
        t_gram = time.perf_counter() - t0
        Jq, hq = Jp + Jl, hp + hl
        P = min(N, proposals)
        if time.perf_counter() - t_wall > budget_s:
            P = min(P, 8)
        t0 = time.perf_counter()
        r.collapsed_resample_a(Jp, hp, Jq, hq, rng.permutation(N)[:P], rng.random(P))
        t_flip = time.perf_counter() - t0
        t0 = time.perf_counter()
        r.resample_W(Jq, hq, rng.standard_normal(D + 1))
        t_w = time.perf_counter() - t0
        per.append((t_act, t_gram, t_flip * (N / P), t_w, P))
        del Jp, Jl, Jq
    t_neuron = float(np.mean([p[0] + p[1] + p[2] + p[3] for p in per]))
    m = np.mean(np.array([p[:4] for p in per]), axis=0)
    return dict(value=1.0 / (N * t_neuron), unit="sweeps/s", cores=os.cpu_count(), kind="port",
                sample="oracle (NumPy/OpenBLAS on all cores + OpenMP PG sampler), %d whole neuron regressions of %d at full T = %d (activation, "
                       "PG draws, full likelihood statistics, weight draw), %s of %d flip proposals each (scaled x%.0f); extrapolated x%d to N "
                       "neurons; mean per-neuron seconds: act+pg %.2f, gram %.2f, flips %.1f (scaled), weights %.2f; measured in %.0f s"
                       % (neurons, N, T, "/".join(str(p[4]) for p in per), N, N / per[0][4], N // neurons, m[0], m[1], m[2], m[3],
                          time.perf_counter() - t_wall))


class PowerWatch(object):
    """samples the GPU's graphics clock and socket power (amdsmi, 5 Hz, ~0.5 ms per sample, on a host thread) while a region runs, so that
    the bench line itself says at which clock and power the roofline number was measured"""

    def __init__(self, pci_bus_id=None, period=0.2, index=None):
        """pci_bus_id: PCI address of the GPU being timed (torch.cuda.get_device_properties(i).pci_bus_id etc.): amdsmi enumerates the
        physical devices, the HIP index counts the visible ones -- the handle is matched by address, and the address goes into the record.
        index: the HIP device index, used only when no address matches (a torch build without the pci_* properties, another address
        format): the record then says `matched_by: index` and why the address did not match"""
        self.period, self.samples, self._stop, self._thread, self.err, self.bdf = period, [], False, None, None, None
        self.matched_by, self.note = None, None
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            self._h = None
            for h in hs:
                bdf = str(amdsmi.amdsmi_get_gpu_device_bdf(h)).lower()
                if pci_bus_id and bdf == pci_bus_id.lower():
                    self._h, self.bdf, self.matched_by = h, bdf, "pci"
            if self._h is None:
                seen = [str(amdsmi.amdsmi_get_gpu_device_bdf(h)).lower() for h in hs]
                self.note = "no amdsmi device with PCI address %r among %s" % (pci_bus_id, seen)
                if len(hs) == 1:
                    k = 0
                elif index is not None and 0 <= index < len(hs):
                    k = index
                else:
                    raise RuntimeError(self.note)
                self._h, self.bdf, self.matched_by = hs[k], seen[k], "only device" if len(hs) == 1 else "index"
            self._smi = amdsmi
        except Exception as e:          # no amdsmi / no permission / no match: the line then simply carries no clock record
            self._smi, self.err = None, repr(e)

    def _loop(self):
        smi = self._smi
        while not self._stop:
            try:
                p = smi.amdsmi_get_power_info(self._h)
                c = smi.amdsmi_get_clock_info(self._h, smi.AmdSmiClkType.GFX)
                w = p.get("current_socket_power", p.get("socket_power"))
                try:        # memory-controller (HBM-side) activity in per cent of the peak bandwidth; "N/A" on parts that do not report it
                    u = float(smi.amdsmi_get_gpu_activity(self._h)["umc_activity"])
                except Exception:
                    u = float("nan")
                self.samples.append((time.perf_counter(), float(w), float(c["clk"]), u))
            except Exception as e:
                self.err = repr(e)
                return
            time.sleep(self.period)

    def start(self):
        if self._smi is None:
            return self
        import threading
        self.samples, self._stop = [], False
        self._thread = threading.Thread(target=self._loop, daemon=True)
        self._thread.start()
        return self

    def stop(self, skip_seconds=0.0):
        """ends the sampling thread (joined BEFORE the samples are touched) and summarises; skip_seconds drops the samples of the region's first
        seconds (the SMU averages over a window)"""
        self._stop = True
        if self._thread is not None:
            self._thread.join()
            self._thread = None
        if skip_seconds > 0 and self.samples:
            t_first = self.samples[0][0]
            self.samples = [x for x in self.samples if x[0] - t_first >= skip_seconds] or self.samples
        if not self.samples:
            return {"available": False, "error": self.err}
        w = np.array([x[1] for x in self.samples])
        c = np.array([x[2] for x in self.samples])
        u = np.array([x[3] for x in self.samples])
        u = u[np.isfinite(u)]
        return {"available": True, "pci_bdf": self.bdf, "matched_by": self.matched_by, "match_note": self.note, "samples": len(w), "period_s": self.period, "sclk_mhz_mean": float(c.mean()), "sclk_mhz_min": float(c.min()),
                "sclk_mhz_max": float(c.max()), "power_w_mean": float(w.mean()), "power_w_max": float(w.max()),
                "umc_activity_pct_mean": float(u.mean()) if u.size else None,
                "source": "amdsmi (amdsmi_get_clock_info GFX, amdsmi_get_power_info current_socket_power, amdsmi_get_gpu_activity umc_activity) sampled "
                          "over the region"}


def box_ubench(seconds_i8=2.0, seconds_f64=1.0):
    """what the matrix cores of THIS box sustain right now (pgl_ubench_mfma: register-only MFMA loops on every CU, power limiter settled):
    i8 on random operand bytes in TOP/s, fp64 in TFLOP/s.  Boxes of the pool differ by ~5 % on power-limited kernels; the roofline fractions
    are quoted against these as well as against the nominal peaks, so that a 2 % kernel gain shows in a driver-run line."""
    import ctypes
    from pyglm_amd._lib import call
    out = {}
    for kind, key, secs, scale in ((0, "i8_tops", seconds_i8, 1e-12), (1, "f64_tflops", seconds_f64, 1e-12)):
        r, ms = ctypes.c_double(), ctypes.c_double()
        try:
            call("pgl_ubench_mfma", kind, secs, ctypes.byref(r), ctypes.byref(ms), None)
            out[key], out[key.split("_")[0] + "_ms"] = r.value * scale, ms.value
        except Exception as e:          # noqa: BLE001
            out[key], out["error"] = None, repr(e)
    out["note"] = ("pgl_ubench_mfma right before the timed region: v_mfma_i32_16x16x64_i8 on random operand bytes (8 x 4 blocks per wave, two waves "
                   "per SIMD) and v_mfma_f64_16x16x4_f64 (8 accumulators, one wave per SIMD), every CU, timed after a settling launch")
    return out


def hbm_side_probe(eng, watch, seconds=3.0):
    """HBM-side traffic of the integer product kernel, MEASURED IN THIS RUN from the SMU's memory-controller activity (amdsmi umc_activity, per cent
    of the peak bandwidth): rocprofv3 exposes no HBM / Infinity-Cache-side counter on this pool (FETCH_SIZE counts the cache's hits too).
    Two legs of `seconds` each, sampled at 20 Hz, the first second dropped (the SMU averages over a window): a 4 GiB fill whose traffic is known
    (calibration: GB/s per per cent), then product launches on the planes of the last group of the last sweep (its residues are scratch)."""
    import ctypes
    import torch
    from pyglm_amd._lib import call, ptr
    ds = eng.datasets[0]
    i8 = eng._i8_scratch
    if watch is None or watch._smi is None or not getattr(ds, "int8", False) or ds.PA is None or not i8 or i8[6]:
        return None
    G, PB, R = i8[2], i8[3], i8[4]

    def leg(fn):
        fn()
        torch.cuda.synchronize()
        watch.period = 0.05
        watch.start()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0, n = time.perf_counter(), 0
        e0.record()
        while time.perf_counter() - t0 < seconds:
            fn()
            fn()
            torch.cuda.synchronize()
            n += 2
        e1.record()
        torch.cuda.synchronize()
        rec = watch.stop(skip_seconds=1.0)
        watch.period = 0.2
        return e0.elapsed_time(e1) / n, rec
    try:
        buf = torch.empty(4 << 30, dtype=torch.uint8, device=eng.dev)
        ms_fill, rec_fill = leg(lambda: buf.fill_(1))
        del buf
        ms_k, rec_k = leg(lambda: call("pgl_i8_gram", ptr(ds.PA), ptr(PB), ptr(R), ds.T, eng.D, G, ds.planes, None))
        u_fill, u_k = rec_fill.get("umc_activity_pct_mean"), rec_k.get("umc_activity_pct_mean")
        if not u_fill or u_k is None:
            return {"available": False, "error": "no umc_activity samples"}
        gbs_fill = (4 << 30) / (ms_fill * 1e-3) * 1e-9
        per_pct = gbs_fill / u_fill                    # GB/s per per cent of memory-controller activity (81.9 = 8192 GB/s / 100 on MI355X)
        gbs_k = per_pct * u_k
        return {"available": True, "hbm_bytes_per_launch": gbs_k * 1e9 * ms_k * 1e-3, "hbm_gb_per_s": gbs_k, "umc_activity_pct": u_k,
                "launch_ms": ms_k, "neurons_per_launch": G, "power_w_mean": rec_k.get("power_w_mean"), "sclk_mhz_mean": rec_k.get("sclk_mhz_mean"),
                "calibration": {"kernel": "fill of 4 GiB", "known_gb_per_s": gbs_fill, "umc_activity_pct": u_fill, "gb_per_s_per_pct": per_pct},
                "source": "amdsmi umc_activity sampled at 20 Hz over %.0f s of back-to-back pgl_i8_gram launches on this run's own planes (the first "
                          "second dropped), scaled by a fill of known traffic measured the same way; the SMU's counter, not a rocprofv3 PMC" % seconds}
    except Exception as e:          # noqa: BLE001
        return {"available": False, "error": repr(e)}


def gather_cost_one_rank(model, eng, reps=3):
    """what the per-sweep exchange costs a rank apart from the wire: pack on the device -> all_gather_into_tensor over RCCL -> read-back of all N
    rows -> unpack, timed with the ONE rank a one-GPU box gives RCCL (a temporary 1-rank "nccl" group unless the run already has one).  With G
    ranks every rank still reads back and unpacks all N rows; what one rank cannot show is the transfer itself, bounded in `wire_note`."""
    import torch
    import torch.distributed as dist
    made = False
    try:
        if not dist.is_initialized():
            import socket
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                port = so.getsockname()[1]
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=eng.dev)
            made = True
        if dist.get_backend() != "nccl" or dist.get_world_size() != 1:
            return {"available": False, "error": "needs a one-rank nccl group"}
        ms = []
        for _ in range(reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.cuda.device(eng.dev):
                h = model._gather_start(eng.packed_state())
            rows = model._gather_finish(h)
            ms.append((time.perf_counter() - t0) * 1e3)
        assert rows[1].shape == (model.N, model.N * model.B)
        pay = int(h[1].numel())
        return {"available": True, "ms": float(np.median(ms[1:])), "ms_all": [round(x, 3) for x in ms], "payload_bytes": pay, "backend": "nccl", "world": 1,
                "what": "packed_state (device) + all_gather_into_tensor + device->host copy of all N rows + unpack, median of %d after one warm-up" % reps,
                "wire_note": "not in `ms`: moving (G-1)/G of the payload between GPUs -- %.2f ms if it all went over ONE xGMI link at 153 GB/s, less on the "
                             "node's point-to-point links used side by side" % (pay / 153e9 * 1e3)}
    except Exception as e:          # noqa: BLE001
        return {"available": False, "error": repr(e)}
    finally:
        if made:
            dist.destroy_process_group()


def scaling_proxy(model, eng, N, B, T, groups=(2, 4, 8)):
    """What a G-GPU run of the same model would take, from THIS GPU: the neurons shard by postsynaptic neuron (models.py:169-171,
    shard_bounds); EVERY one of the G shards is swept here on its own (engine.sweep(nrun, nfirst): pgl_sweep over neurons [lo, hi), batched as
    the owning rank would batch them, from the bench chain's current state, which is not advanced), because a sweep ends when the SLOWEST
    rank does -- flip and weight time follow each row's active-set size.  Then the replicated host-side network prior (timed) and the
    per-sweep gather as far as one rank can time it (gather_cost_one_rank).  projected = 1 / (max over shards + network + gather).
    NOT measured scaling: no second GPU was involved."""
    import torch
    from pyglm_amd.models import shard_bounds, state_row_layout
    inputs = model._sweep_inputs()
    state = model.get_state()
    stats = eng.row_stats().cpu().numpy()          # (of the state the last sweep left on the device: for the timing any state of this density does)
    t0 = time.perf_counter()
    model.resample_network(_row_stats=stats)       # as inside resample_model(): from the per-row statistics every rank holds after the gather
    t_net = time.perf_counter() - t0
    model.set_state(state)
    del state
    gather = gather_cost_one_rank(model, eng)
    t_gather = gather["ms"] * 1e-3 if gather.get("available") else 0.0
    rows = []
    for G in groups:
        shard_ms, tables = [], []
        for r in range(G):
            lo, hi = shard_bounds(N, G, r)
            eng.profile = TOP_STAGES
            eng.collect_timings()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.sweep(*inputs, model.seed, model.sweeps_done, nrun=hi - lo, nfirst=lo)
            torch.cuda.synchronize()
            shard_ms.append((time.perf_counter() - t0) * 1e3)
            st = eng.collect_timings()
            eng.profile = False
            tables.append({n: round(v["ms"], 1) for n, v in st.items() if n in TOP_STAGES})
        worst, best = int(np.argmax(shard_ms)), int(np.argmin(shard_ms))
        mx, mean = max(shard_ms), float(np.mean(shard_ms))
        rows.append({"gpus": G, "neurons_per_rank": N // G, "batches_per_rank": -(-(N // G) // eng.nb),
                     "shard_sweep_ms": [round(x, 1) for x in shard_ms], "max_ms": mx, "mean_ms": mean, "imbalance_max_over_mean": mx / mean,
                     "sum_ms": float(np.sum(shard_ms)),
                     "rank_sweep_plus_network_plus_gather_ms": mx + (t_net + t_gather) * 1e3,
                     "projected_sweeps_per_s": 1.0 / (mx * 1e-3 + t_net + t_gather),
                     "slowest_shard": {"rank": worst, "stages_ms": tables[worst]}, "fastest_shard": {"rank": best, "stages_ms": tables[best]}})
    return {"note": "projection, not measured scaling: every shard of a G-rank job swept on THIS GPU from the bench chain's current state (one sweep "
                    "each; the chain is not advanced); projected = 1 / (slowest shard + replicated host-side network prior + the per-sweep gather as one "
                    "rank can time it)",
            "host_network_prior_ms": t_net * 1e3, "gather": gather, "allgather_payload_bytes": int(N * state_row_layout(N, B)[-1]),
            "per_gpu_count": rows}


def self_launch(n):
    """`python bench.py --gpus N` without torchrun: N child processes of this same script, one per GPU, rendezvous on 127.0.0.1 at a free
    port.  Children inherit stdout/stderr (only rank 0 prints the JSON line).  Any rank failing ends the others; exit code = first failure."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # the host side of a rank (NumPy / BLAS for the priors and the random inputs) must not claim every core of the node n times over
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                sys.stderr.write("bench.py: rank %d exited with code %d; stopping the other ranks\n" % (r, code))
                for o in live:
                    procs[o].terminate()
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--N", type=int)
    ap.add_argument("--T", type=int)
    ap.add_argument("--B", type=int)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--gram", default="auto", choices=["auto", "fp64", "int8"],
                    help="likelihood Gram: auto (the engine's default: exact integer arithmetic on the int8 MFMA where that is faster, DESIGN.md "
                         "section 8c), or forced onto the fp64 MFMA kernel / the int8 path")
    ap.add_argument("--planes", type=int, default=None, help="residue planes (moduli) of the integer Gram (default: the engine's, 13)")
    ap.add_argument("--no-fp64-compare", action="store_true",
                    help="skip the extra sweeps (untimed for `value`) that fill int8_vs_fp64 and fp64_gram_path")
    ap.add_argument("--fp64-steps", type=int, default=1, help="sweeps timed with the Gram on the fp64-MFMA kernel for fp64_gram_path")
    ap.add_argument("--neurons", type=int, default=None,
                    help="time only the first k neurons of this rank's shard (a config whose shard is hours of work per sweep: cfg5); the line is "
                         "then seconds per neuron, labelled extrapolated")
    ap.add_argument("--xi", type=float, default=2.0,
                    help="shape of the negative-binomial model (cfg4): PG(y + xi, psi) draws; a fractional xi takes Windle's alternate sampler for the "
                         "fractional part (pgl_rng.h), an integer xi Devroye draws only")
    ap.add_argument("--no-fixed-state", action="store_true",
                    help="skip fixed_state (the sweep after the timed region run twice from the same chain state, once fully instrumented)")
    ap.add_argument("--no-box-ubench", action="store_true",
                    help="skip the ~4 s of register-only MFMA loops before the timed region (roofline.box_ubench_tops / frac_vs_this_box)")
    ap.add_argument("--no-hbm-probe", action="store_true",
                    help="skip hbm_side (umc_activity sampled over ~6 s of fill / product launches after the timed region)")
    ap.add_argument("--no-scaling-proxy", action="store_true",
                    help="skip scaling_proxy (one rank's shard of a 2 / 4 / 8-GPU run, timed on this GPU; N = 1 only)")
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    for k in ("N", "T", "B"):
        if getattr(args, k):
            cfg[k] = getattr(args, k)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: start one fresh process per GPU ourselves.  This parent has not touched the GPU (torch is not even
        # imported yet) and never will: it only waits for the ranks and passes rank 0's JSON line through.
        sys.exit(self_launch(args.gpus))

    # stdout carries ONE line, the JSON record: libraries that print there (RCCL's version banner at the first collective) go to stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
    # test hooks for boxes with a single GPU: PGL_BENCH_DEVICE puts every rank on one device, PGL_DIST_BACKEND=gloo replaces RCCL
    # (which refuses two ranks on one device) -- the launch path, sharding, gathers and the max-over-ranks timing are then the real ones
    if os.environ.get("PGL_BENCH_DEVICE"):
        local = int(os.environ["PGL_BENCH_DEVICE"])
    torch.cuda.set_device(local)
    use_dist = world > 1 or bool(os.environ.get("PGL_FORCE_DIST"))   # PGL_FORCE_DIST: exercise the RCCL path with one rank
    backend = os.environ.get("PGL_DIST_BACKEND", "nccl")
    if use_dist:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    cdev = "cuda" if backend == "nccl" else "cpu"

    def allmax(x):
        if not use_dist:
            return float(x)
        t = torch.tensor([float(x)], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    from pyglm_amd.models import SparseBernoulliGLM, NegativeBinomialGLM, SparseGaussianGLM, state_row_layout
    N, B, T, L = cfg["N"], cfg["B"], cfg["T"], cfg["L"]
    np.random.seed(0)
    basis, Y = synth(N, B, T, L)
    t_setup = time.perf_counter()
    ekw = {}
    if os.environ.get("PGL_BENCH_DEVICE") and world > 1:
        ekw["device_share"] = world                # ranks sharing one GPU (test hook): each engine budgets 1/world of its memory
    if args.batch:
        ekw["batch"] = args.batch
    if args.gram != "auto":
        ekw["gram"] = args.gram
    if args.planes:
        ekw["planes"] = args.planes
    ekw = ekw or None
    if cfg.get("obs") == "negbin":
        Y = np.random.default_rng(1).negative_binomial(2, 0.85, size=(T, N)).astype(np.float64)     # counts, mean 0.35
        model = NegativeBinomialGLM(N, basis=basis, regression_kwargs=dict(S_w=1.0, mu_b=-2.0, xi=args.xi), seed=0, engine_kwargs=ekw)
    elif cfg.get("obs") == "gaussian":
        rg = np.random.default_rng(1)
        Y = rg.standard_normal((T, N)) + 2.0 * Y                                                      # real-valued activity
        model = SparseGaussianGLM(N, basis=basis, regression_kwargs=dict(S_w=10.0, a_0=2.0, b_0=2.0), seed=0, engine_kwargs=ekw)
    else:
        mkw = {}
        k_neurons = args.neurons or cfg.get("neurons")
        if k_neurons:
            # k neurons of this rank's shard (rank r of `world` owns N / world neurons starting at lo)
            lo = rank * (N // max(world, 1))
            mkw["shard"] = (lo, min(N, lo + k_neurons))
        if cfg.get("dense"):
            from pyglm_amd.networks import NIWDenseNetwork
            mkw["network"] = NIWDenseNetwork(N, B)          # rho = 1: regression.py:153-155 -> deterministic rows, one dense draw per neuron
        model = SparseBernoulliGLM(N, basis=basis, regression_kwargs=dict(S_w=10.0, mu_b=-2.0), seed=0, engine_kwargs=ekw, **mkw)
    model.add_data(Y)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup
    eng = model.engine

    props = torch.cuda.get_device_properties(local)
    pci = "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0))
    watch = PowerWatch(pci, index=local) if rank == 0 else None
    # who took part in the collectives: backend, world size and every rank's device (what a multi-GPU record is checked against)
    me = {"rank": rank, "device": "cuda:%d" % local, "name": props.name, "pci": pci}
    ranks_info = [me]
    if use_dist:
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, me)
    collective = {"backend": (dist.get_backend() if use_dist else None), "world": world if use_dist else 1, "devices": ranks_info,
                  "per_sweep": "one all_gather_into_tensor of the shard's packed rows (W | b | eta or status | row statistics | a bytes: %d B per neuron) + none "
                               "on the data path; log_likelihood(): one all_reduce of the N per-neuron fp64 values" % state_row_layout(N, B)[-1]}

    # (sweeps of a second or more: ~550 event pairs are noise; configs[0] / configs[1] -- 3 / 50 ms per sweep -- keep the dominant kernel's only)
    timed_stages = TIMED_STAGES if float(N) * N * B * T >= 1e11 else TIMED_STAGES_SMALL

    def timed(steps, profile=timed_stages, watched=False):
        """`steps` sweeps bracketed by barrier + synchronize on both sides -> (max-over-ranks seconds, this rank's seconds, stage table,
        seconds this rank spent inside collectives).  profile: the stages whose launches are bracketed by HIP events on the launch
        stream (True: all of them -- a few thousand event pairs per sweep, kept out of the region `value` is measured on)"""
        eng.profile = profile
        eng.collect_timings()
        c0 = model.comm_seconds
        w0, o0, l0 = getattr(eng, "wait_seconds", 0.0), getattr(eng, "overlap_seconds", 0.0), getattr(eng, "launch_seconds", 0.0)
        barrier()
        if watched and watch:
            watch.start()
        t0 = time.perf_counter()
        for _ in range(steps):
            model.resample_model()
        barrier()
        mine = time.perf_counter() - t0
        pw = watch.stop() if (watched and watch) else None
        st = eng.collect_timings()
        eng.profile = False
        timed.launch = getattr(eng, "launch_seconds", 0.0) - l0
        timed.host_busy = (mine - (model.comm_seconds - c0) - (getattr(eng, "wait_seconds", 0.0) - w0) - (getattr(eng, "overlap_seconds", 0.0) - o0)
                           - timed.launch)
        return allmax(mine), mine, st, model.comm_seconds - c0, pw

    for _ in range(args.warmup):
        model.resample_model()
    ubench = box_ubench() if rank == 0 and not args.no_box_ubench else None
    dt, dt_mine, stages, comm_s, power = timed(args.steps, watched=True)
    host_busy, launch_s = timed.host_busy, timed.launch
    stages = {k_: dict(v_, ms=v_["ms"] / args.steps, calls=v_["calls"] / args.steps, work=v_["work"] / args.steps) for k_, v_ in stages.items()}

    # per-rank breakdown: wall time, time inside collectives, GPU time of the top-level stages, and what is left (host-only share)
    gpu_ms = sum(v["ms"] for k, v in stages.items() if k in TOP_STAGES)            # per sweep
    mine = {"rank": rank, "neurons": model.n1 - model.n0, "ms_per_step": dt_mine / args.steps * 1e3,
            "collectives_ms_per_step": comm_s / args.steps * 1e3, "gpu_stage_ms_per_step": gpu_ms,
            "host_only_ms_per_step": max(0.0, dt_mine / args.steps * 1e3 - gpu_ms),
            # wall time less the waits for the GPU (pgl_get_state), the waits inside collectives, the host work done while the GPU was busy
            # (the next sweep's random inputs) and the pgl_sweep call itself (its ~3 000 launches overlap the GPU's work; when several ranks share
            # one GPU the call also waits for room in the queue): what the host adds to a sweep on this rank
            "host_busy_ms_per_step": host_busy / args.steps * 1e3, "launch_call_ms_per_step": launch_s / args.steps * 1e3}
    per_rank = [mine]
    if use_dist:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    barrier()
    t_ll = time.perf_counter()
    ll = model.log_likelihood()                # the chain state right after the timed region (warmup + steps sweeps from the initial state)
    barrier()
    t_ll = time.perf_counter() - t_ll          # log_likelihood() on its own: activation + fused reduction (+ scalar all-reduce)

    # ---- two more sweeps from ONE chain state (not part of `value`): the state the timed region ended in, restored in between.  `value`
    # follows the chain, whose adjacency is still thinning out during the timed sweeps (the flip and weight stages follow its density); this
    # table is one state swept twice -- once with the top-level stage events only, once fully instrumented (the pieces of the flip stage) --
    # so the two agree on what they measure, and with the driver's fixed --warmup / --steps it is the same point of the same seeded chain
    # from round to round
    fixed, stages_full = None, {}
    if not args.no_fixed_state and not model._shard_override:
        keep_state = model.get_state()
        dens0 = float(model.adjacency.mean())
        d_a, _, st_a, _, _ = timed(1)
        dens1 = float(model.adjacency.mean())
        model.set_state(keep_state)
        d_b, _, st_b, _, _ = timed(1, profile=True)
        model.set_state(keep_state)
        del keep_state
        stages_full = st_b
        fixed = {"note": "the sweep after the timed region (chain sweep %d), run twice from the same state (top-level stage events only / fully "
                         "instrumented); not part of `value`, the chain is not advanced" % (args.warmup + args.steps),
                 "ms_per_step": d_a * 1e3, "ms_per_step_instrumented": d_b * 1e3, "adjacency_density_before": dens0, "adjacency_density_after": dens1,
                 "top_stages_ms": {k_: round(v_["ms"], 3) for k_, v_ in st_a.items()},
                 "stages_ms": {k_: round(v_["ms"], 3) for k_, v_ in st_b.items()},
                 "stages_ms_note": "flips.init / flips.decide / flips.apply are the pieces of `flips`"}

    # ---- HBM-side traffic of the product kernel from the SMU's memory-controller counter (rank 0; not part of `value`)
    hbm_side = hbm_side_probe(eng, watch) if rank == 0 and not args.no_hbm_probe else None

    # ---- what ONE rank of a 2 / 4 / 8-GPU run does, timed here (the driver's multi-GPU run is the measurement; this is the stand-in a
    # one-GPU box can give): the sweep of the first N/G neurons of this model from its current state (not advanced), twice each
    scaling = None
    if world == 1 and not args.no_scaling_proxy and not model._shard_override and cfg.get("obs") is None and N % 8 == 0 and N >= 64:
        scaling = scaling_proxy(model, eng, N, B, T)
        if fixed:
            # the proxy sweeps start from the chain state fixed_state was measured at (sweep warmup + steps), which is sparser than the states
            # `value` averages over: compare a rank's time with THIS one-GPU time, not with ms_per_step
            scaling["one_gpu_sweep_ms_same_state"] = fixed["ms_per_step"]

    # ---- the two Gram paths from the same state (consistency), then the fp64 path timed on its own
    cmp64 = consistency = None
    took_i8 = bool(allmax(float(any(ds.int8 for ds in eng.datasets))))      # every rank runs the extra sweeps, or none
    if took_i8 and not args.no_fp64_compare and not model._shard_override:
        state = model.get_state()
        model.resample_model()
        ll_i8 = model.log_likelihood()
        A_i8, W_i8 = model.adjacency, model.weights
        model.set_state(state)
        for ds in eng.datasets:
            ds.int8 = False
        d1, _, st64, _, _ = timed(1)
        ll_f64 = model.log_likelihood()
        A_f64, W_f64 = model.adjacency, model.weights
        consistency = {"note": "one sweep from the same chain state with the Gram on the integer matrix cores and on the fp64 kernel",
                       "log_likelihood_int8": ll_i8, "log_likelihood_fp64": ll_f64, "rel_diff": abs(ll_i8 - ll_f64) / abs(ll_f64),
                       "adjacency_equal": bool(np.array_equal(A_i8, A_f64)), "flips_differing": int((A_i8 != A_f64).sum()),
                       "max_abs_weight_diff": float(np.abs(W_i8 - W_f64).max())}
        del A_i8, W_i8, A_f64, W_f64, state
        more = max(0, args.fp64_steps - 1)
        d2, st2 = 0.0, {}
        if more:
            d2, _, st2, _, _ = timed(more)
        nst = 1 + more
        dt64 = d1 + d2
        g64 = {k: st64.get("gram", {}).get(k, 0.0) + st2.get("gram", {}).get(k, 0.0) for k in ("ms", "calls", "work")}
        a64 = (g64["work"] / (g64["ms"] * 1e-3) * 1e-12) if g64["ms"] > 0 else None
        cmp64 = {"value": nst / dt64, "unit": "sweeps/s", "ms_per_step": dt64 / nst * 1e3, "steps": nst,
                 "note": "further sweeps of the same chain with the Gram on the fp64-MFMA kernel (engine gram='fp64')",
                 "roofline": {"bound": "mfma", "kernel": "gemm_tn_f64_persistent<2,2,2,weighted,3-stage,DMA> (omega-weighted Gram)", "achieved": a64,
                              "peak": PEAK_F64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": (a64 / PEAK_F64_MFMA_TFLOPS) if a64 else None,
                              "launches": g64["calls"], "avg_launch_ms": (g64["ms"] / g64["calls"]) if g64["calls"] else None}}
        for ds in eng.datasets:
            ds.int8 = True

    if rank == 0:
        def stage(name):
            return stages.get(name, dict(ms=0.0, calls=0, work=0.0))
        g = stage("gram")
        achieved = (g["work"] / (g["ms"] * 1e-3) * 1e-12) if g["ms"] > 0 else None
        # HBM-side bytes per Gram launch come from separate rocprofv3 --pmc passes of this same command (profiles/gram_pmc.json, committed);
        # they are NOT measured in this run, and only quoted for the workload and launch geometry they were collected on
        pmc = {}
        pmc_path = os.path.join(ROOT, "profiles", "gram_pmc.json")
        if os.path.exists(pmc_path):
            try:
                pmc = json.load(open(pmc_path))
            except Exception:
                pmc = {}
        # ... and only while the kernel sources are the ones the counters were taken on: the files record pyglm_amd._lib.source_hash()
        from pyglm_amd._lib import source_hash
        src_hash = source_hash()
        pmc_ok = args.config == "cfg3" and world == 1 and (N, B, T) == (1024, 5, 100000)
        pmc_at_head = pmc.get("int8", {}).get("source_hash") == src_hash
        out = {
            "metric": "Gibbs sweeps/sec (full resample_model)", "value": args.steps / dt, "unit": "sweeps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s N=%d B=%d L=%d T=%d, synthetic i.i.d. %s, neurons sharded over %d GPU(s)"
                                   % (type(model).__name__, N, B, L, T, {"negbin": "NB(2, 0.85) counts", "gaussian": "N(2 s, 1) activity"}.get(cfg.get("obs"), "Bernoulli(0.08) spikes"), world), "N": N, "B": B, "T": T, "parallelism": "neuron-shard x%d" % world,
                       "neurons_per_batch": eng.nb},
            "roofline": {"bound": "mfma", "kernel": "gemm_tn_f64_persistent<2,2,2,weighted,3-stage,DMA> (omega-weighted Gram)", "achieved": achieved,
                         "peak": PEAK_F64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": (achieved / PEAK_F64_MFMA_TFLOPS) if achieved else None,
                         "traffic": None, "launches": g["calls"], "avg_launch_ms": (g["ms"] / g["calls"]) if g["calls"] else None,
                         "box_ubench_tflops": (ubench or {}).get("f64_tflops"),
                         "frac_vs_this_box": (achieved / ubench["f64_tflops"]) if achieved and ubench and ubench.get("f64_tflops") else None},
            "stages_ms_rank0": {k: round(v["ms"], 3) for k, v in stages.items()},
            "stages_note": "ms per sweep on rank 0 from HIP events inside the timed region (%s; averaged over its %d sweeps); the pieces of the "
                           "flip stage are in fixed_state.stages_ms (the next sweep of the chain, fully instrumented, not part of `value`)"
                           % ("the top-level stages" if timed_stages is TIMED_STAGES else "the dominant kernel's only at this size", args.steps),
            "fixed_state": fixed, "collective": collective,
            "per_rank": per_rank,
            "setup_s": round(t_setup, 2), "log_likelihood_after": ll, "log_likelihood_ms": round(t_ll * 1e3, 2),
        }
        gi = stage("gram.int8")
        if gi["ms"] > g["ms"]:
            # the Gram went through the integer matrix cores: `planes` int8 residue-plane products per neuron, T D (D+1) operations each
            # (lower triangle, 2 per multiply-add); HIP events around i8_gram_kernel alone on the launch stream (column statistics =
            # gram.stats, conversion = gram.planes, CRT = gram.crt).  With random operand bytes the package power limit holds the bare
            # MFMA loop to 4.0 POP/s (tools/ubench_i8.hip, profiles/archive/r01_power_limit.md)
            npl = eng.datasets[0].planes
            grp = eng._i8_scratch[2] if eng._i8_scratch else 0
            ach = npl * gi["work"] / (gi["ms"] * 1e-3) * 1e-12
            i8p = pmc.get("int8", {})
            tr_ok = pmc_ok and pmc_at_head and i8p.get("planes") == npl and i8p.get("group") == grp
            out["dtype"] = "f64 (likelihood Gram: exact integer arithmetic on %d i8 residue planes, CRT back to f64)" % npl
            out["roofline"] = {"bound": "mfma", "kernel": "i8_gram_kernel (v_mfma_i32_16x16x64_i8; %d residue planes of %d neurons per launch)" % (npl, grp),
                               "achieved": ach, "peak": PEAK_I8_MFMA_TOPS, "unit": "TOP/s", "frac": ach / PEAK_I8_MFMA_TOPS,
                               "traffic": i8p.get("hbm_bytes_per_launch") if tr_ok else None,
                               "traffic_source": ("profiles/gram_pmc.json: fabric-side FETCH_SIZE + WRITE_SIZE (Infinity-Cache hits included) from rocprofv3 --pmc "
                                                  "passes of this command, committed, taken at these kernel sources (source_hash %s); not measured in "
                                                  "this run -- the HBM side, measured in this run, is hbm_side" % src_hash) if tr_ok else
                                                 ("null: profiles/gram_pmc.json was taken at other kernel sources (%s, now %s) or another geometry"
                                                  % (i8p.get("source_hash"), src_hash)),
                               "hbm_side": hbm_side,
                               "launches": gi["calls"], "avg_launch_ms": gi["ms"] / gi["calls"], "planes": npl,
                               "algorithmic_ops_per_launch": npl * gi["work"] / gi["calls"],
                               "fp64_equivalent_tflops": gi["work"] / (gi["ms"] * 1e-3) * 1e-12,
                               "frac_vs_ubench": ach / UBENCH_I8_MFMA_TOPS, "ubench_tops": UBENCH_I8_MFMA_TOPS,
                               "box_ubench_tops": (ubench or {}).get("i8_tops"),
                               "frac_vs_this_box": (ach / ubench["i8_tops"]) if ubench and ubench.get("i8_tops") else None,
                               "box_ubench": ubench}
            busy_path = os.path.join(ROOT, "profiles", "i8_busy_pmc.json")
            if tr_ok and os.path.exists(busy_path):
                try:      # effective clock / MFMA-busy / wait counters of this kernel at this geometry: committed rocprofv3 --pmc passes, NOT measured in this run
                    busy = json.load(open(busy_path))
                    if busy.get("source_hash") == src_hash:
                        out["roofline"]["issue_counters"] = dict(busy, source="profiles/i8_busy_pmc.json (committed, taken at these kernel sources; not measured in this run)")
                except Exception:
                    pass
            gp = stage("gram.planes")
            if gp["ms"] > 0:
                # i8_planes_kernel: reads X once per group (8 B per element) and stores `planes` bytes per element and neuron
                byts = gp["work"] + 8.0 * T * (N * B) * gp["calls"]
                gbs = byts / (gp["ms"] * 1e-3) * 1e-9
                out["hbm_stage"] = {"bound": "hbm", "kernel": "i8_planes_kernel (fp64 -> %d residue planes, %d neurons per pass over X)" % (npl, grp),
                                    "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                                    "launches": gp["calls"], "avg_launch_ms": gp["ms"] / gp["calls"], "bytes_per_launch": byts / gp["calls"]}
            if consistency:
                out["int8_vs_fp64"] = consistency
            if cmp64:
                out["fp64_gram_path"] = cmp64
        out["roofline"]["clock_power"] = power            # graphics clock and socket power sampled while the timed region ran
        if scaling is not None:
            out["scaling_proxy"] = scaling
        if model._shard_override:
            k_ = model.n1 - model.n0
            out["shard_sample"] = {"neurons_timed": k_, "first_neuron": model.n0, "s_per_neuron": dt / args.steps / k_,
                                   "note": "`value` and ms_per_step are for these %d neurons only (one rank's shard of this config is %d neurons at "
                                           "%d GPUs: %.0f s per sweep per rank by extrapolation -- neurons are independent and of equal cost, "
                                           "models.py:169-171); the network prior (host, replicated on every rank) is in the timed region once per sweep"
                                           % (k_, N // 8, 8, dt / args.steps / k_ * (N // 8))}
        pl = stages.get("pg_loglik") or stages_full.get("pg_loglik") or dict(ms=0.0, work=0.0)
        if pl["ms"] > 0:
            out["stage_E_note"] = ("pg_loglik_kernel: %.1f M PG draws/s; bound by the sampler's transcendental VALU work (rejection loops), not by "
                                   "HBM (40 B per draw = %.2f TB/s)" % (pl["work"] / (pl["ms"] * 1e-3) * 1e-6, 40.0 * pl["work"] / (pl["ms"] * 1e-3) * 1e-12))
        if cfg.get("obs") == "gaussian":
            # no per-neuron Gram here: X'X is formed once in add_data and only scaled per sweep (HBM-bound streaming store)
            gs = stage("gram_scale")
            ach = gs["work"] / (gs["ms"] * 1e-3) * 1e-9 if gs["ms"] > 0 else None
            out["roofline"] = {"bound": "hbm", "kernel": "scaled_gram_kernel (J[n] = X'X / eta_n, lower triangle)", "achieved": ach,
                               "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS if ach else None, "traffic": None,
                               "launches": gs["calls"], "avg_launch_ms": gs["ms"] / gs["calls"] if gs["calls"] else None}
        if not args.no_cpu_baseline and world == 1 and cfg.get("obs") is None and not model._shard_override:
            out["cpu_baseline"] = cpu_baseline(model, cfg)
        else:
            out["cpu_baseline"] = None
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
