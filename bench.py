#!/usr/bin/env python
"""Headline benchmark: full Gibbs sweeps/sec (SparseBernoulliGLM.resample_model) at N=1024, T=100k, B=5 on synthetic
spike trains, neurons sharded over --gpus MI355X (one process per GPU; launch with torch.distributed.run for N>1).

One JSON line on rank 0 (contract in the task statement): value = sweeps/s of the WHOLE model, `roofline` for the dominant
kernel (omega-weighted fp64 Gram, MFMA-bound), `cpu_baseline` = the oracle timed on this box's host cores on a bounded sample.
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
}
PEAK_HBM_GBS = 8000.0
PEAK_I8_MFMA_TOPS = 5000.0
PEAK_F64_MFMA_TFLOPS = 78.6   # 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz; v_mfma_f64_16x16x4_f64 = 64 cyc (tools/ubench2_f64.hip, measured)


def synth(N, B, T, L, seed=0):
    """i.i.d. Bernoulli(0.08) spikes, cosine basis / L, priors of examples/synthetic.py:40-42."""
    from pyglm_amd.utils.basis import cosine_basis
    rng = np.random.default_rng(seed)
    basis = cosine_basis(B, L=L) / L
    Y = (rng.random((T, N)) < 0.08).astype(np.float64)
    return basis, Y


def cpu_baseline(model, cfg, budget_s=20.0):
    """The oracle (NumPy/BLAS restatement of the reference, oracle/pyglm_oracle.py) on this host's cores, for ONE neuron of the
    same workload, on a bounded sample: T_s time bins of the Gram/activation (cost linear in T) and P of the N flip proposals
    (each proposal costs the same), extrapolated to one full sweep of N neurons.  Labelled as extrapolated."""
    from oracle import pyglm_oracle as orc
    N, B, T = cfg["N"], cfg["B"], cfg["T"]
    D = N * B
    eng = model.engine
    Ts = int(min(T, max(2000, 4e10 / (2.0 * D * D))))         # ~4e10 flop of dgemm
    P = min(N, 8)
    X = eng.datasets[0].X[:Ts, :D].cpu().numpy()
    om = eng.datasets[0].OK[:Ts, 0].cpu().numpy()
    y = model.data_list[0][1][:Ts, model.n0].astype(float)
    r0 = model.regressions[model.n0]
    r = orc.Regression(N, B, rho=r0.rho.copy(), mu_w=r0.mu_w.copy(), S_w=r0.S_w.copy(), mu_b=r0.mu_b.copy(), S_b=r0.S_b.copy())
    r.a, r.W, r.b = r0.a.copy(), r0.W.copy(), r0.b.copy()
    rng = np.random.default_rng(1)
    t0 = time.perf_counter()
    psi = r.activation(X)
    _ = orc.pg_draw(None, psi, 1, 0)
    t_act = time.perf_counter() - t0
    t0 = time.perf_counter()
    Jp, hp = r.prior_stats()
    Jl, hl = r.lkhd_stats([(X, y)], [om])
    t_gram = time.perf_counter() - t0
    Jq, hq = Jp + Jl * (T / Ts), hp + hl * (T / Ts)
    t0 = time.perf_counter()
    r.collapsed_resample_a(Jp, hp, Jq, hq, rng.permutation(N)[:P], rng.random(P))
    t_flip = time.perf_counter() - t0
    t0 = time.perf_counter()
    r.resample_W(Jq, hq, rng.standard_normal(D + 1))
    t_w = time.perf_counter() - t0
    t_neuron = (t_act + t_gram) * (T / Ts) + t_flip * (N / P) + t_w
    return dict(value=1.0 / (N * t_neuron), unit="sweeps/s", cores=os.cpu_count(), kind="port",
                sample="oracle (NumPy/OpenBLAS all cores + OpenMP PG), 1 of %d neurons, T_s=%d of %d bins for activation/PG/Gram "
                       "(x%.0f), %d of %d flip proposals (x%.0f), full weight draw; extrapolated to N neurons; "
                       "per-neuron s: act+pg %.3f gram %.3f flips %.3f weights %.3f" %
                       (N, Ts, T, T / Ts, P, N, N / P, t_act * T / Ts, t_gram * T / Ts, t_flip * N / P, t_w))


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
    ap.add_argument("--no-fp64-compare", action="store_true",
                    help="skip the extra (untimed for `value`) sweep with the fp64-MFMA Gram that fills the fp64_gram_path object")
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    for k in ("N", "T", "B"):
        if getattr(args, k):
            cfg[k] = getattr(args, k)

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node N)" % (args.gpus, world)
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

    from pyglm_amd.models import SparseBernoulliGLM, NegativeBinomialGLM, SparseGaussianGLM
    N, B, T, L = cfg["N"], cfg["B"], cfg["T"], cfg["L"]
    np.random.seed(0)
    basis, Y = synth(N, B, T, L)
    t_setup = time.perf_counter()
    ekw = dict(batch=args.batch) if args.batch else {}
    if args.gram != "auto":
        ekw["gram"] = args.gram
    ekw = ekw or None
    if cfg.get("obs") == "negbin":
        Y = np.random.default_rng(1).negative_binomial(2, 0.85, size=(T, N)).astype(np.float64)     # counts, mean 0.35
        model = NegativeBinomialGLM(N, basis=basis, regression_kwargs=dict(S_w=1.0, mu_b=-2.0, xi=2.0), seed=0, engine_kwargs=ekw)
    elif cfg.get("obs") == "gaussian":
        rg = np.random.default_rng(1)
        Y = rg.standard_normal((T, N)) + 2.0 * Y                                                      # real-valued activity
        model = SparseGaussianGLM(N, basis=basis, regression_kwargs=dict(S_w=10.0, a_0=2.0, b_0=2.0), seed=0, engine_kwargs=ekw)
    else:
        model = SparseBernoulliGLM(N, basis=basis, regression_kwargs=dict(S_w=10.0, mu_b=-2.0), seed=0, engine_kwargs=ekw)
    model.add_data(Y)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        model.resample_model()
    model.engine.profile = True
    model.engine.collect_timings()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.resample_model()
    barrier()
    dt = time.perf_counter() - t0
    stages = model.engine.collect_timings()
    model.engine.profile = False
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    barrier()
    t_ll = time.perf_counter()
    ll = model.log_likelihood()
    barrier()
    t_ll = time.perf_counter() - t_ll          # log_likelihood() on its own: activation + fused reduction (+ scalar all-reduce)
    # the same sampler with the Gram forced onto the fp64-MFMA kernel, one more sweep of the same chain (not part of `value`)
    cmp64 = None
    took_i8 = any(ds.int8 for ds in model.engine.datasets)
    if use_dist:      # the extra sweep contains collectives: every rank runs it, or none (a rank short of memory may have stayed on fp64)
        t = torch.tensor([float(took_i8)], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        took_i8 = bool(t.item())
    if took_i8 and not args.no_fp64_compare:
        for ds in model.engine.datasets:
            ds.int8 = False
        model.engine.profile = True
        model.engine.collect_timings()
        barrier()
        t1 = time.perf_counter()
        model.resample_model()
        barrier()
        dt64 = time.perf_counter() - t1
        st64 = model.engine.collect_timings()
        model.engine.profile = False
        if use_dist:
            t = torch.tensor([dt64], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt64 = float(t.item())
        g64 = st64.get("gram", dict(ms=0.0, calls=0, work=0.0))
        a64 = (g64["work"] / (g64["ms"] * 1e-3) * 1e-12) if g64["ms"] > 0 else None
        cmp64 = {"value": 1.0 / dt64, "unit": "sweeps/s", "ms_per_step": dt64 * 1e3, "steps": 1,
                 "note": "one further sweep of the same chain with the Gram on the fp64-MFMA kernel (engine gram='fp64')",
                 "roofline": {"bound": "mfma", "kernel": "gemm_tn_f64_persistent<2,2,2,weighted,3-stage,DMA> (omega-weighted Gram)", "achieved": a64,
                              "peak": PEAK_F64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": (a64 / PEAK_F64_MFMA_TFLOPS) if a64 else None,
                              "launches": g64["calls"], "avg_launch_ms": (g64["ms"] / g64["calls"]) if g64["calls"] else None}}

    if rank == 0:
        g = stages.get("gram", dict(ms=0.0, calls=0, work=0.0))
        achieved = (g["work"] / (g["ms"] * 1e-3) * 1e-12) if g["ms"] > 0 else None
        # HBM bytes per Gram launch from the committed rocprofv3 --pmc passes (profiles/gram_pmc.json); only valid for the
        # workload and launch geometry they were collected on (cfg3, one GPU)
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "gram_pmc.json")
        pmc_ok = os.path.exists(pmc) and args.config == "cfg3" and world == 1 and (N, B, T) == (1024, 5, 100000)
        if pmc_ok:
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Gibbs sweeps/sec (full resample_model)", "value": args.steps / dt, "unit": "sweeps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s N=%d B=%d L=%d T=%d, synthetic i.i.d. %s, neurons sharded over %d GPU(s)"
                                   % (type(model).__name__, N, B, L, T, {"negbin": "NB(2, 0.85) counts", "gaussian": "N(2 s, 1) activity"}.get(cfg.get("obs"), "Bernoulli(0.08) spikes"), world), "N": N, "B": B, "T": T, "parallelism": "neuron-shard x%d" % world,
                       "neurons_per_batch": model.engine.nb},
            "roofline": {"bound": "mfma", "kernel": "gemm_tn_f64_persistent<2,2,2,weighted,3-stage,DMA> (omega-weighted Gram)", "achieved": achieved,
                         "peak": PEAK_F64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": (achieved / PEAK_F64_MFMA_TFLOPS) if achieved else None,
                         "traffic": traffic, "launches": g["calls"], "avg_launch_ms": (g["ms"] / g["calls"]) if g["calls"] else None},
            "stages_ms_rank0": {k: round(v["ms"], 3) for k, v in stages.items()},
            "setup_s": round(t_setup, 2), "log_likelihood_after": ll, "log_likelihood_ms": round(t_ll * 1e3, 2),
        }
        gi = stages.get("gram.int8")
        if gi and gi["ms"] > g["ms"]:
            # the Gram went through the integer matrix cores: 15 int8 residue-plane products per neuron, T D (D+1) operations each (lower
            # triangle, 2 per multiply-add); HIP events around i8_gram_kernel alone (conversion = stage gram.planes, CRT = gram.crt).
            # peak: dense i8 MFMA 5 POP/s (2x the bf16 figure of MI355X_MICROARCH.md; 4.92 measured at 2.39 GHz with constant operands;
            # with random operand bytes the package power limit holds the bare MFMA loop to 3.45 POP/s at 1.77 GHz: tools/ubench_i8.hip)
            ach = 15.0 * gi["work"] / (gi["ms"] * 1e-3) * 1e-12
            tr = None
            if pmc_ok:
                try:
                    tr = json.load(open(pmc)).get("int8", {}).get("hbm_bytes_per_launch")
                except Exception:
                    tr = None
            out["dtype"] = "f64 (likelihood Gram: exact integer arithmetic on i8 residue planes, CRT back to f64)"
            out["roofline"] = {"bound": "mfma", "kernel": "i8_gram_kernel (v_mfma_i32_16x16x64_i8; 15 residue planes of %d neurons per launch)" % (model.engine._i8_scratch[2] if model.engine._i8_scratch else 0),
                               "achieved": ach, "peak": PEAK_I8_MFMA_TOPS, "unit": "TOP/s", "frac": ach / PEAK_I8_MFMA_TOPS, "traffic": tr,
                               "launches": gi["calls"], "avg_launch_ms": gi["ms"] / gi["calls"],
                               "fp64_equivalent_tflops": gi["work"] / (gi["ms"] * 1e-3) * 1e-12,
                               "power_limited_mfma_only_tops": 4000.0}
            if cmp64:
                out["fp64_gram_path"] = cmp64
        if cfg.get("obs") == "gaussian":
            # no per-neuron Gram here: X'X is formed once in add_data and only scaled per sweep (HBM-bound streaming store)
            gs = stages.get("gram_scale", dict(ms=0.0, calls=0, work=0.0))
            ach = gs["work"] / (gs["ms"] * 1e-3) * 1e-9 if gs["ms"] > 0 else None
            out["roofline"] = {"bound": "hbm", "kernel": "scaled_gram_kernel (J[n] = X'X / eta_n, lower triangle)", "achieved": ach,
                               "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS if ach else None, "traffic": None,
                               "launches": gs["calls"], "avg_launch_ms": gs["ms"] / gs["calls"] if gs["calls"] else None}
        if not args.no_cpu_baseline and world == 1 and cfg.get("obs") is None:
            out["cpu_baseline"] = cpu_baseline(model, cfg)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
