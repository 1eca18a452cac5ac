"""bench.py's own launch path: `python bench.py --gpus N` with no launcher around it starts N ranks itself (bench.py::self_launch).  On a
one-GPU box the ranks share the device (PGL_BENCH_DEVICE=0) and talk over gloo (RCCL refuses two ranks on one device); sharding
(models.py:169-171 by postsynaptic neuron), the all_gather of the rows, the scalar all_reduce and the max-over-ranks timing are the real
ones."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(gpus, *extra, **envkw):
    env = dict(os.environ, PGL_BENCH_DEVICE="0", PGL_DIST_BACKEND="gloo")
    env.update(envkw)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--config", "cfg2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-fp64-compare", "--no-scaling-proxy"] + list(extra)
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout (rank 0): %r" % p.stdout[-500:]
    return json.loads(lines[0])


def test_self_launch_two_ranks_equals_one_rank():
    one = _bench(1)
    two = _bench(2)
    for d, n in ((one, 1), (two, 2)):
        assert d["n_gpus"] == n and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "sweeps/s" and d["value"] > 0
        assert abs(d["value"] - 2 / (d["ms_per_step"] * 2e-3)) < 1e-9 * d["value"]
        assert len(d["per_rank"]) == n and sorted(r["rank"] for r in d["per_rank"]) == list(range(n))
        assert sum(r["neurons"] for r in d["per_rank"]) == 128
        # (this size goes through the integer Gram since gram="auto" compares the two paths' tile areas below 1024 columns: TOP/s)
        assert d["roofline"]["achieved"] > 0 and d["roofline"]["unit"] in ("TFLOP/s", "TOP/s")
    assert two["per_rank"][0]["collectives_ms_per_step"] > 0
    # the chain does not depend on the number of ranks (random inputs are keyed by the global neuron), and since round 6 neither does the
    # log-likelihood total in any bit: the ranks all_reduce the N per-neuron values and sum them in neuron order
    assert one["log_likelihood_after"] == two["log_likelihood_after"]
    assert one["collective"]["backend"] is None and two["collective"] == dict(two["collective"], backend="gloo", world=2)
    assert [d["rank"] for d in two["collective"]["devices"]] == [0, 1] and all(d["device"] == "cuda:0" for d in two["collective"]["devices"])
    fs = one["fixed_state"]
    assert fs["ms_per_step"] > 0 and fs["stages_ms"]["flips"] > 0 and 0 < fs["adjacency_density_after"] < 1
    # bench.py over RCCL with the one rank this box has (PGL_FORCE_DIST: the process group is initialised with backend "nccl", world_size 1):
    # the same line, the same chain, and the record of who took part
    rccl = _bench(1, PGL_FORCE_DIST="1", PGL_DIST_BACKEND="nccl")
    assert rccl["collective"]["backend"] == "nccl" and rccl["collective"]["world"] == 1 and len(rccl["collective"]["devices"]) == 1
    assert rccl["log_likelihood_after"] == one["log_likelihood_after"]
    assert rccl["per_rank"][0]["collectives_ms_per_step"] > 0


def test_bench_line_carries_the_box_calibration_and_the_hbm_side_probe():
    """round 5: `roofline.box_ubench_tops` / `frac_vs_this_box` from pgl_ubench_mfma (register-only MFMA loops, run right before the timed
    region), `roofline.hbm_side` from the SMU's memory-controller activity (amdsmi), measured in the run; the committed PMC counters are quoted
    only while their recorded source hash matches the kernel sources (this configuration is not the one they were taken on: null + reason);
    `per_rank` carries the exposed host share"""
    d = _bench(1, "--no-fixed-state")
    r = d["roofline"]
    assert 2000 < r["box_ubench_tops"] < 5100 and 0 < r["frac_vs_this_box"] < 1 and r["frac"] < r["frac_vs_this_box"]
    assert 40 < r["box_ubench"]["f64_tflops"] < 79
    assert r["traffic"] is None and "null" in r["traffic_source"]
    hb = r["hbm_side"]
    assert hb is not None
    if hb.get("available"):                      # (amdsmi reports umc_activity on MI355X; a box without it says why)
        assert hb["hbm_bytes_per_launch"] > 0 and 70 < hb["calibration"]["gb_per_s_per_pct"] < 95
    pr = d["per_rank"][0]
    assert pr["host_busy_ms_per_step"] < pr["ms_per_step"] and pr["launch_call_ms_per_step"] > 0


def test_scaling_proxy_times_every_shard():
    """round 6: scaling_proxy sweeps EVERY shard of a 2 / 4 / 8-rank job on this GPU (pgl_sweep_t.nfirst) and projects from the slowest one,
    plus the replicated network prior and the gather as one rank over RCCL can time it"""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "PGL_BENCH_DEVICE", "PGL_DIST_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "cfg2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-fp64-compare",
           "--no-box-ubench", "--no-hbm-probe"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    sp = d["scaling_proxy"]
    assert sp["gather"]["available"] and sp["gather"]["backend"] == "nccl" and sp["gather"]["ms"] > 0
    assert sp["gather"]["payload_bytes"] == sp["allgather_payload_bytes"] == 128 * (8 * 640 + 16 + 8 * (1 + 5 + 25) + 128)
    for row in sp["per_gpu_count"]:
        G = row["gpus"]
        assert len(row["shard_sweep_ms"]) == G and row["neurons_per_rank"] == 128 // G
        assert abs(row["max_ms"] - max(row["shard_sweep_ms"])) < 0.06 and 1.0 <= row["imbalance_max_over_mean"] < 3.0
        assert abs(row["projected_sweeps_per_s"] - 1e3 / row["rank_sweep_plus_network_plus_gather_ms"]) < 1e-6 * row["projected_sweeps_per_s"]
        assert row["slowest_shard"]["stages_ms"]["flips"] > 0
    # more ranks, shorter slowest shard; and the shards of a G-rank job together cost about one whole sweep
    mx = [r["max_ms"] for r in sp["per_gpu_count"]]
    assert mx[0] > mx[1] > mx[2]


def test_ubench_entry_point():
    """pgl_ubench_mfma (include/pyglm_hip.h): plausible rates for both instructions, argument checks"""
    import ctypes
    from pyglm_amd._lib import call, load, PglError
    r, ms = ctypes.c_double(), ctypes.c_double()
    call("pgl_ubench_mfma", 0, 0.2, ctypes.byref(r), ctypes.byref(ms), None)
    assert 2e15 < r.value < 5.2e15 and ms.value > 0
    call("pgl_ubench_mfma", 1, 0.2, ctypes.byref(r), None, None)
    assert 4e13 < r.value < 7.9e13
    with pytest.raises(PglError):
        call("pgl_ubench_mfma", 2, 0.2, ctypes.byref(r), None, None)
    assert b"argument check" in load().pgl_last_error()
