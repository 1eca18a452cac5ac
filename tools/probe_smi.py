#!/usr/bin/env python
"""HBM-side activity of the integer product kernel from the SMU's own counters (amdsmi gpu_metrics: average_umc_activity, mem_activity_acc),
which rocprofv3 does not expose on this pool (no MALL / HBM-side PMC: FETCH_SIZE counts Infinity-Cache hits too).

The memory-controller activity is a percentage; it is calibrated here against kernels whose HBM traffic is known -- a device-to-device copy
(reads + writes), a read-only reduction and the residue-plane conversion (store-dominated, 57 GB per group) -- and then sampled over a loop of
product launches on the bench's own data (one group of 8 neurons at the cfg3 shape, as tools/probe_i8_real.py).

    python tools/probe_smi.py [seconds per leg = 4] > gpurun_out/smi.json
"""
import ctypes
import json
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import os                                           # noqa: E402
import pyglm_amd._lib as _l                         # noqa: E402
if os.environ.get("PGL_PROBE_LIB"):                 # "ab": lib/libpyglm_hip_ab.so, whose tuning knobs read the environment (PGL_I8_KPARTS ...)
    _l.LIB_PATH = _l.LIB_PATH.replace("libpyglm_hip.so", "libpyglm_hip_%s.so" % os.environ["PGL_PROBE_LIB"])
from pyglm_amd.engine import GibbsEngine            # noqa: E402
from pyglm_amd._lib import call, ptr                # noqa: E402
from pyglm_amd.utils.basis import cosine_basis      # noqa: E402

import amdsmi                                       # noqa: E402

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
amdsmi.amdsmi_init()
H = amdsmi.amdsmi_get_processor_handles()[0]


class Sampler(object):
    FIELDS = ("average_umc_activity", "average_gfx_activity", "mem_activity_acc", "gfx_activity_acc", "accumulation_counter", "current_uclk",
              "average_uclk_frequency", "current_socket_power", "average_socket_power", "energy_accumulator", "firmware_timestamp", "current_gfxclk")

    def __init__(self, period=0.05):
        self.period, self.rows, self._stop = period, [], False

    def _loop(self):
        while not self._stop:
            try:
                m = amdsmi.amdsmi_get_gpu_metrics_info(H)
                a = amdsmi.amdsmi_get_gpu_activity(H)
                self.rows.append((time.perf_counter(), {k: m.get(k) for k in self.FIELDS}, a))
            except Exception as e:          # noqa: BLE001
                self.rows.append((time.perf_counter(), {"error": repr(e)}, {}))
                return
            time.sleep(self.period)

    def __enter__(self):
        self.rows, self._stop = [], False
        self._t = threading.Thread(target=self._loop, daemon=True)
        self._t.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        self._t.join()

    def summary(self):
        rows = [r for r in self.rows if "error" not in r[1]]
        if len(rows) < 3:
            return {"error": self.rows[-1][1] if self.rows else "no samples"}
        rows = rows[len(rows) // 4:]            # the first quarter: the averages of the previous leg are still in the SMU's window
        out = {"samples": len(rows)}

        def num(v):
            return float(v) if isinstance(v, (int, float)) else None
        for k in ("average_umc_activity", "average_gfx_activity", "current_uclk", "average_uclk_frequency", "current_socket_power", "current_gfxclk"):
            vals = [num(r[1].get(k)) for r in rows]
            vals = [v for v in vals if v is not None]
            if vals:
                out[k + "_mean"] = float(np.mean(vals))
        vals = [num(r[2].get("umc_activity")) for r in rows]
        vals = [v for v in vals if v is not None]
        if vals:
            out["activity_umc_mean"] = float(np.mean(vals))
        f, l = rows[0][1], rows[-1][1]
        for k in ("mem_activity_acc", "gfx_activity_acc", "accumulation_counter", "energy_accumulator", "firmware_timestamp"):
            if num(f.get(k)) is not None and num(l.get(k)) is not None:
                out[k + "_delta"] = num(l[k]) - num(f[k])
        if out.get("accumulation_counter_delta"):
            out["mem_activity_acc_per_count"] = out.get("mem_activity_acc_delta", 0.0) / out["accumulation_counter_delta"]
        out["wall_s"] = rows[-1][0] - rows[0][0]
        return out


def leg(name, fn, bytes_per_call=None):
    fn()
    torch.cuda.synchronize()
    n = 0
    with Sampler() as s:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < SECS:
            for _ in range(4):
                fn()
            torch.cuda.synchronize()
            n += 4
        dt = time.perf_counter() - t0
    r = s.summary()
    r.update(leg=name, calls=n, ms_per_call=dt / n * 1e3)
    if bytes_per_call:
        r.update(known_bytes_per_call=bytes_per_call, known_gb_per_s=bytes_per_call * n / dt * 1e-9)
    print(json.dumps(r), flush=True)
    return r


static = {}
try:
    m = amdsmi.amdsmi_get_gpu_metrics_info(H)
    static = {k: m.get(k) for k in ("vram_max_bandwidth", "current_uclk", "num_partition")}
    static["vram"] = {k: str(v) for k, v in amdsmi.amdsmi_get_gpu_vram_info(H).items()}
except Exception as e:          # noqa: BLE001
    static["error"] = repr(e)
print(json.dumps({"static": static}), flush=True)

res = []
nbytes = 8 << 30
src = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
dst = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
src.random_(0, 255)
res.append(leg("idle", lambda: time.sleep(0.05)))
res.append(leg("copy 8 GiB d2d (read + write)", lambda: dst.copy_(src), 2 * nbytes))
s64 = src.view(torch.float64)
res.append(leg("sum of 8 GiB (read only)", lambda: torch.sum(s64), nbytes))
res.append(leg("fill 8 GiB (write only)", lambda: dst.fill_(3), nbytes))
del src, dst, s64
torch.cuda.empty_cache()

k, N, B, T, nl = 13, 1024, 5, 100000, 8
rng = np.random.default_rng(0)
Y = (rng.random((T, N)) < 0.08).astype(float)
eng = GibbsEngine(N, B, 0, nl, batch=nl, gram="int8", planes=k)
ds = eng.add_data(Y, basis=cosine_basis(B, L=100) / 100)
eng._upload_weights(np.ones((nl, N), bool), rng.standard_normal((nl, N, B)) * 0.05, np.full(nl, -2.0))
with torch.cuda.device(eng.dev):
    eng._psi_pass(True, 3, 0)
    D, Dp, ldj = eng.D, eng.Dp, eng.ldj
    _, _, G, PB, R, stat = eng._i8_scratch[:6]
    om = ctypes.c_void_p(ds.OK.data_ptr())
    ldo = 2 * eng.ldn
    for c0 in range(0, nl, 8):
        call("pgl_i8_colstats", ptr(ds.X), Dp, ctypes.c_void_p(om.value + 8 * c0), ldo, T, D, min(8, nl - c0), ptr(stat[0][c0:]), ptr(stat[1][c0:]), None)
    call("pgl_i8_scales", ptr(stat[0]), ptr(stat[1]), nl * D, T, k, ptr(stat[2]), None)
    Dq, Kp = (D + 319) // 320 * 320, (T + 63) // 64 * 64
    planes_bytes = k * nl * Dq * Kp + 8 * T * D
    res.append(leg("plane conversion (8 B read per element of X, 13 B stored per element and neuron)",
                   lambda: call("pgl_i8_planes_t", ptr(ds.Xt), ds.Tp, om, ldo, ptr(stat[2]), ptr(PB), T, D, nl, k, 0, None), planes_bytes))
    alg = k * Dq * Kp * (1 + nl) + k * nl * Dq * (Dq + 320) // 2
    r = leg("i8_gram_kernel (13 planes x 8 neurons per launch)", lambda: call("pgl_i8_gram", ptr(ds.PA), ptr(PB), ptr(R), T, D, nl, k, None))
    r["algorithmic_bytes_per_call"] = alg
    res.append(r)

# calibration: GB/s per per cent of memory-controller activity from the legs with known traffic, then the product kernel's HBM bytes per launch
cal = [(x["known_gb_per_s"], x.get("average_umc_activity_mean")) for x in res if x.get("known_gb_per_s") and x.get("average_umc_activity_mean")]
out = {"static": static, "legs": res}
if cal and res[-1].get("average_umc_activity_mean") is not None:
    per_pct = [g / u for g, u in cal if u > 0]
    g = res[-1]
    out["calibration_gb_per_s_per_percent"] = per_pct
    lo, hi = min(per_pct), max(per_pct)
    out["i8_gram_hbm_gb_per_s_range"] = [lo * g["average_umc_activity_mean"], hi * g["average_umc_activity_mean"]]
    out["i8_gram_hbm_bytes_per_launch_range"] = [lo * g["average_umc_activity_mean"] * 1e9 * g["ms_per_call"] * 1e-3,
                                                 hi * g["average_umc_activity_mean"] * 1e9 * g["ms_per_call"] * 1e-3]
print(json.dumps(out))
