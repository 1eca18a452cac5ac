"""Reads the per-wave cycle accounting of the Gram kernel (ablation build, PGL_GRAM_ABLATE=64): share of a wave's loop time
spent in the DMA wait (s_waitcnt vmcnt) and at the mid-tile barrier.  Waves w and w+4 share a SIMD."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ["PGL_GRAM_ABLATE"] = "64"
import runpy
runpy.run_path(os.path.join(os.path.dirname(__file__), "probe_gram.py"), run_name="__main__")
from pyglm_amd import _lib
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(8 * 256 * 4, dtype=np.int64)
rc = lib.pgl_debug_read(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
d = buf.reshape(256, 8, 4).astype(float)
ok = d[:, 0, 3] > 0
d = d[ok]
tot, vm, bar = d[:, :, 0], d[:, :, 1], d[:, :, 2]
print("workgroups sampled", d.shape[0], "tiles", int(d[0, 0, 3]), "cycles/tile/wave %.0f" % (tot.mean() / d[0, 0, 3]))
for w in range(8):
    print("wave %d (SIMD %d): vmcnt wait %.3f  barrier wait %.3f of loop time" % (w, w % 4, (vm[:, w] / tot[:, w]).mean(), (bar[:, w] / tot[:, w]).mean()))
pair_min = np.minimum(bar[:, :4] + vm[:, :4], bar[:, 4:] + vm[:, 4:]) / tot[:, :4]
print("per SIMD: min over its two waves of (vmcnt+barrier wait) share = lower bound on pipe idle from sync: %s" % pair_min.mean(0).round(3))
