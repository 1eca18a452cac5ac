import ctypes, sys
sys.path.insert(0, ".")
import torch
from pyglm_amd._lib import call
for kind, secs in ((0, 2.0), (1, 1.0), (0, 2.0), (1, 1.0)):
    r, ms = ctypes.c_double(), ctypes.c_double()
    call("pgl_ubench_mfma", kind, secs, ctypes.byref(r), ctypes.byref(ms), None)
    print("ubench kind %d: %.1f T(FL)OP/s in %.0f ms" % (kind, r.value * 1e-12, ms.value))
