"""ctypes binding of libpyglm_hip.so (C ABI: include/pyglm_hip.h). Fails loudly when the library is absent."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpyglm_hip.so")

c_p, c_l, c_i, c_d, c_u64, c_u32, c_sz = (ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_double,
                                          ctypes.c_uint64, ctypes.c_uint32, ctypes.c_size_t)


class FlipState(ctypes.Structure):   # pgl_flip_t
    _fields_ = [("M", c_p), ("ldj", c_l), ("strideM", c_l), ("nb", c_i), ("N", c_i), ("B", c_i),
                ("perm", c_p), ("u", c_p), ("rho", c_p), ("c0", c_p), ("a", c_p), ("skip", c_p),
                ("d_idx", c_p), ("d_sign", c_p), ("d_cnt", c_p), ("batch_k", c_p), ("G", c_p), ("Lws", c_p),
                ("Ut", c_p), ("Wt", c_p), ("ldu", c_l), ("status", c_p), ("visit_order", c_i), ("logodds", c_p)]


class CholState(ctypes.Structure):   # pgl_chol_t
    _fields_ = [("J", c_p), ("ldj", c_l), ("strideJ", c_l), ("a", c_p), ("act", c_p), ("ldact", c_l), ("na", c_p),
                ("Ac", c_p), ("ldc", c_l), ("strideC", c_l), ("hc", c_p), ("Tinv", c_p), ("z", c_p), ("ldz", c_l),
                ("W", c_p), ("b", c_p), ("nb", c_i), ("N", c_i), ("B", c_i), ("status", c_p)]


class Dataset(ctypes.Structure):     # pgl_dataset_t
    _fields_ = [("T", c_i), ("Tp", c_i), ("X", c_p), ("Xt", c_p), ("Y", c_p), ("Psi", c_p), ("OK", c_p), ("llpart", c_p), ("elem0", c_u64),
                ("int8", c_i), ("planes", c_i), ("sA", c_p), ("PA", c_p), ("omega_override", c_p), ("xmax", c_p)]


NSTAGES = 16


class StageTimes(ctypes.Structure):  # pgl_stage_times_t
    _fields_ = [("ms", c_d * NSTAGES), ("work", c_d * NSTAGES), ("calls", c_i * NSTAGES), ("pending", c_p), ("mask", c_u32)]


class Sweep(ctypes.Structure):       # pgl_sweep_t
    _fields_ = [("N", c_i), ("B", c_i), ("n0", c_i), ("nloc", c_i), ("nb", c_i), ("obs", c_i), ("xi", c_d), ("visit_order", c_i),
                ("planes", c_i), ("i8_group", c_i), ("datasets", ctypes.POINTER(Dataset)), ("ndatasets", c_i),
                ("a", c_p), ("W", c_p), ("b", c_p), ("rho", c_p), ("Jw", c_p), ("hw", c_p), ("label", c_p), ("Jb", c_p), ("hb", c_p), ("c0", c_p),
                ("perm", c_p), ("u", c_p), ("z", c_p), ("inv_eta", c_p), ("G0", c_p), ("ll", c_p), ("status", c_p), ("logodds", c_p),
                ("Wt", c_p), ("bias", c_p), ("border", c_p), ("skip", c_p), ("c0_dense", c_p),
                ("Jbuf", c_p), ("Mtab", c_p), ("Ac", c_p), ("hc", c_p), ("Tinv", c_p), ("G", c_p), ("Lws", c_p), ("Ut", c_p), ("Wt_ws", c_p),
                ("d_idx", c_p), ("d_sign", c_p), ("d_cnt", c_p), ("batch_k", c_p), ("act", c_p), ("na", c_p),
                ("i8_PB", c_p), ("i8_R", c_p), ("i8_stat", c_p), ("i8_slice", c_i), ("i8_PAs", c_p), ("i8_Rx", c_p), ("i8_norm", c_p), ("nrun", c_i), ("nfirst", c_i),
                ("all_deterministic", c_i), ("init_rows_bound", c_i), ("active_rows_bound", c_i), ("flip_single_pass", c_i),
                ("times", ctypes.POINTER(StageTimes))]


# symbol -> argument types; every function returns int status unless noted. Mirrors include/pyglm_hip.h 1:1.
SIGNATURES = {
    "pgl_abi_version": [],
    "pgl_last_error": [],
    "pgl_philox_words": [c_u64, c_u32, c_u32, c_u64, c_u64, c_p, c_sz, c_p],
    "pgl_pg_draw": [c_p, c_p, c_p, c_sz, c_u64, c_u64, c_u64, c_p],
    "pgl_design_matrix": [c_p, c_l, c_p, c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_i, c_i, c_p],
    "pgl_transpose": [c_p, c_l, c_p, c_l, c_i, c_i, c_p],
    "pgl_activation": [c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_p],
    "pgl_pg_loglik": [c_p, c_l, c_p, c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_p, c_i, c_i, c_i, c_i, c_d, c_u64, c_u64, c_u64, c_u64, c_p],
    "pgl_pg_loglik_partials": [c_i],
    "pgl_gaussian_stats": [c_p, c_l, c_p, c_p, c_l, c_p, c_p, c_l, c_p, c_l, c_p, c_p, c_i, c_i, c_i, c_p],
    "pgl_scaled_gram": [c_p, c_l, c_p, c_p, c_l, c_l, c_i, c_i, c_p],
    "pgl_weighted_gram": [c_p, c_l, c_i, c_p, c_l, c_i, c_i, c_i, c_p, c_l, c_l, c_i, c_p],
    "pgl_i8_plane_bytes": [c_i, c_i],
    "pgl_i8_residue_bytes": [c_i],
    "pgl_i8_max_planes": [],
    "pgl_i8_padded_rows": [c_i],
    "pgl_i8_min_planes": [c_i],
    "pgl_i8_norm_bits": [c_i, c_i],
    "pgl_i8_norm_limit": [c_i, c_i],
    "pgl_i8_colstats": [c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_p, c_p, c_p],
    "pgl_i8_scales": [c_p, c_p, c_l, c_i, c_i, c_p, c_p],
    "pgl_i8_planes": [c_p, c_l, c_p, c_l, c_p, c_p, c_i, c_i, c_i, c_i, c_l, c_p],
    "pgl_i8_planes_t": [c_p, c_l, c_p, c_l, c_p, c_p, c_i, c_i, c_i, c_i, c_l, c_p],
    "pgl_i8_gram": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "pgl_i8_gram_slice": [c_p, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "pgl_i8_crt": [c_p, c_p, c_p, c_p, c_l, c_l, c_i, c_i, c_i, c_i, c_i, c_p],
    "pgl_contract_tn": [c_p, c_l, c_i, c_p, c_l, c_i, c_p, c_l, c_i, c_i, c_i, c_d, c_d, c_p],
    "pgl_contract_tn_batched": [c_p, c_l, c_l, c_i, c_p, c_l, c_l, c_i, c_p, c_l, c_l, c_i, c_i, c_i, c_i, c_p, c_d, c_d, c_i, c_i, c_p],
    "pgl_assemble_posterior": [c_p, c_l, c_l, c_p, c_p, c_l, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "pgl_flip_kmax": [],
    "pgl_flip_window_blocks": [c_i],
    "pgl_flip_apply": [ctypes.POINTER(FlipState), c_p],
    "pgl_flip_apply_chunk": [ctypes.POINTER(FlipState), c_i, c_p],
    "pgl_flip_apply_window": [ctypes.POINTER(FlipState), c_i, c_p],
    "pgl_flip_visit_order": [ctypes.POINTER(FlipState), c_p, c_l, c_l, c_p],
    "pgl_flip_decide": [ctypes.POINTER(FlipState), c_i, c_p],
    "pgl_row_stats": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "pgl_sweep_dims": [c_i, c_i, c_i, ctypes.POINTER(c_i), ctypes.POINTER(c_i), ctypes.POINTER(c_i)],
    "pgl_sweep": [ctypes.POINTER(Sweep), c_u64, c_u64, c_p],
    "pgl_get_state": [ctypes.POINTER(Sweep), c_p, c_p, c_p, c_p, c_p, c_p],
    "pgl_stage_name": [c_i],
    "pgl_stage_times_collect": [ctypes.POINTER(StageTimes)],
    "pgl_active_index": [ctypes.POINTER(CholState), c_p],
    "pgl_sample_weights": [ctypes.POINTER(CholState), c_i, c_p],
    "pgl_ubench_mfma": [c_i, c_d, ctypes.POINTER(c_d), ctypes.POINTER(c_d), c_p],
}

ABI_VERSION = 11
_lib = None


class PglError(RuntimeError):
    pass


def load():
    """Load the HIP library (once). No fallback: a missing library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PglError("libpyglm_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "or `make -C pyglm_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
    import torch  # noqa: F401  -- FIRST: PyTorch-ROCm carries its own HIP runtime; a process must bind one runtime only, and the library's
    #                      dependency on libamdhip64 has to resolve to the copy torch has loaded (loading the library first leaves
    #                      two runtimes in the process: "no ROCm-capable device is detected" at the first launch)
    lib = ctypes.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError if the export is missing
        fn.argtypes = args
        fn.restype = (ctypes.c_char_p if name in ("pgl_last_error", "pgl_stage_name") else ctypes.c_size_t if name in ("pgl_i8_plane_bytes", "pgl_i8_residue_bytes")
                      else ctypes.c_double if name == "pgl_i8_norm_limit" else ctypes.c_int)
    if lib.pgl_abi_version() != ABI_VERSION:
        raise PglError("libpyglm_hip.so ABI version %d != %d (rebuild: make -C pyglm_amd/csrc)" % (lib.pgl_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def source_hash():
    """sha256 (first 16 hex digits) over the kernel sources of the library -- pyglm_amd/csrc/*.hip, *.h and the Makefile, in name order, plus
    include/pyglm_hip.h: committed hardware-counter summaries (profiles/*.json) record the hash they were taken at, and bench.py quotes them
    only while it still matches"""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(_HERE, "csrc")
    files = [os.path.join(src, f) for f in sorted(os.listdir(src)) if f.endswith((".hip", ".h")) or f == "Makefile"]
    files.append(os.path.join(os.path.dirname(_HERE), "include", "pyglm_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise PglError("%s failed (%d): %s" % (name, rc, lib.pgl_last_error().decode()))
    return rc


def ptr(t):
    """device pointer of a torch tensor (or None)"""
    return None if t is None else ctypes.c_void_p(t.data_ptr())
