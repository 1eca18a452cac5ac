"""The C-ABI library loads and exports every symbol include/pyglm_hip.h declares (no compute calls: no GPU needed)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pyglm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pgl_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ["pgl_pg_draw", "pgl_weighted_gram", "pgl_activation", "pgl_pg_loglik", "pgl_flip_decide", "pgl_sample_weights", "pgl_last_error"]:
        assert must in syms


def test_library_exports_every_declared_symbol():
    from pyglm_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(lib, s), "libpyglm_hip.so does not export %s" % s
    assert sorted(_lib.SIGNATURES) == declared_symbols()      # the ctypes table mirrors the header 1:1
    assert _lib.load().pgl_abi_version() == _lib.ABI_VERSION == 11
    assert _lib.load().pgl_flip_kmax() == 512 and _lib.load().pgl_flip_window_blocks(5) == 64
    # host-side constants of the integer Gram: bits of the column norms per number of moduli, fewest moduli at the fp64 level
    lib = _lib.load()
    assert [lib.pgl_i8_norm_bits(k, 100000) for k in (12, 13, 14, 15)] == [46, 50, 54, 58]
    assert lib.pgl_i8_max_planes() == 15 and lib.pgl_i8_min_planes(100000) == 13 and 50.7 < __import__('math').log2(lib.pgl_i8_norm_limit(13, 100000)) < 50.8


def test_struct_layouts_match_header_field_order():
    from pyglm_amd import _lib
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "pyglm_hip.h")).read(), flags=re.S)
    for cname, cls in [("pgl_flip_t", _lib.FlipState), ("pgl_chol_t", _lib.CholState), ("pgl_dataset_t", _lib.Dataset), ("pgl_sweep_t", _lib.Sweep),
                       ("pgl_stage_times_t", _lib.StageTimes)]:
        body = re.search(r"typedef struct \{([^{}]*)\} %s;" % cname, text).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                names.append(re.findall(r"([A-Za-z_0-9]+)\s*$", re.sub(r"\[[^\]]*\]", "", part).strip())[0])
        assert names == [f[0] for f in cls._fields_], (cname, names)


def test_shipped_library_reads_no_environment_variable():
    """tuning knobs live behind -DPGL_AB (`make -C pyglm_amd/csrc ab`), which the package never loads: no getenv in the product sources outside
    that fence, and none imported by the shipped library"""
    import subprocess
    from pyglm_amd import _lib
    src = os.path.join(ROOT, "pyglm_amd", "csrc")
    for f in sorted(os.listdir(src)):
        if f.endswith((".hip", ".h")):
            text = open(os.path.join(src, f)).read()
            text = re.sub(r"#ifdef PGL_AB\n.*?#else", "", text, flags=re.S)
            assert "getenv" not in text, f
    nm = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True)
    if nm.returncode == 0:
        assert "getenv" not in nm.stdout


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from pyglm_amd._lib import PglError
    from pyglm_amd.engine import GibbsEngine
    with pytest.raises(PglError):
        GibbsEngine(4, 1)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pyglm_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
