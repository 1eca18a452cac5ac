"""Small host helpers with the reference's names (pyglm/utils/utils.py:3-27)."""
import numpy as np


def logistic(x):
    return 1.0 / (1.0 + np.exp(-x))


def expand_scalar(x, shp):
    """scalar -> array of shape shp; arrays must already have that shape (utils.py:7-12)"""
    if np.isscalar(x):
        return np.full(shp, float(x))
    x = np.asarray(x, dtype=float)
    assert x.shape == tuple(shp), "expected shape %s, got %s" % (tuple(shp), x.shape)
    return x


def expand_cov(c, shp):
    """scalar c -> c*I broadcast over the leading dims (utils.py:15-27)"""
    shp = tuple(shp)
    assert len(shp) >= 2 and shp[-2] == shp[-1]
    if np.isscalar(c):
        out = np.zeros(shp)
        idx = np.arange(shp[-1])
        out[..., idx, idx] = float(c)
        return out
    c = np.asarray(c, dtype=float)
    assert c.shape == shp
    return c


def fingerprint(arr):
    """(shape, dtype, 64-bit content checksum) of an array: what the engine caches key on.  Python ids are reused as soon as an array is
    freed and say nothing about in-place edits, so cached device copies are matched by content (a pass over the bytes, cheaper than the
    upload it saves)."""
    import zlib
    a = np.ascontiguousarray(arr)
    try:
        import xxhash
        h = xxhash.xxh3_64_intdigest(memoryview(a).cast("B"))
    except ImportError:
        buf = memoryview(a).cast("B")
        h = (zlib.crc32(buf) << 32) | zlib.adler32(buf)
    return (a.shape, a.dtype.str, h)
