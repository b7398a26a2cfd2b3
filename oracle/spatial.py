"""CPU restatement of wavebreaking/processing/spatial.py (TEST INFRASTRUCTURE -- never imported by the product).

SURVEY.md 8(f) rows 3 and 4: the two pre-processing steps either side of the hot path.

* ``smooth_field``  follows ``calculate_smoothed_field`` (spatial.py:60-128): per time step, ``passes`` times
  ``scipy.ndimage.convolve(temp, weights, mode) / np.sum(weights)``, then the rows ``arange(-b, b)``,
  ``b = int(weights.shape[0] / 2 + 0.5)``, are set to NaN (spatial.py:104-107).
  scipy is a third-party dependency (unpinned, setup.py:19); its algorithm (ni_filters.c, NI_Correlate) is restated
  here: the kernel is flipped, weights with ``|w| <= DBL_EPSILON`` are skipped, the remaining taps are accumulated
  in C order of the flipped kernel as ``tmp += w * (double)x`` (no FMA in the x86-64 wheels) and ``tmp`` is cast to
  the input dtype.  With numpy >= 2 (NEP 50) ``float32_array / np.int64(6)`` is float64, so from the first
  division on everything is float64; that is the behaviour pinned here (tests/test_oracle_spatial.py runs this
  file against the scipy 1.15 / numpy 2.2 of the build image, tests/golden/spatial.npz holds their outputs).
* ``momentum_flux`` follows ``calculate_momentum_flux`` (spatial.py:27-57): ``(u - u.mean(lon)) * (v - v.mean(lon))``.
  xarray's float mean is ``np.nanmean`` (no bottleneck in the reference's requirements): NaNs count as 0, the
  row sum is numpy's PAIRWISE summation in the array's own dtype (``pairwise_sum`` below restates
  numpy/_core/src/umath/loops_utils.h.src), divided by the count of non-NaN cells.
"""
import numpy as np

MODES = {"reflect": 0, "grid-mirror": 0, "constant": 1, "grid-constant": 1, "nearest": 2, "mirror": 3, "wrap": 4,
         "grid-wrap": 4}


def extend_index(i, n, mode):
    """Source index of position ``i`` outside ``[0, n)`` (scipy ni_support.c NI_ExtendLine semantics); -1 = cval."""
    m = MODES[mode]
    if 0 <= i < n:
        return i
    if m == 1:
        return -1
    if m == 2:
        return 0 if i < 0 else n - 1
    if m == 4:
        return i % n
    if m == 0:                                  # d c b a | a b c d | d c b a
        p = 2 * n
        i %= p
        return i if i < n else p - 1 - i
    if n == 1:                                  # mirror of a single cell
        return 0
    p = 2 * n - 2                               # d c b | a b c d | c b a
    i %= p
    return i if i < n else p - i


def convolve2d(a, weights, mode="wrap"):
    """scipy.ndimage.convolve(a, weights, mode=mode) for a 2-D float array, restated (see module docstring)."""
    a = np.asarray(a)
    wf = np.asarray(weights, dtype=np.float64)[::-1, ::-1]
    kh, kw = wf.shape
    nr, nc = a.shape
    # convolve(): origin of the flipped kernel is -origin, minus one for even sizes (scipy/ndimage/_filters.py)
    ch = kh // 2 - (0 if kh & 1 else 1)
    cw = kw // 2 - (0 if kw & 1 else 1)
    acc = np.zeros(a.shape, np.float64)
    for i in range(kh):
        ri = np.array([extend_index(r + i - ch, nr, mode) for r in range(nr)])
        for j in range(kw):
            if not abs(wf[i, j]) > np.finfo(np.float64).eps:
                continue
            cj = np.array([extend_index(c + j - cw, nc, mode) for c in range(nc)])
            src = a[np.maximum(ri, 0)][:, np.maximum(cj, 0)].astype(np.float64)
            src[ri < 0, :] = 0.0                # cval
            src[:, cj < 0] = 0.0
            acc = acc + wf[i, j] * src
    return acc.astype(a.dtype)


def smooth_field(data, passes, weights=np.array([[0, 1, 0], [1, 2, 1], [0, 1, 0]]), mode="wrap"):
    """calculate_smoothed_field (spatial.py:98-109) on a (time, lat, lon) cube -> stacked result."""
    weights = np.asarray(weights)
    out = []
    for t in range(data.shape[0]):
        temp = data[t]
        for _ in range(int(passes)):
            temp = convolve2d(temp, weights, mode) / np.sum(weights)
        temp = np.array(temp, copy=True)
        b = int(weights.shape[0] / 2 + 0.5)
        temp[np.arange(-b, b), :] = np.nan
        out.append(temp)
    return np.array(out)


def pairwise_sum(a):
    """numpy's pairwise summation of a contiguous 1-D float array in its own dtype (loops_utils.h.src *_pairwise_sum).
    This is ONE call of numpy's inner loop: a reduction over more than 8192 elements (the ufunc buffer size) is handed over in
    blocks of 8192 whose pairwise sums are added one after the other (csrc/wb_core.h:wb_np_pairwise_sum restates that, checked
    against numpy in tests/test_hostsim_parity.py); the rows summed here are at most 4096 long."""
    dt = a.dtype.type
    n = a.size
    if n < 8:
        res = dt(0.0)
        for i in range(n):
            res = dt(res + a[i])
        return res
    if n <= 128:
        r = [a[k] for k in range(8)]
        i = 8
        while i < n - (n % 8):
            for k in range(8):
                r[k] = dt(r[k] + a[i + k])
            i += 8
        res = dt(dt(dt(r[0] + r[1]) + dt(r[2] + r[3])) + dt(dt(r[4] + r[5]) + dt(r[6] + r[7])))
        while i < n:
            res = dt(res + a[i])
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return dt(pairwise_sum(a[:n2]) + pairwise_sum(a[n2:]))


def zonal_mean(x):
    """np.nanmean(x, axis=-1) of a (..., nlon) float array, restated: NaN -> 0 in the sum, divided by the non-NaN count."""
    x = np.asarray(x)
    flat = x.reshape(-1, x.shape[-1])
    out = np.empty(flat.shape[0], dtype=x.dtype)
    with np.errstate(invalid="ignore", divide="ignore"):
        for k in range(flat.shape[0]):
            row = flat[k]
            nan = np.isnan(row)
            tot = pairwise_sum(np.where(nan, x.dtype.type(0), row))
            out[k] = x.dtype.type(np.float64(tot) / np.float64(int((~nan).sum())))   # true_divide in double, cast back
    return out.reshape(x.shape[:-1])


def momentum_flux(u, v):
    """calculate_momentum_flux (spatial.py:49-56) on (time, lat, lon) cubes of one float dtype."""
    up = u - zonal_mean(u)[..., None]
    vp = v - zonal_mean(v)[..., None]
    return up * vp
