"""Golden vectors for SURVEY.md 8(f) rows 3-4 (G5): outputs of the THIRD-PARTY libraries the reference calls --
scipy.ndimage.convolve (spatial.py:101) and numpy's nanmean (xarray mean, spatial.py:50-51) -- on small seeded inputs.
Run in the build image (scipy 1.15.3, numpy 2.2.6):   python tests/golden/make_golden_spatial.py
Nothing from /root/reference is imported or copied; only inputs and expected outputs are stored.
"""
import os

import numpy as np
from scipy import ndimage

HERE = os.path.dirname(os.path.abspath(__file__))


def smoothed(data, passes, weights, mode):
    # the per-step recipe of spatial.py:98-109 written against scipy / numpy directly
    out = []
    for t in range(data.shape[0]):
        temp = data[t]
        for _ in range(passes):
            temp = ndimage.convolve(temp, weights=weights, mode=mode) / np.sum(weights)
        temp = np.array(temp, copy=True)
        b = int(weights.shape[0] / 2 + 0.5)
        temp[np.arange(-b, b), :] = np.nan
        out.append(temp)
    return np.array(out)


def main():
    rng = np.random.default_rng(85)
    out = {}
    five = np.array([[0, 1, 0], [1, 2, 1], [0, 1, 0]])
    cases = [
        ("f32_p1", np.float32, (2, 9, 12), 1, five, "wrap"),
        ("f32_p5", np.float32, (2, 14, 20), 5, five, "wrap"),
        ("f64_p3", np.float64, (1, 11, 16), 3, five, "wrap"),
        ("f64_p7_small", np.float64, (1, 6, 5), 7, five, "wrap"),
        ("f32_full3", np.float32, (1, 10, 13), 2, rng.standard_normal((3, 3)), "wrap"),
        ("f64_5x3_reflect", np.float64, (1, 12, 9), 2, rng.standard_normal((5, 3)), "reflect"),
        ("f32_3x5_mirror", np.float32, (1, 9, 11), 2, rng.standard_normal((3, 5)), "mirror"),
        ("f64_nearest", np.float64, (1, 8, 8), 3, five, "nearest"),
        ("f32_constant", np.float32, (1, 8, 10), 2, five, "constant"),
        ("f64_even", np.float64, (1, 9, 10), 1, rng.standard_normal((2, 4)), "wrap"),
    ]
    names = []
    for name, dt, shape, passes, w, mode in cases:
        data = (rng.standard_normal(shape) * 3).astype(dt)
        if name == "f32_p5":
            data[0, 5, 7] = np.nan                      # NaN spreads one cell per pass
        out[name + "_in"] = data
        out[name + "_w"] = np.asarray(w)
        out[name + "_out"] = smoothed(data, passes, np.asarray(w), mode)
        names.append("%s|%d|%s" % (name, passes, mode))
    out["smooth_cases"] = np.array(names)
    # zonal means / momentum flux: numpy nanmean along the last (contiguous) axis
    mf = []
    for name, dt, nlon in (("mf32_a", np.float32, 7), ("mf32_b", np.float32, 100), ("mf32_c", np.float32, 360),
                           ("mf32_d", np.float32, 1440), ("mf64_a", np.float64, 129), ("mf64_b", np.float64, 1440),
                           ("mf32_e", np.float32, 1001)):
        u = (rng.standard_normal((2, 2, nlon)) * 20 + 10).astype(dt)
        v = (rng.standard_normal((2, 2, nlon)) * 8).astype(dt)
        if name == "mf32_b":
            u[0, 1, 17] = np.nan
            v[1, 1, :3] = np.nan
        with np.errstate(all="ignore"):
            um, vm = np.nanmean(u, axis=2), np.nanmean(v, axis=2)
            out[name + "_u"], out[name + "_v"] = u, v
            out[name + "_um"], out[name + "_vm"] = um, vm
            out[name + "_mflux"] = (u - um[..., None]) * (v - vm[..., None])
        mf.append(name)
    out["mflux_cases"] = np.array(mf)
    np.savez_compressed(os.path.join(HERE, "spatial.npz"), **out)
    print("wrote spatial.npz:", len(cases), "smoothing cases,", len(mf), "momentum-flux cases")


if __name__ == "__main__":
    main()
