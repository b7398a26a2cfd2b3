"""Synthetic ERA5-shaped PV fields -- numpy side.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The device-side generator
``wb_synth_fill`` (wavebreaking_amd/csrc/wb_synth.hip) draws the same
parameters and the same per-cell noise from the same counter-based hash; the
smooth part goes through cos/tanh, so host and device fields agree to f32
rounding, not bit for bit -- parity checks therefore always run both paths on
ONE field (tests: the numpy field uploaded; bench: the device field copied
back for the sampled steps).

    PV(t, phi, lam) = sgn(phi) * max(0, 2 + 6 tanh((|phi| + D - 45) / 6)) + eps * eta
    D = sum_{k=3..9} a_k cos(k lam + psi_k + omega_k t + tau_k k (|phi_rad| - 0.8))
    (psi_k differs between the hemispheres so that the two are not mirror images)

Deviation from the literal SURVEY formula: the ``max(0, .)`` clamp.  Without it
the tropics change sign (2 - 6 = -4), which puts a second full-span +-2 PVU
contour in each hemisphere plus a straight one along the equator; real
isentropic PV has one dynamical-tropopause contour per hemisphere and level.
"""
import numpy as np

MASK64 = (1 << 64) - 1
SEED = 20240
AVOID_LEVELS = (-2.0, 2.0)


def splitmix64(x):
    """splitmix64 finaliser on numpy uint64 arrays (wraps mod 2**64)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _u01(h):
    """top 53 bits -> [0, 1) double; exact on host and device."""
    return (np.asarray(h, dtype=np.uint64) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _key(seed, stream, a, b=0):
    with np.errstate(over="ignore"):
        k = splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream))
        k = splitmix64(k ^ np.asarray(a, dtype=np.uint64))
        k = splitmix64(k ^ np.asarray(b, dtype=np.uint64))
    return k


def wave_params(seed=SEED):
    """The 7 planetary-wave parameter sets (k = 3..9): a_k, psi_k, tau_k, omega_k, psi_k(south).

    Returned as a (7, 5) float64 table that is handed verbatim to the device
    generator, so both sides use identical parameters."""
    k = np.arange(3, 10, dtype=np.uint64)
    u = 0.5 + 0.5 * _u01(_key(seed, 1, k))
    psi = 2.0 * np.pi * _u01(_key(seed, 2, k))
    # Box-Muller from two uniforms
    u1 = np.maximum(_u01(_key(seed, 3, k)), 1e-300)
    u2 = _u01(_key(seed, 4, k))
    tau = 3.0 * np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    omega = (0.02 + 0.06 * _u01(_key(seed, 5, k))) / k.astype(np.float64)
    a = 15.0 * u / np.sqrt(7.0)
    psi_s = 2.0 * np.pi * _u01(_key(seed, 7, k))      # southern-hemisphere phase (breaks the mirror symmetry)
    return np.stack([a, psi, tau, omega, psi_s], axis=1)


def grid(dlon, dlat=None, lat_min=-90.0, lat_max=90.0):
    dlat = dlon if dlat is None else dlat
    nlon = int(round(360.0 / dlon))
    nlat = int(round((lat_max - lat_min) / dlat)) + 1
    lon = -180.0 + dlon * np.arange(nlon, dtype=np.float64)
    lat = lat_min + dlat * np.arange(nlat, dtype=np.float64)
    return lat, lon


def times(nt, start="1980-01-01T00", step_hours=6):
    return np.datetime64(start, "ns") + np.arange(nt) * np.timedelta64(step_hours, "h")


def noise(seed, t, nlat, nlon):
    """eta in U(-1, 1), one draw per (t, j, i)."""
    j = np.arange(nlat, dtype=np.uint64)[:, None]
    i = np.arange(nlon, dtype=np.uint64)[None, :]
    cell = j * np.uint64(nlon) + i
    return 2.0 * _u01(_key(seed, 6, np.uint64(t), cell)) - 1.0


def pv_field(t0, nt, lat, lon, eps=0.0, seed=SEED, dtype=np.float32):
    """(nt, nlat, nlon) synthetic PV, time index t0 .. t0+nt-1."""
    dtype = np.dtype(dtype).type
    par = wave_params(seed)
    phi = np.abs(lat)[:, None]
    phir = np.radians(phi)
    lam = np.radians(lon)[None, :]
    sgn = np.sign(lat)[:, None]
    south = (lat < 0)[:, None]
    out = np.empty((nt, lat.size, lon.size), dtype=dtype)
    for n in range(nt):
        t = float(t0 + n)
        D = np.zeros((lat.size, lon.size))
        for m, k in enumerate(range(3, 10)):
            a, psi, tau, om, psi_s = par[m]
            ph = np.where(south, psi_s, psi)
            D += a * np.cos(k * lam + ph + om * t + tau * k * (phir - 0.8))
        pv = sgn * np.maximum(0.0, 2.0 + 6.0 * np.tanh((phi + D - 45.0) / 6.0))
        if eps:
            pv = pv + eps * noise(seed, t0 + n, lat.size, lon.size)
        v = pv.astype(dtype)
        # generator contract: no cell equals a contour level exactly.  2 + 6 tanh(x) rounds
        # to exactly 2.0f for |x| < 1e-8, so such cells are moved one ulp up (same rule on the device).
        for lv in AVOID_LEVELS:
            v = np.where(v == dtype(lv), np.nextafter(dtype(lv), dtype(np.inf)), v)
        out[n] = v
    return out


def assert_no_level_hits(field, levels):
    """The tracer's edge-linking equals scikit-image's coordinate joins only when
    no cell equals a contour level exactly -- generator contract."""
    for lv in np.atleast_1d(levels):
        if np.any(field == np.asarray(lv, dtype=field.dtype)):
            raise AssertionError("synthetic field hits level %r exactly" % (lv,))
