"""The synthetic ERA5-shaped workload of SURVEY.md 8(d): grid, time axis and the parameter table of the device-side PV
generator (``wb_synth_fill`` in csrc/wb_api.hip evaluates

    PV(t, phi, lam) = sgn(phi) * max(0, 2 + 6 tanh((|phi| + D - 45) / 6)) + eps * eta
    D = sum_{k=3..9} a_k cos(k lam + psi_k + omega_k t + tau_k k (|phi_rad| - 0.8))

with a_k, psi_k, tau_k, omega_k drawn here from a counter-based splitmix64 hash, so every rank and every box generates the
same record).  Used by bench.py and benchmarks/; the test oracle carries its own numpy statement of the same field."""
import numpy as np

SEED = 20240


def _splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _u01(h):
    return (np.asarray(h, dtype=np.uint64) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _key(seed, stream, a):
    with np.errstate(over="ignore"):
        k = _splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream))
        k = _splitmix64(k ^ np.asarray(a, dtype=np.uint64))
        k = _splitmix64(k ^ np.uint64(0))
    return k


def wave_params(seed=SEED):
    """(7, 5) float64 table for wave numbers k = 3..9: a_k, psi_k, tau_k, omega_k, psi_k of the southern hemisphere."""
    k = np.arange(3, 10, dtype=np.uint64)
    u = 0.5 + 0.5 * _u01(_key(seed, 1, k))
    psi = 2.0 * np.pi * _u01(_key(seed, 2, k))
    u1 = np.maximum(_u01(_key(seed, 3, k)), 1e-300)
    u2 = _u01(_key(seed, 4, k))
    tau = 3.0 * np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    omega = (0.02 + 0.06 * _u01(_key(seed, 5, k))) / k.astype(np.float64)
    a = 15.0 * u / np.sqrt(7.0)
    psi_s = 2.0 * np.pi * _u01(_key(seed, 7, k))
    return np.stack([a, psi, tau, omega, psi_s], axis=1)


def grid(dlon, dlat=None, lat_min=-90.0, lat_max=90.0):
    """(lat, lon) of the regular global grid: lon = -180 + i dlon, lat = lat_min + j dlat, both ascending."""
    dlat = dlon if dlat is None else dlat
    nlon = int(round(360.0 / dlon))
    nlat = int(round((lat_max - lat_min) / dlat)) + 1
    return lat_min + dlat * np.arange(nlat, dtype=np.float64), -180.0 + dlon * np.arange(nlon, dtype=np.float64)


def times(nt, start="1980-01-01T00", step_hours=6):
    return np.datetime64(start, "ns") + np.arange(nt) * np.timedelta64(step_hours, "h")
