"""development probe: wb_fingerprint throughput on the GPU box's host by thread count (one cube, and two cubes side by side)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavebreaking_amd import _hip
from concurrent.futures import ThreadPoolExecutor
L = _hip.default_library()
a = np.random.default_rng(0).random((365, 721, 1440), dtype=np.float32)
b = a.copy()
print("cube %.2f GB, cpus %d" % (a.nbytes / 1e9, os.cpu_count()))
for th in (8, 16, 32, 64, 96, 128):
    ts = []
    for rep in range(3):
        t = time.perf_counter(); L.fingerprint(a, th); ts.append(time.perf_counter() - t)
    with ThreadPoolExecutor(2) as pool:
        t = time.perf_counter(); list(pool.map(lambda x: L.fingerprint(x, th), (a, b))); t2 = time.perf_counter() - t
    print("threads %3d: one cube %.1f ms = %.0f GB/s; two side by side %.1f ms = %.0f GB/s" % (th, 1e3 * min(ts), a.nbytes / min(ts) / 1e9, 1e3 * t2, 2 * a.nbytes / t2 / 1e9), flush=True)
