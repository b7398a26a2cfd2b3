"""Scan kernels alone on the bench batch: duration per variant (development tool; run on the GPU box)."""
import sys, numpy as np, time
sys.path.insert(0, '/root/repo')
from wavebreaking_amd import _hip, synthetic as synth
ctx = _hip.Context(0)
ctx.set_option("timers", 2)              # (every stage timed: the library times the field scan only by default)
lat, lon = synth.grid(0.25)
nt = 1460
d = ctx.alloc(nt * lat.size * lon.size * 4)
ctx.synth_fill(d, np.float32, 0, nt, lat, lon, synth.wave_params(), 0.0, synth.SEED)
def run(split, variant, debug, halo=1):
    ctx.set_option("scan_split", split); ctx.set_option("scan_variant", variant); ctx.set_option("scan_debug", debug)
    ctx.set_option("scan_halo_lds", halo)
    best = 1e9
    for k in range(4):
        try:
            b = ctx.contours(d, lat, lon, [-2.0, 2.0], 0.25, 0.25, device_shape=(nt, lat.size, lon.size), device_dtype=np.float32)
        except RuntimeError as e:
            return str(e)[:40]
        t = dict(b.timings()); b.close()
        if t["scan"] < best:
            best, bb = t["scan"], t.get("scan_build", 0.0)
    return "%.3f ms  %.2f TB/s  build %.3f ms" % (best, nt*lat.size*lon.size*4/best/1e9, bb)
configs = [(0,0,0),(1,0,0),(1,1,0),(1,2,0),(1,3,0),(1,1,1)]
if len(sys.argv) >= 4:                                 # one configuration (e.g. under rocprofv3 --pmc): split variant debug [halo]
    configs = [tuple(int(v) for v in sys.argv[1:])]
if len(sys.argv) == 2 and sys.argv[1] == "halo":       # same-box A/B of the halo exchange: (variant, halo) alternating
    configs = [(1, v, 0, h) for _ in range(3) for v, h in ((2, 1), (1, 1), (5, 1), (0, 0))]
for cfg in configs:
    print(*cfg, run(*cfg), flush=True)
