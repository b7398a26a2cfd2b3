"""development probe: the device-resident result of wb_indices viewed through torch (single process)"""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
from wavebreaking_amd import _hip, synthetic as synth
from wavebreaking_amd import distributed as wd
torch.cuda.set_device(0)
ctx = _hip.Context(0)
ctx.set_option("keep_device", 1)
lat, lon = synth.grid(0.25)
nt = 64
d = ctx.alloc(nt * lat.size * lon.size * 4)
ctx.synth_fill(d, np.float32, 0, nt, lat, lon, synth.wave_params(), 0.0, synth.SEED)
for rep in range(3):
    b = ctx.contours(d, lat, lon, [-2.0, 2.0], 0.25, 0.25, device_shape=(nt, lat.size, lon.size), device_dtype=np.float32)
    b.set_step_offset(100)
    b.indices()
    cat = wd.gather_device_tables(b, None)
    for k in (0, 1, 2):
        ev = b.events(k)
        assert np.array_equal(cat[k]["step"], ev["step"] + 100) and np.array_equal(cat[k]["sums"], ev["sums"])
        assert np.array_equal(cat[k]["poly_x"], ev["poly_x"]) and np.array_equal(cat[k]["poly_offset"], ev["poly_offset"])
    b.close()
print("device gather view OK", [c["step"].size for c in cat], flush=True)
ctx.close()
print("closed", flush=True)
