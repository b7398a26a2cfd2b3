"""Timing probe of wb_track_events on a synthetic catalogue (development tool; run on the GPU box)."""
import sys, time
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from wavebreaking_amd import _hip
import test_tracking as tt
ctx=_hip.Context(0)
n,steps=200000,5000
t=time.time(); dates,polys,com=tt._tracking_case(7,2000,50,True)
# tile the 2000 polygons to n events over many steps
rng=np.random.RandomState(1)
idx=rng.randint(0,2000,n)
polys=[polys[i] for i in idx]
step=np.sort(rng.randint(0,steps,n))
dates=(np.datetime64("1979-01-01T00","ns")+step*np.timedelta64(6,"h")).astype("datetime64[ns]")
com=np.c_[rng.uniform(-180,180,n),rng.uniform(-80,80,n)]
print('setup',time.time()-t)
for m,kw in (("by_overlap",dict(overlap=0.0,rings_per_event=polys)),("by_distance",dict(distance=1000.0,com=com))):
    for rep in range(2):
        t=time.time(); label,area,stats=ctx.track_events(dates,m,6.0,**kw); w=time.time()-t
        print(m,'wall %.3f s'%w,'device %.2f ms'%ctx.last_op_ms(),stats,'labels',label.max()+1)
