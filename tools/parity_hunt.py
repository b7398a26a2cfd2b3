"""Hunt for residual parity mismatches: many random single-step fields, host-simulation library against the oracle, in worker
processes (development tool).   python tools/parity_hunt.py <first seed> <count> [dlon]     WB_HUNT_GPU=1: the HIP library on the GPU box
(the oracle is the slow part: WB_HUNT_WORKERS processes share the GPU)"""
import os, sys, time
import multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def case(args):
    seed, dlon = args
    import numpy as np
    import helpers
    from oracle import synth
    from wavebreaking_amd import _hip
    rng = np.random.RandomState(seed)
    eps = float(rng.choice([0.0, 0.1, 0.2, 0.3, 0.45]))
    t0 = int(rng.randint(0, 5000))
    levels = [[-2, 2], [2], [-2, 2, 1.5], [1.0, 3.0]][int(rng.randint(0, 4))]
    lat, lon = synth.grid(dlon)
    nt = int(os.environ.get("WB_HUNT_NT", 1))
    pv = synth.pv_field(t0, nt, lat, lon, eps=eps, seed=1000 + seed)
    if os.environ.get("WB_HUNT_F64"):
        pv = pv.astype(np.float64)
    ctx = _hip.Context(0) if os.environ.get("WB_HUNT_GPU") else _hip.Context(0, helpers.hostsim_library())
    try:
        kw = {}
        if os.environ.get("WB_HUNT_PARAMS"):           # random index parameters and periodic extension
            me = float(rng.choice([3.0, 5.0, 8.0]))
            kw = dict(geo_dis=float(rng.choice([400.0, 800.0, 1200.0])), cont_dis=float(rng.choice([1000.0, 1500.0, 2500.0])),
                      range_group=float(rng.choice([3.0, 5.0, 8.0])), ot_min_exp=me, co_min_exp=me,
                      periodic_add=float(rng.choice([60.0, 120.0, 180.0])))
        helpers.full_parity(ctx, pv, lat, lon, synth.times(nt), levels, intensity=pv, **kw)
        return (seed, eps, t0, levels, "ok")
    except Exception as e:
        return (seed, eps, t0, levels, "FAIL " + type(e).__name__ + " " + str(e)[:120])
    finally:
        ctx.close()


if __name__ == "__main__":
    first, count = int(sys.argv[1]), int(sys.argv[2])
    dlon = float(sys.argv[3]) if len(sys.argv) > 3 else 0.25
    import helpers
    if not os.environ.get("WB_HUNT_GPU"):
        helpers.build_hostsim()
    t = time.time()
    bad = 0
    with mp.get_context("spawn").Pool(int(os.environ.get("WB_HUNT_WORKERS", 6))) as pool:
        for r in pool.imap_unordered(case, [(s, dlon) for s in range(first, first + count)]):
            if r[-1] != "ok":
                bad += 1
                print(r, flush=True)
    print("hunt", first, count, dlon, "failures", bad, "%.0fs" % (time.time() - t), flush=True)
