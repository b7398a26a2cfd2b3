"""Hunt over fields with cells EQUAL to the level: every job must match the oracle (skimage's coordinate joins, the
order-dependent constellations included: those are joined by wb_k_trace_walk) or carry STATUS_LEVEL_KEYERROR exactly where
the oracle's restatement of skimage's walk raises KeyError (development tool; WB_HUNT_GPU=1 for the HIP build).   python tools/levelhit_hunt.py <first seed> <count> [dlon]"""
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
    pv, lat, lon = helpers.level_hit_field(seed, dlon=dlon, nhits=(4 + seed % 9) * int(os.environ.get("WB_HUNT_HITS", 1)))
    if os.environ.get("WB_HUNT_QUANT"):                  # quantised PV (packed data): cells equal to the level by the dozen
        q = float(os.environ["WB_HUNT_QUANT"])
        pv = (np.round(pv / q) * q).astype(pv.dtype)
    ctx = _hip.Context(0) if os.environ.get("WB_HUNT_GPU") else _hip.Context(0, helpers.hostsim_library())
    try:
        got = helpers.run_abi(ctx, pv, lat, lon, [2.0], intensity=pv)
        st = int(got["status"][0])
        try:
            helpers.full_parity(ctx, pv, lat, lon, synth.times(1), [2.0], intensity=pv)
        except KeyError:                                   # the oracle's restatement of skimage's walk raises: so must the job
            return (seed, "keyerror-ok" if st & _hip.STATUS_LEVEL_KEYERROR else "FAIL oracle raises KeyError, status %d" % st)
        if st & _hip.STATUS_LEVEL_KEYERROR:
            return (seed, "FAIL status KEYERROR but the oracle returns contours")
        if st & _hip.STATUS_LEVEL_ORDER:
            return (seed, "walked-ok" if st & _hip.STATUS_LEVEL_WALKED else "FAIL flagged but not walked")
        return (seed, "ok" if st & _hip.STATUS_LEVEL_HIT else "ok-nohit")
    except Exception as e:
        return (seed, "FAIL " + type(e).__name__ + " " + str(e)[:120])
    finally:
        ctx.close()


if __name__ == "__main__":
    first, count = int(sys.argv[1]), int(sys.argv[2])
    dlon = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    import helpers
    if not os.environ.get("WB_HUNT_GPU"):
        helpers.build_hostsim()
    t = time.time()
    tally = {}
    with mp.get_context("spawn").Pool(int(os.environ.get("WB_HUNT_WORKERS", 6))) as pool:
        for seed, res in pool.imap_unordered(case, [(s, dlon) for s in range(first, first + count)]):
            tally[res.split()[0]] = tally.get(res.split()[0], 0) + 1
            if res.startswith("FAIL"):
                print(seed, res, flush=True)
    print("levelhit hunt", first, count, dlon, tally, "%.0fs" % (time.time() - t), flush=True)
