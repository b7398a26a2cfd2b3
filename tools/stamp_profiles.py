#!/usr/bin/env python
"""Copy the summaries of one tools/gpu_round.sh session (gpurun_out/<name>) into profiles/ as rNN_*, each with a header that
says which command produced it.      python tools/stamp_profiles.py gpurun_out/r2d r02"""
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
what = {
    "kernel_stats.csv": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --cpu-sample 0 --e2e-steps 0 --stagger-steps 0   (default: "
                        "20 timed + 3 warm-up passes with two batches in flight, after 3 set-up passes alone on the device; wb_k_synth is "
                        "the input generator)",
    "kernel_stats_one_batch.csv": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --cpu-sample 0 --e2e-steps 0 "
                                  "--streams 1 --stagger-steps 0   (one batch at a time: every launch alone on the device)",
    "hbm_traffic_pmc.txt": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) -- python3 bench.py --steps 1 --warmup 0 "
                           "--cpu-sample 0 --e2e-steps 0 --streams 1 --stagger-steps 0   (3 passes: 2 set-up + 1 step; counters are SUMS over the listed calls, KB "
                           "as reported; FETCH_SIZE doubled in the 'corrected' column per MI355X_MICROARCH.md, HBM section)",
    "sq_counters.txt": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM "
                       "SQ_INSTS_LDS -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --e2e-steps 0 --streams 1 --stagger-steps 0   (sums over the calls)",
}
for name, cmd in what.items():
    p = os.path.join(src, name)
    if not os.path.exists(p):
        continue
    body = open(p).read()
    with open(os.path.join(dst, "%s_%s" % (tag, name)), "w") as f:
        f.write("# %s\n# MI355X, tools/gpu_round.sh + tools/profile_summary.py\n" % cmd)
        f.write(body)
for name, out in (("bench.json", "bench.json"), ("bench_s1.json", "bench_one_batch.json"), ("bench_config1.json", "bench_config1_one_degree.json"),
                  ("bench_dist1.json", "bench_rccl_library_1rank.json")):
    p = os.path.join(src, name)
    if os.path.exists(p):
        line = [l for l in open(p).read().splitlines() if l.startswith("{")][-1]
        json.loads(line)
        open(os.path.join(dst, "%s_%s" % (tag, out)), "w").write(line + "\n")
print("stamped", sorted(f for f in os.listdir(dst) if f.startswith(tag)))
