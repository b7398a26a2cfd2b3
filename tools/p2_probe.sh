#!/bin/bash
# development: phase stamps and launch-shape variants of wb_k_streamer_pairs2 on the GPU box
cd "$(dirname "$0")/.."
O=gpurun_out/p2probe; mkdir -p $O
WB_HIP_LIBRARY=$PWD/wavebreaking_amd/csrc/libwb_hip_stamps.so WB_DEBUG_STAMPS=1 WB_DEBUG_TIMING=1 python bench.py --steps 1 --warmup 0 --cpu-sample 0 --e2e-steps 0 --streams 1 2>&1 | grep -E "stamps|pairs2:" | tail -4 > $O/stamps.txt
for opts in "p2_split=1" "p2_split=2" "pairs_lds=0"; do
  WB_BENCH_OPTS=$opts python bench.py --steps 10 --warmup 2 --cpu-sample 0 --e2e-steps 0 --streams 1 > /tmp/p2.json 2>/tmp/p2.err || tail -3 /tmp/p2.err
  python - "$opts" <<'PY' >> $O/variants.txt
import json, sys
d = json.load(open("/tmp/p2.json")); s = d["stage_ms"]
print(sys.argv[1], "ms/step %.2f" % d["ms_per_step"], "pairs %.3f prep %.3f f1 %.3f" % (s["streamer_pairs"], s["streamer_prep"], s["streamer_f1"]))
PY
done
cat $O/stamps.txt $O/variants.txt
