#!/bin/bash
# same-box comparison: the tree of the end of round 3 (git worktree build/r03, its own bench.py and library) against this tree
cd "$(dirname "$0")/.."
O=gpurun_out/r04v; mkdir -p $O
run() {  # label, dir, streams
  (cd $2 && timeout 300 python bench.py --steps 40 --warmup 6 --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 --streams $3 > /tmp/ab.json 2> /tmp/ab.err || tail -5 /tmp/ab.err)
  python - "$1" $3 <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
s = d["stage_ms"]
print(sys.argv[1], "streams", sys.argv[2], "value %d  ms/step %.3f frac %.3f alone %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_alone"]), " ".join("%s %.2f" % (k, v) for k, v in s.items()), flush=True)
PY
}
{
for rep in 1 2 3; do
run r03 build/r03 2
run r04 . 2
run r03 build/r03 1
run r04 . 1
done
} 2>&1 | tee $O/ab.txt
