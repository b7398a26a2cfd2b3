cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "default:X=1" "blitsize0:GPU_FORCE_BLIT_COPY_SIZE=0" "sdma1:HSA_ENABLE_SDMA=1" "engine2:GPU_BLIT_ENGINE_TYPE=2" "cpdma0:GPU_CP_DMA_COPY_SIZE=0"; do
  label=${cfg%%:*}; envs=${cfg#*:}
  rm -rf /tmp/prof_$label
  export $envs
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$label -- python3 $R/bench.py --steps 6 --warmup 2 --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 --streams 1 > /tmp/prof_$label.log 2>&1
  unset ${envs%%=*}
  f=$(find /tmp/prof_$label -name "*kernel_stats.csv" | head -1)
  echo "== $label: $(grep -c . $f) kernels; copyBuffer: $(grep -i copyBuffer $f | cut -d, -f1-4)"; grep '"value"' /tmp/prof_$label.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['stage_ms'].get('geometry'))"
done
