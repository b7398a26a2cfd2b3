cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/ovraw
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ovraw -- python3 $R/bench.py --steps 30 --warmup 6 --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 > /tmp/ov.log 2>&1
python3 $R/tools/overlap.py /tmp/ovraw | head -60
