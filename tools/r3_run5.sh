mkdir -p gpurun_out/r3c
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for q in 8 16; do
  echo "=== GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python3 tools/inflight_bench.py 300 2>&1 | grep -E "one at a time|in flight" 
done | tee gpurun_out/r3c/inflight_hwq.log
