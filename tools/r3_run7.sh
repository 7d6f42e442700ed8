mkdir -p gpurun_out/r3d
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "=== LOCO_GEMM_NOSPLITK=1"
LOCO_GEMM_NOSPLITK=1 timeout -k 10 300 python3 tools/inflight_bench.py 300 2>&1 | grep -E "one at a time|in flight" | tee gpurun_out/r3d/inflight_nosplitk.log
