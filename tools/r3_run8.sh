mkdir -p gpurun_out/r3d
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( echo "=== default"; timeout -k 10 200 python3 tools/enqueue_cost.py 2>&1 | grep -v amdgpu.ids
echo "=== LOCO_GEMM_NOSPLITK=1"; LOCO_GEMM_NOSPLITK=1 timeout -k 10 200 python3 tools/enqueue_cost.py 2>&1 | grep -v amdgpu.ids
echo "=== GPU_MAX_HW_QUEUES=16 LOCO_GEMM_NOSPLITK=1"; GPU_MAX_HW_QUEUES=16 LOCO_GEMM_NOSPLITK=1 timeout -k 10 200 python3 tools/enqueue_cost.py 2>&1 | grep -v amdgpu.ids
echo "=== AMD_LOG_LEVEL=0 HIP_LAUNCH_BLOCKING unset; DEBUG_HIP_GRAPH? none" ) | tee gpurun_out/r3d/enqueue_cost.log
