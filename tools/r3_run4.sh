set -x
mkdir -p gpurun_out/r3c
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "row_max or non_finite or configs4 or g10" > gpurun_out/r3c/pytest_sel.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r3c/pytest_sel.log
timeout -k 10 400 python3 tools/inflight_bench.py 300 --threads > gpurun_out/r3c/inflight_bench.log 2>&1; echo "inflight rc=$?"
cat gpurun_out/r3c/inflight_bench.log | grep -v amdgpu.ids
