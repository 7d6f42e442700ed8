set -x
mkdir -p gpurun_out/r3d
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 tools/cli_bench.py 2000 > gpurun_out/r3d/cli_bench.log 2>&1; echo "cli rc=$?"
grep -E "^--inflight|corpus|identical|Error|error|Traceback" gpurun_out/r3d/cli_bench.log | tail -12
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "capturable or two_stream or cli or inflight" > gpurun_out/r3d/pytest_sel.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r3d/pytest_sel.log
