set -x
mkdir -p gpurun_out/r3a
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/r3a/pytest.rc
tail -15 gpurun_out/r3a/pytest.log
timeout -k 10 300 python3 tools/cli_bench.py 2000 > gpurun_out/r3a/cli_bench.log 2>&1; echo "cli rc=$?"
grep -E "inflight|corpus|identical|Error|error" gpurun_out/r3a/cli_bench.log | tail -12
timeout -k 10 240 python3 tools/power_probe.py gpurun_out/r3a/power 5 > gpurun_out/r3a/power_probe.log 2>&1; echo "power rc=$?"
tail -16 gpurun_out/r3a/power_probe.log
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3a/hg_trace -- python3 tools/hipgraph_trace.py > gpurun_out/r3a/hg_trace.log 2>&1; echo "hg rc=$?"
tail -3 gpurun_out/r3a/hg_trace.log
python3 tools/hipgraph_trace.py --analyse gpurun_out/r3a/hg_trace > gpurun_out/r3a/hg_analysis.txt 2>&1
head -8 gpurun_out/r3a/hg_analysis.txt
find gpurun_out/r3a/hg_trace -name "*.csv" -size +20M -delete
