set -x
mkdir -p gpurun_out/r3b
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r3b/pytest.log 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/r3b/pytest.log
cat gpurun_out/parity_figures.jsonl | cut -c1-600
