set -x
mkdir -p gpurun_out/r3a
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/r3a/pytest.rc
tail -5 gpurun_out/r3a/pytest.log
