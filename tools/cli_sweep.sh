mkdir -p gpurun_out/r3i
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
LOCO_EXTRACT_PROFILE=1 timeout -k 10 400 python3 tools/cli_bench.py 2000 2>&1 | grep -E "^--inflight|main thread|identical" | tee gpurun_out/r3i/cli_profile.log
