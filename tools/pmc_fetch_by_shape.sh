#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2n
rm -rf $O && mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_encoder.py -x -q -m gpu -k "conv or golden or matches or gemm" > $O/tests.log 2>&1; tail -2 $O/tests.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_FETCH_SIZE --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-two-streams > $O/pmc.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, re
f = sorted(glob.glob("gpurun_out/r2n/pmc_FETCH_SIZE/*/*counter_collection.csv"))[-1]
rows = [(r['Kernel_Name'], float(r['Counter_Value'])) for r in csv.DictReader(open(f)) if 'gemm_f16x3_dma_kernel' in r['Kernel_Name']][-68:]
names = ["conv1","conv2","conv3","conv4","conv5","conv6","featproj","posconv","qkv","qp","out","ffn1","ffn2"]
print(" ".join("%s %.0f |" % (n, rows[i][1]*1024/1e6) for i, n in enumerate(names)))
PY
