#!/bin/bash
# rocprofv3 kernel stats of the packed forward at the reference's operating point (run on the GPU box from the repo root):
# 8 packs of 32 pairs of 2-6 s utterances, the library's own per-kernel events on (single stream) -- tools/packed_profile.py.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/packed_prof
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 tools/packed_profile.py 32 > $O/packed_profile.txt 2>&1 || exit 1
cp $(ls $O/stats/*/*kernel_stats.csv) $O/kernel_stats.csv
grep -v amdgpu $O/packed_profile.txt
head -12 $O/kernel_stats.csv | cut -c1-160
find $O -name "*kernel_trace.csv" -size +20M -delete
