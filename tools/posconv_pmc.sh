#!/bin/bash
# LDS bank conflicts of the positional conv's resident-A kernel (run on the GPU box from the repo root): one PMC pass over three
# bench steps, summary per kernel.  Round 3: SQ_LDS_BANK_CONFLICT = 7.55e7 of SQ_BUSY_CYCLES = 7.57e7 per launch.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/posconv_pmc
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-two-streams --no-packed > $O/pmc.log 2>&1 || exit 1
python3 tools/pmc_summary.py "$O/pmc/*/*counter_collection.csv" > $O/summary.txt
grep -E "pos_conv_resident|F16_S6_S3" $O/summary.txt
find $O -name "*kernel_trace.csv" -size +20M -delete
find $O -name "*counter_collection.csv" -size +30M -delete
