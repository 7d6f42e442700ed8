#!/bin/bash
# SQ counter passes over tools/attn_bench.py (run on the GPU box from the repo root): wave time split, MFMA/VALU overlap, LDS conflicts.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/attn_pmc1 gpurun_out/attn_pmc2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d gpurun_out/attn_pmc1 --output-format csv -- python3 tools/attn_bench.py > gpurun_out/attn_pmc1.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d gpurun_out/attn_pmc2 --output-format csv -- python3 tools/attn_bench.py > gpurun_out/attn_pmc2.log 2>&1
python3 tools/pmc_summary.py "gpurun_out/attn_pmc1/*/*counter_collection.csv" "gpurun_out/attn_pmc2/*/*counter_collection.csv" | grep -v WSGR
