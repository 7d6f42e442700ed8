#!/bin/bash
# End-of-round evidence run (on the GPU box, from the repo root): bench lines, rocprofv3 kernel stats of the same
# command, and the PMC passes (one counter set per pass, --kernel-trace only) that profiles/ summarises.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/final
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 bench.py --steps 5 --no-cpu-baseline --no-alt --no-two-streams --no-packed > $O/stats.log 2>&1 || exit 1
echo "stats done"
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$n --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-two-streams --no-packed > $O/pmc_$n.log 2>&1 || exit 1
  python3 tools/pmc_summary.py "$O/pmc_$n/*/*counter_collection.csv" > $O/pmc_${n}_summary.txt
  echo "pmc $n done"
done
python3 tools/pmc_traffic.py gemm_f16x3_dma_kernel $(ls $O/pmc_FETCH_SIZE/*/*counter_collection.csv) $(ls $O/pmc_WRITE_SIZE/*/*counter_collection.csv) $O/gemm_f16x3_traffic.json \
  --exclude "<0, false, 2, 2, 2," --exclude "<4, false, 8, 1," --workload 30sx32
python3 tools/pmc_traffic.py attention_f16x3_kernel $(ls $O/pmc_FETCH_SIZE/*/*counter_collection.csv) $(ls $O/pmc_WRITE_SIZE/*/*counter_collection.csv) $O/attention_f16x3_traffic.json --workload 30sx32
cp $(ls $O/stats/*/*kernel_stats.csv) $O/kernel_stats.csv
# 10 min x 4: PMC traffic of the attention kernel at that shape (bench.py keys roofline.traffic by workload)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $O/pmc10_$c --output-format csv -- python3 bench.py --clip-seconds 600 --batch 4 --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-two-streams --no-packed > $O/pmc10_$c.log 2>&1 || exit 1
  echo "pmc 10min $c done"
done
python3 tools/pmc_traffic.py attention_f16x3_kernel $(ls $O/pmc10_FETCH_SIZE/*/*counter_collection.csv) $(ls $O/pmc10_WRITE_SIZE/*/*counter_collection.csv) $O/attention_f16x3_traffic_10minx4.json --workload 10minx4
python3 tools/pmc_traffic.py gemm_f16x3_dma_kernel $(ls $O/pmc10_FETCH_SIZE/*/*counter_collection.csv) $(ls $O/pmc10_WRITE_SIZE/*/*counter_collection.csv) $O/gemm_f16x3_traffic_10minx4.json --exclude "<4, false, 8, 1," --workload 10minx4
# the bench lines LAST: bench.py copies roofline.traffic from profiles/*_traffic.json, so the files of THIS call go there first
# (the round prefix is the newest one present in profiles/)
R=${LOCO_ROUND:-$(ls profiles | sed -n 's/^\(r[0-9][0-9]\)_bench_30sx32.json$/\1/p' | sort | tail -1)}  # LOCO_ROUND=r04 for a round without a bench file yet
cp $O/gemm_f16x3_traffic.json profiles/${R}_gemm_f16x3_traffic.json
cp $O/attention_f16x3_traffic.json profiles/${R}_attention_f16x3_traffic.json
cp $O/attention_f16x3_traffic_10minx4.json profiles/${R}_10minx4_attention_f16x3_traffic.json
cp $O/gemm_f16x3_traffic_10minx4.json profiles/${R}_10minx4_gemm_f16x3_traffic.json
python3 bench.py > $O/bench_30sx32.json 2> $O/bench_30sx32.err || exit 1
echo "bench 30sx32 done" && tail -c 400 $O/bench_30sx32.json
python3 bench.py --clip-seconds 600 --batch 4 --steps 3 --warmup 1 --no-cpu-baseline --no-alt > $O/bench_10minx4.json 2> $O/bench_10minx4.err || exit 1
echo "bench 10minx4 done"
find $O -name "*kernel_trace.csv" -size +20M -delete
find $O -name "*counter_collection.csv" -size +30M -delete
