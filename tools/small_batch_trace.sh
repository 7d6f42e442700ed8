#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2s; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/s --output-format csv -- python3 tools/stream_mode_trace.py ${1:-2} ${2:-5} 1 30 > $O/s.log 2>&1
grep "batch" $O/s.log
f=$(ls $O/s/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel time %.2f ms over %d forwards" % (tot/1e6, 33))
for r in rows[:22]:
    print("   %-100s calls %5s total %7.2f ms (%4.1f %%) avg %7.1f us" % (r['Name'][:100], r['Calls'], float(r['TotalDurationNs'])/1e6, 100*float(r['TotalDurationNs'])/tot, float(r['AverageNs'])/1e3))
PY
