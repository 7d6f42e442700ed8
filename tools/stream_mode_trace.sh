#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2p; rm -rf $O; mkdir -p $O
for n in 1 2; do
  rocprofv3 --kernel-trace --stats -d $O/s$n --output-format csv -- python3 tools/stream_mode_trace.py 6 10 $n 20 > $O/s$n.log 2>&1
  grep "batch" $O/s$n.log
  f=$(ls $O/s$n/*/*kernel_stats.csv | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:12]:
    print("   %-90s calls %5s total %8.2f ms avg %8.1f us" % (r['Name'][:90], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
done
