set -x
mkdir -p gpurun_out/r3e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "attention or encoder or long or g1 or g2 or g3 or g5 or g10" > gpurun_out/r3e/pytest_sel.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/r3e/pytest_sel.log
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-alt > gpurun_out/r3e/bench.json 2> gpurun_out/r3e/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r3e/bench.json'))
print(d['value'], d['ms_per_step'], {k: round(v['ms_per_step'],3) for k,v in d['kernels'].items()}, d['two_streams'])
PY
timeout -k 10 300 python3 bench.py --clip-seconds 600 --batch 4 --steps 3 --warmup 1 --no-cpu-baseline --no-alt > gpurun_out/r3e/bench_10min.json 2> gpurun_out/r3e/bench_10min.err; echo "bench10 rc=$?"
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r3e/bench_10min.json'))
print(d['value'], d['ms_per_step'], {k: round(v['ms_per_step'],3) for k,v in d['kernels'].items()})
PY
