mkdir -p gpurun_out/r3e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "attention or g2 or g3 or long" > gpurun_out/r3e/pytest_sel2.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r3e/pytest_sel2.log
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-alt > gpurun_out/r3e/bench2.json 2> gpurun_out/r3e/bench2.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r3e/bench2.json'))
print(d['value'], d['ms_per_step'], {k: round(v['ms_per_step'],3) for k,v in d['kernels'].items()}, d['two_streams'])
PY
