mkdir -p gpurun_out/r5
export PYTHONUNBUFFERED=1
TAG=${1:-tr1}
( timeout 900 python -m pytest tests/test_gpu_wgrad.py tests/test_gpu_train.py -q -x 2>&1 | tail -8 ) > gpurun_out/r5/test_$TAG.log 2>&1
( timeout 600 python bench.py --mode train --train-steps 20 --no-cpu-baseline --no-stock-baseline ) > gpurun_out/r5/bench_${TAG}_train.log 2>&1
tail -8 gpurun_out/r5/test_$TAG.log; python - <<'PY'
import json,sys
tag=sys.argv[1] if len(sys.argv)>1 else 'tr1'
import glob
f=sorted(glob.glob('gpurun_out/r5/bench_*_train.log'))[-1]
d=json.loads([l for l in open(f) if l.startswith('{')][-1])
print(f, d['value'], d['ms_per_step'], d.get('iters_per_sec'))
for k,v in list(d['kernels'].items())[:14]: print('   ',k, v.get('launches_per_step'), v.get('ms_per_step'))
PY
