#!/bin/bash
mkdir -p gpurun_out/r6
export PYTHONUNBUFFERED=1
O=gpurun_out/r6
timeout 1200 python -m pytest tests/test_gpu_dcn.py tests/test_gpu_dcn1.py tests/test_gpu_compat_ext.py tests/test_gpu_train.py tests/test_gpu_ddp.py -x -q --tb=short 2>&1 | tail -4 > $O/c24.log
timeout 600 python scripts/r6/train_motion_table.py 2>&1 | grep -v amdgpu.ids | head -6 >> $O/c24.log
timeout 900 python bench.py --no-cpu-baseline --no-stock-baseline --no-configs --no-target-4k --no-batch4 --no-fp32-leg --no-roofline 2>/dev/null | tail -1 > $O/c24_bench.json
cat $O/c24.log; python - <<'PY'
import json
f=json.load(open('gpurun_out/r6/c24_bench.json'))
print(f['value'], f['train'])
print(f.get('trained_like'))
PY
