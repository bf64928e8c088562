mkdir -p gpurun_out/r4q
F="--no-cpu-baseline --no-stock-baseline --no-train-leg --no-batch4 --no-target-4k --no-trained-like --no-configs --steps 5 --warmup 2"
for b in 9 10 17 25; do
  ( timeout 300 python bench.py $F --batch $b ) 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels']
print('batch', $b, 'clips/s', d['value'], 'ms', d['ms_per_step'], 'F4 frac', k['conv3x3_winograd_f4_kernel']['frac_of_mfma_peak'], 'dcn', k['dcnv2_fwd[dcn_tapwin_fwd_kernel]']['frac_of_mfma_peak'])
" >> gpurun_out/r4q/batch_sweep.log 2>&1
done
cat gpurun_out/r4q/batch_sweep.log
