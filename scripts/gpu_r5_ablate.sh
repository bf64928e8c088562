mkdir -p gpurun_out/r5
export PYTHONUNBUFFERED=1
LOG=gpurun_out/r5/${1:-f4s_ablations}.log
: > $LOG
for v in "${@:2}"; do
  EDVR_AMD_LIB=edvr_amd/lib/variants/libedvr_amd_$v.so timeout 120 python scripts/bench_f4s_time.py $v 2>&1 | grep -v amdgpu.ids >> $LOG
done
cat $LOG
