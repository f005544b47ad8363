# A/B of library variants on another bench workload: ARGS="--ppd 2048 --plt 1" bash scripts/ab_c3.sh "" _variant
for v in "$@"; do
  echo "== lib$v" >> gpurun_out/ab_c3.log
  ZD_LIB_PATH=$PWD/zeldovich_plt_amd/csrc/build/libzeldovich_hip$v.so python bench.py $ARGS --steps 4 --warmup 1 --no-cpu-baseline --no-isolated 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k = j['kernels']
        print('step %.1f ms  z %.1f  y %.1f  x %.1f' % (j['ms_per_step'], k['z_stage']['ms_per_step'], k['k_yfft']['ms_per_step'], k['k_xfft']['ms_per_step']))
" >> gpurun_out/ab_c3.log
done
cat gpurun_out/ab_c3.log
