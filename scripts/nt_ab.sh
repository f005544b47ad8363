#!/bin/bash
# A/B of non-temporal access bits (zd_kernels.hip ZD_NTBIT) on product flags: build each variant first, e.g.
#   (cd zeldovich_plt_amd/csrc && make variant NAME=nt32 FLAGS=-DZD_NT_FORCE=32)
# then on the GPU box:  bash scripts/nt_ab.sh "" _nt32 _nt128 ""
for v in "$@"; do
  echo "== lib$v" >> gpurun_out/ntab.log
  ZD_LIB_PATH=$PWD/zeldovich_plt_amd/csrc/build/libzeldovich_hip$v.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-isolated 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k = j['kernels']
        print('step %.1f ms  z %.1f  y %.1f  x %.1f' % (j['ms_per_step'], k['z_stage']['ms_per_step'], k['k_yfft']['ms_per_step'], k['k_xfft']['ms_per_step']))
" >> gpurun_out/ntab.log
done
cat gpurun_out/ntab.log
