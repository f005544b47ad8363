#!/bin/bash
# A/B of library variants on timed runs of scripts/np2_times.py:  SPECS="3456 5120" bash scripts/ab_np2.sh "" _variant
for v in "$@"; do
  echo "== lib$v" >> gpurun_out/ab_np2.log
  ZD_LIB_PATH=$PWD/zeldovich_plt_amd/csrc/build/libzeldovich_hip$v.so python scripts/np2_times.py $SPECS 2>/dev/null | grep k_cutoff >> gpurun_out/ab_np2.log
done
cat gpurun_out/ab_np2.log
