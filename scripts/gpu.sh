#!/bin/bash
# build what is stale, check that every library carries the sha of the sources, then run "$@" on the GPU box through gpurun
#   scripts/gpu.sh <timeout s> <log name> '<command>'      -> gpurun_out/<log name>.log (the command's output), .gpurun (gpurun's)
cd /root/repo || exit 1
T=$1; NAME=$2; shift 2
make -C zeldovich_plt_amd/csrc -j6 all testing tuning > /tmp/gpu_sh_make.log 2>&1 || { tail -20 /tmp/gpu_sh_make.log; exit 1; }
make -C zeldovich_plt_amd/csrc -j6 all testing tuning > /tmp/gpu_sh_make2.log 2>&1   # (a second pass: nothing may be left to do)
SHA=$(make -s -C zeldovich_plt_amd/csrc srcsha)
for l in libzeldovich_hip libzeldovich_hip_testing libzeldovich_hip_tuning; do
  [ "$(cat zeldovich_plt_amd/csrc/build/$l.so.srcsha)" = "$SHA" ] || { echo "stale $l"; exit 1; }
done
mkdir -p gpurun_out
/usr/local/graft/bin/gpurun --timeout $T -- "$* > gpurun_out/$NAME.log 2>&1" > gpurun_out/$NAME.gpurun 2>&1
tail -3 gpurun_out/$NAME.gpurun
