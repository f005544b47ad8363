# round 4, GPU probe 1 (through gpurun): go / no-go numbers for a fused generator + z FFT, and the two-store A/B
#   bash scripts/r04_probe1.sh > gpurun_out/r04_probe1.log 2>&1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
hipcc -w --offload-arch=gfx950 -O3 scripts/microbench/z_store_merge.hip -o gpurun_out/z_store_merge && timeout -k 5 120 gpurun_out/z_store_merge || exit 1
hipcc -w --offload-arch=gfx950 -O3 scripts/microbench/mall_handoff.hip -o gpurun_out/mall_handoff && timeout -k 5 200 gpurun_out/mall_handoff || exit 1
rm -f gpurun_out/z_store_merge gpurun_out/mall_handoff
echo "== generator alone (isolated pass, serial_z) against its persistent workgroups per CU =="
export ZD_LIB_PATH=$GRAFT_REPO_ROOT/zeldovich_plt_amd/csrc/build/libzeldovich_hip_tuning.so
for w in 1 2 3 4; do
  ZD_GEN_WGS=$w timeout -k 10 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('GEN_WGS=$w step', round(d['ms_per_step']), {k: round(v['ms_per_step']) for k,v in d['kernels'].items()}, 'isolated', {k: round(v['ms_per_step']) for k,v in d['kernels_isolated'].items()})" || exit 1
done
unset ZD_LIB_PATH
echo "== one GPU, two stores: R = 8 serial (default) against R = 16 with the Z stage of pass p+1 beside the XY stages of pass p =="
for args in "" "--stream 16" "--stream 16 --two-stores"; do
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-isolated $args 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$args', 'R', d['config']['stream_factor'], 'step', round(d['ms_per_step']), {k: round(v['ms_per_step']) for k,v in d['kernels'].items()})" || exit 1
done
