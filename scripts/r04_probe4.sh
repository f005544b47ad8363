# round 4, GPU probe 4: the kz-paired PLT generator (genf_tile_kz) against the unpaired one (tuning library, ZD_GEN_NO_KZPAIR=1)
#   bash scripts/r04_probe4.sh > gpurun_out/r04_probe4.log 2>&1
cd $GRAFT_REPO_ROOT
export ZD_LIB_PATH=$GRAFT_REPO_ROOT/zeldovich_plt_amd/csrc/build/libzeldovich_hip_tuning.so
for nk in 0 1; do
  if [ $nk = 1 ]; then export ZD_GEN_NO_KZPAIR=1; fi
  for args in "--ppd 2048 --plt 1" "--ppd 4096 --plt 1"; do
  timeout -k 10 300 python bench.py $args --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('NO_KZPAIR=$nk $args', 'R', d['config']['stream_factor'], 'step', round(d['ms_per_step']), {k: round(v['ms_per_step']) for k,v in d['kernels'].items()}, 'isolated', {k: round(v['ms_per_step']) for k,v in d['kernels_isolated'].items()})" || exit 1
  done
done
