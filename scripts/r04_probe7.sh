cd $GRAFT_REPO_ROOT
export ZD_LIB_PATH=$GRAFT_REPO_ROOT/zeldovich_plt_amd/csrc/build/libzeldovich_hip_tuning.so
for nk in 0 1 0; do
  if [ $nk = 1 ]; then export ZD_GEN_NO_KZPAIR=1; else unset ZD_GEN_NO_KZPAIR; fi
  timeout -k 10 300 python bench.py --ppd 4096 --plt 1 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('NO_KZPAIR=$nk PPD=4096 PLT', 'step', round(d['ms_per_step']), {k: round(v['ms_per_step']) for k,v in d['kernels'].items()}, 'isolated gen', round(d['kernels_isolated']['k_gen']['ms_per_step']))" || exit 1
done
