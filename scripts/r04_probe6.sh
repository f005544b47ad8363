# composite-grid timings after the tree reduction + kz-pair A/B on ONE box (tuning library)
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python scripts/np2_times.py 3456 3456:1:dens 3456:1:plt || exit 1
export ZD_LIB_PATH=$GRAFT_REPO_ROOT/zeldovich_plt_amd/csrc/build/libzeldovich_hip_tuning.so
for nk in 0 1 0 1; do
  if [ $nk = 1 ]; then export ZD_GEN_NO_KZPAIR=1; else unset ZD_GEN_NO_KZPAIR; fi
  timeout -k 10 300 python bench.py --ppd 4096 --plt 1 --steps 1 --warmup 1 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('NO_KZPAIR=$nk PPD=4096 PLT', 'step', round(d['ms_per_step']), {k: round(v['ms_per_step']) for k,v in d['kernels'].items()})" || exit 1
done
