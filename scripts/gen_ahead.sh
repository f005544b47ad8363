# Generator run-ahead (tuning library, ZD_Y_AHEAD=1: the generator of pass p+1 fills spare folded-input slabs while the XY stages
# of pass p run) against the default, on bench.py --ppd 2048 --plt 1.  bash scripts/gen_ahead.sh   (through gpurun)
cd $GRAFT_REPO_ROOT
export ZD_LIB_PATH=$GRAFT_REPO_ROOT/zeldovich_plt_amd/csrc/build/libzeldovich_hip_tuning.so
for a in 0 1; do
  if [ $a = 1 ]; then export ZD_Y_AHEAD=1; fi
  timeout -k 10 200 python bench.py --ppd 2048 --plt 1 --steps 3 --warmup 1 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('AHEAD=$a', round(d['ms_per_step'],1), {k: round(v['ms_per_step']) for k,v in d['kernels'].items()})" || exit 1
done
