# round 4, GPU probe 2: does a small folded-input slab (generator -> z FFT hand-off inside the Infinity Cache) shorten the Z stage
# of the REAL pipeline?  PPD=1024 PLT (one row of folded inputs = 50 MB) and PPD=1024 ZA (a group of 8 rows = 268 MB), tuning
# library, ZD_SLAB_MB.     bash scripts/r04_probe2.sh > gpurun_out/r04_probe2.log 2>&1
cd $GRAFT_REPO_ROOT
export ZD_LIB_PATH=$GRAFT_REPO_ROOT/zeldovich_plt_amd/csrc/build/libzeldovich_hip_tuning.so
for plt in 1 0; do
for mb in 50 100 200 400 1500; do
  ZD_SLAB_MB=$mb timeout -k 10 200 python bench.py --ppd 1024 --plt $plt --steps 5 --warmup 2 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('PPD=1024 plt=$plt SLAB_MB=$mb step %.2f ms' % d['ms_per_step'], {k: round(v['ms_per_step'],2) for k,v in d['kernels'].items()})" || exit 1
done
done
