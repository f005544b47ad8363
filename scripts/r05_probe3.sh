# round 5, GPU probe 3: where the fused PLT Z stage's time goes (tuning library ablations) and the x stage's pairing distance
cd $GRAFT_REPO_ROOT
export ZD_LIB_PATH=$GRAFT_REPO_ROOT/zeldovich_plt_amd/csrc/build/libzeldovich_hip_tuning.so
run() {
  env "$@" timeout -k 10 200 python bench.py --ppd 2048 --plt 1 --steps 2 --warmup 1 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', 'step', round(d['ms_per_step'],1), {k: round(v['ms_per_step'],1) for k,v in d['kernels'].items() if k in ('z_stage','k_yfft','k_xfft')})" || exit 1
}
run ZD_PRUNE=7
echo "== Z stage ablations: 15 = draws only (no mode arithmetic), 135 = no transform, 23 = no stores, 151 = neither, 159 = draws + barriers only =="
for p in 15 135 23 151 159; do run ZD_PRUNE=$p; done
echo "== x stage: positions between the two workgroups of a plane group on an XCD =="
for d in 1 2 4 8 16 64; do run ZD_PRUNE=7 ZD_XQ_DIST=$d; done
