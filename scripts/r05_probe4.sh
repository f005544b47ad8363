# round 5, GPU probe 4: fused PLT Z stage after the LDS twiddles / no loads in the hot loop: parity, timing, ablations
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python scripts/r05_fz_check.py 1024 2048 2>&1 | grep "worst\|OK\|Error\|error" || exit 1
timeout -k 10 300 python bench.py --ppd 2048 --plt 1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('product step', round(d['ms_per_step'],1), {k: round(v['ms_per_step'],1) for k,v in d['kernels'].items()})" || exit 1
export ZD_LIB_PATH=$GRAFT_REPO_ROOT/zeldovich_plt_amd/csrc/build/libzeldovich_hip_tuning.so
run() {
  env "$@" timeout -k 10 200 python bench.py --ppd 2048 --plt 1 --steps 2 --warmup 1 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', 'step', round(d['ms_per_step'],1), {k: round(v['ms_per_step'],1) for k,v in d['kernels'].items() if k in ('z_stage','k_yfft','k_xfft')})" || exit 1
}
for p in 7 15 135 23 151 159; do run ZD_PRUNE=$p; done
echo "== row pad (elements per plane row) =="
for pad in 6 12 48; do run ZD_PRUNE=7 ZD_PAD=$pad; done
