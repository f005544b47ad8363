# round 5, GPU probe 7: x stage of the interleaved rows with all four planes of a group per workgroup (A/B, tuning library)
cd $GRAFT_REPO_ROOT
export ZD_LIB_PATH=$GRAFT_REPO_ROOT/zeldovich_plt_amd/csrc/build/libzeldovich_hip_tuning.so
run() {
  env "$@" timeout -k 10 200 python bench.py --ppd 2048 --plt 1 --steps 2 --warmup 1 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', 'step', round(d['ms_per_step'],1), {k: round(v['ms_per_step'],1) for k,v in d['kernels'].items() if k in ('z_stage','k_yfft','k_xfft')})" || exit 1
}
run ZD_X=0
run ZD_XQ_NP4=1
run ZD_X=0
run ZD_XQ_NP4=1
timeout -k 10 200 python scripts/r05_fz_check.py 1024 2>&1 | grep "worst\|OK"
ZD_XQ_NP4=1 timeout -k 10 200 python scripts/r05_fz_check.py 1024 2>&1 | grep "worst\|OK"
