# round 5, GPU probe 6: three A/Bs on the headline Z stage (PPD=4096 ZA), tuning library
cd $GRAFT_REPO_ROOT
export ZD_LIB_PATH=$GRAFT_REPO_ROOT/zeldovich_plt_amd/csrc/build/libzeldovich_hip_tuning.so
run() {
  env "$@" timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', 'step', round(d['ms_per_step'],1), {k: round(v['ms_per_step'],1) for k,v in d['kernels'].items() if k in ('z_stage','k_yfft','k_xfft','k_gen','k_zfft')})" || exit 1
}
run ZD_X=0
echo "== (i) CU partition: z FFT on n CUs of every XCD, generator (4 workgroups per CU) on the rest =="
for n in 4 6 8 12; do run ZD_CU_SPLIT=$n ZD_GEN_WGS=4; done
run ZD_CU_SPLIT=8 ZD_GEN_WGS=3
echo "== (ii) z FFT with 8 elements per thread (512 threads, half the registers) beside 3 / 4 generator workgroups per CU =="
run ZD_ZFFT_E8=1
run ZD_ZFFT_E8=1 ZD_GEN_WGS=4
