# Z-stage time against the number of persistent generator workgroups per CU (tuning library: ZD_GEN_WGS)
#   bash scripts/zstage_wgs.sh [bench.py arguments, e.g. --ppd 2048 --plt 1]
cd $GRAFT_REPO_ROOT
export ZD_LIB_PATH=$GRAFT_REPO_ROOT/zeldovich_plt_amd/csrc/build/libzeldovich_hip_tuning.so
for w in 1 2 3 4; do
  ZD_GEN_WGS=$w timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-isolated "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('GEN_WGS=$w', round(d['ms_per_step']), {k: round(v['ms_per_step']) for k,v in d['kernels'].items()})" || exit 1
done
