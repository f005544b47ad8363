# round 4, GPU probe 10: paced run-ahead — generator slabs of the next pass at one workgroup per CU beside the x phases (tuning library)
cd $GRAFT_REPO_ROOT
export ZD_LIB_PATH=$GRAFT_REPO_ROOT/zeldovich_plt_amd/csrc/build/libzeldovich_hip_tuning.so
run() { timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'step', round(d['ms_per_step']), {k: round(v['ms_per_step']) for k,v in d['kernels'].items()})"; }
run "default" || exit 1
ZD_Y_SLABS=34 ZD_AHEAD_PER_GROUP=1 run "slabs=34 ahead/group=1" || exit 1
ZD_Y_SLABS=34 ZD_AHEAD_PER_GROUP=1 ZD_AHEAD_PRIO=3 run "slabs=34 ahead/group=1 prio=3" || exit 1
ZD_Y_SLABS=34 ZD_AHEAD_PER_GROUP=1 ZD_AHEAD_PRIO=1 run "slabs=34 ahead/group=1 prio=1" || exit 1
run "default again" || exit 1
