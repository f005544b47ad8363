# round 4: the large / composite configurations of DESIGN §4 on the final build (zd_generate with the NULL sink)
#   bash scripts/r04_big_runs.sh > gpurun_out/r04_big_runs.log 2>&1
cd $GRAFT_REPO_ROOT
F='WARNING|spectrum|extrap|make sure|that goes|it\) to|sigma|amdgpu.ids|^$'
timeout -k 10 200 python scripts/run_8192.py 8192 2.0 auto 2 2 1 2>&1 | grep -Ev "$F" || exit 1
timeout -k 10 100 python scripts/run_8192.py 8192 2.0 2>&1 | grep -Ev "$F" || exit 1
timeout -k 10 400 python scripts/np2_times.py 6912:1:plt 6912 6912:2 2>&1 | grep -Ev "$F" || exit 1
