# round 4, GPU probe 3: (a) ZD_qdensity = 1 on a composite grid, six-field store against the plain run; (b) PPD=8192 PLT after the
# max_disp change (k_xfft_two spills); (c) PPD=1024 slab-size scan (Infinity Cache hand-off in the real pipeline)
#   bash scripts/r04_probe3.sh > gpurun_out/r04_probe3.log 2>&1
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python scripts/np2_times.py 3456 3456:1:dens 2880 2880:1:dens || exit 1
timeout -k 10 300 python scripts/run_8192.py 8192 2.0 auto 2 2 1 || exit 1
bash scripts/r04_probe2.sh
