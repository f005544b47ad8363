# round 5, GPU probe 1: the store side of a fused PLT generator + z FFT (go / no-go, VERDICT r4 #1b)
#   bash scripts/r05_probe1.sh > gpurun_out/r05_probe1.log 2>&1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
hipcc -w --offload-arch=gfx950 -O3 scripts/microbench/fused_plt_store.hip -o gpurun_out/fused_plt_store && timeout -k 5 300 gpurun_out/fused_plt_store 96 || exit 1
rm -f gpurun_out/fused_plt_store
