# PMC counters of the field-store y stage (tuning library): HBM bytes and L2 hit rate per launch
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export ZD_LIB_PATH=$R/zeldovich_plt_amd/csrc/build/libzeldovich_hip_tuning.so
export VARIANTS="${VARIANTS:-full}"
cd /tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/yfpmc_$tag -- python3 $R/scripts/yf_ablate.py > $R/gpurun_out/yfpmc_$tag.log 2>&1 || { tail -5 $R/gpurun_out/yfpmc_$tag.log; }
done
cd $R
python3 - <<PY
import csv, glob, collections, os
R=os.environ['GRAFT_REPO_ROOT']
for d in sorted(glob.glob(f"{R}/gpurun_out/yfpmc_*/")):
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
        for row in csv.DictReader(open(f)):
            k=row["Kernel_Name"].split("(")[0].replace("void ","").strip()[:40]
            a=agg[k][row["Counter_Name"]]; a[0]+=1; a[1]+=float(row["Counter_Value"])
        for k,v in agg.items():
            if "yfft_f" in k or "xfft" in k or "zfft_f" in k:
                print(k, {c:(n, round(t/n,1)) for c,(n,t) in v.items()})
PY
