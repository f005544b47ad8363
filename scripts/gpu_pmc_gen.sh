# hardware counters of the generator alone (no two-stream overlap), PPD=2048 forced R=8
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3-avail list 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|GRBM_[A-Z_]*" | sort -u > $R/gpurun_out/pmc_avail.txt
wc -l $R/gpurun_out/pmc_avail.txt
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F64"; do
i=$((i+1))
ZD_NO_OVERLAP=1 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmcgen_$i -- python3 $R/scripts/run_one.py 2048 8 > $R/gpurun_out/pmcgen_$i.log 2>&1 || tail -3 $R/gpurun_out/pmcgen_$i.log
done
cd $R
python3 - <<'PY'
import csv, glob, collections, os
R=os.environ['GRAFT_REPO_ROOT']
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(int)
for f in glob.glob(f"{R}/gpurun_out/pmcgen_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"].split("<")[0].replace("void ","").strip()
        agg[k][row["Counter_Name"]]+=float(row["Counter_Value"])
for k,v in agg.items():
    if k.startswith("k_"): print(k, {a: "%.4g"%b for a,b in sorted(v.items())})
PY
