# One rocprofv3 --pmc pass with SQ issue counters for the default bench workload (one step): where the waves' cycles go per kernel.
#   gpurun -- 'TAG=r03d bash scripts/gpu_pmc_sq.sh'   -> gpurun_out/pmc_<TAG>_SQ_summary.csv
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r03d}
ARGS=${ARGS:-}
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_SQ -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-isolated $ARGS > $R/gpurun_out/pmc_${TAG}_SQ.log 2>&1 || { tail -5 $R/gpurun_out/pmc_${TAG}_SQ.log; exit 1; }
cd $R
python3 - <<PY
import csv, glob, collections, os
R=os.environ['GRAFT_REPO_ROOT']; TAG="$TAG"
f=glob.glob(f"{R}/gpurun_out/pmc_{TAG}_SQ/**/*counter_collection.csv", recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for row in csv.DictReader(open(f)):
    k=row["Kernel_Name"].split("<")[0].split("(")[0].replace("void ","").strip()
    agg[k][row["Counter_Name"]]+=float(row["Counter_Value"])
    if row["Counter_Name"]=="SQ_WAVE_CYCLES": n[k]+=1
names=["SQ_WAVE_CYCLES","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_INSTS_VALU"]
with open(f"{R}/gpurun_out/pmc_{TAG}_SQ_summary.csv","w") as o:
    o.write("kernel,launches,"+",".join(names)+",wait_any_frac,wait_inst_frac,active_frac,valu_frac_of_wave_cycles\n")
    for k,v in agg.items():
        if not k.startswith("k_"): continue
        wc=v["SQ_WAVE_CYCLES"] or 1
        line='"%s",%d,'%(k,n[k])+",".join("%.4g"%v[c] for c in names)+",%.3f,%.3f,%.3f,%.3f"%(v["SQ_WAIT_ANY"]/wc,v["SQ_WAIT_INST_ANY"]/wc,v["SQ_ACTIVE_INST_ANY"]/wc,v["SQ_ACTIVE_INST_VALU"]/wc)
        o.write(line+"\n"); print(line)
PY
