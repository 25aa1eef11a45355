cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/wgp1 $R/gpurun_out/wgp2 $R/gpurun_out/wgp3
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/wgp1 -o c -- python3 $R/tools/wgrad_bench.py --iters 2 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM --output-format csv -d $R/gpurun_out/wgp2 -o c -- python3 $R/tools/wgrad_bench.py --iters 2 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/wgp3 -o c -- python3 $R/tools/wgrad_bench.py --iters 2 > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv,collections,glob
def load(d):
    f=glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv",recursive=True)[0]
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if "wgrad_f16" not in n: continue
        agg[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg
for d in ("wgp1","wgp2","wgp3"):
    a=load(d)
    for g,c in a.items():
        print(d,g,{k:round(sum(v)/len(v),1) for k,v in c.items()})
PY
