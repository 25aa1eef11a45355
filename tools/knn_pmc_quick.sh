#!/bin/bash
# quick SQ counter passes of tools/knn_bench.py (coarse kernel analysis); output: gpurun_out/knnq_<pass>/
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
run() { d=$R/gpurun_out/knnq_$1; shift; rm -rf $d; mkdir -p $d
  timeout 200 rocprofv3 "$@" --output-format csv -d $d -o c -- python3 $R/tools/knn_bench.py --prec 4 --reps 5 > $d/stdout.txt 2>&1; }
run a --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run b --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES
run c --pmc SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_ANY
cd $R
python3 - <<'PY'
import csv, glob, collections
for p in "abc":
    fs = glob.glob(f"gpurun_out/knnq_{p}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(p, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"][:60]
        if "coarse" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        print(p, k)
        for c, v in d.items():
            print(f"    {c}: {v / n[(k, c)]:.4g} per launch ({n[(k, c)]} launches)")
PY
