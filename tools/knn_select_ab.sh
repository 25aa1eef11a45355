#!/bin/bash
# A/B of two library builds (agplace_amd/lib/variants/libA.so, libB.so) on one box: rocprofv3 average kernel times of tools/knn_bench.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for v in A B; do
  rm -rf /tmp/kab_$v; AGP_HIP_LIB=$R/agplace_amd/lib/variants/lib$v.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kab_$v -o t -- python3 $R/tools/knn_bench.py --prec 4 --reps 20 > /dev/null 2>&1
  python3 - "$v" "$(find /tmp/kab_$v -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[2])):
    if "select_rerank" in r["Name"] or "coarse_f16_w4" in r["Name"]:
        print("lib" + sys.argv[1], r["Name"].split("(")[0][-40:], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
done; done
