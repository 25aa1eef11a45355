#!/usr/bin/env python3
"""gpurun_out/<tag>_knn{,_fetch,_write,_sq} (tools/collect_knn_pmc.sh) -> profiles/<tag>_pmc_knn.json + <tag>_knn_kernel_stats.csv.
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE reports half of wide coalesced reads,
MI355X_MICROARCH.md, HBM section); counters in separate passes."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_inputs  # noqa: E402


def short(name):
    for k in ("coarse_f16_w4_kernel", "coarse_f16_kernel", "select_rerank_kernel", "q_prep_kernel", "db_prep_kernel"):
        if k in name:
            return k
    return None


def counters(tag, sub):
    d = os.path.join(ROOT, "gpurun_out", f"{tag}_{sub}")
    fs = glob.glob(os.path.join(d, "*", "*_counter_collection.csv")) + glob.glob(os.path.join(d, "*_counter_collection.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        k = short(r["Kernel_Name"])
        if k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    out = os.path.join(ROOT, "profiles")
    d = os.path.join(ROOT, "gpurun_out", f"{tag}_knn")
    fs = glob.glob(os.path.join(d, "*", "*_kernel_stats.csv")) + glob.glob(os.path.join(d, "*_kernel_stats.csv"))
    stats = max(fs, key=os.path.getmtime)
    shutil.copy(stats, os.path.join(out, f"{tag}_knn_kernel_stats.csv"))
    avg_ns = {short(r["Name"]): float(r["AverageNs"]) for r in csv.DictReader(open(stats)) if short(r["Name"])}
    fe, wr, sq = counters(tag, "knn_fetch"), counters(tag, "knn_write"), counters(tag, "knn_sq")
    summ = {"note": "per-launch averages over `tools/knn_bench.py --prec 4 --reps 20` (100k x 256 database, 4096 queries, k = 20); "
                    "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction); counters from separate rocprofv3 --pmc passes",
            "kernels": {}}
    for k in fe:
        f, w, m = fe[k]["FETCH_SIZE"], wr[k]["WRITE_SIZE"], sq[k]
        rec = {"launches_profiled": len(f), "avg_us": avg_ns.get(k, 0) / 1e3, "fetch_size_kib_avg": sum(f) / len(f),
               "write_size_kib_avg": sum(w) / len(w), "hbm_mb_per_launch": (2 * sum(f) / len(f) + sum(w) / len(w)) * 1024 / 1e6}
        if m.get("GRBM_GUI_ACTIVE"):
            rec["mfma_busy_frac"] = sum(m["SQ_VALU_MFMA_BUSY_CYCLES"]) / (sum(m["GRBM_GUI_ACTIVE"]) / 8 * 1024)
            rec["sq_wait_inst_lds_per_wave_cycle"] = sum(m["SQ_WAIT_INST_LDS"]) / max(sum(m["SQ_WAVE_CYCLES"]), 1)
            rec["lds_insts_per_mfma"] = sum(m["SQ_INSTS_LDS"]) / max(sum(m["SQ_INSTS_MFMA"]), 1)
        summ["kernels"][k] = rec
    summ["csrc_sha16"] = bench_inputs.kernel_source_sha16(ROOT)
    json.dump(summ, open(os.path.join(out, f"{tag}_pmc_knn.json"), "w"), indent=1)
    print(json.dumps(summ, indent=1))


if __name__ == "__main__":
    main()
