#!/usr/bin/env python3
"""Turn the rocprofv3 outputs under gpurun_out/<tag>_* into the committed summaries in profiles/.

    python tools/summarize_profiles.py r01
expects (all produced by `rocprofv3 ... -- python bench.py --no-cpu-baseline --no-knn --graph 0 --streams 1`):
    gpurun_out/<tag>_trace      --kernel-trace --stats
    gpurun_out/<tag>_pmc_fetch  --pmc FETCH_SIZE          (separate pass)
    gpurun_out/<tag>_pmc_write  --pmc WRITE_SIZE          (separate pass)
    gpurun_out/<tag>_pmc_mfma   --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_MFMA
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE reports half of
the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kind(name):
    if "fblock64_kernel" in name:
        return "fblock64_kernel (fused 64-channel BasicBlock: two 3x3 s1 convs per launch)"
    if "igemm_kxr" in name:     # igemm_kxr_kernel and igemm_kxr2_kernel
        return "igemm_kxr_kernel (3x3 s1 convs)"
    if "stem_walk" in name:
        return "stem_walk_kernel (the stem: conv 7x7/2 + BN + ReLU + max-pool, walking along a strip)"
    if "igemm_d16" in name or "stem_pool_lds" in name:
        return "stem kernels (igemm_d16 / stem_pool_lds)"
    if "igemm_s2_kernel" in name:
        return "igemm_s2_kernel (stage entry: 3x3/s2 conv + 1x1/s2 downsample)"
    if "igemm_kernel" in name or "igemm_group_kernel" in name:
        return "igemm_kernel (1x1 / stride-2 convs, kNN coarse pass)"
    return None


def counters(tag, sub):
    d = os.path.join(ROOT, "gpurun_out", f"{tag}_{sub}")
    f = max(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")) + glob.glob(os.path.join(d, "*_counter_collection.csv")), key=os.path.getmtime)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = kind(r["Kernel_Name"])
        if k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    prec = sys.argv[2] if len(sys.argv) > 2 else "2"
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    d = os.path.join(ROOT, "gpurun_out", f"{tag}_trace")
    stats = max(glob.glob(os.path.join(d, "*", "*_kernel_stats.csv")) + glob.glob(os.path.join(d, "*_kernel_stats.csv")), key=os.path.getmtime)
    shutil.copy(stats, os.path.join(out, f"{tag}_bench_kernel_stats.csv"))
    for f_ in (f"{tag}_train_step_kernels.txt", f"{tag}_trainvox_step_kernels.txt", f"{tag}_bench_line.json", f"{tag}_vox_line.json",
               f"{tag}_pmc_train.json"):
        if os.path.exists(os.path.join(ROOT, "gpurun_out", f_)):
            shutil.copy(os.path.join(ROOT, "gpurun_out", f_), os.path.join(out, f_))
    fe, wr, mf = counters(tag, "pmc_fetch"), counters(tag, "pmc_write"), counters(tag, "pmc_mfma")
    summ = {"note": f"per-launch averages over every conv launch of `bench.py --no-knn --graph 0 --streams 1 --prec {prec}` (b=64); "
                    "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction)", "kernels": {}}
    tf = tw = n = 0
    for k in fe:
        f, w = fe[k]["FETCH_SIZE"], wr[k]["WRITE_SIZE"]
        m = mf[k]
        busy = sum(m["SQ_VALU_MFMA_BUSY_CYCLES"]) / (sum(m["GRBM_GUI_ACTIVE"]) / 8 * 1024)
        summ["kernels"][k] = {
            "launches_profiled": len(f), "fetch_size_kib_avg": sum(f) / len(f), "write_size_kib_avg": sum(w) / len(w),
            "hbm_mb_per_launch": (2 * sum(f) / len(f) + sum(w) / len(w)) * 1024 / 1e6,
            "mfma_busy_frac": busy}
        try:
            wt = counters(tag, "pmc_wait")[k]
            wc = sum(wt["SQ_WAVE_CYCLES"])
            summ["kernels"][k]["wave_cycles_split"] = {
                "note": "fractions of SQ_WAVE_CYCLES (quad-cycles summed over waves): parked on s_waitcnt / s_barrier, issue stalls "
                        "(of which LDS issue stalls), issuing",
                "wait_any": sum(wt["SQ_WAIT_ANY"]) / wc, "wait_inst_any": sum(wt["SQ_WAIT_INST_ANY"]) / wc,
                "wait_inst_lds": sum(wt["SQ_WAIT_INST_LDS"]) / wc, "active_inst_any": sum(wt["SQ_ACTIVE_INST_ANY"]) / wc,
                "lds_insts_per_mfma": sum(wt["SQ_INSTS_LDS"]) / max(sum(wt["SQ_INSTS_MFMA"]), 1)}
        except Exception:
            pass
        if "kNN" not in k or True:
            tf, tw, n = tf + sum(f), tw + sum(w), n + len(f)
    summ["conv_hbm_bytes_per_launch"] = (2 * tf + tw) * 1024 / n
    # the 3x3 stride-1 family of bench.py's `roofline` (fused blocks + the other 3x3 convs): bytes per launch over its launches
    fam = [k for k in fe if "fblock64" in k or "igemm_kxr" in k]
    ff = sum(sum(fe[k]["FETCH_SIZE"]) for k in fam)
    fw = sum(sum(wr[k]["WRITE_SIZE"]) for k in fam)
    fn = sum(len(fe[k]["FETCH_SIZE"]) for k in fam)
    if fn:
        summ["conv3x3_family_hbm_bytes_per_launch"] = (2 * ff + fw) * 1024 / fn
    sys.path.insert(0, ROOT)
    import bench_inputs
    summ["csrc_sha16"] = bench_inputs.kernel_source_sha16(ROOT)      # bench.py quotes these figures only while this matches
    json.dump(summ, open(os.path.join(out, f"{tag}_pmc_conv_p{prec}.json"), "w"), indent=1)
    print(json.dumps(summ, indent=1))


if __name__ == "__main__":
    main()
