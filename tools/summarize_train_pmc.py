#!/usr/bin/env python3
"""gpurun_out/<tag>_trainpmc_* (tools/collect_train_pmc.sh) -> one JSON with a per-kernel entry for the training step:
launches per step, average duration (kernel trace), HBM bytes per launch ((2 * FETCH_SIZE + WRITE_SIZE) * 1024: the gfx950
FETCH_SIZE correction of MI355X_MICROARCH.md), MFMA-busy fraction, the wave-cycle split and LDS bank-conflict share.
Steps are delimited by the fused-Adam launches; only the LAST step of every pass is summarised (steady state).

    python3 tools/summarize_train_pmc.py r05 > profiles/r05_pmc_train.json
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    return name.replace("void ", "").split("(")[0][:80]


def last_step(rows, key):
    """rows of the last step: between the last two groups of multi_tensor_apply launches"""
    rows.sort(key=key)
    ad = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r["Kernel_Name"]]
    groups = []
    for i in ad:
        if not groups or i - groups[-1][-1] > 50:
            groups.append([i])
        else:
            groups[-1].append(i)
    return rows[groups[-2][-1] + 1: groups[-1][-1] + 1]


def load(tag, sub, pat):
    d = os.path.join(ROOT, "gpurun_out", f"{tag}_trainpmc_{sub}")
    fs = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return list(csv.DictReader(open(max(fs, key=os.path.getmtime))))


def counters(tag, sub):
    rows = load(tag, sub, "*counter_collection.csv")
    # one row per (dispatch, counter): order dispatches by id
    disp = collections.OrderedDict()
    for r in rows:
        disp.setdefault(int(r["Dispatch_Id"]), {"Kernel_Name": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
    lst = [dict(v, Dispatch_Id=k) for k, v in disp.items()]
    step = last_step(lst, key=lambda r: r["Dispatch_Id"])
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in step:
        for c, v in r.items():
            if c not in ("Kernel_Name", "Dispatch_Id"):
                agg[short(r["Kernel_Name"])][c].append(v)
    return agg


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    tr = last_step(load(tag, "trace", "*kernel_trace.csv"), key=lambda r: int(r["Start_Timestamp"]))
    dur = collections.defaultdict(list)
    for r in tr:
        dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = sum(sum(v) for v in dur.values())
    fe, wr, mf, wt = (counters(tag, s) for s in ("fetch", "write", "mfma", "wait"))
    try:
        ld = counters(tag, "lds")
    except Exception:
        ld = {}
    out = {"note": "last (steady-state) step of `tools/train_bench.py --batch 16 --streams 1` (BF16X3 maps; 16 queries x (1 panorama "
                   "+ 11 tiles of 256^2)); durations from the kernel trace, counters from separate --pmc passes; hbm bytes = "
                   "(2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / "
                   "(GRBM_GUI_ACTIVE / 8 XCD * 1024 SIMD); wave-cycle fractions are of SQ_WAVE_CYCLES",
           "step_kernel_ms": total / 1e6, "kernels": {}}
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        if sum(v) < 0.004 * total:
            continue
        e = {"launches_per_step": len(v), "us_per_launch": sum(v) / len(v) / 1e3, "ms_per_step": sum(v) / 1e6,
             "share_of_step": sum(v) / total}
        if k in fe and k in wr:
            f, w = fe[k]["FETCH_SIZE"], wr[k]["WRITE_SIZE"]
            e["hbm_mb_per_launch"] = (2 * sum(f) / len(f) + sum(w) / len(w)) * 1024 / 1e6
            e["hbm_gb_per_s"] = e["hbm_mb_per_launch"] / e["us_per_launch"] * 1e3 / 1e3
        if k in mf and sum(mf[k]["GRBM_GUI_ACTIVE"]) > 0:
            m = mf[k]
            e["mfma_busy_frac"] = sum(m["SQ_VALU_MFMA_BUSY_CYCLES"]) / (sum(m["GRBM_GUI_ACTIVE"]) / 8 * 1024)
            e["mfma_insts_per_launch"] = sum(m["SQ_INSTS_MFMA"]) / len(m["SQ_INSTS_MFMA"])
        if k in wt and sum(wt[k]["SQ_WAVE_CYCLES"]) > 0:
            t = wt[k]
            wc = sum(t["SQ_WAVE_CYCLES"])
            e["wave_cycles_split"] = {"wait_any": sum(t["SQ_WAIT_ANY"]) / wc, "wait_inst_any": sum(t["SQ_WAIT_INST_ANY"]) / wc,
                                      "wait_inst_lds": sum(t["SQ_WAIT_INST_LDS"]) / wc,
                                      "active_inst_any": sum(t["SQ_ACTIVE_INST_ANY"]) / wc,
                                      "lds_insts_per_mfma": sum(t["SQ_INSTS_LDS"]) / max(sum(t["SQ_INSTS_MFMA"]), 1)}
        if k in ld and sum(ld[k].get("SQ_LDS_IDX_ACTIVE", [0])) > 0:
            t = ld[k]
            e["lds_bank_conflict_frac"] = sum(t["SQ_LDS_BANK_CONFLICT"]) / sum(t["SQ_LDS_IDX_ACTIVE"])
            e["valu_insts_per_launch"] = sum(t["SQ_INSTS_VALU"]) / len(t["SQ_INSTS_VALU"])
        out["kernels"][k] = e
    sys.path.insert(0, ROOT)
    import bench_inputs
    out["csrc_sha16"] = bench_inputs.kernel_source_sha16(ROOT)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
