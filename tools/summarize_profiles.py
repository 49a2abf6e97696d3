"""Condense rocprofv3 outputs under gpurun_out/ into the small tracked summaries under profiles/.
usage: python tools/summarize_profiles.py r01"""
import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
G = os.path.join(ROOT, "gpurun_out")
sys.path.insert(0, ROOT)
from pase_amd import build as _B  # noqa: E402
# the digest of the kernel sources these profiles were taken of: bench.py reports `roofline.traffic` from this file
# only while it matches the library that is loaded
out = {"tag": tag, "lib_digest": _B.hip_digest()}


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:90]


def last_step(rows):
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    if len(idx) < 14:
        return rows
    return rows[idx[-14] + 1: idx[-1] + 1]


def counters(path):
    agg = defaultdict(lambda: defaultdict(float))
    f = os.path.join(G, path + "_" + tag, "r_counter_collection.csv")
    if not os.path.exists(f):
        f = os.path.join(G, path, "r_counter_collection.csv")
    if not os.path.exists(f):
        return None
    rows = list(csv.DictReader(open(f)))
    # restrict to the last training step: dispatch ids after the 14th-from-last adam launch
    adam = sorted({int(r["Dispatch_Id"]) for r in rows if "adam_kernel" in r["Kernel_Name"]})
    lo = adam[-14] if len(adam) >= 14 else -1
    for r in rows:
        if int(r["Dispatch_Id"]) > lo:
            agg[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
            agg[short(r["Kernel_Name"])]["_dispatches_x_counters"] += 1
    return agg


# --- kernel time per family for one step
kt = None
for cand in ("prof_" + tag, "prof4", "prof3", "prof2"):
    f = os.path.join(G, cand, "r1_kernel_trace.csv")
    if os.path.exists(f):
        kt = f
        break
if kt:
    step = last_step(list(csv.DictReader(open(kt))))
    fam = defaultdict(lambda: [0, 0.0])
    for r in step:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        fam[short(r["Kernel_Name"])][0] += 1
        fam[short(r["Kernel_Name"])][1] += d
    tot = sum(v[1] for v in fam.values())
    span = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3
    out["step_kernel_time"] = {"source": os.path.relpath(kt, ROOT), "step_span_us": round(span, 1),
                               "sum_kernel_us": round(tot, 1),
                               "families": [{"kernel": k, "calls": v[0], "total_us": round(v[1], 1),
                                             "avg_us": round(v[1] / v[0], 1), "pct": round(100 * v[1] / tot, 2)}
                                            for k, v in sorted(fam.items(), key=lambda x: -x[1][1])[:28]]}
# --- HBM traffic (FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x,
#     MI355X_MICROARCH.md section HBM -> both the raw and the corrected figure are given)
fe, wr = counters("pmc_fetch"), counters("pmc_write")
if fe and wr:
    tr = []
    for k in fe:
        f_kib = fe[k].get("FETCH_SIZE", 0.0)
        w_kib = wr.get(k, {}).get("WRITE_SIZE", 0.0)
        tr.append({"kernel": k, "fetch_MB_raw": round(f_kib * 1024 / 1e6, 1), "fetch_MB_x2": round(2 * f_kib * 1024 / 1e6, 1),
                   "write_MB": round(w_kib * 1024 / 1e6, 1)})
    tr.sort(key=lambda x: -(x["fetch_MB_x2"] + x["write_MB"]))
    out["hbm_traffic_per_step"] = tr[:24]
sq = counters("pmc_sq")
if sq:
    rows = []
    for k, c in sq.items():
        if c.get("SQ_INSTS_MFMA", 0) > 0:
            rows.append({"kernel": k, "mfma_insts": c["SQ_INSTS_MFMA"], "valu_insts": c["SQ_INSTS_VALU"],
                         "valu_per_mfma": round(c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"], 2),
                         "mfma_busy_cycles": c["SQ_VALU_MFMA_BUSY_CYCLES"], "sq_busy_cycles": c["SQ_BUSY_CYCLES"],
                         # SQ_BUSY_CYCLES is summed over 32 shader engines, MFMA busy over 1024 SIMDs
                         "mfma_util": round((c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (c["SQ_BUSY_CYCLES"] / 32), 4),
                         "lds_bank_conflict_cycles": c.get("SQ_LDS_BANK_CONFLICT", 0.0),
                         "wait_inst_any_frac": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3)})
    out["sq_counters_per_step"] = rows
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
dst = os.path.join(ROOT, "profiles", "summary_%s.json" % tag)
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
