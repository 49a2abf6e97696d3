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


def nth_step(rows, j):
    """kernels of training step j (0-based): between the 13 Adam launches that end step j - 1 and those that end step j"""
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    if len(idx) < 13 * (j + 1):
        return None
    lo = idx[13 * j - 1] + 1 if j > 0 else 0
    return rows[lo: idx[13 * (j + 1) - 1] + 1]


def union_us(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s_, e_ in iv:
        if cur_e is None or s_ > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s_, e_
        else:
            cur_e = max(cur_e, e_)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot / 1e3


# (template arguments <NPOS, KGS, TM, ZP, NARROW>: TM = true marks a weight gradient; first match wins)
FAMILY = (("split-bf16 weight gradients", ("conv_x6c_kernel<128, 3, true", "conv_x6c_kernel<128, 4, true", "conv_x6c_kernel<128, 5, true",
                                           "x6c_wgrad_sym_kernel", "sinc_x6_wgrad")),
          ("split-bf16 convolutions / data gradients", ("conv_x6c_kernel<192", "conv_x6c_kernel<128, 3, false", "conv_x6c_kernel<320",
                                                         "sinc_x6_fwd")),
          ("exact-fp32 GEMMs", ("conv_gemm_kernel", "wgrad_gemm_kernel", "wgrad_flat_kernel")),
          ("operand packs", ("pack_",)),
          ("elementwise / normalisation / scan / heads", ("act_bwd", "bn_", "qrnn_", "head1_", "adam", "commit_cols", "rownorm",
                                                          "ctx_loss", "sinc_filters")),
          ("stock torch kernels", ("at::native", "__amd_rocclr")))


def family_of(name):
    for fam, keys in FAMILY:
        if any(k in name for k in keys):
            return fam
    return "other"


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
    # the OVERLAPPED step the headline is measured on (bench.py: W warm-up + K timed steps run with the weight-gradient /
    # head / pooling side streams on; the two steps behind them are the serialised per-launch timing steps summarised above):
    # step W + K - 1.  Per stream: kernels and busy time (union of its kernels' intervals); `gpu_busy_us` = union over all
    # streams; sum_kernel_us > step_span_us is what the overlap buys.
    nsteps = len([r for r in csv.DictReader(open(kt)) if "adam_kernel" in r["Kernel_Name"]]) // 13
    ov = nth_step(list(csv.DictReader(open(kt))), nsteps - 3) if nsteps >= 4 else None
    if ov:
        iv_all, per = [], defaultdict(list)
        fam2 = defaultdict(lambda: [0, 0.0])
        for r in ov:
            s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            iv_all.append((s_, e_))
            per[(r.get("Queue_Id", "?"), r.get("Stream_Id", "?"))].append((s_, e_))
            fam2[short(r["Kernel_Name"])][0] += 1
            fam2[short(r["Kernel_Name"])][1] += (e_ - s_) / 1e3
        span2 = (max(e for _, e in iv_all) - min(s_ for s_, _ in iv_all)) / 1e3
        out["overlapped_step"] = {
            "which": "training step %d of %d in the trace (last timed step of bench.py)" % (nsteps - 3, nsteps),
            "step_span_us": round(span2, 1), "sum_kernel_us": round(sum(v[1] for v in fam2.values()), 1),
            "gpu_busy_us": round(union_us(iv_all), 1), "launches": len(ov),
            "streams": [{"queue/stream": "%s/%s" % k, "kernels": len(v), "busy_us": round(union_us(v), 1),
                         "sum_kernel_us": round(sum(e - s_ for s_, e in v) / 1e3, 1)}
                        for k, v in sorted(per.items(), key=lambda kv: -len(kv[1]))],
            "families": [{"kernel": k, "calls": v[0], "total_us": round(v[1], 1), "avg_us": round(v[1] / v[0], 1)}
                         for k, v in sorted(fam2.items(), key=lambda x: -x[1][1])[:14]]}
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
    byfam = defaultdict(lambda: [0.0, 0.0])
    for row in tr:
        byfam[family_of(row["kernel"])][0] += row["fetch_MB_x2"]
        byfam[family_of(row["kernel"])][1] += row["write_MB"]
    out["hbm_traffic_by_family_GB"] = {k: {"fetch_x2": round(v[0] / 1e3, 2), "write": round(v[1] / 1e3, 2),
                                           "total": round((v[0] + v[1]) / 1e3, 2)} for k, v in byfam.items()}
    out["hbm_traffic_total_GB"] = round(sum(v[0] + v[1] for v in byfam.values()) / 1e3, 2)
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
