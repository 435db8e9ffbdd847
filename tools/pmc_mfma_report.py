"""Per kernel instantiation, from two rocprofv3 --pmc passes (tools/pmc_mfma.sh, tools/pmc_headline.sh):
  matrix-pipe busy  = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs) / (GRBM_GUI_ACTIVE / 8 XCDs)
                      (MI355X_MICROARCH.md: the counter is in cycles, 32 per v_mfma_*_32x32x16 per SIMD; GRBM_GUI_ACTIVE sums the 8 XCDs)
  effective clock   = (GRBM_GUI_ACTIVE / 8) / kernel duration of the SAME (profiled) pass
  L2 hit rate       = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum); fabric reads = TCC_EA0_RDREQ_sum x 64 B (x 2 for wide streaming
                      reads on gfx950, the guide's FETCH_SIZE correction), writes = TCC_EA0_WRREQ_sum x 64 B
usage: pmc_mfma_report.py <dir of the SQ/GRBM pass> <dir of the TCC pass> <output stem> [kernel-name filter]"""
import collections
import csv
import glob
import json
import os
import sys

FAMILIES = (   # bench.py's bracket families -> the kernel instantiations behind them (rocprof names)
    ("conv_h2d_fwd", "conv_h2d_kernel<4, false, false"), ("conv_h2d_dgrad", "conv_h2d_kernel<4, false, true"),
    ("conv_wgrad_h2d", "igemm_wgrad_h2d_kernel<4"), ("conv_h2p_fwd", "conv_h2p_kernel<false>"), ("conv_h2p_dgrad", "conv_h2p_kernel<true>"),
    ("conv_x3_128x256", "igemm_conv_x3_kernel<2, 4, 2, 2, false, false"), ("conv_dgrad_wide", "igemm_conv_x3_kernel<2, 4, 2, 2, false, true"),
    ("conv_wgrad_h2t4", "igemm_wgrad_h2t_kernel<4"),
    ("conv_bf16_fwd_ep0", "conv_bf16_kernel<2, 4, 2, 2, false, false, true, true, 3, 1, 0>"),
    ("conv_bf16_fwd_ep1", "conv_bf16_kernel<2, 4, 2, 2, false, false, true, true, 3, 1, 1>"),
    ("conv_bf16_dgrad_ep0", "conv_bf16_kernel<2, 4, 2, 2, false, true, true, true, 3, 1, 0>"),
    ("conv_bf16_dgrad_ep2", "conv_bf16_kernel<2, 4, 2, 2, false, true, true, true, 3, 1, 2>"),
    ("conv_bf16_dgrad_ep3", "conv_bf16_kernel<2, 4, 2, 2, false, true, true, true, 3, 1, 3>"),
    ("conv_bf16_wgrad4", "wgrad_bf16_dma_kernel<4"))


def short(name):
    n = name.replace("void (anonymous namespace)::", "")
    cut = n.find("((anonymous")
    if cut < 0:
        cut = n.find("(")
    return (n[:cut] if cut > 0 else n)[:110]


def load(d):
    """-> {kernel: {counter: [values per dispatch]}}, {kernel: [durations ns]}"""
    ctr = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            ctr[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r and "End_Timestamp" in r and r.get("Dispatch_Id") not in seen:
                seen.add(r.get("Dispatch_Id"))
                dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    if not dur:
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                dur[short(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return ctr, dur


def main():
    d1, d2, stem = sys.argv[1:4]
    filt = sys.argv[4] if len(sys.argv) > 4 else ""
    c1, t1 = load(d1)
    c2, _ = load(d2)
    rows = []
    for k, c in c1.items():
        if filt and filt not in k:
            continue
        n = len(c.get("GRBM_GUI_ACTIVE", []))
        if not n or k not in t1:
            continue
        mean = lambda v: sum(v) / max(len(v), 1)
        grbm, busy, nm = mean(c["GRBM_GUI_ACTIVE"]) / 8.0, mean(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])), mean(c.get("SQ_INSTS_MFMA", [0]))
        us = mean(t1[k]) / 1e3
        wc = mean(c.get("SQ_WAVE_CYCLES", [0]))
        row = {"kernel": k, "launches": n, "avg_us": round(us, 2), "total_ms": round(us * n / 1e3, 3),
               "mfma_busy": round(busy / 1024.0 / grbm, 4) if grbm else None, "clock_ghz": round(grbm / (us * 1e3), 3) if us else None,
               "mfma_insts": round(nm), "wait_inst_frac": round(mean(c.get("SQ_WAIT_INST_ANY", [0])) / wc, 3) if wc else None,
               "active_inst_frac": round(mean(c.get("SQ_ACTIVE_INST_ANY", [0])) / wc, 3) if wc else None}
        t = c2.get(k)
        if t and t.get("TCC_HIT_sum"):
            hit, miss = mean(t["TCC_HIT_sum"]), mean(t["TCC_MISS_sum"])
            row.update({"l2_hit": round(hit / max(hit + miss, 1.0), 4), "l2_req": round(hit + miss),
                        "ea_rd_MB": round(mean(t.get("TCC_EA0_RDREQ_sum", [0])) * 64 / 1e6, 2),
                        "ea_wr_MB": round(mean(t.get("TCC_EA0_WRREQ_sum", [0])) * 64 / 1e6, 2)})
        rows.append(row)
    rows.sort(key=lambda r: -r["total_ms"])
    fam = {}
    for name, pat in FAMILIES:
        sel = [r for r in rows if pat in r["kernel"]]
        if sel:
            tot = sum(r["total_ms"] for r in sel)
            w = lambda key: round(sum((r.get(key) or 0) * r["total_ms"] for r in sel) / tot, 4)
            fam[name] = {"kernel": " + ".join(r["kernel"] for r in sel), "launches": sum(r["launches"] for r in sel), "total_ms": round(tot, 3),
                         "mfma_busy": w("mfma_busy"), "clock_ghz": w("clock_ghz"), "l2_hit": w("l2_hit")}
    with open(stem + ".txt", "w") as o:
        o.write("# " + __doc__.strip().replace("\n", "\n# ") + "\n")
        o.write("kernel | launches | avg us (profiled pass) | total ms | MFMA busy | clock GHz | wait-inst / active-inst of wave cycles | L2 hit | fabric rd MB (x64 B) | wr MB\n")
        for r in rows[:70]:
            o.write(f"{r['kernel']} | {r['launches']} | {r['avg_us']} | {r['total_ms']} | {r['mfma_busy']} | {r['clock_ghz']} | "
                    f"{r['wait_inst_frac']} / {r['active_inst_frac']} | {r.get('l2_hit')} | {r.get('ea_rd_MB')} | {r.get('ea_wr_MB')}\n")
        o.write("\n# by bench.py family (time-weighted over the family's instantiations)\n")
        for k, v in fam.items():
            o.write(f"{k}: launches {v['launches']} total {v['total_ms']} ms MFMA busy {v['mfma_busy']} clock {v['clock_ghz']} GHz L2 hit {v['l2_hit']}\n")
    json.dump({"by_family": fam, "kernels": rows[:70]}, open(stem + ".json", "w"), indent=1)
    print(open(stem + ".txt").read()[:6000])


if __name__ == "__main__":
    main()
