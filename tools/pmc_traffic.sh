#!/bin/bash
# HBM traffic of the dominant conv kernel: separate FETCH_SIZE / WRITE_SIZE passes over bench.py (MI355X_MICROARCH.md §HBM)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do   # RCF_XCD_RANGES etc. pass through the environment
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-stage2 > $R/gpurun_out/pmc_$c.log 2>&1
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.Counter())
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$R/gpurun_out/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c: continue
            agg[r["Kernel_Name"]][c] += float(r["Counter_Value"]); cnt[r["Kernel_Name"]][c] += 1
rows = []
for k, d in agg.items():
    n = max(cnt[k].values())
    f, w = d.get("FETCH_SIZE", 0) / max(cnt[k]["FETCH_SIZE"], 1), d.get("WRITE_SIZE", 0) / max(cnt[k]["WRITE_SIZE"], 1)
    rows.append(((2 * f + w) * 1024 * n, k, n, f, w))
rows.sort(reverse=True)
with open("$R/gpurun_out/pmc_hbm_traffic.txt", "w") as o:
    o.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-stage2\n")
    o.write("# counter unit KiB; gfx950 correction: hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md HBM section)\n")
    o.write("kernel | launches | FETCH_SIZE avg KiB | WRITE_SIZE avg KiB | corrected HBM MB per launch\n")
    for tot, k, n, f, w in rows[:40]:
        o.write(f"{k[:120]} | {n} | {f:.1f} | {w:.1f} | {(2*f+w)*1024/1e6:.1f}\n")
out = {}
# a family = every kernel whose name starts like the pattern (template variants of one kernel: 128 x 256 and 256 x 256 tiles)
for fam, pat in (("conv_h2d_dgrad", "conv_h2d_kernel<4, false, true,"), ("conv_h2d_fwd", "conv_h2d_kernel<4, false, false,"),
                 ("conv_wgrad_h2d", "igemm_wgrad_h2d_kernel<4, false"), ("conv_wgrad_h2t4", "igemm_wgrad_h2t_kernel<4, false, true"), ("conv_x3_128x256", "igemm_conv_x3_kernel<2, 4, 2, 2, false, false, 2, true, true,"),
                 ("conv_dgrad_wide", "igemm_conv_x3_kernel<2, 4, 2, 2, false, true, 2, true, true,"),
                 ("conv_h2p_fwd", "conv_h2p_kernel<false>"), ("conv_h2p_dgrad", "conv_h2p_kernel<true>"),
                 ("conv_bf16_wgrad4", "wgrad_bf16_dma_kernel<4, false, true"), ("conv_bf16_fwd", "conv_bf16_kernel<2, 4, 2, 2, false, false, true, true, 3, 1, 0>"),
                 ("conv_bf16_fwd_folded", "conv_bf16_kernel<2, 4, 2, 2, false, false, true, true, 3, 1, 1>"),
                 ("conv_bf16_dgrad_wide", "conv_bf16_kernel<2, 4, 2, 2, false, true, true, true, 3, 1, 0>"),
                 ("conv_bf16_dgrad_masked", "conv_bf16_kernel<2, 4, 2, 2, false, true, true, true, 3, 1, 2>")):
    sel = [(k, n, f, w) for tot, k, n, f, w in rows if pat in k]
    if sel:
        nn = sum(n for _, n, _, _ in sel)
        bytes_ = sum((2 * f + w) * 1024 * n for _, n, f, w in sel)
        out[fam] = {"kernel": " + ".join(k for k, _, _, _ in sel), "launches": nn, "hbm_bytes_per_launch": bytes_ / nn}
        print(fam, nn, round(bytes_ / nn / 1e6, 1), "MB/launch")
# bench.py reads the entry of the family its headline roofline brackets (fp32 leg: the 256-wide plane forward kernel)
head = "conv_h2d_fwd" if "conv_h2d_fwd" in out else "conv_wgrad_h2t4"
if head in out:
    d = dict(out[head], family=head, by_family=out,
             note="(2*FETCH_SIZE + WRITE_SIZE)*1024 per launch, separate rocprofv3 --pmc passes over bench.py --steps 1 --warmup 1; "
                  "the split-K partial sums this kernel writes and the reduction kernel reads are part of its traffic")
    json.dump(d, open("$R/gpurun_out/pmc_traffic.json", "w"), indent=1)
PY
