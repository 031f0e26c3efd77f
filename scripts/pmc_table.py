#!/usr/bin/env python3
"""Per-kernel counter table from the separate rocprofv3 --pmc passes of scripts/pmc_kernels.sh (gpurun_out/pmck_<tag>_*/):
   python scripts/pmc_table.py <tag> <out.md> [title]
Means per launch, plus the derived figures the roofline discussion uses: matrix-pipe busy share of the launch
(SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs)), wave time parked / issue-stalled / issuing, instructions
per MFMA, LDS bank-conflict share, L2 hit rate.  (GRBM_GUI_ACTIVE / 8 over-reads the duration of launches shorter than ~0.3 ms —
MI355X_MICROARCH.md, DVFS — so the busy share of short kernels is a lower bound.)"""
import collections, csv, glob, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, out_path = sys.argv[1], sys.argv[2]
title = sys.argv[3] if len(sys.argv) > 3 else tag
acc = collections.defaultdict(lambda: collections.defaultdict(list))


def short(n):
    n = n.replace("void ", "").replace("fh::", "").replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", n)[:64]


for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"pmck_{tag}_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
m = lambda d, c: (sum(d[c]) / len(d[c])) if d.get(c) else None
lines = [f"# {title}", "",
         "Separate `rocprofv3 --pmc <group> --kernel-trace` passes (scripts/pmc_kernels.sh: the program directly behind `--`), means per launch.",
         "`mfma busy` = SQ_VALU_MFMA_BUSY_CYCLES ÷ (GRBM_GUI_ACTIVE ÷ 8 × 1024 SIMDs); `parked / stalled / issuing` = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / "
         "SQ_ACTIVE_INST_ANY ÷ SQ_WAVE_CYCLES; instruction counts per MFMA instruction (per 1 000 VALU where a kernel has no MFMA).", "",
         "| kernel | launches | GRBM cycles / 8 | mfma busy | parked | stalled | issuing | VALU / MFMA | LDS / MFMA | SALU / MFMA | VMEM / MFMA | LDS conflict share | L2 hit | FETCH MB (x2) | WRITE MB |",
         "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
rows = []
for k, d in acc.items():
    g = m(d, "GRBM_GUI_ACTIVE")
    if not g:
        continue
    cyc = g / 8
    n = max(len(v) for v in d.values())
    wc = m(d, "SQ_WAVE_CYCLES") or 0
    mf = m(d, "SQ_INSTS_MFMA") or 0
    den = mf if mf > 0 else (m(d, "SQ_INSTS_VALU") or 1) / 1000.0
    f3 = lambda x: "-" if x is None else f"{x:.3f}"
    f2 = lambda x: "-" if x is None else f"{x:.2f}"
    busy = (m(d, "SQ_VALU_MFMA_BUSY_CYCLES") or 0) / (cyc * 1024)
    hit, miss = m(d, "TCC_HIT_sum"), m(d, "TCC_MISS_sum")
    rows.append((cyc * n, f"| {k} | {n} | {cyc:,.0f} | {f3(busy)} | {f3((m(d, 'SQ_WAIT_ANY') or 0) / wc if wc else None)} | "
                 f"{f3((m(d, 'SQ_WAIT_INST_ANY') or 0) / wc if wc else None)} | {f3((m(d, 'SQ_ACTIVE_INST_ANY') or 0) / wc if wc else None)} | "
                 f"{f2((m(d, 'SQ_INSTS_VALU') or 0) / den)} | {f2((m(d, 'SQ_INSTS_LDS') or 0) / den)} | {f2((m(d, 'SQ_INSTS_SALU') or 0) / den)} | "
                 f"{f2((m(d, 'SQ_INSTS_VMEM') or 0) / den)} | {f3((m(d, 'SQ_LDS_BANK_CONFLICT') or 0) / (m(d, 'SQ_LDS_IDX_ACTIVE') or 1))} | "
                 f"{f3(hit / (hit + miss) if hit is not None and miss is not None and hit + miss > 0 else None)} | "
                 f"{f2(2 * (m(d, 'FETCH_SIZE') or 0) * 1024 / 1e6)} | {f2((m(d, 'WRITE_SIZE') or 0) * 1024 / 1e6)} |"))
for _, l in sorted(rows, reverse=True):
    lines.append(l)
open(os.path.join(ROOT, out_path), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:14]))
