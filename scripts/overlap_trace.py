#!/usr/bin/env python3
"""Co-residency of the two HIP streams of the default (streaming) bench step, from a rocprofv3 --kernel-trace of it
(gpurun_out/r04_overlap, written by scripts/round4_collect.sh): per stream the time its kernels occupy, the time BOTH streams have a
kernel in flight, and which kernels are co-resident with which.   python scripts/overlap_trace.py [tag]  -> profiles/<tag>_overlap_trace.md"""
import collections, csv, glob, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
files = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_overlap", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(files[-1])))


def short(n):
    n = n.replace("void ", "").replace("fh::", "").replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", n)[:44]


# the streaming step runs on two streams of its own; the default stream (Stream_Id 0) carries set-up and bench.py's serial reference pass
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], short(r["Kernel_Name"])) for r in rows
      if "fh::" in r["Kernel_Name"] and r["Stream_Id"] != "0"]
ev.sort()
byq = collections.defaultdict(list)
for s, e, q, k in ev:
    byq[q].append((s, e, k))
queues = sorted(byq, key=lambda q: -len(byq[q]))[:2]
rec_q = max(queues, key=lambda q: sum(1 for _, _, k in byq[q] if k.startswith("wino_gemm")))   # IResNet's kernels: the recogniser's stream
det_q = [q for q in queues if q != rec_q][0]
# steady state: from 30 % into the two streams' common span (warm-up steps) to its end
a0, a1 = max(byq[det_q][0][0], byq[rec_q][0][0]), min(byq[det_q][-1][1], byq[rec_q][-1][1])
t0 = a0 + int(0.3 * (a1 - a0)); t1 = a1


def merged(iv):
    out = []
    for s, e in sorted(iv):
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def clip(iv):
    return [(max(s, t0), min(e, t1)) for s, e, *_ in iv if e > t0 and s < t1]


md, mr = merged(clip(byq[det_q])), merged(clip(byq[rec_q]))
busy_d, busy_r = sum(e - s for s, e in md), sum(e - s for s, e in mr)
both = 0
i = j = 0
while i < len(md) and j < len(mr):
    s, e = max(md[i][0], mr[j][0]), min(md[i][1], mr[j][1])
    if e > s:
        both += e - s
    if md[i][1] < mr[j][1]:
        i += 1
    else:
        j += 1
span = t1 - t0
# per detector kernel: how much of its time a recogniser kernel was in flight too
co = collections.defaultdict(lambda: [0, 0])
for s, e, k in byq[det_q]:
    if e <= t0:
        continue
    ov = sum(max(0, min(e, b) - max(s, a)) for a, b in mr if b > s and a < e)
    co[k][0] += e - s; co[k][1] += ov
bench = ""
bj = os.path.join(ROOT, "gpurun_out", f"{tag}_overlap_bench.json")
if os.path.exists(bj):
    bench = open(bj).read().strip()
out = [f"# Two-stream co-residency of the default bench step ({tag}) — rocprofv3 --kernel-trace of `python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing`", "",
       f"Steady-state window: the last 70 % of the span in which both streams are active ({span / 1e6:.2f} ms).  A stream is 'busy' while one of its kernels is between its start and end timestamps.", "",
       "| | ms | share of the window |", "|---|---|---|",
       f"| detector stream busy (queue {det_q}) | {busy_d / 1e6:.3f} | {busy_d / span:.3f} |",
       f"| recogniser stream busy (queue {rec_q}) | {busy_r / 1e6:.3f} | {busy_r / span:.3f} |",
       f"| BOTH busy (kernels of the two streams co-resident) | {both / 1e6:.3f} | {both / span:.3f} |",
       f"| neither busy | {(span - busy_d - busy_r + both) / 1e6:.3f} | {(span - busy_d - busy_r + both) / span:.3f} |", "",
       f"Fraction of the detector's kernel time spent beside a recogniser kernel: {both / max(busy_d, 1):.3f}.  Sum of the two streams' busy times ÷ window = "
       f"{(busy_d + busy_r) / span:.3f}: kernels that run side by side each take LONGER than alone (they share CUs, LDS and HBM), so a value above 1 is the "
       "stretched time, not throughput.", "",
       "Detector kernels, their time in the window and the part of it overlapped by a recogniser kernel:", "",
       "| kernel | ms | overlapped |", "|---|---|---|"]
for k, (t, ov) in sorted(co.items(), key=lambda kv: -kv[1][0])[:14]:
    out.append(f"| {k} | {t / 1e6:.3f} | {ov / max(t, 1):.2f} |")
if bench:
    out += ["", "bench.py line of the traced run:", "", "```json", bench, "```"]
open(os.path.join(ROOT, "profiles", f"{tag}_overlap_trace.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out[:24]))
