#!/usr/bin/env python3
"""Turns the scratch rocprofv3 output of scripts/profile_round.sh (gpurun_out/prof_<tag>/) into the
tracked, judged artefacts under profiles/:

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (verbatim)
  profiles/<tag>_traffic.csv        per kernel: launches, FETCH_SIZE / WRITE_SIZE sums and per launch
  profiles/<tag>_summary.md         both, readable, with the bench line measured under rocprofv3
  profiles/traffic.json             {workload: {kernel: bytes per launch}} read back by bench.py

Units (MI355X_MICROARCH.md §HBM): FETCH_SIZE / WRITE_SIZE are in KiB and count the L2's
fabric-side requests (Infinity-Cache hits included).  On gfx950 FETCH_SIZE reports exactly half
the bytes of a wide (16 B/lane) streaming read, so the read side is doubled here, as the guide
prescribes; WRITE_SIZE is exact for 16-byte stores.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
workload = sys.argv[2] if len(sys.argv) > 2 else "e2e"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"(?:void )?(?:fh::)?(\w+)(<[^>]*>)?", name)
    if not m:
        return name[:60]
    base, targs = m.group(1), m.group(2) or ""
    if base == "conv_igemm_kernel":
        a = [x.strip() for x in targs.strip("<>").split(",")]
        grouped = len(a) > 6 and a[6] == "true"              # the 36-GEMM launch of the Winograd form
        return f"conv_igemm_kernel<{a[0]},{a[1]},{a[2]},{a[3]}>" + (" grouped (Winograd GEMM)" if grouped else "")
    if base == "conv_fixup_kernel":
        return "conv_fixup_kernel"
    if base == "dwpw_kernel":                                 # <BN, WM, WN, depthwise stride, float4 columns, direct, fused stem>
        a = [x.strip() for x in targs.strip("<>").split(",")]
        return f"dwpw_kernel<{','.join(a)}>".replace("false", "0").replace("true", "1")
    return base + (targs if len(targs) < 24 else "")


def newest(pattern):
    """gpurun merges every call's output into the same scratch tree: take the latest run, not the first the glob returns."""
    return sorted(glob.glob(pattern), key=os.path.getmtime)[-1:]


stats = newest(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(stats)))
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as f:
    f.write(open(stats).read())

agg = collections.defaultdict(lambda: {"launches": 0, "fetch_kib": 0.0, "write_kib": 0.0})
for kind, key in (("fetch", "fetch_kib"), ("write", "write_kib")):
    files = newest(os.path.join(src, kind, "*", "*_counter_collection.csv"))
    if not files:
        continue
    seen = collections.Counter()
    for r in csv.DictReader(open(files[0])):
        k = short(r["Kernel_Name"])
        agg[k][key] += float(r["Counter_Value"])
        seen[k] += 1
    for k, n in seen.items():
        agg[k]["launches"] = max(agg[k]["launches"], n)

traffic = {}
with open(os.path.join(dst, f"{tag}_traffic.csv"), "w") as f:
    f.write("kernel,launches,fetch_KiB_raw_sum,write_KiB_sum,read_bytes_per_launch_corrected,write_bytes_per_launch,total_bytes_per_launch\n")
    for k, v in sorted(agg.items(), key=lambda kv: -(kv[1]["fetch_kib"] + kv[1]["write_kib"])):
        n = max(v["launches"], 1)
        rd = 2.0 * v["fetch_kib"] * 1024 / n
        wr = v["write_kib"] * 1024 / n
        traffic[k] = rd + wr
        f.write(f"\"{k}\",{n},{v['fetch_kib']:.0f},{v['write_kib']:.0f},{rd:.0f},{wr:.0f},{rd + wr:.0f}\n")

# bench.py books the Winograd GEMM's tile widths under one timer tag: the same aggregate here (launch-weighted)
wg = [(k, v) for k, v in agg.items() if k.startswith("wino_gemm_kernel<")]
if wg:
    n = sum(max(v["launches"], 1) for _, v in wg)
    traffic["wino_gemm_kernel<64, 3> / <128, 2>"] = sum(2.0 * v["fetch_kib"] * 1024 + v["write_kib"] * 1024 for _, v in wg) / n
tj_path = os.path.join(dst, "traffic.json")
tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
tj[workload] = traffic
# provenance of this workload's numbers: the sources they were measured on (stamp written on the GPU box by profile_round.sh) and the
# commit checked out here when they were summarised; bench.py copies it into roofline.traffic_source and reports traffic = null when
# csrc/ no longer hashes to csrc_sha16
stamp = {}
sp = os.path.join(src, "stamp.json")
if os.path.exists(sp):
    stamp = json.load(open(sp))
try:
    import subprocess
    stamp["commit_at_summary"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    stamp["csrc_dirty_at_summary"] = bool(subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "facerecognizeonnx_amd/csrc"], text=True).strip())
except Exception:  # noqa: BLE001 - no git here: the source hash alone is the stamp
    pass
stamp["profile_tag"] = tag
tj.setdefault("_stamp", {})[workload] = stamp
tj["_note"] = ("bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 / launches, from separate rocprofv3 --pmc passes "
               "(scripts/profile_round.sh); L2 fabric-side traffic, Infinity-Cache hits included")
json.dump(tj, open(tj_path, "w"), indent=1, sort_keys=True)

bench = ""
bj = os.path.join(src, "bench_under_rocprof.json")
if os.path.exists(bj):
    bench = open(bj).read().strip()
with open(os.path.join(dst, f"{tag}_summary.md"), "w") as f:
    f.write(f"# rocprofv3 summary {tag} — `python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --serial` ({workload})\n\n")
    f.write("`rocprofv3 --kernel-trace --stats` (all steps of the run: warm-up + timed + instrumented; a profiled run is slower than an un-profiled one):\n\n")
    f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for r in rows[:16]:
        f.write(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |\n")
    f.write("\nL2-fabric traffic per launch (separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes, read side x2 per the gfx950 correction):\n\n")
    f.write("| kernel | launches | read MB/launch | write MB/launch |\n|---|---|---|---|\n")
    for k, v in sorted(agg.items(), key=lambda kv: -(kv[1]["fetch_kib"] + kv[1]["write_kib"]))[:12]:
        n = max(v["launches"], 1)
        f.write(f"| {k} | {n} | {2 * v['fetch_kib'] * 1024 / n / 1e6:.1f} | {v['write_kib'] * 1024 / n / 1e6:.1f} |\n")
    if bench:
        f.write("\nbench.py line printed under rocprofv3 (HIP-event timing of the same launches):\n\n```json\n" + bench + "\n```\n")
print("wrote", sorted(os.listdir(dst)))
