"""Formats gpurun_out/secondary/*.json (scripts/secondary_configs.sh) into profiles/<tag>_secondary_configs.md"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
rows = [("headline", "headline: 128 frames 640x640, detect+align+embed, F=1 (streaming form)", "(default)"),
        ("c2_embed", "C2: ArcFace w600k_r50, 256 pre-aligned crops", "`--workload embed`"),
        ("c3_detect", "C3: SCRFD det_500m + decode + NMS, 128 frames", "`--workload detect`"),
        ("f4", "headline with 4 faces per frame", "`--faces-per-frame 4`"),
        ("c4_gallery", "C4: 64 frames end-to-end + top-16 of a 1 M x 512 gallery", "`--gallery 1000000 --frames 64`"),
        ("c4_match", "C4 match stage alone: 64 queries vs 1 M x 512 gallery, top-16 (one scan kernel + list merge)", "`--workload match --gallery 1000000 --queries 64 --topk 16`"),
        ("c5_rank", "C5 per-rank work on one GPU: the LAST of 8 shards of a 10 M x 512 gallery (1.25 M rows, index base 8.75 M), 64 queries, top-16 (the 8-rank exchange itself is not run)", "`--workload match --gallery 10000000 --as-rank 7 --of-world 8 --queries 64 --topk 16`"),
        ("from_host", "PCIe-inclusive headline (pinned host frames, double-buffered H2D)", "`--from-host`"),
        ("serial", "headline, one batch at a time on one stream (the default streams: detector of batch k+1 beside the recogniser of batch k)", "`--serial`"),
        ("latency", "batch-1 latency through the blocking C-ABI (`fh_det_detect` + `fh_rec_extract` from host memory, HIP-graph replay)", "`--workload latency`"),
        ("mbf_embed", "MobileFaceNet (w600k_mbf, the buffalo_sc recogniser) instead of w600k_r50: 256 pre-aligned crops", "`--workload embed --recogniser mbf`"),
        ("mbf_e2e", "headline pipeline with MobileFaceNet as the recogniser", "`--recogniser mbf`")]
out = [f"# Round {tag} — other BASELINE.json configurations (same build as profiles/{tag}_summary.md)", "",
       "`python bench.py --steps 10 --warmup 3 --no-cpu-baseline <args>` on one MI355X, HBM-resident inputs unless stated, fp32.", "",
       "| config | args | value | ms / step | dominant kernel (achieved, frac of its roofline) | all conv launches |", "|---|---|---|---|---|---|"]
for name, desc, args in rows:
    p = os.path.join(ROOT, "gpurun_out", "secondary", name + ".json")
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e:
        out.append(f"| {desc} | {args} | (no result: {e}) | | | |"); continue
    r = d.get("roofline") or {}
    dom = f"{r.get('kernel', '—')} {r.get('achieved', 0):.1f} {r.get('unit', '')}, {r.get('frac', 0):.3f}" if r else "—"
    allc = r.get("all_conv_igemm", {})
    out.append(f"| {desc} | {args} | {d['value']:,.0f} {d['unit']} | {d['ms_per_step']:.2f} | {dom} | {allc.get('achieved', 0):.1f} TF |")
open(os.path.join(ROOT, "profiles", f"{tag}_secondary_configs.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
