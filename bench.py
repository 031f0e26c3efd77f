#!/usr/bin/env python3
"""bench.py — faces/sec of the end-to-end detect -> align -> embed path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1: launched through
torch.distributed.run, one rank per GPU) prints ONE JSON line on rank 0.

* workload (BASELINE.json `metric`): a batch of 128 synthetic 640x640 BGR frames resident in HBM,
  SCRFD det_500m (synthetic weights) -> anchor decode -> NMS -> first F faces per frame ->
  5-point alignment to 112x112 -> ArcFace IResNet-50 (synthetic weights) -> L2-normalise.
  value = faces embedded per second, whole job (all ranks).  One step = one batch.
* multi-GPU: frames are sharded one batch per rank (weak scaling); the headline path has no
  exchange step, so no collective is on the data path — only the timing barrier / max-reduce.
* roofline: the dominant kernel is the f32-MFMA implicit-GEMM convolution; every launch of it is
  bracketed by HIP events inside the library (fh_timing_*, on the stream the kernels run on) in an
  instrumented pass of the same steps right after the timed region (the ~250 event records per
  step cost ~6 % of a step, so they are kept out of `value`); achieved = algorithmic FLOP of
  those launches / their summed duration, peak = 157.3 TFLOP/s (f32 MFMA).
* cpu_baseline: the CPU oracle (oracle/, a restatement — NOT ONNX Runtime) timed on a bounded
  sample on the host cores with the reference's 4 threads (src/face_detector.cpp:10).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2516.8    # the same table: BF16 MFMA = 16x the f32 matrix rate, dense (~2.5 PF)
CFG_NAMES = ["conv_igemm_kernel<128,128,2,2>", "conv_igemm_kernel<256,64,4,1>", "conv_igemm_kernel<128,32,4,1>",
             "conv_igemm_kernel<64,64,2,2>", "dwpw_kernel (incl. the fused stem front) + dwconv3x3 kernels", "other graph ops (incl. stem_conv_u8)",
             "conv_fixup_kernel", "wino_gemm_kernel<64, 3> / <128, 2>", "wino_input_kernel + wino_output_kernel + wino_fused_kernel + wino_mix_kernel",
             "conv3x3_halo_kernel", "conv_tall_kernel<256,64,4,1> / <128,32,4,1>", "conv_pw_kernel<96|64|32>",
             "wino2_kernel<4|2> (fused F(2x2,3x3))"]
NTAGS = len(CFG_NAMES)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=128, help="frames per batch per GPU")
    ap.add_argument("--faces-per-frame", type=int, default=1)
    ap.add_argument("--workload", default="e2e", choices=["e2e", "embed", "detect", "match", "latency"])
    ap.add_argument("--queries", type=int, default=64, help="--workload match: query embeddings per step (config C4: one per frame of a 64-frame batch)")
    ap.add_argument("--crops", type=int, default=256, help="--workload embed: pre-aligned crops per batch (config 2)")
    ap.add_argument("--score-thr", type=float, default=0.5)
    ap.add_argument("--nms-thr", type=float, default=0.4)
    ap.add_argument("--as-rank", type=int, default=-1, help="--workload match on ONE GPU playing rank R of --of-world N of a row-sharded gallery "
                    "(config C5's per-rank work: its shard only, uploaded with its global index base; no collective)")
    ap.add_argument("--of-world", type=int, default=8)
    ap.add_argument("--stream-priority", default="rec", choices=["rec", "det", "none"],
                    help="streaming e2e form: which of the two HIP streams gets the higher priority (tuning)")
    ap.add_argument("--sustain-steps", type=int, default=200, help="extra steps of the same form right after the timed region, reported as "
                    "`sustained` beside `value` (0 = off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--cpu-sample-frames", type=int, default=32)
    ap.add_argument("--cpu-budget-s", type=float, default=25.0, help="wall-clock bound of each CPU baseline leg (the sample stops early)")
    ap.add_argument("--gallery", type=int, default=0, help="config C4/C5: also match every embedding against a gallery of this many "
                    "512-d rows (row-sharded over the ranks; per-rank top-k all-gathered and merged)")
    ap.add_argument("--topk", type=int, default=16)
    ap.add_argument("--exchange", default="auto", choices=["auto", "cabi", "torch"], help="--gallery with N > 1: where the two all-gathers of "
                    "the sharded top-k run.  cabi = behind the C ABI (fh_comm_* over librccl, fh_gallery_topk_sharded_dev: queries "
                    "all-gather -> scan -> one top-k all-gather -> merge, on the launch stream); torch = torch.distributed + "
                    "fh_topk_merge_dev (the only form gloo rehearsals can take); auto = cabi on nccl, torch otherwise")
    ap.add_argument("--serial", action="store_true", help="e2e: one batch at a time on one stream (fh_pipeline_run_dev).  The default since "
                    "round 3 is the library's streaming form (fh_pipeline_submit_dev): the detector of batch k+1 runs on its own HIP stream "
                    "beside the recogniser of batch k (+4.4 %% measured); every batch is complete inside the timed region (device-wide "
                    "synchronise on both sides).  The per-kernel roofline leg always runs the serial form, so that a kernel's duration is "
                    "its own; the line carries both rates")
    ap.add_argument("--overlap", action="store_true", help="(accepted for compatibility: the streaming form is the default now)")
    ap.add_argument("--from-host", action="store_true", help="secondary measurement: frames start in pinned HOST memory and are "
                    "uploaded over PCIe, double-buffered on a side stream (the PCIe-inclusive rate; never the headline value)")
    ap.add_argument("--recogniser", default="r50", choices=["r50", "mbf"], help="r50 = w600k_r50 (the reference's model, headline); "
                    "mbf = w600k_mbf (MobileFaceNet, the buffalo_s / buffalo_sc recogniser): a secondary measurement")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x2"], help="fp32 = the reference's arithmetic (headline). bf16x2 = "
                    "SECONDARY line: opt-in split-bf16 Winograd GEMMs (fh_rec_set_precision, gated on 1 - cos < 1e-3 against fp32)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) on a real multi-GPU node; gloo only to rehearse the N>1 code path")
    ap.add_argument("--all-ranks-on-device0", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--dry-run", action="store_true", help="launcher / rendezvous rehearsal without a GPU: ranks start, meet on gloo, "
                    "run the barrier + max-over-ranks timing protocol around EMPTY steps and rank 0 prints a line with value = null and "
                    "\"dry_run\": true.  Not a measurement; exists so that the N > 1 launch path is testable on a CPU-only host")
    return ap.parse_args()


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no RANK in the environment: start the N ranks ourselves, exactly as the
    driver would (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py <same args>`), as a CHILD of this process — which has not touched the GPU (no torch.cuda / libfacehip import
    yet; replacing a GPU-initialised process by exec is forbidden on this pool).  Rank 0's JSON line goes to our stdout."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")                 # dmabuf IPC only on this host driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_run(args, rank, world):
    """No GPU work at all: the rendezvous, the barrier + synchronise bracket and the MAX / SUM reductions of the real run."""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    for _ in range(args.warmup):
        pass
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    units = 0
    for _ in range(args.steps):
        units += args.frames * args.faces_per_frame
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    tot, mx = float(units), dt
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
        u = torch.tensor([float(units)], dtype=torch.float64); dist.all_reduce(u, op=dist.ReduceOp.SUM)
        mx, tot = float(t.item()), float(u.item())
    if rank == 0:
        print(json.dumps({"metric": "faces/sec end-to-end (detect+align+embed), batch=128 640x640", "value": None, "unit": "faces/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "none", "dry_run": True,
                          "units_all_ranks": tot, "config": {"workload": "DRY RUN: launcher + rendezvous rehearsal, no GPU work",
                                                             "parallelism": f"frame-sharded x{world}"}}), flush=True)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


def rec_model(args):
    from facerecognizeonnx_amd.synth import models
    if args.recogniser == "mbf":
        return models.cached("w600k_mbf_seed300.onnx", models.make_w600k_mbf)
    return models.cached("w600k_r50_seed200.onnx", models.make_w600k_r50)


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def traffic_from_profile(workload, kernel):
    """`roofline.traffic` = HBM-side bytes per launch of `kernel` from the committed counter passes (profiles/traffic.json, written by
    scripts/summarize_profile.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs).  The file carries the fingerprint of
    the sources it was measured on; when csrc/ has changed since, the number is stale and is NOT reported (null + a warning)."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return None, None
    from facerecognizeonnx_amd._lib import csrc_fingerprint
    tj = json.load(open(tpath))
    src = dict(tj.get("_stamp", {}).get(workload, {}))
    src["file"] = "profiles/traffic.json"
    live = csrc_fingerprint()
    src["csrc_sha16_now"] = live
    if src.get("csrc_sha16") != live:
        src["stale"] = True
        print(f"[bench] profiles/traffic.json [{workload}] was measured on csrc {src.get('csrc_sha16', 'unstamped')}, the tree is {live}: "
              f"roofline.traffic = null (re-run scripts/profile_round.sh + scripts/summarize_profile.py)", file=sys.stderr, flush=True)
        return None, src
    return tj.get(workload, {}).get(kernel), src


def cpu_baseline(det_path, rec_path, frames_np, args, threads=None, engine="oracle"):
    """Bounded CPU sample of the same workload: 4 threads like the reference (src/face_detector.cpp:10), or `threads`.
    engine "oracle": the C restatement end to end (`kind: port`).  engine "torch": the same pre / post-processing, but the two graphs
    evaluated by torch.nn.functional on CPU tensors (fp32, oneDNN convolutions) — a proxy for what an optimised CPU engine like the
    ONNX Runtime CPU EP the reference uses would deliver; it is NOT the reference either."""
    from oracle import oracle
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = min(4, ncpu) if threads is None else threads
    oracle.set_threads(threads)
    od, orc = oracle.OracleDetector(), oracle.OracleRecognizer()
    assert od.loadModel(det_path) and orc.loadModel(rec_path)
    if engine == "torch":
        import torch
        from oracle import torch_graph
        prev_threads = torch.get_num_threads()
        torch.set_num_threads(threads)
        tdet, trec = torch_graph.TorchGraph(od.g, torch.float32), torch_graph.TorchGraph(orc.g, torch.float32)
        dname, rname, rout = od.g.inputs[0][0], orc.g.inputs[0][0], orc.g.outputs[0][0]

        def det_net(inp):
            o = tdet.run({dname: inp[None]})
            return [o[n] for n, _ in od.g.outputs]
        od.run_network = det_net
        orc.embed_aligned = lambda aligned: oracle.l2_normalize(trec.run({rname: oracle.rec_preprocess(aligned)[None]})[rout].reshape(-1))
        # one untimed unit: oneDNN creates its primitives on first use (1.2 s for the detector graph here, against 25 ms per frame after)
        if args.workload != "embed":
            w = od.detect(frames_np[0], args.score_thr, args.nms_thr)
            if args.workload == "e2e" and len(w):
                orc.extractFeature(frames_np[0], w[0])
        else:
            orc.embed_aligned(frames_np[0])
    n = max(1, min(args.cpu_sample_frames if engine == "oracle" else 4 * args.cpu_sample_frames, len(frames_np)))
    t0 = time.perf_counter()
    faces = 0
    done = 0
    for i in range(n):
        if args.workload == "embed":
            orc.embed_aligned(frames_np[i])
            faces += 1
        else:
            det = od.detect(frames_np[i], args.score_thr, args.nms_thr)
            if args.workload == "detect":
                faces += 1
            else:
                for f in det[:args.faces_per_frame]:
                    if orc.extractFeature(frames_np[i], f).size:
                        faces += 1
        done = i + 1
        el = time.perf_counter() - t0
        if done % 4 == 0 or el > args.cpu_budget_s:
            print(f"[bench] cpu baseline ({engine}, {threads} threads): {done}/{n} sample units, {el:.1f} s", file=sys.stderr, flush=True)
        if el > args.cpu_budget_s:                       # bounded sample: the default run must finish within minutes on any host
            break
    n = done
    dt = time.perf_counter() - t0
    unit = "frames/s" if args.workload == "detect" else "faces/s"
    what = {"e2e": f"{n} of the batch's 640x640 frames: detect + decode + NMS + align + embed of the first "
                   f"{args.faces_per_frame} face(s) per frame",
            "embed": f"{n} of the batch's 112x112 crops: preprocess + IResNet-50 + L2-normalise",
            "detect": f"{n} of the batch's 640x640 frames: SCRFD + decode + NMS"}[args.workload]
    if engine == "torch":
        torch.set_num_threads(prev_threads)
        return {"value": faces / dt, "unit": unit, "cores": threads, "threads": threads,
                "kind": "torch-cpu proxy for the ORT CPU EP, not the reference", "cpu": cpu_model_name(), "host_cores_available": ncpu,
                "engine": f"torch {torch.__version__} CPU fp32 (torch.nn.functional per ONNX node, no graph fusion), oracle pre / post-processing",
                "sample": what + f" ({dt:.1f} s)"}
    return {"value": faces / dt, "unit": unit, "cores": threads, "threads": threads, "kind": "port", "cpu": cpu_model_name(), "host_cores_available": ncpu,
            "sample": what + f" ({dt:.1f} s, CPU oracle = restatement of the reference, not ONNX Runtime)"}


def bench_match(args, rank, world, local, dist, cdev, fa, torch):
    """1:N compareFaces leg (config C4's last stage / C5's sharded form): Q query embeddings against a gallery of G rows
    (row-sharded over the ranks), top-k.  One step = one query batch.  The scan kernel reads every gallery row once, so it is
    priced against HBM with G x dim x 4 algorithmic bytes; its 2*Q*G*dim FLOP on the f32 matrix cores are reported beside it."""
    from facerecognizeonnx_amd import distributed as fd
    G = args.gallery or 1_000_000
    Q, k, dim = args.queries, args.topk, 512
    gb, ge = fd.gallery_shard_base(G, rank, world)
    if args.as_rank >= 0 and world == 1:                 # one GPU plays rank R of N: C5's per-rank shard at its own size and index base
        gb, ge = fd.gallery_shard_base(G, args.as_rank, args.of_world)
    gen = torch.Generator(device="cuda"); gen.manual_seed(4 + rank)
    gal = torch.randn((ge - gb, dim), device="cuda", generator=gen)
    gal /= gal.norm(dim=1, keepdim=True)
    gallery = fa.Gallery(dim)
    gallery.upload(gal.data_ptr(), ge - gb, True, gb)
    del gal
    q = torch.randn((Q, dim), device="cuda", generator=gen); q /= q.norm(dim=1, keepdim=True)
    sc = torch.zeros((Q, k), device="cuda"); ix = torch.zeros((Q, k), dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        gallery.topk_dev(q.data_ptr(), Q, k, sc.data_ptr(), ix.data_ptr(), stream)
        if world > 1:
            fd.allgather_topk(sc, ix, k, comm_device=cdev)
        return Q
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    units = 0
    for _ in range(args.steps):
        units += step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    # kernel-level: HIP events on the stream the scan runs on, around the library call only (scan + list merge)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(5, min(args.steps, 20))
    e0.record()
    for _ in range(reps):
        gallery.topk_dev(q.data_ptr(), Q, k, sc.data_ptr(), ix.data_ptr(), stream)
    e1.record(); torch.cuda.synchronize()
    kms = e0.elapsed_time(e1) / reps
    max_dt, tot = dt, float(units)
    if dist is not None:
        t = torch.tensor([dt], device=cdev, dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
        max_dt = float(t.item())                   # every rank answers the SAME Q queries against its shard: units are not summed
    if rank == 0:
        rows_local = ge - gb
        gbs = rows_local * dim * 4 / (kms * 1e-3) / 1e9
        tf = 2.0 * Q * rows_local * dim / (kms * 1e-3) / 1e12
        out = {"metric": "queries/sec 1:N compareFaces top-k against a 512-d gallery", "value": tot / max_dt, "unit": "queries/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * max_dt / max(args.steps, 1), "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"C4 match stage: {Q} L2-normalised 512-d queries vs {G} gallery rows ({rows_local} on this rank), top-{k} "
                                      f"by (dot+1)/2", "gallery_rows": G, "queries": Q, "topk": k,
                          "parallelism": f"gallery row-sharded x{world}" + (", one all-gather of per-rank top-k + kernel merge" if world > 1 else "") +
                                         (f"; this GPU plays rank {args.as_rank} of {args.of_world} (rows [{gb}, {ge}), global indices; the exchange itself is not run)"
                                          if args.as_rank >= 0 and world == 1 else "")},
               "roofline": None}
        # the scan is priced against whichever of its two floors is higher: every gallery row once from HBM (rows x dim x 4 B at 8 TB/s) or
        # 2 Q rows dim FLOP on the f32 matrix cores (157.3 TFLOP/s) — at Q = 64 the matrix cores bind (0.43 ms against 0.26 ms per 1 M rows)
        t_hbm, t_mfma = rows_local * dim * 4 / 8000e9, 2.0 * Q * rows_local * dim / (F32_MFMA_PEAK_TFLOPS * 1e12)
        common = {"kernel": "gallery_topk_kernel (+ topk_merge_kernel)", "traffic": None, "avg_launch_us": 1e3 * kms,
                  "algorithmic_mbytes_per_launch": rows_local * dim * 4 / 1e6, "algorithmic_gflop_per_launch": 2.0 * Q * rows_local * dim / 1e9,
                  "hbm_gbs": gbs, "hbm_frac": gbs / 8000.0, "mfma_tflops": tf, "mfma_frac": tf / F32_MFMA_PEAK_TFLOPS,
                  "floor_ms": {"hbm": 1e3 * t_hbm, "mfma": 1e3 * t_mfma}}
        if t_mfma >= t_hbm:
            out["roofline"] = {"bound": "mfma", "achieved": tf, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / F32_MFMA_PEAK_TFLOPS, **common}
        else:
            out["roofline"] = {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0, **common}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


def bench_latency(args, fa, torch, models):
    """SECONDARY line: the reference's own mode — one image per call through the host-pointer entry points the C++ shim binds
    (`FaceDetector::detect`, `FaceRecognizer::extractFeature`, reference src/face_detector.cpp:139-222 / src/face_recognizer.cpp:236-304;
    callers src/main.cpp:88-104).  Host image in, host results out, PCIe and synchronisation included.  One step = one detect() +
    one extractFeature() of the best face on a 640x640 frame; per-call latencies (p50 / p99 over `--steps` x 20 calls) for the
    HIP-graph replay (default) and for the eager launch sequence, with the launches each call stands for."""
    import ctypes as C
    L = fa.lib()
    det, rec = fa.FaceDetector(), fa.FaceRecognizer()
    if not det.loadModel(models.cached("det_500m_seed100.onnx", models.make_det_500m)) or not rec.loadModel(rec_model(args)):
        raise SystemExit("model load failed: " + fa._lib.last_error())
    rng = np.random.default_rng(0)
    imgs = [rng.integers(0, 256, (640, 640, 3), dtype=np.uint8) for _ in range(8)]
    face0 = det.detect_records(imgs[0], args.score_thr, args.nms_thr)
    if not len(face0):
        raise SystemExit("latency: the synthetic detector found no face on the probe frame")
    out_f = np.zeros(face0.shape[0] + 16800, fa.FACE_DTYPE)
    calls = max(40, args.steps * 20)

    def sample(fn):
        for _ in range(max(5, args.warmup)):
            fn(0)
        ts = np.empty(calls)
        for i in range(calls):
            t0 = time.perf_counter(); fn(i); ts[i] = time.perf_counter() - t0
        return {"p50_us": float(np.percentile(ts, 50) * 1e6), "p99_us": float(np.percentile(ts, 99) * 1e6), "mean_us": float(ts.mean() * 1e6)}

    def f_det(i):
        return det.detect_records(imgs[i % len(imgs)], args.score_thr, args.nms_thr)

    def f_rec(i):
        return rec.extractFeature(imgs[0], face0[0])
    res = {}
    for mode in ("graph", "eager"):
        L.fh_set_graph_replay(1 if mode == "graph" else 0)
        res[mode] = {"detect": sample(f_det), "extractFeature": sample(f_rec)}
    L.fh_set_graph_replay(1)
    f_det(0); f_det(0); f_det(0); f_rec(0); f_rec(0); f_rec(0)
    n = C.c_longlong(0)
    det_nodes = L.fh_det_graph_stats(det.handle, C.byref(n)); rec_nodes = L.fh_rec_graph_stats(rec.handle, C.byref(n))
    step_us = res["graph"]["detect"]["p50_us"] + res["graph"]["extractFeature"]["p50_us"]
    out = {"metric": "faces/sec, batch-1 drop-in calls (detect + extractFeature of the best face, host image in / host results out)",
           "value": 1e6 / step_us, "unit": "faces/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_us / 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "secondary": True,
           "config": {"workload": "latency: one 640x640 frame per call, SCRFD det_500m detect() then IResNet-50 extractFeature() of faces[0] "
                                  "(reference mode: src/main.cpp:88-104), PCIe + synchronisation included",
                      "calls_per_sample": calls, "faces_on_probe_frame": int(len(face0))},
           "latency_us": res, "graph_nodes_per_call": {"detect": det_nodes, "extractFeature": rec_nodes},
           "note": "graph = one hipGraphLaunch per call (captured per call shape); eager = the same kernels launched one by one. "
                   "Not the headline: BASELINE.json's metric is the 128-frame batch."}
    print(json.dumps(out), flush=True)


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(args))                           # before anything touches the GPU
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} ... bench.py --gpus {args.gpus})")
    if args.dry_run:
        return dry_run(args, rank, world)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if args.all_ranks_on_device0:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.dist_backend)
    cdev = torch.device("cuda", local) if args.dist_backend == "nccl" else torch.device("cpu")   # where collectives run

    import facerecognizeonnx_amd as fa
    from facerecognizeonnx_amd.synth import models
    L = fa.lib()
    fa._lib.check(L.fh_init(local), "fh_init")

    if args.workload == "match":                                       # no networks involved: gallery scan only
        return bench_match(args, rank, world, local, dist, cdev, fa, torch)
    if args.workload == "latency":
        if world > 1:
            raise SystemExit("--workload latency is a single-GPU, single-call measurement")
        return bench_latency(args, fa, torch, models)
    # synthetic models (seeded; the genuine .onnx files are not available offline)
    if local == 0:
        det_path = models.cached("det_500m_seed100.onnx", models.make_det_500m)
        rec_path = rec_model(args)
    if dist is not None:
        dist.barrier()
    det_path = models.cached("det_500m_seed100.onnx", models.make_det_500m)
    rec_path = rec_model(args)
    det, rec = fa.FaceDetector(), fa.FaceRecognizer()
    if not det.loadModel(det_path) or not rec.loadModel(rec_path):
        raise SystemExit("model load failed: " + fa._lib.last_error())
    prec_gate = rec.set_precision(args.precision) if args.precision != "fp32" else None   # raises if the library's own gate refuses

    B, F = args.frames, args.faces_per_frame
    rng = np.random.default_rng(rank)
    stream = torch.cuda.current_stream().cuda_stream
    if args.workload == "embed":
        nunits = args.crops
        host = rng.integers(0, 256, (nunits, 112, 112, 3), dtype=np.uint8)
        data = torch.from_numpy(host).cuda()
        emb = torch.zeros((nunits, 512), device="cuda")

        def step():
            rec.embed_aligned_dev(data.data_ptr(), nunits, emb.data_ptr(), 0, stream)
            return nunits
    else:
        host = rng.integers(0, 256, (B, 640, 640, 3), dtype=np.uint8)
        data = torch.from_numpy(host).cuda()
        faces = torch.zeros((B * F, 15), device="cuda")
        frame_of = torch.zeros(B * F, dtype=torch.int32, device="cuda")
        emb = torch.zeros((B * F, 512), device="cuda")
        counts = torch.zeros(B, dtype=torch.int32, device="cuda")
        if args.workload == "detect":
            def step():
                det.detect_batch_dev(data.data_ptr(), B, 640, 640, faces.data_ptr(), F, counts.data_ptr(),
                                     args.score_thr, args.nms_thr, stream=stream)
                return B
        else:
            def step_serial():
                return fa.pipeline_run_dev(det, rec, data.data_ptr(), B, 640, 640, F, faces.data_ptr(), frame_of.data_ptr(),
                                           emb.data_ptr(), args.score_thr, args.nms_thr, stream)
            step = step_serial
        pipelined = args.workload == "e2e" and not args.serial and not args.from_host and not (args.gallery and world > 1)
        if pipelined:
            # Streaming form (fh_pipeline_submit_dev): the detector of batch k+1 is queued on its own HIP stream and
            # runs beside the recogniser of batch k (HBM-bound next to MFMA-bound work); no host sync per batch.
            # A ring of 3 per-batch result buffers with event back-pressure keeps at most 3 batches in flight.
            RING = 3
            pr = {"rec": (0, -1), "det": (-1, 0), "none": (0, 0)}[args.stream_priority]
            s_det, s_rec = torch.cuda.Stream(priority=pr[0]), torch.cuda.Stream(priority=pr[1])
            rf = [torch.zeros((B * F, 15), device="cuda") for _ in range(RING)]
            ro = [torch.zeros(B * F, dtype=torch.int32, device="cuda") for _ in range(RING)]
            re_ = [torch.zeros((B * F, 512), device="cuda") for _ in range(RING)]
            rt = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(RING)]
            done = [None] * RING
            state = {"k": 0}
            emb = re_[0]

            def step():
                k = state["k"]; slot = k % RING
                if done[slot] is not None:
                    s_det.wait_event(done[slot])                 # the slot's previous batch has been embedded
                n = fa.pipeline_submit_dev(det, rec, data.data_ptr(), B, 640, 640, F, rf[slot].data_ptr(), ro[slot].data_ptr(),
                                           re_[slot].data_ptr(), rt[slot].data_ptr(), s_det.cuda_stream, s_rec.cuda_stream,
                                           args.score_thr, args.nms_thr)
                done[slot] = torch.cuda.Event(); done[slot].record(s_rec)
                state["k"] = k + 1
                return n

    if args.from_host and args.workload == "e2e":
        # streaming-caller shape (reference main.cpp:214-258 generalised), through the library's own front end (fh_stream_*):
        # batch k+1 is copied H2D on the object's copy stream while batch k computes; results are collected one batch later.
        pinned = torch.from_numpy(host).pin_memory()
        fstream = fa.FrameStream(det, rec, B, 640, 640, F)
        state = {"inflight": 0}

        def step():                                   # noqa: F811
            n = fstream.submit((pinned.data_ptr(), B), args.score_thr, args.nms_thr)
            state["inflight"] += 1
            if state["inflight"] == 2:                # ring of 2: retire the older batch
                fstream.collect_count()
                state["inflight"] -= 1
            return n

        def drain():
            while state["inflight"]:
                fstream.collect_count()
                state["inflight"] -= 1
            return 0

    if args.gallery > 0 and args.workload != "detect":
        from facerecognizeonnx_amd import distributed as fd
        gb, ge = fd.gallery_shard_base(args.gallery, rank, world)
        grng = torch.Generator(device="cuda"); grng.manual_seed(4 + rank)
        gal = torch.randn((ge - gb, 512), device="cuda", generator=grng)
        gal /= gal.norm(dim=1, keepdim=True)
        gallery = fa.Gallery(512)
        gallery.upload(gal.data_ptr(), ge - gb, True, gb)
        del gal
        nq = emb.shape[0]
        k = args.topk
        sc = torch.zeros((nq * world, k), device="cuda"); ix = torch.zeros((nq * world, k), dtype=torch.int32, device="cuda")
        inner = step
        exchange = args.exchange if args.exchange != "auto" else ("cabi" if args.dist_backend == "nccl" else "torch")
        comm = None
        if world > 1 and exchange == "cabi":
            # the communicator behind the C ABI: rank 0's id travels through the process group that already exists
            uid = [fa.Comm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            comm = fa.Comm(rank, world, uid[0], local)

        if locals().get("pipelined"):
            # streaming form with the 1:N match: the scan of batch k is queued on the recogniser's stream right behind its embeddings
            # (per-slot result buffers), so it runs beside the detector of batch k+1 like the rest of the recogniser's work
            gsc = [torch.zeros((nq, k), device="cuda") for _ in range(RING)]
            gix = [torch.zeros((nq, k), dtype=torch.int32, device="cuda") for _ in range(RING)]

            def step():                               # noqa: F811
                slot = state["k"] % RING
                n = inner()
                for off in range(0, nq, 256):
                    m = min(256, nq - off)
                    gallery.topk_dev(re_[slot][off:off + m].data_ptr(), m, k, gsc[slot][off:off + m].data_ptr(),
                                     gix[slot][off:off + m].data_ptr(), s_rec.cuda_stream)
                done[slot] = torch.cuda.Event(); done[slot].record(s_rec)      # the slot is free once its scan has read the embeddings
                return n

            serial_inner = step_serial

            def step_serial():                        # noqa: F811
                n = serial_inner()
                for off in range(0, nq, 256):
                    m = min(256, nq - off)
                    gallery.topk_dev(emb[off:off + m].data_ptr(), m, k, sc[off:off + m].data_ptr(), ix[off:off + m].data_ptr(), stream)
                return n
        else:
            def step():                               # noqa: F811
                n = inner()
                if comm is not None:                  # both all-gathers, the scan and the merge behind the boundary, on the launch stream
                    comm.gallery_topk_sharded_dev(gallery, emb.data_ptr(), nq, k, sc.data_ptr(), ix.data_ptr(), stream)
                    return n
                if world > 1:                         # every rank scores ALL queries against its own gallery shard
                    q = fd.allgather_queries(emb.to(cdev)).to("cuda")
                else:
                    q = emb
                for off in range(0, q.shape[0], 256):
                    m = min(256, q.shape[0] - off)
                    gallery.topk_dev(q[off:off + m].data_ptr(), m, k, sc[off:off + m].data_ptr(), ix[off:off + m].data_ptr(), stream)
                if world > 1:                         # ONE all-gather of the per-rank lists, merged by the library's kernel
                    fd.allgather_topk(sc, ix, k, comm_device=cdev)
                return n

    drain = locals().get("drain", lambda: 0)       # host-frame streaming form: retire the batches still in flight
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    drain()
    timing = not args.no_kernel_timing
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    # per-step HIP events on the stream the kernels run on (torch's current stream = `stream`): the timed region is a fraction of a
    # second, so min / median / max over its steps say whether `ms_per_step` is one lucky wall-clock sample (a record costs ~2 us)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    units = 0
    mark_stream = locals().get("s_rec")                                   # streaming form: a batch ends on the recogniser's stream
    marks[0].record(mark_stream) if mark_stream is not None else marks[0].record()
    for i in range(args.steps):
        units += step()
        marks[i + 1].record(mark_stream) if mark_stream is not None else marks[i + 1].record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    units += drain()
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    # Kernel-level roofline leg: the same steps again with the library's per-launch HIP events switched on.
    # It runs right AFTER the timed region (not inside it) because the ~250 event records per step cost
    # ~6 % of the step (22.8 vs 21.5 ms); `value` must not carry the instrumentation.
    # Sustained leg: the timed region above is a fraction of a second (K = 20 steps of ~10 ms), which says little about clocks that settle
    # over seconds (MI355X_MICROARCH.md, DVFS give-back) and is invisible to a 5 s utilisation sampler.  The SAME step, same form, is
    # therefore repeated for --sustain-steps more steps right after it and reported beside `value` (never as `value`).
    sustained = None
    if args.sustain_steps > 0 and args.workload in ("e2e", "embed", "detect") and not args.from_host:
        torch.cuda.synchronize()
        tq0 = time.perf_counter()
        su = 0
        for _ in range(args.sustain_steps):
            su += step()
        torch.cuda.synchronize()
        sdt = time.perf_counter() - tq0
        sustained = {"steps": args.sustain_steps, "ms_per_step": 1e3 * sdt / args.sustain_steps, "value": su / sdt,
                     "unit": "frames/s" if args.workload == "detect" else "faces/s",
                     "what": "the same step in the same form, repeated right after the timed region on rank 0 (clock settling, visibility to "
                             "utilisation samplers); `value` stays the K timed steps"}
    serial_dt = None
    if locals().get("pipelined"):                                         # the same K steps once more, one batch at a time on one stream
        step = step_serial
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        ts0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        serial_dt = time.perf_counter() - ts0
    ms, fl, by = (C.c_double * NTAGS)(), (C.c_double * NTAGS)(), (C.c_double * NTAGS)()
    ln = (C.c_longlong * NTAGS)()
    isteps = 0
    instr_dt = 0.0
    if timing:
        isteps = max(1, min(args.steps, 5))
        L.fh_timing_enable(1)
        torch.cuda.synchronize()
        ti = time.perf_counter()
        for _ in range(isteps):
            step()
        torch.cuda.synchronize()
        instr_dt = time.perf_counter() - ti
        drain()
        L.fh_timing_enable(0)
        fa._lib.check(L.fh_timing_collect(ms, fl, by, ln, NTAGS), "fh_timing_collect")

    total_units, max_dt = float(units), dt
    if dist is not None:
        t = torch.tensor([dt], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        u = torch.tensor([float(units)], device=cdev, dtype=torch.float64)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
        max_dt, total_units = float(t.item()), float(u.item())

    if rank == 0:
        per_step_faces = units / max(args.steps, 1)
        out = {
            "metric": "faces/sec end-to-end (detect+align+embed), batch=128 640x640" if args.workload == "e2e" else
                      ("faces/sec ArcFace w600k_r50 embed, pre-aligned crops" if args.workload == "embed" else
                       "frames/sec SCRFD det_500m detect+decode+NMS"),
            "value": total_units / max_dt,
            "unit": "frames/s" if args.workload == "detect" else "faces/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * max_dt / max(args.steps, 1),
            "timed_step_ms_min": step_ms[0] if step_ms else None, "timed_step_ms_median": step_ms[len(step_ms) // 2] if step_ms else None,
            "timed_step_ms_max": step_ms[-1] if step_ms else None,
            "timed_step_ms_source": "HIP events around each of the timed steps on the launch stream, rank 0 (value / ms_per_step stay the "
                                    "barrier-bracketed wall clock, max over ranks)",
            "sustained": sustained,
            "serial_reference": None if serial_dt is None else {
                "ms_per_step": 1e3 * serial_dt / max(args.steps, 1), "value": per_step_faces * args.steps / serial_dt, "unit": "faces/s",
                "what": "the same steps run one batch at a time on one stream (fh_pipeline_run_dev) right after the timed region, rank 0: "
                        "the form the per-kernel roofline numbers below are measured in"},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": {"e2e": f"C-headline: {B} frames 640x640 per GPU, SCRFD det_500m + decode + NMS, first {F} "
                                           f"face(s)/frame aligned to 112x112, IResNet-50 (w600k_r50) embed, L2-norm",
                                    "embed": f"C2: {args.crops} pre-aligned 112x112 crops per GPU, IResNet-50 (w600k_r50), fp32",
                                    "detect": f"C3: {B} frames 640x640 per GPU, SCRFD det_500m + decode + NMS"}[args.workload],
                       "frames_per_gpu": B, "faces_per_frame": F, "faces_per_step_rank0": per_step_faces,
                       "score_thr": args.score_thr, "nms_thr": args.nms_thr,
                       "weights": "synthetic seeded (det seed 100, rec seed 200)" if args.recogniser == "r50" else
                                  "synthetic seeded (det seed 100), recogniser = MobileFaceNet w600k_mbf (seed 300) instead of the reference's w600k_r50",
                       "pipelining": "serial, one stream" if not locals().get("pipelined") else
                                     "detector of batch k+1 on its own HIP stream beside the recogniser of batch k (<= 3 batches in flight)",
                       "input_residency": "pinned host memory, double-buffered H2D over PCIe (PCIe-inclusive)" if args.from_host
                                          else "HBM-resident before the timed region",
                       "gallery_rows": args.gallery, "topk": args.topk if args.gallery else 0,
                       "parallelism": f"frame-sharded x{world}, " + ("gallery row-sharded, all-gather of queries + per-rank top-k"
                                                                     if args.gallery and world > 1 else "no data-path collective"),
                       "exchange": (locals().get("exchange") if args.gallery and world > 1 else None)},
        }
        if timing:
            conv = [(ms[i], fl[i], ln[i], i) for i in (0, 1, 2, 3, 7, 9, 10, 11, 12) if ln[i] > 0]
            if conv:
                dom = max(conv)
                tf = dom[1] / (dom[0] * 1e-3) / 1e12
                # fix-up and Winograd-transform time counts against the convs; FLOPs = what the matrix cores EXECUTE
                allms, allfl = sum(c[0] for c in conv) + ms[6] + ms[8], sum(c[1] for c in conv)
                # the same launches priced with the direct-form FLOPs of the layers they compute (Winograd GEMMs stand for 4x
                # their own work, fused F(2x2) launches for 2.25x; the timer carries that figure in the bytes slot of tags 7 / 12)
                algfl = sum(fl[i] for i in (0, 1, 2, 3, 9, 10, 11)) + by[7] + by[12]
                traffic, traffic_source = traffic_from_profile(args.workload, CFG_NAMES[dom[3]])
                out["roofline"] = {"bound": "mfma", "kernel": CFG_NAMES[dom[3]], "achieved": tf, "peak": F32_MFMA_PEAK_TFLOPS,
                                   "unit": "TFLOP/s", "frac": tf / F32_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_source,
                                   "launches": int(dom[2]), "avg_launch_us": 1e3 * dom[0] / dom[2],
                                   "algorithmic_gflop_per_launch": dom[1] / dom[2] / 1e9,
                                   "flops_counted": "executed by the matrix cores (wino_gemm_kernel = the 36 GEMMs of a Winograd F(4x4,3x3) layer, "
                                                    "booked with its own product, not with the 4x larger direct-form count of the layer)",
                                   # the same launches priced with the direct-form (2*MAC) FLOPs of the layers they compute
                                   "direct_form_achieved": (by[7] if dom[3] == 7 else dom[1]) / (dom[0] * 1e-3) / 1e12,
                                   "direct_form_frac": (by[7] if dom[3] == 7 else dom[1]) / (dom[0] * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS,
                                   "all_conv_igemm": {"achieved": allfl / (allms * 1e-3) / 1e12,
                                                      "frac": allfl / (allms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS,
                                                      "ms_per_step": allms / isteps,
                                                      "direct_form_equivalent_tflops": algfl / (allms * 1e-3) / 1e12},
                                   "per_kernel_ms_per_step": {CFG_NAMES[i]: ms[i] / isteps for i in range(NTAGS) if ln[i] > 0},
                                   "measured_on": f"{isteps} instrumented steps run right after the timed region (same inputs); "
                                                  f"instrumented step = {1e3 * instr_dt / isteps:.2f} ms"}
            if args.workload == "detect" and ln[4] > 0 and "roofline" in out and ms[4] > max(c[0] for c in conv):
                # SCRFD alone: the fused depthwise blocks take more of the step than any MFMA kernel, and they are bandwidth-bound
                # (SURVEY.md 8d): price them against HBM with the planner's algorithmic activation bytes (in + out of each fused op)
                gbs = by[4] / (ms[4] * 1e-3) / 1e9
                out["roofline"].update({"bound": "hbm", "kernel": CFG_NAMES[4], "achieved": gbs, "peak": 8000.0, "unit": "GB/s",
                                        "frac": gbs / 8000.0, "traffic": None, "launches": int(ln[4]), "avg_launch_us": 1e3 * ms[4] / ln[4],
                                        "algorithmic_gflop_per_launch": None,
                                        "algorithmic_mbytes_per_launch": by[4] / ln[4] / 1e6})
        if args.precision != "fp32":
            out["dtype"] = "f32 with split-bf16 (bf16 hi + bf16 mid per operand, 3 bf16 MFMAs, f32 accumulate) Winograd GEMMs"
            out["secondary"] = True
            out["config"]["precision"] = {"mode": args.precision, "gate": "max(1 - cos) vs the fp32 path on the library's fixed 64-crop batch < 1e-3",
                                          "measured_max_1_minus_cos": prec_gate}
            r = out.get("roofline")
            if r and r.get("kernel") == CFG_NAMES[7]:
                # the GEMM's product is executed as three bf16 MFMAs: price those against the dense bf16 peak
                r.update({"f32_equivalent_achieved": r["achieved"], "achieved": 3 * r["achieved"], "peak": BF16_MFMA_PEAK_TFLOPS,
                          "frac": 3 * r["achieved"] / BF16_MFMA_PEAK_TFLOPS,
                          "flops_counted": "3 bf16 products (hi*hi, hi*mid, mid*hi) per f32-equivalent product of the 36 Winograd GEMMs"})
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(det_path, rec_path, host, args)                  # the reference's own setting: 4 threads
            try:                                             # second leg: an optimised CPU engine's view of the same sample (torch / oneDNN)
                out["cpu_baseline_torch"] = cpu_baseline(det_path, rec_path, host, args, engine="torch")
            except Exception as e:  # noqa: BLE001 - the proxy leg must never cost the line
                out["cpu_baseline_torch"] = {"value": None, "error": f"{type(e).__name__}: {e}"}
            ncpu = out["cpu_baseline"]["host_cores_available"]
            if ncpu > 4:                                     # all cores of this process's CPU share (a GPU box gives 16 per GPU;
                # more OpenMP threads than that only spin against the cgroup quota)
                allc = cpu_baseline(det_path, rec_path, host, args, threads=min(ncpu, 16))
                # `cores` is the contract's key for "threads actually used"; on a 256-CPU host this leg is capped at the 16 a one-GPU box
                # grants, so it is labelled as what it is: a thread count, not the host's core count
                allc["threads_note"] = f"{allc['threads']} OpenMP threads (cap 16 = the CPU share of a one-GPU box) on a host reporting {ncpu} logical CPUs"
                out["cpu_baseline_all_cores"] = allc
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
