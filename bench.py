#!/usr/bin/env python3
"""bench.py -- whole-job throughput of the hot path on N MI355X GPUs of one node.

Metric (BASELINE.json): frames/s of ORB extract + match at 640x480, ~2k keypoints per frame.
One "step" = one pass of the front end over a batch of B = 1024 synthetic frames resident in HBM:
  svgpu_orb_extract_batch_device  (pyramid, blur, per-cell FAST, grid selection, orientation, rBRIEF)
  svgpu_match_consecutive_batch_device  (frame t+1 against frame t in a ring: robust::brute_force_match with the
                                        reference's robust_match_based_track settings 0.8 / orientation check); the matcher of a batch runs on a
                                        second stream, enqueued one step late, behind the pyramid + blur of the NEXT batch's extraction
Frames shard one batch per GPU (no data-path collective; RCCL is used for the barrier and the MAX of the
per-rank times only) -> "scaling": "weak".  Rank 0 prints ONE JSON line.

`python bench.py --gpus N` without a torchrun environment starts its own N ranks (torch.distributed.run, 127.0.0.1) and
relays rank 0's line.

Extra objects on that line:
  roofline      dominant kernel (picked by a per-kernel HIP-event pre-pass), timed with HIP events on the launch stream over
                the timed region: achieved = algorithmic bytes per launch / mean launch time; `kernels` = the same figures for
                EVERY kernel class of the step from the pre-pass (describe and the matcher kernels the north_star's 0.6 bar
                names included); traffic / valu_issue from the committed PMC passes, only while the kernel sources still
                hash to what those passes measured
  latency       one frame at a time through the host-buffer entry points (what tracking_module calls per frame)
  stereo        BASELINE config 4 shape: stereo pairs at 1241x376, both extractions + match::stereo::compute per step
  local_ba      LM iterations/s on config 3 (20 KF / 10k landmarks / ~60k observations), with its own roofline
  global_ba     LM iterations/s on config 5 (500 KF / 200k landmarks / 1.2M observations; block-Jacobi PCG); with N > 1 the
                landmark-sharded solve over the library's own RCCL communicator (svgpu_comm_init), all ranks, same problem
  global_ba_large  the same loop with 1.6 M landmarks / 9.6 M observations (the size at which sharding the observations can pay), with the
                per-phase projection for 2 / 4 / 8 ranks beside the single-GPU measurement; N > 1: keyframe-segment shards, bytes exchanged
  cpu_baseline  the oracle ("port" of the reference CPU path), single thread, on bounded samples
"""
from __future__ import annotations

import argparse
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 640, 480
LOWE, CHECK_ORI = 0.8, 1  # module/frame_tracker.cc:98
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
I8_MFMA_PEAK_TOPS = 5000.0  # dense int8 MFMA = 2 x the 2.5 PFLOP/s bf16 figure (measured ceiling >= 3.94 POP/s)


def csrc_hash() -> str:
    """sha256 over the kernel / host sources of libsvgpu that the front end's kernels are built from: PMC-derived figures (traffic, VALU issue,
    LDS / MFMA counters of the ORB and matcher kernels) are only reported for the sources they were measured on.  The bundle adjusters'
    own translation units (ba_*.hip / ba_*.h / svgpu_ba.hip) define none of those kernels and are left out, so that work on the adjusters
    does not void the front end's counters; every header the front end includes (svgpu_internal.h among them) is in."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "stella_vslam_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.startswith("ba_") or name == "svgpu_ba.hip":
            continue
        if name.endswith((".hip", ".h", ".inc")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` outside torchrun: start the N ranks ourselves and relay rank 0's JSON line."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + [a for a in sys.argv[1:]]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="frames per GPU per step (the kernels' tails and the latency-bound matcher kernels amortise over a larger batch: 64 -> 256 is +7 %, 256 -> 1024 another +7 %)")
    ap.add_argument("--batch2", type=int, default=256, help="second batch point printed beside the headline (0 = off)")
    ap.add_argument("--detail", default=os.environ.get("BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json")),
                    help="where the full result object goes (the stdout line is the <= 4 KB summary of it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ba", action="store_true")
    ap.add_argument("--no-ba-large", action="store_true", help="skip the 9.6 M-observation global-BA leg (about 20 s of scene generation per rank)")
    ap.add_argument("--no-extra", action="store_true", help="skip the latency / stereo legs")
    ap.add_argument("--leg-timeout", type=int, default=420, help="seconds the secondary legs (latency, stereo, BA, CPU baseline) may take before the line is printed without them")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        print(f"WORLD_SIZE={world} != --gpus {args.gpus}", file=sys.stderr)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (the product path has no CPU fallback)", file=sys.stderr)
        return 2
    n_dev = torch.cuda.device_count()
    local_rank %= n_dev  # one rank per GPU on a real node; lets the multi-rank path be exercised on fewer GPUs
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" IS RCCL on ROCm; BENCH_DIST_BACKEND=gloo only exists to test the multi-rank control flow on a single GPU
        dist.init_process_group(os.environ.get("BENCH_DIST_BACKEND", "nccl"), rank=rank, world_size=world)

    from stella_vslam_amd import feature, synthetic
    from stella_vslam_amd._lib import lib

    B = args.batch
    ctx = feature.Context(local_rank, priority=1)  # extraction is the longer leg of the two-stream pipeline: it gets the CUs first
    L = lib()

    def barrier():
        if world > 1:
            dist.barrier()

    # ---- synthetic input, resident in HBM before the timed region
    frames_np = synthetic.frame_sequence(B, W, H, seed=0x5EED + 7919 * rank)
    fe = run_front_end(ctx, L, frames_np, B, args.steps, args.warmup, barrier, world)
    src_hash = csrc_hash()
    kernels, dom = roofline_entries(fe, src_hash, B, world)

    result = {
        "metric": "frames/s ORB-extract+match @640x480,2k kpts",
        "value": round(B * world * args.steps / fe["dt"], 2),
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(fe["dt"] / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": "synthetic 640x480 frame sequence (BASELINE configs[1] proxy: EuRoC imagery unavailable "
                               "offline), ORB extract + brute-force match vs previous frame",
                   "frames_per_gpu_per_step": B, "keypoints_per_frame": round(fe["n_kp"], 1),
                   "matches_per_pair": round(fe["n_match"], 1), "parallelism": f"frames sharded x{world}, no collective; extraction and matcher on two HIP streams (every step = one extraction + one matcher pass, the matcher's of the previous batch)",
                   "input_residency": ("the same %d frames (%.0f MB) are re-read every step: %s "
                                      "(no kernel of the step is byte-bound; see roofline.kernels[].traffic)" % (B, B * W * H / 1e6, "they fit the 256 MB Infinity Cache, so level-0 reads need not reach HBM"
                                                                                                       if B * W * H <= 256e6 else "more than the 256 MB Infinity Cache holds, so level-0 reads do reach HBM")),
                   "per_kernel_timing": "HIP events around EVERY kernel class on its launch stream inside the timed region (svgpu_profile_select \"*\"); "
                                        "the same steps re-timed with the events off: %.4f ms per step" % fe["ms_per_step_unprofiled"]},
        "roofline": dict(dom, csrc_hash=src_hash, per_kernel_ms_per_step={k["kernel"]: round(k["mean_launch_ms"] * k["launches_per_step"], 4) for k in kernels},
                         kernels=kernels),
    }
    del fe
    # the operating point of rounds 1-3 (256 frames per step) beside the headline's, so that rounds stay comparable
    if args.batch2 and args.batch2 != B:
        B2 = min(args.batch2, B)
        fe2 = run_front_end(ctx, L, frames_np[:B2], B2, args.steps, args.warmup, barrier, world, profile=False)
        result["second_batch_point"] = {"frames_per_gpu_per_step": B2, "value": round(B2 * world * args.steps / fe2["dt"], 2), "unit": "frames/s",
                                        "ms_per_step": round(fe2["dt"] / args.steps * 1e3, 4), "keypoints_per_frame": round(fe2["n_kp"], 1)}
        del fe2
    # the same step fed from page-locked host memory (the upload of the next batch on a copy stream): what the link allows, beside `value`
    if not args.no_extra:
        try:
            n_h = max(4, args.steps // 4)
            feh = run_front_end(ctx, L, frames_np, B, n_h, 2, barrier, world, profile=False, h2d=True)
            result["with_h2d"] = {"value": round(B * world * n_h / feh["dt"], 2), "unit": "frames/s", "ms_per_step": round(feh["dt"] / n_h * 1e3, 4), "steps": n_h,
                                  "what": "the headline step with the batch uploaded from page-locked host memory every step (copy stream, two device images): %.0f MB per step" % (B * W * H / 1e6)}
            del feh
        except Exception as e:  # (a secondary leg)
            result["with_h2d"] = {"error": repr(e)}

    torch.cuda.empty_cache()

    # The secondary legs must never cost the headline: if one of them hangs (the sharded global BA is the only code of this file that
    # cannot be exercised on the one-GPU test box at N > 1), every rank's watchdog ends its process after rank 0 has printed the line
    # it has.  ctypes and torch.distributed calls release the GIL, so the timer thread runs while the main thread is blocked.
    import threading
    legs_done = threading.Event()

    def bail():
        if legs_done.is_set():
            return
        if rank == 0:
            result.setdefault("global_ba", {"error": "secondary legs timed out after %d s; headline unaffected" % args.leg_timeout})
            emit(result, args.detail)
        os._exit(0)

    watchdog = threading.Timer(args.leg_timeout, bail)
    watchdog.daemon = True
    watchdog.start()

    if rank == 0 and world == 1 and not args.no_extra:
        for key, fn in (("natural_images", lambda: bench_natural(ctx, L, B, want_cpu=not args.no_cpu_baseline)),
                        ("tracked_frame", lambda: bench_tracked_frame(frames_np, want_cpu=not args.no_cpu_baseline)),
                        ("latency", lambda: bench_latency(ctx, frames_np)), ("stereo", lambda: bench_stereo(local_rank))):
            try:
                result[key] = fn()
            except Exception as e:  # a secondary leg must never hide the headline number
                result[key] = {"error": repr(e)}
    if not args.no_ba:
        try:
            lb = bench_local_ba(ctx, rank, world)
            if rank == 0:
                result["local_ba"] = lb
        except Exception as e:
            if rank == 0:
                result["local_ba"] = {"error": repr(e)}
        if rank == 0 and world == 1:
            try:
                result["mapping_keyframe"] = bench_mapping_keyframe(want_cpu=not args.no_cpu_baseline)
            except Exception as e:
                result["mapping_keyframe"] = {"error": repr(e)}
        try:
            gb = bench_global_ba(local_rank, rank, world)
            if rank == 0:
                result["global_ba"] = gb
        except Exception as e:
            if rank == 0:
                result["global_ba"] = {"error": repr(e)}
        if not args.no_ba_large:
            try:
                gl = bench_global_ba(local_rank, rank, world, large=True)
                if rank == 0:
                    result["global_ba_large"] = gl
            except Exception as e:
                if rank == 0:
                    result["global_ba_large"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(frames_np, want_ba=not args.no_ba)
    legs_done.set()
    watchdog.cancel()
    if rank == 0:
        emit(result, args.detail)
    if world > 1:
        closer = threading.Timer(60, lambda: os._exit(0))  # the line is out: a stuck teardown must not keep the launcher waiting
        closer.daemon = True
        closer.start()
        dist.barrier()
        dist.destroy_process_group()
    return 0


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def compact_line(r):
    """The ONE stdout line: <= 4 KB (the driver's capture holds 8 KB of tail; round 4's 20.8 KB line was cut and counted as unmeasured).  Every key the
    contract names, the dominant kernel's roofline with the two alternates as three-number objects, the CPU baseline, and one-line summaries
    of the secondary legs.  Everything else (per-kernel arrays, projections, per-phase legs) is in the detail file and on stderr."""
    out = {k: r[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data") if k in r}
    cfg = r.get("config", {})
    out["config"] = {"workload": "synthetic 640x480 sequence (configs[1] proxy), ORB extract + brute-force match vs previous frame, resident in HBM",
                     **_pick(cfg, ("frames_per_gpu_per_step", "keypoints_per_frame", "matches_per_pair"))}
    rf = r.get("roofline", {})
    ro = _pick(rf, ("kernel", "rocprof_kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "mean_launch_ms", "selection"))
    if isinstance(rf.get("valu_bound"), dict):
        ro["valu_issue_frac"] = rf["valu_bound"].get("frac")
    alts = {}
    for key in ("overlapped_stream_longest", "matrix_core_kernel"):
        if isinstance(rf.get(key), dict):
            alts[rf[key].get("kernel", key)] = _pick(rf[key], ("achieved", "frac", "mean_launch_ms"))
            if isinstance(rf[key].get("standalone"), dict):  # the kernel's own duration (extraction stream idle) beside its elapsed time in the pipeline
                alts[rf[key].get("kernel", key)]["standalone"] = _pick(rf[key]["standalone"], ("achieved", "frac", "mean_launch_ms"))
    for k in rf.get("kernels", []):   # the two kernels north_star's 0.6 bar names beside the matcher's
        if k["kernel"] in ("k_describe", "k_fast") and k["kernel"] != ro.get("kernel"):
            hb = k.get("hbm_context") or k
            alts[k["kernel"]] = {"achieved": hb["achieved"], "frac": hb["frac"], "mean_launch_ms": k["mean_launch_ms"]}
    if alts:
        ro["alternates"] = alts
    if rf.get("per_kernel_ms_per_step"):
        ro["ms_per_step"] = rf["per_kernel_ms_per_step"]
    out["roofline"] = ro
    # the rule is frozen: `roofline` = the longest kernel of the critical (extraction) stream, `roofline_overlapped` = the longest
    # kernel per step of the stream that runs beside it (the matcher), each with the name it carries in a rocprofv3 trace
    if isinstance(rf.get("overlapped_stream_longest"), dict):
        out["roofline_overlapped"] = dict(_pick(rf["overlapped_stream_longest"], ("kernel", "rocprof_kernel", "bound", "achieved", "peak", "unit", "frac", "mean_launch_ms", "mfma_busy_frac")),
                                          selection="largest time per step among the kernels of the overlapped (matcher) stream; elapsed time beside the extraction stream")
        if isinstance(rf["overlapped_stream_longest"].get("standalone"), dict):
            out["roofline_overlapped"]["standalone"] = _pick(rf["overlapped_stream_longest"]["standalone"], ("achieved", "frac", "mean_launch_ms"))
    if "second_batch_point" in r:
        out["second_batch_point"] = _pick(r["second_batch_point"], ("frames_per_gpu_per_step", "value", "ms_per_step"))
    if "with_h2d" in r:
        out["with_h2d"] = _pick(r["with_h2d"], ("value", "ms_per_step", "error"))
    cb = r.get("cpu_baseline")
    if isinstance(cb, dict):
        c = _pick(cb, ("value", "unit", "cores", "kind", "error"))
        if "sample" in cb:
            c["sample"] = cb["sample"][:120]
        for k in ("local_ba", "global_ba"):
            if isinstance(cb.get(k), dict):
                c[k] = _pick(cb[k], ("value", "unit", "ms_per_call", "error"))
        if isinstance(cb.get("all_host_cores"), dict):
            c["all_host_cores"] = _pick(cb["all_host_cores"], ("value", "cores"))
        out["cpu_baseline"] = c
    for k in ("local_ba", "global_ba", "global_ba_large", "mapping_keyframe"):
        if isinstance(r.get(k), dict):
            out[k] = _pick(r[k], ("value", "unit", "ms_per_call", "iters_per_call", "n_gpus", "setup_ms", "gpu_ms", "cpu_ms", "host_share", "error"))
            if isinstance(r[k].get("error"), str):
                out[k]["error"] = r[k]["error"][:120]
    tf = r.get("tracked_frame")
    if isinstance(tf, dict):
        t = {}
        for key, name in (("chain", "chain_ms"), ("chain_stereo", "chain_stereo_ms"), ("resident_frames", "per_call_ms")):
            if isinstance(tf.get(key), dict) and "ms_per_frame" in tf[key]:
                t[name] = tf[key]["ms_per_frame"]
        if isinstance(tf.get("cpu_port"), dict) and "ms_per_frame" in tf["cpu_port"]:
            t["cpu_ms"] = tf["cpu_port"]["ms_per_frame"]
        if isinstance(tf.get("cpu_port_stereo"), dict) and "ms_per_frame" in tf["cpu_port_stereo"]:
            t["cpu_stereo_ms"] = tf["cpu_port_stereo"]["ms_per_frame"]
        if "error" in tf:
            t["error"] = str(tf["error"])[:120]
        out["tracked_frame"] = t
    if isinstance(r.get("stereo"), dict):
        out["stereo"] = _pick(r["stereo"], ("pairs_per_s", "ms_per_step", "error"))
    if isinstance(r.get("latency"), dict):
        out["latency"] = _pick(r["latency"], ("extract_ms", "match_ms", "frames_per_s", "error"))
    if isinstance(r.get("natural_images"), dict):
        out["natural_images"] = _pick(r["natural_images"], ("frames_per_s", "keypoints_per_frame", "error"))
    out["detail"] = "bench_detail.json (full per-kernel arrays, projections and per-phase legs; also on stderr)"
    line = json.dumps(out, separators=(",", ":"))
    if len(line) > 4000:   # never let a leg's growth cost the headline again: drop the summaries, keep the contract keys
        for k in ("natural_images", "latency", "stereo", "tracked_frame", "mapping_keyframe", "global_ba_large", "with_h2d", "second_batch_point"):
            out.pop(k, None)
            line = json.dumps(out, separators=(",", ":"))
            if len(line) <= 4000:
                break
    return line


def emit(result, detail_path):
    """Full object -> detail file + stderr; the <= 4 KB summary -> the LAST stdout line."""
    full = json.dumps(result)
    try:
        with open(detail_path, "w") as f:
            f.write(full + "\n")
    except OSError as e:
        print(f"bench.py: could not write {detail_path}: {e}", file=sys.stderr)
    print(full, file=sys.stderr, flush=True)
    print(compact_line(result), flush=True)


CRITICAL_STREAM = ("k_resize", "k_blur", "k_fast", "k_select", "k_describe")  # the extraction chain: its kernels add up to the step
KERNEL_CLASSES = ("k_resize", "k_blur", "k_fast", "k_select", "k_describe", "k_bf_binsort", "k_bf_topk", "k_bf_replay")
# profiling class (svgpu_profile_read_class) -> the kernel symbol rocprofv3 lists for it in profiles/*_kernel_stats.csv, where the two differ
# class name of the profiler (svgpu_profile_*) -> kernel name in a rocprofv3 trace (profiles/r06_kernel_stats.csv)
ROCPROF_KERNEL = {"k_resize": "k_pyramid_lds", "k_blur": "k_blur<64>", "k_fast": "k_fast", "k_select": "k_select", "k_describe": "k_describe_bands",
                  "k_bf_binsort": "k_bf_binsort", "k_bf_topk": "k_bf_mfma", "k_bf_replay": "k_bf_replay<5, 512>"}


PACKED_ANGLES = os.environ.get("BENCH_PACKED_ANGLES", "1") != "0"  # (A/B aid: 0 = the matcher reads the angles out of the keypoint records)
_MATCHER_STREAM = None


def matcher_stream():
    """The matcher's stream, created ONCE per process (as a tracking thread would).  Measured (round 5, tools/stream_alias_exp.py, 256 frames/step, 8 calls
    in one process): with one stream kept, 212-215 k frames/s on every call; with a fresh torch.cuda.Stream per call, calls 2 and 5 drop to 163-166 k —
    some streams of torch's pool land on a hardware queue that serialises against the extraction stream's (k_select doubles, k_resize gets faster:
    the overlap pattern changes, not the kernels).  Rounds 1-5's 'second_batch_point' was such a second stream."""
    global _MATCHER_STREAM
    import torch
    if _MATCHER_STREAM is None:
        _MATCHER_STREAM = torch.cuda.Stream(priority=int(os.environ.get("BENCH_MATCH_PRIORITY", "0")))
    return _MATCHER_STREAM


def run_front_end(ctx, L, frames_np, B, steps, warmup, barrier, world, profile=True, h2d=False):
    """The step of this bench on one rank: ORB extraction of the B resident frames + brute-force match of each against the previous one
    (ring of B pairs).  Two HIP streams: extraction of step t+1 (stream A = the context's) overlaps the matcher of step t (stream B),
    which leaves most CUs idle during its sort / greedy-replay kernels.  Two output buffer sets alternate; events order
    extract(t) -> match(t) and match(t) -> extract(t+2) (the next writer of that set).
    The timed region (barrier + synchronize on both sides, MAX over ranks) runs with HIP events around EVERY kernel class, so the
    per-kernel figures are those of the kernels inside this very pipeline; the same number of steps is then re-timed with the events
    off to show what the bracketing costs."""
    import torch
    from stella_vslam_amd import feature
    from stella_vslam_amd.distributed import max_over_ranks
    Wf, Hf = frames_np.shape[2], frames_np.shape[1]
    params = feature.orb_params()
    NL = params.num_levels_
    ctx.check(L.svgpu_orb_configure(ctx.handle, Wf, Hf, B, C.c_float(params.scale_factor_), NL, params.ini_fast_thr_,
                                    params.min_fast_thr_, C.c_uint(800)), "svgpu_orb_configure")
    cap = L.svgpu_orb_max_keypoints(ctx.handle)
    level_px = []
    for l in range(NL):
        w_, h_ = C.c_int(), C.c_int()
        L.svgpu_orb_level_size(ctx.handle, l, C.byref(w_), C.byref(h_))
        level_px.append(w_.value * h_.value)
    stream = torch.cuda.ExternalStream(ctx.stream)
    stream_b = matcher_stream()
    # the matcher of batch t is enqueued one step LATE, behind a stage of the extraction of batch t+1 (svgpu_orb_stream_wait_stage): -1 = right
    # behind its own extraction (rounds 1-4), 0 = when FAST starts, 1 = when the descriptor kernel starts
    # Measured (same box, 100 steps, twice each): -1: 218.4 / 218.6 k frames/s, 0: 224.1 / 224.0 k, 1: 218.7 k.  Since the pyramid became an efficient kernel
    # (round 5) nothing on the extraction stream is latency-bound enough to hide the matcher in: the step is close to the SUM of the two
    # streams' work, and the placement only decides whose tails overlap.
    match_stage = int(os.environ.get("BENCH_MATCH_STAGE", "0"))
    NBUF = 2 if match_stage < 0 else 3
    nc = 1 + NL
    with torch.cuda.stream(stream):
        frames = torch.from_numpy(np.ascontiguousarray(frames_np)).cuda()
        bufs = [dict(kps=torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda"),
                     desc=torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda"),
                     counts=torch.zeros(B * nc, dtype=torch.int32, device="cuda"),
                     matched=torch.zeros(B * cap, dtype=torch.int32, device="cuda"),
                     nmatch=torch.zeros(B, dtype=torch.int32, device="cuda"),
                     angles=torch.zeros(B * cap, dtype=torch.float32, device="cuda"),
                     ev_ext=torch.cuda.Event(), ev_match=torch.cuda.Event(), used=False) for _ in range(NBUF)]
    stream.synchronize()
    state = {"i": 0}
    up = None
    if h2d:
        # the PCIe-inclusive variant: the batch lives in page-locked HOST memory and goes up every step -- the upload of batch t + 1 on a copy
        # stream beside the work on batch t, two device images of the batch alternating
        stream_c = torch.cuda.Stream()
        host = torch.from_numpy(np.ascontiguousarray(frames_np)).pin_memory()
        with torch.cuda.stream(stream_c):
            dev = [frames, torch.empty_like(frames)]
        up = dict(stream=stream_c, host=host, dev=dev, ev_up=[torch.cuda.Event(), torch.cuda.Event()], ev_free=[torch.cuda.Event(), torch.cuda.Event()], primed=False)
        stream_c.synchronize()

    def upload(j):
        with torch.cuda.stream(up["stream"]):
            if up["primed"]:
                up["stream"].wait_event(up["ev_free"][j])  # the extraction that read this image has finished
            up["dev"][j].copy_(up["host"], non_blocking=True)
            up["ev_up"][j].record(up["stream"])

    def step():
        """Extraction of batch t on stream A, then its matcher on stream B: the matcher of batch t overlaps the extraction of batch
        t+1.  (Measured alternative, rejected: holding the matcher back until the next batch's pyramid kernel has finished: 148 k
        instead of 157 k frames/s; the latency-bound pyramid is exactly where the matcher's kernels fit best.)"""
        bf = bufs[state["i"] % NBUF]
        state["i"] += 1
        kps, desc, counts = bf["kps"], bf["desc"], bf["counts"]
        if bf["used"]:
            stream.wait_event(bf["ev_match"])  # the matcher of two steps ago has finished reading this buffer set
        bf["used"] = True
        src = frames
        if up is not None:
            j = (state["i"] - 1) & 1
            if not up["primed"]:
                upload(j)
            stream.wait_event(up["ev_up"][j])
            src = up["dev"][j]
        # (the extractor also leaves the angles as a packed array: the matcher's angle-bin sort then reads 4 instead of 28 bytes per keypoint)
        ctx.check(L.svgpu_orb_extract_batch_device_angles(ctx.handle, C.c_void_p(src.data_ptr()), B, C.c_size_t(Wf * Hf), Wf, None,
                                                          C.c_size_t(0), 0, C.c_void_p(kps.data_ptr()), C.c_void_p(desc.data_ptr()),
                                                          cap, C.c_void_p(counts.data_ptr()), C.c_void_p(bf["angles"].data_ptr()) if PACKED_ANGLES else None, None), "extract_batch")
        bf["ev_ext"].record(stream)
        if up is not None:
            up["ev_free"][j].record(stream)
            up["ev_free"][j ^ 1].record(stream) if not up["primed"] else None
            up["primed"] = True
            upload(j ^ 1)  # the next step's batch, while this one is worked on
        if match_stage >= 0:
            prev = state.get("pending")
            state["pending"] = bf
            if prev is None:
                return
            bf = prev
            kps, desc, counts = bf["kps"], bf["desc"], bf["counts"]
            ctx.check(L.svgpu_orb_stream_wait_stage(ctx.handle, match_stage, C.c_void_p(stream_b.cuda_stream)), "stream_wait_stage")
        stream_b.wait_event(bf["ev_ext"])
        # pair t = (frame (t + 1) % B, keyframe = frame t), t = 0..B-1, straight from the extractor's batch layout
        ctx.check(L.svgpu_match_consecutive_batch_device_angles(
            ctx.handle, B, C.c_void_p(desc.data_ptr()), C.c_void_p(kps.data_ptr()), C.c_void_p(bf["angles"].data_ptr()) if PACKED_ANGLES else None,
            C.c_void_p(counts.data_ptr()), cap, nc, None,
            C.c_float(LOWE), CHECK_ORI, C.c_void_p(bf["matched"].data_ptr()), C.c_void_p(bf["nmatch"].data_ptr()),
            C.c_void_p(stream_b.cuda_stream)), "match_batch")
        bf["ev_match"].record(stream_b)

    def sync_all():
        ctx.synchronize()
        torch.cuda.synchronize()

    def timed(n):
        sync_all()
        barrier()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        sync_all()
        barrier()
        return time.perf_counter() - t0

    for _ in range(max(warmup, 1)):
        step()
    sync_all()
    n_kp = bufs[0]["counts"].view(B, nc)[:, 0].float().mean().item()
    n_match = bufs[0]["nmatch"].float().mean().item()
    # ---- timed region: exactly K steps between barrier + synchronize, every kernel class bracketed by HIP events on its stream
    per_kernel, ops, alone = {}, C.c_ulonglong(), {}
    if profile:
        L.svgpu_profile_select(ctx.handle, b"*")
        dt = max_over_ranks(timed(steps))  # MAX over ranks (RCCL when world > 1)
        for name in KERNEL_CLASSES:
            ms, n = C.c_double(), C.c_longlong()
            L.svgpu_profile_read_class(ctx.handle, name.encode(), C.byref(ms), C.byref(n))
            per_kernel[name] = (ms.value, n.value)
        L.svgpu_profile_mfma_ops(ctx.handle, C.byref(ops))
        L.svgpu_profile_select(ctx.handle, None)
        dt_plain = max_over_ranks(timed(steps))
        # the matcher kernels ALONE (outside the timed region, on the buffers of the last batch): inside the pipeline they run on the issue slots the
        # extraction stream leaves, so their elapsed time there is 2 - 3 x their own; both figures are reported
        sync_all()
        bf = bufs[(state["i"] - 1) % NBUF]
        L.svgpu_profile_select(ctx.handle, b"*")
        for _ in range(4):
            ctx.check(L.svgpu_match_consecutive_batch_device_angles(
                ctx.handle, B, C.c_void_p(bf["desc"].data_ptr()), C.c_void_p(bf["kps"].data_ptr()), C.c_void_p(bf["angles"].data_ptr()) if PACKED_ANGLES else None,
                C.c_void_p(bf["counts"].data_ptr()), cap, nc, None,
                C.c_float(LOWE), CHECK_ORI, C.c_void_p(bf["matched"].data_ptr()), C.c_void_p(bf["nmatch"].data_ptr()), C.c_void_p(stream_b.cuda_stream)), "match_batch")
        sync_all()
        for name in ("k_bf_binsort", "k_bf_topk", "k_bf_replay"):
            ms, n = C.c_double(), C.c_longlong()
            L.svgpu_profile_read_class(ctx.handle, name.encode(), C.byref(ms), C.byref(n))
            if n.value:
                alone[name] = ms.value / n.value
        L.svgpu_profile_select(ctx.handle, None)
    else:
        dt = dt_plain = max_over_ranks(timed(steps))
    del bufs, frames
    return {"dt": dt, "ms_per_step_unprofiled": dt_plain / steps * 1e3, "per_kernel": per_kernel, "mfma_ops": ops.value, "steps": steps, "matcher_alone_ms": alone,
            "n_kp": n_kp, "n_match": n_match, "alg": algorithmic_bytes(level_px, n_kp, B), "level_px": level_px}


def roofline_entries(fe, src_hash, B, world):
    """roofline.kernels[] (one entry per kernel class, all from the timed region) and the dominant kernel's object."""
    # PMC-derived figures (rocprofv3 passes of this same command, tools/pmc_traffic.py / tools/pmc_valu.py) are valid only for the
    # kernel sources they were taken on: every profiles/*_traffic.json / *_valu_issue.json carries the csrc hash of its run
    traffic_json, valu_json = load_profile_json("*_traffic.json", src_hash, B, world), load_profile_json("*_valu_issue.json", src_hash, B, world)
    lds_json = load_profile_json("*_lds_mfma.json", src_hash, B, world)  # LDS busy / bank-conflict and matrix-pipe busy fractions (tools/pmc_lds_mfma.py)

    def traffic_of(name):
        return None if traffic_json is None else traffic_json["kernels"].get(name, {}).get("total")

    def valu_of(name):
        if valu_json is None or name not in valu_json["kernels"]:
            return None
        kv = valu_json["kernels"][name]
        return {"wave_insts_per_launch": kv["valu_wave_insts"], "cycles_per_wave_inst": valu_json["cycles_per_valu_wave_inst"],
                "simds": valu_json["simds"], "kernel_cycles": kv["kernel_cycles"], "frac": kv["valu_issue_frac"]}

    kernels = []
    for name, (kms, kn) in fe["per_kernel"].items():
        if kn == 0:
            continue
        mean_ms = kms / kn
        lps = kn / fe["steps"]
        b_, u_, p_ = ROOFS.get(name, ("hbm", "GB/s", HBM_PEAK_GBS))
        if b_ == "mfma":
            # primary figure: the int8 operations the matrix cores EXECUTED (counted by the kernel: it multiplies only the pairs inside
            # the +-30 degree orientation windows of robust.cc:279); context: the reference's all-pairs work over the same time
            per_launch = fe["mfma_ops"] / kn
            allpairs = fe["alg"][name] / lps
        else:
            per_launch = fe["alg"][name] / lps
        ach = per_launch / (mean_ms * 1e-3) / (1e9 if b_ == "hbm" else 1e12)
        entry = {"kernel": name, "rocprof_kernel": ROCPROF_KERNEL.get(name, name), "bound": b_, "unit": u_, "peak": p_, "achieved": round(ach, 2), "frac": round(ach / p_, 5),
                 "mean_launch_ms": round(mean_ms, 5), "launches_per_step": lps,
                 ("algorithmic_bytes_per_launch" if b_ == "hbm" else "executed_int8_ops_per_launch"): int(per_launch),
                 "traffic": traffic_of(name), "valu_issue": valu_of(name)}
        if b_ == "mfma":
            entry["all_pairs_context"] = {"ops_per_launch": int(allpairs), "achieved": round(allpairs / (mean_ms * 1e-3) / 1e12, 2),
                                          "frac": round(allpairs / (mean_ms * 1e-3) / 1e12 / p_, 5),
                                          "note": "2 x 256 x N1 x N2 per pair = the work of robust.cc:271-314 if every pair were multiplied"}
        if name in fe.get("matcher_alone_ms", {}):  # overlapped stream: the kernel's own duration, measured with the extraction stream idle
            a_ms = fe["matcher_alone_ms"][name]
            a_ach = per_launch / (a_ms * 1e-3) / (1e9 if b_ == "hbm" else 1e12)
            entry["standalone"] = {"mean_launch_ms": round(a_ms, 5), "achieved": round(a_ach, 2), "frac": round(a_ach / p_, 5)}
        vi = entry["valu_issue"]
        if b_ == "hbm" and vi is not None and vi["frac"] >= 0.5 and vi["frac"] > entry["frac"]:
            # the kernel is bound by VALU issue, not by bytes: the issue fraction (VALU wave-instructions x 4 cycles over the SIMD-cycles of the
            # launch, from the committed PMC pass) is the primary figure; the byte fraction stays as context
            entry["hbm_context"] = {"achieved": entry["achieved"], "unit": entry["unit"], "peak": entry["peak"], "frac": entry["frac"]}
            entry.update({"bound": "valu", "unit": "VALU issue slots", "peak": 1.0, "achieved": vi["frac"], "frac": vi["frac"]})
        if lds_json is not None and name in lds_json["kernels"]:
            lk = lds_json["kernels"][name]
            entry["lds_util"], entry["lds_bank_conflict_frac"] = lk["lds_util"], lk["lds_bank_conflict_frac"]
            if lk.get("mfma_busy_frac"):
                entry["mfma_busy_frac"] = lk["mfma_busy_frac"]
        kernels.append(entry)
    # The step is two chains on two streams: the extraction of batch t+1 (high-priority stream; its kernels add up to the step time) and the
    # matcher of batch t, which runs in its shadow -- a matcher kernel's elapsed time includes waiting for compute units the extraction holds
    # (k_bf_replay: 0.06 ms per 256 pairs stand-alone, 8x that at B = 1024 in the pipeline, without costing the step anything).  The line's
    # `roofline` object is the kernel with the largest time per step ON THE CRITICAL STREAM; the longest kernel of the overlapped stream is
    # reported beside it (`overlapped_stream_longest`) and every kernel is in `kernels`.
    crit = [k for k in kernels if k["kernel"] in CRITICAL_STREAM] or kernels
    dk = max(crit, key=lambda k: k["mean_launch_ms"] * k["launches_per_step"])
    hb = dk.get("hbm_context")  # the top-level object keeps the byte roofline (contract: "hbm" | "mfma"); the issue bound rides along
    dom = {"kernel": dk["kernel"], "rocprof_kernel": dk["rocprof_kernel"], "bound": "hbm" if hb else dk["bound"], "achieved": hb["achieved"] if hb else dk["achieved"],
           "peak": hb["peak"] if hb else dk["peak"], "unit": hb["unit"] if hb else dk["unit"], "frac": hb["frac"] if hb else dk["frac"],
           "traffic": dk["traffic"], "valu_issue": dk["valu_issue"], "mean_launch_ms": dk["mean_launch_ms"],
           "selection": "largest time per step among the kernels of the critical (extraction) stream",
           "pmc_profiles_match_sources": traffic_json is not None}
    if hb:
        dom["valu_bound"] = {"frac": dk["frac"], "note": "this kernel is bound by VALU issue (kernels[].bound == \"valu\"); the byte fraction above says how little it needs memory"}
    over = [k for k in kernels if k["kernel"] not in CRITICAL_STREAM]
    if over:
        ok = max(over, key=lambda k: k["mean_launch_ms"] * k["launches_per_step"])
        dom["overlapped_stream_longest"] = {k: ok[k] for k in ("kernel", "rocprof_kernel", "bound", "unit", "peak", "achieved", "frac", "mean_launch_ms", "standalone", "mfma_busy_frac") if k in ok}
        mf = next((k for k in over if k["bound"] == "mfma"), None)
        if mf is not None:
            dom["matrix_core_kernel"] = {k: mf[k] for k in ("kernel", "rocprof_kernel", "bound", "unit", "peak", "achieved", "frac", "mean_launch_ms", "executed_int8_ops_per_launch", "mfma_busy_frac", "standalone") if k in mf}
    for k in ("algorithmic_bytes_per_launch", "executed_int8_ops_per_launch"):
        if k in dk:
            dom[k] = dk[k]
    return kernels, dom


def natural_sequence(B, width=W, height=H):
    """B frames of NATURAL image statistics: 640x480 crops of the reference's own two 1920x960 test images
    (tests/golden/equirect_00{1,2}_gray.png = test/data/equirectangular_image_00{1,2}.jpg as 8-bit luma).  B/16 tracks (8 crop
    origins per image, shifted a few pixels per round of 16 tracks) of 16 consecutive frames each; inside a track the crop moves by (3, 1) px per frame, so consecutive frames
    match like a translating camera.  The ring pair that closes a track (last frame against the next track's first) matches little,
    as a scene cut would."""
    from PIL import Image
    imgs = [np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", f"equirect_00{i}_gray.png")), dtype=np.uint8) for i in (1, 2)]
    per = 16                                       # frames per track; a batch of B frames is B / 16 tracks
    out = np.empty((B, height, width), np.uint8)
    origins = [(0, 120), (420, 60), (840, 200), (1230, 150), (100, 440), (520, 400), (900, 430), (1200, 380)]
    for t in range(B):
        k = t // per                               # track: crop origin k % 8 of image (k // 8) % 2, shifted by (5, 9) px per round of 16 tracks
        im = imgs[(k // 8) % 2]
        ox, oy = origins[k % 8]
        rnd, f = k // 16, t % per
        x0, y0 = min(ox + 5 * rnd + 3 * f, im.shape[1] - width), min(oy + 9 * rnd + f, im.shape[0] - height)
        out[t] = im[y0:y0 + height, x0:x0 + width]
    return out


def quick_test_pass_rate(img, thr):
    """Fraction of the interior pixels of one level-0 image that pass k_fast's 5-pixel quick test at threshold thr (the necessary
    condition for a 9-arc on the 16-ring: (p0 | p8) & (p4 | p12) brighter, or darker)."""
    v = img[3:-3, 3:-3].astype(np.int16)
    p0, p8 = img[:-6, 3:-3].astype(np.int16), img[6:, 3:-3].astype(np.int16)
    p4, p12 = img[3:-3, 6:].astype(np.int16), img[3:-3, :-6].astype(np.int16)
    br = np.minimum(np.maximum(p0, p8), np.maximum(p4, p12)) - v > thr
    dk = v - np.maximum(np.minimum(p0, p8), np.minimum(p4, p12)) > thr
    return float((br | dk).mean())


def bench_natural(ctx, L, B, want_cpu=True):
    """The same two-stream batch pipeline on natural imagery: per-kernel ms, keypoints / frame, quick-test pass
    rate, and the CPU port on the same frames."""
    from stella_vslam_amd import synthetic
    nat = natural_sequence(B)
    fe = run_front_end(ctx, L, nat, B, 10, 2, lambda: None, 1)
    kernels, dom = roofline_entries(fe, "-", B, 1)
    syn = synthetic.frame_sequence(4, W, H, seed=0x5EED)
    out = {"what": "the headline pipeline on 640x480 crops of the reference's two 1920x960 test images (tracks of 16 frames, crop moving (3,1) px per frame)",
           "frames_per_s": round(B * fe["steps"] / fe["dt"], 1), "ms_per_step": round(fe["dt"] / fe["steps"] * 1e3, 4),
           "frames_per_step": B, "keypoints_per_frame": round(fe["n_kp"], 1), "matches_per_pair": round(fe["n_match"], 1),
           "quick_test_pass_rate_level0": {"natural_thr20": round(float(np.mean([quick_test_pass_rate(nat[i], 20) for i in range(0, B, max(1, B // 16))])), 4),
                                           "natural_thr7": round(float(np.mean([quick_test_pass_rate(nat[i], 7) for i in range(0, B, max(1, B // 16))])), 4),
                                           "synthetic_thr20": round(float(np.mean([quick_test_pass_rate(f, 20) for f in syn])), 4),
                                           "synthetic_thr7": round(float(np.mean([quick_test_pass_rate(f, 7) for f in syn])), 4)},
           "dominant_kernel": dom["kernel"],
           "kernels": [{k: e[k] for k in ("kernel", "mean_launch_ms", "launches_per_step", "bound", "achieved", "frac", "unit")} for e in kernels]}
    if want_cpu:
        per = max(1, B // 16)
        out["cpu_port"] = cpu_front_end(np.concatenate([nat[0:min(8, per)], nat[9 * per:9 * per + min(8, per)]]), reps=1)
    return out


def load_profile_json(pattern, src_hash, B, world):
    """Latest profiles/<pattern> whose `csrc_hash` equals the hash of the sources being run (and whose batch / rank count match)."""
    import glob
    try:
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
            j = json.load(open(f))
            if j.get("csrc_hash") == src_hash and j.get("batch", 64) == B and world == 1:
                return j
    except Exception:
        pass
    return None


# Roofline that bounds each kernel class.  Everything streams bytes (HBM, 8 TB/s) except the brute-force distance kernel,
# which computes the reference's all-pairs 256-bit Hamming distances on the matrix cores: int8 MFMA, dense peak = 2 x the
# 2.5 PFLOP/s bf16 figure of MI355X_MICROARCH.md (its measured 32x32x32 i8 rate is 4.4 POP/s).
ROOFS = {"k_bf_topk": ("mfma", "TFLOP/s", I8_MFMA_PEAK_TOPS)}


def algorithmic_bytes(level_px, n_kp, B):
    """ALGORITHMIC bytes (k_bf_topk: integer operations) per step for each kernel class, for B frames."""
    pyr = sum(level_px[:-1]) + sum(level_px[1:])          # read L0..L6, write L1..L7
    fast = sum(level_px)                                    # every level read once
    blur = 2 * sum(level_px)                                # read + write every level
    desc = n_kp * (749 + 512 + 32 + 28)                     # IC-angle patch + BRIEF samples + descriptor + record
    select = n_kp * 16 + 8 * 2463
    bf_ops = 2 * 256 * n_kp * n_kp                          # all pairs x 256 bit positions x (multiply, add): the work of robust.cc:271-314
    sort = 2 * n_kp * (32 + 4 + 4)                          # ring mode: every frame is angle-sorted ONCE (read + write descriptors, angles, indices)
    return {"k_resize": pyr * B, "k_blur": blur * B, "k_fast": fast * B, "k_select": select * B, "k_describe": desc * B,
            "k_bf_binsort": sort * B, "k_bf_topk": bf_ops * B, "k_bf_replay": (n_kp * 16 * 4 + n_kp * 8) * B}


def bench_latency(ctx, frames_np):
    """What tracking_module does per frame: ONE frame through the host-buffer entry points (H2D of the image, D2H of keypoints and
    descriptors, then robust::brute_force_match against the previous frame, host in / host out)."""
    from stella_vslam_amd import feature, match
    ext = feature.orb_extractor(feature.orb_params(), ctx=feature.Context(ctx.device))
    m = match.robust(LOWE, bool(CHECK_ORI), ext.ctx)
    k0, d0 = ext.extract(frames_np[0])
    k1, d1 = ext.extract(frames_np[1])
    m.brute_force_match(d1, k1["angle"], d0, k0["angle"])
    reps = 30
    t0 = time.perf_counter()
    for i in range(reps):
        k1, d1 = ext.extract(frames_np[1 + (i & 1)])
    t1 = time.perf_counter()
    for i in range(reps):
        m.brute_force_match(d1, k1["angle"], d0, k0["angle"])
    t2 = time.perf_counter()
    e, b = (t1 - t0) / reps * 1e3, (t2 - t1) / reps * 1e3
    return {"what": "single 640x480 frame, host buffers in and out (PCIe included): svgpu_orb_extract, then svgpu_match_bruteforce vs the previous frame",
            "extract_ms": round(e, 4), "match_ms": round(b, 4), "frames_per_s": round(1e3 / (e + b), 1)}


def bench_tracked_frame(frames_np, want_cpu=True):
    """What tracking_module does per image (tracking_module.cc:533-608), through the C++ DROP-IN CLASSES of stella_vslam_amd/host/drop_in
    (libsvgpu_host.so, svgpu_host_tracked_frame): extract -> frame observation -> projection::match_current_and_last_frames ->
    pose_optimizer -> can_observe loop -> projection::match_frame_and_landmarks -> pose_optimizer, on a plane scene that makes the
    synthetic sequence geometrically consistent.  Run with device-resident frames (the extractor's output is adopted on the device and bound
    by every matcher: the frame's descriptors cross PCIe once, downwards) and without (every matcher call uploads its frame), next to the
    CPU oracle chain on the same images."""
    host = C.CDLL(os.path.join(ROOT, "stella_vslam_amd", "host", "libsvgpu_host.so"))
    seq = np.ascontiguousarray(frames_np[:4])
    out = {"what": "one tracked 640x480 frame through the drop-in classes: extract, frame observation, match_current_and_last_frames, pose optimizer, "
                   "can_observe loop, match_frame_and_landmarks, pose optimizer (host objects in and out, PCIe included)"}
    names = ("extract", "frame_observation", "match_current_and_last_frames", "pose_optimizer_1", "can_observe", "match_frame_and_landmarks", "pose_optimizer_2", "total")
    host.svgpu_host_tracked_frame_counters.restype = None
    for key, resident in (("chain", 2), ("resident_frames", 1), ("uploads_per_call", 0)):
        ms, cnt = np.zeros(8), np.zeros(8, np.int32)
        rc = host.svgpu_host_tracked_frame(C.c_void_p(seq.ctypes.data), len(seq), W, H, 30, resident, C.c_void_p(ms.ctypes.data), C.c_void_p(cnt.ctypes.data))
        if rc != 0:
            out[key] = {"error": "svgpu_host_tracked_frame failed"}
            continue
        out[key] = {"ms_per_frame": round(float(ms[7]), 4), "frames_per_s": round(1e3 / float(ms[7]), 1), "ms": {n: round(float(v), 4) for n, v in zip(names[:7], ms[:7])},
                    "keypoints": int(cnt[0]), "last_frame_landmarks": int(cnt[1]), "matches_1": int(cnt[2]), "inliers_1": int(cnt[3]),
                    "local_landmarks_visible": int(cnt[4]), "matches_2": int(cnt[5]), "inliers_2": int(cnt[6]), "translation_error_um": int(cnt[7])}
        if resident == 2:
            # the chain (svgpu_track_motion / svgpu_track_local_map behind drop_in/tracking_hip.h): landmark ids instead of flattened landmarks, the
            # resident landmark table, two submissions per frame.  Its two timed legs are the two halves.
            la, sy = C.c_double(0), C.c_double(0)
            host.svgpu_host_tracked_frame_counters(C.byref(la), C.byref(sy))
            out[key]["ms"] = {"motion_half (extract + frame observation + match_current_and_last_frames + pose optimizer)": round(float(ms[0]), 4),
                              "local_map_half (can_observe + match_frame_and_landmarks + pose optimizer)": round(float(ms[4]), 4)}
            out[key]["launches_per_frame"] = round(la.value, 2)   # kernel launches + runtime copies
            out[key]["host_syncs"] = round(sy.value, 2)           # stream synchronisations per frame
    if want_cpu:
        try:
            out["cpu_port"] = cpu_tracked_frame(seq)
        except Exception as e:
            out["cpu_port"] = {"error": repr(e)}
    # ---- the same chain for a STEREO frame (BASELINE configs[3] shape: 1241 x 376, ini_fast_threshold 12): svgpu_track_motion_stereo + svgpu_track_local_map
    try:
        from stella_vslam_amd import synthetic
        Wk, Hk, disp = 1241, 376, 17
        big = synthetic.frame_sequence(4, Wk + 64, Hk, seed=0x5EED + 4)
        sl, sr = np.ascontiguousarray(big[:, :, 8:8 + Wk]), np.ascontiguousarray(big[:, :, 8 + disp:8 + disp + Wk])
        ms, cnt = np.zeros(8), np.zeros(8, np.int32)
        rc = host.svgpu_host_tracked_frame_stereo(C.c_void_p(sl.ctypes.data), C.c_void_p(sr.ctypes.data), len(sl), Wk, Hk, C.c_double(disp), 12, 30, C.c_void_p(ms.ctypes.data),
                                                  C.c_void_p(cnt.ctypes.data))
        if rc != 0:
            out["chain_stereo"] = {"error": "svgpu_host_tracked_frame_stereo failed"}
        else:
            out["chain_stereo"] = {"what": "one tracked 1241x376 STEREO frame through tracked_frame_chain: left + right extraction, stereo::compute, frame observation, "
                                           "match_current_and_last_frames, pose optimizer | can_observe, match_frame_and_landmarks, pose optimizer (two submissions, PCIe included)",
                                   "ms_per_frame": round(float(ms[7]), 4), "frames_per_s": round(1e3 / float(ms[7]), 1),
                                   "ms": {"motion_half": round(float(ms[0]), 4), "local_map_half": round(float(ms[4]), 4)}, "keypoints": int(cnt[0]),
                                   "last_frame_landmarks": int(cnt[1]), "matches_1": int(cnt[2]), "inliers_1": int(cnt[3]), "keypoints_with_stereo_partner": int(cnt[4]),
                                   "matches_2": int(cnt[5]), "inliers_2": int(cnt[6]), "translation_error_um": int(cnt[7])}
        if want_cpu:
            out["cpu_port_stereo"] = cpu_tracked_frame(sl, seq_right=sr, disparity=disp, fx=718.856, ini_thr=12, margin1=10.0, reps=2)
    except Exception as e:
        out.setdefault("chain_stereo", {"error": repr(e)})
    return out


def cpu_tracked_frame(seq, reps=3, seq_right=None, disparity=0.0, fx=500.0, ini_thr=20, margin1=20.0):
    """The same chain with the oracle (C restatements of the reference's methods), pinned to one core: medians over `reps` frames.
    With seq_right: the stereo frame's (both extractions + match::stereo::compute in the `extract` leg, stereo gates / edges behind)."""
    from oracle import oracle as O
    n_frames, h, w = seq.shape
    fy = fx
    cx, cy, Z, sx, sy = 0.5 * w, 0.5 * h, 5.0, 3.0, 1.0
    stereo = seq_right is not None
    bl = disparity * Z / fx if stereo else 0.0
    fxb = fx * bl
    cam = O.make_camera(O.CAM_PERSPECTIVE, w, h, fx, fy, cx, cy, (0, 0, 0, 0, 0), fxb) if stereo else O.make_camera(O.CAM_PERSPECTIVE, w, h, fx, fy, cx, cy, (0, 0, 0, 0, 0))
    sf, _, lss, _ = O.scale_tables(1.2, 8)
    sf = np.asarray(sf, np.float32)
    inv_sigma = (1.0 / np.asarray(lss, np.float32)).astype(np.float32)
    lsf = float(np.log(np.float32(1.2)))

    def pose(t):
        return np.eye(3), np.array([-t * sx * Z / fx, -t * sy * Z / fy, 0.0])

    def backproject(t, xy):
        return np.stack([(xy[:, 0] - cx) / fx * Z + t * sx * Z / fx, (xy[:, 1] - cy) / fy * Z + t * sy * Z / fy, np.full(len(xy), Z)], 1)

    maps = []
    for t in range(n_frames - 1):
        k, d, _ = O.orb_extract(seq[t], ini_thr=ini_thr)
        xy = np.stack([k["x"], k["y"]], 1).astype(np.float64)
        pw = backproject(t, xy)
        c = -pose(t)[1]
        dist = np.linalg.norm(pw - c, axis=1)
        lvl = k["octave"].astype(np.int64)
        maps.append(dict(k=k, d=d, pw=pw, nrm=(pw - c) / dist[:, None], mx=(dist * sf[lvl]).astype(np.float32),
                         mn=((dist * sf[lvl]) / sf[7]).astype(np.float32)))
    last = maps[-1]
    loc = {key: np.concatenate([m[key] for m in maps[:-1]]) for key in ("pw", "nrm", "mx", "mn", "d")}
    Rl, tl = pose(n_frames - 2)
    Rg, tg = pose(n_frames - 1)
    tg = tg + np.array([0.004, -0.003, 0.002])
    K = np.array([fx, fy, cx, cy, fxb])
    huber = np.float32(np.sqrt(7.81473 if stereo else 5.991))
    old = _pin(2)
    tt = {n: [] for n in ("extract", "frame_observation", "match_current_and_last_frames", "pose_optimizer_1", "can_observe", "match_frame_and_landmarks", "pose_optimizer_2")}
    try:
        for rep in range(reps + 1):
            t0 = time.perf_counter()
            if stereo:
                k, d, _, pl = O.orb_extract(seq[-1], ini_thr=ini_thr, want_pyramid=True)
                kr, dr, _, pr = O.orb_extract(seq_right[-1], ini_thr=ini_thr, want_pyramid=True)
                xr_cur, _ = O.stereo_match(k, d, kr, dr, pl, pr, fxb, bl)
            else:
                k, d, _ = O.orb_extract(seq[-1])
                xr_cur = None
            t1 = time.perf_counter()
            xy = np.stack([k["x"], k["y"]], 1)
            und = O.undistort_keypoints(cam, xy)
            O.keypoints_to_bearings(cam, und)
            O.assign_keypoints_to_grid(und[:, 0], und[:, 1], (cam.min_x, cam.max_x, cam.min_y, cam.max_y))
            t2 = time.perf_counter()
            m1, n1 = O.match_current_and_last_frames(True, cam, Rg, tg, Rl, tl, last["pw"], np.ones(len(last["pw"]), np.uint8), last["d"], last["k"]["octave"],
                                                     last["k"]["angle"], sf, margin1, d, und, k["octave"], k["angle"], **(dict(t_xright=xr_cur, is_monocular=False, true_baseline=bl) if stereo else {}))
            t3 = time.perf_counter()
            sel = m1 >= 0
            kp = m1[sel]
            uvr = np.stack([und[kp, 0], und[kp, 1], xr_cur[kp] if stereo else np.full(len(kp), -1.0)], 1).astype(np.float32)
            nv1, p1, outl, _ = O.pose_optimize(np.hstack([Rg, tg[:, None]]).reshape(12), last["pw"][sel], uvr, inv_sigma[k["octave"][kp]], np.full(len(kp), huber), K)
            t4 = time.perf_counter()
            P = p1.reshape(3, 4)
            vis, rp, xr, lv = O.can_observe(cam, P[:, :3], P[:, 3], loc["pw"], loc["nrm"], loc["mn"], loc["mx"], 0.5, 8, lsf)
            t5 = time.perf_counter()
            occ = np.zeros(len(d), np.uint8)
            occ[kp[outl == 0]] = 1
            m2, n2 = O.match_frame_and_landmarks(cam, vis, rp, xr, lv, loc["d"], sf, 5.0, 0.8, d, und, k["octave"], occupied=occ, **(dict(t_xright=xr_cur) if stereo else {}))
            t6 = time.perf_counter()
            sel2 = m2 >= 0
            kp2 = np.concatenate([kp[outl == 0], m2[sel2]])
            pw2 = np.concatenate([last["pw"][sel][outl == 0], loc["pw"][sel2]])
            uvr2 = np.stack([und[kp2, 0], und[kp2, 1], xr_cur[kp2] if stereo else np.full(len(kp2), -1.0)], 1).astype(np.float32)
            nv2, p2, _, _ = O.pose_optimize(p1, pw2, uvr2, inv_sigma[k["octave"][kp2]], np.full(len(kp2), huber), K)
            t7 = time.perf_counter()
            if rep == 0:
                continue
            for name, dt in zip(tt, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6)):
                tt[name].append(dt * 1e3)
    finally:
        _unpin(old)
    med = {n: round(float(np.median(v)), 3) for n, v in tt.items()}
    total = sum(med.values())
    return {"ms_per_frame": round(total, 3), "frames_per_s": round(1e3 / total, 2), "ms": med, "cores": 1, "kind": "port", "keypoints": int(len(k)), "matches_1": int(n1),
            "inliers_1": int(nv1), "local_landmarks_visible": int(vis.sum()), "matches_2": int(n2), "inliers_2": int(nv2),
            "translation_error_um": int(round(float(np.abs(p2.reshape(3, 4)[:, 3] - pose(n_frames - 1)[1]).max()) * 1e6))}


def bench_stereo(device):
    """BASELINE config 4 shape (KITTI 00 stereo: 1241x376, ini_fast_threshold 12; fx 718.856, baseline 0.537 m): both images of 8 pairs
    extracted in device-resident batches on two contexts / streams, then match::stereo::compute for the 8 pairs in one launch."""
    import torch
    from stella_vslam_amd import feature, pipeline, synthetic
    Wk, Hk, P, disp = 1241, 376, 8, 17
    big = synthetic.frame_sequence(P, Wk + 64, Hk, seed=0x5EED + 4)
    prm = feature.orb_params(ini_fast_thr=12)
    el = pipeline.BatchExtractor(Wk, Hk, P, prm, device=device, priority=1)
    er = pipeline.BatchExtractor(Wk, Hk, P, prm, device=device, priority=1)
    el.upload(np.ascontiguousarray(big[:, :, 8:8 + Wk]))
    er.upload(np.ascontiguousarray(big[:, :, 8 + disp:8 + disp + Wk]))
    ev = torch.cuda.Event()
    out = None

    def one():
        nonlocal out
        el.extract()
        er.extract()
        ev.record(er.stream)
        el.stream.wait_event(ev)
        out = pipeline.stereo_batch(el, er, 718.856 * 0.537, 0.537, out=out)
    for _ in range(3):
        one()
    el.ctx.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        one()
    el.ctx.synchronize()
    dt = (time.perf_counter() - t0) / reps
    xr = out[0].cpu().numpy().reshape(P, el.cap)
    n_kp = float(el.counts.cpu().numpy().reshape(P, el.nc)[:, 0].mean())
    return {"what": "8 stereo pairs 1241x376 (ini_fast_threshold 12) per step: left + right ORB extraction on two contexts, stereo::compute for all pairs in one launch, resident in HBM",
            "pairs_per_s": round(P / dt, 1), "ms_per_step": round(dt * 1e3, 4), "keypoints_per_image": round(n_kp, 1),
            "stereo_matches_per_pair": round(float((xr >= 0).sum(1).mean()), 1)}


def ba_roofline(sc, iters, seconds, free_poses):
    """compulsory bytes per LM iteration = E x 29 + L x 48 + P_free x 112 (observations, landmark state + Hll/bl, pose blocks)."""
    E, Lm = len(sc["obs_pose"]), len(sc["points"])
    per_iter = E * 29 + Lm * 48 + free_poses * 112
    ach = per_iter * iters / seconds / 1e9
    return {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": round(ach, 3), "frac": round(ach / HBM_PEAK_GBS, 6),
            "algorithmic_bytes_per_iteration": int(per_iter),
            "note": "whole optimize() call incl. host set-up divided by its LM iterations; at these sizes an iteration is a chain of dependent "
                    "launches (latency-bound), not a byte stream"}


def bench_local_ba(ctx, rank=0, world=1):
    """Config 3.  With N > 1 ranks: INDEPENDENT windows, one per GPU (replicas) -- a 20-keyframe window cannot beat one GPU when sharded
    (its all-reduce costs more than its damping trial), and a SLAM system has one mapping thread per map anyway; the figure
    says how many maps' local BA a node serves."""
    from stella_vslam_amd import distributed, optimize, synthetic
    sc = synthetic.ba_scene(seed=1234 + rank)  # 20 KF / 10k landmarks / ~60k observations
    ba = optimize.local_bundle_adjuster(ctx=ctx)
    for _ in range(3):
        ba.optimize_flat(sc)  # warm-up
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    reps, iters, per_call = 20, 0, []
    t0 = time.perf_counter()
    for _ in range(reps):
        t1 = time.perf_counter()
        res = ba.optimize_flat(sc)
        per_call.append(time.perf_counter() - t1)
        iters += res["stats"]["iters_stage1"] + res["stats"]["iters_stage2"]
    dt = time.perf_counter() - t0
    if world > 1:
        dt = distributed.max_over_ranks(dt)
    out = {"metric": "local-BA LM iterations/s @20 KF / 10k landmarks / %d obs" % len(sc["obs_pose"]),
           "value": round(world * iters / dt, 2), "unit": "iters/s", "ms_per_call": round(dt / reps * 1e3, 3),
           "median_ms_per_call": round(float(np.median(per_call)) * 1e3, 3),
           "iters_per_call": iters / reps, "dtype": "f64", "lm_trials_per_call": res["stats"]["lm_trials"], "n_gpus": world,
           "sharding": "none" if world == 1 else "replicas: one independent config-3 window per GPU, no collective (value = sum over ranks)",
           "roofline": ba_roofline(sc, iters, dt, int((np.asarray(sc["pose_fixed"]) == 0).sum()))}
    return out


XGMI_LINK_GBS = 153.0  # MI355X_MICROARCH.md: per xGMI link and direction (7 links per GPU); a ring all-reduce moves 2 (N - 1) / N of the payload over one


def bench_global_ba(device, rank, world, large=False):
    """BASELINE config 5 (or, `large`, the same 500-keyframe loop with 1.6 M landmarks / 9.6 M observations: the size at which the sharded
    part of a trial outweighs its serial part).  One rank: svgpu_global_ba.  N ranks: every rank holds the same scene, takes the observations
    of the landmarks whose keyframe segment it owns (distributed.shard_by_keyframe_segment; SVGPU_BENCH_BA_SHARD=mod: l % N) and runs
    svgpu_global_ba_sharded over the library's own RCCL communicator."""
    import torch
    from stella_vslam_amd import distributed, feature, optimize, synthetic
    sc = synthetic.ba_scene_large(num_lm=1600000) if large else synthetic.ba_scene_large()   # 500 KF / 200k landmarks / 1.2M observations (seed 5005)
    ctx = feature.Context(device)
    ba = optimize.local_bundle_adjuster(ctx=ctx)
    by_segment = os.environ.get("SVGPU_BENCH_BA_SHARD", "segment") != "mod"
    partition = None
    if world > 1:
        import torch.distributed as dist
        use_lib_comm = dist.get_backend() == "nccl" and torch.cuda.device_count() >= world
        cb = None
        keep = None
        if use_lib_comm:
            distributed.init_comm(ctx)
        else:
            cb, keep = distributed.make_allreduce_callback()
        shard = (distributed.shard_by_keyframe_segment if by_segment else distributed.shard_by_landmark)(sc, rank, world)
        partition = shard.get("_partition")
        run = lambda: ba.optimize_global_flat_sharded(shard, rank, world, cb, num_iter=10)
    else:
        run = lambda: ba.optimize_global_flat(sc, num_iter=10)
    res = run()  # warm-up
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    # per-call times, MEDIAN reported (x reps below, so that every figure derived from `dt` keeps its meaning): one call in a few dozen
    # takes twice as long -- 15 instead of 7.5 ms, sporadically among the first calls of a process -- and a mean over three calls then
    # says 9.9 ms for a 7.5 ms call (round 5, tools: 40 calls, median 7.47, max 15.1)
    reps, iters, per_call = (3 if large else 7), 0, []
    for _ in range(reps):
        t0 = time.perf_counter()
        res = run()
        per_call.append(time.perf_counter() - t0)
        iters += res["stats"]["iters_stage1"]
    dt = float(np.median(per_call)) * reps
    if world > 1:
        dt = distributed.max_over_ranks(dt)
    free = int((np.asarray(sc["pose_fixed"]) == 0).sum())
    try:
        plan = ba.last_envelope_plan()
    except Exception:
        plan = None
    exchange = ba.last_exchange() if world > 1 else None
    kernels, projection = None, None
    E, Lm, P = len(sc["obs_pose"]), len(sc["points"]), len(sc["pose_cw"])
    if world == 1:
        # one more call with HIP events around every kernel class of the LM loop (svgpu_profile_select "*"): mean launch time and the
        # ALGORITHMIC bytes of a launch (what the formulation has to move once: observations, the W blocks at 144 B per edge, landmark
        # blocks, the pair list) against the 8 TB/s HBM roof; the envelope solve is a dependency chain over 1.5 MB (no byte figure)
        from stella_vslam_amd._lib import lib
        L = lib()
        L.svgpu_profile_select(ctx.handle, b"*")
        r2 = run()
        pairs = 0
        lm_deg = np.bincount(np.asarray(sc["obs_point"]), minlength=Lm).astype(np.int64)
        pairs = int((lm_deg * (lm_deg + 1) // 2).sum())
        alg = {"ba_linearize": E * (29 + 24 + 144) + Lm * (24 + 48 + 24) + P * 96,      # observations twice (landmark- and pose-major), W, Hll, bl, points, poses
               "ba_schur": E * 144 + Lm * 48 + pairs * 12,                               # W and Hll once, the (edge, edge, landmark) pair list
               "ba_update": E * 144 + Lm * (48 + 24 + 24 + 24) + P * 96,                 # W, Hll, bl, point in / out
               "ba_chi2": E * 25 + Lm * 24 + P * 96,                                     # observations, points, poses
               "ba_solve": None}
        kernels = []
        class_ms = {}
        for name, nbytes in alg.items():
            ms, n = C.c_double(), C.c_longlong()
            L.svgpu_profile_read_class(ctx.handle, name.encode(), C.byref(ms), C.byref(n))
            if n.value == 0:
                continue
            mean = ms.value / n.value
            class_ms[name] = ms.value
            ent = {"class": name, "launches_per_call": n.value, "mean_launch_ms": round(mean, 4)}
            if nbytes is not None:
                ach = nbytes / (mean * 1e-3) / 1e9
                ent.update({"bound": "hbm", "algorithmic_bytes_per_launch": int(nbytes), "achieved": round(ach, 1), "unit": "GB/s", "peak": HBM_PEAK_GBS,
                            "frac": round(ach / HBM_PEAK_GBS, 4)})
            else:
                ent.update({"bound": "latency", "note": "segmented block envelope Cholesky: jobs + separator system + backward substitution, a dependency chain of ~100 block columns"})
            kernels.append(ent)
        L.svgpu_profile_select(ctx.handle, None)
        # Projection for N ranks from THIS GPU's measurements (nothing here is measured at N > 1):
        #   set-up       MEASURED: rank 0's real 1/N shard of the partition through svgpu_global_ba_sharded with zero iterations and a null
        #                exchange (host staging, uploads, structure, first chi2, read-back of a shard are what a rank does before / after its loop)
        #   loop         the observation-proportional phases scale with the largest shard's share of the observations, the envelope solve and
        #                the loop's remaining launch gaps stay as measured at N = 1
        #   exchange     every collective priced as a ring all-reduce over one xGMI link + a fixed cost per call
        trials = max(int(r2["stats"]["lm_trials"]), 1)
        call_ms = dt / reps * 1e3
        shard_ms = sum(class_ms.get(k, 0.0) for k in ("ba_linearize", "ba_schur", "ba_update", "ba_chi2"))
        solve_ms = class_ms.get("ba_solve", 0.0)

        def timed_ms(fn, n=3):
            fn()
            t = time.perf_counter()
            for _ in range(n):
                fn()
            return (time.perf_counter() - t) / n * 1e3

        setup1_ms = timed_ms(lambda: ba.optimize_global_flat(sc, num_iter=0))
        gaps_ms = max(call_ms - setup1_ms - shard_ms - solve_ms, 0.0)
        null_cb = distributed.ALLREDUCE_FN(lambda user, buf, count, stream: 0)
        FIXED_US, PER_TRIAL_CALLS = 20.0, 5  # pose blocks + damping slots | system (separator blocks) | job contributions | solution | trial sums (round 6: the slots ride with the pose blocks)
        projection = {"assumptions": {"collective_fixed_us": FIXED_US, "xgmi_link_GBs": XGMI_LINK_GBS, "collectives_per_trial": PER_TRIAL_CALLS,
                                      "setup": "measured on this GPU: rank 0's shard, zero iterations, null exchange",
                                      "loop": "observation phases x largest shard share; envelope solve and launch gaps as at N = 1"},
                      "measured_N1_ms": {"call": round(call_ms, 2), "setup": round(setup1_ms, 2), "observation_phases": round(shard_ms, 2),
                                         "envelope_solve": round(solve_ms, 2), "loop_gaps": round(gaps_ms, 2)},
                      "by_ranks": {}}
        for n in (2, 4, 8):
            shard0 = distributed.shard_by_keyframe_segment(sc, 0, n)
            info = shard0["_partition"]
            lm_rank = sc["_kfseg"][0]
            share = float(np.bincount(lm_rank[np.asarray(sc["obs_point"])], minlength=n).max()) / max(E, 1)
            setup_n_ms = timed_ms(lambda: ba.optimize_global_flat_sharded(shard0, 0, n, null_cb, num_iter=0))
            seg_bytes = 288 * info["separator_blocks"] + 48 * info["separator_keyframes"] + 8 * info["job_exchange_doubles"] + 48 * info["free_keyframes"] + 32 + 8 * n
            lin_bytes = 42 * 8 * info["free_keyframes"]
            full_bytes = seg_bytes - 288 * info["separator_blocks"] - 48 * info["separator_keyframes"] + 288 * info["kept_blocks"] + 48 * info["free_keyframes"]
            setup_xch = 8 * (Lm + 1) + 8 * 4 * Lm  # by-landmark contract check + the points of the other ranks at the end
            ent = {"segmented": info["segmented"], "largest_shard_share_of_observations": round(share, 4), "separator_keyframes": info["separator_keyframes"],
                   "landmarks_on_separators": info["landmarks_on_separators"], "setup_ms_of_a_shard": round(setup_n_ms, 2)}
            ring = lambda nbytes: nbytes * 2.0 * (n - 1) / n / (XGMI_LINK_GBS * 1e9) * 1e3
            for label, per_trial in (("keyframe_segments", seg_bytes), ("l_mod_N", full_bytes)):
                wire_ms = trials * (ring(per_trial + lin_bytes) + PER_TRIAL_CALLS * FIXED_US * 1e-3) + ring(setup_xch) + 4 * FIXED_US * 1e-3
                t_ms = setup_n_ms + gaps_ms + solve_ms + shard_ms * share + wire_ms
                ent[label] = {"bytes_per_trial": int(per_trial), "bytes_per_linearisation": int(lin_bytes), "setup_exchange_bytes": int(setup_xch),
                              "exchange_ms_per_call": round(wire_ms, 3), "projected_ms_per_call": round(t_ms, 2), "projected_speedup": round(call_ms / t_ms, 2)}
            projection["by_ranks"][str(n)] = ent
    name = "global-BA LM iterations/s @500 KF / %s landmarks / %d obs" % ("1.6M" if large else "200k", len(sc["obs_pose"]))
    shard_note = "none"
    if world > 1:
        shard_note = ("observations by keyframe segment (a landmark follows the piece of the keyframe graph that owns its keyframes): per damping trial only the "
                      "separator blocks, what the jobs leave on the separators and the solution cross ranks over RCCL") if by_segment else \
                     ("observations by landmark (l % N): all-reduce of the kept Schur blocks per damping trial over RCCL; the factorisation of the reduced system "
                      "is distributed (every rank eliminates the envelope jobs it owns, separator contributions and solution exchanged)")
    return {"metric": name, "value": round(iters / dt, 2), "unit": "iters/s",
            "envelope_plan": plan, "kernels": kernels,
            "ms_per_call": round(dt / reps * 1e3, 2), "ms_per_call_all": [round(t * 1e3, 2) for t in per_call], "timing": "median of %d calls" % reps,
            "iters_per_call": iters / reps, "lm_trials_per_call": int(res["stats"]["lm_trials"]), "dtype": "f64", "n_gpus": world,
            "sharding": shard_note, "partition": partition, "exchange": exchange, "projection": projection,
            "linear_solver": "block envelope Cholesky of the reduced camera system (direct)" if res["stats"]["pcg_iterations"] == 0 else "block-Jacobi PCG",
            "pcg_iterations_per_call": res["stats"]["pcg_iterations"], "chi2_final": res["stats"]["chi2_final"],
            "roofline": ba_roofline(sc, iters, dt, free)}


def bench_mapping_keyframe(want_cpu=True):
    """The mapping thread's per-keyframe call, END TO END on an object graph (mapping_module.cc:206): optimize::local_bundle_adjuster_hip::optimize(
    map_db, keyfrm, &flag) on the config-3 scene as stand-in keyframe / landmark objects (stella_vslam_amd/host/mapping_keyframe.cpp): gather from
    the graph, flatten, device solve, write-back under the map mutex, flush of the resident landmark table -- per-phase host milliseconds.
    Beside it the REFERENCE's own optimize/local_bundle_adjuster_g2o.cc (compiled where it lies into oracle/_ref/libsvref_ba.so; g2o's optimize() is
    the oracle's LM) on the same map, one core."""
    from stella_vslam_amd import synthetic
    host = C.CDLL(os.path.join(ROOT, "stella_vslam_amd", "host", "libsvgpu_host.so"))
    sc = synthetic.ba_scene()
    P, Lm, E = len(sc["pose_cw"]), len(sc["points"]), len(sc["obs_pose"])
    pose = np.ascontiguousarray(sc["pose_cw"], np.float64)
    fixed = np.ascontiguousarray(sc["pose_fixed"], np.uint8)
    pts = np.ascontiguousarray(sc["points"], np.float64)
    op, ol = np.ascontiguousarray(sc["obs_pose"], np.int32), np.ascontiguousarray(sc["obs_point"], np.int32)
    uvr = np.ascontiguousarray(sc["obs_uvr"], np.float32)
    octave = np.clip(np.rint(np.log(1.0 / np.asarray(sc["obs_inv_sigma_sq"], np.float64)) / np.log(1.44)), 0, 7).astype(np.int32)
    intr = np.ascontiguousarray(sc["intr"][0], np.float64)
    _v = lambda a: C.c_void_p(a.ctypes.data)
    ms, st = np.zeros(7), np.zeros(8, np.int32)
    rc = host.svgpu_host_mapping_keyframe(P, Lm, E, _v(pose), _v(fixed), _v(pts), _v(op), _v(ol), _v(uvr), _v(octave), _v(intr), 0, 5, _v(ms), _v(st))
    if rc != 0:
        return {"error": "svgpu_host_mapping_keyframe failed"}
    names = ("gather", "flatten", "solve", "write_back", "flush_map")
    out = {"what": "local_bundle_adjuster_hip::optimize(map_db, keyfrm, &flag) on %d local + %d fixed keyframes / %d landmarks / %d observations of stand-in objects, mean of 5 calls on fresh maps"
                   % (st[0], st[1], st[2], st[3]),
           "ms_per_call": round(float(ms[5]), 3), "gpu_ms": round(float(ms[2]), 3), "host_share": round(float(1.0 - ms[2] / ms[5]), 3),
           "phase_ms": {n: round(float(v), 3) for n, v in zip(names, ms[:5])}, "iters_stage1": int(st[4]), "iters_stage2": int(st[5]), "observations_erased": int(st[6]),
           "object_graph_construction_ms_untimed": round(float(ms[6]), 1)}
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libsvref_ba.so")
    if want_cpu and os.path.exists(ref_so):
        try:
            ref = C.CDLL(ref_so)
            K = P
            idx, seen = np.zeros(E, np.int32), np.zeros(K, np.int64)
            for e in range(E):
                idx[e] = seen[op[e]]
                seen[op[e]] += 1
            free = np.flatnonzero(fixed == 0)
            curr = int(free[-1])
            covis = np.ascontiguousarray(free[:-1], np.int32)
            a = dict(kf_id=np.arange(K, dtype=np.uint32), kf_flags=np.zeros(K, np.uint8), lm_id=np.arange(Lm, dtype=np.uint32), lm_erased=np.zeros(Lm, np.uint8),
                     uv=np.ascontiguousarray(uvr[:, :2]), xr=np.ascontiguousarray(uvr[:, 2]))
            o = dict(counts=np.zeros(3, np.int32), pose_order=np.full(K, -1, np.int32), point_order=np.full(Lm, -1, np.int32), edge_order=np.full(2 * E, -1, np.int32),
                     kf_pose=np.zeros((K, 12)), lm_pos=np.zeros((Lm, 3)), n_erased=C.c_int(0), erased=np.zeros(2 * E + 2, np.int32), lm_cnt=np.zeros((Lm, 4), np.int32),
                     kf_set=np.zeros(K, np.int32), iters=np.zeros(2, np.int32), stop=np.zeros(1, np.uint8))
            old = _pin(2)
            try:
                ts = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    r = ref.svref_local_ba(0, 0, 752, 480, _v(intr), C.c_float(1.2), 8, K, _v(a["kf_id"]), _v(pose), _v(a["kf_flags"]), Lm, _v(a["lm_id"]), _v(pts),
                                           _v(a["lm_erased"]), E, _v(op), _v(ol), _v(idx), _v(a["uv"]), _v(a["xr"]), _v(octave), curr, len(covis), _v(covis), 0, 0, 5, 10, 0,
                                           _v(o["counts"]), _v(o["pose_order"]), _v(o["point_order"]), _v(o["edge_order"]), _v(o["kf_pose"]), _v(o["lm_pos"]),
                                           C.byref(o["n_erased"]), _v(o["erased"]), _v(o["lm_cnt"]), _v(o["kf_set"]), _v(o["iters"]), _v(o["stop"]))
                    ts.append(time.perf_counter() - t0)
                    assert r == 0
            finally:
                _unpin(old)
            out["cpu_ms"] = round(float(np.median(ts)) * 1e3, 1)
            out["cpu_reference"] = {"ms_per_call": out["cpu_ms"], "cores": 1, "kind": "reference", "iters": [int(o["iters"][0]), int(o["iters"][1])],
                                    "sample": "median of 3 calls of the reference's compiled local_bundle_adjuster_g2o::optimize (oracle/_ref/libsvref_ba.so: its gather, graph, schedule and "
                                              "write-back; g2o's optimize() played by the oracle's LM) on the same map, object construction of the fixture included, pinned to core 2"}
        except Exception as e:
            out["cpu_reference"] = {"error": repr(e)}
    return out


def _pin(core):
    """taskset -c <core> for this process; returns the previous affinity, or None where pinning is not available."""
    try:
        old = os.sched_getaffinity(0)
        if core in old:
            os.sched_setaffinity(0, {core})
            return old
    except (AttributeError, OSError):
        pass
    return None


def _unpin(old):
    if old is not None:
        try:
            os.sched_setaffinity(0, old)
        except OSError:
            pass


def cpu_front_end(seq, reps=2, warm=1):
    """The oracle's orb_extract + brute_force_match (vs the previous frame) over `seq`, pinned to one core: per-frame MEDIANS after warm-up."""
    from oracle import oracle as O
    old = _pin(2)
    try:
        for i in range(warm):
            O.orb_extract(seq[i % len(seq)])
        t_ext, t_bf, prev = [], [], None
        for r in range(reps):
            for img in seq:
                t0 = time.perf_counter()
                k, d, _ = O.orb_extract(img)
                t1 = time.perf_counter()
                if prev is not None:
                    O.brute_force_match(d, k["angle"], prev[1], prev[0]["angle"], None, LOWE, bool(CHECK_ORI))
                    t_bf.append(time.perf_counter() - t1)
                t_ext.append(t1 - t0)
                prev = (k, d)
    finally:
        _unpin(old)
    e, b = float(np.median(t_ext)), float(np.median(t_bf)) if t_bf else 0.0
    return {"value": round(1.0 / (e + b), 3), "unit": "frames/s", "cores": 1, "kind": "port", "pinned_to_core": 2 if old is not None else None,
            "extract_ms_median": round(e * 1e3, 2), "match_ms_median": round(b * 1e3, 2), "frames": len(t_ext),
            "keypoints_per_frame": round(float(len(k)), 1)}


def cpu_baseline(frames_np, want_ba=True):
    """Oracle (CPU restatement of the reference path), 1 thread pinned to one core (taskset -c 2), medians after warm-up, bounded samples."""
    from oracle import oracle as O
    fe = cpu_front_end(frames_np[:25], reps=2, warm=5)
    out = {"value": fe["value"], "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": f"{fe['frames']} frame passes (25 frames x 2 after 5 warm-ups) of the same synthetic 640x480 sequence, process pinned to core 2: oracle orb_extract "
                     f"(median {fe['extract_ms_median']} ms/frame) + brute_force_match vs previous frame (median {fe['match_ms_median']} ms/pair), "
                     "gcc -O3, no -march=native, no OpenMP"}
    # context figure: the same port on ALL host cores, frame-parallel (frames are independent; one process per core,
    # each on its own slice of the sequence) -- the reference's optional OpenMP pragmas parallelise inside a frame instead
    try:
        ncore = max(1, min(os.cpu_count() or 1, 16))
        code = ("import sys,time; sys.path.insert(0, %r); from oracle import oracle as O; from stella_vslam_amd import synthetic as S; import numpy as np;"
                "seq=S.frame_sequence(6,640,480,seed=0x5EED+int(sys.argv[1])); O.orb_extract(seq[0]); t0=time.perf_counter(); n=0; prev=None\n"
                "while time.perf_counter()-t0 < 6.0:\n"
                "    k,d,_=O.orb_extract(seq[n%%6])\n"
                "    if prev is not None: O.brute_force_match(d,k['angle'],prev[1],prev[0]['angle'],None,%r,%r)\n"
                "    prev=(k,d); n+=1\n"
                "print(n, time.perf_counter()-t0)") % (ROOT, LOWE, bool(CHECK_ORI))
        procs = [subprocess.Popen([sys.executable, "-c", code, str(i)], stdout=subprocess.PIPE, text=True) for i in range(ncore)]
        rates = []
        for pr in procs:
            o, _ = pr.communicate(timeout=120)
            nf, dt = o.split()
            rates.append(int(nf) / float(dt))
        out["all_host_cores"] = {"value": round(sum(rates), 2), "unit": "frames/s", "cores": ncore, "how": f"one single-threaded oracle process per core on {ncore} of the host's {os.cpu_count()} cores, 6 s each"}
    except Exception as e:
        out["all_host_cores"] = {"error": str(e)}
    if want_ba:
        old = _pin(2)
        try:
            from stella_vslam_amd import synthetic
            sc = synthetic.ba_scene()
            for _ in range(2):
                O.local_ba(sc)
            ts = []
            for _ in range(20):
                t0 = time.perf_counter()
                r = O.local_ba(sc)
                ts.append(time.perf_counter() - t0)
            dt = float(np.median(ts))
            out["local_ba"] = {"value": round((r["stats"][2] + r["stats"][3]) / dt, 3), "unit": "iters/s", "ms_per_call": round(dt * 1e3, 1), "cores": 1, "kind": "port",
                               "sample": "median of 20 calls after 2 warm-ups on the config-3 scene, pinned to core 2"}
            sg = synthetic.ba_scene_large()
            O.local_ba(sg, iters1=10, iters2=0)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                r = O.local_ba(sg, iters1=10, iters2=0)
                ts.append(time.perf_counter() - t0)
            dt = float(np.median(ts))
            out["global_ba"] = {"value": round(r["stats"][2] / dt, 3), "unit": "iters/s", "ms_per_call": round(dt * 1e3, 1), "cores": 1, "kind": "port",
                                "sample": "median of 3 calls after 1 warm-up on the config-5 scene (500 KF / 200k landmarks), pinned to core 2; the oracle factors the reduced system with an envelope Cholesky"}
        except Exception as e:
            out.setdefault("local_ba", {"error": str(e)})
        finally:
            _unpin(old)
    return out


if __name__ == "__main__":
    sys.exit(main())
