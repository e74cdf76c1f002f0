#!/usr/bin/env python3
"""bench.py -- whole-job throughput of the hot path on N MI355X GPUs of one node.

Metric (BASELINE.json): frames/s of ORB extract + match at 640x480, ~2k keypoints per frame.
One "step" = one pass of the front end over a batch of B synthetic frames resident in HBM:
  svgpu_orb_extract_batch_device  (pyramid, blur, per-cell FAST, grid selection, orientation, rBRIEF)
  svgpu_match_consecutive_batch_device  (frame t+1 against frame t in a ring: robust::brute_force_match with the
                                        reference's robust_match_based_track settings 0.8 / orientation check)
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
    """sha256 over the kernel / host sources of libsvgpu: PMC-derived figures are only reported for the sources they were measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "stella_vslam_amd", "csrc")
    for name in sorted(os.listdir(d)):
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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="frames per GPU per step (the latency-bound matcher kernels amortise over a larger batch: 64 -> 256 is +7 %)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ba", action="store_true")
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
    params = feature.orb_params()
    NL = params.num_levels_
    ctx.check(L.svgpu_orb_configure(ctx.handle, W, H, B, C.c_float(params.scale_factor_), NL, params.ini_fast_thr_,
                                    params.min_fast_thr_, C.c_uint(800)), "svgpu_orb_configure")
    cap = L.svgpu_orb_max_keypoints(ctx.handle)
    level_px = []
    for l in range(NL):
        w_, h_ = C.c_int(), C.c_int()
        L.svgpu_orb_level_size(ctx.handle, l, C.byref(w_), C.byref(h_))
        level_px.append(w_.value * h_.value)

    # ---- synthetic input, resident in HBM before the timed region
    frames_np = synthetic.frame_sequence(B, W, H, seed=0x5EED + 7919 * rank)
    # Two HIP streams: extraction of step t+1 (stream A = the context's) overlaps the matcher of step t (stream B), which
    # leaves most CUs idle during its sort / greedy-replay kernels.  Two output buffer sets alternate; events order
    # extract(t) -> match(t) and match(t) -> extract(t+2) (the next writer of that set).
    stream = torch.cuda.ExternalStream(ctx.stream)
    stream_b = torch.cuda.Stream()
    NBUF = 2
    with torch.cuda.stream(stream):
        frames = torch.from_numpy(frames_np).cuda()
        bufs = [dict(kps=torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda"),
                     desc=torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda"),
                     counts=torch.zeros(B * (1 + NL), dtype=torch.int32, device="cuda"),
                     matched=torch.zeros(B * cap, dtype=torch.int32, device="cuda"),
                     nmatch=torch.zeros(B, dtype=torch.int32, device="cuda"),
                     ev_ext=torch.cuda.Event(), ev_match=torch.cuda.Event(), used=False) for _ in range(NBUF)]
    stream.synchronize()
    nc = 1 + NL
    state = {"i": 0}

    def match(bf):
        kps, desc, counts = bf["kps"], bf["desc"], bf["counts"]
        stream_b.wait_event(bf["ev_ext"])
        # pair t = (frame (t + 1) % B, keyframe = frame t), t = 0..B-1, straight from the extractor's batch layout
        ctx.check(L.svgpu_match_consecutive_batch_device(
            ctx.handle, B, C.c_void_p(desc.data_ptr()), C.c_void_p(kps.data_ptr()), C.c_void_p(counts.data_ptr()), cap, nc, None,
            C.c_float(LOWE), CHECK_ORI, C.c_void_p(bf["matched"].data_ptr()), C.c_void_p(bf["nmatch"].data_ptr()),
            C.c_void_p(stream_b.cuda_stream)), "match_batch")
        bf["ev_match"].record(stream_b)

    def step():
        """Extraction of batch t on stream A, then its matcher on stream B: the matcher of batch t overlaps the extraction of batch
        t+1.  (Measured alternative, rejected: holding the matcher back until the next batch's pyramid kernel -- whose LDS-resident
        level bands exclude the matcher's distance kernel from a CU -- has finished: 148 k instead of 157 k frames/s; the
        latency-bound pyramid is exactly where the matcher's kernels fit best.)"""
        bf = bufs[state["i"] % NBUF]
        state["i"] += 1
        kps, desc, counts = bf["kps"], bf["desc"], bf["counts"]
        if bf["used"]:
            stream.wait_event(bf["ev_match"])  # the matcher of two steps ago has finished reading this buffer set
        bf["used"] = True
        ctx.check(L.svgpu_orb_extract_batch_device(ctx.handle, C.c_void_p(frames.data_ptr()), B, C.c_size_t(W * H), W, None,
                                                   C.c_size_t(0), 0, C.c_void_p(kps.data_ptr()), C.c_void_p(desc.data_ptr()),
                                                   cap, C.c_void_p(counts.data_ptr()), None), "extract_batch")
        bf["ev_ext"].record(stream)
        match(bf)

    def sync_all():
        ctx.synchronize()
        torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()

    # ---- warm-up (untimed) + per-kernel pre-pass to find the dominant kernel
    for _ in range(max(args.warmup, 1)):
        step()
    sync_all()
    n_kp = bufs[0]["counts"].view(B, nc)[:, 0].float().mean().item()
    n_match = bufs[0]["nmatch"].float().mean().item()
    alg = algorithmic_bytes(level_px, n_kp, B)
    per_kernel = {}
    for name in alg:
        L.svgpu_profile_select(ctx.handle, name.encode())
        step()
        ms, n = C.c_double(), C.c_longlong()
        L.svgpu_profile_read(ctx.handle, C.byref(ms), C.byref(n))
        per_kernel[name] = (ms.value, n.value)
    dominant = max(per_kernel, key=lambda k: per_kernel[k][0])
    L.svgpu_profile_select(ctx.handle, dominant.encode())

    # ---- timed region: exactly K steps between barrier + synchronize
    sync_all()
    barrier()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    barrier()
    dt = time.perf_counter() - t0
    from stella_vslam_amd.distributed import max_over_ranks
    dt = max_over_ranks(dt)  # MAX over ranks (RCCL when world > 1)
    ms, n = C.c_double(), C.c_longlong()
    L.svgpu_profile_read(ctx.handle, C.byref(ms), C.byref(n))
    L.svgpu_profile_select(ctx.handle, None)
    k_ms = ms.value / max(n.value, 1)  # mean duration of one launch of the dominant kernel
    launches_per_step = max(n.value, 1) / args.steps
    bytes_per_launch = alg[dominant] / launches_per_step   # bytes, or integer operations for the MFMA-bound kernel
    bound, unit, peak = ROOFS.get(dominant, ("hbm", "GB/s", HBM_PEAK_GBS))
    achieved = bytes_per_launch / (k_ms * 1e-3) / (1e9 if bound == "hbm" else 1e12)

    # PMC-derived figures (rocprofv3 passes of this same command, tools/pmc_traffic.py / tools/pmc_valu.py) are valid only for the
    # kernel sources they were taken on: every profiles/*_traffic.json / *_valu_issue.json carries the csrc hash of its run
    src_hash = csrc_hash()
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
    for name, (kms, kn) in per_kernel.items():
        if kn == 0:
            continue
        b_, u_, p_ = ROOFS.get(name, ("hbm", "GB/s", HBM_PEAK_GBS))
        per_launch = alg[name] / kn
        ach = per_launch / (kms / kn * 1e-3) / (1e9 if b_ == "hbm" else 1e12)
        entry = {"kernel": name, "bound": b_, "unit": u_, "peak": p_, "achieved": round(ach, 2), "frac": round(ach / p_, 5),
                 "mean_launch_ms": round(kms / kn, 5), "launches_per_step": kn,
                 ("algorithmic_bytes_per_launch" if b_ == "hbm" else "algorithmic_ops_per_launch"): int(per_launch),
                 "traffic": traffic_of(name)}
        if lds_json is not None and name in lds_json["kernels"]:
            lk = lds_json["kernels"][name]
            entry["lds_util"], entry["lds_bank_conflict_frac"] = lk["lds_util"], lk["lds_bank_conflict_frac"]
            if lk.get("mfma_busy_frac"):
                entry["mfma_busy_frac"] = lk["mfma_busy_frac"]
        if name == "k_bf_topk":
            # the distance kernel multiplies only the candidate pairs inside the +-30 degree angle windows of robust.cc:279
            entry["note"] = ("algorithmic ops = all N1 x N2 pairs x 256 bit positions x 2 (the reference's work); the kernel multiplies only the "
                             "pairs inside the orientation windows (~22-28 % of them), so the matrix pipe's own utilisation is ~ frac x 0.25")
        kernels.append(entry)

    result = {
        "metric": "frames/s ORB-extract+match @640x480,2k kpts",
        "value": round(B * world * args.steps / dt, 2),
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": "synthetic 640x480 frame sequence (BASELINE configs[1] proxy: EuRoC imagery unavailable "
                               "offline), ORB extract + brute-force match vs previous frame",
                   "frames_per_gpu_per_step": B, "keypoints_per_frame": round(n_kp, 1),
                   "matches_per_pair": round(n_match, 1), "parallelism": f"frames sharded x{world}, no collective; extraction and matcher on two HIP streams"},
        "roofline": {"kernel": dominant, "bound": bound, "achieved": round(achieved, 2), "peak": peak, "unit": unit,
                     "frac": round(achieved / peak, 5), "traffic": traffic_of(dominant), "valu_issue": valu_of(dominant),
                     ("algorithmic_bytes_per_launch" if bound == "hbm" else "algorithmic_ops_per_launch"): int(bytes_per_launch),
                     "mean_launch_ms": round(k_ms, 5), "csrc_hash": src_hash,
                     "pmc_profiles_match_sources": traffic_json is not None,
                     "per_kernel_ms_per_step": {k: round(v[0], 4) for k, v in per_kernel.items()},
                     "kernels": kernels},
    }

    # free the front-end buffers before the other legs
    del bufs, frames
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
            print(json.dumps(result), flush=True)
        os._exit(0)

    watchdog = threading.Timer(args.leg_timeout, bail)
    watchdog.daemon = True
    watchdog.start()

    if rank == 0 and world == 1 and not args.no_extra:
        for key, fn in (("latency", lambda: bench_latency(ctx, frames_np)), ("stereo", lambda: bench_stereo(local_rank))):
            try:
                result[key] = fn()
            except Exception as e:  # a secondary leg must never hide the headline number
                result[key] = {"error": repr(e)}
    if not args.no_ba:
        if rank == 0 and world == 1:
            try:
                result["local_ba"] = bench_local_ba(ctx)
            except Exception as e:
                result["local_ba"] = {"error": repr(e)}
        try:
            gb = bench_global_ba(local_rank, rank, world)
            if rank == 0:
                result["global_ba"] = gb
        except Exception as e:
            if rank == 0:
                result["global_ba"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(frames_np, want_ba=not args.no_ba)
    legs_done.set()
    watchdog.cancel()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        closer = threading.Timer(60, lambda: os._exit(0))  # the line is out: a stuck teardown must not keep the launcher waiting
        closer.daemon = True
        closer.start()
        dist.barrier()
        dist.destroy_process_group()
    return 0


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


def bench_local_ba(ctx):
    from stella_vslam_amd import optimize, synthetic
    sc = synthetic.ba_scene()  # 20 KF / 10k landmarks / ~60k observations, seed 1234
    ba = optimize.local_bundle_adjuster(ctx=ctx)
    ba.optimize_flat(sc)  # warm-up
    t0 = time.perf_counter()
    reps, iters = 5, 0
    for _ in range(reps):
        res = ba.optimize_flat(sc)
        iters += res["stats"]["iters_stage1"] + res["stats"]["iters_stage2"]
    dt = time.perf_counter() - t0
    return {"metric": "local-BA LM iterations/s @20 KF / 10k landmarks / %d obs" % len(sc["obs_pose"]),
            "value": round(iters / dt, 2), "unit": "iters/s", "ms_per_call": round(dt / reps * 1e3, 3),
            "iters_per_call": iters / reps, "dtype": "f64", "lm_trials_per_call": res["stats"]["lm_trials"],
            "roofline": ba_roofline(sc, iters, dt, int((np.asarray(sc["pose_fixed"]) == 0).sum()))}


def bench_global_ba(device, rank, world):
    """BASELINE config 5.  One rank: svgpu_global_ba.  N ranks: every rank holds the same scene, takes the observations of the landmarks
    l % N == rank and runs svgpu_global_ba_sharded over the library's own RCCL communicator."""
    import torch
    from stella_vslam_amd import distributed, feature, optimize, synthetic
    sc = synthetic.ba_scene_large()   # 500 KF / 200k landmarks / 1.2M observations (seed 5005)
    ctx = feature.Context(device)
    ba = optimize.local_bundle_adjuster(ctx=ctx)
    if world > 1:
        import torch.distributed as dist
        use_lib_comm = dist.get_backend() == "nccl" and torch.cuda.device_count() >= world
        cb = None
        keep = None
        if use_lib_comm:
            distributed.init_comm(ctx)
        else:
            cb, keep = distributed.make_allreduce_callback()
        shard = distributed.shard_by_landmark(sc, rank, world)
        run = lambda: ba.optimize_global_flat_sharded(shard, rank, world, cb, num_iter=10)
    else:
        run = lambda: ba.optimize_global_flat(sc, num_iter=10)
    res = run()  # warm-up
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    reps, iters = 3, 0
    t0 = time.perf_counter()
    for _ in range(reps):
        res = run()
        iters += res["stats"]["iters_stage1"]
    dt = time.perf_counter() - t0
    if world > 1:
        dt = distributed.max_over_ranks(dt)
    free = int((np.asarray(sc["pose_fixed"]) == 0).sum())
    return {"metric": "global-BA LM iterations/s @500 KF / 200k landmarks / %d obs" % len(sc["obs_pose"]), "value": round(iters / dt, 2), "unit": "iters/s",
            "ms_per_call": round(dt / reps * 1e3, 2), "iters_per_call": iters / reps, "dtype": "f64", "n_gpus": world,
            "sharding": "none" if world == 1 else "observations by landmark (l % N), all-reduce of the kept Schur blocks per damping trial over RCCL",
            "linear_solver": "block envelope Cholesky of the reduced camera system (direct)" if res["stats"]["pcg_iterations"] == 0 else "block-Jacobi PCG",
            "pcg_iterations_per_call": res["stats"]["pcg_iterations"], "chi2_final": res["stats"]["chi2_final"],
            "roofline": ba_roofline(sc, iters, dt, free)}


def cpu_baseline(frames_np, want_ba=True):
    """Oracle (CPU restatement of the reference path), 1 thread, bounded sample (~10-20 s)."""
    from oracle import oracle as O
    n = len(frames_np)
    budget, done, t_ext, t_bf = 12.0, 0, 0.0, 0.0
    prev = None
    t_start = time.perf_counter()
    while time.perf_counter() - t_start < budget and done < 4 * n:
        img = frames_np[done % n]
        t0 = time.perf_counter()
        k, d, _ = O.orb_extract(img)
        t1 = time.perf_counter()
        if prev is not None:
            O.brute_force_match(d, k["angle"], prev[1], prev[0]["angle"], None, LOWE, bool(CHECK_ORI))
        t2 = time.perf_counter()
        t_ext += t1 - t0
        t_bf += t2 - t1
        prev = (k, d)
        done += 1
    total = t_ext + t_bf
    out = {"value": round(done / total, 3), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": f"{done} frames of the same synthetic 640x480 sequence: oracle orb_extract "
                     f"({t_ext / done * 1e3:.1f} ms/frame) + brute_force_match vs previous frame ({t_bf / max(done - 1, 1) * 1e3:.1f} ms/pair), "
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
        try:
            from stella_vslam_amd import synthetic
            sc = synthetic.ba_scene()
            t0 = time.perf_counter()
            r = O.local_ba(sc)
            dt = time.perf_counter() - t0
            out["local_ba"] = {"value": round((r["stats"][2] + r["stats"][3]) / dt, 3), "unit": "iters/s", "ms_per_call": round(dt * 1e3, 1), "cores": 1, "kind": "port"}
            sg = synthetic.ba_scene_large()
            t0 = time.perf_counter()
            r = O.local_ba(sg, iters1=10, iters2=0)
            dt = time.perf_counter() - t0
            out["global_ba"] = {"value": round(r["stats"][2] / dt, 3), "unit": "iters/s", "ms_per_call": round(dt * 1e3, 1), "cores": 1, "kind": "port",
                                "sample": "one call on the config-5 scene (500 KF / 200k landmarks); the oracle factors the reduced system with an envelope Cholesky"}
        except Exception as e:
            out["local_ba"] = {"error": str(e)}
    return out


if __name__ == "__main__":
    sys.exit(main())
