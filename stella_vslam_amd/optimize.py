"""Python mirror of stella_vslam::optimize::local_bundle_adjuster over the C ABI.

Reference interface: optimize/local_bundle_adjuster.h:15-24 -- optimize(map_db, curr_keyfrm, force_stop_flag);
created by local_bundle_adjuster_factory (backend "g2o" / "gtsam"; this one registers as "hip").  The object
graph gather (local_bundle_adjuster_g2o.cc:38-147) and the write-back (:352-430) are host-side steps of the C++
adaptor (stella_vslam_amd/host/drop_in/flat_optimizers.h); the flat problem they exchange is what
`optimize_flat` takes.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import lib
from .feature import Context


class _BaProblem(C.Structure):
    _fields_ = [("num_poses", C.c_int32), ("num_points", C.c_int32), ("num_obs", C.c_int32),
                ("pose_cw", C.c_void_p), ("pose_fixed", C.c_void_p), ("points", C.c_void_p), ("point_fixed", C.c_void_p),
                ("obs_pose", C.c_void_p), ("obs_point", C.c_void_p), ("obs_uvr", C.c_void_p),
                ("obs_inv_sigma_sq", C.c_void_p), ("obs_huber_delta", C.c_void_p), ("intrinsics", C.c_void_p),
                ("num_first_iter", C.c_int32), ("num_second_iter", C.c_int32), ("gain_threshold", C.c_double)]


class _BaStats(C.Structure):
    _fields_ = [("chi2_initial", C.c_double), ("chi2_final", C.c_double), ("iters_stage1", C.c_int32),
                ("iters_stage2", C.c_int32), ("stage2_entered", C.c_int32), ("num_gated", C.c_int32),
                ("lm_trials", C.c_int32), ("cholesky_failures", C.c_int32), ("lambda_final", C.c_double),
                ("stopped_by_terminate_action", C.c_int32), ("pcg_iterations", C.c_int32)]


def partition_keyframe_segments(scene: dict, world: int):
    """svgpu_ba_partition_keyframe_segments: landmark -> rank over the keyframe segments the `world`-rank solve will cut (host only).
    Returns (landmark_rank int32[L], info dict)."""
    a = lambda k, t: np.ascontiguousarray(scene[k], t)
    pose, pts = a("pose_cw", np.float64), a("points", np.float64)
    keep = [a("pose_fixed", np.uint8), a("obs_pose", np.int32), a("obs_point", np.int32)]
    pf = scene.get("point_fixed")
    pf = None if pf is None else np.ascontiguousarray(pf, np.uint8)
    ptr = lambda x: None if x is None else x.ctypes.data
    prob = _BaProblem(len(pose), len(pts), len(keep[1]), ptr(pose), ptr(keep[0]), ptr(pts), ptr(pf), ptr(keep[1]), ptr(keep[2]), None, None, None, None, 0, 0, 0.0)
    ranks, info = np.zeros(max(len(pts), 1), np.int32), np.zeros(12, np.int32)
    rc = lib().svgpu_ba_partition_keyframe_segments(C.byref(prob), int(world), C.c_void_p(ranks.ctypes.data), C.c_void_p(info.ctypes.data))
    if rc:
        raise RuntimeError(f"svgpu_ba_partition_keyframe_segments failed: {rc}")
    keys = ("segmented", "jobs", "cuts", "separator_keyframes", "landmarks_on_separators", "landmarks_on_separators_only", "free_keyframes", "kept_blocks", "separator_blocks", "job_exchange_doubles")
    return ranks[:len(pts)], {k: int(v) for k, v in zip(keys, info)}


def local_bundle_adjuster_factory_create(backend: str = "hip", **kw):
    """optimize/local_bundle_adjuster_factory.h:17-32 -- the backend switch."""
    if backend != "hip":
        raise RuntimeError(f"Invalid backend: {backend} (this build provides 'hip' only)")
    return local_bundle_adjuster(**kw)


SOLVER_AUTO, SOLVER_CHOLESKY, SOLVER_PCG, SOLVER_DENSE, SOLVER_PCG_MULTI, SOLVER_ENVELOPE, SOLVER_CHOLESKY_MFMA = 0, 1, 2, 3, 4, 6, 7  # svgpu_ba_solver


class local_bundle_adjuster:
    def __init__(self, num_first_iter: int = 5, num_second_iter: int = 10, ctx: Context | None = None):
        self.num_first_iter_ = num_first_iter
        self.num_second_iter_ = num_second_iter
        self.ctx = ctx or Context()

    def set_solver(self, solver: int = SOLVER_AUTO, pcg_tolerance: float = 1e-10, pcg_max_iterations: int = 0):
        """Linear solver of the reduced camera system (svgpu_ba_set_solver): on-chip LL^T / block-Jacobi PCG / dense LL^T in global memory / block envelope Cholesky."""
        self.ctx.check(lib().svgpu_ba_set_solver(self.ctx.handle, int(solver), C.c_double(pcg_tolerance), int(pcg_max_iterations)),
                       "svgpu_ba_set_solver")
        return self

    def last_envelope_plan(self) -> dict:
        """How the context's last envelope factorisation was planned (svgpu_ba_last_envelope_plan)."""
        info = np.zeros(8, np.int32)
        self.ctx.check(lib().svgpu_ba_last_envelope_plan(self.ctx.handle, C.c_void_p(info.ctypes.data)), "svgpu_ba_last_envelope_plan")
        return dict(kind=("one-sided", "two-sided", "segmented")[int(info[0])], rows=int(info[1]), widest_column=int(info[2]), cuts=int(info[3]),
                    jobs=int(info[4]), jobs_local=int(info[5]), separator_rows=int(info[6]), longest_job=int(info[7]))

    def last_exchange(self) -> dict:
        """What the context's last sharded solve all-reduced (svgpu_ba_last_exchange); bytes are payload per rank."""
        info = np.zeros(10, np.int64)
        self.ctx.check(lib().svgpu_ba_last_exchange(self.ctx.handle, C.c_void_p(info.ctypes.data)), "svgpu_ba_last_exchange")
        trials, lins = max(int(info[8]), 1), max(int(info[9]), 1)
        per_trial = (int(info[3]) + int(info[4]) + int(info[5]) + int(info[6])) / trials + int(info[2]) / lins
        return dict(mode=("none", "whole reduced system", "keyframe segments")[int(info[0])], setup_bytes=int(info[1]), pose_block_bytes=int(info[2]),
                    reduced_system_bytes=int(info[3]), separator_bytes=int(info[4]), solution_bytes=int(info[5]), sums_bytes=int(info[6]),
                    allreduce_calls=int(info[7]), trials=int(info[8]), linearisations=int(info[9]), bytes_per_trial=per_trial)

    def optimize_global_flat(self, scene: dict, num_iter: int = 10, force_stop_flag: np.ndarray | None = None, gain_threshold: float = 1e-3):
        """global_bundle_adjuster core: one LM run of `num_iter` iterations over the whole graph (no outlier stage)."""
        saved = self.num_first_iter_
        self.num_first_iter_ = num_iter
        try:
            return self.optimize_flat(scene, force_stop_flag, gain_threshold, _global=True)
        finally:
            self.num_first_iter_ = saved

    def optimize_flat_sharded(self, shard: dict, rank: int, world: int, allreduce_cb=None, force_stop_flag: np.ndarray | None = None,
                              gain_threshold: float = 1e-3):
        """Multi-GPU variant: `shard` holds this rank's observations (distributed.shard_by_landmark) and the complete
        pose / point arrays; `allreduce_cb` is a svgpu_allreduce_fn (distributed.make_allreduce_callback) or None = the
        context's own RCCL communicator (distributed.init_comm)."""
        return self.optimize_flat(shard, force_stop_flag, gain_threshold, _sharded=(rank, world, allreduce_cb))

    def optimize_global_flat_sharded(self, shard: dict, rank: int, world: int, allreduce_cb=None, num_iter: int = 10,
                                     force_stop_flag: np.ndarray | None = None, gain_threshold: float = 1e-3):
        """svgpu_global_ba_sharded: the single-stage global-BA run over landmark shards."""
        saved = self.num_first_iter_
        self.num_first_iter_ = num_iter
        try:
            return self.optimize_flat(shard, force_stop_flag, gain_threshold, _sharded=(rank, world, allreduce_cb), _global=True)
        finally:
            self.num_first_iter_ = saved

    def optimize_flat(self, scene: dict, force_stop_flag: np.ndarray | None = None, gain_threshold: float = 1e-3, _sharded=None,
                      _global=False):
        """scene keys: pose_cw (P,12) f64, pose_fixed (P) u8, points (L,3) f64, [point_fixed (L) u8], obs_pose / obs_point (E) i32,
        obs_uvr (E,3) f32, obs_inv_sigma_sq (E) f32, obs_huber (E) f32, intr (P,5) f64.
        force_stop_flag: None or a writable uint8[1] (the caller's abort flag; may be SET by the terminate rule)."""
        a = lambda k, t: np.ascontiguousarray(scene[k], t)
        pose, pts = a("pose_cw", np.float64), a("points", np.float64)
        keep = [pose, pts, a("pose_fixed", np.uint8), a("obs_pose", np.int32), a("obs_point", np.int32), a("obs_uvr", np.float32),
                a("obs_inv_sigma_sq", np.float32), a("obs_huber", np.float32), a("intr", np.float64)]
        pf = scene.get("point_fixed")
        pf = None if pf is None else np.ascontiguousarray(pf, np.uint8)
        P, L, E = len(pose), len(pts), len(keep[3])
        ptr = lambda x: None if x is None else x.ctypes.data
        prob = _BaProblem(P, L, E, ptr(pose), ptr(keep[2]), ptr(pts), ptr(pf), ptr(keep[3]), ptr(keep[4]), ptr(keep[5]),
                          ptr(keep[6]), ptr(keep[7]), ptr(keep[8]), self.num_first_iter_, self.num_second_iter_, gain_threshold)
        pose_out, pts_out = np.zeros_like(pose), np.zeros_like(pts)
        outl = np.zeros(1 if _global else max(E, 1), np.uint8)  # (the global adjuster has no outlier list: no 10 MB of flags to fault in and copy at 9.6 M observations)
        st = _BaStats()
        stop = None if force_stop_flag is None else C.c_void_p(force_stop_flag.ctypes.data)
        outs = (C.c_void_p(pose_out.ctypes.data), C.c_void_p(pts_out.ctypes.data), C.c_void_p(outl.ctypes.data), C.byref(st))
        if _global and _sharded is not None:
            rank, world, cb = _sharded
            rc = lib().svgpu_global_ba_sharded(self.ctx.handle, C.byref(prob), rank, world, cb, None, stop, outs[0], outs[1], outs[3])
        elif _global:
            rc = lib().svgpu_global_ba(self.ctx.handle, C.byref(prob), stop, outs[0], outs[1], outs[3])
        elif _sharded is None:
            rc = lib().svgpu_local_ba(self.ctx.handle, C.byref(prob), stop, *outs)
        else:
            rank, world, cb = _sharded
            rc = lib().svgpu_local_ba_sharded(self.ctx.handle, C.byref(prob), rank, world, cb, None, stop, *outs)
        self.ctx.check(rc, "svgpu_local_ba", ok=(0, 7))
        return dict(rc=rc, pose_cw=pose_out, points=pts_out, outlier=(np.zeros(0, np.uint8) if _global else outl[:E].copy()),
                    stats={f: getattr(st, f) for f, _ in _BaStats._fields_})


class pose_optimizer:
    """optimize/pose_optimizer.h (g2o backend defaults of pose_optimizer_factory.h:18-26): motion-only BA of one frame."""

    def __init__(self, num_trials_robust: int = 2, num_trials: int = 2, num_each_iter: int = 10, ctx: Context | None = None,
                 reset_stop_flag_each_round: bool = False):
        self.num_trials_robust_, self.num_trials_, self.num_each_iter_ = num_trials_robust, num_trials, num_each_iter
        self.reset_stop_flag_each_round_ = reset_stop_flag_each_round
        self.ctx = ctx or Context()

    def optimize_flat(self, pose_cw, pos_w, uvr, inv_sigma_sq, huber, intr):
        """Returns (num_valid_obs, optimized pose 3x4 flat, outlier_flags, lm_iterations)."""
        pose = np.ascontiguousarray(pose_cw, np.float64).reshape(12)
        pw, uv = np.ascontiguousarray(pos_w, np.float64), np.ascontiguousarray(uvr, np.float32)
        w, hb = np.ascontiguousarray(inv_sigma_sq, np.float32), np.ascontiguousarray(huber, np.float32)
        K = np.ascontiguousarray(intr, np.float64).reshape(5)
        n = len(pw)
        out, outl = np.zeros(12), np.zeros(max(n, 1), np.uint8)
        nv, it = C.c_int(0), C.c_int(0)
        p = lambda a: C.c_void_p(a.ctypes.data)
        self.ctx.check(lib().svgpu_pose_optimize(self.ctx.handle, p(pose), n, p(pw), p(uv), p(w), p(hb), p(K), self.num_trials_robust_,
                                                 self.num_trials_, self.num_each_iter_, int(self.reset_stop_flag_each_round_), p(out),
                                                 p(outl), C.byref(nv), C.byref(it)), "svgpu_pose_optimize")
        return nv.value, out, outl[:n].copy(), it.value

    def optimize_device(self, pose_cw, n: int, pos_w_dev, uvr_dev, inv_sigma_sq_dev, huber_dev, intr):
        """Same with the observation arrays already on the device (torch tensors or raw device addresses)."""
        pose = np.ascontiguousarray(pose_cw, np.float64).reshape(12)
        K = np.ascontiguousarray(intr, np.float64).reshape(5)
        out, outl = np.zeros(12), np.zeros(max(n, 1), np.uint8)
        nv, it = C.c_int(0), C.c_int(0)
        p = lambda a: C.c_void_p(a.ctypes.data)
        d = lambda t: C.c_void_p(t.data_ptr() if hasattr(t, "data_ptr") else int(t))
        self.ctx.check(lib().svgpu_pose_optimize_device(self.ctx.handle, p(pose), n, d(pos_w_dev), d(uvr_dev), d(inv_sigma_sq_dev), d(huber_dev), p(K),
                                                        self.num_trials_robust_, self.num_trials_, self.num_each_iter_,
                                                        int(self.reset_stop_flag_each_round_), p(out), p(outl), C.byref(nv), C.byref(it)),
                       "svgpu_pose_optimize_device")
        return nv.value, out, outl[:n].copy(), it.value
