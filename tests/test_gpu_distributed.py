"""Two PROCESSES (torch.distributed, gloo rendezvous on 127.0.0.1) sharing the one GPU of the test box run
svgpu_local_ba_sharded / svgpu_global_ba_sharded through the shipped binding (distributed.make_allreduce_callback): the
multi-rank control flow of the library -- unconditional collectives on the stream, decisions from the all-reduced control
block only -- is exercised end to end and compared with the single-rank solve.  (RCCL itself refuses two ranks on one
device; its code path is covered with world = 1 in tests/test_gpu_ba.py and by the driver's multi-GPU bench.)"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from stella_vslam_amd import distributed as D, feature, optimize, synthetic as S
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
cb, keep = D.make_allreduce_callback()
adj = optimize.local_bundle_adjuster(ctx=feature.Context(0))
out = {}
sc = S.ba_scene(num_kf=12, num_lm=2000, obs_per_lm=5, num_fixed=3, seed=21)
res = adj.optimize_flat_sharded(D.shard_by_landmark(sc, rank, world), rank, world, cb)
flag = np.zeros(1, np.uint8)
if rank == 1:
    flag[0] = 1   # only ONE rank's caller raises the stop flag: the vote must stop every rank at the same boundary
stopped = adj.optimize_flat_sharded(D.shard_by_landmark(sc, rank, world), rank, world, cb, force_stop_flag=flag)
sg = S.ba_scene(num_kf=40, num_lm=3000, obs_per_lm=6, num_fixed=1, seed=32, loop=True)
resg = adj.optimize_global_flat_sharded(D.shard_by_landmark(sg, rank, world), rank, world, cb, num_iter=10)
# north_star's partition: keyframe-segment shards of a system long enough for the segmented plan; only the separators cross the processes
os.environ["SVGPU_SKY_SEGMENTS"] = "5"
adjk = optimize.local_bundle_adjuster(ctx=feature.Context(0)).set_solver(optimize.SOLVER_ENVELOPE)
sk = S.ba_scene_large(num_kf=200, num_lm=24000, obs_per_lm=4)
resk = adjk.optimize_global_flat_sharded(D.shard_by_keyframe_segment(sk, rank, world), rank, world, cb, num_iter=10)
xk = adjk.last_exchange()
resm = adjk.optimize_global_flat_sharded(D.shard_by_landmark(sk, rank, world), rank, world, cb, num_iter=10)
xm = adjk.last_exchange()
if rank == 0:
    single = adj.optimize_flat(sc)
    singleg = adj.optimize_global_flat(sg, num_iter=10)
    out = dict(rc=res["rc"], rcg=resg["rc"], stopped_rc=stopped["rc"],
               dpose=float(np.abs(res["pose_cw"] - single["pose_cw"]).max()), dpts=float(np.abs(res["points"] - single["points"]).max()),
               iters=[res["stats"]["iters_stage1"], res["stats"]["iters_stage2"], single["stats"]["iters_stage1"], single["stats"]["iters_stage2"]],
               gated=[res["stats"]["num_gated"], single["stats"]["num_gated"]],
               dposeg=float(np.abs(resg["pose_cw"] - singleg["pose_cw"]).max()), itersg=[resg["stats"]["iters_stage1"], singleg["stats"]["iters_stage1"]],
               pcg=resg["stats"]["pcg_iterations"])
    singlek = adjk.optimize_global_flat(sk, num_iter=10)
    out.update(rck=resk["rc"], dposek=float(np.abs(resk["pose_cw"] - singlek["pose_cw"]).max()),
               dptsk=float(np.abs(resk["points"] - singlek["points"]).max() / np.abs(singlek["points"]).max()),
               trialsk=[resk["stats"]["lm_trials"], singlek["stats"]["lm_trials"]], modes=[xk["mode"], xm["mode"]],
               system_bytes=[xk["reduced_system_bytes"], xm["reduced_system_bytes"]])
allr = [None] * world
dist.all_gather_object(allr, [res["pose_cw"].tobytes(), resg["pose_cw"].tobytes(), stopped["rc"], int(flag[0]), resk["pose_cw"].tobytes() + resk["points"].tobytes()])
if rank == 0:
    out["identical"] = all(a[0] == allr[0][0] and a[1] == allr[0][1] and a[4] == allr[0][4] for a in allr)
    out["stopped"] = [a[2] for a in allr]
    out["flags"] = [a[3] for a in allr]
    json.dump(out, open(sys.argv[1], "w"))
dist.barrier()
dist.destroy_process_group()
"""


def test_two_process_sharded_ba_on_one_gpu(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT))
    outp = tmp_path / "out.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script), str(outp)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.load(open(outp))
    assert out["rc"] == 0 and out["rcg"] == 0
    assert out["identical"]                      # every rank returns the same bits
    assert out["iters"][0] == out["iters"][2] and out["iters"][1] == out["iters"][3] and out["gated"][0] == out["gated"][1]
    assert out["dpose"] < 1e-9 and out["dpts"] < 1e-9 and out["dposeg"] < 1e-7
    assert out["itersg"][0] == out["itersg"][1] and out["pcg"] == 0   # the global solve takes the envelope Cholesky on every rank
    assert out["stopped"] == [7, 7] and out["flags"] == [1, 1]   # SVGPU_STOPPED on both ranks, the flag propagated to rank 0's caller
    # keyframe-segment shards: recognised, equal to the single-rank solve, a fraction of the l % N exchange
    assert out["rck"] == 0 and out["modes"] == ["keyframe segments", "whole reduced system"], out
    assert out["dposek"] < 1e-7 and out["dptsk"] < 1e-7 and out["trialsk"][0] == out["trialsk"][1], out
    assert out["system_bytes"][0] * 4 < out["system_bytes"][1], out
