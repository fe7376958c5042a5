"""The landmark-sharded LM loop end to end with 2 and 3 ranks (processes) on ONE GPU: gloo carries the exchanges (the
hook stages device buffers through host memory), everything else is the product path the RCCL runs use -- shard,
upload, global covisibility / nested plan, one exchange of the assembled system per attempt, common accept / reject
decisions.  Ranks must agree bit for bit on the cameras and match the single-process run."""
import os
import socket
import sys

import numpy as np
import pytest

import surikatoko_amd as sa

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# BASELINE config 4 = the headline scene (1000 cams / 100 000 pts / 2 000 000 obs) with its landmarks sharded: the same
# workload, shard sizes per rank of a 2- and a 4-GPU run, through Scene.shard + band exchange + nested plan
C3 = dict(n_frames=1000, grid_nx=400, grid_ny=250, vis_window=20)
# BASELINE config 5 (4000 cams / 1M pts / 20M obs, the HBM-bound stress configuration of the 8-GPU line) through the sharded
# path: shard sizes of a 2- and a 4-GPU run, 32-chunk plan, 400 MB band per damping factor, three reduced camera systems of
# 12.8 GB per rank (all ranks on ONE GPU here: 4 x 3 x 12.8 GB of systems alone); two iterations
C5 = dict(n_frames=4000, grid_nx=1000, grid_ny=1000, vis_window=20, _iters=2)


# schedule "dp" (default at world >= 2): damping-parallel -- every rank builds the round's two or three damping factors on its
# shard, band k is reduced to rank k, rank k solves and broadcasts its corrections, all ranks score every factor;
# "allreduce": the round-2 schedule (band all-reduced, every rank solves redundantly, speculative pairs)
@pytest.mark.timeout(1100)
@pytest.mark.parametrize("world,spec_kwargs,min_chunks,schedule", [
    (2, dict(n_frames=30, grid_nx=23, grid_ny=17, vis_window=7), 0, "dp"),                       # single skyline chain
    (2, dict(n_frames=30, grid_nx=23, grid_ny=17, vis_window=7), 0, "allreduce"),
    (3, dict(n_frames=400, grid_nx=60, grid_ny=40, vis_window=10, noise_uv_pix=0.2), 2, "dp"),   # nested plan, three factors a round
    (3, dict(n_frames=400, grid_nx=60, grid_ny=40, vis_window=10, noise_uv_pix=0.2), 2, "allreduce"),
    (2, C3, 8, "dp"),
    (4, C3, 8, "dp"),
    (2, C5, 16, "dp"),
    (4, C5, 16, "dp"),
    # frame numbers shuffled: every rank is given the numbering found on the whole scene (srk_ba_set_frame_order)
    (2, dict(n_frames=400, grid_nx=60, grid_ny=40, vis_window=10, noise_uv_pix=0.2, _shuffle=3), 2, "dp"),
    # ragged tracks (20 % of the observations dropped): the shards' runs are unions of frame lists (masked Schur / derivative kernels)
    (3, dict(n_frames=400, grid_nx=80, grid_ny=50, vis_window=16, noise_uv_pix=0.2, _drop=0.2), 2, "dp"),
], ids=["w2_30cam", "w2_30cam_allreduce", "w3_400cam", "w3_400cam_allreduce", "w2_C3_1kcam_100kpt", "w4_C3_1kcam_100kpt",
        "w2_C5_4kcam_1Mpt", "w4_C5_4kcam_1Mpt", "w2_400cam_frames_shuffled", "w3_400cam_ragged"])
def test_sharded_run_matches_single_process(tmp_path, world, spec_kwargs, min_chunks, schedule):
    import torch.multiprocessing as mp
    import _dist_gpu_worker
    kw = dict(spec_kwargs)
    iters = kw.pop("_iters", 3)  # far from convergence: no accept / reject decision is a near tie that summation order could flip
    shuffle = kw.pop("_shuffle", None)
    drop = kw.pop("_drop", None)
    spec = sa.SceneSpec(**kw)
    ref = sa.generate_scene(spec)
    if drop is not None:
        ref = sa.drop_observations(ref, drop, seed=11)
    if shuffle is not None:
        ref = sa.renumber_frames(ref, np.random.RandomState(shuffle).permutation(ref.M))
    ba = sa.BundleAdjustmentKanatani(0)
    crit = sa.BundleAdjustmentKanataniTermCriteria()
    crit.AllowedReprojErrRelativeChange(1e-7)
    ok_ref = ba.ComputeInplace(spec.f0, ref, crit, iters)
    rep = ba.report
    ref_rep = (rep.iterations, rep.attempts, rep.err_initial, rep.err_final, rep.seen, rep.status)
    ba.close()

    mp.spawn(_dist_gpu_worker.run, args=(world, _free_port(), str(tmp_path), spec_kwargs, iters, schedule), nprocs=world,
             join=True)
    res = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    assert res[0]["lo"] == 0 and res[-1]["hi"] == ref.N
    for r in range(world):
        z = res[r]
        assert bool(z["ok"]) == ok_ref
        got = (int(z["iterations"]), int(z["attempts"]), int(z["seen"]), int(z["status"]))
        assert got == (ref_rep[0], ref_rep[1], ref_rep[4], ref_rep[5]), (r, got, ref_rep, float(z["err_final"]))
        assert float(z["err_initial"]) == pytest.approx(ref_rep[2], rel=1e-12)
        assert float(z["err_final"]) == pytest.approx(ref_rep[3], rel=1e-8)
        # every rank applied the same corrections (dp: broadcast from the rank that solved; allreduce: every rank solved
        # the same all-reduced system with the same deterministic solver)
        assert np.array_equal(z["cam_R"], res[0]["cam_R"]) and np.array_equal(z["cam_T"], res[0]["cam_T"]), \
            (r, float(np.abs(z["cam_T"] - res[0]["cam_T"]).max()))
        dT, dR = float(np.abs(z["cam_T"] - ref.cam_T).max()), float(np.abs(z["cam_R"] - ref.cam_R).max())
        assert dT < 1e-7 and dR < 1e-7, (r, dT, dR)
        lo, hi = int(z["lo"]), int(z["hi"])
        assert np.abs(z["points"] - ref.points[lo:hi]).max() < 1e-7
        if r:
            assert lo == int(res[r - 1]["hi"])
    for r in range(world):
        assert int(res[r]["chunks"]) >= min_chunks, (r, int(res[r]["chunks"]))
