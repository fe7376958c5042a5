"""Worker of tests/test_gpu_multirank.py: one process per rank, all on cuda:0, gloo backend (the hook stages the
device buffers through host memory).  Runs the product's sharded LM loop end to end and stores this rank's result."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run(rank, world, port, out_dir, spec_kwargs, iters, schedule="dp"):
    os.environ["SRK_DEBUG"] = "1"  # per-attempt trace on stderr: shows up in the pytest log when an assertion fails
    import torch.distributed as dist
    import surikatoko_amd as sa
    from surikatoko_amd.ba import covisibility, frame_order, revert_normalization
    from surikatoko_amd.dist import make_allreduce_hook

    # file rendezvous inside the test's own directory: no TCP port to race for (`port` is kept for the signature)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world,
                            init_method="file://" + os.path.join(out_dir, "rendezvous"))
    try:
        spec_kwargs = dict(spec_kwargs)
        spec_kwargs.pop("_iters", None)
        shuffle = spec_kwargs.pop("_shuffle", None)
        drop = spec_kwargs.pop("_drop", None)
        spec = sa.SceneSpec(**spec_kwargs)
        full = sa.generate_scene(spec)
        if drop is not None:  # ragged tracks: hardly two landmarks see the same frames
            full = sa.drop_observations(full, drop, seed=11)
        if shuffle is not None:  # an unordered image set: the frame numbers say nothing about covisibility
            full = sa.renumber_frames(full, np.random.RandomState(shuffle).permutation(full.M))
        ok, nrm = sa.normalize_scene_inplace(full)
        assert ok
        shard, (lo, hi) = full.shard(rank, world)
        ba = sa.BundleAdjustmentKanatani(0)
        ba.set_multi_schedule(schedule)  # "dp" (default) or "allreduce"; before the upload (it sizes the attempt slots)
        hook = make_allreduce_hook(None, "cuda:0")
        ba.set_allreduce(hook, rank, world)
        # the numbering is found on the WHOLE scene and given to every rank (a shard's own would differ from rank to rank)
        order = frame_order(full)
        assert (order is not None) == (shuffle is not None)
        ba.set_frame_order(order)
        assert ba.upload(spec.f0, shard, already_normalized=True)
        ba.set_covisibility(covisibility(full, order))
        crit = sa.BundleAdjustmentKanataniTermCriteria()
        crit.AllowedReprojErrRelativeChange(1e-7)
        ok = ba.optimize(crit, iters)
        print(f"rank {rank}: iterations {ba.report.iterations} attempts {ba.report.attempts} err {ba.report.err_initial!r} -> "
              f"{ba.report.err_final!r} status {ba.report.status}", flush=True)
        out = shard.copy()
        ba.download(out, revert_normalization=False)
        revert_normalization(out, nrm)
        r = ba.report
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ok=ok, lo=lo, hi=hi, points=out.points, cam_R=out.cam_R,
                 cam_T=out.cam_T, iterations=r.iterations, attempts=r.attempts, err_initial=r.err_initial,
                 err_final=r.err_final, seen=r.seen, chunks=ba.rcs_chunks(), status=r.status)
        ba.close()
    finally:
        dist.destroy_process_group()
