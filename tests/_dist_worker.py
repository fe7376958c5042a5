"""Worker of tests/test_dist_gloo.py: one process per rank, gloo backend, CPU only.

Exercises the product's multi-GPU glue (Scene.shard, covisibility, the ctypes all-reduce hook of
surikatoko_amd/dist.py) with HOST pointers; the per-shard numbers come from the CPU oracle (allowed in tests)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run(rank, world, port, out_dir):
    import torch.distributed as dist
    import surikatoko_amd as sa
    from surikatoko_amd.dist import make_allreduce_hook
    from oracle import oracle as orc

    # file rendezvous inside the test's own directory: no TCP port to race for (`port` is kept for the signature)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world,
                            init_method="file://" + os.path.join(out_dir, "rendezvous"))
    try:
        spec = sa.SceneSpec(n_frames=14, grid_nx=13, grid_ny=9, vis_window=5)
        full = sa.generate_scene(spec)
        ok, _ = sa.normalize_scene_inplace(full)
        assert ok
        shard, (lo, hi) = full.shard(rank, world)
        hook = make_allreduce_hook(None, None)  # host pointers

        def allreduce(a):
            a = np.ascontiguousarray(a, dtype=np.float64)
            rc = hook(None, a.ctypes.data_as(C.c_void_p).value, a.size)
            assert rc == 0
            return a

        def o(sc):
            return orc.Scene(sc.points, sc.cam_R, sc.cam_T, sc.K, sc.shared_k, sc.row_ptr, sc.obs_frame, sc.obs_uv)

        f0, c = spec.f0, 1e-3
        so, sf = o(shard), o(full)
        # exchange 3: error scalar and observation count
        e_loc, seen_loc = orc.reproj_error(f0, so)
        e_sum = allreduce(np.array([e_loc, float(seen_loc)]))
        e_full, seen_full = orc.reproj_error(f0, sf)
        # exchange 2: frame blocks + frame gradients
        g_l, V_l, U_l, W_l = orc.derivatives(f0, so)
        U_sum = allreduce(U_l.copy())
        gf_sum = allreduce(g_l[3 * shard.N:].copy())
        g_f, V_f, U_f, W_f = orc.derivatives(f0, sf)
        # exchange 1: reduced camera system + rhs are linear in the shard's blocks
        _, _, S_l, rhs_l = orc.two_phase(so, g_l, V_l, U_l, W_l, c, want_system=True)
        S_sum = allreduce(S_l.copy())
        rhs_sum = allreduce(rhs_l.copy())
        _, corr_f, S_f, rhs_f = orc.two_phase(sf, g_f, V_f, U_f, W_f, c, want_system=True)

        def rel(a, b):
            return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))

        res = dict(rank=rank, lo=lo, hi=hi, obs=shard.O,
                   err=rel(e_sum[0:1], np.array([e_full])), seen=int(e_sum[1]) == seen_full,
                   U=rel(U_sum, U_f), gf=rel(gf_sum, g_f[3 * full.N:]), S=rel(S_sum, S_f), rhs=rel(rhs_sum, rhs_f),
                   pts_block=rel(V_l, V_f[lo:hi]), min_cv=sa.ba.covisibility(full).tolist())
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.array([res], dtype=object), allow_pickle=True)
    finally:
        dist.destroy_process_group()
