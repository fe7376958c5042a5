"""World-size-2 gloo test (CPU) of the landmark-sharding path: shard bounds, covisibility skyline, and the
three all-reduce exchanges of SURVEY 8e driven through the product's ctypes hook with host pointers."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(300)
def test_sharded_exchanges_sum_to_the_unsharded_blocks(tmp_path):
    import torch.multiprocessing as mp
    import _dist_worker
    world = 2
    port = _free_port()
    mp.spawn(_dist_worker.run, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(os.path.join(tmp_path, f"rank{r}.npy"), allow_pickle=True)[0] for r in range(world)]
    assert res[0]["lo"] == 0 and res[0]["hi"] == res[1]["lo"] and res[1]["hi"] == 13 * 9
    assert abs(res[0]["obs"] - res[1]["obs"]) <= 5  # balanced by observation count
    for r in res:
        assert r["seen"]
        assert r["err"] < 1e-13
        assert r["U"] < 1e-13 and r["gf"] < 1e-12
        assert r["S"] < 1e-12 and r["rhs"] < 1e-11
        assert r["pts_block"] < 1e-15  # point blocks are shard-local
    mc = res[0]["min_cv"]
    assert mc == res[1]["min_cv"] and all(0 <= m <= j for j, m in enumerate(mc))


def test_covisibility_matches_bruteforce():
    import surikatoko_amd as sa
    from surikatoko_amd.ba import covisibility
    sc = sa.generate_scene(sa.SceneSpec(n_frames=17, grid_nx=9, grid_ny=7, vis_window=4))
    mc = covisibility(sc)
    brute = list(range(sc.M))
    for i in range(sc.N):
        fr = sc.obs_frame[sc.row_ptr[i]:sc.row_ptr[i + 1]]
        for j in fr:
            brute[j] = min(brute[j], int(fr[0]))
    assert mc.tolist() == brute
