"""Internal frame order (srk_frame_order: reverse Cuthill-McKee on the covisibility graph, include/srk_ba.h).  CPU only: the
decision and the numbering are host code.  The reference has nothing to compare with -- its dense system
(bundle-adj-kanatani.cpp:1911) is indifferent to the numbering -- so these pin the properties the solver relies on."""
import ctypes as C

import numpy as np
import pytest

import surikatoko_amd as sa


def _order(scene, mode=-1):
    to_int = np.zeros(scene.M, np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = sa.lib().srk_frame_order(C.c_int(mode), C.c_int64(scene.N), C.c_int32(scene.M), p(scene.row_ptr), p(scene.obs_frame), p(to_int))
    assert rc in (0, 1)
    assert sorted(to_int.tolist()) == list(range(scene.M))
    return rc, to_int


def _bandwidth(scene, to_int=None):
    f = scene.obs_frame if to_int is None else to_int[scene.obs_frame]
    rp = scene.row_ptr[:-1][np.diff(scene.row_ptr) > 0]
    return int((np.maximum.reduceat(f, rp) - np.minimum.reduceat(f, rp)).max())


BAND = sa.SceneSpec(n_frames=120, grid_nx=30, grid_ny=20, vis_window=8)


def test_a_time_ordered_sequence_keeps_the_callers_numbering():
    sc = sa.generate_scene(BAND)
    rc, to_int = _order(sc)
    assert rc == 0 and np.array_equal(to_int, np.arange(sc.M))
    # the demos' all-visible scenes: nothing to gain either
    rc, _ = _order(sa.config_scene("demo_circle_grid"))
    assert rc == 0


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_a_shuffled_sequence_gets_its_band_back(seed):
    sc = sa.generate_scene(BAND)
    shuffled = sa.renumber_frames(sc, np.random.RandomState(seed).permutation(sc.M))
    assert _bandwidth(shuffled) > 80
    rc, to_int = _order(shuffled)
    assert rc == 1
    assert _bandwidth(shuffled, to_int) <= _bandwidth(sc) + 2   # (the band of the time order: window - 1)


def test_a_closed_loop_becomes_a_band_of_two_to_three_times_the_width():
    sc = sa.loop_scene(sa.SceneSpec(n_frames=90, grid_nx=20, grid_ny=15, vis_window=0), window=6)
    assert _bandwidth(sc) == sc.M - 1
    rc, to_int = _order(sc)
    assert rc == 1
    assert _bandwidth(sc, to_int) <= 3 * 6 - 3   # (the best numbering of a ring of reach 5 has bandwidth 10; level-wise numbering gives 10-15)


def test_modes_and_degenerate_inputs():
    sc = sa.generate_scene(BAND)
    assert _order(sc, mode=0)[0] == 0
    rc, to_int = _order(sc, mode=1)     # forced: the ordering of a band is the band itself, possibly reversed
    assert _bandwidth(sc, to_int) <= _bandwidth(sc) + 2
    shuffled = sa.renumber_frames(sc, np.random.RandomState(5).permutation(sc.M))
    assert _order(shuffled, mode=0)[0] == 0
    # frames nobody observes, landmarks with one or no observation
    keep = (sc.obs_frame % 7) != 3
    cnt = np.add.reduceat(keep.astype(np.int64), sc.row_ptr[:-1])
    holes = sa.Scene(sc.points, sc.cam_R, sc.cam_T, sc.K, sc.shared_k, np.concatenate([[0], np.cumsum(cnt)]), sc.obs_frame[keep], sc.obs_uv[keep])
    rc, to_int = _order(sa.renumber_frames(holes, np.random.RandomState(3).permutation(sc.M)))
    assert rc == 1
    empty = sa.Scene(sc.points[:3], sc.cam_R, sc.cam_T, sc.K, sc.shared_k, np.zeros(4, np.int64), np.zeros(0, np.int32), np.zeros((0, 2)))
    assert _order(empty)[0] == 0


def test_covisibility_in_a_supplied_numbering_is_that_of_the_renumbered_scene():
    """What a sharded caller hands to srk_ba_set_covisibility after srk_ba_set_frame_order: min_cv of the scene as the
    library stores it."""
    from surikatoko_amd.ba import covisibility, frame_order
    sc = sa.generate_scene(BAND)
    shuffled = sa.renumber_frames(sc, np.random.RandomState(4).permutation(sc.M))
    order = frame_order(shuffled)
    assert order is not None and frame_order(sc) is None
    renumbered = sa.renumber_frames(shuffled, order)
    assert np.array_equal(covisibility(shuffled, order), covisibility(renumbered))
    mc = covisibility(shuffled, order)
    assert np.all(mc <= np.arange(sc.M)) and (np.arange(sc.M) - mc).max() <= BAND.vis_window + 1
