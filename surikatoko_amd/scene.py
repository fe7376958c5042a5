"""Synthetic circle-grid scenes through the C ABI (srk_scene_generate, surikatoko_amd/csrc/srk_scene.cpp),
restating cpp_impl/demos/demo-bundle-adj-circle-grid.cpp:86-257 of the reference."""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from ._lib import SceneSpecC, lib


@dataclass
class SceneSpec:
    n_frames: int
    grid_nx: int
    grid_ny: int
    vis_window: int = 20
    half_extent_x: float = 1.0
    half_extent_y: float = 1.0
    f0: float = 600.0
    noise_x3d_hi: float = 0.005
    noise_r_hi: float = 0.005
    noise_uv_pix: float = 0.0
    seed: int = 1234

    def to_c(self):
        return SceneSpecC(self.n_frames, self.grid_nx, self.grid_ny, self.vis_window, self.half_extent_x,
                          self.half_extent_y, self.f0, self.noise_x3d_hi, self.noise_r_hi, self.noise_uv_pix,
                          self.seed)


# BASELINE.json configs (SURVEY 8d grid sizes); C1 is the labelled synthetic stand-in for the missing dino files
CONFIGS = {
    # 36 cams, 4983 pts; config_scene() trims the 4-frame windows to the 16432 observations the dino flagfile states
    "C1_dino_standin": SceneSpec(36, 33, 151, vis_window=4),
    "C2_200cam_20kpt": SceneSpec(200, 200, 100, vis_window=20),      # 400k obs
    # config 2 the way the reference's demo itself builds it (demo-bundle-adj-circle-grid.cpp:196-207: every point projected
    # into every frame): 4M obs, a DENSE 1993^2 reduced camera system ("one run with full visibility")
    "C2_all_visible": SceneSpec(200, 200, 100, vis_window=0),
    "C3_1kcam_100kpt": SceneSpec(1000, 400, 250, vis_window=20),     # 2M obs (headline)
    "C5_4kcam_1Mpt": SceneSpec(4000, 1000, 1000, vis_window=20),     # 20M obs
    "demo_circle_grid": SceneSpec(36, 5, 5, vis_window=0),           # the demo's own 36-frame all-visible scene
}


def generate_scene(spec: SceneSpec, with_gt=False):
    """Returns a surikatoko_amd.ba.Scene (and the ground truth arrays when with_gt)."""
    from .ba import Scene
    L = lib()
    cs = spec.to_c()
    O = L.srk_scene_num_observations(C.byref(cs))
    if O < 0:
        raise ValueError("bad scene spec")
    N, M = spec.grid_nx * spec.grid_ny, spec.n_frames
    pts = np.zeros((N, 3))
    pts_gt = np.zeros((N, 3))
    R = np.zeros((M, 9))
    T = np.zeros((M, 3))
    Rg = np.zeros((M, 9))
    Tg = np.zeros((M, 3))
    K = np.zeros((M, 9))
    row_ptr = np.zeros(N + 1, dtype=np.int64)
    obs_frame = np.zeros(O, dtype=np.int32)
    obs_uv = np.zeros((O, 2))
    dp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rc = L.srk_scene_generate(C.byref(cs), dp(pts), dp(pts_gt), dp(R), dp(T), dp(Rg), dp(Tg), dp(K), dp(row_ptr),
                              dp(obs_frame), dp(obs_uv))
    if rc != 0:
        raise RuntimeError(f"srk_scene_generate failed: {rc}")
    sc = Scene(pts, R, T, K, False, row_ptr, obs_frame, obs_uv)
    if with_gt:
        return sc, pts_gt, Rg, Tg
    return sc


DINO_OBSERVATIONS = 16432  # cpp_impl/flagfile-demo-dino.txt:7 ("16432*0.01^2/600^2")


def config_scene(name, with_gt=False):
    """The scene of a BASELINE.json config.  C1 is the labelled synthetic stand-in for the oxfvisgeom dinosaur files
    (not in the reference tree): 36 turntable cameras, 4983 points (demo-bundle-adj-dinosaur.cpp:97,116)
    and exactly 16432 observations: 1483 landmarks keep their 4-frame window, the other 3500 lose its last frame
    (a deterministic choice; every landmark keeps >= 3 consecutive frames)."""
    spec = CONFIGS[name]
    out = generate_scene(spec, with_gt=with_gt)
    if name != "C1_dino_standin":
        return out
    from .ba import Scene
    sc = out[0] if with_gt else out
    N = sc.N
    assert N == 4983 and sc.O == 4 * N
    keep4 = (np.arange(N, dtype=np.int64) * 7919) % N < DINO_OBSERVATIONS - 3 * N
    keep = np.ones(sc.O, dtype=bool)
    keep[sc.row_ptr[1:][~keep4] - 1] = False
    counts = np.where(keep4, 4, 3).astype(np.int64)
    rp = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    trimmed = Scene(sc.points, sc.cam_R, sc.cam_T, sc.K, sc.shared_k, rp, sc.obs_frame[keep], sc.obs_uv[keep])
    assert trimmed.O == DINO_OBSERVATIONS
    return (trimmed,) + tuple(out[1:]) if with_gt else trimmed


def drop_observations(scene, fraction, seed=0, keep_min=2):
    """A copy of `scene` with about `fraction` of its observations removed at random (every landmark keeps at least
    `keep_min`, its first and last ones among them): ragged tracks as a feature tracker produces them, where hardly any
    two landmarks see exactly the same frames."""
    import numpy as np
    from .ba import Scene
    rng = np.random.RandomState(seed)
    keep = rng.rand(scene.O) >= fraction
    rp = scene.row_ptr
    for i in range(scene.N):
        lo, hi = int(rp[i]), int(rp[i + 1])
        if hi - lo <= keep_min:
            keep[lo:hi] = True
        else:
            keep[lo] = keep[hi - 1] = True
    counts = np.add.reduceat(keep.astype(np.int64), rp[:-1]) if scene.N else np.zeros(0, np.int64)
    counts[rp[:-1] == rp[1:]] = 0
    new_rp = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    return Scene(scene.points, scene.cam_R, scene.cam_T, scene.K, scene.shared_k, new_rp, scene.obs_frame[keep],
                 scene.obs_uv[keep])


def renumber_frames(scene, new_index):
    """The same scene with its frames numbered differently: frame j becomes frame new_index[j] (cameras and intrinsics
    move with it, every landmark's observations are re-sorted by the new index).  What an unordered image set looks like
    to the solver; the reference (a dense system, bundle-adj-kanatani.cpp:1911) is indifferent to it."""
    import numpy as np
    from .ba import Scene
    new_index = np.asarray(new_index, np.int64)
    M = scene.M
    assert sorted(new_index.tolist()) == list(range(M))
    old_of_new = np.argsort(new_index)
    K = scene.K if scene.shared_k else scene.K[old_of_new]
    f = new_index[scene.obs_frame]
    lm = np.repeat(np.arange(scene.N), np.diff(scene.row_ptr))
    order = np.lexsort((f, lm))  # by landmark, then by new frame index
    return Scene(scene.points, scene.cam_R[old_of_new], scene.cam_T[old_of_new], K, scene.shared_k, scene.row_ptr,
                 f[order].astype(np.int32), scene.obs_uv[order])


def loop_scene(spec: SceneSpec, window):
    """A sequence that closes a loop: the all-visible scene of `spec` (vis_window 0) cut down so that landmark i is seen by
    the `window` cyclically consecutive frames starting at a pseudo-random frame -- the last frames share landmarks with
    the first ones, so the reduced camera system is a band plus two corner blocks."""
    import numpy as np
    from .ba import Scene
    assert spec.vis_window == 0
    sc = generate_scene(spec)
    M, N = sc.M, sc.N
    assert np.all(np.diff(sc.row_ptr) == M) and 2 <= window < M
    start = (np.arange(N, dtype=np.uint64) * np.uint64(2654435761) % np.uint64(2 ** 32) % np.uint64(M)).astype(np.int64)
    frames = np.sort((start[:, None] + np.arange(window)[None, :]) % M, axis=1)  # [N, window] ascending
    idx = (np.arange(N)[:, None] * M + frames).ravel()
    row_ptr = np.arange(N + 1, dtype=np.int64) * window
    return Scene(sc.points, sc.cam_R, sc.cam_T, sc.K, sc.shared_k, row_ptr, frames.ravel().astype(np.int32), sc.obs_uv[idx])
