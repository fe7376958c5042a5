"""the dinosaur loader and its pre-processing (host code behind the C ABI), checked against golden
vectors from the reference's Python prototype (obs_geom.py DecomposeProjMat / Triangulate3DPointByLeastSquares) and
by a file round trip (the oxfvisgeom files themselves are not in the reference tree)."""
import os

import numpy as np
import pytest

import surikatoko_amd as sa
from surikatoko_amd import io as sio
from conftest import load_golden


def test_decompose_proj_mat_vs_prototype():
    g = load_golden("pyproto_dino_prep")
    for P, sc, K, R, T in zip(g["P"], g["scale"], g["K"], g["R_direct"], g["T_direct"]):
        ok, s, Kd, Rd, Td = sio.decompose_proj_mat(P)
        assert ok
        assert s == pytest.approx(float(sc), rel=1e-10)
        assert np.abs(Kd - K).max() < 1e-8 * np.abs(K).max()
        assert np.abs(Rd - R).max() < 1e-10 and np.abs(Td - T).max() < 1e-9
        back = s * Kd @ Rd.T @ np.hstack([np.eye(3), -Td.reshape(3, 1)])   # P = scale K R^T [I | -t]  (:666-670)
        assert np.abs(back - P).max() < 1e-8 * np.abs(P).max()
        assert np.linalg.det(Rd) == pytest.approx(1.0, abs=1e-10)


def test_triangulate_vs_prototype():
    g = load_golden("pyproto_dino_prep")
    f0 = float(g["f0"])
    for n, uv, P, X in zip(g["tri_n"], g["tri_uv"], g["tri_P"], g["tri_X"]):
        Xg = sio.triangulate_least_squares(uv[:n], P[:n], f0)
        assert np.abs(Xg - X).max() < 1e-9 * max(1.0, np.abs(X).max())   # colPivQR vs lstsq (SVD)
    with pytest.raises(ValueError):   # CHECK(frames_count >= 2)
        sio.triangulate_least_squares(g["tri_uv"][0][:1], g["tri_P"][0][:1], f0)


def test_read_matrix_from_file(tmp_path):
    p = tmp_path / "m.txt"
    p.write_text("1\t2.5\t-3\n4e2\t5\t6\n")
    m = sio.read_matrix_from_file(p, "\t")
    assert m.shape == (2, 3) and m[1, 0] == 400.0 and m[0, 1] == 2.5
    p.write_text("1 2 3\n4 5\n")
    with pytest.raises(ValueError, match="inconsistent number of columns"):
        sio.read_matrix_from_file(p, " ")
    p.write_text("1 2x 3\n")
    with pytest.raises(ValueError, match="Can't parse number"):
        sio.read_matrix_from_file(p, " ")
    with pytest.raises(ValueError, match="Can't open file"):
        sio.read_matrix_from_file(tmp_path / "missing.txt", " ")
    p.write_text("")
    assert sio.read_matrix_from_file(p, " ").size == 0


def _write_dino_files(directory, sc, pts_gt, Rg, Tg, f0):
    """Files in the oxfvisgeom formats (demo-bundle-adj-dinosaur.cpp:85-116): 3 tab-separated rows of 4 per frame,
    and one row per point with 'x y' per frame, -1 -1 when unseen."""
    M = sc.M
    with open(os.path.join(directory, "dinoPs_as_mat108x4.txt"), "w") as f:
        for j in range(M):
            Kpix = np.diag([f0, f0, 1.0]) @ sc.K[j].reshape(3, 3)
            P = Kpix @ np.hstack([Rg[j].reshape(3, 3), Tg[j].reshape(3, 1)])
            for r in range(3):
                f.write("\t".join(repr(float(v)) for v in P[r]) + "\n")
    with open(os.path.join(directory, "viff.xy"), "w") as f:
        for i in range(sc.N):
            row = -np.ones(2 * M)
            for o in range(sc.row_ptr[i], sc.row_ptr[i + 1]):
                row[2 * sc.obs_frame[o]:2 * sc.obs_frame[o] + 2] = sc.obs_uv[o]
            f.write(" ".join(repr(float(v)) for v in row) + "\n")


def test_dino_loader_round_trip(tmp_path):
    spec = sa.SceneSpec(n_frames=9, grid_nx=6, grid_ny=5, vis_window=4, noise_x3d_hi=0.0, noise_r_hi=0.0)
    sc, pts_gt, Rg, Tg = sa.generate_scene(spec, with_gt=True)
    _write_dino_files(str(tmp_path), sc, pts_gt, Rg, Tg, spec.f0)
    loaded = sio.load_dino_scene(tmp_path, spec.f0)
    assert (loaded.N, loaded.M, loaded.O) == (sc.N, sc.M, sc.O)
    assert np.array_equal(loaded.row_ptr, sc.row_ptr) and np.array_equal(loaded.obs_frame, sc.obs_frame)
    assert np.abs(loaded.obs_uv - sc.obs_uv).max() < 1e-9
    assert np.abs(loaded.K - sc.K).max() < 1e-9            # K(0,1) is 0 in the synthetic scene already
    assert np.abs(loaded.cam_R - Rg).max() < 1e-9 and np.abs(loaded.cam_T - Tg).max() < 1e-8
    assert np.abs(loaded.points - pts_gt).max() < 1e-7     # exact pixels -> triangulation recovers the landmarks


@pytest.mark.gpu
def test_dino_demo_path_on_gpu(tmp_path):
    """demo-dino end to end on a stand-in written in the oxfvisgeom file formats: load -> ComputeInplace."""
    spec = sa.SceneSpec(n_frames=12, grid_nx=9, grid_ny=8, vis_window=4, noise_uv_pix=0.3)
    sc, pts_gt, Rg, Tg = sa.generate_scene(spec, with_gt=True)
    _write_dino_files(str(tmp_path), sc, pts_gt, Rg, Tg, spec.f0)
    scene = sio.load_dino_scene(tmp_path, spec.f0)
    ba = sa.BundleAdjustmentKanatani(0)
    crit = sa.BundleAdjustmentKanataniTermCriteria()
    crit.AllowedReprojErrRelativeChange(1e-5)       # the demo's --allowed_repr_err default (:66)
    e0, seen = ba.ReprojError(spec.f0, scene)
    ok = ba.ComputeInplace(spec.f0, scene, crit, 50)
    assert ba.OptimizationStatusString() in ("small relative err change", "max iterations", "abs err threshold")
    assert ba.report.err_final <= e0 and seen == scene.O
    ba.close()


def _run(cmd, cwd):
    import json
    import subprocess
    p = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads(p.stdout.strip().splitlines()[-1]), p.stderr


def _read_scene_dump(path, orc):
    """demos/scene_dump.hpp: the flat scene a demo handed to (or got back from) ComputeInplace -> (f0, oracle scene)."""
    raw = open(path, "rb").read()
    N, M, O, shared = np.frombuffer(raw, dtype=np.int64, count=4)
    off = 32
    f0 = float(np.frombuffer(raw, dtype=np.float64, count=1, offset=off)[0])
    off += 8

    def take(dtype, n):
        nonlocal off
        a = np.frombuffer(raw, dtype=dtype, count=int(n), offset=off)
        off += a.nbytes
        return a
    pts, R, T = take(np.float64, 3 * N), take(np.float64, 9 * M), take(np.float64, 3 * M)
    K = take(np.float64, 9 * (1 if shared else M))
    row_ptr, frames, uv = take(np.int64, N + 1), take(np.int32, O), take(np.float64, 2 * O)
    assert off == len(raw)
    return f0, orc.Scene(pts, R, T, K, int(shared), row_ptr, frames, uv)


@pytest.mark.gpu
def test_demo_circle_grid_cli_with_reference_flagfile(orc, tmp_path):
    """The C++ drop-in of demo-circle-grid with the reference's flagfile values (81 points x 36 frames, all visible,
    rotation noise only; cpp_impl/flagfile-demo-circle-grid.txt, call at demo-bundle-adj-circle-grid.cpp:285-293)
    against the CPU oracle run on exactly the scene the demo handed to ComputeInplace: same result flag, status string,
    iteration and attempt counts, error rel 1e-6, output scene abs 1e-6."""
    from conftest import ROOT
    exe = os.path.join(ROOT, "demos", "demo-circle-grid")
    if not os.path.exists(exe):
        pytest.skip("demos not built")
    before, after = str(tmp_path / "before.bin"), str(tmp_path / "after.bin")
    out, log = _run([exe, "--flagfile=" + os.path.join(ROOT, "demos", "flagfile-demo-circle-grid.txt"),
                     "--max_iterations=25", "--dump_scene_before=" + before, "--dump_scene_after=" + after], ROOT)
    assert out["frames"] == 36 and out["points"] == 81
    assert "bundle adjustment finished with result" in log
    f0, so = _read_scene_dump(before, orc)
    _, sg = _read_scene_dump(after, orc)
    assert (f0, so.N, so.M, so.O, so.shared_k) == (600.0, 81, 36, 81 * 36, 0)
    rc_o, rep_o = orc.compute_inplace(f0, so, 2.25e-12, None, 25)   # the flagfile's --allowed_repr_err
    assert out["result"] == int(rc_o == 0)
    assert out["status"] == orc.status_string(rep_o.status)
    assert (out["iterations"], out["attempts"]) == (rep_o.iterations, rep_o.attempts)
    assert out["err_initial"] == pytest.approx(rep_o.err_initial, rel=1e-12)
    assert out["err_final"] == pytest.approx(rep_o.err_final, rel=1e-6, abs=1e-18)
    assert out["iterations"] >= 1 and out["err_final"] < out["err_initial"]
    assert np.abs(sg.points - so.points).max() < 1e-6
    assert np.abs(sg.cam_R - so.cam_R).max() < 1e-6
    assert np.abs(sg.cam_T - so.cam_T).max() < 1e-6


@pytest.mark.gpu
def test_demo_dino_cli(tmp_path):
    from conftest import ROOT
    exe = os.path.join(ROOT, "demos", "demo-dino")
    if not os.path.exists(exe):
        pytest.skip("demos not built")
    spec = sa.SceneSpec(n_frames=10, grid_nx=8, grid_ny=7, vis_window=4, noise_uv_pix=0.2)
    sc, pts_gt, Rg, Tg = sa.generate_scene(spec, with_gt=True)
    d = tmp_path / "oxfvisgeom" / "dinosaur"
    d.mkdir(parents=True)
    _write_dino_files(str(d), sc, pts_gt, Rg, Tg, spec.f0)
    out, log = _run([exe, f"--testdata={tmp_path}", "--f0=600", "--allowed_repr_err=4.56e-8", "--max_iterations=30"], ROOT)
    assert out["seen"] == sc.O and out["err_final"] <= out["err_initial"]
    # same scene through the Python mirror gives the same numbers
    scene = sio.load_dino_scene(d, 600.0)
    ba = sa.BundleAdjustmentKanatani(0)
    crit = sa.BundleAdjustmentKanataniTermCriteria()
    crit.AllowedReprojErrRelativeChange(4.56e-8)
    ba.ComputeInplace(600.0, scene, crit, 30)
    assert ba.report.iterations == out["iterations"] and ba.report.attempts == out["attempts"]
    assert ba.report.err_final == pytest.approx(out["err_final"], rel=1e-9)
    ba.close()


@pytest.mark.gpu
def test_cpp_adapter_mvf_call_contract(orc, tmp_path):
    """the C++ mirror class in the multi-view-factorization call contract (shared K, f0 = 1,
    threshold 1e-3, salient points created out of track order, one track without a salient point;
    multi-view-factorization.cpp:379-394) against the CPU oracle on the scene the adapter flattened, and against the
    flat C-ABI call."""
    from conftest import ROOT
    exe = os.path.join(ROOT, "demos", "test-adapter")
    if not os.path.exists(exe):
        pytest.skip("adapter test not built")
    before, after = str(tmp_path / "before.bin"), str(tmp_path / "after.bin")
    out, _ = _run([exe, before, after], ROOT)
    assert out["ok"] == (out["rc"] == 0)
    assert out["seen"] == 42 * 5 and out["points"] == 42 and out["vars"] == 3 * 42 + 90 and out["normalized_vars"] == out["vars"] - 7
    # the adapter against the oracle
    f0, so = _read_scene_dump(before, orc)
    _, sg = _read_scene_dump(after, orc)
    assert (f0, so.N, so.M, so.O, so.shared_k) == (1.0, 42, 9, 210, 1)
    e0, seen = orc.reproj_error(f0, so)
    assert seen == out["seen"] and out["err0"] == pytest.approx(e0, rel=1e-12)
    rc_o, rep_o = orc.compute_inplace(f0, so, 1e-3, None, 0)
    assert out["ok"] == int(rc_o == 0) and out["status"] == orc.status_string(rep_o.status)
    assert (out["iterations"], out["attempts"]) == (rep_o.iterations, rep_o.attempts)
    assert out["err_final"] == pytest.approx(rep_o.err_final, rel=1e-6, abs=1e-18)
    assert np.abs(sg.points - so.points).max() < 1e-6
    assert np.abs(sg.cam_R - so.cam_R).max() < 1e-6
    assert np.abs(sg.cam_T - so.cam_T).max() < 1e-6
    # the adapter against the flat C ABI (same library underneath: must agree to rounding)
    assert out["err0"] == pytest.approx(out["err0_c"], rel=1e-12)
    assert out["iterations"] == out["iterations_c"]
    assert out["err_final"] == pytest.approx(out["err_final_c"], rel=1e-9)
    assert out["maxdiff"] < 1e-9


@pytest.mark.gpu
def test_demo_multi_view_factorization_cli(orc, tmp_path):
    """The drop-in of demo-multi-view-factorization (reference flags, cpp_impl/demos/demo-multi-view-factorization.cpp:
    351-370; frame loop :529-656; driver multi-view-factorization.cpp:255-397) with the reference flagfile's values:
    two ground-truth frames, then frames integrated by srk_mvf_relative_motion / srk_mvf_estimate_depths, bundle
    adjustment whenever the driver's score exceeds 1e-3 -- every such call (shared K, f0 = 1, threshold 1e-3) is
    replayed through the CPU oracle on the scene the demo handed over and must give the same result."""
    import glob
    from conftest import ROOT
    exe = os.path.join(ROOT, "demos", "demo-multi-view-factorization")
    if not os.path.exists(exe):
        pytest.skip("demos not built")
    prefix = str(tmp_path / "ba")
    out, log = _run([exe, "--flagfile=" + os.path.join(ROOT, "demos", "flagfile-demo-multi-view-factorization.txt"),
                     "--max_frames=13", "--ba_max_iterations=6", "--dump_ba_prefix=" + prefix], ROOT)
    assert out["world_points"] == 81 * 41 and out["frames"] + out["failed_frames"] == 13
    assert out["integrated_frames"] >= 9 and out["salient_points"] > 500 and out["max_pose_diff"] < 1e-3
    assert "anchored on f=" in log and "reconstructed_salient_points_count=" in log
    calls = sorted(glob.glob(prefix + "_*_before.bin"))
    assert len(calls) == out["ba_calls"] and out["ba_calls"] >= 1
    its = atts = 0
    for before in calls:
        f0, so = _read_scene_dump(before, orc)
        _, sg = _read_scene_dump(before.replace("_before", "_after"), orc)
        assert f0 == 1.0 and so.shared_k == 1 and so.M >= 3
        rc_o, rep_o = orc.compute_inplace(1.0, so, 1e-3, None, 6)
        its += rep_o.iterations
        atts += rep_o.attempts
        scale = max(1.0, float(np.abs(so.points).max()))
        assert np.abs(sg.points - so.points).max() < 1e-6 * scale
        assert np.abs(sg.cam_R - so.cam_R).max() < 1e-6
        assert np.abs(sg.cam_T - so.cam_T).max() < 1e-6 * scale
    assert (out["ba_iterations"], out["ba_attempts"]) == (its, atts)


@pytest.mark.gpu
def test_demo_multi_view_factorization_full_flagfile_ends_at_frame_29():
    """The reference flagfile's whole camera path (60 frames, cpp_impl/flagfile-demo-multi-view-factorization.txt) through
    the drop-in: where tracking ends and why.  Frames 0-1 come from the ground truth, frames 2..28 are integrated (nine BA
    calls on the way); frame 29 -- the first one after the viewer's path turns its first corner -- shares only 4
    reconstructed tracks with its best anchor (frame 27).  Four points give 8 of the 11 equations the 12-unknown motion
    system needs (multi-view-factorization.cpp:107-189): srk_mvf_relative_motion refuses (< 6 points) and the drop-in stops
    integrating, logging the reason.  The reference driver only gives up when there are NO common points (:262-270); with
    4 it would take whatever null vector its SVD returns and carry on with a meaningless pose ("diverged cam localiz",
    :283-295).  No fixture of the reference pins either behaviour."""
    from conftest import ROOT
    exe = os.path.join(ROOT, "demos", "demo-multi-view-factorization")
    if not os.path.exists(exe):
        pytest.skip("demos not built")
    out, log = _run([exe, "--flagfile=" + os.path.join(ROOT, "demos", "flagfile-demo-multi-view-factorization.txt"),
                     "--ba_max_iterations=50"], ROOT)
    assert out["world_points"] == 81 * 41
    assert (out["frames"], out["integrated_frames"], out["failed_frames"]) == (29, 27, 31)
    assert "f=29 anchored on f=27 using common_points=4" in log
    assert "tracking lost at frame 29: relative motion from 4 common points failed" in log
    assert out["ba_calls"] >= 5 and out["max_pose_diff"] < 1e-3 and out["last_reproj_err"] < 1e-3
    assert out["ba_last_frames"] == 29
