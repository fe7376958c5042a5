#!/usr/bin/env python3
"""bench.py -- BA iterations/sec of the MI355X bundle-adjustment core on the BASELINE.json headline scene.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one accepted outer Levenberg-Marquardt iteration of the reference loop (bundle-adj-kanatani.cpp:756-891)
on the resident scene: derivatives (Jacobian / normal-equation blocks) -> [reduced camera system (Schur) -> dense solve
-> back-substitution -> apply -> reprojection error] per attempt, with as many attempts as the damping adaptation of
that iteration needs.  The timed region is ONE optimise call of K iterations continuing from the uploaded state (the
first iteration needs one attempt on the synthetic scenes, later ones two to three; "first_iteration" in the JSON line
carries the one-attempt time on its own).  Inputs are resident in HBM before the timed region.  At N > 1 the SAME scene is sharded by landmark over the ranks (strong scaling); the exchange steps are
all-reduces over RCCL/xGMI through torch.distributed.

Rank 0 prints ONE JSON line with the driver's contract plus "roofline" (dominant kernel), "kernels" (per-phase
rooflines) and, at N = 1, "cpu_baseline" (the CPU oracle timed on this box's host cores on a bounded sample).
"""
import argparse
import json

import numpy as np
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X datasheet FP64 matrix (v_mfma_f64_16x16x4_f64: 2048 flop / 64 cycles / SIMD)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C3_1kcam_100kpt")
    ap.add_argument("--drop", type=float, default=0.0,
                    help="remove this fraction of the observations at random (ragged tracks; not the headline workload)")
    ap.add_argument("--schur-fp32", action="store_true",
                    help="opt-in mixed precision (fp32 run sums in the Schur kernel); never the headline configuration")
    ap.add_argument("--store-f32", action="store_true",
                    help="opt-in f32 storage of the point-frame blocks W (arithmetic stays fp64); never the headline configuration")
    ap.add_argument("--sequential-attempts", action="store_true",
                    help="one attempt slot (srk_ba_set_speculation off): kernels of different attempts never overlap, "
                         "so per-kernel durations under rocprofv3 are those of the kernel alone (profiles/)")
    ap.add_argument("--deterministic", action="store_true",
                    help="srk_ba_set_deterministic: ordered sums instead of fp64 atomics (never the default line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-one-call", action="store_true", help="skip the one-call (upload + LM + download) latency probe")
    ap.add_argument("--cpu-sample", default="C2_200cam_20kpt")
    ap.add_argument("--no-dense-probe", action="store_true")
    ap.add_argument("--rcs", choices=["chunks", "skyline", "dense"], default="chunks",
                    help="reduced-camera-system solver: exploit the covisibility skyline (exact) or treat it as dense")
    return ap.parse_args()


def algorithmic_bytes(N, M, O, ld, rcs_fill=1.0):
    """SURVEY 8(d) per-call algorithmic bytes (fp64).  The reduced camera system counts only the entries inside the
    covisibility skyline (rcs_fill = 1 when it is treated as dense)."""
    k2 = O * (4 + 16) + (N + 1) * 8 + N * 24 + M * (72 + 24 + 72) + O * 240 + N * (72 + 24) + M * (800 + 80)
    k1 = O * 20 + (N + 1) * 8 + N * 24 + M * 168 + 8
    n = 10 * M - 7
    k3 = O * 240 + N * 96 + int(rcs_fill * n * n * 8)
    k5 = O * 240 + N * (72 + 24 + 24) + M * 80
    k4_flops = n ** 3 / 3.0
    return dict(jacobian=k2, error=k1, schur=k3, backsub=k5, solve_flops=k4_flops, solve_bytes=n * n * 8)


def _oracle_scene(orc, sc):
    return orc.Scene(sc.points, sc.cam_R, sc.cam_T, sc.K, sc.shared_k, sc.row_ptr, sc.obs_frame, sc.obs_uv)


def _gpu_rate(sa, name, steps, crit=None, f0=None):
    """The GPU path on a CPU-baseline sample scene, same run: K iterations of one continuing optimise call."""
    spec = sa.CONFIGS[name]
    sc = sa.config_scene(name)
    ba = sa.BundleAdjustmentKanatani(0)
    try:
        assert ba.upload(spec.f0 if f0 is None else f0, sc)
        ba.optimize(crit, max_iterations=2)   # warm-up
        ba.reset()
        import torch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ba.optimize(crit, max_iterations=steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        r = ba.report
        return {"iterations_per_s": r.iterations / dt if dt > 0 else None, "iterations": int(r.iterations),
                "attempts": int(r.attempts), "seconds": dt}
    finally:
        ba.close()


def cpu_baseline(sample_name, steps):
    """The CPU oracle (plain-C restatement of the reference) timed on this box's host cores on bounded samples, next to
    the GPU path on the SAME sample scenes in the same run.  BASELINE.md section 3 variants:
      top-level "value": ONE complete outer LM iteration of the HEADLINE scene on one core -- the reference's arithmetic on
            skyline storage with a skyline Cholesky (variant ii as a competent CPU port would do it; the literal 9993^2
            Householder QR would take hours), and "allcore": three iterations with OpenMP on the host's cores
      (ii)  block-sparse storage, literal Householder QR, ONE thread, on the C2 sample ("literal_qr_sample")
      (iii) the same with OpenMP over points / rows / columns on the host's cores (bit-identical results)
      (i)   the reference's literal dense storage and per-point n x n product (bundle-adj-kanatani.cpp:581,1891,1911),
            one thread, on config 1 (n = 353); infeasible beyond config 2."""
    import surikatoko_amd as sa
    from oracle import oracle as orc
    spec = sa.CONFIGS[sample_name]
    sc = sa.config_scene(sample_name)
    host_cpus = os.cpu_count() or 1

    def phases(rep):
        return {"derivatives": rep.sec_derivatives, "schur": rep.sec_schur, "solve": rep.sec_solve,
                "backsub": rep.sec_backsub, "apply": rep.sec_apply, "error": rep.sec_error}

    def run(scene, f0, threads=1, max_it=1, dense=False, allowed=None):
        orc.set_threads(threads)
        try:
            so = _oracle_scene(orc, scene)
            t0 = time.perf_counter()
            rc, rep = orc.compute_inplace(f0, so, allowed, None, max_it, dense_literal=dense)
            dt = time.perf_counter() - t0
        finally:
            orc.set_threads(1)
        return {"iterations_per_s": rep.iterations / dt if dt > 0 and rep.iterations else None,
                "iterations": int(rep.iterations), "attempts": int(rep.attempts), "seconds": dt,
                "phase_seconds": phases(rep), "cores": threads}

    threads = max(1, min(host_cpus, 64))
    one = run(sc, spec.f0)                                                   # (ii), literal QR
    allc = run(sc, spec.f0, threads=threads)                                 # (iii)
    c1 = sa.config_scene("C1_dino_standin")
    dense = run(c1, 600.0, max_it=2, dense=True, allowed=4.56e-8)            # (i)
    c1_sparse = run(c1, 600.0, max_it=2, allowed=4.56e-8)
    # the HEADLINE scene, complete iterations: the reference's arithmetic on skyline storage with a skyline Cholesky
    # (BASELINE.md variant ii as "a competent CPU port would do"; oracle: orc_two_phase_skyline, checked against the
    # literal QR path by tests/test_oracle_skyline.py) -- the literal 9993^2 Householder QR alone would take hours
    head1 = headN = None
    try:
        orc.set_solver(1)
        hs = sa.config_scene("C3_1kcam_100kpt")
        hf0 = sa.CONFIGS["C3_1kcam_100kpt"].f0
        head1 = run(hs, hf0, threads=1, max_it=1)
        headN = run(hs, hf0, threads=threads, max_it=3)
        for hd in (head1, headN):
            hd["attempts_per_s"] = hd["attempts"] / hd["seconds"] if hd["seconds"] > 0 else None
    except Exception as e:  # noqa: BLE001
        head1 = head1 or {"failed": repr(e)}
        headN = headN or {"failed": repr(e)}
    finally:
        orc.set_solver(0)
    crit = sa.BundleAdjustmentKanataniTermCriteria()
    crit.AllowedReprojErrRelativeChange(4.56e-8)
    gpu_sample = _gpu_rate(sa, sample_name, steps)
    gpu_c1 = _gpu_rate(sa, "C1_dino_standin", 10, crit=crit, f0=600.0)
    ok_head = isinstance(head1, dict) and head1.get("iterations_per_s")
    return {
        # the headline workload itself, one complete outer LM iteration on ONE core (its first iteration: one attempt)
        "value": head1["iterations_per_s"] if ok_head else one["iterations_per_s"],
        "unit": "iterations/s",
        "cores": 1,
        "kind": "port",
        "sample": ((f"1 complete outer LM iteration ({head1['attempts']} attempt) of the headline scene C3_1kcam_100kpt on one "
                    "core: CPU oracle, reference arithmetic on skyline storage + skyline Cholesky (BASELINE.md variant ii); "
                    f"all-core: 3 iterations on {threads} threads")
                   if ok_head else
                   f"1 outer LM iteration of {sample_name} (literal Householder QR); the headline-scene run failed"),
        "seconds": head1["seconds"] if ok_head else one["seconds"],
        "attempts_per_s": head1.get("attempts_per_s") if ok_head else None,
        "phase_seconds": head1["phase_seconds"] if ok_head else one["phase_seconds"],
        "host_cpus": host_cpus,
        "allcore": ({"iterations_per_s": headN.get("iterations_per_s"), "attempts_per_s": headN.get("attempts_per_s"),
                     "cores": threads, "iterations": headN.get("iterations"), "attempts": headN.get("attempts"),
                     "seconds": headN.get("seconds")} if isinstance(headN, dict) else None),
        "literal_qr_sample": {"scene": sample_name, "iterations_per_s": one["iterations_per_s"], "seconds": one["seconds"],
                              "attempts": one["attempts"], "cores": 1,
                              "note": f"1 outer LM iteration of {sample_name} with the reference's Householder QR of the "
                                      f"{10 * sc.M - 7}^2 reduced system (BA:1911), one core"},
        "gpu_on_sample": gpu_sample,
        "gpu_over_cpu_on_sample": (gpu_sample["iterations_per_s"] / one["iterations_per_s"]
                                   if gpu_sample["iterations_per_s"] and one["iterations_per_s"] else None),
        "variants": {
            "headline_scene_skyline_cholesky_1core": head1,
            "headline_scene_skyline_cholesky_allcore": headN,
            "ii_sparse_1core": dict(one, scene=sample_name),
            "iii_sparse_allcore": dict(allc, scene=sample_name,
                                       note="OpenMP: derivatives over points / frames, Schur sum by row ownership (every "
                                            "entry keeps its sequential summation order), QR reflector application over "
                                            "columns, back-substitution over points; results bit-identical to one core"),
            "i_dense_literal_1core": dict(dense, scene="C1_dino_standin", sparse_1core_same_run=c1_sparse,
                                          gpu_same_scene=gpu_c1,
                                          note="reference storage: dense 3 x n row-block per point and an n x n product "
                                               "per point (n = 353); 2 outer iterations with the dino flagfile's threshold"),
        },
    }


def one_call_latency(sa):
    """Wall time of ONE srk_ba_compute_inplace call (validation, landmark sort, grouping, solver plan, allocation, upload,
    LM loop, download, revert) at the sizes the reference's callers use; never part of `value`.
    * config 1 (demo-dino stand-in, per-frame K, f0 = 600, the dino flagfile's threshold);
    * the BA calls of the multi-view-factorization driver on the reference flagfile's scene (see below)."""
    import torch
    out = {}

    def timed(ba, f0, scene, crit, max_it=0):
        sg = scene.copy()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ok = ba.ComputeInplace(f0, sg, crit, max_it)
        dt = 1e3 * (time.perf_counter() - t0)
        r = ba.report
        return {"ms": dt, "ms_lm_loop": r.ms_total, "iterations": int(r.iterations), "attempts": int(r.attempts),
                "result": bool(ok), "status": ba.OptimizationStatusString()}

    crit = sa.BundleAdjustmentKanataniTermCriteria()
    crit.AllowedReprojErrRelativeChange(4.56e-8)
    c1 = sa.config_scene("C1_dino_standin")
    ba = sa.BundleAdjustmentKanatani(0)
    try:
        out["C1_dino_standin_first_call"] = timed(ba, 600.0, c1, crit)
        out["C1_dino_standin_warm_handle"] = timed(ba, 600.0, c1, crit)
    finally:
        ba.close()
    # every track in every frame, at the size of the MVF flagfile's world (60 frames / 3321 points): the shape the demos'
    # all-visible scenes have (tracks longer than 24 frames: k_schur_long)
    av = sa.generate_scene(sa.SceneSpec(60, 81, 41, vis_window=0))
    ba = sa.BundleAdjustmentKanatani(0)
    try:
        out["all_visible_60x3321_first_call"] = timed(ba, 600.0, av, crit, 5)
        out["all_visible_60x3321_warm_handle"] = timed(ba, 600.0, av, crit, 5)
    finally:
        ba.close()
    # the multi-view-factorization caller itself: the drop-in of the reference demo with the reference flagfile's values
    # (60 frames / 3321 points; cpp_impl/flagfile-demo-multi-view-factorization.txt) hands BA the scene it has built so
    # far whenever its score exceeds 1e-3 (multi-view-factorization.cpp:379-394); wall time of those calls as the demo
    # measures it around ComputeInplace (shared K, f0 = 1, threshold 1e-3, at most 50 LM iterations a call)
    exe = os.path.join(ROOT, "demos", "demo-multi-view-factorization")
    if os.path.exists(exe):
        import subprocess
        p = subprocess.run([exe, "--flagfile=" + os.path.join(ROOT, "demos", "flagfile-demo-multi-view-factorization.txt"),
                            "--ba_max_iterations=50"], capture_output=True, text=True, timeout=300, cwd=ROOT)
        if p.returncode == 0:
            j = json.loads(p.stdout.strip().splitlines()[-1])
            out["mvf_demo_60_frames"] = {
                "ba_calls": j["ba_calls"], "ba_ms_total": j["ba_ms_total"], "ba_ms_max": j["ba_ms_max"],
                "ba_ms_mean": j["ba_ms_total"] / max(j["ba_calls"], 1), "ba_iterations": j["ba_iterations"],
                "ba_attempts": j["ba_attempts"], "last_call": {"points": j["ba_last_points"], "frames": j["ba_last_frames"],
                                                               "observations": j["ba_last_observations"]},
                "integrated_frames": j["integrated_frames"], "failed_frames": j["failed_frames"]}
        else:
            out["mvf_demo_60_frames"] = {"failed": p.stderr[-300:]}
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import surikatoko_amd as sa

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    # SRK_BENCH_BACKEND=gloo: rehearsal of the N > 1 path with several ranks on ONE GPU (the exchanges are staged through
    # host memory; RCCL does not run two ranks on one device).  Never the measured configuration.
    backend = os.environ.get("SRK_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend=backend)

    spec = sa.CONFIGS[args.config]
    scene = sa.config_scene(args.config)
    if args.drop > 0:
        scene = sa.drop_observations(scene, args.drop, seed=1)
    N_total, M, O_total = scene.N, scene.M, scene.O
    # gauge-normalise ONCE on the whole scene so that every shard sees the same normalised cameras
    ok, _nrm = sa.normalize_scene_inplace(scene)
    assert ok, "scene cannot be normalised"
    if world > 1:
        shard, (lo, hi) = scene.shard(rank, world)
    else:
        shard, (lo, hi) = scene, (0, scene.N)

    ba = sa.BundleAdjustmentKanatani(local_rank)
    ba.set_profile(0)  # the timed region carries no instrumentation; phases are timed in separate steps below
    if args.schur_fp32:
        ba.set_schur_precision(True)
    if args.store_f32:
        ba.set_storage_precision(True)
    if args.sequential_attempts:
        ba.set_speculation(False)
    if args.deterministic:
        ba.set_deterministic(True)
    exchange = "none"
    if world > 1:
        # N > 1: the library's own RCCL all-reduces on its streams (srk_ba_rccl_init; the unique id travels through
        # torch.distributed, nothing else does).  SRK_BENCH_EXCHANGE=torch keeps the Python callback of round 1
        # (torch.distributed.all_reduce with a host synchronisation on either side); the gloo rehearsal needs it.
        if backend == "nccl" and os.environ.get("SRK_BENCH_EXCHANGE", "rccl") == "rccl":
            # Every rank takes part in every collective of the set-up, whatever happened to it locally, and the ranks agree
            # (all-reduce MIN) after each step before the next one starts.  A native set-up that fails on any rank ends the
            # run with a non-zero exit on EVERY rank: a SCALE run must never silently time the host-synchronised callback
            # path (SRK_BENCH_EXCHANGE=torch selects that path on purpose).
            def agree(ok):
                flag = torch.tensor([1 if ok else 0], device=f"cuda:{local_rank}", dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                return int(flag.item()) == 1

            def new_id():
                ident = [None]
                if rank == 0:
                    try:
                        ident[0] = ba.rccl_unique_id()
                    except Exception as e:  # noqa: BLE001
                        print("bench.py: srk_ba_rccl_get_unique_id failed:", repr(e), file=sys.stderr, flush=True)
                dist.broadcast_object_list(ident, src=0)
                return ident[0]

            def native_step(what, fn):
                ok = True
                try:
                    fn()
                except Exception as e:  # noqa: BLE001
                    ok = False
                    print(f"bench.py[rank {rank}]: {what} failed: {e!r}", file=sys.stderr, flush=True)
                if not agree(ok):
                    ba.close()
                    dist.barrier()
                    dist.destroy_process_group()
                    raise SystemExit(f"bench.py: native RCCL set-up failed at '{what}' on at least one rank "
                                     "(SRK_BENCH_EXCHANGE=torch selects the torch.distributed callback instead)")

            id1 = new_id()
            native_step("srk_ba_rccl_init", lambda: ba.rccl_init(id1, rank, world))
            ba.set_multi_schedule(os.environ.get("SRK_BENCH_SCHEDULE", "dp"))  # "dp" (default) or "allreduce"
            exchange = "native RCCL"
        else:
            from surikatoko_amd.dist import make_allreduce_hook
            ba.set_allreduce(make_allreduce_hook(None, f"cuda:{local_rank}"), rank, world)
            exchange = f"torch.distributed all_reduce callback ({backend})"
    assert ba.upload(spec.f0, shard, already_normalized=True)
    RCS_MODE = {"dense": 0, "skyline": 1, "chunks": 2}
    if args.rcs != "chunks":
        ba.set_rcs_mode(RCS_MODE[args.rcs])
    if args.rcs != "dense" and world > 1:
        from surikatoko_amd.ba import covisibility
        ba.set_covisibility(covisibility(scene))  # global skyline: every rank factorises the same all-reduced system
    deterministic_on = ba.deterministic()
    rcs_fill = ba.rcs_fill()
    mfma_flops = ba.solve_mfma_flops()
    rcs_chunks = ba.rcs_chunks()
    jac_kernel = ba.jacobian_kernel()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run(k):
        """one LM run of k outer iterations from the uploaded state"""
        ba.reset()
        ba.optimize(None, max_iterations=k)
        return ba.report

    def step():
        return run(1)

    # A step = one accepted outer LM iteration of a CONTINUING run, with the rejected attempts it needs (on C3 the first
    # iteration takes one attempt, the later ones two to three while the damping factor re-adapts): the timed region is
    # one optimise call of K iterations from the uploaded state.  Warm-up = W iterations of the same run, then a reset.
    if args.warmup > 0:
        run(args.warmup)
    ba.reset()
    barrier()
    t0 = time.perf_counter()
    ba.optimize(None, max_iterations=args.steps)
    barrier()
    dt = time.perf_counter() - t0
    iterations = ba.report.iterations
    attempts_timed = ba.report.attempts
    iter_log = ba.iteration_log()
    # the exchange schedule that actually ran: the first native damping-parallel round of a handle checks itself (checksums
    # beside the rooted collectives) and the handle falls back to the all-reduce schedule when that fails
    schedule_used = ba.multi_schedule() if world > 1 else "none"
    if world > 1 and exchange == "native RCCL":
        exchange = {"dp": "native RCCL, damping-parallel schedule (no native round ran)",
                    "dp (self-check passed)": "native RCCL, damping-parallel schedule: band k reduced to rank k, corrections "
                                              "broadcast, one all-reduce of the status words; first round self-checked",
                    "allreduce": "native RCCL all-reduce of the band on the attempt's stream (round-2 schedule)",
                    "allreduce (dp self-check failed)": "native RCCL all-reduce schedule AFTER the damping-parallel round failed its "
                                                        "self-check: " + ba.last_error()[:200]}[schedule_used]
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    err_final = ba.report.err_final
    err_initial = ba.report.err_initial

    # outside the timed region: the first iteration alone (one attempt on the synthetic scenes), for reference
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    r1 = step()
    torch.cuda.synchronize()
    first_iteration = {"ms": 1e3 * (time.perf_counter() - t1), "attempts": int(r1.attempts),
                       "err_after": r1.err_final}

    # outside the timed region: the same run again with the library's HIP-event instrumentation on (one event pair
    # per phase, one per MFMA trailing-update launch) -- every event costs a few microseconds on the stream, so the
    # phase times below add up to slightly more than ms_per_step
    ba.set_profile(2)
    run(1)
    r = run(args.steps)
    acc = {k: getattr(r, k) for k in ("ms_jacobian", "ms_schur", "ms_solve", "ms_backsub", "ms_apply", "ms_error",
                                      "ms_jacobian_kernel", "ms_solve_syrk", "solve_mfma_flops")}
    attempts = r.attempts
    prof_steps = max(int(r.iterations), 1)

    # outside the timed region: one more step with the reduced camera system treated as DENSE, so that every bench
    # line carries the fp64-MFMA trailing update at full size (the north-star's "dense RCS GEMM" evidence)
    dense_probe = None
    chain_probe = None
    if world == 1 and args.rcs != "dense" and not args.no_dense_probe:
        ba.set_rcs_mode(0)
        step()
        r = step()
        dense_probe = {"ms_solve": r.ms_solve / max(r.attempts, 1), "ms_trail": r.ms_solve_syrk / max(r.attempts, 1),
                       "flops": r.solve_mfma_flops / max(r.attempts, 1)}
        if rcs_chunks >= 2:  # the same skyline factorised as ONE panel chain, for the chunking gain
            ba.set_rcs_mode(1)
            step()
            r = step()
            chain_probe = {"ms_solve": r.ms_solve / max(r.attempts, 1),
                           "ms_trail": r.ms_solve_syrk / max(r.attempts, 1)}
        ba.set_rcs_mode(RCS_MODE[args.rcs])

    if rank == 0:
        K = max(iterations, 1)
        ms_per_step = 1e3 * dt / K
        ld = ((10 * M + 63) // 64) * 64
        ab = algorithmic_bytes(shard.N, M, shard.O, ld, rcs_fill)
        per_it = {k: v / prof_steps for k, v in acc.items()}
        per_attempt = {k: v / max(attempts, 1) for k, v in acc.items()}

        # HBM traffic per launch from the committed PMC passes of this configuration (profiles/, rocprofv3 --pmc in
        # separate FETCH_SIZE / WRITE_SIZE runs, gfx950 FETCH x2 correction); None when no profile matches
        pmc = {}
        pmc_source = None
        for rnd in ("r4", "r3", "r2", "r1"):
            pmc_path = os.path.join(ROOT, "profiles", rnd, f"pmc_{args.config}.json")
            if world == 1 and os.path.exists(pmc_path):
                try:
                    pmc = json.load(open(pmc_path))["kernels"]
                    pmc_source = (f"profiles/{rnd}/pmc_{args.config}.json: committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                  "passes of this command (gfx950 FETCH x2 correction), NOT measured in this run; a kernel "
                                  "that changed since keeps no entry")
                    break
                except Exception:
                    pmc = {}

        def traffic(*names):
            vals = [pmc[n]["hbm_bytes_per_launch"] for n in names if n in pmc]
            return sum(vals) if vals else None

        def hbm(bytes_, ms, *kernel_names):
            a = bytes_ / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            tr = traffic(*kernel_names)
            # frac: SURVEY 8(d)'s algorithmic bytes / time / peak (the contract's figure); frac_moved: the bytes the kernel
            # actually moves (PMC passes) / time / peak -- the layout stores 168 instead of 240 B per point-frame block, so
            # the first over-credits the memory system and the second is the honest bandwidth figure
            moved = (tr / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (tr is not None and ms > 0) else None
            return {"bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS,
                    "frac_moved": moved, "traffic": tr, "traffic_source": pmc_source if tr is not None else None, "ms": ms,
                    "algorithmic_bytes": bytes_}

        jac_names = {3: ("k_jac_runs",), 2: ("k_jac_runs",), 1: ("k_jac_fused",), 0: ("k_jac_points", "k_jac_frames")}.get(jac_kernel, ())
        # (fp64 storage keeps the point-frame blocks as their 21 rank-2 factors: the derivative pass WRITES and the Schur and
        # back-substitution passes READ 168 instead of SURVEY 8(d)'s 240 bytes per observation; `achieved` stays on the
        # algorithmic figure, `stored_bytes` says what the layout moves)
        w_stored = 84 if args.store_f32 else 168  # the 21 rank-2 factors of a point-frame block, as floats or doubles
        kernels = {
            "jacobian_phase": hbm(ab["jacobian"], per_it["ms_jacobian"], *jac_names),
            "jacobian_kernel": hbm(ab["jacobian"], per_it["ms_jacobian_kernel"], *jac_names),
            "schur_phase": hbm(ab["schur"], per_attempt["ms_schur"], "k_schur_mm", "k_schur_ws", "k_schur_grouped", "k_schur", "k_env_zero",
                               "k_assemble"),
            "backsub_phase": hbm(ab["backsub"], per_attempt["ms_backsub"], "k_backsub_obs", "k_point_update"),
            "error_phase": hbm(ab["error"], per_attempt["ms_error"], "k_error", "k_error_staged"),
        }
        kernels["jacobian_kernel"]["stored_bytes"] = ab["jacobian"] - shard.O * (240 - w_stored)
        kernels["backsub_phase"]["stored_bytes"] = ab["backsub"] - shard.O * (240 - w_stored)
        ms_syrk = per_attempt["ms_solve_syrk"]
        # flops actually executed by the MFMA trailing-update launches (= n^3/3 up to blocking when dense), counted by
        # the library while it issues them; their time is the sum of the HIP event pairs around those launches
        mfma_flops = per_attempt["solve_mfma_flops"]
        tf = mfma_flops / (ms_syrk * 1e-3) / 1e12 if ms_syrk > 0 else 0.0
        kernels["solve_syrk_mfma"] = {"bound": "mfma", "achieved": tf, "peak": FP64_MFMA_PEAK_TFLOPS,
                                      "unit": "TFLOP/s", "frac": tf / FP64_MFMA_PEAK_TFLOPS, "traffic": None,
                                      "ms": ms_syrk, "algorithmic_flops": mfma_flops,
                                      "dense_flops_n3_over_3": ab["solve_flops"],
                                      "ms_solve_phase": per_attempt["ms_solve"], "rcs_mode": args.rcs,
                                      "rcs_fill": rcs_fill, "rcs_chunks": rcs_chunks}
        if chain_probe:
            kernels["solve_single_chain_probe"] = {
                "bound": "latency", "ms_solve_phase": chain_probe["ms_solve"], "ms_trail": chain_probe["ms_trail"],
                "note": "one untimed step with the skyline factorised as one panel chain (--rcs skyline)"}
        if dense_probe and dense_probe["ms_trail"] > 0:
            tfd = dense_probe["flops"] / (dense_probe["ms_trail"] * 1e-3) / 1e12
            kernels["solve_trail_mfma_dense_probe"] = {
                "bound": "mfma", "achieved": tfd, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": tfd / FP64_MFMA_PEAK_TFLOPS, "traffic": None, "ms": dense_probe["ms_trail"],
                "algorithmic_flops": dense_probe["flops"], "ms_solve_phase": dense_probe["ms_solve"],
                "note": "one untimed step with the reduced camera system forced dense (--rcs dense gives the same)"}
        kernels["solve_panel_chain"] = {
            "bound": "latency", "ms": per_attempt["ms_solve"] - per_attempt["ms_solve_syrk"],
            "note": "outer-step kernels (k_step256: the four 64-column panels of a 256-column step as one launch; the "
                    "factorisation chain stays inside one workgroup) + backward substitution of the blocked Cholesky (+ gather / "
                    "reduce of the nested-dissection levels): a dependency chain of pivots, bound by per-pivot "
                    "latency, not by HBM or MFMA throughput"}
        # The Schur sum is compute-bound (22 flop per algorithmic byte against a machine balance of ~10): price it against
        # the fp64 MFMA peak.  Useful flops: every landmark's lower block triangle, nf (nf + 1) / 2 blocks of 100 entries,
        # 3 multiply-adds each (k_schur_mm runs them as v_mfma_f64_16x16x4 over the 16x16 tiles of a run's lower triangle:
        # 91 tiles for 20 frames = 11 % more than the useful entries).
        nf = np.diff(shard.row_ptr).astype(np.float64)
        schur_flops = float((nf * (nf + 1) / 2).sum() * 600.0)
        tfs = schur_flops / (per_attempt["ms_schur"] * 1e-3) / 1e12 if per_attempt["ms_schur"] > 0 else 0.0
        kernels["schur_kernel_fp64"] = {
            "bound": "mfma", "achieved": tfs, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": tfs / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic("k_schur_mm", "k_schur_ws", "k_schur_grouped", "k_schur"),
            "ms": per_attempt["ms_schur"], "algorithmic_flops": schur_flops, "algorithmic_bytes": ab["schur"],
            "note": "fp64 MFMA (v_mfma_f64_16x16x4: a run's landmark sum as a (10 nf)^2 x (3 np) matrix product); the time "
                    "is the Schur phase = this kernel + ~20 us of zeroing / assembly"}
        # The factorisation + substitution phase as a whole, priced twice: the flops its launches execute (MFMA trailing
        # and rank-64 updates as counted by the library while it issues them + n_sky pivots' worth of panel work, ~2 n b^2
        # for a band of half-width b -- the MFMA count is the dominant term and the one used) against the fp64 MFMA peak,
        # and the skyline of the system (read and written once) against HBM.  Neither roof is near: the phase is a chain
        # of dependent launches, which is exactly why it carries the largest share of the step.
        skyline_bytes = 2.0 * rcs_fill * 0.5 * (10 * M) ** 2 * 8
        ms_sol = per_attempt["ms_solve"]
        tf_sol = mfma_flops / (ms_sol * 1e-3) / 1e12 if ms_sol > 0 else 0.0
        kernels["solve_phase"] = {
            "bound": "mfma", "achieved": tf_sol, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": tf_sol / FP64_MFMA_PEAK_TFLOPS, "traffic": None, "ms": ms_sol, "algorithmic_flops": mfma_flops,
            "hbm_view": {"algorithmic_bytes": skyline_bytes,
                         "achieved_GBs": skyline_bytes / (ms_sol * 1e-3) / 1e9 if ms_sol > 0 else 0.0,
                         "frac_of_hbm_peak": (skyline_bytes / (ms_sol * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms_sol > 0 else 0.0},
            "ms_mfma_updates": ms_syrk, "ms_panel_chain": ms_sol - ms_syrk,
            "note": "whole solve phase of one attempt (all levels of the nested dissection: gather, fused outer steps, "
                    "trailing updates, separator reduction, backward substitution); executed MFMA flops / phase time"}
        # dominant = the largest share of the step among the kernels / kernel classes (the solve phase is one candidate)
        shares = {"jacobian_kernel": per_it["ms_jacobian_kernel"], "schur_kernel_fp64": per_it["ms_schur"],
                  "solve_phase": per_it["ms_solve"], "backsub_phase": per_it["ms_backsub"],
                  "error_phase": per_it["ms_error"]}
        dominant = max(shares, key=shares.get)
        roofline = dict(kernels[dominant])
        roofline["kernel"] = dominant
        roofline["share_of_step_ms"] = shares
        # converging phase: the iterations before the first one that needs more than three attempts (late in a run on the
        # synthetic scenes the error stagnates, the damping factor climbs through a dozen rejected attempts and the rest
        # are rounding-level ties: legitimate by the reference's control flow, but a different regime)
        att_log = [int(a) for a in iter_log["attempts"]]
        n_conv = next((k for k, a in enumerate(att_log) if a > 3), len(att_log))
        converging = None
        if n_conv > 0:
            ms_conv = float(iter_log["ms"][n_conv - 1])
            converging = {"iterations": n_conv, "attempts": int(sum(att_log[:n_conv])),
                          "iterations_per_s": n_conv / (ms_conv * 1e-3) if ms_conv > 0 else None,
                          "ms_per_iteration": ms_conv / n_conv,
                          "note": "iterations before the first one that needs > 3 attempts; host time from the call's start "
                                  "(includes the initial error evaluation)"}
        out = {
            "metric": "BA iterations/sec",
            "value": iterations / dt if dt > 0 else 0.0,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": ("f64" if not args.schur_fp32 else "f64 with fp32 Schur run sums (opt-in mixed precision)") +
                     (" / point-frame blocks stored as f32 (opt-in)" if args.store_f32 else ""),
            "data": "synthetic",
            "config": {"workload": (f"ragged tracks ({args.drop:.0%} of the observations dropped) of " if args.drop > 0 else "") +
                                   f"{args.config}: {M} cams / {N_total} pts / {O_total} obs (circle-grid, "
                                   f"{spec.vis_window}-frame visibility window, f0={spec.f0:g}); a step = one accepted "
                                   "outer LM iteration (with its rejected attempts) of one continuing run from the "
                                   "uploaded state",
                       "parallelism": f"landmark shards x{world}" if world > 1 else "single GPU",
                       "exchange": exchange,
                       "points_per_rank": shard.N, "obs_per_rank": shard.O, "rcs_dim": 10 * M - 7,
                       "rcs_solver": args.rcs, "rcs_fill": rcs_fill, "rcs_chunks": rcs_chunks,
                       "rcs_outer_step": "one launch per 256-column outer step: a diagonal-block workgroup owns the 256 x 256 block for "
                                         "all four sub-steps and hands tiles one way to helper / row workgroups "
                                         "(srk_ba_set_solver_fusion); solves repeated unfused after a hand-off timeout: "
                                         f"{ba.solver_sync_timeouts()}",
                       "lm_attempts": "one attempt at a time (--sequential-attempts)" if args.sequential_attempts else
                                      "two attempt slots: the next damping factor runs beside the current one and is "
                                      "judged in the reference's order (srk_ba_set_speculation)"},
            "deterministic_mode": bool(deterministic_on),
            "iterations_done": iterations,
            "attempts": int(attempts_timed),
            "attempts_by_iteration": att_log,
            "converging_phase": converging,
            "attempts_per_iteration": attempts_timed / max(iterations, 1),
            "attempts_per_s": attempts_timed / dt if dt > 0 else 0.0,
            "ms_per_attempt": 1e3 * dt / max(attempts_timed, 1),
            "first_iteration": first_iteration,
            "profiled_steps": prof_steps,
            "ms_per_iter": {"jacobian": per_it["ms_jacobian"], "schur": per_it["ms_schur"],
                            "solve": per_it["ms_solve"], "backsub": per_it["ms_backsub"],
                            "apply": per_it["ms_apply"], "error": per_it["ms_error"]},
            "err_initial": err_initial,
            "err_final": err_final,
            "roofline": roofline,
            "kernels": kernels,
            "solver_sync_timeouts": ba.solver_sync_timeouts(),
        }
        # the side measurements use handles of their own: the main handle goes first.  (HIP multiplexes a process's streams
        # onto a few hardware queues, GPU_MAX_HW_QUEUES = 4 by default; with the main handle's two streams still open the
        # probe handles' two attempt streams landed on ONE queue and their speculative pairs ran one after the other:
        # config 1's one call took 18.6 instead of 12.3 ms.)
        ba.close()
        ba = None
        if world == 1 and not args.no_one_call:
            try:
                out["one_call_ms"] = one_call_latency(sa)
            except Exception as e:  # noqa: BLE001
                out["one_call_ms"] = {"failed": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.steps)
            except Exception as e:  # the checker must never take the bench down
                out["cpu_baseline"] = {"value": None, "unit": "iterations/s", "cores": 1, "kind": "port",
                                       "sample": f"failed: {e!r}"}
        # The driver keeps the LAST stdout line: it stays under 4 KB.  Everything else (per-phase rooflines with their notes,
        # CPU-baseline variants, one-call latencies, probes) goes to bench_detail.json next to this file and to stderr.
        timeouts = out["solver_sync_timeouts"]

        def brief(k):
            e = kernels[k]
            b = {"frac": round(e["frac"], 4), "ms": round(e["ms"], 4), "bound": e["bound"]}
            if e.get("traffic") is not None:
                b["traffic"] = e["traffic"]   # HBM bytes per launch from the committed PMC passes (profiles/)
            if e.get("frac_moved") is not None:
                b["frac_moved"] = round(e["frac_moved"], 4)
            return b

        rl = {k: roofline.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "ms")}
        rl["algorithmic"] = roofline.get("algorithmic_flops", roofline.get("algorithmic_bytes"))
        line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
        line["config"] = {"workload": f"{args.config}: {M} cams / {N_total} pts / {O_total} obs" +
                                      (f", {args.drop:.0%} of the observations dropped" if args.drop > 0 else ""),
                          "step": "one accepted outer LM iteration with its rejected attempts, one continuing run",
                          "parallelism": out["config"]["parallelism"], "exchange": exchange, "rcs_solver": args.rcs,
                          "rcs_chunks": rcs_chunks,
                          "schedule": schedule_used, "deterministic": bool(deterministic_on),
                          "lm_attempts": ("sequential" if args.sequential_attempts and world == 1 else
                                          ("speculative pairs" if world == 1 or not schedule_used.startswith("dp") else
                                           f"{min(3, world)} damping factors a round, one per rank"))}
        line.update({"iterations_done": iterations, "attempts": int(attempts_timed),
                     "converging_phase": None if converging is None else {k: converging[k] for k in ("iterations", "attempts", "iterations_per_s")},
                     "attempts_per_s": out["attempts_per_s"],
                     "attempts_per_iteration": out["attempts_per_iteration"], "solver_sync_timeouts": timeouts,
                     "roofline": rl,
                     "kernels": {k: brief(k) for k in ("jacobian_kernel", "schur_kernel_fp64", "backsub_phase", "solve_phase")},
                     "ms_per_iter": {k: round(v, 4) for k, v in out["ms_per_iter"].items()},
                     "err_initial": err_initial, "err_final": err_final})
        cb = out.get("cpu_baseline")
        if cb is not None:
            line["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample", "seconds",
                                                           "attempts_per_s", "host_cpus", "allcore", "literal_qr_sample")}
            if isinstance(line["cpu_baseline"].get("literal_qr_sample"), dict):
                line["cpu_baseline"]["literal_qr_sample"].pop("note", None)
        line["detail"] = "bench_detail.json"
        try:
            with open(os.path.join(ROOT, "bench_detail.json"), "w") as f:
                json.dump(out, f, indent=1)
        except OSError as e:
            print("bench.py: bench_detail.json not written:", repr(e), file=sys.stderr, flush=True)
        print(json.dumps(out), file=sys.stderr, flush=True)
        text = json.dumps(line)
        # the line must fit whatever the optional fields hold: drop them, least important first, until it does
        for drop in (("cpu_baseline", "literal_qr_sample"), ("cpu_baseline", "allcore"), ("config", "exchange"), ("kernels",),
                     ("ms_per_iter",), ("cpu_baseline", "sample")):
            if len(text) < 4096:
                break
            tgt = line
            for k in drop[:-1]:
                tgt = tgt.get(k) if isinstance(tgt, dict) else None
            if isinstance(tgt, dict) and drop[-1] in tgt:
                if drop[-1] in ("sample", "exchange") and isinstance(tgt[drop[-1]], str):
                    tgt[drop[-1]] = tgt[drop[-1]][:120]
                else:
                    tgt.pop(drop[-1])
                text = json.dumps(line)
        print(text, flush=True)
    if ba is not None:
        ba.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
