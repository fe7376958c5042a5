#!/bin/bash
# round-2 artefacts on the GPU box: bench lines, rocprofv3 kernel stats (speculative and sequential attempts), PMC passes
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2
python bench.py > gpurun_out/r2/bench_C3.json 2> gpurun_out/r2/bench_C3.err
bash tools/gpu_prof.sh r2_final --no-dense-probe --no-cpu-baseline --no-one-call > gpurun_out/r2/prof_final.txt 2>&1
bash tools/gpu_prof.sh r2_final_seq --sequential-attempts --no-dense-probe --no-cpu-baseline --no-one-call > gpurun_out/r2/prof_final_seq.txt 2>&1
bash tools/gpu_pmc.sh C3_1kcam_100kpt --sequential-attempts --no-dense-probe --no-one-call --steps 3 --warmup 1 > gpurun_out/r2/pmc_C3.txt 2>&1
for cfg in C1_dino_standin C2_200cam_20kpt C5_4kcam_1Mpt; do
  python bench.py --config $cfg --no-cpu-baseline --no-one-call --no-dense-probe > gpurun_out/r2/bench_$cfg.json 2> gpurun_out/r2/bench_$cfg.err
done
python bench.py --drop 0.1 --no-cpu-baseline --no-one-call --no-dense-probe > gpurun_out/r2/bench_C3_drop10.json 2> gpurun_out/r2/bench_C3_drop10.err
python tools/dbg_long_tracks.py > gpurun_out/r2/long_tracks.txt 2>&1
SRK_SCHUR_NO_LONG=1 python tools/dbg_long_tracks.py > gpurun_out/r2/long_tracks_per_landmark_kernel.txt 2>&1
SRK_CHOL_FUSED=0 python bench.py --no-cpu-baseline --no-one-call --no-dense-probe > gpurun_out/r2/bench_C3_unfused_solve.json 2> gpurun_out/r2/bench_C3_unfused_solve.err
bash tools/step_stamps.sh > gpurun_out/r2/step_stamps.txt 2>&1 || true
echo collected
