#!/bin/bash
# copies what tools/r4_collect.sh left under gpurun_out/ into profiles/r4/ (run in the build container after the gpurun call)
cd "$(dirname "$0")/.."
P=profiles/r4
mkdir -p $P
cp gpurun_out/prof_r4_final/bench.json $P/C3_bench_under_rocprof.json
cp "$(ls -t gpurun_out/prof_r4_final/runc/*_kernel_stats.csv | head -1)" $P/C3_kernel_stats.csv
cp gpurun_out/prof_r4_final_seq/bench.json $P/C3_sequential_attempts_bench_under_rocprof.json
cp "$(ls -t gpurun_out/prof_r4_final_seq/runc/*_kernel_stats.csv | head -1)" $P/C3_sequential_attempts_kernel_stats.csv
cp gpurun_out/r4/bench_C3.json $P/bench_C3_1kcam_100kpt.json
cp gpurun_out/r4/bench_detail_C3.json $P/bench_detail_C3_1kcam_100kpt.json
cp gpurun_out/r4/bench_C3_K10.json $P/bench_C3_1kcam_100kpt_K10.json
for c in C1_dino_standin C2_200cam_20kpt C5_4kcam_1Mpt; do cp gpurun_out/r4/bench_$c.json $P/bench_$c.json; done
cp gpurun_out/r4/bench_C3_drop10.json $P/bench_C3_drop10.json
cp gpurun_out/r4/bench_C3_store_f32.json $P/bench_C3_store_f32.json
cp gpurun_out/pmc_C3_1kcam_100kpt.json $P/pmc_C3_1kcam_100kpt.json
cp gpurun_out/r4/pmc_C3.txt $P/pmc_C3_1kcam_100kpt.txt
cp gpurun_out/r4/bench_C3_deterministic.json $P/bench_C3_deterministic.json
cp gpurun_out/r4/bench_C2_all_visible.json $P/bench_C2_all_visible.json
grep -v amdgpu.ids gpurun_out/r4/long_tracks.txt > $P/long_tracks.txt
cp gpurun_out/r4/run_len_probe.txt $P/run_len_probe.txt
grep -A12 "^k_step256" gpurun_out/r4/step_stamps.txt > $P/step_stamps.txt
ls -la $P
