#!/bin/bash
# round-3 artefacts on the GPU box: bench lines, rocprofv3 kernel stats (speculative and sequential attempts), PMC passes
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
python bench.py --steps 20 --warmup 5 > gpurun_out/r4/bench_C3.json 2> gpurun_out/r4/bench_C3.err; cp bench_detail.json gpurun_out/r4/bench_detail_C3.json
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-one-call --no-dense-probe > gpurun_out/r4/bench_C3_K10.json 2> /dev/null
bash tools/gpu_prof.sh r4_final --steps 20 --warmup 5 --no-dense-probe --no-cpu-baseline --no-one-call > gpurun_out/r4/prof_final.txt 2>&1
bash tools/gpu_prof.sh r4_final_seq --sequential-attempts --steps 20 --warmup 5 --no-dense-probe --no-cpu-baseline --no-one-call > gpurun_out/r4/prof_final_seq.txt 2>&1
bash tools/gpu_pmc.sh C3_1kcam_100kpt --sequential-attempts --no-dense-probe --no-one-call --steps 3 --warmup 1 > gpurun_out/r4/pmc_C3.txt 2>&1
for cfg in C1_dino_standin C2_200cam_20kpt C5_4kcam_1Mpt; do
  python bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline --no-one-call --no-dense-probe > gpurun_out/r4/bench_$cfg.json 2> gpurun_out/r4/bench_$cfg.err
done
python bench.py --drop 0.1 --steps 10 --warmup 2 --no-cpu-baseline --no-one-call --no-dense-probe > gpurun_out/r4/bench_C3_drop10.json 2> gpurun_out/r4/bench_C3_drop10.err
python bench.py --store-f32 --steps 10 --warmup 2 --no-cpu-baseline --no-one-call --no-dense-probe > gpurun_out/r4/bench_C3_store_f32.json 2> gpurun_out/r4/bench_C3_store_f32.err
python bench.py --deterministic --steps 20 --warmup 5 --no-cpu-baseline --no-one-call --no-dense-probe > gpurun_out/r4/bench_C3_deterministic.json 2> gpurun_out/r4/bench_C3_deterministic.err
python bench.py --config C2_all_visible --steps 5 --warmup 1 --no-cpu-baseline --no-one-call --no-dense-probe > gpurun_out/r4/bench_C2_all_visible.json 2> gpurun_out/r4/bench_C2_all_visible.err
python tools/dbg_long_tracks.py > gpurun_out/r4/long_tracks.txt 2>&1
SRK_BA_LIBRARY=$PWD/surikatoko_amd/libsrk_ba_dev.so python tools/run_len_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4/run_len_probe.txt
timeout -k 10 300 bash tools/step_stamps.sh > gpurun_out/r4/step_stamps.txt 2>&1
for f in gpurun_out/r4/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], round(j["value"],1), "it/s", round(j["ms_per_step"],3), "ms", j.get("attempts_per_iteration"), {k:(v["ms"],v["frac"]) for k,v in j.get("kernels",{}).items()} if "kernels" in j else "")
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
done
echo collected
