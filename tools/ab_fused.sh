#!/bin/bash
# A/B of the fused outer-step kernel (k_step256) against the k_panel / k_upd64 launch sequence on the bench workload
python bench.py --no-cpu-baseline > gpurun_out/ab_bench_fused.json 2> gpurun_out/ab_bench_fused.err
SRK_CHOL_FUSED=0 python bench.py --no-cpu-baseline --no-one-call > gpurun_out/ab_bench_unfused.json 2> gpurun_out/ab_bench_unfused.err
python - <<PY
import json
for n in ("fused", "unfused"):
    d = json.load(open("gpurun_out/ab_bench_%s.json" % n))
    print(n, round(d["value"], 1), "it/s", round(d["ms_per_step"], 3), "ms/it", d["attempts_per_iteration"], "att/it",
          {k: round(v, 3) for k, v in d["ms_per_iter"].items()}, "solve/attempt", round(d["kernels"]["solve_phase"]["ms"], 4))
PY
