#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ab
run() { tag=$1; shift; env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-one-call --no-dense-probe > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err; python - <<PY
import json
j=json.loads(open("gpurun_out/ab/$tag.json").read().strip().splitlines()[-1])
print("$tag", round(j["value"],1), "it/s", round(j["ms_per_step"],3), "ms/step att/it", j["attempts_per_iteration"], "solve", j["kernels"]["solve_phase"]["ms"])
PY
}
run two_a SRK_X=1
run one_a SRK_STEP_PAD_LDS=1
run two_b SRK_X=1
run one_b SRK_STEP_PAD_LDS=1
run two_c SRK_X=1
run one_c SRK_STEP_PAD_LDS=1
