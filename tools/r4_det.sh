#!/bin/bash
# deterministic mode on the GPU box: its tests, then the driver's bench line with and without it (three runs each)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/det
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "deterministic" > gpurun_out/det/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/det/tests.log
run() { tag=$1; shift; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-one-call --no-dense-probe "$@" > gpurun_out/det/$tag.json 2> gpurun_out/det/$tag.err; python - <<PY
import json
j=json.loads(open("gpurun_out/det/$tag.json").read().strip().splitlines()[-1])
print("$tag", round(j["value"],1), "it/s", round(j["ms_per_step"],3), "ms/step attempts", j["attempts"], "det", j.get("deterministic_mode", j["config"].get("deterministic")), "conv", j["converging_phase"], "kernels", {k:v["ms"] for k,v in j["kernels"].items()}, "err_final", repr(j["err_final"]))
PY
}
for k in 1 2 3; do run default_$k; run det_$k --deterministic; done
