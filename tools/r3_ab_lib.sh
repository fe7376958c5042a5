#!/bin/bash
# usage (GPU box): tools/r3_ab_lib.sh <variant.so> [runs]: the driver's bench line with the product library and with a variant
# build of it (loaded through SRK_BA_LIBRARY), interleaved
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ab
var=$1; n=${2:-3}
run() { tag=$1; shift; env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-one-call --no-dense-probe > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err; python - <<PY
import json
j=json.loads(open("gpurun_out/ab/$tag.json").read().strip().splitlines()[-1])
print("$tag", round(j["value"],1), "it/s", round(j["ms_per_step"],3), "ms/step att/it", j["attempts_per_iteration"], "solve", j["kernels"]["solve_phase"]["ms"])
PY
}
for k in $(seq 1 $n); do
  run product_$k SRK_X=1
  run variant_$k SRK_BA_LIBRARY=$PWD/$var
done
