#!/bin/bash
# usage (GPU box): tools/r3_ab_ragged.sh <variant.so>: bench lines of the ragged scenes (C3 with 10 % of the observations
# dropped, the dino stand-in C1) with the product library and a variant build, interleaved
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ab
var=$1
run() { tag=$1; shift; env "$@" > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err; python - <<PY
import json
j=json.loads(open("gpurun_out/ab/$tag.json").read().strip().splitlines()[-1])
print("$tag", round(j["value"],1), "it/s", round(j["ms_per_step"],3), "ms/step att/it", j["attempts_per_iteration"], "schur", j["kernels"]["schur_kernel_fp64"]["ms"])
PY
}
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-one-call --no-dense-probe"
for k in 1 2; do
  run drop_product_$k SRK_X=1 $B --drop 0.1
  run drop_variant_$k SRK_BA_LIBRARY=$PWD/$var $B --drop 0.1
  run c1_product_$k SRK_X=1 $B --config C1_dino_standin
  run c1_variant_$k SRK_BA_LIBRARY=$PWD/$var $B --config C1_dino_standin
done
