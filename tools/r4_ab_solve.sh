#!/bin/bash
# usage (GPU box): tools/r4_ab_solve.sh <variant.so> : solve-phase A/B of the product library against a variant, plus bitwise
# comparison of their dense solutions
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ab
var=$1
timeout -k 10 300 python tools/r4_ab_solve.py product > gpurun_out/ab/solve_product.log 2>&1; echo "product rc=$?"; grep -v "^$" gpurun_out/ab/solve_product.log | tail -8
SRK_BA_LIBRARY=$PWD/$var timeout -k 10 300 python tools/r4_ab_solve.py variant > gpurun_out/ab/solve_variant.log 2>&1; echo "variant rc=$?"; tail -8 gpurun_out/ab/solve_variant.log
python - <<PY
import numpy as np
a=np.load("gpurun_out/ab/product_x.npy"); b=np.load("gpurun_out/ab/variant_x.npy")
print("dense solutions bit-identical:", np.array_equal(a,b), "max abs diff", np.abs(a-b).max())
PY
