#!/bin/bash
# usage (GPU box, repo root): tools/gpu_pmc.sh <tag> [bench args]  -- two separate PMC passes (FETCH_SIZE, WRITE_SIZE)
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  out=gpurun_out/pmc_${tag}_$c
  mkdir -p $out
  timeout -k 10 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -- python bench.py --no-cpu-baseline "$@" > $out/bench.json 2> $out/err.log || { tail -5 $out/err.log; exit 1; }
done
python tools/pmc_summary.py gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE gpurun_out/pmc_${tag}.json
