#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_prof.sh <tag> [bench args...]
# rocprofv3 kernel trace + stats of bench.py; the summary is printed and kept under gpurun_out/prof_<tag>/
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python bench.py --no-cpu-baseline "$@" > $out/bench.json 2> $out/err.log
python tools/prof_summary.py $out 22
