#!/bin/bash
# usage (GPU box, repo root): tools/r3_eval.sh <tag> [pytest -k expression | "all" | "none"]
# round-3 evaluation of one state of the code: parity tests, the driver's bench line, rocprofv3 kernel stats of a
# sequential-attempts run (per-kernel durations); everything under gpurun_out/<tag>/
tag=$1; sel=${2:-all}
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
if [ "$sel" = "all" ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -4 $out/tests.log
elif [ "$sel" != "none" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$sel" > $out/tests.log 2>&1; echo "tests rc=$?"; tail -4 $out/tests.log
fi
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-one-call --no-dense-probe > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - <<PY
import json
j=json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
print("value", round(j["value"],1), "it/s  ms/step", round(j["ms_per_step"],3), "att/it", j["attempts_per_iteration"], {k:(v["ms"],v["frac"]) for k,v in j["kernels"].items()})
PY
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_seq -- python bench.py --sequential-attempts --steps 20 --warmup 5 --no-cpu-baseline --no-one-call --no-dense-probe > $out/bench_seq.json 2> $out/bench_seq.err
python tools/prof_summary.py $out/prof_seq 24 | tee $out/prof_seq_summary.txt
