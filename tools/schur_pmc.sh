#!/bin/bash
# development (GPU box, repo root): SQ counters of the Schur kernels over a few Schur phases of C3
# usage: [SRK_PMC_CONFIG=C2_all_visible] tools/schur_pmc.sh "<counter list 1>" "<counter list 2>" ...   (one rocprofv3 pass per list)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cat > tools/_schur_only.py <<'PY'
import sys, os; sys.path.insert(0, os.getcwd())
import surikatoko_amd as sa
spec=sa.CONFIGS[os.environ.get("SRK_PMC_CONFIG", "C3_1kcam_100kpt")]; sc=sa.generate_scene(spec)
ba=sa.BundleAdjustmentKanatani(0); ba.upload(spec.f0, sc)
ba.phase_error(); ba.phase_derivatives()
for _ in range(4): ba.phase_schur(1e-4)
PY
i=0
for c in "$@"; do
  out=gpurun_out/schur_pmc_$i; rm -rf $out; mkdir -p $out
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -- python tools/_schur_only.py > $out/out.log 2> $out/err.log || { tail -5 $out/err.log; exit 1; }
  python - "$out" <<'PY'
import sys, glob, csv, collections
f=glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"]
    if "schur" not in k: continue
    k=k.split("(")[0][:40]
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
for k,v in acc.items():
    print(k, {c: "%.4g" % (x/n[(k,c)]) for c,x in v.items()}, "(mean per launch)")
PY
  i=$((i+1))
done
