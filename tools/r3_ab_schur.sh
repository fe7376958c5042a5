#!/bin/bash
# usage (GPU box): tools/r3_ab_schur.sh <variant.so> [runs]: the Schur phase of C3 (40 phases, incl. zeroing / assembly / sync)
# with the product library and with a variant build of it (SRK_BA_LIBRARY), interleaved
cd "$GRAFT_REPO_ROOT"
var=$1; n=${2:-3}
one() { python - <<PY
import surikatoko_amd as sa, time
spec=sa.CONFIGS["C3_1kcam_100kpt"]; sc=sa.generate_scene(spec)
ba=sa.BundleAdjustmentKanatani(0); ba.upload(spec.f0, sc)
ba.phase_error(); ba.phase_derivatives()
for _ in range(5): ba.phase_schur(1e-4)
t=time.perf_counter()
for _ in range(40): ba.phase_schur(1e-4)
print("$1", round((time.perf_counter()-t)/40*1e6, 1), "us per Schur phase")
PY
}
for k in $(seq 1 $n); do one product; SRK_BA_LIBRARY=$PWD/$var one variant; done
