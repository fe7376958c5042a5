#!/bin/bash
# development: rebuild libsrk_ba.so with one ablation macro at a time on the GPU box and time the derivative kernel
# (HIP events around 20 launches of the derivatives phase; SRK_JR_* = k_jac_runs without its W stores / frame sums /
# landmark sums: wrong results by design)
set -e
cd "$GRAFT_REPO_ROOT/surikatoko_amd/csrc"
# the variant is built to a temporary path and loaded through SRK_BA_LIBRARY: the product library stays untouched
export SRK_BA_LIBRARY=/tmp/libsrk_ba_variant.so
for abl in ${ABLS:-NONE}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -D${abl//+/ -D} -c srk_ba_kernels.hip -o /tmp/k_abl.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$SRK_BA_LIBRARY" /tmp/k_abl.o srk_chol.o srk_ba_host.o srk_scene.o srk_io.o
  (cd "$GRAFT_REPO_ROOT" && python - <<PY
import surikatoko_amd as sa, time, torch
spec=sa.CONFIGS["${CONFIG:-C3_1kcam_100kpt}"]; sc=sa.config_scene("${CONFIG:-C3_1kcam_100kpt}")
ba=sa.BundleAdjustmentKanatani(0); ba.upload(spec.f0, sc)
for _ in range(3): ba.phase_derivatives()
t=time.perf_counter()
for _ in range(20): ba.phase_derivatives()
print("$abl", "kernel", ba.jacobian_kernel(), (time.perf_counter()-t)/20*1e6, "us per derivatives phase (incl. memsets+sync)")
PY
)
done
