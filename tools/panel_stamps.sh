#!/bin/bash
# development: in-kernel clock stamps of k_panel (d = 40, last workgroup) on the GPU box
set -e
cd "$GRAFT_REPO_ROOT/surikatoko_amd/csrc"
# the variant is built to a temporary path and loaded through SRK_BA_LIBRARY: the product library stays untouched
export SRK_BA_LIBRARY=/tmp/libsrk_ba_variant.so
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DSRK_PANEL_STAMPS -c srk_chol.hip -o /tmp/chol_st.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$SRK_BA_LIBRARY" srk_ba_kernels.o /tmp/chol_st.o srk_ba_host.o srk_scene.o srk_io.o
(cd "$GRAFT_REPO_ROOT" && python - <<'PY'
import ctypes as C, numpy as np
import surikatoko_amd as sa
spec=sa.CONFIGS["C3_1kcam_100kpt"]; sc=sa.generate_scene(spec)
ba=sa.BundleAdjustmentKanatani(0); ba.set_rcs_mode(1); ba.upload(spec.f0, sc)
for _ in range(2):
    ba.reset(); ba.optimize(None, max_iterations=1)
out=(C.c_longlong*16)()
sa.lib().srk_dbg_panel_stamps(out)
t=[out[i] for i in range(7)]
names=["load diag","potrf64","store+fwd","row load","row sweep","row store"]
print("wall_clock64 ticks (100 MHz => x10 ns); note: the stamps themselves perturb the unrolled sweep")
for n,a,b in zip(names,t[:-1],t[1:]): print(f"  {n:12s} {b-a:8d} ticks = {(b-a)*10/1000:.2f} us")
print("  sweep quarters:", out[7]-out[4], out[8]-out[7], out[9]-out[8], out[5]-out[9])
print("  total", (t[-1]-t[0])*10/1000, "us")
print("inverse workgroup: load + potrf %.2f us, inverse %.2f us, total %.2f us" % ((out[11]-out[10])*10/1000, (out[12]-out[11])*10/1000, (out[12]-out[10])*10/1000))
PY
)
