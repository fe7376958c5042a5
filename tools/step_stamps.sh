#!/bin/bash
# development: in-kernel clock stamps of k_step256 (item 0, first outer step of the last solve) on the GPU box
set -e
cd "$GRAFT_REPO_ROOT/surikatoko_amd/csrc"
export SRK_BA_LIBRARY=/tmp/libsrk_ba_variant.so
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DSRK_STEP_STAMPS -c srk_chol.hip -o /tmp/chol_st.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$SRK_BA_LIBRARY" srk_ba_kernels.o /tmp/chol_st.o srk_ba_host.o srk_scene.o srk_io.o -ldl
(cd "$GRAFT_REPO_ROOT" && python - <<'PY'
import ctypes as C, numpy as np
import surikatoko_amd as sa
spec=sa.CONFIGS["C3_1kcam_100kpt"]; sc=sa.generate_scene(spec)
ba=sa.BundleAdjustmentKanatani(0); ba.set_speculation(False); ba.upload(spec.f0, sc)
for _ in range(2):
    ba.reset(); ba.optimize(None, max_iterations=1)
out=(C.c_longlong*256)()
sa.lib().srk_dbg_step_stamps(out)
st=np.array(list(out)).reshape(8,32)
t0=st[:, 0].min()
names={0:"start",25:"end"}
for d in range(4):
    names.update({1+6*d:f"d{d} begin",2+6*d:f"d{d} L here/potrf",3+6*d:f"d{d} L loaded/F pub",4+6*d:f"d{d} swept/Y pub",5+6*d:f"d{d} X pub",6+6*d:f"d{d} need here"})
print("k_step256, item 0, outer step 0 (LAST launch with K == 0 = last level of the last solve); us since the first workgroup's start")
for r in range(8):
    row=[(k, (st[r,k]-t0)/100.0) for k in range(26) if st[r,k]>0]
    print("role", r, " ".join(f"[{names.get(k,k)}] {v:.1f}" for k,v in row))
PY
)
