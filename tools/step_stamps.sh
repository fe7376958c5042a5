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
t0=st[:, 0][st[:, 0] > 0].min()
rows={0:"diag chain",1:"diag passengers",2:"inverse",3:"inverse (idle half)",4:"forward",5:"forward (idle half)",6:"helper (2,2)",7:"helper (2,2) (waves 4-7)"}
diag={0:"start",1:"potrf/passengers done",2:"F published",3:"Y published",4:"drained, F set",5:"update done"}
roww={0:"start",1:"begin",2:"flags seen",3:"L loaded",4:"swept",5:"updates done"}
print("k_step256, item 0, outer step 0 (LAST launch with K == 0 = last level of the last solve); us since the first workgroup's start")
for r in range(8):
    names = diag if r < 2 else roww
    row=sorted([(k, (st[r,k]-t0)/100.0) for k in range(32) if st[r,k]>0], key=lambda x: x[1])
    def nm(k):
        if k == 0: return "start"
        if k == 25: return "end"
        if r < 2 and 26 <= k <= 28: return "d%d stores issued" % (k - 26)
        if r < 2 and 29 <= k <= 31: return "d%d G seen, C loads issued" % (k - 29)
        if r < 2 and k % 6 == 0: return "d%d mfma issued" % (k // 6 - 1)
        return "d%d %s" % ((k-1)//6, names.get((k-1)%6+1, k))
    print(rows[r], " ".join(f"[{nm(k)}] {v:.1f}" for k,v in row))
PY
)
