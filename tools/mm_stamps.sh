#!/bin/bash
# development: in-kernel clock stamps of k_schur_mm (one multiplier and one helper lane of every workgroup) on the GPU box
set -e
cd "$GRAFT_REPO_ROOT/surikatoko_amd/csrc"
# the variant is built to a temporary path and loaded through SRK_BA_LIBRARY: the product library stays untouched
export SRK_BA_LIBRARY=/tmp/libsrk_ba_variant.so
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DSRK_MM_STAMPS $EXTRA -c srk_ba_kernels.hip -o /tmp/k_st.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$SRK_BA_LIBRARY" /tmp/k_st.o srk_chol.o srk_ba_host.o srk_scene.o srk_io.o
(cd "$GRAFT_REPO_ROOT" && python - <<'PY'
import ctypes as C, numpy as np
import surikatoko_amd as sa
spec=sa.CONFIGS["C3_1kcam_100kpt"]; sc=sa.generate_scene(spec)
ba=sa.BundleAdjustmentKanatani(0); ba.upload(spec.f0, sc)
ba.phase_error(); ba.phase_derivatives()
for _ in range(5): ba.phase_schur(1e-4)
out=(C.c_longlong*32768)()
sa.lib().srk_dbg_mm_stamps(out)
t=np.array(out[:]).reshape(2048,16).astype(np.float64); t=t[t[:,0]>0]
us=lambda x: x*10/1000
t0=t[:,0].min()
print("workgroups", len(t)); print("kernel span %.1f us (first start to last end)" % us(t[:,4].max()-t0))
print("start times: median %.1f us, 75%% %.1f, max %.1f" % tuple(us(np.percentile(t[:,0]-t0,[50,75,100]))))
dur=t[:,4]-t[:,0]
print("per workgroup (median / max, us): total %.1f / %.1f" % (us(np.median(dur)), us(dur.max())))
for name,a in (("prologue (E inverse, W of rounds 0-1)",t[:,1]-t[:,0]),("Y of round 0",t[:,2]-t[:,1]),("rounds",t[:,3]-t[:,2]),
               ("  Y of the next round",t[:,5]),("  multiply",t[:,6]),("  barrier",t[:,7]),("flush",t[:,4]-t[:,3])):
    print("  %-22s %7.2f / %7.2f" % (name, us(np.median(a)), us(a.max())))
for name,a in (("loader: stage",t[:,10]),("loader: Y of next round",t[:,13]),("loader: issue loads",t[:,11]),("loader: barrier wait",t[:,12])):
    print("  %-24s %7.2f / %7.2f" % (name, us(np.median(a)), us(a.max())))
print("core clock over the workgroup: median %.2f GHz" % np.median(t[:,9]/(dur*10)))
PY
)
