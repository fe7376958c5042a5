"""development: run the 3-rank sharded worker of tests/_dist_gpu_worker.py repeatedly and print every rank's outcome"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def main():
    import torch.multiprocessing as mp
    import _dist_gpu_worker
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    kw = dict(n_frames=400, grid_nx=60, grid_ny=40, vis_window=10, noise_uv_pix=0.2)
    for it in range(n):
        d = tempfile.mkdtemp()
        mp.spawn(_dist_gpu_worker.run, args=(3, 0, d, kw, 3), nprocs=3, join=True)
        z = [np.load(os.path.join(d, f"rank{r}.npz")) for r in range(3)]
        print(it, [(int(x["iterations"]), int(x["attempts"]), int(x["status"]), float(x["err_final"])) for x in z], flush=True)

if __name__ == "__main__":
    main()
