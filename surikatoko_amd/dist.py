"""Multi-GPU glue: one process per GPU, landmarks sharded, torch.distributed (backend "nccl" = RCCL over xGMI)
for the exchange steps of SURVEY 8e.

The C library calls back (srk_allreduce_fn) with a pointer + element count whenever a buffer has to be summed
across the landmark shards: twice per LM attempt -- the packed skyline of the assembled reduced camera system with
its right-hand side behind it, and the error scalar with the solver / point-update status flags.  The pointer is device memory on the GPU path
(wrapped zero-copy as a torch tensor through __cuda_array_interface__) and host memory in the CPU (gloo) tests.
"""
import ctypes as C

import numpy as np

from ._lib import ALLREDUCE_FN


class _DeviceArray:
    """Zero-copy view of `count` fp64 values at a raw device pointer (CUDA/HIP array interface v2)."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def tensor_from_pointer(ptr, count, device=None):
    import torch
    if device is None:  # host memory (gloo tests)
        buf = (C.c_double * int(count)).from_address(int(ptr))
        return torch.from_numpy(np.frombuffer(buf, dtype=np.float64, count=int(count)))
    return torch.as_tensor(_DeviceArray(ptr, count), device=device)


def make_allreduce_hook(group=None, device=None, chunk_elems=1 << 27):
    """Returns an ALLREDUCE_FN for BundleAdjustmentKanatani.set_allreduce.  device: torch device of this rank
    (e.g. "cuda:3") or None for host pointers.  Large buffers (the n^2 reduced camera system) are reduced in
    chunks of `chunk_elems` doubles (1 GiB) so that RCCL's staging stays bounded."""
    import torch
    import torch.distributed as dist

    views = {}  # (pointer, count) -> tensor view: the library exchanges the same few buffers every attempt

    def _view(ptr, n):
        key = (int(ptr), int(n))
        t = views.get(key)
        if t is None:
            if len(views) > 64:
                views.clear()
            t = views[key] = tensor_from_pointer(ptr, n, device)
        return t

    def _hook(_ctx, ptr, count):
        try:
            # a CPU-only backend (gloo) with device pointers: stage through host memory (used to rehearse the
            # multi-rank path with several processes on one GPU; RCCL reduces in place)
            stage = device is not None and dist.get_backend(group) == "gloo"
            done = 0
            while done < count:
                n = min(int(chunk_elems), int(count) - done)
                t = _view(int(ptr) + 8 * done, n)
                if stage:
                    th = t.cpu()
                    dist.all_reduce(th, op=dist.ReduceOp.SUM, group=group)
                    t.copy_(th)
                else:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                done += n
            if device is not None:
                torch.cuda.current_stream(torch.device(device)).synchronize()
            return 0
        except Exception as e:  # never let an exception cross the C ABI
            print("surikatoko_amd.dist: allreduce hook failed:", repr(e), flush=True)
            return 1

    return ALLREDUCE_FN(_hook)
