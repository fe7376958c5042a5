import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os
sys_path_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
order = sys.argv[1]
if order == "lib":
    import surikatoko_amd as sa
    h = sa.BundleAdjustmentKanatani(0)
    import torch
    print("after lib:", torch.cuda.is_available(), torch.cuda.device_count())
else:
    import torch
    print("torch first:", torch.cuda.is_available(), torch.cuda.device_count())
    import surikatoko_amd as sa
    h = sa.BundleAdjustmentKanatani(0)
    print("ok")
