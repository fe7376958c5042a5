"""surikatoko_amd -- MI355X-native bundle-adjustment core (drop-in for suriko's BundleAdjustmentKanatani).

Only the hot path lives here: csrc/ (HIP kernels + the C ABI of include/srk_ba.h) and a thin Python
mirror of the reference's operator interface (ba.py) used by the tests and bench.py.  There is no CPU
fallback: every compute entry point needs libsrk_ba.so and a HIP device.
"""
from ._lib import lib, library_path, build_library, LibraryNotBuilt  # noqa: F401
from .ba import (BundleAdjustmentKanatani, BundleAdjustmentKanataniTermCriteria, Report, Scene,  # noqa: F401
                 normalize_scene_inplace, check_world_is_normalized, status_string, device_count)
from .scene import (SceneSpec, generate_scene, config_scene, drop_observations, renumber_frames, loop_scene,  # noqa: F401
                    CONFIGS)
