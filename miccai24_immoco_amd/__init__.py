"""miccai24_immoco_amd — MI355X-native IM-MoCo inner optimisation loop.

Host-side mirror of the reference interface for ONE hot path
(src/models/immoco.py, src/utils/{data_utils,losses,motion_utils}.py of
multimodallearning/MICCAI24_IMMoCo) over hand-written HIP kernels reached
through the C-ABI library ``csrc/libimmoco_hip.so`` (include/immoco_hip.h).
There is no CPU or PyTorch fallback: every operator raises if the library is
missing or a tensor is not on the GPU.
"""
from . import _lib  # noqa: F401
from .models.immoco import (IMMoCo, encoding_config, imcoco_motion_correction,  # noqa: F401
                            imcoco_motion_correction_batch, make_grids,
                            mot_network_config, network_config)
from .models.autofocusing import Autofocusing  # noqa: F401
from .models.kld_net import get_unet  # noqa: F401
from .pipeline import correct_slice, evaluate_slices  # noqa: F401
from .utils.evaluate import calmetric2D  # noqa: F401
from .tcnn import NetworkWithInputEncoding  # noqa: F401
from .utils.data_utils import FFT, IFFT  # noqa: F401
from .utils.losses import GradientEntropyLoss  # noqa: F401
from .utils.motion_utils import extract_movement_groups  # noqa: F401

__version__ = "0.1.0"
