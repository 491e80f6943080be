"""vq_amd: MI355X-native nearest-prototype assignment (the hot path of GerritLuimstra/vq).

The product is vq_amd/libvqhip.so (hand-written HIP for gfx950 behind include/vqhip.h).
This package is the ctypes handle over it plus the host-side mirror of the reference's
`predict` interface.  There is no CPU or PyTorch fallback: importing works without a GPU,
computing does not.
"""
from ._lib import (KERNEL_AUTO, KERNEL_MFMA, KERNEL_SCALAR, LIB_PATH, OPT_PROJECTION, OPT_RANKING, PROJECT_LAMBDA, PROJECT_OMEGA,  # noqa: F401
                   VALUE_BF16, VALUE_DEFAULT, VALUE_F32, VqhipError)
from .context import Context, MultiContext  # noqa: F401
