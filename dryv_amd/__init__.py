"""dryv_amd — MI355X (gfx950) macroblock-reconstruction backend for dryv's AVC intra decode path.

Only what the path needs: csrc/ (HIP kernels + the extern "C" boundary declared in
include/dryv_recon.h), the host-side mirror of the reference's `Frame` interface (frame.py,
host/frame.hpp) and the synthetic batch generator used to exercise it (synth.py).
"""
from .abi import (DRYV_OK, DRYV_E_INVALID, DRYV_E_UNSUPPORTED, DRYV_E_DEVICE, DRYV_E_NOMEM,  # noqa: F401
                  DRYV_E_STATE, DRYV_E_NODEVICE, FrameParams, MbDesc, MB_DESC_DTYPE, COEFFS_PER_MB,
                  load_library, make_frame_params, strerror)
from .frame import Frame, ReconContext, ReconError  # noqa: F401
