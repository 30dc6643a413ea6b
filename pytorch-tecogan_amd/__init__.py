"""MI355X-native TecoGAN training hot path (generator + pseudo-flow/warp + spatio-temporal discriminator + losses +
Adam) behind the reference's Python surface.  The directory name is not a Python identifier, so the package is
imported as `pytorch_tecogan_amd` through the loader module of that name at the repository root."""
import os as _os

# The step runs as TWO lanes of launches on two HIP streams (step.TecoGANStep).  The HIP runtime maps streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4); once an RCCL process group exists (its own streams take queues) the two
# lane streams landed on ONE hardware queue with the default and ran serially - 7.14 instead of 4.52 ms per step with a
# process group that does not even issue a collective (profiles/r03_c_dp_hw_queues.log).  Every explicit value measured
# (2, 8, 16, 24) keeps them apart; 8 is set unless the user chose one.  Read by the runtime at its first HIP call, so this
# must run before anything touches the GPU: the package is imported first by main.py, bench.py and the code/ entry points.
if "GPU_MAX_HW_QUEUES" not in _os.environ:
    _os.environ["GPU_MAX_HW_QUEUES"] = "8"
    import sys as _sys
    _t = _sys.modules.get("torch")
    if _t is not None and _t.cuda.is_initialized():   # too late for this process: the runtime has read its flags
        import warnings as _w
        _w.warn("pytorch_tecogan_amd imported after the GPU runtime was initialised: GPU_MAX_HW_QUEUES keeps the runtime's default "
                "(4), with which the training step's two lanes can share one hardware queue beside an RCCL process group and "
                "run serially (~1.6x slower).  Import the package, or export GPU_MAX_HW_QUEUES=8, before the first CUDA call.")

from . import _lib  # noqa: E402,F401

__all__ = ["_lib"]
