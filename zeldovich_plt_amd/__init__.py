"""zeldovich_plt_amd — MI355X-native grid->displacements path of zeldovich-PLT.

csrc/   hand-written HIP kernels (gfx950) + the C ABI declared in include/zeldovich_hip.h + the
        `zeldovich <param_file>` drop-in CLI
api.py  ctypes mirror of the reference's call sites (Parameters / PowerSpectrum / ZeldovichZ+XY)
parallel.py  one-process-per-GPU driver (torch.distributed all-to-all between the Z and XY stages)
"""
from .api import (ICFORMATS, KERNEL_NAMES, RECORD_DTYPES, Plan, PowerSpectrum, generate, load_library,
                  make_params, params_from_file)

__all__ = ["ICFORMATS", "KERNEL_NAMES", "RECORD_DTYPES", "Plan", "PowerSpectrum", "generate", "load_library",
           "make_params", "params_from_file"]
