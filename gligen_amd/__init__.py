"""gligen_amd — MI355X (gfx950) native GLIGEN denoising path.

Hand-written HIP kernels + a C++ engine behind a C ABI (include/gligen_amd.h), bound from
Python with ctypes. The reference's entry points (ldm.*, grounding_input.*, gligen_inference)
live at the repo root and route their forward passes through this package.
"""
from ._lib import GligenAmdError, LIB_PATH  # noqa: F401

__all__ = ["GligenAmdError", "LIB_PATH"]
