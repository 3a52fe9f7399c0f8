"""MI355X-native STTN inpainting hot path behind the reference's plugin surface.

Layout
  csrc/      hand-written gfx950 kernels + host engine, built into lib/libvsr_hip.so
             (C-ABI: include/vsr_hip.h)
  _lib.py    ctypes binding of that C-ABI (fails loudly when the library is missing)
  engine.py  thin torch-tensor convenience layer over the C handle (device memory, streams)
  backend/   mirror of the reference's backend/ interface for this path: config values,
             InpaintMode, CLI arguments, mask helpers, STTNInpaint / STTNAutoInpaint plugins

The directory name follows the project name and is not a Python identifier; import it as
``vsr_amd`` (repo-root shim ``vsr_amd.py``).
"""
__version__ = "0.1.0"
