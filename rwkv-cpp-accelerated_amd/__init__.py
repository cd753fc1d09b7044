"""rwkv-cpp-accelerated_amd -- MI355X-native (gfx950) RWKV-v4 uint8 inference engine.

The product is csrc/librwkv_mi355x.so (C-ABI: include/rwkv_mi355x.h), written in HIP/C++.
This Python package only holds what the hot path needs on the host side:
  build      -- compile the HIP extension (+ the pybind module) for gfx950
  modelfile  -- the converter's model.bin format + seeded synthetic models
  engine     -- ctypes binding and a mirror of the reference's `RWKV` / `RWKVState` classes
  converter  -- .pth / state-dict -> model.bin (numpy only)
  pipeline   -- layer pipeline across the GPUs of one node
(the reference's pybind module `rwkv` is csrc/pybind_module.cpp, built next to the engine)

The directory name has a hyphen (it mirrors the reference repo's name); import it through the
`rwkv_cpp_accelerated_amd` shim at the repository root.
"""
from . import build, modelfile  # noqa: F401

__all__ = ["build", "modelfile", "engine", "converter", "pipeline"]
