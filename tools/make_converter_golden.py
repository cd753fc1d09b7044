#!/usr/bin/env python3
"""Run the REFERENCE converter (/root/reference/converter/convert_model.py + cpp_save_tensor.cpp, imported
where they lie; builds its torch C++ extension once, ~40 s) on a seeded synthetic checkpoint and record,
per tensor of the model.bin it writes, a sha256 and -- for the small float tensors -- the values.
Only runs in the authoring container (needs /root/reference).  Output: tests/golden/converter_L2_D64.npz, or with
`169M` as argument tests/golden/converter_169M.npz (BASELINE config 1's shape, L=12, D=768: hashes of every tensor, values
of the eight offset vectors only -- the one place where torch's and numpy's reductions may round differently)."""
import hashlib, os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rwkv_cpp_accelerated_amd import converter, modelfile as mf
REF = "/root/reference/converter"
sys.path.insert(0, REF)
import convert_model as ref                                   # the reference's module, unmodified
from torch.utils.cpp_extension import load as torch_load_cpp

L, D, SEED, NAME = 2, 64, 4242, "L2_D64"
if len(sys.argv) > 1 and sys.argv[1] == "169M":
    L, D, SEED, NAME = 12, 768, 169, "169M"
w = {k: torch.from_numpy(v) for k, v in converter.synthetic_state_dict(L, D, SEED).items()}
td = tempfile.mkdtemp()
torch_load_cpp(name="wkv_cuda_export", sources=[os.path.join(REF, "cpp_save_tensor.cpp")],
               extra_include_paths=[os.path.join(REF, "../include/")], build_directory=td)
ref.current_path = td                                          # the script writes <current_path>/model.bin (convert_model.py:127)
inst = ref.ConvertRWKV(w, D, L)
inst.save_converted()
a, b, tens = mf.read_bin(os.path.join(td, "model.bin"), mmap=False)
assert (a, b) == (L, D)
out = dict(L=L, D=D, seed=SEED, file_bytes=os.path.getsize(os.path.join(td, "model.bin")))
for i, t in enumerate(tens):
    out[f"sha_{i}"] = hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest()
    if NAME == "L2_D64":
        if t.dtype != np.uint8 and t.size <= 4096 * 8:
            out[f"val_{i}"] = t
    elif i in (mf.O1, mf.O2, mf.O3, mf.ATTOUTO, mf.FFNKO, mf.FFNVO, mf.FFNRO, mf.HEADO):
        out[f"val_{i}"] = t
np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"converter_{NAME}.npz"), **out)
print(f"wrote tests/golden/converter_{NAME}.npz", out["file_bytes"])
