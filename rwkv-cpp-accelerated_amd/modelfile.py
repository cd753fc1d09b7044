"""model.bin -- the reference converter's on-disk format, and synthetic models in that format.

Format contract (reference include/rwkv/rwkv/rwkv.h:10-56,84,124-128,
include/rwkv/enums/enum.h:7-55, converter/cpp_save_tensor.cpp:75-95, loader rwkv.cu:649-711):
little-endian, no padding: u64 n_layers, u64 n_embed, then 46 raw tensors in enum order.

This module is host-side plumbing (numpy / optional torch): tensor table, reader, writer, and
the seeded synthetic-checkpoint generator used by tests and bench.py (there is no network to
fetch real checkpoints).  It is not on the hot path.
"""
from __future__ import annotations

import numpy as np

VOCAB = 50277
N_TENSORS = 46

# reference rwkv.h:10-56
NAMES = [
    "xbuf", "embed", "layernorms", "state_xy", "state_aa", "state_bb", "state_pp", "state_dd",
    "buffer1", "buffer2", "buffer3", "buffer4", "mix_k", "mix_v", "mix_r", "km", "vm", "rm",
    "kr", "vr", "rr", "o1", "o2", "o3", "att_out", "att_out_r", "att_out_o", "ffn_mix_k",
    "ffn_mix_v", "ffn_k", "ffn_v", "ffn_r", "ffn_kr", "ffn_vr", "ffn_rr", "ffn_ko", "ffn_vo",
    "ffn_ro", "ffn_k_buffer", "ffn_v_buffer", "ffn_r_buffer", "decay", "bonus", "head", "head_r",
    "head_o",
]
(X, EMBED, LAYERNORMS, STATEXY, STATEAA, STATEBB, STATEPP, STATEDD, BUFFER1, BUFFER2, BUFFER3,
 BUFFER4, MIXK, MIXV, MIXR, KM, VM, RM, KR, VR, RR, O1, O2, O3, ATTOUT, ATTOUTR, ATTOUTO, FFNMIXK,
 FFNMIXV, FFNK, FFNV, FFNR, FFNKR, FFNVR, FFNRR, FFNKO, FFNVO, FFNRO, FFNKBUFFER, FFNVBUFFER,
 FFNRBUFFER, DECAY, BONUS, HEAD, HEADR, HEADO) = range(N_TENSORS)

_D, _F, _G = np.float64, np.float32, np.uint8
# reference rwkv.h:84
DTYPES = [_D, _F, _D, _D, _D, _D, _D, _D, _D, _F, _F, _F, _D, _D, _D, _G, _G, _G, _F, _F, _F, _F,
          _F, _F, _G, _F, _F, _D, _D, _G, _G, _G, _F, _F, _F, _F, _F, _F, _D, _D, _F, _D, _D, _G,
          _F, _F]
# slots that hold scratch / state, not weights (rwkv.cu:656-670)
BUFFER_SLOTS = (X, STATEXY, STATEAA, STATEBB, STATEPP, STATEDD, BUFFER1, BUFFER2, BUFFER3, BUFFER4,
                FFNKBUFFER, FFNVBUFFER, FFNRBUFFER)


def sizes(a: int, b: int) -> list[int]:
    """element counts of the 46 tensors (reference rwkv.h:124-128; a = n_layers, b = n_embed)."""
    V = VOCAB
    return [b, V * b, 4 * (a + 1) * b, a * b, a * b, a * b, a * b, a * b, b, V, b, b,
            a * b, a * b, a * b, a * b * b, a * b * b, a * b * b, a * b, a * b, a * b, a * b, a * b, a * b,
            a * b * b, a * b, a * b, a * b, a * b, a * b * b * 4, a * b * b * 4, a * b * b,
            a * b, a * b * 4, a * b, a * b, a * b * 4, a * b, b, b, b * 4, a * b, a * b, V * b, b, b]


def file_bytes(a: int, b: int) -> int:
    return 16 + sum(n * np.dtype(dt).itemsize for n, dt in zip(sizes(a, b), DTYPES))


def bytes_per_token(a: int, b: int) -> int:
    """ALGORITHMIC HBM bytes of one token."""
    return 13 * a * b * b + VOCAB * b + 168 * a * b + 40 * b


def write_bin(path: str, n_layers: int, n_embed: int, tensors) -> None:
    sz = sizes(n_layers, n_embed)
    with open(path, "wb") as f:
        f.write(np.array([n_layers, n_embed], dtype="<u8").tobytes())
        for i, t in enumerate(tensors):
            arr = np.ascontiguousarray(np.asarray(t), dtype=DTYPES[i]).reshape(-1)
            if arr.size != sz[i]:
                raise ValueError(f"tensor {i} ({NAMES[i]}): {arr.size} elements, format wants {sz[i]}")
            arr.tofile(f)


def read_bin(path: str, mmap: bool = True):
    """-> (n_layers, n_embed, [46 numpy arrays (memory-mapped, flat)])"""
    hdr = np.fromfile(path, dtype="<u8", count=2)
    if hdr.size != 2:
        raise IOError(f"{path}: truncated header")
    a, b = int(hdr[0]), int(hdr[1])
    out, off = [], 16
    for n, dt in zip(sizes(a, b), DTYPES):
        if mmap:
            out.append(np.memmap(path, dtype=dt, mode="r", offset=off, shape=(n,)))
        else:
            out.append(np.fromfile(path, dtype=dt, count=n, offset=off))
        off += n * np.dtype(dt).itemsize
    return a, b, out


def _buffers(a: int, b: int):
    """contents the converter writes into the scratch/state slots (convert_model.py:19-25,98-105)."""
    t = [None] * N_TENSORS
    t[X] = np.arange(b, dtype=_D)
    for s in (STATEXY, STATEAA, STATEBB, STATEDD):
        t[s] = np.zeros(a * b, dtype=_D)
    # torch.tensor([...python floats...]) is float32, then .double() (convert_model.py:19-25): -1e30 rounded through f32
    t[STATEPP] = np.full(a * b, np.float64(np.float32(-1e30)), dtype=_D)
    t[BUFFER1] = np.arange(b, dtype=_D)
    t[BUFFER2] = np.arange(VOCAB, dtype=_F)
    t[BUFFER3] = np.arange(b, dtype=_F)
    t[BUFFER4] = np.arange(b, dtype=_F)
    t[FFNKBUFFER] = np.arange(b, dtype=_D)
    t[FFNVBUFFER] = np.arange(b, dtype=_D)
    t[FFNRBUFFER] = np.arange(4 * b, dtype=_F)
    return t


def synthetic_tensors(n_layers: int, n_embed: int, seed: int = 0, head_scale: float = 6.0):
    """Seeded synthetic model in FILE layout: quantised matrices
    u8 ~ U{0..255} with per-input-row scale r = 2a/255 and offset o ~ -a (a ~ c/sqrt(N), jittered
    per row so r and o are not constant), LN weight 1 + 0.1 N(0,1), bias 0.1 N(0,1), time-mix
    U(0,1), decay = -exp(U(-6,1)), bonus = N(0,0.3), embedding N(0,1).  All f64 tensors except
    decay hold f32-representable values, as converted checkpoints do (convert_model.py:44-60)."""
    a, b = n_layers, n_embed
    rng = np.random.default_rng(seed)
    t = _buffers(a, b)

    def f32d(x):
        return x.astype(_F).astype(_D)

    t[EMBED] = rng.standard_normal(VOCAB * b, dtype=_F)
    ln = np.empty((4 * (a + 1), b), dtype=_D)
    ln[0::2] = f32d(1.0 + 0.1 * rng.standard_normal((2 * (a + 1), b)))
    ln[1::2] = f32d(0.1 * rng.standard_normal((2 * (a + 1), b)))
    t[LAYERNORMS] = ln.reshape(-1)
    for s in (MIXK, MIXV, MIXR, FFNMIXK, FFNMIXV):
        t[s] = f32d(rng.random(a * b))
    t[DECAY] = -np.exp(f32d(rng.uniform(-6.0, 1.0, a * b)))
    t[BONUS] = f32d(0.3 * rng.standard_normal(a * b))

    def qmat(wslot, rslot, oslot, layers, n_in, n_out, c):
        t[wslot] = rng.integers(0, 256, size=layers * n_in * n_out, dtype=_G)
        amp = (c / np.sqrt(n_in)) * (0.5 + rng.random(layers * n_in))
        t[rslot] = (2.0 * amp / 255.0).astype(_F)
        t[oslot] = (-amp * (1.0 + 0.1 * rng.standard_normal(layers * n_in))).astype(_F)

    qmat(KM, KR, O1, a, b, b, 1.0)
    qmat(VM, VR, O2, a, b, b, 1.0)
    qmat(RM, RR, O3, a, b, b, 1.0)
    qmat(ATTOUT, ATTOUTR, ATTOUTO, a, b, b, 1.0)
    qmat(FFNK, FFNKR, FFNKO, a, b, 4 * b, 1.0)
    qmat(FFNV, FFNVR, FFNVO, a, 4 * b, b, 1.0)
    qmat(FFNR, FFNRR, FFNRO, a, b, b, 1.0)
    qmat(HEAD, HEADR, HEADO, 1, b, VOCAB, head_scale)
    return t


def synthetic_tensors_torch(n_layers: int, n_embed: int, seed: int = 0, device="cuda", head_scale: float = 6.0):
    """Same distribution as synthetic_tensors, generated with torch on `device` (bench.py uses it
    for the 7B / 14B shapes: 8-15 GB of uint8 are produced in milliseconds on the GPU instead of
    tens of seconds on the host).  Returns 46 torch tensors in FILE layout; the scratch/state slots
    are None (rwkv_load_tensors accepts that)."""
    import torch

    a, b = n_layers, n_embed
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    t = [None] * N_TENSORS

    def randn(*shape, dtype=torch.float32):
        return torch.randn(*shape, generator=g, device=device, dtype=dtype)

    def rand(*shape, dtype=torch.float32):
        return torch.rand(*shape, generator=g, device=device, dtype=dtype)

    t[EMBED] = randn(VOCAB * b)
    ln = torch.empty((4 * (a + 1), b), device=device, dtype=torch.float64)
    ln[0::2] = (1.0 + 0.1 * randn(2 * (a + 1), b)).double()
    ln[1::2] = (0.1 * randn(2 * (a + 1), b)).double()
    t[LAYERNORMS] = ln.reshape(-1)
    for s in (MIXK, MIXV, MIXR, FFNMIXK, FFNMIXV):
        t[s] = rand(a * b).double()
    t[DECAY] = -torch.exp((rand(a * b) * 7.0 - 6.0).double())
    t[BONUS] = (0.3 * randn(a * b)).double()

    def qmat(wslot, rslot, oslot, layers, n_in, n_out, c):
        t[wslot] = torch.randint(0, 256, (layers * n_in * n_out,), generator=g, device=device, dtype=torch.uint8)
        amp = (c / float(np.sqrt(n_in))) * (0.5 + rand(layers * n_in))
        t[rslot] = (2.0 * amp / 255.0).float()
        t[oslot] = (-amp * (1.0 + 0.1 * randn(layers * n_in))).float()

    qmat(KM, KR, O1, a, b, b, 1.0)
    qmat(VM, VR, O2, a, b, b, 1.0)
    qmat(RM, RR, O3, a, b, b, 1.0)
    qmat(ATTOUT, ATTOUTR, ATTOUTO, a, b, b, 1.0)
    qmat(FFNK, FFNKR, FFNKO, a, b, 4 * b, 1.0)
    qmat(FFNV, FFNVR, FFNVO, a, 4 * b, b, 1.0)
    qmat(FFNR, FFNRR, FFNRO, a, b, b, 1.0)
    qmat(HEAD, HEADR, HEADO, 1, b, VOCAB, head_scale)
    return t


# named shapes of the BASELINE.json configs
SHAPES = {
    "169M": (12, 768),
    "430M": (24, 1024),
    "1B5": (24, 2048),
    "3B": (32, 2560),
    "7B": (32, 4096),
    "14B": (40, 5120),
}
