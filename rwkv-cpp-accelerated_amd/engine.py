"""ctypes binding of librwkv_mi355x.so (include/rwkv_mi355x.h) and a Python mirror of the
reference's host class `RWKV` (include/rwkv/rwkv/rwkv.h:245-429).

Nothing here computes: every forward goes through the C-ABI into the HIP kernels.  There is no
CPU fallback -- if the shared library is missing or no HIP device is present the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import modelfile as mf

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RWKV_LIB") or os.path.join(_HERE, "csrc", "librwkv_mi355x.so")   # RWKV_LIB: tuning variants

MODE_PARRALEL, MODE_GPT = 0, 1   # reference enums/enum.h:2-5
SAMPLE_BAN0, SAMPLE_RECIPE = 1, 2   # include/rwkv_mi355x.h
N_KCLASS = 7
ABI_VERSION = 6                   # RWKV_MI355X_ABI_VERSION of include/rwkv_mi355x.h
KCLASS_NAMES = ["first", "att_kvr_wkv", "att_out", "ffn_rk", "ffn_v", "head", "argmax"]

# every entry point declared in include/rwkv_mi355x.h (tests check the library exports all of them)
ABI_SYMBOLS = [
    "rwkv_create", "rwkv_load_file", "rwkv_load_tensors", "rwkv_n_layers", "rwkv_n_embed", "rwkv_max_ctx",
    "rwkv_forward", "rwkv_set_state", "rwkv_get_output", "rwkv_reset_state", "rwkv_decode_greedy",
    "rwkv_free", "rwkv_last_error", "rwkv_logits_device", "rwkv_state_device", "rwkv_stream",
    "rwkv_bytes_per_token", "rwkv_profile_token", "rwkv_debug_launch", "rwkv_debug_read", "rwkv_debug_write", "rwkv_debug_grid", "rwkv_debug_timeline", "rwkv_profile_batched", "rwkv_abi_version", "rwkv_resident_bytes", "rwkv_decode_form", "rwkv_set_layer_range", "rwkv_stage_forward", "rwkv_x_device", "rwkv_sample_typical", "rwkv_decode_typical",
    "rwkv_stage_chunk", "rwkv_xseq_device", "rwkv_xseq_copy", "rwkv_sync", "rwkv_pipe_rccl_path", "rwkv_pipe_unique_id", "rwkv_pipe_init", "rwkv_pipe_decode", "rwkv_pipe_decode_streams", "rwkv_pipe_profile", "rwkv_pipe_hop_stats",
    "rwkv_pipe_prefill", "rwkv_pipe_free", "rwkv_tensor_device", "rwkv_pipe_info", "rwkv_pipe_decode_dual",
]

_lib = None


class RWKVError(RuntimeError):
    pass


def lib():
    """dlopen the engine; fail loudly when it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RWKVError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
    L = C.CDLL(LIB_PATH)
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
    L.rwkv_create.argtypes = [C.POINTER(vp), i32]; L.rwkv_create.restype = i32
    L.rwkv_load_file.argtypes = [vp, C.c_char_p, u64]; L.rwkv_load_file.restype = i32
    L.rwkv_load_tensors.argtypes = [vp, u64, u64, C.POINTER(vp), i32, u64]; L.rwkv_load_tensors.restype = i32
    for f in ("rwkv_n_layers", "rwkv_n_embed", "rwkv_max_ctx", "rwkv_bytes_per_token"):
        getattr(L, f).argtypes = [vp]; getattr(L, f).restype = u64
    L.rwkv_forward.argtypes = [vp, C.POINTER(u64), u64, i32]; L.rwkv_forward.restype = i32
    L.rwkv_set_state.argtypes = [vp] + [vp] * 5 + [u64]; L.rwkv_set_state.restype = i32
    L.rwkv_get_output.argtypes = [vp] + [vp] * 6 + [u64]; L.rwkv_get_output.restype = i32
    L.rwkv_reset_state.argtypes = [vp]; L.rwkv_reset_state.restype = i32
    L.rwkv_decode_greedy.argtypes = [vp, u64, u64, C.POINTER(u64)]; L.rwkv_decode_greedy.restype = i32
    L.rwkv_free.argtypes = [vp]; L.rwkv_free.restype = None
    L.rwkv_last_error.argtypes = []; L.rwkv_last_error.restype = C.c_char_p
    L.rwkv_logits_device.argtypes = [vp]; L.rwkv_logits_device.restype = vp
    L.rwkv_state_device.argtypes = [vp, i32]; L.rwkv_state_device.restype = vp
    L.rwkv_stream.argtypes = [vp]; L.rwkv_stream.restype = vp
    L.rwkv_profile_token.argtypes = [vp, u64, i32, C.POINTER(C.c_double), C.POINTER(u64), C.POINTER(C.c_uint32)]
    L.rwkv_profile_token.restype = i32
    L.rwkv_debug_launch.argtypes = [vp, i32, u64, u64, C.c_uint32]; L.rwkv_debug_launch.restype = i32
    L.rwkv_debug_read.argtypes = [vp, i32, vp, u64]; L.rwkv_debug_read.restype = i32
    L.rwkv_debug_write.argtypes = [vp, i32, vp, u64]; L.rwkv_debug_write.restype = i32
    L.rwkv_debug_grid.argtypes = [vp]; L.rwkv_debug_grid.restype = u64
    L.rwkv_set_layer_range.argtypes = [vp, u64, u64]; L.rwkv_set_layer_range.restype = i32
    L.rwkv_stage_forward.argtypes = [vp, u64, C.c_uint32, C.POINTER(u64)]; L.rwkv_stage_forward.restype = i32
    L.rwkv_x_device.argtypes = [vp]; L.rwkv_x_device.restype = vp
    L.rwkv_profile_batched.argtypes = [vp, u64, i32, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]; L.rwkv_profile_batched.restype = i32
    L.rwkv_debug_timeline.argtypes = [vp, u64, vp, u64]; L.rwkv_debug_timeline.restype = i32
    L.rwkv_abi_version.argtypes = []; L.rwkv_abi_version.restype = i32
    L.rwkv_resident_bytes.argtypes = [vp]; L.rwkv_resident_bytes.restype = u64
    L.rwkv_decode_form.argtypes = [vp]; L.rwkv_decode_form.restype = i32
    if L.rwkv_abi_version() != ABI_VERSION:
        raise RWKVError(f"{LIB_PATH} has C-ABI version {L.rwkv_abi_version()}, this binding expects {ABI_VERSION}: rebuild it")
    L.rwkv_sample_typical.argtypes = [vp, u64, C.c_float, C.c_float, C.c_double, i32, C.POINTER(u64)]; L.rwkv_sample_typical.restype = i32
    L.rwkv_decode_typical.argtypes = [vp, u64, u64, C.c_float, C.c_float, u64, i32, C.POINTER(u64)]; L.rwkv_decode_typical.restype = i32
    L.rwkv_stage_chunk.argtypes = [vp, C.POINTER(u64), u64, u64, i32]; L.rwkv_stage_chunk.restype = i32
    L.rwkv_xseq_device.argtypes = [vp, i32]; L.rwkv_xseq_device.restype = vp
    L.rwkv_xseq_copy.argtypes = [vp, i32, vp, i32, u64]; L.rwkv_xseq_copy.restype = i32
    L.rwkv_sync.argtypes = [vp]; L.rwkv_sync.restype = i32
    L.rwkv_pipe_rccl_path.argtypes = [C.c_char_p, u64]; L.rwkv_pipe_rccl_path.restype = i32
    L.rwkv_pipe_unique_id.argtypes = [vp]; L.rwkv_pipe_unique_id.restype = i32
    L.rwkv_pipe_init.argtypes = [vp, vp, i32, i32]; L.rwkv_pipe_init.restype = i32
    L.rwkv_pipe_decode.argtypes = [vp, C.POINTER(u64), u64, C.POINTER(u64)]; L.rwkv_pipe_decode.restype = i32
    L.rwkv_pipe_decode_streams.argtypes = [vp, C.POINTER(u64), u64, u64, C.POINTER(u64)]; L.rwkv_pipe_decode_streams.restype = i32
    L.rwkv_pipe_profile.argtypes = [vp, i32]; L.rwkv_pipe_profile.restype = i32
    L.rwkv_pipe_hop_stats.argtypes = [vp, C.POINTER(C.c_double)]; L.rwkv_pipe_hop_stats.restype = i32
    L.rwkv_pipe_prefill.argtypes = [vp, C.POINTER(u64), u64]; L.rwkv_pipe_prefill.restype = i32
    L.rwkv_pipe_free.argtypes = [vp]; L.rwkv_pipe_free.restype = None
    L.rwkv_pipe_info.argtypes = [vp, C.c_char_p, u64]; L.rwkv_pipe_info.restype = i32
    L.rwkv_pipe_decode_dual.argtypes = [vp, C.POINTER(u64), u64, C.POINTER(u64)]; L.rwkv_pipe_decode_dual.restype = i32
    L.rwkv_tensor_device.argtypes = [vp, i32]; L.rwkv_tensor_device.restype = vp
    _lib = L
    return L


def _chk(rc: int):
    if rc != 0:
        raise RWKVError(lib().rwkv_last_error().decode(errors="replace") + f" (status {rc})")


def _ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


class RWKVState:
    """Mirror of reference class RWKVState (rwkv.h:140-242): five host arrays [stateSize][L][D] f64."""

    def __init__(self, num_layers: int, num_embed: int, stateSize: int = 1):
        self.num_layers, self.num_embed, self.stateSize = num_layers, num_embed, stateSize
        n = num_layers * num_embed * stateSize
        self.statexy = np.zeros(n); self.stateaa = np.zeros(n); self.statebb = np.zeros(n)
        self.statepp = np.zeros(n); self.statedd = np.zeros(n)

    def arrays(self):
        return [self.statexy, self.stateaa, self.statebb, self.statepp, self.statedd]

    def copy(self) -> "RWKVState":
        s = RWKVState(self.num_layers, self.num_embed, self.stateSize)
        for d, a in zip(s.arrays(), self.arrays()):
            d[:] = a
        return s

    def getSubState(self, offset: int = 0) -> "RWKVState":                      # rwkv.h:223-228
        if offset >= self.stateSize:
            raise RuntimeError(f"State get offset out of bounds, max offset is {self.stateSize}")
        n = self.num_layers * self.num_embed
        s = RWKVState(self.num_layers, self.num_embed, 1)
        for d, a in zip(s.arrays(), self.arrays()):
            d[:] = a[offset * n:(offset + 1) * n]   # (the reference's ctor ignores the L*D stride, rwkv.h:205; fixed)
        return s

    def setSubState(self, other: "RWKVState", offset: int = 0):                 # rwkv.h:231-240
        n = self.num_layers * self.num_embed
        for d, a in zip(self.arrays(), other.arrays()):
            d[offset * n:(offset + 1) * n] = a[:n]


class RWKV:
    """Mirror of reference class RWKV (rwkv.h:245-429) on top of the C-ABI.

    Default semantics are the reference's: the HOST state is authoritative and is uploaded before
    / downloaded after every forward (rwkv.h:353,372).  `resident=True` keeps the state on the
    device between calls (sync explicitly with pull_state/push_state): the engine's fast path."""

    def __init__(self, device: int = 0, resident: bool = False):
        self._h = C.c_void_p()
        _chk(lib().rwkv_create(C.byref(self._h), device))
        self.ready = False
        self.resident = resident
        self.num_layers = self.num_embed = 0
        self.maxContext = 1
        self.state = None
        self.out = None

    # -- loading ---------------------------------------------------------------------------
    def _after_load(self):
        L = lib()
        self.num_layers = int(L.rwkv_n_layers(self._h)); self.num_embed = int(L.rwkv_n_embed(self._h))
        self.maxContext = int(L.rwkv_max_ctx(self._h))
        self.state = RWKVState(self.num_layers, self.num_embed, self.maxContext)
        self.out = np.zeros(mf.VOCAB * self.maxContext, dtype=np.float32)
        self.ready = True

    def loadFile(self, filename: str, maxGPT: int = 1):                          # rwkv.h:281-310
        if self.ready:
            raise RuntimeError("RWKV already loaded")
        _chk(lib().rwkv_load_file(self._h, os.fsencode(filename), maxGPT))
        self._after_load()

    def loadTensors(self, n_layers: int, n_embed: int, tensors, maxGPT: int = 1):
        """46 tensors in file layout: numpy arrays (host) or torch CUDA tensors (device), not mixed;
        scratch/state slots may be None."""
        if self.ready:
            raise RuntimeError("RWKV already loaded")
        ptrs = (C.c_void_p * mf.N_TENSORS)()
        on_device = None
        keep = []
        for i, t in enumerate(tensors):
            if t is None:
                if i not in mf.BUFFER_SLOTS:
                    raise ValueError(f"tensor {i} ({mf.NAMES[i]}) is required")
                ptrs[i] = None
                continue
            if isinstance(t, np.ndarray):
                a = np.ascontiguousarray(t, dtype=mf.DTYPES[i]); keep.append(a)
                ptrs[i] = a.ctypes.data; dev = False
            else:   # torch tensor
                tt = t.contiguous(); keep.append(tt)
                ptrs[i] = tt.data_ptr(); dev = tt.is_cuda
                if tt.numel() * tt.element_size() != mf.sizes(n_layers, n_embed)[i] * np.dtype(mf.DTYPES[i]).itemsize:
                    raise ValueError(f"tensor {i} ({mf.NAMES[i]}) has the wrong byte size")
            if i in mf.BUFFER_SLOTS:
                continue
            if on_device is None:
                on_device = dev
            elif on_device != dev:
                raise ValueError("tensors must be all host or all device")
        if on_device:
            import torch
            torch.cuda.synchronize()
        _chk(lib().rwkv_load_tensors(self._h, n_layers, n_embed, ptrs, 1 if on_device else 0, maxGPT))
        self._after_load()

    # -- layer pipeline ---------------------------------------------------------------------
    def set_layer_range(self, l0: int, l1: int):
        _chk(lib().rwkv_set_layer_range(self._h, l0, l1))

    def stage_forward(self, token: int, slot: int = 0, want_pick: bool = False):
        pick = C.c_uint64(0)
        _chk(lib().rwkv_stage_forward(self._h, int(token), slot, C.byref(pick) if want_pick else None))
        return int(pick.value) if want_pick else None

    def x_device_ptr(self) -> int:
        return int(lib().rwkv_x_device(self._h) or 0)

    def stage_chunk(self, tokens, n: int, row0: int = 0, buf: int = 0):
        """one prompt chunk (n <= 32 tokens) through this stage's layers on the mm8_seq path (asynchronous)"""
        arr = (C.c_uint64 * n)(*[int(t) for t in tokens]) if tokens is not None else None
        _chk(lib().rwkv_stage_chunk(self._h, arr, n, row0, buf))

    def xseq_copy_from(self, src: "RWKV", rows: int, buf: int = 0):
        """same-device hand-over of a chunk's residual stream from the previous stage's context"""
        _chk(lib().rwkv_xseq_copy(self._h, buf, src._h, buf, rows))

    def sync(self):
        _chk(lib().rwkv_sync(self._h))

    # -- native pipeline transport (RCCL inside the engine) ------------------------------------
    @staticmethod
    def pipe_rccl_path() -> str:
        """the RCCL shared object the pipeline transport binds in this process (RWKV_RCCL_LIB, else the one already loaded -- torch's --, else the system's)"""
        buf = C.create_string_buffer(4096)
        _chk(lib().rwkv_pipe_rccl_path(buf, len(buf)))
        return buf.value.decode()

    @staticmethod
    def pipe_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _chk(lib().rwkv_pipe_unique_id(buf))
        return buf.raw

    def pipe_init(self, unique_id: bytes, rank: int, world: int):
        assert len(unique_id) == 128
        _chk(lib().rwkv_pipe_init(self._h, C.create_string_buffer(unique_id, 128), rank, world))

    def pipe_info(self) -> dict:
        """this rank's end of the transport (rwkv_pipe_info): rank, world, layers, device, PCI bus id, arch, RCCL version + path, prefill rows"""
        import json
        buf = C.create_string_buffer(2048)
        _chk(lib().rwkv_pipe_info(self._h, buf, len(buf)))
        return json.loads(buf.value.decode())

    def pipe_decode(self, first_tokens, n_steps: int, world: int, last: bool, n_streams: int | None = None):
        """greedy decode of n_streams (default: world) streams over the stages; [world][n_steps] ids on the last rank"""
        ft = (C.c_uint64 * world)(*([int(t) for t in first_tokens] + [0] * world)[:world]) if first_tokens is not None else None
        picks = (C.c_uint64 * (world * n_steps))() if last else None
        _chk(lib().rwkv_pipe_decode_streams(self._h, ft, n_steps, world if n_streams is None else n_streams, picks))
        return np.frombuffer(picks, dtype=np.uint64).reshape(world, n_steps).astype(np.int64) if last else None

    def pipe_decode_dual(self, first_tokens, n_steps: int, world: int, last: bool):
        """greedy decode of 2 * world streams, two per stage in flight on two communicators (the hop of one under the stage of the other);
        [2 * world][n_steps] ids on the last rank"""
        n = 2 * world
        ft = (C.c_uint64 * n)(*([int(t) for t in first_tokens] + [0] * n)[:n]) if first_tokens is not None else None
        picks = (C.c_uint64 * (n * n_steps))() if last else None
        _chk(lib().rwkv_pipe_decode_dual(self._h, ft, n_steps, picks))
        return np.frombuffer(picks, dtype=np.uint64).reshape(n, n_steps).astype(np.int64) if last else None

    def pipe_profile(self, on: bool = True):
        _chk(lib().rwkv_pipe_profile(self._h, 1 if on else 0))

    def pipe_hop_stats(self):
        """{n, mean_us, min_us, max_us} of the event pairs around the per-tick RCCL group of the last pipe_decode"""
        out = (C.c_double * 4)()
        _chk(lib().rwkv_pipe_hop_stats(self._h, out))
        return dict(n=int(out[0]), mean_us=float(out[1]), min_us=float(out[2]), max_us=float(out[3]))

    def pipe_prefill(self, tokens, n_tokens: int):
        arr = (C.c_uint64 * n_tokens)(*[int(t) for t in tokens]) if tokens is not None else None
        _chk(lib().rwkv_pipe_prefill(self._h, arr, n_tokens))

    def tensor_device_ptr(self, slot: int) -> int:
        return int(lib().rwkv_tensor_device(self._h, slot) or 0)

    # -- state sync --------------------------------------------------------------------------
    def push_state(self, n_slots: int | None = None):
        n = self.maxContext if n_slots is None else n_slots
        _chk(lib().rwkv_set_state(self._h, *[_ptr(a) for a in self.state.arrays()], n))

    def pull_state(self, n_slots: int | None = None):
        n = self.maxContext if n_slots is None else n_slots
        _chk(lib().rwkv_get_output(self._h, None, *[_ptr(a) for a in self.state.arrays()], n))

    def reset_state(self):
        for a in self.state.arrays():
            a[:] = 0
        _chk(lib().rwkv_reset_state(self._h))

    # -- forward -----------------------------------------------------------------------------
    def forward(self, token, mode: int = MODE_GPT):                               # rwkv.h:339-388
        if not self.ready:
            raise RuntimeError("RWKV not loaded")
        toks = [int(token)] if np.isscalar(token) else [int(t) for t in token]
        if len(toks) > self.maxContext:
            raise RuntimeError(f"Context too large, max context is {self.maxContext}")
        T = len(toks)
        arr = (C.c_uint64 * T)(*toks)
        if not self.resident:
            self.push_state(T)                                                    # setState, rwkv.h:353
        _chk(lib().rwkv_forward(self._h, arr, T, mode))
        if self.resident:
            _chk(lib().rwkv_get_output(self._h, _ptr(self.out), None, None, None, None, None, T))
        else:
            _chk(lib().rwkv_get_output(self._h, _ptr(self.out), *[_ptr(a) for a in self.state.arrays()], T))   # rwkv.h:372
        return self.out

    def decode_greedy(self, first_token: int, n_tokens: int) -> np.ndarray:
        """device-side greedy continuation on the resident state (slot 0); returns the picked ids."""
        if not self.ready:
            raise RuntimeError("RWKV not loaded")
        out = (C.c_uint64 * n_tokens)()
        if not self.resident:
            self.push_state(1)          # host state is authoritative: continue from it ...
        _chk(lib().rwkv_decode_greedy(self._h, int(first_token), n_tokens, out))
        if not self.resident:
            self.pull_state(1)          # ... and leave it where the generated tokens ended
        return np.frombuffer(out, dtype=np.uint64).copy()

    def sample_typical(self, temp: float = 0.9, tau: float = 0.8, u: float = 0.5, row: int = 0, ban0: bool = False, recipe: bool = False) -> int:
        """the reference's typical() ON THE DEVICE from the logits of the last forward (reference typical.h:20-58); u in
        [0, 1) is the caller's uniform -- the draw is the inverse CDF in token order.  Default: what the reference computes,
        a draw from softmax^(1/temp) (its cut at tau is a no-op, typical.h:50); recipe=True applies the documented cut."""
        tok = C.c_uint64(0)
        flags = (SAMPLE_BAN0 if ban0 else 0) | (SAMPLE_RECIPE if recipe else 0)
        _chk(lib().rwkv_sample_typical(self._h, int(row), float(temp), float(tau), float(u), flags, C.byref(tok)))
        return int(tok.value)

    def decode_typical(self, first_token: int, n_tokens: int, temp: float = 0.9, tau: float = 0.8, seed: int = 0, recipe: bool = False) -> np.ndarray:
        """device-side sampled continuation (storygen's loop with the device sampler; logit 0 banned)"""
        out = (C.c_uint64 * n_tokens)()
        if not self.resident:
            self.push_state(1)
        _chk(lib().rwkv_decode_typical(self._h, int(first_token), n_tokens, float(temp), float(tau), int(seed),
                                       SAMPLE_RECIPE if recipe else 0, out))
        if not self.resident:
            self.pull_state(1)
        return np.frombuffer(out, dtype=np.uint64).copy()

    def logits(self, n_tokens: int = 1) -> np.ndarray:
        _chk(lib().rwkv_get_output(self._h, _ptr(self.out), None, None, None, None, None, n_tokens))
        return self.out[: n_tokens * mf.VOCAB]

    def emptyState(self) -> RWKVState:                                            # rwkv.h:390-393
        return RWKVState(self.num_layers, self.num_embed, 1)

    def getTensorSize(self, i: int) -> int:                                       # rwkv.h:325-328
        return mf.sizes(self.num_layers, self.num_embed)[i]

    def getTensorTypes(self, i: int) -> int:                                      # rwkv.h:331-334
        return np.dtype(mf.DTYPES[i]).itemsize

    # -- measurement -------------------------------------------------------------------------
    def bytes_per_token(self) -> int:
        return int(lib().rwkv_bytes_per_token(self._h))

    def profile_token(self, token: int = 1, reps: int = 8):
        ms = (C.c_double * N_KCLASS)(); by = (C.c_uint64 * N_KCLASS)(); ln = (C.c_uint32 * N_KCLASS)()
        _chk(lib().rwkv_profile_token(self._h, token, reps, ms, by, ln))
        return [dict(name=KCLASS_NAMES[k], ms_total=ms[k], reps=reps, launches_per_token=int(ln[k]),
                     bytes_per_launch=int(by[k])) for k in range(N_KCLASS)]

    def profile_batched(self, token: int = 1, reps: int = 4):
        """per-class launch duration from one event pair around reps x L back-to-back launches (resets the state)"""
        ms = (C.c_double * N_KCLASS)(); n = (C.c_uint32 * N_KCLASS)()
        _chk(lib().rwkv_profile_batched(self._h, token, reps, ms, n))
        by = [p["bytes_per_launch"] for p in self.profile_token(token, 1)]
        return [dict(name=KCLASS_NAMES[k], us=1e3 * ms[k] / max(1, n[k]), launches=int(n[k]), bytes_per_launch=by[k]) for k in range(N_KCLASS)]

    def debug_timeline(self, token: int = 1, grid: int = 256):
        buf = np.zeros(512 * 8 * 8, dtype=np.uint64)
        _chk(lib().rwkv_debug_timeline(self._h, token, _ptr(buf), buf.size))
        return buf

    def resident_bytes(self) -> int:
        return int(lib().rwkv_resident_bytes(self._h))

    def decode_form(self) -> int:
        """Mask of the per-layer decode kernel classes that stream the tile image (bit 0 K/V/R, 1 att_out, 2 ffn k/r, 3 ffn_v)."""
        return int(lib().rwkv_decode_form(self._h))

    # ---- per-kernel parity hooks (include/rwkv_mi355x.h rwkv_debug_launch): the production decode kernels one launch at a time ----
    DBG = dict(x=(0, np.float64), ybuf=(1, np.float32), part_att=(2, np.float64), pmax_att=(3, np.float32), hbuf=(4, np.float32),
               rgate=(5, np.float32), part_ffn=(6, np.float64), pmax_ffn=(7, np.float32), lnstat=(8, np.float64))

    def debug_launch(self, cls, layer=0, token=0, slot=0):
        _chk(lib().rwkv_debug_launch(self._h, int(cls), int(layer), int(token), int(slot)))

    def debug_grid(self):
        return int(lib().rwkv_debug_grid(self._h))

    def debug_read(self, name):
        which, dt = self.DBG[name]
        D, G = self.num_embed, self.debug_grid()
        n = dict(x=D, ybuf=D, part_att=G, pmax_att=G, hbuf=4 * D, rgate=D, part_ffn=G, pmax_ffn=G, lnstat=6)[name]
        out = np.zeros(n, dt)
        _chk(lib().rwkv_debug_read(self._h, which, out.ctypes.data, out.nbytes))
        return out

    def debug_write(self, name, arr):
        which, dt = self.DBG[name]
        a = np.ascontiguousarray(arr, dt)
        _chk(lib().rwkv_debug_write(self._h, which, a.ctypes.data, a.nbytes))

    def close(self):
        if self._h:
            lib().rwkv_free(self._h)
            self._h = C.c_void_p()
            self.ready = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
