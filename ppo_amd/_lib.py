"""ctypes binding of libppo_xp.so (include/ppo_xp.h).

The library is HIP only.  If it is missing, or no GPU is usable, calls fail loudly with
PpoXpError: there is no CPU fallback in this package.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PPO_XP_LIB: another build of the same ABI (tests load libppo_xp_experiments.so this way); default = the product library
LIB_PATH = os.environ.get("PPO_XP_LIB") or os.path.join(_HERE, "libppo_xp.so")

OK, EINVAL, ENOMEM, EIO, ENODEV, ESTATE = 0, -22, -12, -5, -19, -1
GAE_SEQUENTIAL, GAE_ASSOC, GAE_NO_PACK = 0, 1, 0x100
K_NAMES = ("gae_scan", "gae_chunk_summary", "gae_carry_scan", "pack_records", "gather", "finalize_stats", "normalize",
           "scalar_gather", "exchange")
K_GAE, K_GAE_SUMMARY, K_GAE_CARRY, K_PACK, K_GATHER, K_FINALIZE, K_NORMALIZE, K_SCALARS, K_EXCHANGE = range(9)
K_COUNT = 9


def gae_chunk(length: int) -> int:
    return (int(length) & 0xFFFF) << 16


class PpoXpError(RuntimeError):
    """A C-ABI call failed.  `.code` is the PPO_XP_E* value."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"ppo_xp error {code}: {msg}")
        self.code = code


class PpoXpPanic(PpoXpError, AssertionError):
    """PPO_XP_EINVAL: the reference would have panicked (assert_eq! on sizes)."""


class ViewsT(C.Structure):
    _fields_ = [("num_steps", C.c_size_t), ("obs_size", C.c_size_t), ("observations", C.c_void_p),
                ("actions", C.c_void_p), ("values", C.c_void_p), ("neglogps", C.c_void_p),
                ("rewards", C.c_void_p), ("dones", C.c_void_p), ("returns", C.c_void_p),
                ("advantages", C.c_void_p)]


class BatchView(C.Structure):
    _fields_ = [("index", C.c_size_t), ("batch_size", C.c_size_t), ("obs_size", C.c_size_t),
                ("observations", C.c_void_p), ("actions", C.c_void_p), ("values", C.c_void_p),
                ("neglogps", C.c_void_p), ("returns", C.c_void_p), ("advantages", C.c_void_p),
                ("adv_mean", C.c_float), ("adv_std", C.c_float)]


class BatchHost(C.Structure):
    _fields_ = [("capacity", C.c_size_t), ("index", C.c_size_t), ("batch_size", C.c_size_t), ("obs_size", C.c_size_t),
                ("observations", C.c_void_p), ("actions", C.c_void_p), ("values", C.c_void_p),
                ("neglogps", C.c_void_p), ("returns", C.c_void_p), ("advantages", C.c_void_p)]


# name -> (restype, argtypes); every symbol include/ppo_xp.h declares.
_vp, _sz, _i, _f, _u64 = C.c_void_p, C.c_size_t, C.c_int, C.c_float, C.c_uint64
SYMBOLS = {
    "ppo_xp_create": (_i, [C.POINTER(_vp), _i, _sz, _sz, _sz, _f, _f]),
    "ppo_xp_destroy": (None, [_vp]),
    "ppo_xp_add": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz]),
    "ppo_xp_add_many": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _sz]),
    "ppo_xp_add_device": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz]),
    "ppo_xp_reset_counter": (None, [_vp]),
    "ppo_xp_len": (_sz, [_vp]),
    "ppo_xp_counter": (_sz, [_vp]),
    "ppo_xp_last_values_idx": (_sz, [_vp]),
    "ppo_xp_returns": (_i, [_vp, _vp, _sz, _i, _vp]),
    "ppo_xp_returns_noalloc": (_i, [_vp, _vp, _sz, _i, _vp]),
    "ppo_xp_views": (_i, [_vp, C.POINTER(ViewsT)]),
    "ppo_xp_pass_begin": (_i, [_vp, _sz, _vp, _u64]),
    "ppo_xp_pass_begin_device": (_i, [_vp, _sz, _vp]),
    "ppo_xp_pass_begin_device_shuffle": (_i, [_vp, _sz, _u64]),
    "ppo_xp_pass_begin_bijection": (_i, [_vp, _sz, _u64]),
    "ppo_xp_explained_variance": (_i, [_vp, C.POINTER(C.c_float)]),
    "ppo_xp_explained_variance_sums": (_i, [_vp, C.POINTER(C.c_double)]),
    "ppo_xp_explained_variance_from_sums": (_i, [C.POINTER(C.c_double), C.POINTER(C.c_float)]),
    "ppo_xp_pass_next": (_i, [_vp, C.POINTER(BatchView)]),
    "ppo_xp_pass_next_host": (_i, [_vp, C.POINTER(BatchHost)]),
    "ppo_xp_pass_next_host_pipelined": (_i, [_vp, C.POINTER(BatchHost), C.POINTER(BatchHost)]),
    "ppo_xp_shuffle_prefetch": (_i, [_vp, _u64]),
    "ppo_xp_pass_gather_kind": (_i, [_vp]),
    "ppo_xp_set_streaming": (_i, [_vp, _i]),
    "ppo_xp_pass_num_batches": (_sz, [_vp]),
    "ppo_xp_pass_perm_device": (_vp, [_vp]),
    "ppo_xp_pass_perm_host": (_i, [_vp, _vp, _sz]),
    "ppo_xp_pass_stats_host": (_i, [_vp, _vp, _sz]),
    "ppo_xp_set_exchange": (_i, [_vp, _i, _vp]),
    "ppo_xp_attach_comm": (_i, [_vp, _vp, _i, _i]),
    "ppo_xp_host_alloc": (_vp, [_sz]),
    "ppo_xp_host_free": (None, [_vp]),
    "ppo_xp_pass_sums_device": (_vp, [_vp]),
    "ppo_xp_pass_finish": (_i, [_vp]),
    "ppo_xp_rccl_unique_id": (_i, [_vp]),
    "ppo_xp_rccl_comm_init": (_i, [C.POINTER(_vp), _vp, _i, _i]),
    "ppo_xp_rccl_comm_count": (_i, [_vp, C.POINTER(_i)]),
    "ppo_xp_rccl_comm_destroy": (_i, [_vp]),
    "ppo_xp_rnorm_create": (_i, [C.POINTER(_vp), _i, _sz, _f, _f, _f]),
    "ppo_xp_rnorm_destroy": (None, [_vp]),
    "ppo_xp_rnorm_step": (_i, [_vp, _vp, _vp, _vp, _sz]),
    "ppo_xp_rnorm_step_device": (_i, [_vp, _vp, _vp, _vp, _sz, _vp]),
    "ppo_xp_rnorm_step_begin_device": (_i, [_vp, _vp, _sz, _vp]),
    "ppo_xp_rnorm_sums_device": (_vp, [_vp]),
    "ppo_xp_rnorm_step_finish_device": (_i, [_vp, _vp, _vp, _vp, _sz, _vp]),
    "ppo_xp_rnorm_state": (_i, [_vp, C.POINTER(_f), C.POINTER(_f), C.POINTER(_f), _vp, _sz]),
    "ppo_xp_stream": (_vp, [_vp]),
    "ppo_xp_set_stream": (_i, [_vp, _vp]),
    "ppo_xp_sync": (_i, [_vp]),
    "ppo_xp_device": (_i, [_vp]),
    "ppo_xp_prof_enable": (_i, [_vp, _i]),
    "ppo_xp_prof_collect": (_i, [_vp, _vp, _vp]),
    "ppo_xp_kernel_name": (C.c_char_p, [_i]),
    "ppo_xp_set_tunable": (_i, [_vp, C.c_char_p, C.c_long]),
    "ppo_xp_last_error": (C.c_char_p, []),
    "ppo_xp_version": (C.c_char_p, []),
}

_lib = None


def _share_hip_runtime_with_torch():
    """One process must hold ONE HIP runtime: torch wheels bundle their own libamdhip64.so (same
    soname as /opt/rocm's).  Whichever copy initialises the GPU first leaves the other without
    devices, so when torch is installed its copy is mapped (RTLD_GLOBAL) before libppo_xp.so,
    whose DT_NEEDED libamdhip64.so.7 then binds to it.  Without torch the system runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load() -> C.CDLL:
    """dlopen libppo_xp.so and type every entry point.  Raises if the library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PpoXpError(ENODEV, f"{LIB_PATH} is not built (run `make -C ppo_amd/csrc`); there is no CPU fallback")
        _share_hip_runtime_with_torch()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error() -> str:
    return (load().ppo_xp_last_error() or b"").decode()


def check(rc: int) -> int:
    if rc < 0:
        msg = last_error()
        raise (PpoXpPanic if rc == EINVAL else PpoXpError)(rc, msg)
    return rc
