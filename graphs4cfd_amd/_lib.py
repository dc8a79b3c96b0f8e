"""ctypes binding of libg4c.so (the C-ABI declared in include/g4c.h).

The product path has no CPU / eager-torch fallback: if the HIP library is missing the import
of any compute entry point raises, and every op rejects non-HIP tensors.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("G4C_LIB_PATH") or os.path.join(_HERE, "lib", "libg4c.so")      # (G4C_LIB_PATH: A/B of two builds)

OK, EINVAL, ELAUNCH, EUNSUPPORTED = 0, -1, -2, -3
ACT_NONE, ACT_SELU, ACT_TANH = 0, 1, 2
MAX_SRC, MAX_LAYERS = 4, 4
MAX_HEADS = 2
NARROW_MAX = 8
KERNEL_MLP_RS = 5
KERNEL_MLP_RS2 = 6
KERNEL_NAMES = {0: "none", 1: "mlp_split_kernel", 2: "mlp_bx6_kernel", 3: "mlp_bx6i_kernel", 4: "mlp_ws_kernel", 5: "mlp_rs1_kernel", 6: "mlp_rs2_kernel"}   # g4c_mlp_last_kernel

_ACT_CODES = {None: ACT_NONE, "none": ACT_NONE, "selu": ACT_SELU, "tanh": ACT_TANH}


def act_code(activation) -> Optional[int]:
    """Map an activation spec to a fused-epilogue code, or None if it cannot be fused
    (an arbitrary callable is then applied by the caller with torch on the HIP tensor)."""
    if activation is None or isinstance(activation, str):
        return _ACT_CODES[activation]
    if activation is torch.tanh or activation is torch.nn.functional.tanh:
        return ACT_TANH
    if activation is torch.nn.functional.selu or activation is torch.selu:
        return ACT_SELU
    return None


class g4c_src_t(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("idx", C.c_void_p), ("width", C.c_int32), ("ld", C.c_int32),
                ("col0", C.c_int32), ("pre_act", C.c_int32), ("additive", C.c_int32), ("seg_mean", C.c_int32),
                ("w", C.c_void_p), ("seg_off", C.c_void_p), ("seg_perm", C.c_void_p), ("dtype", C.c_int32)]


class g4c_mlp_t(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("k_pad", C.c_int32 * MAX_LAYERS), ("n_pad", C.c_int32 * MAX_LAYERS),
                ("w", C.c_void_p * MAX_LAYERS), ("b", C.c_void_p * MAX_LAYERS),
                ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("ln_eps", C.c_float), ("n_out", C.c_int32), ("w_format", C.c_int32),
                ("range_flag", C.c_void_p), ("range_slot", C.c_int32)]


_SIGNATURES = {
    "g4c_version": (C.c_int, []),
    "g4c_device_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "g4c_last_error": (C.c_char_p, []),
    "g4c_plan_csr": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "g4c_plan_pool_edge": (C.c_int64, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "g4c_plan_pool_edge_ordered": (C.c_int64, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_void_p]),
    "g4c_segment_reduce": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "g4c_weighted_segment_mean": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                            C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "g4c_mlp_pack_layer": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                     C.c_int32, C.c_int32, C.c_void_p]),
    "g4c_mlp_forward_bf16": (C.c_int, [C.POINTER(g4c_mlp_t), C.POINTER(g4c_src_t), C.c_int32, C.c_int64, C.c_void_p,
                                       C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "g4c_mlp_pack_layer_bx6": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                         C.c_int32, C.c_int32, C.c_void_p]),
    "g4c_mlp_pack_layer_f16x3": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                           C.c_int32, C.c_int32, C.c_void_p]),
    "g4c_mlp_forward_bx6": (C.c_int, [C.POINTER(g4c_mlp_t), C.POINTER(g4c_src_t), C.c_int32, C.c_int64, C.c_void_p,
                                      C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "g4c_mlp_forward_bx6_save": (C.c_int, [C.POINTER(g4c_mlp_t), C.POINTER(g4c_src_t), C.c_int32, C.c_int64, C.c_void_p, C.c_int32,
                                           C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.c_int32,
                                           C.POINTER(C.c_void_p), C.c_int32, C.c_void_p]),
    "g4c_mlp_forward_heads_bx6": (C.c_int, [C.POINTER(g4c_mlp_t), C.POINTER(g4c_src_t), C.c_int32, C.c_int64, C.c_void_p, C.c_int32,
                                            C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.c_int32, C.c_void_p]),
    "g4c_mlp_forward_heads_bf16": (C.c_int, [C.POINTER(g4c_mlp_t), C.POINTER(g4c_src_t), C.c_int32, C.c_int64, C.c_void_p, C.c_int32,
                                            C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.c_int32, C.c_void_p]),
    "g4c_mlp_forward_heads_bf16_out": (C.c_int, [C.POINTER(g4c_mlp_t), C.POINTER(g4c_src_t), C.c_int32, C.c_int64, C.c_void_p, C.c_int32,
                                                 C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_void_p]),
    "g4c_mlp_forward_heads_bf16_rows": (C.c_int, [C.POINTER(g4c_mlp_t), C.POINTER(g4c_src_t), C.c_int32, C.c_int64, C.c_void_p, C.c_int32,
                                                  C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_void_p]),
    "g4c_mp_layer_forward_bx6": (C.c_int, [C.POINTER(g4c_mlp_t), C.POINTER(g4c_src_t), C.c_int32, C.c_int64, C.c_void_p, C.c_int32,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                           C.POINTER(g4c_mlp_t), C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                           C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.c_int32, C.c_void_p]),
    "g4c_mlp_forward_bf16_out": (C.c_int, [C.POINTER(g4c_mlp_t), C.POINTER(g4c_src_t), C.c_int32, C.c_int64, C.c_void_p, C.c_int32,
                                           C.c_int32, C.c_int32, C.c_void_p]),
    "g4c_plan_tiles": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64]),
    "g4c_mlp_forward_bx6_agg": (C.c_int, [C.POINTER(g4c_mlp_t), C.POINTER(g4c_src_t), C.c_int32, C.c_int64, C.c_void_p, C.c_int32,
                                          C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                          C.c_void_p]),
    "g4c_mlp_forward_bf16_agg": (C.c_int, [C.POINTER(g4c_mlp_t), C.POINTER(g4c_src_t), C.c_int32, C.c_int64, C.c_void_p, C.c_int32,
                                           C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                           C.c_int32, C.c_void_p]),
    "g4c_mlp_forward": (C.c_int, [C.POINTER(g4c_mlp_t), C.POINTER(g4c_src_t), C.c_int32, C.c_int64, C.c_void_p,
                                  C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "g4c_mlp_bx6i_enable": (C.c_int, [C.c_int]),
    "g4c_mlp_ws_enable": (C.c_int, [C.c_int]),
    "g4c_mlp_small_launch_tiles": (C.c_int, [C.c_int]),
    "g4c_layer_norm": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "g4c_debug_mean_div": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "g4c_mlp_last_kernel": (C.c_int, []),
    "g4c_mlp_forward_rows": (C.c_int, [C.POINTER(g4c_mlp_t), C.POINTER(g4c_src_t), C.c_int32, C.c_int64, C.c_int64, C.c_int64,
                                       C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                       C.c_int32, C.c_void_p]),
    "g4c_mlp_forward_heads": (C.c_int, [C.POINTER(g4c_mlp_t), C.POINTER(g4c_src_t), C.c_int32, C.c_int64, C.c_void_p, C.c_int32,
                                        C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.c_int32, C.c_void_p]),
    "g4c_project_to_edges": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                       C.c_void_p, C.c_int32, C.c_void_p]),
    "g4c_edge_scalar_to_node_vector": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.c_int32,
                                                 C.c_void_p, C.c_int32, C.c_void_p]),
    "g4c_knn_grid": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p,
                               C.c_void_p, C.c_float, C.c_int32, C.c_void_p, C.c_void_p]),
    "g4c_knn_grid_query": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_float, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "g4c_rollout_advance": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                      C.c_void_p, C.c_int64, C.c_void_p]),
    "g4c_activation_inplace": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "g4c_add_cols": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                               C.c_int32, C.c_int64, C.c_void_p]),
    "g4c_copy_cols": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                C.c_int32, C.c_int64, C.c_void_p]),
    # training path (train_ops.hip)
    "g4c_train_gather": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                   C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_void_p]),
    "g4c_act_grad": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                               C.c_int32, C.c_int64, C.c_void_p]),
    "g4c_layernorm_grad_partials": (C.c_int32, [C.c_int64]),
    "g4c_layernorm_grad": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                     C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_void_p]),
    "g4c_colsum_partials": (C.c_int32, [C.c_int64]),
    "g4c_colsum": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "g4c_weight_grad_partials": (C.c_int32, [C.c_int64]),
    "g4c_weight_grad_scratch_floats": (C.c_int64, [C.c_int64]),
    "g4c_weight_grad": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32,
                                  C.c_void_p]),
    "g4c_segment_broadcast": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_void_p, C.c_int32, C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def load() -> C.CDLL:
    """Load libg4c.so (built in-tree by `__graft_entry__.build()` / `make -C graphs4cfd_amd/csrc`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"graphs4cfd_amd: HIP library not found at {LIB_PATH}. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)   # AttributeError if the library does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(code: int) -> None:
    if code == OK:
        return
    msg = load().g4c_last_error().decode("utf-8", "replace")
    if code == EINVAL:
        raise ValueError(msg)
    if code == EUNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError(msg)


def require_hip(*tensors: torch.Tensor) -> torch.device:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("graphs4cfd_amd kernels run on an MI355X (HIP) device only; got a tensor on "
                               f"'{t.device}'. Move the model and Graph to 'cuda' (there is no CPU fallback).")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"tensors on different devices: {dev} and {t.device}")
    return dev


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def stream_handle(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream
