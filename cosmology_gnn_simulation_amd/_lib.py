"""ctypes binding of ``libcgnn_hip.so`` (the C ABI in ``include/cgnn.h``).

There is deliberately no fallback: if the shared library is missing the first
compute call raises, telling the user how to build it.  Python never touches
device memory itself; it passes ``tensor.data_ptr()`` and the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# the in-tree library (developer scripts that compare two builds assign another path before the first load())
LIB_PATH = os.path.join(_HERE, "libcgnn_hip.so")

MAX_HIDDEN_LAYERS = 6
F32, BF16, BF16_N16, F32X3, F32X3_N16, F16X2_N16, F16X2 = 0, 1, 2, 3, 4, 5, 6  # cgnn_precision
N16_NODE = (F32X3_N16, F16X2_N16)       # node-kernel packings whose epilogue fuses the next round's projections
P_F32, P_BF16_S32, P_BF16_S16, P_F16_S32 = 0, 1, 2, 3  # cgnn_ptable
STREAM_FOLDED = 1  # cgnn_edge_stream_run_w8 flags: CGNN_STREAM_FOLDED
LDS_WEIGHT_BUDGET = 152 * 1024           # CGNN_LDS_WEIGHT_BUDGET in csrc/mlp_device.hpp
PRECISIONS = {"fp32": F32, "f32": F32, "float32": F32, "bf16": BF16, "bfloat16": BF16, "bf16_n16": BF16_N16,
              "fp32x3": F32X3, "f32x3": F32X3, "fp32x3_n16": F32X3_N16, "fp16x2_n16": F16X2_N16,
              # "fp16x2" = f32-level accuracy from two fp16 terms: the 32-row packing, accepted wherever "fp32x3" is
              # (graph_network._PackedProcessor swaps in "fp16x2_n16" where the 16-row ring kernels apply)
              "fp16x2": F16X2, "f16x2": F16X2}
F32_EMULATED = (F32X3, F16X2)           # 32-row packings that emulate f32 on the bf16 / fp16 matrix cores

# every symbol include/cgnn.h declares (tests check the library exports all of them)
EXPORTS = (
    "cgnn_version", "cgnn_arch", "cgnn_last_error", "cgnn_packed_linear_bytes", "cgnn_pack_linear",
    "cgnn_mlp_rows", "cgnn_project_nodes", "cgnn_edge_block", "cgnn_aggregate", "cgnn_node_block",
    "cgnn_knn_workspace_bytes", "cgnn_knn_periodic", "cgnn_knn_sorted_order", "cgnn_segment_colsum",
    "cgnn_gather_rows", "cgnn_scatter_rows", "cgnn_tiled_rows", "cgnn_relayout", "cgnn_window_features",
    "cgnn_mlp_backward", "cgnn_weight_grad", "cgnn_weight_grad_x3_workspace_bytes", "cgnn_weight_grad_x3",
    "cgnn_col_dot", "cgnn_col_dot2", "cgnn_col_dot_workspace_bytes", "cgnn_col_dot_ordered", "cgnn_weight_grad_workspace_bytes", "cgnn_weight_grad_ordered", "cgnn_csr_workspace_bytes", "cgnn_csr_build",
    "cgnn_aggregate_csr", "cgnn_aggregate_csr_add", "cgnn_edge_stream", "cgnn_edge_stream_image_bytes", "cgnn_edge_stream_image_build",
    "cgnn_edge_stream_run", "cgnn_edge_stream_w8_supported", "cgnn_edge_stream_image_build_w8", "cgnn_edge_stream_run_w8", "cgnn_aggregate_plan_bytes", "cgnn_aggregate_plan_build", "cgnn_aggregate_planned", "cgnn_aggregate_planned_rows",
)
ROWS, TILED32 = 0, 1


class Linear(C.Structure):
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p), ("in_dim", C.c_int32), ("out_dim", C.c_int32)]


class Mlp(C.Structure):
    _fields_ = [("precision", C.c_int32), ("num_hidden_layers", C.c_int32),
                ("layer", Linear * (MAX_HIDDEN_LAYERS + 1)),
                ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p)]


class MlpBwdBuffers(C.Structure):
    _fields_ = [("h", C.c_void_p * MAX_HIDDEN_LAYERS), ("g_a", C.c_void_p * MAX_HIDDEN_LAYERS),
                ("g_o", C.c_void_p), ("zhat", C.c_void_p)]


class CgnnError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """dlopen the library (no GPU needed for this) and declare signatures."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise CgnnError(
            f"{LIB_PATH} not found. Build the gfx950 kernels first: "
            "`python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C cosmology_gnn_simulation_amd/csrc -j8`. There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, sz, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t, C.c_float
    lib.cgnn_version.restype = C.c_int
    lib.cgnn_arch.restype = C.c_char_p
    lib.cgnn_last_error.restype = C.c_char_p
    lib.cgnn_packed_linear_bytes.restype = sz
    lib.cgnn_packed_linear_bytes.argtypes = [i32, i32, i32]
    lib.cgnn_pack_linear.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp]
    lib.cgnn_mlp_rows.argtypes = [C.POINTER(Mlp), vp, i64, i32, vp, i32, i32, vp]
    lib.cgnn_tiled_rows.restype = i64
    lib.cgnn_tiled_rows.argtypes = [i64]
    lib.cgnn_relayout.argtypes = [vp, i32, vp, i32, i64, i32, vp]
    lib.cgnn_project_nodes.argtypes = [C.POINTER(Linear), C.POINTER(Linear), i32, vp, i64, vp, vp, i32, vp]
    lib.cgnn_edge_block.argtypes = [C.POINTER(Mlp), vp, vp, vp, vp, i64, vp, vp, vp, i32, i32, vp, vp, i32, vp]
    lib.cgnn_edge_stream.argtypes = [C.POINTER(Mlp), i32, vp, vp, i64, vp, vp, i64, vp, vp, i32, C.POINTER(Mlp), vp, i32, vp]
    lib.cgnn_edge_stream_image_bytes.restype = sz
    lib.cgnn_edge_stream_image_bytes.argtypes = [i32, i32, i32, i32]
    lib.cgnn_edge_stream_image_build.argtypes = [C.POINTER(Mlp), i32, C.POINTER(Mlp), i32, vp, sz, vp]
    lib.cgnn_edge_stream_run.argtypes = [vp, sz, i32, i32, i32, i32, vp, vp, i64, vp, vp, i64, vp, vp, vp, i32, vp]
    lib.cgnn_edge_stream_w8_supported.argtypes = [i32, i32, i32]
    lib.cgnn_edge_stream_image_build_w8.argtypes = [C.POINTER(Mlp), i32, C.POINTER(Mlp), i32, vp, sz, vp]
    lib.cgnn_edge_stream_run_w8.argtypes = [vp, sz, i32, i32, i32, i32, vp, vp, i64, vp, vp, i64, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.cgnn_aggregate.argtypes = [vp, i32, vp, vp, i64, i32, i64, i32, vp, vp]
    lib.cgnn_aggregate_plan_bytes.restype = sz
    lib.cgnn_aggregate_plan_bytes.argtypes = [i64, i32]
    lib.cgnn_aggregate_plan_build.argtypes = [vp, i64, i32, vp, vp]
    lib.cgnn_aggregate_planned.argtypes = [vp, vp, vp, i64, i32, i32, vp, vp]
    lib.cgnn_aggregate_planned_rows.argtypes = [vp, i64, vp, vp, i64, i32, i32, vp, vp]
    lib.cgnn_node_block.argtypes = [C.POINTER(Mlp), C.POINTER(Linear), C.POINTER(Linear), vp, vp, i64, vp, i32, i32,
                                    C.POINTER(Linear), C.POINTER(Linear), i32, vp, vp, i32, vp]
    lib.cgnn_knn_workspace_bytes.restype = sz
    lib.cgnn_knn_workspace_bytes.argtypes = [i64, i32]
    lib.cgnn_knn_periodic.argtypes = [vp, i64, f32, i32, vp, i64, vp, vp, vp, sz, vp]
    lib.cgnn_knn_sorted_order.argtypes = [vp, i64, vp, vp]
    lib.cgnn_segment_colsum.argtypes = [vp, vp, i64, i32, i32, vp, vp]
    lib.cgnn_window_features.argtypes = [vp, vp, vp, vp, i32, i64, f32, f32, f32, f32, f32, f32, vp, vp, vp]
    lib.cgnn_gather_rows.argtypes = [vp, vp, i64, i32, vp, vp]
    lib.cgnn_scatter_rows.argtypes = [vp, vp, i64, i32, vp, vp]
    lib.cgnn_mlp_backward.argtypes = [C.POINTER(Mlp), C.POINTER(Linear), C.POINTER(Mlp), C.POINTER(Linear), vp, i32, vp,
                                      i32, vp, i32, i64, C.POINTER(MlpBwdBuffers), vp, i32, vp, i32, vp]
    lib.cgnn_weight_grad.argtypes = [vp, i32, i32, vp, i32, i32, i64, vp, i32, i32, vp, vp]
    lib.cgnn_csr_workspace_bytes.restype = sz
    lib.cgnn_csr_workspace_bytes.argtypes = [i64]
    lib.cgnn_csr_build.argtypes = [vp, vp, i64, i64, vp, vp, vp, sz, vp]
    lib.cgnn_aggregate_csr.argtypes = [vp, vp, vp, i64, i32, vp, vp]
    lib.cgnn_aggregate_csr_add.argtypes = [vp, vp, vp, i64, i32, vp, vp, vp, vp]
    lib.cgnn_weight_grad_x3_workspace_bytes.argtypes = []
    lib.cgnn_weight_grad_x3_workspace_bytes.restype = C.c_size_t
    lib.cgnn_weight_grad_x3.argtypes = [vp, i32, vp, i32, i64, vp, i32, i32, vp, vp, C.c_size_t, vp]
    lib.cgnn_col_dot.argtypes = [vp, i32, vp, i32, i64, i32, vp, vp]
    lib.cgnn_col_dot2.argtypes = [vp, i32, vp, i32, i64, i32, vp, vp, vp]
    lib.cgnn_col_dot_workspace_bytes.restype = sz
    lib.cgnn_weight_grad_workspace_bytes.restype = sz
    lib.cgnn_weight_grad_workspace_bytes.argtypes = [i64, i32, i32]
    lib.cgnn_weight_grad_ordered.argtypes = [vp, i32, i32, vp, i32, i32, i64, vp, i32, i32, vp, vp, sz, vp]
    lib.cgnn_col_dot_workspace_bytes.argtypes = [i64, i32]
    lib.cgnn_col_dot_ordered.argtypes = [vp, i32, vp, i32, i64, i32, vp, vp, vp, sz, vp]
    missing = [name for name in EXPORTS if not hasattr(lib, name)]
    if missing:
        raise CgnnError(f"{LIB_PATH} lacks {missing}: rebuild it (make -C cosmology_gnn_simulation_amd/csrc)")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().cgnn_last_error().decode("utf-8", "replace")
        raise CgnnError(f"{what} failed (status {rc}): {msg}")


def require_device(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise CgnnError(f"{name} must live on a HIP device (got {t.device}); this engine has no CPU path")


def stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    """float32 + contiguous view of a device tensor (copy only when needed)."""
    require_device(t, name)
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def i32c(t: torch.Tensor, name: str) -> torch.Tensor:
    require_device(t, name)
    if t.dtype != torch.int32:
        t = t.to(torch.int32)
    return t if t.is_contiguous() else t.contiguous()
