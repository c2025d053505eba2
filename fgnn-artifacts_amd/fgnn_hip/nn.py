"""torch.autograd wrapper of fgnn_block_aggregate (csrc/block_aggregate.hip): the message-passing step of the
DGL-free layers in examples/models.py.  out[col[e]] += w[e] * h[row[e]]; backward is the same kernel with the two
index arrays swapped.  Index tensors are the int32 row / col tensors the engine returns (no copies)."""
import ctypes as C

import torch

from . import lib as _lib


def _launch(src_idx, dst_idx, w, h, out):
    L = _lib.load()
    st = C.c_void_p(torch.cuda.current_stream(h.device).cuda_stream)
    _lib._check(L.fgnn_block_aggregate(C.c_void_p(src_idx.data_ptr()), C.c_void_p(dst_idx.data_ptr()),
                                       C.c_void_p(w.data_ptr()) if w is not None else None,
                                       C.c_size_t(src_idx.numel()), C.c_void_p(h.data_ptr()), C.c_size_t(h.shape[1]),
                                       C.c_void_p(out.data_ptr()), st), "fgnn_block_aggregate")


class _BlockAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, row, col, w, num_dst):
        h = h.contiguous()
        out = torch.zeros((num_dst, h.shape[1]), dtype=torch.float32, device=h.device)
        _launch(row, col, w, h, out)
        ctx.save_for_backward(row, col, w if w is not None else torch.empty(0, device=h.device))
        ctx.has_w = w is not None
        ctx.num_src = h.shape[0]
        return out

    @staticmethod
    def backward(ctx, grad_out):
        row, col, w = ctx.saved_tensors
        grad_h = None
        if ctx.needs_input_grad[0]:
            g = grad_out.contiguous()
            grad_h = torch.zeros((ctx.num_src, g.shape[1]), dtype=torch.float32, device=g.device)
            _launch(col, row, w if ctx.has_w else None, g, grad_h)
        return grad_h, None, None, None, None


def block_aggregate(h, row, col, num_dst, edge_weight=None):
    """sum_e edge_weight[e] * h[row[e]] into out[col[e]]; h float32 [num_src, D], row / col int32 [E]."""
    assert h.dtype == torch.float32 and row.dtype == torch.int32 and col.dtype == torch.int32
    assert row.is_contiguous() and col.is_contiguous() and row.numel() == col.numel()
    if edge_weight is not None:
        edge_weight = edge_weight.to(torch.float32).contiguous()
    return _BlockAggregate.apply(h, row, col, edge_weight, num_dst)


def aggregate_into(out, h, src_idx, dst_idx, edge_weight=None, in_degree=None):
    """out[dst_idx[e]] += edge_weight[e] * h[src_idx[e]] into a caller-initialised fp32 tensor: no autograd, for layers
    that write their own backward (examples/models.py: FusedSAGEConv).  `out` may be a column block of a wider
    row-major matrix (unit stride inside a row); in_degree (fp32 [num_dst], caller-zeroed) receives the edge count of
    every destination on the way."""
    assert h.dtype == torch.float32 and out.dtype == torch.float32 and h.is_contiguous()
    assert out.dim() == 2 and out.stride(1) == 1 and out.shape[1] == h.shape[1]
    assert src_idx.dtype == torch.int32 and dst_idx.dtype == torch.int32
    L = _lib.load()
    st = C.c_void_p(torch.cuda.current_stream(h.device).cuda_stream)
    _lib._check(L.fgnn_block_aggregate_ex(
        C.c_void_p(src_idx.data_ptr()), C.c_void_p(dst_idx.data_ptr()),
        C.c_void_p(edge_weight.data_ptr()) if edge_weight is not None else None, C.c_size_t(src_idx.numel()),
        C.c_void_p(h.data_ptr()), C.c_size_t(h.shape[1]), C.c_void_p(out.data_ptr()), C.c_size_t(out.stride(0)),
        C.c_void_p(in_degree.data_ptr()) if in_degree is not None else None, st), "fgnn_block_aggregate_ex")
    return out
