"""GNN layers on the COO blocks the engine returns, written with plain torch ops (DGL has no ROCm wheel in this image;
with DGL installed `get_dgl_blocks` returns DGLBlocks and the dgl.nn layers work unchanged).

Block convention (samgraph/torch/adapter.py): row[e] = local id of the sampled neighbour (source), col[e] = local id
of the seed (destination); the first number_of_dst_nodes() source nodes are the destination nodes themselves."""
import os

import torch as th
import torch.nn as nn
import torch.nn.functional as F

try:  # fused gather + segment-sum kernel of this repo (fgnn_hip/nn.py); FGNN_TORCH_AGGREGATE=1 forces the torch ops
    from fgnn_hip.nn import block_aggregate as _fused_aggregate
except ImportError:
    _fused_aggregate = None
if os.environ.get("FGNN_TORCH_AGGREGATE"):
    _fused_aggregate = None


class _TallLinearFn(th.autograd.Function):
    """y = x W^T + b for x with very many rows (10^5 nodes x 10^2 features).  The weight gradient gy^T x reduces over
    the rows; the library GEMM picked for that shape is slow (0.26 ms for 88 K x 128 x 256 on MI355X), a batched GEMM
    over 32 row slices followed by a sum takes 0.06 ms."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        y = x.mm(weight.t())
        return y + bias if bias is not None else y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gy.mm(weight) if ctx.needs_input_grad[0] else None
        m, s = x.shape[0], 32
        mp = (m // s) * s
        if mp >= 8192 and x.is_contiguous():
            gw = th.bmm(gy[:mp].view(s, mp // s, -1).transpose(1, 2), x[:mp].view(s, mp // s, -1)).sum(0)
            if mp < m:
                gw = gw + gy[mp:].t().mm(x[mp:])
        else:
            gw = gy.t().mm(x)
        gb = gy.sum(0) if ctx.needs_input_grad[2] else None
        return gx, gw, gb


class TallLinear(nn.Linear):
    def forward(self, x):
        if x.dim() == 2 and x.is_cuda and x.dtype == th.float32:
            return _TallLinearFn.apply(x, self.weight, self.bias)
        return super().forward(x)


def _sum_to_dst(block, h, weight=None):
    num_dst = block.number_of_dst_nodes()
    if (_fused_aggregate is not None and h.is_cuda and h.dtype == th.float32 and getattr(block.row, "dtype", None) == th.int32
            and block.col.dtype == th.int32):
        return _fused_aggregate(h, block.row, block.col, num_dst, weight)
    row, col = block.row.long(), block.col.long()
    msg = h[row] if weight is None else h[row] * weight.unsqueeze(1)
    return th.zeros((num_dst, h.shape[1]), dtype=h.dtype, device=h.device).index_add_(0, col, msg)


def _in_degree(block, dtype, weight=None):
    num_dst = block.number_of_dst_nodes()
    col = block.col.long()
    ones = th.ones_like(col, dtype=dtype) if weight is None else weight
    return th.zeros(num_dst, dtype=dtype, device=col.device).index_add_(0, col, ones)


class SAGEConvMean(nn.Module):
    """h_dst' = W_self h_dst + W_neigh mean_{(u->v)} h_u   (dgl.nn.SAGEConv(..., 'mean'))"""

    def __init__(self, in_feats, out_feats):
        super().__init__()
        self.fc_self = TallLinear(in_feats, out_feats, bias=False)
        self.fc_neigh = TallLinear(in_feats, out_feats, bias=True)

    def forward(self, block, h):
        num_dst = block.number_of_dst_nodes()
        agg = _sum_to_dst(block, h) / _in_degree(block, h.dtype).clamp(min=1).unsqueeze(1)
        return self.fc_self(h[:num_dst]) + self.fc_neigh(agg)


class GraphConv(nn.Module):
    """dgl.nn.GraphConv(norm='both', allow_zero_in_degree=True) as the reference's GCN uses it
    (example/samgraph/multi_gpu/train_gcn.py:24-47): D_out^-1/2 on the sources, D_in^-1/2 on the destinations."""

    def __init__(self, in_feats, out_feats, activation=None):
        super().__init__()
        self.fc = TallLinear(in_feats, out_feats, bias=True)
        self.activation = activation

    def forward(self, block, h):
        row = block.row.long()
        out_deg = th.zeros(h.shape[0], dtype=h.dtype, device=h.device).index_add_(
            0, row, th.ones_like(row, dtype=h.dtype)).clamp(min=1)
        h = h * out_deg.pow(-0.5).unsqueeze(1)
        agg = _sum_to_dst(block, h) * _in_degree(block, h.dtype).clamp(min=1).pow(-0.5).unsqueeze(1)
        out = self.fc(agg)
        return self.activation(out) if self.activation else out


class WeightedSAGEConv(nn.Module):
    """PinSAGE's weighted aggregator (example/samgraph/multi_gpu/train_pinsage.py:24-75): neighbours weighted by
    their random-walk visit counts (block.edata['weights']), concatenated with the destination's own state."""

    def __init__(self, in_feats, hidden, out_feats, dropout):
        super().__init__()
        self.Q = TallLinear(in_feats, hidden)
        self.W = TallLinear(in_feats + hidden, out_feats)
        self.dropout = nn.Dropout(dropout)

    def forward(self, block, h):
        num_dst = block.number_of_dst_nodes()
        w = block.edata["weights"].to(h.dtype)
        n = _sum_to_dst(block, F.relu(self.Q(self.dropout(h))), w)
        ws = _in_degree(block, h.dtype, w).clamp(min=1).unsqueeze(1)
        z = F.relu(self.W(self.dropout(th.cat([n / ws, h[:num_dst]], 1))))
        z_norm = z.norm(2, 1, keepdim=True)
        return z / th.where(z_norm == 0, th.ones_like(z_norm), z_norm)


class SAGE(nn.Module):
    def __init__(self, in_feats, n_hidden, n_classes, n_layers, dropout):
        super().__init__()
        dims = [in_feats] + [n_hidden] * (n_layers - 1) + [n_classes]
        self.layers = nn.ModuleList(SAGEConvMean(dims[i], dims[i + 1]) for i in range(n_layers))
        self.dropout = nn.Dropout(dropout)

    def forward(self, blocks, x):
        h = x
        for l, (layer, block) in enumerate(zip(self.layers, blocks)):
            h = layer(block, h)
            if l != len(self.layers) - 1:
                h = self.dropout(F.relu(h))
        return h


class GCN(nn.Module):
    def __init__(self, in_feats, n_hidden, n_classes, n_layers, dropout):
        super().__init__()
        dims = [in_feats] + [n_hidden] * (n_layers - 1) + [n_classes]
        self.layers = nn.ModuleList(GraphConv(dims[i], dims[i + 1], F.relu if i < n_layers - 1 else None)
                                    for i in range(n_layers))
        self.dropout = nn.Dropout(dropout)

    def forward(self, blocks, x):
        h = x
        for i, (layer, block) in enumerate(zip(self.layers, blocks)):
            if i != 0:
                h = self.dropout(h)
            h = layer(block, h)
        return h


class PinSAGE(nn.Module):
    def __init__(self, in_feats, n_hidden, n_classes, n_layers, dropout):
        super().__init__()
        dims = [in_feats] + [n_hidden] * (n_layers - 1) + [n_classes]
        self.layers = nn.ModuleList(WeightedSAGEConv(dims[i], n_hidden, dims[i + 1], dropout) for i in range(n_layers))

    def forward(self, blocks, x):
        h = x
        for layer, block in zip(self.layers, blocks):
            h = layer(block, h)
        return h


MODELS = {"graphsage": SAGE, "gcn": GCN, "pinsage": PinSAGE}
