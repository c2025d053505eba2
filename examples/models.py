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


# The weight gradient of a layer, gw = gy^T z, reduces over the rows (tens of thousands of nodes) into a small matrix:
# a handful of output tiles, so the library GEMM runs at a fraction of the chip unless the reduction is split -- a
# batched GEMM over s row slices followed by a sum.  Which s wins is erratic (MI355X, fp32, tools/gw_microbench.py:
# 22 500 x 256 x 256: library 111 us, s = 8 55 us, s = 32 181 us; 8 000 x 256 x 256: library 49 us, s = 8 231 us), so
# the first call of a shape times the candidates once and the choice is kept per (rows / 2048, widths).
_GW_CHOICE = {}


def _gw_slices(gy, z, s):
    m = gy.shape[0]
    mp = (m // s) * s
    gw = th.bmm(gy[:mp].view(s, mp // s, -1).transpose(1, 2), z[:mp].view(s, mp // s, -1)).sum(0)
    return gw + gy[mp:].t().mm(z[mp:]) if mp < m else gw


def _weight_grad(gy, z):
    m = gy.shape[0]
    if not gy.is_cuda or m < 4096 or not (gy.is_contiguous() and z.is_contiguous()):
        return gy.t().mm(z)
    key = (m >> 11, gy.shape[1], z.shape[1], gy.dtype)
    s = _GW_CHOICE.get(key)
    if s is None and th.cuda.is_current_stream_capturing():
        s = 0  # no timing inside a capture: the library GEMM (eager steps before the capture have usually chosen already)
    if s is None:
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        best = (float("inf"), 0)
        for cand in (0, 8, 16, 64):
            fn = (lambda: gy.t().mm(z)) if cand == 0 else (lambda: _gw_slices(gy, z, cand))
            fn()
            e0.record()
            for _ in range(3):
                fn()
            e1.record()
            e1.synchronize()
            best = min(best, (e0.elapsed_time(e1), cand))
        s = _GW_CHOICE[key] = best[1]
    return _gw_slices(gy, z, s) if s else gy.t().mm(z)


class _TallLinearFn(th.autograd.Function):
    """y = x W^T + b for x with very many rows (10^5 nodes x 10^2 features); the weight gradient through _weight_grad."""

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
        gw = _weight_grad(gy, x.contiguous())
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


class _FusedSageFn(th.autograd.Function):
    """SAGEConv('mean') as ONE autograd node: z = [h_dst | mean_{(u->v)} h_u], out = z W^T + b with W = [W_self |
    W_neigh].  One GEMM forward instead of two, two backward instead of four; the neighbour sums land straight in the
    right half of z, the in-degrees are counted by the same launch (once per block), and there is no autograd
    bookkeeping for the dozen small ops in between -- a training step is bound by the number of ops Python launches,
    not by their GPU time (profiles/r04_c_train_*).  Same arithmetic as SAGEConvMean (fp32; the float additions of the
    GEMM are grouped differently: rtol 1e-4)."""

    @staticmethod
    def forward(ctx, h, weight, bias, block, num_dst):
        from fgnn_hip.nn import aggregate_into, sage_finish_z
        h = h.contiguous()
        din = h.shape[1]
        # z and the in-degree counts out of ONE zeroed buffer (one fill), the neighbour sums and the counts by one
        # launch, the mean's scaling and the copy of the destinations' own rows by one more (fgnn_sage_finish_z)
        buf = th.zeros(num_dst * (2 * din + 1), dtype=h.dtype, device=h.device)
        z = buf[:num_dst * 2 * din].view(num_dst, 2 * din)
        deg = buf[num_dst * 2 * din:]
        aggregate_into(z[:, din:], h, block.row, block.col, in_degree=deg)
        if din % 4 == 0:
            inv = sage_finish_z(z, h, deg, num_dst, din)
        else:
            inv = deg.clamp(min=1).reciprocal_()
            z[:, din:] *= inv.unsqueeze(1)
            z[:, :din] = h[:num_dst]
        ctx.save_for_backward(z, weight, block.row, block.col, inv)
        ctx.num_src = h.shape[0]
        return th.addmm(bias, z, weight.t())

    @staticmethod
    def backward(ctx, gout):
        from fgnn_hip.nn import aggregate_into, sage_grad_prep
        z, weight, row, col, inv_deg = ctx.saved_tensors
        gout = gout.contiguous()
        din = z.shape[1] // 2
        gw = _weight_grad(gout, z) if ctx.needs_input_grad[1] else None
        gb = gout.sum(0) if ctx.needs_input_grad[2] else None
        gh = None
        if ctx.needs_input_grad[0]:
            gz = gout.mm(weight)
            if din % 4 == 0:  # self path + zero rows + scaled neighbour gradients in one launch
                gh, gagg = sage_grad_prep(gz, inv_deg, ctx.num_src, din)
                aggregate_into(gh, gagg, col, row)
            else:
                gagg = gz[:, din:] * inv_deg.unsqueeze(1)  # (contiguous: a new tensor)
                gh = aggregate_into(th.zeros((ctx.num_src, din), dtype=gz.dtype, device=gz.device), gagg, col, row)
                gh[:z.shape[0]] += gz[:, :din]
        return gh, gw, gb, None, None


class FusedSAGEConv(nn.Module):
    """SAGEConvMean with the layer's two linear maps as one weight [out, 2 in] = [W_self | W_neigh] and the whole layer
    as one autograd node (_FusedSageFn); falls back to the op-by-op formulation off the GPU / without the HIP
    library."""

    def __init__(self, in_feats, out_feats):
        super().__init__()
        self.in_feats = in_feats
        self.weight = nn.Parameter(th.empty(out_feats, 2 * in_feats))
        self.bias = nn.Parameter(th.zeros(out_feats))
        for half in (self.weight[:, :in_feats], self.weight[:, in_feats:]):  # each map initialised like nn.Linear's
            nn.init.kaiming_uniform_(half, a=5 ** 0.5)
        bound = 1 / in_feats ** 0.5
        nn.init.uniform_(self.bias, -bound, bound)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        """accepts the op-by-op layout too (fc_self.weight / fc_neigh.weight / fc_neigh.bias: SAGEConvMean here, DGL's
        SAGEConv in the reference's checkpoints): the two maps side by side are this layer's weight"""
        ks, kn, kb = prefix + "fc_self.weight", prefix + "fc_neigh.weight", prefix + "fc_neigh.bias"
        ksb = prefix + "fc_self.bias"
        if ks in state_dict and kn in state_dict and prefix + "weight" not in state_dict:
            state_dict[prefix + "weight"] = th.cat([state_dict.pop(ks), state_dict.pop(kn)], 1)
            # DGL < 0.8 keeps a bias in BOTH maps (their sum is what the layer adds); DGL >= 0.8 a separate `bias`
            # (already under this layer's name); SAGEConvMean here: fc_neigh.bias only
            parts = [state_dict.pop(k) for k in (kb, ksb) if k in state_dict]
            if parts and prefix + "bias" not in state_dict:
                state_dict[prefix + "bias"] = parts[0] if len(parts) == 1 else parts[0] + parts[1]
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def split_state_dict(self, prefix=""):
        """this layer's parameters in the op-by-op layout (what SAGEConvMean / DGL's SAGEConv load)"""
        d = self.in_feats
        return {prefix + "fc_self.weight": self.weight.detach()[:, :d].clone(),
                prefix + "fc_neigh.weight": self.weight.detach()[:, d:].clone(),
                prefix + "fc_neigh.bias": self.bias.detach().clone()}

    def forward(self, block, h):
        num_dst = block.number_of_dst_nodes()
        if (_fused_aggregate is not None and h.is_cuda and h.dtype == th.float32
                and getattr(block.row, "dtype", None) == th.int32 and block.col.dtype == th.int32):
            return _FusedSageFn.apply(h, self.weight, self.bias, block, num_dst)
        agg = _sum_to_dst(block, h) / _in_degree(block, h.dtype).clamp(min=1).unsqueeze(1)
        return th.cat([h[:num_dst], agg], 1).mm(self.weight.t()) + self.bias


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
    def __init__(self, in_feats, n_hidden, n_classes, n_layers, dropout, fused=True):
        super().__init__()
        dims = [in_feats] + [n_hidden] * (n_layers - 1) + [n_classes]
        # fused=False: the op-by-op layer (two nn.Linear-like maps per layer, the layout a DGL SAGEConv checkpoint has)
        conv = FusedSAGEConv if fused else SAGEConvMean
        self.layers = nn.ModuleList(conv(dims[i], dims[i + 1]) for i in range(n_layers))
        self.dropout = nn.Dropout(dropout)
        # set by a training loop that uses fgnn_hip.nn.Adam: its device-side step count keys the dropout masks, so
        # ReLU + dropout run as ONE launch (fgnn_relu_dropout) and a replayed graph still draws a fresh mask per step
        self.dropout_step = None
        self.dropout_seed = 0x5A4D47

    def forward(self, blocks, x):
        h = x
        for l, (layer, block) in enumerate(zip(self.layers, blocks)):
            h = layer(block, h)
            if l != len(self.layers) - 1:
                if self.dropout_step is not None and h.is_cuda and h.dtype == th.float32 and h.numel() % 4 == 0:
                    from fgnn_hip.nn import relu_dropout
                    h = relu_dropout(h, self.dropout.p, self.training, self.dropout_seed, self.dropout_step, l)
                else:
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
