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


# ---- fused pieces of a GraphSAGE training step (csrc/train_ops.hip) ------------------------------------------------------
# One launch each for work a step otherwise spends two to four torch ops on: the step is replayed as a captured HIP
# graph (examples/graphed_step.py), where a node costs the GPU 15-20 us whatever it does.

def _st(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def sage_finish_z(z, h, deg, num_dst, din):
    """z[:, :din] = h[:num_dst]; inv = 1 / max(deg, 1); z[:, din:] *= inv.  Returns inv (fp32 [num_dst])."""
    inv = torch.empty(num_dst, dtype=torch.float32, device=z.device)
    _lib._check(_lib.load().fgnn_sage_finish_z(C.c_void_p(z.data_ptr()), C.c_size_t(z.stride(0)), C.c_void_p(h.data_ptr()),
                                               C.c_size_t(h.stride(0)), C.c_void_p(deg.data_ptr()),
                                               C.c_void_p(inv.data_ptr()), C.c_size_t(num_dst), C.c_size_t(din), _st(z)),
                "fgnn_sage_finish_z")
    return inv


def sage_grad_prep(gz, inv, num_src, din):
    """(gh, gagg): gh[:num_dst] = gz[:, :din], gh[num_dst:] = 0; gagg = gz[:, din:] * inv[:, None]"""
    num_dst = gz.shape[0]
    gh = torch.empty((num_src, din), dtype=torch.float32, device=gz.device)
    gagg = torch.empty((num_dst, din), dtype=torch.float32, device=gz.device)
    _lib._check(_lib.load().fgnn_sage_grad_prep(C.c_void_p(gz.data_ptr()), C.c_size_t(gz.stride(0)),
                                                C.c_void_p(inv.data_ptr()), C.c_void_p(gh.data_ptr()),
                                                C.c_void_p(gagg.data_ptr()), C.c_size_t(num_dst), C.c_size_t(num_src),
                                                C.c_size_t(din), _st(gz)), "fgnn_sage_grad_prep")
    return gh, gagg


class _ReluDropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, d_step, tag):
        x = x.contiguous()
        y = torch.empty_like(x)
        _lib._check(_lib.load().fgnn_relu_dropout(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), C.c_size_t(x.numel()),
                                                  C.c_float(p), C.c_uint64(seed),
                                                  C.c_void_p(d_step.data_ptr()) if d_step is not None else None,
                                                  C.c_uint32(tag), _st(x)), "fgnn_relu_dropout")
        ctx.save_for_backward(y)
        ctx.p = p
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        gy = gy.contiguous()
        gx = torch.empty_like(gy)
        _lib._check(_lib.load().fgnn_relu_dropout_backward(C.c_void_p(y.data_ptr()), C.c_void_p(gy.data_ptr()),
                                                           C.c_void_p(gx.data_ptr()), C.c_size_t(y.numel()),
                                                           C.c_float(ctx.p), _st(y)), "fgnn_relu_dropout_backward")
        return gx, None, None, None, None


def relu_dropout(x, p, training, seed=0x5A4D47, d_step=None, tag=0):
    """dropout(relu(x), p) as one launch (and one in the backward pass).  The mask comes from Philox keyed by (seed,
    *d_step, tag, element): d_step = an fgnn_hip.nn.Adam's `.step_count` (device, advanced once per step) makes every
    training step draw a fresh mask inside a replayed graph; without it the caller varies `seed`.  Eval mode, CPU
    tensors and sizes that are not a multiple of 4 take the torch ops."""
    if not training or p <= 0.0:
        return torch.relu(x)
    if not (x.is_cuda and x.dtype == torch.float32 and x.numel() % 4 == 0):
        return torch.nn.functional.dropout(torch.relu(x), p, True)
    return _ReluDropout.apply(x, float(p), int(seed), d_step, int(tag))


class _XentWs:
    """scratch of fgnn_softmax_xent per (device, n): arrival counter (zeroed once) + row losses"""
    cache = {}

    @classmethod
    def get(cls, device, n):
        key = (torch.device(device), n)
        ws = cls.cache.get(key)
        if ws is None:
            nbytes = _lib.load().fgnn_softmax_xent_scratch_bytes(C.c_size_t(n))
            ws = cls.cache[key] = torch.zeros((nbytes + 3) // 4, dtype=torch.int32, device=device)
        return ws


def xent_scratch(device, n):
    """creates the scratch softmax_xent uses for n rows on `device` (arrival counter zeroed once + partial sums): call
    it BEFORE a graph capture whose first softmax_xent call would otherwise allocate -- and zero -- it inside the
    capture (a memset node in every replay, a buffer of the graph's private pool in this module's cache)"""
    return _XentWs.get(torch.device(device), n)


_XENT_GRAD = {}


def xent_grad_buffer(device, rows, num_class):
    """the reusable, zero-initialised gradient buffer softmax_xent(..., pad_rows) hands out for this shape: create it
    BEFORE a graph capture that will use it (an allocation inside a capture belongs to that graph's pool and its
    zeroing would be replayed)"""
    key = (torch.device(device), rows, num_class)
    g = _XENT_GRAD.get(key)
    if g is None:
        g = _XENT_GRAD[key] = torch.zeros((rows, num_class), dtype=torch.float32, device=device)
    return g


def softmax_xent(logits, labels, pad_rows=0):
    """(loss, dlogits): CrossEntropyLoss(reduction='mean')(logits, labels) and d loss / d logits, one launch.  Use as
    `loss, g = softmax_xent(out, y); out.backward(g)`: no scalar multiply node between the loss and the first GEMM.
    pad_rows > 0: dlogits has that many extra all-zero rows behind the n real ones (the gradient of a padded output
    whose first n rows carry the loss -- no zeros + slice-backward pair) and lives in a buffer that is reused by the
    next call with the same shape: consume it (backward) before calling again."""
    assert logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2 and logits.stride(1) == 1
    assert labels.dtype == torch.int64 and labels.is_contiguous() and labels.numel() == logits.shape[0]
    n, c = logits.shape
    L = _lib.load()
    L.fgnn_softmax_xent_scratch_bytes.restype = C.c_size_t
    ws = _XentWs.get(logits.device, n)
    loss = torch.empty((), dtype=torch.float32, device=logits.device)
    if pad_rows:
        g = xent_grad_buffer(logits.device, n + pad_rows, c)
    else:
        g = torch.empty((n, c), dtype=torch.float32, device=logits.device)
    _lib._check(L.fgnn_softmax_xent(C.c_void_p(logits.data_ptr()), C.c_size_t(logits.stride(0)), C.c_void_p(labels.data_ptr()),
                                    C.c_size_t(n), C.c_size_t(c), C.c_void_p(loss.data_ptr()), C.c_void_p(g.data_ptr()),
                                    C.c_size_t(c), C.c_void_p(ws.data_ptr()), C.c_size_t(ws.numel() * 4), _st(logits)),
                "fgnn_softmax_xent")
    return loss, g


class Adam:
    """torch.optim.Adam's update (lr, betas, eps, weight_decay; no amsgrad) for a handful of fp32 tensors as ONE launch
    per step, the step count on the device (`step_count`: two int64 words) -- capturable by construction.  The
    interface the training loops use: zero_grad(set_to_none=True), step(), param_groups[0]['lr'], state_dict()."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params if p.requires_grad]
        assert self.params and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in self.params)
        dev = self.params[0].device
        self.param_groups = [dict(params=self.params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)]
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self.step_count = torch.zeros(2, dtype=torch.int64, device=dev)

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def step(self):
        g = self.param_groups[0]
        L = _lib.load()
        live = [(p, m, v) for p, m, v in zip(self.params, self.exp_avg, self.exp_avg_sq) if p.grad is not None]
        for a in range(0, len(live), 8):  # eight tensors per launch
            chunk = live[a:a + 8]
            k = len(chunk)
            grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p, _, _ in chunk]
            arr = lambda ts: (C.c_void_p * k)(*[t.data_ptr() for t in ts])  # noqa: E731
            numel = (C.c_size_t * k)(*[p.numel() for p, _, _ in chunk])
            # (only the LAST launch of a step advances the count: earlier ones get a scratch counter that starts equal)
            last = a + 8 >= len(live)
            cnt = self.step_count if last else self.step_count.clone()
            _lib._check(L.fgnn_adam_step(arr([p for p, _, _ in chunk]), arr(grads), arr([m for _, m, _ in chunk]),
                                         arr([v for _, _, v in chunk]), numel, C.c_int(k), C.c_float(g["lr"]),
                                         C.c_float(g["betas"][0]), C.c_float(g["betas"][1]), C.c_float(g["eps"]),
                                         C.c_float(g["weight_decay"]), C.c_void_p(cnt.data_ptr()), _st(self.step_count)),
                        "fgnn_adam_step")

    def state_dict(self):
        return {"step": int(self.step_count[0]), "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}]}
