"""A GraphSAGE training step replayed as a captured HIP graph.

One step on a papers100M-shaped batch is ~45 small kernels (0.5 ms of GPU time) that Python needs 0.6-1.2 ms to launch:
the step is bound by the host (the reference trains through DGL + PyTorch eager the same way,
example/samgraph/multi_gpu/train_graphsage.py:300-330).  A captured graph replays the whole step -- forward, loss,
backward, fused Adam -- with one launch.  Graphs want static shapes and a batch's sizes differ from batch to batch, so:

  * the step runs on the batch buffers' FULL-capacity tensors (they never move: fgnn_hip.lib.Batch.graph_buffers),
    sliced to the batch's sizes rounded UP to a few coarse buckets -- one graph per (buffer, bucket combination), a
    handful in practice;
  * the edges between the real count and the bucket are pointed at an extra, discarded destination row (source row 0)
    right before the replay, so they contribute to nothing that is kept; destination rows between the real count and
    the bucket are computed from whatever the buffer holds and are never referenced by the next layer's real edges;
    the loss reads the first `batch_size` output rows only.
Arithmetic on the real rows is the eager step's (same kernels, same order; the weight-gradient GEMMs see a few padded
all-but-irrelevant rows: they multiply rows of the upstream gradient that are exactly zero).  Batches that do not fit
the pattern (a short last batch of an epoch, sizes beyond a buffer) run eagerly.
"""
import os
import tempfile

import torch as th


def _bucket(n, floor):
    """n rounded up to eight steps per octave (at most 12.5 % padding), never finer than `floor`: batches of very
    different sizes still fall into a handful of buckets"""
    n = max(int(n), 1)
    g = max(1 << max(n.bit_length() - 4, 0), floor)
    return (n + g - 1) // g * g


class GraphedSageStep:
    def __init__(self, model, opt, loss_fcn, batch_size, edge_bucket=32768, node_bucket=32768, inner_bucket=4096,
                 max_graphs=16, tune_gemms=False):
        """opt must be capturable (torch.optim.Adam(..., fused=True, capturable=True)).  *_bucket: the finest rounding
        of edge counts / input rows / inner-layer rows; max_graphs: batches of further shapes run eagerly (a capture
        costs ~0.1 s: a workload whose shapes never repeat must not capture per batch).
        tune_gemms: before a shape is captured, its forward + backward run once with PyTorch's TunableOp choosing among
        the rocBLAS / hipBLASLt solutions for every GEMM shape of the step (static shapes: each is tuned once, ~1 s
        apiece), and the capture records the chosen kernels -- the library's default picks run the step's tall-skinny
        fp32 GEMMs at about half the fp32 matrix peak (papers100M-shaped step: 164 -> 120 us of GEMMs, replay 0.363 ->
        0.320 ms; profiles/r05_l_train_gemm_tuning.txt).  Same fp32 arithmetic, another summation order inside the
        GEMMs (like any change of library version)."""
        self.model, self.opt, self.loss_fcn, self.batch_size = model, opt, loss_fcn, batch_size
        self.gE, self.gS, self.gI, self.max_graphs = edge_bucket, node_bucket, inner_bucket, max_graphs
        self.graphs = {}
        self.eager_steps = self.replays = 0
        self._primed = False
        self.tune_gemms = bool(tune_gemms) and hasattr(th.cuda, "tunable")
        self.tuned_shapes = 0

    def _tune(self, blocks, x, y):
        """one forward + backward of the padded shapes with GEMM tuning on; no optimizer step, the gradients are dropped
        -- nothing of the training state moves (the dropout masks are keyed by the optimizer's step count)"""
        tn = th.cuda.tunable
        tn.enable(True)
        if not os.environ.get("PYTORCH_TUNABLEOP_FILENAME") and not self.tuned_shapes:
            # (TunableOp logs its picks to a file as it goes: not into the working directory)
            tn.set_filename(os.path.join(tempfile.gettempdir(), "fgnn_tunableop_%d.csv" % os.getpid()))
        if not self.tuned_shapes:  # bound the tuner's own time: a few timed runs per candidate kernel are enough
            tn.set_max_tuning_duration(15)
            tn.set_max_tuning_iterations(20)
        tn.tuning_enable(True)
        try:
            out = self.model(blocks, x)
            self._loss_backward(out, y)
            self.opt.zero_grad(set_to_none=True)
            th.cuda.synchronize()
        finally:
            tn.tuning_enable(False)  # the picks stay in use while the switch is on: the capture that follows records them
        self.tuned_shapes += 1

    def _loss_backward(self, out, y):
        """loss of the first batch_size rows of `out` + backward.  CrossEntropyLoss (mean, unweighted) on the GPU goes
        through fgnn_softmax_xent: loss and the logits' gradient from one launch, the gradient already padded with
        zero rows to out's shape -- log_softmax, nll_loss, their two backward ops, the slice's zeros + copy: six nodes
        less in the replayed graph."""
        n = min(self.batch_size, out.shape[0])
        lf = self.loss_fcn
        # (the kernel ignores every row whose label is outside [0, C) -- torch's ignore_index semantics for the default
        # -100 and any other negative value; an ignore_index that names a real class stays with torch)
        if (isinstance(lf, th.nn.CrossEntropyLoss) and lf.reduction == "mean" and lf.weight is None
                and lf.label_smoothing == 0.0 and lf.ignore_index < 0 and out.is_cuda and out.dtype == th.float32 and y.dtype == th.int64):
            try:
                from fgnn_hip.nn import softmax_xent
            except ImportError:
                softmax_xent = None
            if softmax_xent is not None:
                loss, g = softmax_xent(out[:n], y[:n].contiguous(), pad_rows=out.shape[0] - n)
                self.opt.zero_grad(set_to_none=True)
                out.backward(g)
                return loss
        loss = lf(out[:n], y[:n])
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        return loss

    def _eager(self, blocks, x, y):
        out = self.model(blocks, x)
        self._ncls = out.shape[1]
        loss = self._loss_backward(out, y)
        self.opt.step()
        self.eager_steps += 1
        # (detached: an autograd graph of an eager step that stays alive keeps its AccumulateGrad nodes bound to the
        # eager stream, and a later capture's backward would then run them outside the capture)
        return loss.detach()

    def step(self, bt, make_block):
        """bt: a waited-for fgnn_hip.lib.Batch with features and labels; make_block(row, col, num_src, num_dst) builds
        the block object the model's layers take.  Returns the loss tensor (valid until the next step)."""
        m = bt.meta
        L = int(m.num_layers)
        ne = [int(m.num_edge[l]) for l in range(L)]
        nsrc = [int(m.num_src[l]) for l in range(L)]
        ndst = [int(m.num_dst[l]) for l in range(L)]
        # samgraph layer numbering: layer 0 is the OUTER one (input features -> first hidden), layer L-1 ends at the seeds
        bufs = [bt.graph_buffers(l) for l in range(L)]
        x_full, y = bt.feat_buffer(), bt.label()
        eb = [_bucket(ne[l], self.gE if l == 0 else self.gI) for l in range(L)]
        db = [_bucket(ndst[l], self.gI) if l < L - 1 else ndst[l] for l in range(L)]  # padded dst rows (+1 dummy each)
        sb0 = _bucket(nsrc[0], self.gS)
        fits = (self._primed and ndst[L - 1] == self.batch_size and sb0 <= x_full.shape[0]
                and all(eb[l] <= bufs[l][0].numel() for l in range(L))
                # padded destination rows of the outer layer must be rows the batch really has (their features enter a
                # GEMM: no uninitialised memory there), and every layer's padded destinations must exist as its sources
                and db[0] + 1 <= nsrc[0] and all(db[l] + 1 <= db[l - 1] + 1 for l in range(1, L)))
        if not fits:
            # the very first step is always eager: lazy initialisation (GEMM workspaces, the weight-gradient choice of
            # examples/models.py, optimizer state) must not happen inside a capture
            self._primed = True
            blocks = [make_block(bufs[l][0][:ne[l]], bufs[l][1][:ne[l]], nsrc[l], ndst[l]) for l in range(L)]
            return self._eager(blocks, x_full[:nsrc[0]], y)
        # padded edges: source row 0 -> the discarded destination row db[l]
        for l in range(L):
            if eb[l] > ne[l]:
                bufs[l][0][ne[l]:eb[l]].zero_()
                bufs[l][1][ne[l]:eb[l]].fill_(db[l])
        key = (id(bt), tuple(eb), sb0, tuple(db))
        entry = self.graphs.get(key)
        if entry is None and len(self.graphs) >= self.max_graphs:
            blocks = [make_block(bufs[l][0][:ne[l]], bufs[l][1][:ne[l]], nsrc[l], ndst[l]) for l in range(L)]
            return self._eager(blocks, x_full[:nsrc[0]], y)
        if entry is None:
            blocks = [make_block(bufs[l][0][:eb[l]], bufs[l][1][:eb[l]], sb0 if l == 0 else db[l - 1] + 1, db[l] + 1)
                      for l in range(L)]
            if getattr(self, "_ncls", None) and x_full.is_cuda:  # the padded gradient buffer of the fused loss: not inside
                try:                                              # the capture
                    from fgnn_hip.nn import xent_grad_buffer, xent_scratch
                    xent_grad_buffer(x_full.device, db[L - 1] + 1, self._ncls)
                    xent_scratch(x_full.device, min(self.batch_size, db[L - 1] + 1))  # (its zeroing must not be replayed)
                except ImportError:
                    pass
            tuned = self.tune_gemms and x_full.is_cuda
            was_on = tuned and th.cuda.tunable.is_enabled()
            if tuned:
                try:
                    self._tune(blocks, x_full[:sb0], y)
                except Exception as e:  # the step works with the libraries' default picks: never lose it to the tuner
                    import warnings
                    warnings.warn("GEMM tuning switched off (%s: %s)" % (type(e).__name__, e))
                    self.tune_gemms = tuned = False
                    th.cuda.tunable.tuning_enable(False)
                    th.cuda.tunable.enable(was_on)
                    self.opt.zero_grad(set_to_none=True)
            g = th.cuda.CUDAGraph()
            with th.cuda.graph(g):
                out = self.model(blocks, x_full[:sb0])
                loss = self._loss_backward(out, y)
                self.opt.step()
            if tuned and not was_on:
                th.cuda.tunable.enable(False)  # the graph holds the chosen kernels; eager code keeps the defaults
            # (the entry keeps the batch object alive: its graphs replay on its buffers' addresses)
            entry = self.graphs[key] = (g, loss.detach(), bt)
            del out, loss, blocks
        entry[0].replay()
        self.replays += 1
        return entry[1]
