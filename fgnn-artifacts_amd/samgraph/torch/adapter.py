"""PyTorch adapter of the MI355X engine: the reference's samgraph/torch/adapter.py API
(config/init/.../get_next_batch/get_dgl_blocks/get_graph_*), bound to fgnn-artifacts_amd/samgraph/torch/c_lib.so.

Tensor getters: the reference builds torch tensors in C++ with torch::from_blob (adapter.cc:48-192); here the
engine returns raw pointers through the C ABI (include/samgraph.h) and they are wrapped without a copy via
__cuda_array_interface__ / ctypes.  Lifetime contract is the reference's: a batch's tensors are valid until
the next get_next_batch().
"""
import ctypes as _C

import numpy as _np
import torch

from samgraph.common import *  # noqa: F401,F403
from samgraph.common import SamGraphBasics

try:  # DGL has no ROCm wheel in this image; the example scripts need it, the engine does not
    import dgl as _dgl
    from dgl.heterograph import DGLBlock as _DGLBlock
except Exception:  # pragma: no cover
    _dgl = None

_basics = SamGraphBasics(__file__, 'c_lib')
_L = _basics.C_LIB_CTYPES

for _name in ('config', 'init', 'start', 'num_class', 'feat_dim', 'num_epoch', 'steps_per_epoch', 'get_next_batch',
              'get_graph_num_src', 'get_graph_num_dst', 'get_graph_num_edge', 'shutdown', 'sample_once', 'log_step',
              'log_step_add', 'log_epoch_add', 'get_log_init_value', 'get_log_step_value', 'get_log_epoch_value',
              'report_init', 'report_step', 'report_step_average', 'report_epoch', 'report_epoch_average',
              'report_node_access', 'trace_step_begin', 'trace_step_end', 'trace_step_begin_now',
              'trace_step_end_now', 'dump_trace', 'forward_barrier', 'wait_one_child', 'switch_init', 'data_init',
              'sample_init', 'train_init', 'extract_start', 'num_local_step', 'ext_queue_stats', 'ext_ring_mapping'):
    globals()[_name] = getattr(_basics, _name)

_TYPESTR = {0: '<f4', 1: '<f8', 2: '<f2', 3: '|u1', 4: '<i4', 5: '|i1', 6: '<i8'}
_TORCH = {0: torch.float32, 1: torch.float64, 2: torch.float16, 3: torch.uint8, 4: torch.int32, 5: torch.int8,
          6: torch.int64}
_NP = {0: _np.float32, 1: _np.float64, 2: _np.float16, 3: _np.uint8, 4: _np.int32, 5: _np.int8, 6: _np.int64}

_sz, _int, _u64 = _C.c_size_t, _C.c_int, _C.c_uint64
for _fn, _args in (('samgraph_torch_get_graph_feat_ptr', (_u64, _C.POINTER(_sz), _C.POINTER(_sz), _C.POINTER(_int),
                                                          _C.POINTER(_int))),
                   ('samgraph_torch_get_graph_label_ptr', (_u64, _C.POINTER(_sz), _C.POINTER(_int), _C.POINTER(_int))),
                   ('samgraph_torch_get_graph_row_ptr', (_u64, _int, _C.POINTER(_sz), _C.POINTER(_int))),
                   ('samgraph_torch_get_graph_col_ptr', (_u64, _int, _C.POINTER(_sz), _C.POINTER(_int))),
                   ('samgraph_torch_get_graph_data_ptr', (_u64, _int, _C.POINTER(_sz), _C.POINTER(_int))),
                   ('samgraph_torch_get_graph_input_nodes_ptr', (_u64, _C.POINTER(_sz), _C.POINTER(_int))),
                   ('samgraph_torch_get_graph_output_nodes_ptr', (_u64, _C.POINTER(_sz), _C.POINTER(_int))),
                   ('samgraph_torch_get_dataset_feat_ptr', (_C.POINTER(_sz), _C.POINTER(_sz), _C.POINTER(_int))),
                   ('samgraph_torch_get_dataset_label_ptr', (_C.POINTER(_sz), _C.POINTER(_int)))):
    getattr(_L, _fn).restype = _C.c_void_p
    getattr(_L, _fn).argtypes = _args


def _wrap(ptr, shape, dtype, device):
    """device >= 0: tensor on cuda:<device> aliasing the engine's buffer; -1: host memory."""
    n = 1
    for s in shape:
        n *= s
    if device >= 0:
        if n == 0 or not ptr:
            return torch.empty(shape, dtype=_TORCH[dtype], device='cuda:{:d}'.format(device))

        class _A(object):
            __cuda_array_interface__ = {'shape': tuple(shape), 'typestr': _TYPESTR[dtype], 'data': (ptr, False),
                                        'version': 2}
        return torch.as_tensor(_A(), device='cuda:{:d}'.format(device))
    if n == 0 or not ptr:
        return torch.empty(shape, dtype=_TORCH[dtype])
    nbytes = n * _np.dtype(_NP[dtype]).itemsize
    buf = (_C.c_char * nbytes).from_address(ptr)
    return torch.from_numpy(_np.frombuffer(buf, dtype=_NP[dtype]).reshape(shape))


def get_graph_feat(batch_key):
    rows, dim, dt, dev = _sz(), _sz(), _int(), _int()
    p = _L.samgraph_torch_get_graph_feat_ptr(batch_key, rows, dim, dt, dev)
    return _wrap(p, (rows.value, dim.value), dt.value, dev.value)


def get_graph_label(batch_key):
    n, dt, dev = _sz(), _int(), _int()
    p = _L.samgraph_torch_get_graph_label_ptr(batch_key, n, dt, dev)
    return _wrap(p, (n.value,), dt.value, dev.value)


def _ids(fn, batch_key, *layer):
    n, dev = _sz(), _int()
    p = fn(batch_key, *layer, n, dev)
    return _wrap(p, (n.value,), 4, dev.value)  # u32 storage viewed as i32, like the reference (kI32)


def get_graph_row(batch_key, layer_idx):
    return _ids(_L.samgraph_torch_get_graph_row_ptr, batch_key, layer_idx)


def get_graph_col(batch_key, layer_idx):
    return _ids(_L.samgraph_torch_get_graph_col_ptr, batch_key, layer_idx)


def get_graph_data(batch_key, layer_idx):
    return _ids(_L.samgraph_torch_get_graph_data_ptr, batch_key, layer_idx)


def get_graph_input_nodes(batch_key):
    return _ids(_L.samgraph_torch_get_graph_input_nodes_ptr, batch_key)


def get_graph_output_nodes(batch_key):
    return _ids(_L.samgraph_torch_get_graph_output_nodes_ptr, batch_key)


def get_dataset_feat():
    rows, dim, dt = _sz(), _sz(), _int()
    p = _L.samgraph_torch_get_dataset_feat_ptr(rows, dim, dt)
    return _wrap(p, (rows.value, dim.value), dt.value, -1)


def get_dataset_label():
    n, dt = _sz(), _int()
    p = _L.samgraph_torch_get_dataset_label_ptr(n, dt)
    return _wrap(p, (n.value,), dt.value, -1)


class CooBlock(object):
    """Stand-in for a DGLBlock when DGL is not installed: the bipartite COO of one layer
    (row = local id of the sampled neighbour in [0, num_src), col = local id of the seed in [0, num_dst))."""

    def __init__(self, row, col, num_src_nodes, num_dst_nodes):
        self.row, self.col = row, col
        self._num_src, self._num_dst = num_src_nodes, num_dst_nodes
        self.edata = {}

    def number_of_src_nodes(self):
        return self._num_src

    def number_of_dst_nodes(self):
        return self._num_dst

    def num_edges(self):
        return int(self.row.numel())


def _create_dgl_block(data, num_src_nodes, num_dst_nodes):
    row, col = data
    if _dgl is None:
        return CooBlock(row, col, num_src_nodes, num_dst_nodes)
    gidx = _dgl.heterograph_index.create_unitgraph_from_coo(2, num_src_nodes, num_dst_nodes, row, col, 'coo')
    return _DGLBlock(gidx, (['_N'], ['_N']), ['_E'])


def get_dgl_blocks(batch_key, num_layers, with_feat=True):
    feat = get_graph_feat(batch_key) if with_feat else None
    label = get_graph_label(batch_key) if with_feat else None
    blocks = []
    for i in range(num_layers):
        blocks.append(_create_dgl_block((get_graph_row(batch_key, i), get_graph_col(batch_key, i)),
                                        get_graph_num_src(batch_key, i), get_graph_num_dst(batch_key, i)))
    return blocks, feat, label


def get_dgl_blocks_with_weights(batch_key, num_layers, with_feat=True):
    blocks, feat, label = get_dgl_blocks(batch_key, num_layers, with_feat)
    for i, block in enumerate(blocks):
        block.edata['weights'] = get_graph_data(batch_key, i)
    return blocks, feat, label


def notify_sampler_ready(barrier):
    barrier.wait()


def wait_for_sampler_ready(barrier):
    barrier.wait()


def load_subtensor(batch_key, feat, label, device):
    input_nodes = get_graph_input_nodes(batch_key).to(feat.device)
    output_nodes = get_graph_output_nodes(batch_key).to(label.device)
    batch_inputs = torch.index_select(feat, 0, input_nodes.long()).to(device)
    batch_labels = torch.index_select(label, 0, output_nodes.long()).to(device)
    return batch_inputs, batch_labels
