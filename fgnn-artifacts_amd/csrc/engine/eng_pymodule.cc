// eng_pymodule.cc -- `from samgraph.torch import c_lib`: the nine tensor getters under the reference's own names.
//
// The reference's c_lib is a pybind11 module living in the same shared object as the C ABI (samgraph/torch/adapter.cc:
// 48-192, PYBIND11_MODULE at :177-189; imported by samgraph/torch/adapter.py:26) and returns torch tensors built with
// torch::from_blob.  This file gives c_lib.so the same module entry point and the same nine functions
//     samgraph_torch_get_graph_feat(key) / _label(key) / _row(key, layer) / _col(key, layer) / _data(key, layer)
//     samgraph_torch_get_dataset_feat() / _label(), samgraph_torch_get_graph_input_nodes(key) / _output_nodes(key)
// so the reference's unmodified adapter.py binds this library.  Each getter calls the plain-C pointer getter of the
// boundary (include/samgraph.h, samgraph_torch_*_ptr) and wraps the pointer WITHOUT a copy: device memory through
// __cuda_array_interface__ + torch.as_tensor (what torch::from_blob does for a foreign device pointer), host memory
// through ctypes + torch.frombuffer.  Lifetime is the reference's contract: valid until the next
// samgraph_get_next_batch (the reference's deleter closures only pin the same buffers, adapter.cc:59).
//
// c_lib.so stays a plain C-ABI library: nothing here links against libpython or libtorch.  The few CPython entry
// points are looked up with dlsym when the interpreter calls PyInit_c_lib (they are in the process by then, by
// definition), so a C host can still dlopen the library; `nm -u c_lib.so` shows no Py* symbol
// (tests/test_capi_exports.py).  Python.h is needed for the struct layouts only.
#include <Python.h>
#include <dlfcn.h>

#include <cstdint>

#include "samgraph.h"

namespace {

// CPython API used, resolved at PyInit time.  No Py_* macro that touches a data symbol or an inline refcount.
struct PyApi {
  PyObject *(*ModuleCreate2)(PyModuleDef *, int);
  PyObject *(*ModuleGetDict)(PyObject *);
  int (*ArgParseTuple)(PyObject *, const char *, ...);
  PyObject *(*CallFunction)(PyObject *, const char *, ...);
  PyObject *(*RunString)(const char *, int, PyObject *, PyObject *, PyCompilerFlags *);
  PyObject *(*EvalGetBuiltins)(void);
  int (*DictSetItemString)(PyObject *, const char *, PyObject *);
  PyObject *(*DictGetItemString)(PyObject *, const char *);
  void (*DecRef)(PyObject *);
  bool ok = false;
} py;

PyObject *g_wrap = nullptr;  // the Python helper below (borrowed from the module dict, which lives as long as the module)

template <typename T>
bool resolve(T &fn, const char *name) {
  fn = reinterpret_cast<T>(dlsym(RTLD_DEFAULT, name));
  return fn != nullptr;
}

bool resolve_python() {
  if (py.ok) return true;
  py.ok = resolve(py.ModuleCreate2, "PyModule_Create2") && resolve(py.ModuleGetDict, "PyModule_GetDict") &&
          resolve(py.ArgParseTuple, "PyArg_ParseTuple") && resolve(py.CallFunction, "PyObject_CallFunction") &&
          resolve(py.RunString, "PyRun_StringFlags") && resolve(py.EvalGetBuiltins, "PyEval_GetBuiltins") &&
          resolve(py.DictSetItemString, "PyDict_SetItemString") &&
          resolve(py.DictGetItemString, "PyDict_GetItemString") && resolve(py.DecRef, "Py_DecRef");
  return py.ok;
}

// _wrap(ptr, rows, dim, dtype code, device): tensor aliasing the engine's buffer.  dim < 0: one-dimensional.
// device >= 0: cuda:<device>; -1: host memory.  dtype codes: include/fgnn_hip.h (F32 F64 F16 U8 I32 I8 I64).
const char *kHelper = R"PY(
def _wrap(ptr, rows, dim, dt, device):
    import torch
    typestr = ('<f4', '<f8', '<f2', '|u1', '<i4', '|i1', '<i8')[dt]
    tdtype = (torch.float32, torch.float64, torch.float16, torch.uint8, torch.int32, torch.int8, torch.int64)[dt]
    shape = (rows,) if dim < 0 else (rows, dim)
    n = rows * (1 if dim < 0 else dim)
    if n == 0 or not ptr:
        return torch.empty(shape, dtype=tdtype, device='cuda:%d' % device if device >= 0 else 'cpu')
    if device >= 0:
        class _A(object):
            __cuda_array_interface__ = {'shape': shape, 'typestr': typestr, 'data': (ptr, False), 'version': 2}
        return torch.as_tensor(_A(), device='cuda:%d' % device)
    import ctypes
    buf = (ctypes.c_char * (n * torch.empty((), dtype=tdtype).element_size())).from_address(ptr)
    return torch.frombuffer(buf, dtype=tdtype, count=n).reshape(shape)
)PY";

PyObject *wrap(const void *ptr, size_t rows, long dim, int dt, int device) {
  return py.CallFunction(g_wrap, "Knlii", (unsigned long long)(uintptr_t)ptr, (Py_ssize_t)rows, dim, dt, device);
}

constexpr int kI32 = 4;  // ids: u32 storage viewed as i32, like the reference (kI32, adapter.cc:86)

PyObject *get_graph_feat(PyObject *, PyObject *args) {
  unsigned long long key;
  if (!py.ArgParseTuple(args, "K", &key)) return nullptr;
  size_t rows = 0, dim = 0;
  int dt = 0, dev = 0;
  const void *p = samgraph_torch_get_graph_feat_ptr(key, &rows, &dim, &dt, &dev);
  return wrap(p, rows, (long)dim, dt, dev);
}

PyObject *get_graph_label(PyObject *, PyObject *args) {
  unsigned long long key;
  if (!py.ArgParseTuple(args, "K", &key)) return nullptr;
  size_t n = 0;
  int dt = 0, dev = 0;
  const void *p = samgraph_torch_get_graph_label_ptr(key, &n, &dt, &dev);
  return wrap(p, n, -1, dt, dev);
}

template <const uint32_t *(*FN)(uint64_t, int, size_t *, int *)>
PyObject *get_layer_ids(PyObject *, PyObject *args) {
  unsigned long long key;
  int layer;
  if (!py.ArgParseTuple(args, "Ki", &key, &layer)) return nullptr;
  size_t n = 0;
  int dev = 0;
  const uint32_t *p = FN(key, layer, &n, &dev);
  return wrap(p, n, -1, kI32, dev);
}

template <const uint32_t *(*FN)(uint64_t, size_t *, int *)>
PyObject *get_ids(PyObject *, PyObject *args) {
  unsigned long long key;
  if (!py.ArgParseTuple(args, "K", &key)) return nullptr;
  size_t n = 0;
  int dev = 0;
  const uint32_t *p = FN(key, &n, &dev);
  return wrap(p, n, -1, kI32, dev);
}

PyObject *get_dataset_feat(PyObject *, PyObject *) {
  size_t rows = 0, dim = 0;
  int dt = 0;
  const void *p = samgraph_torch_get_dataset_feat_ptr(&rows, &dim, &dt);
  return wrap(p, rows, (long)dim, dt, -1);
}

PyObject *get_dataset_label(PyObject *, PyObject *) {
  size_t n = 0;
  int dt = 0;
  const void *p = samgraph_torch_get_dataset_label_ptr(&n, &dt);
  return wrap(p, n, -1, dt, -1);
}

PyMethodDef kMethods[] = {
    {"samgraph_torch_get_graph_feat", get_graph_feat, METH_VARARGS, "f32[U, D] on the trainer device (adapter.cc:48)"},
    {"samgraph_torch_get_graph_label", get_graph_label, METH_VARARGS, "i64[B] on the trainer device (adapter.cc:66)"},
    {"samgraph_torch_get_graph_row", get_layer_ids<samgraph_torch_get_graph_row_ptr>, METH_VARARGS,
     "i32[E_l] on the trainer device (adapter.cc:82)"},
    {"samgraph_torch_get_graph_col", get_layer_ids<samgraph_torch_get_graph_col_ptr>, METH_VARARGS,
     "i32[E_l] on the trainer device (adapter.cc:96)"},
    {"samgraph_torch_get_graph_data", get_layer_ids<samgraph_torch_get_graph_data_ptr>, METH_VARARGS,
     "i32[E_l] random-walk visit counts (adapter.cc:110)"},
    {"samgraph_torch_get_dataset_feat", get_dataset_feat, METH_NOARGS, "f32[N, D] host (adapter.cc:124)"},
    {"samgraph_torch_get_dataset_label", get_dataset_label, METH_NOARGS, "i64[N] host (adapter.cc:139)"},
    {"samgraph_torch_get_graph_input_nodes", get_ids<samgraph_torch_get_graph_input_nodes_ptr>, METH_VARARGS,
     "i32[U] on the sampler device (adapter.cc:153)"},
    {"samgraph_torch_get_graph_output_nodes", get_ids<samgraph_torch_get_graph_output_nodes_ptr>, METH_VARARGS,
     "i32[B] on the sampler device (adapter.cc:168)"},
    {nullptr, nullptr, 0, nullptr}};

PyModuleDef kModule = {PyModuleDef_HEAD_INIT, "c_lib",
                       "tensor getters of the MI355X sampling engine under the reference's names", -1, kMethods,
                       nullptr, nullptr, nullptr, nullptr};

}  // namespace

extern "C" __attribute__((visibility("default"))) PyObject *PyInit_c_lib(void) {
  if (!resolve_python()) return nullptr;  // not inside a CPython process
  PyObject *m = py.ModuleCreate2(&kModule, PYTHON_API_VERSION);
  if (!m) return nullptr;
  PyObject *d = py.ModuleGetDict(m);  // borrowed
  if (!d || py.DictSetItemString(d, "__builtins__", py.EvalGetBuiltins()) != 0) {
    py.DecRef(m);
    return nullptr;
  }
  PyObject *r = py.RunString(kHelper, Py_file_input, d, d, nullptr);
  if (!r) {
    py.DecRef(m);
    return nullptr;
  }
  py.DecRef(r);
  g_wrap = py.DictGetItemString(d, "_wrap");  // borrowed; the module dict keeps it alive
  if (!g_wrap) {
    py.DecRef(m);
    return nullptr;
  }
  return m;
}
