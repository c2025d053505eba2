"""Host-side mirror of the reference's samgraph/common/__init__.py (enum block :47-265, ctypes binding
:268-500): same constant names and values, same `SamGraphBasics` method surface, bound to the
`samgraph_*` C ABI of this repo's HIP engine library instead of the CUDA extension.

The numeric values are part of the API contract (scripts index profiler tables with them); they are
pinned by tests/test_python_constants.py against tests/golden/py_constants.json, which was produced by
importing the reference module in the build container.
"""
import ctypes
import os
import sys

_THIS = sys.modules[__name__]

# The engine runs up to five HIP streams per process; the HIP runtime maps a process's streams onto 4 hardware queues
# unless told otherwise, and reads the variable at its first call -- usually after this import (samgraph_config sets the
# same default for C callers; a value set by the user wins)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# the samplers' HBM message rings are shared with the trainer processes through hipIpcGetMemHandle / hipIpcOpenMemHandle,
# which on hosts whose driver only supports dmabuf IPC fail ("invalid argument") unless the legacy IPC mode is off -- the
# hand-off then falls back to the reference's pinned host ring.  Read by the runtime at its first call; a user's value wins
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def _enum(names, start=0):
    for i, n in enumerate(names):
        setattr(_THIS, n, start + i)
    return len(names)


# DeviceType, SampleType, RunArch, CachePolicy -- common.h:38-92
_enum(["kCPU", "kMMAP", "kGPU"])
_enum(["kKHop0", "kKHop1", "kWeightedKHop", "kRandomWalk", "kWeightedKHopPrefix", "kKHop2", "kWeightedKHopHashDedup"])
_enum(["kArch%d" % i for i in range(8)])
_enum(["kCacheByDegree", "kCacheByHeuristic", "kCacheByPreSample", "kCacheByDegreeHop", "kCacheByPreSampleStatic",
       "kCacheByFakeOptimal", "kDynamicCache", "kCacheByRandom"])


def cpu(device_id=0):
    return "cpu:{:}".format(device_id)


def gpu(device_id=0):
    return "cuda:{:}".format(device_id)


def simple_hash(x):
    return hash(x)


sample_types = {
    "khop0": kKHop0, "khop1": kKHop1, "khop2": kKHop2, "random_walk": kRandomWalk,  # noqa: F821
    "weighted_khop": kWeightedKHop, "weighted_khop_prefix": kWeightedKHopPrefix,  # noqa: F821
    "weighted_khop_hash_dedup": kWeightedKHopHashDedup,  # noqa: F821
}

builtin_archs = {
    "arch0": {"arch": kArch0, "sampler_ctx": cpu(), "trainer_ctx": gpu(0)},  # noqa: F821
    "arch1": {"arch": kArch1, "sampler_ctx": gpu(0), "trainer_ctx": gpu(0)},  # noqa: F821
    "arch2": {"arch": kArch2, "sampler_ctx": gpu(0), "trainer_ctx": gpu(0)},  # noqa: F821
    "arch3": {"arch": kArch3, "sampler_ctx": gpu(0), "trainer_ctx": gpu(1)},  # noqa: F821
    "arch4": {"arch": kArch4, "sampler_ctx": gpu(1), "trainer_ctx": gpu(0)},  # noqa: F821
    "arch5": {"arch": kArch5},  # noqa: F821
    "arch6": {"arch": kArch6},  # noqa: F821
    "arch7": {"arch": kArch7},  # noqa: F821
}

cache_policies = {
    "degree": kCacheByDegree, "heuristic": kCacheByHeuristic, "pre_sample": kCacheByPreSample,  # noqa: F821
    "degree_hop": kCacheByDegreeHop, "presample_static": kCacheByPreSampleStatic,  # noqa: F821
    "fake_optimal": kCacheByFakeOptimal, "dynamic_cache": kDynamicCache, "random": kCacheByRandom,  # noqa: F821
}

# profiler.h:30-131 -- init / step / epoch log items and trace events
_INIT_ITEMS = ["L1Common", "L1Sampler", "L1Trainer", "L2LoadDataset", "L2DistQueue", "L2Presample", "L2InternalState",
               "L2BuildCache", "L3LoadDatasetMMap", "L3LoadDatasetCopy", "L3DistQueueAlloc", "L3DistQueuePin",
               "L3DistQueuePush", "L3PresampleInit", "L3PresampleSample", "L3PresampleCopy", "L3PresampleCount",
               "L3PresampleSort", "L3PresampleReset", "L3PresampleGetRank", "L3InternalStateCreateCtx",
               "L3InternalStateCreateStream"]
kNumLogInitItems = _enum(["kLogInit" + n for n in _INIT_ITEMS])

_STEP_ITEMS = (["L1NumSample", "L1NumNode", "L1SampleTime", "L1SendTime", "L1RecvTime", "L1CopyTime", "L1ConvertTime",
                "L1TrainTime", "L1FeatureBytes", "L1LabelBytes", "L1IdBytes", "L1GraphBytes", "L1MissBytes",
                "L1PrefetchAdvanced", "L1GetNeighbourTime", "L2ShuffleTime", "L2LastLayerTime", "L2LastLayerSize",
                "L2CoreSampleTime", "L2IdRemapTime", "L2GraphCopyTime", "L2IdCopyTime", "L2ExtractTime",
                "L2FeatCopyTime", "L2CacheCopyTime", "L3KHopSampleCooTime", "L3KHopSampleSortCooTime",
                "L3KHopSampleCountEdgeTime", "L3KHopSampleCompactEdgesTime", "L3RandomWalkSampleCooTime",
                "L3RandomWalkTopKTime"] + ["L3RandomWalkTopKStep%dTime" % i for i in range(1, 12)] +
               ["L3RemapFillUniqueTime", "L3RemapPopulateTime", "L3RemapMapNodeTime", "L3RemapMapEdgeTime",
                "L3CacheGetIndexTime", "L3CacheCopyIndexTime", "L3CacheExtractMissTime", "L3CacheCopyMissTime",
                "L3CacheCombineMissTime", "L3CacheCombineCacheTime"])
# the reference spells two names with a capital K (common/__init__.py:225,235); scripts use those spellings
_enum([("KLog" if n == "L3CacheCopyIndexTime" else "kLog") + n for n in _STEP_ITEMS])

_EPOCH_ITEMS = ["SampleTime", "SampleGetCacheMissIndexTime", "SampleSendTime", "SampleTotalTime", "CopyTime",
                "ConvertTime", "TrainTime", "TotalTime", "FeatureBytes", "MissBytes"]
_enum([("KLogEpoch" if n == "SampleGetCacheMissIndexTime" else "kLogEpoch") + n for n in _EPOCH_ITEMS])

_enum(["kL0Event_Train_Step", "kL1Event_Sample", "kL2Event_Sample_Shuffle", "kL2Event_Sample_Core",
       "kL2Event_Sample_IdRemap", "kL1Event_Copy", "kL2Event_Copy_Id", "kL2Event_Copy_Graph", "kL2Event_Copy_Extract",
       "kL2Event_Copy_FeatCopy", "kL2Event_Copy_CacheCopy", "kL3Event_Copy_CacheCopy_GetIndex",
       "kL3Event_Copy_CacheCopy_CopyIndex", "kL3Event_Copy_CacheCopy_ExtractMiss", "kL3Event_Copy_CacheCopy_CopyMiss",
       "kL3Event_Copy_CacheCopy_CombineMiss", "kL3Event_Copy_CacheCopy_CombineCache", "kL1Event_Convert",
       "kL1Event_Train"])


# ---------------------------------------------------------------------------------------------------
# ctypes binding of the samgraph_* C ABI (reference common/__init__.py:268-500; include/samgraph.h)

def _get_ext_suffix():
    import sysconfig
    return sysconfig.get_config_var('EXT_SUFFIX') or sysconfig.get_config_var('SO') or '.so'


def _get_extension_full_path(pkg_path, *args):
    """<dir of pkg_path>/<args...>.so -- the reference appends the CPython EXT_SUFFIX
    (common/__init__.py:31-36); this build's engine is a plain shared library, so `.so` is tried first."""
    assert len(args) >= 1
    dir_path = os.path.join(os.path.dirname(pkg_path), *args[:-1])
    plain = os.path.join(dir_path, args[-1] + '.so')
    if os.path.exists(plain):
        return plain
    return os.path.join(dir_path, args[-1] + _get_ext_suffix())


_u64, _int, _dbl, _sz, _cstr = ctypes.c_uint64, ctypes.c_int, ctypes.c_double, ctypes.c_size_t, ctypes.c_char_p

# name -> (restype, argtypes); argtypes/restypes as the reference declares them (common/__init__.py:273-341)
_SIGNATURES = {
    'samgraph_config': (None, (ctypes.POINTER(_cstr), ctypes.POINTER(_cstr), _sz)),
    'samgraph_init': (None, ()), 'samgraph_start': (None, ()), 'samgraph_shutdown': (None, ()),
    'samgraph_num_epoch': (_sz, ()), 'samgraph_steps_per_epoch': (_sz, ()), 'samgraph_num_class': (_sz, ()),
    'samgraph_feat_dim': (_sz, ()), 'samgraph_get_next_batch': (_u64, ()), 'samgraph_sample_once': (None, ()),
    'samgraph_get_graph_num_src': (_sz, (_u64, _int)), 'samgraph_get_graph_num_dst': (_sz, (_u64, _int)),
    'samgraph_get_graph_num_edge': (_sz, (_u64, _int)),
    'samgraph_log_step': (None, (_u64, _u64, _int, _dbl)), 'samgraph_log_step_add': (None, (_u64, _u64, _int, _dbl)),
    'samgraph_log_epoch_add': (None, (_u64, _int, _dbl)), 'samgraph_get_log_init_value': (_dbl, (_int,)),
    'samgraph_get_log_step_value': (_dbl, (_u64, _u64, _int)), 'samgraph_get_log_epoch_value': (_dbl, (_u64, _int)),
    'samgraph_report_init': (None, ()), 'samgraph_report_step': (None, (_u64, _u64)),
    'samgraph_report_step_average': (None, (_u64, _u64)), 'samgraph_report_epoch': (None, (_u64,)),
    'samgraph_report_epoch_average': (None, (_u64,)), 'samgraph_report_node_access': (None, ()),
    'samgraph_trace_step_begin': (None, (_u64, _int, _u64)), 'samgraph_trace_step_end': (None, (_u64, _int, _u64)),
    'samgraph_trace_step_begin_now': (None, (_u64, _int)), 'samgraph_trace_step_end_now': (None, (_u64, _int)),
    'samgraph_dump_trace': (None, ()), 'samgraph_forward_barrier': (None, ()), 'samgraph_data_init': (None, ()),
    'samgraph_sample_init': (None, (_int, _cstr)), 'samgraph_train_init': (None, (_int, _cstr)),
    'samgraph_extract_start': (None, (_int,)), 'samgraph_switch_init': (None, (_int, _cstr, _dbl)),
    'samgraph_num_local_step': (_sz, ()), 'samgraph_wait_one_child': (_int, ()),
}


# entry points the reference does not have (include/samgraph_ext.h); nothing reference-shaped depends on them
_EXT_SIGNATURES = {
    'samgraph_ext_queue_stats': (_int, (_int, ctypes.POINTER(_u64))),
    'samgraph_ext_ring_mapping': (_int, (_int, ctypes.POINTER(ctypes.c_int64))),
}


class SamGraphBasics(object):
    """Same method surface as the reference class (common/__init__.py:268-500)."""

    def __init__(self, pkg_path, *args):
        full_path = _get_extension_full_path(pkg_path, *args)
        if not os.path.exists(full_path):
            raise ImportError(full_path + " is missing: build it with `python __graft_entry__.py build` "
                              "(the engine has no Python or CPU fallback)")
        self.C_LIB_CTYPES = ctypes.CDLL(full_path, mode=ctypes.RTLD_GLOBAL)
        for name, (res, argt) in _SIGNATURES.items():
            fn = getattr(self.C_LIB_CTYPES, name)
            fn.restype = res
            fn.argtypes = argt
        for name, (res, argt) in _EXT_SIGNATURES.items():
            fn = getattr(self.C_LIB_CTYPES, name)
            fn.restype = res
            fn.argtypes = argt
        # every samgraph_X becomes method X, except the ones with marshalling below
        for name in _SIGNATURES:
            short = name[len('samgraph_'):]
            if not hasattr(self, short):
                setattr(self, short, getattr(self.C_LIB_CTYPES, name))

    def config(self, run_config: dict):
        keys = [str.encode(str(k)) for k in run_config.keys()]
        vals = [str.encode(' '.join(str(x) for x in v) if isinstance(v, list) else str(v))
                for v in run_config.values()]
        n = len(keys)
        return self.C_LIB_CTYPES.samgraph_config((_cstr * n)(*keys), (_cstr * n)(*vals), n)

    def ext_queue_stats(self, ring):
        """hand-off statistics of sampler `ring` (include/samgraph_ext.h), or None without a queue / ring"""
        out = (_u64 * 6)()
        if self.C_LIB_CTYPES.samgraph_ext_queue_stats(ring, out) != 0:
            return None
        return dict(zip(("ring_slots", "sent_device", "sent_host", "spilled", "verified", "check_failed"),
                        (int(x) for x in out)))

    def ext_ring_mapping(self, ring):
        """how THIS process reads sampler `ring`'s payloads (include/samgraph_ext.h), or None without a queue / ring"""
        out = (ctypes.c_int64 * 3)()
        if self.C_LIB_CTYPES.samgraph_ext_ring_mapping(ring, out) != 0:
            return None
        how = ("not read yet", "own ring", "mapped: device-to-device reads (hipIpcOpenMemHandle, lazy peer access)",
               "mapping refused: payloads copied back to the pinned host slot")[int(out[0])]
        return {"state": int(out[0]), "how": how, "ring_device": int(out[1]), "reader_device": int(out[2])}

    def sample_init(self, worker_id, ctx):
        return self.C_LIB_CTYPES.samgraph_sample_init(worker_id, str.encode(ctx))

    def train_init(self, worker_id, ctx):
        return self.C_LIB_CTYPES.samgraph_train_init(worker_id, str.encode(ctx))

    def switch_init(self, worker_id, ctx, cache_percentage):
        return self.C_LIB_CTYPES.samgraph_switch_init(worker_id, str.encode(ctx), cache_percentage)
