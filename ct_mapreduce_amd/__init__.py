"""ct_mapreduce_amd — MI355X-native map/reduce hot path of jcjones/ct-mapreduce.

The product is libctmr.so (hand-written HIP for gfx950 behind the C ABI of include/ctmr.h).
This package is the Python binding of that ABI: `Engine` (batched insertCTWorker + RemoteCache set
methods), `remote_cache` (storage.RemoteCache over the engine + the Redis-protocol export of the sets)
and `distributed` (the native ctmr_group_* layer: log-index shards, global dedup over RCCL).
"""
from . import _native  # noqa: F401
from .engine import Engine, Batch, BatchResult, CtmrError  # noqa: F401
