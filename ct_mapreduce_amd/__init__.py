"""ct_mapreduce_amd — MI355X-native map/reduce hot path of jcjones/ct-mapreduce.

The product is libctmr.so (hand-written HIP for gfx950 behind the C ABI of include/ctmr.h).
This package is the Python host mirror over that ABI: `Engine` (batched insertCTWorker +
RemoteCache set methods), `storage` (KnownCertificates / IssuerMetadata-style helpers with the
reference's names) and `distributed` (log-index sharding over torch.distributed/RCCL).
"""
from . import _native  # noqa: F401
from .engine import Engine, Batch, BatchResult, CtmrError  # noqa: F401
