"""art_amd: MI355X-native exact MIPS top-k search behind ART's evidence-index API.

Public surface (mirrors /root/reference/megatron/data/art_index.py):
    detach, OpenRetreivalDataStore, DistributedBruteForceIndex
Engine level (ctypes over include/artmips.h):
    art_amd.engine.MipsEngine, art_amd.distributed.spmd_search
"""
from .art_index import DistributedBruteForceIndex, OpenRetreivalDataStore, detach

__all__ = ["detach", "OpenRetreivalDataStore", "DistributedBruteForceIndex"]
__version__ = "0.1.0"
