"""ctypes binding of libartmips.so (C ABI: include/artmips.h).

This is the whole FFI surface: plain pointers and sizes, no torch types.  The
library is built in-tree by ``art_amd/csrc/build.sh`` (``__graft_entry__.build``)
and there is no fallback: if it is missing, importing this module fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product loads lib/libartmips.so and nothing else.  Another build of the same ABI (the measurement
# scripts under tools/ use lib/libartmips_diag.so, the only build that reads the ARTMIPS_* diagnostic
# knobs) is loaded only on a double opt-in: ARTMIPS_DIAGNOSTICS=1 AND ARTMIPS_LIBRARY=path.
# A stray ARTMIPS_LIBRARY alone changes nothing (it is reported once and ignored).
LIB_PATH = os.path.join(_HERE, "lib", "libartmips.so")
if os.environ.get("ARTMIPS_LIBRARY"):
    if os.environ.get("ARTMIPS_DIAGNOSTICS") == "1":
        LIB_PATH = os.environ["ARTMIPS_LIBRARY"]
    else:
        import warnings

        warnings.warn("ARTMIPS_LIBRARY is set but ARTMIPS_DIAGNOSTICS=1 is not: ignored, loading %s" % LIB_PATH,
                      RuntimeWarning)
DIAG_LIB_PATH = os.path.join(_HERE, "lib", "libartmips_diag.so")
NOALLOC_SCRATCH_BYTES = 64 << 20  # ARTMIPS_NOALLOC_SCRATCH_BYTES

F16, BF16, F32 = 0, 1, 2
MAX_DEVICES = 16
MAX_K = 2048
SORT_MAX = 8192
PACK_TAIL = 4  # ARTMIPS_PACK_TAIL
ID_SENTINEL = -2139062144  # 0x80808080: id slot of a row that was never uploaded (artmips.h)

# every symbol include/artmips.h declares; tests check that the library exports each of them
SYMBOLS = (
    "artmips_create",
    "artmips_destroy",
    "artmips_begin",
    "artmips_upload",
    "artmips_upload_device",
    "artmips_upload_scatter",
    "artmips_finalize",
    "artmips_search",
    "artmips_search_async",
    "artmips_search_local",
    "artmips_search_local_noalloc",
    "artmips_merge",
    "artmips_packed_words",
    "artmips_search_local_async",
    "artmips_merge_packed",
    "artmips_merge_packed_sync",
    "artmips_fetch_rows",
    "artmips_generation",
    "artmips_download",
    "artmips_grow_lists",
    "artmips_set_stats",
    "artmips_info",
    "artmips_last_error",
    "artmips_version",
)


class ArtmipsInfo(ctypes.Structure):
    _fields_ = [
        ("n_dev", ctypes.c_int32),
        ("dim", ctypes.c_int32),
        ("storage_dtype", ctypes.c_int32),
        ("n_cu", ctypes.c_int32),
        ("n_rows", ctypes.c_int64),
        ("rows_per_shard", ctypes.c_int64 * MAX_DEVICES),
        ("index_bytes", ctypes.c_int64),
        ("workspace_bytes", ctypes.c_int64),
        ("last_scan_ms", ctypes.c_float),
        ("last_total_ms", ctypes.c_float),
        ("last_scan_launches", ctypes.c_int32),
        ("last_retries", ctypes.c_int32),
        ("last_candidates", ctypes.c_int64),
        ("last_host_enqueue_us", ctypes.c_float),
        ("last_host_sync_us", ctypes.c_float),
    ]


class ArtmipsError(RuntimeError):
    """A libartmips call returned a negative status; carries the C status string."""

    def __init__(self, code, message):
        super().__init__("artmips error %d: %s" % (code, message))
        self.code = code


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libartmips.so not found at %s. Build it with art_amd/csrc/build.sh "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no CPU fallback for the search path." % LIB_PATH
        )
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    lib.artmips_create.argtypes = [ctypes.POINTER(ctypes.c_int), i32, i32, i32, ctypes.POINTER(vp)]
    lib.artmips_create.restype = i32
    lib.artmips_destroy.argtypes = [vp]
    lib.artmips_destroy.restype = None
    lib.artmips_begin.argtypes = [vp, i64]
    lib.artmips_begin.restype = i32
    lib.artmips_upload.argtypes = [vp, vp, vp, i64, i32]
    lib.artmips_upload.restype = i32
    lib.artmips_upload_device.argtypes = [vp, i32, vp, vp, i64, i64, vp]
    lib.artmips_upload_device.restype = i32
    lib.artmips_upload_scatter.argtypes = [vp, i32, vp, i32, vp, i64, i64, vp]
    lib.artmips_upload_scatter.restype = i32
    lib.artmips_finalize.argtypes = [vp]
    lib.artmips_finalize.restype = i32
    lib.artmips_search.argtypes = [vp, vp, i32, i64, i32, vp, i32, vp, vp]
    lib.artmips_search.restype = i32
    lib.artmips_search_async.argtypes = [vp, vp, i32, i64, i32, vp, i32, vp, vp, vp]
    lib.artmips_search_async.restype = i32
    lib.artmips_search_local_noalloc.argtypes = [vp, i32, vp, i32, i64, i32, vp, vp]
    lib.artmips_search_local_noalloc.restype = i32
    lib.artmips_search_local.argtypes = [vp, i32, vp, i32, i64, i32, vp, vp, vp]
    lib.artmips_search_local.restype = i32
    lib.artmips_merge.argtypes = [i32, vp, vp, i32, i64, i32, vp, i32, vp, vp]
    lib.artmips_merge.restype = i32
    lib.artmips_packed_words.argtypes = [i64, i32]
    lib.artmips_packed_words.restype = i64
    lib.artmips_search_local_async.argtypes = [vp, i32, vp, i32, i64, i32, vp, vp]
    lib.artmips_search_local_async.restype = i32
    lib.artmips_merge_packed.argtypes = [i32, vp, i32, i64, i32, vp, i32, vp, vp, vp]
    lib.artmips_merge_packed.restype = i32
    lib.artmips_merge_packed_sync.argtypes = [i32, vp, i32, i64, i32, vp, i32, vp, ctypes.POINTER(ctypes.c_int32), vp]
    lib.artmips_merge_packed_sync.restype = i32
    lib.artmips_fetch_rows.argtypes = [vp, vp, i64, vp]
    lib.artmips_fetch_rows.restype = i32
    lib.artmips_generation.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    lib.artmips_generation.restype = i32
    lib.artmips_download.argtypes = [vp, ctypes.c_uint64, i64, i64, vp, vp]
    lib.artmips_download.restype = i32
    lib.artmips_grow_lists.argtypes = [vp]
    lib.artmips_grow_lists.restype = i32
    lib.artmips_set_stats.argtypes = [vp, i32]
    lib.artmips_set_stats.restype = i32
    lib.artmips_info.argtypes = [vp, ctypes.POINTER(ArtmipsInfo)]
    lib.artmips_info.restype = i32
    lib.artmips_last_error.argtypes = []
    lib.artmips_last_error.restype = ctypes.c_char_p
    lib.artmips_version.argtypes = []
    lib.artmips_version.restype = i32
    return lib


lib = _load()


def check(rc):
    if rc != 0:
        raise ArtmipsError(rc, lib.artmips_last_error().decode("utf-8", "replace"))
