// artmips.hip — host side of libartmips.so (C ABI declared in include/artmips.h).
//
// Replaces the device orchestration of DistributedBruteForceIndex
// (/root/reference/megatron/data/art_index.py:200-304): shard ownership, upload,
// per-shard search pipeline, candidate exchange and merge.  No torch types; the
// Python mirror of the reference classes lives in art_amd/art_index.py.
#include "artmips_kernels.h"
#include "../../include/artmips.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>
#include <unistd.h>

using namespace artk;

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess)                                                                 \
            return fail(ARTMIPS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                        __FILE__, __LINE__);                                                   \
    } while (0)

// Diagnostic knobs (A/B timing experiments) exist only in the diagnostic build
// (-DARTMIPS_DIAG -> art_amd/lib/libartmips_diag.so).  The shipped library never reads the
// environment: a stray ARTMIPS_* variable in a training job cannot change a plan or a result.
#ifdef ARTMIPS_DIAG
static int diag_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
#else
static inline int diag_int(const char*, int dflt) { return dflt; }
#endif

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// ---------------------------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct Generation {
    void* rows = nullptr;   // [n_rows, dim] storage dtype
    int32_t* ids = nullptr; // [n_rows]
    int64_t n_rows = 0;
    int64_t row0 = 0;       // global index of the shard's first row
    unsigned long long* meta = nullptr;  // [4] device words: rows scattered, duplicates, unfilled
};

struct Shard {
    int dev = 0;
    int n_cu = 0;
    Generation live, next;
    hipStream_t stream = nullptr;  // used when the caller's stream lives on another device
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;       // whole pipeline
    std::vector<hipEvent_t> ev_scan;                   // pairs around each FILTER scan
    // workspace
    DevBuf qpad, thr, gmax, counts, lists, keys, status, qin, loc_scores, loc_ids, loc_packed;
    unsigned gang_epoch = 0;   // gang rendezvous: changes with every FILTER launch (ScanArgs::gang_epoch)
    int* status_host = nullptr;  // pinned mirror of status[4] (+ cand_total as 2 ints), written by the
                                 // last select launch of a search itself (SelectArgs::publish)
    int* status_host_dev = nullptr;  // the same words as the device sees them
};

// Device scratch of one shard pipeline: pointers into the shard's own buffers (grown on demand), or
// into the static scratch array of the allocation-free entry point.
struct Workspace {
    void* qpad = nullptr;     // [QT, dim] storage dtype (only used when the queries need converting)
    float* thr = nullptr;     // [QT]
    float* gmax = nullptr;    // pass 0: [QT][grid0 * 4 * topt]
    int* counts = nullptr;    // [QT][grid * 4]
    int2* lists = nullptr;    // [grid * nq * NT][cap]
    u64* keys = nullptr;      // select_kernel's global key scratch (may be absent)
    size_t keys_bytes = 0;
    int* status = nullptr;    // 4 words (+ pad)
};

// One persistent helper thread per extra shard of a one-process / many-shards handle: the host needs
// ~19 us to plan and enqueue one shard's pipeline (peer copy of the queries, four launches, status
// copy), so a single thread would start the eighth GPU ~135 us after the first (a fifth of a 0.65 ms
// step).  The workers only enqueue; the calling thread collects.
struct ShardWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = false, stop = false;
    int rc = 0;
    char err[512] = "";

    void loop() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return has_job || stop; });
            if (stop) return;
            std::function<int()> j = std::move(job);
            has_job = false;
            lk.unlock();
            const int r = j();
            lk.lock();
            rc = r;
            if (r) {
                strncpy(err, g_err, sizeof(err) - 1);
                err[sizeof(err) - 1] = 0;
            }
            done = true;
            cv.notify_all();
        }
    }
    void post(std::function<int()> j) {
        {
            std::lock_guard<std::mutex> lk(m);
            job = std::move(j);
            has_job = true;
            done = false;
        }
        cv.notify_all();
    }
    // returns the job's status; on failure the worker's message becomes the caller's last error
    int wait() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return done; });
        done = false;
        return rc ? fail(rc, "%s", err) : 0;
    }
    void shutdown() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};

struct artmips {
    int n_dev = 0;
    int dim = 0;
    int dtype = 0;
    Shard sh[ARTMIPS_MAX_DEVICES];
    bool building = false;
    int64_t next_total = 0, next_chunk = 0, cursor = 0;
    int64_t n_rows = 0, chunk = 0;
    // staging for host uploads
    void* pin[2] = {nullptr, nullptr};
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    size_t pin_bytes = 0;
    std::vector<std::unique_ptr<ShardWorker>> workers;  // [n_dev], slot 0 unused; created on first use
    // artmips_download: its own pinned staging and streams, so that it can run beside searches; the
    // mutex orders it against the generation swap of artmips_finalize / artmips_destroy
    std::mutex gen_mutex;
    uint64_t generation = 0;
    void* dl_pin[2] = {nullptr, nullptr};
    size_t dl_pin_bytes = 0;
    hipStream_t dl_stream[ARTMIPS_MAX_DEVICES][2] = {};
    // gather buffer on slot 0 for the multi-shard merge: [active shards][packed words]
    DevBuf g_packed, m_keys, m_status;
    int32_t* flag_host = nullptr;  // pinned + mapped: the multi-shard merge writes its status flag here
    int32_t* flag_dev = nullptr;
    hipEvent_t ev_src[ARTMIPS_MAX_DEVICES * 4] = {};  // upload_scatter: "batch ready", one per source device
    // stats
    float last_scan_ms = 0, last_total_ms = 0;
    int last_scan_launches = 0, last_retries = 0;
    int64_t last_candidates = 0;
    int cap_boost = 1;  // sticky multiplier learnt from overflow retries
    bool stats_enabled = true;         // artmips_set_stats: record HIP events around the kernels
    bool stats_pending = false;
    bool stats_async = false;          // the pending stats belong to artmips_search_local_async
    hipStream_t stats_stream = nullptr;
    double host_enqueue_us = 0, host_sync_us = 0;
    int stats_slot = 0, stats_tiles = 0;
};

static size_t elem_size(int dtype) { return dtype == ARTMIPS_F32 ? 4 : 2; }

static int ensure(DevBuf& b, size_t bytes) {
    if (b.bytes >= bytes && b.p) return 0;
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
    size_t want = bytes;  // callers size for the plan's worst case, so growth is rare (hipFree syncs)
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return fail(ARTMIPS_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    b.bytes = want;
    return 0;
}

static void release(DevBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}

static void free_generation(Generation& g) {
    if (g.rows) (void)hipFree(g.rows);
    if (g.ids) (void)hipFree(g.ids);
    if (g.meta) (void)hipFree(g.meta);
    g = Generation();
}

// ---------------------------------------------------------------------------------------------
// kernel launch tables
// ---------------------------------------------------------------------------------------------
// mode: MODE_FILTER, MODE_GROUPMAX, or MODE_GROUPMAX4 (pass 0 keeping four values per lane; built
// where large k occurs: 2-byte storage, dim 768)
enum { MODE_GROUPMAX4 = 2 };

// grid == 0 is the PREPARE form: the kernel's dynamic-LDS attribute is set for the current device and
// nothing is launched.  artmips_create prepares every instantiation its handle can launch
// (prepare_kernels), so that no search call ever changes a function attribute: a search can then be
// enqueued under HIP graph capture, and the first call of a shape costs what every other one does.
// The flags keep the lazy path as a fallback for a device the handle was not created for.
template <typename T, int DIM, int NSTAGE, int AUX, int NQ = 1, int STG = 1, int NW = 4>
static int launch_scan_t(int mode, const ScanArgs& a, int grid, hipStream_t st) {
    const size_t lds = (size_t)NSTAGE * TILE_ROWS * DIM * sizeof(T) + STAGING_BYTES;
    // set once per device and instantiation; shard worker threads may race here, which is harmless
    // (the attribute call is idempotent) as long as the flag itself is not torn
    static std::atomic<bool> attr_done[ARTMIPS_MAX_DEVICES * 4][3] = {};
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= ARTMIPS_MAX_DEVICES * 4) return fail(ARTMIPS_ERR_ARG, "device %d", dev);
    auto go = [&](auto kfn, int which) -> int {
        if (!attr_done[dev][which]) {
            HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds));
            attr_done[dev][which] = true;
        }
        if (grid > 0) hipLaunchKernelGGL(kfn, dim3(grid), dim3(NW * 64), lds, st, a);
        return 0;
    };
    int rc = 0;
    if (mode == MODE_FILTER) {
        rc = go(scan_kernel<T, DIM, NSTAGE, MODE_FILTER, AUX, NQ, STG, NW>, 0);
    } else if (mode == MODE_GROUPMAX4) {
        if constexpr (DIM == 768 && sizeof(T) == 2 && STG == 0) {
            rc = go(scan_kernel<T, DIM, NSTAGE, MODE_GROUPMAX, AUX, NQ, STG, NW, 4>, 2);
        } else {
            return fail(ARTMIPS_ERR_UNSUPPORTED, "internal: no top-4 pass 0 for this shape");
        }
    } else {
        rc = go(scan_kernel<T, DIM, NSTAGE, MODE_GROUPMAX, AUX, NQ, STG, NW>, 1);
    }
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}

// Non-temporal DMA loads (AUX = 2: the rows are read exactly once per call,
// profiles/r01_b_scan_sweep.txt).
// stg: stage survivors in LDS (1, few survivors per lane: k <= 128) or store them straight to the
// HBM lists (0, many survivors per lane: the staging would flush all the time)
template <typename T, int STG>
static int launch_scan_dt(int dim, int nq, int nw, int mode, const ScanArgs& a, int grid, hipStream_t st) {
    if (dim == 768 && nw == 8) {
        if (nq == 2 && a.gang > 1) return launch_scan_t<T, 768, 6, 0, 2, 0, 8>(mode, a, grid, st);
        if (nq == 2 && a.gang <= 1) return launch_scan_t<T, 768, 6, 2, 2, 0, 8>(mode, a, grid, st);
    }
    // gang launches re-read every tile from the XCD's L2: default cache policy instead of nt
    if (a.gang > 1 && dim == 768 && nw == 4 && nq == 4) return launch_scan_t<T, 768, 6, 0, 4, STG>(mode, a, grid, st);
    if (a.gang > 1) return fail(ARTMIPS_ERR_UNSUPPORTED, "internal: no gang kernel for this shape");
    if (nw != 4) return fail(ARTMIPS_ERR_UNSUPPORTED, "no %d-wave scan kernel for dim=%d nq=%d", nw, dim, nq);
    if (dim == 768) {
        if (nq == 1) {
            return launch_scan_t<T, 768, 6, 2, 1, STG>(mode, a, grid, st);
        }
        if (nq == 2) return launch_scan_t<T, 768, 6, 2, 2, STG>(mode, a, grid, st);
        if (nq == 3) return launch_scan_t<T, 768, 6, 2, 3, STG>(mode, a, grid, st);
        if (nq == 4) return launch_scan_t<T, 768, 6, 2, 4, STG>(mode, a, grid, st);
    }
    if (dim == 1024) {
        if (nq == 1) return launch_scan_t<T, 1024, 4, 2, 1, STG>(mode, a, grid, st);
        if (nq == 2) return launch_scan_t<T, 1024, 4, 2, 2, STG>(mode, a, grid, st);
    }
    // the other multiples of 128 below 1,024: one column group per sweep (a search with more than 64
    // queries streams the shard once per 64 of them); ring depth = what 144 KiB of LDS hold
    if (nq == 1) {
        if (dim == 256) return launch_scan_t<T, 256, 8, 2, 1, STG>(mode, a, grid, st);
        if (dim == 384) return launch_scan_t<T, 384, 8, 2, 1, STG>(mode, a, grid, st);
        if (dim == 512) return launch_scan_t<T, 512, 8, 2, 1, STG>(mode, a, grid, st);
        if (dim == 640) return launch_scan_t<T, 640, 7, 2, 1, STG>(mode, a, grid, st);
        if (dim == 896) return launch_scan_t<T, 896, 5, 2, 1, STG>(mode, a, grid, st);
    }
    if (dim == 128) {
        if (nq == 1) return launch_scan_t<T, 128, 8, 2, 1, STG>(mode, a, grid, st);
        if (nq == 2) return launch_scan_t<T, 128, 8, 2, 2, STG>(mode, a, grid, st);
        if (nq == 3) return launch_scan_t<T, 128, 8, 2, 3, STG>(mode, a, grid, st);
        if (nq == 4) return launch_scan_t<T, 128, 8, 2, 4, STG>(mode, a, grid, st);
    }
    return fail(ARTMIPS_ERR_UNSUPPORTED, "no scan kernel for dim=%d nq=%d", dim, nq);
}

static int launch_scan(int dtype, int dim, int nq, int nw, bool stage_lds, int mode, const ScanArgs& a,
                       int grid, hipStream_t st) {
    if (dtype == ARTMIPS_F16)
        return stage_lds ? launch_scan_dt<_Float16, 1>(dim, nq, nw, mode, a, grid, st)
                         : launch_scan_dt<_Float16, 0>(dim, nq, nw, mode, a, grid, st);
    if (dtype == ARTMIPS_BF16)
        return stage_lds ? launch_scan_dt<__bf16, 1>(dim, nq, nw, mode, a, grid, st)
                         : launch_scan_dt<__bf16, 0>(dim, nq, nw, mode, a, grid, st);
    if (dtype == ARTMIPS_F32 && nq == 1 && nw == 4) {
        // fp32 rows: 48 KiB stages, exact-f32 MFMA (1/16 of the f16 rate): one column group only
        if (dim == 768) return launch_scan_t<float, 768, 3, 2, 1, 1>(mode, a, grid, st);
        if (dim == 128) return launch_scan_t<float, 128, 8, 2, 1, 1>(mode, a, grid, st);
    }
    return fail(ARTMIPS_ERR_UNSUPPORTED, "no scan kernel for storage_dtype=%d dim=%d nq=%d", dtype, dim, nq);
}

// column groups per pass the kernels are built for at this width
static int max_nq(int dim, int dtype) {
    if (dtype == ARTMIPS_F32) return 1;
    if (dim == 768 || dim == 128) return MAX_NQ;
    return dim == 1024 ? 2 : 1;
}

static bool dim_supported(int dim, int dtype) {
    if (dtype == ARTMIPS_F32) return dim == 768 || dim == 128;
    return dim >= 128 && dim <= 1024 && dim % 128 == 0;   // f16 / bf16: every multiple of 128 up to 1,024
}

static std::atomic<bool> g_kth_attr[ARTMIPS_MAX_DEVICES * 4] = {};

// With device `dev` current: set the function attributes of every kernel instantiation a handle of
// this (dim, dtype) can launch (see launch_scan_t).  Shapes that are not built answer
// ARTMIPS_ERR_UNSUPPORTED from the dispatch, which is not an error here.
static int prepare_kernels(int dev, int dim, int dtype) {
    ScanArgs a;
    memset(&a, 0, sizeof(a));
    for (int gang = 1; gang <= 2; ++gang) {       // gang > 1 selects the gang instantiations
        a.gang = gang;
        for (int nw = 4; nw <= 8; nw += 4)
            for (int nq = 1; nq <= MAX_NQ; ++nq)
                for (int stg = 0; stg <= 1; ++stg)
                    for (int mode = 0; mode <= 2; ++mode) {
                        const int rc = launch_scan(dtype, dim, nq, nw, stg != 0, mode, a, 0, nullptr);
                        if (rc == ARTMIPS_ERR_HIP) return rc;
                    }
    }
    if (dev >= 0 && dev < ARTMIPS_MAX_DEVICES * 4 && !g_kth_attr[dev]) {
        // 16,384 sample values (k in (1024, 2048]) are 64 KiB of dynamic LDS on top of the static
        // histogram: above the default limit of a launch
        HIP_TRY(hipFuncSetAttribute((const void*)kth_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    128 * 1024));
        g_kth_attr[dev] = true;
    }
    g_err[0] = 0;   // the "unsupported" answers above are not this call's error
    return 0;
}

// ---------------------------------------------------------------------------------------------
// static device scratch of artmips_search_local_noalloc
// ---------------------------------------------------------------------------------------------
// Callers that may not allocate (HIP graph capture, serving loops, harnesses that forbid hipMalloc)
// get the per-shard workspace from this array instead of the shard's grown-on-demand buffers.  Every
// GPU the library runs on holds its own copy until the process ends.  In host code the NAME of the
// array is a host-side stand-in, not a device address: the address on each device is asked for once
// (artmips_create, with that device current) and kept in g_scratch_ptr; only that pointer is ever
// handed to kernels.
static __device__ __align__(256) unsigned char scratch[ARTMIPS_NOALLOC_SCRATCH_BYTES];
static void* g_scratch_ptr[ARTMIPS_MAX_DEVICES * 4] = {};
static std::mutex g_scratch_mutex;

// with device `dev` current: look the array's address on that device up once
static int scratch_init_device(int dev) {
    if (dev < 0 || dev >= ARTMIPS_MAX_DEVICES * 4) return fail(ARTMIPS_ERR_ARG, "device %d", dev);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[dev]) return 0;
    void* ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr)
        return fail(ARTMIPS_ERR_HIP, "hipGetSymbolAddress(scratch) on device %d failed: %s", dev,
                    hipGetErrorString(e));
    g_scratch_ptr[dev] = ptr;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// create / destroy
// ---------------------------------------------------------------------------------------------
extern "C" int artmips_version(void) { return 2; }

extern "C" const char* artmips_last_error(void) { return g_err; }

extern "C" int artmips_create(const int* device_ids, int n_dev, int dim, int storage_dtype,
                              artmips_t** out) {
    if (!out) return fail(ARTMIPS_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!device_ids || n_dev < 1 || n_dev > ARTMIPS_MAX_DEVICES)
        return fail(ARTMIPS_ERR_ARG, "n_dev must be in [1, %d]", ARTMIPS_MAX_DEVICES);
    if (storage_dtype != ARTMIPS_F16 && storage_dtype != ARTMIPS_BF16 && storage_dtype != ARTMIPS_F32)
        return fail(ARTMIPS_ERR_UNSUPPORTED, "storage dtype %d: rows are f16 (0), bf16 (1) or f32 (2)",
                    storage_dtype);
    if (!dim_supported(dim, storage_dtype))
        return fail(ARTMIPS_ERR_UNSUPPORTED,
                    "dim %d: kernels are built for the multiples of 128 up to 1024 (f16/bf16) or 128 and 768 (f32)", dim);
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    for (int i = 0; i < n_dev; ++i)
        if (device_ids[i] < 0 || device_ids[i] >= count)
            return fail(ARTMIPS_ERR_ARG, "device id %d out of range (%d devices)", device_ids[i],
                        count);
    artmips* h = new (std::nothrow) artmips();
    if (!h) return fail(ARTMIPS_ERR_NOMEM, "out of host memory");
    h->n_dev = n_dev;
    h->dim = dim;
    h->dtype = storage_dtype;
    for (int i = 0; i < n_dev; ++i) {
        Shard& s = h->sh[i];
        s.dev = device_ids[i];
        DeviceGuard g(s.dev);
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, s.dev) != hipSuccess) {
            artmips_destroy(h);
            return fail(ARTMIPS_ERR_HIP, "hipGetDeviceProperties(%d) failed", s.dev);
        }
        s.n_cu = prop.multiProcessorCount;
        if (scratch_init_device(s.dev) != 0) {
            artmips_destroy(h);
            return ARTMIPS_ERR_HIP;
        }
        if (prepare_kernels(s.dev, dim, storage_dtype) != 0) {
            artmips_destroy(h);
            return ARTMIPS_ERR_HIP;
        }
        if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&s.ev_ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming) != hipSuccess ||
            hipEventCreate(&s.ev_t0) != hipSuccess || hipEventCreate(&s.ev_t1) != hipSuccess ||
            hipHostMalloc((void**)&s.status_host, 64, hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void**)&s.status_host_dev, s.status_host, 0) != hipSuccess) {
            artmips_destroy(h);
            return fail(ARTMIPS_ERR_HIP, "stream/event creation failed on device %d", s.dev);
        }
        memset(s.status_host, 0, 64);
    }
    // One process, several devices (the reference's model, art_index.py:204,254-255): queries and
    // candidate lists travel between the shards' devices by peer copies.
    for (int i = 0; i < n_dev; ++i) {
        for (int j = 0; j < n_dev; ++j) {
            const int a = h->sh[i].dev, b = h->sh[j].dev;
            if (a == b) continue;
            // Best effort: no kernel of this library dereferences another device's memory; queries,
            // candidate lists and scatter batches cross devices by hipMemcpyPeerAsync only, which the
            // runtime stages through the host where peer access is unavailable (IOMMU / ACS, mixed
            // topologies).  Peer access makes those copies direct; a platform without it still works,
            // and a copy that really cannot be made fails where it is issued.
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, a, b) != hipSuccess || !can) {
                (void)hipGetLastError();
                fprintf(stderr, "[artmips] no peer access from device %d to device %d: copies between them "
                                "will be staged by the runtime\n", a, b);
                continue;
            }
            DeviceGuard g(a);
            const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
            if (e != hipSuccess) {
                (void)hipGetLastError();  // hipErrorPeerAccessAlreadyEnabled (torch or an earlier handle): fine
                if (e != hipErrorPeerAccessAlreadyEnabled)
                    fprintf(stderr, "[artmips] hipDeviceEnablePeerAccess(%d -> %d): %s; copies between them "
                                    "will be staged by the runtime\n", a, b, hipGetErrorString(e));
            }
        }
    }
    {
        DeviceGuard g(h->sh[0].dev);
        if (hipHostMalloc((void**)&h->flag_host, 64, hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void**)&h->flag_dev, h->flag_host, 0) != hipSuccess) {
            artmips_destroy(h);
            return fail(ARTMIPS_ERR_HIP, "pinned flag word allocation failed");
        }
        memset(h->flag_host, 0, 64);
    }
    *out = h;
    return 0;
}

extern "C" void artmips_destroy(artmips_t* h) {
    if (!h) return;
    for (auto& w : h->workers)
        if (w) w->shutdown();
    h->workers.clear();
    for (int i = 0; i < h->n_dev; ++i) {
        Shard& s = h->sh[i];
        DeviceGuard g(s.dev);
        (void)hipDeviceSynchronize();
        free_generation(s.live);
        free_generation(s.next);
        DevBuf* bufs[] = {&s.qpad, &s.thr, &s.gmax, &s.counts, &s.lists, &s.keys,
                          &s.status, &s.qin, &s.loc_scores, &s.loc_ids, &s.loc_packed};
        for (DevBuf* b : bufs) release(*b);
        for (hipEvent_t e : s.ev_scan) (void)hipEventDestroy(e);
        if (s.ev_ready) (void)hipEventDestroy(s.ev_ready);
        if (s.ev_done) (void)hipEventDestroy(s.ev_done);
        if (s.ev_t0) (void)hipEventDestroy(s.ev_t0);
        if (s.ev_t1) (void)hipEventDestroy(s.ev_t1);
        if (s.stream) (void)hipStreamDestroy(s.stream);
        if (s.status_host) (void)hipHostFree(s.status_host);
    }
    {
        DeviceGuard g(h->sh[0].dev);
        release(h->g_packed);
        release(h->m_keys);
        release(h->m_status);
        if (h->flag_host) (void)hipHostFree(h->flag_host);
        for (int i = 0; i < 2; ++i) {
            if (h->pin[i]) (void)hipHostFree(h->pin[i]);
            if (h->pin_ev[i]) (void)hipEventDestroy(h->pin_ev[i]);
            if (h->dl_pin[i]) (void)hipHostFree(h->dl_pin[i]);
        }
    }
    for (int i = 0; i < h->n_dev; ++i) {
        DeviceGuard g(h->sh[i].dev);
        for (int b = 0; b < 2; ++b)
            if (h->dl_stream[i][b]) (void)hipStreamDestroy(h->dl_stream[i][b]);
    }
    for (int d = 0; d < ARTMIPS_MAX_DEVICES * 4; ++d)
        if (h->ev_src[d]) {
            DeviceGuard g(d);
            (void)hipEventDestroy(h->ev_src[d]);
        }
    delete h;
}

// ---------------------------------------------------------------------------------------------
// building an index generation
// ---------------------------------------------------------------------------------------------
extern "C" int artmips_begin(artmips_t* h, int64_t n_rows_total) {
    if (!h) return fail(ARTMIPS_ERR_ARG, "handle is NULL");
    if (n_rows_total < 1) return fail(ARTMIPS_ERR_ARG, "n_rows_total must be >= 1");
    if (n_rows_total > 0x7fffffffLL * (int64_t)h->n_dev)
        return fail(ARTMIPS_ERR_ARG, "more than 2^31-1 rows per shard");
    // torch.chunk semantics (art_index.py:251-253): chunk = ceil(N/G); trailing shards may be
    // short or empty
    const int64_t chunk = (n_rows_total + h->n_dev - 1) / h->n_dev;
    for (int i = 0; i < h->n_dev; ++i) {
        Shard& s = h->sh[i];
        DeviceGuard g(s.dev);
        free_generation(s.next);
        const int64_t lo = std::min<int64_t>(chunk * i, n_rows_total);
        const int64_t hi = std::min<int64_t>(chunk * (i + 1), n_rows_total);
        s.next.row0 = lo;
        s.next.n_rows = hi - lo;
        if (s.next.n_rows > 0) {
            // whole tiles: the scan's DMA reads 16-row tiles without a clamp (artmips_kernels.h)
            const size_t row_bytes = (size_t)h->dim * elem_size(h->dtype);
            const int64_t n_pad = (s.next.n_rows + TILE_ROWS - 1) / TILE_ROWS * TILE_ROWS;
            const size_t rb = (size_t)n_pad * row_bytes;
            hipError_t e = hipMalloc(&s.next.rows, rb);
            if (e == hipSuccess && n_pad > s.next.n_rows)
                e = hipMemsetAsync((char*)s.next.rows + (size_t)s.next.n_rows * row_bytes, 0,
                                   (size_t)(n_pad - s.next.n_rows) * row_bytes, s.stream);
            if (e == hipSuccess) e = hipMalloc((void**)&s.next.ids, (size_t)s.next.n_rows * 4);
            if (e == hipSuccess) e = hipMalloc((void**)&s.next.meta, 32);
            // every id starts as ID_SENTINEL; artmips_finalize refuses a generation that still has one
            // The fills run on the shard's own stream, the one the upload copies use, so that they are
            // ordered before them (s.stream is non-blocking: the null stream orders nothing against it);
            // the wait covers uploads that arrive on a caller's stream (upload_device / upload_scatter).
            if (e == hipSuccess) e = hipMemsetAsync(s.next.ids, 0x80, (size_t)s.next.n_rows * 4, s.stream);
            if (e == hipSuccess) e = hipMemsetAsync(s.next.meta, 0, 32, s.stream);
            if (e == hipSuccess) e = hipStreamSynchronize(s.stream);
            if (e != hipSuccess) {
                free_generation(s.next);
                return fail(ARTMIPS_ERR_NOMEM, "hipMalloc of shard %d (%zu bytes) failed: %s", i, rb,
                            hipGetErrorString(e));
            }
        }
    }
    h->building = true;
    h->next_total = n_rows_total;
    h->next_chunk = chunk;
    h->cursor = 0;
    return 0;
}

extern "C" int artmips_upload(artmips_t* h, const void* host_rows, const int32_t* ids,
                              int64_t n_rows, int flags) {
    (void)flags;
    if (!h || !host_rows || !ids) return fail(ARTMIPS_ERR_ARG, "NULL argument");
    if (!h->building) return fail(ARTMIPS_ERR_STATE, "artmips_upload without artmips_begin");
    if (n_rows < 0 || h->cursor + n_rows > h->next_total)
        return fail(ARTMIPS_ERR_ARG, "upload of %lld rows at cursor %lld exceeds the %lld rows announced",
                    (long long)n_rows, (long long)h->cursor, (long long)h->next_total);
    for (int64_t i = 0; i < n_rows; ++i)
        if (ids[i] == ID_SENTINEL)
            return fail(ARTMIPS_ERR_ARG, "passage id %d is reserved (the \"row not uploaded yet\" marker)",
                        (int)ID_SENTINEL);
    const size_t row_bytes = (size_t)h->dim * elem_size(h->dtype);
    // pinned double buffer: sized for ~32 MiB of rows plus their ids
    const int64_t rows_per_buf = std::max<int64_t>(1, (32ll << 20) / (int64_t)row_bytes);
    const size_t need = (size_t)rows_per_buf * (row_bytes + 4);
    {
        DeviceGuard g(h->sh[0].dev);
        if (h->pin_bytes < need) {
            for (int i = 0; i < 2; ++i) {
                if (h->pin[i]) (void)hipHostFree(h->pin[i]);
                h->pin[i] = nullptr;
                HIP_TRY(hipHostMalloc(&h->pin[i], need, hipHostMallocDefault));
                if (!h->pin_ev[i]) HIP_TRY(hipEventCreateWithFlags(&h->pin_ev[i], hipEventDisableTiming));
            }
            h->pin_bytes = need;
        }
    }
    int which = 0;
    bool used[2] = {false, false};
    int64_t done = 0;
    while (done < n_rows) {
        const int64_t gpos = h->cursor + done;
        const int slot = (int)(gpos / h->next_chunk);
        Shard& s = h->sh[slot];
        const int64_t in_shard = gpos - s.next.row0;
        const int64_t n = std::min<int64_t>(std::min<int64_t>(rows_per_buf, n_rows - done),
                                            s.next.n_rows - in_shard);
        if (used[which]) HIP_TRY(hipEventSynchronize(h->pin_ev[which]));
        char* stage = (char*)h->pin[which];
        memcpy(stage, (const char*)host_rows + (size_t)done * row_bytes, (size_t)n * row_bytes);
        memcpy(stage + (size_t)rows_per_buf * row_bytes, ids + done, (size_t)n * 4);
        {
            DeviceGuard g(s.dev);
            HIP_TRY(hipMemcpyAsync((char*)s.next.rows + (size_t)in_shard * row_bytes, stage,
                                   (size_t)n * row_bytes, hipMemcpyHostToDevice, s.stream));
            HIP_TRY(hipMemcpyAsync(s.next.ids + in_shard, stage + (size_t)rows_per_buf * row_bytes,
                                   (size_t)n * 4, hipMemcpyHostToDevice, s.stream));
            HIP_TRY(hipEventRecord(h->pin_ev[which], s.stream));
        }
        used[which] = true;
        which ^= 1;
        done += n;
    }
    for (int i = 0; i < 2; ++i)
        if (used[i]) HIP_TRY(hipEventSynchronize(h->pin_ev[i]));
    h->cursor += n_rows;
    return 0;
}

extern "C" int artmips_upload_device(artmips_t* h, int slot, const void* dev_rows,
                                     const int32_t* dev_ids, int64_t n, int64_t row_offset,
                                     void* stream) {
    if (!h || !dev_rows || !dev_ids) return fail(ARTMIPS_ERR_ARG, "NULL argument");
    if (!h->building) return fail(ARTMIPS_ERR_STATE, "artmips_upload_device without artmips_begin");
    if (slot < 0 || slot >= h->n_dev) return fail(ARTMIPS_ERR_ARG, "slot %d out of range", slot);
    Shard& s = h->sh[slot];
    if (n < 0 || row_offset < 0 || row_offset + n > s.next.n_rows)
        return fail(ARTMIPS_ERR_ARG, "rows [%lld, %lld) outside shard %d of %lld rows",
                    (long long)row_offset, (long long)(row_offset + n), slot, (long long)s.next.n_rows);
    if (n == 0) return 0;
    DeviceGuard g(s.dev);
    const size_t row_bytes = (size_t)h->dim * elem_size(h->dtype);
    // asynchronous on the caller's stream (the one that produced the rows); artmips_finalize
    // synchronises the device before the generation goes live
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync((char*)s.next.rows + (size_t)row_offset * row_bytes, dev_rows,
                           (size_t)n * row_bytes, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(s.next.ids + row_offset, dev_ids, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
    return 0;
}

template <typename T>
static void launch_scatter(const void* src, int src_dtype, const long long* ids, long long n,
                           long long id_base, const Generation& g, int dim, hipStream_t st) {
    // 16-byte accesses when both sides allow it (every shape the index supports does; a caller's
    // source buffer may still be unaligned), else the element-wise kernel
    const size_t src_row = (size_t)dim * elem_size(src_dtype), dst_row = (size_t)dim * sizeof(T);
    const bool vec = (((uintptr_t)src) & 15) == 0 && src_row % 16 == 0 && dst_row % 16 == 0 &&
                     (((uintptr_t)g.rows) & 15) == 0 &&
                     (sizeof(T) == 2 || src_dtype == ARTMIPS_F32 || src_row % 8 == 0);
    if (vec) {
        // enough workgroups for every CU even from a 1,024-row batch, 8 rows each for large ones
        // (at most 2,048 workgroups: each ends with one atomic add on a word all of them share)
        const int rpw = (int)std::max<long long>(1, std::min<long long>(8, n / 1024));
        const int grid = (int)std::min<long long>((n + rpw - 1) / rpw, 2048);
#define SCATTER_VEC(S)                                                                                  \
    hipLaunchKernelGGL((scatter_rows_vec_kernel<T, S>), dim3(grid), dim3(256), 0, st, (const S*)src, ids, n, \
                       id_base, (long long)g.row0, (long long)g.n_rows, dim, (T*)g.rows, g.ids, g.meta, rpw)
        if (src_dtype == ARTMIPS_F16) SCATTER_VEC(_Float16);
        else if (src_dtype == ARTMIPS_BF16) SCATTER_VEC(__bf16);
        else SCATTER_VEC(float);
#undef SCATTER_VEC
        return;
    }
    const int grid = (int)std::min<long long>(n, 4096);
    hipLaunchKernelGGL(scatter_rows_kernel<T>, dim3(grid), dim3(256), 0, st, src, src_dtype, ids, n, id_base,
                       (long long)g.row0, (long long)g.n_rows, dim, (T*)g.rows, g.ids, g.meta);
}

extern "C" int artmips_upload_scatter(artmips_t* h, int src_device, const void* dev_rows, int src_dtype,
                                      const int64_t* dev_ids, int64_t n, int64_t id_base, void* stream) {
    if (!h || !dev_rows || !dev_ids) return fail(ARTMIPS_ERR_ARG, "NULL argument");
    if (!h->building) return fail(ARTMIPS_ERR_STATE, "artmips_upload_scatter without artmips_begin");
    if (src_dtype < 0 || src_dtype > 2) return fail(ARTMIPS_ERR_ARG, "bad source dtype %d", src_dtype);
    if (n < 0) return fail(ARTMIPS_ERR_ARG, "n must be >= 0");
    if (n == 0) return 0;
    // ids are stored as int32: every id this generation can hold must fit, and none may be the marker
    if (id_base < -0x80000000LL || id_base + h->next_total - 1 > 0x7fffffffLL)
        return fail(ARTMIPS_ERR_ARG, "passage ids [%lld, %lld] do not fit int32", (long long)id_base,
                    (long long)(id_base + h->next_total - 1));
    if (src_device < 0 || src_device >= ARTMIPS_MAX_DEVICES * 4)
        return fail(ARTMIPS_ERR_ARG, "source device %d", src_device);
    hipStream_t st = (hipStream_t)stream;
    const size_t src_bytes = (size_t)n * h->dim * elem_size(src_dtype);
    // Shards on another device than the batch get it through a staging copy on their own stream.
    // Two events tie that stream to the caller's: `ready` (created on the source device, recorded on
    // the caller's stream) lets the copy start only when the batch exists, and `ev_done` (recorded
    // behind the copy, waited for by the caller's stream) keeps stream-ordered frees of the sources
    // on the caller's stream behind the copy that still reads them.
    bool recorded = false;
    const bool force_staging = diag_int("ARTMIPS_FORCE_STAGING", 0) != 0;  // diagnostic build: tests
    for (int i = 0; i < h->n_dev; ++i) {
        Shard& s = h->sh[i];
        if (s.next.n_rows == 0) continue;
        const void* rows_here = dev_rows;
        const int64_t* ids_here = dev_ids;
        hipStream_t run = st;
        if (s.dev != src_device || force_staging) {
            if (!recorded) {
                DeviceGuard gsrc(src_device);
                if (!h->ev_src[src_device])
                    HIP_TRY(hipEventCreateWithFlags(&h->ev_src[src_device], hipEventDisableTiming));
                HIP_TRY(hipEventRecord(h->ev_src[src_device], st));
                recorded = true;
            }
            DeviceGuard g(s.dev);
            int rc = 0;
            // the previous batch may still be scattering out of this staging buffer
            HIP_TRY(hipStreamSynchronize(s.stream));
            if ((rc = ensure(s.qin, src_bytes))) return rc;
            if ((rc = ensure(s.loc_ids, (size_t)n * 8))) return rc;
            HIP_TRY(hipStreamWaitEvent(s.stream, h->ev_src[src_device], 0));
            HIP_TRY(hipMemcpyPeerAsync(s.qin.p, s.dev, dev_rows, src_device, src_bytes, s.stream));
            HIP_TRY(hipMemcpyPeerAsync(s.loc_ids.p, s.dev, dev_ids, src_device, (size_t)n * 8, s.stream));
            HIP_TRY(hipEventRecord(s.ev_done, s.stream));
            {
                DeviceGuard gsrc(src_device);
                HIP_TRY(hipStreamWaitEvent(st, s.ev_done, 0));
            }
            rows_here = s.qin.p;
            ids_here = (const int64_t*)s.loc_ids.p;
            run = s.stream;
        }
        DeviceGuard g(s.dev);
        if (h->dtype == ARTMIPS_F16)
            launch_scatter<_Float16>(rows_here, src_dtype, (const long long*)ids_here, n, id_base, s.next, h->dim, run);
        else if (h->dtype == ARTMIPS_BF16)
            launch_scatter<__bf16>(rows_here, src_dtype, (const long long*)ids_here, n, id_base, s.next, h->dim, run);
        else
            launch_scatter<float>(rows_here, src_dtype, (const long long*)ids_here, n, id_base, s.next, h->dim, run);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

extern "C" int artmips_finalize(artmips_t* h) {
    if (!h) return fail(ARTMIPS_ERR_ARG, "handle is NULL");
    if (!h->building) return fail(ARTMIPS_ERR_STATE, "artmips_finalize without artmips_begin");
    // every row must have been uploaded exactly once (the reference asserts the same after its merge:
    // indexer.py:161 and art_index.py:59,90)
    unsigned long long missing = 0, dups = 0;
    for (int i = 0; i < h->n_dev; ++i) {
        Shard& s = h->sh[i];
        if (s.next.n_rows == 0) continue;
        DeviceGuard g(s.dev);
        HIP_TRY(hipDeviceSynchronize());  // uploads landed
        hipLaunchKernelGGL(count_unfilled_kernel, dim3(1024), dim3(256), 0, 0, s.next.ids,
                           (long long)s.next.n_rows, s.next.meta + 2);
        HIP_TRY(hipGetLastError());
        unsigned long long meta[4] = {0, 0, 0, 0};
        HIP_TRY(hipMemcpy(meta, s.next.meta, 32, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemset(s.next.meta + 2, 0, 8));
        dups += meta[1];
        missing += meta[2];
    }
    if (dups) return fail(ARTMIPS_ERR_STATE, "Unexpectedly tried to overwrite block data: %llu rows were uploaded twice", dups);
    if (missing) return fail(ARTMIPS_ERR_STATE, "%llu of %lld rows were never uploaded", missing, (long long)h->next_total);
    std::lock_guard<std::mutex> swap_lock(h->gen_mutex);  // no artmips_download chunk is in flight
    ++h->generation;
    for (int i = 0; i < h->n_dev; ++i) {
        Shard& s = h->sh[i];
        DeviceGuard g(s.dev);
        HIP_TRY(hipDeviceSynchronize());  // no search is using the old rows
        free_generation(s.live);
        s.live = s.next;
        s.next = Generation();
    }
    h->n_rows = h->next_total;
    h->chunk = h->next_chunk;
    h->building = false;
    h->cap_boost = 1;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// per-shard search pipeline
// ---------------------------------------------------------------------------------------------
// a shard's status block: 16 status words, then (at STATUS_SYNC_OFFSET) the gang rendezvous' progress words,
// 8 per logical workgroup of a gang launch (at most n_cu / 2 of them)
constexpr size_t STATUS_SYNC_OFFSET = 256;
constexpr size_t STATUS_BLOCK_BYTES = STATUS_SYNC_OFFSET + 256 * 8 * 4;
constexpr int STATUS_UNPUBLISHED = -1;   // never a value of the flag word (bits 0 and 1 only)

struct Plan {
    bool pass0;
    int grid0;            // blocks of the GROUPMAX scan
    long long n_sample;   // sample tiles
    long long sample_stride;
    int grid;             // blocks of the FILTER scan
    int cap;              // entries per candidate list
    long long n_tiles;
    int nq;               // column groups per scan pass
    int nw;               // waves per scan workgroup (4 or 8); a column group is 16*nw queries
    bool stage_lds;       // stage survivors in LDS (few per lane) instead of storing them directly
    int topt;             // sample values each pass-0 lane keeps per query (1, or 4 for large k)
    int refine;           // 0, or the period P (4 or 8) of the two-part FILTER scan: every P-th tile,
                          // a threshold refinement, then the other tiles (large k)
    int gang;             // workgroups that walk the same tiles with different query blocks (1, 2, 4, 8);
                          // grid and grid0 are LOGICAL (per gang member): launches use grid * gang
};

static int round_up_pow2(long long x) {
    long long p = 1;
    while (p < x) p <<= 1;
    return (int)std::min<long long>(p, 1ll << 30);
}

static Plan make_plan(const Shard& s, int64_t n_rows, size_t row_bytes, int k, int cap_boost, int nq, int nw,
                      bool top4_built, int gang) {
    Plan p;
    p.nq = nq;
    p.nw = nw;
    p.gang = gang;
    // Large k wants 8k sample values per query.  One per lane means thousands of pass-0 workgroups,
    // each reloading the whole query operand (384 KiB at 256 queries) for a handful of tiles; four per
    // lane reach the same count with a quarter of the workgroups.
    p.topt = (top4_built && k > 128) ? 4 : 1;
    if (const int t = diag_int("ARTMIPS_TOPT", 0)) p.topt = (t == 4 && top4_built) ? 4 : 1;
    p.n_tiles = (n_rows + TILE_ROWS - 1) / TILE_ROWS;
    const int grid_mult = std::max(1, diag_int("ARTMIPS_GRID_MULT", 1));
    p.grid = (int)std::min<long long>(p.n_tiles, (long long)std::max(1, s.n_cu) * grid_mult / gang);
    p.grid = std::min(p.grid, MAX_LISTS / 4);  // select_kernel keeps one offset per list in LDS
    const long long tiles_per_block = (p.n_tiles + p.grid - 1) / p.grid;
    const long long cap_max = tiles_per_block * 4;  // rows one lane can ever see
    // pass 0: groups = lanes of the sample scan that share a query = grid0 * 4.  Want >= 8k groups
    // (so that the k-th largest group maximum is close to the k-th best of the sample) and a sample
    // of about 1/128 of the rows, but never more than 1/4 of them.
    long long groups = std::max<long long>(1024, 8ll * k);
    const long long per_wg = 4ll * p.topt;  // sample values one pass-0 workgroup contributes per query
    long long grid0 = std::min<long long>(std::max<long long>(256, (groups + per_wg - 1) / per_wg), 8192);
    if (gang > 1) grid0 = (grid0 + 7) / 8 * 8;  // gang members are dealt out over the 8 XCDs
    // sample = n_tiles / sample_div.  Pass 0 costs scan_time / sample_div; every survivor of the
    // filter costs scan time (a wave that keeps one holds its workgroup at the next barrier) and
    // select time, and the expected survivors per query are k * sample_div whatever the shard size.
    // Minimising the sum gives sample_div = sqrt(c * n_tiles / (k * column groups per wave)) with
    // c = 0.45 for 1536-byte rows, fitted on shards of 2.6M..21M rows at k = 32 and 100
    // (profiles/r01_o_sample_by_size.txt): 128 at 21M rows and k = 32, 48 on a 1/8 shard.  Since round 3
    // c = 0.70 (60 on a 1/8 shard): the selection and the survivor path have become cheaper since the fit, and a
    // coarser sample is 0.8 % faster on a 1/8 shard, 0.2-0.6 % on 8.8M-10.5M rows and in 16-query chunks, and
    // changes nothing at 21M rows (clamped to 128) or in the gang launches (profiles/r03_h_sample_constant.txt;
    // 1.4 is too much: the top-100 gangs then leave the two-part scan and lose 7 %).  fp32 rows of width 768 keep
    // 0.45: their exact-f32 scan is MFMA-bound at any batch size and measured 2 % slower with the coarser sample.
    const double row_scale = (double)row_bytes / 1536.0;
    const double groups_per_wave = (double)nq * (double)nw / 4.0;
    long long sample_div = (long long)std::llround(
        std::sqrt(0.01 * diag_int("ARTMIPS_SAMPLE_C100", row_bytes > 2048 ? 45 : 70) * row_scale * (double)p.n_tiles /
                  ((double)k * groups_per_wave)));
    sample_div = std::min<long long>(128, std::max<long long>(16, sample_div));
    if (const int sd = diag_int("ARTMIPS_SAMPLE_DIV", 0)) sample_div = std::max(4, sd);
    const int staging_slots = STAGING_BYTES / (nw * 64 * nq * 8);
    // Fills in the pass-0 part of the plan for a sample of n_tiles / sd tiles; `survivors(kept)` is the
    // expected number of candidates per query given the tiles per query-sample `kept` = n_tiles / n_sample
    auto with_sample = [&](long long sd, int period) {
        long long sample = std::max<long long>(grid0, p.n_tiles / sd);
        p.pass0 = (grid0 * per_wg >= k) && (sample * 4 <= p.n_tiles);
        p.refine = 0;
        if (!p.pass0) {
            p.stage_lds = false;
            p.grid0 = 0;
            p.n_sample = 0;
            p.sample_stride = 1;
            p.cap = (int)cap_max;  // threshold is -inf: every row is a candidate
            return;
        }
        const long long t0 = std::max<long long>(1, sample / grid0);
        p.grid0 = (int)grid0;
        p.n_sample = grid0 * t0;
        p.sample_stride = p.n_tiles / p.n_sample;
        // expected survivors per query: k * n_tiles / n_sample (loose by the group approximation) in
        // one part; in two parts the first 1/P of the tiles keeps that rate and the rest is filtered
        // at the k-th best of the first part, i.e. at rank k * P of the shard
        const double rate = (double)k * (double)p.n_tiles / (double)p.n_sample;
        const double expect = period ? rate / period + (double)k * (period - 1) : rate;
        const double per_list = expect / ((double)p.grid * 4.0);
        long long cap = round_up_pow2((long long)std::ceil(per_list * 8.0) + 16);
        cap *= cap_boost;
        p.cap = (int)std::min<long long>(cap, cap_max);
        // LDS staging holds 8/nq survivors per lane and column group; it pays while a lane flushes
        // at most a few times per launch (profiles/r01_j_lds_staging_ab.txt)
        // (8 waves x 2 groups has no registers left for the staging bookkeeping: it would spill in the loop)
        // (gang launches always write their lists directly: the staging area is where their rendezvous keeps
        // its words, a two-part scan needs direct writes, and staged gangs measured 0-4 % slower:
        // profiles/r03_e_gang_sync.txt)
        p.stage_lds = gang == 1 && !(nw == 8 && nq == 2) && per_list <= 4.0 * staging_slots;
        p.refine = period;
    };
    // Where the lists are written directly (many survivors per lane: large k, or large batches at any
    // k): pass 0 at the fitted sample_div (16 at top-1000 where the scan is MFMA-bound, 38 at top-100)
    // is 1/16 of the scan's own cost and still leaves k * 16 survivors per query.  The FILTER scan therefore runs in two
    // parts: every P-th tile first, then the k-th best score among THOSE survivors (rank k * P of the
    // whole shard) replaces the threshold for the other tiles.  That makes a 4 x coarser sample
    // affordable: with P = 8 and sample_div = 64 the two parts keep 8k + 7k candidates per query at
    // top-1000 (16k before) and pass 0 shrinks from 6.3 % to 1.6 % of the scan.  Still exact: the new
    // threshold is the score of k distinct rows already seen.  Needs direct list writes (the LDS staging
    // has no continuation across launches) and enough tiles for both parts to fill the grid.
    int period = 8;
    if (const int r = diag_int("ARTMIPS_REFINE", -1); r >= 0) period = (r == 4 || r == 8) ? r : 0;
    bool two_part = false;
    if (period && p.n_tiles >= 64ll * p.grid) {
        // (part A's survivors, k * sd_r / P per query, must fit the refinement's LDS: SCORE_MAX)
        long long sd_r = std::min<long long>(std::min<long long>(128, sample_div * 4), 20000ll * period / k);
        if (const int sd = diag_int("ARTMIPS_SAMPLE_DIV", 0)) sd_r = std::max(4, sd);
        if (sd_r >= 2 * sample_div) {   // worth a second launch only if pass 0 at least halves
            with_sample(sd_r, period);
            two_part = p.pass0 && !p.stage_lds;
        }
    }
    if (!two_part) with_sample(sample_div, 0);
    if (p.cap < 1) p.cap = 1;
    return p;
}

// ARTMIPS_DEBUG_SYNC=1 (diagnostics): wait for every kernel of the pipeline and say so on stderr, so
// that a kernel that never ends can be named.
static bool debug_sync() {
    static const bool on = diag_int("ARTMIPS_DEBUG_SYNC", 0) != 0;
    return on;
}
#define DEBUG_STEP(st, ...)                                            \
    do {                                                               \
        if (debug_sync()) {                                            \
            fprintf(stderr, "[artmips %d] ", (int)getpid());           \
            fprintf(stderr, __VA_ARGS__);                              \
            fprintf(stderr, " ...");                                   \
            fflush(stderr);                                            \
            hipError_t e__ = hipStreamSynchronize(st);                 \
            fprintf(stderr, " %s\n", hipGetErrorString(e__));          \
            fflush(stderr);                                            \
        }                                                              \
    } while (0)

template <typename T>
static void launch_pack(const void* q, int q_dtype, int q_count, int q_tile, int dim, void* out,
                        hipStream_t st) {
    hipLaunchKernelGGL(pack_queries_kernel<T>, dim3(q_tile), dim3(256), 0, st, q, q_dtype,
                       q_count, q_tile, dim, (T*)out);
}

// Enqueue the pipeline of one query tile (<= 64 queries) on stream st of shard s.
static int enqueue_tile(artmips* h, Shard& s, const Workspace& ws, const Plan& p, const void* q_tile,
                        int q_dtype, int q_count, int k, void* out_scores, int score_dtype,
                        int32_t* out_ids, long long out_row0, hipStream_t st, int tile_index, bool timed,
                        int* status, bool publish) {
    const Generation& gen = s.live;
    const int qm = p.nq * p.nw * 16;   // queries per workgroup
    const int qt = qm * p.gang;        // queries of this launch (gang members hold qm each)
    int* clear = (tile_index == 0) ? status : nullptr;  // cleared by kth / fill_thr below
    // queries already in the storage dtype are read where they lie (columns past q_count re-read the
    // last row and are never selected); anything else is converted into the padded tile first
    const void* q_src = q_tile;
    int q_rows = q_count;
    if (q_dtype != h->dtype || ((uintptr_t)q_tile & 15) != 0) {
        if (h->dtype == ARTMIPS_F16) launch_pack<_Float16>(q_tile, q_dtype, q_count, qt, h->dim, ws.qpad, st);
        else if (h->dtype == ARTMIPS_F32) launch_pack<float>(q_tile, q_dtype, q_count, qt, h->dim, ws.qpad, st);
        else launch_pack<__bf16>(q_tile, q_dtype, q_count, qt, h->dim, ws.qpad, st);
        HIP_TRY(hipGetLastError());
        DEBUG_STEP(st, "pack q_count=%d qt=%d", q_count, qt);
        q_src = ws.qpad;
        q_rows = qt;
    }

    ScanArgs a;
    memset(&a, 0, sizeof(a));
    a.rows = gen.rows;
    a.n_rows = gen.n_rows;
    a.qpad = q_src;
    a.q_rows = q_rows;
    a.q_count = q_count;
    a.thr = ws.thr;
    a.lists = ws.lists;
    a.counts = ws.counts;
    a.cap = p.cap;
    a.gmax = ws.gmax;
    a.gang = p.gang;
    a.q_tile = qm;
    a.lists_member_stride = (long long)p.grid * p.nq * (p.nw * 64) * (long long)p.cap;

    if (p.pass0) {
        a.n_work = p.n_sample;
        a.tile_stride = p.sample_stride;
        int rc = (p.topt == 4)
            ? launch_scan(h->dtype, h->dim, p.nq, p.nw, false, MODE_GROUPMAX4, a, p.grid0 * p.gang, st)
            : launch_scan(h->dtype, h->dim, p.nq, p.nw, p.stage_lds, MODE_GROUPMAX, a, p.grid0 * p.gang, st);
        if (rc) return rc;
        DEBUG_STEP(st, "scan<GROUPMAX> grid=%d nq=%d nw=%d topt=%d n_work=%lld stride=%lld", p.grid0, p.nq,
                   p.nw, p.topt, a.n_work, a.tile_stride);
        const int n_values = p.grid0 * 4 * p.topt;
        size_t kth_lds = 8;
        while (kth_lds < (size_t)n_values) kth_lds <<= 1;
        const int kth_threads = (n_values <= 4096) ? 256 : SEL_THREADS;
        if (kth_lds * 4 > 48 * 1024 && !g_kth_attr[s.dev]) {   // (set in artmips_create: prepare_kernels)
            HIP_TRY(hipFuncSetAttribute((const void*)kth_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        128 * 1024));
            g_kth_attr[s.dev] = true;
        }
        hipLaunchKernelGGL(kth_kernel, dim3(qt), dim3(kth_threads), kth_lds * 4, st,
                           ws.gmax, n_values, p.nq, k, q_count, ws.thr, clear);
        HIP_TRY(hipGetLastError());
        DEBUG_STEP(st, "kth grid=%d threads=%d", qt, kth_threads);
    } else {
        hipLaunchKernelGGL(fill_thr_kernel, dim3(1), dim3(256), 0, st, ws.thr, q_count, qt, clear);  // (loops over qt)
        HIP_TRY(hipGetLastError());
        DEBUG_STEP(st, "fill_thr q_count=%d qt=%d", q_count, qt);
    }

    SelectArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.lists = ws.lists;
    sa.counts = ws.counts;
    sa.cap = p.cap;
    sa.scan_grid = p.grid;
    sa.nq = p.nq;
    sa.threads = p.nw * 64;
    sa.q_member = p.gang > 1 ? qm : 0;
    sa.lists_member_stride = a.lists_member_stride;
    sa.ids = gen.ids;
    sa.keys_ws = ws.keys;
    sa.keys_stride = (long long)(ws.keys_bytes / 8 / (size_t)qt);
    sa.k = k;
    sa.out_scores = out_scores;
    sa.score_dtype = score_dtype;
    sa.out_ids = out_ids;
    sa.out_row0 = out_row0;
    sa.status = status;
    sa.cand_total = (unsigned long long*)(status + 2);
    sa.stop_after = diag_int("ARTMIPS_SELECT_STOP", 0);   // diagnostic build only
    // last launch of a search that wants its status on the host: the select publishes it itself
    sa.publish = publish ? s.status_host_dev : nullptr;
    sa.done_counter = ws.status + 8;   // word 8 of the shard's own status block (zeroed when allocated)

    // gang rendezvous: the progress words live behind the status words of the same block
    // (256 logical workgroups have words there; a larger grid, which only a diagnostic knob produces, runs without)
    a.gang_progress = (p.gang > 1 && p.grid <= 256) ? (unsigned int*)((char*)ws.status + STATUS_SYNC_OFFSET) : nullptr;
    a.sync_lead = diag_int("ARTMIPS_SYNC_LEAD", 0);   // diagnostic build only
    a.ablate = diag_int("ARTMIPS_ABLATE", 0);  // diagnostic build only: results are then meaningless
    a.thr_shift = 0.01f * (float)diag_int("ARTMIPS_THR_SHIFT", 0);
    if (timed) {
        while ((int)s.ev_scan.size() < 2 * (tile_index + 1)) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            s.ev_scan.push_back(e);
        }
        HIP_TRY(hipEventRecord(s.ev_scan[2 * tile_index], st));
    }
    int rc = 0;
    if (p.refine) {
        // part A: tiles 0, P, 2P, ...
        const long long part_a = (p.n_tiles + p.refine - 1) / p.refine;
        a.n_work = part_a;
        a.tile_stride = p.refine;
        a.gang_epoch = (++s.gang_epoch) & 0xFFFu;
        if ((rc = launch_scan(h->dtype, h->dim, p.nq, p.nw, p.stage_lds, MODE_FILTER, a, p.grid * p.gang, st))) return rc;
        DEBUG_STEP(st, "scan<FILTER> part A grid=%d tiles=%lld", p.grid * p.gang, a.n_work);
        // thr[q] <- max(thr[q], k-th best score among the survivors of part A)
        SelectArgs ra = sa;
        ra.refine_thr = ws.thr;
        ra.publish = nullptr;
        hipLaunchKernelGGL(select_kernel, dim3(q_count), dim3(SEL_THREADS), 0, st, ra);
        HIP_TRY(hipGetLastError());
        DEBUG_STEP(st, "refine grid=%d", q_count);
        // part B: the other tiles, appended to the same lists
        a.n_work = p.n_tiles - part_a;
        a.tile_stride = 1;
        a.tile_skip = p.refine;
        a.counts_continue = 1;
    } else {
        a.n_work = p.n_tiles;
        a.tile_stride = 1;
    }
#ifdef ARTMIPS_STAMPS_BUILD
    // ARTMIPS_STAMPS=1: clock stamps of one iteration of one workgroup of this launch, printed per wave as
    // cycles since the iteration's first stamp (tools/r3/stamps.sh); the search waits for the launch
    static unsigned long long* stamp_buf = nullptr;
    const bool stamps = diag_int("ARTMIPS_STAMPS", 0) != 0;
    if (stamps) {
        if (!stamp_buf) HIP_TRY(hipMalloc((void**)&stamp_buf, 8 * 24 * 8));
        HIP_TRY(hipMemsetAsync(stamp_buf, 0, 8 * 24 * 8, st));
        a.stamps = stamp_buf;
    }
#endif
    a.gang_epoch = (++s.gang_epoch) & 0xFFFu;
    rc = launch_scan(h->dtype, h->dim, p.nq, p.nw, p.stage_lds, MODE_FILTER, a, p.grid * p.gang, st);
    if (rc) return rc;
#ifdef ARTMIPS_STAMPS_BUILD
    if (stamps) {
        unsigned long long hst[8 * 24];
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpy(hst, stamp_buf, sizeof(hst), hipMemcpyDeviceToHost));
        for (int w = 0; w < p.nw; ++w) {
            fprintf(stderr, "[stamps nw=%d nq=%d gang=%d wave %d]", p.nw, p.nq, p.gang, w);
            for (int i = 0; i < 24; ++i)
                if (hst[w * 24 + i]) fprintf(stderr, " %d:%lld", i, (long long)(hst[w * 24 + i] - hst[w * 24]));
            fprintf(stderr, "\n");
        }
    }
#endif
    if (timed) HIP_TRY(hipEventRecord(s.ev_scan[2 * tile_index + 1], st));
    DEBUG_STEP(st, "scan<FILTER> grid=%d nq=%d nw=%d stg=%d cap=%d tiles=%lld rows=%lld q_count=%d q_rows=%d",
               p.grid, p.nq, p.nw, (int)p.stage_lds, p.cap, p.n_tiles, (long long)gen.n_rows, q_count, q_rows);

    // small k: the selection is the gather of the candidates plus a few barriers; 8 waves halve the
    // gather of 4 and keep the barriers cheap (select at the headline shape 28.6 us with 256 threads,
    // 21.3 with 512, 19.6 with 1,024; on a 1/8 shard 18.7 / 15.6 / 16.0)
    const int sel_threads = (k <= 128 && p.pass0) ? std::max(64, diag_int("ARTMIPS_SEL_THREADS", 512)) : SEL_THREADS;
    hipLaunchKernelGGL(select_kernel, dim3(q_count), dim3(sel_threads), 0, st, sa);
    HIP_TRY(hipGetLastError());
    DEBUG_STEP(st, "select grid=%d threads=%d k=%d", q_count, sel_threads, k);
    return 0;
}

// Full local search of one shard for n_q queries, enqueued on st; outputs [n_q, k].
// Returns after the stream has been synchronised and overflow retries (if any) are done.
struct ShardCall {
    int slot;
    const void* q_dev;
    int q_dtype;
    int64_t n_q;
    int k;
    void* out_scores;
    int score_dtype;
    int32_t* out_ids;
    hipStream_t st;
    bool record_stats;
    bool host_status = true;  // the last select launch publishes the status words to pinned host memory
    int* status_dev = nullptr;  // 16 status words on the shard's device, 8-byte aligned; NULL = the
                                // shard's own block (the async path raises them in the packed tail)
    unsigned char* static_scratch = nullptr;  // allocation-free form: the workspace is carved from this
                                              // device array (ARTMIPS_NOALLOC_SCRATCH_BYTES) instead
                                              // of the shard's grown-on-demand buffers
    int* status_used = nullptr;  // out: where the pipeline raises its status words
    int n_tiles_q = 0;
    std::chrono::steady_clock::time_point t_begin, t_enq;
};

// Plan, size the scratch and enqueue the whole pipeline of one shard (every query tile) plus the
// publication of its status words in pinned host memory (by the last select launch).  Does not wait.
static int shard_enqueue(artmips* h, ShardCall& c) {
    Shard& s = h->sh[c.slot];
    const Generation& gen = s.live;
    DeviceGuard g(s.dev);
    if (gen.n_rows == 0) return fail(ARTMIPS_ERR_STATE, "shard %d is empty", c.slot);
    const size_t q_row_bytes = (size_t)h->dim * elem_size(c.q_dtype);
    // The queries are cut into launches.  Shapes (waves x column groups) of one workgroup:
    //   4 x 1..3  64..192 queries, one wave per SIMD (4 x 1 is the HBM-bound training step);
    //   8 x 2     256 queries as two groups of 128: two waves per SIMD overlap one wave's DMA issue,
    //             LDS waits and filter with the other's MFMAs;
    //   4 x 4     256 queries, one wave per SIMD holding 64 queries: half the LDS read bytes per flop
    //             of 8 x 2.  Used where no 8-wave kernel is built (dim != 768) and in gangs.
    // Large batches run in gangs of 2, 4 or 8 workgroups (ScanArgs::gang): the members of a gang share
    // an XCD and walk the same tiles with different 256-query blocks, so one HBM read of the shard
    // serves 512 ... 2,048 queries and the other reads are L2 hits.  With the HBM traffic gone the
    // regime is limited by power, and for small k the 4 x 4 shape (fewer LDS bytes per flop) is the
    // faster one there: 2,048 queries x top-100 over 21M rows 62.9 ms as 8 x 2 in gangs of 4, 60.6 in
    // gangs of 8; 59.7 / 58.2 ms as 4 x 4 (without gangs 68.3 / 68.7).  At top-1000 every lane writes
    // survivors all the time and two waves per SIMD hide that better: 8 x 2 in gangs of 8 62.9 ms,
    // 4 x 4 63.1 ms (27.5 vs 29.1 ms on 8.8M rows) (profiles/r02_c_gang_shapes.txt).  A gang is used
    // only where it is at least ~7/8 full; what is left runs as smaller launches.
    struct Launch {
        int64_t q0;
        int qc;
        Plan plan;
    };
    const bool gang_dim = (h->dim == 768 && h->dtype != ARTMIPS_F32);
    const int nw_knob = diag_int("ARTMIPS_NW", 8);   // diagnostic build: 4 = no wide shapes, 88 / 44 = gangs of 8 x 2 / 4 x 4 whatever k
    const bool wide_ok = gang_dim && nw_knob != 4;
    const int64_t shard_tiles = (gen.n_rows + TILE_ROWS - 1) / TILE_ROWS;
    int max_gang = 1;
    if (wide_ok && shard_tiles >= 8ll * s.n_cu) {
        if (s.n_cu % 64 == 0) max_gang = 8;
        else if (s.n_cu % 32 == 0) max_gang = 4;
        else if (s.n_cu % 16 == 0) max_gang = 2;
    }
    max_gang = std::min(max_gang, std::max(1, diag_int("ARTMIPS_GANG", 8)));
    const int nq_cap = std::max(1, diag_int("ARTMIPS_MAX_NQ", MAX_NQ));
    c.t_begin = std::chrono::steady_clock::now();
    Launch launches[64];
    int n_launches = 0;
    Plan cache[8];          // plans by shape: [0..3] = 4 x (1..4), [4] = 8 x 2, [5..7] = gangs of 2, 4, 8
    bool have[8] = {};
    size_t sz_qpad = 0, sz_thr = 0, sz_gmax = 0, sz_counts = 0, sz_lists = 0, sz_keys = 0;
    std::vector<Launch> overflow_launches;   // batches of more than 64 launches (> ~16k queries)
    for (int64_t q0 = 0; q0 < c.n_q;) {
        const int64_t left = c.n_q - q0;
        int nw = 4, nq = 1, gang = 1, slot_i;
        if (wide_ok && left > 3 * QGROUP && nq_cap >= 2) {
            if (max_gang >= 8 && left > 7 * 256) gang = 8;
            else if (max_gang >= 4 && left > 3 * 256) gang = 4;
            else if (max_gang >= 2 && left > 256 + 128) gang = 2;
            // (4 x 4 up to k = 256: 1 % ahead of 8 x 2 at k = 200, level at 300, 1.4 % behind at 500, 6 % at 1,000:
            // profiles/r03_e_gang_sync.txt)
            if (gang > 1 && nw_knob != 88 && (c.k <= 256 || nw_knob == 44)) {
                nw = 4;
                nq = 4;
            } else {
                nw = 8;
                nq = 2;
            }
            slot_i = gang == 8 ? 7 : gang == 4 ? 6 : gang == 2 ? 5 : 4;
        } else {
            nq = (int)std::min<int64_t>((left + QGROUP - 1) / QGROUP, std::min(nq_cap, max_nq(h->dim, h->dtype)));
            slot_i = nq - 1;
        }
        if (!have[slot_i]) {
            cache[slot_i] = make_plan(s, gen.n_rows, (size_t)h->dim * elem_size(h->dtype), c.k, h->cap_boost, nq,
                                      nw, h->dim == 768 && h->dtype != ARTMIPS_F32, gang);
            have[slot_i] = true;
            const Plan& p = cache[slot_i];
            // (the scan kernels address a lane's list entries with a 32-bit byte offset inside its column group)
            if ((unsigned long long)nw * 64ull * (unsigned long long)p.cap * 8ull >= (1ull << 32))
                return fail(ARTMIPS_ERR_UNSUPPORTED, "internal: candidate lists of %d entries per lane", p.cap);
            const size_t QT = (size_t)nq * nw * 16 * gang, NT = (size_t)nw * 64;
            sz_qpad = std::max(sz_qpad, QT * h->dim * elem_size(h->dtype));
            sz_thr = std::max(sz_thr, QT * 4);
            if (p.pass0) sz_gmax = std::max(sz_gmax, (size_t)gang * p.grid0 * nq * NT * 4 * p.topt);
            sz_counts = std::max(sz_counts, (size_t)gang * p.grid * nq * NT * 4);
            sz_lists = std::max(sz_lists, (size_t)gang * p.grid * nq * NT * (size_t)p.cap * 8);
            // global key scratch: worst case every list of a query is full
            const size_t keys_per_q = (size_t)p.grid * 4 * (size_t)p.cap;
            if (keys_per_q > (size_t)SORT_MAX) sz_keys = std::max(sz_keys, keys_per_q * 8 * QT);
        }
        const int qc = (int)std::min<int64_t>(left, (int64_t)nq * nw * 16 * gang);
        const Launch l{q0, qc, cache[slot_i]};
        if (n_launches < 64) launches[n_launches++] = l;
        else overflow_launches.push_back(l);
        q0 += qc;
    }
    int rc = 0;
    Workspace ws;
    if (c.static_scratch) {
        // carve the workspace out of the static device array; nothing is allocated, and a shape that
        // does not fit is refused before anything is launched
        size_t off = 0;
        auto take = [&](size_t bytes) -> void* {
            void* at = c.static_scratch + off;
            off += (bytes + 255) & ~(size_t)255;
            return at;
        };
        ws.status = (int*)take(STATUS_BLOCK_BYTES);
        ws.thr = (float*)take(sz_thr);
        ws.qpad = take(sz_qpad);
        ws.gmax = (float*)take(sz_gmax);
        ws.counts = (int*)take(sz_counts);
        ws.lists = (int2*)take(sz_lists);
        if (off > (size_t)ARTMIPS_NOALLOC_SCRATCH_BYTES)
            return fail(ARTMIPS_ERR_UNSUPPORTED,
                        "this search needs %zu bytes of scratch (%lld rows, %lld queries, top-%d); the "
                        "allocation-free entry point has %lld (ARTMIPS_NOALLOC_SCRATCH_BYTES)",
                        off, (long long)gen.n_rows, (long long)c.n_q, c.k,
                        (long long)ARTMIPS_NOALLOC_SCRATCH_BYTES);
        // key scratch (needed only when one query keeps more candidates than the LDS routes hold):
        // whatever is left; select_kernel raises status bit 1 rather than use more than it is given
        ws.keys = (u64*)(c.static_scratch + off);
        ws.keys_bytes = std::min(sz_keys, (size_t)ARTMIPS_NOALLOC_SCRATCH_BYTES - off);
    } else {
        if ((rc = ensure(s.qpad, sz_qpad))) return rc;
        if ((rc = ensure(s.thr, sz_thr))) return rc;
        if (!s.status.p) {
            if ((rc = ensure(s.status, STATUS_BLOCK_BYTES))) return rc;
            // word 8 is select_kernel's done counter; zeroed on the stream the pipeline runs on (the
            // null stream orders nothing against a non-blocking stream)
            HIP_TRY(hipMemsetAsync(s.status.p, 0, STATUS_BLOCK_BYTES, c.st));
        }
        if (sz_gmax && (rc = ensure(s.gmax, sz_gmax))) return rc;
        if ((rc = ensure(s.counts, sz_counts))) return rc;
        if ((rc = ensure(s.lists, sz_lists))) return rc;
        if (sz_keys && (rc = ensure(s.keys, sz_keys))) return rc;
        ws.qpad = s.qpad.p;
        ws.thr = (float*)s.thr.p;
        ws.gmax = (float*)s.gmax.p;
        ws.counts = (int*)s.counts.p;
        ws.lists = (int2*)s.lists.p;
        ws.keys = (u64*)s.keys.p;
        ws.keys_bytes = s.keys.bytes;
        ws.status = (int*)s.status.p;
    }
    // HIP-event timing creates events on first use: never on the allocation-free path
    const bool timed = c.record_stats && h->stats_enabled && !c.static_scratch;
    if (timed) HIP_TRY(hipEventRecord(s.ev_t0, c.st));
    int* const status = c.status_dev ? c.status_dev : ws.status;
    c.status_used = status;
    c.n_tiles_q = 0;
    const int total_launches = n_launches + (int)overflow_launches.size();
    // the last select launch publishes the status words to the pinned host block: a sentinel there
    // makes a publication that never happened visible (shard_complete) instead of letting the previous
    // search's "no overflow" stand in for it
    if (c.host_status && !c.static_scratch) ((volatile int*)s.status_host)[0] = STATUS_UNPUBLISHED;
    for (int li = 0; li < total_launches; ++li, ++c.n_tiles_q) {
        const Launch& l = li < 64 ? launches[li] : overflow_launches[li - 64];
        const bool publish = c.host_status && !c.static_scratch && li == total_launches - 1;
        rc = enqueue_tile(h, s, ws, l.plan, (const char*)c.q_dev + (size_t)l.q0 * q_row_bytes, c.q_dtype, l.qc,
                          c.k, c.out_scores, c.score_dtype, c.out_ids, l.q0, c.st, c.n_tiles_q, timed, status,
                          publish);
        if (rc) return rc;
    }
    if (timed) HIP_TRY(hipEventRecord(s.ev_t1, c.st));
    c.t_enq = std::chrono::steady_clock::now();
    return 0;
}

// Wait for a shard's pipeline and read its status.  *overflow is set when a candidate list
// overflowed: the result may have lost a top-k member and the shard must be searched again with
// larger lists.
static int shard_complete(artmips* h, ShardCall& c, bool* overflow) {
    Shard& s = h->sh[c.slot];
    DeviceGuard g(s.dev);
    HIP_TRY(hipStreamSynchronize(c.st));
    const auto t_syn = std::chrono::steady_clock::now();
    *overflow = false;
    const int status = ((volatile int*)s.status_host)[0];
    if (status == STATUS_UNPUBLISHED)
        return fail(ARTMIPS_ERR_STATE, "internal: the pipeline ended without publishing its status words");
    if (status & 2) return fail(ARTMIPS_ERR_STATE, "internal: key scratch too small");
    if (status & 1) { *overflow = true; return 0; }
    if (c.record_stats) {
        h->host_enqueue_us = std::chrono::duration<double, std::micro>(c.t_enq - c.t_begin).count();
        h->host_sync_us = std::chrono::duration<double, std::micro>(t_syn - c.t_enq).count();
        h->stats_slot = c.slot;
        h->stats_tiles = c.n_tiles_q;
        h->stats_pending = h->stats_enabled;  // event times are read when artmips_info asks for them
        h->stats_async = false;
        unsigned long long n = 0;
        memcpy(&n, s.status_host + 2, 8);
        h->last_candidates = (int64_t)n;
    }
    return 0;
}

// Full local search of one shard; returns with the stream synchronised and overflow retries (if
// any) done.  At cap == rows-per-lane overflow is impossible, so the retry loop ends.
static int search_shard(artmips* h, ShardCall c, int retries_so_far = 0) {
    int retries = retries_so_far;
    for (;;) {
        int rc = shard_enqueue(h, c);
        if (rc) return rc;
        bool overflow = false;
        if ((rc = shard_complete(h, c, &overflow))) return rc;
        if (!overflow) {
            if (c.record_stats) h->last_retries = retries;
            return 0;
        }
        ++retries;
        if (h->cap_boost >= (1 << 24)) return fail(ARTMIPS_ERR_STATE, "overflow retry did not converge");
        h->cap_boost *= 8;
    }
}

static int check_search_args(artmips* h, const void* q, int q_dtype, int64_t n_q, int k,
                             const void* out_scores, const void* out_ids) {
    if (!h || !q || !out_scores || !out_ids) return fail(ARTMIPS_ERR_ARG, "NULL argument");
    if (h->n_rows < 1) return fail(ARTMIPS_ERR_STATE, "search on an empty index (no artmips_finalize yet)");
    if (q_dtype < 0 || q_dtype > 2) return fail(ARTMIPS_ERR_ARG, "bad query dtype %d", q_dtype);
    if (n_q < 1) return fail(ARTMIPS_ERR_ARG, "n_q must be >= 1");
    if (k < 1) return fail(ARTMIPS_ERR_ARG, "top_k must be >= 1 (got %d)", k);
    return 0;
}

extern "C" int artmips_search_local(artmips_t* h, int slot, const void* q_dev, int q_dtype,
                                    int64_t n_q, int k, float* out_scores, int32_t* out_ids,
                                    void* stream) {
    int rc = check_search_args(h, q_dev, q_dtype, n_q, k, out_scores, out_ids);
    if (rc) return rc;
    if (slot < 0 || slot >= h->n_dev) return fail(ARTMIPS_ERR_ARG, "slot %d out of range", slot);
    const int64_t n_local = h->sh[slot].live.n_rows;
    if (k > ARTMIPS_MAX_K && n_local > SORT_MAX)
        return fail(ARTMIPS_ERR_UNSUPPORTED, "top_k %d > %d on a shard of %lld rows", k, ARTMIPS_MAX_K,
                    (long long)n_local);
    if (k > SORT_MAX) return fail(ARTMIPS_ERR_UNSUPPORTED, "top_k %d > %d", k, SORT_MAX);
    ShardCall c{slot, q_dev, q_dtype, n_q, k, out_scores, ARTMIPS_F32, out_ids, (hipStream_t)stream, true};
    return search_shard(h, c);
}

// Scratch of the handle-less artmips_merge: one (status, keys) pair per device, kept for the life
// of the process.  A buffer only ever grows, and growing waits for the device first, so a merge
// still in flight on some stream never loses its scratch.
struct MergeScratch {
    DevBuf keys, status;
    int32_t* flag_host = nullptr;  // pinned words the merge kernel writes over the bus (merge_packed_sync)
    int32_t* flag_dev = nullptr;   // their address as the device sees them
    unsigned next_flag = 0;        // calls take the words round-robin, so two calls in flight differ
    std::mutex m;                  // one lock per device: a one-process caller's devices do not serialise
};
constexpr int MERGE_FLAG_WORDS = 64;
static MergeScratch g_merge_scratch[ARTMIPS_MAX_DEVICES * 4];

static int run_merge(int device, const float* scores, const int32_t* ids, int n_lists, int64_t n_q,
                     int k, void* out_scores, int score_dtype, int32_t* out_ids, hipStream_t st,
                     DevBuf* keys, DevBuf* status_buf, long long list_stride = 0,
                     const int* status_in = nullptr, long long status_stride = 0, int* flag_out = nullptr) {
    DeviceGuard g(device);
    MergeArgs m;
    memset(&m, 0, sizeof(m));
    m.scores = scores;
    m.ids = ids;
    m.n_lists = n_lists;
    m.n_q = n_q;
    m.k = k;
    m.out_scores = out_scores;
    m.score_dtype = score_dtype;
    m.out_ids = out_ids;
    m.list_stride = list_stride ? list_stride : (long long)n_q * k;
    m.status_in = status_in;
    m.status_stride = status_stride;
    m.flag_out = flag_out;
    const long long total = (long long)n_lists * k;
    int rc = 0;
    if (status_buf->bytes < 64) {
        if ((rc = ensure(*status_buf, 64))) return rc;
        HIP_TRY(hipMemset(status_buf->p, 0, 64));
    }
    if (total > SORT_MAX) {
        const size_t need = (size_t)total * 8 * (size_t)n_q;
        if (keys->bytes < need) {
            HIP_TRY(hipDeviceSynchronize());
            if ((rc = ensure(*keys, need))) return rc;
        }
        m.keys_ws = (u64*)keys->p;
        m.keys_stride = total;
    }
    // the kernel can only raise status bit 1 if keys_stride were too small, which the sizing above
    // rules out; the merge is therefore left asynchronous on `st`
    m.status = (int*)status_buf->p;
    hipLaunchKernelGGL(merge_kernel, dim3((unsigned)n_q), dim3(SEL_THREADS), 0, st, m);
    HIP_TRY(hipGetLastError());
    DEBUG_STEP(st, "merge n_q=%lld lists=%d k=%d", (long long)n_q, n_lists, k);
    return 0;
}

extern "C" int artmips_merge(int device, const float* scores, const int32_t* ids, int n_lists,
                             int64_t n_q, int k, void* out_scores, int score_dtype,
                             int32_t* out_ids, void* stream) {
    if (!scores || !ids || !out_scores || !out_ids) return fail(ARTMIPS_ERR_ARG, "NULL argument");
    if (n_lists < 1 || n_q < 1 || k < 1) return fail(ARTMIPS_ERR_ARG, "n_lists, n_q and k must be >= 1");
    if (k > SORT_MAX) return fail(ARTMIPS_ERR_UNSUPPORTED, "top_k %d > %d", k, SORT_MAX);
    if (score_dtype < 0 || score_dtype > 2) return fail(ARTMIPS_ERR_ARG, "bad score dtype");
    if (device < 0 || device >= ARTMIPS_MAX_DEVICES * 4) return fail(ARTMIPS_ERR_ARG, "device %d", device);
    MergeScratch& ms = g_merge_scratch[device];
    std::lock_guard<std::mutex> lock(ms.m);
    return run_merge(device, scores, ids, n_lists, n_q, k, out_scores, score_dtype, out_ids,
                     (hipStream_t)stream, &ms.keys, &ms.status);
}

static_assert(ARTMIPS_PACK_TAIL == 4, "the kernels clear and raise exactly 4 status words");

extern "C" int64_t artmips_packed_words(int64_t n_q, int k) {
    return 2 * n_q * (int64_t)k + ARTMIPS_PACK_TAIL;
}

extern "C" int artmips_search_local_async(artmips_t* h, int slot, const void* q_dev, int q_dtype,
                                          int64_t n_q, int k, int32_t* out_packed, void* stream) {
    int rc = check_search_args(h, q_dev, q_dtype, n_q, k, out_packed, out_packed);
    if (rc) return rc;
    if (slot < 0 || slot >= h->n_dev) return fail(ARTMIPS_ERR_ARG, "slot %d out of range", slot);
    const int64_t n_local = h->sh[slot].live.n_rows;
    if (k > ARTMIPS_MAX_K && n_local > SORT_MAX)
        return fail(ARTMIPS_ERR_UNSUPPORTED, "top_k %d > %d on a shard of %lld rows", k, ARTMIPS_MAX_K,
                    (long long)n_local);
    if (k > SORT_MAX) return fail(ARTMIPS_ERR_UNSUPPORTED, "top_k %d > %d", k, SORT_MAX);
    ShardCall c{slot, q_dev, q_dtype, n_q, k, out_packed, ARTMIPS_F32, out_packed + n_q * (int64_t)k,
                (hipStream_t)stream, true};
    c.host_status = h->stats_enabled;  // only the candidate count of artmips_info needs it here
    // The shard's status words travel with the candidates, so that whoever merges them sees the
    // overflow bit: the kernels raise them in the tail of out_packed itself (an 8-byte aligned tail;
    // otherwise they are copied there behind the pipeline).
    int32_t* const tail = out_packed + 2 * n_q * (int64_t)k;
    const bool in_place = (((uintptr_t)tail) & 7) == 0;
    if (in_place) c.status_dev = (int*)tail;
    if ((rc = shard_enqueue(h, c))) return rc;
    Shard& s = h->sh[slot];
    DeviceGuard g(s.dev);
    if (!in_place)
        HIP_TRY(hipMemcpyAsync(tail, c.status_used, ARTMIPS_PACK_TAIL * 4, hipMemcpyDeviceToDevice, c.st));
    h->host_enqueue_us = std::chrono::duration<double, std::micro>(c.t_enq - c.t_begin).count();
    h->host_sync_us = 0.0;
    h->stats_slot = slot;
    h->stats_tiles = c.n_tiles_q;
    h->stats_pending = h->stats_enabled;
    h->stats_async = h->stats_enabled;
    h->stats_stream = c.st;
    h->last_retries = 0;
    return 0;
}

extern "C" int artmips_search_local_noalloc(artmips_t* h, int slot, const void* q_dev, int q_dtype,
                                            int64_t n_q, int k, int32_t* out_packed, void* stream) {
    int rc = check_search_args(h, q_dev, q_dtype, n_q, k, out_packed, out_packed);
    if (rc) return rc;
    if (slot < 0 || slot >= h->n_dev) return fail(ARTMIPS_ERR_ARG, "slot %d out of range", slot);
    const int64_t n_local = h->sh[slot].live.n_rows;
    if (k > ARTMIPS_MAX_K && n_local > SORT_MAX)
        return fail(ARTMIPS_ERR_UNSUPPORTED, "top_k %d > %d on a shard of %lld rows", k, ARTMIPS_MAX_K,
                    (long long)n_local);
    if (k > SORT_MAX) return fail(ARTMIPS_ERR_UNSUPPORTED, "top_k %d > %d", k, SORT_MAX);
    int32_t* const tail = out_packed + 2 * n_q * (int64_t)k;
    if ((((uintptr_t)tail) & 7) != 0)
        return fail(ARTMIPS_ERR_ARG, "out_packed must be 8-byte aligned (its status tail is raised in place)");
    Shard& s = h->sh[slot];
    unsigned char* const base = (unsigned char*)g_scratch_ptr[s.dev];   // looked up in artmips_create
    if (!base) return fail(ARTMIPS_ERR_STATE, "static scratch of device %d was never initialised", s.dev);
    ShardCall c{slot, q_dev, q_dtype, n_q, k, out_packed, ARTMIPS_F32, out_packed + n_q * (int64_t)k,
                (hipStream_t)stream, false};
    c.host_status = false;
    c.status_dev = (int*)tail;
    c.static_scratch = base;
    // shard_enqueue refuses a shape whose workspace exceeds the array before it launches anything;
    // the kernels overwrite every word of the workspace they later read (thresholds, maxima, list
    // counters, status tail), so the previous call's contents need no clearing
    return shard_enqueue(h, c);
}

extern "C" int artmips_set_stats(artmips_t* h, int enabled) {
    if (!h) return fail(ARTMIPS_ERR_ARG, "NULL handle");
    h->stats_enabled = enabled != 0;
    return 0;
}

extern "C" int artmips_grow_lists(artmips_t* h) {
    if (!h) return fail(ARTMIPS_ERR_ARG, "NULL handle");
    if (h->cap_boost >= (1 << 24)) return fail(ARTMIPS_ERR_STATE, "overflow retry did not converge");
    h->cap_boost *= 8;
    return 0;
}

extern "C" int artmips_merge_packed(int device, const int32_t* packed, int n_lists, int64_t n_q, int k,
                                    void* out_scores, int score_dtype, int32_t* out_ids,
                                    int32_t* out_flag, void* stream) {
    if (!packed || !out_scores || !out_ids) return fail(ARTMIPS_ERR_ARG, "NULL argument");
    if (n_lists < 1 || n_q < 1 || k < 1) return fail(ARTMIPS_ERR_ARG, "n_lists, n_q and k must be >= 1");
    if (k > SORT_MAX) return fail(ARTMIPS_ERR_UNSUPPORTED, "top_k %d > %d", k, SORT_MAX);
    if (score_dtype < 0 || score_dtype > 2) return fail(ARTMIPS_ERR_ARG, "bad score dtype");
    if (device < 0 || device >= ARTMIPS_MAX_DEVICES * 4) return fail(ARTMIPS_ERR_ARG, "device %d", device);
    const long long plane = (long long)n_q * k;
    const long long stride = 2 * plane + ARTMIPS_PACK_TAIL;
    MergeScratch& ms = g_merge_scratch[device];
    std::lock_guard<std::mutex> lock(ms.m);
    return run_merge(device, (const float*)packed, packed + plane, n_lists, n_q, k, out_scores, score_dtype,
                     out_ids, (hipStream_t)stream, &ms.keys, &ms.status, stride,
                     out_flag ? packed + 2 * plane : nullptr, stride, out_flag);
}

extern "C" int artmips_merge_packed_sync(int device, const int32_t* packed, int n_lists, int64_t n_q, int k,
                                         void* out_scores, int score_dtype, int32_t* out_ids,
                                         int32_t* flag_host_out, void* stream) {
    if (!packed || !out_scores || !out_ids || !flag_host_out) return fail(ARTMIPS_ERR_ARG, "NULL argument");
    if (n_lists < 1 || n_q < 1 || k < 1) return fail(ARTMIPS_ERR_ARG, "n_lists, n_q and k must be >= 1");
    if (k > SORT_MAX) return fail(ARTMIPS_ERR_UNSUPPORTED, "top_k %d > %d", k, SORT_MAX);
    if (score_dtype < 0 || score_dtype > 2) return fail(ARTMIPS_ERR_ARG, "bad score dtype");
    if (device < 0 || device >= ARTMIPS_MAX_DEVICES * 4) return fail(ARTMIPS_ERR_ARG, "device %d", device);
    const long long plane = (long long)n_q * k;
    const long long stride = 2 * plane + ARTMIPS_PACK_TAIL;
    MergeScratch& ms = g_merge_scratch[device];
    DeviceGuard g(device);
    // The per-device lock covers the enqueue.  It is kept across the wait only when the merge uses
    // the device's global key scratch (more than SORT_MAX candidates per query), which a second
    // merge on another stream of the same device would otherwise overwrite; the flag word is this
    // call's own (taken round-robin from a ring of pinned words).
    std::unique_lock<std::mutex> lock(ms.m);
    if (!ms.flag_host) {
        HIP_TRY(hipHostMalloc((void**)&ms.flag_host, MERGE_FLAG_WORDS * 4, hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer((void**)&ms.flag_dev, ms.flag_host, 0));
    }
    const unsigned slot = ms.next_flag++ % MERGE_FLAG_WORDS;
    int rc = run_merge(device, (const float*)packed, packed + plane, n_lists, n_q, k, out_scores, score_dtype,
                       out_ids, (hipStream_t)stream, &ms.keys, &ms.status, stride, packed + 2 * plane, stride,
                       ms.flag_dev + slot);
    if (rc) return rc;
    if ((long long)n_lists * k <= SORT_MAX) lock.unlock();
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));  // kernel end makes the word visible to the host
    *flag_host_out = ((volatile int32_t*)ms.flag_host)[slot];
    return 0;
}

// common argument checks of artmips_search / artmips_search_async; fills `used` with the shards that
// hold rows
static int check_global_search(artmips* h, const void* q_dev, int q_dtype, int64_t n_q, int k,
                               const void* out_scores, int score_dtype, const void* out_ids,
                               std::vector<int>* used) {
    int rc = check_search_args(h, q_dev, q_dtype, n_q, k, out_scores, out_ids);
    if (rc) return rc;
    if (score_dtype < 0 || score_dtype > 2) return fail(ARTMIPS_ERR_ARG, "bad score dtype");
    if ((int64_t)k > h->n_rows)
        return fail(ARTMIPS_ERR_ARG, "top_k %d exceeds the %lld rows of the index", k, (long long)h->n_rows);
    if (k > SORT_MAX) return fail(ARTMIPS_ERR_UNSUPPORTED, "top_k %d > %d", k, SORT_MAX);
    for (int i = 0; i < h->n_dev; ++i) {
        if (h->sh[i].live.n_rows > 0) used->push_back(i);
        if (k > ARTMIPS_MAX_K && h->sh[i].live.n_rows > SORT_MAX)
            return fail(ARTMIPS_ERR_UNSUPPORTED, "top_k %d > %d on a shard of %lld rows", k,
                        ARTMIPS_MAX_K, (long long)h->sh[i].live.n_rows);
    }
    return 0;
}

// Several shards driven by this one process (the reference's model, art_index.py:273-291): every
// shard searches on its own stream into a packed candidate list (scores | ids | status tail, the
// layout of the SPMD exchange), the lists are brought to slot 0's device by peer copies (Q*k*8 bytes
// per shard instead of the reference's [Q, N/G] score slabs) and merged there on the caller's stream.
// Everything is enqueued, nothing waits for the host: the peer copy of every shard is queued behind
// its pipeline before any shard is waited for, and the merge writes the OR of the shards' status
// words to `flag_dev` (a device-visible int32).  A shard on slot 0's device writes its list
// straight into the gather buffer and reads the queries where they lie.
static int multi_enqueue(artmips* h, const std::vector<int>& used, const void* q_dev, int q_dtype,
                         int64_t n_q, int k, void* out_scores, int score_dtype, int32_t* out_ids,
                         int32_t* flag_dev, hipStream_t st, ShardCall* stats_call) {
    Shard& s0 = h->sh[0];
    int rc = 0;
    const size_t q_bytes = (size_t)n_q * h->dim * elem_size(q_dtype);
    const int64_t plane = n_q * (int64_t)k;
    const int64_t words = 2 * plane + ARTMIPS_PACK_TAIL;   // a multiple of 2: every list is 8-byte aligned
    {
        DeviceGuard g(s0.dev);
        if ((rc = ensure(h->g_packed, (size_t)words * 4 * used.size()))) return rc;
        HIP_TRY(hipEventRecord(s0.ev_ready, st));   // the queries exist from here on
    }
    int32_t* const gathered = (int32_t*)h->g_packed.p;
    // Each shard's enqueue (wait for the queries' producer, peer copy of the queries, the pipeline,
    // peer copy of its list) is one job: the first shard's runs here, the others on their worker
    // threads, because planning + enqueuing a pipeline costs the host ~19 us and one thread would
    // start the eighth GPU ~135 us after the first.
    std::vector<ShardCall> calls(used.size());
    auto enqueue_one = [h, q_dev, q_dtype, n_q, k, q_bytes, plane, words, gathered, &s0, &calls](
                           size_t u, int slot, bool stats) -> int {
        Shard& s = h->sh[slot];
        DeviceGuard g(s.dev);
        int r = 0;
        // diagnostic build: ARTMIPS_FORCE_REMOTE=1 takes the other-device branch (peer copy of the
        // queries, private list, peer copy of the list) for shards on slot 0's own device too, so that
        // a one-GPU box can execute it (tests/test_search3_gpu.py)
        const bool remote = (s.dev != s0.dev) || diag_int("ARTMIPS_FORCE_REMOTE", 0) != 0;
        const void* q_here = q_dev;
        int32_t* list = gathered + (size_t)u * words;
        HIP_TRY(hipStreamWaitEvent(s.stream, s0.ev_ready, 0));
        if (remote) {
            if ((r = ensure(s.qin, q_bytes))) return r;
            if ((r = ensure(s.loc_packed, (size_t)words * 4))) return r;
            HIP_TRY(hipMemcpyPeerAsync(s.qin.p, s.dev, q_dev, s0.dev, q_bytes, s.stream));
            q_here = s.qin.p;
            list = (int32_t*)s.loc_packed.p;
        }
        ShardCall& c = calls[u];
        c = ShardCall{slot, q_here, q_dtype, n_q, k, list, ARTMIPS_F32, list + plane, s.stream, stats};
        c.status_dev = (int*)(list + 2 * plane);   // raised in the list's own tail: it travels with it
        c.host_status = stats && h->stats_enabled;  // candidate count for artmips_info only
        if ((r = shard_enqueue(h, c))) return r;
        if (remote)
            HIP_TRY(hipMemcpyPeerAsync(gathered + (size_t)u * words, s0.dev, list, s.dev, (size_t)words * 4,
                                       s.stream));
        HIP_TRY(hipEventRecord(s.ev_done, s.stream));
        return 0;
    };
    const bool threaded = used.size() > 1 && diag_int("ARTMIPS_ENQUEUE_THREADS", 1) != 0;
    if (threaded) {
        if (h->workers.size() != (size_t)h->n_dev) h->workers.resize(h->n_dev);
        for (size_t u = 1; u < used.size(); ++u) {
            std::unique_ptr<ShardWorker>& w = h->workers[used[u]];
            if (!w) {
                w.reset(new ShardWorker());
                ShardWorker* raw = w.get();
                raw->th = std::thread([raw] { raw->loop(); });
            }
            const int slot = used[u];
            w->post([enqueue_one, u, slot] { return enqueue_one(u, slot, false); });
        }
        rc = enqueue_one(0, used[0], true);
        for (size_t u = 1; u < used.size(); ++u) {  // wait for every worker, even after a failure
            const int r = h->workers[used[u]]->wait();
            if (r && !rc) rc = r;
        }
    } else {
        for (size_t u = 0; u < used.size() && !rc; ++u) rc = enqueue_one(u, used[u], u == 0);
    }
    if (rc) {
        // do not return with pipelines still running on buffers the caller may free
        for (int i : used) { DeviceGuard g(h->sh[i].dev); (void)hipStreamSynchronize(h->sh[i].stream); }
        return rc;
    }
    if (stats_call) *stats_call = calls[0];
    DeviceGuard g(s0.dev);
    for (int i : used) HIP_TRY(hipStreamWaitEvent(st, h->sh[i].ev_done, 0));
    return run_merge(s0.dev, (const float*)gathered, gathered + plane, (int)used.size(), n_q, k, out_scores,
                     score_dtype, out_ids, st, &h->m_keys, &h->m_status, words, gathered + 2 * plane, words,
                     flag_dev);
}

extern "C" int artmips_search(artmips_t* h, const void* q_dev, int q_dtype, int64_t n_q, int k,
                              void* out_scores, int score_dtype, int32_t* out_ids, void* stream) {
    std::vector<int> used;
    int rc = check_global_search(h, q_dev, q_dtype, n_q, k, out_scores, score_dtype, out_ids, &used);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (used.size() == 1 && used[0] == 0) {
        ShardCall c{0, q_dev, q_dtype, n_q, k, out_scores, score_dtype, out_ids, st, true};
        return search_shard(h, c);
    }
    // One host wait per attempt: the merge leaves the OR of the shards' overflow bits in a pinned
    // host word.  A set bit (rare: the lists are sized 8x the expectation) repeats the call with 8x
    // larger lists on every shard; at cap == rows-per-lane overflow is impossible, so the loop ends.
    Shard& s0 = h->sh[0];
    int retries = 0;
    for (;;) {
        const auto t_entry = std::chrono::steady_clock::now();
        ShardCall c0;
        if ((rc = multi_enqueue(h, used, q_dev, q_dtype, n_q, k, out_scores, score_dtype, out_ids, h->flag_dev,
                                st, &c0)))
            return rc;
        const auto t_enq = std::chrono::steady_clock::now();
        {
            DeviceGuard g(s0.dev);
            HIP_TRY(hipStreamSynchronize(st));  // the search entry points return with the stream drained
        }
        const auto t_syn = std::chrono::steady_clock::now();
        const int flag = *(volatile int32_t*)h->flag_host;
        if (flag & 2) return fail(ARTMIPS_ERR_STATE, "internal: key scratch too small");
        if (!(flag & 1)) {
            h->host_enqueue_us = std::chrono::duration<double, std::micro>(t_enq - t_entry).count();
            h->host_sync_us = std::chrono::duration<double, std::micro>(t_syn - t_enq).count();
            h->stats_slot = c0.slot;
            h->stats_tiles = c0.n_tiles_q;
            h->stats_pending = h->stats_enabled;
            h->stats_async = false;
            h->last_retries = retries;
            if (h->stats_enabled) {
                // the first shard's candidate count: its pipeline and the copy behind it have ended
                // (the merge on `st` waited for them)
                unsigned long long n = 0;
                memcpy(&n, h->sh[c0.slot].status_host + 2, 8);
                h->last_candidates = (int64_t)n;
            }
            return 0;
        }
        ++retries;
        if (h->cap_boost >= (1 << 24)) return fail(ARTMIPS_ERR_STATE, "overflow retry did not converge");
        h->cap_boost *= 8;
    }
}

extern "C" int artmips_search_async(artmips_t* h, const void* q_dev, int q_dtype, int64_t n_q, int k,
                                    void* out_scores, int score_dtype, int32_t* out_ids,
                                    int32_t* out_status, void* stream) {
    std::vector<int> used;
    int rc = check_global_search(h, q_dev, q_dtype, n_q, k, out_scores, score_dtype, out_ids, &used);
    if (rc) return rc;
    if (!out_status || (((uintptr_t)out_status) & 7) != 0)
        return fail(ARTMIPS_ERR_ARG, "out_status must be 4 int32 words on the device, 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    h->stats_pending = false;
    h->last_retries = 0;
    if (used.size() == 1 && used[0] == 0) {
        ShardCall c{0, q_dev, q_dtype, n_q, k, out_scores, score_dtype, out_ids, st, false};
        c.host_status = false;
        c.status_dev = (int*)out_status;   // the kernels clear the 4 words and raise bits in word 0
        return shard_enqueue(h, c);
    }
    return multi_enqueue(h, used, q_dev, q_dtype, n_q, k, out_scores, score_dtype, out_ids, out_status, st,
                         nullptr);
}

template <typename T>
static void launch_gather(const Generation& g, int dim, const long long* pos, long long n, float* out,
                          hipStream_t st) {
    const int grid = (int)std::min<long long>(n, 4096);
    hipLaunchKernelGGL(gather_rows_kernel<T>, dim3(grid), dim3(256), 0, st, (const T*)g.rows, dim, pos, n,
                       (long long)g.n_rows, out);
}

extern "C" int artmips_fetch_rows(artmips_t* h, const int64_t* positions, int64_t n, float* out_host) {
    if (!h || !positions || !out_host) return fail(ARTMIPS_ERR_ARG, "NULL argument");
    if (h->n_rows < 1) return fail(ARTMIPS_ERR_STATE, "fetch from an empty index (no artmips_finalize yet)");
    if (n < 0) return fail(ARTMIPS_ERR_ARG, "n must be >= 0");
    for (int64_t i = 0; i < n; ++i)
        if (positions[i] < 0 || positions[i] >= h->n_rows)
            return fail(ARTMIPS_ERR_ARG, "row position %lld outside the %lld rows of the index",
                        (long long)positions[i], (long long)h->n_rows);
    // rarely used, so kept simple: per shard, the positions that fall inside it are sent over, the
    // rows come back as fp32 and are put where they belong in out_host
    std::vector<long long> local, where;
    std::vector<float> tmp;
    for (int sidx = 0; sidx < h->n_dev; ++sidx) {
        Shard& s = h->sh[sidx];
        if (s.live.n_rows == 0) continue;
        local.clear();
        where.clear();
        for (int64_t i = 0; i < n; ++i) {
            const int64_t p = positions[i] - s.live.row0;
            if (p >= 0 && p < s.live.n_rows) { local.push_back(p); where.push_back(i); }
        }
        if (local.empty()) continue;
        DeviceGuard g(s.dev);
        const size_t m = local.size();
        // the shard's grow-on-demand scratch (kept between calls): positions in, fp32 rows out
        int rc = 0;
        HIP_TRY(hipStreamSynchronize(s.stream));   // nothing of an earlier call still uses the two buffers
        if ((rc = ensure(s.loc_ids, m * 8))) return rc;
        if ((rc = ensure(s.loc_scores, m * (size_t)h->dim * 4))) return rc;
        void* const d_pos = s.loc_ids.p;
        void* const d_out = s.loc_scores.p;
        hipError_t e = hipMemcpyAsync(d_pos, local.data(), m * 8, hipMemcpyHostToDevice, s.stream);
        if (e == hipSuccess) {
            if (h->dtype == ARTMIPS_F16) launch_gather<_Float16>(s.live, h->dim, (const long long*)d_pos, (long long)m, (float*)d_out, s.stream);
            else if (h->dtype == ARTMIPS_BF16) launch_gather<__bf16>(s.live, h->dim, (const long long*)d_pos, (long long)m, (float*)d_out, s.stream);
            else launch_gather<float>(s.live, h->dim, (const long long*)d_pos, (long long)m, (float*)d_out, s.stream);
            e = hipGetLastError();
        }
        tmp.resize(m * (size_t)h->dim);
        if (e == hipSuccess) e = hipMemcpyAsync(tmp.data(), d_out, m * (size_t)h->dim * 4, hipMemcpyDeviceToHost, s.stream);
        if (e == hipSuccess) e = hipStreamSynchronize(s.stream);
        if (e != hipSuccess) return fail(ARTMIPS_ERR_HIP, "row fetch on shard %d failed: %s", sidx, hipGetErrorString(e));
        for (size_t j = 0; j < m; ++j)
            memcpy(out_host + (size_t)where[j] * h->dim, tmp.data() + j * (size_t)h->dim, (size_t)h->dim * 4);
    }
    return 0;
}

extern "C" int artmips_generation(artmips_t* h, uint64_t* out) {
    if (!h || !out) return fail(ARTMIPS_ERR_ARG, "NULL argument");
    std::lock_guard<std::mutex> lock(h->gen_mutex);
    *out = h->generation;
    return 0;
}

extern "C" int artmips_download(artmips_t* h, uint64_t generation, int64_t row_offset, int64_t n_rows,
                                void* host_rows, int32_t* host_ids) {
    if (!h || !host_rows || !host_ids) return fail(ARTMIPS_ERR_ARG, "NULL argument");
    if (n_rows < 0 || row_offset < 0) return fail(ARTMIPS_ERR_ARG, "negative range");
    const size_t row_bytes = (size_t)h->dim * elem_size(h->dtype);
    const int64_t rows_per_buf = std::max<int64_t>(1, (16ll << 20) / (int64_t)row_bytes);
    const size_t need = (size_t)rows_per_buf * (row_bytes + 4);
    // Two pinned buffers, two streams per shard: while chunk i travels device -> pinned buffer, chunk
    // i-1 is copied from the other pinned buffer to its place in the caller's memory.
    struct Pending {
        bool on = false;
        int slot = 0, buf = 0;
        int64_t at = 0, n = 0;
    } prev;
    auto drain = [&](const Pending& p) -> int {
        DeviceGuard g(h->sh[p.slot].dev);
        HIP_TRY(hipStreamSynchronize(h->dl_stream[p.slot][p.buf]));
        const char* stage = (const char*)h->dl_pin[p.buf];
        memcpy((char*)host_rows + (size_t)p.at * row_bytes, stage, (size_t)p.n * row_bytes);
        memcpy(host_ids + p.at, stage + (size_t)rows_per_buf * row_bytes, (size_t)p.n * 4);
        return 0;
    };
    int64_t done = 0;
    int which = 0;
    int rc = 0;
    while (done < n_rows) {
        Pending cur;
        {
            // one chunk is issued per lock hold, so that a refresh that finishes meanwhile waits
            // microseconds; the generation check makes a swapped index an error, not a mix.  A chunk
            // in flight when the lock is released is safe: artmips_finalize synchronises the device
            // before it frees the old generation.
            std::lock_guard<std::mutex> lock(h->gen_mutex);
            if (h->generation != generation) {
                rc = fail(ARTMIPS_ERR_STATE, "the index was replaced while it was being downloaded");
                break;
            }
            if (row_offset + n_rows > h->n_rows) {
                rc = fail(ARTMIPS_ERR_ARG, "rows [%lld, %lld) outside the %lld rows of the index",
                          (long long)row_offset, (long long)(row_offset + n_rows), (long long)h->n_rows);
                break;
            }
            const int64_t gpos = row_offset + done;
            const int slot = (int)(gpos / h->chunk);
            Shard& s = h->sh[slot];
            const int64_t in_shard = gpos - s.live.row0;
            const int64_t n = std::min<int64_t>(std::min<int64_t>(rows_per_buf, n_rows - done),
                                                s.live.n_rows - in_shard);
            DeviceGuard g(s.dev);
            if (h->dl_pin_bytes < need) {
                for (int i = 0; i < 2; ++i) {
                    if (h->dl_pin[i]) (void)hipHostFree(h->dl_pin[i]);
                    h->dl_pin[i] = nullptr;
                    HIP_TRY(hipHostMalloc(&h->dl_pin[i], need, hipHostMallocDefault));
                }
                h->dl_pin_bytes = need;
            }
            hipStream_t& st = h->dl_stream[slot][which];
            if (!st) HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            char* stage = (char*)h->dl_pin[which];
            HIP_TRY(hipMemcpyAsync(stage, (const char*)s.live.rows + (size_t)in_shard * row_bytes,
                                   (size_t)n * row_bytes, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(stage + (size_t)rows_per_buf * row_bytes, s.live.ids + in_shard,
                                   (size_t)n * 4, hipMemcpyDeviceToHost, st));
            cur.on = true;
            cur.slot = slot;
            cur.buf = which;
            cur.at = done;
            cur.n = n;
        }
        if (prev.on && (rc = drain(prev))) { prev = cur; break; }
        prev = cur;
        which ^= 1;
        done += cur.n;
    }
    if (prev.on) {
        const int r = drain(prev);   // also on the error paths: nothing may stay in flight into pinned memory
        if (r && !rc) rc = r;
    }
    return rc;
}

extern "C" int artmips_info(artmips_t* h, artmips_info_t* out) {
    if (!h || !out) return fail(ARTMIPS_ERR_ARG, "NULL argument");
    if (h->stats_pending) {
        Shard& s = h->sh[h->stats_slot];
        DeviceGuard g(s.dev);
        if (h->stats_async) {  // nobody has waited for that pipeline on the library's behalf yet
            HIP_TRY(hipStreamSynchronize(h->stats_stream));
            unsigned long long n = 0;
            memcpy(&n, s.status_host + 2, 8);
            h->last_candidates = (int64_t)n;
            h->stats_async = false;
        }
        float ms = 0, total = 0;
        for (int t = 0; t < h->stats_tiles; ++t) {
            float x = 0;
            HIP_TRY(hipEventElapsedTime(&x, s.ev_scan[2 * t], s.ev_scan[2 * t + 1]));
            ms += x;
        }
        HIP_TRY(hipEventElapsedTime(&total, s.ev_t0, s.ev_t1));
        h->last_scan_ms = ms;
        h->last_total_ms = total;
        h->last_scan_launches = h->stats_tiles;
        h->stats_pending = false;
    }
    memset(out, 0, sizeof(*out));
    out->n_dev = h->n_dev;
    out->dim = h->dim;
    out->storage_dtype = h->dtype;
    out->n_cu = h->sh[0].n_cu;
    out->n_rows = h->n_rows;
    size_t ws = h->g_packed.bytes + h->m_keys.bytes + h->m_status.bytes;
    for (int i = 0; i < h->n_dev; ++i) {
        const Shard& s = h->sh[i];
        out->rows_per_shard[i] = s.live.n_rows;
        out->index_bytes += (int64_t)s.live.n_rows * h->dim * (int64_t)elem_size(h->dtype);
        ws += s.qpad.bytes + s.thr.bytes + s.gmax.bytes + s.counts.bytes + s.lists.bytes +
              s.keys.bytes + s.status.bytes + s.qin.bytes + s.loc_scores.bytes + s.loc_ids.bytes +
              s.loc_packed.bytes;
    }
    out->workspace_bytes = (int64_t)ws;
    out->last_scan_ms = h->last_scan_ms;
    out->last_total_ms = h->last_total_ms;
    out->last_scan_launches = h->last_scan_launches;
    out->last_retries = h->last_retries;
    out->last_candidates = h->last_candidates;
    out->last_host_enqueue_us = (float)h->host_enqueue_us;
    out->last_host_sync_us = (float)h->host_sync_us;
    return 0;
}
