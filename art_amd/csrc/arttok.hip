// libarttok.so — evidence token store in HBM + batched id -> tokens gather (include/arttok.h).
// gfx950 only.  One wave per fetched item; HBM-bound byte work, no MFMA.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <new>

#include "../../include/arttok.h"

static thread_local char g_tok_err[512] = "";

static int tok_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_tok_err, sizeof(g_tok_err), fmt, ap);
    va_end(ap);
    return code;
}

#define TOK_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess)                                                                 \
            return tok_fail(ARTTOK_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                            __FILE__, __LINE__);                                               \
    } while (0)

struct TokDeviceGuard {
    int prev = -1;
    explicit TokDeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~TokDeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

struct arttok {
    int device = 0;
    int elem_bytes = 0;
    int64_t n_items = 0, n_tokens = 0;
    void* tokens = nullptr;     // [n_tokens] elements
    int64_t* offsets = nullptr; // [n_items] ELEMENT offset of every item
    int32_t* sizes = nullptr;   // [n_items]
};

// One wave (64 lanes) per output row; rows are grid-strided.  Lanes read consecutive tokens of the
// item (coalesced) and write consecutive int32 of the padded row.
template <typename E>
__device__ __forceinline__ void gather_tokens_body(const E* __restrict__ tokens,
                                                   const int64_t* __restrict__ offsets,
                                                   const int32_t* __restrict__ sizes,
                                                   int64_t n_items, const int32_t* __restrict__ ids,
                                                   int64_t n, int32_t id_base, int max_len,
                                                   int32_t pad, int32_t* __restrict__ out_tokens,
                                                   int32_t* __restrict__ out_lens) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t i = wave; i < n; i += n_waves) {
        const int64_t item = (int64_t)ids[i] - id_base;
        int32_t* row = out_tokens + i * max_len;
        int len = -1;
        const E* src = nullptr;
        if (item >= 0 && item < n_items) {
            len = sizes[item] < max_len ? sizes[item] : max_len;
            src = tokens + offsets[item];
        }
        for (int t = lane; t < max_len; t += 64) row[t] = (t < len) ? (int32_t)src[t] : pad;
        if (lane == 0) out_lens[i] = len;
    }
}
template <typename E>
__global__ __launch_bounds__(256) void gather_tokens_kernel(const E* __restrict__ tokens,
                                                            const int64_t* __restrict__ offsets,
                                                            const int32_t* __restrict__ sizes,
                                                            int64_t n_items, const int32_t* __restrict__ ids,
                                                            int64_t n, int32_t id_base, int max_len,
                                                            int32_t pad, int32_t* __restrict__ out_tokens,
                                                            int32_t* __restrict__ out_lens) {
    gather_tokens_body<E>(tokens, offsets, sizes, n_items, ids, n, id_base, max_len, pad, out_tokens, out_lens);
}

// The same gather with 16-byte stores: every lane takes 4 consecutive tokens of an item (its four loads
// are independent and the lanes' loads cover one contiguous run of the token file) and writes one int4
// of the padded row.  An item is a chain of dependent loads (id -> size and offset -> tokens), about
// 2.5 us with nothing else in flight, and one item per wave at a time made that chain the whole cost
// (361,000 items: 58 us even for 16-token rows).  So a wave works on several items at once: `lpi` lanes
// per item (the power of two that covers max_len / 4, at most 64), 64 / lpi items side by side, and four
// such groups per step, all their metadata requested before any token is copied.
// Needs 16-byte aligned output rows (max_len % 4 == 0, aligned base); the host falls back to
// gather_tokens_kernel otherwise.
template <typename E>
__device__ __forceinline__ void gather_tokens_vec_body(const E* __restrict__ tokens,
                                                       const int64_t* __restrict__ offsets,
                                                       const int32_t* __restrict__ sizes,
                                                       int64_t n_items, const int32_t* __restrict__ ids,
                                                       int64_t n, int32_t id_base, int max_len,
                                                       int32_t pad, int32_t* __restrict__ out_tokens,
                                                       int32_t* __restrict__ out_lens, int lpi) {
    const int lane = threadIdx.x & 63;
    const int sub = lane / lpi, sl = lane - sub * lpi;   // item slot of this lane in the wave, lane in the item
    const int ipw = 64 / lpi;                            // items per wave and group
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    constexpr int U = 4;
    for (int64_t base = wave * ipw * U; base < n; base += n_waves * ipw * U) {
        int64_t idx[U];
        int len[U];
        const E* src[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            idx[u] = base + u * ipw + sub;
            len[u] = -1;
            src[u] = nullptr;
            if (idx[u] < n) {
                const int64_t item = (int64_t)ids[idx[u]] - id_base;
                if (item >= 0 && item < n_items) {
                    len[u] = sizes[item] < max_len ? sizes[item] : max_len;
                    src[u] = tokens + offsets[item];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (idx[u] >= n) continue;
            int4* row = (int4*)(out_tokens + idx[u] * max_len);
            for (int t = sl * 4; t < max_len; t += lpi * 4) {
                int4 v;
                v.x = (t < len[u]) ? (int32_t)src[u][t] : pad;
                v.y = (t + 1 < len[u]) ? (int32_t)src[u][t + 1] : pad;
                v.z = (t + 2 < len[u]) ? (int32_t)src[u][t + 2] : pad;
                v.w = (t + 3 < len[u]) ? (int32_t)src[u][t + 3] : pad;
                typedef int v4i __attribute__((ext_vector_type(4)));
                const v4i vv = {v.x, v.y, v.z, v.w};
                __builtin_nontemporal_store(vv, (v4i*)&row[t >> 2]);   // written once, read by another kernel later
            }
            if (sl == 0) out_lens[idx[u]] = len[u];
        }
    }
}
template <typename E>
__global__ __launch_bounds__(256) void gather_tokens_vec_kernel(const E* __restrict__ tokens,
                                                                const int64_t* __restrict__ offsets,
                                                                const int32_t* __restrict__ sizes,
                                                                int64_t n_items, const int32_t* __restrict__ ids,
                                                                int64_t n, int32_t id_base, int max_len,
                                                                int32_t pad, int32_t* __restrict__ out_tokens,
                                                                int32_t* __restrict__ out_lens, int lpi) {
    gather_tokens_vec_body<E>(tokens, offsets, sizes, n_items, ids, n, id_base, max_len, pad, out_tokens,
                              out_lens, lpi);
}

// Several datasets, one launch (the four token datasets a training step fetches for the same [Q, k]
// winners: art_model.py:367-381).  blockIdx.y picks the dataset; its workgroups run the single-dataset
// body, so the four dependent-load chains (id -> size and offset -> tokens) of an item are in flight
// beside each other instead of one launch after the other, and three launches disappear (at the
// training shape, 2,048 ids, the launches WERE the cost).
struct TokSet {
    const void* tokens;
    const int64_t* offsets;
    const int32_t* sizes;
    int64_t n_items;
    int32_t* out_tokens;
    int32_t* out_lens;
    int elem_bytes;
    int max_len;
    int32_t pad;
    int lpi;   // 0: element-wise body (unaligned output), else lanes per item of the vector body
};
struct TokSets {
    TokSet s[ARTTOK_MAX_SETS];
};
__global__ __launch_bounds__(256) void gather_tokens_multi_kernel(TokSets sets, const int32_t* __restrict__ ids,
                                                                  int64_t n, int32_t id_base) {
    const TokSet& S = sets.s[blockIdx.y];
#define BODY(E)                                                                                              \
    do {                                                                                                     \
        if (S.lpi)                                                                                           \
            gather_tokens_vec_body<E>((const E*)S.tokens, S.offsets, S.sizes, S.n_items, ids, n, id_base,    \
                                      S.max_len, S.pad, S.out_tokens, S.out_lens, S.lpi);                    \
        else                                                                                                 \
            gather_tokens_body<E>((const E*)S.tokens, S.offsets, S.sizes, S.n_items, ids, n, id_base,        \
                                  S.max_len, S.pad, S.out_tokens, S.out_lens);                               \
    } while (0)
    switch (S.elem_bytes) {   // uniform per workgroup
        case 1: BODY(uint8_t); break;
        case 2: BODY(uint16_t); break;
        case 4: BODY(int32_t); break;
        default: BODY(int64_t); break;
    }
#undef BODY
}

extern "C" const char* arttok_last_error(void) { return g_tok_err; }

extern "C" int arttok_create(int device, const void* tokens_host, int64_t n_tokens, int elem_bytes,
                             const int64_t* pointers, const int32_t* sizes, int64_t n_items,
                             arttok_t** out) {
    if (!out || !pointers || !sizes || (!tokens_host && n_tokens > 0))
        return tok_fail(ARTTOK_ERR_ARG, "NULL argument");
    if (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4 && elem_bytes != 8)
        return tok_fail(ARTTOK_ERR_ARG, "elem_bytes must be 1, 2, 4 or 8 (got %d)", elem_bytes);
    if (n_items < 1 || n_tokens < 0) return tok_fail(ARTTOK_ERR_ARG, "empty dataset");
    int n_dev = 0;
    TOK_HIP(hipGetDeviceCount(&n_dev));
    if (device < 0 || device >= n_dev) return tok_fail(ARTTOK_ERR_ARG, "device %d of %d", device, n_dev);
    // byte pointers -> element offsets, with the bounds check the kernel relies on
    int64_t* off = new (std::nothrow) int64_t[n_items];
    if (!off) return tok_fail(ARTTOK_ERR_NOMEM, "host allocation failed");
    for (int64_t i = 0; i < n_items; ++i) {
        const int64_t p = pointers[i];
        if (p < 0 || p % elem_bytes != 0 || sizes[i] < 0 || p / elem_bytes + sizes[i] > n_tokens) {
            delete[] off;
            return tok_fail(ARTTOK_ERR_ARG, "item %lld (pointer %lld, size %d) lies outside the %lld tokens",
                            (long long)i, (long long)p, sizes[i], (long long)n_tokens);
        }
        off[i] = p / elem_bytes;
    }
    TokDeviceGuard g(device);
    arttok* h = new (std::nothrow) arttok;
    if (!h) {
        delete[] off;
        return tok_fail(ARTTOK_ERR_NOMEM, "host allocation failed");
    }
    h->device = device;
    h->elem_bytes = elem_bytes;
    h->n_items = n_items;
    h->n_tokens = n_tokens;
    const size_t tok_bytes = (size_t)(n_tokens > 0 ? n_tokens : 1) * elem_bytes;
    hipError_t e = hipMalloc(&h->tokens, tok_bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&h->offsets, (size_t)n_items * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&h->sizes, (size_t)n_items * 4);
    if (e == hipSuccess && n_tokens > 0)
        e = hipMemcpy(h->tokens, tokens_host, (size_t)n_tokens * elem_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(h->offsets, off, (size_t)n_items * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(h->sizes, sizes, (size_t)n_items * 4, hipMemcpyHostToDevice);
    delete[] off;
    if (e != hipSuccess) {
        arttok_destroy(h);
        return tok_fail(e == hipErrorOutOfMemory ? ARTTOK_ERR_NOMEM : ARTTOK_ERR_HIP, "upload failed: %s",
                        hipGetErrorString(e));
    }
    *out = h;
    return 0;
}

extern "C" void arttok_destroy(arttok_t* h) {
    if (!h) return;
    TokDeviceGuard g(h->device);
    if (h->tokens) (void)hipFree(h->tokens);
    if (h->offsets) (void)hipFree(h->offsets);
    if (h->sizes) (void)hipFree(h->sizes);
    delete h;
}

extern "C" int arttok_gather(arttok_t* h, const int32_t* ids_dev, int64_t n, int32_t id_base, int max_len,
                             int32_t pad, int32_t* out_tokens, int32_t* out_lens, void* stream) {
    if (!h || !ids_dev || !out_tokens || !out_lens) return tok_fail(ARTTOK_ERR_ARG, "NULL argument");
    if (n < 1) return tok_fail(ARTTOK_ERR_ARG, "n must be >= 1");
    if (max_len < 1 || max_len > (1 << 20)) return tok_fail(ARTTOK_ERR_ARG, "max_len %d out of range", max_len);
    TokDeviceGuard g(h->device);
    hipStream_t st = (hipStream_t)stream;
    const int waves_per_block = 4;
    const bool vec = (max_len % 4 == 0) && (((uintptr_t)out_tokens) & 15) == 0;
    // vector kernel: lanes per item = the power of two covering max_len / 4 (at most a whole wave)
    int lpi = 1;
    while (lpi < 64 && lpi * 4 < max_len) lpi <<= 1;
    const int64_t per_wave = vec ? 4 * (64 / lpi) : 1;   // items a wave takes per step
    const int64_t want = (n + waves_per_block * per_wave - 1) / (waves_per_block * per_wave);
    const unsigned grid = (unsigned)(want < 4096 ? want : 4096);
#define LAUNCH(E)                                                                                      \
    do {                                                                                               \
        if (vec)                                                                                       \
            hipLaunchKernelGGL(gather_tokens_vec_kernel<E>, dim3(grid), dim3(64 * waves_per_block), 0, st, \
                               (const E*)h->tokens, (const int64_t*)h->offsets, (const int32_t*)h->sizes,  \
                               h->n_items, ids_dev, n, id_base, max_len, pad, out_tokens, out_lens, lpi); \
        else                                                                                           \
            hipLaunchKernelGGL(gather_tokens_kernel<E>, dim3(grid), dim3(64 * waves_per_block), 0, st, \
                               (const E*)h->tokens, (const int64_t*)h->offsets, (const int32_t*)h->sizes,  \
                               h->n_items, ids_dev, n, id_base, max_len, pad, out_tokens, out_lens);   \
    } while (0)
    switch (h->elem_bytes) {
        case 1: LAUNCH(uint8_t); break;
        case 2: LAUNCH(uint16_t); break;
        case 4: LAUNCH(int32_t); break;
        default: LAUNCH(int64_t); break;
    }
#undef LAUNCH
    TOK_HIP(hipGetLastError());
    return 0;
}

// launch geometry of one dataset's gather: lanes per item of the vector body (0 = element-wise body)
// and the workgroups it wants
static void gather_shape(int max_len, const int32_t* out_tokens, int64_t n, int* lpi_out, int64_t* want_out) {
    const int waves_per_block = 4;
    const bool vec = (max_len % 4 == 0) && (((uintptr_t)out_tokens) & 15) == 0;
    int lpi = 1;
    while (lpi < 64 && lpi * 4 < max_len) lpi <<= 1;
    const int64_t per_wave = vec ? 4 * (64 / lpi) : 1;   // items a wave takes per step
    *lpi_out = vec ? lpi : 0;
    *want_out = (n + waves_per_block * per_wave - 1) / (waves_per_block * per_wave);
}

extern "C" int arttok_gather_multi(arttok_t* const* handles, int n_sets, const int32_t* ids_dev, int64_t n,
                                   int32_t id_base, const int32_t* max_lens, const int32_t* pads,
                                   int32_t* const* out_tokens, int32_t* const* out_lens, void* stream) {
    if (!handles || !ids_dev || !max_lens || !pads || !out_tokens || !out_lens)
        return tok_fail(ARTTOK_ERR_ARG, "NULL argument");
    if (n_sets < 1 || n_sets > ARTTOK_MAX_SETS)
        return tok_fail(ARTTOK_ERR_ARG, "n_sets must be in [1, %d]", ARTTOK_MAX_SETS);
    if (n < 1) return tok_fail(ARTTOK_ERR_ARG, "n must be >= 1");
    TokSets sets;
    memset(&sets, 0, sizeof(sets));
    int64_t want_max = 1;
    for (int i = 0; i < n_sets; ++i) {
        arttok* h = handles[i];
        if (!h || !out_tokens[i] || !out_lens[i]) return tok_fail(ARTTOK_ERR_ARG, "NULL argument (set %d)", i);
        if (h->device != handles[0]->device)
            return tok_fail(ARTTOK_ERR_ARG, "set %d lives on device %d, set 0 on device %d: one launch serves "
                            "one device", i, h->device, handles[0]->device);
        if (max_lens[i] < 1 || max_lens[i] > (1 << 20))
            return tok_fail(ARTTOK_ERR_ARG, "max_len %d out of range (set %d)", max_lens[i], i);
        TokSet& S = sets.s[i];
        S.tokens = h->tokens;
        S.offsets = h->offsets;
        S.sizes = h->sizes;
        S.n_items = h->n_items;
        S.out_tokens = out_tokens[i];
        S.out_lens = out_lens[i];
        S.elem_bytes = h->elem_bytes;
        S.max_len = max_lens[i];
        S.pad = pads[i];
        int64_t want = 1;
        gather_shape(max_lens[i], out_tokens[i], n, &S.lpi, &want);
        want_max = want > want_max ? want : want_max;
    }
    TokDeviceGuard g(handles[0]->device);
    const unsigned grid = (unsigned)(want_max < 4096 ? want_max : 4096);
    hipLaunchKernelGGL(gather_tokens_multi_kernel, dim3(grid, (unsigned)n_sets), dim3(256), 0, (hipStream_t)stream,
                       sets, ids_dev, n, id_base);
    TOK_HIP(hipGetLastError());
    return 0;
}

extern "C" int arttok_info(arttok_t* h, int64_t* n_items, int64_t* n_tokens, int64_t* device_bytes) {
    if (!h) return tok_fail(ARTTOK_ERR_ARG, "NULL handle");
    if (n_items) *n_items = h->n_items;
    if (n_tokens) *n_tokens = h->n_tokens;
    if (device_bytes) *device_bytes = h->n_tokens * h->elem_bytes + h->n_items * 12;
    return 0;
}
