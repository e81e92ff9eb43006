// artmips_kernels.h — gfx950 (MI355X, CDNA4) device code for exact MIPS top-k.
//
// Replaces the GPU side of DistributedBruteForceIndex.search_mips_index
// (/root/reference/megatron/data/art_index.py:278-294): instead of a full
// [Q, N] score matrix followed by torch.topk, the evidence rows are streamed
// ONCE from HBM through an LDS ring (LDS-DMA, 1 KiB per wave-instruction), scored
// on MFMA with fp32 accumulation, and filtered in registers against a per-query
// lower bound of the k-th best score.  Only the survivors (a few thousand per
// query out of 21M rows) are written out and sorted.
//
// Kernels
//   pack_queries_kernel   queries (f16/bf16/f32) -> padded [64, DIM] tile in the storage dtype
//   scan_kernel<GROUPMAX> pass 0: per-lane running max over a strided row sample
//   kth_kernel            pass 0: tau[q] = k-th largest group maximum  (a valid lower bound)
//   scan_kernel<FILTER>   the hot kernel: stream all rows, keep (score,row) with score >= tau[q]
//   select_kernel         per query: gather candidates, exact top-k by (score desc, id asc)
//   merge_kernel          per query: top-k of the union of per-shard top-k lists
//
// Layout facts the kernels rely on (checked on the host before every launch):
//   rows: [n_rows rounded up to whole tiles, DIM] row-major, 16-byte aligned, DIM % 128 == 0; the rows
//         past n_rows are zero
//   a "tile" is 16 consecutive rows; a "piece" is 8 rows x 128 B = one LDS-DMA wave-instruction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace artk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

constexpr int QGROUP = 64;       // queries per column group (4 waves x 16 MFMA columns)
constexpr int MAX_NQ = 4;        // column groups per scan pass: a pass scores up to 256 queries
constexpr int SCAN_THREADS = 256;
constexpr int TILE_ROWS = 16;    // rows per LDS stage (one 16x16x32 MFMA row block)
constexpr int SORT_MAX = 8192;   // 64-bit keys sorted in LDS by select/merge (64 KiB)
constexpr int SEL_THREADS = 1024;
constexpr int STAGING_BYTES = 16384;  // LDS behind the scan's ring: survivors staged per lane

enum { MODE_FILTER = 0, MODE_GROUPMAX = 1 };

// ---------------------------------------------------------------------------------------------
// element helpers
// ---------------------------------------------------------------------------------------------
// hazard pad in front of every inline-asm MFMA (see Elem<>::mfma_asm)
#ifdef ARTMIPS_NO_MFMA_PAD
#define ARTMIPS_MFMA_PAD ""
#else
#define ARTMIPS_MFMA_PAD "s_nop 1\n\t"
#endif

#ifndef ARTMIPS_GANG_SYNC
#define ARTMIPS_GANG_SYNC 1   // 0: no rendezvous between the members of a gang (A/B builds only)
#endif
#ifndef ARTMIPS_PHASED
#define ARTMIPS_PHASED 1   // 0: the 8-wave loop of rounds 1-2 (A/B builds only)
#endif
template <typename T> struct Elem;
template <> struct Elem<_Float16> {
    typedef half8 V;
    static __device__ __forceinline__ f32x4 mfma(V a, V b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ _Float16 from_float(float x) { return (_Float16)x; }
    // inline-asm forms used when the query operand must live in the accumulator register file.
    // The leading s_nop 1 is the VALU-write -> MFMA-operand-read hazard pad: hipcc may re-materialise
    // an operand (v_mov / v_accvgpr_read) right in front of the statement and pads nothing inside asm.
    template <bool B_AGPR, bool FIRST>
    static __device__ __forceinline__ void mfma_asm(f32x4& c, const V& a, const V& b) {
        if constexpr (FIRST) {
            if constexpr (B_AGPR) asm volatile(ARTMIPS_MFMA_PAD "v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "a"(b));
            else asm volatile(ARTMIPS_MFMA_PAD "v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
        } else {
            if constexpr (B_AGPR) asm volatile(ARTMIPS_MFMA_PAD "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
            else asm volatile(ARTMIPS_MFMA_PAD "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
        }
    }
};
template <> struct Elem<__bf16> {
    typedef bf16x8 V;
    static __device__ __forceinline__ f32x4 mfma(V a, V b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ __bf16 from_float(float x) { return (__bf16)x; }
    template <bool B_AGPR, bool FIRST>
    static __device__ __forceinline__ void mfma_asm(f32x4& c, const V& a, const V& b) {
        if constexpr (FIRST) {
            if constexpr (B_AGPR) asm volatile(ARTMIPS_MFMA_PAD "v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "a"(b));
            else asm volatile(ARTMIPS_MFMA_PAD "v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
        } else {
            if constexpr (B_AGPR) asm volatile(ARTMIPS_MFMA_PAD "v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
            else asm volatile(ARTMIPS_MFMA_PAD "v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
        }
    }
};

// fp32 storage (BASELINE config 5's tolerance check): exact-f32 MFMA, K = 4 per instruction; a
// 16-byte fragment holds 4 consecutive k of one row, consumed by 4 chained MFMAs (lane group g of
// MFMA t multiplies element t of its fragment, for the row operand and the query operand alike).
template <> struct Elem<float> {
    typedef f32x4 V;
    static __device__ __forceinline__ f32x4 mfma(V a, V b, f32x4 c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
        return c;
    }
    static __device__ __forceinline__ float from_float(float x) { return x; }
};

// Orderable 32-bit image of a float: larger float <=> larger unsigned. -0 is folded into +0 so
// that the two compare equal in the key as they do as floats.
__device__ __forceinline__ uint32_t f2ord(float f) {
    if (f == 0.0f) f = 0.0f;
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}
// 64-bit selection key: larger key == better candidate (higher score, then LOWER passage id).
__device__ __forceinline__ u64 make_key(float score, int32_t id) {
    uint32_t lo = ~(((uint32_t)id) ^ 0x80000000u);
    return ((u64)f2ord(score) << 32) | (u64)lo;
}
__device__ __forceinline__ float key_score(u64 key) { return ord2f((uint32_t)(key >> 32)); }
__device__ __forceinline__ int32_t key_id(u64 key) {
    return (int32_t)((~(uint32_t)key) ^ 0x80000000u);
}

__device__ __forceinline__ void store_score(void* out, int dtype, int64_t idx, float s) {
    if (dtype == 0) ((_Float16*)out)[idx] = (_Float16)s;
    else if (dtype == 1) ((__bf16*)out)[idx] = (__bf16)s;
    else ((float*)out)[idx] = s;
}

// ---------------------------------------------------------------------------------------------
// pack_queries_kernel: [q_count, dim] (f16 / bf16 / f32) -> [q_tile, dim] storage dtype, zero padded.
// Only launched when the queries are not in the storage dtype already: otherwise the scan reads the
// caller's buffer directly (ScanArgs::q_rows).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_queries_kernel(const void* __restrict__ q, int q_dtype, int q_count, int q_tile,
                                    int dim, T* __restrict__ out) {
    const int total = q_tile * dim;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int row = i / dim;
        float v = 0.0f;
        if (row < q_count) {
            if (q_dtype == 0) v = (float)((const _Float16*)q)[i];
            else if (q_dtype == 1) v = (float)((const __bf16*)q)[i];
            else v = ((const float*)q)[i];
        }
        out[i] = Elem<T>::from_float(v);
    }
}

// ---------------------------------------------------------------------------------------------
// scan_kernel
// ---------------------------------------------------------------------------------------------
struct ScanArgs {
    const void* rows;     // [n_rows, DIM] storage dtype
    long long n_rows;
    long long n_work;     // tiles (or sample tiles) to visit in this launch
    long long tile_stride;// actual tile = work index * tile_stride  (1 for the full scan)
    const void* qpad;     // [q_rows, DIM] storage dtype: the caller's queries, or the packed tile
    int q_rows;           // rows readable at qpad; query columns beyond it re-read row q_rows-1
    int q_count;          // real queries in the tile (<= 64*NQ)
    const float* thr;     // [64*NQ] FILTER: keep score >= thr[q]
    int2* lists;          // FILTER: [grid*NQ*256][cap] (score bits, local row)
    int* counts;          // FILTER: [64*NQ][grid*4] candidates seen by each lane (may exceed cap)
    int cap;
    float* gmax;          // GROUPMAX: [64*NQ][grid*4] running max of each lane
    // Gang mode (large batches): `gang` (2 or 4) workgroups that share an XCD walk the SAME tile
    // sequence, each with its own block of q_tile queries, so that one HBM read of a tile serves
    // gang * q_tile queries (the first reader misses, the others hit that XCD's L2: loads use the
    // default cache policy then).  Member m of a gang uses queries [m*q_tile, (m+1)*q_tile) and the
    // m-th region of thr / lists / counts / gmax, each region laid out as for a launch of grid/gang
    // workgroups; with the regions back to back the per-query arrays are simply flat over
    // gang * q_tile queries.  gang <= 1: off.
    int gang;
    int q_tile;           // queries per workgroup (64 * NQ * NW / 4)
    long long lists_member_stride;  // int2 entries between the list regions of two members
    // Two-part FILTER scan (large k): part A visits every P-th tile (tile_stride = P; P = 4 or 8), the
    // k-th best score among its survivors then raises thr, and part B visits the other tiles (tile_skip
    // = P: work index w -> tile (w / (P-1)) * P + w % (P-1) + 1) and appends to the same lists
    // (counts_continue: the lanes' counters start at what part A left in `counts`; direct list writes
    // only, STG == 0).
    int tile_skip;
    int counts_continue;
    unsigned long long* stamps;  // diagnostic build only (ARTMIPS_STAMPS): [8 waves][24] shader-clock stamps of one
                                 // iteration of one workgroup (see ARTMIPS_STAMP in scan_kernel); NULL = off
    // Gang rendezvous (8-wave gang kernels): one progress word per gang member, [logical workgroup][8], value
    // (epoch << 20) | stage; NULL = off.  Words of other launches (other epochs) read as "not started".
    unsigned int* gang_progress;
    unsigned int gang_epoch;   // 12 bits, changes with every launch that shares the words
    int sync_lead;             // stages a member may be ahead of the slowest before it pauses (0 = default)
    float thr_shift;      // diagnostic build only (ARTMIPS_THR_SHIFT, in 1/100): added to every threshold, to see how the
                          // scan time follows the number of survivors (results are then incomplete)
    int ablate;           // read by the diagnostic build only (-DARTMIPS_DIAG, ARTMIPS_ABLATE).  Bits 1..32 switch
                          // parts of the loop off (results are then meaningless): 1 = no refill DMA, 2 = no
                          // barrier, 4 = no compute, 8 = no LDS reads, 16 = no filter, 32 = reads never waited
                          // for.  Bits 64..256 select code variants with identical results, for A/B runs in one
                          // process: 64 = DMA pieces one by one, 256 = no one-branch pre-check in the filter.
                          // Again meaningless results: 512 = refill issued but never waited for, 1024 = every refill
                          // re-reads one tile (served by the CU's vector L1), 2048 = threshold +inf and nothing else,
                          // 16384 = survivors found and counted but not stored; 8192 = no gang rendezvous (results valid)
};

// LDS image of a stage (16 rows): the rows as they lie in memory (row-major, contiguous), cut into
// 1-KiB pieces; a DMA wave-instruction fetches 1 KiB of consecutive addresses.  The low four bits of
// every 16-byte chunk index are XORed with the row number, which spreads the 16 rows of a column over
// the 16 bank slots (conflict-free ds_read_b128, profiles/r01_k_pmc_sq_counters.json).  Lane group g
// multiplies chunk 4s+g in k-step s.  (An "8 rows x 128 B per piece" image was measured too and was
// no faster: profiles/r01_b_scan_sweep.txt.)
// LDS-DMA with a scalar base and a 32-bit per-lane offset (no vector address arithmetic at all).
// hipcc does not select this addressing form for the builtin, hence inline asm.  M0 (the LDS
// destination) is compiler-reserved: it is saved, written and restored inside the one statement
// that uses it.  hipcc neither counts these loads nor pads their hazards: the opening s_nop 4 covers a
// base that a vector instruction produced, s_nop 0 the M0 write, and the caller waits with vmcnt.
// The instruction adds its immediate offset field to BOTH addresses (the
// global source and the LDS destination), so consecutive 1-KiB pieces of a stage can share one M0 value:
// piece i of a group uses offset i*1024 - DMA_BIAS, its lane offset is pre-biased by the opposite amount
// and M0 holds the group's first destination + DMA_BIAS (offsets span [-2048, 3072] of the 13-bit signed
// field, so a group has up to 6 pieces).  One M0 save / write / restore and one hazard pad per GROUP
// instead of per piece: the scalar instructions around a piece cost the issuing wave as much as the
// piece itself when it is the only wave on its SIMD.
constexpr int DMA_BIAS = 2048;
// (the gang kernels' refill with the cache policy bits sc1, sc0, sc0 sc1 instead of none: +-0.5 %,
// profiles/r03_d_phased_ab.txt)
#define ARTMIPS_GLDS_LINE(op, off, nt) "global_load_lds_dwordx4 %" #op ", %2 offset:" #off nt "\n\t"
#define ARTMIPS_GLDS_HEAD "s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
#define ARTMIPS_GLDS_TAIL "s_mov_b32 m0, %0"
template <int AUX, int N>
__device__ __forceinline__ void glds16_group(const void* base, const uint32_t* voff, uint32_t m0_val) {
    static_assert(AUX == 0 || AUX == 2, "asm DMA path knows the default and the nt policy");
    static_assert(N >= 1 && N <= 6 && N != 5, "a group is 1, 2, 3, 4 or 6 pieces");
    uint32_t keep;
#define ARTMIPS_GLDS_BODY(NT)                                                                                 \
    if constexpr (N == 1)                                                                                      \
        asm volatile(ARTMIPS_GLDS_HEAD ARTMIPS_GLDS_LINE(3, -2048, NT) ARTMIPS_GLDS_TAIL                        \
                     : "=&s"(keep) : "s"(m0_val), "s"(base), "v"(voff[0]) : "memory");                         \
    else if constexpr (N == 2)                                                                                 \
        asm volatile(ARTMIPS_GLDS_HEAD ARTMIPS_GLDS_LINE(3, -2048, NT) ARTMIPS_GLDS_LINE(4, -1024, NT)          \
                     ARTMIPS_GLDS_TAIL                                                                         \
                     : "=&s"(keep) : "s"(m0_val), "s"(base), "v"(voff[0]), "v"(voff[1]) : "memory");           \
    else if constexpr (N == 3)                                                                                 \
        asm volatile(ARTMIPS_GLDS_HEAD ARTMIPS_GLDS_LINE(3, -2048, NT) ARTMIPS_GLDS_LINE(4, -1024, NT)          \
                     ARTMIPS_GLDS_LINE(5, 0, NT) ARTMIPS_GLDS_TAIL                                             \
                     : "=&s"(keep) : "s"(m0_val), "s"(base), "v"(voff[0]), "v"(voff[1]), "v"(voff[2])          \
                     : "memory");                                                                              \
    else if constexpr (N == 4)                                                                                 \
        asm volatile(ARTMIPS_GLDS_HEAD ARTMIPS_GLDS_LINE(3, -2048, NT) ARTMIPS_GLDS_LINE(4, -1024, NT)          \
                     ARTMIPS_GLDS_LINE(5, 0, NT) ARTMIPS_GLDS_LINE(6, 1024, NT) ARTMIPS_GLDS_TAIL               \
                     : "=&s"(keep) : "s"(m0_val), "s"(base), "v"(voff[0]), "v"(voff[1]), "v"(voff[2]),         \
                       "v"(voff[3]) : "memory");                                                               \
    else                                                                                                       \
        asm volatile(ARTMIPS_GLDS_HEAD ARTMIPS_GLDS_LINE(3, -2048, NT) ARTMIPS_GLDS_LINE(4, -1024, NT)          \
                     ARTMIPS_GLDS_LINE(5, 0, NT) ARTMIPS_GLDS_LINE(6, 1024, NT) ARTMIPS_GLDS_LINE(7, 2048, NT)  \
                     ARTMIPS_GLDS_LINE(8, 3072, NT) ARTMIPS_GLDS_TAIL                                          \
                     : "=&s"(keep) : "s"(m0_val), "s"(base), "v"(voff[0]), "v"(voff[1]), "v"(voff[2]),         \
                       "v"(voff[3]), "v"(voff[4]), "v"(voff[5]) : "memory");
    if constexpr (AUX == 2) { ARTMIPS_GLDS_BODY(" nt") } else { ARTMIPS_GLDS_BODY("") }
#undef ARTMIPS_GLDS_BODY
}
// one piece of a group on its own (the refill that rides between the MFMA chunks, mg_dma_after_chunk):
// I = index of the piece inside its group, voff = its biased lane offset, dst = its LDS destination
template <int AUX, int I>
__device__ __forceinline__ void glds16_piece(const void* base, uint32_t voff, uint32_t dst) {
    static_assert(AUX == 0 || AUX == 2, "asm DMA path knows the default and the nt policy");
    static_assert(I >= 0 && I < 6, "piece index inside a group");
    constexpr int OFF = I * 1024 - DMA_BIAS;
    const uint32_t m0_val = dst - (uint32_t)OFF;
    uint32_t keep;
    if constexpr (AUX == 2)
        asm volatile(ARTMIPS_GLDS_HEAD "global_load_lds_dwordx4 %3, %2 offset:%4 nt\n\t" ARTMIPS_GLDS_TAIL
                     : "=&s"(keep) : "s"(m0_val), "s"(base), "v"(voff), "n"(OFF) : "memory");
    else
        asm volatile(ARTMIPS_GLDS_HEAD "global_load_lds_dwordx4 %3, %2 offset:%4\n\t" ARTMIPS_GLDS_TAIL
                     : "=&s"(keep) : "s"(m0_val), "s"(base), "v"(voff), "n"(OFF) : "memory");
}
// One dword per lane into LDS, read at device scope (past the CU's vector L1): the gang rendezvous reads the
// other members' progress words with it.  A DMA, not a register load: a register load would have to be
// waited for, and vmcnt retires in order, i.e. behind every refill piece in flight.
__device__ __forceinline__ void glds4_device_scope(const void* base, uint32_t voff, uint32_t dst) {
    uint32_t keep;
    asm volatile(ARTMIPS_GLDS_HEAD "global_load_lds_dword %3, %2 sc0 sc1\n\t" ARTMIPS_GLDS_TAIL
                 : "=&s"(keep) : "s"(dst), "s"(base), "v"(voff) : "memory");
}
// diagnostic build (ARTMIPS_ABLATE & 64): the pieces of a stage one by one, each with its own M0 round trip
template <int AUX, int PPW, int GS, int PP>
__device__ __forceinline__ void glds16_pieces(const void* base, const uint32_t* voff, uint32_t dst0) {
    if constexpr (PP < PPW) {
        glds16_piece<AUX, PP % GS>(base, voff[PP], dst0 + PP * 1024);
        glds16_pieces<AUX, PPW, GS, PP + 1>(base, voff, dst0);
    }
}
// pieces per M0 value for a wave that issues PPW pieces of a stage
constexpr int dma_group_size(int ppw) { return ppw <= 4 ? ppw : (ppw % 6 == 0 ? 6 : (ppw % 4 == 0 ? 4 : (ppw % 3 == 0 ? 3 : 1))); }

// Pre-check of the filter for NQ column groups in ONE asm statement: does any lane of the wave hold a score
// >= its threshold in any group?  Two instructions find a lane's largest score of a group (NaNs are ignored
// as fmaxf ignores them; MFMA results are never signalling), one compares it, the lane masks are ORed
// on the scalar unit.  Written as one statement because every instruction outside the MFMAs costs a
// lone wave 6-8 cycles: left to hipcc the same test took ~45 instructions (pads after each asm VALU
// result, a ballot rebuilt through v_cndmask / v_cmp) and 325 cycles a stage (profiles/r03_c_stamps.txt).
template <int NQ>
__device__ __forceinline__ bool any_survivor(const f32x4* acc, const float* thr) {
    static_assert(NQ >= 2 && NQ <= 4, "multi-group shapes");
    unsigned long long m0, m1, m2 = 0, m3 = 0;
    float t0, t1;
    if constexpr (NQ == 2) {
        asm volatile("v_max3_f32 %2, %4, %5, %6\n\t"
                     "v_max3_f32 %3, %8, %9, %10\n\t"
                     "v_max_f32 %2, %2, %7\n\t"
                     "v_max_f32 %3, %3, %11\n\t"
                     "v_cmp_ge_f32 %0, %2, %12\n\t"
                     "v_cmp_ge_f32 %1, %3, %13\n\t"
                     "s_or_b64 %0, %0, %1"
                     : "=&s"(m0), "=&s"(m1), "=&v"(t0), "=&v"(t1)
                     : "v"(acc[0][0]), "v"(acc[0][1]), "v"(acc[0][2]), "v"(acc[0][3]),
                       "v"(acc[1][0]), "v"(acc[1][1]), "v"(acc[1][2]), "v"(acc[1][3]), "v"(thr[0]), "v"(thr[1]));
    } else if constexpr (NQ == 3) {
        asm volatile("v_max3_f32 %3, %5, %6, %7\n\t"
                     "v_max3_f32 %4, %9, %10, %11\n\t"
                     "v_max_f32 %3, %3, %8\n\t"
                     "v_max_f32 %4, %4, %12\n\t"
                     "v_cmp_ge_f32 %0, %3, %17\n\t"
                     "v_cmp_ge_f32 %1, %4, %18\n\t"
                     "v_max3_f32 %3, %13, %14, %15\n\t"
                     "v_max_f32 %3, %3, %16\n\t"
                     "v_cmp_ge_f32 %2, %3, %19\n\t"
                     "s_or_b64 %0, %0, %1\n\t"
                     "s_or_b64 %0, %0, %2"
                     : "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&v"(t0), "=&v"(t1)
                     : "v"(acc[0][0]), "v"(acc[0][1]), "v"(acc[0][2]), "v"(acc[0][3]),
                       "v"(acc[1][0]), "v"(acc[1][1]), "v"(acc[1][2]), "v"(acc[1][3]),
                       "v"(acc[2][0]), "v"(acc[2][1]), "v"(acc[2][2]), "v"(acc[2][3]),
                       "v"(thr[0]), "v"(thr[1]), "v"(thr[2]));
    } else {
        asm volatile("v_max3_f32 %4, %6, %7, %8\n\t"
                     "v_max3_f32 %5, %10, %11, %12\n\t"
                     "v_max_f32 %4, %4, %9\n\t"
                     "v_max_f32 %5, %5, %13\n\t"
                     "v_cmp_ge_f32 %0, %4, %22\n\t"
                     "v_cmp_ge_f32 %1, %5, %23\n\t"
                     "v_max3_f32 %4, %14, %15, %16\n\t"
                     "v_max3_f32 %5, %18, %19, %20\n\t"
                     "v_max_f32 %4, %4, %17\n\t"
                     "v_max_f32 %5, %5, %21\n\t"
                     "v_cmp_ge_f32 %2, %4, %24\n\t"
                     "v_cmp_ge_f32 %3, %5, %25\n\t"
                     "s_or_b64 %0, %0, %1\n\t"
                     "s_or_b64 %2, %2, %3\n\t"
                     "s_or_b64 %0, %0, %2"
                     : "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&v"(t0), "=&v"(t1)
                     : "v"(acc[0][0]), "v"(acc[0][1]), "v"(acc[0][2]), "v"(acc[0][3]),
                       "v"(acc[1][0]), "v"(acc[1][1]), "v"(acc[1][2]), "v"(acc[1][3]),
                       "v"(acc[2][0]), "v"(acc[2][1]), "v"(acc[2][2]), "v"(acc[2][3]),
                       "v"(acc[3][0]), "v"(acc[3][1]), "v"(acc[3][2]), "v"(acc[3][3]),
                       "v"(thr[0]), "v"(thr[1]), "v"(thr[2]), "v"(thr[3]));
    }
    return m0 != 0ull;
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}


// ---- multi-group compute in inline asm (NQ >= 2) ----------------------------------------------
// With 2..4 column groups the query operand (NQ*KSTEPS fragments of 4 registers) no longer fits the
// 256 architectural VGPRs beside the row fragments.  hipcc then parks fragments in AGPRs and copies
// each one back with four v_accvgpr_read before its MFMA, which doubles the issue cost.  MFMA can
// read its B operand straight from an AGPR, so this path says so explicitly: fragments with index
// j*KSTEPS+s >= VB are bound to AGPRs ("a"), the rest to VGPRs.  LDS reads are inline asm too, so
// the next chunk of row fragments is requested before the MFMAs of the current chunk and waited for
// with a counted lgkmcnt.
template <int OFF> __device__ __forceinline__ void lds_read128_asm(half8& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int OFF> __device__ __forceinline__ void lds_read128_asm(bf16x8& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}

// request the CH row fragments of chunk C (k-steps C*CH .. C*CH+CH-1) into af[0..CH)
template <int C, int CH, int I, typename V>
__device__ __forceinline__ void mg_load_chunk(V* af, const uint32_t (&base)[4]) {
    if constexpr (I < CH) {
        constexpr int s = C * CH + I;
        lds_read128_asm<(s >> 2) * 256>(af[I], base[s & 3]);
        mg_load_chunk<C, CH, I + 1, V>(af, base);
    }
}

// wait until all but the N youngest LDS reads are back; naming every fragment as a read-write
// operand keeps the MFMAs that consume them below this statement
template <int N, int CH, int I, typename V>
__device__ __forceinline__ void mg_wait_chunk(V* af) {
    if constexpr (I < CH) {
        if constexpr (I == 0) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(af[0]) : "n"(N));
        else asm volatile("" : "+v"(af[I]));
        mg_wait_chunk<N, CH, I + 1, V>(af);
    }
}

// MFMAs of k-step s = C*CH+I for the NQ column groups
template <typename T, int NQ, int KSTEPS, int VB, int S, int J, typename V>
__device__ __forceinline__ void mg_mfma_groups(f32x4* acc, const V& a, const V (*bq)[KSTEPS]) {
    if constexpr (J < NQ) {
        Elem<T>::template mfma_asm<(J * KSTEPS + S >= VB), S == 0>(acc[J], a, bq[J][S]);
        mg_mfma_groups<T, NQ, KSTEPS, VB, S, J + 1, V>(acc, a, bq);
    }
}
// (Tried in round 3 and dropped: the fragment of k-step I of chunk C + 2 requested right behind the MFMAs
// of k-step I of chunk C, i.e. the LDS reads riding in the issue gaps between MFMAs instead of four
// together in front of a chunk.  The stamped iteration got no shorter and the launch 3-5 % slower:
// profiles/r03_a_gang_ab.txt, build v2.)
template <typename T, int NQ, int KSTEPS, int VB, int C, int CH, int I, typename V>
__device__ __forceinline__ void mg_mfma_chunk(f32x4* acc, const V* af, const V (*bq)[KSTEPS]) {
    if constexpr (I < CH) {
        mg_mfma_groups<T, NQ, KSTEPS, VB, C * CH + I, 0, V>(acc, af[I], bq);
        mg_mfma_chunk<T, NQ, KSTEPS, VB, C, CH, I + 1, V>(acc, af, bq);
    }
}
// The refill of the ring (this wave's PPW pieces of the stage NSTAGE-1 ahead) rides inside the
// compute phase: piece pp is issued after chunk FIRST + pp*SPAN/PPW.  With 8 waves the two waves
// that share a SIMD (w and w+4) use different halves of the phase (FIRST = 0 / NCH/2), so one
// wave's DMA issue (60-185 cycles a piece, MI355X_MICROARCH.md) overlaps the other's MFMAs.
struct MgDma {
    const void* tile;        // uniform: first byte of the tile to fetch
    const uint32_t* lane_off;// [PPW] byte offset of this lane's chunk inside the tile
    uint32_t dst0;           // uniform: LDS address of this wave's first piece
    bool on;                 // uniform: false = the caller issued the stage itself
    bool late;               // uniform: this wave issues in the second half of the phase
    bool noread;             // diagnostic build only (ARTMIPS_ABLATE & 8): skip the LDS reads and waits
    bool nowait;             // diagnostic build only (ARTMIPS_ABLATE & 32): issue the LDS reads, never wait
    bool stamp;              // diagnostic build only: this iteration records clock stamps into st[]
    unsigned long long* st;
};
// In-kernel stamps (a build of their own: -DARTMIPS_DIAG -DARTMIPS_STAMPS_BUILD -> libartmips_stamps.so, run
// with ARTMIPS_STAMPS=1): s_memtime into a scalar pair, no wait (the counter it shares with the LDS reads is
// drained once, at the end of the stamped iteration), one iteration of one workgroup only.  The stamped
// iteration itself is slightly disturbed (a counted lgkmcnt wait may be satisfied by a stamp instead of a
// row fragment: its scores are then garbage) and the uniform branch around every stamp costs all other
// iterations a few per cent: the build is for reading where an iteration's cycles go, nothing else.
#ifdef ARTMIPS_STAMPS_BUILD
#define ARTMIPS_STAMP(on, arr, i) do { if (on) asm volatile("s_memtime %0" : "=s"((arr)[i])); } while (0)
#else
#define ARTMIPS_STAMP(on, arr, i) do { } while (0)
#endif
template <int AUX, int PPW, int NCH, int NW, int C, int PP>
__device__ __forceinline__ void mg_dma_after_chunk(const MgDma& d) {
    if constexpr (PP < PPW) {
        constexpr int SPAN = (NW == 8) ? NCH / 2 : NCH;
        constexpr int AT = (PP * SPAN) / PPW;          // chunk (inside the span) after which piece PP goes
        constexpr int GI = PP % dma_group_size(PPW);   // index of the piece inside its group (lane_off is biased for it)
        if constexpr (NW == 8) {
            if constexpr (C == AT) { if (d.on && !d.late) glds16_piece<AUX, GI>(d.tile, d.lane_off[PP], d.dst0 + PP * 1024); }
            if constexpr (C == AT + SPAN) { if (d.on && d.late) glds16_piece<AUX, GI>(d.tile, d.lane_off[PP], d.dst0 + PP * 1024); }
        } else {
            if constexpr (C == AT) { if (d.on) glds16_piece<AUX, GI>(d.tile, d.lane_off[PP], d.dst0 + PP * 1024); }
        }
        mg_dma_after_chunk<AUX, PPW, NCH, NW, C, PP + 1>(d);
    }
}
// `pre` runs once per stage, behind the LDS read requests of the first two chunks and in front of the
// wait for them: the place where the previous stage's filter epilogue costs least (its VALU work
// overlaps the LDS latency, and the MFMA results it reads have settled while the wave sat at the
// barrier).
// ROLE 1 (the second wave of a SIMD in the phased 8-wave loop, see scan_kernel): chunks 0 and 1 were
// requested by the caller in front of the stage's barrier, and `pre` does not run here.
// The row fragments live in NBUF register buffers of CH k-steps; chunk C + NBUF - 1 is requested in front
// of the MFMAs of chunk C (NBUF = 2: the next chunk; NBUF = 4 with CH = 1: three k-steps ahead, which
// covers the LDS latency with the few registers the 8-wave shape has left, even while the other wave of
// the SIMD is not computing).
template <int NBUF, int CH, int K, typename V>
__device__ __forceinline__ void mg_load_first(V (*af)[CH], const uint32_t (&base)[4]) {
    if constexpr (K < NBUF - 1) {
        mg_load_chunk<K, CH, 0, V>(af[K], base);
        mg_load_first<NBUF, CH, K + 1, V>(af, base);
    }
}
template <typename T, int NQ, int KSTEPS, int VB, int NCH, int CH, int NBUF, int AUX, int PPW, int NW, int C,
          int ROLE, typename V, typename Pre>
__device__ __forceinline__ void mg_compute(f32x4* acc, V (*af)[CH], const V (*bq)[KSTEPS],
                                           const uint32_t (&base)[4], const MgDma& d, const Pre& pre) {
    if constexpr (C < NCH) {
        constexpr int PD = NBUF - 1;   // chunks of look-ahead
        static_assert(NCH >= NBUF - 1, "more look-ahead than chunks");
        if (!d.noread) {
            if constexpr (C == 0 && ROLE == 0) mg_load_first<NBUF, CH, 0, V>(af, base);
            if constexpr (C + PD < NCH) mg_load_chunk<C + PD, CH, 0, V>(af[(C + PD) % NBUF], base);
        }
        if constexpr (C == 0 && ROLE == 0) {
            ARTMIPS_STAMP(d.stamp, d.st, 4);
            pre();
            ARTMIPS_STAMP(d.stamp, d.st, 5);
        }
        if (!d.noread && !d.nowait) {
            // LDS returns in order: with the chunks younger than C still out, chunk C is back
            constexpr int YOUNGER = (NCH - 1 - C < PD) ? NCH - 1 - C : PD;
            mg_wait_chunk<YOUNGER * CH, CH, 0, V>(af[C % NBUF]);
        }
        mg_mfma_chunk<T, NQ, KSTEPS, VB, C, CH, 0, V>(acc, af[C % NBUF], bq);
        constexpr int SDIV = NCH > 12 ? NCH / 12 : 1;   // at most 12 stamps inside a stage
        if constexpr (C % SDIV == SDIV - 1) ARTMIPS_STAMP(d.stamp, d.st, 6 + C / SDIV);
        mg_dma_after_chunk<AUX, PPW, NCH, NW, C, 0>(d);
        mg_compute<T, NQ, KSTEPS, VB, NCH, CH, NBUF, AUX, PPW, NW, C + 1, ROLE, V, Pre>(acc, af, bq, base, d, pre);
    }
}
// MFMA results must not be read by ordinary VALU code before the pipeline has written them:
// hipcc pads nothing after inline asm
template <int NQ, int J>
__device__ __forceinline__ void mg_settle(f32x4* acc) {
    if constexpr (J < NQ) {
        if constexpr (J == 0) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]));
        else asm volatile("" : "+v"(acc[J]));
        mg_settle<NQ, J + 1>(acc);
    }
}

// the same in front of the deferred epilogue: the wave has since passed a counted wait, the barrier,
// the loop head and four LDS read requests (well over the 11 wait states an 8-pass MFMA needs before
// VALU may read its result); the pad is margin against the compiler moving that code
template <int NQ, int J>
__device__ __forceinline__ void mg_settle_short(f32x4* acc) {
    if constexpr (J < NQ) {
        if constexpr (J == 0) asm volatile("s_nop 7" : "+v"(acc[0]));
        else asm volatile("" : "+v"(acc[J]));
        mg_settle_short<NQ, J + 1>(acc);
    }
}

// Load the query operand.  Fragment f = j*KSTEPS + s goes to a VGPR tuple (ordinary load) or, on the
// asm path for f >= VB, straight into an AGPR tuple (global_load can target AGPRs; an ordinary load
// followed by a copy would be re-materialised by hipcc inside the loop).
template <typename T, int NQ, int KSTEPS, int VB, bool ASM_PATH, int QG, int F, typename V>
__device__ __forceinline__ void load_queries(V (*bq)[KSTEPS], const T* qbase, int col, int q_rows, int g) {
    if constexpr (F < NQ * KSTEPS) {
        constexpr int j = F / KSTEPS, s = F % KSTEPS;
        const int chunk = 4 * s + g;
        constexpr int EPC = 16 / (int)sizeof(T);            // elements per 16-byte chunk
        constexpr int DIMQ = KSTEPS * 4 * EPC;              // row width in elements
        const int qrow = (j * QG + col < q_rows) ? (j * QG + col) : (q_rows - 1);
        const T* src = qbase + (size_t)qrow * DIMQ + chunk * EPC;
        if constexpr (ASM_PATH && F >= VB) {
            asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(bq[j][s]) : "v"(src) : "memory");
        } else {
            bq[j][s] = *(const V*)src;
        }
        load_queries<T, NQ, KSTEPS, VB, ASM_PATH, QG, F + 1, V>(bq, qbase, col, q_rows, g);
    }
}
template <int NQ, int KSTEPS, int VB, bool ASM_PATH, int F, typename V>
__device__ __forceinline__ void settle_queries(V (*bq)[KSTEPS]) {
    if constexpr (F < NQ * KSTEPS) {
        constexpr int j = F / KSTEPS, s = F % KSTEPS;
        if constexpr (ASM_PATH && F >= VB) asm volatile("" : "+a"(bq[j][s]));
        else asm volatile("" ::"v"(bq[j][s]));
        settle_queries<NQ, KSTEPS, VB, ASM_PATH, F + 1, V>(bq);
    }
}

// A workgroup has NW waves (4 or 8); QG = 16*NW queries form a column group, NT = 64*NW threads.
// Query q of the pass lives in column group j = q / QG, wave w = (q % QG) / 16, MFMA column r = q % 16.
// List of (workgroup b, group j, thread tid): index (b*NQ + j)*NT + tid.
// NW = 8 puts two waves on every SIMD: while one issues its LDS-DMA pieces or waits for LDS, the
// other keeps the matrix pipe busy (profiles/r01_k_pmc_sq_counters.json: with one wave per SIMD the
// pipe idles 55 % of the time at 256 queries per sweep).
// TOPT (GROUPMAX only): sample scores each lane keeps per query.  1 = its maximum; 4 = its four best,
// which lets pass 0 reach the same number of sample values per query with a quarter of the
// workgroups (used for large k, where the value count 8k makes pass 0 launch thousands of workgroups
// that each reload the whole query operand).
template <typename T, int DIM, int NSTAGE, int MODE, int AUX, int NQ, int STG = 1, int NW = 4, int TOPT = 1>
__global__ __launch_bounds__(NW * 64) void scan_kernel(ScanArgs a) {
    typedef typename Elem<T>::V V;
    constexpr int ES = (int)sizeof(T);     // bytes per stored element
    constexpr int EPC = 16 / ES;           // elements per 16-byte chunk
    constexpr int ROWB = DIM * ES;         // bytes per row
    constexpr int KSTEPS = ROWB / 64;      // k-steps per row: 4 lane groups x one 16-byte chunk each
    constexpr int CPR = ROWB / 16;         // 16-byte chunks per row
    constexpr int PIECES = TILE_ROWS * ROWB / 1024;  // 1-KiB pieces per 16-row stage
    constexpr int NT = NW * 64;            // threads per workgroup
    constexpr int QG = NW * 16;            // queries per column group
    constexpr int PPW = PIECES / NW;       // pieces each wave issues per stage
    constexpr int STAGE_BYTES = TILE_ROWS * ROWB;
    constexpr int MPF = (ES == 4) ? 4 : 1; // MFMA instructions per fragment
    static_assert(DIM % 128 == 0, "DIM must be a multiple of 128");
    static_assert(PIECES % NW == 0, "pieces must split evenly over the waves");
    static_assert((NSTAGE - 2) * PPW <= 63, "vmcnt field is 6 bits");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15;  // MFMA row (row operand) / column (query operand) of this lane
    const int g = lane >> 4;  // MFMA k-group of this lane
#ifdef ARTMIPS_DIAG
    const int ablate = a.ablate;   // timing experiments of the diagnostic build
    const float thr_shift = a.thr_shift;
#else
    constexpr int ablate = 0;      // the shipped library has no ablation code
    constexpr float thr_shift = 0.0f;
#endif
    // ---- which tiles, which queries (gang mode: see ScanArgs)
    long long b = blockIdx.x;       // logical workgroup index: tiles b, b + grid, ...
    long long grid = gridDim.x;     // logical grid
    int q_count = a.q_count, q_rows = a.q_rows;
    const T* qbase = (const T*)a.qpad;
    const float* thr_base = a.thr;
    int2* lists_base = a.lists;
    size_t per_query_off = 0;       // into counts / gmax: (first query of this member) * values per query
    if (a.gang > 1) {
        // workgroups b and b + 8 run on the same XCD under round-robin dispatch (a speed assumption
        // only: any placement gives the same results)
        const int xcd = (int)(b & 7), l = (int)(b >> 3);
        const int member = l % a.gang, grp = l / a.gang;
        grid = grid / a.gang;
        b = (long long)grp * 8 + xcd;
        const int q0 = member * a.q_tile;
        q_count = a.q_count - q0;
        if (q_count > a.q_tile) q_count = a.q_tile;
        if (q_count > 0) {
            q_rows = a.q_rows - q0;
            qbase += (size_t)q0 * DIM;
        } else {
            q_count = 0;            // a member without queries streams along and keeps nothing
        }
        thr_base += q0;
        lists_base += (size_t)member * (size_t)a.lists_member_stride;
        per_query_off = (size_t)q0;
    }
    const bool wave_active = (w * 16) < q_count && !(ablate & 4);   // group 0 of this wave has a real query
    // row-operand fragments are read in NCH chunks when the query operand leaves too few registers
    constexpr bool PHASED = (NW == 8) && (NQ >= 2) && (ARTMIPS_PHASED != 0);   // see below
    constexpr int NCH = (KSTEPS >= 24) ? ((NW == 8) ? ((NQ >= 2) ? (PHASED ? KSTEPS : 12) : 8) : (NQ >= 4) ? 6 : 4) : 1;
    constexpr int CH = KSTEPS / NCH;
    constexpr int NBUF = PHASED ? 4 : 2;   // register buffers of CH k-steps of row fragments
    static_assert(KSTEPS % NCH == 0, "k-steps must split evenly into chunks");
    // query fragments j*KSTEPS+s < VB live in VGPRs, the others in AGPRs (asm path, NQ >= 2)
    constexpr int VB = (NW == 8) ? NQ * KSTEPS : (NQ <= 3) ? KSTEPS : KSTEPS + (KSTEPS * 3) / 8;
    static_assert((NQ * KSTEPS - VB) * 4 <= 252 || NQ == 1, "query operand does not fit the AGPR file");
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    const T* __restrict__ rows = (const T*)a.rows;
    const long long n_rows = a.n_rows;

    // ---- query operand: KSTEPS fragments of 8 elements, resident in registers for the whole launch
    V bq[NQ][KSTEPS];
    float thr[NQ];
    constexpr bool ASM_PATH = (NQ >= 2);
    // refill issued between the MFMA chunks (mg_compute) instead of in one burst behind the barrier
    // (tried for the 4 x 4 shape too: one piece behind each of its six chunks was 5 % slower than the burst)
    // PHASED (8 waves, two per SIMD): the waves of the first half (role A) do a stage's non-MFMA work
    // (refill DMA, LDS read requests, filter of the previous stage) in FRONT of their MFMAs, the waves of
    // the second half (role B) BEHIND theirs (filter of this stage, refill DMA, and the first LDS read
    // requests of the NEXT stage, which therefore has to have landed one barrier early).  Right after the
    // barrier B's MFMAs run while A issues memory instructions, at the end A's run while B does; in between
    // the two share the pipe.  With both waves in the same order (rounds 1-2) the pipe idled while both
    // were outside their MFMAs and each wave got every second slot while both were inside.
    static_assert(!PHASED || NSTAGE >= 4, "the phased loop gives one stage of look-ahead away");
    // (the phased loop issues its refill in two bursts, role A's behind the barrier and role B's at the end of
    // the stage; with the pieces riding inside the stretches where a wave computes alone it was 4 % slower:
    // profiles/r03_d_phased_ab.txt)
    constexpr bool INLINE_DMA = (NW == 8) && !PHASED;
    load_queries<T, NQ, KSTEPS, VB, ASM_PATH, QG, 0>(bq, qbase, w * 16 + r, q_rows, g);
#pragma unroll
    for (int j = 0; j < NQ; ++j)
        thr[j] = (MODE == MODE_FILTER) ? ((ablate & (63 | 512 | 1024 | 2048)) ? INFINITY : thr_base[j * QG + w * 16 + r] + thr_shift) : 0.0f;
    // (these loads are drained below, together with the DMA prologue that is issued behind them)

    // ---- per-lane constants of the LDS image
    // MFMA side: lane (r,g), k-step s reads byte r*ROWB + 256*(s>>2) + 16*((4*(s&3)+g) ^ r) of the stage
    int rd_off[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) rd_off[m] = r * ROWB + 16 * ((4 * m + g) ^ r);

    const long long n_work = a.n_work;
    const long long stride = a.tile_stride;
    const long long n_it = (b < n_work) ? (n_work - b + grid - 1) / grid : 0;
    // Work index -> tile (ScanArgs::tile_skip; tile counts fit 32 bits: n_rows < 2^31): with P = tile_skip,
    // wk -> (wk / (P-1)) * P + wk % (P-1) + 1, otherwise wk * stride.  This workgroup visits the work
    // indices b, b + grid, ... in order, so the tile is kept as a running (quotient, remainder) pair
    // that one add, one compare and two selects advance: no division and no branch in the streaming
    // loop (a single wave per SIMD pays for every scalar instruction between its MFMAs).
    const uint32_t t_mod = a.tile_skip ? (uint32_t)a.tile_skip - 1u : 1u;
    const uint32_t t_mult = a.tile_skip ? (uint32_t)a.tile_skip : (uint32_t)stride;
    const uint32_t t_add = a.tile_skip ? 1u : 0u;
    // (tile_skip is 0, 4 or 8: divisions by constants stay on the scalar unit, a division by a run-time
    // value would be expanded on the vector unit; readfirstlane tells hipcc the results are uniform)
    auto div_mod = [&](uint32_t x, uint32_t* rem) __attribute__((always_inline)) -> uint32_t {
        uint32_t qq = x, rr = 0;
        if (a.tile_skip == 8) { qq = x / 7u; rr = x % 7u; }
        else if (a.tile_skip) { qq = x / 3u; rr = x % 3u; }
        *rem = (uint32_t)__builtin_amdgcn_readfirstlane((int)rr);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)qq);
    };
    uint32_t t_dr, t_r0;
    const uint32_t t_dq = div_mod((uint32_t)grid, &t_dr);
    const uint32_t t_q0 = div_mod((uint32_t)b, &t_r0);
    struct TileWalk {
        uint32_t q, r;
    };
    const uint32_t t_last = (uint32_t)((n_rows + TILE_ROWS - 1) / TILE_ROWS - 1);   // last tile of the shard
    // the current tile, then one step on.  Past the workgroup's last item (the ring's tail refills) the
    // walk runs on and is clamped to the shard's last tile: any valid address keeps the vmcnt
    // arithmetic uniform, the data is never used.
    auto walk_next = [&](TileWalk& t) __attribute__((always_inline)) -> long long {
        uint32_t tile = t.q * t_mult + t.r + t_add;
        tile = tile < t_last ? tile : t_last;
        if (ablate & 1024) tile = (uint32_t)b & 7u;   // diagnostic build: every refill re-reads one L2-resident tile
        // (readfirstlane: the DMA takes its base from scalar registers, and hipcc must not move this
        // chain of uniform values to the vector unit)
        tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)tile);
        uint32_t nr = t.r + t_dr;
        const uint32_t carry = nr >= t_mod ? 1u : 0u;
        t.r = nr - (carry ? t_mod : 0u);
        t.q += t_dq + carry;
        return (long long)tile;
    };
    TileWalk walk_issue{t_q0, t_r0};      // stages are issued in order: prologue, then one per iteration
    // the filter asks for a stage's tile only when the stage has survivors: direct formula there, without
    // branches (the slow path is on the critical path of its workgroup): wk / d by a multiply-high with a
    // 32-bit magic number (d = 3: 0xAAAAAAAB >> 33, exact for every 32-bit wk; d = 7: 0x92492493 >> 34, exact below
    // 2^34 / 5 = 3.4e9; work indices are tile counts, below 2^27: tests/test_tile_formula_cpu.py),
    // and tile = q * (d + 1) + (wk - q * d) + 1; without a skip the magic is 0, so q = 0 and tile = wk * stride
    // (8-wave kernels only: the 4-wave shapes keep the branching form below; with the branch-free one the 4 x 4
    // gang scan measured 1.7 % slower, the 8 x 2 one 1.6 % faster: profiles/r03_f_slow_path.txt)
    auto tile_of = [&](long long wk) __attribute__((always_inline)) -> long long {
        if constexpr (NW == 8) {
            const uint32_t td_magic = a.tile_skip == 8 ? 0x92492493u : a.tile_skip ? 0xAAAAAAABu : 0u;
            const uint32_t td_shift = a.tile_skip == 8 ? 2u : 1u;
            const uint32_t w32 = (uint32_t)wk;
            const uint32_t qq = (uint32_t)(((unsigned long long)w32 * td_magic) >> 32) >> td_shift;
            return (long long)(qq * t_mult + (w32 - qq * t_mod) * (a.tile_skip ? 1u : t_mult) + t_add);
        } else {
            if (a.tile_skip) {
                const uint32_t w32 = (uint32_t)wk;
                return (a.tile_skip == 8) ? (long long)((w32 / 7u) * 8u + w32 % 7u + 1u)
                                          : (long long)((w32 / 3u) * 4u + w32 % 3u + 1u);
            }
            return wk * stride;
        }
    };

    // byte offset, inside a tile, of the 16-byte chunk that lane `lane` fetches for its pp-th piece
    // (DMA side): slot pos = p*64+lane of the stage holds chunk (pos % CPR) ^ (row & 15) of row pos / CPR
    // (biased for the grouped DMA issue: piece pp is the (pp % GS)-th of its group, see glds16_group)
    constexpr int GS = dma_group_size(PPW);
    static_assert(PPW % GS == 0, "pieces must split evenly into DMA groups");
    uint32_t lane_off[PPW];
#pragma unroll
    for (int pp = 0; pp < PPW; ++pp) {
        const int pos = (w * PPW + pp) * 64 + lane;
        const int trow = pos / CPR;
        lane_off[pp] = (uint32_t)(trow * ROWB + (((pos - trow * CPR) ^ (trow & 15)) * 16))
                       + (uint32_t)(DMA_BIAS - (pp % GS) * 1024);
    }

    // The shard's row buffer is allocated in whole tiles (artmips_begin pads it with zero rows), so a
    // DMA never needs a per-lane clamp: uniform tile base + per-lane constant.  Rows past n_rows are
    // scored like any other and dropped by the `row < n_rows` test of the filter.
    int fill_slot = 0;   // ring slot the next issued stage goes to
    auto issue_stage = [&]() __attribute__((always_inline)) {
        const long long row0 = walk_next(walk_issue) * TILE_ROWS;
        const char* tile = (const char*)rows + row0 * ROWB;
        const uint32_t dst0 = lds_base + (uint32_t)(fill_slot * STAGE_BYTES + w * PPW * 1024);
        // (the HBM-bound one-group kernel is 0.3 % faster with the pieces spaced by their scalar
        // instructions than with the burst: profiles/r03_a_gang_ab.txt)
        if (!ASM_PATH || (ablate & 64)) {
            glds16_pieces<AUX, PPW, GS, 0>(tile, lane_off, dst0);
            return;
        }
#pragma unroll
        for (int gi = 0; gi < PPW / GS; ++gi)
            glds16_group<AUX, GS>(tile, &lane_off[gi * GS], dst0 + (uint32_t)(gi * GS * 1024 + DMA_BIAS));
    };

    static_assert(TOPT == 1 || MODE == MODE_GROUPMAX, "TOPT belongs to pass 0");
    int cnt[NQ];
    float gm[NQ][TOPT];   // descending
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        cnt[j] = 0;
        if (MODE == MODE_FILTER && !STG && a.counts_continue)
            cnt[j] = a.counts[(per_query_off + (size_t)(j * QG + w * 16 + r)) * (size_t)(grid * 4) + (size_t)(b * 4 + g)];
#pragma unroll
        for (int t = 0; t < TOPT; ++t) gm[j][t] = -INFINITY;
    }
    // list of (b, group j, tid) starts at ((b*NQ + j)*256 + tid) * cap
    // (rebuilt from the thread index at every use: kept in a register pair for the whole loop, the 8-wave
    // shape spills it, and a reload from scratch waits for every DMA piece in flight)
    auto lane_id_opaque = [&]() __attribute__((always_inline)) -> int {   // thread index, not hoisted out of the loop
        int t = tid;
        asm volatile("" : "+v"(t));
        return t;
    };
    auto my_list = [&]() __attribute__((always_inline)) -> int2* {
        int t = tid;
        asm volatile("" : "+v"(t));
        return lists_base + ((size_t)b * NQ * NT + (size_t)t) * (size_t)a.cap;
    };
    const size_t list_stride = (size_t)NT * (size_t)a.cap;  // between column groups
    // Survivors are staged in LDS (SLOTS entries per lane and column group, behind the ring) and
    // copied to the lane's list in HBM only when the staging is full and at the end of the launch.
    // A global store inside the loop would sit in the vmcnt queue behind the DMAs in flight and make
    // the counted wait of the next stage retire one DMA piece too many (~one stage period lost per
    // store instruction, profiles/r01_e_sample_sweep.txt).
    constexpr int SLOTS = STAGING_BYTES / (NT * NQ * 8);
    static_assert(SLOTS >= 1, "staging area too small");
    int2* const stg = (int2*)(smem + NSTAGE * STAGE_BYTES);   // [NQ][SLOTS][NT]
    auto flush = [&](int j, int first, int last) __attribute__((always_inline)) {  // copy staged entries [first, last) to HBM
        for (int e = first; e < last; ++e)
            if (e < a.cap) my_list()[j * list_stride + e] = stg[(j * SLOTS + (e % SLOTS)) * NT + tid];
    };

    // Filter (or pass-0 maxima) of one finished stage: acc[j][i] = score(row = tile*16 + g*4 + i,
    // query = j*QG + w*16 + r)
    f32x4 acc[NQ];   // multi-group path: lives across iterations (its filter runs one stage late)
    auto epilogue = [&](long long it_e, f32x4* acc) __attribute__((always_inline)) {
        if constexpr (MODE == MODE_FILTER && NQ > 1) {
            // the common case (no survivor in any column group of this wave) leaves through ONE branch
            if (!(ablate & (16 | 256))) {
                if (!any_survivor<NQ>(acc, thr)) return;
            }
        }
        // (only a stage with survivors, and pass 0, need to know which rows these are)
        // (rows fit 31 bits: 32-bit row arithmetic and compares in the slow path)
        const int row_base = (int)(tile_of(b + it_e * grid) * TILE_ROWS) + g * 4;
        const int n_rows32 = (int)n_rows;
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            if (ablate & 16) {
                asm volatile("" ::"v"(acc[j]));   // diagnostic build: no filter, scores kept live
            } else if (MODE == MODE_FILTER) {
                const float m = fmaxf(fmaxf(acc[j][0], acc[j][1]), fmaxf(acc[j][2], acc[j][3]));
                if (__builtin_amdgcn_ballot_w64(m >= thr[j]) != 0ull) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = row_base + i;
                        if (acc[j][i] >= thr[j] && row < n_rows32) {
                            if (STG) {
                                const int slot_e = cnt[j] % SLOTS;
                                if (slot_e == 0 && cnt[j] > 0) flush(j, cnt[j] - SLOTS, cnt[j]);
                                stg[(j * SLOTS + slot_e) * NT + tid] = make_int2(__float_as_int(acc[j][i]), (int)row);
                            } else if (!(ablate & 16384)) {   // (16384: everything but the store)
                                // uniform base of this (workgroup, column group)'s lists + a 32-bit lane offset:
                                // one multiply-add instead of 64-bit pointer arithmetic per survivor
                                char* const lb = (char*)(lists_base + ((size_t)b * NQ + j) * list_stride);
                                if constexpr (NW == 8) {
                                    // a full list takes the entry in its last slot instead of testing for room:
                                    // the count runs on, and a count above the capacity reports the overflow
                                    // (8 waves only: the 4 x 4 shape measured 0.6 % slower with it)
                                    const uint32_t at = min((uint32_t)cnt[j], (uint32_t)a.cap - 1u);
                                    *(int2*)(lb + ((uint32_t)lane_id_opaque() * (uint32_t)a.cap + at) * 8u) =
                                        make_int2(__float_as_int(acc[j][i]), row);
                                } else if (cnt[j] < a.cap) {
                                    *(int2*)(lb + ((uint32_t)lane_id_opaque() * (uint32_t)a.cap + (uint32_t)cnt[j]) * 8u) =
                                        make_int2(__float_as_int(acc[j][i]), row);
                                }
                            }
                            ++cnt[j];
                        }
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = row_base + i;
                    // insertion into the lane's descending top-TOPT (NaN never enters)
                    float x = (row < n_rows32) ? acc[j][i] : -INFINITY;
#pragma unroll
                    for (int t = 0; t < TOPT; ++t) {
                        const float hi = fmaxf(gm[j][t], x);
                        x = fminf(gm[j][t], x);
                        gm[j][t] = hi;
                    }
                }
            }
        }
    };

    if (n_it > 0) {
        // prologue: NSTAGE-1 stages in flight, issued while the query operand is still on its way
#pragma unroll
        for (int s = 0; s < NSTAGE - 1; ++s) {
            issue_stage();
            ++fill_slot;
        }
        // Drain the operand loads (and with them the prologue, whose latency they share) before the
        // streaming loop, so that hipcc never has a reason to put a vmcnt(0) inside it (the asm loads
        // into AGPRs are not counted by hipcc at all).  The ring refills from iteration 0 on.
        wait_vmcnt<0>();
        settle_queries<NQ, KSTEPS, VB, ASM_PATH, 0>(bq);
#pragma unroll
        for (int j = 0; j < NQ; ++j) asm volatile("" ::"v"(thr[j]));
        // (role B reads stage 0 in front of iteration 0's barrier: every wave's prologue pieces have landed)
        if constexpr (PHASED) asm volatile("s_barrier" ::: "memory");
        const bool role_b = PHASED && wave_active && (w >= NW / 2);
        // Gang rendezvous.  The members of a gang share their tiles through L2 only while they are within
        // a few dozen stages of each other, and nothing kept them there: a member whose queries meet fewer
        // survivors runs ahead, the ring's look-ahead hides its misses from it, and at top-1000 the shard
        // came from HBM 3.6 times per sweep (L2 hit rate 0.53 where lock-step gives 0.875:
        // profiles/r03_e_gang_sync.txt).  So every fourth (8 waves) or eighth (4 waves) stage wave 0 of a
        // member publishes its stage number and asks for the members' words with a DMA into LDS; one period
        // later, when the counted wait has covered that DMA anyway, it looks at them and pauses (bounded: two
        // s_sleep at most, whatever the others do) if somebody is more than `sync_lead` stages behind (what it
        // sees is one to two periods old, so the lead must exceed that: with 8 the 4-wave shape pauses all the
        // time, 16 and 32 measure alike).
        constexpr bool GANG_SYNC = (AUX == 0) && (NQ >= 2) && (STG == 0) && (MODE == MODE_FILTER) && (ARTMIPS_GANG_SYNC != 0);
        // (period in stages: with two waves per SIMD the few instructions hide behind the partner's MFMAs; a lone
        // wave pays for them.  The request is covered by the counted wait `period` stages later as long as the
        // period is not shorter than the ring's look-ahead.)
        constexpr int SYNC_SHIFT = (NW == 8) ? 2 : 3;
        static_assert(!GANG_SYNC || (1 << SYNC_SHIFT) >= NSTAGE - (PHASED ? 3 : 2), "rendezvous period shorter than the look-ahead");
        constexpr uint32_t SYNC_LDS = (uint32_t)(NSTAGE * STAGE_BYTES);   // the staging area (unused: STG == 0)
        const bool sync_on = GANG_SYNC && w == 0 && a.gang > 1 && a.gang_progress != nullptr && !(ablate & 8192);
        const unsigned int* sync_words = nullptr;
        uint32_t sync_lead = 0;
        if constexpr (GANG_SYNC) {
            sync_words = a.gang_progress + (size_t)__builtin_amdgcn_readfirstlane((int)(b * 8));
            // (16 everywhere but the 8-wave gangs of 2: sixteen of those share an XCD's L2, which then holds
            // fewer stages per gang: profiles/r03_e_gang_sync.txt)
            sync_lead = a.sync_lead > 0 ? (uint32_t)a.sync_lead : (NW == 8 && a.gang <= 2 ? 8u : 16u);
        }

        int slot = 0;
#ifdef ARTMIPS_STAMPS_BUILD
        unsigned long long st[24];
        const bool stamp_wg = a.stamps != nullptr && blockIdx.x == 137u % gridDim.x;
#endif
        // The streaming loop, once per role: two loops of their own (a wave runs one of them) instead of
        // one loop with the role tested inside, which left hipcc short of registers (26 spilled, and every
        // reload of a spilled value waits for ALL the DMA pieces in flight).  Each iteration of either
        // loop passes exactly one barrier.
        auto stream = [&](auto role_tag) __attribute__((always_inline)) {
        constexpr bool RB = decltype(role_tag)::value == 1;
        // LDS addresses of this lane's four fragment columns in the current slot: carried from stage to
        // stage (four adds) instead of rebuilt from four loop-invariant offsets, which would hold four more
        // registers for the whole loop (the 8-wave shape has none to spare)
        uint32_t base[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) base[m] = lds_base + (uint32_t)rd_off[m];
        for (long long it = 0; it < n_it; ++it) {
#ifdef ARTMIPS_STAMPS_BUILD
            const bool stamp_now = stamp_wg && it == n_it / 2;
#else
            constexpr bool stamp_now = false;
            unsigned long long* const st = nullptr;
#endif
            ARTMIPS_STAMP(stamp_now, st, 0);
            [[maybe_unused]] V af[NBUF][CH];   // multi-group path: row fragments, NBUF chunks in registers
            if constexpr (RB) {
                if (!(ablate & 8)) mg_load_first<NBUF, CH, 0, V>(af, base);
            }
            // this wave's pieces of stage `it` (phased: `it + 1`) have landed; the barrier makes it true
            // for all waves
            constexpr int AHEAD = NSTAGE - (PHASED ? 3 : 2);   // stages that may still be on their way
            if (!(ablate & 512)) wait_vmcnt<AHEAD * PPW>();   // (512: refill issued, never waited for)
            ARTMIPS_STAMP(stamp_now, st, 1);
            if (!(ablate & 2)) asm volatile("s_barrier" ::: "memory");
            ARTMIPS_STAMP(stamp_now, st, 2);
            // every wave has finished reading the slot consumed in the previous iteration: refill it.
            // On the multi-group path a working wave spreads the refill over its compute phase
            // (mg_compute) when the tile is whole; otherwise all pieces go out here.
            MgDma dma;
            dma.on = false;
            dma.noread = (ablate & 8) != 0;
            dma.nowait = (ablate & 32) != 0;
            dma.stamp = stamp_now;
            dma.st = st;
            if constexpr (ASM_PATH && INLINE_DMA) {
                dma.on = wave_active && !(ablate & 1);
                // (an idle wave, or the diagnostic build without the refill, issues through issue_stage or
                // not at all: the walk advances exactly once per iteration either way)
                const long long row0 = dma.on ? walk_next(walk_issue) * TILE_ROWS : 0;
                dma.tile = (const char*)rows + row0 * ROWB;
                dma.lane_off = lane_off;
                dma.dst0 = lds_base + (uint32_t)(fill_slot * STAGE_BYTES + w * PPW * 1024);
                dma.late = (w >= NW / 2);
            }
            // (tried in round 3 and dropped, profiles/r03_a_gang_ab.txt: the burst BEHIND the stage's MFMAs
            // instead of in front of them, -0.7 % on one box and +2.9 % on another; and the first row
            // fragments of a stage requested before its wait and barrier, with one stage less of
            // look-ahead: +-0.5 % in the gang shapes)
            if constexpr (GANG_SYNC && !RB) {
                if (sync_on && (it & ((1 << SYNC_SHIFT) - 1)) == 0) {
                    if (it >= (1 << SYNC_SHIFT)) {
                        // (asm: through a generic pointer hipcc reads LDS with a flat load and waits vmcnt(0))
                        uint32_t v;
                        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                                     : "=v"(v)
                                     : "v"(lds_base + SYNC_LDS + (uint32_t)(((it >> SYNC_SHIFT) - 1) & 7) * 256u + (uint32_t)(lane & 7) * 4u)
                                     : "memory");
                        // a member counts while it runs in this launch and is not hopelessly far behind (one that
                        // has not started, or started late because the GPU is shared, is not waited for)
                        const uint32_t peer = v & 0xFFFFFu;
                        const bool counts = (lane & 7) < a.gang && (v >> 20) == a.gang_epoch && (uint32_t)it < peer + 256u;
                        if (__builtin_amdgcn_ballot_w64(counts && (uint32_t)it > peer + sync_lead) != 0ull) {
                            __builtin_amdgcn_s_sleep(127);
                            if (__builtin_amdgcn_ballot_w64(counts && (uint32_t)it > peer + 2 * sync_lead) != 0ull)
                                __builtin_amdgcn_s_sleep(127);
                        }
                    }
                    if (lane == 0)
                        __hip_atomic_store(const_cast<unsigned int*>(sync_words) + (blockIdx.x >> 3) % a.gang,
                                           (a.gang_epoch << 20) | (uint32_t)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    glds4_device_scope(sync_words, (uint32_t)(lane & 7) * 4u,
                                       (uint32_t)__builtin_amdgcn_readfirstlane(
                                           (int)(lds_base + SYNC_LDS + (uint32_t)((it >> SYNC_SHIFT) & 7) * 256u)));
                }
            }
            if (!dma.on && !RB && !(ablate & 1)) issue_stage();
            ARTMIPS_STAMP(stamp_now, st, 3);

            if (wave_active) {
                const char* sb = smem + slot * STAGE_BYTES;
                if constexpr (NQ == 1) {
                    f32x4 acc[NQ];
#pragma unroll
                    for (int j = 0; j < NQ; ++j) acc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    // the row fragments of a chunk are all requested before its first MFMA, so the LDS
                    // latency is paid once per chunk and not once per k-step
                    constexpr int CHB = (KSTEPS > 32) ? 12 : KSTEPS;
                    static_assert(KSTEPS % CHB == 0, "k-steps must split evenly");
#pragma unroll
                    for (int c0 = 0; c0 < KSTEPS; c0 += CHB) {
                        V af[CHB];
#pragma unroll
                        for (int i = 0; i < CHB; ++i) {
                            const int s = c0 + i;
                            af[i] = *(const V*)(sb + (s >> 2) * 256 + rd_off[s & 3]);
                        }
#pragma unroll
                        for (int i = 0; i < CHB; ++i) {
#pragma unroll
                            for (int j = 0; j < NQ; ++j) acc[j] = Elem<T>::mfma(af[i], bq[j][c0 + i], acc[j]);
                        }
                        // pin the order in the emitted code: the ds_read_b128s, then the MFMAs behind
                        // counted lgkmcnt waits (left alone, hipcc alternates read / wait(0) / MFMA)
                        __builtin_amdgcn_sched_group_barrier(0x100, CHB, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, CHB * NQ * MPF, 0);
                    }
                    epilogue(it, acc);
                } else {
                    // chunks of CH k-steps, row fragments double-buffered in registers; the filter of
                    // the PREVIOUS stage runs inside, behind the first LDS read requests (mg_compute)
                    auto pre = [&]() __attribute__((always_inline)) {
                        if (it > 0) {
                            mg_settle_short<NQ, 0>(acc);
                            epilogue(it - 1, acc);
                        }
                    };
                    if constexpr (RB) {
                        mg_compute<T, NQ, KSTEPS, VB, NCH, CH, NBUF, AUX, PPW, NW, 0, 1, V>(acc, af, bq, base, dma, pre);
                        ARTMIPS_STAMP(stamp_now, st, 4);
                        mg_settle<NQ, 0>(acc);
                        epilogue(it, acc);
                        ARTMIPS_STAMP(stamp_now, st, 5);
                    } else {
                        mg_compute<T, NQ, KSTEPS, VB, NCH, CH, NBUF, AUX, PPW, NW, 0, 0, V>(acc, af, bq, base, dma, pre);
                    }
                }
            }
            if constexpr (RB) {
                if (!(ablate & 1)) issue_stage();
            }
            fill_slot = (fill_slot + 1 == NSTAGE) ? 0 : fill_slot + 1;
            ARTMIPS_STAMP(stamp_now, st, 20);
#ifdef ARTMIPS_STAMPS_BUILD
            if (stamp_now) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) {
#pragma unroll
                    for (int i = 0; i < 24; ++i)
                        a.stamps[w * 24 + i] = (i <= 5 || i == 20 || (i >= 6 && i < 6 + (NCH > 12 ? 12 : NCH))) ? st[i] : 0ull;
                }
            }
#endif
            {
                const uint32_t step = (slot + 1 == NSTAGE) ? (uint32_t)(-(NSTAGE - 1) * STAGE_BYTES) : (uint32_t)STAGE_BYTES;
#pragma unroll
                for (int m = 0; m < 4; ++m) base[m] += step;
            }
            slot = (slot + 1 == NSTAGE) ? 0 : slot + 1;
        }
        if constexpr (NQ > 1 && !RB) {
            if (wave_active) {   // the last stage's filter (every other stage's ran inside the next one)
                mg_settle<NQ, 0>(acc);
                epilogue(n_it - 1, acc);
            }
        }
        };
        if constexpr (PHASED) {
            if (role_b) stream(std::integral_constant<int, 1>());
            else stream(std::integral_constant<int, 0>());
        } else {
            stream(std::integral_constant<int, 0>());
        }
        // the tail DMAs (clamped re-reads) must land before the workgroup's LDS is released
        wait_vmcnt<0>();
    }
    if (MODE == MODE_FILTER && STG) {
#pragma unroll
        for (int j = 0; j < NQ; ++j)
            if (cnt[j] > 0) flush(j, ((cnt[j] - 1) / SLOTS) * SLOTS, cnt[j]);
    }
    // counters / maxima are stored query-major ([query][workgroup*4 + lane group]) so that the
    // selection kernels read the L values of a query as one contiguous run
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const size_t o = (per_query_off + (size_t)(j * QG + w * 16 + r)) * (size_t)(grid * 4) + (size_t)(b * 4 + g);
        if (MODE == MODE_FILTER) {
            a.counts[o] = cnt[j];
        } else {  // [query][(workgroup*4 + lane group)*TOPT + t]
#pragma unroll
            for (int t = 0; t < TOPT; ++t) a.gmax[o * TOPT + t] = gm[j][t];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// selection primitives (one workgroup of SEL_THREADS threads)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int next_pow2(int x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

// bitonic sort, descending, of P (power of two) keys in LDS
template <typename K>
__device__ void bitonic_sort_desc(K* s, int P) {
    const int tid = threadIdx.x;
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < (P >> 1); i += blockDim.x) {
                const int pos = 2 * i - (i & (stride - 1));
                const int par = pos + stride;
                const bool desc = ((pos & size) == 0);
                const K x = s[pos], y = s[par];
                if ((x < y) == desc) { s[pos] = y; s[par] = x; }
            }
            __syncthreads();
        }
    }
}

// Radix-select helper, run by the first wave after the histogram is complete: walking the 256
// bins from the top, find the bin in which the running count reaches kk.  Four bins per lane and
// a wave-wide inclusive scan replace a 256-step serial walk by one thread.
// Writes sel_out[0] = bin, sel_out[1] = kk minus the count of all higher bins.
__device__ __forceinline__ void pick_bin_wave(const int* hist, int kk, int* sel_out) {
    const int lane = threadIdx.x & 63;
    // lane l owns bins 255-4l .. 252-4l (descending)
    int c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = hist[255 - 4 * lane - i];
    const int mine = c[0] + c[1] + c[2] + c[3];
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d, 64);
        if (lane >= d) incl += v;
    }
    const int excl = incl - mine;
    if (excl < kk && kk <= incl) {
        int cum = excl;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (cum < kk && kk <= cum + c[i]) { sel_out[0] = 255 - 4 * lane - i; sel_out[1] = kk - cum; }
            cum += c[i];
        }
    }
}

// k-th largest (1-based k <= n) of n keys, 8 bits per pass from the most significant byte.  Keys
// may live in LDS or in global memory.  hist: 256 ints of LDS; ctl: 4 ints of LDS.  All threads
// return the same value.
template <typename K>
__device__ K radix_kth(const K* keys, int n, int k, int* hist, int* ctl) {
    const int tid = threadIdx.x;
    K prefix = 0;
    int kk = k;
    for (int byte = (int)sizeof(K) - 1; byte >= 0; --byte) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const K mask = (byte == (int)sizeof(K) - 1) ? (K)0 : (K)(~(K)0 << ((byte + 1) * 8));
        for (int i = tid; i < n; i += blockDim.x) {
            const K key = keys[i];
            if ((key & mask) == prefix) atomicAdd(&hist[(int)((key >> (byte * 8)) & (K)255)], 1);
        }
        __syncthreads();
        if (tid < 64) pick_bin_wave(hist, kk, ctl);
        __syncthreads();
        prefix |= (K)ctl[0] << (byte * 8);
        kk = ctl[1];
        __syncthreads();
    }
    return prefix;
}

// exclusive prefix sum over the workgroup of one int per thread; wsum: 32 ints of LDS.
// Returns the exclusive offset of this thread; *total gets the sum over the workgroup.
__device__ __forceinline__ int block_excl_scan(int v, int* wsum, int* total) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int u = __shfl_up(incl, d, 64);
        if (lane >= d) incl += u;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    if (tid < 64) {
        const int nw = blockDim.x >> 6;
        int x = (tid < nw) ? wsum[tid] : 0;
        int xi = x;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int u = __shfl_up(xi, d, 64);
            if (lane >= d) xi += u;
        }
        if (tid < nw) wsum[tid] = xi - x;
        if (tid == nw - 1) wsum[31] = xi;
    }
    __syncthreads();
    const int off = wsum[wv] + incl - v;
    *total = wsum[31];
    __syncthreads();
    return off;
}

// Sort one 32-bit value per lane across the wave, descending, in registers (bitonic network over
// __shfl_xor: 21 exchange steps, no LDS, no barrier).  Lane i returns the (i+1)-th largest.
__device__ __forceinline__ uint32_t wave_sort_desc(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const uint32_t o = __shfl_xor(v, stride, 64);
            const bool desc = ((lane & size) == 0);      // direction of this lane's block
            const bool lower = ((lane & stride) == 0);   // lower lane of the pair
            v = (lower == desc) ? (v > o ? v : o) : (v < o ? v : o);
        }
    }
    return v;
}

// Small-k selection without a radix pass.  `mine` is the largest ordered score among the values this
// thread looked at (0 if it saw none); the values of different threads belong to different rows.  Each
// wave sorts its 64 thread maxima in registers and offers its r-th largest, r = ceil(k / waves); the
// minimum B of the offers is a lower bound of the k-th best score of ALL values: every wave has r
// threads whose maximum is >= its offer >= B, which makes waves * r >= k distinct rows at or above B.
// On random data about 1.3-1.5 k values survive `>= B`, few enough to rank by counting.  B == 0 (a
// wave with fewer than r threads that saw anything) keeps everything, which is still correct.
// wsum: 32 words of LDS.  Needs ceil(k / waves) <= 64.  All threads return the same B.
__device__ __forceinline__ uint32_t wave_offer_bound(uint32_t mine, int k, uint32_t* wsum) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nw = blockDim.x >> 6;
    const int r = (k + nw - 1) / nw;
    const uint32_t sorted = wave_sort_desc(mine);
    const uint32_t offer = __shfl(sorted, r - 1, 64);
    if (lane == 0) wsum[wv] = offer;
    __syncthreads();
    uint32_t b = 0xffffffffu;
    for (int w = 0; w < nw; ++w) b = wsum[w] < b ? wsum[w] : b;
    __syncthreads();
    return b;
}

constexpr int RANK_MAX = 512;   // survivors ranked by counting (RANK_MAX^2 / threads LDS reads per thread)

// ---------------------------------------------------------------------------------------------
// kth_kernel (pass 0): tau[q] = k-th largest of the per-lane sample values (the lane's maximum, or
// its four best) that belong to query q.  Each value is the score of a distinct row, so k of them >= tau proves that the k-th best
// score over the whole shard is >= tau: dropping every score < tau loses no top-k member.
// One block per query slot (64 blocks); dynamic LDS = next_pow2(n_groups) * 4 bytes.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SEL_THREADS) void kth_kernel(const float* __restrict__ gmax,
                                                          int n_values, int nq, int k, int q_count,
                                                          float* __restrict__ thr,
                                                          int* __restrict__ status_to_clear) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* vals = (uint32_t*)smem;
    __shared__ int hist[256];
    __shared__ int ctl[4];
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const int G = n_values;  // pass-0 workgroups * 4 lane groups * values kept per lane
    // first query tile of a search: clear the 4 status words (flags, pad, 64-bit candidate count) the
    // select raises; they may be the status tail of a packed candidate list, which is 4 words long
    if (status_to_clear != nullptr && q == 0 && tid < 4) status_to_clear[tid] = 0;
    if (q >= q_count) {
        if (tid == 0) thr[q] = INFINITY;  // padded query: never keep anything
        return;
    }
    if (G < k) {
        if (tid == 0) thr[q] = -INFINITY;
        return;
    }
    uint32_t mine = 0;
    for (int i = tid; i < G; i += blockDim.x) {
        const uint32_t v = f2ord(gmax[(size_t)q * G + i]);
        vals[i] = v;
        mine = v > mine ? v : mine;
    }
    __syncthreads();
    if (k <= 128 && (k + (int)(blockDim.x >> 6) - 1) / (int)(blockDim.x >> 6) <= 64) {
        // small k: bound from the waves' sorted thread maxima, then the exact k-th largest of the few
        // values at or above it, by counting (no radix passes)
        __shared__ uint32_t surv[RANK_MAX];
        __shared__ uint32_t wofs[32];
        const uint32_t bound = wave_offer_bound(mine, k, wofs);
        if (tid == 0) ctl[2] = 0;
        __syncthreads();
        for (int i = tid; i < G; i += blockDim.x) {
            const uint32_t v = vals[i];
            if (v >= bound) {
                const int pos = atomicAdd(&ctl[2], 1);
                if (pos < RANK_MAX) surv[pos] = v;
            }
        }
        __syncthreads();
        const int got = ctl[2];   // >= k (uniform)
        if (got <= RANK_MAX) {
            for (int i = tid; i < got; i += blockDim.x) {
                const uint32_t v = surv[i];
                int gt = 0, ge = 0;
                for (int j = 0; j < got; ++j) {
                    const uint32_t u = surv[j];
                    gt += (u > v);
                    ge += (u >= v);
                }
                if (gt < k && k <= ge) thr[q] = ord2f(v);   // equal values may write: the same number
            }
            return;
        }
        __syncthreads();
    }
    const uint32_t kth = radix_kth<uint32_t>(vals, G, k, hist, ctl);
    if (tid == 0) thr[q] = ord2f(kth);
}

// thresholds when pass 0 is skipped (small shards): keep everything for real queries
__global__ void fill_thr_kernel(float* thr, int q_count, int q_tile, int* status_to_clear) {
    if (status_to_clear != nullptr && threadIdx.x < 4) status_to_clear[threadIdx.x] = 0;
    for (int q = threadIdx.x; q < q_tile; q += blockDim.x) thr[q] = (q < q_count) ? -INFINITY : INFINITY;
}

// ---------------------------------------------------------------------------------------------
// exact top-k of a set of distinct 64-bit keys
// ---------------------------------------------------------------------------------------------
constexpr int SUB_MAX = 2048;   // keys of the second-level sample / survivor buffer (16 KiB)

struct TopkOut {
    void* scores;     // [*, k] score_dtype
    int score_dtype;
    int32_t* ids;     // [*, k]
    long long row;    // output row (query index)
    int k;
};

struct TopkScratch {
    u64* skeys;   // SORT_MAX keys of LDS
    u64* sub;     // SUB_MAX keys of LDS
    int* hist;    // 256 ints
    int* ctl;     // 4 ints
    int stop;     // diagnostic build: return after phase `stop` (timing ablation); 0 = run to the end
};
// phase markers of the selection kernels: in the diagnostic build a kernel can be told to return after
// phase n (results are then meaningless), which prices the phases by difference
#ifdef ARTMIPS_DIAG
#define SEL_PHASE(stop, n) do { if ((stop) == (n)) return; } while (0)
#else
#define SEL_PHASE(stop, n) do { } while (0)
#endif

__device__ __forceinline__ void write_topk(const u64* sorted, int n_sorted, const TopkOut& o) {
    for (int i = threadIdx.x; i < o.k; i += blockDim.x) {
        float s = -INFINITY;
        int32_t id = 0x7fffffff;
        if (i < n_sorted) { const u64 key = sorted[i]; s = key_score(key); id = key_id(key); }
        store_score(o.scores, o.score_dtype, o.row * (long long)o.k + i, s);
        o.ids[o.row * (long long)o.k + i] = id;
    }
}

// A "pre-key" is ordered(score) << 32 | local row: enough to filter on the score.  Turning it into
// the final key needs the passage id of the row (a dependent HBM load), so that is done only for
// the keys that survive the score filter.  row_ids == nullptr means the keys are final already.
__device__ __forceinline__ u64 finish_key(u64 pre, const int32_t* row_ids) {
    if (row_ids == nullptr) return pre;
    const int32_t id = row_ids[(uint32_t)pre];
    return (pre & 0xffffffff00000000ull) | (u64)(uint32_t)(~(((uint32_t)id) ^ 0x80000000u));
}

// The M (pre-)keys are in keys[0..M): in LDS (== sc.skeys) when in_lds, else in global memory.
// Writes the k best in order; output slots beyond M get (-inf, INT32_MAX).
//
// Three routes, all exact:
//   two-level  (small k, many keys): radix-select the k-th largest SCORE of the first ns keys (a
//              second lower bound of the k-th best score), keep the few keys at or above it,
//              finish and sort those;
//   full sort  (M <= SORT_MAX keys in LDS);
//   radix      (keys in global memory): radix-select the k-th key, gather the k keys above it, sort.
__device__ void topk_finish(u64* keys, int M, bool in_lds, const int32_t* row_ids,
                            const TopkScratch& sc, const TopkOut& o) {
    const int tid = threadIdx.x;
    const int k = o.k;
    if (M <= 0) { write_topk(sc.sub, 0, o); return; }
    // ---- wave-offer route (small k): a score bound from the waves' sorted thread maxima
    // (wave_offer_bound), the keys at or above it finished and ranked by counting: one pass over the
    // keys, no radix pass, no sorting network in LDS.
    if (k <= 128 && (k + (int)(blockDim.x >> 6) - 1) / (int)(blockDim.x >> 6) <= 64) {
        uint32_t mine = 0;
        for (int i = tid; i < M; i += blockDim.x) {
            const uint32_t v = (uint32_t)(keys[i] >> 32);
            mine = v > mine ? v : mine;
        }
        const uint32_t bound = wave_offer_bound(mine, k, (uint32_t*)sc.hist);
        if (tid == 0) sc.ctl[2] = 0;
        __syncthreads();
        for (int i = tid; i < M; i += blockDim.x) {
            const u64 key = keys[i];
            if ((uint32_t)(key >> 32) >= bound) {
                const int pos = atomicAdd(&sc.ctl[2], 1);
                if (pos < RANK_MAX) sc.sub[pos] = finish_key(key, row_ids);
            }
        }
        __syncthreads();
        const int got = sc.ctl[2];
        __syncthreads();
        SEL_PHASE(sc.stop, 4);
        if (got <= RANK_MAX) {
            // the number of keys in front of a key is its output position.  Keys of a well-formed index
            // are distinct (unique passage ids); equal keys (a caller's duplicate ids, or a merge fed the
            // same list twice) are ordered by survivor slot, so that the ranks stay a permutation and no
            // output slot is left unwritten
            for (int i = tid; i < got; i += blockDim.x) {
                const u64 key = sc.sub[i];
                int rank = 0;
                for (int j = 0; j < got; ++j) {
                    const u64 other = sc.sub[j];
                    rank += (other > key) || (other == key && j < i);
                }
                if (rank < k) {
                    store_score(o.scores, o.score_dtype, o.row * (long long)k + rank, key_score(key));
                    o.ids[o.row * (long long)k + rank] = key_id(key);
                }
            }
            for (int i = got + tid; i < k; i += blockDim.x) {   // fewer than k rows in all: pads
                store_score(o.scores, o.score_dtype, o.row * (long long)k + i, -INFINITY);
                o.ids[o.row * (long long)k + i] = 0x7fffffff;
            }
            return;
        }
        // massive ties or a degenerate bound: the routes below sort it out
    }
    // ---- two-level route
    int ns = next_pow2(4 * k);
    if (ns < 512) ns = 512;
    if (ns <= SUB_MAX && M >= 2 * ns) {
        SEL_PHASE(sc.stop, 6);
        uint32_t* sample = (uint32_t*)sc.sub;
        for (int i = tid; i < ns; i += blockDim.x) sample[i] = (uint32_t)(keys[i] >> 32);
        __syncthreads();
        // k sampled rows score >= bound, so the true k-th best score does too
        const uint32_t bound = radix_kth<uint32_t>(sample, ns, k, sc.hist, sc.ctl);
        SEL_PHASE(sc.stop, 3);
        if (tid == 0) sc.ctl[2] = 0;
        __syncthreads();
        for (int i = tid; i < M; i += blockDim.x) {
            const u64 key = keys[i];
            if ((uint32_t)(key >> 32) >= bound) {
                const int pos = atomicAdd(&sc.ctl[2], 1);
                if (pos < SUB_MAX) sc.sub[pos] = finish_key(key, row_ids);
            }
        }
        __syncthreads();
        const int got = sc.ctl[2];
        __syncthreads();
        SEL_PHASE(sc.stop, 4);
        if (got <= SUB_MAX) {
            const int P = next_pow2(got < 2 ? 2 : got);
            for (int i = got + tid; i < P; i += blockDim.x) sc.sub[i] = 0ull;
            __syncthreads();
            bitonic_sort_desc<u64>(sc.sub, P);
            SEL_PHASE(sc.stop, 5);
            write_topk(sc.sub, got, o);
            return;
        }
        // too many survivors (skewed sample or massive score ties): unconditional routes below
    }
    SEL_PHASE(sc.stop, 7);
    if (row_ids != nullptr) {
        for (int i = tid; i < M; i += blockDim.x) keys[i] = finish_key(keys[i], row_ids);
        __syncthreads();
    }
    if (in_lds) {
        const int P = next_pow2(M < 2 ? 2 : M);
        for (int i = M + tid; i < P; i += blockDim.x) sc.skeys[i] = 0ull;
        __syncthreads();
        bitonic_sort_desc<u64>(sc.skeys, P);
        write_topk(sc.skeys, M, o);
        return;
    }
    // ---- radix route over global keys (k <= SORT_MAX)
    const int kk = k < M ? k : M;
    const u64 kth = radix_kth<u64>(keys, M, kk, sc.hist, sc.ctl);
    if (tid == 0) sc.ctl[2] = 0;
    __syncthreads();
    for (int i = tid; i < M; i += blockDim.x) {
        const u64 key = keys[i];
        if (key >= kth) {
            const int pos = atomicAdd(&sc.ctl[2], 1);
            if (pos < SORT_MAX) sc.skeys[pos] = key;
        }
    }
    __syncthreads();
    int got = sc.ctl[2];  // == kk because keys are distinct
    if (got > SORT_MAX) got = SORT_MAX;
    const int P = next_pow2(got < 2 ? 2 : got);
    for (int i = got + tid; i < P; i += blockDim.x) sc.skeys[i] = 0ull;
    __syncthreads();
    bitonic_sort_desc<u64>(sc.skeys, P);
    write_topk(sc.skeys, got, o);
}

// ---------------------------------------------------------------------------------------------
// select_kernel: one block per query of the tile.
// ---------------------------------------------------------------------------------------------
constexpr int MAX_LISTS = 4096;  // candidate lists per query = scan grid * 4 (grid <= 1024)

struct SelectArgs {
    const int2* lists;
    const int* counts;
    int cap;
    int scan_grid;          // blocks of the FILTER scan (<= 1024)
    int nq;                 // column groups of the FILTER scan
    int threads;            // threads per workgroup of the FILTER scan (256 or 512)
    int q_member;           // gang mode: queries per gang member (0 = no gang); query q of the launch
                            // then belongs to member q / q_member, whose lists start
                            // lists_member_stride entries further on
    long long lists_member_stride;
    float* refine_thr;      // non-NULL: refine mode.  Nothing is selected or written out; instead
                            // refine_thr[q] is raised to the k-th best score among the candidates kept so
                            // far (scores of k distinct rows: still a lower bound of the final k-th best)
    const int32_t* ids;     // [n_rows] passage id of each local row
    u64* keys_ws;           // [q_tile][keys_stride] global key scratch for M > SORT_MAX
    long long keys_stride;
    int k;
    void* out_scores;
    int score_dtype;
    int32_t* out_ids;
    long long out_row0;     // first output row of this query tile
    int* status;            // bit0: a candidate list overflowed its cap; bit1: key scratch too small
    unsigned long long* cand_total;
    int stop_after;         // diagnostic build only (SEL_PHASE)
    // Optional: the last workgroup of the launch to finish copies the 4 status words to `publish`
    // (pinned host memory as the device sees it), which saves the search a copy command behind the
    // pipeline.  done_counter is a device word that is zero before the launch and zero again after it.
    int* publish;
    int* done_counter;
};

// 32-bit scores radix-selected in LDS by select_kernel's "score route" (112 KiB, shared with skeys)
constexpr int SCORE_MAX = 28672;

__device__ __forceinline__ void select_body(const SelectArgs& a) {
    __shared__ __attribute__((aligned(16))) uint32_t big[SCORE_MAX];
    static_assert(SCORE_MAX * 4 >= SORT_MAX * 8, "skeys lives in the same LDS as the scores");
    u64* const skeys = (u64*)big;
    __shared__ __attribute__((aligned(16))) u64 sub[SUB_MAX];
    __shared__ int loff[MAX_LISTS + 1];
    __shared__ int hist[256];
    __shared__ int ctl[4];
    __shared__ int wsum[32];
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const int L = a.scan_grid * 4;  // candidate lists of this query
    const int member = a.q_member > 0 ? q / a.q_member : 0;
    const int ql = a.q_member > 0 ? q % a.q_member : q;   // index of the query inside its member's block
    const int2* const lists = a.lists + (size_t)member * (size_t)a.lists_member_stride;
    auto list_of = [&](int li) -> size_t {
        const int qg = a.threads >> 2;  // queries per column group
        return ((size_t)(li >> 2) * a.nq + ql / qg) * a.threads + ((ql % qg) >> 4) * 64 + (li & 3) * 16 + (ql & 15);
    };
    // ---- exclusive offsets of the lists: thread t owns lists [t*per, t*per+per)
    const int nt = blockDim.x;
    const int per = (L + nt - 1) / nt;
    int mine = 0, ovf = 0;
    for (int i = 0; i < per; ++i) {
        const int li = tid * per + i;
        if (li < L) {
            int v = a.counts[(size_t)q * L + li];
            if (v > a.cap) { ovf = 1; v = a.cap; }
            mine += v;
        }
    }
    int M;
    int off = block_excl_scan(mine, wsum, &M);
    for (int i = 0; i < per; ++i) {
        const int li = tid * per + i;
        if (li < L) {
            int v = a.counts[(size_t)q * L + li];
            if (v > a.cap) v = a.cap;
            loff[li] = off;
            off += v;
        }
    }
    if (tid == 0) loff[L] = M;
    if (a.refine_thr != nullptr) {
        // refine mode: M and every branch below are uniform over the workgroup
        if (M < a.k || M > SCORE_MAX) return;   // too few rows seen for a bound / too many for LDS: keep thr
        __syncthreads();
        const int cpt_r = (M + nt - 1) / nt;
        const int eb = tid * cpt_r, ee = (eb + cpt_r < M) ? eb + cpt_r : M;
        if (eb < ee) {
            int lo = 0, hi = L;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (loff[mid] <= eb) lo = mid; else hi = mid;
            }
            for (int e = eb; e < ee; ++e) {
                while (loff[lo + 1] <= e) ++lo;
                big[e] = f2ord(__int_as_float(lists[list_of(lo) * (size_t)a.cap + (e - loff[lo])].x));
            }
        }
        __syncthreads();
        const uint32_t kth = radix_kth<uint32_t>(big, M, a.k, hist, ctl);
        if (tid == 0) a.refine_thr[q] = fmaxf(a.refine_thr[q], ord2f(kth));
        return;
    }
    SEL_PHASE(a.stop_after, 1);
    if (ovf) atomicOr(a.status, 1);
    if (tid == 0) atomicAdd(a.cand_total, (unsigned long long)M);
    __syncthreads();
    const bool in_lds = (M <= SORT_MAX);
    u64* gk = a.keys_ws + (size_t)q * a.keys_stride;
    // list holding candidate e (binary search over the offsets): loff[lo] <= e < loff[lo + 1]
    auto find_list = [&](int e) -> int {
        int lo = 0, hi = L;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (loff[mid] <= e) lo = mid; else hi = mid;
        }
        return lo;
    };
    // ---- balanced walk: thread t takes candidates [t*cpt, t*cpt+cpt), one binary search and then a
    // linear walk over the lists; all loads are independent
    const int cpt = (M + nt - 1) / nt;
    const int e_begin = tid * cpt;
    const int e_end = (e_begin + cpt < M) ? e_begin + cpt : M;
    TopkScratch sc{skeys, sub, hist, ctl, a.stop_after};
    TopkOut o{a.out_scores, a.score_dtype, a.out_ids, a.out_row0 + q, a.k};

    // ---- score route (more candidates than the LDS sort holds, e.g. k = 1000): only the 32-bit
    // scores go to LDS, the exact k-th best score is radix-selected there, and only the candidates at
    // or above it (k of them, plus ties on that score) are fetched again, given their passage id and
    // sorted.  Nothing is written to global memory.
    if (!in_lds && M <= SCORE_MAX && a.k <= SUB_MAX) {
        if (e_begin < e_end) {
            int lo = find_list(e_begin);
            for (int e = e_begin; e < e_end; ++e) {
                while (loff[lo + 1] <= e) ++lo;
                big[e] = f2ord(__int_as_float(lists[list_of(lo) * (size_t)a.cap + (e - loff[lo])].x));
            }
        }
        __syncthreads();
        const uint32_t kth = radix_kth<uint32_t>(big, M, a.k, hist, ctl);   // M > SORT_MAX >= k
        if (tid == 0) ctl[2] = 0;
        __syncthreads();
        for (int e = tid; e < M; e += nt) {
            if (big[e] >= kth) {
                const int pos = atomicAdd(&ctl[2], 1);
                if (pos < SUB_MAX) sub[pos] = (u64)e;
            }
        }
        __syncthreads();
        const int got = ctl[2];   // >= k; larger only when several rows score exactly the k-th best
        if (got <= SUB_MAX) {
            for (int i = tid; i < got; i += nt) {
                const int e = (int)sub[i];
                const int lo = find_list(e);
                const int2 ent = lists[list_of(lo) * (size_t)a.cap + (e - loff[lo])];
                sub[i] = make_key(__int_as_float(ent.x), a.ids[ent.y]);
            }
            const int P = next_pow2(got < 2 ? 2 : got);
            for (int i = got + tid; i < P; i += nt) sub[i] = 0ull;
            __syncthreads();
            bitonic_sort_desc<u64>(sub, P);
            write_topk(sub, got, o);
            return;
        }
        __syncthreads();   // massive ties on the k-th score: the general route below sorts them out
    }

    // ---- general routes: the keys written here are pre-keys (score, row); passage ids are looked
    // up only for the survivors
    if (!in_lds && (long long)M > a.keys_stride) {
        // cannot happen when the host sized the scratch as lists*cap (the allocation-free entry
        // point hands over what its static array has left): report instead of faulting
        if (tid == 0) atomicOr(a.status, 2);
        return;
    }
    u64* dst = in_lds ? skeys : gk;
    if (e_begin < e_end) {
        int lo = find_list(e_begin);
        for (int e = e_begin; e < e_end; ++e) {
            while (loff[lo + 1] <= e) ++lo;
            const int2 ent = lists[list_of(lo) * (size_t)a.cap + (e - loff[lo])];
            dst[e] = ((u64)f2ord(__int_as_float(ent.x)) << 32) | (u64)(uint32_t)ent.y;
        }
    }
    __syncthreads();
    SEL_PHASE(a.stop_after, 2);
    topk_finish(dst, M, in_lds, a.ids, sc, o);
}

__global__ __launch_bounds__(SEL_THREADS) void select_kernel(SelectArgs a) {
    select_body(a);
    if (a.publish != nullptr) {
        // every thread of this workgroup has raised what it had to raise in the status words
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            const int done = atomicAdd(a.done_counter, 1);
            if (done == (int)gridDim.x - 1) {   // all other workgroups' status updates are visible
                __threadfence();
                const int flags = atomicOr(a.status, 0);
                const unsigned long long cand = atomicAdd(a.cand_total, 0ull);
                volatile int* out = a.publish;
                out[0] = flags;
                out[1] = 0;
                out[2] = (int)(uint32_t)cand;
                out[3] = (int)(uint32_t)(cand >> 32);
                atomicExch(a.done_counter, 0);
                __threadfence_system();
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// merge_kernel: one block per query; union of n_lists sorted lists of k (score, id) each.
// Input layout [n_lists, n_q, k].  Pads (id == INT32_MAX) are skipped.
// ---------------------------------------------------------------------------------------------
struct MergeArgs {
    const float* scores;
    const int32_t* ids;
    int n_lists;
    long long n_q;
    int k;
    u64* keys_ws;           // [n_q][keys_stride] when n_lists*k > SORT_MAX
    long long keys_stride;
    void* out_scores;
    int score_dtype;
    int32_t* out_ids;
    int* status;
    long long list_stride;  // elements between the [n_q][k] planes of consecutive lists
    const int* status_in;   // optional: one status word per list (status_in[l * status_stride])
    long long status_stride;
    int* flag_out;          // optional: OR of the lists' status words
};

__global__ __launch_bounds__(SEL_THREADS) void merge_kernel(MergeArgs a) {
    __shared__ __attribute__((aligned(16))) u64 skeys[SORT_MAX];
    __shared__ __attribute__((aligned(16))) u64 sub[SUB_MAX];
    __shared__ int hist[256];
    __shared__ int ctl[4];
    __shared__ int s_fill;
    const long long q = blockIdx.x;
    const int tid = threadIdx.x;
    const int total = a.n_lists * a.k;
    const bool in_lds = (total <= SORT_MAX);
    if (a.flag_out && q == 0 && tid == 0) {
        int f = 0;
        for (int l = 0; l < a.n_lists; ++l) f |= a.status_in[(size_t)l * a.status_stride];
        a.flag_out[0] = f;
    }
    u64* gk = a.keys_ws ? a.keys_ws + (size_t)q * a.keys_stride : nullptr;
    if (!in_lds && (gk == nullptr || (long long)total > a.keys_stride)) {
        if (tid == 0) atomicOr(a.status, 2);
        return;
    }
    if (tid == 0) s_fill = 0;
    __syncthreads();
    u64* dst = in_lds ? skeys : gk;
    for (int i = tid; i < total; i += blockDim.x) {
        const int l = i / a.k, j = i - l * a.k;
        const size_t src = (size_t)l * a.list_stride + (size_t)q * a.k + j;
        const int32_t id = a.ids[src];
        if (id == 0x7fffffff) continue;
        const u64 key = make_key(a.scores[src], id);
        dst[atomicAdd(&s_fill, 1)] = key;
    }
    __syncthreads();
    const int M = s_fill;
    TopkScratch sc{skeys, sub, hist, ctl, 0};
    TopkOut o{a.out_scores, a.score_dtype, a.out_ids, q, a.k};
    topk_finish(dst, M, in_lds, nullptr, sc, o);
}

// ---------------------------------------------------------------------------------------------
// index-refresh upload kernels
// ---------------------------------------------------------------------------------------------
constexpr int32_t ID_SENTINEL = (int32_t)0x80808080;  // "row not uploaded yet" (memset byte 0x80)

// Scatter a batch of freshly encoded rows into a shard: row i goes to local position
// ids[i] - id_base - shard_row0 if that falls inside the shard, converted to the storage dtype.
// meta[0] counts rows written, meta[1] counts duplicates (a position written twice).
// One 256-thread workgroup per batch row (grid-strided).
template <typename T>
__global__ __launch_bounds__(256) void scatter_rows_kernel(const void* __restrict__ src, int src_dtype,
                                                           const long long* __restrict__ ids, long long n,
                                                           long long id_base, long long shard_row0,
                                                           long long shard_rows, int dim,
                                                           T* __restrict__ rows, int32_t* __restrict__ row_ids,
                                                           unsigned long long* __restrict__ meta) {
    unsigned long long fresh = 0, dups = 0;   // thread 0: rows this workgroup wrote / wrote over
    for (long long i = blockIdx.x; i < n; i += gridDim.x) {
        const long long id = ids[i];
        const long long pos = id - id_base - shard_row0;
        if (pos < 0 || pos >= shard_rows) continue;  // another shard owns this row (uniform per block)
        T* dst = rows + pos * dim;
        for (int c = threadIdx.x; c < dim; c += blockDim.x) {
            float v;
            if (src_dtype == 0) v = (float)((const _Float16*)src)[i * dim + c];
            else if (src_dtype == 1) v = (float)((const __bf16*)src)[i * dim + c];
            else v = ((const float*)src)[i * dim + c];
            dst[c] = Elem<T>::from_float(v);
        }
        if (threadIdx.x == 0) {
            const int32_t old = atomicExch(&row_ids[pos], (int32_t)id);
            if (old != ID_SENTINEL) ++dups; else ++fresh;
        }
    }
    // one add per workgroup: a counter every row adds to is one contended word (262,144 rows took
    // 3 ms for that alone)
    if (threadIdx.x == 0) {
        if (fresh) atomicAdd(&meta[0], fresh);
        if (dups) atomicAdd(&meta[1], dups);
    }
}

// The same scatter with 16-byte accesses: a workgroup takes rows_per_wg batch rows at a time and its
// threads walk the (row, 16-byte output chunk) pairs of that block, so that every lane loads 16 or 32
// source bytes and stores 16 (HBM-bound byte work: coalesced wide accesses are all there is to it).
// Needs 16-byte aligned source and destination rows (dim * element size a multiple of 16 on both
// sides and an aligned source base); the host falls back to scatter_rows_kernel otherwise.
template <typename T, typename S>
__global__ __launch_bounds__(256) void scatter_rows_vec_kernel(const S* __restrict__ src,
                                                               const long long* __restrict__ ids, long long n,
                                                               long long id_base, long long shard_row0,
                                                               long long shard_rows, int dim,
                                                               T* __restrict__ rows, int32_t* __restrict__ row_ids,
                                                               unsigned long long* __restrict__ meta,
                                                               int rows_per_wg) {
    constexpr int EPC = 16 / (int)sizeof(T);      // elements per 16-byte output chunk
    const int cpr = dim / EPC;                    // chunks per row
    const int items = rows_per_wg * cpr;
    unsigned long long fresh = 0, dups = 0;   // rows this lane wrote / wrote over
    typedef T TV __attribute__((ext_vector_type(EPC)));
    // U chunks per thread and step: their ids and source chunks are all requested before the first is
    // used (the kernel is a chain of dependent latencies otherwise: id -> position -> store, 3 us a link
    // with nothing else in flight)
    constexpr int U = 4;
    for (long long r0 = (long long)blockIdx.x * rows_per_wg; r0 < n; r0 += (long long)gridDim.x * rows_per_wg) {
        for (int w0 = threadIdx.x; w0 < items; w0 += blockDim.x * U) {
            long long id[U];
            int col[U];
            bool live[U];
            TV o[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int w = w0 + u * (int)blockDim.x;
                const int lr = w / cpr;
                col[u] = w - lr * cpr;
                const long long i = r0 + lr;
                live[u] = (w < items) && (i < n);
                id[u] = live[u] ? ids[i] : 0;
                const S* sp = src + (live[u] ? i : 0) * dim + col[u] * EPC;
                if constexpr (sizeof(S) == 4) {
#pragma unroll
                    for (int q = 0; q < EPC / 4; ++q) {
                        const float4 x = ((const float4*)sp)[q];
                        o[u][4 * q] = Elem<T>::from_float(x.x);
                        o[u][4 * q + 1] = Elem<T>::from_float(x.y);
                        o[u][4 * q + 2] = Elem<T>::from_float(x.z);
                        o[u][4 * q + 3] = Elem<T>::from_float(x.w);
                    }
                } else {
                    typedef S SV __attribute__((ext_vector_type(EPC)));
                    const SV x = *(const SV*)sp;
#pragma unroll
                    for (int e = 0; e < EPC; ++e) o[u][e] = Elem<T>::from_float((float)x[e]);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long pos = id[u] - id_base - shard_row0;
                if (!live[u] || pos < 0 || pos >= shard_rows) continue;   // (another shard owns this row)
                // (non-temporal: the rows are not read again before the next search streams the whole shard)
                __builtin_nontemporal_store(o[u], (TV*)(rows + pos * dim + col[u] * EPC));
                if (col[u] == 0) {
                    const int32_t old = atomicExch(&row_ids[pos], (int32_t)id[u]);
                    if (old != ID_SENTINEL) ++dups; else ++fresh;
                }
            }
        }
    }
    // one add per WORKGROUP on the shared words: atomics on one address serialise in L2 at ~10 ns each,
    // which made 32k per-wave adds (0.35 ms) the whole cost of a 262,144-row batch
    __shared__ unsigned long long wg_sum[2][4];
    for (int d = 32; d > 0; d >>= 1) {
        fresh += __shfl_down(fresh, d, 64);
        dups += __shfl_down(dups, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        wg_sum[0][threadIdx.x >> 6] = fresh;
        wg_sum[1][threadIdx.x >> 6] = dups;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long f = wg_sum[0][0] + wg_sum[0][1] + wg_sum[0][2] + wg_sum[0][3];
        const unsigned long long d2 = wg_sum[1][0] + wg_sum[1][1] + wg_sum[1][2] + wg_sum[1][3];
        if (f) atomicAdd(&meta[0], f);
        if (d2) atomicAdd(&meta[1], d2);
    }
}

// how many rows of a shard were never uploaded
__global__ void count_unfilled_kernel(const int32_t* __restrict__ row_ids, long long n,
                                      unsigned long long* __restrict__ out) {
    unsigned long long mine = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        mine += (row_ids[i] == ID_SENTINEL) ? 1ull : 0ull;
    for (int d = 32; d > 0; d >>= 1) mine += __shfl_down(mine, d, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(out, mine);
}

// Fetch rows by local position as fp32 (the "reconstruct" half of the faiss-shaped sibling API,
// art_index.py:190-193).  One 256-thread workgroup per fetched row (grid-strided).
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ rows, int dim,
                                                          const long long* __restrict__ pos, long long n,
                                                          long long n_rows, float* __restrict__ out) {
    for (long long i = blockIdx.x; i < n; i += gridDim.x) {
        const long long p = pos[i];
        for (int c = threadIdx.x; c < dim; c += blockDim.x)
            out[i * dim + c] = (p >= 0 && p < n_rows) ? (float)rows[p * dim + c] : 0.0f;
    }
}

}  // namespace artk
