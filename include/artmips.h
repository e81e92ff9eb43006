/*
 * artmips.h — C ABI of libartmips.so: MI355X-native exact maximum-inner-product
 * top-k search for ART's evidence index.
 *
 * This is the drop-in boundary for the path that the reference implements in
 * megatron/data/art_index.py (DistributedBruteForceIndex + the store upload that
 * feeds it).  Every entry point cites the reference lines it replaces.  The
 * reference is Python, so "the reference's FFI" is a ctypes binding; the stub a
 * maintainer would add is shown in INTEGRATION.md and shipped as
 * art_amd/_lib.py.
 *
 * Conventions
 *   - return 0 on success, a negative ARTMIPS_ERR_* on failure; the message of
 *     the last failure on the calling thread is artmips_last_error().
 *   - no exceptions cross the ABI, no torch types: plain pointers and sizes.
 *   - device pointers are raw HIP device addresses (torch: tensor.data_ptr()).
 *   - `stream` is a hipStream_t passed as void* (torch:
 *     torch.cuda.current_stream().cuda_stream); NULL is the default stream.
 *   - all device work is enqueued on / ordered against `stream`; artmips_search and
 *     artmips_search_local synchronise that stream before they return (they must read a
 *     device status word to guarantee exactness); the *_async forms,
 *     artmips_search_local_noalloc, artmips_merge and artmips_merge_packed only enqueue and
 *     hand the status word to the caller instead.
 *   - rows are `dim` elements of the storage dtype, row-major, contiguous.
 *   - passage ids are int32 (the reference returns int32: art_index.py:297) and must be UNIQUE
 *     over the index, as the keys of the reference's dict are (art_index.py:56-61).  The scatter
 *     upload detects a repeated id (artmips_finalize fails); artmips_upload / artmips_upload_device
 *     do not look for them.  A repeated id does not break a search: rows that share an id and a
 *     score are both returned, in unspecified relative order, and every output slot is written.
 *   - a handle is used by one thread at a time (the reference drives its index from one
 *     Python thread of one process per node: art_model.py:327,351), with one exception:
 *     artmips_download / artmips_generation may run on a second thread.  Different handles are
 *     independent, and the handle-less merges serialise per process on an internal lock.
 *   - top-k order: score descending, ties broken by LOWER passage id
 *     (BASELINE.json north_star); scores are fp32-accumulated dot products.
 */
#ifndef ARTMIPS_H
#define ARTMIPS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct artmips artmips_t;

/* element types (storage_dtype, q_dtype, score_dtype) */
enum {
    ARTMIPS_F16 = 0, /* IEEE half  — the reference's storage type (art_index.py:61,247) */
    ARTMIPS_BF16 = 1,
    ARTMIPS_F32 = 2  /* storage too (dim 128/768; exact-f32 MFMA, BASELINE config 5's check) */
};

/* error codes */
enum {
    ARTMIPS_OK = 0,
    ARTMIPS_ERR_ARG = -1,         /* bad argument (shape, dtype, k, NULL)           */
    ARTMIPS_ERR_HIP = -2,         /* a HIP runtime call failed                       */
    ARTMIPS_ERR_STATE = -3,       /* call sequence error (no index, upload w/o begin) */
    ARTMIPS_ERR_UNSUPPORTED = -4, /* dim / dtype / k outside what the kernels cover  */
    ARTMIPS_ERR_NOMEM = -5
};

#define ARTMIPS_MAX_DEVICES 16
#define ARTMIPS_MAX_K 2048 /* for indexes larger than ARTMIPS_SORT_MAX rows */
#define ARTMIPS_PACK_TAIL 4 /* int32 status words behind a packed candidate list */
/* bytes of the static device array behind artmips_search_local_noalloc (held by every GPU the
 * library runs on until the process ends) */
#define ARTMIPS_NOALLOC_SCRATCH_BYTES (64 << 20)

typedef struct artmips_info {
    int32_t n_dev;                              /* shards (one per device slot)           */
    int32_t dim;
    int32_t storage_dtype;
    int32_t n_cu;                               /* compute units of device slot 0         */
    int64_t n_rows;                             /* rows in the live index                 */
    int64_t rows_per_shard[ARTMIPS_MAX_DEVICES];
    int64_t index_bytes;                        /* bytes of row storage over all shards   */
    int64_t workspace_bytes;                    /* device scratch currently allocated     */
    /* timing of the last search call, measured with HIP events on the streams the
     * kernels were launched on */
    float last_scan_ms;   /* sum over query tiles of the main scan kernel, slot 0        */
    float last_total_ms;  /* first kernel start .. last kernel end, slot 0                */
    int32_t last_scan_launches;
    int32_t last_retries; /* candidate-list overflow retries taken by the last call      */
    int64_t last_candidates; /* candidates kept by the filter, summed over queries        */
    float last_host_enqueue_us; /* host time to plan + enqueue the last search (all shards) */
    float last_host_sync_us;    /* host time spent waiting for the stream afterwards        */
} artmips_info_t;

/* Replaces DistributedBruteForceIndex.__init__ (art_index.py:201-209): one handle
 * owning one row shard per entry of device_ids (art_index.py:204 uses every visible
 * GPU).  The same device id may be listed more than once (two shards on one GPU);
 * n_dev == 1 is the per-rank form used by the SPMD path.
 * dim: any multiple of 128 up to 1024 for f16 / bf16 rows (768 = the reference's BERT-base width,
 * 128 and 768 also score up to 256 queries per sweep and 768 runs large batches in gangs, 1024 up to
 * 128 per sweep, the other widths 64); 128 or 768 for f32 rows.  The call also sets the function
 * attributes of every kernel the handle can launch, so that no search ever does (a search can be
 * enqueued under HIP graph capture).  Devices without peer access to one another are accepted: the
 * copies between them are then staged by the runtime. */
int artmips_create(const int* device_ids, int n_dev, int dim, int storage_dtype, artmips_t** out);

/* Frees every device buffer.  Replaces `del self.evidence_embeds`
 * (art_index.py:221,233). */
void artmips_destroy(artmips_t* h);

/* Start building the next generation of the index with n_rows_total rows, split
 * into contiguous ceil-div row chunks per shard exactly like torch.chunk
 * (art_index.py:251-253).  The live index keeps serving until artmips_finalize. */
int artmips_begin(artmips_t* h, int64_t n_rows_total);

/* Append n_rows host rows (storage dtype) and their passage ids at the upload
 * cursor of the generation being built; pinned, double-buffered H2D copies.  May
 * be called repeatedly.  Replaces `np.array(block_embeds)` + `.to('cuda:i')` and
 * the id_map loop (art_index.py:244-259).  flags: reserved, pass 0. */
int artmips_upload(artmips_t* h, const void* host_rows, const int32_t* ids, int64_t n_rows,
                   int flags);

/* Copy n rows that already live on a device into shard `slot` at row_offset
 * (offset inside that shard), asynchronously on `stream`.  The index-refresh path
 * without the pickle round trip (indexer.py:146-165 -> art_index.py:231-265) when
 * the caller already knows where its rows go. */
int artmips_upload_device(artmips_t* h, int slot, const void* dev_rows, const int32_t* dev_ids,
                          int64_t n, int64_t row_offset, void* stream);

/* Scatter a batch of freshly encoded rows (on device `src_device`, dtype src_dtype,
 * converted to the storage dtype) into the generation being built: the row with
 * passage id `id` lands at global position id - id_base, i.e. the index ends up in
 * id order whatever order the encoder ranks deliver their batches in.  Replaces
 * detach() + add_block_data (indexer.py:146-149, art_index.py:56-61) + the shard
 * pickles (art_index.py:63-100) + the re-upload (art_index.py:231-265).
 * Asynchronous on `stream`.  A row uploaded twice or never is reported by
 * artmips_finalize (the reference's ValueError at art_index.py:59 and the
 * indexer's completeness assert at indexer.py:161).  Ids must not be
 * -2139062144 (the "not uploaded yet" marker). */
int artmips_upload_scatter(artmips_t* h, int src_device, const void* dev_rows, int src_dtype,
                           const int64_t* dev_ids, int64_t n, int64_t id_base, void* stream);

/* Make the generation being built the live index and free the old one.
 * Fails (ARTMIPS_ERR_STATE, generation kept) if a row was never uploaded or was
 * scattered twice.  Replaces the tail of add_embed_data / update_index
 * (art_index.py:231-265). */
int artmips_finalize(artmips_t* h);

/* Replaces DistributedBruteForceIndex.search_mips_index (art_index.py:267-304):
 * exact top-k of q·Eᵀ over every shard, merged on the device of slot 0.
 *   q_dev       [n_q, dim] on the device of slot 0, dtype q_dtype
 *   out_scores  [n_q, k]   on the device of slot 0, dtype score_dtype
 *   out_ids     [n_q, k]   int32 passage ids */
int artmips_search(artmips_t* h, const void* q_dev, int q_dtype, int64_t n_q, int k,
                   void* out_scores, int score_dtype, int32_t* out_ids, void* stream);

/* artmips_search (art_index.py:267-304) without the host wait and without the overflow retry:
 * everything up to the final merge is enqueued on `stream` (and, for several shards, on the shards'
 * own streams, which `stream` then waits for) and the call returns.  out_status: 4 int32 words on
 * the device of slot 0, 8-byte aligned, written by the kernels; once `stream` has passed the call,
 * word 0 holds bit 0 = a candidate list overflowed (the result may have lost a top-k member: call
 * artmips_grow_lists and repeat), bit 1 = internal scratch too small.  Queries and outputs must
 * stay alive until `stream` has passed the call. */
int artmips_search_async(artmips_t* h, const void* q_dev, int q_dtype, int64_t n_q, int k,
                         void* out_scores, int score_dtype, int32_t* out_ids, int32_t* out_status,
                         void* stream);

/* One shard only: the local top-k a rank contributes in the SPMD path (one
 * process per GPU).  Outputs are fp32 scores + int32 passage ids on that shard's
 * device, ready for an RCCL all-gather.  Replaces the per-GPU matmul of
 * art_index.py:278-280 plus the part of torch.topk (art_index.py:294) that can be
 * done before the exchange.  If the shard holds fewer than k rows the tail is
 * filled with (-inf, INT32_MAX). */
int artmips_search_local(artmips_t* h, int slot, const void* q_dev, int q_dtype, int64_t n_q,
                         int k, float* out_scores, int32_t* out_ids, void* stream);

/* k-way merge of n_lists candidate lists, each [n_q, k] (fp32 score, int32 id),
 * laid out [n_lists, n_q, k] as an all-gather leaves them.  Runs on `device`,
 * asynchronously on `stream` (the only entry point that does not synchronise).
 * Replaces the gather of score slabs to GPU-0 and torch.topk over them
 * (art_index.py:283-294) with a merge over k*n_lists candidates per query. */
int artmips_merge(int device, const float* scores, const int32_t* ids, int n_lists, int64_t n_q,
                  int k, void* out_scores, int score_dtype, int32_t* out_ids, void* stream);

/* ---- the SPMD step without a host round trip in the middle -------------------------
 * art_model.py:345-365 (gather queries, search on one rank, two broadcasts) becomes,
 * on every rank:  artmips_search_local_async -> ONE all-gather of the packed buffer
 * (RCCL, torch.distributed.all_gather_into_tensor) -> artmips_merge_packed -> read
 * one int32 flag.  Nothing waits for the host between the three steps, so their
 * launch latencies hide behind the scan.
 *
 * A packed candidate list is artmips_packed_words(n_q, k) int32 words:
 *   [n_q*k] fp32 score bits | [n_q*k] int32 passage ids | ARTMIPS_PACK_TAIL status words
 * (word 0: bit 0 = a candidate list overflowed, the result may be inexact;
 *  bit 1 = internal scratch too small). */
int64_t artmips_packed_words(int64_t n_q, int k);

/* artmips_search_local (art_index.py:278-280,294) without the wait and without the
 * overflow retry: enqueues the shard's pipeline on `stream` and returns.  The
 * overflow bit travels in the status tail of out_packed instead (the kernels raise it
 * there directly when out_packed is 8-byte aligned). */
int artmips_search_local_async(artmips_t* h, int slot, const void* q_dev, int q_dtype, int64_t n_q,
                               int k, int32_t* out_packed, void* stream);

/* artmips_search_local_async for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the
 * reference.  Same arguments, same packed result, bit for bit; but the shard's workspace
 * (thresholds, pass-0 maxima, candidate lists and counters) lives in a static device array of
 * ARTMIPS_NOALLOC_SCRATCH_BYTES (64 MiB) per GPU instead of buffers the library grows on demand, and
 * the call performs no hipMalloc / hipFree / event creation.  A search whose workspace does not fit
 * is refused with ARTMIPS_ERR_UNSUPPORTED before anything is launched (the ART training step fits:
 * 64 queries x top-32 over 21M rows needs ~34 MiB; top-1000 batches do not).  out_packed must be
 * 8-byte aligned.  Every call on a GPU shares that GPU's array: calls that overlap on one GPU
 * (two streams, two handles, two threads) race; calls on one stream are ordered and safe. */
int artmips_search_local_noalloc(artmips_t* h, int slot, const void* q_dev, int q_dtype, int64_t n_q,
                                 int k, int32_t* out_packed, void* stream);

/* artmips_merge over n_lists packed lists laid out back to back (what the all-gather
 * leaves).  If out_flag (device int32) is given it receives the OR of the lists'
 * status words: every rank reads the same value, so all ranks agree on a retry.
 * Asynchronous on `stream`. */
int artmips_merge_packed(int device, const int32_t* packed, int n_lists, int64_t n_q, int k,
                         void* out_scores, int score_dtype, int32_t* out_ids, int32_t* out_flag,
                         void* stream);

/* artmips_merge_packed followed by the step's one host wait (where the reference's caller
 * waits in its two broadcasts and .tolist(), art_model.py:361-365, evaluate.py:173-185): the merge writes the OR of
 * the lists' status words straight into a pinned host word (no device-to-host copy is
 * queued), `stream` is synchronised and the word is returned in *flag_host_out (a HOST
 * pointer).  On return the merged lists are complete.  Calls on different devices do not
 * serialise; on one device only their enqueue does (and the wait too when n_lists * k > 8192). */
int artmips_merge_packed_sync(int device, const int32_t* packed, int n_lists, int64_t n_q, int k,
                              void* out_scores, int score_dtype, int32_t* out_ids,
                              int32_t* flag_host_out, void* stream);

/* After a flag with bit 0 set: make the candidate lists of this handle 8x larger
 * (what artmips_search does internally before it retries), then repeat the step. */
int artmips_grow_lists(artmips_t* h);

/* Read n rows of the live index back to the host as fp32: positions[i] is a global row
 * position (0-based, in upload order; the reference's id_map maps them to passage ids,
 * art_index.py:257-259), out_host is [n, dim] floats.  Serves the `reconstruct=True`
 * half of the sibling index's search (FaissMIPSIndex.search_mips_index,
 * art_index.py:190-193: faiss's search_and_reconstruct).  Synchronous. */
int artmips_fetch_rows(artmips_t* h, const int64_t* positions, int64_t n, float* out_host);

/* Generation counter of the live index: it increases with every successful
 * artmips_finalize, i.e. with every refresh (update_evidence_embedding, art_model.py:336-342).
 * Pair with artmips_download.  No counterpart in the reference, whose index is rebuilt
 * in place. */
int artmips_generation(artmips_t* h, uint64_t* out);

/* Copy rows [row_offset, row_offset + n_rows) of the live index (global row positions,
 * across shards) and their passage ids to host memory, in the storage dtype.  This is the
 * way back to the reference's on-disk artefact after a device-side refresh: the reference
 * keeps its index "checkpoint" as the pickle at --embedding-path, rewritten on every refresh
 * (indexer.py:156-163, art_index.py:72-100).  `generation` is the value artmips_generation
 * returned when the export began: if a refresh replaces the index meanwhile the call fails with
 * ARTMIPS_ERR_STATE instead of mixing two generations.  Uses its own streams and pinned
 * staging and takes the handle's generation lock per 16 MiB chunk, so it MAY run on another
 * thread beside searches and beside the upload calls of the next refresh. */
int artmips_download(artmips_t* h, uint64_t generation, int64_t row_offset, int64_t n_rows,
                     void* host_rows, int32_t* host_ids);

/* Turn the HIP-event timing of the search pipeline on (default) or off.  The event records sit
 * between the kernels in the stream and cost a few microseconds each, which is noise on a 32 GB
 * shard and 2 % on a 4 GB one; with timing off artmips_info keeps reporting the last timed call.
 * No counterpart in the reference (it has no instrumentation on this path). */
int artmips_set_stats(artmips_t* h, int enabled);

/* Shard sizes, memory and the HIP-event timings of the last search. */
int artmips_info(artmips_t* h, artmips_info_t* out);

/* Message of the last error on this thread ("" if none). */
const char* artmips_last_error(void);

/* ABI version of this header. */
int artmips_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ARTMIPS_H */
