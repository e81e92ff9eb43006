/* arttok.h — C ABI of libarttok.so: the evidence TOKEN store resident in HBM and the batched
 * id -> tokens fetch that follows every search.
 *
 * Reference path replaced: the Python loop of PreComputedEvidenceDocsRetriever.get_topk
 * (megatron/model/art_model.py:367-381), which reads Q*k*4 items one by one from four mmap token
 * datasets (megatron/data/indexed_dataset.py:481-486, MMapIndexedDataset.__getitem__) and turns
 * each into a Python list.  Here a dataset's tokens live on the GPU next to the index and one kernel
 * launch gathers the rows of all Q*k winners into a padded [n, max_len] int32 matrix plus lengths.
 *
 * Conventions as in artmips.h: 0 on success, negative code on failure, arttok_last_error() for
 * the message; plain pointers, no torch types; `stream` is a hipStream_t as void*;
 * arttok_gather is asynchronous on `stream`.
 */
#ifndef ARTTOK_H
#define ARTTOK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct arttok arttok_t;

enum {
    ARTTOK_OK = 0,
    ARTTOK_ERR_ARG = -1,
    ARTTOK_ERR_HIP = -2,
    ARTTOK_ERR_NOMEM = -5
};

/* Upload one token dataset (the .bin / .idx pair of indexed_dataset.py:335-512) to `device`.
 *   tokens_host  the .bin file's bytes: n_tokens elements of elem_bytes (1, 2, 4 or 8) each,
 *                unsigned for 1 and 2 bytes (the reference writes uint16 for vocabularies below
 *                65,500: indexed_dataset.py:24-28), signed for 4 and 8
 *   pointers     [n_items] BYTE offset of every item in the .bin (Index._pointers, :416)
 *   sizes        [n_items] element count of every item (Index._sizes, :410)
 * Every item must lie inside the token array. */
int arttok_create(int device, const void* tokens_host, int64_t n_tokens, int elem_bytes,
                  const int64_t* pointers, const int32_t* sizes, int64_t n_items, arttok_t** out);

void arttok_destroy(arttok_t* h);

/* For each of the n ids (device int32, e.g. the [Q*k] topkindex of artmips_search) fetch item
 * `id - id_base` (art_model.py:374 uses idx - 1): out_tokens[i, :len] = its first
 * min(size, max_len) tokens widened to int32, the rest of the row = pad; out_lens[i] = that
 * length.  An id outside [id_base, id_base + n_items) gives length -1 and a row of pad.
 * out_tokens is [n, max_len] int32, out_lens [n] int32, both on the store's device. */
int arttok_gather(arttok_t* h, const int32_t* ids_dev, int64_t n, int32_t id_base, int max_len,
                  int32_t pad, int32_t* out_tokens, int32_t* out_lens, void* stream);

/* The same fetch for up to ARTTOK_MAX_SETS datasets on one device in ONE launch: the four token
 * datasets of a training step (passages / titles, BERT / T0 vocabularies) are looked up for the same
 * ids (art_model.py:367-381: the reference's loop reads all four per winner).  handles, max_lens,
 * pads, out_tokens and out_lens are HOST arrays of n_sets entries; outputs as for arttok_gather. */
#define ARTTOK_MAX_SETS 4
int arttok_gather_multi(arttok_t* const* handles, int n_sets, const int32_t* ids_dev, int64_t n,
                        int32_t id_base, const int32_t* max_lens, const int32_t* pads,
                        int32_t* const* out_tokens, int32_t* const* out_lens, void* stream);

/* items, tokens and bytes held on the device */
int arttok_info(arttok_t* h, int64_t* n_items, int64_t* n_tokens, int64_t* device_bytes);

const char* arttok_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* ARTTOK_H */
