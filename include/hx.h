/*
 * hx.h -- C ABI of the MI355X-native hybrid retrieval engine (libhx.so).
 *
 * This is the drop-in boundary for the Qdrant-backed search path of
 * VivekMalipatel/RAG_Application.  The reference has no FFI of its own: its
 * seam is the Python class QdrantHandler, which forwards every operation to a
 * Qdrant server over HTTP.  Each entry point below cites the reference call it
 * replaces (paths relative to the reference root).  INTEGRATION.md shows the
 * ctypes stub a maintainer adds on the reference side.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; the message is
 *     available from hx_last_error() (thread-local).
 *   - "host" pointers are ordinary process memory; "dev" pointers are HIP device
 *     memory on the index's device.  `stream` is a hipStream_t passed as void*
 *     (NULL = the default stream).  Device entry points enqueue work on
 *     `stream`; they synchronise it only where stated.
 *   - a ranked list is exchanged between stages as an array of 64-bit KEYS,
 *     `keys[b*stride + r]`, r = rank, sorted best-first, with `counts[b]` valid
 *     entries; key = (orderable(score) << 32) | (0xFFFFFFFF - id), so that the
 *     descending integer order IS the engine's total order (score descending,
 *     id ascending).  0 marks an empty slot.  ids are global row ids, < 2^32 - 1:
 *     id_base + local row, or -- a shard that holds slices of many batches -- the
 *     insertion-order ids named batch by batch with hx_set_next_id.
 *   - no torch types, no C++ types: plain pointers and sizes only.
 */
#ifndef HX_H
#define HX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HX_ABI_VERSION 3   /* 2: hx_set_next_id, hx_add_rows_dev, hx_truncate; max_terms dropped from two entries
                            * 3: the candidates-first sharded H1 (hx_h1_plan / _nominate_async / _rescore_async /
                            *    _finish), hx_sparse_wmax / hx_set_sparse_wmax */

typedef struct hx_index hx_index;

/* The 8-key search_params contract
 * (app/services/agents/hybrid_search_workflow.py:8-19,
 *  app/api/v1/endpoints/mcp/qdrant_search_mcp_endpoint.py:19-28) plus the
 * switches for semantics inherited from Qdrant (oracle/oracle.py names them). */
typedef struct hx_params {
  int32_t matryoshka_64_limit;   /* limit of the innermost prefix stage (first matryoshka size)   */
  int32_t matryoshka_128_limit;  /* second prefix stage                                           */
  int32_t matryoshka_256_limit;  /* third prefix stage                                            */
  int32_t dense_limit;
  int32_t quantized_limit;
  int32_t sparse_limit;
  int32_t final_limit;
  int32_t hnsw_ef;               /* accepted, unused: every stage is exact (qdrant_handler.py:369) */
  float   rrf_k;                 /* 2.0  : Qdrant RRF constant                                      */
  int32_t rrf_rank_base;         /* 0    : 0-based ranks                                            */
  int32_t rrf_limit;             /* 10   : default limit of a Prefetch without `limit`              */
  int32_t mode;                  /* HX_MODE_TREE or HX_MODE_H1                                      */
} hx_params;

#define HX_MODE_TREE 0  /* the reference query tree, qdrant_handler.py:305-372                  */
#define HX_MODE_H1   1  /* dense top-dense_limit (+) sparse top-sparse_limit -> RRF -> final_limit */

/* ---- lifecycle ---------------------------------------------------------- */

/* create_collection (qdrant_handler.py:24-117): one index = one user collection
 * holding the named vectors dense / quantized / matryoshka_* and the sparse
 * vector.  `msizes` = matryoshka prefix sizes (ascending, <= 3 of them, each a
 * multiple of 64 and <= dim); n_msizes may be 0.  `id_base` = global id of local
 * row 0 (row sharding).  `device` = HIP device ordinal. */
int hx_create(int32_t dim, const int32_t* msizes, int32_t n_msizes, int32_t device,
              int64_t id_base, hx_index** out);
/* delete_collection (qdrant_handler.py:430-439) */
int hx_destroy(hx_index* h);
const char* hx_last_error(void);
int hx_abi_version(void);

/* ---- ingest: store_document_vectors / store_chat_vectors ------------------
 * (qdrant_handler.py:120-198, 200-267 -> AsyncQdrantClient.upsert :190-193) */

/* optional: pre-size device storage for `n_rows` rows and `nnz` sparse entries */
int hx_reserve(hx_index* h, int64_t n_rows, int64_t nnz);
/* append n raw dense rows [n x dim] (host fp32).  Derives on device, per row:
 * the L2-normalised "dense" vector, the normalised prefixes, the int8
 * "quantized" copy trunc(127*x) (qdrant_handler.py:144-150) and its norm. */
int hx_add_dense(hx_index* h, const float* rows_host, int64_t n);
/* the same for rows already on the device -- the output of an encoder on PyTorch-ROCm
 * (embedding_handler.py:64-99 produces them; qdrant_handler.py:120-198 stores them): read in place,
 * no staging copy.  Returns when the rows are stored. */
int hx_add_dense_dev(hx_index* h, const float* rows_dev, int64_t n, void* stream);
/* append the sparse vectors of the same n rows as doc-major CSR (host):
 * indptr[n+1], idx[nnz] (term ids in [0, 2^31)), val[nnz].  Indices must be
 * unique within a row (Qdrant rejects duplicates).  Rows must be added in the
 * same order as hx_add_dense; a row may be empty.  Values must be finite with
 * |v| <= 1e18 (anything else is refused, nothing is stored).  Sparse vectors are the
 * vectors of the NEXT rows: call it BEFORE hx_add_dense of the same rows (rows that got
 * no sparse vector earlier are padded as empty documents first; a second call before the
 * dense rows of the first arrived is refused). */
int hx_add_sparse(hx_index* h, const int64_t* indptr_host, const int32_t* idx_host,
                  const float* val_host, int64_t n);
/* one chunk batch, dense and sparse together, all or nothing (store_document_vectors builds
 * one PointStruct per chunk carrying both, qdrant_handler.py:152-188): indptr_host NULL = no
 * sparse vectors.  A failure leaves the index exactly as it was. */
int hx_add_rows(hx_index* h, const float* rows_host, const int64_t* indptr_host, const int32_t* idx_host,
                const float* val_host, int64_t n);
/* hx_add_rows for dense rows that already lie on the device (the output of an encoder on PyTorch-ROCm): the same
 * all-or-nothing contract, the sparse CSR still comes from the host (bm25 runs on the host cores). */
int hx_add_rows_dev(hx_index* h, const float* rows_dev, const int64_t* indptr_host, const int32_t* idx_host,
                    const float* val_host, int64_t n, void* stream);
/* Row sharding with insertion-order ids (the reference upserts a collection in MANY batches,
 * app/services/file_processor/text_processor.py:357 -> qdrant_handler.py:190-193; a shard then holds a slice of
 * every batch): the NEXT add call's rows get the global ids first_id, first_id + 1, ...  Ids must ascend with the
 * rows of a shard (first_id >= every id given so far).  Without this call a batch continues the ids of the previous
 * one, starting at hx_create's id_base.  Every key that leaves the index carries global ids, and hx_rescore takes
 * global ids; a failed add consumes the call. */
int hx_set_next_id(hx_index* h, int64_t first_id);
/* Roll the collection back to its first n_rows rows (dense and sparse): how the shards that stored their slice of
 * a batch undo it when another shard could not (one upsert = one request in the reference, :190-193). */
int hx_truncate(hx_index* h, int64_t n_rows);
/* build the on-device inverted index over everything added so far; searches
 * call it implicitly when the index is stale. */
int hx_finalize(hx_index* h);
/* get_collection_chunk_count (qdrant_handler.py:441-481) */
int hx_count(hx_index* h, int64_t* n_rows);
int hx_nnz(hx_index* h, int64_t* nnz);

/* fill rows [row0, row0+n) with the synthetic corpus of SURVEY.md 8(d),
 * generated on the device (oracle/oracle.py synth_dense / synth_sparse_docs give
 * the same values).  Global row r = id_base + local row.  cdf_u32[V] and
 * len_u16[256] are the shared lookup tables (host).  with_sparse=0 skips the
 * sparse side. */
int hx_synth_fill(hx_index* h, int64_t n, uint32_t seed_dense, uint32_t seed_sparse,
                  const uint32_t* cdf_u32_host, int32_t V, const uint16_t* len_u16_host,
                  int32_t with_sparse);
/* synthetic query batch on the device: q_dev [B x dim] fp32 rows q0..q0+B-1 */
int hx_synth_queries_dense(int32_t dim, int64_t q0, int32_t B, uint32_t seed,
                           float* q_dev, void* stream);

/* ---- whole-collection stages (device in, device out) ---------------------- */

/* Prefetch(query=dense_vector[:prefix], using="matryoshka_<prefix>"|"dense", limit)
 * (qdrant_handler.py:311-315, 327-329, 366-368).  q_dev: B raw (un-normalised)
 * query rows [B x dim] fp32.  prefix = 0 searches the full vector.  Output: keys
 * [B x limit] + counts[B].  Synchronises `stream` (exactness certificate). */
int hx_search_dense(hx_index* h, const float* q_dev, int32_t B, int32_t prefix, int32_t limit,
                    uint64_t* keys_dev, int32_t* counts_dev, void* stream);
/* Prefetch(query=quantized_query, using="quantized", limit) (qdrant_handler.py:299-302,335-339) */
int hx_search_i8(hx_index* h, const float* q_dev, int32_t B, int32_t limit,
                 uint64_t* keys_dev, int32_t* counts_dev, void* stream);
/* Prefetch(query=SparseVector, using="sparse", limit) (qdrant_handler.py:347-354).
 * Query batch as CSR on the device: indptr[B+1] (int64), idx (int32, strictly ascending
 * within a query: the exact score is a running fp32 sum in that order; the device checks it
 * and the call fails for a query that is not), val (fp32, finite). */
int hx_search_sparse(hx_index* h, const int64_t* q_indptr_dev, const int32_t* q_idx_dev,
                     const float* q_val_dev, int32_t B, int32_t limit,
                     uint64_t* keys_dev, int32_t* counts_dev, void* stream);

/* The three stages above WITHOUT their host round trip (the reference has no multi-device path; these serve the
 * row-sharded form of its query tree, qdrant_handler.py:305-372, where every rank enqueues a level, the ranks exchange
 * the level's lists, and nobody should wait for a flag in between): everything is enqueued on `stream` and the call
 * returns; the number of queries whose lists are NOT final (a stage flagged them for a retry or the exact path) is
 * ADDED to *flag_dev (device).  The caller zeroes the word before the first stage of a batch, reads it when it suits
 * it -- once, behind the whole tree -- and runs the batch again through the synchronous entries when it is not zero. */
int hx_search_dense_async(hx_index* h, const float* q_dev, int32_t B, int32_t prefix, int32_t limit,
                          uint64_t* keys_dev, int32_t* counts_dev, int32_t* flag_dev, void* stream);
int hx_search_i8_async(hx_index* h, const float* q_dev, int32_t B, int32_t limit,
                       uint64_t* keys_dev, int32_t* counts_dev, int32_t* flag_dev, void* stream);
int hx_search_sparse_async(hx_index* h, const int64_t* q_indptr_dev, const int32_t* q_idx_dev,
                           const float* q_val_dev, int32_t B, int32_t limit,
                           uint64_t* keys_dev, int32_t* counts_dev, int32_t* flag_dev, void* stream);

/* ---- candidate stages ------------------------------------------------------ */

/* outer level of a nested Prefetch / the root query (qdrant_handler.py:307-330,
 * 333-344, 363-372): re-score the candidate ids in cand_keys (scores ignored, ids
 * outside this shard skipped, duplicates merged) with one named vector and keep
 * `limit`. */
int hx_rescore(hx_index* h, const float* q_dev, int32_t B, int32_t prefix,
               const uint64_t* cand_keys_dev, int32_t cand_stride, const int32_t* cand_counts_dev,
               int32_t limit, uint64_t* keys_dev, int32_t* counts_dev, void* stream);
/* FusionQuery(Fusion.RRF) over two ranked lists (qdrant_handler.py:357-360) */
int hx_rrf(int32_t device, const uint64_t* a_keys_dev, int32_t a_stride, const int32_t* a_counts_dev,
           const uint64_t* b_keys_dev, int32_t b_stride, const int32_t* b_counts_dev,
           int32_t B, float rrf_k, int32_t rank_base, int32_t limit,
           uint64_t* keys_dev, int32_t* counts_dev, void* stream);
/* union / cross-shard merge: per query, top `limit` of the `stride` slots of
 * in_keys (0 = empty), duplicates optionally dropped.  in_counts may be NULL
 * (then every slot is examined). */
int hx_merge(int32_t device, const uint64_t* in_keys_dev, int32_t stride, const int32_t* in_counts_dev,
             int32_t B, int32_t limit, int32_t dedupe,
             uint64_t* keys_dev, int32_t* counts_dev, void* stream);
/* Row-sharded H1 query, one process per GPU (the reference has no
 * multi-device path -- these two calls bracket the one all-gather of the step):
 *  hx_h1_local  the shard's dense top-`dense_limit` and sparse top-`sparse_limit` of every
 *               query, side by side in keys_dev [B x (dense_limit + sparse_limit)]
 *               (qdrant_handler.py:347-354 and the dense Prefetch of the H1 configuration;
 *               0 = empty slot, ids are global: id_base + row);
 *  hx_h1_fuse   gathered_dev [world x B x (dense_limit + sparse_limit)] (rank-major, as
 *               all_gather_into_tensor leaves it): per query the global dense and sparse
 *               lists (top of the union of the shards' lists), then RRF as hx_rrf;
 *  hx_h1_local_async  hx_h1_local without its host round trip: everything is enqueued and the
 *               call returns; keys_dev is [(B + 1) x (dense_limit + sparse_limit)], row B holds
 *               in element 0 the number of queries whose lists are NOT final (a stage flagged
 *               them for a retry or the exact path) and zeros after it.  The caller reads that
 *               word when it suits it (it travels through the exchange with the lists, so every
 *               rank sees every rank's word) and redoes the batch through hx_h1_local when any
 *               rank's word is not zero.  Keeps consecutive batches back to back on the device. */
int hx_h1_local(hx_index* h, const float* q_dev, const int64_t* q_indptr_dev, const int32_t* q_idx_dev,
                const float* q_val_dev, int32_t B, int32_t dense_limit, int32_t sparse_limit,
                uint64_t* keys_dev, void* stream);
int hx_h1_local_async(hx_index* h, const float* q_dev, const int64_t* q_indptr_dev, const int32_t* q_idx_dev,
                      const float* q_val_dev, int32_t B, int32_t dense_limit, int32_t sparse_limit,
                      uint64_t* keys_dev, void* stream);
int hx_h1_fuse(int32_t device, const uint64_t* gathered_dev, int32_t world, int32_t B,
               int32_t dense_limit, int32_t sparse_limit, int32_t limit, float rrf_k, int32_t rank_base,
               uint64_t* keys_dev, int32_t* counts_dev, void* stream);
/* Row-sharded H1 with the exchange BEFORE the exact scores ("candidates first").  What a shard
 * repeats per QUERY whatever its row count -- the exact re-score of L' dense candidates, the exact re-score of the sparse
 * margin set, the compaction of full-size buffers -- is divided by the number of shards: a shard only nominates.
 * Three calls around two collectives per batch, everything enqueued (no host round trip):
 *  hx_h1_plan            list sizes for `world` shards: k1 = dense nominations per shard and query (its binomial share of
 *                        the global L' + 10 sigma), k2 = integer BM25 scores it sends, lp = L' (the global dense
 *                        candidate count the certificate needs for dense_limit), k3 = exact sparse keys it returns,
 *                        lout = the stride of its private integer-score list;
 *  hx_h1_nominate_async  nom_dev [B*k1 dense keys | B*k2 sparse keys | B*2 meta words | B*lout + B private words]: the
 *                        shard's best k1 rows by the int8 candidate score (hx_search_dense's candidate pass, no exact
 *                        score) and its k2 best integer BM25 scores of the select pass (sparse2.hip); the private tail
 *                        (its whole integer-score list) stays on this rank for hx_h1_rescore_async;
 *                        -> all-gather of the first B*(k1+k2+2) words over the shards;
 *  hx_h1_rescore_async   gathered_dev [world x B*(k1+k2+2)] + this rank's own nom_dev: the global cuts (top-lp by int8
 *                        score; the global L-th integer score, hence the margin-set threshold), the check that no
 *                        shard's list was cut above them, and the EXACT scores (spec_dot; upstream-order fp32 sparse sum)
 *                        of THIS shard's rows: res_dev [B*lp dense keys at their positions in the global list |
 *                        B*world*k3 sparse keys, this rank's best k3 in slot `rank` | B*world candidate counts | B*4
 *                        meta], 0 elsewhere;
 *                        -> all-reduce (SUM, as int64) of res_dev: every key slot has one owner, the others hold 0; the
 *                        four meta words per query are the same on every rank and come back multiplied by `world`;
 *  hx_h1_finish          reduced_dev: exact dense top-dense_limit with the certificate m + eps < e_L evaluated once on the
 *                        global list, exact sparse top-sparse_limit, RRF as hx_rrf.  *nfail_dev += queries whose lists
 *                        are not final (a shard's list was cut too short, a buffer overflowed, the certificate does not
 *                        hold): the caller then redoes the batch through hx_h1_local, as after hx_h1_local_async.
 * The integer BM25 scores of different shards are comparable only under one scale: hx_set_sparse_wmax gives every shard
 * the largest document weight of ANY shard (hx_sparse_wmax reads the shard's own; also whether it holds a non-positive
 * weight).  Needs the int8 candidate copy (the default). */
int hx_h1_plan(int32_t dense_limit, int32_t sparse_limit, int32_t world, int32_t* k1, int32_t* k2, int32_t* lp,
               int32_t* k3, int32_t* lout);
int hx_sparse_wmax(hx_index* h, float* wmax, int32_t* nonpos);
int hx_set_sparse_wmax(hx_index* h, float wmax);
int hx_h1_nominate_async(hx_index* h, const float* q_dev, const int64_t* q_indptr_dev, const int32_t* q_idx_dev,
                         const float* q_val_dev, int32_t B, int32_t dense_limit, int32_t sparse_limit, int32_t k1,
                         int32_t k2, uint64_t* nom_dev, void* stream);
int hx_h1_rescore_async(hx_index* h, const float* q_dev, const int64_t* q_indptr_dev, const int32_t* q_idx_dev,
                        const float* q_val_dev, int32_t B, const uint64_t* nom_dev, const uint64_t* gathered_dev,
                        int32_t world, int32_t rank, int32_t dense_limit, int32_t sparse_limit, int32_t k1, int32_t k2,
                        int32_t lp, int32_t k3, uint64_t* res_dev, void* stream);
int hx_h1_finish(int32_t device, const uint64_t* reduced_dev, int32_t world, int32_t B, int32_t lp, int32_t k3,
                 int32_t dense_limit, int32_t sparse_limit, int32_t limit, float rrf_k, int32_t rank_base,
                 uint64_t* keys_dev, int32_t* counts_dev, int32_t* nfail_dev, void* stream);
/* keys -> (fp32 score, int64 id); empty slots give (-inf, -1) */
int hx_unpack(int32_t device, const uint64_t* keys_dev, int64_t n, float* scores_dev,
              int64_t* ids_dev, void* stream);

/* ---- whole query, one shard (host in, host out) ----------------------------
 * QdrantHandler.hybrid_search -> query_points (qdrant_handler.py:296-372).
 * q_dense_host [B x dim] raw queries; sparse queries as CSR (indices need not be
 * sorted).  Outputs: scores/ids [B x final_limit], counts[B]. */
int hx_hybrid_query_host(hx_index* h, const float* q_dense_host,
                         const int64_t* q_indptr_host, const int32_t* q_idx_host,
                         const float* q_val_host, int32_t B, const hx_params* p,
                         float* scores_host, int64_t* ids_host, int32_t* counts_host);
/* same, device-resident inputs and outputs (bench path; sparse idx strictly ascending per query, checked) */
int hx_hybrid_query_dev(hx_index* h, const float* q_dense_dev,
                        const int64_t* q_indptr_dev, const int32_t* q_idx_dev,
                        const float* q_val_dev, int32_t B, const hx_params* p,
                        uint64_t* keys_dev, int32_t* counts_dev, void* stream);

/* ---- sparse text provider (host cores) ---------------------------------------
 * EmbeddingHandler.encode_sparse (app/core/embedding/embedding_handler.py:101-142 -> fastembed
 * Qdrant/bm25 :41, :123), batched (the reference's TODO :100): n texts -> CSR of (term id, weight)
 * rows.  texts[i] = lens[i] UTF-8 bytes.  flags[i] = 1 marks a text with non-ASCII bytes: its row
 * is empty and the caller handles it (rag_application_amd/bm25.py).  cap = capacity of idx/val
 * (sum of lens[i]/2 + 1 suffices).  threads <= 0: one per core, at most 16. */
int hx_bm25_embed_batch(const char* const* texts, const int64_t* lens, int64_t n, double k, double b,
                        double avg_len, int32_t threads, int64_t* indptr, int32_t* idx, double* val,
                        int64_t cap, int32_t* flags);

/* ---- persistence ------------------------------------------------------------
 * The reference asks Qdrant for on-disk vectors (qdrant_handler.py:47-55, 62: on_disk=True,
 * memmap_threshold).  hx_save writes the collection to one file (stored vectors as they stand plus
 * the sparse CSR); hx_load creates a new index from it -- searches return the same lists, bit for
 * bit; the inverted index is rebuilt on the device. */
int hx_save(hx_index* h, const char* path);
int hx_load(const char* path, int32_t device, hx_index** out);

/* ---- introspection (tests, bench) ------------------------------------------ */
typedef struct hx_stats {
  int64_t n_rows, nnz, n_segments;
  int64_t n_groups;       /* live terms of the inverted index */
  int64_t hash_capacity;  /* entries of its [live term x segment] offset table (field names kept from ABI v1) */
  int64_t bytes_dense_f32, bytes_dense_f16, bytes_i8, bytes_prefix, bytes_sparse;
  int64_t dense_fallback_queries;   /* queries whose certificate failed so far */
  int64_t i8_fallback_queries;
  int64_t retry_queries;            /* queries re-run with the safe geometry (overflow, underflow, certificate) */
  int64_t sparse_fallback_queries;  /* sparse queries served document-at-a-time (non-positive weights, > 64 terms, overflow) */
  /* ABI 2: the dense stage's candidate pass on the int8 matrix pipe (a per-row-scaled int8 copy of the normalised rows;
   * final scores stay exact fp32).  Queries it took, queries whose certificate failed (re-run through the fp16 scan). */
  int64_t bytes_i8_cand;
  int64_t cand8_queries;
  int64_t cand8_uncertified_queries;
  double  cand8_row_error_max;      /* largest ||x - scale * x8||_2 of any stored row: what the certificate is built from */
  int64_t tree_batches_redone;      /* HX_MODE_TREE batches whose deferred flag word was set: run again stage by stage */
  /* ABI 3: the two speculative paths switch themselves off on a collection they keep failing on (rows the int8 grid
   * resolves badly make the certificate's radius large for EVERY query; a flagged query costs the whole batch twice):
   * 1 once more than 1 query in 20 of a 4096-query window was uncertified (the fp16 copy nominates from then on),
   * 1 once 4 of 16 consecutive tree batches were redone (the tree reads its stages' flags one by one from then on). */
  int64_t cand8_switched_off;
  int64_t tree_deferral_switched_off;
} hx_stats;
int hx_get_stats(hx_index* h, hx_stats* out);
/* HIP-event profile of the hot kernels, measured on the stream they run on.
 * Index 0 = fp16 scan (k_scan<F16>), 1 = int8 scan of the "quantized" stage (k_scan<I8>), 3 = int8 candidate scan of
 * the dense stage (the same kernel over the per-row-scaled copy), 4 = the ingest kernel (K1/K2: k_prep_rows; bytes =
 * the raw row read once + every derived copy written once), 5 = spare, 2 = sparse scoring
 * (k_sparse_select: the pass over the inverted index; bytes = 8 per posting of the queries' terms).  flops/bytes are ALGORITHMIC: 2*B*rows*D and rows*row_bytes +
 * B*row_bytes per scan launch.  hx_profile_read drains what was recorded
 * since the last read (it synchronises the recorded events). */
#define HX_PROF_SLOTS 6
typedef struct hx_prof {
  int64_t launches[HX_PROF_SLOTS];
  double ms[HX_PROF_SLOTS];
  double flops[HX_PROF_SLOTS];
  double bytes[HX_PROF_SLOTS];
} hx_prof;
/* Which copy nominates the candidates of the full-vector dense stage: 1 = the per-row-scaled int8 copy (the
 * default; a query its certificate does not cover is re-run on the fp16 copy), 0 = the fp16 copy.  The lists are the
 * same either way (final scores are exact fp32): this is a measurement and test switch. */
int hx_set_dense_candidates(hx_index* h, int32_t kind);
/* Where the sparse stage of a hybrid call runs: 1 (the default) = on the index's second stream, beside the dense scans;
 * 0 = every stage on the caller's stream, one kernel at a time -- a measurement switch: a kernel's duration (hx_profile)
 * is its own only when nothing runs beside it.  The lists do not depend on it. */
int hx_set_stream_overlap(hx_index* h, int32_t on);
/* Build the inverted index again from the stored sparse vectors (K9; hx_finalize builds it once and keeps it): a
 * measurement aid for the index-build rate -- the first build of a process also pays for its temporary allocations. */
int hx_rebuild_sparse(hx_index* h);
int hx_profile(hx_index* h, int32_t enable);
int hx_profile_read(hx_index* h, hx_prof* out);
/* copy the derived row `row` (local) of one named vector to the host:
 * which = 0 dense f32 [dim], 1..3 prefix f32 [msizes[which-1]], 4 int8 [dim] (the "quantized" vector),
 * 5 int8 [dim] the candidate-pass copy of the normalised row, 6 f32 [1] its scale */
int hx_debug_row(hx_index* h, int32_t which, int64_t row, void* out_host);

#ifdef __cplusplus
}
#endif
#endif /* HX_H */
