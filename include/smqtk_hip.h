/*
 * smqtk_hip.h -- C ABI of libsmqtk_hip.so, the MI355X (gfx950) kNN hot path
 * behind SMQTK-Indexing's plugin surface.
 *
 * The reference (Kitware/SMQTK-Indexing v0.18.0) is pure Python and has no
 * FFI of its own; each entry point below names the reference code whose
 * arithmetic it replaces (paths relative to the reference root).  The only
 * callers are the plugin classes in smqtk_indexing_amd/impls/ (through
 * ctypes, smqtk_indexing_amd/_lib.py) -- see INTEGRATION.md for the stub a
 * reference maintainer would add.
 *
 * Conventions
 *   - every function returns SQ_OK (0) or a negative SQ_ERR_* code; the
 *     message of the last error on the calling thread is sq_last_error().
 *     Nothing throws across the ABI.
 *   - `mem` says where caller buffers live: SQ_MEM_HOST (library stages them
 *     through its own device workspace; the call is synchronous) or
 *     SQ_MEM_DEVICE (pointers are HIP device pointers on the current device;
 *     work is enqueued on `stream`).  The search calls (sq_dense_search,
 *     sq_hamming_search) return when the results are final: each query's
 *     answer is certified on the device and the host reads the status words
 *     (an uncertified query is redone on the exact path), so the call waits
 *     for its own kernels.  sq_itq_hash / sq_dense_distances with device
 *     buffers only enqueue.  SQ_MEM_DEVICE_ASYNC (sq_dense_search,
 *     sq_hamming_search) is the pipelined form: see sq_dense_search.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - all matrices are C-contiguous, row-major.
 *   - hash codes are uint64[n][words], word 0 most significant, bit 0 of the
 *     SMQTK bit vector = most significant bit of the code, code right-aligned
 *     (smqtk_indexing/utils/bits.py:4-20; impls/lsh_functor/itq.py:46-50).
 *   - result order is (distance ascending, then row id ascending); ids are
 *     `id_base + local row`.  When k exceeds the number of rows the tail is
 *     filled with id -1 and distance +inf / INT32_MAX.
 *   - handles own device memory; they are bound to the device that was
 *     current at create time and are freed only by *_destroy.
 */
#ifndef SMQTK_HIP_H
#define SMQTK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SQ_OK 0
#define SQ_ERR_INVALID (-1)     /* bad argument */
#define SQ_ERR_HIP (-2)         /* HIP runtime error (no device, launch failure, ...) */
#define SQ_ERR_NOMEM (-3)       /* device allocation failed */
#define SQ_ERR_UNSUPPORTED (-4) /* shape outside what the kernels cover */
#define SQ_ERR_INTERNAL (-5)    /* invariant violated (bug) */

#define SQ_MEM_HOST 0
#define SQ_MEM_DEVICE 1
#define SQ_MEM_DEVICE_ASYNC 2 /* sq_dense_search / sq_hamming_search: enqueue and return, results final one call later */

#define SQ_METRIC_L2 0     /* utils/metrics.py:73-86 euclidean_distance */
#define SQ_METRIC_COSINE 1 /* utils/metrics.py:89-137 cosine_distance (pos_vectors=True) */

#define SQ_DTYPE_F32 0
#define SQ_DTYPE_F64 1

/* ItqFunctor(normalize=...) (impls/lsh_functor/itq.py:172-191: numpy.linalg.norm(v, ord, axis, keepdims), zero
 * norms replaced by 1, v / norm in v's dtype).  These orders are evaluated on the device in numpy's own arithmetic;
 * any other order numpy accepts (general p) is normalised by the caller with numpy itself and hashed with
 * SQ_NORM_NONE (HipItqFunctor does that): its |x|**p needs libm's pow exactly as numpy calls it. */
#define SQ_NORM_NONE (-1)        /* normalize=None                 */
#define SQ_NORM_L0 0             /* normalize=0: count of non-zeros */
#define SQ_NORM_L1 1             /* normalize=1                    */
#define SQ_NORM_L2 2             /* normalize=2                    */
#define SQ_NORM_INF 1000         /* normalize=numpy.inf            */
#define SQ_NORM_NEG_INF (-1000)  /* normalize=-numpy.inf           */

#define SQ_MAX_K 16384    /* k up to here is answered by the one-workgroup select; larger k (the reference has no
                           * limit on n) is answered too, by a full device sort of the candidate keys: slower */

typedef int64_t sq_handle_t;

typedef struct sq_stats {
    double scan_ms;        /* last search: duration of the scan kernel(s) (hipEvent, when profiling is on) */
    double total_ms;       /* last search: whole enqueue-to-done time on the stream (when profiling is on) */
    int64_t scan_launches; /* kernels that streamed the database in the last search */
    int64_t candidates;    /* sum over queries of candidates the scan emitted */
    int64_t fallback_queries; /* queries that took the exact full-keys path */
    int64_t bytes_scanned; /* algorithmic bytes the scan streamed (rows * row bytes * passes) */
    double rerank_ms;      /* last dense search: duration of the exact re-rank kernel (when profiling is on; 0 otherwise) */
    int64_t mid_tier_queries; /* last dense search: queries the first filter could not certify that took the second,
                               * tighter filter pass; those it could not certify either are in fallback_queries */
} sq_stats_t;

/* ------------------------------------------------------------------ misc */
const char* sq_last_error(void);
int sq_version(void);
int sq_device_count(int* out_n);
int sq_device_name(int device, char* out_name, int name_len, int64_t* out_total_mem, int* out_cu_count);
/* options: "profile" (0/1: record hipEvents so sq_get_stats reports ms),
 * "sample_stride" (0 = auto), "candidate_cap" (0 = auto),
 * "force_fallback" (0/1: every query takes the exact full-keys path),
 * "merge_threads" (host threads of sq_merge_topk, 0 = by size),
 * "spin_wait_us" (a search polls its stream this long before it blocks; default 2000, 0 = block at once);
 * measurement / test knobs of the dense scan (0 = auto): "dense_qt" (1, 2 or 4
 * query tiles per scan wave), "dense_waves" (4 or 8), "dense_stages" (ring
 * depth), "dense_blocks" (row blocks), "dense_rerank_segments" (survivor
 * segments per re-rank workgroup), "dense_debug" (ablation bits). */
int sq_set_option(const char* name, int64_t value);
/* The same options for ONE handle (any kind): an override that wins over the process-wide value for every later call
 * on that handle, read under the handle's lock.  Two indexes searched from two threads keep their own pipeline depth,
 * candidate lists, profiling ... -- the reference's contract is "implementations should be thread safe"
 * (interfaces/nearest_neighbor_index.py:22-23), and its options are per instance (constructor arguments).
 * sq_handle_reset_options drops the handle's overrides. */
/* SQ_ERR_UNSUPPORTED where nothing would read the override: row-matrix and fit handles read no option, an ITQ model
 * reads "itq_exact" / "dense_debug" only, "spin_wait_us" and "merge_threads" are process-wide. */
int sq_handle_set_option(sq_handle_t h, const char* name, int64_t value);
int sq_handle_reset_options(sq_handle_t h);
int sq_get_stats(sq_handle_t h, sq_stats_t* out);

/* ------------------------------------------------------------------- ITQ
 * Replaces ItqFunctor.get_hash + _norm_vector applied to n descriptors
 * (impls/lsh_functor/itq.py:172-191, 389-408) followed by
 * bit_vector_to_int_large (utils/bits.py:4-20), i.e. the per-descriptor body
 * of LSHNearestNeighborIndex._build_index (impls/nn_index/lsh.py:316-321):
 *   z = (norm(x) - mean) . rotation   (float64),  bit = z >= 0
 * x: [n][d] of x_dtype; mean: [d] values as f64; rotation: [d][bits] f64;
 * out_codes: [n][ceil(bits/64)].
 * mean_dtype: dtype of the MODEL's mean vector.  numpy evaluates
 * `norm(x) - mean` in the promoted dtype of the two operands (itq.py:404), so
 * a float32 model applied to float32 descriptors subtracts in float32 (a model
 * ItqFunctor.fit trained on float32 descriptors has a float32 mean_vec); every
 * other combination subtracts in float64. */
int sq_itq_hash(const void* x, int x_dtype, int64_t n, int d,
                const double* mean, int mean_dtype, const double* rotation, int bits, int norm_ord,
                uint64_t* out_codes, int mem, void* stream);

/* --------------------------------------------------------------- ITQ fit
 * The O(n) products of ItqFunctor.fit (impls/lsh_functor/itq.py:291-387) and
 * _find_itq_rotation (itq.py:239-289) with the descriptor matrix resident on
 * the device; the d x d eigen-decomposition and the b x b SVD per iteration stay
 * with the caller (numpy, as in the reference).  d <= 512, bits <= 256 (the outputs are tiled 128 x 128).
 *   create:   x [n][d] of dtype; norm_ord as sq_itq_hash; out_mean[d] = column
 *             means of norm(x) (itq.py:330)
 *   set_mean: the mean values the model keeps (numpy stores them in x's dtype)
 *   cov:      out_cov[d][d] = np.cov((norm(x) - mean).T)          (itq.py:337)
 *   project:  v = (norm(x) - mean) . pc, pc [d][bits]; v stays on the device
 *                                                                  (itq.py:362)
 *   iterate:  ux = sign(v . r), out_c[bits][bits] = ux^T . v   (itq.py:271-275)
 */
int sq_itqfit_create(const void* x, int dtype, int64_t n, int d, int norm_ord, int mem,
                     double* out_mean, sq_handle_t* out);
int sq_itqfit_set_mean(sq_handle_t h, const double* mean);
int sq_itqfit_cov(sq_handle_t h, double* out_cov);
int sq_itqfit_project(sq_handle_t h, const double* pc, int bits);
int sq_itqfit_iterate(sq_handle_t h, const double* r, double* out_c);
int sq_itqfit_destroy(sq_handle_t h);

/* --------------------------------------------------------------- Hamming
 * Replaces LinearHashIndex._nn (impls/hash_index/linear.py:206-244):
 * heapq.nsmallest over the set of unique codes keyed by
 * metrics.hamming_distance (utils/metrics.py:140-155).  `codes` are the
 * UNIQUE packed codes (the caller dedups, linear.py:163); with
 * SQ_MEM_DEVICE the buffer is borrowed, not copied, and must outlive the
 * handle. */
int sq_hamming_create(const uint64_t* codes, int64_t n, int words, int mem,
                      int64_t id_base, sq_handle_t* out);
/* queries: [nq][words]; out_dist int32 [nq][k] (differing bits, NOT yet
 * divided by the bit length -- linear.py:243 does that on the host);
 * out_idx int64 [nq][k]. */
int sq_hamming_search(sq_handle_t h, const uint64_t* queries, int nq, int k,
                      int32_t* out_dist, int64_t* out_idx, int mem, void* stream);
/* mem = SQ_MEM_DEVICE_ASYNC: the pipelined form, with the contract of sq_dense_search's (below): the call enqueues
 * on an internal stream of its slot and returns; its results are final when the (depth - 1)-th later call on the
 * handle -- or sq_hamming_sync, a mutation, the destroy -- returns; until then `queries`, `out_dist` and `out_idx`
 * stay valid and untouched.  Options "hamming_async_depth" (2 .. 4), "hamming_async_wait", "hamming_async_order" as
 * their dense_* counterparts.  The shards of BASELINE config 5 (125 M x 256-bit codes per GPU) pipeline their
 * histogram / threshold / compaction / select kernels under the neighbouring calls' scans this way.
 * Calls of up to 32 queries over 64 .. 256-bit codes (k <= 2048) run as three launches -- sampled histogram, stream with
 * the thresholds computed in its prologue, one pick kernel per query (option "hamming_fused", 1 by default; 0 = the general
 * chain of five).  Their stream threshold is taken at a lower sample rank than k ("hamming_tighten", 1 by default): a bet
 * the pick kernel checks -- a call whose threshold admitted fewer than k codes is redone with the safe rule (counted in
 * sq_stats_t.fallback_queries); results are integer-exact either way. */
int sq_hamming_sync(sq_handle_t h);
/* Incremental mutation of an index that owns its device copy (created from host memory, or from a device array of
 * at least 4096 codes, which is copied): what LinearHashIndex._update_index / _remove_from_index do with a set
 * union / difference (impls/hash_index/linear.py:167-204), without re-uploading the whole code array.  Row ids are
 * ranks in the caller's sorted order of unique codes, so the caller says where things rank:
 *   append: new_codes [m][words] (host memory), ascending, none of them in the index; insert_pos[m] (host) =
 *           number of codes currently in the index that are smaller than new code j (ascending).  Afterwards the
 *           ids are the ranks in the merged order.
 *   remove: ranks[m] (host), strictly ascending current row ids that leave; the remaining codes close ranks.
 * Only the m new codes / m ranks cross PCIe; the device appends physically (or fills the holes from the tail) and
 * keeps an explicit rank per code (4 bytes, read for survivors only).  SQ_ERR_UNSUPPORTED for a borrowed array. */
int sq_hamming_append(sq_handle_t h, const uint64_t* new_codes, int64_t m, const int64_t* insert_pos);
int sq_hamming_remove(sq_handle_t h, const int64_t* ranks, int64_t m);
int sq_hamming_info(sq_handle_t h, int64_t* out_n, int* out_words); /* codes held, 64-bit words per code */
int sq_hamming_destroy(sq_handle_t h);

/* ----------------------------------------------------------------- dense
 * Exact brute-force kNN over a float32 matrix: the contract
 * FaissNearestNeighborsIndex expresses through faiss "IDMap,Flat"
 * (impls/nn_index/faiss.py:486-559, 681-701, 751-831) and the exact re-rank
 * tail of LSHNearestNeighborIndex._nn (impls/nn_index/lsh.py:505-519):
 * distance per row with metrics.euclidean_distance / cosine_distance, stable
 * ascending sort, first k.  L2 distances are float32 and bit-identical to
 * numpy's evaluation of metrics.py:86 (same subtraction, square, pairwise
 * summation order and sqrt); cosine distances are float64.
 * With SQ_MEM_DEVICE the matrix is borrowed (row stride d; the pointer must
 * be 16-byte aligned and d % 4 == 0, otherwise pass host memory and the
 * library pads the rows).
 * Any d: rows of up to 8192 padded dimensions are filtered from a bfloat16 copy (up to 512: the ring kernels, and an
 * int8 copy for small batches; beyond: dense_wide_scan_kernel, the widths of the reference's own examples --
 * 2048- / 4096-dimensional descriptors, docs/examples/caffe_build_index.rst:35); wider rows take the exact path. */
int sq_dense_create(const float* db, int64_t n, int d, int metric, int mem,
                    int64_t id_base, sq_handle_t* out);
/* The same with options of THIS index given at create, as name / value arrays: they become the handle's overrides
 * (sq_handle_set_option) BEFORE anything is built, so create-time choices are per index as the reference's are per
 * instance (constructor arguments: impls/nn_index/faiss.py:182-258) -- "dense_int8" = 0: no int8 copy is built or
 * kept (footprint 1.5x the matrix instead of 1.77x), "dense_no_center" = 1.  Two indexes of one process may differ. */
int sq_dense_create_opts(const float* db, int64_t n, int d, int metric, int mem, int64_t id_base,
                         const char* const* opt_names, const int64_t* opt_values, int n_opts, sq_handle_t* out);
/* What the index keeps resident and what its build cost; out[SQ_DENSE_INFO_FIELDS] =
 * { rows, d, bytes of the float32 rows, 1 if the library owns them (0: borrowed from the caller), bytes of the
 *   bfloat16 scan copy, bytes of the int8 scan copy + its row terms, bytes of the row statistics, 1 if the int8 first
 *   stage is in use, microseconds sq_dense_create took, microseconds of that spent on the int8 copy }.
 * (The build side of FaissNearestNeighborsIndex._build_index, impls/nn_index/faiss.py:486-559.) */
#define SQ_DENSE_INFO_FIELDS 10
int sq_dense_info(sq_handle_t h, int64_t* out, int n_out);
/* Append n_add rows ([n_add][d] float32, host or device) to an index that owns its matrix (created from
 * host memory); the new rows get the next row ids (id_base + old n ...).  Only the new rows cross PCIe and only
 * their statistics / scan-copy rows are built; the L2 filter keeps the origin chosen at create.
 * What FaissNearestNeighborsIndex._update_index does with add_with_ids of the new vectors only
 * (impls/nn_index/faiss.py:561-640), in place of a rebuild, when no existing descriptor is replaced.
 * SQ_ERR_UNSUPPORTED for an index that borrows a device matrix (the caller owns that allocation). */
int sq_dense_append(sq_handle_t h, const float* rows, int64_t n_add, int mem);

/* queries: [nq][d] f32.  out_dist: float32 [nq][k] for SQ_METRIC_L2,
 * float64 [nq][k] for SQ_METRIC_COSINE.  out_idx int64 [nq][k]. */
int sq_dense_search(sq_handle_t h, const float* queries, int nq, int k,
                    void* out_dist, int64_t* out_idx, int mem, void* stream);
/* mem = SQ_MEM_DEVICE_ASYNC: the call enqueues its kernels and returns without waiting for them; `queries`
 * are read in the order of `stream`.  The results of call i are final -- certified, uncertified queries redone,
 * complete in out_dist / out_idx -- when the NEXT call on the handle (another search, sq_dense_sync,
 * sq_dense_append, sq_dense_destroy) returns; until then the call's `queries`, `out_dist` and `out_idx` must
 * stay valid and untouched (so consecutive asynchronous calls alternate between two output buffers).  The device
 * never idles between calls, and with option "dense_async_streams" = 2 (default) consecutive calls run on two
 * internal streams so that the short kernels ending call i overlap those starting call i + 1.  sq_get_stats
 * reports the last FINISHED call.  sq_dense_sync finishes every call in flight.
 * Option "dense_async_depth" (2 by default, up to 4) keeps that many calls in flight: call i is then final when
 * call i + depth - 1 returns (callers rotate `depth` output buffers); small matrices -- the shards of a multi-GPU
 * index -- gain from 3, where the short kernels of a call weigh as much as its scan.  A change of the option takes
 * effect at the next asynchronous call, which first finishes the calls in flight.  Option "dense_async_wait" = 0
 * makes the call return right after enqueueing: the wait for the oldest call moves to the start of the next call, so
 * call i is final when call i + depth returns, and host work between two calls (a collective, a merge hand-off)
 * no longer delays the next enqueue.  Option "dense_async_order" = 0: the caller guarantees that `queries` are
 * complete when the call is made (no producer still running on `stream`); the call then skips the event that orders
 * its internal stream behind `stream` (worth ~10 us of start latency per call on a small matrix).
 * Option "dense_int8" (-1 by default): indexes of up to 512 dimensions and at least 65536 rows also keep an int8 copy
 * of the rows (128 / 256 / 512 + 4 bytes per row) and calls of up to 32 queries filter on it -- half the bytes of the bfloat16
 * pass; the results are the same bits (exact re-rank + certificate, as ever).  0 at create (process-wide, or for one
 * index through sq_dense_create_opts): no copy; 0 on a handle: the copy is not used; -1: after three calls in a row whose
 * candidate lists overflowed the filter is SUSPENDED (bfloat16 answers; the copy stays and follows appends) until the
 * index has doubled and chooses its clamp again, or until a call with 1 on the handle re-arms it; 1: never suspended.
 * Option "dense_int8_batch" (64 by default): the largest batch the int8 filter takes (128-byte rows; 32 = one query tile only).
 * Option "dense_graph" (1 by default): asynchronous int8 calls of one shape replay a captured graph (one launch per call).
 * Option "dense_mid_tier" (1 by default): queries the first filter cannot certify take a second, tighter filter over the
 * float32 rows (L2 and cosine, d <= 512, 16-byte aligned rows) before the exact all-rows path; after three calls
 * in a row in which most candidate lists overflowed, calls skip the first filter and start there (it is tried again after
 * 16 such calls, then 32, 64 ... 1024 while it keeps overflowing; the first probe that does not overflow re-arms it).  0: uncertified queries go straight to the exact path.  Answers are the same bits either way
 * (metrics.py:73-86, 120-137 arithmetic in the re-rank; every tier certifies or hands on). */
int sq_dense_sync(sq_handle_t h);
int sq_dense_destroy(sq_handle_t h);

/* Distances from one query to n gathered candidate rows, in the reference's
 * arithmetic: the `distances = list(map(comp_descr_dist, neighbor_vectors))`
 * step of lsh.py:511.  query: [d], rows: [n][d], both of `dtype`
 * (SQ_DTYPE_F32 / SQ_DTYPE_F64).  out: L2 -> [n] of the same dtype
 * (metrics.py:73-86 is dtype preserving); cosine -> float64[n]. */
int sq_dense_distances(const void* query, const void* rows, int dtype, int64_t n, int d,
                       int metric, void* out, int mem, void* stream);

/* The same hashing with the MODEL resident: mean (values as f64, `mean_dtype` = the model's dtype, see above) and
 * rotation [d][bits] f64 are uploaded once; sq_itq_model_hash then moves only the rows and the codes (small
 * batches through pinned staging).  What ItqFunctor.get_hash costs per query vector in
 * LSHNearestNeighborIndex._nn (impls/nn_index/lsh.py:473): without this every call re-uploads the rotation. */
int sq_itq_model_create(const double* mean, int mean_dtype, const double* rotation, int d, int bits,
                        int norm_ord, sq_handle_t* out);
int sq_itq_model_hash(sq_handle_t model, const void* x, int x_dtype, int64_t n, uint64_t* out_codes,
                      int mem, void* stream);
int sq_itq_model_destroy(sq_handle_t model);

/* ------------------------------------------------------- LSH re-rank stage
 * Replaces the tail of LSHNearestNeighborIndex._nn (impls/nn_index/lsh.py:
 * 499-519): fetch every candidate descriptor, one distance call per row
 * (lsh.py:511), stable sort by distance, first n (lsh.py:513-518).
 * sq_rows_create keeps the descriptor matrix ([n][d] float32 or float64,
 * row order chosen by the caller) on the device.  sq_rows_rerank takes, for
 * nq queries (same dtype as the rows, host memory), the concatenated candidate
 * row ids of all queries and their offsets [nq+1] (host memory), computes each
 * candidate's distance in the reference arithmetic (as sq_dense_distances) and
 * returns per query the k smallest in (distance, position in that query's
 * candidate list) order -- the order of the reference's stable sort.
 * out_dist: float32[nq][k] for float32 rows with SQ_METRIC_L2, float64[nq][k]
 * otherwise (+inf padding); out_pos: int64[nq][k] positions into the query's
 * candidate list (-1 padding). */
int sq_rows_create(const void* rows, int dtype, int64_t n, int d, int mem, sq_handle_t* out);
/* Append n_add rows (the handle's dtype, host or device memory) behind a matrix created from host memory; they
 * get the next row numbers.  The LSH index's update_index (impls/nn_index/lsh.py:331-383) then uploads only
 * the new descriptors instead of the whole set. */
int sq_rows_append(sq_handle_t h, const void* rows, int64_t n_add, int mem);
int sq_rows_rerank(sq_handle_t h, const void* queries, int nq, int metric,
                   const int64_t* cand_rows, const int64_t* cand_offsets, int k,
                   void* out_dist, int64_t* out_pos, void* stream);
/* The whole query path of LSHNearestNeighborIndex._nn (impls/nn_index/lsh.py:452-519) for a batch, on the device:
 * hash the query descriptors (ItqFunctor.get_hash, lsh.py:473) -> the n nearest hash codes (HashIndex.nn,
 * lsh.py:480-487) -> bucket expansion through the hash -> uuids store (lsh.py:489-501) -> distance per candidate row,
 * stable sort, first n (lsh.py:505-519).  Nothing but the queries goes up and the winners come down; between the
 * stages only two integers (candidates in total, longest list) visit the host.
 *   sq_rows_set_buckets: the store as a CSR map over the row matrix: csr_off[n_codes + 1] (code id -> first entry),
 *       csr_rows[csr_off[n_codes]] (row numbers, bucket by bucket, row order inside a bucket; every row at most once --
 *       rows no bucket lists, e.g. descriptors removed from the store, are never candidates); code id = row id of `hamming`
 *       (rank of the code in the sorted unique codes).  Host arrays are copied, device arrays borrowed.
 *   sq_lsh_query: queries [nq][d] in the rows' dtype; `itq` a resident model (sq_itq_model_create) whose codes have
 *       the width of `hamming`'s; n_codes_wanted = the n of HashIndex.nn; k_out = results per query.
 *       out_dist: float32[nq][k_out] for float32 rows with SQ_METRIC_L2, else float64[nq][k_out] (+inf padding);
 *       out_rows: int64[nq][k_out] row numbers of the winners (-1 padding).  Order: (distance, position in the
 *       candidate list) = the reference's stable sort.  mem: where queries / outputs live. */
int sq_rows_set_buckets(sq_handle_t rows, const int64_t* csr_off, int64_t n_codes, const int64_t* csr_rows, int mem);
int sq_lsh_query(sq_handle_t rows, sq_handle_t hamming, sq_handle_t itq, const void* queries, int nq,
                 int n_codes_wanted, int metric, int k_out, void* out_dist, int64_t* out_rows, int mem, void* stream);
int sq_rows_destroy(sq_handle_t h);

/* ----------------------------------------------------------------- merge
 * Host-side k-way merge of per-shard top-k lists after the RCCL all-gather
 * (BASELINE.json north_star; no reference counterpart: the reference is
 * single-process).  dist/idx: [nshards][nq][k_in] (host memory); entries with
 * idx < 0 are padding.  Order: (distance, id).  dist_dtype: 0 = float32,
 * 1 = float64, 2 = int32. */
int sq_merge_topk(const void* dist, const int64_t* idx, int dist_dtype,
                  int nshards, int nq, int k_in, int k_out,
                  void* out_dist, int64_t* out_idx);
/* The same merge reading each shard's [nq][k_in] block at a byte stride: the
 * all-gather can then deliver ids and distances of a shard in ONE buffer
 * ([ids int64 nq*k_in][dist nq*k_in] per shard) and the merge reads the pinned
 * receive buffer in place. */
int sq_merge_topk_strided(const void* dist, const int64_t* idx, int dist_dtype,
                          int nshards, int nq, int k_in, int k_out,
                          int64_t dist_shard_stride_bytes, int64_t idx_shard_stride_bytes,
                          void* out_dist, int64_t* out_idx);

#ifdef __cplusplus
}
#endif
#endif /* SMQTK_HIP_H */
