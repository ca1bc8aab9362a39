/*
 * ivfadc_hip.h -- C ABI of libivfadc_hip.so: the MI355X (gfx950) implementation of
 * IVFADC.jl's knn_search hot path.
 *
 * The reference (pure Julia, /root/reference) has no FFI; its boundary for this path
 * is Julia multiple dispatch.  Each entry point below names the reference interface it
 * replaces.  A Julia shim or the Python mirror (ivfadc.jl_amd/) binds
 * exactly these symbols via ccall / ctypes.
 *
 * Conventions
 *   - every function returns 0 (IVFADC_OK) or a non-zero ivfadc_status; no C++
 *     exception crosses the boundary; ivfadc_last_error() gives the message of the
 *     last failure on the calling thread.
 *   - the caller owns every host buffer; the library owns device memory.
 *   - matrices are Julia-native column-major: "d x n" means n contiguous vectors of d.
 *   - T = Float32, U = UInt8 (ksub <= 256), I = UInt32, distance = SqEuclidean for both
 *     the coarse and the residual quantizer (defaults.jl:6,8), NaiveQuantizer only.
 *   - one HIP stream per handle.  Threads: every entry point locks its handle, so calls on ONE handle from several
 *     threads are serialised (correct, not concurrent); calls on DIFFERENT handles -- an index and its views
 *     (ivfadc_clone_view), unrelated indexes -- run concurrently.  A mutator (push! / delete / set_lists ...) first
 *     waits for the calls running on the index's views and keeps new ones out until it is done; the first view search
 *     afterwards refuses (the index changed).  Destroying a handle while another thread is inside a call on it is
 *     the caller's error.  (The reference is single-threaded, index.jl:269; tests: test_threads_index_view_and_a_mutator.)
 *   - IVFADC_ERR_ASSERT marks the conditions the reference raises AssertionError for.
 *
 * Environment.  The library reads these variables and no others (csrc/ivfadc_hip.hip: env_knob); each is a switch a deployment may
 * need without a rebuild, none changes a result:
 *   IVFADC_EXACT_TABLES=1     reference-order f32 tables in every lane (= ivfadc_set_table_mode(h, 1) on every handle): validation
 *   IVFADC_COARSE_EXACT=1     exact coarse kernels only, no matrix-core filters (= ivfadc_set_coarse_mode(h, 1)): validation
 *   IVFADC_NO_PRUNE=1         scan every probed list (= ivfadc_set_pruning(h, 0)): the reference's own byte count
 *   IVFADC_NO_SMALLQ=1        no single-launch latency path for small batches
 *   IVFADC_NO_PIPELINE=1      ivfadc_search_batches keeps one batch in flight
 *   IVFADC_NO_ZERO_COPY=1     host-pointer entries stage through the library's pinned blocks even for known memory
 *   IVFADC_NO_STREAM_PROBE=1  no probing for streams that share a hardware queue (views, the batches entry)
 *   IVFADC_ABORT_TRACE=path   append a native backtrace to `path` when the process aborts inside the library (test infrastructure)
 * The A/B switches of closed experiments (tools/experiments/, HISTORY.md) exist only in the diagnostic build (-DIVFADC_DEBUG).
 */
#ifndef IVFADC_HIP_H
#define IVFADC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ivfadc_index ivfadc_t;

typedef enum {
    IVFADC_OK = 0,
    IVFADC_ERR_ASSERT = 1,      /* reference @assert would fail (k<1, w<1, wrong dim, capacity) */
    IVFADC_ERR_INVALID = 2,     /* argument the reference cannot express / library limit      */
    IVFADC_ERR_HIP = 3,         /* HIP runtime error (no device, out of memory, launch fail)  */
    IVFADC_ERR_STATE = 4        /* call not valid in the handle's state (e.g. no lists yet)   */
} ivfadc_status;

/* Bumped whenever an exported prototype changes in place or a struct grows (the round-4 change of ivfadc_set_next_queries, which gained
 * its token argument, was version 2 -> 3).  A binding built against another version must not call into this library: the Julia shim and
 * tests/c/abi_smoke.c compare ivfadc_abi_version() with the value they were written for before anything else.                           */
#define IVFADC_ABI_VERSION 4
int ivfadc_abi_version(void);

/* Reach of the selection kernels.  Beyond either (any K, any w <= kc) the library takes the generic path: every
 * (query, probed point) key is written out and sorted, the first K are the result -- same semantics, slower. */
#define IVFADC_MAX_K 2048
#define IVFADC_MAX_W 2048

/* Replaces: the data of IVFADCIndex.coarse_quantizer (NaiveQuantizer.vectors,
 * coarsequantizers.jl:18-20) and .residual_quantizer.codebooks (index.jl:39-48).
 *   centroids   d x kc column-major (centroid c at centroids + c*d)
 *   codebooks   m blocks; block i is codebooks[i].vectors, dsub x ksub column-major
 *   code_labels m x ksub: codebooks[i].codes (the byte that names codeword c of block i);
 *               labels of one block must be distinct (LittleDict keys, index.jl:235)
 * Requires d % m == 0 (QuantizedArrays.rowrange for other shapes is third-party and
 * unverifiable), 1 <= ksub <= 256, kc >= 1.                                            */
int ivfadc_create(ivfadc_t **out, int device, int d, int kc, int m, int ksub,
                  const float *centroids, const float *codebooks, const uint8_t *code_labels);

/* Replaces (statistically, not bit for bit: both are third-party and unseeded in the reference) the training half
 * of the IVFADCIndex constructor, index.jl:127-147: Clustering.kmeans(data, kc; init=:kmpp, maxiter) for the coarse
 * quantizer and QuantizedArrays.build_quantizer(residuals; k, m, method=:pq, maxiter), i.e. one k-means per sub-space.
 *   data d x n; out_centroids d x kc; out_codebooks m blocks of dsub x k (the layout ivfadc_create takes; labels are
 *   0..k-1).  Runs on the device: k-means++ seeding, exact-distance assignment (the search path's coarse kernel), and
 *   order-independent fixed-point sums, so the result is deterministic for a given seed.
 * Constructor checks of index.jl:118-123 (kc >= 2, k <= n, 1 <= m <= d, maxiter > 0) -> IVFADC_ERR_ASSERT.          */
int ivfadc_train(int device, int d, int64_t n, const float *data, int kc, int k, int m,
                 int coarse_maxiter, int quant_maxiter, uint64_t seed,
                 float *out_centroids, float *out_codebooks);

/* Replaces: IVFADCIndex.inverse_index (index.jl:8-11,23; built at index.jl:178-194).
 *   offsets kc+1 point offsets (list c = [offsets[c], offsets[c+1]))
 *   codes   n x m bytes, list order, m bytes per point (the order persistency.jl:74-76 writes)
 *   ids     n ids, 0-based as stored by the reference (index.jl:189)
 * Replaces any previous lists.  Every code byte must be a label of its block.           */
int ivfadc_set_lists(ivfadc_t *h, const int64_t *offsets, const uint8_t *codes, const uint32_t *ids);

/* Bench / test utility (no reference counterpart): fills the lists ON THE DEVICE with
 * counter-based pseudo-random code bytes (byte b of global position g is
 * mix64(seed + ((g*m+b)>>3)*0x9E3779B97F4A7C15) >> 8*((g*m+b)&7)), ids = global position.
 * The oracle regenerates any list from the same rule.  Requires ksub == 256 with identity
 * labels.  No host mirror is kept, so ivfadc_append / ivfadc_get_lists return ERR_STATE. */
int ivfadc_synth_lists(ivfadc_t *h, const int64_t *offsets, uint64_t seed);

/* Replaces: _encode_point (utils.jl:148-161): coarse_search(cq, point, 1) -> residual ->
 * quantize_data.  pts is d x n.  out_list: 0-based cluster per point; out_codes: n x m.
 * Does not modify the index.                                                             */
int ivfadc_encode(ivfadc_t *h, int64_t n, const float *pts, int32_t *out_list, uint8_t *out_codes);

/* Replaces: push!(ivfadc, point) (utils.jl:114, _push! :127-145) for a batch: encodes and
 * appends (ids[i], code_i) to the END of list out_list[i], in order i = 0..nnew-1.  The
 * caller chooses ids (push! uses id = length(ivfadc)); out_list / out_codes (may be NULL)
 * return the assignment so a host mirror stays coherent.  Device side: every list carries
 * spare capacity (max(32, len/8) points); when all target lists have room the codes and
 * ids are scattered in place (O(nnew)), otherwise the next search re-lays the lists out.  */
int ivfadc_append(ivfadc_t *h, int64_t nnew, const float *pts, const uint32_t *ids,
                  int32_t *out_list, uint8_t *out_codes);

/* Replaces: delete_from_index!(ivfadc, points) (utils.jl:90-105), and with a single id pop! / popfirst!
 * (utils.jl:29-68): removes the stored entries whose id is listed (0-based ids; unknown ids are ignored), keeps the
 * order of the survivors in every list, and lowers each surviving id by the number of removed ids below it
 * (_shift_inverse_index!, utils.jl:11-27).  Done in place on the device (one workgroup per list) and on the host
 * mirror.  out_removed (may be NULL): how many entries went.                                                  */
int ivfadc_delete_ids(ivfadc_t *h, int64_t ndel, const uint32_t *ids, int64_t *out_removed);

/* Replaces: _shift_up_inverse_index!(ivfadc, 1) of pushfirst! (utils.jl:1-9, 123): adds delta to every stored id. */
int ivfadc_shift_ids(ivfadc_t *h, int32_t delta);

/* Replaces: knn_search(ivfadc, points::Vector{Vector{T}}, k; w) (index.jl:261-273), i.e.
 * knn_search (index.jl:204-258) for every query.  queries is d x nq.
 *   out_ids / out_dists  K slots per query (query q at + q*K), ascending (distance, visit order)
 *   out_counts           neighbours found for query q (<= K; "at most k", index.jl:200)
 * K < 1 or w < 1 -> IVFADC_ERR_ASSERT (index.jl:210-211); w is clamped to kc (index.jl:216). */
int ivfadc_search(ivfadc_t *h, int64_t nq, const float *queries, int K, int w,
                  uint32_t *out_ids, float *out_dists, int32_t *out_counts);

/* Page-locked host memory for the host-pointer entries (ivfadc_search, ivfadc_search_batches, ivfadc_mg_search).
 * The reference's call takes and returns host arrays (index.jl:261-265: Vector{Vector{T}} in, Vector{Vector{I}} / Vector{Vector{T}} out),
 * so a binding packs the queries into one d x nq matrix and unpacks the results anyway.  When that matrix / those result arrays lie in
 * memory the GPU can address, the library needs no staging copy of its own:
 *   queries     are ingested from the caller's array by the first launch of the call (at most 64 queries on the latency path: read in
 *               place, no ingest at all);
 *   results     are written into the caller's out_ids / out_dists / out_counts by the final kernel of the search -- no device-side
 *               result block, no device-to-host copy (all three arrays must be known, otherwise the library's pinned block is used).
 * Arrays the library does not know are staged through its own pinned buffers as before; results are identical either way.
 *   ivfadc_host_alloc       page-locked memory owned by the library (any device may read / write it); free with ivfadc_host_free
 *   ivfadc_host_register    page-locks an array of the CALLER's (e.g. a Julia Matrix{Float32} that a serving loop refills): the caller
 *                           keeps it alive and unmoved until ivfadc_host_unregister(p).  Registering costs tens of microseconds per
 *                           megabyte: do it once per buffer, not per call.
 * Thread-safe (one process-wide table).  IVFADC_ERR_INVALID for a pointer the table does not hold / an overlapping registration. */
int ivfadc_host_alloc(size_t bytes, void **out);
int ivfadc_host_free(void *p);
int ivfadc_host_register(void *p, size_t bytes);
int ivfadc_host_unregister(void *p);

/* Where the host time of the host-pointer entries went (cumulative since creation / the last reset; microseconds of wall time on the
 * calling thread): stage_in = copies of unregistered queries into pinned memory, enqueue = issuing ingest + kernels, wait = polling for
 * completion, stage_out = copies of results out of the library's pinned block; and how many calls found their queries / results in
 * known memory.  ivfadc_reset_host_stats zeroes them.                                                                         */
typedef struct {
    double  stage_in_us, enqueue_us, wait_us, stage_out_us;
    int64_t calls;             /* ivfadc_search + ivfadc_search_batches calls */
    int64_t batches;           /* batches searched by them */
    int64_t queries_direct;    /* calls whose queries were read from the caller's own (known) array */
    int64_t results_direct;    /* calls whose results were written straight into the caller's arrays */
    int64_t zero_copy;         /* calls whose queries were read in place by the search kernels (latency path) */
    int64_t streams_replaced;  /* since creation: streams of this handle's views / copy lane that were found to share a hardware queue
                                * with a stream they must run beside, and were replaced (probe at ivfadc_clone_view / first batch run) */
} ivfadc_host_stats;
int ivfadc_get_host_stats(ivfadc_t *h, ivfadc_host_stats *out);
int ivfadc_reset_host_stats(ivfadc_t *h);

/* A read-only VIEW of an index: a second handle on the SAME device arrays (quantizers, derived tables, inverted lists -- nothing is copied)
 * with a stream and a workspace of its own, so that two batches can be in flight on one replica: searches on h and on the view overlap
 * on the device.  (The reference's knn_search is a pure function of the index, index.jl:261-273: concurrent searches are independent.)
 * A view cannot change anything (push!/delete/set_lists/save return IVFADC_ERR_STATE) and keeps no host mirror.  Any change to h
 * afterwards (ivfadc_append, ivfadc_delete_ids, ivfadc_shift_ids, ivfadc_set_lists, ivfadc_synth_lists) makes every view of it refuse to
 * search (IVFADC_ERR_STATE: take a new one); a mutator first waits for the searches still in flight on h's views (their streams).
 * Settings (pruning, tuning, table mode, ...) are copied when the view is taken and can be set on it separately afterwards.
 * Destroy views with ivfadc_destroy, before or after h (a view that outlives h refuses to search).                                  */
int ivfadc_clone_view(ivfadc_t *h, ivfadc_t **out_view);

/* Same, with every buffer already resident in device memory of the handle's GPU.
 * Asynchronous on the handle's stream; pair with ivfadc_sync().                          */
int ivfadc_search_device(ivfadc_t *h, int64_t nq, const float *d_queries, int K, int w,
                         uint32_t *d_out_ids, float *d_out_dists, int32_t *d_out_counts);

/* Replaces: a loop of knn_search(ivfadc, points, k; w) calls over consecutive batches (index.jl:261-273 once per batch) -- ONE call for
 * the whole run.  batch_nq[b] queries per batch; queries is d x sum(batch_nq), the batches back to back; outputs are laid out like
 * ivfadc_search's for the concatenated queries (K slots per query).  Every batch's results are exactly what ivfadc_search returns
 * for it.  Inside, batch b is searched with batch b + 1 named as its successor (ivfadc_set_next_queries below, on device buffers and
 * tokens the library owns), so plans with the rider form run one launch per batch.  From two batches on, TWO are in flight: the odd
 * ones run on an internal view of the handle (ivfadc_clone_view: second stream, second workspace, same device arrays), each lane naming
 * its own next batch as the successor -- a launch's ramp and tail leave the chip half empty, a second stream fills them.  Results are
 * unchanged bit for bit; IVFADC_NO_PIPELINE=1 (environment) keeps one batch in flight, and so does profiling (ivfadc_set_profiling: the
 * kernel timings are this handle's).  The cumulative counters of ivfadc_get_stats include the view's share.                              */
int ivfadc_search_batches(ivfadc_t *h, int nbatches, const int64_t *batch_nq, const float *queries, int K, int w,
                          uint32_t *out_ids, float *out_dists, int32_t *out_counts);

int ivfadc_sync(ivfadc_t *h);

/* Single-process multi-device front end: one replica of the index per listed device, contiguous
 * query blocks per device, every device's block in flight at once; results come back in query order.  The
 * one-process-per-GPU form (torch.distributed / RCCL all-gather of the packed top-k) is ivfadc.jl_amd/distributed.py.
 * Same argument meaning and status codes as the single-device calls.                                         */
typedef struct ivfadc_mg ivfadc_mg_t;
int ivfadc_mg_create(ivfadc_mg_t **out, int ndev, const int *devices, int d, int kc, int m, int ksub,
                     const float *centroids, const float *codebooks, const uint8_t *code_labels);
int ivfadc_mg_set_lists(ivfadc_mg_t *g, const int64_t *offsets, const uint8_t *codes, const uint32_t *ids);   /* replicas upload concurrently */
/* every replica synthesises the same lists on its own device (see ivfadc_synth_lists): how the billion-point
 * benchmark shapes are loaded into the single-process front end                                              */
int ivfadc_mg_synth_lists(ivfadc_mg_t *g, const int64_t *offsets, uint64_t seed);
int ivfadc_mg_num_devices(ivfadc_mg_t *g);
/* Result merge of ivfadc_mg_search.  0 (default): each device's block is copied to the caller's arrays as it
 * completes.  1: the final top-k merge of the batch is ONE ncclAllGather (RCCL over xGMI; ncclCommInitAll, one
 * stream per device) of the packed per-device blocks [ids | dists | counts] -- no reduction: devices own disjoint
 * queries (index.jl:269-271) -- after which every device holds the batch's results and the host reads device
 * 0's copy.  Needs distinct devices; RCCL is bound at run time (dlopen) and IVFADC_ERR_STATE is returned if absent. */
int ivfadc_mg_set_gather(ivfadc_mg_t *g, int mode);
int ivfadc_mg_collectives(ivfadc_mg_t *g, int64_t *out);   /* all-gathers issued so far (mode 1) */
int ivfadc_mg_append(ivfadc_mg_t *g, int64_t nnew, const float *pts, const uint32_t *ids,
                     int32_t *out_list, uint8_t *out_codes);
int ivfadc_mg_delete_ids(ivfadc_mg_t *g, int64_t ndel, const uint32_t *ids, int64_t *out_removed);   /* every replica */
int ivfadc_mg_shift_ids(ivfadc_mg_t *g, int32_t delta);
int ivfadc_mg_search(ivfadc_mg_t *g, int64_t nq, const float *queries, int K, int w,
                     uint32_t *out_ids, float *out_dists, int32_t *out_counts);
void ivfadc_mg_destroy(ivfadc_mg_t *g);

/* One process per GPU: the final top-k merge of a batch inside the library.  Every rank holds a replica of the
 * index and a contiguous block of the batch's queries (index.jl:269-271: queries are independent).
 *   ivfadc_comm_unique_id   rank 0 creates the 128-byte RCCL id; the host framework carries it to the other ranks
 *   ivfadc_comm_init        ncclCommInitRank on the handle's device (collective: every rank calls it)
 *   ivfadc_search_device_allgather
 *                           ivfadc_search_device of this rank's nq queries into d_block -- packed [ids nq*K | dists nq*K |
 *                           counts nq] int32 -- followed by ONE ncclAllGather of that block into d_gathered (nranks blocks, rank
 *                           order) on a side stream of the handle: the collective overlaps the next batch's kernels.  Every
 *                           rank passes the same nq and K within one call (the collective's contract: equal blocks; a single rank
 *                           cannot detect a mismatch); the block size may change from call to call if every rank changes it alike.
 *                           d_gathered holds nranks x nq x (2K + 1) words.  slot in [0, 8) names the buffer pair; a
 *                           slot's previous collective is waited for on the device before the slot is written again -- rotate all
 *                           eight slots: one wait then serves several steps (collectives finish in issue order, and a wait in the
 *                           search stream costs the next kernel ~6 us of idle queue whether its event has fired or not).
 *   ivfadc_comm_wait        the search stream waits for every collective issued so far; out_collectives (may be NULL) counts them
 * RCCL is bound at run time (dlopen); IVFADC_ERR_STATE if it is absent or the communicator has not been set up.      */
int ivfadc_comm_unique_id(uint8_t *out_id128);
int ivfadc_comm_init(ivfadc_t *h, int nranks, int rank, const uint8_t *id128);
int ivfadc_search_device_allgather(ivfadc_t *h, int64_t nq, const float *d_queries, int K, int w,
                                   int32_t *d_block, int32_t *d_gathered, int slot);
/* The same with the search on `searcher` -- h itself or a view of it (ivfadc_clone_view: two batches in flight per rank) -- and the
 * collective where the communicator lives, on h's side stream, in call order.                                        */
int ivfadc_search_device_allgather_on(ivfadc_t *h, ivfadc_t *searcher, int64_t nq, const float *d_queries, int K, int w,
                                      int32_t *d_block, int32_t *d_gathered, int slot);
int ivfadc_comm_wait(ivfadc_t *h, int64_t *out_collectives);
int ivfadc_comm_destroy(ivfadc_t *h);

/* List-partitioned multi-GPU mode: strong scaling of a FIXED global batch.  Query shards over replicas stop scaling once a rank streams
 * most of the index for its few queries; here every rank keeps the replica and ALL nq queries, runs the coarse search (identical on every
 * rank: coarsequantizers.jl:33-37), scans only the probed lists l with l % nparts == part -- the probes of a query are independent given
 * the bound, index.jl:228-255 -- and leaves per query the K smallest KEYS of its lists (distance bits << 32 | visit order: visit orders are
 * global, so keys of different ranks compare and name one stored point each).  The index is still replicated and RCCL still carries only the
 * final top-k merge: ONE all-gather of nq x K keys per rank, then the K-way merge on every rank.
 *   ivfadc_set_list_partition      nparts = 1 switches the mode off.  While it is on, searches take the list-major plan, the batch must fit
 *                                  one sub-batch, K <= 2048 and w <= 2048 (IVFADC_ERR_INVALID otherwise); a plain ivfadc_search_device then
 *                                  returns the top-K of this rank's lists alone
 *   ivfadc_search_device_partial   d_keys nq x K (ascending per query), d_counts nq
 *   ivfadc_merge_partials_device   d_keys_all nparts x nq x K, d_counts_all nparts x nq -> ids, distances, counts of the whole scan; on the
 *                                  handle that ran the batch's partial search (its probe arrays translate visit orders into ids)
 *   ivfadc_search_device_listpart  the three steps in one call: partial search into d_block ([keys | counts], ivfadc_listpart_block_words
 *                                  int32 words), ncclAllGather into d_gathered (nranks blocks), merge.  Needs ivfadc_comm_init and
 *                                  ivfadc_set_list_partition(h, nranks, rank); every rank passes the same nq queries, K and w           */
int ivfadc_set_list_partition(ivfadc_t *h, int nparts, int part);
int ivfadc_search_device_partial(ivfadc_t *h, int64_t nq, const float *d_queries, int K, int w, uint64_t *d_keys, int32_t *d_counts);
int ivfadc_merge_partials_device(ivfadc_t *h, int64_t nq, int K, int nparts, const uint64_t *d_keys_all, const int32_t *d_counts_all,
                                 uint32_t *d_ids, float *d_dists, int32_t *d_counts);
int64_t ivfadc_listpart_block_words(int64_t nq, int K);
int ivfadc_search_device_listpart(ivfadc_t *h, int64_t nq, const float *d_queries, int K, int w, int32_t *d_block, int32_t *d_gathered,
                                  uint32_t *d_ids, float *d_dists, int32_t *d_counts);

/* Run on a caller-owned hipStream_t (e.g. the host framework's current stream) instead of the
 * handle's own stream, so searches order naturally with the caller's kernels and collectives. */
int ivfadc_set_stream(ivfadc_t *h, void *hip_stream);

/* Replaces: length(ivfadc) (index.jl:56) and the per-list lengths. list_sizes: kc entries or NULL. */
int ivfadc_ntotal(ivfadc_t *h, int64_t *out_n, int64_t *list_sizes);

/* Copies the host mirror of the lists back out (layout of ivfadc_set_lists).            */
int ivfadc_get_lists(ivfadc_t *h, int64_t *offsets, uint8_t *codes, uint32_t *ids);

/* The quantizers of a handle (what ivfadc_create took, or what ivfadc_load_index read: src/index.jl:39-48): dimensions first
 * (any pointer may be NULL), then the arrays in ivfadc_create's layout.                                               */
int ivfadc_get_dims(ivfadc_t *h, int *d, int *kc, int *m, int *ksub);
int ivfadc_get_quantizers(ivfadc_t *h, float *centroids /* kc x d */, float *codebooks /* m x ksub x dsub */, uint8_t *code_labels /* m x ksub */);

/* Replaces: save_ivfadc_index(filename, ivfadc) (persistency.jl:1-78) for a NaiveQuantizer / UInt8 / Float32 index:
 * byte for byte the reference's file (rotation matrix: identity, or the one the index was loaded with).  index_bits = width
 * of the reference's index type I (8, 16 or 32).                                                                 */
int ivfadc_save_index(ivfadc_t *h, const char *path, int index_bits);

/* Replaces: load_ivfadc_index(filename) (persistency.jl:82-134): reads a file written by IVFADC.jl (NaiveQuantizer,
 * U = UInt8, I <= 32 bits; Float64 values are narrowed) into a new handle, lists included.
 * out_index_bits (may be NULL) returns the width of I.                                                          */
int ivfadc_load_index(ivfadc_t **out, int device, const char *path, int *out_index_bits);

/* The rotation of the residual quantizer (QuantizedArrays' `rot`, persistency.jl:62-64, 110-117).  knn_search never reads it
 * (index.jl:204-258), so an index built with :opq is SEARCHED like any other: ivfadc_load_index keeps its matrix (and
 * ivfadc_save_index writes it back).  quantize_data -- push! (utils.jl:148-161) -- does read it, through third-party arithmetic
 * that cannot be verified here: ivfadc_encode / ivfadc_append return IVFADC_ERR_STATE on such a handle (delete / pop / shift are
 * fine: they touch ids only).  out_rotated: 0 = identity (:pq); out_rot (may be NULL): d x d floats, column by column.     */
int ivfadc_get_rotation(ivfadc_t *h, int *out_rotated, float *out_rot);

/* Measurement.  When profiling is on, every scan-kernel launch is bracketed by HIP events
 * on the handle's stream; ivfadc_get_stats synchronises and reports the totals since the
 * last ivfadc_reset_stats.                                                               */
typedef struct {
    double   scan_ms;          /* sum of scan-kernel durations (HIP events)               */
    double   coarse_ms;        /* sum of coarse-distance kernel durations                 */
    int64_t  scan_launches;
    int64_t  scanned_points;   /* sum over (query, probe) of len(list): x m = B_alg bytes  */
    int64_t  queries;
    int32_t  last_qg;          /* queries sharing one code stream in the last launch (0 = query-major kernel) */
    int32_t  last_chunk;       /* points per work item in the last launch                  */
    int32_t  last_scan_grid;
    int32_t  last_scan_lds;
    int64_t  coarse_fallbacks; /* queries whose MFMA-filter certificate failed (exact recompute taken)  */
    int32_t  coarse_mfma;      /* 1: the last batch used the MFMA filter + certified exact refine        */
    int32_t  inplace_appends;  /* ivfadc_append calls since creation that were written in place on the device (no re-layout) */
    int32_t  last_striped;     /* 1: the last list-major launch used bank-striped tables + rotated-order filter sums */
    int32_t  coarse_listed;    /* 1: the last batch's coarse filter wrote per-tile records (four smallest keys of every
                                * (query, 64-centroid tile)) instead of the score matrix, and the top-w enumerated them */
    int64_t  pruned_points;    /* points of probed lists that were NOT scanned: the list's coarse distance already lay above
                                * the K-th best key (ivfadc_set_pruning).  scanned_points counts every probed list, as
                                * SURVEY.md 8(d) defines B_alg; scanned_points - pruned_points were actually read        */
    int64_t  lb_survivors;     /* points whose 8-bit lower-bound sum passed the filter of the matrix-core table rounds and got
                                * their reference-order sum from the f32 codebook (ivfadc_set_table_mode)  */
    int32_t  last_lb;          /* 1: the last query-major launch built its ADC tables on the matrix cores (lower bounds) */
    int32_t  last_rider;       /* 1: the last query-major scan launch also carried the exact coarse tiles of a hinted next batch
                                * (ivfadc_set_next_queries) */
    double   lb_build_ms;      /* profiling level 2 only: sum of the durations of the table build run ALONE (an extra launch per batch
                                * of the same build code over the same probes; results are unaffected)                      */
    int64_t  lb_build_launches;
    int32_t  coarse_prefetched; /* 1: the last search found its coarse distances already computed (by the riders of the search before it) */
    int32_t  last_nf;          /* 1: the last list-major launch was the narrow-field kernel (4-bit integer filter, eight queries per code
                                * stream, no resident f32 tables: nfscan.hip.h); lb_survivors then counts the (point, query) pairs that got
                                * their reference-order sum from the f32 codebook */
    int32_t  last_twolevel;    /* 1: the last batch's coarse stage was the certified two-level search (ivfadc_set_coarse_mode) */
    int32_t  twolevel_groups;  /* groups the centroids are in when that search is in use (0: it is not) */
    int64_t  coarse_visited;   /* two-level search: exact centroid distances computed (of queries x kc an exhaustive search computes) */
    float    twolevel_probe_fraction;   /* the self-probe's visited fraction when the grouping was built (-1: not built) */
    int32_t  coarse_f16;       /* 1: the last batch's matrix-core coarse filter used ONE f16 product per score (round 5) instead of the
                                * three-product bf16 split (ivfadc_set_coarse_mode(h, 8) asks for the split) */
} ivfadc_stats;

/* on: 0 off, 1 events around the coarse and scan kernels, 2 = 1 + the matrix-core table build timed alone (lb_build_ms) */
int ivfadc_set_profiling(ivfadc_t *h, int on);
int ivfadc_reset_stats(ivfadc_t *h);
int ivfadc_get_stats(ivfadc_t *h, ivfadc_stats *out);

/* Tuning knobs (0 = automatic).  qg: -1 forces the query-major scan kernel (one workgroup per
 * query), 1 / 2 / 4 force the list-major kernel with that many queries per code stream, 8 the narrow-field list-major kernel (eight
 * queries per code stream; m = 8, dsub = 16, ksub = 256, K <= 64 -- other shapes fall back to 4), -2 forces the generic
 * dump-and-sort path (an independent second implementation, used as a cross-check in the tests), -3 the query-major
 * kernel behind the stand-alone top-w selection whatever the batch size (what large batches take: tests);
 * chunk_points: points per list-major work item.  Results never depend on these.            */
int ivfadc_set_tuning(ivfadc_t *h, int qg, int chunk_points);

/* Coarse search implementation: 0 = automatic (matrix-core score filter + certified exact refine when w <= 48,
 * kc >= 2048 and d % 4 == 0 -- split-bf16 MFMA when the problem fills the chip with 128 x 128 tiles, f32 MFMA below
 * that; the 3-op VALU kernel otherwise), 1 = always the exact VALU kernel, 2 = the filter from kc >= 128 on (tests),
 * 3 = as 0 with the f32 MFMA filter only (A/B runs), 4 = as 0 but the split-bf16 filter always writes the whole score
 * matrix (no per-tile records: A/B runs and tests), 5 = as 0, and the small-batch path (at most 64 queries, nq x w <= 512, K and
 * w <= 64: ONE launch, a workgroup per (query, probe, chunk), last-arriver merge) also searches a coarse quantizer of at most 2048
 * cells inside that launch instead of running the exact coarse kernel first (measured slower: the default keeps the separate
 * kernel).  8 = as 0 with the three-product bf16 split in the large matrix-core filter instead of the one-product f16 form (A/B runs,
 * tests; the f16 form -- scaled f16 operands, 2^-11 relative score error, queries that leave the f16 range flagged and recomputed
 * exactly -- is the default where the bf16 kernel ran before).
 * 6 / 7 = the CERTIFIED TWO-LEVEL coarse search always / never (w <= 64, d % 4 == 0).  What the reference reaches for when the
 * quantizer is large is an HNSW graph (coarsequantizers.jl:58-92: approximate); this is the exact counterpart: the centroids are grouped
 * once (k-means over the centroids, kc / 64 groups, a radius per group), a query visits the groups in ascending order of the lower bound
 * max(0, ||q - g|| - r_g)^2 on its members' distances, computes the members' distances in the reference's order and stops at the first
 * group whose bound exceeds the w-th best distance found.  In automatic mode (0) the grouping is built on the first search of a
 * quantizer with kc >= 4096 and kept only if a self-probe (centroids as queries) computes at most 2 % of the kc distances (a visited
 * centroid is read once per query, not once per query tile: the break-even) -- a quantizer trained on clustered data: 0.1 %, three and a
 * half times faster than the matrix-core filter at kc = 65 536; N(0,1) centroids in high dimension: all of them, so the exhaustive kernels stay
 * (stats: last_twolevel, coarse_visited, twolevel_probe_fraction).  Results are identical in every mode (whatever a filter or a bound
 * lets through is recomputed in the reference's order, and what a bound skips cannot be in the result, ties included).              */
int ivfadc_set_coarse_mode(ivfadc_t *h, int mode);

/* Probe pruning in the query-major scan (default on): a point's ADC sum starts from its list's coarse distance and every
 * table entry is >= 0 (index.jl:242-244), so once the K-th best key found so far lies below the coarse distance of the next
 * probe -- probes come in ascending coarse distance (coarsequantizers.jl:35-36) -- no later list can contribute and the
 * query ends.  Exact: ids and distances are those of the full scan.  0 = scan every probed list.             */
int ivfadc_set_pruning(ivfadc_t *h, int on);

/* Hint: the search AFTER the next one on this handle (ivfadc_search_device / ivfadc_search_device_allgather) will search the nq
 * queries at d_queries (device memory, already holding them now and unchanged until that search has run).  A query-major scan launch
 * leaves CUs idle while its last workgroups finish; with the hint, the search made right after this call also computes the hinted
 * batch's exact coarse distances (coarsequantizers.jl:34, the same kernel code, tile by tile) behind its own scan in the same grid,
 * and the hinted search then starts at its top-w selection.
 *   token   the caller's generation number of the buffer's CONTENTS (any value != 0 that the caller changes whenever it refills the
 *           buffer).  The rows are picked up only by the very next search, only if that search's pointer and count equal the hint's,
 *           AND only if the caller declared the same token for it with ivfadc_set_query_token: a staging buffer that was refilled after
 *           the hint carries another token and its search computes its own rows.  Without a declared token (the default) no search
 *           ever picks prefetched rows up.
 * One hint serves one search; a search on other queries, on any other path (small batch, generic, sub-batched), a plan without the
 * rider form, or no further search simply leaves the rows unused.  Results are unchanged bit for bit (stats: last_rider,
 * coarse_prefetched).  nq = 0, d_queries = NULL or token = 0 withdraws the hint.
 * ivfadc_search_batches (below) drives both calls for host-resident batches, which is how the Julia shim reaches this path. */
int ivfadc_set_next_queries(ivfadc_t *h, int64_t nq, const float *d_queries, uint64_t token);
/* Declares the generation of the queries the NEXT search on this handle reads (see above); consumed by that search. */
int ivfadc_set_query_token(ivfadc_t *h, uint64_t token);

/* ADC tables: 0 = automatic -- list-major scan: bank-striped tables with rotated-order sums (m = 16) or 16-bit integer tables
 * (m = 8) as a filter where those forms exist (four queries per code stream), or 4-bit narrow-field tables with eight
 * queries per code stream and no resident f32 tables (m = 8, dsub = 16: nfscan.hip.h); query-major scan: 8-bit lower-bound
 * tables built on the matrix cores where that pays (m = 48, K <= 64, w <= 32: DESIGN.md 4.4).  1 = the reference's f32 tables and
 * sum order in every lane (round-1 kernels; A/B runs and an independent cross-check in the tests).  2 = as 0, and the matrix-core
 * rounds for every shape they are instantiated for (also m = 16 / dsub = 6, where they are slower than the exact tables:
 * measurement and tests).  3 / 4 = as 0 / 2 with the matrix-core tables built from the three-product bf16 split of rounds 3-4 instead
 * of ONE f16 product per entry (round 5: power-of-two-scaled f16 operands, half the codeword bytes; the bound is looser by up to half a
 * table unit per entry, which the survivors' exact sums absorb).  5 / 6 = as 0 with the eight-wave list-major kernel (m = 8, dsub = 16,
 * K <= 64: wg8scan.hip.h) never / wherever it is instantiated; 7 = as 6 with its eight-query form (wg8q8scan.hip.h).  Results are identical in every mode: whatever a filter lets through is recomputed in the reference's
 * order -- from the f32 tables or, in the matrix-core rounds, from the f32 codebook -- before it meets the bound.       */
int ivfadc_set_table_mode(ivfadc_t *h, int mode);

/* Test hook of the matrix-core table build: the 8-bit lower-bound table of ONE (query, cell) pair, built by
 * the code path the search uses (src/index.jl:232-236 is what it bounds).  out_table: m x 256 bytes (slot = code label);
 * out_consts: 3 + 2 m floats -- scale inv, sum of bases, sum of norms, base[m], ||residual_ii||^2[m].  The contract the tests
 * check: base[ii] + q / inv <= (reference entry)(1 + (dsub + 2) 2^-24) for every entry, and <= base[ii] + (q + 1) / inv + 2^-13.4 N
 * from above.  IVFADC_ERR_STATE when the handle's shape has no such kernels.                                  */
int ivfadc_debug_lb_table(ivfadc_t *h, const float *query, int cell, uint8_t *out_table, float *out_consts);

/* Upper bound of the per-batch device workspace (default 8 GiB).  Larger batches are processed in
 * sub-batches of queries; results never depend on it.                                               */
int ivfadc_set_workspace_limit(ivfadc_t *h, uint64_t bytes);

void ivfadc_destroy(ivfadc_t *h);

const char *ivfadc_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* IVFADC_HIP_H */
