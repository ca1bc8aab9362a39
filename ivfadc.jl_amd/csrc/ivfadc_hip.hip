// ivfadc_hip.hip -- host runtime behind the C ABI of include/ivfadc_hip.h.
//
// Replaces, for the knn_search hot path, the Julia functions
//   knn_search            /root/reference/src/index.jl:204-273
//   coarse_search & co.   /root/reference/src/coarsequantizers.jl:33-48
//   _encode_point/_push!  /root/reference/src/utils.jl:127-161
// There is no CPU fallback: without a HIP device every compute entry point fails.
#include "../../include/ivfadc_hip.h"
#include "kernels.hip.h"
#include "train.hip.h"
#include "generic.hip.h"
#include "twolevel.hip.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/stat.h>
#include <unistd.h>
#include <rccl/rccl.h>   // types and prototypes only: the library is bound with dlopen (see rccl_api)

#include <rocprim/device/device_segmented_radix_sort.hpp>   // generic path only (after <cstring>: its headers use memset)

// Diagnostic (off unless IVFADC_ABORT_TRACE=<file> is set when the library is loaded; tests/conftest.py sets it): a process that dies in
// abort() -- the HIP runtime after a queue error, glibc after a heap check, an escaped exception -- under a test runner that captures fd 2
// leaves nothing behind but the signal.  The handler appends the native backtrace of the aborting thread to the file, restores the default
// action and returns into the abort.  Async-signal-safe calls only (open / write / backtrace_symbols_fd).
namespace {
char g_abort_trace_path[512];
struct sigaction g_abort_prev;   // whoever handled SIGABRT before (Python's faulthandler under pytest): called afterwards
void abort_trace_handler(int sig)
{
    const int fd = open(g_abort_trace_path, O_WRONLY | O_CREAT | O_APPEND, 0644);
    if (fd >= 0) {
        static const char head[] = "---- SIGABRT: native backtrace of the aborting thread (libivfadc_hip abort trace) ----\n";
        (void)!write(fd, head, sizeof head - 1);
        void *frames[64];
        const int n = backtrace(frames, 64);
        backtrace_symbols_fd(frames, n, fd);
        close(fd);
    }
    sigaction(sig, &g_abort_prev, nullptr);
    if (g_abort_prev.sa_handler != SIG_DFL && g_abort_prev.sa_handler != SIG_IGN && !(g_abort_prev.sa_flags & SA_SIGINFO))
        g_abort_prev.sa_handler(sig);
}
struct AbortTraceInit {
    AbortTraceInit()
    {
        const char *p = getenv("IVFADC_ABORT_TRACE");
        if (!p || !*p || strlen(p) >= sizeof g_abort_trace_path) return;
        strcpy(g_abort_trace_path, p);
        void *warm[4];
        (void)backtrace(warm, 4);   // loads libgcc's unwinder now, not inside the handler
        struct sigaction sa;
        memset(&sa, 0, sizeof sa);
        sa.sa_handler = abort_trace_handler;
        sigaction(SIGABRT, &sa, &g_abort_prev);
    }
} g_abort_trace_init;
}   // namespace

using namespace ivf;

static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(IVFADC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// No C++ exception crosses the C ABI: every extern "C" entry point is a function-try-block ending in IVF_CATCH
// (std::vector growth on hostile sizes, std::bad_alloc, ...).
static int on_exception() noexcept
{
    try { throw; }
    catch (const std::bad_alloc &) { return fail(IVFADC_ERR_INVALID, "out of host memory"); }
    catch (const std::exception &e) { return fail(IVFADC_ERR_INVALID, "C++ exception: %s", e.what()); }
    catch (...) { return fail(IVFADC_ERR_INVALID, "unknown C++ exception"); }
}
#define IVF_CATCH catch (...) { return on_exception(); }

#define TRY(expr)                  \
    do {                           \
        int rc_ = (expr);          \
        if (rc_ != IVFADC_OK) return rc_; \
    } while (0)

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    bool owned = true;   // false: an alias of another handle's buffer (ivfadc_clone_view) -- never freed, never grown
    int ensure(size_t need)
    {
        if (need <= bytes) return IVFADC_OK;
        if (!owned) return fail(IVFADC_ERR_STATE, "a view cannot grow a buffer of the index it was taken from");
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        size_t want = need + need / 4;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            e = hipMalloc(&p, need);
            want = need;
        }
        if (e != hipSuccess) { p = nullptr; return fail(IVFADC_ERR_HIP, "hipMalloc(%zu) failed: %s", need, hipGetErrorString(e)); }
        bytes = want;
        return IVFADC_OK;
    }
    void release()
    {
        if (p && owned) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        owned = true;
    }
    void forget() { p = nullptr; bytes = 0; owned = true; }   // (a copied struct: the memory belongs to the original)
    void alias() { owned = false; }
    template <class T> T *as() const { return (T *)p; }
};

// Page-locked host ranges the library knows to be readable and writable by kernels at their host address: what ivfadc_host_alloc
// handed out and what the caller pinned with ivfadc_host_register.  A host-pointer search whose arrays lie inside one skips the
// staging copy for that array (queries are ingested from it directly, results are written into it by the final kernel).
struct HostRange { uintptr_t lo, hi; bool owned; };
std::mutex g_host_mu;
std::vector<HostRange> g_host_ranges;

bool host_known(const void *p, size_t bytes)
{
    if (!p) return false;
    const uintptr_t a = (uintptr_t)p, b = a + bytes;
    std::lock_guard<std::mutex> lk(g_host_mu);
    for (const HostRange &r : g_host_ranges)
        if (a >= r.lo && b <= r.hi) return true;
    return false;
}

struct PinnedBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need)
    {
        if (need <= bytes) return IVFADC_OK;
        if (p) { (void)hipHostFree(p); p = nullptr; bytes = 0; }
        const size_t want = need + need / 4;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocPortable | hipHostMallocMapped);
        if (e != hipSuccess) { p = nullptr; return fail(IVFADC_ERR_HIP, "hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e)); }
        bytes = want;
        return IVFADC_OK;
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
    }
    void forget() { p = nullptr; bytes = 0; }
};

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
// ---- host <-> device copies of PAGEABLE memory go through the library's own page-locked bounce buffers ---------------------------------
// hipMemcpy{,Async} stages pageable copies of up to 1 MB itself; beyond that the runtime page-locks the CALLER's pages, lets the copy
// engine read or write them in place ("HSA Copy Using Pinned resource", rocblit.cpp; tools/pin_probe.py), and keeps the locked mapping in
// a cache keyed by address.  The callers here are short-lived std::vectors and numpy arrays on the malloc heap: freed, trimmed, handed out
// again at the same address -- and one run in four of the `-m gpu` suite died with "Memory access fault by GPU ... on address <a heap
// page>", the host inside ivfadc_set_lists uploading a 1-2 MB staging vector (round 5; found with the abort trace below and
// pytest --capture=sys).  So nothing pageable larger than BOUNCE_DIRECT_MAX is handed to the runtime: it is copied in 4 MB pieces through
// two page-locked buffers of the library's own (one process-wide pair, a mutex around a copy; the memcpy of piece i + 1 overlaps the
// transfer of piece i).  Both calls return when the data has arrived (they synchronise `s`).
constexpr size_t BOUNCE_DIRECT_MAX = (size_t)256 << 10;
constexpr size_t BOUNCE_PIECE = (size_t)4 << 20;
struct Bounce {
    std::mutex mu;
    void *buf[2] = {nullptr, nullptr};
    int ensure()
    {
        for (int i = 0; i < 2; ++i)
            if (!buf[i]) {
                hipError_t e = hipHostMalloc(&buf[i], BOUNCE_PIECE, hipHostMallocPortable | hipHostMallocMapped);
                if (e != hipSuccess) { buf[i] = nullptr; return fail(IVFADC_ERR_HIP, "hipHostMalloc(bounce buffer) failed: %s", hipGetErrorString(e)); }
            }
        return IVFADC_OK;
    }
};
Bounce g_bounce;

int h2d_copy(void *dst, const void *src, size_t bytes, hipStream_t s)
{
    if (bytes == 0) return IVFADC_OK;
    if (bytes <= BOUNCE_DIRECT_MAX) {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
        return IVFADC_OK;
    }
    std::lock_guard<std::mutex> lk(g_bounce.mu);
    TRY(g_bounce.ensure());
    int i = 0;
    for (size_t off = 0; off < bytes; off += BOUNCE_PIECE, i ^= 1) {
        const size_t n = std::min(BOUNCE_PIECE, bytes - off);
        memcpy(g_bounce.buf[i], (const char *)src + off, n);                 // (buf[i]'s previous transfer was waited for below)
        if (off) HIP_TRY(hipStreamSynchronize(s));                           // the transfer out of the OTHER buffer: it overlapped this memcpy
        HIP_TRY(hipMemcpyAsync((char *)dst + off, g_bounce.buf[i], n, hipMemcpyHostToDevice, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    return IVFADC_OK;
}

// (for code that threads a hipError_t through a chain of uploads: ivfadc_create)
hipError_t h2d_hip(void *dst, const void *src, size_t bytes, hipStream_t s) { return h2d_copy(dst, src, bytes, s) == IVFADC_OK ? hipSuccess : hipErrorUnknown; }

int d2h_copy(void *dst, const void *src, size_t bytes, hipStream_t s)
{
    if (bytes == 0) return IVFADC_OK;
    if (bytes <= BOUNCE_DIRECT_MAX) {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        return IVFADC_OK;
    }
    std::lock_guard<std::mutex> lk(g_bounce.mu);
    TRY(g_bounce.ensure());
    int i = 0;
    size_t prev_off = 0, prev_n = 0;
    for (size_t off = 0; off < bytes; off += BOUNCE_PIECE, i ^= 1) {
        const size_t n = std::min(BOUNCE_PIECE, bytes - off);
        HIP_TRY(hipMemcpyAsync(g_bounce.buf[i], (const char *)src + off, n, hipMemcpyDeviceToHost, s));
        if (prev_n) memcpy((char *)dst + prev_off, g_bounce.buf[i ^ 1], prev_n);   // the previous piece (waited for at the end of its turn)
        HIP_TRY(hipStreamSynchronize(s));
        prev_off = off;
        prev_n = n;
    }
    if (prev_n) memcpy((char *)dst + prev_off, g_bounce.buf[i ^ 1], prev_n);
    return IVFADC_OK;
}

static inline int pow2ceil(int x) { int p = 1; while (p < x) p <<= 1; return p; }

}  // namespace

// ---- environment -------------------------------------------------------------------------------------------------------------------
// The library reads EIGHT environment variables (include/ivfadc_hip.h lists them): switches a deployment may need without a rebuild --
// validation (exact kernels only), safety valves (no zero-copy host access, no stream probing, one batch in flight), the abort trace.
// Every other IVFADC_* name in this file is an A/B switch of a measured-and-closed experiment: it is looked up only in the diagnostic
// build (-DIVFADC_DEBUG, libivfadc_hip_dbg.so); the production library answers "not set" without touching the environment.
static const char *env_knob(const char *name)
{
    static const char *const product[] = {"IVFADC_NO_PRUNE", "IVFADC_NO_PIPELINE", "IVFADC_EXACT_TABLES", "IVFADC_NO_ZERO_COPY",
                                          "IVFADC_NO_STREAM_PROBE", "IVFADC_COARSE_EXACT", "IVFADC_NO_SMALLQ"};
#ifndef IVFADC_DEBUG
    bool ok = false;
    for (const char *k : product) ok = ok || strcmp(k, name) == 0;
    if (!ok) return nullptr;
#else
    (void)product;
#endif
    return getenv(name);
}

// ---- thread safety ------------------------------------------------------------------------------------------------------------------
// Every entry point locks its handle (a recursive mutex: entry points call each other), so calls on ONE handle from several threads
// are serialised; calls on DIFFERENT handles -- an index and its views, or unrelated indexes -- run concurrently, which is what
// ivfadc_clone_view is for.  What ties handles together is guarded by one process-wide mutex, g_topo_mu: the list of an index's views,
// a view's link to its index, the generation numbers.  Lock order: handle (index) -> g_topo_mu -> handle (view); a view's own calls
// take g_topo_mu only hand-over-hand in front of their own mutex (HandleLock) and NEVER behind it (check_view_current reads the link and
// the generation under the view's own mutex: their writers hold it), a mutator holds g_topo_mu and every view's mutex for
// as long as it edits the lists (MutationScope): no view search starts, or is still being enqueued, while the lists change, and the
// first one afterwards sees the new generation and refuses.
struct HandleMutex {
    std::recursive_mutex m;
    HandleMutex() = default;
    HandleMutex(const HandleMutex &) {}                        // (a view is made by copying the index's struct: it gets a mutex of its own)
    HandleMutex &operator=(const HandleMutex &) { return *this; }
};
static std::recursive_mutex g_topo_mu;

// an int that may be read without the handle's lock (make_plan's hint); copies with the handle (a view starts from its index's fields)
struct RelaxedInt {
    std::atomic<int> v{0};
    RelaxedInt() = default;
    RelaxedInt(const RelaxedInt &o) : v(o.v.load(std::memory_order_relaxed)) {}
    RelaxedInt &operator=(const RelaxedInt &o) { v.store(o.v.load(std::memory_order_relaxed), std::memory_order_relaxed); return *this; }
    int load() const { return v.load(std::memory_order_relaxed); }
    void store(int x) { v.store(x, std::memory_order_relaxed); }
    void fetch_add(int x) { v.fetch_add(x, std::memory_order_relaxed); }
    int fetch_add_get(int x) { return v.fetch_add(x, std::memory_order_relaxed); }
};

struct ivfadc_index {
    HandleMutex mu;
    int device = 0;
    int d = 0, kc = 0, m = 0, ksub = 0, dsub = 0, cs = 0;
    int num_cu = 256;
    hipStream_t stream = nullptr;

    DevBuf centroids, codebooks, codebooks_t, codebooks_p, labels, cnorm, tmin, tlist;
    // lower-bound tables on the matrix cores (lbscan.hip.h): bf16 split of the codebook, ||codeword||^2 and the f32 codewords, all in
    // label order, and max ||codeword|| per sub-quantizer; present for the shapes lb_shape() names
    DevBuf lb_split, lb_n2, lb_lab, lb_maxn;
    DevBuf lb_f16, lb_isc;       // round 5: f16 codewords x 2^e_ii and 2^-e_ii for the one-product build (lb_build_tables_f16)
    bool lb_use_f16 = true;      // IVFADC_LB_BF16=1 / ivfadc_set_table_mode(h, 3): the three-product bf16 split instead (A/B)
    // narrow-field list-major scan (nfscan.hip.h; m = 8, dsub = 16, ksub = 256): ||codeword||^2 by codeword index, f32 codewords by label
    DevBuf nf_n2, nf_lab;
    bool allow_nf = true;
    // eight-wave list-major scan (wg8scan.hip.h; m = 8, dsub = 16, ksub = 256, K <= 64): the work items' f32 tables, 32 KB per workgroup
    DevBuf wg8_tabs, wg8_items;  // (wg8_items: work item -> list, written by bucket_scan_kernel)
    int wg8_mode = 0;            // ivfadc_set_tuning(h, 4, chunk) keeps the plan's choice; wg8_mode: 0 = where it pays, 1 = wherever it exists, -1 = never
    // list-partitioned multi-GPU mode (ivfadc_set_list_partition): this handle scans the probed lists l with l % part_n == part_i only and
    // leaves partial top-K keys; partial_keys: where the running call wants them (null: ids as usual); the batch whose probe arrays stand
    int part_n = 1, part_i = 0;
    uint64_t *partial_keys = nullptr;
    int64_t partial_nq = -1;
    int partial_w = 0, partial_K = 0;
    DevBuf sq_keys, sq_cnt, sq_arrive, cent_t;   // small-batch path (smallq.hip.h): partial results, arrival counters
    bool allow_sq = true, sq_inside = false;
    bool allow_lb = true;
    bool force_lb = false;       // ivfadc_set_table_mode(h, 2): the matrix-core rounds wherever they are instantiated, not only where they pay
    DevBuf cent_hi, cent_lo, q_hi, q_lo;   // bf16 split operands of coarse_bf16_kernel ([rows][dp], dp = d rounded up to 32)
    // round 5: the one-product f16 form of the same filter: scaled f16 centroids, the batch's scaled f16 queries, per-query overflow flags
    DevBuf cent_f16, q_f16, q_flags;
    float f16_scale = 0.f;       // power of two: 2^11 <= scale * max|centroid component| <= 2^12 (0: the form is not available)
    bool allow_f16 = true, last_coarse_f16 = false;
    int dp32 = 0;
    bool allow_bf16 = true, last_coarse_bf16 = false;
    bool allow_prune = true;     // query-major scan: skip probes whose coarse distance exceeds the K-th best key (exact)
    bool allow_listed = true, last_listed = false;   // listed mode: per-tile records instead of the score matrix (run_coarse)
    // one-process-per-GPU result merge inside the library (ivfadc_comm_*): this rank's communicator, a side stream for the
    // collectives and one completion event per result slot
    void *comm = nullptr;
    int comm_ranks = 0, comm_rank = 0;
    hipStream_t comm_stream = nullptr;
    hipEvent_t comm_ready = nullptr;
    static constexpr int COMM_SLOTS = 8;
    hipEvent_t comm_done[COMM_SLOTS] = {};
    bool comm_busy[COMM_SLOTS] = {};
    int64_t comm_collectives = 0;
    // collectives complete in issue order on the side stream: seq numbers them, comm_slot_seq[s] = the last one that read slot s, and a
    // search stream (this handle's, or a view's: its own copy of comm_waited) has waited for every collective up to comm_waited
    int64_t comm_seq = 0, comm_slot_seq[COMM_SLOTS] = {}, comm_waited = 0;
    bool allow_filt = true;       // striped tables + rotated-order filter sums in the list-major kernels (ivfadc_set_table_mode)
    // certified two-level coarse search (twolevel.hip.h): group centres [G][d], slot ranges, radii, grouped centroids, slot -> cluster id;
    // tl_mode: 0 = automatic (built on the first search of a large quantizer, used if a self-probe says the bounds cut), 1 = on, -1 = off
    DevBuf tl_centres, tl_off, tl_rad, tl_cent, tl_slot, tl_gdist;
    int tl_G = 0, tl_mode = 0;
    bool dev_entry = false;   // the running search came in through a device-pointer entry AND another lane of this replica searched since this
                              // handle's previous search (the caller drives several lanes side by side: see make_plan)
    // feedback for make_plan's probes-per-round choice: how much of what the query-major scans probed was pruned (see fb_poll / fb_snapshot)
    PinnedBuf fb_pin;          // 4 KB: a snapshot of the sharded counters
    hipEvent_t fb_ev = nullptr;
    bool fb_pending = false;
    int fb_countdown = 0;
    int64_t fb_sp = 0, fb_pp = 0;   // counters at the last snapshot that was read
    float prune_est = -1.0f;        // smoothed pruned fraction of the probed points (< 0: not known yet)
    RelaxedInt lane_ticket;   // root index only: one ticket per device-entry search on the index or any of its views
    int my_ticket = -1;       // this handle's last ticket
    RelaxedInt n_views;   // live views of this index (a hint for make_plan: several batches are in flight on this replica)
    bool tl_tried = false, tl_use = false;
    float tl_eps = 0.f, tl_probe_fraction = -1.f;
    int64_t visited_base = 0;
    DevBuf gen_a, gen_b, gen_tmp, gen_off, gen_tot;   // generic path: key buffers (sort in/out), rocPRIM scratch, offsets
    int tlist_ldq = 0;
    int tmin_tiles = 0, tmin_tile_w = 0;   // set by run_coarse when the last coarse launch wrote tile minima
    float cmaxn = 0.f;            // >= max ||centroid||, for the MFMA filter's error bound
    bool allow_mfma = true;
    int mfma_min_kc = 2048;
    // lists (device layout)
    int64_t n = 0;
    bool have_lists = false;
    bool synthetic = false;
    bool dirty = false;           // host mirror changed, device copy stale
    int64_t maxlen = 0;
    int64_t inplace_appends = 0;
    DevBuf list_pos, list_len, list_codeoff, codes, ids, app_stage;
    // host mirror: one (codes, ids) pair per list -- the source of truth; the device copy gives every list spare
    // capacity so push! writes in place (re-layout only when a list overflows)
    std::vector<int64_t> h_len;                  // [kc] points per list (kept for synthetic lists too)
    std::vector<std::vector<uint8_t>> hl_codes;  // [kc] len x m bytes
    std::vector<std::vector<uint32_t>> hl_ids;   // [kc]
    std::vector<int64_t> d_cap, d_pos, d_codeoff;   // device layout: capacity, id offset, code byte offset per list
    int64_t ntotal() const
    {
        int64_t t = 0;
        for (int64_t v : h_len) t += v;
        return t;
    }
    std::vector<uint8_t> h_label_ok;   // m x 256 validity
    bool identity_labels = false;
    // the residual quantizer's rotation as loaded from an index file (nrows x nrows, column by column as persistency.jl:62-64 writes it);
    // empty = identity (:pq).  knn_search never reads it (index.jl:204-258): a rotated (:opq) index is SEARCHED as it is; quantize_data --
    // push! / encode -- would need it (third-party arithmetic, unverifiable here), so those entries refuse on such a handle.
    std::vector<float> rot;

    // workspace
    // ivfadc_set_next_queries: the hinted batch (good for one search), and the batch whose exact coarse rows stand in cdist2 -- written
    // by the tiles that rode behind the previous search's scan launch
    // A batch is named by (pointer, count, token): the token is the caller's generation number of the buffer's CONTENTS, so rows computed
    // from a buffer that has been refilled since can never be taken for the new contents.  pf_*: what the last search's riders wrote;
    // avail_*: what the running search may pick up (set at the top of search_dev, on every path, from pf_* -- rows serve the very next
    // search or none); cur_token: the token the caller declared for the next search's queries (ivfadc_set_query_token; 0 = undeclared)
    const float *hint_q = nullptr, *pf_q = nullptr, *avail_q = nullptr;
    int64_t hint_nq = 0, pf_nq = 0, avail_nq = 0;
    uint64_t hint_token = 0, pf_token = 0, cur_token = 0, own_token = 0;   // own_token: numbering of ivfadc_search_batches
    hipEvent_t hint_ev = nullptr;   // ivfadc_search_batches: the hinted rows are on the device once this event has fired (null: they are)
    DevBuf cdist2;
    DevBuf q_stage, cdist, probe_list, probe_dc, probe_base, list_cnt, bucket_off, wi_off, cursor, bucket_items, misc,
        qthr, part_keys, part_cnt, out_ids, out_dists, out_counts, assign, enc_codes, pts_stage, dbg;
    PinnedBuf pin_in, pin_out;   // host staging of ivfadc_search: pageable user buffers <-> pinned (the kernels read / write it in place)
    bool hc_out_direct = false, hc_legacy = false;   // the running host-pointer call: results go straight into the caller's arrays / old copy chain
    hipStream_t copy_stream = nullptr;               // ivfadc_search_batches: query ingest ahead of the searches
    hipStream_t copy_probed_a = nullptr, copy_probed_b = nullptr;   // ... probed to run beside these two (ensure_overlap)
    std::vector<hipEvent_t> ingest_ev;               // ... one event per upload group
    ivfadc_host_stats hstats{};
    // cumulative counters of internal views that no longer exist (a stale second lane of ivfadc_search_batches)
    int64_t carry_queries = 0, carry_scanned = 0, carry_pruned = 0, carry_surv = 0, carry_fallbacks = 0, carry_launches = 0;
    size_t qthr_armed = 0;       // entries of qthr known to hold KEY_MAX
    bool list_cnt_armed = false;
    size_t ws_budget = (size_t)8 << 30;

    // profiling
    bool profiling = false;
    int profiling_level = 0;
    struct EvPair { hipEvent_t a, b; int kind; };
    std::vector<EvPair> pending;
    std::vector<EvPair> free_ev;
    ivfadc_stats stats{};
    int64_t scanned_base = 0, fallback_base = 0, pruned_base = 0, surv_base = 0;
    int force_qg = 0, force_chunk = 0, force_pg = 0;
    bool own_stream = true;
    struct FnCfg { const void *fn; size_t lds; int occ; };
    std::vector<FnCfg> fn_cfg;

    // Read-only views (ivfadc_clone_view): a second handle on the SAME device arrays of quantizers and lists, with a stream and a
    // workspace of its own, so that two batches can be in flight on one replica (a SIFT1M-shape launch is 1024 workgroups that all start
    // together: its ramp and its tail leave the chip half empty, and a second stream fills them).  A view holds no host mirror and cannot
    // change anything; any change to the index it was taken from (generation) makes it refuse to search.
    ivfadc_index *view_of = nullptr;
    bool is_view = false, orphan = false;
    uint64_t generation = 0, view_gen = 0;
    std::vector<ivfadc_index *> views;
    ivfadc_index *pipe_view = nullptr;            // ivfadc_search_batches' second lane (created on first use)
    hipEvent_t pipe_ev_in = nullptr, pipe_ev_out = nullptr;
};

// the lock of an entry point on handle h (null: nothing to lock).  A view takes the topology mutex first and lets it go once its own
// mutex is held: a mutator of its index, which holds the topology mutex while it waits for the views, is never waited for in a cycle.
struct HandleLock {
    ivfadc_index *h;
    explicit HandleLock(ivfadc_index *x) : h(x)
    {
        if (!h) return;
        if (h->is_view) {
            std::lock_guard<std::recursive_mutex> topo(g_topo_mu);
            h->mu.m.lock();
        } else {
            h->mu.m.lock();
        }
    }
    ~HandleLock() { if (h) h->mu.m.unlock(); }
    HandleLock(const HandleLock &) = delete;
    HandleLock &operator=(const HandleLock &) = delete;
};

namespace {

// Low-latency wait for short batches: hipStreamSynchronize can take a sleep/interrupt path that adds a few
// hundred microseconds; poll for up to ~2 ms first.
int wait_stream(ivfadc_index *h)
{
    for (int i = 0; i < 200000; ++i) {
        const hipError_t e = hipStreamQuery(h->stream);
        if (e == hipSuccess) return IVFADC_OK;
        if (e != hipErrorNotReady) return fail(IVFADC_ERR_HIP, "hipStreamQuery failed: %s", hipGetErrorString(e));
        if (i > 20000) break;
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    return IVFADC_OK;
}

int set_device(ivfadc_index *h)
{
    HIP_TRY(hipSetDevice(h->device));
    return IVFADC_OK;
}

// ---- events ------------------------------------------------------------------------------
int ev_begin(ivfadc_index *h, int kind, ivfadc_index::EvPair &ep)
{
    if (!h->free_ev.empty()) {
        ep = h->free_ev.back();
        h->free_ev.pop_back();
    } else {
        HIP_TRY(hipEventCreate(&ep.a));
        HIP_TRY(hipEventCreate(&ep.b));
    }
    ep.kind = kind;
    HIP_TRY(hipEventRecord(ep.a, h->stream));
    return IVFADC_OK;
}

int ev_end(ivfadc_index *h, ivfadc_index::EvPair &ep)
{
    HIP_TRY(hipEventRecord(ep.b, h->stream));
    h->pending.push_back(ep);
    return IVFADC_OK;
}

int ev_fold(ivfadc_index *h)
{
    if (h->pending.empty()) return IVFADC_OK;
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (auto &ep : h->pending) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ep.a, ep.b));
        if (ep.kind == 0) { h->stats.scan_ms += ms; h->stats.scan_launches++; }
        else if (ep.kind == 2) { h->stats.lb_build_ms += ms; h->stats.lb_build_launches++; }
        else h->stats.coarse_ms += ms;
        h->free_ev.push_back(ep);
    }
    h->pending.clear();
    return IVFADC_OK;
}

// ---- device layout of the lists --------------------------------------------------------------
constexpr size_t CODE_SLACK = 64 << 10;

int code_stride(int m) { return (int)align_up((size_t)m, 4); }   // every kernel variant reads this stride

// capacities -> offsets of every list in the id array and in the code buffer (256-B aligned blocks)
int layout_from_caps(ivfadc_index *h, size_t &code_bytes, int64_t &id_slots)
{
    const int kc = h->kc;
    h->d_pos.assign(kc, 0);
    h->d_codeoff.assign(kc, 0);
    size_t run = 0;
    int64_t pos = 0, maxlen = 0, n = 0;
    for (int l = 0; l < kc; ++l) {
        h->d_pos[l] = pos;
        h->d_codeoff[l] = (int64_t)run;
        pos += h->d_cap[l];
        run += align_up((size_t)h->d_cap[l] * h->cs, 256);
        maxlen = std::max(maxlen, h->h_len[l]);
        n += h->h_len[l];
    }
    if (n > (int64_t)0xFFFFFFFFll) return fail(IVFADC_ERR_ASSERT, "index capacity of UInt32 ids exceeded");
    if (pos > (int64_t)0xFFFFFFFFll) return fail(IVFADC_ERR_INVALID, "id array with spare capacity exceeds 2^32 slots");
    h->maxlen = maxlen;
    h->n = n;
    code_bytes = run + CODE_SLACK;
    id_slots = pos;
    return IVFADC_OK;
}

int upload_list_tables(ivfadc_index *h)
{
    const int kc = h->kc;
    std::vector<uint32_t> len32((size_t)kc);
    for (int l = 0; l < kc; ++l) len32[l] = (uint32_t)h->h_len[l];
    TRY(h->list_pos.ensure((size_t)kc * 8));
    TRY(h->list_len.ensure((size_t)kc * 4));
    TRY(h->list_codeoff.ensure((size_t)kc * 8));
    TRY(h2d_copy(h->list_pos.p, h->d_pos.data(), (size_t)kc * 8, h->stream));       // (synchronous: h2d_copy)
    TRY(h2d_copy(h->list_len.p, len32.data(), (size_t)kc * 4, h->stream));
    TRY(h2d_copy(h->list_codeoff.p, h->d_codeoff.data(), (size_t)kc * 8, h->stream));
    return IVFADC_OK;
}

// full re-layout: host mirror -> device, every list gets spare capacity behind it
int upload_lists(ivfadc_index *h)
{
    const int kc = h->kc, m = h->m, cs = h->cs;
    h->d_cap.assign(kc, 0);
    for (int l = 0; l < kc; ++l) h->d_cap[l] = h->h_len[l] + std::max<int64_t>(32, h->h_len[l] / 8);
    size_t total = 0;
    int64_t slots = 0;
    TRY(layout_from_caps(h, total, slots));
    std::vector<uint8_t> stage(total, 0);
    std::vector<uint32_t> idstage((size_t)std::max<int64_t>(1, slots), 0);
    for (int l = 0; l < kc; ++l) {
        const int64_t len = h->h_len[l];
        uint8_t *dst = stage.data() + h->d_codeoff[l];
        const uint8_t *src = h->hl_codes[l].data();
        if (cs == m) {
            if (len) memcpy(dst, src, (size_t)len * m);
        } else {
            for (int64_t p = 0; p < len; ++p) memcpy(dst + (size_t)p * cs, src + (size_t)p * m, m);
        }
        if (len) memcpy(idstage.data() + h->d_pos[l], h->hl_ids[l].data(), (size_t)len * 4);
    }
    TRY(h->codes.ensure(total));
    TRY(h->ids.ensure(idstage.size() * 4));
    TRY(h2d_copy(h->codes.p, stage.data(), total, h->stream));
    TRY(h2d_copy(h->ids.p, idstage.data(), idstage.size() * 4, h->stream));
    TRY(upload_list_tables(h));
    h->dirty = false;
    h->have_lists = true;
    h->synthetic = false;
    return IVFADC_OK;
}

// ---- scan kernel dispatch ----------------------------------------------------------------------
typedef void (*scan_fn_t)(const ScanArgs);
typedef void (*qscan_fn_t)(const QScanArgs);

// Specialised (m, dsub) pairs; every other shape runs the fully generic <0, 0> kernels.
template <int M, int DS, bool SMALL> scan_fn_t scan_fn_qg(int qg, bool stripe)
{
    if constexpr ((M == 8 || M == 16) && DS > 0) {
        if (stripe && qg == 4) return scan_kernel<M, DS, 4, SMALL, true>;
    }
    switch (qg) {
    case 1: return scan_kernel<M, DS, 1, SMALL>;
    case 2: return scan_kernel<M, DS, 2, SMALL>;
    default: return scan_kernel<M, DS, 4, SMALL>;
    }
}

template <int M, int DS, bool SMALL> qscan_fn_t qscan_fn_pg(int pg)
{
    switch (pg) {
    case 1: return qscan_kernel<M, DS, 1, SMALL>;
    case 2: return qscan_kernel<M, DS, 2, SMALL>;
    default: return qscan_kernel<M, DS, 4, SMALL>;
    }
}

#define IVF_SHAPES(X) X(8, 16) X(16, 6) X(16, 8) X(48, 16)

// shapes the matrix-core lower-bound rounds (qscan_kernel<..., LB = true>) are instantiated for
bool lb_shape(int m, int dsub) { return (m == 48 && dsub == 16) || (m == 16 && dsub == 6); }

template <int M, int DS> qscan_fn_t pick_qscan_lb_pg(int pg)
{
    switch (pg) {
    case 1: return qscan_kernel<M, DS, 1, true, true>;
    case 2: return qscan_kernel<M, DS, 2, true, true>;
    case 3: return qscan_kernel<M, DS, 3, true, true>;
    default: return qscan_kernel<M, DS, 4, true, true>;
    }
}

qscan_fn_t pick_qscan_lb(int m, int dsub, int pg)
{
    if (m == 48 && dsub == 16) return pick_qscan_lb_pg<48, 16>(pg);
    if (m == 16 && dsub == 6) return pick_qscan_lb_pg<16, 6>(pg);
    return nullptr;
}

// mirrors LbCfg<M, DS, PG>::END + the tail of qscan_kernel's carve (scnt, swi, sthr, probe cache)
size_t lb_lds_bytes(int m, int dsub, int pg)
{
    size_t b = align_up((size_t)pg * ((size_t)m * 256 + 32), 16);
    b += (size_t)m * pg * ((dsub + 3) & ~3) * 4;   // f32 residuals of the round's probes (rows padded to 16-byte groups)
    b += 2 * (size_t)m * pg * 4 + 128 + 256;   // norms, bases, per-probe constants of the round and of the query
    b += (size_t)m * dsub * 4;              // query
    b += (size_t)4 * (pg >= 4 ? 54 : 16) * (m / 4 + 2) * 4;   // parking pools (LbCfg::PCAP entries per wave)
    b += 4 * 64 * 8;                        // upper-bound keys of the four waves
    b += (size_t)4 * pg * 4 + 16 + (size_t)pg * 40 + 3 * 256;
    return b;
}

template <bool SMALL> scan_fn_t pick_scan_s(int m, int dsub, int qg, bool stripe)
{
#define X(M_, D_) if (m == M_ && dsub == D_) return scan_fn_qg<M_, D_, SMALL>(qg, stripe);
    IVF_SHAPES(X)
#undef X
    return scan_fn_qg<0, 0, SMALL>(qg, false);
}

template <bool SMALL> qscan_fn_t pick_qscan_s(int m, int dsub, int pg)
{
#define X(M_, D_) if (m == M_ && dsub == D_) return qscan_fn_pg<M_, D_, SMALL>(pg);
    IVF_SHAPES(X)
#undef X
    return qscan_fn_pg<0, 0, SMALL>(pg);
}

scan_fn_t pick_scan(int m, int dsub, int qg, bool small, bool stripe)
{
    return small ? pick_scan_s<true>(m, dsub, qg, stripe) : pick_scan_s<false>(m, dsub, qg, stripe);
}

qscan_fn_t pick_qscan(int m, int dsub, int pg, bool small)
{
    return small ? pick_qscan_s<true>(m, dsub, pg) : pick_qscan_s<false>(m, dsub, pg);
}

// query-major scan with the next batch's coarse tiles behind it (qscan_coarse_kernel): the register-selector kernels of the shapes below
typedef void (*qscan_coarse_fn_t)(const QScanArgs, const CoarseNext);
qscan_coarse_fn_t pick_qscan_coarse(int m, int dsub, int pg)
{
    if (pg != 1 && pg != 2) return nullptr;
    if (m == 8 && dsub == 16) return pg == 1 ? qscan_coarse_kernel<8, 16, 1> : qscan_coarse_kernel<8, 16, 2>;
    if (m == 16 && dsub == 8) return pg == 1 ? qscan_coarse_kernel<16, 8, 1> : qscan_coarse_kernel<16, 8, 2>;
    if (m == 16 && dsub == 6) return pg == 1 ? qscan_coarse_kernel<16, 6, 1> : qscan_coarse_kernel<16, 6, 2>;
    if (m == 48 && dsub == 16) return nullptr;   // three workgroups per CU at 168 registers: not worth a second instantiation
    return pg == 1 ? qscan_coarse_kernel<0, 0, 1> : qscan_coarse_kernel<0, 0, 2>;
}

typedef void (*sq_fn_t)(const SqArgs);
sq_fn_t pick_sq(int m, int dsub)
{
#define X(M_, D_) if (m == M_ && dsub == D_) return sq_kernel<M_, D_>;
    IVF_SHAPES(X)
#undef X
    return sq_kernel<0, 0>;
}

// shapes the striped list-major kernels exist for (IVF_SHAPES with m = 8 / 16)
bool filt_shape(int m, int dsub) { return (m == 8 && dsub == 16) || (m == 16 && (dsub == 6 || dsub == 8)); }

// the shape the narrow-field list-major kernel exists for (nfscan.hip.h)
bool nf_shape(int m, int dsub) { return m == 8 && dsub == 16; }

// mirrors carve_lds() in kernels.hip.h
size_t scan_lds_bytes(const ivfadc_index *h, int qg, int cap, bool small, bool list_major = false)
{
    size_t b = (size_t)std::max(h->m, 2) * 256 * qg * 4;
    b += align_up((size_t)h->d * qg, 4) * 4;
    if (!small) b += (size_t)4 * (list_major ? qg : 1) * cap * 8;   // LDS selectors: per wave and query (query-major: one query)
    b += (size_t)4 * qg * 4 + 16;
    b = align_up(b, 8) + (size_t)qg * 40;  // workgroup-shared thresholds + the four waves' quarter keys per slot (STHR_WORDS)
    b += 3 * 256;                           // query-major kernel: LDS copy of the query's probes (see qscan_kernel)
    if (list_major && qg == 4 && (h->m == 8 || h->m == 16))
        b += 4 * 16 * 5 * 4;                // striped list-major kernels: 4 waves x CAND_CAP parked points x <= 5 dwords
    if (list_major && qg == 4 && h->m == 8 && h->dsub == 16 && h->allow_filt && h->ksub == 256)
        b += 256 * 64;                      // m = 8 striped: the 16-bit integer filter table behind the float tables (QF_BYTES)
    return b;
}

constexpr size_t LDS_MAX = 160 << 10;
// the eight-wave list-major kernel (wg8scan.hip.h) as the plan's own choice on long lists (ivfadc_set_table_mode(h, 6) asks for it anywhere)
#ifndef W9_MIN_PPL
#define W9_MIN_PPL 8.0      // probes per list from which the eight-query form of the eight-wave kernel is planned
#endif
#ifndef W8_DEFAULT_ON
#define W8_DEFAULT_ON 1
#endif
// misc device block: [0, 4096) 64 scanned-point counters at a 64-B stride; [4096] work-queue head; [4096 + 64] coarse fallbacks;
// [4096 + 256, + 512) the eight per-XCD queue heads of the narrow-field kernel, 64 B apart
constexpr size_t MISC_BYTES = 4096 + 256 + 512;

struct Plan {
    bool coarse_mfma;   // coarse scores on the matrix cores + certified exact refine (w <= 48)
    bool fuse_topw;   // query-major only: top-w selection runs inside the scan kernel
    bool query_major;
    bool lb;            // query-major rounds with 8-bit lower-bound tables from the matrix cores (lbscan.hip.h)
    bool nf;            // list-major with the narrow-field integer filter, eight queries per code stream (nfscan.hip.h)
    bool wg8q8;         // ... its eight-query form (wg8q8scan.hip.h)
    bool wg8;           // list-major, eight waves per workgroup on four conflict-free copies of the integer filter table (wg8scan.hip.h)
    bool lanes;         // several batches in flight on this replica: stand-alone top-w, a wave per query (see make_plan)
    bool twolevel;      // coarse stage: certified two-level search (twolevel.hip.h) instead of the exhaustive kernels + top-w
    bool small_k, small_w;
    int qg, cap, capw, maxch;
    uint32_t CH;
    size_t lds;
    int64_t nb;   // queries per sub-batch
    bool fits;    // false: the selection kernels' LDS need exceeds the CU's 160 KB -> the caller takes the generic path
};

// ---- how many probes a query-major round takes is a question about the DATA ---------------------------------------------------------
// Two probes per round share every codeword fetch between two tables -- right when most probes are scanned; one probe per round lets
// exact pruning look at the bound after EVERY list -- right when the bound the closest cells leave prunes most of the rest.  SIFT1M
// shape, two batches in flight (profiles/r05_pg12_matrix.txt): clustered data (93 % of the probed points pruned) 49.6 M q/s with one
// probe per round against 44.8 M with two, w = 32 41.5 against 31.5 M; data where nothing prunes 19.5 against 22.5 M.  The scan kernels
// count probed and pruned points anyway (ivfadc_stats); every few searches a 4 KB snapshot of the counters follows the scan on the
// stream into pinned memory, and a later search that finds it arrived (hipEventQuery: no wait, ever) folds the pruned fraction into
// h->prune_est.  Plans change with it, results cannot (every plan returns the reference's bytes).
constexpr float PG1_MIN_PRUNED = 0.6f;
int fb_poll(ivfadc_index *h)
{
    if (!h->fb_pending) return IVFADC_OK;
    const hipError_t q = hipEventQuery(h->fb_ev);
    if (q == hipErrorNotReady) return IVFADC_OK;
    if (q != hipSuccess) return fail(IVFADC_ERR_HIP, "hipEventQuery failed: %s", hipGetErrorString(q));
    h->fb_pending = false;
    const int64_t *sh = (const int64_t *)h->fb_pin.p;
    int64_t sp = 0, pp = 0;
    for (int i = 0; i < 64; ++i) { sp += sh[i * 8]; pp += sh[i * 8 + 1]; }
    const int64_t dsp = sp - h->fb_sp, dpp = pp - h->fb_pp;
    h->fb_sp = sp;
    h->fb_pp = pp;
    if (dsp > 0 && dpp >= 0 && dpp <= dsp) {
        const float f = (float)((double)dpp / (double)dsp);
        h->prune_est = h->prune_est < 0.0f ? f : 0.5f * h->prune_est + 0.5f * f;
    }
    return IVFADC_OK;
}
int fb_snapshot(ivfadc_index *h)   // behind a query-major scan launch, on its stream
{
    static const bool off = env_knob("IVFADC_NO_PG_FEEDBACK") != nullptr;
    if (off || h->fb_pending || !h->misc.p) return IVFADC_OK;
    if (--h->fb_countdown > 0) return IVFADC_OK;
    h->fb_countdown = 8;
    TRY(h->fb_pin.ensure(4096));
    if (!h->fb_ev) HIP_TRY(hipEventCreateWithFlags(&h->fb_ev, hipEventDisableTiming));
    HIP_TRY(hipMemcpyAsync(h->fb_pin.p, h->misc.p, 4096, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipEventRecord(h->fb_ev, h->stream));
    h->fb_pending = true;
    return IVFADC_OK;
}

int make_plan(ivfadc_index *h, int64_t nq, int K, int w, Plan &pl)
{
    pl.fits = true;
    pl.lanes = false;
    pl.small_k = K <= 64;
    pl.small_w = w <= 64;
    pl.cap = pl.small_k ? 64 : std::max(128, pow2ceil(K + 64));
    pl.capw = pl.small_w ? 64 : std::max(128, pow2ceil(w + 64));
    const double avg_len = (double)h->n / std::max(1, h->kc);
    // list-major (queries that probe a list share its code stream; per-item grouping/merge overhead) wins when
    // the lists are very long (a query's probe must be split over workgroups anyway), or when a list is probed by
    // enough queries to fill the groups AND the batch makes enough work items to fill the chip; otherwise
    // query-major (one workgroup per query, no grouping, no partial results).  Measured crossovers, m = 8,
    // kc = 8192, 16 probes/list: list-major ahead from 10 KB lists on (1.74 vs 1.90 ms); SIFT1M-shape (8
    // probes/list but only 2048 work items): query-major 92 vs 190 us; Deep1B-shape (4.9 probes/list): query-major.
    const double ppl_ = (double)nq * w / std::max(1, h->kc);
    const bool long_lists = avg_len * h->m > 256.0 * 1024.0;
    // ... and the lists are long enough to amortise a fresh selection per (query, list): kc = 8192, 16 probes/list:
    // 3.7k-point lists 1.88 vs 2.01 ms, 12k 2.70 vs 3.28 ms for list-major; SIFT1M-shape (1k-point lists) with 16k-64k
    // queries: query-major 18-20 M q/s vs 12 M
    const bool shared = ppl_ >= 6.0 && (double)nq * w / 4.0 >= 64.0 * h->num_cu && avg_len >= 3072.0;
    // ... or when a handful of queries with heavy probes would leave most CUs idle at one workgroup per query: list-major
    // spreads a query's probes over workgroups (HD-shape, 64 queries: w = 8 0.39 vs 0.31 M q/s, w = 32 0.21 vs 0.10;
    // SIFT1M- and Deep1B-shape probes are too light for it: 1.4 vs 3.0 M and 0.2 vs 0.7 M q/s)
    const bool few_heavy = nq <= h->num_cu && w >= 8 && avg_len * h->m >= 64.0 * 1024.0;   // 256 queries: +10 % (w = 8), +60 % (w = 32); 512: even
    // ... or a very small batch with several probes each: one workgroup per query walks its w probes in sequence, list-major
    // (ungrouped: one work item per (query, probe), no bucket kernels) runs them side by side.  SIFT1M-shape, measured
    // query-major vs list-major per batch: 1 query w = 8 42 vs 35 us, w = 16 64 vs 42 us, w = 32 118 vs 56 us; even at
    // 64 queries for w = 8 / 16; w <= 4 stays query-major (30 vs 33 us).  Not with a large kc, where the per-list
    // bookkeeping of the list-major plan costs more than it buys (Deep1B-shape 380 vs 234 us)
    const bool few_many = w >= 8 && nq <= 32 + (int64_t)w && nq <= h->num_cu / 4 && h->kc <= 8192;
    pl.query_major = !(long_lists || shared || few_heavy || few_many);
    if (h->force_qg == -1 || h->force_qg == -3) pl.query_major = true;
    if (h->part_n > 1) pl.query_major = false;   // list-partitioned mode: the list-major plan is the one that skips other ranks' lists
    const bool forced = (h->force_qg == 1 || h->force_qg == 2 || h->force_qg == 4 || h->force_qg == 8);
    if (forced) pl.query_major = false;
    pl.CH = 0;
    pl.maxch = 1;
    pl.wg8 = false;
    pl.wg8q8 = false;
    pl.fuse_topw = false;
    pl.lb = false;
    pl.nf = false;
    // The filter pays when the coarse search is large: below ~2k centroids the extra selection + refine work in the
    // scan prologue costs more than the VALU kernel it replaces (SIFT1M-shape: 92 -> 121 us per batch).
    pl.coarse_mfma = h->allow_mfma && w <= 48 && h->kc >= h->mfma_min_kc && (h->d & 3) == 0;
    pl.twolevel = h->tl_use && h->tl_G > 0 && w <= 64 && (h->d & 3) == 0 && nq <= ((int64_t)1 << 30) / std::max(1, h->tl_G) &&
                  (size_t)4 * h->d * 4 + 4 * 64 * 8 <= (size_t)(96 << 10);   // (its four waves keep their queries in LDS)
    if (pl.twolevel) pl.coarse_mfma = false;
    if (pl.query_major) {
        static const bool no_fuse = env_knob("IVFADC_NO_FUSE_TOPW") != nullptr;
        // large kc: the selection is a 4*kc-byte stream per query, better done by the lean stand-alone kernel
        pl.fuse_topw = pl.small_w && !no_fuse && h->kc <= 8192 && h->force_qg != -3 && !pl.twolevel;
        // Several batches in flight on this replica (the index has views, or this IS a view: ivfadc_search_batches' second lane, a serving
        // loop's lanes): what bounds the chip then is register-file time, not one launch's latency -- and the fused selection holds a scan
        // workgroup's four waves and 126 VGPRs each for the 10 k cycles (28 % of its life on the SIFT1M shape, IVFADC_DEBUG_STAMPS) in which
        // wave 0 selects and the others wait.  A stand-alone selection, one lean wave per query, costs a launch and gives the scan its
        // registers back: 42.5 -> 44.7 M q/s with two batches in flight (profiles/r05_topw_probe.txt); one batch at a time keeps the fused form.
        static const bool lanes_fused = env_knob("IVFADC_LANES_FUSE_TOPW") != nullptr;
        // (device-pointer entries only, and only while the lanes really run side by side -- h->dev_entry, set by the entry from the root
        // index's tickets: the host entries are bound by the host's enqueue time, where one more launch per batch costs
        // ivfadc_search_batches 27.8 -> 25.8 M q/s, and a caller who owns views but searches one batch at a time would pay a launch for nothing)
        pl.lanes = h->dev_entry && (h->is_view || h->n_views.load() > 0) && nq >= 4 * (int64_t)h->num_cu && h->kc >= 512;
        if (pl.lanes && !lanes_fused) pl.fuse_topw = false;
        // probes per round: share each codeword fetch between PG tables, keep >= 4 workgroups per CU when possible
        int pg = w >= 2 ? 2 : 1;   // measured: PG=2 beats PG=4 (register pressure halves the occupancy at 4)
        if (pl.small_k && h->allow_prune && h->prune_est >= PG1_MIN_PRUNED) pg = 1;   // (fb_poll: most of what is probed gets pruned)
        if (h->force_pg == 1 || h->force_pg == 2 || h->force_pg == 4) pg = h->force_pg;
        static const int big_cap_kb = env_knob("IVFADC_PG_LDS_CAP_KB") ? atoi(env_knob("IVFADC_PG_LDS_CAP_KB")) : 40;
        const size_t pg_lds_cap = (h->force_pg == 4) ? LDS_MAX : (pl.small_k ? (size_t)(40 << 10) : (size_t)big_cap_kb << 10);   // forcing 4 lifts the 4-workgroups-per-CU cap
        while (pg > 1 && scan_lds_bytes(h, pg, pl.cap, pl.small_k) > pg_lds_cap) pg >>= 1;
        pl.qg = pg;
        pl.lds = scan_lds_bytes(h, pg, pl.cap, pl.small_k);
        // lower-bound tables on the matrix cores: four probes per round share one pass over the codebook (a quarter of the exact
        // build's L1 traffic, a fraction of its vector-ALU work); register selectors and the LDS probe copy only
        static const bool no_lb = env_knob("IVFADC_NO_LB") != nullptr;
        static const bool lb_everywhere = env_knob("IVFADC_LB_EVERYWHERE") != nullptr;   // = ivfadc_set_table_mode(h, 2), for A/B runs of bench.py
        // (measured: m = 48, where the exact build re-reads 768 KB of codewords per probe, +20 % on the HD shape; m = 16 with 1.5 k-point
        // lists -- the Deep1B shape -- loses 12 %: four barriers and the round's setup per four 24 KB lists cost what the cheaper tables
        // save, so there the rounds run only on request, ivfadc_set_table_mode(h, 2))
        pl.lb = !no_lb && h->allow_lb && h->allow_filt && h->lb_split.p != nullptr && lb_shape(h->m, h->dsub) && pl.small_k && w <= 32 &&
                h->ksub == 256 && (h->m >= 32 || h->force_lb || lb_everywhere);
        if (pl.lb) {
            // the top-w selection of a large batch runs as its own launch, one wave per query at full occupancy (per-tile records,
            // no score matrix); inside this kernel -- two workgroups per CU, three waves idle -- it was a sixth of the launch
            static const bool lb_fuse = env_knob("IVFADC_LB_FUSE_TOPW") != nullptr;
            if ((nq >= 4 * (int64_t)h->num_cu && !lb_fuse) || pl.twolevel) pl.fuse_topw = false;
            pl.qg = w >= 3 ? 4 : w;
            if (h->force_pg >= 1 && h->force_pg <= 4) pl.qg = std::min(w, h->force_pg);
            pl.lds = lb_lds_bytes(h->m, h->dsub, pl.qg);
        }
        if (pl.lds > LDS_MAX) { pl.fits = false; return IVFADC_OK; }   // e.g. m = 48 with K near 2048: tables + selector buffers
    } else {
        // query-group width from the expected number of probes per list
        const double ppl = (double)nq * w / std::max(1, h->kc);
        int qg = ppl >= 2.5 ? 4 : (ppl >= 1.25 ? 2 : 1);
        // billion-scale lists (a list is megabytes: the stream, not the per-item work, is what a group shares): wider groups pay
        // much earlier.  SIFT1B shape, step in ms at QG = 1 / 2 / 4 (profiles/r03_a_sift1b_plan_sweep.txt): probes per list 0.125:
        // 0.26 / 0.30 / 0.38; 0.25: 0.45 / 0.42 / 0.54; 0.5: 0.90 / 0.70 / 0.81; 1.0: 1.55 / 1.07 / 1.15; 2.0: 2.93 / 1.81 / 1.57
        // -- the regime of one rank of an 8-GPU run on a 16 384-query batch (2048 queries, w = 8)
        if (long_lists) qg = ppl >= 1.5 ? 4 : (ppl >= 0.2 ? 2 : 1);
        // ... and where the eight-wave kernel exists (it is planned below for groups of four) it pays from half a probe per list: scan ms of
        // the SIFT1B shape at 0.25 / 0.5 / 1 probes per list, eight-wave kernel against scan_kernel<QG=2>: 0.355 / 0.525 / 0.70-0.75 against
        // 0.316 / 0.544 / 0.92-0.93 (round 6)
        const bool w8_shape = pl.small_k && h->allow_filt && h->wg8_mode >= 0 && h->m == 8 && h->dsub == 16 && h->ksub == 256 && h->d == 128 &&
                              h->maxlen < ((int64_t)1 << 28) && h->part_n <= 1 && W8_DEFAULT_ON && avg_len >= 8192.0;
        if (long_lists && w8_shape && ppl >= 0.5) qg = 4;
        if (forced) qg = h->force_qg;
        // eight queries per code stream behind the 4-bit narrow-field filter (nfscan.hip.h): conflict-free gathers, a third of the vector
        // instructions per (query, point), every list streamed once per eight queries -- where the shape has the kernel, K fits the
        // register selectors and the lists are probed often enough to fill the groups
        // (measured, SIFT1B shape, 16 384 x w = 8: 7.76 ms against 7.44 ms for the 16-bit four-query kernel -- DESIGN.md 4.11 -- so the kernel
        // runs on request only: ivfadc_set_tuning(h, 8, 0), or IVFADC_NF=1 for A/B runs of bench.py)
        static const bool nf_auto = env_knob("IVFADC_NF") != nullptr;
        const bool nf_ok = h->allow_nf && h->allow_filt && h->nf_n2.p != nullptr && nf_shape(h->m, h->dsub) && h->ksub == 256 && pl.small_k;
        static const double nf_min_ppl = env_knob("IVFADC_NF_MIN_PPL") ? atof(env_knob("IVFADC_NF_MIN_PPL")) : 3.0;
        if (h->force_qg == 8 && !nf_ok) qg = 4;
        pl.nf = nf_ok && (h->force_qg == 8 || (nf_auto && !forced && ppl >= nf_min_ppl && avg_len >= 2048.0));
        if (pl.nf) {
            qg = 8;
            pl.qg = 8;
            pl.lds = (size_t)NfLds::END;
        } else {
        // keep two workgroups per CU when possible (a forced width only yields to the hard LDS limit)
        while (qg > 1 && scan_lds_bytes(h, qg, pl.cap, pl.small_k, true) > (forced ? LDS_MAX : (size_t)(80 << 10))) qg >>= 1;
        if (scan_lds_bytes(h, qg, pl.cap, pl.small_k, true) > LDS_MAX) { pl.fits = false; return IVFADC_OK; }
        pl.qg = qg;
        pl.lds = scan_lds_bytes(h, qg, pl.cap, pl.small_k, true);
        // four queries per code stream on long lists of the m = 8 / dsub = 16 shape: the eight-wave kernel (a work item must feed
        // eight waves: lists of at least 8 K points).  Measured on the SIFT1B shape against the four-wave kernel (profiles/r06_w8_sweep.txt,
        // the eight-wave kernel with its workgroup pool): 16 384 queries, scan ms at w = 1 / 8: 1.36 / 5.73 against 1.68 / 7.46;
        // 2048 x w = 8: 1.06 against 1.47.
        // (positions and byte offsets of a list are 28- / 31-bit quantities in the kernel: lists of fewer than 2^28 points; the
        // list-partitioned mode keeps the four-wave kernel it was validated with)
        pl.wg8 = qg == 4 && pl.small_k && h->allow_filt && h->wg8_mode >= 0 && h->m == 8 && h->dsub == 16 && h->ksub == 256 && h->d == 128 &&
                 h->maxlen < ((int64_t)1 << 28) && h->part_n <= 1 &&
                 (h->wg8_mode > 0 || (W8_DEFAULT_ON && avg_len >= 8192.0));   // (w = 1 too since the workgroup pool: 1.36 against the four-wave kernel's 1.67 ms)
        if (pl.wg8) pl.lds = (size_t)W8Lds::END;
        // ... and EIGHT queries per code stream (wg8q8scan.hip.h: 16-byte entries, 32 instead of 48 instructions per point and eight queries)
        // where the lists are probed often enough to fill groups of eight (table mode 7: wherever the kernel exists)
        pl.wg8q8 = pl.wg8 && (h->wg8_mode == 2 || (h->wg8_mode == 0 && !forced && ppl >= W9_MIN_PPL));
        if (pl.wg8q8) {
            qg = 8;
            pl.qg = 8;
            pl.lds = (size_t)W9Lds::END;
        }
        }
        // chunk size: enough work items to fill the chip, as few table rebuilds as possible.  Two items per CU is the
        // measured optimum on billion-scale lists (SIFT1B-shape, 16..1024 queries, w = 1 and 8: every case at or within
        // 5 % of its best chunk size; sixteen per CU rebuilt tables up to 15 times per probe)
        const double items_target = 2.0 * h->num_cu;
        const double ch = (double)nq * w * avg_len / qg / items_target;
        uint32_t CH = 4096;
        // (cap 2^18 points: on the billion-scale shapes a chunk then covers a whole list -- one table build and one selector
        // warm-up per (list, query group) instead of two: SIFT1B-shape scan 8.64 -> 7.45 ms at w = 8, 1.87 -> 1.69 ms at w = 1)
        while ((double)CH < ch && CH < (1u << 18)) CH <<= 1;
        if (h->force_chunk > 0) CH = (uint32_t)align_up((size_t)h->force_chunk, 1024);
        while ((h->maxlen + CH - 1) / CH > 64) CH <<= 1;   // bound the partial-result slots per probe
        pl.CH = CH;
        pl.maxch = (int)std::max<int64_t>(1, (h->maxlen + CH - 1) / CH);
    }
    // sub-batch so the workspace stays inside the budget
    const size_t per_q = (pl.twolevel ? (size_t)h->tl_G : (size_t)h->kc) * 4 + (pl.query_major ? 0 : (size_t)w * pl.maxch * ((size_t)K * 8 + 4)) +
                         (size_t)w * 20 + (size_t)K * 8 + 64;
    int64_t nb = (int64_t)std::max<size_t>(64, h->ws_budget / per_q);
    nb = std::min<int64_t>(nb, (int64_t)1 << 22);                                   // kernel arguments are 32-bit
    nb = std::min<int64_t>(nb, std::max<int64_t>(64, ((int64_t)1 << 30) / std::max(1, w * pl.maxch)));
    pl.nb = std::min<int64_t>(nq, nb);
    return IVFADC_OK;
}

int ensure_common_ws(ivfadc_index *h)
{
    const int kc = h->kc;
    TRY(h->list_cnt.ensure((size_t)kc * 4));
    TRY(h->bucket_off.ensure((size_t)(kc + 1) * 4));
    TRY(h->wi_off.ensure((size_t)(kc + 1) * 4));
    TRY(h->cursor.ensure((size_t)kc * 4));
    TRY(h->misc.ensure(MISC_BYTES));
    if (!h->list_cnt_armed) {
        HIP_TRY(hipMemsetAsync(h->list_cnt.p, 0, (size_t)kc * 4, h->stream));
        HIP_TRY(hipMemsetAsync(h->misc.p, 0, MISC_BYTES, h->stream));
        h->list_cnt_armed = true;
    }
    return IVFADC_OK;
}

// want_tmin: also write the per-tile minimum scores (stand-alone top-w with one wave per query reads them)
// want_listed: the only reader is the stand-alone top-w with one wave per query (select_listed): when the split-bf16 kernel
// runs, it writes the four smallest keys of every (query, 64-centroid tile) and NO score matrix (Deep1B shape: 0.16 GB
// instead of 2.7 GB per batch); otherwise ignored
int run_coarse(ivfadc_index *h, const float *d_q, int64_t nb, bool mfma, bool want_tmin = false, bool want_listed = false, int w = 0)
{
    h->tmin_tiles = 0;
    h->last_listed = false;
    const bool big128 = (int64_t)((h->kc + 127) / 128) * ((nb + 127) / 128) >= 2 * (int64_t)h->num_cu;
    // (records pay when the bound -- the w-th smallest of the tile minima -- cuts most tiles: at least 4 w tiles of 64 centroids;
    // with fewer, every tile qualifies and select_listed would fall back to the exact recompute for every query)
    const bool listed = mfma && want_tmin && want_listed && h->allow_listed && big128 && h->allow_bf16 && (h->kc + 63) / 64 >= 4 * std::max(1, w);
    if (!listed) TRY(h->cdist.ensure((size_t)nb * h->kc * 4));
    ivfadc_index::EvPair ep;
    if (h->profiling) TRY(ev_begin(h, 1, ep));
    if (mfma) {
        // scores ||c||^2 - 2 q.c (coarse_mfma_kernel); 64-wide tiles when 128-wide ones would leave CUs idle
        const int64_t wg128 = (int64_t)((h->kc + 127) / 128) * ((nb + 127) / 128);
        const bool big = wg128 >= 2 * (int64_t)h->num_cu;
        const int tb = big ? 128 : 64, tile_w = tb / 2;
        const int ntiles = (h->kc + tile_w - 1) / tile_w;
        float *tmin = nullptr;
        if (want_tmin) {
            TRY(h->tmin.ensure((size_t)nb * ntiles * 4));
            tmin = h->tmin.as<float>();
            h->tmin_tiles = ntiles;
            h->tmin_tile_w = tile_w;
        }
        dim3 grid((h->kc + tb - 1) / tb, (unsigned)((nb + tb - 1) / tb));
        // small problems (every workgroup resident at once): 64-deep chunks, so a workgroup's chain is 2-3 trips to memory
        const bool deep = !big && (int64_t)grid.x * grid.y <= 4 * (int64_t)h->num_cu && h->d >= 64;
        // (small problems stay on the f32 kernels: at kc = 1024 x 1024 queries a 64 x 64-tile bf16 variant plus the split pass took 13.9 us
        // against 15.3 us for the exact VALU kernel -- launch-bound either way -- and the refine costs the scan prologue 4 us)
        h->last_coarse_bf16 = big && h->allow_bf16;
        h->last_coarse_f16 = h->last_coarse_bf16 && h->allow_f16 && h->f16_scale > 0.f && h->cent_f16.p != nullptr;
        if (h->last_coarse_f16) {
            // one f16 product per score (kernels.hip.h, "round 5"): scaled f16 queries + overflow flags, then the same tile kernel
            const int dp = h->dp32;
            TRY(h->q_f16.ensure((size_t)nb * dp * 2));
            TRY(h->q_flags.ensure((size_t)nb * 4));
            HIP_TRY(hipMemsetAsync(h->q_flags.p, 0, (size_t)nb * 4, h->stream));
            hipLaunchKernelGGL(split_f16_kernel, dim3((unsigned)std::min<int64_t>(4096, (nb * dp + 255) / 256)), dim3(256), 0, h->stream, d_q,
                               (int64_t)nb, h->d, dp, h->f16_scale, h->q_f16.as<unsigned short>(), h->q_flags.as<u32>());
            HIP_TRY(hipGetLastError());
            uint4 *tl = nullptr;
            if (listed) {
                TRY(h->tlist.ensure((size_t)ntiles * nb * 16));
                tl = h->tlist.as<uint4>();
                h->last_listed = true;
                h->tlist_ldq = (int)nb;
            }
            const float neg2 = -2.0f / (h->f16_scale * h->f16_scale);
            hipLaunchKernelGGL((coarse_bf16_kernel<128, 0, true>), grid, dim3(256), 0, h->stream, h->q_f16.as<unsigned short>(),
                               (const unsigned short *)nullptr, h->cent_f16.as<unsigned short>(), (const unsigned short *)nullptr,
                               h->cnorm.as<float>(), listed ? (float *)nullptr : h->cdist.as<float>(), (int)nb, h->kc, dp, tmin, ntiles, tl,
                               (int)nb, neg2);
        } else if (h->last_coarse_bf16) {
            // split-bf16 filter: 3 bf16 MFMAs per product instead of one f32 MFMA at a sixteenth of the rate
            const int dp = h->dp32;
            TRY(h->q_hi.ensure((size_t)nb * dp * 2));
            TRY(h->q_lo.ensure((size_t)nb * dp * 2));
            hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)std::min<int64_t>(4096, (nb * dp + 255) / 256)), dim3(256), 0, h->stream, d_q,
                               (int64_t)nb, h->d, dp, h->q_hi.as<unsigned short>(), h->q_lo.as<unsigned short>());
            HIP_TRY(hipGetLastError());
            uint4 *tl = nullptr;
            if (listed) {
                TRY(h->tlist.ensure((size_t)ntiles * nb * 16));
                tl = h->tlist.as<uint4>();
                h->last_listed = true;
                h->tlist_ldq = (int)nb;
            }
#ifdef IVFADC_DEBUG
            // (read per launch: tools/coarse_probe.py runs a correct search first, then sets the variable)
            const int cdbg = env_knob("IVFADC_COARSE_DBG") ? atoi(env_knob("IVFADC_COARSE_DBG")) : 0;
#define IVFADC_COARSE_LAUNCH(D) hipLaunchKernelGGL((coarse_bf16_kernel<128, D>), grid, dim3(256), 0, h->stream, h->q_hi.as<unsigned short>(), \
                               h->q_lo.as<unsigned short>(), h->cent_hi.as<unsigned short>(), h->cent_lo.as<unsigned short>(), \
                               h->cnorm.as<float>(), listed ? (float *)nullptr : h->cdist.as<float>(), (int)nb, h->kc, dp, tmin, ntiles, tl, \
                               (int)nb, -2.0f);
            if (listed && cdbg == 1) { IVFADC_COARSE_LAUNCH(1) }
            else if (listed && cdbg == 2) { IVFADC_COARSE_LAUNCH(2) }
            else if (listed && cdbg == 3) { IVFADC_COARSE_LAUNCH(3) }
            else if (listed && cdbg == 5) { IVFADC_COARSE_LAUNCH(5) }
            else if (listed && cdbg == 6) { IVFADC_COARSE_LAUNCH(6) }
            else if (listed && cdbg == 7) { IVFADC_COARSE_LAUNCH(7) }
            else if (listed && cdbg == 8) { IVFADC_COARSE_LAUNCH(8) }
            else
#undef IVFADC_COARSE_LAUNCH
#endif
            hipLaunchKernelGGL(coarse_bf16_kernel<128>, grid, dim3(256), 0, h->stream, h->q_hi.as<unsigned short>(),
                               h->q_lo.as<unsigned short>(), h->cent_hi.as<unsigned short>(), h->cent_lo.as<unsigned short>(),
                               h->cnorm.as<float>(), listed ? (float *)nullptr : h->cdist.as<float>(), (int)nb, h->kc, dp, tmin, ntiles, tl,
                               (int)nb, -2.0f);
        } else if (big)
            hipLaunchKernelGGL((coarse_mfma_kernel<128, 16>), grid, dim3(256), 0, h->stream, d_q, h->centroids.as<float>(),
                               h->cnorm.as<float>(), h->cdist.as<float>(), (int)nb, h->kc, h->d, tmin, ntiles);
        else if (deep)
            hipLaunchKernelGGL((coarse_mfma_kernel<64, 64>), grid, dim3(256), 0, h->stream, d_q, h->centroids.as<float>(),
                               h->cnorm.as<float>(), h->cdist.as<float>(), (int)nb, h->kc, h->d, tmin, ntiles);
        else
            hipLaunchKernelGGL((coarse_mfma_kernel<64, 16>), grid, dim3(256), 0, h->stream, d_q, h->centroids.as<float>(),
                               h->cnorm.as<float>(), h->cdist.as<float>(), (int)nb, h->kc, h->d, tmin, ntiles);
    } else {
        // small batches: narrower query tiles multiply the workgroup count until every SIMD has its four waves
        const int64_t wg64 = (int64_t)((h->kc + CO_T - 1) / CO_T) * ((nb + 63) / 64);
        int tq = wg64 >= 4 * (int64_t)h->num_cu ? 64 : (2 * wg64 >= 4 * (int64_t)h->num_cu ? 32 : 16);
        static const int force_tq = env_knob("IVFADC_COARSE_TQ") ? atoi(env_knob("IVFADC_COARSE_TQ")) : 0;
        if (force_tq == 16 || force_tq == 32 || force_tq == 64) tq = force_tq;
        dim3 grid((h->kc + CO_T - 1) / CO_T, (unsigned)((nb + tq - 1) / tq));
        // small problems (every workgroup resident at once): centroid per lane, queries in SGPRs -- no LDS traffic to
        // speak of, the wave's instruction stream is the VALU minimum (SIFT1M-shape, 1024 queries: 14.9 -> see DESIGN 4.1)
        static const int sgpr_mode = env_knob("IVFADC_COARSE_SGPR") ? atoi(env_knob("IVFADC_COARSE_SGPR")) : -1;
        const bool sgpr = (h->d & 7) == 0 && (nb + 15) / 16 <= 65535 && (sgpr_mode < 0 ? tq < 64 : sgpr_mode > 0);
        if (sgpr)
            hipLaunchKernelGGL(coarse_sgpr_kernel<4>, dim3((h->kc + 63) / 64, (unsigned)((nb + 15) / 16)), dim3(256), 0, h->stream, d_q,
                               h->centroids.as<float>(), h->cdist.as<float>(), (int)nb, h->kc, h->d);
        else if (tq == 16)
            hipLaunchKernelGGL(coarse_dist_kernel<16>, grid, dim3(256), 0, h->stream, d_q, h->centroids.as<float>(),
                               h->cdist.as<float>(), (int)nb, h->kc, h->d, h->d);
        else if (tq == 32)
            hipLaunchKernelGGL(coarse_dist_kernel<32>, grid, dim3(256), 0, h->stream, d_q, h->centroids.as<float>(),
                               h->cdist.as<float>(), (int)nb, h->kc, h->d, h->d);
        else
            hipLaunchKernelGGL(coarse_dist_kernel<64>, grid, dim3(256), 0, h->stream, d_q, h->centroids.as<float>(),
                               h->cdist.as<float>(), (int)nb, h->kc, h->d, h->d);
    }
    HIP_TRY(hipGetLastError());
    if (h->profiling) TRY(ev_end(h, ep));
    return IVFADC_OK;
}

RefineArgs refine_args(const ivfadc_index *h, const float *d_q)
{
    RefineArgs r;
    r.queries = d_q;
    r.centroids = h->centroids.as<float>();
    r.d = h->d;
    r.kc = h->kc;
    r.cmaxn = h->cmaxn;
    const float u = 5.9604645e-8f;   // 2^-24
    // score error / (||c|| + ||q||)^2.  f32 MFMA: rounded norms + the fma chain on q.c (refine_probes).  Split bf16: the
    // representation x = hi + lo + O(2^-18 |x|) costs (2 * 2^-18 + 2^-18 (dropped lo.lo)) |q||c| <= 0.76 * 2^-18 (||c|| + ||q||)^2
    // on q.c, doubled in the score: 1.51 * 2^-18 = 96.6 u; the f32 accumulation of 3 d products (+ padding) another
    // 0.51 (3 d + 40) u; norms and final roundings (d + 8) u as before.
    // One f16 product (kernels.hip.h, "round 5"): x^ = fl16(s x) / s, |x^ - x| <= 2^-11 |x| + 2^-25 / s, so q^ . c^ misses q . c by at most
    // (2^-10 + 2^-22) |q||c| <= 0.25 (2^-10 + 2^-22)(||c|| + ||q||)^2, doubled in the score: 8192 (1 + 2^-12) u; the subnormal term stays
    // below 2^-36 sqrt(d) of the same square (2 u with room to spare); f32 accumulation of d products and the norms as before.
    r.eps_coef = h->last_coarse_f16 ? 2.0f * (8194.0f + 2.0f + 0.51f * (float)(h->d + 40) + (float)(h->d + 8)) * u
               : h->last_coarse_bf16 ? 2.0f * (97.0f + 0.51f * (float)(3 * h->d + 40) + (float)(h->d + 8)) * u
                                     : 2.0f * (float)(h->d + 3) * u;
    r.qflags = h->last_coarse_f16 ? h->q_flags.as<u32>() : (const u32 *)nullptr;
    // listed mode: a key carries the score with its low 6 bits replaced (< 64 ulp = 2^-17 |score|, |score| <= (||c|| + ||q||)^2)
    if (h->last_listed) r.eps_coef += 128.0f * u;
    r.tlist = h->last_listed ? h->tlist.as<uint4>() : (const uint4 *)nullptr;
    r.ldq = h->tlist_ldq;
    r.gam = 4.0f * (float)(h->d + 2) * u;
    r.fallbacks = (u64 *)((char *)h->misc.p + 4096 + 64);
    r.tmin = h->tmin_tiles > 0 ? h->tmin.as<float>() : (const float *)nullptr;
    r.ntiles = h->tmin_tiles;
    r.tile_w = h->tmin_tile_w;
    return r;
}

IndexView index_view(const ivfadc_index *h)
{
    IndexView ix;
    ix.centroids = h->centroids.as<float>();
    ix.codebooks = h->codebooks.as<float>();
    ix.codebooks_t = h->codebooks_t.as<float>();
    static const bool no_pk = env_knob("IVFADC_NO_PK_BUILD") != nullptr;
    ix.codebooks_p = (h->codebooks_p.p && !no_pk) ? h->codebooks_p.as<float>() : (const float *)nullptr;
    ix.labels = h->labels.as<uint8_t>();
    ix.codes = h->codes.as<uint8_t>();
    ix.list_pos = h->list_pos.as<int64_t>();
    ix.list_len = h->list_len.as<u32>();
    ix.list_codeoff = h->list_codeoff.as<int64_t>();
    ix.ids = h->synthetic ? (const u32 *)nullptr : h->ids.as<u32>();
    ix.d = h->d; ix.kc = h->kc; ix.m = h->m; ix.ksub = h->ksub; ix.dsub = h->dsub; ix.cs = h->cs;
    ix.identity_labels = h->identity_labels ? 1 : 0;
#ifdef IVFADC_DEBUG
    static const int dbg_flags = env_knob("IVFADC_DEBUG_FLAGS") ? atoi(env_knob("IVFADC_DEBUG_FLAGS")) : 0;
    ix.dbg_flags = dbg_flags;
#else
    ix.dbg_flags = 0;
#endif
    return ix;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (function, device), not of a handle: one process-wide table
// per device records the largest size ever requested for a function and only ever RAISES the attribute, so two live
// handles that share a kernel instantiation with different LDS sizes (another K, another m through the <0, 0> kernels)
// cannot leave each other with an attribute below their launch size.  Occupancy is cached per (function, size).
struct FnAttr { int device; const void *fn; size_t max_lds; bool checked; };
std::mutex g_fn_mu;
std::vector<FnAttr> g_fn_attr;

int fn_raise_lds(int device, const void *fn, size_t lds, bool need_no_static)
{
    std::lock_guard<std::mutex> lk(g_fn_mu);
    FnAttr *e = nullptr;
    for (auto &c : g_fn_attr)
        if (c.device == device && c.fn == fn) e = &c;
    if (!e) {
        g_fn_attr.push_back({device, fn, 0, false});
        e = &g_fn_attr.back();
    }
    if (need_no_static && !e->checked) {
        // the scan kernels address their tables by absolute LDS offsets (lds_load_abs): the dynamic segment must start at 0
        hipFuncAttributes fa;
        HIP_TRY(hipFuncGetAttributes(&fa, fn));
        if (fa.sharedSizeBytes != 0) return fail(IVFADC_ERR_STATE, "scan kernel carries %zu B of static LDS", (size_t)fa.sharedSizeBytes);
        e->checked = true;
    }
    if (lds > e->max_lds) {
        HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        e->max_lds = lds;
    }
    return IVFADC_OK;
}

int fn_occupancy(ivfadc_index *h, const void *fn, size_t lds, int &occ, bool abs_lds = true, int threads = 256)
{
    TRY(fn_raise_lds(h->device, fn, lds, abs_lds));
    occ = 0;
    for (auto &c : h->fn_cfg)
        if (c.fn == fn && c.lds == lds) occ = c.occ;
    if (occ == 0) {
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, threads, lds));
        occ = std::max(1, std::min(occ, 8));
        h->fn_cfg.push_back({fn, lds, occ});
    }
    return IVFADC_OK;
}

// coarse stage by the certified two-level search: distances to the G group centres (the exact tile kernel), then one wave per query
int run_twolevel(ivfadc_index *h, const float *d_q, int64_t nb, int w, int *d_probe_list, float *d_probe_dc, u32 *d_probe_base, u32 *d_list_cnt,
                 u64 *d_scanned)
{
    const int G = h->tl_G;
    TRY(h->tl_gdist.ensure((size_t)nb * G * 4));
    ivfadc_index::EvPair ep;
    if (h->profiling) TRY(ev_begin(h, 1, ep));
    {
        const int64_t wg64 = (int64_t)((G + CO_T - 1) / CO_T) * ((nb + 63) / 64);
        const int tq = wg64 >= 4 * (int64_t)h->num_cu ? 64 : (2 * wg64 >= 4 * (int64_t)h->num_cu ? 32 : 16);
        dim3 grid((G + CO_T - 1) / CO_T, (unsigned)((nb + tq - 1) / tq));
        if (tq == 16)
            hipLaunchKernelGGL(coarse_dist_kernel<16>, grid, dim3(256), 0, h->stream, d_q, h->tl_centres.as<float>(), h->tl_gdist.as<float>(), (int)nb, G, h->d, h->d);
        else if (tq == 32)
            hipLaunchKernelGGL(coarse_dist_kernel<32>, grid, dim3(256), 0, h->stream, d_q, h->tl_centres.as<float>(), h->tl_gdist.as<float>(), (int)nb, G, h->d, h->d);
        else
            hipLaunchKernelGGL(coarse_dist_kernel<64>, grid, dim3(256), 0, h->stream, d_q, h->tl_centres.as<float>(), h->tl_gdist.as<float>(), (int)nb, G, h->d, h->d);
        HIP_TRY(hipGetLastError());
    }
    TwoLevelView tv;
    tv.gdist = h->tl_gdist.as<float>();
    tv.g_off = h->tl_off.as<u32>();
    tv.g_rad = h->tl_rad.as<float>();
    tv.cent_g = h->tl_cent.as<float>();
    tv.slot_id = h->tl_slot.as<u32>();
    tv.G = G;
    tv.d = h->d;
    tv.eps = h->tl_eps;
    const size_t lds = (size_t)4 * h->d * 4 + 4 * 64 * 8;
    if (lds > (size_t)(48 << 10)) TRY(fn_raise_lds(h->device, (const void *)twolevel_topw_kernel, lds, false));
    hipLaunchKernelGGL(twolevel_topw_kernel, dim3((unsigned)((nb + 3) / 4)), dim3(256), lds, h->stream, d_q, tv, (int)nb, w, h->list_len.as<u32>(),
                       d_probe_list, d_probe_dc, d_probe_base, d_list_cnt, d_scanned, h->part_n, h->part_i);
    HIP_TRY(hipGetLastError());
    if (h->profiling) TRY(ev_end(h, ep));
    return IVFADC_OK;
}

int search_subbatch(ivfadc_index *h, const Plan &pl, int64_t nb, const float *d_q, int K, int w, uint32_t *d_ids,
                    float *d_dists, int32_t *d_counts, bool single)   // single: the call's whole batch (hints and prefetched rows apply)
{
    const int kc = h->kc;
    const size_t np = (size_t)nb * w;
    TRY(ensure_common_ws(h));
    TRY(h->probe_list.ensure(np * 4));
    TRY(h->probe_dc.ensure(np * 4));
    TRY(h->probe_base.ensure(np * 4));
    u64 *d_scanned = h->misc.as<u64>();          // 64 sharded counters
    u32 *d_qhead = (u32 *)((char *)h->misc.p + 4096);
    // list-major with one query per code stream: every (query, probe) pair is a work item of its own, nothing to group
    // by list -- no probe histogram, no bucket kernels
    const bool direct = !pl.query_major && pl.qg == 1 && np * (size_t)pl.maxch < ((size_t)1 << 31) && h->part_n <= 1;

    // one wave per query leaves the chip empty on small batches: the stand-alone top-w uses a workgroup per query there
    static const bool wpq1_env = env_knob("IVFADC_TOPW_WPQ1") != nullptr;   // A/B: a wave per query also on small batches (throughput runs with several batches in flight)
    const bool wpq4 = !wpq1_env && !pl.lanes && nb * 1 < (int64_t)8 * h->num_cu * 4 && h->kc >= 512 && !(pl.lb && !pl.fuse_topw);
    static const bool no_tmin = env_knob("IVFADC_NO_TILE_MIN") != nullptr;
    // the rows of these very queries may stand already: written by the previous search's launch behind a hint (ivfadc_set_next_queries)
    const bool have_rows = single && !pl.coarse_mfma && !pl.twolevel && h->avail_q == d_q && h->avail_nq == nb && h->cdist2.p != nullptr;
    h->avail_q = nullptr;
    h->stats.last_rider = 0;
    h->stats.coarse_prefetched = have_rows ? 1 : 0;
    h->stats.last_twolevel = pl.twolevel ? 1 : 0;
    if (pl.twolevel) {
        u32 *lc = (pl.query_major || direct) ? (u32 *)nullptr : h->list_cnt.as<u32>();
        TRY(run_twolevel(h, d_q, nb, w, h->probe_list.as<int>(), h->probe_dc.as<float>(), h->probe_base.as<u32>(), lc, d_scanned));
        h->tmin_tiles = 0;
        h->last_listed = false;
    } else if (have_rows) {
        std::swap(h->cdist, h->cdist2);
        h->tmin_tiles = 0;
        h->last_listed = false;
    } else {
        TRY(run_coarse(h, d_q, nb, pl.coarse_mfma, pl.coarse_mfma && !no_tmin && (pl.fuse_topw ? h->m > 16 : !wpq4),   // who reads them
                       !pl.fuse_topw && !wpq4, w));
    }
    // riders: the hinted batch will take the exact small-problem coarse kernel whatever its K and w (no matrix-core filter at this kc)
    const bool ride = single && !h->tl_use && h->hint_q != nullptr && h->hint_nq > 0 && pl.query_major && !pl.lb && pl.small_k && (h->d & 7) == 0 &&
                      (!h->allow_mfma || kc < h->mfma_min_kc) && (h->hint_nq + 4 * RIDER_QW - 1) / (4 * RIDER_QW) <= 65535 &&
                      (size_t)h->hint_nq * kc * 4 <= h->ws_budget / 4;

    if (!pl.fuse_topw && !pl.twolevel) {
        u32 *lc = (pl.query_major || direct) ? (u32 *)nullptr : h->list_cnt.as<u32>();   // probe histogram: grouped list-major only
        const size_t lds = (size_t)4 * pl.capw * 8;
        void (*fn)(const float *, int, int, int, int, const u32 *, int *, float *, u32 *, u32 *, u64 *, const RefineArgs, int, int);
        if (pl.coarse_mfma)   // implies w <= 48: register selectors
            fn = wpq4 ? topw_select_kernel<true, 4, true> : topw_select_kernel<true, 1, true>;
        else if (pl.small_w)
            fn = wpq4 ? topw_select_kernel<true, 4, false> : topw_select_kernel<true, 1, false>;
        else
            fn = wpq4 ? topw_select_kernel<false, 4, false> : topw_select_kernel<false, 1, false>;
        const unsigned grid = wpq4 ? (unsigned)nb : (unsigned)((nb + 3) / 4);
        if (lds > (size_t)(32 << 10)) { int occ_unused = 0; TRY(fn_occupancy(h, (const void *)fn, lds, occ_unused, false)); }
        hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, h->stream, h->cdist.as<float>(), (int)nb, kc, w, pl.capw,
                           h->list_len.as<u32>(), h->probe_list.as<int>(), h->probe_dc.as<float>(), h->probe_base.as<u32>(), lc,
                           d_scanned, refine_args(h, d_q), h->part_n, h->part_i);
        HIP_TRY(hipGetLastError());
    }
    h->stats.last_qg = pl.query_major ? 0 : pl.qg;
    h->stats.coarse_mfma = pl.coarse_mfma ? 1 : 0;
    h->stats.coarse_f16 = (pl.coarse_mfma && h->last_coarse_f16 && !have_rows) ? 1 : 0;
    h->stats.coarse_listed = h->last_listed ? 1 : 0;
    h->stats.last_chunk = (int)pl.CH;
    h->stats.last_scan_lds = (int)pl.lds;

    if (pl.query_major) {
        QScanArgs a;
        a.ix = index_view(h);
        a.queries = d_q;
        a.nq = (int)nb; a.w = w; a.K = K; a.cap = pl.cap;
        a.probe_list = h->probe_list.as<int>();
        a.probe_dc = h->probe_dc.as<float>();
        a.probe_base = h->probe_base.as<u32>();
        a.out_ids = d_ids;
        a.out_dists = d_dists;
        a.out_counts = d_counts;
        a.cdist = pl.fuse_topw ? h->cdist.as<float>() : (const float *)nullptr;
        a.scanned_points = d_scanned;
        a.approx = pl.coarse_mfma ? 1 : 0;
        a.rf = refine_args(h, d_q);
        a.prune = h->allow_prune ? 1 : 0;
        a.dbg = nullptr;
        a.lb.cb_split = h->lb_split.as<uint4>();
        a.lb.cb_n2 = h->lb_n2.as<float>();
        a.lb.cb_lab = h->lb_lab.as<float>();
        a.lb.cb_maxn = h->lb_maxn.as<float>();
        a.lb.cb_f16 = (h->lb_use_f16 && h->lb_f16.p) ? h->lb_f16.as<uint4>() : (const uint4 *)nullptr;
        a.lb.cb_isc = h->lb_isc.as<float>();
        a.lb.mu = a.lb.cb_f16 ? 1.48e-3f : 9.2e-5f;
        h->stats.last_lb = pl.lb ? 1 : 0;
#ifdef IVFADC_DEBUG
        static const bool dbg_on = env_knob("IVFADC_DEBUG_STAMPS") != nullptr;
#else
        constexpr bool dbg_on = false;
#endif
        if (dbg_on) {
            TRY(h->dbg.ensure((size_t)nb * 128));
            a.dbg = h->dbg.as<u64>();
        }
        qscan_fn_t fn = pl.lb ? pick_qscan_lb(h->m, h->dsub, pl.qg) : pick_qscan(h->m, h->dsub, pl.qg, pl.small_k);
        int occ = 0;
        TRY(fn_occupancy(h, (const void *)fn, pl.lds, occ));
        ivfadc_index::EvPair ep;
        // a hinted next batch (ivfadc_set_next_queries): its exact coarse tiles ride behind this batch's scan in the same grid
        void (*fk)(const QScanArgs, const CoarseNext) = nullptr;
        if (ride) fk = pick_qscan_coarse(h->m, h->dsub, pl.qg);
        if (fk) {
            TRY(h->cdist2.ensure((size_t)h->hint_nq * kc * 4));
            CoarseNext cn;
            cn.queries = h->hint_q; cn.out = h->cdist2.as<float>(); cn.nq = (int)h->hint_nq; cn.ncx = (kc + 63) / 64;
            const size_t lds = std::max<size_t>(pl.lds, (size_t)64 * 132 * 4);
            TRY(fn_raise_lds(h->device, (const void *)fk, lds, true));
            const unsigned grid = (unsigned)(nb + (int64_t)cn.ncx * ((h->hint_nq + 4 * RIDER_QW - 1) / (4 * RIDER_QW)));
            if (h->hint_ev) {   // the hinted rows' ingest (ivfadc_search_batches): in front of the one launch that reads them
                HIP_TRY(hipStreamWaitEvent(h->stream, h->hint_ev, 0));
                h->hint_ev = nullptr;
            }
            if (h->profiling) TRY(ev_begin(h, 0, ep));
            hipLaunchKernelGGL(fk, dim3(grid), dim3(256), lds, h->stream, a, cn);
            HIP_TRY(hipGetLastError());
            if (h->profiling) TRY(ev_end(h, ep));
            h->pf_q = h->hint_q;
            h->pf_nq = h->hint_nq;
            h->pf_token = h->hint_token;
            h->stats.last_rider = 1;
        } else {
            if (h->profiling) TRY(ev_begin(h, 0, ep));
            hipLaunchKernelGGL(fn, dim3((unsigned)nb), dim3(256), pl.lds, h->stream, a);
            HIP_TRY(hipGetLastError());
            if (h->profiling) TRY(ev_end(h, ep));
        }
        h->stats.last_scan_grid = (int)nb;
        if (!pl.lb) TRY(fb_snapshot(h));
        if (h->profiling_level >= 2 && pl.lb && !pl.fuse_topw && pl.qg == 4) {
            // the table build alone, over the probes this batch used (measurement only)
            void (*bk)(const IndexView, const LbView, const float *, const int *, int, u32 *) =
                (h->m == 48) ? lb_build_only_kernel<48, 16, 4> : lb_build_only_kernel<16, 6, 4>;
            TRY(fn_raise_lds(h->device, (const void *)bk, pl.lds, false));
            TRY(h->dbg.ensure((size_t)nb * 4));
            ivfadc_index::EvPair eb;
            TRY(ev_begin(h, 2, eb));
            hipLaunchKernelGGL(bk, dim3((unsigned)nb), dim3(256), pl.lds, h->stream, a.ix, a.lb, d_q, h->probe_list.as<int>(), w,
                               h->dbg.as<u32>());
            HIP_TRY(hipGetLastError());
            TRY(ev_end(h, eb));
        }
        if (dbg_on) {
            std::vector<u64> st((size_t)nb * 16);
            HIP_TRY(hipMemcpyAsync(st.data(), h->dbg.p, st.size() * 8, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            double acc[16] = {0};
            for (int64_t i = 0; i < nb; ++i)
                for (int k = 0; k < 16; ++k) acc[k] += (double)st[i * 16 + k];
            fprintf(stderr, "[ivfadc stamps] prologue (wave 0): to first barrier=%.0f row select+store=%.0f barrier=%.0f merge=%.0f lens+prefix=%.0f "
                            "barrier=%.0f\n", acc[8] / nb, acc[9] / nb, acc[10] / nb, acc[11] / nb, acc[12] / nb, acc[13] / nb);
            fprintf(stderr, "[ivfadc stamps] per-WG mean cycles: wait_prev=%.0f resid=%.0f table=%.0f scan=%.0f | prologue+loop=%.0f "
                            "tail=%.0f | top-w row select (wave 0)=%.0f\n", acc[0] / nb, acc[1] / nb, acc[2] / nb, acc[3] / nb, acc[4] / nb,
                    acc[5] / nb, acc[6] / nb);
            if (pl.lb)
                fprintf(stderr, "[ivfadc stamps] lower-bound rounds (wave 0): resid = setup (norms, scales, bf16 residuals), table = MFMA build, "
                                "scan = integer scan + drains; final drains=%.0f rounds=%.2f\n", acc[14] / nb, acc[15] / nb);
            // balance: every workgroup is resident from the start, so the launch lasts as long as its slowest one
            std::vector<double> dur((size_t)nb);
            u64 t_first = ~0ull, t_last = 0;
            for (int64_t i = 0; i < nb; ++i) {
                const u64 du = st[i * 16 + 4] + st[i * 16 + 5];
                dur[i] = (double)du;
                t_first = std::min(t_first, st[i * 16 + 7] - du);
                t_last = std::max(t_last, st[i * 16 + 7]);
            }
            std::sort(dur.begin(), dur.end());
            double mean = 0;
            for (double v : dur) mean += v;
            mean /= (double)nb;
            fprintf(stderr, "[ivfadc stamps] workgroup duration cycles: mean=%.0f p50=%.0f p90=%.0f p99=%.0f max=%.0f | first start -> last end=%.0f\n",
                    mean, dur[nb / 2], dur[(size_t)(nb * 0.9)], dur[(size_t)(nb * 0.99)], dur[nb - 1], (double)(t_last - t_first));
        }
    } else {
        TRY(h->bucket_items.ensure(np * 4));
        TRY(h->part_keys.ensure(np * pl.maxch * K * 8));
        TRY(h->part_cnt.ensure(np * pl.maxch * 4));
        {
            const size_t before = h->qthr.bytes;
            TRY(h->qthr.ensure((size_t)nb * 8));
            if (h->qthr.bytes != before) h->qthr_armed = 0;
            if (h->qthr_armed < (size_t)nb) {
                const size_t cnt = h->qthr.bytes / 8;
                hipLaunchKernelGGL(fill_u64_kernel, dim3(256), dim3(256), 0, h->stream, h->qthr.as<u64>(), cnt, (u64)KEY_MAX);
                HIP_TRY(hipGetLastError());
                h->qthr_armed = cnt;
            }
        }
        if (!direct) {
            u32 *item_list = nullptr;
            if (pl.wg8) {
                TRY(h->wg8_items.ensure(np * (size_t)pl.maxch * 4));
                item_list = h->wg8_items.as<u32>();
            }
            hipLaunchKernelGGL(bucket_scan_kernel, dim3(1), dim3(1024), 0, h->stream, h->list_cnt.as<u32>(), h->list_len.as<u32>(),
                               kc, pl.qg, pl.CH, h->bucket_off.as<u32>(), h->wi_off.as<u32>(), h->cursor.as<u32>(), d_qhead, item_list);
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL(bucket_scatter_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, h->stream,
                               h->probe_list.as<int>(), (int)np, h->bucket_off.as<u32>(), h->cursor.as<u32>(), h->bucket_items.as<u32>(),
                               h->part_n, h->part_i);
            HIP_TRY(hipGetLastError());
        }

        ScanArgs a;
        a.ix = index_view(h);
        a.queries = d_q;
        a.w = w; a.K = K; a.cap = pl.cap;
        a.probe_dc = h->probe_dc.as<float>();
        a.probe_base = h->probe_base.as<u32>();
        a.list_cnt = h->list_cnt.as<u32>();
        a.bucket_off = h->bucket_off.as<u32>();
        a.wi_off = h->wi_off.as<u32>();
        a.bucket_items = h->bucket_items.as<u32>();
        a.queue_head = d_qhead;
        a.qthr = h->qthr.as<u64>();
        a.part_keys = h->part_keys.as<u64>();
        a.part_cnt = h->part_cnt.as<u32>();
        a.maxch = pl.maxch;
        a.CH = pl.CH;
        a.probe_list = h->probe_list.as<int>();
        a.direct_items = direct ? (u32)(np * (size_t)pl.maxch) : 0u;
        a.prune = h->allow_prune ? 1 : 0;
        a.scanned_points = d_scanned;

        const bool stripe = h->allow_filt && filt_shape(h->m, h->dsub) && pl.qg == 4 && h->ksub == 256;
        h->stats.last_striped = stripe ? 1 : 0;
        h->stats.last_nf = pl.nf ? 1 : 0;
        const size_t upper = np * (size_t)pl.maxch;
        ivfadc_index::EvPair ep;
        if (pl.wg8 && !direct) {
            void (*wk)(const ScanArgs, float *, const u32 *, u32 *, int) = pl.wg8q8 ? wg8q8_scan_kernel : wg8_scan_kernel;
            u32 *xq = (u32 *)((char *)h->misc.p + 4096 + 256);     // eight queue heads, 64 B apart (as the narrow-field kernel's)
            static const bool one_queue = env_knob("IVFADC_W8_NO_XCD") != nullptr;   // A/B (debug build): one queue for all workgroups
#ifdef W8_ONE_QUEUE
            const int nranges = 1;
#else
            const int nranges = one_queue ? 1 : 8;
#endif
            HIP_TRY(hipMemsetAsync(xq, 0, 512, h->stream));
            int occ = 0;
            TRY(fn_occupancy(h, (const void *)wk, pl.lds, occ, true, W8_THREADS));
            const unsigned grid = (unsigned)std::max<size_t>(1, std::min<size_t>(upper, (size_t)h->num_cu * occ));
            TRY(h->wg8_tabs.ensure((size_t)h->num_cu * 8 * W9_GTAB_FLOATS * 4 + 256));   // (occupancy is clamped to 8 workgroups per CU; 256 B: W8_PROF's counters)
#ifdef W8_PROF
            HIP_TRY(hipMemsetAsync((char *)h->wg8_tabs.p + (size_t)grid * (pl.wg8q8 ? W9_GTAB_FLOATS : W8_GTAB_FLOATS) * 4, 0, 128, h->stream));
#endif
            if (h->profiling) TRY(ev_begin(h, 0, ep));
            hipLaunchKernelGGL(wk, dim3(grid), dim3(W8_THREADS), pl.lds, h->stream, a, h->wg8_tabs.as<float>(), h->wg8_items.as<u32>(), xq, nranges);
            HIP_TRY(hipGetLastError());
            if (h->profiling) TRY(ev_end(h, ep));
            h->stats.last_scan_grid = (int)grid;
            h->stats.last_striped = pl.wg8q8 ? 3 : 2;
#ifdef W8_PROF
            {
                u64 pc[16];
                HIP_TRY(hipMemcpyAsync(pc, (char *)h->wg8_tabs.p + (size_t)grid * (pl.wg8q8 ? W9_GTAB_FLOATS : W8_GTAB_FLOATS) * 4, 128, hipMemcpyDeviceToHost, h->stream));
                HIP_TRY(hipStreamSynchronize(h->stream));
                static int shown = 0;
                if (shown++ % 8 == 4) {
                    const double wv_ = (double)grid * (double)W8_NW;
                    fprintf(stderr, "[w8prof] per wave: loop %.0f setup %.0f build %.0f scan %.0f (cand %.0f, drains %.0f of it waiting %.0f) merge %.0f cycles | steps %.0f cand-steps %.0f drains %.0f "
                                    "drained %.0f crowds %.0f build: residuals %.0f entries %.0f items %.1f\n", pc[0] / wv_, pc[1] / wv_, pc[2] / wv_, pc[3] / wv_, pc[4] / wv_, pc[5] / wv_, pc[7] / wv_, pc[6] / wv_,
                            pc[8] / wv_, pc[9] / wv_, pc[10] / wv_, pc[11] / wv_, pc[12] / wv_, pc[13] / wv_, (pc[15] - pc[13]) / wv_, pc[14] / wv_);
                }
            }
#endif
        } else if (pl.nf) {
            // four points per lane and step (measured: two 9.2 ms, eight 8.1 ms -- and 74 spilled registers -- against 7.76 ms)
            void (*nk)(const ScanArgs, const NfView) = nf_scan_kernel<4>;
            NfView nv;
            nv.n2 = h->nf_n2.as<float>();
            nv.cb_lab = h->nf_lab.as<float>();
            nv.maxn2 = h->nf_n2.as<float>() + (size_t)h->m * 256;
            nv.xq = (u32 *)((char *)h->misc.p + 4096 + 256);
            static const bool no_xcd = env_knob("IVFADC_NF_NO_XCD") != nullptr;   // A/B: one queue for all workgroups
            nv.nranges = no_xcd ? 1 : 8;
            HIP_TRY(hipMemsetAsync(nv.xq, 0, 512, h->stream));
            int occ = 0;
            TRY(fn_occupancy(h, (const void *)nk, pl.lds, occ));
            const unsigned grid = (unsigned)std::max<size_t>(1, std::min<size_t>(upper, (size_t)h->num_cu * occ));
            if (h->profiling) TRY(ev_begin(h, 0, ep));
            hipLaunchKernelGGL(nk, dim3(grid), dim3(256), pl.lds, h->stream, a, nv);
            HIP_TRY(hipGetLastError());
            if (h->profiling) TRY(ev_end(h, ep));
            h->stats.last_scan_grid = (int)grid;
        } else {
        scan_fn_t fn = pick_scan(h->m, h->dsub, pl.qg, pl.small_k, stripe);
        int occ = 0;
        // IVFADC_LDS_PAD (bytes, diagnostic): unused LDS behind the kernel's own, to see what a workgroup per CU fewer costs
        static const size_t lds_pad = env_knob("IVFADC_LDS_PAD") ? (size_t)atol(env_knob("IVFADC_LDS_PAD")) : 0;
        const size_t lds_launch = std::min<size_t>(LDS_MAX, pl.lds + lds_pad);
        TRY(fn_occupancy(h, (const void *)fn, lds_launch, occ));
        const unsigned grid = (unsigned)std::max<size_t>(1, std::min<size_t>(upper, (size_t)h->num_cu * occ));
        if (h->profiling) TRY(ev_begin(h, 0, ep));
        hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds_launch, h->stream, a);
        HIP_TRY(hipGetLastError());
        if (h->profiling) TRY(ev_end(h, ep));
        h->stats.last_scan_grid = (int)grid;
        }

        const size_t mlds = pl.small_k ? 0 : (size_t)4 * pl.cap * 8;
        const u32 *idp = h->synthetic ? (const u32 *)nullptr : h->ids.as<u32>();
        if (mlds > (size_t)(32 << 10)) { int occ_unused = 0; TRY(fn_occupancy(h, (const void *)merge_kernel<false>, mlds, occ_unused, false)); }
        if (pl.small_k)
            hipLaunchKernelGGL(merge_kernel<true>, dim3((unsigned)((nb + 3) / 4)), dim3(256), mlds, h->stream, (int)nb, w, K, pl.cap,
                               pl.maxch, pl.CH, kc, h->probe_list.as<int>(), h->probe_base.as<u32>(), h->list_pos.as<int64_t>(),
                               h->list_len.as<u32>(), idp,
                               h->part_keys.as<u64>(), h->part_cnt.as<u32>(), d_ids, d_dists, d_counts, h->qthr.as<u64>(),
                               h->list_cnt.as<u32>(), d_qhead, h->part_n, h->part_i, (u64 *)h->partial_keys);
        else
            hipLaunchKernelGGL(merge_kernel<false>, dim3((unsigned)((nb + 3) / 4)), dim3(256), mlds, h->stream, (int)nb, w, K, pl.cap,
                               pl.maxch, pl.CH, kc, h->probe_list.as<int>(), h->probe_base.as<u32>(), h->list_pos.as<int64_t>(),
                               h->list_len.as<u32>(), idp,
                               h->part_keys.as<u64>(), h->part_cnt.as<u32>(), d_ids, d_dists, d_counts, h->qthr.as<u64>(),
                               h->list_cnt.as<u32>(), d_qhead, h->part_n, h->part_i, (u64 *)h->partial_keys);
        HIP_TRY(hipGetLastError());
    }
    h->stats.queries += nb;
    if (h->profiling && h->pending.size() > 2048) TRY(ev_fold(h));
    return IVFADC_OK;
}

// mutators: not on a view; every other holder of the device arrays learns that they changed
int check_mutable(ivfadc_index *h, const char *what)
{
    if (h->is_view) return fail(IVFADC_ERR_STATE, "%s: this handle is a read-only view (ivfadc_clone_view)", what);
    return IVFADC_OK;
}

// A mutator's critical section.  begin() is called once the arguments have been validated and the call is known to change something (a
// failing or no-op call leaves the views, and the internal second lane of ivfadc_search_batches, valid): from there until the scope ends
// the topology mutex and every view's mutex are held -- no view call is running or can start -- the views' streams have drained, and
// the generation has moved on, so the first view search afterwards refuses.
struct MutationScope {
    ivfadc_index *h = nullptr;
    std::vector<ivfadc_index *> held;
    bool topo = false;
    int begin(ivfadc_index *x, const char *what)
    {
        TRY(check_mutable(x, what));
        if (h) return IVFADC_OK;   // (once per call)
        h = x;
        g_topo_mu.lock();
        topo = true;
        for (ivfadc_index *v : h->views) { v->mu.m.lock(); held.push_back(v); }
        h->generation++;
        // searches still in flight on views read the arrays that are about to change: they finish first
        if (!h->views.empty()) TRY(set_device(h));
        for (ivfadc_index *v : h->views)
            if (v->stream) HIP_TRY(hipStreamSynchronize(v->stream));
        return IVFADC_OK;
    }
    ~MutationScope()
    {
        for (ivfadc_index *v : held) v->mu.m.unlock();
        if (topo) g_topo_mu.unlock();
    }
};

// Called with the view's own mutex held (every entry point's HandleLock), and with that alone: whoever changes what is read here -- a
// mutator bumping the generation (MutationScope), the index's destructor cutting the link -- holds this view's mutex while doing so, so
// the topology mutex is NOT taken (taking it here, behind the view's own mutex, would invert the order a mutator takes the two in).
int check_view_current(ivfadc_index *h)
{
    if (!h->is_view) return IVFADC_OK;
    if (h->orphan || !h->view_of) return fail(IVFADC_ERR_STATE, "the index this view was taken from has been destroyed");
    if (h->view_of->generation != h->view_gen) return fail(IVFADC_ERR_STATE, "the index has changed since this view was taken: take a new one");
    return IVFADC_OK;
}

int check_search_args(ivfadc_index *h, int64_t nq, int K, int &w)
{
    if (!h) return fail(IVFADC_ERR_INVALID, "null handle");
    TRY(check_view_current(h));
    if (K < 1) return fail(IVFADC_ERR_ASSERT, "Number of neighbors must be k >= 1");
    if (w < 1) return fail(IVFADC_ERR_ASSERT, "Number of clusters to search in must be w >= 1");
    if (nq < 0) return fail(IVFADC_ERR_INVALID, "nq < 0");
    if (!h->have_lists && !h->dirty) return fail(IVFADC_ERR_STATE, "no inverted lists set");
    w = std::min(w, h->kc);
    if ((int64_t)nq * K > ((int64_t)1 << 40)) return fail(IVFADC_ERR_INVALID, "nq x K = %lld result slots", (long long)nq * K);
    return IVFADC_OK;
}

// ---- generic path (generic.hip.h): dump every key, sort, take the first K ------------------------------------
int gen_sort(ivfadc_index *h, const u64 *in, u64 *out, int64_t total, int segments, const u32 *d_off)
{
    if (total == 0) return IVFADC_OK;
    size_t tmp = 0;
    HIP_TRY(rocprim::segmented_radix_sort_keys(nullptr, tmp, in, out, (unsigned)total, (unsigned)segments, d_off, d_off + 1, 0, 64,
                                               h->stream));
    TRY(h->gen_tmp.ensure(std::max<size_t>(tmp, 16)));
    HIP_TRY(rocprim::segmented_radix_sort_keys(h->gen_tmp.p, tmp, in, out, (unsigned)total, (unsigned)segments, d_off, d_off + 1, 0, 64,
                                               h->stream));
    return IVFADC_OK;
}

int search_generic(ivfadc_index *h, int64_t nq, const float *d_q, int K, int w, uint32_t *d_ids, float *d_dists, int32_t *d_counts)
{
    const int kc = h->kc;
    const size_t lds = (align_up((size_t)h->d, 4) + (size_t)h->m * 256) * 4;
    if (lds > LDS_MAX) return fail(IVFADC_ERR_INVALID, "m=%d needs %zu B of LDS in the generic path (> %zu)", h->m, lds, LDS_MAX);
    TRY(fn_raise_lds(h->device, (const void *)gen_dump_kernel, lds, false));
    TRY(ensure_common_ws(h));
    u64 *d_scanned = h->misc.as<u64>();
    const IndexView ix = index_view(h);
    const u32 *idp = h->synthetic ? (const u32 *)nullptr : h->ids.as<u32>();
    // stage A batches: the kc-key rows of a batch are sorted in one call (< 2^32 keys, inside the workspace budget)
    const int64_t per_q = (int64_t)kc * (4 + 16) + (int64_t)w * 12 + 64;
    int64_t nba = std::max<int64_t>(1, (int64_t)(h->ws_budget / (size_t)per_q));
    nba = std::min<int64_t>(std::min<int64_t>(nba, nq), std::min<int64_t>(65535, ((int64_t)1 << 31) / std::max(1, kc)));
    std::vector<u32> tot, off;
    for (int64_t a0 = 0; a0 < nq; a0 += nba) {
        const int64_t na = std::min(nba, nq - a0);
        const float *qa = d_q + (size_t)a0 * h->d;
        // coarse distances (exact VALU kernel), rows -> keys -> sorted rows -> probes
        TRY(run_coarse(h, qa, na, false));
        const int64_t rk = na * kc;
        TRY(h->gen_a.ensure((size_t)rk * 8));
        TRY(h->gen_b.ensure((size_t)rk * 8));
        hipLaunchKernelGGL(gen_row_keys_kernel, dim3((unsigned)std::min<int64_t>(8192, (rk + 255) / 256)), dim3(256), 0, h->stream,
                           h->cdist.as<float>(), rk, kc, h->gen_a.as<u64>());
        HIP_TRY(hipGetLastError());
        off.resize((size_t)na + 1);
        for (int64_t q = 0; q <= na; ++q) off[q] = (u32)(q * kc);
        TRY(h->gen_off.ensure(((size_t)na + 1) * 4));
        TRY(h2d_copy(h->gen_off.p, off.data(), ((size_t)na + 1) * 4, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));   // `off` is reused below
        TRY(gen_sort(h, h->gen_a.as<u64>(), h->gen_b.as<u64>(), rk, (int)na, h->gen_off.as<u32>()));
        const size_t np = (size_t)na * w;
        TRY(h->probe_list.ensure(np * 4));
        TRY(h->probe_dc.ensure(np * 4));
        TRY(h->probe_base.ensure(np * 4));
        TRY(h->gen_tot.ensure((size_t)na * 4));
        hipLaunchKernelGGL(gen_probes_kernel, dim3((unsigned)na), dim3(256), 0, h->stream, h->gen_b.as<u64>(), kc, w, h->list_len.as<u32>(),
                           h->probe_list.as<int>(), h->probe_dc.as<float>(), h->probe_base.as<u32>(), h->gen_tot.as<u32>(), d_scanned);
        HIP_TRY(hipGetLastError());
        tot.resize((size_t)na);
        TRY(d2h_copy(tot.data(), h->gen_tot.p, (size_t)na * 4, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        // stage B groups: as many queries as fit the key budget (two key buffers + sort scratch), at least one
        const int64_t cap_keys = std::max<int64_t>(1 << 20, (int64_t)(h->ws_budget / 24));
        for (int64_t g0 = 0; g0 < na;) {
            int64_t g1 = g0, keys = 0;
            while (g1 < na && (g1 == g0 || keys + tot[g1] <= cap_keys) && keys + tot[g1] < ((int64_t)1 << 32)) keys += tot[g1++];
            if (g1 == g0) return fail(IVFADC_ERR_INVALID, "a single query probes %u points: too many for one sort", tot[g0]);
            const int64_t ng = g1 - g0;
            off.resize((size_t)ng + 1);
            off[0] = 0;
            for (int64_t q = 0; q < ng; ++q) off[q + 1] = off[q] + tot[g0 + q];
            TRY(h->gen_a.ensure((size_t)std::max<int64_t>(1, keys) * 8));
            TRY(h->gen_b.ensure((size_t)std::max<int64_t>(1, keys) * 8));
            TRY(h->gen_off.ensure(((size_t)ng + 1) * 4));
            TRY(h2d_copy(h->gen_off.p, off.data(), ((size_t)ng + 1) * 4, h->stream));
            const int *pl = h->probe_list.as<int>() + (size_t)g0 * w;
            const float *pd = h->probe_dc.as<float>() + (size_t)g0 * w;
            const u32 *pb = h->probe_base.as<u32>() + (size_t)g0 * w;
            for (int64_t y0 = 0; y0 < ng; y0 += 32768) {   // grid.y limit
                const int64_t ny = std::min<int64_t>(32768, ng - y0);
                hipLaunchKernelGGL(gen_dump_kernel, dim3((unsigned)w, (unsigned)ny), dim3(256), lds, h->stream, ix,
                                   qa + (size_t)(g0 + y0) * h->d, w, pl + (size_t)y0 * w, pd + (size_t)y0 * w, pb + (size_t)y0 * w,
                                   h->gen_off.as<u32>() + y0, h->gen_a.as<u64>());
                HIP_TRY(hipGetLastError());
            }
            TRY(gen_sort(h, h->gen_a.as<u64>(), h->gen_b.as<u64>(), keys, (int)ng, h->gen_off.as<u32>()));
            hipLaunchKernelGGL(gen_emit_kernel, dim3((unsigned)ng), dim3(256), 0, h->stream, h->gen_b.as<u64>(), h->gen_off.as<u32>(),
                               h->gen_tot.as<u32>() + g0, w, K, pl, pb, h->list_pos.as<int64_t>(), idp, d_ids + (size_t)(a0 + g0) * K,
                               d_dists + (size_t)(a0 + g0) * K, d_counts + a0 + g0);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(h->stream));   // `off` and the key buffers are reused by the next group
            g0 = g1;
        }
        h->stats.queries += na;
    }
    h->stats.last_qg = -2;
    h->stats.coarse_mfma = 0;
    return IVFADC_OK;
}

// The latency path (smallq.hip.h): a handful of queries in ONE launch, (query, probe, chunk)-parallel with a last-arriver merge
// (two launches when the coarse quantizer is too large to be searched by every workgroup for itself).
// Largest coarse quantizer the small-batch launch searches inside itself.  Measured (SIFT1M shape, one query, w = 8): 32.9 us with the
// search inside -- a lane walking its own centroid rows touches 64 cache lines per load instruction -- against 24.8 us with the exact
// coarse kernel one launch earlier: the default is the separate kernel; ivfadc_set_coarse_mode(h, 5) takes the single-launch form.
int sq_inside_kc(const ivfadc_index *h)
{
    static const bool env_on = env_knob("IVFADC_SQ_INSIDE") != nullptr;   // A/B runs
    return ((h->sq_inside || env_on) && h->cent_t.p) ? SQ_COARSE_INSIDE : 0;
}

bool sq_eligible(const ivfadc_index *h, int64_t nq, int K, int w)
{
    static const bool off = env_knob("IVFADC_NO_SMALLQ") != nullptr;
    if (off || !h->allow_sq || h->force_qg != 0) return false;
    if (K > 64 || w > 64 || nq > 64 || nq * w > 512) return false;
    if ((h->d & 3) != 0) return false;
    const size_t lds = scan_lds_bytes(h, 1, 64, true) + 64 + (h->kc <= sq_inside_kc(h) ? (size_t)h->kc * 4 : 0);
    return lds <= LDS_MAX;
}

int search_small(ivfadc_index *h, int64_t nq, const float *d_q, int K, int w, uint32_t *d_ids, float *d_dists, int32_t *d_counts)
{
    TRY(ensure_common_ws(h));
    const bool inside = h->kc <= sq_inside_kc(h);
    if (!inside) {
        static const bool no_lane = env_knob("IVFADC_SQ_TILE_COARSE") != nullptr;   // A/B: the tiled small-problem kernel instead
        if (h->cent_t.p && !no_lane && nq <= 65535) {
            // exact distances, a lane per (centroid, query): half the time of the tiled kernel for a handful of queries (smallq.hip.h)
            TRY(h->cdist.ensure((size_t)nq * h->kc * 4));
            ivfadc_index::EvPair ec;
            if (h->profiling) TRY(ev_begin(h, 1, ec));
            void (*ck)(const float4 *, const float *, float *, int, int, int) =
                h->d == 128 ? coarse_lane_kernel<32> : (h->d == 96 ? coarse_lane_kernel<24> : (h->d == 64 ? coarse_lane_kernel<16> : coarse_lane_kernel<0>));
            hipLaunchKernelGGL(ck, dim3((unsigned)((h->kc + 255) / 256), (unsigned)nq), dim3(256), 0, h->stream,
                               h->cent_t.as<float4>(), d_q, h->cdist.as<float>(), (int)nq, h->kc, h->d);
            HIP_TRY(hipGetLastError());
            if (h->profiling) TRY(ev_end(h, ec));
        } else {
            TRY(run_coarse(h, d_q, nq, false));   // exact distances (the 3-op VALU kernel): nothing to refine
        }
    }
    // chunks: about two workgroups per CU in all, a chunk no shorter than 4096 points
    const int64_t items = nq * w;
    int nch = (int)std::max<int64_t>(1, std::min<int64_t>((h->maxlen + 4095) / 4096, (2 * (int64_t)h->num_cu) / std::max<int64_t>(1, items)));
    uint32_t CH = (uint32_t)align_up((size_t)std::max<int64_t>(1, (h->maxlen + nch - 1) / nch), 1024);
    nch = (int)std::max<int64_t>(1, (h->maxlen + CH - 1) / CH);
    const size_t slots = (size_t)items * nch;
    TRY(h->sq_keys.ensure(slots * K * 8));
    TRY(h->sq_cnt.ensure(slots * 4));
    if (!h->sq_arrive.p) {
        TRY(h->sq_arrive.ensure(64 * 4));
        HIP_TRY(hipMemsetAsync(h->sq_arrive.p, 0, h->sq_arrive.bytes, h->stream));
    }
    SqArgs a;
    a.ix = index_view(h);
    a.queries = d_q;
    a.nq = (int)nq; a.w = w; a.K = K; a.nch = nch; a.CH = CH;
    a.cdist = inside ? (const float *)nullptr : h->cdist.as<float>();
    a.centroids_t = h->cent_t.as<float>();
    a.part_keys = h->sq_keys.as<u64>();
    a.part_cnt = h->sq_cnt.as<u32>();
    a.arrive = h->sq_arrive.as<u32>();
    a.out_ids = d_ids; a.out_dists = d_dists; a.out_counts = d_counts;
    a.scanned_points = h->misc.as<u64>();
    const size_t lds = scan_lds_bytes(h, 1, 64, true) + 64 + (inside ? (size_t)h->kc * 4 : 0);
    sq_fn_t fn = pick_sq(h->m, h->dsub);
    int occ = 0;
    TRY(fn_occupancy(h, (const void *)fn, lds, occ));
    ivfadc_index::EvPair ep;
    if (h->profiling) TRY(ev_begin(h, 0, ep));
    hipLaunchKernelGGL(fn, dim3((unsigned)slots), dim3(256), lds, h->stream, a);
    HIP_TRY(hipGetLastError());
    if (h->profiling) TRY(ev_end(h, ep));
    h->stats.queries += nq;
    h->stats.last_qg = -3;
    h->stats.coarse_mfma = 0;
    h->stats.coarse_listed = 0;
    h->stats.last_lb = 0;
    h->stats.last_chunk = (int)CH;
    h->stats.last_scan_grid = (int)slots;
    h->stats.last_scan_lds = (int)lds;
    return IVFADC_OK;
}

// clears the hint of ivfadc_set_next_queries when a search ends, however it ends (a hint is good for ONE search)
struct HintScope {
    ivfadc_index *h;
    ~HintScope()
    {
        // (a wait that no rider launch consumed still belongs in the stream: the NEXT search on this lane reads those rows as its own)
        if (h->hint_ev && h->stream) (void)hipStreamWaitEvent(h->stream, h->hint_ev, 0);
        h->hint_ev = nullptr;
        h->hint_q = nullptr; h->hint_nq = 0; h->hint_token = 0; h->avail_q = nullptr; h->avail_nq = 0;
    }
};

// ---- certified two-level coarse search: grouping of the centroids (twolevel.hip.h) ---------------------------------------------------
int twolevel_group(ivfadc_index *h, int G, std::vector<int> &assign);   // (with the trainer, below: k-means of the centroids into h->tl_centres)

constexpr int TL_AUTO_MIN_KC = 4096;   // automatic mode: quantizers whose exhaustive coarse stage is a visible share of a step
// ... and only where the self-probe computes at most this fraction of the kc distances.  Measured on a trained kc = 65 536, d = 96
// quantizer, 10 000 queries, w = 32 (profiles/r05_two_level_coarse.json): 0.1 % -> 0.18 ms against 0.63 ms for the matrix-core filter +
// refine + top-w; with a poorer grouping, 4.5 % -> 0.92 ms.  A visited centroid costs its d floats from L2 / MALL once per QUERY (the
// exhaustive kernels read a centroid once per 128-query tile), so the break-even sits near 2.5 %.
constexpr float TL_AUTO_MAX_FRACTION = 0.02f;

int build_twolevel(ivfadc_index *h)
{
    h->tl_tried = true;
    h->tl_use = false;
    const int kc = h->kc, d = h->d;
    if ((d & 3) != 0 || kc < 128) return IVFADC_OK;
    int G = std::max(16, std::min(2048, kc / 64));
    TRY(set_device(h));
    std::vector<float> cent((size_t)kc * d);
    std::vector<int> assign;
    TRY(h->tl_centres.ensure((size_t)G * d * 4));
    // k-means over the centroids themselves (the trainer's own code: k-means++ seeding, exact assignment, deterministic sums)
    TRY(twolevel_group(h, G, assign));
    TRY(d2h_copy(cent.data(), h->centroids.p, cent.size() * 4, h->stream));
    // ---- refinement on the host.  k-means++ over the centroids leaves some groups that span two natural clusters (a cluster that drew
    // no seed is shared out among its neighbours): their radius is the distance BETWEEN clusters, their bound is useless, and -- distances
    // between cluster centres concentrate in high dimension -- every query ends up visiting every such group (measured on a trained
    // kc = 65 536 quantizer: 46 of 1024 groups visited per query, 4.5 % of the distances).  So: groups whose radius stands out against
    // the median are split by 2-means until none does; centres are the members' means, radii are taken against those centres.
    std::vector<std::vector<int>> mem((size_t)G);
    for (int c = 0; c < kc; ++c) mem[assign[c]].push_back(c);
    auto centre_of = [&](const std::vector<int> &l, std::vector<float> &out) {
        std::vector<double> acc((size_t)d, 0.0);
        for (int c : l)
            for (int i = 0; i < d; ++i) acc[i] += (double)cent[(size_t)c * d + i];
        out.resize((size_t)d);
        for (int i = 0; i < d; ++i) out[i] = (float)(acc[i] / (double)std::max<size_t>(1, l.size()));
    };
    auto dist2 = [&](int c, const float *g) {
        double r2 = 0.0;
        for (int i = 0; i < d; ++i) { const double t = (double)cent[(size_t)c * d + i] - (double)g[i]; r2 += t * t; }
        return r2;
    };
    auto radius_of = [&](const std::vector<int> &l, const std::vector<float> &g) {
        double r2 = 0.0;
        for (int c : l) r2 = std::max(r2, dist2(c, g.data()));
        return std::sqrt(r2);
    };
    const size_t gmax = (size_t)std::min(8192, std::max(G, kc / 8));
    for (int round = 0; round < 6; ++round) {
        std::vector<double> rr;
        std::vector<std::vector<float>> ctr(mem.size());
        for (size_t g = 0; g < mem.size(); ++g) {
            if (mem[g].empty()) { rr.push_back(0.0); continue; }
            centre_of(mem[g], ctr[g]);
            rr.push_back(radius_of(mem[g], ctr[g]));
        }
        std::vector<double> nz;
        for (size_t g = 0; g < mem.size(); ++g)
            if (mem[g].size() >= 2) nz.push_back(rr[g]);
        if (nz.empty()) break;
        std::nth_element(nz.begin(), nz.begin() + nz.size() / 2, nz.end());
        const double med = nz[nz.size() / 2];
        bool any = false;
        const size_t ng0 = mem.size();
        for (size_t g = 0; g < ng0 && mem.size() < gmax; ++g) {
            if (mem[g].size() < 4 || rr[g] <= 1.3 * med) continue;
            // 2-means: seeds = the member farthest from the centre and the member farthest from that one
            int a = mem[g][0];
            double best = -1.0;
            for (int c : mem[g]) { const double v = dist2(c, ctr[g].data()); if (v > best) { best = v; a = c; } }
            int b = a;
            best = -1.0;
            for (int c : mem[g]) { const double v = dist2(c, &cent[(size_t)a * d]); if (v > best) { best = v; b = c; } }
            if (a == b) continue;
            std::vector<float> ca(cent.begin() + (size_t)a * d, cent.begin() + (size_t)(a + 1) * d), cb(cent.begin() + (size_t)b * d, cent.begin() + (size_t)(b + 1) * d);
            std::vector<int> la, lb;
            for (int it = 0; it < 8; ++it) {
                la.clear(); lb.clear();
                for (int c : mem[g]) (dist2(c, ca.data()) <= dist2(c, cb.data()) ? la : lb).push_back(c);
                if (la.empty() || lb.empty()) break;
                centre_of(la, ca);
                centre_of(lb, cb);
            }
            if (la.empty() || lb.empty()) continue;
            mem[g].swap(la);
            mem.push_back(lb);
            any = true;
        }
        if (!any) break;
    }
    mem.erase(std::remove_if(mem.begin(), mem.end(), [](const std::vector<int> &l) { return l.empty(); }), mem.end());
    const int G2 = (int)mem.size();
    std::vector<float> gc((size_t)G2 * d);
    std::vector<u32> off((size_t)G2 + 1, 0);
    for (int g = 0; g < G2; ++g) {
        std::vector<float> c_;
        centre_of(mem[g], c_);
        std::copy(c_.begin(), c_.end(), gc.begin() + (size_t)g * d);
        off[g + 1] = off[g] + (u32)mem[g].size();
    }
    // slots in group order (a group's members side by side, no padding: the kernel masks the lanes past a group's end); radii in
    // double against the float centres the device will use, rounded up
    const size_t slots = off[G2];
    std::vector<u32> slot_id(std::max<size_t>(slots, 1), 0xFFFFFFFFu);
    std::vector<float> grouped(std::max<size_t>(slots, 1) * d, 0.0f), rad((size_t)G2, 0.0f);
    for (int g = 0; g < G2; ++g) {
        const u32 ns = off[g + 1] - off[g];
        float *blk = grouped.data() + (size_t)off[g] * d;
        double r2max = 0.0;
        for (u32 sl = 0; sl < ns; ++sl) {
            const int c = mem[g][sl];
            slot_id[off[g] + sl] = (u32)c;
            for (int i = 0; i < d; ++i) blk[((size_t)(i >> 2) * ns + sl) * 4 + (i & 3)] = cent[(size_t)c * d + i];
            r2max = std::max(r2max, dist2(c, &gc[(size_t)g * d]));
        }
        rad[g] = std::nextafter((float)(std::sqrt(r2max) * (1.0 + 1e-6)), INFINITY);
    }
    TRY(h->tl_centres.ensure(gc.size() * 4));
    TRY(h->tl_off.ensure(off.size() * 4));
    TRY(h->tl_rad.ensure(rad.size() * 4));
    TRY(h->tl_cent.ensure(grouped.size() * 4 + 64));
    TRY(h->tl_slot.ensure(slot_id.size() * 4));
    TRY(h2d_copy(h->tl_centres.p, gc.data(), gc.size() * 4, h->stream));
    TRY(h2d_copy(h->tl_off.p, off.data(), off.size() * 4, h->stream));
    TRY(h2d_copy(h->tl_rad.p, rad.data(), rad.size() * 4, h->stream));
    TRY(h2d_copy(h->tl_cent.p, grouped.data(), grouped.size() * 4, h->stream));
    TRY(h2d_copy(h->tl_slot.p, slot_id.data(), slot_id.size() * 4, h->stream));
    G = G2;
    h->tl_G = G;
    h->tl_eps = (float)(d + 16) * 1.1920929e-7f;   // (d + 16) 2^-23
    // Self-probe: up to 1024 centroids as queries (what a query of a trained index looks like from the quantizer's side), w = 32: the
    // fraction of the kc distances the search still computes.  Automatic mode keeps the two-level search only if the bounds leave at
    // most TL_AUTO_MAX_FRACTION of them -- N(0,1) quantizers in high dimension show ~1.0 here (distance concentration) and stay exhaustive.
    {
        const int64_t ns = std::min<int64_t>(1024, kc);
        const int wp = std::min(32, kc);
        TRY(ensure_common_ws(h));
        DevBuf pl_, pd_, pb_;
        int rc = pl_.ensure((size_t)ns * wp * 4);
        if (rc == IVFADC_OK) rc = pd_.ensure((size_t)ns * wp * 4);
        if (rc == IVFADC_OK) rc = pb_.ensure((size_t)ns * wp * 4);
        u64 before[512], after[512];
        if (rc == IVFADC_OK) {
            HIP_TRY(hipStreamSynchronize(h->stream));
            HIP_TRY(hipMemcpy(before, h->misc.p, sizeof(before), hipMemcpyDeviceToHost));
            // (every 64th... a contiguous block of centroid rows is a valid query matrix: [ns][d])
            const bool prof = h->profiling;
            h->profiling = false;
            rc = run_twolevel(h, h->centroids.as<float>(), ns, wp, pl_.as<int>(), pd_.as<float>(), pb_.as<u32>(), nullptr, h->misc.as<u64>());
            h->profiling = prof;
        }
        if (rc == IVFADC_OK) {
            HIP_TRY(hipStreamSynchronize(h->stream));
            HIP_TRY(hipMemcpy(after, h->misc.p, sizeof(after), hipMemcpyDeviceToHost));
            u64 vis = 0;
            for (int i = 0; i < 64; ++i) vis += after[i * 8 + 3] - before[i * 8 + 3];
            // the probe's counts do not belong to any search: take them out again
            HIP_TRY(h2d_hip(h->misc.p, before, sizeof(before), h->stream));
            h->tl_probe_fraction = (float)((double)vis / ((double)ns * (double)kc));
        }
        pl_.release(); pd_.release(); pb_.release();
        if (rc != IVFADC_OK) return rc;
    }
    h->tl_use = h->tl_mode > 0 || (h->tl_mode == 0 && h->tl_probe_fraction >= 0.f && h->tl_probe_fraction <= TL_AUTO_MAX_FRACTION);
    return IVFADC_OK;
}

int search_dev(ivfadc_index *h, int64_t nq, const float *d_q, int K, int w, uint32_t *d_ids, float *d_dists, int32_t *d_counts)
{
    HintScope hint_scope{h};
    // Rows the previous search's riders left serve THIS search or none, whatever path it takes (small-batch, generic, sub-batched, failing):
    // they are taken over here and forgotten.  They are offered only when the caller declared this search's queries to be the very
    // generation the rows were computed from (same token, != 0): a staging buffer refilled in between carries another token.
    const bool declared = h->cur_token != 0 && h->cur_token == h->pf_token;
    h->avail_q = declared ? h->pf_q : nullptr;
    h->avail_nq = declared ? h->pf_nq : 0;
    h->pf_q = nullptr; h->pf_nq = 0; h->pf_token = 0; h->cur_token = 0;
    h->stats.coarse_prefetched = 0;
    h->stats.last_rider = 0;
    h->partial_nq = -1;   // the probe arrays a pending ivfadc_merge_partials_device would read are about to be overwritten
    TRY(set_device(h));
    if (h->dirty) TRY(upload_lists(h));
    if (nq == 0) return IVFADC_OK;
    // the grouping of a large quantizer is built on its first search (automatic mode), or on request (ivfadc_set_coarse_mode(h, 6))
    static const bool tl_env_off = env_knob("IVFADC_NO_TWOLEVEL") != nullptr;
    if (!h->tl_tried && !h->is_view && !tl_env_off && h->tl_mode >= 0 && (h->tl_mode > 0 || h->kc >= TL_AUTO_MIN_KC)) TRY(build_twolevel(h));
    const bool parted = h->part_n > 1;
    if (!parted && sq_eligible(h, nq, K, w)) return search_small(h, nq, d_q, K, w, d_ids, d_dists, d_counts);
    if (K > IVFADC_MAX_K || w > IVFADC_MAX_W || h->force_qg == -2) {
        if (parted) return fail(IVFADC_ERR_INVALID, "list-partitioned mode reaches K <= %d and w <= %d", IVFADC_MAX_K, IVFADC_MAX_W);
        return search_generic(h, nq, d_q, K, w, d_ids, d_dists, d_counts);
    }
    Plan pl;
    TRY(fb_poll(h));
    TRY(make_plan(h, nq, K, w, pl));
    if (!pl.fits) {
        if (parted) return fail(IVFADC_ERR_INVALID, "list-partitioned mode: the selection kernels' LDS need exceeds a CU for this m and K");
        return search_generic(h, nq, d_q, K, w, d_ids, d_dists, d_counts);   // "any K and w" holds for every m
    }
    if (parted && pl.nb < nq)
        return fail(IVFADC_ERR_INVALID, "list-partitioned mode: the batch must fit one sub-batch (raise ivfadc_set_workspace_limit or split the batch)");
    for (int64_t b0 = 0; b0 < nq; b0 += pl.nb) {
        const int64_t nb = std::min(pl.nb, nq - b0);
        TRY(search_subbatch(h, pl, nb, d_q + (size_t)b0 * h->d, K, w, d_ids + (size_t)b0 * K, d_dists + (size_t)b0 * K,
                            d_counts + b0, pl.nb >= nq));
    }
    return IVFADC_OK;
}

int encode_dev(ivfadc_index *h, int64_t n, const float *pts, int32_t *out_list, uint8_t *out_codes)
{
    // batches keep the n x kc distance matrix bounded
    const int64_t bmax = std::max<int64_t>(256, (int64_t)(((size_t)1 << 30) / ((size_t)h->kc * 4)));
    const size_t lds = align_up((size_t)h->d, 4) * 4 + (size_t)h->m * 8;
    for (int64_t b0 = 0; b0 < n; b0 += bmax) {
        const int64_t nb = std::min(bmax, n - b0);
        TRY(h->pts_stage.ensure((size_t)nb * h->d * 4));
        TRY(h->assign.ensure((size_t)nb * 4));
        TRY(h->enc_codes.ensure((size_t)nb * h->m));
        TRY(h2d_copy(h->pts_stage.p, pts + (size_t)b0 * h->d, (size_t)nb * h->d * 4, h->stream));
        TRY(run_coarse(h, h->pts_stage.as<float>(), nb, false));   // push! path: exact distances
        hipLaunchKernelGGL(argmin_rows_kernel, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, h->stream, h->cdist.as<float>(), (int)nb,
                           h->kc, h->assign.as<int>());
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(encode_kernel, dim3((unsigned)nb), dim3(256), lds, h->stream, h->pts_stage.as<float>(), h->assign.as<int>(),
                           h->d, h->m, h->ksub, h->dsub, h->centroids.as<float>(), h->codebooks.as<float>(), h->labels.as<uint8_t>(),
                           h->enc_codes.as<uint8_t>());
        HIP_TRY(hipGetLastError());
        TRY(d2h_copy(out_list + b0, h->assign.p, (size_t)nb * 4, h->stream));
        TRY(d2h_copy(out_codes + (size_t)b0 * h->m, h->enc_codes.p, (size_t)nb * h->m, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return IVFADC_OK;
}

}  // namespace

// ---- training (train.hip.h) ---------------------------------------------------------------------
namespace {

struct TrainCtx {
    hipStream_t stream = nullptr;
    DevBuf cdist, assign, mind, partial, acc, counts, flag, blockmax;
};

// k-means of the n x dcols window (leading dimension ld) of d_x into d_centres [k][dcols]
int kmeans_dev(TrainCtx &t, const float *d_x, int64_t n, int dcols, int ld, int k, int maxiter, uint64_t seed, float *d_centres)
{
    // ---- k-means++ over a strided subsample
    const int S = (int)std::min<int64_t>(n, std::max<int64_t>(32768, (int64_t)32 * k));
    const int nblk = (S + 255) / 256;
    TRY(t.mind.ensure((size_t)S * 4));
    TRY(t.partial.ensure((size_t)nblk * 8));
    for (int j = 0; j < k; ++j) {
        if (j > 0) {
            hipLaunchKernelGGL(tr_kmpp_update_kernel, dim3(nblk), dim3(256), 0, t.stream, d_x, n, dcols, ld, S,
                               d_centres + (size_t)(j - 1) * dcols, t.mind.as<float>(), t.partial.as<double>(), j == 1 ? 1 : 0);
            HIP_TRY(hipGetLastError());
        }
        hipLaunchKernelGGL(tr_kmpp_pick_kernel, dim3(1), dim3(64), 0, t.stream, d_x, n, dcols, ld, S, t.mind.as<float>(),
                           t.partial.as<double>(), nblk, (u64)seed, j, d_centres);
        HIP_TRY(hipGetLastError());
    }
    // ---- fixed-point scale: |sum| <= n * maxabs must stay below 2^62
    const int mb = 1024;
    TRY(t.blockmax.ensure((size_t)mb * 4));
    hipLaunchKernelGGL(tr_maxabs_kernel, dim3(mb), dim3(256), 0, t.stream, d_x, n, dcols, ld, t.blockmax.as<float>());
    HIP_TRY(hipGetLastError());
    std::vector<float> bm(mb);
    HIP_TRY(hipMemcpyAsync(bm.data(), t.blockmax.p, (size_t)mb * 4, hipMemcpyDeviceToHost, t.stream));
    HIP_TRY(hipStreamSynchronize(t.stream));
    double maxabs = 1e-30;
    for (float v : bm) maxabs = std::max(maxabs, (double)v);
    const int ex = 61 - (int)std::ceil(std::log2((double)n * maxabs + 1.0));
    const double scale = std::ldexp(1.0, std::max(-60, std::min(ex, 60)));

    // ---- Lloyd
    const int64_t chunk = std::max<int64_t>(256, (int64_t)(((size_t)1 << 30) / ((size_t)k * 4)));
    TRY(t.cdist.ensure((size_t)std::min<int64_t>(n, chunk) * k * 4));
    TRY(t.assign.ensure((size_t)n * 4));
    TRY(t.acc.ensure((size_t)k * dcols * 8));
    TRY(t.counts.ensure((size_t)k * 4));
    TRY(t.flag.ensure(4));
    for (int it = 0; it < maxiter; ++it) {
        for (int64_t p0 = 0; p0 < n; p0 += chunk) {
            const int64_t nb = std::min(chunk, n - p0);
            dim3 grid((k + CO_T - 1) / CO_T, (unsigned)((nb + 63) / 64));
            hipLaunchKernelGGL(coarse_dist_kernel<64>, grid, dim3(256), 0, t.stream, d_x + (size_t)p0 * ld, d_centres,
                               t.cdist.as<float>(), (int)nb, k, dcols, ld);
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL(argmin_rows_kernel, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, t.stream, t.cdist.as<float>(),
                               (int)nb, k, t.assign.as<int>() + p0);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipMemsetAsync(t.acc.p, 0, (size_t)k * dcols * 8, t.stream));
        HIP_TRY(hipMemsetAsync(t.counts.p, 0, (size_t)k * 4, t.stream));
        HIP_TRY(hipMemsetAsync(t.flag.p, 0, 4, t.stream));
        hipLaunchKernelGGL(tr_accumulate_kernel, dim3(2048), dim3(256), 0, t.stream, d_x, n, dcols, ld, t.assign.as<int>(), scale,
                           t.acc.as<long long>(), t.counts.as<u32>());
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(tr_finalize_kernel, dim3((unsigned)(((size_t)k * dcols + 255) / 256)), dim3(256), 0, t.stream, d_x, n,
                           dcols, ld, k, t.acc.as<long long>(), t.counts.as<u32>(), 1.0 / scale, (u64)seed, it, d_centres,
                           t.flag.as<int>());
        HIP_TRY(hipGetLastError());
        int changed = 0;
        HIP_TRY(hipMemcpyAsync(&changed, t.flag.p, 4, hipMemcpyDeviceToHost, t.stream));
        HIP_TRY(hipStreamSynchronize(t.stream));
        if (!changed) break;   // Lloyd's fixed point
    }
    return IVFADC_OK;
}

// the centroids of an index grouped for the two-level coarse search: G centres into h->tl_centres, every centroid's group into `assign`
int twolevel_group(ivfadc_index *h, int G, std::vector<int> &assign)
{
    const int kc = h->kc, d = h->d;
    TrainCtx t;
    HIP_TRY(hipStreamCreateWithFlags(&t.stream, hipStreamNonBlocking));
    auto body = [&]() -> int {
        HIP_TRY(hipStreamSynchronize(h->stream));
        TRY(kmeans_dev(t, h->centroids.as<float>(), kc, d, d, G, 12, 0x7107ull, h->tl_centres.as<float>()));
        // final assignment against the final centres
        TRY(t.cdist.ensure((size_t)kc * G * 4));
        TRY(t.assign.ensure((size_t)kc * 4));
        dim3 grid((G + CO_T - 1) / CO_T, (unsigned)((kc + 63) / 64));
        hipLaunchKernelGGL(coarse_dist_kernel<64>, grid, dim3(256), 0, t.stream, h->centroids.as<float>(), h->tl_centres.as<float>(),
                           t.cdist.as<float>(), kc, G, d, d);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(argmin_rows_kernel, dim3((unsigned)((kc + 3) / 4)), dim3(256), 0, t.stream, t.cdist.as<float>(), kc, G, t.assign.as<int>());
        HIP_TRY(hipGetLastError());
        assign.resize((size_t)kc);
        HIP_TRY(hipMemcpyAsync(assign.data(), t.assign.p, (size_t)kc * 4, hipMemcpyDeviceToHost, t.stream));
        HIP_TRY(hipStreamSynchronize(t.stream));
        for (int c = 0; c < kc; ++c)
            if (assign[c] < 0 || assign[c] >= G) return fail(IVFADC_ERR_STATE, "two-level grouping: centroid %d assigned to group %d", c, assign[c]);
        return IVFADC_OK;
    };
    const int rc = body();
    (void)hipStreamSynchronize(t.stream);
    DevBuf *bufs[] = {&t.cdist, &t.assign, &t.mind, &t.partial, &t.acc, &t.counts, &t.flag, &t.blockmax};
    for (DevBuf *b : bufs) b->release();
    (void)hipStreamDestroy(t.stream);
    return rc;
}

int train_impl(int device, int d, int64_t n, const float *data, int kc, int k, int m, int coarse_maxiter, int quant_maxiter,
               uint64_t seed, float *out_centroids, float *out_codebooks)
{
    if (d < 1 || n < 1 || !data || !out_centroids || !out_codebooks) return fail(IVFADC_ERR_INVALID, "bad argument");
    if (kc < 2) return fail(IVFADC_ERR_ASSERT, "Number of coarse clusters has to be >= 2");                   // index.jl:118
    if (k > n) return fail(IVFADC_ERR_ASSERT, "Number of quantization levels  has to be <= %lld", (long long)n);   // :119
    if (m < 1 || m > d) return fail(IVFADC_ERR_ASSERT, "Number of codebooks has to be between 1 and %d", d);       // :120
    if (coarse_maxiter < 1 || quant_maxiter < 1) return fail(IVFADC_ERR_ASSERT, "Number of clustering iterations has to be > 0");
    if (d % m != 0) return fail(IVFADC_ERR_INVALID, "d %% m != 0 is not supported");
    if (k < 1 || k > 256) return fail(IVFADC_ERR_INVALID, "k must be in 1..256");
    if (kc > n) return fail(IVFADC_ERR_INVALID, "kc > n");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(IVFADC_ERR_HIP, "no such HIP device %d", device);
    HIP_TRY(hipSetDevice(device));
    TrainCtx t;
    HIP_TRY(hipStreamCreateWithFlags(&t.stream, hipStreamNonBlocking));
    DevBuf x, resid, cent, cbs;
    const int dsub = d / m;
    int rc = x.ensure((size_t)n * d * 4);
    if (rc == IVFADC_OK) rc = resid.ensure((size_t)n * d * 4);
    if (rc == IVFADC_OK) rc = cent.ensure((size_t)kc * d * 4);
    if (rc == IVFADC_OK) rc = cbs.ensure((size_t)k * dsub * 4);
    auto body = [&]() -> int {
        TRY(h2d_copy(x.p, data, (size_t)n * d * 4, t.stream));
        TRY(kmeans_dev(t, x.as<float>(), n, d, d, kc, coarse_maxiter, seed, cent.as<float>()));
        // final assignment -> residuals (index.jl:138,168-175): t.assign holds the last Lloyd assignment only if the
        // loop stopped at a fixed point, so assign once more against the final centres
        const int64_t chunk = std::max<int64_t>(256, (int64_t)(((size_t)1 << 30) / ((size_t)kc * 4)));
        for (int64_t p0 = 0; p0 < n; p0 += chunk) {
            const int64_t nb = std::min(chunk, n - p0);
            dim3 grid((kc + CO_T - 1) / CO_T, (unsigned)((nb + 63) / 64));
            hipLaunchKernelGGL(coarse_dist_kernel<64>, grid, dim3(256), 0, t.stream, x.as<float>() + (size_t)p0 * d, cent.as<float>(),
                               t.cdist.as<float>(), (int)nb, kc, d, d);
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL(argmin_rows_kernel, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, t.stream, t.cdist.as<float>(),
                               (int)nb, kc, t.assign.as<int>() + p0);
            HIP_TRY(hipGetLastError());
        }
        hipLaunchKernelGGL(tr_residual_kernel, dim3(2048), dim3(256), 0, t.stream, x.as<float>(), n, d, t.assign.as<int>(),
                           cent.as<float>(), resid.as<float>());
        HIP_TRY(hipGetLastError());
        TRY(d2h_copy(out_centroids, cent.p, (size_t)kc * d * 4, t.stream));
        for (int i = 0; i < m; ++i) {
            TRY(kmeans_dev(t, resid.as<float>() + (size_t)i * dsub, n, dsub, d, k, quant_maxiter, seed + 1 + (uint64_t)i,
                           cbs.as<float>()));
            TRY(d2h_copy(out_codebooks + (size_t)i * k * dsub, cbs.p, (size_t)k * dsub * 4, t.stream));
            HIP_TRY(hipStreamSynchronize(t.stream));
        }
        HIP_TRY(hipStreamSynchronize(t.stream));
        return IVFADC_OK;
    };
    if (rc == IVFADC_OK) rc = body();
    (void)hipStreamSynchronize(t.stream);
    DevBuf *bufs[] = {&x, &resid, &cent, &cbs, &t.cdist, &t.assign, &t.mind, &t.partial, &t.acc, &t.counts, &t.flag, &t.blockmax};
    for (DevBuf *b : bufs) b->release();
    (void)hipStreamDestroy(t.stream);
    return rc;
}

}  // namespace

// =============================================================================================
extern "C" {

const char *ivfadc_last_error(void) { return g_err.c_str(); }

int ivfadc_create(ivfadc_t **out, int device, int d, int kc, int m, int ksub, const float *centroids, const float *codebooks,
                  const uint8_t *code_labels)
try {
    if (!out) return fail(IVFADC_ERR_INVALID, "out is null");
    *out = nullptr;
    if (d < 1 || kc < 1 || m < 1 || ksub < 1) return fail(IVFADC_ERR_INVALID, "d, kc, m, ksub must be >= 1");
    if (m > d) return fail(IVFADC_ERR_ASSERT, "Number of codebooks has to be between 1 and %d", d);
    if (d % m != 0) return fail(IVFADC_ERR_INVALID, "d %% m != 0 is not supported (rowrange for ragged sub-spaces is unverifiable)");
    if (ksub > 256) return fail(IVFADC_ERR_INVALID, "ksub > 256 does not fit UInt8 codes");
    if (!centroids || !codebooks || !code_labels) return fail(IVFADC_ERR_INVALID, "null array");
    // Quantizers must be finite: every bound the filters certify (coarse score filter, lower-bound tables) is computed from their norms.
    // (Queries are not scanned -- that would cost the hot path a pass over every batch: a query with a NaN or infinite component gets
    // unspecified neighbours, valid ids and counts, and leaves the other queries of its batch untouched; tests/test_gpu_parity.py.)
    for (size_t i = 0; i < (size_t)d * kc; ++i)
        if (!std::isfinite(centroids[i])) return fail(IVFADC_ERR_INVALID, "centroid %zu has a non-finite component", i / (size_t)d);
    for (size_t i = 0; i < (size_t)d * ksub; ++i)
        if (!std::isfinite(codebooks[i])) return fail(IVFADC_ERR_INVALID, "codebook %zu has a non-finite component", i / ((size_t)(d / m) * ksub));
    std::vector<uint8_t> ok((size_t)m * 256, 0);
    for (int i = 0; i < m; ++i)
        for (int c = 0; c < ksub; ++c) {
            uint8_t lab = code_labels[(size_t)i * ksub + c];
            if (ok[(size_t)i * 256 + lab]) return fail(IVFADC_ERR_INVALID, "duplicate label %d in codebook %d", (int)lab, i);
            ok[(size_t)i * 256 + lab] = 1;
        }
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(IVFADC_ERR_HIP, "no HIP device");
    if (device < 0 || device >= ndev) return fail(IVFADC_ERR_INVALID, "device %d out of range [0,%d)", device, ndev);
    ivfadc_index *h = new ivfadc_index();
    h->device = device;
    h->d = d; h->kc = kc; h->m = m; h->ksub = ksub; h->dsub = d / m; h->cs = code_stride(m);
    h->h_label_ok.swap(ok);
    h->identity_labels = true;
    for (int i = 0; i < m && h->identity_labels; ++i)
        for (int c = 0; c < ksub; ++c)
            if (code_labels[(size_t)i * ksub + c] != (uint8_t)c) { h->identity_labels = false; break; }
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) { delete h; return fail(IVFADC_ERR_HIP, "device init failed: %s", hipGetErrorString(e)); }
    h->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    int rc = h->centroids.ensure((size_t)d * kc * 4);
    if (rc == IVFADC_OK) rc = h->codebooks.ensure((size_t)d * ksub * 4);
    if (rc == IVFADC_OK) rc = h->codebooks_t.ensure((size_t)m * (((d / m) + 3) & ~3) * ksub * 4);
    if (rc == IVFADC_OK) rc = h->labels.ensure((size_t)m * ksub);
    if (rc == IVFADC_OK) {
        e = h2d_hip(h->centroids.p, centroids, (size_t)d * kc * 4, h->stream);
        if (e == hipSuccess) e = h2d_hip(h->codebooks.p, codebooks, (size_t)d * ksub * 4, h->stream);
        if (e == hipSuccess) {
            // regrouped copy for the table build (IndexView::codebooks_t)
            const int dsub = d / m, dp = (dsub + 3) & ~3;   // [m][dp / 4][ksub][4], zero-padded
            std::vector<float> t((size_t)m * dp * ksub, 0.0f);
            if (dsub == 6 && (m & 1) == 0) {
                // pair-packed (build_tables_t, DSUB == 6): [m / 2][3][ksub][4], element e = 0..11 of pair p, codeword c =
                // dimension e of sub-quantizer 2p (e < 6) or dimension e - 6 of sub-quantizer 2p + 1
                for (int p = 0; p < m / 2; ++p)
                    for (int c = 0; c < ksub; ++c)
                        for (int e2 = 0; e2 < 12; ++e2) {
                            const int ii = 2 * p + e2 / 6, x = e2 % 6;
                            t[(size_t)p * 12 * ksub + ((size_t)(e2 / 4) * ksub + c) * 4 + (e2 % 4)] = codebooks[((size_t)ii * ksub + c) * dsub + x];
                        }
            } else
            for (int ii = 0; ii < m; ++ii)
                for (int c = 0; c < ksub; ++c)
                    for (int x = 0; x < dsub; ++x)
                        t[(size_t)ii * dp * ksub + ((size_t)(x / 4) * ksub + c) * 4 + (x % 4)] =
                            codebooks[((size_t)ii * ksub + c) * dsub + x];
            e = h2d_hip(h->codebooks_t.p, t.data(), t.size() * 4, h->stream);
            if (e == hipSuccess && m == 48 && dsub == 16) {
                // pair-interleaved copy for the packed-FP32 table build of the m = 48 query-major kernel (IndexView::codebooks_p)
                std::vector<float> pp((size_t)m * dsub * ksub);
                for (int p = 0; p < m / 2; ++p)
                    for (int c = 0; c < ksub; ++c)
                        for (int x = 0; x < dsub; ++x)
                            for (int hh = 0; hh < 2; ++hh)
                                pp[(size_t)p * dsub * 2 * ksub + ((size_t)(x / 2) * ksub + c) * 4 + (x % 2) * 2 + hh] =
                                    codebooks[((size_t)(2 * p + hh) * ksub + c) * dsub + x];
                rc = h->codebooks_p.ensure(pp.size() * 4);
                if (rc == IVFADC_OK) e = h2d_hip(h->codebooks_p.p, pp.data(), pp.size() * 4, h->stream);
            }
        }
        if (e == hipSuccess) e = h2d_hip(h->labels.p, code_labels, (size_t)m * ksub, h->stream);
        if (e != hipSuccess) rc = fail(IVFADC_ERR_HIP, "upload failed: %s", hipGetErrorString(e));
        if (rc == IVFADC_OK && lb_shape(m, d / m) && ksub == 256) {
            // operands of the lower-bound table build (lbscan.hip.h), everything in LABEL order (table slot = code byte):
            // split[ii][g][p][lane] = 8 bf16 of label 64 g + lane -- parts p < NP / 2: hi pieces of dimensions 8 p .. 8 p + 7,
            // the others the lo pieces (x = hi + lo + O(2^-18 |x|)); n2 = ||codeword||^2 in double, rounded once
            const int dsub = d / m, dsp = (dsub + 7) & ~7, np = dsp / 4;
            auto to_bf16 = [](float x) {
                uint32_t b;
                memcpy(&b, &x, 4);
                return (uint16_t)((b + 0x7FFFu + ((b >> 16) & 1u)) >> 16);
            };
            std::vector<uint16_t> sp((size_t)m * 4 * np * 64 * 8, 0);
            std::vector<float> n2((size_t)m * 256, 0.0f), lab((size_t)m * 256 * dsub, 0.0f), mx((size_t)m, 0.0f);
            for (int ii = 0; ii < m; ++ii) {
                double mxn = 0.0;
                for (int c = 0; c < ksub; ++c) {
                    const int L = code_labels[(size_t)ii * ksub + c], g = L >> 6, ln = L & 63;
                    const float *cw = codebooks + ((size_t)ii * ksub + c) * dsub;
                    double acc = 0.0;
                    for (int t = 0; t < dsub; ++t) {
                        const float v = cw[t];
                        acc += (double)v * v;
                        lab[((size_t)ii * 256 + L) * dsub + t] = v;
                        const uint16_t hb = to_bf16(v);
                        const uint32_t hb32 = (uint32_t)hb << 16;
                        float hf;
                        memcpy(&hf, &hb32, 4);
                        const uint16_t lb16 = to_bf16(v - hf);
                        const size_t base = (((size_t)(ii * 4 + g) * np) * 64) * 8;
                        sp[base + ((size_t)(t >> 3) * 64 + ln) * 8 + (t & 7)] = hb;
                        sp[base + ((size_t)(np / 2 + (t >> 3)) * 64 + ln) * 8 + (t & 7)] = lb16;
                    }
                    n2[(size_t)ii * 256 + L] = (float)acc;
                    mxn = std::max(mxn, acc);
                }
                mx[ii] = (float)(std::sqrt(mxn) * (1.0 + 1e-6));
            }
            {
                // f16 operand of the one-product build: [ii][g][p][lane] = 8 f16 of label 64 g + lane, dimensions 8 p .. 8 p + 7, scaled by
                // 2^e_ii with max |cb 2^e_ii| in [2^10, 2^11]
                const int nph = dsp / 8;
                std::vector<uint16_t> hf((size_t)m * 4 * nph * 64 * 8, 0);
                std::vector<float> isc((size_t)m, 1.0f);
                for (int ii = 0; ii < m; ++ii) {
                    double mxa = 0.0;
                    for (int c = 0; c < ksub; ++c)
                        for (int t = 0; t < dsub; ++t) mxa = std::max(mxa, (double)std::fabs(codebooks[((size_t)ii * ksub + c) * dsub + t]));
                    int ex = 0;
                    if (mxa > 0.0) ex = std::max(-100, std::min(100, (int)std::floor(std::log2(2048.0 / mxa))));
                    const float sc = std::ldexp(1.0f, ex);
                    isc[ii] = std::ldexp(1.0f, -ex);
                    for (int c = 0; c < ksub; ++c) {
                        const int L = code_labels[(size_t)ii * ksub + c], g = L >> 6, ln = L & 63;
                        for (int t = 0; t < dsub; ++t) {
                            const _Float16 v = (_Float16)(codebooks[((size_t)ii * ksub + c) * dsub + t] * sc);
                            memcpy(&hf[((((size_t)(ii * 4 + g) * nph) + (t >> 3)) * 64 + ln) * 8 + (t & 7)], &v, 2);
                        }
                    }
                }
                rc = h->lb_f16.ensure(hf.size() * 2);
                if (rc == IVFADC_OK) rc = h->lb_isc.ensure(isc.size() * 4);
                if (rc == IVFADC_OK) {
                    e = h2d_hip(h->lb_f16.p, hf.data(), hf.size() * 2, h->stream);
                    if (e == hipSuccess) e = h2d_hip(h->lb_isc.p, isc.data(), isc.size() * 4, h->stream);
                    if (e != hipSuccess) rc = fail(IVFADC_ERR_HIP, "upload failed: %s", hipGetErrorString(e));
                }
                if (env_knob("IVFADC_LB_BF16") != nullptr) h->lb_use_f16 = false;
            }
            if (rc == IVFADC_OK) rc = h->lb_split.ensure(sp.size() * 2);
            if (rc == IVFADC_OK) rc = h->lb_n2.ensure(n2.size() * 4);
            if (rc == IVFADC_OK) rc = h->lb_lab.ensure(lab.size() * 4);
            if (rc == IVFADC_OK) rc = h->lb_maxn.ensure(mx.size() * 4);
            if (rc == IVFADC_OK) {
                e = h2d_hip(h->lb_split.p, sp.data(), sp.size() * 2, h->stream);
                if (e == hipSuccess) e = h2d_hip(h->lb_n2.p, n2.data(), n2.size() * 4, h->stream);
                if (e == hipSuccess) e = h2d_hip(h->lb_lab.p, lab.data(), lab.size() * 4, h->stream);
                if (e == hipSuccess) e = h2d_hip(h->lb_maxn.p, mx.data(), mx.size() * 4, h->stream);
                if (e != hipSuccess) rc = fail(IVFADC_ERR_HIP, "upload failed: %s", hipGetErrorString(e));
            }
        }
    }
    if (rc == IVFADC_OK && nf_shape(m, d / m) && ksub == 256) {
        // operands of the narrow-field list-major kernel (nfscan.hip.h): ||codeword||^2 by CODEWORD index (double, rounded once) for the
        // filter tables, and the f32 codewords in LABEL order (table slot = code byte) for the exact sums of what the filter lets through
        const int dsub = d / m;
        std::vector<float> n2((size_t)m * 256 + m, 0.0f), lab((size_t)m * 256 * dsub, 0.0f);   // n2[m][256], then max ||codeword||^2 per block
        for (int ii = 0; ii < m; ++ii)
            for (int c = 0; c < ksub; ++c) {
                const int L = code_labels[(size_t)ii * ksub + c];
                const float *cw = codebooks + ((size_t)ii * ksub + c) * dsub;
                double acc = 0.0;
                for (int t = 0; t < dsub; ++t) {
                    acc += (double)cw[t] * cw[t];
                    lab[((size_t)ii * 256 + L) * dsub + t] = cw[t];
                }
                n2[(size_t)ii * 256 + c] = (float)acc;
                n2[(size_t)m * 256 + ii] = std::max(n2[(size_t)m * 256 + ii], (float)(acc * (1.0 + 1e-6)));
            }
        rc = h->nf_n2.ensure(n2.size() * 4);
        if (rc == IVFADC_OK) rc = h->nf_lab.ensure(lab.size() * 4);
        if (rc == IVFADC_OK) {
            hipError_t e2 = h2d_hip(h->nf_n2.p, n2.data(), n2.size() * 4, h->stream);
            if (e2 == hipSuccess) e2 = h2d_hip(h->nf_lab.p, lab.data(), lab.size() * 4, h->stream);
            if (e2 != hipSuccess) rc = fail(IVFADC_ERR_HIP, "upload failed: %s", hipGetErrorString(e2));
        }
    }
    if (rc == IVFADC_OK) {
        // ||c||^2 in double, rounded once: error <= u ||c||^2 (see refine_probes)
        std::vector<float> cn((size_t)kc);
        double mx = 0.0;
        for (int c = 0; c < kc; ++c) {
            double acc = 0.0;
            for (int i = 0; i < d; ++i) acc += (double)centroids[(size_t)c * d + i] * centroids[(size_t)c * d + i];
            cn[c] = (float)acc;
            mx = std::max(mx, acc);
        }
        h->cmaxn = (float)(std::sqrt(mx) * (1.0 + 1e-6));
        if ((d & 3) == 0 && (size_t)d * kc <= ((size_t)16 << 20)) {
            // regrouped copy for the latency path's coarse search (smallq.hip.h: coarse_lane_kernel, and the search inside sq_kernel): [d / 4][kc][4]
            std::vector<float> ct((size_t)d * kc);
            for (int c = 0; c < kc; ++c)
                for (int i = 0; i < d; ++i) ct[((size_t)(i >> 2) * kc + c) * 4 + (i & 3)] = centroids[(size_t)c * d + i];
            rc = h->cent_t.ensure(ct.size() * 4);
            if (rc == IVFADC_OK) {
                e = h2d_hip(h->cent_t.p, ct.data(), ct.size() * 4, h->stream);
                if (e != hipSuccess) rc = fail(IVFADC_ERR_HIP, "upload failed: %s", hipGetErrorString(e));
            }
        }
        // bf16 split of the centroids for coarse_bf16_kernel: x = hi + lo + O(2^-18 |x|), rows zero-padded to 32 dimensions
        if ((d & 3) == 0 && kc >= 2048) {
            const int dp = (d + 31) & ~31;
            h->dp32 = dp;
            auto to_bf16 = [](float x) {
                uint32_t b;
                memcpy(&b, &x, 4);
                return (uint16_t)((b + 0x7FFFu + ((b >> 16) & 1u)) >> 16);
            };
            std::vector<uint16_t> hi((size_t)kc * dp, 0), lo((size_t)kc * dp, 0);
            for (int c = 0; c < kc; ++c)
                for (int i = 0; i < d; ++i) {
                    const float v = centroids[(size_t)c * d + i];
                    const uint16_t hb = to_bf16(v);
                    const uint32_t hb32 = (uint32_t)hb << 16;
                    float hf;
                    memcpy(&hf, &hb32, 4);
                    hi[(size_t)c * dp + i] = hb;
                    lo[(size_t)c * dp + i] = to_bf16(v - hf);
                }
            {
                // the f16 form: one power-of-two scale for the whole quantizer (and for the queries, which live in the same space)
                double maxabs = 0.0;
                for (size_t i = 0; i < (size_t)kc * d; ++i) maxabs = std::max(maxabs, (double)std::fabs(centroids[i]));
                if (maxabs > 0.0 && std::isfinite(maxabs)) {
                    const int ex = (int)std::floor(std::log2(4096.0 / maxabs));
                    if (ex > -100 && ex < 100) {
                        const float sc = std::ldexp(1.0f, ex);
                        std::vector<uint16_t> hf((size_t)kc * dp, 0);
                        for (int c = 0; c < kc; ++c)
                            for (int i = 0; i < d; ++i) {
                                const _Float16 v = (_Float16)(centroids[(size_t)c * d + i] * sc);
                                memcpy(&hf[(size_t)c * dp + i], &v, 2);
                            }
                        rc = h->cent_f16.ensure(hf.size() * 2);
                        if (rc == IVFADC_OK) {
                            e = h2d_hip(h->cent_f16.p, hf.data(), hf.size() * 2, h->stream);
                            if (e != hipSuccess) rc = fail(IVFADC_ERR_HIP, "upload failed: %s", hipGetErrorString(e));
                            else h->f16_scale = sc;
                        }
                    }
                }
            }
            if (rc == IVFADC_OK) rc = h->cent_hi.ensure(hi.size() * 2);
            if (rc == IVFADC_OK) rc = h->cent_lo.ensure(lo.size() * 2);
            if (rc == IVFADC_OK) {
                e = h2d_hip(h->cent_hi.p, hi.data(), hi.size() * 2, h->stream);
                if (e == hipSuccess) e = h2d_hip(h->cent_lo.p, lo.data(), lo.size() * 2, h->stream);
                if (e != hipSuccess) rc = fail(IVFADC_ERR_HIP, "upload failed: %s", hipGetErrorString(e));
            }
        } else {
            h->allow_bf16 = false;
        }
        if (env_knob("IVFADC_COARSE_F32") != nullptr) h->allow_bf16 = false;
        if (env_knob("IVFADC_COARSE_BF16") != nullptr) h->allow_f16 = false;   // A/B: the three-product bf16 split instead of the f16 form
        if (env_knob("IVFADC_NO_LISTED") != nullptr) h->allow_listed = false;
        if (env_knob("IVFADC_NO_PRUNE") != nullptr) h->allow_prune = false;
        if (rc == IVFADC_OK) rc = h->cnorm.ensure((size_t)kc * 4);
        if (rc == IVFADC_OK) {
            e = h2d_hip(h->cnorm.p, cn.data(), (size_t)kc * 4, h->stream);
            if (e != hipSuccess) rc = fail(IVFADC_ERR_HIP, "upload failed: %s", hipGetErrorString(e));
        }
        h->allow_filt = env_knob("IVFADC_EXACT_TABLES") == nullptr;
        h->allow_mfma = env_knob("IVFADC_COARSE_EXACT") == nullptr;
        if (const char *e = env_knob("IVFADC_MFMA_MIN_KC")) h->mfma_min_kc = std::max(128, atoi(e));   // tuning knob
    }
    if (rc != IVFADC_OK) { ivfadc_destroy(h); return rc; }
    // an index starts with kc empty lists
    h->h_len.assign((size_t)kc, 0);
    h->hl_codes.assign((size_t)kc, {});
    h->hl_ids.assign((size_t)kc, {});
    h->dirty = true;
    *out = h;
    return IVFADC_OK;
} IVF_CATCH

void ivfadc_destroy(ivfadc_t *h)
{
    if (!h) return;
    // (destroying a handle that another thread is still calling into is the caller's bug; what IS guarded is the topology: a view leaving
    // its index's list while a mutator of that index walks it, an index telling its views that it is gone)
    std::unique_lock<std::recursive_mutex> topo(g_topo_mu);
    h->mu.m.lock();
    h->mu.m.unlock();   // a call still running on this handle (a mutator holding a view's mutex) has finished
    (void)hipSetDevice(h->device);
    if (h->pipe_view) { ivfadc_destroy(h->pipe_view); h->pipe_view = nullptr; }
    if (h->fb_ev) { (void)hipEventSynchronize(h->fb_ev); (void)hipEventDestroy(h->fb_ev); }
    h->fb_pin.release();
    if (h->pipe_ev_in) (void)hipEventDestroy(h->pipe_ev_in);
    if (h->pipe_ev_out) (void)hipEventDestroy(h->pipe_ev_out);
    if (h->copy_stream) { (void)hipStreamSynchronize(h->copy_stream); (void)hipStreamDestroy(h->copy_stream); }
    for (hipEvent_t e : h->ingest_ev) (void)hipEventDestroy(e);
    // views that outlive the index keep dangling aliases: they are told, and refuse to search
    // (under each view's own mutex -- topology first, then the view, the order of every mutator: a view call reads its link and the
    // generation with nothing but its own mutex held)
    for (ivfadc_index *v : h->views) {
        std::lock_guard<std::recursive_mutex> vl(v->mu.m);
        v->orphan = true;
        v->view_of = nullptr;
    }
    if (h->is_view && h->view_of) {
        auto &vs = h->view_of->views;
        vs.erase(std::remove(vs.begin(), vs.end(), h), vs.end());
        h->view_of->n_views.store((int)vs.size());
    }
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->comm || h->comm_stream) (void)ivfadc_comm_destroy(h);
    for (auto &ep : h->pending) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
    for (auto &ep : h->free_ev) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
    DevBuf *bufs[] = {&h->lb_f16, &h->lb_isc, &h->cent_f16, &h->q_f16, &h->q_flags, &h->tl_centres, &h->tl_off, &h->tl_rad, &h->tl_cent, &h->tl_slot, &h->tl_gdist, &h->cdist2, &h->cent_t, &h->sq_keys, &h->sq_cnt, &h->sq_arrive, &h->lb_split, &h->lb_n2, &h->lb_lab, &h->lb_maxn, &h->nf_n2, &h->nf_lab, &h->wg8_tabs, &h->wg8_items, &h->centroids, &h->codebooks, &h->codebooks_t, &h->codebooks_p, &h->labels, &h->cnorm, &h->tmin, &h->tlist, &h->cent_hi, &h->cent_lo, &h->q_hi, &h->q_lo, &h->gen_a, &h->gen_b, &h->gen_tmp, &h->gen_off, &h->gen_tot, &h->list_pos, &h->list_len, &h->list_codeoff, &h->codes, &h->ids, &h->app_stage, &h->q_stage,
                      &h->cdist, &h->probe_list, &h->probe_dc, &h->probe_base, &h->list_cnt, &h->bucket_off, &h->wi_off, &h->cursor,
                      &h->bucket_items, &h->misc, &h->qthr, &h->part_keys, &h->part_cnt, &h->out_ids, &h->out_dists, &h->out_counts,
                      &h->assign, &h->enc_codes, &h->pts_stage, &h->dbg};
    for (DevBuf *b : bufs) b->release();
    h->pin_in.release();
    h->pin_out.release();
    if (h->stream && h->own_stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

// ---- streams that really run side by side --------------------------------------------------------------------------------------
// The runtime maps the streams of a process onto a handful of hardware queues (GPU_MAX_HW_QUEUES, 4 by default): a new stream takes the
// queue with the fewest users, and two streams on one queue run their kernels one after the other.  Whether a view's stream lands beside
// its index's or behind it depends on what else the process has created -- measured in bench.py's process: ivfadc_search_batches 696 us
// per 16-batch call where a process with only the library's streams runs it in 615.  So a lane is PROBED when it is made: a 60 us spin
// kernel on each of the two streams; side by side they end after ~75 us, on one queue after 120 and more.  A stream that shares a queue
// with one it must overlap is replaced (the rejected ones are kept until the end, so the runtime hands out other queues) -- at most
// six tries, ~0.2 ms once per lane.  IVFADC_NO_STREAM_PROBE=1 skips it.
static int streams_serialised(hipStream_t a, hipStream_t b, bool &serial)
{
    serial = false;
    double best = 1e30;
    for (int rep = 0; rep < 2; ++rep) {
        HIP_TRY(hipStreamSynchronize(a));
        HIP_TRY(hipStreamSynchronize(b));
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(spin_us_kernel, dim3(1), dim3(64), 0, a, 60u);
        hipLaunchKernelGGL(spin_us_kernel, dim3(1), dim3(64), 0, b, 60u);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(a));
        HIP_TRY(hipStreamSynchronize(b));
        best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    serial = best > 105.0;
    return IVFADC_OK;
}

// *cand ends up on a hardware queue of its own with respect to every stream in fixed[] (as far as six tries reach)
static int ensure_overlap(const hipStream_t *fixed, int nfixed, hipStream_t *cand, int64_t *replaced)
{
    static const bool off = env_knob("IVFADC_NO_STREAM_PROBE") != nullptr;
    if (off) return IVFADC_OK;
    std::vector<hipStream_t> rejected;
    int rc = IVFADC_OK;
    for (int attempt = 0; attempt < 6; ++attempt) {
        bool clash = false;
        for (int i = 0; i < nfixed && !clash && rc == IVFADC_OK; ++i) {
            if (!fixed[i] || fixed[i] == *cand) continue;
            rc = streams_serialised(fixed[i], *cand, clash);
        }
        if (rc != IVFADC_OK || !clash) break;
        hipStream_t next = nullptr;
        if (hipStreamCreateWithFlags(&next, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
        rejected.push_back(*cand);
        *cand = next;
        if (replaced) ++*replaced;
    }
    for (hipStream_t s : rejected) (void)hipStreamDestroy(s);
    return rc;
}

// A read-only view of `src`: the same device arrays (quantizers, derived tables, lists), a stream and a workspace of its own.
static int clone_view(ivfadc_index *src, ivfadc_index **out)
{
    *out = nullptr;
    if (src->is_view) return fail(IVFADC_ERR_STATE, "a view of a view: take it from the index itself");
    TRY(set_device(src));
    if (src->dirty) TRY(upload_lists(src));   // the device copy is what the view shares
    if (!src->have_lists) return fail(IVFADC_ERR_STATE, "no inverted lists set");
    HIP_TRY(hipStreamSynchronize(src->stream));   // uploads and in-place edits have landed
    // (the host mirror stays with the index: moved aside while the struct is copied)
    std::vector<std::vector<uint8_t>> keep_codes;
    std::vector<std::vector<uint32_t>> keep_ids;
    keep_codes.swap(src->hl_codes);
    keep_ids.swap(src->hl_ids);
    ivfadc_index *v = nullptr;
    try { v = new ivfadc_index(*src); } catch (...) { keep_codes.swap(src->hl_codes); keep_ids.swap(src->hl_ids); throw; }
    keep_codes.swap(src->hl_codes);
    keep_ids.swap(src->hl_ids);
    DevBuf *shared[] = {&v->lb_f16, &v->lb_isc, &v->cent_f16, &v->tl_centres, &v->tl_off, &v->tl_rad, &v->tl_cent, &v->tl_slot, &v->centroids, &v->codebooks, &v->codebooks_t, &v->codebooks_p, &v->labels, &v->cnorm, &v->lb_split, &v->lb_n2, &v->lb_lab,
                        &v->lb_maxn, &v->nf_n2, &v->nf_lab, &v->cent_t, &v->cent_hi, &v->cent_lo, &v->list_pos, &v->list_len, &v->list_codeoff,
                        &v->codes, &v->ids};
    for (DevBuf *b : shared) b->alias();
    DevBuf *scratch[] = {&v->wg8_tabs, &v->wg8_items, &v->q_f16, &v->q_flags, &v->tl_gdist, &v->cdist2, &v->sq_keys, &v->sq_cnt, &v->sq_arrive, &v->tmin, &v->tlist, &v->q_hi, &v->q_lo, &v->gen_a, &v->gen_b, &v->gen_tmp,
                         &v->gen_off, &v->gen_tot, &v->app_stage, &v->q_stage, &v->cdist, &v->probe_list, &v->probe_dc, &v->probe_base, &v->list_cnt,
                         &v->bucket_off, &v->wi_off, &v->cursor, &v->bucket_items, &v->misc, &v->qthr, &v->part_keys, &v->part_cnt, &v->out_ids,
                         &v->out_dists, &v->out_counts, &v->assign, &v->enc_codes, &v->pts_stage, &v->dbg};
    for (DevBuf *b : scratch) b->forget();
    v->pin_in.forget();
    v->pin_out.forget();
    v->pending.clear();
    v->free_ev.clear();
    v->views.clear();
    v->n_views.store(0);
    v->my_ticket = -1;
    v->fb_pin.forget();      // (a view has counters, snapshots and an event of its own; it starts from its index's estimate)
    v->fb_ev = nullptr;
    v->fb_pending = false;
    v->fb_countdown = 0;
    v->fb_sp = v->fb_pp = 0;
    v->pipe_view = nullptr;
    v->pipe_ev_in = v->pipe_ev_out = nullptr;
    v->copy_stream = nullptr;
    v->copy_probed_a = v->copy_probed_b = nullptr;
    v->ingest_ev.clear();
    v->hstats = ivfadc_host_stats{};
    v->carry_queries = v->carry_scanned = v->carry_pruned = v->carry_surv = v->carry_fallbacks = v->carry_launches = 0;
    v->comm = nullptr;
    v->comm_stream = nullptr;
    v->comm_ready = nullptr;
    for (int i = 0; i < ivfadc_index::COMM_SLOTS; ++i) { v->comm_done[i] = nullptr; v->comm_busy[i] = false; }
    v->comm_ranks = 0;
    v->comm_rank = 0;
    v->comm_collectives = 0;
    v->comm_seq = 0;
    v->comm_waited = 0;
    for (int i = 0; i < ivfadc_index::COMM_SLOTS; ++i) v->comm_slot_seq[i] = 0;
    v->hint_ev = nullptr;
    v->hint_q = v->pf_q = v->avail_q = nullptr;
    v->hint_nq = v->pf_nq = v->avail_nq = 0;
    v->hint_token = v->pf_token = v->cur_token = 0;
    v->partial_keys = nullptr;
    v->partial_nq = -1;
    v->qthr_armed = 0;
    v->list_cnt_armed = false;
    v->profiling = false;
    v->profiling_level = 0;
    v->stats = ivfadc_stats{};
    v->scanned_base = v->fallback_base = v->pruned_base = v->surv_base = 0;
    v->visited_base = 0;
    v->inplace_appends = 0;
    v->stream = nullptr;
    v->own_stream = true;
    v->is_view = true;
    v->view_of = src;
    v->view_gen = src->generation;
    const hipError_t e = hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete v;   // (nothing of its own yet: every buffer is an alias or empty)
        return fail(IVFADC_ERR_HIP, "hipStreamCreateWithFlags failed: %s", hipGetErrorString(e));
    }
    // the point of a view is a second batch IN FLIGHT: its stream must not share a hardware queue with the index's
    {
        const hipStream_t fixed[1] = {src->stream};
        const int rc = ensure_overlap(fixed, 1, &v->stream, &src->hstats.streams_replaced);
        if (rc != IVFADC_OK) { (void)hipStreamDestroy(v->stream); delete v; return rc; }
    }
    {
        std::lock_guard<std::recursive_mutex> topo(g_topo_mu);
        src->views.push_back(v);
        src->n_views.fetch_add(1);
    }
    *out = v;
    return IVFADC_OK;
}

// settings that change how a search runs, copied to the internal second lane before every use
static void copy_search_config(ivfadc_index *dst, const ivfadc_index *src)
{
    dst->wg8_mode = src->wg8_mode;
    dst->allow_nf = src->allow_nf; dst->allow_sq = src->allow_sq; dst->sq_inside = src->sq_inside; dst->allow_lb = src->allow_lb;
    dst->force_lb = src->force_lb; dst->allow_bf16 = src->allow_bf16; dst->allow_f16 = src->allow_f16; dst->lb_use_f16 = src->lb_use_f16; dst->allow_prune = src->allow_prune; dst->allow_listed = src->allow_listed;
    dst->allow_filt = src->allow_filt; dst->allow_mfma = src->allow_mfma; dst->mfma_min_kc = src->mfma_min_kc; dst->ws_budget = src->ws_budget;
    dst->force_qg = src->force_qg; dst->force_chunk = src->force_chunk; dst->force_pg = src->force_pg;
    dst->part_n = src->part_n; dst->part_i = src->part_i;
    dst->tl_use = src->tl_use && dst->tl_cent.p != nullptr; dst->tl_mode = src->tl_mode;
}

int ivfadc_clone_view(ivfadc_t *h, ivfadc_t **out)
try {
    HandleLock lk_(h);
    if (!h || !out) return fail(IVFADC_ERR_INVALID, "null argument");
    return clone_view(h, out);
} IVF_CATCH

int ivfadc_set_lists(ivfadc_t *h, const int64_t *offsets, const uint8_t *codes, const uint32_t *ids)
try {
    HandleLock lk_(h);
    if (!h || !offsets) return fail(IVFADC_ERR_INVALID, "null argument");
    TRY(check_mutable(h, "ivfadc_set_lists"));
    const int kc = h->kc, m = h->m;
    if (offsets[0] != 0) return fail(IVFADC_ERR_INVALID, "offsets[0] must be 0");
    for (int l = 0; l < kc; ++l)
        if (offsets[l + 1] < offsets[l]) return fail(IVFADC_ERR_INVALID, "offsets must be non-decreasing (list %d)", l);
    const int64_t n = offsets[kc];
    if (n > (int64_t)0xFFFFFFFFll) return fail(IVFADC_ERR_ASSERT, "index capacity of UInt32 ids exceeded");
    if (n > 0 && (!codes || !ids)) return fail(IVFADC_ERR_INVALID, "null codes/ids");
    if (h->ksub < 256) {
        for (int64_t p = 0; p < n; ++p)
            for (int i = 0; i < m; ++i)
                if (!h->h_label_ok[(size_t)i * 256 + codes[(size_t)p * m + i]])
                    return fail(IVFADC_ERR_INVALID, "code byte %d of point %lld is not a label of codebook %d",
                                (int)codes[(size_t)p * m + i], (long long)p, i);
    }
    TRY(set_device(h));
    MutationScope ms;
    TRY(ms.begin(h, "ivfadc_set_lists"));
    for (int l = 0; l < kc; ++l) {
        const int64_t a = offsets[l], b = offsets[l + 1];
        h->h_len[l] = b - a;
        h->hl_codes[l].assign(codes + (size_t)a * m, codes + (size_t)b * m);
        h->hl_ids[l].assign(ids + a, ids + b);
    }
    return upload_lists(h);
} IVF_CATCH

int ivfadc_synth_lists(ivfadc_t *h, const int64_t *offsets, uint64_t seed)
try {
    HandleLock lk_(h);
    if (!h || !offsets) return fail(IVFADC_ERR_INVALID, "null argument");
    TRY(check_mutable(h, "ivfadc_synth_lists"));
    if (h->ksub != 256) return fail(IVFADC_ERR_INVALID, "synthetic lists need ksub == 256");
    TRY(set_device(h));
    std::vector<uint8_t> lab((size_t)h->m * 256);
    HIP_TRY(hipMemcpy(lab.data(), h->labels.p, lab.size(), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < lab.size(); ++i)
        if (lab[i] != (uint8_t)(i & 255)) return fail(IVFADC_ERR_INVALID, "synthetic lists need identity labels");
    const int kc = h->kc;
    for (int l = 0; l < kc; ++l)
        if (offsets[l + 1] < offsets[l]) return fail(IVFADC_ERR_INVALID, "offsets must be non-decreasing (list %d)", l);
    MutationScope ms;
    TRY(ms.begin(h, "ivfadc_synth_lists"));
    // no spare capacity: the id of a point is its canonical global position (ids == nullptr in the kernels)
    for (int l = 0; l < kc; ++l) {
        h->h_len[l] = offsets[l + 1] - offsets[l];
        std::vector<uint8_t>().swap(h->hl_codes[l]);
        std::vector<uint32_t>().swap(h->hl_ids[l]);
    }
    h->d_cap = h->h_len;
    size_t total = 0;
    int64_t slots = 0;
    TRY(layout_from_caps(h, total, slots));
    TRY(h->codes.ensure(total));
    TRY(upload_list_tables(h));
    HIP_TRY(hipMemsetAsync((uint8_t *)h->codes.p + (total - CODE_SLACK), 0, CODE_SLACK, h->stream));
    const int64_t maxdw = h->maxlen * (h->cs / 4);
    const unsigned gy = (unsigned)std::max<int64_t>(1, std::min<int64_t>(1024, (maxdw + 256 * 8 - 1) / (256 * 8)));
    hipLaunchKernelGGL(synth_codes_kernel, dim3((unsigned)kc, gy), dim3(256), 0, h->stream, h->codes.as<uint8_t>(),
                       h->list_pos.as<int64_t>(), h->list_len.as<u32>(), h->list_codeoff.as<int64_t>(), kc, h->m, h->cs, (u64)seed);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->have_lists = true;
    h->synthetic = true;
    h->dirty = false;
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_encode(ivfadc_t *h, int64_t n, const float *pts, int32_t *out_list, uint8_t *out_codes)
try {
    HandleLock lk_(h);
    if (!h) return fail(IVFADC_ERR_INVALID, "null handle");
    if (n < 0) return fail(IVFADC_ERR_INVALID, "n < 0");
    if (n == 0) return IVFADC_OK;
    if (!pts || !out_list || !out_codes) return fail(IVFADC_ERR_INVALID, "null argument");
    if (!h->rot.empty()) return fail(IVFADC_ERR_STATE, "ivfadc_encode: the residual quantizer carries a rotation (:opq); this index is served for search only");
    TRY(set_device(h));
    return encode_dev(h, n, pts, out_list, out_codes);
} IVF_CATCH

// The mirror edit of push!: `lst` / `cod` are the (already computed) assignment and codes of the new points.  The
// host mirror is the source of truth; the device copy is marked stale BEFORE the mirror changes and declared current
// again only after the in-place device update has fully succeeded, so a failure half-way (allocation, copy, launch)
// leaves a handle whose next search re-lays the lists out from the mirror instead of one that searches stale lists.
static int append_encoded(ivfadc_t *h, int64_t nnew, const int32_t *lst, const uint8_t *cod, const uint32_t *ids)
{
    const int m = h->m, cs = h->cs;
    const int64_t n_old = h->ntotal();
    MutationScope ms;
    TRY(ms.begin(h, "ivfadc_append"));
    TRY(set_device(h));
    // In place when the device layout is current and every target list has room; otherwise the host mirror takes
    // the points and the next search re-lays the lists out with fresh spare capacity.
    bool inplace = h->have_lists && !h->dirty && env_knob("IVFADC_NO_INPLACE_APPEND") == nullptr;
    if (inplace) {
        std::vector<int64_t> add((size_t)h->kc, 0);
        for (int64_t i = 0; i < nnew; ++i) add[lst[i]]++;
        for (int l = 0; l < h->kc && inplace; ++l)
            if (h->h_len[l] + add[l] > h->d_cap[l]) inplace = false;
    }
    std::vector<int64_t> dst;
    if (inplace) dst.resize((size_t)nnew * 2);
    // reserve first: the mirror edit below must not stop half-way on an allocation failure
    {
        std::vector<int64_t> add((size_t)h->kc, 0);
        for (int64_t i = 0; i < nnew; ++i) add[lst[i]]++;
        for (int l = 0; l < h->kc; ++l)
            if (add[l]) {
                h->hl_codes[l].reserve(h->hl_codes[l].size() + (size_t)add[l] * m);
                h->hl_ids[l].reserve(h->hl_ids[l].size() + (size_t)add[l]);
            }
    }
    h->dirty = true;
    // new points go to the END of their list, in call order (utils.jl:143-144)
    for (int64_t i = 0; i < nnew; ++i) {
        const int l = lst[i];
        if (inplace) {
            dst[2 * i] = h->d_codeoff[l] + h->h_len[l] * cs;
            dst[2 * i + 1] = h->d_pos[l] + h->h_len[l];
        }
        h->hl_codes[l].insert(h->hl_codes[l].end(), cod + (size_t)i * m, cod + (size_t)(i + 1) * m);
        h->hl_ids[l].push_back(ids[i]);
        h->h_len[l]++;
        h->maxlen = std::max(h->maxlen, h->h_len[l]);
    }
    h->n = n_old + nnew;
    if (inplace) {
        const size_t o_ids = (size_t)nnew * 16, o_codes = o_ids + align_up((size_t)nnew * 4, 16);
        const size_t bytes = o_codes + (size_t)nnew * m;
        std::vector<uint8_t> stage(bytes);
        memcpy(stage.data(), dst.data(), (size_t)nnew * 16);
        memcpy(stage.data() + o_ids, ids, (size_t)nnew * 4);
        memcpy(stage.data() + o_codes, cod, (size_t)nnew * m);
        TRY(h->app_stage.ensure(bytes));
        TRY(h2d_copy(h->app_stage.p, stage.data(), bytes, h->stream));
        const uint8_t *base = (const uint8_t *)h->app_stage.p;
        hipLaunchKernelGGL(append_scatter_kernel, dim3((unsigned)((nnew + 255) / 256)), dim3(256), 0, h->stream, nnew, m,
                           (const int64_t *)base, base + o_codes, (const u32 *)(base + o_ids), h->codes.as<uint8_t>(), h->ids.as<u32>());
        HIP_TRY(hipGetLastError());
        // lengths: only the touched lists change, but kc u32 is a single small copy
        std::vector<uint32_t> len32((size_t)h->kc);
        for (int l = 0; l < h->kc; ++l) len32[l] = (uint32_t)h->h_len[l];
        TRY(h2d_copy(h->list_len.p, len32.data(), (size_t)h->kc * 4, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));   // staging vectors are stack-owned
        h->inplace_appends++;
        h->dirty = false;                           // the device copy is current again
    }
    return IVFADC_OK;
}

static int append_check(ivfadc_t *h, int64_t nnew, const float *pts, const uint32_t *ids)
{
    if (!h) return fail(IVFADC_ERR_INVALID, "null handle");
    if (h->synthetic) return fail(IVFADC_ERR_STATE, "append is not available on device-synthesised lists");
    if (nnew < 0) return fail(IVFADC_ERR_INVALID, "nnew < 0");
    if (nnew > 0 && (!pts || !ids)) return fail(IVFADC_ERR_INVALID, "null argument");
    if (h->ntotal() + nnew > (int64_t)0xFFFFFFFFll) return fail(IVFADC_ERR_ASSERT, "Cannot index, exceeding index capacity of UInt32");
    return IVFADC_OK;
}

int ivfadc_append(ivfadc_t *h, int64_t nnew, const float *pts, const uint32_t *ids, int32_t *out_list, uint8_t *out_codes)
try {
    HandleLock lk_(h);
    TRY(append_check(h, nnew, pts, ids));
    TRY(check_mutable(h, "ivfadc_append"));
    if (!h->rot.empty()) return fail(IVFADC_ERR_STATE, "ivfadc_append: the residual quantizer carries a rotation (:opq); this index is served for search only");
    if (nnew == 0) return IVFADC_OK;
    TRY(set_device(h));
    std::vector<int32_t> lst((size_t)nnew);
    std::vector<uint8_t> cod((size_t)nnew * h->m);
    TRY(encode_dev(h, nnew, pts, lst.data(), cod.data()));   // nothing has changed yet if this fails
    TRY(append_encoded(h, nnew, lst.data(), cod.data(), ids));
    if (out_list) memcpy(out_list, lst.data(), (size_t)nnew * 4);
    if (out_codes) memcpy(out_codes, cod.data(), (size_t)nnew * h->m);
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_delete_ids(ivfadc_t *h, int64_t ndel, const uint32_t *del_ids, int64_t *out_removed)
try {
    HandleLock lk_(h);
    if (!h) return fail(IVFADC_ERR_INVALID, "null handle");
    TRY(check_mutable(h, "ivfadc_delete_ids"));
    if (h->synthetic) return fail(IVFADC_ERR_STATE, "delete is not available on device-synthesised lists");
    if (ndel < 0 || (ndel > 0 && !del_ids)) return fail(IVFADC_ERR_INVALID, "bad argument");
    if (out_removed) *out_removed = 0;
    if (ndel == 0) return IVFADC_OK;
    TRY(set_device(h));
    std::vector<uint32_t> want(del_ids, del_ids + ndel);
    std::sort(want.begin(), want.end());
    want.erase(std::unique(want.begin(), want.end()), want.end());
    // stale until the device-side compaction has succeeded (see append_encoded)
    const bool was_dirty = h->dirty;
    h->dirty = true;
    // host mirror, pass 1: drop the entries (stable) and collect the ids that were really there
    const int kc = h->kc, m = h->m;
    std::vector<uint32_t> rem;
    for (int l = 0; l < kc; ++l) {
        auto &lid = h->hl_ids[l];
        auto &lco = h->hl_codes[l];
        size_t wr = 0;
        for (size_t p = 0; p < lid.size(); ++p) {
            if (std::binary_search(want.begin(), want.end(), lid[p])) { rem.push_back(lid[p]); continue; }
            if (wr != p) {
                lid[wr] = lid[p];
                memmove(lco.data() + wr * m, lco.data() + p * m, m);
            }
            ++wr;
        }
        lid.resize(wr);
        lco.resize(wr * m);
    }
    if (rem.empty()) { h->dirty = was_dirty; return IVFADC_OK; }   // nothing was stored under these ids: mirror unchanged, views stay valid
    MutationScope ms;
    TRY(ms.begin(h, "ivfadc_delete_ids"));
    std::sort(rem.begin(), rem.end());
    // pass 2: every surviving id drops by the number of removed ids below it (_shift_inverse_index!, utils.jl:11-27)
    int64_t maxlen = 0, n = 0;
    for (int l = 0; l < kc; ++l) {
        for (uint32_t &v : h->hl_ids[l]) v -= (uint32_t)(std::lower_bound(rem.begin(), rem.end(), v) - rem.begin());
        h->h_len[l] = (int64_t)h->hl_ids[l].size();
        maxlen = std::max(maxlen, h->h_len[l]);
        n += h->h_len[l];
    }
    h->maxlen = maxlen;
    h->n = n;
    if (out_removed) *out_removed = (int64_t)rem.size();
    // device copy: the same compaction in place, one workgroup per list
    if (h->have_lists && !was_dirty) {
        TRY(h->app_stage.ensure(rem.size() * 4));
        TRY(h2d_copy(h->app_stage.p, rem.data(), rem.size() * 4, h->stream));
        hipLaunchKernelGGL(delete_compact_kernel, dim3((unsigned)kc), dim3(256), 0, h->stream, h->app_stage.as<u32>(), (u32)rem.size(),
                           h->list_pos.as<int64_t>(), h->list_codeoff.as<int64_t>(), h->list_len.as<u32>(), h->codes.as<uint8_t>(),
                           h->ids.as<u32>(), h->cs);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(h->stream));   // rem is stack-owned
        h->dirty = false;
    }
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_shift_ids(ivfadc_t *h, int32_t delta)
try {
    HandleLock lk_(h);
    if (!h) return fail(IVFADC_ERR_INVALID, "null handle");
    TRY(check_mutable(h, "ivfadc_shift_ids"));
    if (h->synthetic) return fail(IVFADC_ERR_STATE, "not available on device-synthesised lists");
    if (delta == 0) return IVFADC_OK;
    TRY(set_device(h));
    MutationScope ms;
    TRY(ms.begin(h, "ivfadc_shift_ids"));
    const bool was_dirty = h->dirty;
    h->dirty = true;
    for (int l = 0; l < h->kc; ++l)
        for (uint32_t &v : h->hl_ids[l]) v += (uint32_t)delta;
    if (h->have_lists && !was_dirty) {
        int64_t slots = 0;
        for (int l = 0; l < h->kc; ++l) slots += h->d_cap[l];
        if (slots > 0) {
            hipLaunchKernelGGL(shift_ids_kernel, dim3((unsigned)std::min<int64_t>(4096, (slots + 255) / 256)), dim3(256), 0, h->stream,
                               h->ids.as<u32>(), slots, (u32)delta);
            HIP_TRY(hipGetLastError());
        }
        h->dirty = false;
    }
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_search_device(ivfadc_t *h, int64_t nq, const float *d_queries, int K, int w, uint32_t *d_out_ids, float *d_out_dists,
                         int32_t *d_out_counts)
try {
    HandleLock lk_(h);
    TRY(check_search_args(h, nq, K, w));
    if (nq > 0 && (!d_queries || !d_out_ids || !d_out_dists || !d_out_counts)) return fail(IVFADC_ERR_INVALID, "null buffer");
    {
        // are several lanes of this replica in use side by side?  Every device-entry search takes a ticket from the root index; a handle
        // whose two consecutive tickets are more than one apart has had a neighbour search in between
        ivfadc_index *root = (h->is_view && h->view_of) ? h->view_of : h;
        const int t = root->lane_ticket.fetch_add_get(1);
        h->dev_entry = h->my_ticket >= 0 && t - h->my_ticket > 1;
        h->my_ticket = t;
    }
    struct Reset { ivfadc_index *h; ~Reset() { h->dev_entry = false; } } reset_{h};   // (also when an exception unwinds into IVF_CATCH)
    return search_dev(h, nq, d_queries, K, w, d_out_ids, d_out_dists, d_out_counts);
} IVF_CATCH

// ---- list-partitioned multi-GPU mode: strong scaling of a FIXED global batch --------------------------------------------------------------
// Queries sharded over replicas cannot scale a fixed batch past the point where every rank streams most of the index for its few
// queries (4.9x at 8 GPUs on the SIFT1B shape).  Here every rank keeps the replica and ALL queries, runs the
// coarse search (identical on every rank), scans only the probed lists l with l % nparts == part -- 1 / nparts of the code bytes, each list
// still shared by as many queries as on one GPU -- and leaves its K smallest KEYS per query (distance bits << 32 | visit order: probes are
// independent given the bound, index.jl:228-255, and visit orders are global).  ONE all-gather of nq x K keys per rank, then the K-way
// merge (partial_merge_kernel) on every rank.
int ivfadc_set_list_partition(ivfadc_t *h, int nparts, int part)
try {
    HandleLock lk_(h);
    if (!h) return fail(IVFADC_ERR_INVALID, "null handle");
    if (nparts < 1 || part < 0 || part >= nparts) return fail(IVFADC_ERR_INVALID, "part %d of %d", part, nparts);
    h->part_n = nparts;
    h->part_i = part;
    h->partial_nq = -1;
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_search_device_partial(ivfadc_t *h, int64_t nq, const float *d_queries, int K, int w, uint64_t *d_keys, int32_t *d_counts)
try {
    HandleLock lk_(h);
    TRY(check_search_args(h, nq, K, w));
    if (nq > 0 && (!d_queries || !d_keys || !d_counts)) return fail(IVFADC_ERR_INVALID, "null buffer");
    if (h->part_n < 2) return fail(IVFADC_ERR_STATE, "ivfadc_set_list_partition(h, nparts >= 2, part) first");
    h->partial_keys = d_keys;
    h->partial_nq = -1;
    const int rc = search_dev(h, nq, d_queries, K, w, nullptr, nullptr, d_counts);
    h->partial_keys = nullptr;
    if (rc == IVFADC_OK) { h->partial_nq = nq; h->partial_w = w; h->partial_K = K; }
    return rc;
} IVF_CATCH

static int merge_partials_dev(ivfadc_t *h, int64_t nq, int K, int nparts, const uint64_t *d_keys, size_t stride_u64, const int32_t *d_counts,
                              size_t stride_i32, uint32_t *d_ids, float *d_dists, int32_t *d_out_counts)
{
    if (h->partial_nq != nq || h->partial_K != K)
        return fail(IVFADC_ERR_STATE, "ivfadc_merge_partials_device: the handle's last partial search was on %lld queries with K = %d", (long long)h->partial_nq, h->partial_K);
    TRY(set_device(h));
    const bool small_k = K <= 64;
    const int cap = small_k ? 64 : std::max(128, pow2ceil(K + 64));
    const size_t mlds = small_k ? 0 : (size_t)4 * cap * 8;
    const u32 *idp = h->synthetic ? (const u32 *)nullptr : h->ids.as<u32>();
    if (mlds > (size_t)(32 << 10)) { int occ_unused = 0; TRY(fn_occupancy(h, (const void *)partial_merge_kernel<false>, mlds, occ_unused, false)); }
    if (small_k)
        hipLaunchKernelGGL(partial_merge_kernel<true>, dim3((unsigned)((nq + 3) / 4)), dim3(256), mlds, h->stream, (int)nq, h->partial_w, K, cap, nparts,
                           (const u64 *)d_keys, stride_u64, (const int *)d_counts, stride_i32, h->probe_list.as<int>(), h->probe_base.as<u32>(),
                           h->list_pos.as<int64_t>(), idp, d_ids, d_dists, d_out_counts);
    else
        hipLaunchKernelGGL(partial_merge_kernel<false>, dim3((unsigned)((nq + 3) / 4)), dim3(256), mlds, h->stream, (int)nq, h->partial_w, K, cap, nparts,
                           (const u64 *)d_keys, stride_u64, (const int *)d_counts, stride_i32, h->probe_list.as<int>(), h->probe_base.as<u32>(),
                           h->list_pos.as<int64_t>(), idp, d_ids, d_dists, d_out_counts);
    HIP_TRY(hipGetLastError());
    return IVFADC_OK;
}

int ivfadc_merge_partials_device(ivfadc_t *h, int64_t nq, int K, int nparts, const uint64_t *d_keys_all, const int32_t *d_counts_all,
                                 uint32_t *d_ids, float *d_dists, int32_t *d_counts)
try {
    HandleLock lk_(h);
    if (!h) return fail(IVFADC_ERR_INVALID, "null handle");
    if (nq < 0 || K < 1 || nparts < 1) return fail(IVFADC_ERR_INVALID, "nq, K, nparts");
    if (nq == 0) return IVFADC_OK;
    if (!d_keys_all || !d_counts_all || !d_ids || !d_dists || !d_counts) return fail(IVFADC_ERR_INVALID, "null buffer");
    return merge_partials_dev(h, nq, K, nparts, d_keys_all, (size_t)nq * K, d_counts_all, (size_t)nq, d_ids, d_dists, d_counts);
} IVF_CATCH

// ---- host-pointer entries ----------------------------------------------------------------------------------------------------
// The reference's contract is host arrays in, host arrays out (index.jl:261-265).  What that costs on top of the device-resident search
// (tools/micro/host_path.hip, one MI355X box, 1024 x 128 f32 queries in, 84 KB out): the copy into pinned memory 7 us, a copy-engine H2D
// 19 us alone and ~10 us more than a copy KERNEL inside a chain, a D2H copy 10 us, and every in-stream hand-off between the SDMA and the
// compute queue a few more.  So: queries are ingested by a kernel that reads page-locked host memory (the caller's own array when the
// library knows it -- ivfadc_host_register / ivfadc_host_alloc -- else pin_in), a handful of queries (latency path) are read in place,
// and the final kernel of the search writes ids / distances / counts straight into page-locked host memory (the caller's arrays when
// known, else pin_out): no device result block and no D2H copy in the chain.  IVFADC_HOST_LEGACY=1 keeps round 4's copy chain (A/B).
static inline double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static bool host_legacy() { return env_knob("IVFADC_HOST_LEGACY") != nullptr; }   // (read per call: bench.py measures both chains in one run)

// page-locked host rows -> device, on stream s
static int ingest_rows(ivfadc_index *h, hipStream_t s, const void *src, void *dst, size_t bytes)
{
    static const bool by_dma = env_knob("IVFADC_INGEST_DMA") != nullptr;   // A/B: the copy engine instead of the compute queue
    if (bytes == 0) return IVFADC_OK;
    if (by_dma || bytes > ((size_t)256 << 20) || (bytes & 3) != 0) {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
        return IVFADC_OK;
    }
    const size_t nwords = bytes / 4;
    const bool vec = ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0;
    const size_t nvec = vec ? nwords / 4 : 0;
    const size_t items = vec ? nvec + 3 : nwords;
    const unsigned grid = (unsigned)std::max<size_t>(1, std::min<size_t>((items + 255) / 256, (size_t)4 * h->num_cu));
    hipLaunchKernelGGL(host_ingest_kernel, dim3(grid), dim3(256), 0, s, (const u32 *)src, (u32 *)dst, nvec, nwords);
    HIP_TRY(hipGetLastError());
    return IVFADC_OK;
}

// host-pointer search in two halves so several handles (devices) can be in flight at once (ivfadc_mg_search)
static int search_enqueue(ivfadc_t *h, int64_t nq, const float *queries, int K, int w, uint32_t *out_ids, float *out_dists, int32_t *out_counts)
try {
    TRY(set_device(h));
    h->dev_entry = false;
    const double t0 = now_us();
    const size_t qbytes = (size_t)nq * h->d * 4;
    const size_t idb = (size_t)nq * K * 4, cb = (size_t)nq * 4;
    const size_t obytes = 2 * idb + cb;
    h->hstats.calls++;
    h->hstats.batches++;
    h->hc_legacy = host_legacy();
    if (h->hc_legacy) {
        // round 4's chain: memcpy -> H2D copy -> kernels -> D2H copy of the packed block -> memcpy
        TRY(h->q_stage.ensure(qbytes));
        TRY(h->out_ids.ensure(obytes));
        TRY(h->pin_in.ensure(qbytes));
        TRY(h->pin_out.ensure(obytes));
        memcpy(h->pin_in.p, queries, qbytes);
        const double t1 = now_us();
        HIP_TRY(hipMemcpyAsync(h->q_stage.p, h->pin_in.p, qbytes, hipMemcpyHostToDevice, h->stream));
        uint8_t *dout = (uint8_t *)h->out_ids.p;
        TRY(search_dev(h, nq, h->q_stage.as<float>(), K, w, (uint32_t *)dout, (float *)(dout + idb), (int32_t *)(dout + 2 * idb)));
        HIP_TRY(hipMemcpyAsync(h->pin_out.p, dout, obytes, hipMemcpyDeviceToHost, h->stream));
        h->hc_out_direct = false;
        h->hstats.stage_in_us += t1 - t0;
        h->hstats.enqueue_us += now_us() - t1;
        return IVFADC_OK;
    }
    const float *src = queries;
    if (host_known(queries, qbytes)) {
        h->hstats.queries_direct++;
    } else {
        TRY(h->pin_in.ensure(qbytes));
        memcpy(h->pin_in.p, queries, qbytes);
        src = (const float *)h->pin_in.p;
    }
    const double t1 = now_us();
    if (h->dirty) TRY(upload_lists(h));
    const float *d_q = src;
    static const bool no_zero_copy = env_knob("IVFADC_NO_ZERO_COPY") != nullptr;
    if (!no_zero_copy && h->part_n <= 1 && (((uintptr_t)src) & 15) == 0 && sq_eligible(h, nq, K, w)) {
        // the latency path: a handful of rows, read in place by the two launches (a few KB over PCIe; no ingest step in the chain)
        h->hstats.zero_copy++;
    } else {
        TRY(h->q_stage.ensure(qbytes));
        TRY(ingest_rows(h, h->stream, src, h->q_stage.p, qbytes));
        d_q = h->q_stage.as<float>();
    }
    uint32_t *oi = out_ids;
    float *od = out_dists;
    int32_t *oc = out_counts;
    h->hc_out_direct = host_known(out_ids, idb) && host_known(out_dists, idb) && host_known(out_counts, cb);
    if (h->hc_out_direct) {
        h->hstats.results_direct++;
    } else {
        TRY(h->pin_out.ensure(obytes));
        uint8_t *po = (uint8_t *)h->pin_out.p;
        oi = (uint32_t *)po;
        od = (float *)(po + idb);
        oc = (int32_t *)(po + 2 * idb);
    }
    TRY(search_dev(h, nq, d_q, K, w, oi, od, oc));
    h->hstats.stage_in_us += t1 - t0;
    h->hstats.enqueue_us += now_us() - t1;
    return IVFADC_OK;
} IVF_CATCH

static int search_finish(ivfadc_t *h, int64_t nq, int K, uint32_t *out_ids, float *out_dists, int32_t *out_counts)
try {
    TRY(set_device(h));
    const double t0 = now_us();
    TRY(wait_stream(h));
    const double t1 = now_us();
    if (!h->hc_out_direct) {
        const size_t idb = (size_t)nq * K * 4, cb = (size_t)nq * 4;
        const uint8_t *hout = (const uint8_t *)h->pin_out.p;
        memcpy(out_ids, hout, idb);
        memcpy(out_dists, hout + idb, idb);
        memcpy(out_counts, hout + 2 * idb, cb);
    }
    h->hstats.wait_us += t1 - t0;
    h->hstats.stage_out_us += now_us() - t1;
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_search(ivfadc_t *h, int64_t nq, const float *queries, int K, int w, uint32_t *out_ids, float *out_dists,
                  int32_t *out_counts)
try {
    HandleLock lk_(h);
    TRY(check_search_args(h, nq, K, w));
    if (nq == 0) return IVFADC_OK;
    if (!queries || !out_ids || !out_dists || !out_counts) return fail(IVFADC_ERR_INVALID, "null buffer");
    const int rc = search_enqueue(h, nq, queries, K, w, out_ids, out_dists, out_counts);
    if (rc != IVFADC_OK) {
        // whatever was enqueued before the failure may still be writing the staging buffers (or the caller's arrays): it ends first
        const std::string msg = g_err;
        if (h->stream && hipSetDevice(h->device) == hipSuccess) (void)hipStreamSynchronize(h->stream);
        g_err = msg;
        return rc;
    }
    return search_finish(h, nq, K, out_ids, out_dists, out_counts);
} IVF_CATCH

// the cumulative counters of an internal view that is about to be destroyed stay in its index's totals
static void fold_view_counters(ivfadc_index *h, ivfadc_index *v)
{
    ivfadc_stats st;
    const std::string keep = g_err;
    if (ivfadc_get_stats(v, &st) == IVFADC_OK) {
        h->carry_queries += st.queries;
        h->carry_scanned += st.scanned_points;
        h->carry_pruned += st.pruned_points;
        h->carry_surv += st.lb_survivors;
        h->carry_fallbacks += st.coarse_fallbacks;
        h->carry_launches += st.scan_launches;
    }
    g_err = keep;
}

// A run of consecutive batches from host memory: what a serving loop of knn_search(ivfadc, points, k; w) calls does (index.jl:261-273 once
// per batch), as ONE call -- batch i is searched with batch i + 1 named as its successor (ivfadc_set_next_queries / ivfadc_set_query_token
// with tokens of the library's own, on buffers the library owns: nothing the caller does can make a stale row match).  Each batch's results
// are exactly ivfadc_search's.  The queries are ingested batch by batch on a copy lane of their own (a stream that carries nothing else),
// a few batches ahead of the searches, and every search writes its results where the caller reads them: the device never waits for the
// host between batches, and the call ends when the two search lanes have drained.
int ivfadc_search_batches(ivfadc_t *h, int nbatches, const int64_t *batch_nq, const float *queries, int K, int w, uint32_t *out_ids,
                          float *out_dists, int32_t *out_counts)
try {
    HandleLock lk_(h);
    if (nbatches < 0 || (nbatches > 0 && !batch_nq)) return fail(IVFADC_ERR_INVALID, "nbatches < 0 or null batch sizes");
    int64_t total = 0;
    for (int b = 0; b < nbatches; ++b) {
        if (batch_nq[b] < 0) return fail(IVFADC_ERR_INVALID, "batch %d: nq < 0", b);
        total += batch_nq[b];
    }
    TRY(check_search_args(h, total, K, w));
    if (total == 0) return IVFADC_OK;
    if (!queries || !out_ids || !out_dists || !out_counts) return fail(IVFADC_ERR_INVALID, "null buffer");
    TRY(set_device(h));
    const double t_enter = now_us();
    const size_t qbytes = (size_t)total * h->d * 4;
    const size_t idb = (size_t)total * K * 4, cb = (size_t)total * 4;
    const size_t obytes = 2 * idb + cb;
    const bool legacy = host_legacy();
    const bool q_known = !legacy && host_known(queries, qbytes);
    const bool o_known = !legacy && host_known(out_ids, idb) && host_known(out_dists, idb) && host_known(out_counts, cb);
    TRY(h->q_stage.ensure(qbytes));
    if (!q_known) TRY(h->pin_in.ensure(qbytes));
    if (!o_known) TRY(h->pin_out.ensure(obytes));
    if (legacy) TRY(h->out_ids.ensure(obytes));
    // where the kernels write: the caller's arrays, the library's pinned block, or (legacy) a device block that is copied back at the end
    uint8_t *ob = legacy ? (uint8_t *)h->out_ids.p : (uint8_t *)h->pin_out.p;
    uint32_t *oi = o_known ? out_ids : (uint32_t *)ob;
    float *od = o_known ? out_dists : (float *)(ob + idb);
    int32_t *oc = o_known ? out_counts : (int32_t *)(ob + 2 * idb);
    const uint8_t *qsrc = q_known ? (const uint8_t *)queries : (const uint8_t *)h->pin_in.p;
    h->hstats.calls++;
    if (q_known) h->hstats.queries_direct++;
    if (o_known) h->hstats.results_direct++;
    std::vector<int64_t> start, cnt;   // the non-empty batches, in order
    int64_t run = 0;
    for (int b = 0; b < nbatches; ++b) {
        if (batch_nq[b] > 0) { start.push_back(run); cnt.push_back(batch_nq[b]); }
        run += batch_nq[b];
    }
    const size_t nbat = start.size();
    h->hstats.batches += (int64_t)nbat;
    const float *dq = h->q_stage.as<float>();
    const uint64_t base = h->own_token;
    h->own_token += nbat;
    auto token_of = [&](size_t i) { return (base + i + 1) | ((uint64_t)1 << 63); };   // never 0; the library's own numbering
    // Two batches in flight: even batches on this handle, odd ones on a view of it (second stream, second workspace), each lane naming
    // ITS next batch (i + 2) as the successor.  Not while profiling (the statistics are this handle's) and not for a view.
    static const bool no_pipe = env_knob("IVFADC_NO_PIPELINE") != nullptr;
    ivfadc_index *lane2 = nullptr;
    if (h->pipe_view && h->pipe_view->view_gen != h->generation) {
        fold_view_counters(h, h->pipe_view);
        ivfadc_destroy(h->pipe_view);
        h->pipe_view = nullptr;
    }
    if (h->dirty) TRY(upload_lists(h));
    if (!no_pipe && !h->is_view && !h->profiling && nbat >= 2) {
        if (!h->pipe_view) TRY(clone_view(h, &h->pipe_view));
        if (!h->pipe_ev_in) HIP_TRY(hipEventCreateWithFlags(&h->pipe_ev_in, hipEventDisableTiming));
        lane2 = h->pipe_view;
        copy_search_config(lane2, h);
        lane2->own_token = 0;
        HIP_TRY(hipEventRecord(h->pipe_ev_in, h->stream));            // in-place edits of the lists queued on this handle's stream come first
        HIP_TRY(hipStreamWaitEvent(lane2->stream, h->pipe_ev_in, 0));
    }
    // (stream priorities do not help here -- measured: a copy lane or a second lane of another priority class is no faster in a process
    // that owns few streams and up to 65 % slower in one that owns many; profiles/r05_batches_stream_priorities.txt)
    if (!h->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    {
        // ... and the copy lane beside both search lanes (probed once per pair of lanes)
        const hipStream_t l2 = lane2 ? lane2->stream : nullptr;
        if (h->copy_probed_a != h->stream || h->copy_probed_b != l2) {
            const hipStream_t fixed[2] = {h->stream, l2};
            TRY(ensure_overlap(fixed, 2, &h->copy_stream, &h->hstats.streams_replaced));
            h->copy_probed_a = h->stream;
            h->copy_probed_b = l2;
        }
    }
    // However this call ends, nothing it enqueued is left running: a failing search returns with both lanes and the copy lane drained
    // (the next call reuses the staging buffers, and the caller's arrays are the caller's again on return).
    struct Drain {
        ivfadc_index *h, *lane2;
        bool armed;
        ~Drain()
        {
            if (!armed) return;
            const std::string keep = g_err;
            if (hipSetDevice(h->device) == hipSuccess) {
                if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
                if (lane2 && lane2->stream) (void)hipStreamSynchronize(lane2->stream);
                if (h->stream) (void)hipStreamSynchronize(h->stream);
            }
            g_err = keep;
        }
    } drain{h, lane2, true};
    const size_t stride = lane2 ? 2 : 1;
    auto lane_of = [&](size_t i) { return (lane2 && (i & 1)) ? lane2 : h; };
    // Upload groups: consecutive batches that travel together (one ingest launch, one event): one batch per lane -- the first searches
    // start as soon as THEIR rows are on the device, and the copy lane, which moves a batch in a third of the time a lane needs to search
    // one, is further ahead with every group -- or more where a call brings more than ~500 batches (at most ~256 groups per call).
    std::vector<size_t> gb;   // group g = batches [gb[g], gb[g + 1])
    {
        // (the very first groups are single batches: lane 0 starts when batch 0 has landed, while batch 1 is still on its way)
        const size_t per = std::max<size_t>(stride, (nbat + 251) / 252);
        size_t at = 0;
        while (at < nbat) {
            gb.push_back(at);
            at += (at < stride) ? 1 : per;
        }
        gb.push_back(nbat);
    }
    const size_t ngroups = gb.size() - 1;
    std::vector<size_t> group_of(nbat);
    for (size_t g = 0; g < ngroups; ++g)
        for (size_t i = gb[g]; i < gb[g + 1]; ++i) group_of[i] = g;
    while (h->ingest_ev.size() < ngroups) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->ingest_ev.push_back(e);
    }
    std::vector<char> ev_seen(ngroups, 0);   // 1: the group's ingest is known to have finished
    size_t up_next = 0;                      // groups [0, up_next) are on their way
    double t_stage = 0.0;
    auto upload_until = [&](size_t want_batches) -> int {
        const size_t last = std::min(want_batches, nbat);
        while (up_next < ngroups && gb[up_next] < last) {
            const size_t b0 = gb[up_next], b1 = gb[up_next + 1];
            const size_t off = (size_t)start[b0] * h->d * 4;
            const size_t bytes = ((size_t)start[b1 - 1] + (size_t)cnt[b1 - 1] - (size_t)start[b0]) * h->d * 4;
            if (!q_known) {
                const double t0 = now_us();
                memcpy((uint8_t *)h->pin_in.p + off, (const uint8_t *)queries + off, bytes);
                t_stage += now_us() - t0;
            }
            if (legacy) HIP_TRY(hipMemcpyAsync((uint8_t *)h->q_stage.p + off, qsrc + off, bytes, hipMemcpyHostToDevice, h->copy_stream));
            else TRY(ingest_rows(h, h->copy_stream, qsrc + off, (uint8_t *)h->q_stage.p + off, bytes));
            HIP_TRY(hipEventRecord(h->ingest_ev[up_next], h->copy_stream));
            ++up_next;
        }
        return IVFADC_OK;
    };
    // A lane is held back only by an ingest that has not finished when its search is issued: events that have fired cost the lane nothing
    // (a wait in a search stream idles its queue for microseconds whether the event has fired or not).  The copy lane is in order, so
    // the newest group a search needs covers the ones before it.  Returns the event the lane still has to wait for (null: none).
    std::vector<size_t> lane_done(2, 0);   // groups [0, lane_done[l]) are known complete to lane l (seen fired, or waited for in its stream)
    auto pending_event = [&](size_t lane_ix, size_t batch, hipEvent_t &ev) -> int {
        ev = nullptr;
        const size_t g = group_of[std::min(batch, nbat - 1)];
        if (lane_done[lane_ix] > g) return IVFADC_OK;
        if (!ev_seen[g]) {
            const hipError_t q = hipEventQuery(h->ingest_ev[g]);
            if (q == hipSuccess) { for (size_t x = 0; x <= g; ++x) ev_seen[x] = 1; }
            else if (q != hipErrorNotReady) return fail(IVFADC_ERR_HIP, "hipEventQuery failed: %s", hipGetErrorString(q));
        }
        if (!ev_seen[g]) ev = h->ingest_ev[g];
        lane_done[lane_ix] = g + 1;          // (the caller puts the wait into the lane's stream, or hands it to the search)
        return IVFADC_OK;
    };
    // From a lane's second search on, the HOST waits for a missing ingest (the lane still has its previous search to run, and the copy
    // lane moves a batch in a third of the time a search takes, so this is rare and short): a wait in the search stream would cost the
    // lane's queue ~6 us per search whether or not the event has fired by the time the queue gets there.  The first search of a lane
    // must start the moment its rows land: there the wait goes into the stream.
    auto host_wait = [&](hipEvent_t &ev) -> int {
        if (!ev) return IVFADC_OK;
        const double t0 = now_us();
        for (;;) {
            const hipError_t q = hipEventQuery(ev);
            if (q == hipSuccess) { ev = nullptr; return IVFADC_OK; }
            if (q != hipErrorNotReady) return fail(IVFADC_ERR_HIP, "hipEventQuery failed: %s", hipGetErrorString(q));
            if (now_us() - t0 > 500.0) return IVFADC_OK;   // (something else owns the copy engine's attention: let the stream wait)
        }
    };
    for (size_t i = 0; i < nbat; ++i) {
        ivfadc_index *ln = lane_of(i);
        const size_t lane_ix = (lane2 && (i & 1)) ? 1 : 0;
        TRY(upload_until(i + 2 * stride + 1));   // batch i, the one it names as its successor, and the group after that are on their way
        hipEvent_t ev = nullptr;
        TRY(pending_event(lane_ix, i, ev));  // this batch's own rows: before the search's first launch
        if (i >= stride) TRY(host_wait(ev));
        if (ev) HIP_TRY(hipStreamWaitEvent(ln->stream, ev, 0));
        ln->cur_token = token_of(i);         // batch i's rows, if any stand, were hinted with this very token by the lane's step before
        ln->hint_ev = nullptr;
        if (i + stride < nbat) {
            ln->hint_q = dq + (size_t)start[i + stride] * h->d;
            ln->hint_nq = cnt[i + stride];
            ln->hint_token = token_of(i + stride);
            // the successor's rows are read by the riders of this search's scan launch only: the wait for them goes in front of THAT
            // launch, behind the search's own coarse stage
            TRY(pending_event(lane_ix, i + stride, ev));
            if (i >= stride) TRY(host_wait(ev));
            ln->hint_ev = ev;
        }
        TRY(search_dev(ln, cnt[i], dq + (size_t)start[i] * h->d, K, w, oi + (size_t)start[i] * K, od + (size_t)start[i] * K, oc + start[i]));
    }
    const double t_issued = now_us();
    if (legacy) {
        if (lane2) {
            if (!h->pipe_ev_out) HIP_TRY(hipEventCreateWithFlags(&h->pipe_ev_out, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(h->pipe_ev_out, lane2->stream));
            HIP_TRY(hipStreamWaitEvent(h->stream, h->pipe_ev_out, 0));
        }
        HIP_TRY(hipMemcpyAsync(h->pin_out.p, ob, obytes, hipMemcpyDeviceToHost, h->stream));
    }
    if (lane2) TRY(wait_stream(lane2));
    TRY(wait_stream(h));
    drain.armed = false;   // (the copy lane's work ended before the searches that read it)
    const double t_done = now_us();
    if (!o_known) {
        const uint8_t *hout = (const uint8_t *)h->pin_out.p;
        memcpy(out_ids, hout, idb);
        memcpy(out_dists, hout + idb, idb);
        memcpy(out_counts, hout + 2 * idb, cb);
    }
    const double t_out = now_us();
    h->hstats.stage_in_us += t_stage;
    h->hstats.enqueue_us += (t_issued - t_enter) - t_stage;
    h->hstats.wait_us += t_done - t_issued;
    h->hstats.stage_out_us += t_out - t_done;
    return IVFADC_OK;
} IVF_CATCH

// ---- single-process multi-device front end: index replicated, contiguous query blocks per device -------------
// (queries are independent, index.jl:269-271.)  Result merge: by default every device's block is
// copied to the caller's host arrays as it completes; with ivfadc_mg_set_gather(g, 1) the packed per-device blocks are
// exchanged by ONE ncclAllGather (RCCL over xGMI, ncclCommInitAll, one stream per device) so that every device holds the
// whole batch's results, and the host reads device 0's copy.  RCCL is bound at run time (dlopen), so the library loads
// on hosts without it and in processes where torch has already loaded its own copy.
extern "C++" {
struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    bool ok = false;
};

static RcclApi &rccl_api()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
        }
        if (!api.lib) return;
        api.CommInitAll = (decltype(api.CommInitAll))dlsym(api.lib, "ncclCommInitAll");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
        api.AllGather = (decltype(api.AllGather))dlsym(api.lib, "ncclAllGather");
        api.GroupStart = (decltype(api.GroupStart))dlsym(api.lib, "ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))dlsym(api.lib, "ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
        api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommInitAll && api.CommDestroy && api.AllGather && api.GroupStart && api.GroupEnd && api.GetErrorString;
    });
    return api;
}

#define NCCL_TRY(expr)                                                                                      \
    do {                                                                                                    \
        ncclResult_t r_ = (expr);                                                                           \
        if (r_ != ncclSuccess) return fail(IVFADC_ERR_HIP, "%s failed: %s", #expr, rccl_api().GetErrorString(r_)); \
    } while (0)

struct ivfadc_mg {
    std::recursive_mutex mu;             // calls on one multi-device handle are serialised (its replicas are only reached through it)
    std::vector<ivfadc_t *> dev;
    int gather_mode = 0;                 // 0: host gather; 1: ncclAllGather of the packed blocks
    std::vector<ncclComm_t> comms;       // one per device (gather_mode 1)
    std::vector<DevBuf> gath;            // per device: the gathered [G][block] results
    PinnedBuf host_gath;
    int64_t collectives = 0;             // statistics: all-gathers issued
};

// run f(r) for every replica on its own host thread (uploads and device-side synthesis of the replicas overlap);
// returns the first failure, with its message
template <class F> static int mg_parallel(ivfadc_mg *g, F f)
{
    const size_t G = g->dev.size();
    std::vector<int> rc(G, IVFADC_OK);
    std::vector<std::string> msg(G);
    std::vector<std::thread> th;
    for (size_t r = 0; r < G; ++r)
        th.emplace_back([&, r] {
            try { rc[r] = f(r); } catch (...) { rc[r] = on_exception(); }
            if (rc[r] != IVFADC_OK) msg[r] = g_err;   // g_err is thread-local
        });
    for (auto &t : th) t.join();
    for (size_t r = 0; r < G; ++r)
        if (rc[r] != IVFADC_OK) { g_err = msg[r]; return rc[r]; }
    return IVFADC_OK;
}

}  // extern "C++"

int ivfadc_mg_create(ivfadc_mg_t **out, int ndev, const int *devices, int d, int kc, int m, int ksub, const float *centroids,
                     const float *codebooks, const uint8_t *code_labels)
try {
    if (!out) return fail(IVFADC_ERR_INVALID, "out is null");
    *out = nullptr;
    if (ndev < 1 || !devices) return fail(IVFADC_ERR_INVALID, "ndev must be >= 1");
    ivfadc_mg *g = new ivfadc_mg();
    for (int r = 0; r < ndev; ++r) {
        ivfadc_t *h = nullptr;
        const int rc = ivfadc_create(&h, devices[r], d, kc, m, ksub, centroids, codebooks, code_labels);
        if (rc != IVFADC_OK) { ivfadc_mg_destroy(g); return rc; }
        g->dev.push_back(h);
    }
    g->gath.resize((size_t)ndev);
    *out = g;
    return IVFADC_OK;
} IVF_CATCH

void ivfadc_mg_destroy(ivfadc_mg_t *g)
{
    if (!g) return;
    for (size_t r = 0; r < g->comms.size(); ++r)
        if (g->comms[r]) { (void)hipSetDevice(g->dev[r]->device); (void)rccl_api().CommDestroy(g->comms[r]); }
    for (size_t r = 0; r < g->gath.size(); ++r) {
        if (r < g->dev.size()) (void)hipSetDevice(g->dev[r]->device);
        g->gath[r].release();
    }
    g->host_gath.release();
    for (ivfadc_t *h : g->dev) ivfadc_destroy(h);
    delete g;
}

int ivfadc_mg_num_devices(ivfadc_mg_t *g) { return g ? (int)g->dev.size() : 0; }

int ivfadc_mg_set_gather(ivfadc_mg_t *g, int mode)
try {
    std::unique_lock<std::recursive_mutex> lkg_;
    if (g) lkg_ = std::unique_lock<std::recursive_mutex>(g->mu);
    if (!g) return fail(IVFADC_ERR_INVALID, "null handle");
    if (mode != 0 && mode != 1) return fail(IVFADC_ERR_INVALID, "mode must be 0 (host gather) or 1 (RCCL all-gather)");
    if (mode == 1 && g->comms.empty()) {
        RcclApi &api = rccl_api();
        if (!api.ok) return fail(IVFADC_ERR_STATE, "librccl.so could not be loaded");
        std::vector<int> devs;
        for (ivfadc_t *h : g->dev) devs.push_back(h->device);
        std::vector<int> srt(devs);
        std::sort(srt.begin(), srt.end());
        if (std::adjacent_find(srt.begin(), srt.end()) != srt.end())
            return fail(IVFADC_ERR_INVALID, "RCCL needs distinct devices (a device is listed twice)");
        std::vector<ncclComm_t> comms(devs.size(), nullptr);
        NCCL_TRY(api.CommInitAll(comms.data(), (int)devs.size(), devs.data()));
        g->comms.swap(comms);
    }
    g->gather_mode = mode;
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_mg_collectives(ivfadc_mg_t *g, int64_t *out)
{
    if (!g || !out) return fail(IVFADC_ERR_INVALID, "null argument");
    *out = g->collectives;
    return IVFADC_OK;
}

int ivfadc_mg_set_lists(ivfadc_mg_t *g, const int64_t *offsets, const uint8_t *codes, const uint32_t *ids)
try {
    std::unique_lock<std::recursive_mutex> lkg_;
    if (g) lkg_ = std::unique_lock<std::recursive_mutex>(g->mu);
    if (!g) return fail(IVFADC_ERR_INVALID, "null handle");
    return mg_parallel(g, [&](size_t r) { return ivfadc_set_lists(g->dev[r], offsets, codes, ids); });
} IVF_CATCH

int ivfadc_mg_synth_lists(ivfadc_mg_t *g, const int64_t *offsets, uint64_t seed)
try {
    std::unique_lock<std::recursive_mutex> lkg_;
    if (g) lkg_ = std::unique_lock<std::recursive_mutex>(g->mu);
    if (!g) return fail(IVFADC_ERR_INVALID, "null handle");
    return mg_parallel(g, [&](size_t r) { return ivfadc_synth_lists(g->dev[r], offsets, seed); });
} IVF_CATCH

// Mutators: a replica that fails does not stop the others -- every replica's host mirror takes the same edit (the
// mirrors never diverge) and a replica whose device-side update failed is left marked stale (re-laid out by its next
// search).  The first failure is reported.
int ivfadc_mg_append(ivfadc_mg_t *g, int64_t nnew, const float *pts, const uint32_t *ids, int32_t *out_list, uint8_t *out_codes)
try {
    std::unique_lock<std::recursive_mutex> lkg_;
    if (g) lkg_ = std::unique_lock<std::recursive_mutex>(g->mu);
    if (!g || g->dev.empty()) return fail(IVFADC_ERR_INVALID, "null handle");
    for (ivfadc_t *h : g->dev) TRY(append_check(h, nnew, pts, ids));   // nothing has changed yet if a check fails
    if (nnew == 0) return IVFADC_OK;
    ivfadc_t *h0 = g->dev[0];
    TRY(set_device(h0));
    std::vector<int32_t> lst((size_t)nnew);
    std::vector<uint8_t> cod((size_t)nnew * h0->m);
    TRY(encode_dev(h0, nnew, pts, lst.data(), cod.data()));             // one encode: every replica appends the same codes
    int first = IVFADC_OK;
    std::string first_msg;
    for (ivfadc_t *h : g->dev) {
        const int rc = append_encoded(h, nnew, lst.data(), cod.data(), ids);
        if (rc != IVFADC_OK && first == IVFADC_OK) { first = rc; first_msg = g_err; }
    }
    if (out_list) memcpy(out_list, lst.data(), (size_t)nnew * 4);
    if (out_codes) memcpy(out_codes, cod.data(), (size_t)nnew * h0->m);
    if (first != IVFADC_OK) g_err = first_msg;
    return first;
} IVF_CATCH

int ivfadc_mg_delete_ids(ivfadc_mg_t *g, int64_t ndel, const uint32_t *ids, int64_t *out_removed)
try {
    std::unique_lock<std::recursive_mutex> lkg_;
    if (g) lkg_ = std::unique_lock<std::recursive_mutex>(g->mu);
    if (!g) return fail(IVFADC_ERR_INVALID, "null handle");
    int first = IVFADC_OK;
    std::string first_msg;
    for (size_t r = 0; r < g->dev.size(); ++r) {
        const int rc = ivfadc_delete_ids(g->dev[r], ndel, ids, r == 0 ? out_removed : nullptr);
        if (rc != IVFADC_OK && first == IVFADC_OK) { first = rc; first_msg = g_err; }
    }
    if (first != IVFADC_OK) g_err = first_msg;
    return first;
} IVF_CATCH

int ivfadc_mg_shift_ids(ivfadc_mg_t *g, int32_t delta)
try {
    std::unique_lock<std::recursive_mutex> lkg_;
    if (g) lkg_ = std::unique_lock<std::recursive_mutex>(g->mu);
    if (!g) return fail(IVFADC_ERR_INVALID, "null handle");
    int first = IVFADC_OK;
    std::string first_msg;
    for (ivfadc_t *h : g->dev) {
        const int rc = ivfadc_shift_ids(h, delta);
        if (rc != IVFADC_OK && first == IVFADC_OK) { first = rc; first_msg = g_err; }
    }
    if (first != IVFADC_OK) g_err = first_msg;
    return first;
} IVF_CATCH

// device r's block of a batch of nq queries split over G devices: contiguous, sizes differ by at most one
static inline int64_t mg_lo(int64_t r, int64_t nq, int64_t G) { return r * (nq / G) + std::min<int64_t>(r, nq % G); }

int ivfadc_mg_search(ivfadc_mg_t *g, int64_t nq, const float *queries, int K, int w, uint32_t *out_ids, float *out_dists,
                     int32_t *out_counts)
try {
    std::unique_lock<std::recursive_mutex> lkg_;
    if (g) lkg_ = std::unique_lock<std::recursive_mutex>(g->mu);
    if (!g || g->dev.empty()) return fail(IVFADC_ERR_INVALID, "null handle");
    int wc = w;
    TRY(check_search_args(g->dev[0], nq, K, wc));
    if (nq == 0) return IVFADC_OK;
    if (!queries || !out_ids || !out_dists || !out_counts) return fail(IVFADC_ERR_INVALID, "null buffer");
    const int64_t G = (int64_t)g->dev.size();
    const int d = g->dev[0]->d;
    if (g->gather_mode == 0) {
        // enqueue every device's block, then collect: the devices run concurrently
        for (int64_t r = 0; r < G; ++r) {
            const int64_t a = mg_lo(r, nq, G), b = mg_lo(r + 1, nq, G);
            if (b <= a) continue;
            const int rc = search_enqueue(g->dev[r], b - a, queries + (size_t)a * d, K, w, out_ids + (size_t)a * K, out_dists + (size_t)a * K, out_counts + a);
            if (rc != IVFADC_OK) {
                // the devices before r (and r itself, up to the failure) are writing the caller's arrays or the staging blocks: they end first
                const std::string msg = g_err;
                for (int64_t q = 0; q <= r; ++q)
                    if (hipSetDevice(g->dev[q]->device) == hipSuccess) (void)hipStreamSynchronize(g->dev[q]->stream);
                g_err = msg;
                return rc;
            }
        }
        for (int64_t r = 0; r < G; ++r) {
            const int64_t a = mg_lo(r, nq, G), b = mg_lo(r + 1, nq, G);
            if (b > a) TRY(search_finish(g->dev[r], b - a, K, out_ids + (size_t)a * K, out_dists + (size_t)a * K, out_counts + a));
        }
        return IVFADC_OK;
    }
    // RCCL: every device leaves its packed block [ids nql*K | dists nql*K | counts nql] (nql = ceil(nq / G): equal
    // blocks, the collective's contract) in device memory; ONE all-gather per batch; the host reads device 0's copy
    RcclApi &api = rccl_api();
    const int64_t nql = (nq + G - 1) / G;
    const size_t blk_words = (size_t)nql * (2 * (size_t)K + 1);
    // a failure on device r must not leave devices 0 .. r-1 with work in flight that nothing waits for, nor an RCCL group open (the
    // next collective of the process would be absorbed into it): remember the first failure, finish the bookkeeping, then report
    int first_rc = IVFADC_OK;
    std::string first_msg;
    auto note = [&](int rc) {
        if (rc != IVFADC_OK && first_rc == IVFADC_OK) { first_rc = rc; first_msg = g_err; }
    };
    int64_t enq = 0;
    for (int64_t r = 0; r < G && first_rc == IVFADC_OK; ++r) {
        ivfadc_t *h = g->dev[r];
        const int64_t a = mg_lo(r, nq, G), b = mg_lo(r + 1, nq, G);
        auto one = [&]() -> int {
            TRY(set_device(h));
            TRY(h->out_ids.ensure(blk_words * 4));
            TRY(g->gath[r].ensure(blk_words * 4 * (size_t)G));
            if (b > a) {
                const size_t qbytes = (size_t)(b - a) * d * 4;
                TRY(h->q_stage.ensure(qbytes));
                const float *src = queries + (size_t)a * d;
                if (!host_known(src, qbytes)) {
                    TRY(h->pin_in.ensure(qbytes));
                    memcpy(h->pin_in.p, src, qbytes);
                    src = (const float *)h->pin_in.p;
                }
                TRY(ingest_rows(h, h->stream, src, h->q_stage.p, qbytes));
                uint32_t *o = h->out_ids.as<uint32_t>();
                TRY(search_dev(h, b - a, h->q_stage.as<float>(), K, w, o, (float *)(o + (size_t)nql * K), (int32_t *)(o + 2 * (size_t)nql * K)));
            }
            return IVFADC_OK;
        };
        note(one());
        enq = r + 1;
    }
    if (first_rc != IVFADC_OK) {
        for (int64_t r = 0; r < enq; ++r)
            if (hipSetDevice(g->dev[r]->device) == hipSuccess) (void)hipStreamSynchronize(g->dev[r]->stream);
        g_err = first_msg;
        return first_rc;
    }
    NCCL_TRY(api.GroupStart());
    for (int64_t r = 0; r < G; ++r) {
        ivfadc_t *h = g->dev[r];
        auto one = [&]() -> int {
            TRY(set_device(h));
            NCCL_TRY(api.AllGather(h->out_ids.p, g->gath[r].p, blk_words, ncclInt32, g->comms[r], h->stream));
            return IVFADC_OK;
        };
        note(one());
    }
    {
        const ncclResult_t ge = api.GroupEnd();   // always closed, whatever happened inside
        if (ge != ncclSuccess && first_rc == IVFADC_OK) note(fail(IVFADC_ERR_HIP, "ncclGroupEnd failed: %s", api.GetErrorString(ge)));
    }
    if (first_rc != IVFADC_OK) {
        for (int64_t r = 0; r < G; ++r)
            if (hipSetDevice(g->dev[r]->device) == hipSuccess) (void)hipStreamSynchronize(g->dev[r]->stream);
        g_err = first_msg;
        return first_rc;
    }
    g->collectives++;
    ivfadc_t *h0 = g->dev[0];
    TRY(set_device(h0));
    TRY(g->host_gath.ensure(blk_words * 4 * (size_t)G));
    HIP_TRY(hipMemcpyAsync(g->host_gath.p, g->gath[0].p, blk_words * 4 * (size_t)G, hipMemcpyDeviceToHost, h0->stream));
    for (int64_t r = 0; r < G; ++r) {   // every device's stream: the collective has finished everywhere before we return
        TRY(set_device(g->dev[r]));
        TRY(wait_stream(g->dev[r]));
    }
    const uint32_t *hg = (const uint32_t *)g->host_gath.p;
    for (int64_t r = 0; r < G; ++r) {
        const int64_t a = mg_lo(r, nq, G), b = mg_lo(r + 1, nq, G);
        const uint32_t *blk = hg + (size_t)r * blk_words;
        memcpy(out_ids + (size_t)a * K, blk, (size_t)(b - a) * K * 4);
        memcpy(out_dists + (size_t)a * K, blk + (size_t)nql * K, (size_t)(b - a) * K * 4);
        memcpy(out_counts + a, blk + 2 * (size_t)nql * K, (size_t)(b - a) * 4);
    }
    return IVFADC_OK;
} IVF_CATCH

// ---- one process per GPU: the final top-k merge of a batch inside the library ------------------------------------------------
// (query batches partitioned across the GPUs of a node, index replicated, RCCL over xGMI only for the final
// merge.)  Every rank owns one handle on its GPU and a contiguous block of the batch's queries.  ivfadc_comm_init joins the
// ranks in an RCCL communicator (ncclCommInitRank; the 128-byte id comes from rank 0's ivfadc_comm_unique_id and travels
// by whatever the host framework uses to talk: torch.distributed's store, MPI, a file);
// ivfadc_search_device_allgather then searches the rank's block and issues ONE ncclAllGather of the packed result block on a
// side stream of the handle -- a few microseconds of host time per batch where a framework-level collective costs tens -- so
// the collective of batch i overlaps the kernels of batch i + 1.  `slot` names one of COMM_SLOTS result buffers in
// flight: a slot's previous collective is waited for (on the device) before the slot is reused.
int ivfadc_comm_unique_id(uint8_t *out_id128)
try {
    if (!out_id128) return fail(IVFADC_ERR_INVALID, "null argument");
    RcclApi &api = rccl_api();
    if (!api.ok) return fail(IVFADC_ERR_STATE, "librccl.so could not be loaded");
    ncclUniqueId id;
    NCCL_TRY(api.GetUniqueId(&id));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(out_id128, &id, sizeof(id));
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_comm_destroy(ivfadc_t *h)
try {
    HandleLock lk_(h);
    if (!h) return fail(IVFADC_ERR_INVALID, "null handle");
    TRY(set_device(h));
    if (h->comm_stream) (void)hipStreamSynchronize(h->comm_stream);
    if (h->comm) { (void)rccl_api().CommDestroy((ncclComm_t)h->comm); h->comm = nullptr; }
    for (int i = 0; i < ivfadc_index::COMM_SLOTS; ++i)
        if (h->comm_done[i]) { (void)hipEventDestroy(h->comm_done[i]); h->comm_done[i] = nullptr; h->comm_busy[i] = false; }
    for (int i = 0; i < ivfadc_index::COMM_SLOTS; ++i) h->comm_slot_seq[i] = 0;   // (comm_seq stays: views compare their comm_waited with it)
    if (h->comm_ready) { (void)hipEventDestroy(h->comm_ready); h->comm_ready = nullptr; }
    if (h->comm_stream) { (void)hipStreamDestroy(h->comm_stream); h->comm_stream = nullptr; }
    h->comm_ranks = 0;
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_comm_init(ivfadc_t *h, int nranks, int rank, const uint8_t *id128)
try {
    HandleLock lk_(h);
    if (!h || !id128) return fail(IVFADC_ERR_INVALID, "null argument");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(IVFADC_ERR_INVALID, "rank %d of %d", rank, nranks);
    RcclApi &api = rccl_api();
    if (!api.ok) return fail(IVFADC_ERR_STATE, "librccl.so could not be loaded");
    if (h->comm) TRY(ivfadc_comm_destroy(h));
    TRY(set_device(h));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    NCCL_TRY(api.CommInitRank(&comm, nranks, id, rank));
    h->comm = (void *)comm;
    h->comm_ranks = nranks;
    h->comm_rank = rank;
    auto rest = [&]() -> int {
        HIP_TRY(hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
        // search -> collective: the event orders the search's writes to the send block before the all-gather, and a PEER GPU may read that
        // block directly over xGMI (RCCL's P2P-read and registered-buffer paths), so the record carries its system-scope release by
        // default.  The relaxed form (no system fence: 2-3 us less per step, measured with a single-rank communicator only) is opt-in,
        // IVFADC_EVENT_NO_SYSFENCE=1, until a run with >= 2 ranks has passed bench.py's gather_check with it.
        static const bool no_sysfence = env_knob("IVFADC_EVENT_NO_SYSFENCE") != nullptr;
        HIP_TRY(hipEventCreateWithFlags(&h->comm_ready, hipEventDisableTiming | (no_sysfence ? (unsigned)hipEventDisableSystemFence : 0u)));
        for (int i = 0; i < ivfadc_index::COMM_SLOTS; ++i) HIP_TRY(hipEventCreateWithFlags(&h->comm_done[i], hipEventDisableTiming));
        return IVFADC_OK;
    };
    const int rc = rest();
    if (rc != IVFADC_OK) {   // no half-initialised communicator is left behind
        const std::string msg = g_err;
        (void)ivfadc_comm_destroy(h);
        g_err = msg;
    }
    return rc;
} IVF_CATCH

int ivfadc_search_device_allgather(ivfadc_t *h, int64_t nq, const float *d_queries, int K, int w, int32_t *d_block, int32_t *d_gathered,
                                   int slot)
{
    return ivfadc_search_device_allgather_on(h, h, nq, d_queries, K, w, d_block, d_gathered, slot);
}

// The same with the SEARCH on `searcher` -- h itself or a view of it (two batches in flight per rank) -- and the collective where the
// communicator lives: on h's side stream, in call order, so every rank issues its all-gathers in the same order whatever the lanes do.
int ivfadc_search_device_allgather_on(ivfadc_t *h, ivfadc_t *searcher, int64_t nq, const float *d_queries, int K, int w, int32_t *d_block,
                                      int32_t *d_gathered, int slot)
try {
    HandleLock lk_(h);
    HandleLock lk2_(searcher != h ? searcher : nullptr);
    if (!h || !searcher) return fail(IVFADC_ERR_INVALID, "null handle");
    if (searcher != h && searcher->view_of != h) return fail(IVFADC_ERR_INVALID, "searcher must be the handle itself or a view of it");
    TRY(check_search_args(searcher, nq, K, w));
    if (!h->comm) return fail(IVFADC_ERR_STATE, "ivfadc_comm_init has not been called");
    if (slot < 0 || slot >= ivfadc_index::COMM_SLOTS) return fail(IVFADC_ERR_INVALID, "slot must be in [0, %d)", ivfadc_index::COMM_SLOTS);
    if (nq < 1 || !d_queries || !d_block || !d_gathered) return fail(IVFADC_ERR_INVALID, "null buffer / empty block");
    // ncclAllGather moves EQUAL blocks: every rank must pass the same nq and K within one call (a different count on one rank hangs the
    // collective or overruns d_gathered); nothing a single rank sees can detect a mismatch, so it is the caller's contract.  The
    // block size may change from call to call (a ragged final batch, another K) as long as every rank changes it alike, and
    // d_gathered must hold nranks x nq x (2K + 1) words.
    TRY(set_device(h));
    // The slot's buffers were read by its previous collective: that must have finished before the search overwrites them.  A wait in the
    // search stream costs the NEXT kernel ~6 us of idle queue, fired event or not (measured), so one wait serves several steps: it names
    // the newest collective that is at least COMM_SLOTS / 2 behind the head (never an older one than needed) -- collectives finish in
    // order, so every earlier one is covered -- and with the slots in rotation the next few searches find theirs already waited for.
    {
        const int64_t need = h->comm_slot_seq[slot];
        if (need > searcher->comm_waited) {
            const int64_t target = std::max<int64_t>(need, h->comm_seq - ivfadc_index::COMM_SLOTS / 2);
            int ts = slot;
            for (int i = 0; i < ivfadc_index::COMM_SLOTS; ++i)
                if (h->comm_slot_seq[i] == target) ts = i;
            HIP_TRY(hipStreamWaitEvent(searcher->stream, h->comm_done[ts], 0));
            searcher->comm_waited = h->comm_slot_seq[ts];
        }
    }
    uint32_t *ids = (uint32_t *)d_block;
    TRY(search_dev(searcher, nq, d_queries, K, w, ids, (float *)(ids + (size_t)nq * K), (int32_t *)(ids + 2 * (size_t)nq * K)));
    HIP_TRY(hipEventRecord(h->comm_ready, searcher->stream));
    HIP_TRY(hipStreamWaitEvent(h->comm_stream, h->comm_ready, 0));
    RcclApi &api = rccl_api();
    NCCL_TRY(api.AllGather(d_block, d_gathered, (size_t)nq * (2 * (size_t)K + 1), ncclInt32, (ncclComm_t)h->comm, h->comm_stream));
    HIP_TRY(hipEventRecord(h->comm_done[slot], h->comm_stream));
    h->comm_busy[slot] = true;
    h->comm_slot_seq[slot] = ++h->comm_seq;
    h->comm_collectives++;
    return IVFADC_OK;
} IVF_CATCH

// List-partitioned step in ONE call (one process per GPU): partial search of ALL nq queries over this rank's lists into d_block
// ([keys nq x K u64 | counts nq i32], ivfadc_listpart_block_words(nq, K) int32 words), ONE ncclAllGather of the blocks into d_gathered
// (nranks blocks, rank order), then the K-way merge of the ranks' keys on this rank: every rank ends with the batch's full results.
// ivfadc_set_list_partition(h, nranks, rank) and ivfadc_comm_init first.
int64_t ivfadc_listpart_block_words(int64_t nq, int K) { return ((nq * (2 * (int64_t)K + 1)) + 1) & ~(int64_t)1; }

int ivfadc_search_device_listpart(ivfadc_t *h, int64_t nq, const float *d_queries, int K, int w, int32_t *d_block, int32_t *d_gathered,
                                  uint32_t *d_ids, float *d_dists, int32_t *d_counts)
try {
    HandleLock lk_(h);
    TRY(check_search_args(h, nq, K, w));
    if (!h->comm) return fail(IVFADC_ERR_STATE, "ivfadc_comm_init has not been called");
    if (h->part_n != h->comm_ranks || h->part_i != h->comm_rank)
        return fail(IVFADC_ERR_STATE, "ivfadc_set_list_partition(h, %d, %d) must name this rank of the communicator", h->comm_ranks, h->comm_rank);
    if (nq < 1 || !d_queries || !d_block || !d_gathered || !d_ids || !d_dists || !d_counts) return fail(IVFADC_ERR_INVALID, "null buffer / empty batch");
    const size_t words = (size_t)ivfadc_listpart_block_words(nq, K);
    uint64_t *keys = (uint64_t *)d_block;
    int32_t *cnts = d_block + (size_t)nq * K * 2;
    if (h->comm_ranks == 1) {
        // one rank owns every list: an ordinary search (ids straight away), no collective
        return ivfadc_search_device(h, nq, d_queries, K, w, d_ids, d_dists, d_counts);
    }
    TRY(ivfadc_search_device_partial(h, nq, d_queries, K, w, keys, cnts));
    HIP_TRY(hipEventRecord(h->comm_ready, h->stream));
    HIP_TRY(hipStreamWaitEvent(h->comm_stream, h->comm_ready, 0));
    RcclApi &api = rccl_api();
    NCCL_TRY(api.AllGather(d_block, d_gathered, words, ncclInt32, (ncclComm_t)h->comm, h->comm_stream));
    HIP_TRY(hipEventRecord(h->comm_done[0], h->comm_stream));
    HIP_TRY(hipStreamWaitEvent(h->stream, h->comm_done[0], 0));
    h->comm_collectives++;
    return merge_partials_dev(h, nq, K, h->comm_ranks, (const uint64_t *)d_gathered, words / 2, d_gathered + (size_t)nq * K * 2, words, d_ids, d_dists,
                              d_counts);
} IVF_CATCH

// makes the handle's search stream wait (on the device) for every collective issued so far; returns how many were issued
int ivfadc_comm_wait(ivfadc_t *h, int64_t *out_collectives)
try {
    HandleLock lk_(h);
    if (!h) return fail(IVFADC_ERR_INVALID, "null handle");
    TRY(set_device(h));
    // (the newest one covers them all: they finish in issue order)
    if (h->comm_seq > h->comm_waited) {
        for (int i = 0; i < ivfadc_index::COMM_SLOTS; ++i)
            if (h->comm_slot_seq[i] == h->comm_seq) HIP_TRY(hipStreamWaitEvent(h->stream, h->comm_done[i], 0));
        h->comm_waited = h->comm_seq;
    }
    for (int i = 0; i < ivfadc_index::COMM_SLOTS; ++i) h->comm_busy[i] = false;
    if (out_collectives) *out_collectives = h->comm_collectives;
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_sync(ivfadc_t *h)
try {
    HandleLock lk_(h);
    if (!h) return fail(IVFADC_ERR_INVALID, "null handle");
    TRY(set_device(h));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_set_stream(ivfadc_t *h, void *stream)
try {
    HandleLock lk_(h);
    if (!h) return fail(IVFADC_ERR_INVALID, "null handle");
    TRY(set_device(h));
    TRY(ev_fold(h));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    h->stream = (hipStream_t)stream;
    h->own_stream = false;
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_ntotal(ivfadc_t *h, int64_t *out_n, int64_t *list_sizes)
try {
    HandleLock lk_(h);
    if (!h) return fail(IVFADC_ERR_INVALID, "null handle");
    if (out_n) *out_n = h->ntotal();
    if (list_sizes)
        for (int l = 0; l < h->kc; ++l) list_sizes[l] = h->h_len[l];
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_get_lists(ivfadc_t *h, int64_t *offsets, uint8_t *codes, uint32_t *ids)
try {
    HandleLock lk_(h);
    if (!h) return fail(IVFADC_ERR_INVALID, "null handle");
    if (h->is_view) return fail(IVFADC_ERR_STATE, "a view keeps no host mirror of the lists");
    if (h->synthetic) return fail(IVFADC_ERR_STATE, "device-synthesised lists keep no host mirror");
    int64_t run = 0;
    for (int l = 0; l < h->kc; ++l) {
        const int64_t len = h->h_len[l];
        if (offsets) offsets[l] = run;
        if (codes && len) memcpy(codes + (size_t)run * h->m, h->hl_codes[l].data(), (size_t)len * h->m);
        if (ids && len) memcpy(ids + run, h->hl_ids[l].data(), (size_t)len * 4);
        run += len;
    }
    if (offsets) offsets[h->kc] = run;
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_get_dims(ivfadc_t *h, int *d, int *kc, int *m, int *ksub)
try {
    HandleLock lk_(h);
    if (!h) return fail(IVFADC_ERR_INVALID, "null handle");
    if (d) *d = h->d;
    if (kc) *kc = h->kc;
    if (m) *m = h->m;
    if (ksub) *ksub = h->ksub;
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_get_quantizers(ivfadc_t *h, float *centroids, float *codebooks, uint8_t *code_labels)
try {
    HandleLock lk_(h);
    if (!h) return fail(IVFADC_ERR_INVALID, "null handle");
    TRY(set_device(h));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (centroids) TRY(d2h_copy(centroids, h->centroids.p, (size_t)h->d * h->kc * 4, h->stream));
    if (codebooks) TRY(d2h_copy(codebooks, h->codebooks.p, (size_t)h->d * h->ksub * 4, h->stream));
    if (code_labels) HIP_TRY(hipMemcpy(code_labels, h->labels.p, (size_t)h->m * h->ksub, hipMemcpyDeviceToHost));
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_abi_version(void) { return IVFADC_ABI_VERSION; }

// ---- page-locked host memory the kernels address directly (see "host-pointer entries") -------------------------------------------
static int host_range_add(void *p, size_t bytes, bool owned)
{
    const uintptr_t a = (uintptr_t)p, b = a + bytes;
    std::lock_guard<std::mutex> lk(g_host_mu);
    for (const HostRange &r : g_host_ranges)
        if (a < r.hi && b > r.lo) return fail(IVFADC_ERR_INVALID, "host range overlaps one that is already registered");
    g_host_ranges.push_back({a, b, owned});
    return IVFADC_OK;
}

// removes the range that STARTS at p; found = 0 when there is none of the wanted kind
static bool host_range_take(void *p, bool owned)
{
    std::lock_guard<std::mutex> lk(g_host_mu);
    for (size_t i = 0; i < g_host_ranges.size(); ++i)
        if (g_host_ranges[i].lo == (uintptr_t)p && g_host_ranges[i].owned == owned) {
            g_host_ranges.erase(g_host_ranges.begin() + (ptrdiff_t)i);
            return true;
        }
    return false;
}

int ivfadc_host_alloc(size_t bytes, void **out)
try {
    if (!out) return fail(IVFADC_ERR_INVALID, "out is null");
    *out = nullptr;
    if (bytes == 0) return fail(IVFADC_ERR_INVALID, "bytes == 0");
    void *p = nullptr;
    const hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocPortable | hipHostMallocMapped);
    if (e != hipSuccess) return fail(IVFADC_ERR_HIP, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    const int rc = host_range_add(p, bytes, true);
    if (rc != IVFADC_OK) { (void)hipHostFree(p); return rc; }
    *out = p;
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_host_free(void *p)
try {
    if (!p) return IVFADC_OK;
    if (!host_range_take(p, true)) return fail(IVFADC_ERR_INVALID, "not a pointer ivfadc_host_alloc returned");
    HIP_TRY(hipHostFree(p));
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_host_register(void *p, size_t bytes)
try {
    if (!p || bytes == 0) return fail(IVFADC_ERR_INVALID, "null pointer or bytes == 0");
    {
        const uintptr_t a = (uintptr_t)p, b = a + bytes;
        std::lock_guard<std::mutex> lk(g_host_mu);
        for (const HostRange &r : g_host_ranges)
            if (a < r.hi && b > r.lo) return fail(IVFADC_ERR_INVALID, "host range overlaps one that is already registered");
    }
    const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterPortable | hipHostRegisterMapped);
    if (e != hipSuccess) return fail(IVFADC_ERR_HIP, "hipHostRegister(%zu) failed: %s", bytes, hipGetErrorString(e));
    // the kernels use the HOST address: it must be the address the device sees too (it is, wherever the runtime maps host memory at
    // its own virtual address; checked rather than assumed)
    void *dp = nullptr;
    const hipError_t e2 = hipHostGetDevicePointer(&dp, p, 0);
    if (e2 != hipSuccess || dp != p) {
        (void)hipHostUnregister(p);
        (void)hipGetLastError();
        return fail(IVFADC_ERR_HIP, "registered host memory is not addressable by the device at its host address");
    }
    const int rc = host_range_add(p, bytes, false);
    if (rc != IVFADC_OK) { (void)hipHostUnregister(p); return rc; }
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_host_unregister(void *p)
try {
    if (!p) return IVFADC_OK;
    if (!host_range_take(p, false)) return fail(IVFADC_ERR_INVALID, "not a pointer ivfadc_host_register took");
    HIP_TRY(hipHostUnregister(p));
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_get_host_stats(ivfadc_t *h, ivfadc_host_stats *out)
try {
    HandleLock lk_(h);
    if (!h || !out) return fail(IVFADC_ERR_INVALID, "null argument");
    *out = h->hstats;
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_reset_host_stats(ivfadc_t *h)
try {
    HandleLock lk_(h);
    if (!h) return fail(IVFADC_ERR_INVALID, "null handle");
    const int64_t keep = h->hstats.streams_replaced;   // (a fact about the handle's streams, not a per-interval counter)
    h->hstats = ivfadc_host_stats{};
    h->hstats.streams_replaced = keep;
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_set_profiling(ivfadc_t *h, int on)
try {
    HandleLock lk_(h);
    if (!h) return fail(IVFADC_ERR_INVALID, "null handle");
    h->profiling = on != 0;
    h->profiling_level = on;
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_reset_stats(ivfadc_t *h)
try {
    HandleLock lk_(h);
    if (!h) return fail(IVFADC_ERR_INVALID, "null handle");
    TRY(set_device(h));
    TRY(ev_fold(h));
    HIP_TRY(hipStreamSynchronize(h->stream));
    int64_t sp = 0, pp = 0, sv = 0, vis = 0;
    if (h->misc.p) {
        int64_t shards[512];
        HIP_TRY(hipMemcpy(shards, h->misc.p, sizeof(shards), hipMemcpyDeviceToHost));
        for (int i = 0; i < 64; ++i) sp += shards[i * 8];
        for (int i = 0; i < 64; ++i) pp += shards[i * 8 + 1];
        for (int i = 0; i < 64; ++i) sv += shards[i * 8 + 2];
        for (int i = 0; i < 64; ++i) vis += shards[i * 8 + 3];
        HIP_TRY(hipMemcpy(&h->fallback_base, (char *)h->misc.p + 4096 + 64, 8, hipMemcpyDeviceToHost));
    }
    h->scanned_base = sp;
    h->pruned_base = pp;
    h->surv_base = sv;
    h->visited_base = vis;
    const int qg = h->stats.last_qg, ch = h->stats.last_chunk, gr = h->stats.last_scan_grid, lds = h->stats.last_scan_lds;
    const int cm = h->stats.coarse_mfma, ls = h->stats.last_striped, cl = h->stats.coarse_listed, llb = h->stats.last_lb, lnf = h->stats.last_nf;
    const int ltl = h->stats.last_twolevel, lf16 = h->stats.coarse_f16;
    h->stats = ivfadc_stats{};
    h->stats.last_twolevel = ltl;
    h->stats.coarse_f16 = lf16;
    h->stats.last_lb = llb;
    h->stats.last_nf = lnf;
    h->stats.coarse_mfma = cm;
    h->stats.last_striped = ls;
    h->stats.coarse_listed = cl;
    h->stats.last_qg = qg; h->stats.last_chunk = ch; h->stats.last_scan_grid = gr; h->stats.last_scan_lds = lds;
    h->carry_queries = h->carry_scanned = h->carry_pruned = h->carry_surv = h->carry_fallbacks = h->carry_launches = 0;
    if (h->pipe_view) TRY(ivfadc_reset_stats(h->pipe_view));
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_get_stats(ivfadc_t *h, ivfadc_stats *out)
try {
    HandleLock lk_(h);
    if (!h || !out) return fail(IVFADC_ERR_INVALID, "null argument");
    TRY(set_device(h));
    TRY(ev_fold(h));
    HIP_TRY(hipStreamSynchronize(h->stream));
    int64_t sp = 0, pp = 0, sv = 0, vis = 0;
    if (h->misc.p) {
        int64_t shards[512];
        HIP_TRY(hipMemcpy(shards, h->misc.p, sizeof(shards), hipMemcpyDeviceToHost));
        for (int i = 0; i < 64; ++i) sp += shards[i * 8];
        for (int i = 0; i < 64; ++i) pp += shards[i * 8 + 1];
        for (int i = 0; i < 64; ++i) sv += shards[i * 8 + 2];
        for (int i = 0; i < 64; ++i) vis += shards[i * 8 + 3];
        int64_t fb = 0;
        HIP_TRY(hipMemcpy(&fb, (char *)h->misc.p + 4096 + 64, 8, hipMemcpyDeviceToHost));
        h->stats.coarse_fallbacks = fb - h->fallback_base;
    }
    h->stats.scanned_points = sp - h->scanned_base;
    h->stats.pruned_points = pp - h->pruned_base;
    h->stats.lb_survivors = sv - h->surv_base;
    h->stats.coarse_visited = vis - h->visited_base;
    h->stats.twolevel_groups = h->tl_use ? h->tl_G : 0;
    h->stats.twolevel_probe_fraction = h->tl_probe_fraction;
    h->stats.inplace_appends = (int32_t)std::min<int64_t>(h->inplace_appends, 0x7fffffff);
    *out = h->stats;
    // (second lanes of ivfadc_search_batches that a push! or delete made stale and that are gone: their share stays in the totals)
    out->queries += h->carry_queries;
    out->scanned_points += h->carry_scanned;
    out->pruned_points += h->carry_pruned;
    out->lb_survivors += h->carry_surv;
    out->coarse_fallbacks += h->carry_fallbacks;
    out->scan_launches += h->carry_launches;
    if (h->pipe_view) {
        // the odd batches of ivfadc_search_batches ran on the internal view: its counters belong to this handle's totals
        ivfadc_stats v;
        TRY(ivfadc_get_stats(h->pipe_view, &v));
        out->queries += v.queries;
        out->scanned_points += v.scanned_points;
        out->pruned_points += v.pruned_points;
        out->lb_survivors += v.lb_survivors;
        out->coarse_fallbacks += v.coarse_fallbacks;
        out->scan_launches += v.scan_launches;
        out->coarse_visited += v.coarse_visited;
    }
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_train(int device, int d, int64_t n, const float *data, int kc, int k, int m, int coarse_maxiter, int quant_maxiter,
                 uint64_t seed, float *out_centroids, float *out_codebooks)
try {
    return train_impl(device, d, n, data, kc, k, m, coarse_maxiter, quant_maxiter, seed, out_centroids, out_codebooks);
} IVF_CATCH

int ivfadc_set_coarse_mode(ivfadc_t *h, int mode)
try {
    HandleLock lk_(h);
    if (!h) return fail(IVFADC_ERR_INVALID, "null handle");
    if (mode < 0 || mode > 8) return fail(IVFADC_ERR_INVALID, "mode must be 0 ... 8");
    h->allow_f16 = mode != 8 && mode != 3 && env_knob("IVFADC_COARSE_BF16") == nullptr;   // 8: as 0 with the three-product bf16 split (A/B, tests)
    // 6: the certified two-level search whatever the self-probe says (built on the next search); 7: never; anything else: automatic
    h->tl_mode = mode == 6 ? 1 : (mode == 7 ? -1 : 0);
    if (h->tl_tried) h->tl_use = h->tl_G > 0 && (h->tl_mode > 0 || (h->tl_mode == 0 && h->tl_probe_fraction >= 0.f && h->tl_probe_fraction <= 0.02f));
    if (h->tl_mode > 0 && h->tl_G == 0) h->tl_tried = false;   // (automatic mode did not build it for a small quantizer: build on request)
    h->sq_inside = mode == 5;
    h->allow_mfma = (mode != 1) && env_knob("IVFADC_COARSE_EXACT") == nullptr;
    h->mfma_min_kc = (mode == 2) ? 128 : 2048;
    h->allow_bf16 = mode != 3 && h->cent_hi.p != nullptr && env_knob("IVFADC_COARSE_F32") == nullptr;
    h->allow_listed = mode != 4 && env_knob("IVFADC_NO_LISTED") == nullptr;
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_set_pruning(ivfadc_t *h, int on)
try {
    HandleLock lk_(h);
    if (!h) return fail(IVFADC_ERR_INVALID, "null handle");
    h->allow_prune = on != 0 && env_knob("IVFADC_NO_PRUNE") == nullptr;
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_set_next_queries(ivfadc_t *h, int64_t nq, const float *d_queries, uint64_t token)
try {
    HandleLock lk_(h);
    if (!h) return fail(IVFADC_ERR_INVALID, "null handle");
    if (nq < 0) return fail(IVFADC_ERR_INVALID, "nq < 0");
    h->hint_q = (nq > 0 && token != 0) ? d_queries : nullptr;
    h->hint_nq = h->hint_q ? nq : 0;
    h->hint_token = h->hint_q ? token : 0;
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_set_query_token(ivfadc_t *h, uint64_t token)
try {
    HandleLock lk_(h);
    if (!h) return fail(IVFADC_ERR_INVALID, "null handle");
    h->cur_token = token;
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_set_table_mode(ivfadc_t *h, int mode)
try {
    HandleLock lk_(h);
    if (!h) return fail(IVFADC_ERR_INVALID, "null handle");
    if (mode < 0 || mode > 7) return fail(IVFADC_ERR_INVALID, "mode must be 0 ... 7");
    h->allow_filt = mode != 1 && env_knob("IVFADC_EXACT_TABLES") == nullptr;
    h->force_lb = mode == 2 || mode == 4;
    // 3 / 4: as 0 / 2 with the matrix-core tables built from the three-product bf16 split instead of one f16 product (A/B runs, tests)
    h->lb_use_f16 = mode != 3 && mode != 4 && env_knob("IVFADC_LB_BF16") == nullptr;
    // 5 / 6: as 0 with the eight-wave list-major kernel (wg8scan.hip.h) never / wherever it is instantiated (A/B runs, tests);
    // 7: as 6 with its eight-query form (wg8q8scan.hip.h) wherever that is instantiated
    h->wg8_mode = mode == 5 ? -1 : (mode == 6 ? 1 : (mode == 7 ? 2 : 0));
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_debug_lb_table(ivfadc_t *h, const float *query, int cell, uint8_t *out_table, float *out_consts)
try {
    HandleLock lk_(h);
    if (!h || !query || !out_table || !out_consts) return fail(IVFADC_ERR_INVALID, "null argument");
    if (cell < 0 || cell >= h->kc) return fail(IVFADC_ERR_INVALID, "cell out of range");
    if (!h->lb_split.p || !lb_shape(h->m, h->dsub)) return fail(IVFADC_ERR_STATE, "no matrix-core table kernels for m=%d dsub=%d", h->m, h->dsub);
    TRY(set_device(h));
    const int m = h->m;
    DevBuf dq, dt, df;
    int rc = dq.ensure((size_t)h->d * 4);
    if (rc == IVFADC_OK) rc = dt.ensure((size_t)m * 256);
    if (rc == IVFADC_OK) rc = df.ensure((size_t)(3 + 2 * m) * 4);
    if (rc == IVFADC_OK) {
        hipError_t e = hipMemcpyAsync(dq.p, query, (size_t)h->d * 4, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) {
            QScanArgs a;
            a.lb.cb_split = h->lb_split.as<uint4>();
            a.lb.cb_n2 = h->lb_n2.as<float>();
            a.lb.cb_lab = h->lb_lab.as<float>();
            a.lb.cb_maxn = h->lb_maxn.as<float>();
        a.lb.cb_f16 = (h->lb_use_f16 && h->lb_f16.p) ? h->lb_f16.as<uint4>() : (const uint4 *)nullptr;
        a.lb.cb_isc = h->lb_isc.as<float>();
        a.lb.mu = a.lb.cb_f16 ? 1.48e-3f : 9.2e-5f;
            const size_t lds = lb_lds_bytes(m, h->dsub, 1);
            void (*dk)(const IndexView, const LbView, const float *, int, unsigned char *, float *) =
                (m == 48) ? lb_debug_kernel<48, 16> : lb_debug_kernel<16, 6>;
            rc = fn_raise_lds(h->device, (const void *)dk, lds, false);
            if (rc == IVFADC_OK) {
                hipLaunchKernelGGL(dk, dim3(1), dim3(256), lds, h->stream, index_view(h), a.lb, dq.as<float>(), cell, dt.as<unsigned char>(),
                                   df.as<float>());
                e = hipGetLastError();
            }
        }
        if (e == hipSuccess) e = hipMemcpyAsync(out_table, dt.p, (size_t)m * 256, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(out_consts, df.p, (size_t)(3 + 2 * m) * 4, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess && rc == IVFADC_OK) rc = fail(IVFADC_ERR_HIP, "debug table: %s", hipGetErrorString(e));
    }
    dq.release(); dt.release(); df.release();
    return rc;
} IVF_CATCH

int ivfadc_set_workspace_limit(ivfadc_t *h, uint64_t bytes)
try {
    HandleLock lk_(h);
    if (!h) return fail(IVFADC_ERR_INVALID, "null handle");
    h->ws_budget = (size_t)std::max<uint64_t>(bytes, 1 << 20);
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_set_tuning(ivfadc_t *h, int qg, int chunk_points)
try {
    HandleLock lk_(h);
    if (h) {
        const char *e = env_knob("IVFADC_FORCE_PG");
        h->force_pg = e ? atoi(e) : 0;
    }
    if (!h) return fail(IVFADC_ERR_INVALID, "null handle");
    if (!(qg == 0 || qg == -1 || qg == -2 || qg == -3 || qg == 1 || qg == 2 || qg == 4 || qg == 8))
        return fail(IVFADC_ERR_INVALID, "qg must be 0, -1, -2, -3, 1, 2, 4 or 8");
    if (chunk_points < 0) return fail(IVFADC_ERR_INVALID, "chunk_points < 0");
    h->force_qg = qg;
    h->force_chunk = chunk_points;
    return IVFADC_OK;
} IVF_CATCH

}  // extern "C"

// ---- persistence: the reference's own on-disk format (src/persistency.jl:1-78 writer, :82-134 loader) ---------
// 9 text lines ("<nrows> <nclusters>" / "<n> <m> <k> <dsub>" / coarse quantizer / quantization / U / I / Dc / Dr / T),
// then raw little-endian binary: centroids column by column (:44-49); per codebook its `codes` (labels) and then
// `vectors` row j across all k codewords (:56-61); the nrows x nrows rotation matrix (:62-64, never read by
// knn_search); per list clsize::Int64, idxs, then every vector's m code bytes (:68-78).
namespace {

struct FileCloser {
    FILE *f;
    ~FileCloser() { if (f) fclose(f); }
};

bool read_exact(FILE *f, void *dst, size_t bytes) { return bytes == 0 || fread(dst, 1, bytes, f) == bytes; }

bool read_line(FILE *f, std::string &out)
{
    out.clear();
    int c;
    while ((c = fgetc(f)) != EOF && c != '\n') out.push_back((char)c);
    while (!out.empty() && (out.back() == '\r' || out.back() == ' ')) out.pop_back();
    return c != EOF || !out.empty();
}

std::string last_component(const std::string &s)
{
    const size_t p = s.rfind('.');
    return p == std::string::npos ? s : s.substr(p + 1);
}

}  // namespace

int ivfadc_save_index(ivfadc_t *h, const char *path, int index_bits)
try {
    HandleLock lk_(h);
    if (!h || !path) return fail(IVFADC_ERR_INVALID, "null argument");
    if (h->is_view) return fail(IVFADC_ERR_STATE, "a view keeps no host mirror of the lists");
    if (h->synthetic) return fail(IVFADC_ERR_STATE, "device-synthesised lists keep no host mirror");
    if (index_bits != 8 && index_bits != 16 && index_bits != 32) return fail(IVFADC_ERR_INVALID, "index_bits must be 8, 16 or 32");
    TRY(set_device(h));
    if (index_bits < 32) {   // the reference asserts the capacity of I (index.jl:124-125, utils.jl:134-135): never truncate an id
        const uint32_t lim = index_bits == 8 ? 0xFFu : 0xFFFFu;
        for (int l = 0; l < h->kc; ++l)
            for (uint32_t v : h->hl_ids[l])
                if (v > lim) return fail(IVFADC_ERR_ASSERT, "id %u does not fit a %d-bit index type", v, index_bits);
    }
    const int d = h->d, kc = h->kc, m = h->m, k = h->ksub, dsub = h->dsub;
    std::vector<float> cent((size_t)kc * d), cbs((size_t)d * k);
    std::vector<uint8_t> lab((size_t)m * k);
    HIP_TRY(hipStreamSynchronize(h->stream));
    TRY(d2h_copy(cent.data(), h->centroids.p, cent.size() * 4, h->stream));
    TRY(d2h_copy(cbs.data(), h->codebooks.p, cbs.size() * 4, h->stream));
    HIP_TRY(hipMemcpy(lab.data(), h->labels.p, lab.size(), hipMemcpyDeviceToHost));
    FileCloser fc{fopen(path, "wb")};
    FILE *f = fc.f;
    if (!f) return fail(IVFADC_ERR_INVALID, "cannot open %s for writing", path);
    const char *iname = index_bits == 8 ? "UInt8" : (index_bits == 16 ? "UInt16" : "UInt32");
    fprintf(f, "%d %d\n%lld %d %d %d\nNaiveQuantizer\nQuantizedArrays.OrthogonalQuantization\nUInt8\n%s\n"
               "Distances.SqEuclidean\nDistances.SqEuclidean\nFloat32\n", d, kc, (long long)h->ntotal(), m, k, dsub, iname);
    bool ok = fwrite(cent.data(), 4, cent.size(), f) == cent.size();                  // centroid c = column c
    std::vector<float> row((size_t)k);
    for (int i = 0; i < m && ok; ++i) {
        ok = fwrite(lab.data() + (size_t)i * k, 1, (size_t)k, f) == (size_t)k;
        for (int j = 0; j < dsub && ok; ++j) {                                        // vectors[j, :]
            for (int c = 0; c < k; ++c) row[c] = cbs[((size_t)i * k + c) * dsub + j];
            ok = fwrite(row.data(), 4, (size_t)k, f) == (size_t)k;
        }
    }
    if (!h->rot.empty()) {                                                            // the rotation the index was loaded with (:opq)
        ok = ok && fwrite(h->rot.data(), 4, h->rot.size(), f) == h->rot.size();
    } else {
        std::vector<float> rot((size_t)d, 0.0f);
        for (int i = 0; i < d && ok; ++i) {                                           // identity rotation, column i
            rot[i] = 1.0f;
            ok = fwrite(rot.data(), 4, (size_t)d, f) == (size_t)d;
            rot[i] = 0.0f;
        }
    }
    std::vector<uint8_t> narrow;
    for (int l = 0; l < kc && ok; ++l) {
        const int64_t len = h->h_len[l];
        ok = fwrite(&len, 8, 1, f) == 1;
        const uint32_t *ids = h->hl_ids[l].data();
        if (index_bits == 32) {
            ok = ok && (len == 0 || fwrite(ids, 4, (size_t)len, f) == (size_t)len);
        } else {
            const int bw = index_bits / 8;
            narrow.resize((size_t)len * bw);
            for (int64_t p = 0; p < len; ++p) {
                if (bw == 1) narrow[p] = (uint8_t)ids[p];
                else { const uint16_t v = (uint16_t)ids[p]; memcpy(&narrow[(size_t)p * 2], &v, 2); }
            }
            ok = ok && (len == 0 || fwrite(narrow.data(), 1, narrow.size(), f) == narrow.size());
        }
        ok = ok && (len == 0 || fwrite(h->hl_codes[l].data(), 1, (size_t)len * m, f) == (size_t)len * m);
    }
    if (!ok) return fail(IVFADC_ERR_INVALID, "short write to %s", path);
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_load_index(ivfadc_t **out, int device, const char *path, int *out_index_bits)
try {
    if (!out || !path) return fail(IVFADC_ERR_INVALID, "null argument");
    *out = nullptr;
    FileCloser fc{fopen(path, "rb")};
    FILE *f = fc.f;
    if (!f) return fail(IVFADC_ERR_INVALID, "cannot open %s", path);
    struct stat sb;
    if (fstat(fileno(f), &sb) != 0 || sb.st_size < 0) return fail(IVFADC_ERR_INVALID, "%s: cannot stat", path);
    const unsigned long long fsize = (unsigned long long)sb.st_size;
    std::string ln[9];
    for (int i = 0; i < 9; ++i)
        if (!read_line(f, ln[i])) return fail(IVFADC_ERR_INVALID, "%s: truncated header", path);
    long long nrows = 0, nclusters = 0, n = 0, m = 0, k = 0, dsub = 0;
    if (sscanf(ln[0].c_str(), "%lld %lld", &nrows, &nclusters) != 2 || sscanf(ln[1].c_str(), "%lld %lld %lld %lld", &n, &m, &k, &dsub) != 4)
        return fail(IVFADC_ERR_INVALID, "%s: bad header", path);
    // What the HIP path can search with the reference's semantics, and nothing else: a file of another quantizer,
    // quantization or distance would load and then be searched with the wrong arithmetic.  Type names come as `X` or
    // `Module.X` (persistency.jl:14-19 writes string(T); :137-144 reads both spellings).
    if (last_component(ln[2]) != "NaiveQuantizer")
        return fail(IVFADC_ERR_INVALID, "only NaiveQuantizer files are supported, got %s", ln[2].c_str());
    if (last_component(ln[3]) != "OrthogonalQuantization")
        return fail(IVFADC_ERR_INVALID, "quantization %s is not supported (only OrthogonalQuantization, i.e. :pq)", ln[3].c_str());
    if (ln[4] != "UInt8") return fail(IVFADC_ERR_INVALID, "quantization element type %s (only UInt8)", ln[4].c_str());
    int ibytes = 0;
    if (ln[5] == "UInt8") ibytes = 1; else if (ln[5] == "UInt16") ibytes = 2; else if (ln[5] == "UInt32") ibytes = 4;
    else return fail(IVFADC_ERR_INVALID, "index type %s is not supported by the HIP path", ln[5].c_str());
    if (last_component(ln[6]) != "SqEuclidean")
        return fail(IVFADC_ERR_INVALID, "coarse distance %s is not supported (only SqEuclidean)", ln[6].c_str());
    if (last_component(ln[7]) != "SqEuclidean")
        return fail(IVFADC_ERR_INVALID, "residual distance %s is not supported (only SqEuclidean)", ln[7].c_str());
    int tbytes = 0;
    if (ln[8] == "Float32") tbytes = 4; else if (ln[8] == "Float64") tbytes = 8;
    else return fail(IVFADC_ERR_INVALID, "element type %s", ln[8].c_str());
    if (nrows < 1 || nclusters < 1 || m < 1 || k < 1 || k > 256 || dsub < 1 || n < 0) return fail(IVFADC_ERR_INVALID, "%s: inconsistent sizes", path);
    // every size is checked against what the file can hold BEFORE anything is allocated from it (a corrupt or hostile
    // header must come back as IVFADC_ERR_INVALID, not as std::bad_alloc); 128-bit products cannot wrap
    typedef unsigned __int128 u128;
    const long pos0 = ftell(f);
    if (pos0 < 0) return fail(IVFADC_ERR_INVALID, "%s: ftell failed", path);
    const u128 remain = (u128)(fsize - std::min<unsigned long long>(fsize, (unsigned long long)pos0));
    if (nrows > (1ll << 24) || nclusters > (1ll << 31) - 1 || m > nrows || dsub > nrows || (u128)m * (u128)dsub != (u128)nrows)
        return fail(IVFADC_ERR_INVALID, "%s: inconsistent sizes", path);
    if (n > (long long)0xFFFFFFFFll) return fail(IVFADC_ERR_INVALID, "%s: %lld vectors exceed UInt32 ids", path, n);
    const u128 quant_bytes = (u128)tbytes * (u128)nrows * (u128)nclusters + (u128)m * ((u128)k + (u128)tbytes * (u128)dsub * (u128)k) +
                             (u128)tbytes * (u128)nrows * (u128)nrows;
    const u128 list_bytes = (u128)nclusters * 8 + (u128)n * (u128)(ibytes + m);
    if (quant_bytes + list_bytes > remain)
        return fail(IVFADC_ERR_INVALID, "%s: header describes more data than the file holds (truncated or corrupt)", path);
    auto read_floats = [&](float *dst, size_t cnt) {
        if (tbytes == 4) return read_exact(f, dst, cnt * 4);
        std::vector<double> tmp(cnt);
        if (!read_exact(f, tmp.data(), cnt * 8)) return false;
        for (size_t i = 0; i < cnt; ++i) dst[i] = (float)tmp[i];
        return true;
    };
    std::vector<float> cent((size_t)nclusters * nrows), cbs((size_t)nrows * k), row((size_t)std::max(k, nrows));
    std::vector<uint8_t> lab((size_t)m * k);
    if (!read_floats(cent.data(), cent.size())) return fail(IVFADC_ERR_INVALID, "%s: truncated centroids", path);
    for (long long i = 0; i < m; ++i) {
        if (!read_exact(f, lab.data() + (size_t)i * k, (size_t)k)) return fail(IVFADC_ERR_INVALID, "%s: truncated codebook", path);
        for (long long j = 0; j < dsub; ++j) {
            if (!read_floats(row.data(), (size_t)k)) return fail(IVFADC_ERR_INVALID, "%s: truncated codebook", path);
            for (long long c = 0; c < k; ++c) cbs[((size_t)i * k + c) * dsub + j] = row[c];
        }
    }
    // rotation matrix (persistency.jl:62-64).  knn_search never reads it (index.jl:204-258): an :opq index is searched as it is.
    // quantize_data -- push! -- does read it: the handle keeps the matrix (it is written back by ivfadc_save_index) and refuses to encode.
    std::vector<float> rotm((size_t)nrows * nrows);
    bool rot_identity = true;
    for (long long i = 0; i < nrows; ++i) {
        if (!read_floats(row.data(), (size_t)nrows)) return fail(IVFADC_ERR_INVALID, "%s: truncated rotation", path);
        for (long long j = 0; j < nrows; ++j) {
            if (!std::isfinite(row[j])) return fail(IVFADC_ERR_INVALID, "%s: non-finite rotation entry", path);
            rotm[(size_t)i * nrows + j] = row[j];
            rot_identity = rot_identity && row[j] == (i == j ? 1.0f : 0.0f);
        }
    }
    std::vector<int64_t> offsets((size_t)nclusters + 1, 0);
    std::vector<uint8_t> codes, raw;
    std::vector<uint32_t> ids;
    codes.reserve((size_t)n * m);
    ids.reserve((size_t)n);
    for (long long l = 0; l < nclusters; ++l) {
        int64_t len = 0;
        if (!read_exact(f, &len, 8) || len < 0) return fail(IVFADC_ERR_INVALID, "%s: truncated list %lld", path, l);
        if (len > n - offsets[l]) return fail(IVFADC_ERR_INVALID, "%s: list %lld holds %lld entries, more than the header's n leaves", path, l, (long long)len);
        raw.resize((size_t)len * ibytes);
        if (!read_exact(f, raw.data(), raw.size())) return fail(IVFADC_ERR_INVALID, "%s: truncated list %lld", path, l);
        for (int64_t p = 0; p < len; ++p) {
            uint32_t v = 0;
            memcpy(&v, raw.data() + (size_t)p * ibytes, ibytes);   // little-endian narrow ids
            ids.push_back(v);
        }
        const size_t at = codes.size();
        codes.resize(at + (size_t)len * m);
        if (!read_exact(f, codes.data() + at, (size_t)len * m)) return fail(IVFADC_ERR_INVALID, "%s: truncated list %lld", path, l);
        offsets[l + 1] = offsets[l] + len;
    }
    if (offsets[nclusters] != n) return fail(IVFADC_ERR_INVALID, "%s: the lists hold %lld entries, the header says %lld", path, (long long)offsets[nclusters], n);
    ivfadc_t *h = nullptr;
    TRY(ivfadc_create(&h, device, (int)nrows, (int)nclusters, (int)m, (int)k, cent.data(), cbs.data(), lab.data()));
    const int rc = ivfadc_set_lists(h, offsets.data(), codes.data(), ids.data());
    if (rc != IVFADC_OK) { ivfadc_destroy(h); return rc; }
    if (!rot_identity) h->rot = std::move(rotm);
    if (out_index_bits) *out_index_bits = ibytes * 8;
    *out = h;
    return IVFADC_OK;
} IVF_CATCH

int ivfadc_get_rotation(ivfadc_t *h, int *out_rotated, float *out_rot)
try {
    HandleLock lk_(h);
    if (!h) return fail(IVFADC_ERR_INVALID, "null handle");
    if (out_rotated) *out_rotated = h->rot.empty() ? 0 : 1;
    if (out_rot) {
        if (!h->rot.empty()) memcpy(out_rot, h->rot.data(), h->rot.size() * 4);
        else
            for (int i = 0; i < h->d; ++i)
                for (int j = 0; j < h->d; ++j) out_rot[(size_t)i * h->d + j] = i == j ? 1.0f : 0.0f;
    }
    return IVFADC_OK;
} IVF_CATCH
