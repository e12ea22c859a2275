// sbwtgpu_capi.cpp -- implementation of the C ABI declared in include/sbwtgpu.h.
// Host-side glue only: builds the device image of an index, owns device memory, and enqueues
// the kernels of sbwt_search.hip and its siblings.  There is deliberately no CPU query path in this library:
// every query entry point runs on the GPU or fails with an error code.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <cstdarg>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <mutex>
#include <cstddef>
#include <cstring>
#include <new>
#include <vector>
#include <thread>

#include "../../include/sbwtgpu.h"
#include "sbwt_device.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(e_ == hipErrorOutOfMemory ? SBWTGPU_ERR_OOM : SBWTGPU_ERR_HIP,         \
                        "%s failed: %s", #expr, hipGetErrorString(e_));                        \
    } while (0)

// Depth of the device-side prefix table: deep enough that most walks start with a nearly unique
// interval (ceil(log4(n_nodes)) + 2), capped at 14 (4 GiB of the 288 GB of HBM) unless
// SBWTGPU_DEVICE_PRECALC asks otherwise.  Measured on MI355X: 12.8 M columns (tools/ab_bench.py)
// 8: 33, 10: 45, 11: 47, 12: 49 G k-mers/s; 142 M columns (bench.py --config 3) 11: 30.2, 12: 32.2,
// 13: 34.1, 14: 36.1 G k-mers/s.
// Depth of the dense device prefix table.  Without the derived structures it is the only accelerator of the
// walks: two levels past log4(n), where most entries are empty and end a probe at once.  With the sparse table
// and the probe filter it only backs exact fall-back walks, and log4(n) does (measured: 12 and 14 give the
// same 7.08 ms on config 2, and 4 GB less image).
int default_device_precalc(int64_t n_nodes, bool derived) {
    const char *e = getenv("SBWTGPU_DEVICE_PRECALC");
    int v;
    if (e) {
        v = atoi(e);
    } else {
        v = 1;
        while (v < 14 && ((int64_t)1 << (2 * v)) < n_nodes) v++;
        if (!derived) v = v + 2 > 14 ? 14 : v + 2;
        // An image with the sparse table and the filter hardly ever walks from the dense table (0.00 dense lookups per read on
        // configs 2, 3 and 5: whole k-mers and probe windows have their own structures), so a table of 4^8 entries serves the
        // rare window that fits neither -- not one entry per column: 21 B per column on config 2, 30 on config 3 (round 5)
        else if (v > 8) v = 8;
    }
    if (v < 0) v = 0;
    if (v > 14) v = 14;
    return v;
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
};
struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace

// tuning knobs (A/B experiments; defaults are the shipped configuration)
// 0 = k_search (reference order), 1 = k_search_cert on the blocks, 4 = k_search_cert along the path order (two passes: encode +
// search; 2 and 3, kernels of earlier rounds, mean 4 now), 5 = the fused route (k_search_fused + the general kernel behind it)
static int tuning_variant() {
    static int v = [] { const char *e = getenv("SBWTGPU_SEARCH_VARIANT"); return e ? atoi(e) : -1; }();
    return v;
}
static int g_variant_override = -1, g_probe_override = -1, g_debug = 0, g_derive_ssup = 1, g_kernel_events = 0;
// the fused kernel with its lanes sorted by state ("fused_sort": 0 off, 1 on; env SBWTGPU_FUSED_SORT; sbwt_search_fused.hip)
static int g_fused_table = 1;      // the fused route's ticket table for batches with many long reads ("fused_table")
static int g_fused_sort = [] { const char *e = getenv("SBWTGPU_FUSED_SORT"); return e ? atoi(e) : SBWT_FUSED_SORT_DEFAULT; }();
// set around a search call by the *_i32 entry points: the kernels of this call write int32 results (SbwtIndexView::out32)
static thread_local int t_out32 = 0;
static int64_t g_ev_count = 0;
// depth of the sparse (hashed) prefix table built at index creation (capped at k and at 31 = one 62-bit key)
// debug aid for the parity tests: fill the result range with a poison pattern before every search, so that a
// result the kernel never writes cannot inherit a correct value from an earlier launch
static int g_poison = [] { const char *e = getenv("SBWTGPU_POISON_RESULTS"); return e ? atoi(e) : 0; }();
static int g_probe_filter = [] { const char *e = getenv("SBWTGPU_PROBE_FILTER"); return e ? atoi(e) : 1; }();
static int g_image_level = [] { const char *e = getenv("SBWTGPU_IMAGE_LEVEL"); return e ? atoi(e) : 0; }();
static int64_t g_max_image_bytes = [] { const char *e = getenv("SBWTGPU_MAX_IMAGE_BYTES"); return e ? (int64_t)atoll(e) : (int64_t)0; }();
static int g_sort_reads = [] { const char *e = getenv("SBWTGPU_SORT_READS"); return e ? atoi(e) : -1; }();   // -1 auto, 0 off, 1 on
static int g_force_mega = 0;    // tests: store every image's block counts relative to mega[c][0] (the dense rank-only layout)
static int g_path_safe = [] { const char *e = getenv("SBWTGPU_PATH_SAFE"); return e ? atoi(e) : 2; }();   // 0 off, 1 narrow rule, 2 wide
static int g_path_lookahead = [] { const char *e = getenv("SBWTGPU_PATH_LOOKAHEAD"); return e ? atoi(e) : 8; }();   // 0: the blind rule
static int g_path_order = [] { const char *e = getenv("SBWTGPU_PATH_ORDER"); return e ? atoi(e) : 1; }();
// the full image (path order, sparse table, filter) for indexes of 2^31 .. 2^32 - 2^24 columns with k <= 31 (round 5): columns
// and positions as full 32-bit unsigned values, read by k_search_fused<false, false, BIG>.  0: such an index gets blocks +
// dense table only, as before round 5 (23 G k-mers/s on 2.25 x 10^9 columns)
// SBWTGPU_VERBOSE=1: index_create says on stderr which part of the image it is building and how long each took (a build of
// 10^9 columns takes 10-30 s; a stall names its phase)
static int g_verbose = [] { const char *e = getenv("SBWTGPU_VERBOSE"); return e ? atoi(e) : 0; }();
struct PhaseLog {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void operator()(const char *what) {
        if (!g_verbose) return;
        (void)hipDeviceSynchronize();
        const auto t1 = std::chrono::steady_clock::now();
        size_t fr = 0, tot = 0;
        (void)hipMemGetInfo(&fr, &tot);
        fprintf(stderr, "sbwtgpu: index_create: %-34s %8.2f s   (%.1f GB of device memory free)\n", what,
                std::chrono::duration<double>(t1 - t0).count(), (double)fr / 1e9);
        t0 = t1;
    }
};
static int g_big_path = [] { const char *e = getenv("SBWTGPU_BIG_PATH"); return e ? atoi(e) : 1; }();
// stitched chains (sbwt_derived.hip): 0 = vertex-disjoint paths only; the shortest stretch worth copying
// long reads are cut into pieces on the device (SbwtPieceTab); 0: one lane per read whatever its length (tests)
static int g_split_long = [] { const char *e = getenv("SBWTGPU_SPLIT_LONG"); return e ? atoi(e) : 1; }();
// batches of reads of different lengths through the fused kernel (it fetches their offsets); 0: only reads of one length
static int g_fused_ragged = [] { const char *e = getenv("SBWTGPU_FUSED_RAGGED"); return e ? atoi(e) : 1; }();
// reads of more than 160 bases through the fused kernel as up to this many pieces of 160 bases (1: such reads go to the general
// kernel).  -1 (default) = 3: for 31 < k <= 63 the fused kernel walks with F_CMP (250-base reads 115 -> 243 G k-mers/s, the
// general kernel has neither bridges nor anchors there); for k <= 31 pieces and the two-pass route came out the same until
// round 5 made the fused kernel's writer cheaper (250-base reads: 235 vs 228 G, lengths 80-250: 203 vs 200 G, NOTES.md)
static int g_fused_pieces = [] { const char *e = getenv("SBWTGPU_FUSED_PIECES"); return e ? atoi(e) : -1; }();
static int g_path_stitch = [] { const char *e = getenv("SBWTGPU_PATH_STITCH"); return e ? atoi(e) : 1; }();
static int g_path_stitch_min = [] { const char *e = getenv("SBWTGPU_PATH_STITCH_MIN"); return e ? atoi(e) : 1; }();
// buckets of the sparse tables per 100 columns (two 16-byte entries each): 125 = 40 % of the slots in use (~6 % of the keys
// overflow their bucket), 100 = 50 % (~10 %): 8 bytes per column against 0.1 more probes per read.  0 (default) = by the index:
// 100 for k <= 31 (round 5, one box: config 2 4.74 vs 4.90 ms, config 3 6.39 vs 6.54 ms -- the smaller table is no slower), 125 where
// whole k-mers take two levels (k = 63: 4.18 ms at 125, 4.26 at 110, 4.29 at 100)
static int g_sparse_buckets_pct = [] { const char *e = getenv("SBWTGPU_SPARSE_BUCKETS_PCT"); int v = e ? atoi(e) : 0; return v <= 0 ? 0 : v < 60 ? 60 : v > 400 ? 400 : v; }();
static int g_sparse_depth = [] { const char *e = getenv("SBWTGPU_SPARSE_PRECALC"); return e ? atoi(e) : 31; }();

struct sbwtgpu_index {
    SbwtBlobHeader h;
    int device = 0;
    char *blob = nullptr;       // device
    bool owns_blob = true;
    // probe length of the certificate walks: long enough that a random string of that length is almost
    // surely absent (log4(#k-mers) + 4), at least one char longer than the device prefix table
    int probe_len(bool allow_override = true) const {
        if (allow_override && g_probe_override >= 0) return g_probe_override < h.k ? g_probe_override : 0;
        int64_t nk = h.n_kmers > 0 ? h.n_kmers : h.n_nodes;
        int L = 4;
        while (L < 40 && ((int64_t)1 << (2 * L)) < nk) L++;
        L += 4;
        if (L < h.p_dev + 2) L = (int)h.p_dev + 2;
        if (L > h.k - 1) return 0;
        return L;
    }
    SbwtIndexView view() const {
        SbwtIndexView v;
        v.blocks = reinterpret_cast<const uint4 *>(blob + h.off_blocks);
        v.ptab = h.p_dev > 0 ? reinterpret_cast<const longlong2 *>(blob + h.off_ptab) : nullptr;
        v.mega = reinterpret_cast<const unsigned long long *>(blob + h.off_mega);
        v.n_nodes = h.n_nodes;
        v.n_pos = h.n_pos > 0 ? h.n_pos : h.n_nodes;
        for (int i = 0; i < 4; i++) v.C[i] = h.C[i];
        v.k = (int)h.k;
        v.p_dev = (int)h.p_dev;
        v.n_mega = (int)h.n_mega;
        v.has_ssup = h.has_ssup;
        v.probe_len = probe_len();
        v.debug = g_debug;
        v.fused_sort = g_fused_sort;
        v.big = h.big_layout;
        v.out32 = t_out32;
        v.force_mega = h.force_mega;
        v.p_sparse = (int)h.p_sparse;
        v.n_sb = (unsigned)h.n_sb;
        v.stab = h.p_sparse > 0 ? reinterpret_cast<const uint4 *>(blob + h.off_stab) : nullptr;
        v.col = h.has_path ? reinterpret_cast<const unsigned *>(blob + h.off_col) : nullptr;
        v.pos = h.has_path ? reinterpret_cast<const unsigned *>(blob + h.off_pos) : nullptr;
        v.pq = h.has_path ? reinterpret_cast<const uint4 *>(blob + h.off_pq) : nullptr;
        v.trans = (h.has_path && h.n_tslots > 0) ? reinterpret_cast<const uint4 *>(blob + h.off_trans) : nullptr;
        v.stab_pos = h.stab_pos;
        v.n_tslots = (unsigned)h.n_tslots;
        v.has_safe = h.has_safe;
        v.stab2 = h.n_sb2 > 0 ? reinterpret_cast<const uint4 *>(blob + h.off_stab2) : nullptr;
        v.n_sb2 = (unsigned)h.n_sb2;
        v.pfil = h.p_filter > 0 ? reinterpret_cast<const uint4 *>(blob + h.off_pfil) : nullptr;
        v.p_filter = (int)h.p_filter;
        v.log2f = (int)h.log2f;
        return v;
    }
};

// Which path-order route by default: the fused one (equal-length batches in one kernel, everything else through the
// general kernel behind it).
static inline int auto_variant(const SbwtBlobHeader &) { return 5; }

extern "C" {

const char *sbwtgpu_version(void) { return "sbwtgpu 0.1 (gfx950)"; }
const char *sbwtgpu_last_error(void) { return g_err; }

int sbwtgpu_set_tuning(const char *key, int64_t value) {
    if (!key) return fail(SBWTGPU_ERR_INVALID_ARG, "key is NULL");
    if (!strcmp(key, "search_variant")) { g_variant_override = (int)value; return SBWTGPU_OK; }
    if (!strcmp(key, "probe_len")) { g_probe_override = (int)value; return SBWTGPU_OK; }
    if (!strcmp(key, "debug")) { g_debug = (int)value; return SBWTGPU_OK; }
    if (!strcmp(key, "kernel_events")) { g_kernel_events = (int)value; if (value) g_ev_count = 0; return SBWTGPU_OK; }
    if (!strcmp(key, "derive_ssup")) { g_derive_ssup = (int)value; return SBWTGPU_OK; }
    if (!strcmp(key, "poison_results")) { g_poison = (int)value; return SBWTGPU_OK; }
    if (!strcmp(key, "probe_filter")) { g_probe_filter = (int)value; return SBWTGPU_OK; }   // indexes created afterwards
    if (!strcmp(key, "big_path")) { g_big_path = (int)value; return SBWTGPU_OK; }           // indexes created afterwards
    if (!strcmp(key, "image_level")) { g_image_level = (int)value; return SBWTGPU_OK; }          // indexes created afterwards
    if (!strcmp(key, "max_image_bytes")) { g_max_image_bytes = value; return SBWTGPU_OK; }     // indexes created afterwards
    if (!strcmp(key, "sort_reads")) { g_sort_reads = (int)value; return SBWTGPU_OK; }
    if (!strcmp(key, "trans_wide")) return SBWTGPU_OK;     // (round 2: wide transition entries; the table is sparse now)
    if (!strcmp(key, "force_mega")) { g_force_mega = (int)value; return SBWTGPU_OK; }     // indexes created afterwards
    if (!strcmp(key, "trans_ext")) return SBWTGPU_OK;      // (round 2: transitions always run on along their quoted steps now)
    if (!strcmp(key, "path_safe")) { g_path_safe = (int)value; return SBWTGPU_OK; }   // indexes created afterwards
    if (!strcmp(key, "path_lookahead")) { g_path_lookahead = value < 0 ? 0 : value > 64 ? 64 : (int)value; return SBWTGPU_OK; }
    if (!strcmp(key, "path_order")) { g_path_order = (int)value; return SBWTGPU_OK; }   // indexes created afterwards
    if (!strcmp(key, "fused_table")) { g_fused_table = (int)value; return SBWTGPU_OK; }
    if (!strcmp(key, "fused_sort")) { g_fused_sort = (int)value; return SBWTGPU_OK; }
    if (!strcmp(key, "fused_ragged")) { g_fused_ragged = (int)value; return SBWTGPU_OK; }
    if (!strcmp(key, "fused_pieces")) { g_fused_pieces = value < 1 ? -1 : value > 3 ? 3 : (int)value; return SBWTGPU_OK; }
    if (!strcmp(key, "split_long")) { g_split_long = (int)value; return SBWTGPU_OK; }
    if (!strcmp(key, "path_stitch")) { g_path_stitch = (int)value; return SBWTGPU_OK; }   // indexes created afterwards
    if (!strcmp(key, "path_stitch_min")) { g_path_stitch_min = (int)value < 1 ? 1 : (int)value; return SBWTGPU_OK; }
    if (!strcmp(key, "sparse_depth")) {      // takes effect for indexes created afterwards
        if (value < 0 || value > 31) return fail(SBWTGPU_ERR_INVALID_ARG, "sparse_depth must be in [0,31]");
        g_sparse_depth = (int)value;
        return SBWTGPU_OK;
    }
    return fail(SBWTGPU_ERR_INVALID_ARG, "unknown tuning key '%s'", key);
}

int sbwtgpu_device_count(int *count) {
    if (!count) return fail(SBWTGPU_ERR_INVALID_ARG, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(SBWTGPU_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return SBWTGPU_OK;
}

static inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }
static inline int64_t a256(int64_t x) { return align256(x); }

// set while sbwtgpu_index_create retries without the derived structures after running out of device memory
// Image levels: 0 = everything (path order, transition table, sparse table, probe filter: 139-168 bytes per column),
// 1 = no path order (sparse table + filter + dense prefix table: ~66 bytes per column), 2 = blocks + dense prefix table
// only (1 byte per column + the table).  index_create starts at "image_level" (tuning / SBWTGPU_IMAGE_LEVEL), and moves
// to the next level when the image would exceed "max_image_bytes" (SBWTGPU_MAX_IMAGE_BYTES) or device memory runs out.
static thread_local int t_image_level = -1;

int sbwtgpu_index_create(const sbwtgpu_index_desc *d, int device, sbwtgpu_index **out) {
    if (!d || !out) return fail(SBWTGPU_ERR_INVALID_ARG, "desc/out is NULL");
    *out = nullptr;
    if (d->n_nodes <= 0 || !d->A_bits || !d->C_bits || !d->G_bits || !d->T_bits)
        return fail(SBWTGPU_ERR_INVALID_ARG, "n_nodes must be > 0 and the four bit vectors non-NULL");
    if (d->k <= 0 || d->k > 255) return fail(SBWTGPU_ERR_INVALID_ARG, "k must be in [1,255]");
    if (d->precalc_k < 0) return fail(SBWTGPU_ERR_INVALID_ARG, "precalc_k < 0");
    if (d->precalc_k > 20)
        return fail(SBWTGPU_ERR_PRECALC_TOO_LONG,
                    "Error: Can't precalc longer than 20-mers (would take over 4^20 = 2^40 bytes");
    if (d->precalc_k > d->k)
        return fail(SBWTGPU_ERR_PRECALC_GT_K, "Error: Precalc length is longer than k (%lld > %lld)",
                    (long long)d->precalc_k, (long long)d->k);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(SBWTGPU_ERR_NO_DEVICE, "no HIP device");
    if (device < 0 || device >= ndev) return fail(SBWTGPU_ERR_NO_DEVICE, "device %d out of range", device);
    DeviceGuard guard(device);
    if (!guard.ok) return fail(SBWTGPU_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);

    const int64_t n = d->n_nodes;
    const int64_t nw = (n + 63) / 64;
    const int64_t n_blocks = n / 64 + 1;
    const int64_t n_mega = (n >> SBWT_MEGA_SHIFT) + 1;
    int64_t p_file = d->precalc_k;
    const int level = t_image_level >= 0 ? t_image_level : (g_image_level < 0 ? 0 : g_image_level > 2 ? 2 : g_image_level);
    const bool t_minimal_image = level >= 2;
    DevBuf d_bits, d_bscr;                             // the uploaded bit vectors and the block builder's scratch (below)
    auto retry_next_level = [&]() {
        // (this frame's device buffers first: the retry uploads its own, and holding both could push a level that fits down
        // another level)
        if (d_bits.p) { (void)hipFree(d_bits.p); d_bits.p = nullptr; }
        if (d_bscr.p) { (void)hipFree(d_bscr.p); d_bscr.p = nullptr; }
        // never silently: a stepped-down image answers the same and three to ten times slower
        fprintf(stderr, "sbwtgpu: the level-%d image of this index (%lld columns, k = %lld) does not fit the device%s: building level %d "
                        "(%s)\n", level, (long long)n, (long long)d->k, g_max_image_bytes > 0 ? " or \"max_image_bytes\"" : "", level + 1,
                level + 1 == 1 ? "no path order: the blocks-only search kernel" : "blocks and dense prefix table only");
        t_image_level = level + 1;
        const int rc2 = sbwtgpu_index_create(d, device, out);
        t_image_level = -1;
        return rc2;
    };
    // 2^31 - 64 <= n < 2^32 - 2^24 (round 5): the same derived structures with full 32-bit unsigned columns and positions,
    // when whole k-mers fit the sparse table (k <= 31: every found k-mer comes with its path position, so no segment source
    // needs bit 31 as a flag) and the marks are there; absolute 32-bit block counts (the mega table stays zero)
    // (round 6: 31 < k <= 63 as well -- the second-level table's entries hold position + 1 and no flags in that layout.
    // "big_path" 2 gives ANY index that layout: the BIG instantiations are tested on small indexes)
    const bool big_range = (n >= ((int64_t)1 << 31) - 64 || g_big_path == 2) && n < ((int64_t)1 << 32) - ((int64_t)1 << 24);
    const bool big_k = d->k <= SBWT_SP_MAX_DEPTH ? g_sparse_depth >= d->k
                                                 : (d->k - SBWT_SP_MAX_DEPTH <= 32 && g_sparse_depth >= SBWT_SP_MAX_DEPTH);
    const bool big_path = big_range && g_big_path && level == 0 && d->k > 16 && big_k &&
                          g_probe_filter && g_path_order && (d->suffix_group_starts || g_derive_ssup);
    const bool derived = !t_minimal_image && g_sparse_depth > 0 && g_probe_filter &&
                         ((n < ((int64_t)1 << 31) - 64 && n_mega == 1) || big_path) && d->k > 16;
    int64_t p_dev = default_device_precalc(n, derived);
    if (p_dev < p_file) p_dev = p_file;
    if (p_dev > d->k) p_dev = d->k;
    if (level >= 2 && g_max_image_bytes > 0) {
        // the smallest kind of image under a cap: the deepest dense table that still fits (each level is 4x the bytes)
        auto est = [&](int64_t pd) {
            return align256(n_blocks * 64) + (pd > 0 ? (int64_t)16 << (2 * pd) : 0) +
                   ((p_file > 0 && p_file != pd) ? (int64_t)16 << (2 * p_file) : 0) + 4 * n_mega * 8 + 1024;
        };
        while (p_dev > p_file && p_dev > 0 && est(p_dev) > g_max_image_bytes) p_dev--;
    }

    sbwtgpu_index *idx = new (std::nothrow) sbwtgpu_index();
    if (!idx) return fail(SBWTGPU_ERR_OOM, "out of host memory");
    SbwtBlobHeader &h = idx->h;
    memset(&h, 0, sizeof(h));
    h.magic = SBWT_BLOB_MAGIC;
    h.n_nodes = n;
    h.n_kmers = d->n_kmers;
    h.k = d->k;
    h.p_file = p_file;
    h.p_dev = p_dev;
    h.n_blocks = n_blocks;
    h.n_mega = n_mega;
    h.has_ssup = d->suffix_group_starts ? 1 : 0;
    h.off_blocks = 0;
    h.off_ptab = align256(n_blocks * 64);
    int64_t ptab_bytes = p_dev > 0 ? (int64_t)16 << (2 * p_dev) : 0;   // revised below for rank-only images
    h.off_ftab = align256(h.off_ptab + ptab_bytes);
    int64_t ftab_bytes = (p_file > 0 && p_file != p_dev) ? (int64_t)16 << (2 * p_file) : 0;
    (void)ptab_bytes;
    if (ftab_bytes == 0) h.off_ftab = h.off_ptab;   // the same table serves both
    h.off_mega = align256(h.off_ftab + (ftab_bytes ? ftab_bytes : ptab_bytes));
    h.blob_bytes = align256(h.off_mega + 4 * n_mega * 8);
    // sparse table one level above the dense one: 32-bit intervals only, one bucket per column on average
    int64_t p_sparse = g_sparse_depth < d->k ? g_sparse_depth : d->k;
    if (t_minimal_image) p_sparse = 0;
    if (p_sparse > SBWT_SP_MAX_DEPTH) p_sparse = SBWT_SP_MAX_DEPTH;
    if (p_sparse <= p_dev || p_dev <= 0 || n >= ((int64_t)1 << 32) || (n_mega > 1 && !big_path)) p_sparse = 0;
    // (2^31 columns and more: whole k-mers with their positions -- in the table itself, or through the second level -- or nothing)
    if (big_path && p_sparse != d->k && !(p_sparse == SBWT_SP_MAX_DEPTH && d->k > p_sparse && d->k - p_sparse <= 32)) p_sparse = 0;
    if (p_sparse > 0) {
        h.p_sparse = (int32_t)p_sparse;
        // (2^31 columns and more with two levels: one bucket per column in both -- at 1.25 the two tables are 180 GB of a 2.25 x 10^9-
        // column image and leave their own builder 7 GB short of its 108 GB of scratch on a 288 GB device)
        const int64_t bpct = g_sparse_buckets_pct > 0 ? g_sparse_buckets_pct
                                                      : (d->k > SBWT_SP_MAX_DEPTH && !(big_path && n >= ((int64_t)1 << 31) - 64) ? 125 : 100);
        h.n_sb = n * bpct / 100 + 64;       // two-entry buckets per column
        h.off_stab = h.blob_bytes;
        h.blob_bytes = align256(h.off_stab + 32 * h.n_sb);
        // second level for 31 < k <= 63 (the remaining k-31 bases fit one 64-bit key): 1.25 two-entry buckets per column, like the
        // first level (round 3: 2 n single 32-byte entries with linear probing -- a miss walked 2.5 of them)
        // (its entries keep their flags in bit 31 of the column / origin words: columns below 2^31 only -- p_sparse > 0 already
        // implies one mega block; said again here because a wider first level must not widen this one by accident)
        if (p_sparse == SBWT_SP_MAX_DEPTH && d->k > p_sparse && d->k - p_sparse <= 32 && (n < ((int64_t)1 << 31) - 64 || big_path)) {
            h.n_sb2 = n * bpct / 100 + 64;
            h.off_stab2 = h.blob_bytes;
            h.blob_bytes = align256(h.off_stab2 + 32 * h.n_sb2);
        }
        // probe filter at the certificate probes' length: 128-bit blocks of 8 .. 16 windows (8 .. 16 bits per column; round 5:
        // half of what it was -- config 2 at 12 windows per block runs as at 6, 4.69 vs 4.66 ms; at 24 it is 2.5 % slower).
        // A filter of more than 128 MB goes up to 20 per block: config 3 (142 M columns) 6.23 ms at 4 per block / 512 MB,
        // 6.18 at 8 / 256 MB, 6.09 at 17 / 128 MB, 6.23 at 34 -- the smaller filter stays in the Infinity Cache, and that is
        // worth more than the false positives cost (profiles/r05_experiments.txt)
        const int L0 = idx->probe_len(false);
        if (g_probe_filter && L0 > p_dev && L0 <= p_sparse) {
            int lf = 4;
            while (((int64_t)16 << lf) < n) lf++;
            if (((int64_t)16 << lf) > ((int64_t)128 << 20) && n <= ((int64_t)10 << lf)) lf--;
            // (experiments: SBWTGPU_FILTER_LOG2_ADJ = -1 halves the filter once more, 1 doubles it)
            { const char *ea = getenv("SBWTGPU_FILTER_LOG2_ADJ"); if (ea) { lf += atoi(ea); if (lf < 4) lf = 4; } }
            h.p_filter = L0;
            h.log2f = lf;
            h.off_pfil = h.blob_bytes;
            h.blob_bytes = align256(h.off_pfil + ((int64_t)16 << lf));
        }
    }
    // path order: needs suffix-group marks (given or derived) and 32-bit columns
    const bool marks = d->suffix_group_starts || (g_derive_ssup && d->k >= 2);
    int64_t pos_cap = n;
    if (g_path_order && level == 0 && marks && ((n < ((int64_t)1 << 31) - 64 && n_mega == 1 && !big_path) || (big_path && p_sparse > 0))) {
        // room for the path order with stitched chains (copies of shared stretches: at most a fifth of the columns, and
        // positions stay below 2^31); the finished image keeps what was used.  (2^31 columns and more: disjoint paths only --
        // the copies' room and the stitching's temporaries are what such an image has no memory for)
        pos_cap = (g_path_stitch && !big_path) ? std::min<int64_t>(n + n / 5 + 64, ((int64_t)1 << 31) - 64) : n;
        if (pos_cap < n) pos_cap = n;
        h.has_path = 1;
        h.big_layout = big_path ? 1 : 0;
        h.off_col = h.blob_bytes;
        h.off_pos = align256(h.off_col + (pos_cap + 4) * 4);
        h.off_pq = align256(h.off_pos + (n + 4) * 4);
        h.off_trans = align256(h.off_pq + sbwt_path_quads(pos_cap) * 16);
        h.blob_bytes = h.off_trans;        // the transition table follows once the path order has said how many entries it needs
    }
    idx->device = device;

    // The bit vectors go to the device once; counting, the per-block prefix counts and the interleaving all happen
    // there (sbwt_build.hip) -- 142 M columns took seconds in host loops.
    const uint64_t *cols[4] = {d->A_bits, d->C_bits, d->G_bits, d->T_bits};
    {
        hipError_t eb = d_bits.alloc((size_t)(5 * nw) * 8);
        if (eb == hipSuccess) eb = d_bscr.alloc((size_t)sbwt_blocks_scratch_bytes(n));
        for (int c = 0; c < 4 && eb == hipSuccess; c++)
            eb = hipMemcpy(static_cast<char *>(d_bits.p) + (size_t)c * (size_t)nw * 8, cols[c], (size_t)nw * 8, hipMemcpyHostToDevice);
        if (eb == hipSuccess && d->suffix_group_starts)
            eb = hipMemcpy(static_cast<char *>(d_bits.p) + (size_t)4 * (size_t)nw * 8, d->suffix_group_starts, (size_t)nw * 8,
                           hipMemcpyHostToDevice);
        if (eb != hipSuccess) {
            delete idx;
            return fail(eb == hipErrorOutOfMemory ? SBWTGPU_ERR_OOM : SBWTGPU_ERR_HIP, "uploading the bit vectors: %s",
                        hipGetErrorString(eb));
        }
    }
    // C array (SBWT.hh:344-349): C[0] = 1 (ghost dollar into the root), C[i+1] = C[i] + rank(n, sigma_i)
    long long tot[5] = {0, 0, 0, 0, 0};
    if (sbwt_blocks_count(static_cast<const unsigned long long *>(d_bits.p), n, d_bscr.p, tot, 0) != 0) {
        delete idx;
        return fail(SBWTGPU_ERR_HIP, "counting the set bits on the device failed");
    }
    h.C[0] = 1;
    for (int c = 1; c < 4; c++) h.C[c] = h.C[c - 1] + tot[c - 1];
    h.path_lookahead = h.has_path ? g_path_lookahead : 0;
    // In an SBWT every column except the root has exactly one incoming edge, so the matrix holds
    // n_nodes - 1 set bits and every LF step stays inside [0, n_nodes).  Arbitrary bit vectors (the
    // stand-alone SubsetMatrixRank use) are still served, but only by rank(): walking them would
    // leave the image.
    const bool consistent = (tot[0] + tot[1] + tot[2] + tot[3] == n - 1);
    h.rank_only = consistent ? 0 : 1;
    // Dense arbitrary bit vectors: C[c] + rank_c can pass 2^32 well before n reaches 2^31 columns.  Then the 32-bit
    // block counts are stored relative to mega[c][0] = C[c] (inside one mega block rank_c <= 2^31 always fits).
    for (int c = 0; c < 4; c++) h.row_ones[c] = tot[c];
    h.force_mega = (n_mega == 1 && ((uint64_t)(h.C[3] + tot[3]) >= (1ull << 32) || (g_force_mega && !consistent))) ? 1 : 0;
    if (!consistent) {
        if (p_file > 0) {
            delete idx;
            return fail(SBWTGPU_ERR_INVALID_ARG,
                        "the four columns hold %lld set bits, an SBWT with %lld columns holds %lld: cannot "
                        "compute a prefix table", (long long)(tot[0] + tot[1] + tot[2] + tot[3]), (long long)n,
                        (long long)(n - 1));
        }
        p_dev = 0;
        h.p_dev = 0;
        h.has_ssup = 0;
        h.p_sparse = 0;
        h.n_sb2 = 0;
        h.p_filter = 0;
        h.has_path = 0;
        h.big_layout = 0;
    }
    if (d->precalc && p_file > 0) {
        const int64_t np = (int64_t)1 << (2 * p_file);
        for (int64_t e = 0; e < np; e++) {
            int64_t l = d->precalc[2 * e], r = d->precalc[2 * e + 1];
            if (!((l == -1 && r == -1) || (l >= 0 && l <= r && r < n))) {
                delete idx;
                return fail(SBWTGPU_ERR_INVALID_ARG, "prefix table entry %lld = (%lld, %lld) is outside the index",
                            (long long)e, (long long)l, (long long)r);
            }
        }
    }

    if (h.rank_only) {   // no tables in a rank-only image
        ptab_bytes = 0;
        ftab_bytes = 0;
        h.off_ptab = h.off_ftab = align256(n_blocks * 64);
        h.off_mega = h.off_ptab;
        h.blob_bytes = align256(h.off_mega + 4 * n_mega * 8);
        h.off_stab = 0;
        h.n_sb = 0;
        h.off_col = h.off_pos = h.off_pq = h.off_trans = 0;
    }
    h.image_level = (h.has_path ? 0 : h.p_sparse > 0 ? 1 : 2);
    if (g_max_image_bytes > 0 && h.blob_bytes > g_max_image_bytes) {
        // the derived structures are optional, and at level 2 the dense table shrinks until the image fits: an index without
        // path order or sparse table (k <= 16, or both switched off) still has that last resort
        if (level < 2) {
            delete idx;
            return retry_next_level();
        }
        delete idx;
        return fail(SBWTGPU_ERR_OOM, "the smallest image of this index (%lld bytes: blocks + depth-%lld prefix table) exceeds "
                    "max_image_bytes = %lld", (long long)h.blob_bytes, (long long)h.p_dev, (long long)g_max_image_bytes);
    }
    // An image with a path order is built in TWO allocations (round 6): first the index proper and the path arrays (at temporary
    // offsets right behind the mega table), then -- the path order says how long col[] is and how many transition entries there
    // are -- the image at its final size, into which the sparse tables and the filter are built directly.  (Before, everything
    // was built in one allocation and moved at the end: twice the image at the peak, which a 164 GB image of 2.25 x 10^9
    // columns at k = 32 does not have.)
    const int64_t f_off_col = h.off_col, f_blob_bytes = h.blob_bytes;       // (the one-allocation layout, for the size checks)
    int64_t p1_bytes = h.blob_bytes;
    if (h.has_path) {
        h.off_col = align256(h.off_mega + 4 * n_mega * 8);
        h.off_pos = align256(h.off_col + (pos_cap + 4) * 4);
        h.off_pq = align256(h.off_pos + (n + 4) * 4);
        p1_bytes = align256(h.off_pq + sbwt_path_quads(pos_cap) * 16);
    }
    (void)f_blob_bytes;
    hipError_t e = hipMalloc((void **)&idx->blob, (size_t)p1_bytes);
    if (e != hipSuccess && level < 2 && (h.has_path || h.p_sparse > 0)) {
        // the derived structures are optional: without them the blocks-only kernel serves
        (void)hipGetLastError();
        delete idx;
        return retry_next_level();
    }
    if (e != hipSuccess) {
        delete idx;
        return fail(SBWTGPU_ERR_OOM, "hipMalloc(%lld bytes) for the index image: %s", (long long)h.blob_bytes,
                    hipGetErrorString(e));
    }
    int rc = SBWTGPU_OK;
    unsigned char *alt_safe = nullptr;                 // per-position verdicts of the safe-bit pass, for the transition table
    // One scratch allocation serves the path order, the sparse tables and the safe bits in turn (round 6): fresh device memory
    // costs 20-30 ms per GB on this driver once an allocation goes beyond what the process has had before (2.25 x 10^9 columns: 128 GB
    // for the path order and 108 GB for the sparse tables were 5.9 s of an 10.8 s image).  It is let go early only where the final
    // image does not fit beside it (2.25 x 10^9 columns at k = 32: 167 GB).  SBWTGPU_SCRATCH_ARENA=0: an allocation per phase, as before.
    static const int use_arena = [] { const char *e = getenv("SBWTGPU_SCRATCH_ARENA"); return e ? atoi(e) : 1; }();
    void *arena = nullptr;
    size_t arena_bytes = 0;
    auto arena_drop = [&] { if (arena) (void)hipFree(arena); arena = nullptr; arena_bytes = 0; };
    auto arena_get = [&](size_t need, size_t later) -> void * {        // (later: what the phases behind this one will ask for)
        if (arena && arena_bytes >= need) return arena;
        arena_drop();
        const size_t want = (use_arena && later > need) ? later : need;
        if (hipMalloc(&arena, want) == hipSuccess) { arena_bytes = want; return arena; }
        (void)hipGetLastError();
        arena = nullptr;
        if (want > need && hipMalloc(&arena, need) == hipSuccess) { arena_bytes = need; return arena; }
        (void)hipGetLastError();
        arena = nullptr;
        return nullptr;
    };
    PhaseLog plog;
    plog("upload, counts, allocation");
    do {
        if ((e = hipMemset(idx->blob, 0, (size_t)p1_bytes)) != hipSuccess) break;
        {
            const long long Cs[4] = {h.C[0], h.C[1], h.C[2], h.C[3]};
            sbwt_blocks_fill(static_cast<const unsigned long long *>(d_bits.p),
                             d->suffix_group_starts ? static_cast<const unsigned long long *>(d_bits.p) + 4 * nw : nullptr, n,
                             d_bscr.p, Cs, ((n_mega > 1 && !(big_path && consistent)) || h.force_mega) ? 1 : 0, (int)n_mega,
                             reinterpret_cast<uint4 *>(idx->blob + h.off_blocks),
                             reinterpret_cast<unsigned long long *>(idx->blob + h.off_mega), 0);
            if ((e = hipDeviceSynchronize()) != hipSuccess) break;
            (void)hipFree(d_bits.p); d_bits.p = nullptr;          // the derived structures need the room
            (void)hipFree(d_bscr.p); d_bscr.p = nullptr;
        }
        SbwtIndexView v = idx->view();
        if (!d->suffix_group_starts && !h.rank_only && g_derive_ssup && d->k >= 2) {
            // no streaming support in the file: derive the marks for internal use (has_ssup stays 0)
            void *scr = nullptr;
            if ((e = hipMalloc(&scr, (size_t)sbwt_derive_scratch_bytes(n))) != hipSuccess) break;
            sbwt_launch_derive_marks(v, reinterpret_cast<uint4 *>(idx->blob + h.off_blocks), scr, 0);
            e = hipDeviceSynchronize();
            (void)hipFree(scr);
            if (e != hipSuccess) break;
            h.ssup_derived = 1;
        }
        plog("blocks (+ derived marks)");
        if (p_dev > 0) sbwt_launch_precalc(v, (int)p_dev, reinterpret_cast<longlong2 *>(idx->blob + h.off_ptab), 0);
        if (ftab_bytes) {
            if (d->precalc) {
                if ((e = hipMemcpy(idx->blob + h.off_ftab, d->precalc, (size_t)ftab_bytes, hipMemcpyHostToDevice)) !=
                    hipSuccess)
                    break;
            } else {
                sbwt_launch_precalc(v, (int)p_file, reinterpret_cast<longlong2 *>(idx->blob + h.off_ftab), 0);
            }
        }
        plog("dense prefix table(s)");
        if (h.has_path) {
            void *scr = arena_get((size_t)sbwt_path_scratch_bytes(n), h.p_sparse > 0 ? (size_t)sbwt_sparse_scratch_bytes(n) : 0);
            if (!scr) { e = hipErrorOutOfMemory; break; }
            if (g_verbose >= 2) plog("  (path order: scratch allocated)");
            long long n_pos = n;
            int prc = sbwt_launch_build_path(v, reinterpret_cast<unsigned *>(idx->blob + h.off_col),
                                             reinterpret_cast<unsigned *>(idx->blob + h.off_pos),
                                             reinterpret_cast<uint4 *>(idx->blob + h.off_pq), pos_cap, &n_pos, g_path_stitch,
                                             g_path_stitch_min, scr, g_path_lookahead, 0);
            if (g_verbose >= 2) plog("  (path order: built)");
            if (!use_arena) arena_drop();
            if (prc != 0) { e = hipErrorUnknown; break; }
            h.n_pos = n_pos;
            // col[n_pos] = 0xFFFFFFFF: the "position" whose column is -1 (the fused kernel's writer loads every result, absent
            // ones included, through col[]; the array has four entries of padding behind the last position)
            if ((e = hipMemset(idx->blob + h.off_col + (size_t)n_pos * 4, 0xFF, 16)) != hipSuccess) break;
            // ---- the final layout: the transition table's size (three entries per branching column, one per successor of a
            //      path's last column: counted now, filled in at the end) and the path arrays at the length the path order
            //      turned out to have; the image moves into its final allocation (the index proper and the path arrays:
            //      a tenth of it), the sparse tables and the filter are built there ----
            long long nb = 0;
            const long long n_ent = sbwt_launch_path_oth(idx->view(), reinterpret_cast<uint4 *>(idx->blob + h.off_pq), &nb, 0, 1);
            if (n_ent < 0) { e = hipErrorUnknown; break; }
            h.n_branch = nb;
            h.n_trans = n_ent;
            const int64_t n_slots = 3 * n_ent + 64;                         // load factor 1/3: ~1.25 probes per lookup
            if (n_slots >= ((int64_t)1 << 32)) { e = hipErrorOutOfMemory; break; }
            h.n_tslots = n_slots;
            const int64_t col_bytes = align256((h.n_pos + 4) * 4), pos_bytes = align256((n + 4) * 4);
            const int64_t pq_bytes = align256(sbwt_path_quads(h.n_pos) * 16);
            const int64_t new_pos = f_off_col + col_bytes, new_pq = new_pos + pos_bytes, new_trans = new_pq + pq_bytes;
            const int64_t full = align256(new_trans + 32 * n_slots);
            if (g_max_image_bytes > 0 && full > g_max_image_bytes && level < 2) { e = hipErrorOutOfMemory; break; }
            char *nblob = nullptr;
            if (g_verbose >= 2) plog("  (transition entries counted)");
            e = hipMalloc((void **)&nblob, (size_t)full);
            if (e != hipSuccess && arena) {             // (no room beside the scratch: the scratch goes first)
                (void)hipGetLastError();
                arena_drop();
                e = hipMalloc((void **)&nblob, (size_t)full);
            }
            if (e != hipSuccess) break;
            if (arena) {                                // (what is still to come -- alt_safe, the builders' small buffers -- must not fail for it)
                size_t fr = 0, tot = 0;
                if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < (size_t)h.n_pos * 2 + ((size_t)4 << 30)) arena_drop();
            }
            if (g_verbose >= 2) plog("  (final allocation)");
            const int64_t head = h.off_col;                                 // blocks, tables, mega: the same offsets in both
            if ((e = hipMemcpy(nblob, idx->blob, (size_t)head, hipMemcpyDeviceToDevice)) != hipSuccess ||
                (e = hipMemset(nblob + head, 0, (size_t)(f_off_col - head))) != hipSuccess ||        // the sparse tables' and the filter's room
                (e = hipMemcpy(nblob + f_off_col, idx->blob + h.off_col, (size_t)col_bytes, hipMemcpyDeviceToDevice)) != hipSuccess ||
                (e = hipMemcpy(nblob + new_pos, idx->blob + h.off_pos, (size_t)pos_bytes, hipMemcpyDeviceToDevice)) != hipSuccess ||
                (e = hipMemcpy(nblob + new_pq, idx->blob + h.off_pq, (size_t)pq_bytes, hipMemcpyDeviceToDevice)) != hipSuccess) {
                (void)hipFree(nblob);
                break;
            }
            if (g_verbose >= 2) plog("  (moved)");
            (void)hipFree(idx->blob);
            idx->blob = nblob;
            h.off_col = f_off_col;
            h.off_pos = new_pos;
            h.off_pq = new_pq;
            h.off_trans = new_trans;
            h.blob_bytes = full;
            v = idx->view();
        }
        plog("path order, final layout");
        if (h.p_sparse > 0) {
            void *scr = arena_get((size_t)sbwt_sparse_scratch_bytes(n), 0);
            if (!scr) { e = hipErrorOutOfMemory; break; }
            if (g_verbose >= 2) plog("  (sparse tables: scratch allocated)");
            int src = sbwt_launch_build_sparse(v, (int)p_dev, (int)h.p_sparse, (long long)h.n_sb,
                                               reinterpret_cast<uint4 *>(idx->blob + h.off_stab), scr,
                                               h.has_path ? reinterpret_cast<const unsigned *>(idx->blob + h.off_pos) : nullptr,
                                               (int)h.p_filter, (int)h.log2f,
                                               h.p_filter > 0 ? reinterpret_cast<uint4 *>(idx->blob + h.off_pfil) : nullptr,
                                               (long long)h.n_sb2,
                                               h.n_sb2 > 0 ? reinterpret_cast<uint4 *>(idx->blob + h.off_stab2) : nullptr, 0);
            e = hipDeviceSynchronize();
            if (g_verbose >= 2) plog("  (sparse tables: built)");
            if (!use_arena) arena_drop();
            if (src == -3) { h.n_sb2 = 0; src = 0; }   // some k-mer spans several columns: no second level
            if (src < 0 && e == hipSuccess) e = hipErrorUnknown;
            if (e != hipSuccess) break;
            h.stab_pos = src > 0 ? 1 : 0;
            plog("sparse table, second level, filter");
            // (2^31 columns and more: the fused kernel's BIG instantiation needs every k-mer's position in its table entry;
            // without them -- the columns are not an SBWT's -- the image steps down like one that does not fit)
            if (big_path && h.has_path && !(h.stab_pos || (h.n_sb2 > 0 && d->k > h.p_sparse))) { e = hipErrorOutOfMemory; break; }
            // substitution-safe bits of the path: need the whole k-mers in the sparse table
            // (31 < k <= 63: whole k-mers live in the two-level table -- the wide kernels, rule 2 only)
            const bool whole_kmers = h.p_sparse == d->k || (h.n_sb2 > 0 && d->k > h.p_sparse);
            if (h.has_path && whole_kmers && g_path_safe) {
                SbwtIndexView v2 = idx->view();
                // (rule 2 keeps the path heads' labels and their list in scratch)
                void *hscr = nullptr;
                const bool wide_safe = d->k > h.p_sparse;
                const size_t safe_bytes = (size_t)sbwt_path_safe_scratch_bytes(h.n_pos, (int)d->k);
                const bool hscr_in_arena = arena && arena_bytes >= safe_bytes;
                if (!hscr_in_arena) arena_drop();       // (too small to serve: its room may be what the allocation below needs)
                if (g_path_safe >= 2 || wide_safe) {
                    if (hscr_in_arena) hscr = arena;
                    else if (hipMalloc(&hscr, safe_bytes) != hipSuccess) {
                        (void)hipGetLastError();
                        hscr = nullptr;                 // no room for rule 2: the narrow rule needs none
                    }
                }
                // ... and hands the per-substitute verdicts to the transition table's negative entries through alt_safe (a byte
                // per position; without it only the steps that are safe for all three substitutes bridge)
                if (hscr && hipMalloc((void **)&alt_safe, (size_t)h.n_pos) != hipSuccess) {
                    (void)hipGetLastError();
                    alt_safe = nullptr;
                }
                sbwt_launch_path_safe(v2, reinterpret_cast<uint4 *>(idx->blob + h.off_pq), hscr ? (wide_safe ? 2 : g_path_safe) : 1, hscr, alt_safe, 0);
                e = hipDeviceSynchronize();
                const bool had_scratch = hscr != nullptr;
                if (hscr && !hscr_in_arena) (void)hipFree(hscr);
                if (e != hipSuccess) break;
                h.has_safe = (wide_safe && !had_scratch) ? 0 : 1;    // (no room for the head labels of long k-mers: no safe bits)
            }
        }
        plog("substitution-safe bits");
        if (h.has_path) {
            // Last: the only-successor bits, the path groups' final encoding, and the transition table (its room was made when
            // the image moved into its final allocation, behind the path order).
            SbwtIndexView v3 = idx->view();
            long long nb = 0;
            const long long n_ent = sbwt_launch_path_oth(v3, reinterpret_cast<uint4 *>(idx->blob + h.off_pq), &nb, 0);
            if (n_ent != h.n_trans) { e = hipErrorUnknown; break; }         // (the count the layout was made for)
            const int64_t n_slots = h.n_tslots;
            plog("only-successor bits");
            sbwt_launch_trans_insert(idx->view(), reinterpret_cast<uint4 *>(idx->blob + h.off_trans), n_slots, alt_safe, 0);
            if ((e = hipDeviceSynchronize()) != hipSuccess) break;
            h.n_paths = sbwt_count_paths(idx->view(), 0);
            if (h.n_paths < 0) { e = hipErrorUnknown; break; }
            plog("transition table");
        }
        if ((e = hipGetLastError()) != hipSuccess) break;
        e = hipDeviceSynchronize();
    } while (0);
    arena_drop();
    if (alt_safe) (void)hipFree(alt_safe);
    if (e == hipErrorOutOfMemory && level < 2 && (h.has_path || h.p_sparse > 0)) {
        (void)hipGetLastError();                       // scratch of a derived structure did not fit: build without them
        (void)hipFree(idx->blob);
        delete idx;
        return retry_next_level();
    }
    if (e != hipSuccess) {
        rc = fail(SBWTGPU_ERR_HIP, "building the index image: %s", hipGetErrorString(e));
        (void)hipFree(idx->blob);
        delete idx;
        return rc;
    }
    *out = idx;
    return SBWTGPU_OK;
}

void sbwtgpu_index_destroy(sbwtgpu_index *idx) {
    if (!idx) return;
    if (idx->blob && idx->owns_blob) {
        DeviceGuard guard(idx->device);
        (void)hipFree(idx->blob);
    }
    delete idx;
}

int sbwtgpu_index_get_info(const sbwtgpu_index *idx, sbwtgpu_index_info *info) {
    if (!idx || !info) return fail(SBWTGPU_ERR_INVALID_ARG, "idx/info is NULL");
    info->n_nodes = idx->h.n_nodes;
    info->n_kmers = idx->h.n_kmers;
    info->k = idx->h.k;
    info->precalc_k = idx->h.p_file;
    for (int i = 0; i < 4; i++) info->C[i] = idx->h.C[i];
    info->has_streaming_support = idx->h.has_ssup;
    info->device = idx->device;
    info->device_precalc_k = idx->h.p_dev;
    info->blob_bytes = idx->h.blob_bytes;
    info->image_level = idx->h.image_level;
    info->n_paths = idx->h.n_paths;
    info->n_branch = idx->h.n_branch;
    info->default_search_variant = !idx->h.has_path ? 1 : auto_variant(idx->h);
    return SBWTGPU_OK;
}

int sbwtgpu_index_get_precalc(const sbwtgpu_index *idx, int64_t *out_pairs) {
    if (!idx || !out_pairs) return fail(SBWTGPU_ERR_INVALID_ARG, "idx/out is NULL");
    if (idx->h.p_file == 0) return SBWTGPU_OK;
    DeviceGuard guard(idx->device);
    HIP_TRY(hipMemcpy(out_pairs, idx->blob + idx->h.off_ftab, (size_t)16 << (2 * idx->h.p_file),
                      hipMemcpyDeviceToHost));
    return SBWTGPU_OK;
}

// ---- replication ---------------------------------------------------------------------------
int sbwtgpu_index_export_header(const sbwtgpu_index *idx, void *header_out, int64_t cap, int64_t *bytes) {
    if (!idx || !bytes) return fail(SBWTGPU_ERR_INVALID_ARG, "idx/bytes is NULL");
    *bytes = (int64_t)sizeof(SbwtBlobHeader);
    if (!header_out) return SBWTGPU_OK;   // size query
    if (cap < (int64_t)sizeof(SbwtBlobHeader)) return fail(SBWTGPU_ERR_INVALID_ARG, "header buffer too small");
    memcpy(header_out, &idx->h, sizeof(SbwtBlobHeader));
    return SBWTGPU_OK;
}

int sbwtgpu_index_blob(const sbwtgpu_index *idx, void **dev_ptr, int64_t *bytes) {
    if (!idx || !dev_ptr || !bytes) return fail(SBWTGPU_ERR_INVALID_ARG, "NULL argument");
    *dev_ptr = idx->blob;
    *bytes = idx->h.blob_bytes;
    return SBWTGPU_OK;
}

int sbwtgpu_index_copy_blob(const sbwtgpu_index *idx, void *dst_dev, int64_t bytes, void *stream) {
    if (!idx || !dst_dev) return fail(SBWTGPU_ERR_INVALID_ARG, "NULL argument");
    if (bytes != idx->h.blob_bytes) return fail(SBWTGPU_ERR_INVALID_ARG, "bytes must equal blob_bytes");
    DeviceGuard guard(idx->device);
    HIP_TRY(hipMemcpyAsync(dst_dev, idx->blob, (size_t)bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    return SBWTGPU_OK;
}

int sbwtgpu_index_adopt(const void *header, int64_t header_bytes, void *dev_blob, int64_t blob_bytes, int device,
                        sbwtgpu_index **out) {
    if (!header || !dev_blob || !out) return fail(SBWTGPU_ERR_INVALID_ARG, "NULL argument");
    if (header_bytes != (int64_t)sizeof(SbwtBlobHeader)) return fail(SBWTGPU_ERR_INVALID_ARG, "bad header size");
    SbwtBlobHeader h;
    memcpy(&h, header, sizeof(h));
    if (h.magic != SBWT_BLOB_MAGIC) return fail(SBWTGPU_ERR_INVALID_ARG, "bad header magic");
    if (h.blob_bytes != blob_bytes) return fail(SBWTGPU_ERR_INVALID_ARG, "blob size does not match its header");
    if (((uintptr_t)dev_blob & 255) != 0) return fail(SBWTGPU_ERR_INVALID_ARG, "the device image must be 256-byte aligned");
    sbwtgpu_index *idx = new (std::nothrow) sbwtgpu_index();
    if (!idx) return fail(SBWTGPU_ERR_OOM, "out of host memory");
    idx->h = h;
    idx->device = device;
    idx->blob = static_cast<char *>(dev_blob);
    idx->owns_blob = false;
    *out = idx;
    return SBWTGPU_OK;
}

// RCCL is loaded lazily so that single-GPU use never needs it.
// devs[] may name a device more than once (several host threads per GPU): the image travels once per DISTINCT
// device and the duplicates share that device's handle.
// The plan of a replication, without touching any device (so that a box without GPUs can test it): the DISTINCT devices of
// devs[] in order of first appearance (uniq_out, n_dev entries of room), slot_out[i] = the index of devs[i] among them, and
// the root device's index (= its rank in the RCCL group, whose ranks are the distinct devices in that order).
int sbwtgpu_debug_bcast_plan(int n_dev, const int *devs, int root_device, int n_visible, int *uniq_out, int *slot_out, int *n_uniq,
                             int *root_rank) {
    if (n_dev <= 0 || !devs || !uniq_out || !slot_out || !n_uniq || !root_rank) return fail(SBWTGPU_ERR_INVALID_ARG, "NULL/empty argument");
    int nu = 0;
    for (int i = 0; i < n_dev; i++) {
        if (devs[i] < 0 || devs[i] >= n_visible) return fail(SBWTGPU_ERR_NO_DEVICE, "device %d out of range", devs[i]);
        int u = 0;
        while (u < nu && uniq_out[u] != devs[i]) u++;
        if (u == nu) uniq_out[nu++] = devs[i];
        slot_out[i] = u;
    }
    *n_uniq = nu;
    *root_rank = -1;
    for (int u = 0; u < nu; u++)
        if (uniq_out[u] == root_device) *root_rank = u;
    if (*root_rank < 0) return fail(SBWTGPU_ERR_INVALID_ARG, "the root's device must be in devs[]");
    return SBWTGPU_OK;
}

int sbwtgpu_index_bcast(sbwtgpu_index *root, int n_dev, const int *devs, sbwtgpu_index **out) {
    if (!root || n_dev <= 0 || !devs || !out) return fail(SBWTGPU_ERR_INVALID_ARG, "NULL/empty argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) ndev = 0;
    std::vector<int> uniq((size_t)n_dev, 0);            // distinct devices, in order of first appearance
    std::vector<int> slot((size_t)n_dev, 0);            // devs[i] -> its index in uniq
    int n_uniq = 0, root_rank = -1;
    {
        const int prc = sbwtgpu_debug_bcast_plan(n_dev, devs, root->device, ndev, uniq.data(), slot.data(), &n_uniq, &root_rank);
        if (prc != SBWTGPU_OK) return prc;
    }
    // SBWTGPU_BCAST_NO_DEDUP=1 (tests): every entry of devs[] is a rank of its own, duplicates included -- the RCCL call
    // sequence below then runs on a one-GPU box (with a stand-in library behind SBWTGPU_RCCL_LIB: RCCL itself refuses two
    // ranks on one device); the root's rank is its device's first entry
    const char *nd_env = getenv("SBWTGPU_BCAST_NO_DEDUP");
    if (nd_env && *nd_env == '1' && n_dev > 1) {
        n_uniq = n_dev;
        root_rank = -1;
        for (int i = 0; i < n_dev; i++) {
            uniq[(size_t)i] = devs[i];
            slot[(size_t)i] = i;
            if (root_rank < 0 && devs[i] == root->device) root_rank = i;
        }
    }
    uniq.resize((size_t)n_uniq);
    if (uniq.size() == 1) {
        for (int i = 0; i < n_dev; i++) out[i] = root;
        return SBWTGPU_OK;
    }
    typedef void *comm_t;
    typedef int (*init_all_t)(comm_t *, int, const int *);
    typedef int (*bcast_t)(const void *, void *, size_t, int, int, comm_t, hipStream_t);
    typedef int (*grp_t)(void);
    typedef int (*destroy_t)(comm_t);
    // (SBWTGPU_RCCL_LIB names another library with RCCL's five entry points: a site's own build, or a stand-in for tests)
    const char *rccl_name = getenv("SBWTGPU_RCCL_LIB");
    void *lib = (rccl_name && *rccl_name) ? dlopen(rccl_name, RTLD_NOW | RTLD_GLOBAL) : dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib && !(rccl_name && *rccl_name)) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return fail(SBWTGPU_ERR_HIP, "cannot load librccl.so: %s", dlerror());
    init_all_t init_all = (init_all_t)dlsym(lib, "ncclCommInitAll");
    bcast_t bcast = (bcast_t)dlsym(lib, "ncclBroadcast");
    grp_t gstart = (grp_t)dlsym(lib, "ncclGroupStart"), gend = (grp_t)dlsym(lib, "ncclGroupEnd");
    destroy_t destroy = (destroy_t)dlsym(lib, "ncclCommDestroy");
    if (!init_all || !bcast || !gstart || !gend || !destroy) {
        dlclose(lib);
        return fail(SBWTGPU_ERR_HIP, "RCCL symbols missing");
    }
    const int nu = (int)uniq.size();
    std::vector<comm_t> comms((size_t)nu, nullptr);
    std::vector<hipStream_t> streams((size_t)nu, nullptr);
    std::vector<char> have_stream((size_t)nu, 0);
    std::vector<sbwtgpu_index *> made((size_t)nu, nullptr);
    bool have_comms = false;
    int rc = SBWTGPU_OK;
    for (int u = 0; u < nu && rc == SBWTGPU_OK; u++) {
        if (hipSetDevice(uniq[u]) != hipSuccess || hipStreamCreate(&streams[u]) != hipSuccess) {
            rc = fail(SBWTGPU_ERR_HIP, "cannot set up device %d", uniq[u]);
            break;
        }
        have_stream[u] = 1;
        if (u != root_rank) {
            sbwtgpu_index *c = new (std::nothrow) sbwtgpu_index();
            if (!c || hipMalloc((void **)&c->blob, (size_t)root->h.blob_bytes) != hipSuccess) {
                delete c;
                rc = fail(SBWTGPU_ERR_OOM, "hipMalloc on device %d", uniq[u]);
                break;
            }
            c->h = root->h;
            c->device = uniq[u];
            made[u] = c;
        }
    }
    if (rc == SBWTGPU_OK) {
        if (init_all(comms.data(), nu, uniq.data()) != 0) rc = fail(SBWTGPU_ERR_HIP, "ncclCommInitAll failed");
        else have_comms = true;
    }
    if (rc == SBWTGPU_OK) {
        gstart();
        for (int u = 0; u < nu; u++) {
            (void)hipSetDevice(uniq[u]);
            void *buf = (u == root_rank) ? (void *)root->blob : (void *)made[u]->blob;
            if (bcast(buf, buf, (size_t)root->h.blob_bytes, /*ncclChar*/ 0, root_rank, comms[u], streams[u]) != 0)
                rc = fail(SBWTGPU_ERR_HIP, "ncclBroadcast failed");
        }
        if (gend() != 0 && rc == SBWTGPU_OK) rc = fail(SBWTGPU_ERR_HIP, "ncclGroupEnd failed");
        for (int u = 0; u < nu; u++) {
            (void)hipSetDevice(uniq[u]);
            if (hipStreamSynchronize(streams[u]) != hipSuccess && rc == SBWTGPU_OK)
                rc = fail(SBWTGPU_ERR_HIP, "broadcast to device %d failed", uniq[u]);
        }
    }
    for (int u = 0; u < nu; u++) {                       // only what was actually created
        if (!have_stream[u] && !have_comms) continue;
        (void)hipSetDevice(uniq[u]);
        if (have_comms) destroy(comms[u]);
        if (have_stream[u]) (void)hipStreamDestroy(streams[u]);
    }
    (void)hipSetDevice(root->device);
    if (rc != SBWTGPU_OK) {
        for (auto *c : made) sbwtgpu_index_destroy(c);
        dlclose(lib);
        return rc;
    }
    // success: librccl stays loaded for the life of the process (it keeps per-process state)
    for (int i = 0; i < n_dev; i++) out[i] = (slot[(size_t)i] == root_rank) ? root : made[(size_t)slot[(size_t)i]];
    return SBWTGPU_OK;
}

// ---- device-pointer entry points -----------------------------------------------------------
// Workspace = header | packed bases (16 bytes per 32 bases) | with "sort_reads" = 1, room to sort the reads by their place
// in the path order (sbwt_sort.hip): four 32-bit arrays of one entry per read + the radix sort's own temporary, sized
// for reads of >= 32 bases on average (a batch of shorter reads is searched unsorted).
static inline int64_t ws_packed_bytes(int64_t total_bases) {
    // + 2 groups that the encoder zero-fills (windows that run past the last base) + 2 that are only ever loaded (the
    // search kernel fetches the packed groups of a read ahead of their use, three at a time)
    const int64_t groups = (total_bases + SBWT_GROUP_BASES - 1) / SBWT_GROUP_BASES + 4;
    return (int64_t)sizeof(SbwtWorkHeader) + groups * 16;
}
static inline int64_t ws_sort_capacity(int64_t total_bases) {      // only when sorting is switched on ("sort_reads" = 1)
    return g_sort_reads > 0 ? total_bases / 32 + 4096 : 0;
}
// the fused route's list of reads handed on to the general kernel: one 32-bit entry per read, reads of >= 32 bases
static inline int64_t ws_defer_bytes(int64_t total_bases) { return align256(total_bases / 8 + 256); }
// the pieces of long reads (SbwtPieceTab, sbwt_device.h): two 16-byte entries per zone.  Zones of 128 k-mers while that
// makes at most 2^18 of them (a few genomes as single reads then fill the chip), doubling up to SBWT_PIECE beyond.
static inline int ws_piece_len(int64_t total_bases) {
    int piece = 128;
    while (piece < SBWT_PIECE && total_bases / piece > ((int64_t)1 << 18)) piece *= 2;
    return piece;
}
// (entries: non-decreasing in total_bases -- a workspace sized for a large batch serves every smaller one)
static inline int64_t ws_piece_cap(int64_t total_bases) {
    return std::max<int64_t>(total_bases / SBWT_PIECE, std::min<int64_t>(total_bases / 128, (int64_t)1 << 18)) + 2;
}
static inline int64_t ws_piece_bytes(int64_t total_bases) { return align256(ws_piece_cap(total_bases) * 32); }
// the fused route's ticket table (SbwtTickTab, round 6): one ticket per 64 bases of the batch at most (a batch of long reads has
// one per 98 .. 144 bases; a batch that would need more goes the general route), 16 + 4 bytes each, and one bit per read of the
// list of reads handed on (one read per 32 bases, like that list)
static inline int64_t ws_tick_cap(int64_t total_bases) { return total_bases / 64 + 64; }
static inline int64_t ws_tick_bytes(int64_t total_bases) {
    return align256(ws_tick_cap(total_bases) * 16) + align256(ws_tick_cap(total_bases) * 4) + align256(total_bases / 256 + 64);
}
static inline int64_t ws_fixed_bytes(int64_t total_bases) {
    return align256(ws_packed_bytes(total_bases)) + ws_defer_bytes(total_bases) + ws_piece_bytes(total_bases) + ws_tick_bytes(total_bases);
}
int64_t sbwtgpu_search_workspace_bytes(int64_t total_bases) {
    if (total_bases < 0) total_bases = 0;
    const int64_t n_cap = ws_sort_capacity(total_bases);
    return ws_fixed_bytes(total_bases) + (n_cap ? 32 * n_cap + ((int64_t)16 << 20) : 0);
}
static SbwtPieceTab ws_piece_tab(void *d_ws, int64_t total_bases) {
    SbwtPieceTab pt;
    if (!g_split_long) return pt;
    char *base = static_cast<char *>(d_ws) + align256(ws_packed_bytes(total_bases)) + ws_defer_bytes(total_bases);
    pt.cap = ws_piece_cap(total_bases);
    pt.piece = ws_piece_len(total_bases);
    pt.pairs = reinterpret_cast<uint4 *>(base);
    pt.outs = pt.pairs + pt.cap;
    return pt;
}

static SbwtTickTab ws_tick_tab(void *d_ws, int64_t total_bases) {
    SbwtTickTab tt;
    char *base = static_cast<char *>(d_ws) + align256(ws_packed_bytes(total_bases)) + ws_defer_bytes(total_bases) + ws_piece_bytes(total_bases);
    tt.cap = ws_tick_cap(total_bases);
    tt.tick = reinterpret_cast<uint4 *>(base);
    tt.tick_read = reinterpret_cast<unsigned *>(base + align256(tt.cap * 16));
    tt.defer_bits = reinterpret_cast<unsigned *>(base + align256(tt.cap * 16) + align256(tt.cap * 4));
    return tt;
}

static const char *RANK_ONLY_MSG =
    "the index columns are not a valid SBWT (set bits != n_nodes - 1): only rank() is available";

// SBWT::search through streaming steps on the device (the marks given with the index, or derived): exact where a streaming step
// and a full search agree on every k-mer of valid bases (tests/test_large.hh:104-115).  Not for k = 1: there every column is in
// the root's suffix group while the edges sit on the first k-mer's column (NodeBOSSInMemoryConstructor.hh:98-154), so the
// reference's own streaming_search misses every second 1-mer that search() finds (round 5, tests/test_gpu_corners.py).
static bool internal_streaming_ok(const SbwtBlobHeader &h) {
    return (h.has_ssup || h.ssup_derived) && g_derive_ssup && h.k >= 2;
}

static int search_dev_check(const sbwtgpu_index *idx, int64_t total_bases, int64_t n_reads, const void *d_ws,
                            int64_t ws_bytes, int streaming) {
    if (!idx) return fail(SBWTGPU_ERR_INVALID_ARG, "idx is NULL");
    if (idx->h.rank_only) return fail(SBWTGPU_ERR_INVALID_ARG, "%s", RANK_ONLY_MSG);
    if (streaming && !idx->h.has_ssup)
        return fail(SBWTGPU_ERR_NO_STREAMING, "Error: streaming search support not built");
    if (n_reads < 0 || total_bases < 0) return fail(SBWTGPU_ERR_INVALID_ARG, "negative size");
    if (total_bases >= ((int64_t)1 << 36))
        return fail(SBWTGPU_ERR_INVALID_ARG, "more than 2^36 bases in one call: split the batch");
    if (!d_ws) return fail(SBWTGPU_ERR_INVALID_ARG, "NULL workspace");
    if (ws_bytes < sbwtgpu_search_workspace_bytes(total_bases))
        return fail(SBWTGPU_ERR_INVALID_ARG, "workspace too small (%lld < %lld)", (long long)ws_bytes,
                    (long long)sbwtgpu_search_workspace_bytes(total_bases));
    if (((uintptr_t)d_ws & 15) != 0) return fail(SBWTGPU_ERR_INVALID_ARG, "workspace must be 16-byte aligned");
    return SBWTGPU_OK;
}

int sbwtgpu_encode_bases_dev(const sbwtgpu_index *idx, const char *d_bases, int64_t total_bases, void *d_ws,
                             int64_t ws_bytes, void *stream) {
    int rc = search_dev_check(idx, total_bases, 0, d_ws, ws_bytes, 0);
    if (rc != SBWTGPU_OK) return rc;
    if (total_bases > 0 && !d_bases) return fail(SBWTGPU_ERR_INVALID_ARG, "NULL device pointer");
    DeviceGuard guard(idx->device);
    SbwtWorkHeader *ws = static_cast<SbwtWorkHeader *>(d_ws);
    uint4 *packed = reinterpret_cast<uint4 *>(static_cast<char *>(d_ws) + sizeof(SbwtWorkHeader));
    sbwt_launch_encode(d_bases, total_bases, packed, ws, static_cast<hipStream_t>(stream));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SBWTGPU_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
    return SBWTGPU_OK;
}

int sbwtgpu_search_encoded_dev(const sbwtgpu_index *idx, int64_t total_bases, const int64_t *d_read_off,
                               int64_t n_reads, int64_t *d_out, const int64_t *d_out_off, void *d_ws,
                               int64_t ws_bytes, int streaming, void *stream) {
    int rc = search_dev_check(idx, total_bases, n_reads, d_ws, ws_bytes, streaming);
    if (rc != SBWTGPU_OK) return rc;
    if (n_reads == 0) return SBWTGPU_OK;
    if (!d_read_off || !d_out_off) return fail(SBWTGPU_ERR_INVALID_ARG, "NULL device pointer");
    DeviceGuard guard(idx->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    SbwtWorkHeader *ws = static_cast<SbwtWorkHeader *>(d_ws);
    // the ticket counter and the work counters restart with every search launch
    hipError_t e = hipMemsetAsync(ws, 0, sizeof(SbwtWorkHeader), st);
    if (e != hipSuccess) return fail(SBWTGPU_ERR_HIP, "hipMemsetAsync: %s", hipGetErrorString(e));
    const uint4 *packed = reinterpret_cast<const uint4 *>(static_cast<char *>(d_ws) + sizeof(SbwtWorkHeader));
    if (g_poison && n_reads > 0) {
        int64_t ends[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(&ends[0], d_out_off, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&ends[1], d_out_off + n_reads, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (ends[1] > ends[0]) {
                    const size_t vb = t_out32 ? 4 : 8;
                    HIP_TRY(hipMemsetAsync(reinterpret_cast<char *>(d_out) + (size_t)ends[0] * vb, 0xA5, (size_t)(ends[1] - ends[0]) * vb, st));
                }
    }
    // kernel: an explicit choice ("search_variant" / SBWTGPU_SEARCH_VARIANT), else by the index: the segment-list writer
    // where reads follow their paths for long (fewer than one column in 64 offers a choice of successors: config 2 has one
    // in 107, config 5 one in 586), the staged writer on branchy indexes (the star pan-genome of config 3: one in 31 --
    // its reads leave their path every 3-4 k-mers, and short segments fill the lists: 146.4 vs 143.8 ms)
    int variant = g_variant_override >= 0 ? g_variant_override : tuning_variant();
    if (variant < 0) variant = idx->h.has_path ? auto_variant(idx->h) : 2;
    if (variant == 5 || variant == 2 || variant == 3) variant = 4;   // already-encoded bases: the general path kernel
    const int eff_streaming = (!streaming && internal_streaming_ok(idx->h)) ? 2 : streaming;
    // reads sorted by their place in the path order (sbwt_sort.hip): when the batch covers the index a few times
    void *sort_scratch = nullptr;
    long long sort_bytes = 0;
    int key_bits = 1;
    while (key_bits < 32 && ((int64_t)1 << key_bits) <= idx->h.n_nodes) key_bits++;
    const bool path_kernel = variant == 4 && idx->h.has_path && eff_streaming && idx->h.stab_pos &&
                             idx->h.n_nodes < ((int64_t)1 << 31) - 128 && n_reads < ((int64_t)1 << 31);
    // Off unless asked for ("sort_reads" = 1): the path order numbers its paths in column order, not along the genome, so
    // reads sorted by path position share lines only within one path (kernel 6.45 -> 6.16 ms on config 2) and the
    // pre-pass costs 0.7 ms; reads that arrive in genome order get the full effect (5.3 ms) without any pre-pass.
    const bool want_sort = path_kernel && g_sort_reads > 0;
    if (want_sort) {
        // the scratch lives in the caller's workspace, behind the packed bases (no allocation, nothing shared between calls)
        const int64_t off = ws_fixed_bytes(total_bases);
        sort_bytes = sbwt_sort_scratch_bytes(n_reads, key_bits);
        if (n_reads <= ws_sort_capacity(total_bases) && off + sort_bytes <= ws_bytes) sort_scratch = static_cast<char *>(d_ws) + off;
    }
    sbwt_launch_search(idx->view(), packed, reinterpret_cast<const long long *>(d_read_off),
                       reinterpret_cast<const long long *>(d_out_off), reinterpret_cast<long long *>(d_out), n_reads,
                       ws, eff_streaming, st, variant, total_bases / SBWT_GROUP_BASES + 2, sort_scratch, sort_bytes, key_bits,
                       ws_piece_tab(d_ws, total_bases));
    e = hipGetLastError();
    if (e != hipSuccess) return fail(SBWTGPU_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
    return SBWTGPU_OK;
}

// HIP events around the dominant kernel of the search calls ("kernel_events" = 1; bench.py's roofline leg): a ring of
// event pairs, so that a timed loop of launches needs no synchronisation in between
static const int EV_RING = 256;
static hipEvent_t g_ev[EV_RING][2];
static int g_ev_made = 0;

int sbwtgpu_kernel_times(double *ms, int64_t cap, int64_t *n) {
    if (!n || (cap > 0 && !ms)) return fail(SBWTGPU_ERR_INVALID_ARG, "NULL argument");
    const int64_t have = g_ev_count < EV_RING ? g_ev_count : EV_RING;
    const int64_t first = g_ev_count - have;
    int64_t w = 0;
    for (int64_t q = first; q < g_ev_count && w < cap; q++, w++) {
        hipEvent_t *e = g_ev[q % EV_RING];
        HIP_TRY(hipEventSynchronize(e[1]));
        float f = 0;
        HIP_TRY(hipEventElapsedTime(&f, e[0], e[1]));
        ms[w] = (double)f;
    }
    *n = w;
    return SBWTGPU_OK;
}

static int search_dev_common(const sbwtgpu_index *idx, const char *d_bases, int64_t total_bases,
                             const int64_t *d_read_off, int64_t n_reads, int64_t *d_out, const int64_t *d_out_off,
                             void *d_ws, int64_t ws_bytes, void *stream, int streaming) {
    int rc = search_dev_check(idx, total_bases, n_reads, d_ws, ws_bytes, streaming);
    if (rc != SBWTGPU_OK) return rc;
    if (n_reads == 0) return SBWTGPU_OK;
    {
        // The fused route (sbwt_search_fused.hip; "search_variant" 5, the default on a path-order index): no separate
        // encode pass.  It decides on the device whether the batch qualifies (reads of one length, 32 .. 160 bases) and
        // runs the general kernel behind the fused one for everything that does not.
        int variant = g_variant_override >= 0 ? g_variant_override : tuning_variant();
        if (variant < 0) variant = idx->h.has_path ? auto_variant(idx->h) : 2;
        const int eff_streaming = (!streaming && internal_streaming_ok(idx->h)) ? 2 : streaming;
        // (an image of 2^31 columns or more has a path order only in the form the fused kernel's BIG instantiation reads:
        // k <= 31, whole k-mers with their positions; int64 results -- the int32 calls refuse such an index)
        const bool big_image = idx->h.big_layout && idx->h.has_path && !t_out32 &&
                               ((idx->h.stab_pos && idx->h.p_sparse == idx->h.k) || (idx->h.n_sb2 > 0 && idx->h.k > idx->h.p_sparse));
        const bool path_kernel = idx->h.has_path && eff_streaming &&
                                 ((idx->h.n_nodes < ((int64_t)1 << 31) - 128 && !idx->h.big_layout) || big_image) && n_reads < ((int64_t)1 << 31) &&
                                 total_bases / SBWT_GROUP_BASES + 2 < ((int64_t)1 << 31) - 4;
        if (variant == 5 && path_kernel && g_sort_reads <= 0 && !(g_debug & 16)) {
            if (!d_read_off || !d_out_off) return fail(SBWTGPU_ERR_INVALID_ARG, "NULL device pointer");
            if (total_bases > 0 && !d_bases) return fail(SBWTGPU_ERR_INVALID_ARG, "NULL device pointer");
            DeviceGuard guard(idx->device);
            hipStream_t st = static_cast<hipStream_t>(stream);
            SbwtWorkHeader *ws = static_cast<SbwtWorkHeader *>(d_ws);
            // (all but the header's last word: the hint the call before left for this one, SbwtWorkHeader::hint -- the only
            // part of a workspace that is read before it is written)
            HIP_TRY(hipMemsetAsync(ws, 0, SBWT_WS_CLEAR_BYTES, st));
            if (g_poison) {
                int64_t ends[2] = {0, 0};
                HIP_TRY(hipMemcpyAsync(&ends[0], d_out_off, 8, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipMemcpyAsync(&ends[1], d_out_off + n_reads, 8, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                if (ends[1] > ends[0]) {
                    const size_t vb = t_out32 ? 4 : 8;
                    HIP_TRY(hipMemsetAsync(reinterpret_cast<char *>(d_out) + (size_t)ends[0] * vb, 0xA5, (size_t)(ends[1] - ends[0]) * vb, st));
                }
            }
            uint4 *packed = reinterpret_cast<uint4 *>(static_cast<char *>(d_ws) + sizeof(SbwtWorkHeader));
            unsigned *defer = reinterpret_cast<unsigned *>(static_cast<char *>(d_ws) + align256(ws_packed_bytes(total_bases)));
            const SbwtTickTab tt_of_call = ws_tick_tab(d_ws, total_bases);
            // (the bits of the reads handed on: cleared with the header)
            HIP_TRY(hipMemsetAsync(tt_of_call.defer_bits, 0, (size_t)(total_bases / 256 + 64), st));
            SbwtTickTab tt_for_launch = tt_of_call;
            if (!(g_fused_table && g_fused_ragged && n_reads <= total_bases / 32)) { tt_for_launch.tick = nullptr; tt_for_launch.tick_read = nullptr; }
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (g_kernel_events) {
                const int slot = (int)(g_ev_count % EV_RING);
                if (slot >= g_ev_made) {
                    HIP_TRY(hipEventCreate(&g_ev[slot][0]));
                    HIP_TRY(hipEventCreate(&g_ev[slot][1]));
                    g_ev_made = slot + 1;
                }
                e0 = g_ev[slot][0]; e1 = g_ev[slot][1];
                g_ev_count++;
            }
            sbwt_launch_search_fused(idx->view(), d_bases, total_bases, packed, reinterpret_cast<const long long *>(d_read_off),
                                     reinterpret_cast<const long long *>(d_out_off), reinterpret_cast<long long *>(d_out), n_reads,
                                     ws, eff_streaming, st, defer, e0, e1, ws_piece_tab(d_ws, total_bases),
                                     // (the list of reads handed on has one entry per 32 bases of the batch)
                                     ((g_fused_ragged && n_reads <= total_bases / 32) ? 1 : 0) |
                                         ((g_fused_pieces > 0 ? g_fused_pieces : 3) << 8),
                                     // (the ticket table of batches with many long reads; "fused_table" 0 switches it off)
                                     tt_for_launch);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return fail(SBWTGPU_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
            return SBWTGPU_OK;
        }
    }
    rc = sbwtgpu_encode_bases_dev(idx, d_bases, total_bases, d_ws, ws_bytes, stream);
    if (rc != SBWTGPU_OK) return rc;
    return sbwtgpu_search_encoded_dev(idx, total_bases, d_read_off, n_reads, d_out, d_out_off, d_ws, ws_bytes,
                                      streaming, stream);
}

int sbwtgpu_streaming_search_dev(const sbwtgpu_index *idx, const char *d_bases, int64_t total_bases,
                                 const int64_t *d_read_off, int64_t n_reads, int64_t *d_out,
                                 const int64_t *d_out_off, void *d_ws, int64_t ws_bytes, void *stream) {
    return search_dev_common(idx, d_bases, total_bases, d_read_off, n_reads, d_out, d_out_off, d_ws, ws_bytes, stream,
                             1);
}

int sbwtgpu_search_dev(const sbwtgpu_index *idx, const char *d_bases, int64_t total_bases, const int64_t *d_read_off,
                       int64_t n_reads, int64_t *d_out, const int64_t *d_out_off, void *d_ws, int64_t ws_bytes,
                       void *stream) {
    return search_dev_common(idx, d_bases, total_bases, d_read_off, n_reads, d_out, d_out_off, d_ws, ws_bytes, stream,
                             0);
}

// The same two calls with int32 results (result compaction): every kernel of the route writes 4 bytes per k-mer
// instead of 8 -- half the write requests of a launch, which are 40 % of its time.
static int search_dev_i32(const sbwtgpu_index *idx, const char *d_bases, int64_t total_bases, const int64_t *d_read_off,
                          int64_t n_reads, int32_t *d_out, const int64_t *d_out_off, void *d_ws, int64_t ws_bytes, void *stream,
                          int streaming) {
    if (!idx) return fail(SBWTGPU_ERR_INVALID_ARG, "idx is NULL");
    if (idx->h.n_nodes >= ((int64_t)1 << 31))
        return fail(SBWTGPU_ERR_INVALID_ARG, "int32 results need an index of fewer than 2^31 columns (this one has %lld)",
                    (long long)idx->h.n_nodes);
    t_out32 = 1;
    const int rc = search_dev_common(idx, d_bases, total_bases, d_read_off, n_reads, reinterpret_cast<int64_t *>(d_out), d_out_off,
                                     d_ws, ws_bytes, stream, streaming);
    t_out32 = 0;
    return rc;
}
int sbwtgpu_streaming_search_dev_i32(const sbwtgpu_index *idx, const char *d_bases, int64_t total_bases,
                                     const int64_t *d_read_off, int64_t n_reads, int32_t *d_out, const int64_t *d_out_off,
                                     void *d_ws, int64_t ws_bytes, void *stream) {
    return search_dev_i32(idx, d_bases, total_bases, d_read_off, n_reads, d_out, d_out_off, d_ws, ws_bytes, stream, 1);
}
int sbwtgpu_search_dev_i32(const sbwtgpu_index *idx, const char *d_bases, int64_t total_bases, const int64_t *d_read_off,
                           int64_t n_reads, int32_t *d_out, const int64_t *d_out_off, void *d_ws, int64_t ws_bytes,
                           void *stream) {
    return search_dev_i32(idx, d_bases, total_bases, d_read_off, n_reads, d_out, d_out_off, d_ws, ws_bytes, stream, 0);
}

int sbwtgpu_rank_dev(const sbwtgpu_index *idx, const int64_t *d_pos, const char *d_sym, int64_t n, int64_t *d_out,
                     void *stream) {
    if (!idx) return fail(SBWTGPU_ERR_INVALID_ARG, "idx is NULL");
    if (n < 0) return fail(SBWTGPU_ERR_INVALID_ARG, "negative size");
    if (n == 0) return SBWTGPU_OK;
    if (!d_pos || !d_sym || !d_out) return fail(SBWTGPU_ERR_INVALID_ARG, "NULL device pointer");
    DeviceGuard guard(idx->device);
    sbwt_launch_rank(idx->view(), reinterpret_cast<const long long *>(d_pos), d_sym, n,
                     reinterpret_cast<long long *>(d_out), static_cast<hipStream_t>(stream));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SBWTGPU_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
    return SBWTGPU_OK;
}

int sbwtgpu_workspace_status(const void *d_ws, void *stream, int *status) {
    if (!d_ws || !status) return fail(SBWTGPU_ERR_INVALID_ARG, "NULL argument");
    SbwtWorkHeader hdr;
    HIP_TRY(hipMemcpyAsync(&hdr, d_ws, sizeof(hdr), hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    *status = hdr.status;
    return SBWTGPU_OK;
}

int sbwtgpu_workspace_stats(const void *d_ws, void *stream, int64_t stats[8]) {
    if (!d_ws || !stats) return fail(SBWTGPU_ERR_INVALID_ARG, "NULL argument");
    SbwtWorkHeader hdr;
    HIP_TRY(hipMemcpyAsync(&hdr, d_ws, sizeof(hdr), hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    stats[0] = (int64_t)hdr.n_stream;
    stats[1] = (int64_t)hdr.n_search;
    stats[2] = (int64_t)hdr.n_lf;
    stats[3] = (int64_t)hdr.n_tab_hit;
    stats[4] = (int64_t)hdr.n_ext;
    stats[5] = (int64_t)hdr.n_bridge;
    stats[6] = stats[7] = 0;
    return SBWTGPU_OK;
}

// ---- host-buffer entry points --------------------------------------------------------------
namespace {
struct Stream {
    hipStream_t s = nullptr;
    ~Stream() {
        if (s) (void)hipStreamDestroy(s);
    }
};

// Small calls (the scalar API of the reference -- SBWT::search(kmer), rank(pos, c), one update_sbwt_interval -- arrives
// here as batches of one): no hipMalloc / hipStreamCreate per call.  Every host thread keeps, per device, one stream,
// one pinned host buffer and one device buffer of SLOT_CAP bytes; a call whose data fits stages its inputs in the pinned
// buffer, moves them with ONE H2D copy, runs its kernels, and brings the results back with ONE D2H copy + one sync.
constexpr size_t SLOT_CAP = (size_t)1 << 20;
bool g_exiting = false;                                  // set at process exit: HIP may be gone, leak instead of freeing
struct SmallSlot {
    int device = -1;
    hipStream_t stream = nullptr;
    char *host = nullptr, *dev = nullptr;
    void release() {
        if (device < 0 || g_exiting) return;
        DeviceGuard guard(device);
        if (stream) (void)hipStreamDestroy(stream);
        if (host) (void)hipHostFree(host);
        if (dev) (void)hipFree(dev);
        stream = nullptr; host = dev = nullptr; device = -1;
    }
};
struct SlotSet {
    std::vector<SmallSlot> v;
    ~SlotSet() { for (auto &s : v) s.release(); }
};
thread_local SlotSet t_slots;
// The calling thread's slot on `device` (the current device must already be `device`), or nullptr when `need` does not
// fit or the buffers cannot be allocated (the caller then takes the general path).
SmallSlot *small_slot(int device, size_t need) {
    if (need > SLOT_CAP) return nullptr;
    static const bool once = [] { atexit([] { g_exiting = true; }); return true; }();
    (void)once;
    for (auto &s : t_slots.v)
        if (s.device == device) return &s;
    SmallSlot s;
    if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipHostMalloc((void **)&s.host, SLOT_CAP, hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void **)&s.dev, SLOT_CAP) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipStreamDestroy(s.stream);
        if (s.host) (void)hipHostFree(s.host);
        return nullptr;
    }
    s.device = device;
    t_slots.v.push_back(s);
    return &t_slots.v.back();
}
inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }
}  // namespace

// (*any_long: some read has more than 2 * PIECE k-mers -- the host entry points cut those into pieces, below)
static int check_reads(const int64_t *read_off, const int64_t *out_off, int64_t n_reads, int64_t k, bool *any_long = nullptr) {
    int64_t longest = 0;
    for (int64_t r = 0; r < n_reads; r++) {
        int64_t len = read_off[r + 1] - read_off[r];
        if (len < 0) return fail(SBWTGPU_ERR_INVALID_ARG, "read_off is not non-decreasing at read %lld", (long long)r);
        if (len >= ((int64_t)1 << 31)) return fail(SBWTGPU_ERR_READ_TOO_LONG, "read %lld has >= 2^31 bases", (long long)r);
        int64_t m = len - k + 1;
        if (m < 0) m = 0;
        if (out_off[r + 1] - out_off[r] != m)
            return fail(SBWTGPU_ERR_INVALID_ARG, "out_off[%lld+1]-out_off[%lld] must be max(0,len-k+1) = %lld",
                        (long long)r, (long long)r, (long long)m);
        if (m > longest) longest = m;
    }
    if (any_long) *any_long = longest > 4096;          // = 2 * PIECE
    return SBWTGPU_OK;
}


// ---- long reads -----------------------------------------------------------------------------------
// One lane walks one read, so a single very long query (a genome as one FASTA record) would keep one
// lane busy for millions of steps.  The host entry points therefore cut reads longer than 2*PIECE
// k-mers into pieces of about PIECE k-mers that overlap by k-1 bases and hand the pieces to the kernel
// as reads of their own; their result ranges tile the read's range, so nothing else changes.  This is
// exact: a piece starts at a k-mer whose window holds no lower-case acgt, and for such a k-mer the
// reference's result does not depend on what came before (streaming step and full search agree,
// tests/test_large.hh:104-115; only lower-case input makes the two differ).
static const int64_t PIECE = 2048;
static_assert(2 * PIECE == 4096, "check_reads flags reads of more than 4096 k-mers");

static inline bool window_has_lower(const char *s, int64_t k) {
    for (int64_t t = 0; t < k; t++) {
        char ch = s[t];
        if (ch == 'a' || ch == 'c' || ch == 'g' || ch == 't') return true;
    }
    return false;
}
// Appends the pieces of read `s` (len bases) to (dst bases, roff, ooff); dst may be NULL to count only.
// Returns the number of bytes appended.
static int64_t append_pieces(const char *s, int64_t len, int64_t k, char *dst, std::vector<int64_t> &roff,
                             std::vector<int64_t> &ooff) {
    const int64_t m = len - k + 1;
    int64_t written = 0;
    if (m <= 2 * PIECE) {
        if (dst && len > 0) memcpy(dst, s, (size_t)len);
        roff.push_back(roff.back() + len);
        ooff.push_back(ooff.back() + (m > 0 ? m : 0));
        return len;
    }
    int64_t start = 0;
    while (start < m) {
        int64_t next = start + PIECE;
        if (m - next < PIECE) next = m;                              // no tiny tail piece
        while (next < m && window_has_lower(s + next, k)) next++;    // a history-independent split point
        const int64_t nb = (next - start) + k - 1;
        if (dst) memcpy(dst + written, s + start, (size_t)nb);
        written += nb;
        roff.push_back(roff.back() + nb);
        ooff.push_back(ooff.back() + (next - start));
        start = next;
    }
    return written;
}
static inline int64_t pieces_bases_bound(int64_t len, int64_t k) {   // bytes append_pieces can write for a read
    const int64_t m = len - k + 1;
    if (m <= 2 * PIECE) return len > 0 ? len : 0;
    return len + (m / PIECE + 2) * (k - 1);
}

// ---- large host batches: chunks pipelined over two streams ------------------------------------------------------
// A batch whose results exceed PIPE_MIN bytes is cut into chunks of reads; chunk c+1's bases go up and its kernels run
// while chunk c's results come down (H2D and D2H use different DMA engines, the kernels are ~30x faster than either).
// Buffers of the caller that are pinned (hipHostMalloc / hipHostRegister / torch pin_memory: hipPointerGetAttributes says
// so) are the DMA's source and target themselves; pageable ones go through pinned staging buffers, copied by a few host
// threads.  The rate is then PCIe's: 8 bytes of results per k-mer.
namespace {
struct PipeSlot {
    hipStream_t st = nullptr;
    char *h_in = nullptr, *h_out = nullptr;     // pinned staging: bases | offsets; results (only for pageable callers)
    int *h_status = nullptr;
    char *d_mem = nullptr;
    int64_t cap_in = 0, cap_out = 0, cap_dev = 0;
    void release() {
        if (st) (void)hipStreamDestroy(st);
        if (h_in) (void)hipHostFree(h_in);
        if (h_out) (void)hipHostFree(h_out);
        if (h_status) (void)hipHostFree(h_status);
        if (d_mem) (void)hipFree(d_mem);
        *this = PipeSlot();
    }
};
std::mutex g_pipe_mutex;
struct ParkedPipe { int device; PipeSlot s[2]; };
std::vector<ParkedPipe> g_pipe_parked;
bool is_pinned(const void *p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}
void parallel_memcpy(char *dst, const char *src, size_t n) {
    const int T = n >= ((size_t)8 << 20) ? 4 : 1;
    if (T == 1) { memcpy(dst, src, n); return; }
    std::vector<std::thread> th;
    const size_t per = (n / T + 4095) & ~(size_t)4095;
    for (int t = 0; t < T; t++) {
        const size_t lo = std::min(n, per * (size_t)t), hi = std::min(n, lo + per);
        if (hi > lo) th.emplace_back([=] { memcpy(dst + lo, src + lo, hi - lo); });
    }
    for (auto &x : th) x.join();
}
const int64_t PIPE_MIN = (int64_t)64 << 20;
}  // namespace

// ro / oo: nv + 1 offsets into src_bases / out (rebased to 0 or not: only ro[0], oo[0] and differences are used)
// out32 != nullptr (and out == nullptr): the kernels write int32 results (SbwtIndexView::out32): half the bytes over PCIe
static int search_host_pipelined(const sbwtgpu_index *idx, const char *src_bases, const int64_t *ro, const int64_t *oo,
                                 int64_t nv, int64_t *out, int streaming, int32_t *out32 = nullptr) {
    const int64_t vb = out32 ? 4 : 8;                   // bytes of a result on its way to the host
    const bool pin_in = is_pinned(src_bases), pin_out = is_pinned(out32 ? (const void *)out32 : (const void *)out);
    // chunks: results <= 512 MiB when they land in the caller's pinned memory, <= 128 MiB when they are staged
    static const int64_t chunk_mb = [] { const char *e = getenv("SBWTGPU_PIPE_CHUNK_MB"); return e ? atoll(e) : 0ll; }();
    const int64_t CH_OUT = (chunk_mb > 0 ? chunk_mb : (pin_out ? (int64_t)512 : (int64_t)128)) << 20;
    std::vector<int64_t> cuts{0};
    int64_t max_bases = 0, max_reads = 0, max_vals = 0;
    for (int64_t lo = 0; lo < nv;) {
        int64_t hi = lo + 1;
        // (offsets are non-decreasing: bisect for the last read whose results still fit)
        int64_t a = lo + 1, b = nv;
        while (a < b) {
            const int64_t mid = a + (b - a + 1) / 2;
            if ((oo[mid] - oo[lo]) * vb <= CH_OUT && ro[mid] - ro[lo] <= ((int64_t)1 << 30)) a = mid; else b = mid - 1;
        }
        hi = a;
        cuts.push_back(hi);
        max_bases = std::max(max_bases, ro[hi] - ro[lo]);
        max_reads = std::max(max_reads, hi - lo);
        max_vals = std::max(max_vals, oo[hi] - oo[lo]);
        lo = hi;
    }
    const int64_t n_chunks = (int64_t)cuts.size() - 1;
    const int64_t ws_bytes = sbwtgpu_search_workspace_bytes(max_bases);
    const int64_t need_in = a256(max_bases + 16) + 2 * a256((max_reads + 1) * 8);
    const int64_t need_out = pin_out ? 0 : a256(max_vals * vb + 8);
    const int64_t need_dev = a256(max_bases + 16) + 2 * a256((max_reads + 1) * 8) + a256(max_vals * 8 + 8) + a256(ws_bytes);
    DeviceGuard guard(idx->device);
    PipeSlot S[2];
    {
        std::lock_guard<std::mutex> lock(g_pipe_mutex);
        for (size_t i = 0; i < g_pipe_parked.size(); i++)
            if (g_pipe_parked[i].device == idx->device) {
                S[0] = g_pipe_parked[i].s[0];
                S[1] = g_pipe_parked[i].s[1];
                g_pipe_parked.erase(g_pipe_parked.begin() + (long)i);
                break;
            }
    }
    int rc = SBWTGPU_OK;
    auto cleanup = [&]() { S[0].release(); S[1].release(); };
    for (int q = 0; q < 2 && rc == SBWTGPU_OK; q++) {
        PipeSlot &P = S[q];
        hipError_t e = hipSuccess;
        if (!P.st) e = hipStreamCreateWithFlags(&P.st, hipStreamNonBlocking);
        if (e == hipSuccess && !P.h_status) e = hipHostMalloc((void **)&P.h_status, 64, hipHostMallocDefault);
        if (e == hipSuccess && P.cap_in < need_in) {
            if (P.h_in) (void)hipHostFree(P.h_in);
            P.h_in = nullptr; P.cap_in = 0;
            if ((e = hipHostMalloc((void **)&P.h_in, (size_t)need_in, hipHostMallocDefault)) == hipSuccess) P.cap_in = need_in;
        }
        if (e == hipSuccess && P.cap_out < need_out) {
            if (P.h_out) (void)hipHostFree(P.h_out);
            P.h_out = nullptr; P.cap_out = 0;
            if ((e = hipHostMalloc((void **)&P.h_out, (size_t)need_out, hipHostMallocDefault)) == hipSuccess) P.cap_out = need_out;
        }
        if (e == hipSuccess && P.cap_dev < need_dev) {
            if (P.d_mem) (void)hipFree(P.d_mem);
            P.d_mem = nullptr; P.cap_dev = 0;
            if ((e = hipMalloc((void **)&P.d_mem, (size_t)need_dev)) == hipSuccess) P.cap_dev = need_dev;
        }
        if (e != hipSuccess)
            rc = fail(e == hipErrorOutOfMemory ? SBWTGPU_ERR_OOM : SBWTGPU_ERR_HIP, "pipeline buffers: %s", hipGetErrorString(e));
    }
    if (rc != SBWTGPU_OK) { (void)hipGetLastError(); cleanup(); return rc; }
    bool bug = false;
    struct Carve { char *bases; int64_t *roff, *ooff, *out; char *ws; };
    auto carve = [&](PipeSlot &P) {
        Carve c;
        char *p = P.d_mem;
        c.bases = p; p += a256(max_bases + 16);
        c.roff = (int64_t *)p; p += a256((max_reads + 1) * 8);
        c.ooff = (int64_t *)p; p += a256((max_reads + 1) * 8);
        c.out = (int64_t *)p; p += a256(max_vals * 8 + 8);
        c.ws = p;
        return c;
    };
    auto submit = [&](int64_t c) -> int {
        PipeSlot &P = S[c & 1];
        const Carve d = carve(P);
        const int64_t lo = cuts[(size_t)c], hi = cuts[(size_t)c + 1], nr = hi - lo, nb = ro[hi] - ro[lo], nvals = oo[hi] - oo[lo];
        int64_t *hro = (int64_t *)(P.h_in + a256(max_bases + 16)), *hoo = (int64_t *)((char *)hro + a256((max_reads + 1) * 8));
        for (int64_t r = 0; r <= nr; r++) { hro[r] = ro[lo + r] - ro[lo]; hoo[r] = oo[lo + r] - oo[lo]; }
        const char *hb = src_bases + ro[lo];
        if (!pin_in) { memcpy(P.h_in, hb, (size_t)nb); hb = P.h_in; }
        hipError_t e;
        if ((e = hipMemcpyAsync(d.bases, hb, (size_t)nb, hipMemcpyHostToDevice, P.st)) != hipSuccess ||
            (e = hipMemcpyAsync(d.roff, hro, (size_t)(nr + 1) * 8, hipMemcpyHostToDevice, P.st)) != hipSuccess ||
            (e = hipMemcpyAsync(d.ooff, hoo, (size_t)(nr + 1) * 8, hipMemcpyHostToDevice, P.st)) != hipSuccess)
            return fail(SBWTGPU_ERR_HIP, "H2D copy: %s", hipGetErrorString(e));
        t_out32 = out32 ? 1 : 0;                        // (int32 results: the kernels write them into the same device range)
        int r2 = search_dev_common(idx, d.bases, nb, d.roff, nr, d.out, d.ooff, d.ws, ws_bytes, P.st, streaming);
        t_out32 = 0;
        if (r2 != SBWTGPU_OK) return r2;
        char *target = pin_out ? (out32 ? (char *)(out32 + oo[lo]) : (char *)(out + oo[lo])) : P.h_out;
        const void *from = d.out;
        if ((e = hipMemcpyAsync(target, from, (size_t)(nvals * vb), hipMemcpyDeviceToHost, P.st)) != hipSuccess ||
            (e = hipMemcpyAsync(P.h_status, d.ws + offsetof(SbwtWorkHeader, status), 4, hipMemcpyDeviceToHost, P.st)) != hipSuccess)
            return fail(SBWTGPU_ERR_HIP, "D2H copy: %s", hipGetErrorString(e));
        return SBWTGPU_OK;
    };
    auto collect = [&](int64_t c) -> int {
        PipeSlot &P = S[c & 1];
        hipError_t e = hipStreamSynchronize(P.st);
        if (e != hipSuccess) return fail(SBWTGPU_ERR_HIP, "stream synchronize: %s", hipGetErrorString(e));
        if (P.h_status[0] != 0) bug = true;
        const int64_t lo = cuts[(size_t)c], hi = cuts[(size_t)c + 1];
        if (!pin_out) parallel_memcpy(out32 ? (char *)(out32 + oo[lo]) : (char *)(out + oo[lo]), P.h_out, (size_t)((oo[hi] - oo[lo]) * vb));
        return SBWTGPU_OK;
    };
    for (int64_t c = 0; c < n_chunks && rc == SBWTGPU_OK; c++) {
        if (c >= 2) rc = collect(c - 2);
        if (rc == SBWTGPU_OK) rc = submit(c);
    }
    for (int64_t c = std::max<int64_t>(0, n_chunks - 2); c < n_chunks && rc == SBWTGPU_OK; c++) rc = collect(c);
    if (rc != SBWTGPU_OK) {
        (void)hipDeviceSynchronize();
        cleanup();
        return rc;
    }
    {
        std::lock_guard<std::mutex> lock(g_pipe_mutex);
        ParkedPipe pp;
        pp.device = idx->device;
        pp.s[0] = S[0];
        pp.s[1] = S[1];
        g_pipe_parked.push_back(pp);
    }
    if (bug) return fail(SBWTGPU_ERR_NOT_SINGLETON, "Bug: k-mer search did not give a singleton interval");
    return SBWTGPU_OK;
}

static int search_host_common(const sbwtgpu_index *idx, const char *bases, const int64_t *read_off, int64_t n_reads,
                              int64_t *out, const int64_t *out_off, int streaming) {
    if (!idx) return fail(SBWTGPU_ERR_INVALID_ARG, "idx is NULL");
    if (streaming && !idx->h.has_ssup)
        return fail(SBWTGPU_ERR_NO_STREAMING, "Error: streaming search support not built");
    if (n_reads < 0) return fail(SBWTGPU_ERR_INVALID_ARG, "negative n_reads");
    if (n_reads == 0) return SBWTGPU_OK;
    if (!read_off || !out_off) return fail(SBWTGPU_ERR_INVALID_ARG, "NULL offsets");
    bool any_long = false;
    int rc = check_reads(read_off, out_off, n_reads, idx->h.k, &any_long);
    if (rc != SBWTGPU_OK) return rc;
    const int64_t base0 = read_off[0], total = read_off[n_reads] - base0;
    const int64_t out0 = out_off[0], n_out = out_off[n_reads] - out0;
    if (total > 0 && !bases) return fail(SBWTGPU_ERR_INVALID_ARG, "bases is NULL");
    if (n_out > 0 && !out) return fail(SBWTGPU_ERR_INVALID_ARG, "out is NULL");
    if (n_out == 0) return SBWTGPU_OK;
    // a large batch of ordinary reads: the pipeline works on the caller's arrays as they are (it only ever uses
    // differences of offsets; every chunk's offsets are rebased into its staging buffer anyway)
    if (!any_long && n_out * 8 >= PIPE_MIN)
        return search_host_pipelined(idx, bases, read_off, out_off, n_reads, out, streaming);

    // offsets are rebased so that device buffers start at 0; long reads are cut into pieces (see above)
    const int64_t kk = idx->h.k;
    std::vector<int64_t> ro, oo;
    std::vector<char> vbases;
    const char *src_bases = bases + base0;
    int64_t nv = n_reads, vtotal = total;
    try {
        if (!any_long) {
            ro.resize((size_t)n_reads + 1);
            oo.resize((size_t)n_reads + 1);
            for (int64_t r = 0; r <= n_reads; r++) {
                ro[(size_t)r] = read_off[r] - base0;
                oo[(size_t)r] = out_off[r] - out0;
            }
        } else {
            int64_t bound = 0;
            for (int64_t r = 0; r < n_reads; r++) bound += pieces_bases_bound(read_off[r + 1] - read_off[r], kk);
            vbases.resize((size_t)bound + 16);
            ro.assign(1, 0);
            oo.assign(1, 0);
            int64_t w = 0;
            for (int64_t r = 0; r < n_reads; r++)
                w += append_pieces(bases + read_off[r], read_off[r + 1] - read_off[r], kk, vbases.data() + w, ro, oo);
            src_bases = vbases.data();
            vtotal = w;
            nv = (int64_t)ro.size() - 1;
        }
    } catch (const std::bad_alloc &) {
        return fail(SBWTGPU_ERR_OOM, "out of host memory");
    }
    if (n_out * 8 >= PIPE_MIN)
        return search_host_pipelined(idx, src_bases, ro.data(), oo.data(), nv, out + out0, streaming);
    DeviceGuard guard(idx->device);
    const int64_t ws_bytes = sbwtgpu_search_workspace_bytes(vtotal);
    {   // small call: layout [roff][ooff][bases] (in) | [out][workspace] (the D2H copy ends with the workspace header)
        const size_t o_roff = 0, o_ooff = up256((size_t)(nv + 1) * 8), o_bases = o_ooff + up256((size_t)(nv + 1) * 8);
        const size_t o_out = o_bases + up256((size_t)vtotal + 16), o_ws = o_out + up256((size_t)n_out * 8);
        SmallSlot *sl = small_slot(idx->device, o_ws + (size_t)ws_bytes);
        if (sl) {
            memcpy(sl->host + o_roff, ro.data(), (size_t)(nv + 1) * 8);
            memcpy(sl->host + o_ooff, oo.data(), (size_t)(nv + 1) * 8);
            memcpy(sl->host + o_bases, src_bases, (size_t)vtotal);
            HIP_TRY(hipMemcpyAsync(sl->dev, sl->host, o_bases + (size_t)vtotal, hipMemcpyHostToDevice, sl->stream));
            rc = search_dev_common(idx, sl->dev + o_bases, vtotal, (const int64_t *)(sl->dev + o_roff), nv,
                                   (int64_t *)(sl->dev + o_out), (const int64_t *)(sl->dev + o_ooff), sl->dev + o_ws, ws_bytes,
                                   sl->stream, streaming);
            if (rc != SBWTGPU_OK) return rc;
            const size_t back = (o_ws - o_out) + sizeof(SbwtWorkHeader);
            HIP_TRY(hipMemcpyAsync(sl->host + o_out, sl->dev + o_out, back, hipMemcpyDeviceToHost, sl->stream));
            HIP_TRY(hipStreamSynchronize(sl->stream));
            memcpy(out + out0, sl->host + o_out, (size_t)n_out * 8);
            if (reinterpret_cast<const SbwtWorkHeader *>(sl->host + o_ws)->status != 0)
                return fail(SBWTGPU_ERR_NOT_SINGLETON, "Bug: k-mer search did not give a singleton interval");
            return SBWTGPU_OK;
        }
    }
    Stream st;
    HIP_TRY(hipStreamCreateWithFlags(&st.s, hipStreamNonBlocking));
    DevBuf d_bases, d_roff, d_ooff, d_out, d_ws;
    HIP_TRY(d_bases.alloc((size_t)vtotal + 16));
    HIP_TRY(d_roff.alloc((size_t)(nv + 1) * 8));
    HIP_TRY(d_ooff.alloc((size_t)(nv + 1) * 8));
    HIP_TRY(d_out.alloc((size_t)n_out * 8));
    HIP_TRY(d_ws.alloc((size_t)ws_bytes));
    HIP_TRY(hipMemcpyAsync(d_bases.p, src_bases, (size_t)vtotal, hipMemcpyHostToDevice, st.s));
    HIP_TRY(hipMemcpyAsync(d_roff.p, ro.data(), (size_t)(nv + 1) * 8, hipMemcpyHostToDevice, st.s));
    HIP_TRY(hipMemcpyAsync(d_ooff.p, oo.data(), (size_t)(nv + 1) * 8, hipMemcpyHostToDevice, st.s));
    rc = search_dev_common(idx, (const char *)d_bases.p, vtotal, (const int64_t *)d_roff.p, nv, (int64_t *)d_out.p,
                           (const int64_t *)d_ooff.p, d_ws.p, ws_bytes, st.s, streaming);
    if (rc != SBWTGPU_OK) return rc;
    HIP_TRY(hipMemcpyAsync(out + out0, d_out.p, (size_t)n_out * 8, hipMemcpyDeviceToHost, st.s));
    SbwtWorkHeader hdr;
    HIP_TRY(hipMemcpyAsync(&hdr, d_ws.p, sizeof(hdr), hipMemcpyDeviceToHost, st.s));
    HIP_TRY(hipStreamSynchronize(st.s));
    if (hdr.status != 0)
        return fail(SBWTGPU_ERR_NOT_SINGLETON, "Bug: k-mer search did not give a singleton interval");
    return SBWTGPU_OK;
}

int sbwtgpu_streaming_search_batch(const sbwtgpu_index *idx, const char *bases, const int64_t *read_off,
                                   int64_t n_reads, int64_t *out, const int64_t *out_off) {
    return search_host_common(idx, bases, read_off, n_reads, out, out_off, 1);
}

int sbwtgpu_search_batch(const sbwtgpu_index *idx, const char *bases, const int64_t *read_off, int64_t n_reads,
                         int64_t *out, const int64_t *out_off) {
    return search_host_common(idx, bases, read_off, n_reads, out, out_off, 0);
}

// Results as int32 (result compaction): for indexes of fewer than 2^31 columns every rank fits, -1 stays -1.  The
// kernels write int32 themselves, so a result costs 4 bytes of HBM writes and of PCIe instead of 8.  Large batches go through
// the same two-stream pipeline as the int64 calls; small ones through the int64 call and a host loop.
static int search_host_i32(const sbwtgpu_index *idx, const char *bases, const int64_t *read_off, int64_t n_reads,
                           int32_t *out, const int64_t *out_off, int streaming) {
    if (!idx) return fail(SBWTGPU_ERR_INVALID_ARG, "idx is NULL");
    if (idx->h.n_nodes >= ((int64_t)1 << 31))
        return fail(SBWTGPU_ERR_INVALID_ARG, "int32 results need an index of fewer than 2^31 columns (this one has %lld)",
                    (long long)idx->h.n_nodes);
    if (streaming && !idx->h.has_ssup) return fail(SBWTGPU_ERR_NO_STREAMING, "Error: streaming search support not built");
    if (n_reads < 0) return fail(SBWTGPU_ERR_INVALID_ARG, "negative n_reads");
    if (n_reads == 0) return SBWTGPU_OK;
    if (!read_off || !out_off) return fail(SBWTGPU_ERR_INVALID_ARG, "NULL offsets");
    bool any_long = false;
    int rc = check_reads(read_off, out_off, n_reads, idx->h.k, &any_long);
    if (rc != SBWTGPU_OK) return rc;
    const int64_t n_out = out_off[n_reads] - out_off[0];
    if (n_out == 0) return SBWTGPU_OK;
    if (!out) return fail(SBWTGPU_ERR_INVALID_ARG, "out is NULL");
    if (!any_long && n_out * 4 >= PIPE_MIN)
        return search_host_pipelined(idx, bases, read_off, out_off, n_reads, nullptr, streaming, out);
    std::vector<int64_t> wide;
    try { wide.resize((size_t)n_out); } catch (const std::bad_alloc &) { return fail(SBWTGPU_ERR_OOM, "out of host memory"); }
    std::vector<int64_t> oo((size_t)n_reads + 1);
    for (int64_t r = 0; r <= n_reads; r++) oo[(size_t)r] = out_off[r] - out_off[0];
    rc = search_host_common(idx, bases, read_off, n_reads, wide.data(), oo.data(), streaming);
    if (rc != SBWTGPU_OK) return rc;
    int32_t *dst = out + out_off[0];
    for (int64_t t = 0; t < n_out; t++) dst[t] = (int32_t)wide[(size_t)t];
    return SBWTGPU_OK;
}
int sbwtgpu_streaming_search_batch_i32(const sbwtgpu_index *idx, const char *bases, const int64_t *read_off,
                                       int64_t n_reads, int32_t *out, const int64_t *out_off) {
    return search_host_i32(idx, bases, read_off, n_reads, out, out_off, 1);
}
int sbwtgpu_search_batch_i32(const sbwtgpu_index *idx, const char *bases, const int64_t *read_off, int64_t n_reads,
                             int32_t *out, const int64_t *out_off) {
    return search_host_i32(idx, bases, read_off, n_reads, out, out_off, 0);
}

int sbwtgpu_rank_batch(const sbwtgpu_index *idx, const int64_t *pos, const char *sym, int64_t n, int64_t *out) {
    if (!idx) return fail(SBWTGPU_ERR_INVALID_ARG, "idx is NULL");
    if (n < 0) return fail(SBWTGPU_ERR_INVALID_ARG, "negative n");
    if (n == 0) return SBWTGPU_OK;
    if (!pos || !sym || !out) return fail(SBWTGPU_ERR_INVALID_ARG, "NULL argument");
    for (int64_t t = 0; t < n; t++)
        if (pos[t] < 0 || pos[t] > idx->h.n_nodes)
            return fail(SBWTGPU_ERR_INVALID_ARG, "pos[%lld] = %lld outside [0, n_nodes]", (long long)t, (long long)pos[t]);
    DeviceGuard guard(idx->device);
    {   // small call: [pos][sym] in, [out] back
        const size_t o_sym = up256((size_t)n * 8), o_out = o_sym + up256((size_t)n);
        SmallSlot *sl = small_slot(idx->device, o_out + (size_t)n * 8);
        if (sl) {
            memcpy(sl->host, pos, (size_t)n * 8);
            memcpy(sl->host + o_sym, sym, (size_t)n);
            HIP_TRY(hipMemcpyAsync(sl->dev, sl->host, o_sym + (size_t)n, hipMemcpyHostToDevice, sl->stream));
            int rc = sbwtgpu_rank_dev(idx, (const int64_t *)sl->dev, sl->dev + o_sym, n, (int64_t *)(sl->dev + o_out), sl->stream);
            if (rc != SBWTGPU_OK) return rc;
            HIP_TRY(hipMemcpyAsync(sl->host + o_out, sl->dev + o_out, (size_t)n * 8, hipMemcpyDeviceToHost, sl->stream));
            HIP_TRY(hipStreamSynchronize(sl->stream));
            memcpy(out, sl->host + o_out, (size_t)n * 8);
            return SBWTGPU_OK;
        }
    }
    Stream st;
    HIP_TRY(hipStreamCreateWithFlags(&st.s, hipStreamNonBlocking));
    DevBuf d_pos, d_sym, d_out;
    HIP_TRY(d_pos.alloc((size_t)n * 8));
    HIP_TRY(d_sym.alloc((size_t)n));
    HIP_TRY(d_out.alloc((size_t)n * 8));
    HIP_TRY(hipMemcpyAsync(d_pos.p, pos, (size_t)n * 8, hipMemcpyHostToDevice, st.s));
    HIP_TRY(hipMemcpyAsync(d_sym.p, sym, (size_t)n, hipMemcpyHostToDevice, st.s));
    int rc = sbwtgpu_rank_dev(idx, (const int64_t *)d_pos.p, (const char *)d_sym.p, n, (int64_t *)d_out.p, st.s);
    if (rc != SBWTGPU_OK) return rc;
    HIP_TRY(hipMemcpyAsync(out, d_out.p, (size_t)n * 8, hipMemcpyDeviceToHost, st.s));
    HIP_TRY(hipStreamSynchronize(st.s));
    return SBWTGPU_OK;
}

int sbwtgpu_update_interval_batch(const sbwtgpu_index *idx, const char *bases, const int64_t *off, int64_t n,
                                  int64_t *first, int64_t *second) {
    if (!idx) return fail(SBWTGPU_ERR_INVALID_ARG, "idx is NULL");
    if (idx->h.rank_only) return fail(SBWTGPU_ERR_INVALID_ARG, "%s", RANK_ONLY_MSG);
    if (n < 0) return fail(SBWTGPU_ERR_INVALID_ARG, "negative n");
    if (n == 0) return SBWTGPU_OK;
    if (!off || !first || !second) return fail(SBWTGPU_ERR_INVALID_ARG, "NULL argument");
    const int64_t base0 = off[0], total = off[n] - base0;
    if (total < 0 || (total > 0 && !bases)) return fail(SBWTGPU_ERR_INVALID_ARG, "bad bases/offsets");
    for (int64_t t = 0; t < n; t++) {
        if (off[t + 1] < off[t]) return fail(SBWTGPU_ERR_INVALID_ARG, "off is not non-decreasing");
        if (first[t] != -1 && (first[t] < 0 || second[t] >= idx->h.n_nodes || second[t] < -1))
            return fail(SBWTGPU_ERR_INVALID_ARG, "interval %lld out of range", (long long)t);
    }
    DeviceGuard guard(idx->device);
    {   // small call: [first][second] (in and back) [off][bases]
        const size_t o_s = up256((size_t)n * 8), o_off = 2 * o_s, o_bases = o_off + up256((size_t)(n + 1) * 8);
        SmallSlot *sl = small_slot(idx->device, o_bases + (size_t)total + 16);
        if (sl) {
            memcpy(sl->host, first, (size_t)n * 8);
            memcpy(sl->host + o_s, second, (size_t)n * 8);
            int64_t *o = reinterpret_cast<int64_t *>(sl->host + o_off);
            for (int64_t t = 0; t <= n; t++) o[t] = off[t] - base0;
            if (total) memcpy(sl->host + o_bases, bases + base0, (size_t)total);
            HIP_TRY(hipMemcpyAsync(sl->dev, sl->host, o_bases + (size_t)total, hipMemcpyHostToDevice, sl->stream));
            sbwt_launch_update_interval(idx->view(), sl->dev + o_bases, (const long long *)(sl->dev + o_off), n,
                                        (long long *)sl->dev, (long long *)(sl->dev + o_s), sl->stream);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(sl->host, sl->dev, o_s + (size_t)n * 8, hipMemcpyDeviceToHost, sl->stream));
            HIP_TRY(hipStreamSynchronize(sl->stream));
            memcpy(first, sl->host, (size_t)n * 8);
            memcpy(second, sl->host + o_s, (size_t)n * 8);
            return SBWTGPU_OK;
        }
    }
    Stream st;
    HIP_TRY(hipStreamCreateWithFlags(&st.s, hipStreamNonBlocking));
    DevBuf d_bases, d_off, d_f, d_s;
    HIP_TRY(d_bases.alloc((size_t)total + 16));
    HIP_TRY(d_off.alloc((size_t)(n + 1) * 8));
    HIP_TRY(d_f.alloc((size_t)n * 8));
    HIP_TRY(d_s.alloc((size_t)n * 8));
    std::vector<int64_t> o((size_t)n + 1);
    for (int64_t t = 0; t <= n; t++) o[(size_t)t] = off[t] - base0;
    if (total) HIP_TRY(hipMemcpyAsync(d_bases.p, bases + base0, (size_t)total, hipMemcpyHostToDevice, st.s));
    HIP_TRY(hipMemcpyAsync(d_off.p, o.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st.s));
    HIP_TRY(hipMemcpyAsync(d_f.p, first, (size_t)n * 8, hipMemcpyHostToDevice, st.s));
    HIP_TRY(hipMemcpyAsync(d_s.p, second, (size_t)n * 8, hipMemcpyHostToDevice, st.s));
    sbwt_launch_update_interval(idx->view(), (const char *)d_bases.p, (const long long *)d_off.p, n,
                                (long long *)d_f.p, (long long *)d_s.p, st.s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(first, d_f.p, (size_t)n * 8, hipMemcpyDeviceToHost, st.s));
    HIP_TRY(hipMemcpyAsync(second, d_s.p, (size_t)n * 8, hipMemcpyDeviceToHost, st.s));
    HIP_TRY(hipStreamSynchronize(st.s));
    return SBWTGPU_OK;
}

int sbwtgpu_forward_batch(const sbwtgpu_index *idx, const int64_t *node, const char *sym, int64_t n, int64_t *out) {
    if (!idx) return fail(SBWTGPU_ERR_INVALID_ARG, "idx is NULL");
    if (idx->h.rank_only) return fail(SBWTGPU_ERR_INVALID_ARG, "%s", RANK_ONLY_MSG);
    if (!idx->h.has_ssup)
        return fail(SBWTGPU_ERR_NO_STREAMING, "Error: Streaming support required for SBWT::forward");
    if (n < 0) return fail(SBWTGPU_ERR_INVALID_ARG, "negative n");
    if (n == 0) return SBWTGPU_OK;
    if (!node || !sym || !out) return fail(SBWTGPU_ERR_INVALID_ARG, "NULL argument");
    for (int64_t t = 0; t < n; t++)
        if (node[t] < 0 || node[t] >= idx->h.n_nodes)
            return fail(SBWTGPU_ERR_INVALID_ARG, "node[%lld] out of range", (long long)t);
    DeviceGuard guard(idx->device);
    {   // small call: [node][sym] in, [out] back
        const size_t o_sym = up256((size_t)n * 8), o_out = o_sym + up256((size_t)n);
        SmallSlot *sl = small_slot(idx->device, o_out + (size_t)n * 8);
        if (sl) {
            memcpy(sl->host, node, (size_t)n * 8);
            memcpy(sl->host + o_sym, sym, (size_t)n);
            HIP_TRY(hipMemcpyAsync(sl->dev, sl->host, o_sym + (size_t)n, hipMemcpyHostToDevice, sl->stream));
            sbwt_launch_forward(idx->view(), (const long long *)sl->dev, sl->dev + o_sym, n, (long long *)(sl->dev + o_out),
                                sl->stream);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(sl->host + o_out, sl->dev + o_out, (size_t)n * 8, hipMemcpyDeviceToHost, sl->stream));
            HIP_TRY(hipStreamSynchronize(sl->stream));
            memcpy(out, sl->host + o_out, (size_t)n * 8);
            return SBWTGPU_OK;
        }
    }
    Stream st;
    HIP_TRY(hipStreamCreateWithFlags(&st.s, hipStreamNonBlocking));
    DevBuf d_node, d_sym, d_out;
    HIP_TRY(d_node.alloc((size_t)n * 8));
    HIP_TRY(d_sym.alloc((size_t)n));
    HIP_TRY(d_out.alloc((size_t)n * 8));
    HIP_TRY(hipMemcpyAsync(d_node.p, node, (size_t)n * 8, hipMemcpyHostToDevice, st.s));
    HIP_TRY(hipMemcpyAsync(d_sym.p, sym, (size_t)n, hipMemcpyHostToDevice, st.s));
    sbwt_launch_forward(idx->view(), (const long long *)d_node.p, (const char *)d_sym.p, n, (long long *)d_out.p, st.s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, d_out.p, (size_t)n * 8, hipMemcpyDeviceToHost, st.s));
    HIP_TRY(hipStreamSynchronize(st.s));
    return SBWTGPU_OK;
}


// One host-buffer call of the small API neighbours: inputs staged contiguously, one H2D, the kernel, one D2H.
// `in` / `back` describe byte ranges of the staging buffer; small calls use the thread's slot, large ones a
// temporary stream + device buffer.
namespace {
struct Staged {
    SmallSlot *sl = nullptr;
    Stream st;
    DevBuf dbuf;
    std::vector<char> hbuf;
    char *host = nullptr, *dev = nullptr;
    hipStream_t stream = nullptr;
    int open(int device, size_t bytes) {
        sl = small_slot(device, bytes);
        if (sl) { host = sl->host; dev = sl->dev; stream = sl->stream; return SBWTGPU_OK; }
        HIP_TRY(hipStreamCreateWithFlags(&st.s, hipStreamNonBlocking));
        HIP_TRY(dbuf.alloc(bytes));
        try { hbuf.resize(bytes); } catch (...) { return fail(SBWTGPU_ERR_OOM, "out of host memory"); }
        host = hbuf.data(); dev = (char *)dbuf.p; stream = st.s;
        return SBWTGPU_OK;
    }
};
}  // namespace

int sbwtgpu_partial_search_batch(const sbwtgpu_index *idx, const char *bases, const int64_t *off, int64_t n,
                                 int64_t *first, int64_t *second, int64_t *matched) {
    if (!idx) return fail(SBWTGPU_ERR_INVALID_ARG, "idx is NULL");
    if (idx->h.rank_only) return fail(SBWTGPU_ERR_INVALID_ARG, "%s", RANK_ONLY_MSG);
    if (n < 0) return fail(SBWTGPU_ERR_INVALID_ARG, "negative n");
    if (n == 0) return SBWTGPU_OK;
    if (!off || !first || !second || !matched) return fail(SBWTGPU_ERR_INVALID_ARG, "NULL argument");
    const int64_t base0 = off[0], total = off[n] - base0;
    if (total < 0 || (total > 0 && !bases)) return fail(SBWTGPU_ERR_INVALID_ARG, "bad bases/offsets");
    for (int64_t t = 0; t < n; t++)
        if (off[t + 1] < off[t]) return fail(SBWTGPU_ERR_INVALID_ARG, "off is not non-decreasing");
    DeviceGuard guard(idx->device);
    // [first][second][matched] (back) [off][bases] (in)
    const size_t o_s = up256((size_t)n * 8), o_m = 2 * o_s, o_off = 3 * o_s, o_bases = o_off + up256((size_t)(n + 1) * 8);
    Staged sg;
    int rc = sg.open(idx->device, o_bases + (size_t)total + 16);
    if (rc != SBWTGPU_OK) return rc;
    int64_t *o = reinterpret_cast<int64_t *>(sg.host + o_off);
    for (int64_t t = 0; t <= n; t++) o[t] = off[t] - base0;
    if (total) memcpy(sg.host + o_bases, bases + base0, (size_t)total);
    HIP_TRY(hipMemcpyAsync(sg.dev + o_off, sg.host + o_off, (o_bases - o_off) + (size_t)total, hipMemcpyHostToDevice, sg.stream));
    sbwt_launch_partial_search(idx->view(), sg.dev + o_bases, (const long long *)(sg.dev + o_off), n, (long long *)sg.dev,
                               (long long *)(sg.dev + o_s), (long long *)(sg.dev + o_m), sg.stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(sg.host, sg.dev, o_m + (size_t)n * 8, hipMemcpyDeviceToHost, sg.stream));
    HIP_TRY(hipStreamSynchronize(sg.stream));
    memcpy(first, sg.host, (size_t)n * 8);
    memcpy(second, sg.host + o_s, (size_t)n * 8);
    memcpy(matched, sg.host + o_m, (size_t)n * 8);
    return SBWTGPU_OK;
}

int sbwtgpu_get_kmer_batch(const sbwtgpu_index *idx, const int64_t *colex_rank, int64_t n, char *out) {
    if (!idx) return fail(SBWTGPU_ERR_INVALID_ARG, "idx is NULL");
    if (idx->h.rank_only) return fail(SBWTGPU_ERR_INVALID_ARG, "%s", RANK_ONLY_MSG);
    if (n < 0) return fail(SBWTGPU_ERR_INVALID_ARG, "negative n");
    if (n == 0) return SBWTGPU_OK;
    if (!colex_rank || !out) return fail(SBWTGPU_ERR_INVALID_ARG, "NULL argument");
    for (int64_t t = 0; t < n; t++)
        if (colex_rank[t] < 0 || colex_rank[t] >= idx->h.n_nodes)
            return fail(SBWTGPU_ERR_INVALID_ARG, "colex_rank[%lld] out of range", (long long)t);
    DeviceGuard guard(idx->device);
    const int64_t k = idx->h.k;
    const size_t o_out = up256((size_t)n * 8);
    Staged sg;
    int rc = sg.open(idx->device, o_out + (size_t)(n * k) + 16);
    if (rc != SBWTGPU_OK) return rc;
    memcpy(sg.host, colex_rank, (size_t)n * 8);
    HIP_TRY(hipMemcpyAsync(sg.dev, sg.host, (size_t)n * 8, hipMemcpyHostToDevice, sg.stream));
    sbwt_launch_get_kmer(idx->view(), (const long long *)sg.dev, n, sg.dev + o_out, sg.stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(sg.host + o_out, sg.dev + o_out, (size_t)(n * k), hipMemcpyDeviceToHost, sg.stream));
    HIP_TRY(hipStreamSynchronize(sg.stream));
    memcpy(out, sg.host + o_out, (size_t)(n * k));
    return SBWTGPU_OK;
}

int sbwtgpu_select_batch(const sbwtgpu_index *idx, const int64_t *j, const char *sym, int64_t n, int64_t *out) {
    if (!idx) return fail(SBWTGPU_ERR_INVALID_ARG, "idx is NULL");
    if (n < 0) return fail(SBWTGPU_ERR_INVALID_ARG, "negative n");
    if (n == 0) return SBWTGPU_OK;
    if (!j || !sym || !out) return fail(SBWTGPU_ERR_INVALID_ARG, "NULL argument");
    for (int64_t t = 0; t < n; t++) {
        const char ch = sym[t];
        const int c = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : -1;
        if (c >= 0 && (j[t] < 1 || j[t] > idx->h.row_ones[c]))
            return fail(SBWTGPU_ERR_INVALID_ARG, "j[%lld] = %lld outside [1, %lld] (ones in row %c)", (long long)t,
                        (long long)j[t], (long long)idx->h.row_ones[c], ch);
    }
    DeviceGuard guard(idx->device);
    const size_t o_sym = up256((size_t)n * 8), o_out = o_sym + up256((size_t)n);
    Staged sg;
    int rc = sg.open(idx->device, o_out + (size_t)n * 8);
    if (rc != SBWTGPU_OK) return rc;
    memcpy(sg.host, j, (size_t)n * 8);
    memcpy(sg.host + o_sym, sym, (size_t)n);
    HIP_TRY(hipMemcpyAsync(sg.dev, sg.host, o_sym + (size_t)n, hipMemcpyHostToDevice, sg.stream));
    long long ones[4] = {idx->h.row_ones[0], idx->h.row_ones[1], idx->h.row_ones[2], idx->h.row_ones[3]};
    sbwt_launch_select(idx->view(), (const long long *)sg.dev, sg.dev + o_sym, n, ones, (long long *)(sg.dev + o_out), sg.stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(sg.host + o_out, sg.dev + o_out, (size_t)n * 8, hipMemcpyDeviceToHost, sg.stream));
    HIP_TRY(hipStreamSynchronize(sg.stream));
    memcpy(out, sg.host + o_out, (size_t)n * 8);
    return SBWTGPU_OK;
}

// ---- construction on the device ---------------------------------------------------
}  // extern "C"
namespace {
// The host part between the builder's two device phases: the dummy prefixes of the predecessor-less k-mers
// (NodeBOSSInMemoryConstructor.hh:70-79) + the root, sorted like Kmer::operator< (label, then length), equal nodes merged.
// KT = the key type of the device builder: uint64_t for k <= 32, unsigned __int128 for 32 < k <= 64.
template <typename KT>
int build_columns_t(SbwtBuildState &S, int64_t k, int build_streaming_support, sbwtgpu_plain_matrix_bits *out) {
    std::vector<KT> nopred, ddata;
    std::vector<unsigned long long> rows;
    std::vector<unsigned> dedges;
    nopred.resize((size_t)S.n_nopred);
    if (sbwt_build_copy_nopred(&S, nopred.data()) != 0) { sbwt_build_release(&S); return fail(SBWTGPU_ERR_HIP, "device builder: copy"); }
    struct Dm { KT data; unsigned len, edges; };
    std::vector<Dm> dm;
    dm.reserve((size_t)S.n_nopred * (size_t)k + 1);
    dm.push_back(Dm{(KT)0, 0, 0});
    const int kbits = 2 * (int)k;
    for (KT zk : nopred)
        for (int j = 0; j < (int)k; j++) {
            KT label = (j == 0) ? (KT)0 : (KT)((zk & ((((KT)1) << (2 * j)) - (KT)1)) << (kbits - 2 * j));
            dm.push_back(Dm{label, (unsigned)j, 1u << (unsigned)((unsigned)(zk >> (2 * j)) & 3u)});
        }
    std::sort(dm.begin(), dm.end(), [](const Dm &a, const Dm &b) { return a.data != b.data ? a.data < b.data : a.len < b.len; });
    size_t wd = 0;
    for (size_t i = 0; i < dm.size(); i++) {
        if (wd > 0 && dm[wd - 1].data == dm[i].data && dm[wd - 1].len == dm[i].len) dm[wd - 1].edges |= dm[i].edges;
        else dm[wd++] = dm[i];
    }
    dm.resize(wd);
    ddata.resize(wd);
    dedges.resize(wd);
    for (size_t i = 0; i < wd; i++) { ddata[i] = dm[i].data; dedges[i] = dm[i].edges; }
    const int64_t n = S.nk + (int64_t)wd, nw = (n + 63) / 64;
    rows.resize((size_t)(5 * nw));
    int rb = sbwt_build_phase_b(&S, ddata.data(), dedges.data(), (long long)wd, build_streaming_support ? 1 : 0, rows.data(), 0);
    const int64_t nk = S.nk;
    sbwt_build_release(&S);
    if (rb != 0) return fail(rb == -8 ? SBWTGPU_ERR_OOM : SBWTGPU_ERR_HIP, "device builder, phase B");
    uint64_t *mem = static_cast<uint64_t *>(malloc((size_t)(5 * nw) * 8 + 8));
    if (!mem) return fail(SBWTGPU_ERR_OOM, "out of host memory");
    memcpy(mem, rows.data(), (size_t)(5 * nw) * 8);
    out->n_nodes = n;
    out->n_kmers = nk;
    out->k = k;
    out->A_bits = mem;
    out->C_bits = mem + nw;
    out->G_bits = mem + 2 * nw;
    out->T_bits = mem + 3 * nw;
    out->suffix_group_starts = build_streaming_support ? mem + 4 * nw : nullptr;
    return SBWTGPU_OK;
}
}  // namespace
extern "C" {

int sbwtgpu_build_plain_matrix(const char *const *seqs, const int64_t *seq_len, int64_t n_seqs, int64_t k, int add_revcomp,
                               int build_streaming_support, int device, sbwtgpu_plain_matrix_bits *out) {
    if (!out) return fail(SBWTGPU_ERR_INVALID_ARG, "out is NULL");
    memset(out, 0, sizeof(*out));
    if (n_seqs < 0 || (n_seqs > 0 && (!seqs || !seq_len))) return fail(SBWTGPU_ERR_INVALID_ARG, "bad sequence list");
    if (k < 2 || k > 64)
        return fail(SBWTGPU_ERR_INVALID_ARG, "the device builder packs a k-mer into 64 or 128 bits: 2 <= k <= 64");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(SBWTGPU_ERR_NO_DEVICE, "no HIP device");
    if (device < 0 || device >= ndev) return fail(SBWTGPU_ERR_NO_DEVICE, "device %d out of range", device);
    DeviceGuard guard(device);
    if (!guard.ok) return fail(SBWTGPU_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
    // one text, sequences separated by a byte that is not ACGT: a k-mer window across a boundary is invalid like one with N
    int64_t n_text = 0;
    for (int64_t i = 0; i < n_seqs; i++) {
        if (seq_len[i] < 0) return fail(SBWTGPU_ERR_INVALID_ARG, "negative sequence length");
        n_text += seq_len[i] + 1;
    }
    std::vector<char> text;
    SbwtBuildState S;
    int rc = SBWTGPU_OK;
    try {
        text.resize((size_t)n_text + 1);
        int64_t w = 0;
        for (int64_t i = 0; i < n_seqs; i++) {
            if (seq_len[i]) memcpy(text.data() + w, seqs[i], (size_t)seq_len[i]);
            w += seq_len[i];
            text[(size_t)w++] = '$';
        }
        int ra = sbwt_build_phase_a(text.data(), n_text, (int)k, add_revcomp ? 1 : 0, &S, 0);
        if (ra != 0) { sbwt_build_release(&S); return fail(ra == -8 ? SBWTGPU_ERR_OOM : SBWTGPU_ERR_HIP, "device builder, phase A"); }
        std::vector<char>().swap(text);
        // the dummy prefixes are expanded on the host, k per predecessor-less k-mer at 16-32 bytes each: about one k-mer per
        // input SEQUENCE, so a genome has a handful and a read set has millions.  Past 2^26 records the caller's host
        // builder (index_builder.hh) is the better tool: report it as "does not fit".
        if ((long long)S.n_nopred * (long long)k > (1ll << 26)) {
            sbwt_build_release(&S);
            return fail(SBWTGPU_ERR_OOM, "device builder: %lld predecessor-less k-mers x k = %lld dummy records (limit 2^26): "
                        "use the host builder for inputs with this many sequences", (long long)S.n_nopred,
                        (long long)S.n_nopred * (long long)k);
        }
        rc = (k <= 32) ? build_columns_t<unsigned long long>(S, k, build_streaming_support, out)
                       : build_columns_t<unsigned __int128>(S, k, build_streaming_support, out);
    } catch (const std::bad_alloc &) {
        sbwt_build_release(&S);
        rc = fail(SBWTGPU_ERR_OOM, "out of host memory");
    }
    return rc;
}

void sbwtgpu_free_plain_matrix(sbwtgpu_plain_matrix_bits *b) {
    if (!b) return;
    free(b->A_bits);                                   // one allocation holds all five rows
    memset(b, 0, sizeof(*b));
}

// ---- device-side formatting + pipelined host path ------------------------------------------------
int64_t sbwtgpu_format_text_bound(const sbwtgpu_index *idx, int64_t n_values, int64_t n_reads) {
    int digits = 1;
    for (int64_t x = idx ? idx->h.n_nodes : INT64_MAX; x >= 10; x /= 10) digits++;
    int tok = digits + 1 > 3 ? digits + 1 : 3;
    return n_values * tok + n_reads + 16;
}
int64_t sbwtgpu_format_scratch_bytes(int64_t n_reads) { return sbwt_format_scratch_bytes(n_reads < 0 ? 0 : n_reads); }

int sbwtgpu_format_results_dev(const sbwtgpu_index *idx, const int64_t *d_values, const int64_t *d_out_off,
                               int64_t n_reads, int64_t n_values, char *d_text, int64_t text_cap, int64_t *d_line_off,
                               void *d_scratch, int64_t scratch_bytes, void *stream) {
    if (!idx) return fail(SBWTGPU_ERR_INVALID_ARG, "idx is NULL");
    if (n_reads < 0 || n_values < 0) return fail(SBWTGPU_ERR_INVALID_ARG, "negative size");
    if (n_reads == 0) return SBWTGPU_OK;
    if (!d_out_off || !d_text || !d_line_off || !d_scratch || (n_values > 0 && !d_values))
        return fail(SBWTGPU_ERR_INVALID_ARG, "NULL device pointer");
    if (text_cap < sbwtgpu_format_text_bound(idx, n_values, n_reads))
        return fail(SBWTGPU_ERR_INVALID_ARG, "text buffer smaller than sbwtgpu_format_text_bound()");
    if (scratch_bytes < sbwtgpu_format_scratch_bytes(n_reads)) return fail(SBWTGPU_ERR_INVALID_ARG, "scratch too small");
    DeviceGuard guard(idx->device);
    sbwt_launch_format(reinterpret_cast<const long long *>(d_values), reinterpret_cast<const long long *>(d_out_off),
                       n_reads, d_text, reinterpret_cast<long long *>(d_line_off), d_scratch,
                       static_cast<hipStream_t>(stream));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SBWTGPU_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
    return SBWTGPU_OK;
}

void sbwtgpu_free_host(void *p) { free(p); }

namespace {
// one pipeline slot: a stream with its device buffers and pinned staging buffers
struct Slot {
    hipStream_t st = nullptr;
    char *h_in = nullptr;            // pinned: bases | read_off | out_off
    char *h_text = nullptr;          // pinned: formatted text
    int64_t *h_total = nullptr;      // pinned: text length, status
    char *d_mem = nullptr;           // one device allocation carved below
    int64_t cap_bases = 0, cap_reads = 0, cap_vals = 0, cap_text = 0;
    // carved device pointers
    char *d_bases = nullptr; int64_t *d_roff = nullptr, *d_ooff = nullptr, *d_out = nullptr, *d_line = nullptr;
    int64_t *d_vooff = nullptr;      // result offsets of the pieces (d_roff holds the pieces' base offsets)
    int64_t cap_vreads = 0;          // pieces per chunk
    char *d_ws = nullptr, *d_text = nullptr, *d_scr = nullptr;
    int64_t ws_bytes = 0, scr_bytes = 0;
    // the chunk in flight
    int64_t n_reads = 0, n_vals = 0, text_len = 0;
    bool busy = false;
    void release() {
        if (st) (void)hipStreamDestroy(st);
        if (h_in) (void)hipHostFree(h_in);
        if (h_text) (void)hipHostFree(h_text);
        if (h_total) (void)hipHostFree(h_total);
        if (d_mem) (void)hipFree(d_mem);
        *this = Slot();
    }
};

// Pinned + device buffers are expensive to create (page pinning), so finished calls park their slots
// here and later calls on the same device reuse them when they are large enough.
std::mutex g_slot_mutex;
struct ParkedSlot { int device; Slot slot; };
std::vector<ParkedSlot> g_parked;

bool take_parked(int device, int64_t bases, int64_t reads, int64_t vals, int64_t text, Slot *out) {
    std::lock_guard<std::mutex> lock(g_slot_mutex);
    for (size_t i = 0; i < g_parked.size(); i++) {
        Slot &S = g_parked[i].slot;
        if (g_parked[i].device == device && S.cap_bases >= bases && S.cap_reads >= reads && S.cap_vals >= vals &&
            S.cap_text >= text && S.ws_bytes >= sbwtgpu_search_workspace_bytes(S.cap_bases)) {   // (tuning may have changed it)
            *out = S;
            g_parked.erase(g_parked.begin() + (long)i);
            return true;
        }
    }
    return false;
}
void park(int device, Slot &S) {
    std::lock_guard<std::mutex> lock(g_slot_mutex);
    if (g_parked.size() >= 8) { S.release(); return; }
    g_parked.push_back(ParkedSlot{device, S});
    S = Slot();
}
}  // namespace

void sbwtgpu_release_cached_buffers(void) {
    {
        std::lock_guard<std::mutex> lock(g_pipe_mutex);
        for (auto &pp : g_pipe_parked) { DeviceGuard guard(pp.device); pp.s[0].release(); pp.s[1].release(); }
        g_pipe_parked.clear();
    }
    std::lock_guard<std::mutex> lock(g_slot_mutex);
    for (auto &ps : g_parked) {
        int prev = -1;
        (void)hipGetDevice(&prev);
        (void)hipSetDevice(ps.device);
        ps.slot.release();
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    g_parked.clear();
    for (auto &sl : t_slots.v) sl.release();            // the calling thread's small-call slots
    t_slots.v.clear();
}

int sbwtgpu_search_text_stream(const sbwtgpu_index *idx, const char *bases, const int64_t *read_off, int64_t n_reads,
                               int streaming, sbwtgpu_text_sink sink, void *sink_ctx, int64_t *n_queries) {
    if (!idx || !sink) return fail(SBWTGPU_ERR_INVALID_ARG, "NULL argument");
    if (n_queries) *n_queries = 0;
    if (streaming && !idx->h.has_ssup)
        return fail(SBWTGPU_ERR_NO_STREAMING, "Error: streaming search support not built");
    if (idx->h.rank_only) return fail(SBWTGPU_ERR_INVALID_ARG, "%s", RANK_ONLY_MSG);
    if (n_reads < 0) return fail(SBWTGPU_ERR_INVALID_ARG, "negative n_reads");
    if (n_reads > 0 && (!read_off || (read_off[n_reads] > read_off[0] && !bases)))
        return fail(SBWTGPU_ERR_INVALID_ARG, "NULL input");
    const int64_t k = idx->h.k;
    // ---- chunking: <= 8 Mi bases and <= 1 Mi reads per chunk (small pinned buffers, deep overlap) ----
    const int64_t CH_BASES = (int64_t)8 << 20, CH_READS = (int64_t)1 << 20;
    std::vector<int64_t> cuts{0};
    int64_t max_bases = 0, max_reads = 0, max_vals = 0, max_vreads = 0;
    {
        // vb / vr: bases and reads of the chunk after long reads are cut into pieces
        int64_t lo = 0, vals = 0, vb = 0, vr = 0;
        for (int64_t r = 0; r < n_reads; r++) {
            int64_t len = read_off[r + 1] - read_off[r];
            if (len < 0) return fail(SBWTGPU_ERR_INVALID_ARG, "read_off is not non-decreasing at read %lld", (long long)r);
            if (len >= ((int64_t)1 << 31)) return fail(SBWTGPU_ERR_READ_TOO_LONG, "read %lld has >= 2^31 bases", (long long)r);
            if (r > lo && (read_off[r + 1] - read_off[lo] > CH_BASES || r - lo >= CH_READS)) {
                max_bases = std::max(max_bases, vb);
                max_reads = std::max(max_reads, r - lo);
                max_vreads = std::max(max_vreads, vr);
                max_vals = std::max(max_vals, vals);
                cuts.push_back(r);
                lo = r;
                vals = vb = vr = 0;
            }
            const int64_t mm = std::max<int64_t>(0, len - k + 1);
            vals += mm;
            vb += pieces_bases_bound(len, k);
            vr += (mm > 2 * PIECE) ? mm / PIECE + 2 : 1;
        }
        if (n_reads > lo) {
            max_bases = std::max(max_bases, vb);
            max_reads = std::max(max_reads, n_reads - lo);
            max_vreads = std::max(max_vreads, vr);
            max_vals = std::max(max_vals, vals);
            cuts.push_back(n_reads);
        }
    }
    const int64_t n_chunks = (int64_t)cuts.size() - 1;
    if (n_chunks == 0) return SBWTGPU_OK;
    DeviceGuard guard(idx->device);
    Slot slots[3];
    int rc = SBWTGPU_OK;
    auto cleanup = [&]() { slots[0].release(); slots[1].release(); slots[2].release(); };
#define PIPE_TRY(expr)                                                                                     \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) {                                                                            \
            rc = fail(e_ == hipErrorOutOfMemory ? SBWTGPU_ERR_OOM : SBWTGPU_ERR_HIP, "%s failed: %s", #expr, \
                      hipGetErrorString(e_));                                                              \
            cleanup();                                                                                     \
            return rc;                                                                                     \
        }                                                                                                  \
    } while (0)
    const int n_slots = n_chunks > 2 ? 3 : (int)n_chunks;
    for (int s = 0; s < n_slots; s++) {
        Slot &S = slots[s];
        if (take_parked(idx->device, max_bases, max_reads, max_vals, sbwtgpu_format_text_bound(idx, max_vals, max_reads), &S) &&
            S.cap_vreads >= max_vreads)
            continue;
        S.release();
        S.cap_bases = max_bases; S.cap_reads = max_reads; S.cap_vals = max_vals; S.cap_vreads = max_vreads;
        S.cap_text = sbwtgpu_format_text_bound(idx, max_vals, max_reads);
        S.ws_bytes = sbwtgpu_search_workspace_bytes(max_bases);
        S.scr_bytes = sbwtgpu_format_scratch_bytes(max_reads);
        const int64_t in_bytes = a256(max_bases + 16) + 2 * a256((max_vreads + 1) * 8) + a256((max_reads + 1) * 8);
        const int64_t dev_bytes = a256(max_bases + 16) + 2 * a256((max_vreads + 1) * 8) + 2 * a256((max_reads + 1) * 8) +
                                  a256(max_vals * 8 + 8) + a256(S.ws_bytes) + a256(S.cap_text) + a256(S.scr_bytes);
        PIPE_TRY(hipStreamCreateWithFlags(&S.st, hipStreamNonBlocking));
        PIPE_TRY(hipHostMalloc((void **)&S.h_in, (size_t)in_bytes, hipHostMallocDefault));
        PIPE_TRY(hipHostMalloc((void **)&S.h_text, (size_t)S.cap_text, hipHostMallocDefault));
        PIPE_TRY(hipHostMalloc((void **)&S.h_total, 64, hipHostMallocDefault));
        PIPE_TRY(hipMalloc((void **)&S.d_mem, (size_t)dev_bytes));
        char *p = S.d_mem;
        S.d_bases = p; p += a256(max_bases + 16);
        S.d_roff = (int64_t *)p; p += a256((max_vreads + 1) * 8);
        S.d_vooff = (int64_t *)p; p += a256((max_vreads + 1) * 8);
        S.d_ooff = (int64_t *)p; p += a256((max_reads + 1) * 8);
        S.d_line = (int64_t *)p; p += a256((max_reads + 1) * 8);
        S.d_out = (int64_t *)p; p += a256(max_vals * 8 + 8);
        S.d_ws = p; p += a256(S.ws_bytes);
        S.d_text = p; p += a256(S.cap_text);
        S.d_scr = p;
    }
    int64_t total_queries = 0;
    bool bug = false;
    // enqueue everything of chunk c on its slot's stream up to the copy of the text length
    auto submit = [&](int64_t c) -> int {
        Slot &S = slots[c % n_slots];
        const int64_t lo = cuts[(size_t)c], hi = cuts[(size_t)c + 1], nr = hi - lo;
        char *hb = S.h_in;
        int64_t *hro = (int64_t *)(S.h_in + a256(S.cap_bases + 16));                    // pieces: base offsets
        int64_t *hvo = (int64_t *)((char *)hro + a256((S.cap_vreads + 1) * 8));         // pieces: result offsets
        int64_t *hoo = (int64_t *)((char *)hvo + a256((S.cap_vreads + 1) * 8));         // reads: result offsets
        int64_t nb = 0, acc = 0, nv = 0;
        hoo[0] = 0;
        bool any_long = false;
        for (int64_t r = 0; r < nr && !any_long; r++) any_long = (read_off[lo + r + 1] - read_off[lo + r] - k + 1) > 2 * PIECE;
        if (!any_long) {
            // the usual chunk: no read long enough to be cut, pieces = reads -- one copy of the bases, offsets by a loop
            const int64_t b0 = read_off[lo];
            nb = read_off[hi] - b0;
            if (nb > S.cap_bases || nr > S.cap_vreads) return fail(SBWTGPU_ERR_HIP, "internal: piece bound exceeded");
            if (nb > 0) memcpy(hb, bases + b0, (size_t)nb);
            hro[0] = 0;
            hvo[0] = 0;
            for (int64_t r = 0; r < nr; r++) {
                const int64_t len = read_off[lo + r + 1] - read_off[lo + r];
                acc += std::max<int64_t>(0, len - k + 1);
                hro[r + 1] = read_off[lo + r + 1] - b0;
                hvo[r + 1] = acc;
                hoo[r + 1] = acc;
            }
            nv = nr;
        } else {
            std::vector<int64_t> vro{0}, voo{0};
            for (int64_t r = 0; r < nr; r++) {
                const int64_t len = read_off[lo + r + 1] - read_off[lo + r];
                nb += append_pieces(bases + read_off[lo + r], len, k, hb + nb, vro, voo);
                acc += std::max<int64_t>(0, len - k + 1);
                hoo[r + 1] = acc;
            }
            nv = (int64_t)vro.size() - 1;
            if (nv > S.cap_vreads || nb > S.cap_bases) return fail(SBWTGPU_ERR_HIP, "internal: piece bound exceeded");
            memcpy(hro, vro.data(), (size_t)(nv + 1) * 8);
            memcpy(hvo, voo.data(), (size_t)(nv + 1) * 8);
        }
        S.n_reads = nr;
        S.n_vals = acc;
        total_queries += acc;
        hipError_t e;
        if ((e = hipMemcpyAsync(S.d_bases, hb, (size_t)nb, hipMemcpyHostToDevice, S.st)) != hipSuccess ||
            (e = hipMemcpyAsync(S.d_roff, hro, (size_t)(nv + 1) * 8, hipMemcpyHostToDevice, S.st)) != hipSuccess ||
            (e = hipMemcpyAsync(S.d_vooff, hvo, (size_t)(nv + 1) * 8, hipMemcpyHostToDevice, S.st)) != hipSuccess ||
            (e = hipMemcpyAsync(S.d_ooff, hoo, (size_t)(nr + 1) * 8, hipMemcpyHostToDevice, S.st)) != hipSuccess)
            return fail(SBWTGPU_ERR_HIP, "H2D copy: %s", hipGetErrorString(e));
        int r2 = search_dev_common(idx, S.d_bases, nb, S.d_roff, nv, S.d_out, S.d_vooff, S.d_ws, S.ws_bytes, S.st, streaming);
        if (r2 != SBWTGPU_OK) return r2;
        r2 = sbwtgpu_format_results_dev(idx, S.d_out, S.d_ooff, nr, acc, S.d_text, S.cap_text, S.d_line, S.d_scr,
                                        S.scr_bytes, S.st);
        if (r2 != SBWTGPU_OK) return r2;
        if ((e = hipMemcpyAsync(&S.h_total[0], S.d_line + nr, 8, hipMemcpyDeviceToHost, S.st)) != hipSuccess ||
            (e = hipMemcpyAsync(&S.h_total[1], S.d_ws + offsetof(SbwtWorkHeader, status), 4, hipMemcpyDeviceToHost, S.st)) != hipSuccess)
            return fail(SBWTGPU_ERR_HIP, "D2H copy: %s", hipGetErrorString(e));
        S.busy = true;
        return SBWTGPU_OK;
    };
    // wait for chunk c's kernels, then start the copy of its text
    auto fetch = [&](int64_t c) -> int {
        Slot &S = slots[c % n_slots];
        hipError_t e = hipStreamSynchronize(S.st);
        if (e != hipSuccess) return fail(SBWTGPU_ERR_HIP, "stream synchronize: %s", hipGetErrorString(e));
        S.text_len = S.h_total[0];
        if ((int)(S.h_total[1] & 0xffffffff) != 0) bug = true;
        if (S.text_len < 0 || S.text_len > S.cap_text) return fail(SBWTGPU_ERR_HIP, "formatted text overflows its bound");
        e = hipMemcpyAsync(S.h_text, S.d_text, (size_t)S.text_len, hipMemcpyDeviceToHost, S.st);
        if (e != hipSuccess) return fail(SBWTGPU_ERR_HIP, "D2H copy: %s", hipGetErrorString(e));
        return SBWTGPU_OK;
    };
    // wait for chunk c's text and hand it to the sink (straight out of the pinned staging buffer)
    auto collect = [&](int64_t c) -> int {
        Slot &S = slots[c % n_slots];
        hipError_t e = hipStreamSynchronize(S.st);
        if (e != hipSuccess) return fail(SBWTGPU_ERR_HIP, "stream synchronize: %s", hipGetErrorString(e));
        if (S.text_len > 0 && sink(sink_ctx, S.h_text, S.text_len) != 0) return fail(SBWTGPU_ERR_INVALID_ARG, "the text sink reported an error");
        S.busy = false;
        return SBWTGPU_OK;
    };
    // Three slots: while the sink has chunk c-2's text (a write to a file takes several times longer than the GPU needs
    // for a chunk), chunk c is on the GPU and chunk c-1's text is on its way down.
    for (int64_t c = 0; c < n_chunks && rc == SBWTGPU_OK; c++) {
        rc = submit(c);                                            // its slot was freed by collect(c - 3) in the last round
        if (rc == SBWTGPU_OK && c >= 1) rc = fetch(c - 1);
        if (rc == SBWTGPU_OK && c >= 2) rc = collect(c - 2);
    }
    if (rc == SBWTGPU_OK) rc = fetch(n_chunks - 1);
    for (int64_t c = std::max<int64_t>(0, n_chunks - 2); c < n_chunks && rc == SBWTGPU_OK; c++) rc = collect(c);
#undef PIPE_TRY
    if (rc != SBWTGPU_OK) {
        (void)hipDeviceSynchronize();
        cleanup();
        return rc;
    }
    for (int s2 = 0; s2 < n_slots; s2++) park(idx->device, slots[s2]);
    if (bug) return fail(SBWTGPU_ERR_NOT_SINGLETON, "Bug: k-mer search did not give a singleton interval");
    if (n_queries) *n_queries = total_queries;
    return SBWTGPU_OK;
}

// The same with the whole text in one malloc'ed buffer: sized by the bound of the batch (untouched pages cost nothing)
// and shrunk at the end, so every chunk's text is copied exactly once.
namespace {
struct TextAcc { char *buf; int64_t len, cap; };
int text_acc_sink(void *ctx, const char *text, int64_t bytes) {
    TextAcc *a = static_cast<TextAcc *>(ctx);
    if (a->len + bytes > a->cap) return 1;
    memcpy(a->buf + a->len, text, (size_t)bytes);
    a->len += bytes;
    return 0;
}
}  // namespace
int sbwtgpu_search_text_batch(const sbwtgpu_index *idx, const char *bases, const int64_t *read_off, int64_t n_reads,
                              int streaming, char **text, int64_t *text_bytes, int64_t *n_queries) {
    if (!idx || !text || !text_bytes) return fail(SBWTGPU_ERR_INVALID_ARG, "NULL argument");
    *text = nullptr;
    *text_bytes = 0;
    if (n_queries) *n_queries = 0;
    if (n_reads < 0) return fail(SBWTGPU_ERR_INVALID_ARG, "negative n_reads");
    if (n_reads > 0 && !read_off) return fail(SBWTGPU_ERR_INVALID_ARG, "NULL input");
    int64_t all_vals = 0;
    for (int64_t r = 0; r < n_reads; r++) all_vals += std::max<int64_t>(0, read_off[r + 1] - read_off[r] - idx->h.k + 1);
    TextAcc acc{nullptr, 0, sbwtgpu_format_text_bound(idx, all_vals, n_reads) + 1};
    acc.buf = (char *)malloc((size_t)acc.cap);
    if (!acc.buf) return fail(SBWTGPU_ERR_OOM, "out of host memory");
    const int rc = sbwtgpu_search_text_stream(idx, bases, read_off, n_reads, streaming, text_acc_sink, &acc, n_queries);
    if (rc != SBWTGPU_OK) { free(acc.buf); return rc; }
    char *shrunk = (char *)realloc(acc.buf, (size_t)(acc.len ? acc.len : 1));
    *text = shrunk ? shrunk : acc.buf;
    *text_bytes = acc.len;
    return SBWTGPU_OK;
}

}  // extern "C"
