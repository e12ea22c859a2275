// sbwt_build.hip -- construction on the device.
//   (1) the interleaved 64-column blocks of the device image from the four bit vectors (+ suffix_group_starts):
//       what SubsetMatrixRank's constructor does with sdsl::util::init_support (SubsetMatrixRank.hh:52-58) and
//       SBWT's constructor with the C array (SBWT.hh:344-349), for this layout;
//   (2) the plain-matrix SBWT columns themselves from sequences, k <= 64: every k-mer packed so that integer order is
//       the colexicographic order (Kmer.hh:108-123), radix sort (rocPRIM), predecessors / edges / suffix groups by
//       searches in the sorted array, dummy prefixes of the predecessor-less k-mers, merged emission of the columns --
//       the node and edge rules of NodeBOSSInMemoryConstructor.hh:98-213 (edges only on suffix-group starts; k-mers
//       without a predecessor get all their proper prefixes as dummy nodes; the empty root always exists; colex
//       order with the shorter label first), bit-identical to host/index_builder.hh and to the oracle's literal
//       restatement of that constructor (tests/test_gpu_build.py).
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include "sbwt_kernels_common.h"
#include "sbwt_scan.h"

// ---------------------------------------------------------------------------------------------
// (1) blocks
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 masked_word(const u64 *__restrict__ v, i64 w, i64 nw, i64 n) {
    if (w >= nw) return 0;
    u64 x = v[w];
    if (w == nw - 1 && (n & 63)) x &= (~0ull) >> (64 - (n & 63));
    return x;
}
// ones of every block, per row: cnt[c * (nb + 1) + b]; and the number of columns that carry two or more chars
// (suffix-group starts with a choice of successors), summed into *n_multi
__global__ void __launch_bounds__(256) k_blk_count(const u64 *__restrict__ bits, i64 nw, i64 n, i64 nb, i64 *__restrict__ cnt,
                                                   unsigned long long *n_multi) {
    const i64 b = (i64)blockIdx.x * 256 + threadIdx.x;
    u64 w[4] = {0, 0, 0, 0};
    if (b < nb) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
            w[c] = masked_word(bits + (i64)c * nw, b, nw, n);
            cnt[(i64)c * (nb + 1) + b] = __popcll(w[c]);
        }
    }
    unsigned multi = (unsigned)__popcll((w[0] & w[1]) | (w[0] & w[2]) | (w[0] & w[3]) | (w[1] & w[2]) | (w[1] & w[3]) | (w[2] & w[3]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) multi += __shfl_down(multi, off);
    if ((threadIdx.x & 63) == 0 && multi) atomicAdd(n_multi, (unsigned long long)multi);
}
// blocks[b] = four quads { bits lo, bits hi, C[c] + ones before the block - mega base, suffix-group piece }
__global__ void __launch_bounds__(256) k_blk_fill(const u64 *__restrict__ bits, const u64 *__restrict__ ssup, i64 nw, i64 n,
                                                  i64 nb, const i64 *__restrict__ pre, i64 C0, i64 C1, i64 C2, i64 C3,
                                                  int use_mega, int n_mega, uint4 *__restrict__ blocks, u64 *__restrict__ mega) {
    const i64 b = (i64)blockIdx.x * 256 + threadIdx.x;
    if (b >= nb) return;
    const i64 bpm = (i64)1 << (SBWT_MEGA_SHIFT - 6);
    const i64 mb = b / bpm;
    const u64 s = ssup ? masked_word(ssup, b, nw, n) : 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const i64 Cc = c == 0 ? C0 : c == 1 ? C1 : c == 2 ? C2 : C3;
        const u64 run = (u64)(Cc + pre[(i64)c * (nb + 1) + b]);
        const u64 base = use_mega ? (u64)(Cc + pre[(i64)c * (nb + 1) + mb * bpm]) : 0;
        if (b % bpm == 0) mega[(i64)c * n_mega + mb] = base;
        const u64 w = masked_word(bits + (i64)c * nw, b, nw, n);
        blocks[4 * b + c] = make_uint4((unsigned)w, (unsigned)(w >> 32), (unsigned)(run - base),
                                       (c & 1) ? (unsigned)(s >> 32) : (unsigned)s);
    }
}

long long sbwt_blocks_scratch_bytes(long long n_nodes) {
    const i64 nb = n_nodes / 64 + 1;
    return 2 * 4 * (nb + 1) * 8 + ((nb + 1023) / 1024 + 2) * 8 + 512;
}
// d_bits: the four rows back to back (nw words each); d_ssup may be null.  Step 1 counts and scans; totals[4] (host)
// receives the ones per row (the caller decides about the count layout); step 2 fills blocks and the mega table.
int sbwt_blocks_count(const unsigned long long *d_bits, long long n_nodes, void *d_scratch, long long totals[5], hipStream_t st) {
    const i64 n = n_nodes, nw = (n + 63) / 64, nb = n / 64 + 1;
    i64 *cnt = static_cast<i64 *>(d_scratch), *pre = cnt + 4 * (nb + 1), *bsum = pre + 4 * (nb + 1);
    unsigned long long *n_multi = reinterpret_cast<unsigned long long *>(bsum + ((nb + 1023) / 1024 + 2));
    (void)hipMemsetAsync(n_multi, 0, 8, st);
    hipLaunchKernelGGL(k_blk_count, dim3(grid_for(nb)), dim3(256), 0, st, reinterpret_cast<const u64 *>(d_bits), nw, n, nb, cnt, n_multi);
    const unsigned gb = (unsigned)((nb + 1023) / 1024);
    for (int c = 0; c < 4; c++) {
        hipLaunchKernelGGL(k_scan_block_sums, dim3(gb), dim3(256), 0, st, cnt + (i64)c * (nb + 1), nb, bsum);
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, st, bsum, (i64)gb);
        hipLaunchKernelGGL(k_scan_apply, dim3(gb), dim3(256), 0, st, cnt + (i64)c * (nb + 1), nb, bsum, pre + (i64)c * (nb + 1));
    }
    i64 t[5];
    for (int c = 0; c < 4; c++)
        if (hipMemcpyAsync(&t[c], pre + (i64)c * (nb + 1) + nb, 8, hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
    if (hipMemcpyAsync(&t[4], n_multi, 8, hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
    if (hipStreamSynchronize(st) != hipSuccess) return -1;
    for (int c = 0; c < 5; c++) totals[c] = t[c];
    return 0;
}
void sbwt_blocks_fill(const unsigned long long *d_bits, const unsigned long long *d_ssup, long long n_nodes, void *d_scratch,
                      const long long C[4], int use_mega, int n_mega, uint4 *d_blocks, unsigned long long *d_mega, hipStream_t st) {
    const i64 n = n_nodes, nw = (n + 63) / 64, nb = n / 64 + 1;
    const i64 *pre = static_cast<const i64 *>(d_scratch) + 4 * (nb + 1);
    hipLaunchKernelGGL(k_blk_fill, dim3(grid_for(nb)), dim3(256), 0, st, reinterpret_cast<const u64 *>(d_bits),
                       reinterpret_cast<const u64 *>(d_ssup), nw, n, nb, pre, (i64)C[0], (i64)C[1], (i64)C[2], (i64)C[3], use_mega,
                       n_mega, d_blocks, reinterpret_cast<u64 *>(d_mega));
}

// ---------------------------------------------------------------------------------------------
// (2) plain-matrix SBWT columns from sequences, k <= 64
// ---------------------------------------------------------------------------------------------
// packed text: the k_encode format (sbwt_search.hip): group = { codes lo, codes hi, validU, validRaw }, 32 bases per
// 16 bytes; validRaw = upper-case ACGT, which is what the reference's constructors accept
// (NodeBOSSInMemoryConstructor.hh:156-159).  Sequences are separated by one non-ACGT byte by the caller, so a k-mer
// window that crosses a boundary is invalid like one that holds an N.
// Keys: char i of the k-mer at bits 2i, so that integer order = colexicographic order (Kmer.hh:108-123); one 64-bit
// word for k <= 32, __uint128_t for 32 < k <= 64 (the kernels are templates on the key type; rocPRIM sorts both).
typedef __uint128_t u128;
template <typename KT> __host__ __device__ __forceinline__ KT key_mask(int bits) {       // bits in [0, 8 * sizeof(KT)]
    return bits >= (int)(8 * sizeof(KT)) ? ~(KT)0 : (((KT)1 << bits) - (KT)1);
}
__device__ __forceinline__ u64 rev2(u64 x) {               // reverses the order of the 32 two-bit groups
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    x = ((x >> 8) & 0x00FF00FF00FF00FFull) | ((x & 0x00FF00FF00FF00FFull) << 8);
    x = ((x >> 16) & 0x0000FFFF0000FFFFull) | ((x & 0x0000FFFF0000FFFFull) << 16);
    return (x >> 32) | (x << 32);
}
__device__ __forceinline__ u64 revcomp_key(u64 key, int k) { return (~rev2(key)) >> (64 - 2 * k); }   // complement = 3 - code
__device__ __forceinline__ u128 revcomp_key(u128 key, int k) {
    const u128 r = ((u128)rev2((u64)key) << 64) | (u128)rev2((u64)(key >> 64));
    return (~r) >> (128 - 2 * k);
}
// every valid k-mer start p -> key; with rc also the reverse complement's key.  Output slots from a wave-aggregated
// counter (the order does not matter: they get sorted).
template <typename KT>
__global__ void __launch_bounds__(256) k_bld_extract(const uint4 *__restrict__ packed, i64 n_pos, int k, int rc,
                                                     KT *__restrict__ keys, unsigned long long *__restrict__ counter) {
    const i64 p = (i64)blockIdx.x * 256 + threadIdx.x;
    bool ok = false;
    KT key = 0;
    if (p < n_pos) {
        const uint4 g0 = packed[p >> 5], g1 = packed[(p >> 5) + 1];
        const int s = (int)(p & 31);
        if (sizeof(KT) == 8) {
            u64 w = quad_bits(g0) >> (2 * s);
            if (s) w |= quad_bits(g1) << (64 - 2 * s);
            const u64 vr = (((u64)g1.w << 32) | (u64)g0.w) >> s;
            const u64 vm = low_mask(k >= 64 ? 63 : k) | (k >= 64 ? (1ull << 63) : 0ull);
            ok = (vr & vm) == vm;
            key = (KT)((k == 32) ? w : (w & low_mask(2 * k)));
        } else {
            const uint4 g2 = packed[(p >> 5) + 2];          // (the text is padded: three groups from any position)
            const u128 lo = (u128)quad_bits(g0) | ((u128)quad_bits(g1) << 64);
            u128 w = lo >> (2 * s);
            if (s) w |= (u128)quad_bits(g2) << (128 - 2 * s);
            const u128 va = (u128)g0.w | ((u128)g1.w << 32) | ((u128)g2.w << 64);
            const u128 vr = va >> s;
            const u128 vm = key_mask<u128>(k);
            ok = (vr & vm) == vm;
            key = (KT)(w & key_mask<u128>(2 * k));
        }
    }
    const u64 m = __ballot(ok);
    if (m == 0) return;
    const int lane = threadIdx.x & 63, cnt = __popcll(m), per = rc ? 2 : 1;
    unsigned long long base = 0;
    if (lane == (int)(__ffsll((i64)m) - 1)) base = atomicAdd(counter, (unsigned long long)(cnt * per));
    base = uniform64(__shfl(base, __ffsll((i64)m) - 1));
    if (ok) {
        const i64 at = (i64)base + (i64)__popcll(m & low_mask(lane)) * per;
        keys[at] = key;
        if (rc) keys[at + 1] = revcomp_key(key, k);
    }
}
// flag[i] = first of its run of equal values (shifted right by `shift`: 0 = distinct k-mers, 2 = suffix groups)
template <typename KT>
__global__ void __launch_bounds__(256) k_bld_flag_first(const KT *__restrict__ v, i64 n, int shift, i64 *__restrict__ flag) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || (v[i] >> shift) != (v[i - 1] >> shift)) ? 1 : 0;
}
template <typename KT>
__global__ void __launch_bounds__(256) k_bld_compact(const KT *__restrict__ v, const i64 *__restrict__ flag,
                                                     const i64 *__restrict__ pos, i64 n, KT *__restrict__ out) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i < n && flag[i]) out[pos[i]] = v[i];
}
// group starts: gstart[g] = index of the first k-mer of suffix group g, gsuf[g] = its (k-1)-suffix
template <typename KT>
__global__ void __launch_bounds__(256) k_bld_groups(const KT *__restrict__ km, const i64 *__restrict__ flag,
                                                    const i64 *__restrict__ pos, i64 n, i64 *__restrict__ gstart,
                                                    KT *__restrict__ gsuf) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i < n && flag[i]) { gstart[pos[i]] = i; gsuf[pos[i]] = km[i] >> 2; }
}
template <typename KT>
__device__ __forceinline__ i64 lower_bound_key(const KT *__restrict__ a, i64 n, KT x) {   // first index with a[i] >= x
    i64 lo = 0, hi = n;
    while (lo < hi) { const i64 mid = (lo + hi) >> 1; if (a[mid] < x) lo = mid + 1; else hi = mid; }
    return lo;
}
// predecessor join (NodeBOSSInMemoryConstructor.hh:113-147): k-mer z = y c has an edge from the suffix group of x iff x's
// (k-1)-suffix equals z's (k-1)-prefix y; otherwise z has no predecessor (nopred[z] = 1)
template <typename KT>
__global__ void __launch_bounds__(256) k_bld_pred(const KT *__restrict__ km, i64 nk, int k, const i64 *__restrict__ gstart,
                                                  const KT *__restrict__ gsuf, i64 ng, unsigned *__restrict__ edges,
                                                  i64 *__restrict__ nopred) {
    const i64 z = (i64)blockIdx.x * 256 + threadIdx.x;
    if (z >= nk) return;
    const KT key = km[z];
    const int c = (int)(key >> (2 * k - 2)) & 3;
    const KT pre = (k > 1) ? (key & key_mask<KT>(2 * k - 2)) : (KT)0;
    const i64 g = lower_bound_key<KT>(gsuf, ng, pre);
    if (g < ng && gsuf[g] == pre) {
        atomicOr(&edges[gstart[g]], 1u << c);
        nopred[z] = 0;
    } else {
        nopred[z] = 1;
    }
}
// merged emission: k-mer i goes to column i + #(dummies with label <= key) (a dummy sorts before the k-mer it is a
// prefix-padding of: shorter first, Kmer.hh:108-123), dummy d to column d + #(k-mers with key < label)
__device__ __forceinline__ void put_column(u64 *__restrict__ rows, i64 nw, i64 col, unsigned e, bool start, int ssup) {
    const u64 bit = 1ull << (col & 63);
#pragma unroll
    for (int c = 0; c < 4; c++)
        if (e & (1u << c)) atomicOr(reinterpret_cast<unsigned long long *>(rows + (i64)c * nw + (col >> 6)), (unsigned long long)bit);
    if (ssup && start) atomicOr(reinterpret_cast<unsigned long long *>(rows + 4 * nw + (col >> 6)), (unsigned long long)bit);
}
template <typename KT>
__global__ void __launch_bounds__(256) k_bld_emit_kmers(const KT *__restrict__ km, i64 nk, const unsigned *__restrict__ edges,
                                                        const KT *__restrict__ ddata, i64 nd, u64 *__restrict__ rows, i64 nw,
                                                        int ssup) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= nk) return;
    const KT key = km[i];
    i64 lo = 0, hi = nd;                                   // dummies with label <= key
    while (lo < hi) { const i64 mid = (lo + hi) >> 1; if (ddata[mid] <= key) lo = mid + 1; else hi = mid; }
    const bool start = (i == 0) || ((key >> 2) != (km[i - 1] >> 2));
    put_column(rows, nw, i + lo, edges[i], start, ssup);
}
template <typename KT>
__global__ void __launch_bounds__(256) k_bld_emit_dummies(const KT *__restrict__ ddata, const unsigned *__restrict__ dedges,
                                                          i64 nd, const KT *__restrict__ km, i64 nk, u64 *__restrict__ rows,
                                                          i64 nw, int ssup) {
    const i64 d = (i64)blockIdx.x * 256 + threadIdx.x;
    if (d >= nd) return;
    put_column(rows, nw, d + lower_bound_key<KT>(km, nk, ddata[d]), dedges[d], true, ssup);   // a dummy is its own group
}

static void scan_i64(const i64 *in, i64 n, i64 *out, i64 *bsum, hipStream_t st) {
    const unsigned gb = (unsigned)((n + 1023) / 1024);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(gb), dim3(256), 0, st, in, n, bsum);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, st, bsum, (i64)gb);
    hipLaunchKernelGGL(k_scan_apply, dim3(gb), dim3(256), 0, st, in, n, bsum, out);
}

// Device part of the builder.  Phase A: text (separator-joined) -> sorted distinct k-mers, suffix groups, edges, the
// list of predecessor-less k-mers (returned to the host, which makes and sorts their dummy prefixes: few).  Phase B:
// the columns.  All device memory is owned by SbwtBuildState.
#define BLD_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err = e_; goto fail; } } while (0)

template <typename KT>
static int build_phase_a(const char *h_text, long long n_text, int k, int rc, SbwtBuildState *S, hipStream_t st) {
    hipError_t err = hipSuccess;
    S->k = k; S->rc = rc; S->n_text = n_text;
    const i64 n_groups = (n_text + 31) / 32 + 4;
    const i64 n_pos = n_text;                              // windows that run past the end meet zero validity bits
    const i64 cap = (rc ? 2 : 1) * n_text;
    char *d_text = nullptr; uint4 *d_packed = nullptr; KT *keys = nullptr, *keys2 = nullptr; unsigned long long *d_cnt = nullptr;
    i64 *flag = nullptr, *pos = nullptr, *bsum = nullptr, *gstart = nullptr; KT *gsuf = nullptr;
    void *tmp = nullptr; size_t tmp_bytes = 0;
    SbwtWorkHeader *ws = nullptr;
    unsigned long long h_cnt = 0;
    i64 nv = 0, nk = 0, ng = 0, nn = 0;
    BLD_TRY(hipMalloc((void **)&d_text, (size_t)n_text + 64));
    BLD_TRY(hipMalloc((void **)&d_packed, (size_t)n_groups * 16));
    BLD_TRY(hipMalloc((void **)&ws, sizeof(SbwtWorkHeader)));
    BLD_TRY(hipMalloc((void **)&d_cnt, 8));
    BLD_TRY(hipMemcpyAsync(d_text, h_text, (size_t)n_text, hipMemcpyHostToDevice, st));
    BLD_TRY(hipMemsetAsync(d_cnt, 0, 8, st));
    BLD_TRY(hipMemsetAsync(d_packed, 0, (size_t)n_groups * 16, st));     // (the groups past the text: no valid base)
    sbwt_launch_encode(d_text, n_text, d_packed, ws, st);
    BLD_TRY(hipMalloc((void **)&keys, (size_t)(cap + 1) * sizeof(KT)));
    hipLaunchKernelGGL(k_bld_extract<KT>, dim3(grid_for(n_pos)), dim3(256), 0, st, d_packed, n_pos, k, rc, keys, d_cnt);
    BLD_TRY(hipMemcpyAsync(&h_cnt, d_cnt, 8, hipMemcpyDeviceToHost, st));
    BLD_TRY(hipStreamSynchronize(st));
    (void)hipFree(d_text); d_text = nullptr;
    (void)hipFree(d_packed); d_packed = nullptr;
    nv = (i64)h_cnt;
    if (nv > 0) {
        BLD_TRY(hipMalloc((void **)&keys2, (size_t)nv * sizeof(KT)));
        BLD_TRY(rocprim::radix_sort_keys(nullptr, tmp_bytes, keys, keys2, (size_t)nv, 0, 2 * k, st));
        BLD_TRY(hipMalloc(&tmp, tmp_bytes + 16));
        BLD_TRY(rocprim::radix_sort_keys(tmp, tmp_bytes, keys, keys2, (size_t)nv, 0, 2 * k, st));
        BLD_TRY(hipStreamSynchronize(st));
        (void)hipFree(tmp); tmp = nullptr;
        (void)hipFree(keys); keys = nullptr;
        // distinct k-mers
        BLD_TRY(hipMalloc((void **)&flag, (size_t)(nv + 1) * 8));
        BLD_TRY(hipMalloc((void **)&pos, (size_t)(nv + 2) * 8));
        BLD_TRY(hipMalloc((void **)&bsum, (size_t)((nv + 1023) / 1024 + 2) * 8));
        hipLaunchKernelGGL(k_bld_flag_first<KT>, dim3(grid_for(nv)), dim3(256), 0, st, keys2, nv, 0, flag);
        scan_i64(flag, nv, pos, bsum, st);
        BLD_TRY(hipMemcpyAsync(&nk, pos + nv, 8, hipMemcpyDeviceToHost, st));
        BLD_TRY(hipStreamSynchronize(st));
        KT *km = nullptr;
        BLD_TRY(hipMalloc((void **)&km, (size_t)nk * sizeof(KT)));
        S->km = km;
        hipLaunchKernelGGL(k_bld_compact<KT>, dim3(grid_for(nv)), dim3(256), 0, st, keys2, flag, pos, nv, km);
        BLD_TRY(hipStreamSynchronize(st));
        (void)hipFree(keys2); keys2 = nullptr;
        // suffix groups
        hipLaunchKernelGGL(k_bld_flag_first<KT>, dim3(grid_for(nk)), dim3(256), 0, st, km, nk, 2, flag);
        scan_i64(flag, nk, pos, bsum, st);
        BLD_TRY(hipMemcpyAsync(&ng, pos + nk, 8, hipMemcpyDeviceToHost, st));
        BLD_TRY(hipStreamSynchronize(st));
        BLD_TRY(hipMalloc((void **)&gstart, (size_t)ng * 8));
        BLD_TRY(hipMalloc((void **)&gsuf, (size_t)ng * sizeof(KT)));
        hipLaunchKernelGGL(k_bld_groups<KT>, dim3(grid_for(nk)), dim3(256), 0, st, km, flag, pos, nk, gstart, gsuf);
        // predecessors and edges
        BLD_TRY(hipMalloc((void **)&S->edges, (size_t)nk * 4));
        BLD_TRY(hipMemsetAsync(S->edges, 0, (size_t)nk * 4, st));
        hipLaunchKernelGGL(k_bld_pred<KT>, dim3(grid_for(nk)), dim3(256), 0, st, km, nk, k, gstart, gsuf, ng, S->edges, flag);
        scan_i64(flag, nk, pos, bsum, st);
        BLD_TRY(hipMemcpyAsync(&nn, pos + nk, 8, hipMemcpyDeviceToHost, st));
        BLD_TRY(hipStreamSynchronize(st));
        if (nn > 0) {
            KT *np = nullptr;
            BLD_TRY(hipMalloc((void **)&np, (size_t)nn * sizeof(KT)));
            S->nopred_keys = np;
            hipLaunchKernelGGL(k_bld_compact<KT>, dim3(grid_for(nk)), dim3(256), 0, st, km, flag, pos, nk, np);
        }
        BLD_TRY(hipStreamSynchronize(st));
    }
    S->nk = nk; S->ng = ng; S->n_nopred = nn;
fail:
    (void)hipFree(d_text); (void)hipFree(d_packed); (void)hipFree(ws); (void)hipFree(d_cnt); (void)hipFree(keys); (void)hipFree(keys2);
    (void)hipFree(tmp); (void)hipFree(flag); (void)hipFree(pos); (void)hipFree(bsum); (void)hipFree(gstart); (void)hipFree(gsuf);
    if (err != hipSuccess) { (void)hipGetLastError(); return err == hipErrorOutOfMemory ? -8 : -3; }
    return 0;
}
int sbwt_build_phase_a(const char *h_text, long long n_text, int k, int rc, SbwtBuildState *S, hipStream_t st) {
    S->key_bytes = k <= 32 ? 8 : 16;
    return k <= 32 ? build_phase_a<u64>(h_text, n_text, k, rc, S, st) : build_phase_a<u128>(h_text, n_text, k, rc, S, st);
}

int sbwt_build_copy_nopred(const SbwtBuildState *S, void *h_keys) {
    if (S->n_nopred == 0) return 0;
    return hipMemcpy(h_keys, S->nopred_keys, (size_t)S->n_nopred * (size_t)S->key_bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}

// Phase B: dummies (sorted by (label, length), edges merged; label top-aligned in 2k bits) + k-mers -> the five rows
// (A, C, G, T, suffix_group_starts; nw words each) in host memory.
template <typename KT>
static int build_phase_b(SbwtBuildState *S, const void *h_ddata, const unsigned *h_dedges, long long nd, int ssup,
                         unsigned long long *h_rows, hipStream_t st) {
    hipError_t err = hipSuccess;
    const i64 n = S->nk + nd, nw = (n + 63) / 64;
    KT *ddata = nullptr; u64 *rows = nullptr; unsigned *dedges = nullptr;
    const KT *km = static_cast<const KT *>(S->km);
    BLD_TRY(hipMalloc((void **)&ddata, (size_t)(nd + 1) * sizeof(KT)));
    BLD_TRY(hipMalloc((void **)&dedges, (size_t)(nd + 1) * 4));
    BLD_TRY(hipMalloc((void **)&rows, (size_t)(5 * nw + 1) * 8));
    BLD_TRY(hipMemcpyAsync(ddata, h_ddata, (size_t)nd * sizeof(KT), hipMemcpyHostToDevice, st));
    BLD_TRY(hipMemcpyAsync(dedges, h_dedges, (size_t)nd * 4, hipMemcpyHostToDevice, st));
    BLD_TRY(hipMemsetAsync(rows, 0, (size_t)(5 * nw) * 8, st));
    if (S->nk > 0)
        hipLaunchKernelGGL(k_bld_emit_kmers<KT>, dim3(grid_for(S->nk)), dim3(256), 0, st, km, S->nk, S->edges, ddata, (i64)nd, rows,
                           nw, ssup);
    hipLaunchKernelGGL(k_bld_emit_dummies<KT>, dim3(grid_for(nd)), dim3(256), 0, st, ddata, dedges, (i64)nd, km, S->nk, rows, nw, ssup);
    BLD_TRY(hipMemcpyAsync(h_rows, rows, (size_t)(5 * nw) * 8, hipMemcpyDeviceToHost, st));
    BLD_TRY(hipStreamSynchronize(st));
fail:
    (void)hipFree(ddata); (void)hipFree(dedges); (void)hipFree(rows);
    if (err != hipSuccess) { (void)hipGetLastError(); return err == hipErrorOutOfMemory ? -8 : -3; }
    return 0;
}
int sbwt_build_phase_b(SbwtBuildState *S, const void *h_ddata, const unsigned *h_dedges, long long nd, int ssup,
                       unsigned long long *h_rows, hipStream_t st) {
    return S->key_bytes == 8 ? build_phase_b<u64>(S, h_ddata, h_dedges, nd, ssup, h_rows, st)
                             : build_phase_b<u128>(S, h_ddata, h_dedges, nd, ssup, h_rows, st);
}

void sbwt_build_release(SbwtBuildState *S) {
    (void)hipFree(S->km); (void)hipFree(S->edges); (void)hipFree(S->nopred_keys);
    S->km = nullptr; S->edges = nullptr; S->nopred_keys = nullptr;
}
