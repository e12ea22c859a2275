// sbwt_search_pool.hip -- k_search_pool: the path-order search (see sbwt_search.hip, k_search_cert<PATH>) with the
// reads kept in a per-wave POOL in LDS and the wave's lanes re-assigned every iteration.
//
// Why.  k_search_cert gives a lane one read and walks it as a state machine; in every iteration the lanes of a wave
// are in five or six different states, so the wave executes the code of ALL of them with a fraction of its lanes
// each: rocprofv3 on config 2 shows 3.3e9 vector instructions per launch with 36 of 64 lanes active on average, the
// vector ALU 82 % busy (SQ_ACTIVE_INST_VALU) -- the kernel is bound by instruction issue as much as by memory
// (profiles/r02_base_*).  Here a wave owns POOL reads (their state: 80 bytes + 16 staged results each, in LDS).  Every
// iteration it counts the reads per state, picks ONE state, hands up to 64 reads in that state to its 64 lanes, and runs
// only that state's code, with all lanes active; the lanes load the reads' state from LDS, do the step (one memory
// round trip, as before), and store the state back.  Same steps, same order per read, same results as k_search_cert
// -- the per-read logic below is that kernel's, specialised to 32-bit positions and the path order; what changed is
// which lane runs which read when.
//
// A workgroup of four waves shares one pool: a wave CLAIMS the reads it picked (compare-and-swap on the slot's state
// word in LDS), so that 16 waves per CU hide the memory round trips while every wave still draws from 256 reads.
// There is no barrier in the loop: a slot is touched only by the wave that holds its claim.
//
// Reference semantics: SBWT::streaming_search include/sbwt/SBWT.hh:544-581, SBWT::search :389-415,
// SBWT::update_sbwt_interval :422-437 (see sbwt_search.hip for the restated rules and the certificates).
#include "sbwt_kernels_common.h"

#define P_IDLE 0                // free slot (a new read can move in)
#define P_FETCH 1               // ragged batches: the read's offsets are being fetched
#define P_INIT 2                // start of a walk: one table lookup (kind in wk)
#define P_STEP 3                // one interval update (SBWT.hh:430-431)
#define P_EXT 4                 // follow the path from position r while the read agrees with it
#define P_TRANS 5               // the read left the path at position r: the streaming step, from the transition table
#define P_POS 6                 // r = pos[l]  (a k-mer was found by a walk: onto its path)
#define P_BRIDGE 7              // the read differs from the path at a substitution-safe base: do the next k-1 agree?
#define P_NMODES 8
#define P_DEAD 8                // no read will ever live here again

#define PEV_NONE 0
#define PEV_EMIT1 1
#define PEV_FAIL 2
#define PEV_END 3
#define PEV_PRES 4
#define PK_NONE 0
#define PK_RELOAD 2
#define PK_MODE 3
#ifndef SBWT_POOL_PIPE
#define SBWT_POOL_PIPE 2        // descriptors per group of 16 lanes and trip (4 needs more than 128 registers)
#endif

#define P_BUSY 0x80u            // or-ed into a slot's state word while a wave works on it

template <int POOL, int SD>
__global__ void __launch_bounds__(256, 4) k_search_pool(SbwtIndexView ix, const uint4 *__restrict__ packed,
                                                    const i64 *__restrict__ read_off, const i64 *__restrict__ out_off,
                                                    i64 *__restrict__ out, i64 n_reads, SbwtWorkHeader *ws, int streaming) {
    constexpr int SPL = POOL / 64;                    // slots every lane looks at in the census
    static_assert(POOL % 64 == 0 && POOL <= 256, "pool slots are addressed with 8 bits");
    static_assert(SD == 8 || SD == 16, "staged results: one 64-byte or one 128-byte line of `out`");
    // read state: q0 = { i, m, pgrp, poff | j << 5 | wk << 16 | g1ok << 19 | cnt << 20 }, q1 = { l, r, b, blo },
    // q2 = { wstart, tag, obase lo, obase hi }, q3 = g0, q4 = g1 (the two cached packed groups of the read)
    __shared__ uint4 st[5][POOL];
    __shared__ unsigned stage[SD][POOL];              // results not written yet (they leave as whole lines of `out`)
    __shared__ unsigned modes[POOL];                  // state word of every slot: P_* (| P_BUSY while claimed)
    __shared__ unsigned char sel_all[4][64];
    __shared__ uint4 desc_all[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned char *sel = sel_all[wv];
    uint4 *desc = desc_all[wv];
    const int k = ix.k, p = ix.p_dev, L0 = ix.probe_len;
    const int ps = ix.p_sparse;
    const bool pfon = ix.pfil && ix.p_filter == L0 && L0 > p;
    const u64 m2 = (k - ps >= 32) ? ~0ull : low_mask(2 * ((k - ps) & 31));
    const int pw = pfon ? L0 : p;
    const int last_node = (int)(ix.n_nodes - 1);
    const bool uni = ws->u_bad == 0 && ws->u_len > 0;
    const i64 u_read0 = ws->u_read0, u_len = ws->u_len, u_out0 = ws->u_out0, u_stride = ws->u_stride;
    const unsigned lmask = SD - 1;                    // results per line of `out`
    const u64 lt = low_mask(lane);
    unsigned c_ext = 0, c_brg = 0;                                // per lane
    unsigned c_stream = 0, c_search = 0, c_lf = 0, c_tab = 0;     // wave-uniform
    unsigned c_steps = 0, c_short = 0;
    bool exhausted = false;                                       // wave-uniform: the ticket counter ran past the last read

    for (int u = threadIdx.x; u < POOL; u += 256) modes[u] = P_IDLE;
    __syncthreads();                                   // the only barrier: from here on the waves run on their own

    for (;;) {
        // ---- census: reads per state; pick one state for this iteration ----
        unsigned mm[SPL];
#pragma unroll
        for (int u = 0; u < SPL; u++) mm[u] = __hip_atomic_load(&modes[lane + 64 * u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        int M = -1, best = 0, c_idle = 0;
#pragma unroll
        for (int X = 0; X < P_NMODES; X++) {
            int c = 0;
#pragma unroll
            for (int u = 0; u < SPL; u++) c += __popcll(__ballot(mm[u] == (unsigned)X));
            if (X == P_IDLE) { c_idle = exhausted ? 0 : c; continue; }
            if (c > best) { best = c; M = X; }
        }
        // refill the pool as soon as a good part of a wave's worth of slots is free (cheap: no memory round trip for
        // fixed-length reads), otherwise the state most reads are in
        if (c_idle >= 32 || (best == 0 && c_idle > 0)) M = P_IDLE;
        if (M < 0) {
            // nothing to pick: done, unless another wave still holds reads (they may come back in any state)
            bool busy = false;
#pragma unroll
            for (int u = 0; u < SPL; u++) busy = busy || (mm[u] & P_BUSY);
            if (__ballot(busy) == 0) break;
            __builtin_amdgcn_s_sleep(16);
            continue;
        }
        int n = 0, slot = 0;
        {
            // candidates in slot order, starting with this wave's own quarter of the pool (the waves of a workgroup pick
            // at the same time: they should not all reach for the same reads)
            int before = 0;
#pragma unroll
            for (int uu = 0; uu < SPL; uu++) {
                const int u = (uu + wv) % SPL;
                unsigned mine = mm[0];
#pragma unroll
                for (int q = 1; q < SPL; q++) mine = (u == q) ? mm[q] : mine;
                const bool is = mine == (unsigned)M;
                const u64 msk = __ballot(is);
                const int rk = before + __popcll(msk & lt);
                if (is && rk < 64) sel[rk] = (unsigned char)(lane + 64 * u);
                before += __popcll(msk);
            }
            n = before < 64 ? before : 64;
        }
        slot = (lane < n) ? (int)sel[lane] : 0;
        // claim: only the wave whose compare-and-swap succeeds touches the slot until it publishes the next state
        bool act = false;
        if (lane < n) {
            unsigned expect = (unsigned)M;
            act = __hip_atomic_compare_exchange_strong(&modes[slot], &expect, (unsigned)M | P_BUSY, __ATOMIC_ACQUIRE,
                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        n = __popcll(__ballot(act));
        if (n == 0) continue;

        // ---- this lane's read ----
        int mode = act ? M : P_DEAD;
        uint4 q0 = st[0][slot], q1 = st[1][slot], q2 = st[2][slot];
        uint4 g0 = st[3][slot], g1 = st[4][slot];
        int i = (int)q0.x, m = (int)q0.y, pgrp = (int)q0.z;
        int poff = (int)(q0.w & 31u), j = (int)((q0.w >> 5) & 0x7FFu), wk = (int)((q0.w >> 16) & 7u), cnt = (int)(q0.w >> 20);
        bool g1ok = ((q0.w >> 19) & 1u) != 0;
        int l = (int)q1.x, r = (int)q1.y, b = (int)q1.z, blo = (int)q1.w;
        int wstart = (int)q2.x, tag = (int)q2.y;
        i64 obase = (i64)((u64)q2.z | ((u64)q2.w << 32));

        if (M == P_IDLE) {
            // ---- new reads move in ----
            u64 t = 0;
            if (lane == 0) t = atomicAdd(&ws->ticket, (u64)n);
            const u64 first = uniform64(t);
            if (act) {
                const i64 rd = (i64)(first + (u64)__popcll(__ballot(act) & lt));
                cnt = 0;
                tag = -2;
                g1ok = false;
                if (rd >= n_reads) {
                    mode = P_DEAD;
                } else if (uni) {
                    const i64 P0 = u_read0 + rd * u_len;
                    obase = u_out0 + rd * u_stride;
                    pgrp = (int)(P0 >> 5);
                    poff = (int)(P0 & 31);
                    m = (int)u_len - k + 1;
                    i = 0; b = -1; blo = -1; wstart = 0; j = 0;
                    wk = (ps > 0) ? 1 : 0;
                    if (m <= 0) mode = P_IDLE;
                    else if (p > 0) mode = P_INIT;
                    else { mode = P_STEP; l = 0; r = last_node; }
                } else {
                    mode = P_FETCH;
                    obase = rd;                        // until the offsets are here
                }
            }
            if (first + (u64)n >= (u64)n_reads) exhausted = true;
        } else {
            // ---- what does this lane gather?  Two 16-byte loads (+ the prefetch of the read's next packed group) ----
            int kind = PK_NONE, ev = PEV_NONE, tfail = 0, c = 0, grp = 0;
            const uint4 *a1 = ix.blocks, *a2 = ix.blocks;
            int res = -1;
            u64 hk = 0;
            const bool ext = (mode == P_EXT), trn = (mode == P_TRANS), brg = (mode == P_BRIDGE);
            bool rknown = false, qshort = false;
            const bool use_q = ix.trans_ext > 0 || (ix.trans_ext < 0 && 4u * c_short >= c_steps);
            int tnext = P_EXT;
            int tpos = -1;
            int seg_n = 0;
            unsigned seg_src = 0;
            bool do_plan = false, force = false;
            if (M == P_FETCH) {
                if (act) {
                    kind = PK_MODE;
                    a1 = reinterpret_cast<const uint4 *>(read_off + obase);
                    a2 = reinterpret_cast<const uint4 *>(out_off + obase);
                }
            } else if (M == P_POS) {
                if (act) {
                    kind = PK_MODE;
                    a1 = reinterpret_cast<const uint4 *>(ix.pos + ((unsigned)l & ~3u));
                    a2 = a1;
                }
            } else if (act) {
                const int woff = (mode == P_INIT && wk == 5) ? ps : 0;
                const int P = poff + ((ext || trn) ? (i + k - 1) : brg ? (i + k) : ((mode == P_INIT) ? (wstart + woff) : (wstart + j)));
                const int s = P & 31;
                const int wl = (wk == 1) ? ps : (wk == 2) ? L0 : (wk == 3) ? pw : (wk == 5) ? k - ps : p;
                grp = pgrp + (P >> 5);
                if (grp == tag + 1 && g1ok && (mode != P_INIT || s + wl <= 32)) {
                    g0 = g1;
                    g1ok = false;
                    tag = grp;
                }
                if (grp != tag || (mode == P_INIT && s + wl > 32 && !g1ok)) {
                    kind = PK_RELOAD;
                    a1 = packed + grp;
                    a2 = a1 + 1;
                } else {
                    kind = PK_MODE;
                    const u64 codes0 = quad_bits(g0);
                    c = (int)((unsigned)(codes0 >> (2 * s)) & 3u);
                    if (M == P_EXT || M == P_BRIDGE) {
                        a1 = ix.pq + (((unsigned)r + (brg ? 1u : 0u)) >> 5);
                        a2 = a1 + 1;
                    } else if (M == P_TRANS) {
                        if (((streaming == 2 ? g0.w : g0.z) >> s) & 1u) {
                            a1 = ix.trans + ((4 * (size_t)(unsigned)r + (unsigned)c) << ix.trans_wide);
                            a2 = a1;
                        } else {
                            ev = PEV_EMIT1;            // non-ACGT (after toupper: SBWT.hh:565-568) -> -1
                            b = blo = i + k - 1;
                        }
                    } else if (M == P_INIT) {
                        u64 w = codes0 >> (2 * s);
                        if (s) w |= quad_bits(g1) << (64 - 2 * s);
                        const u64 vr = (((u64)g1.w << 32) | (u64)g0.w) >> s;
                        const u64 vm = low_mask(wl);
                        if ((vr & vm) == vm) {
                            if (wk == 1) {
                                const u64 key = w & low_mask(2 * ps);
                                hk = key;
                                const u64 bkt = (((key * SBWT_SP_HASH) >> (64 - ix.log2b)) + (u64)j) & low_mask(ix.log2b);
                                a1 = ix.stab + 2 * bkt;
                                a2 = a1 + 1;
                            } else if (wk == 5) {
                                hk = w & m2;
                                const u64 bkt = ((sp2_hash((unsigned)l, hk) >> (64 - ix.log2b2)) + (u64)j) & low_mask(ix.log2b2);
                                a1 = ix.stab2 + 2 * bkt;
                                a2 = a1 + 1;
                            } else if (wk == 2 || (wk == 3 && pfon)) {
                                const u64 h = sbwt_pf_hash(w & low_mask(2 * L0));
                                hk = (u64)sbwt_pf_bits(h);
                                a1 = ix.pfil + (h >> (64 - ix.log2f));
                                a2 = a1;
                            } else {
                                a1 = reinterpret_cast<const uint4 *>(ix.ptab + (w & low_mask(2 * p)));
                                a2 = a1;
                            }
                        } else {
                            ev = PEV_FAIL;             // a non-ACGT char inside the table window (SBWT.hh:398-399)
                            tfail = wstart + woff + (__ffsll((i64)(~vr & vm)) - 1);
                        }
                    } else {   // P_STEP
                        if ((g0.w >> s) & 1u) {
                            a1 = ix.blocks + ((((i64)l >> 6) << 2) + c);
                            a2 = ix.blocks + (((((i64)r + 1) >> 6) << 2) + c);
                        } else {
                            ev = PEV_FAIL;             // SBWT.hh:427-428
                            tfail = wstart + j;
                        }
                    }
                }
            }
            const bool have = (kind == PK_MODE && ev == PEV_NONE);
            if (M == P_INIT || M == P_STEP)
                c_search = uniform32(c_search + (unsigned)__popcll(__ballot(kind == PK_MODE && (mode == P_INIT || (p == 0 && mode == P_STEP && j == 0)))));
            if (M == P_STEP) c_lf = uniform32(c_lf + (unsigned)__popcll(__ballot(have)));
            if (M == P_EXT || M == P_TRANS) c_steps = uniform32(c_steps + (unsigned)n);

            // ---- the one memory round trip of this iteration ----
            const bool pf = !g1ok && kind == PK_MODE && M != P_FETCH && M != P_POS;
            const uint4 *a3 = pf ? (packed + (tag + 1)) : a1;
            const uint4 v1 = *a1;
            const uint4 v2 = *a2;
            const uint4 v3 = *a3;
            if (pf) { g1 = v3; g1ok = true; }

            // ---- consume ----
            bool tabhit = false;
            bool imprecise = false;
            int burst_to = -1;
            if (kind == PK_RELOAD) {
                g0 = v1;
                g1 = v2;
                g1ok = true;
                tag = grp;
            } else if (M == P_FETCH) {
                if (act) {
                    const i64 P0 = (i64)quad_bits(v1);
                    obase = (i64)quad_bits(v2);
                    pgrp = (int)(P0 >> 5);
                    poff = (int)(P0 & 31);
                    m = (int)((i64)((u64)v1.z | ((u64)v1.w << 32)) - P0) - k + 1;
                    i = 0; b = -1; blo = -1;
                    if (m > 0) { do_plan = true; force = true; }
                    else mode = P_IDLE;
                }
            } else if (M == P_POS) {
                if (act) {
                    const unsigned sl = (unsigned)l & 3u;
                    r = (int)(sl == 0 ? v1.x : sl == 1 ? v1.y : sl == 2 ? v1.z : v1.w);
                    mode = P_EXT;
                }
            } else if (M == P_TRANS) {
                if (have) {
                    ev = PEV_EMIT1;
                    if (v1.x == 0xFFFFFFFFu) {
                        b = blo = i + k - 1;
                    } else {
                        res = (int)v1.x;
                        r = (int)v1.y;
                        rknown = true;
                        const int P1 = poff + i + k, s1 = P1 & 31;
                        if (use_q && ((s1 != 0 && (s1 <= 24 || g1ok)) || (s1 == 0 && g1ok))) {
                            u64 rw, rv;
                            const u64 va = (streaming == 2) ? (((u64)g1.w << 32) | (u64)g0.w) : (((u64)g1.z << 32) | (u64)g0.z);
                            if (s1 != 0) {
                                rw = (quad_bits(g0) >> (2 * s1)) | (quad_bits(g1) << (64 - 2 * s1));
                                rv = va >> s1;
                            } else {
                                rw = quad_bits(g1);
                                rv = va >> 32;
                            }
                            const unsigned x = ((unsigned)rw ^ v1.z) & 0xFFFFu;
                            const unsigned mmk = (x | (x >> 1)) & 0x5555u;
                            const int nm = mmk ? ((__ffs((int)mmk) - 1) >> 1) : 8;
                            const unsigned okb = (unsigned)rv & (v1.z >> 16) & 0xFFu;
                            const int nv = __ffs((int)(~okb | 0x100u)) - 1;
                            int n2 = nm < nv ? nm : nv;
                            bool stop2 = n2 < 8;
                            if (n2 > m - 1 - i) { n2 = m - 1 - i; stop2 = false; }
                            if (n2 > 31 - cnt) { n2 = 31 - cnt; stop2 = false; }
                            if (n2 < 0) n2 = 0;
                            seg_n = n2;
                            seg_src = (unsigned)r + 1u;
                            r += n2;
                            c_ext += (unsigned)n2;
                            if (stop2) tnext = (ix.has_safe && nm < nv && ((v1.z >> 24 >> nm) & 1u)) ? P_BRIDGE : P_TRANS;
                            qshort = stop2;
                        }
                    }
                }
            } else if (M == P_BRIDGE) {
                if (have) {
                    const int P = poff + i + k, s = P & 31, sp = (int)(((unsigned)r + 1u) & 31u);
                    u64 rw = quad_bits(g0) >> (2 * s), pwd = quad_bits(v1) >> (2 * sp);
                    if (s) rw |= quad_bits(g1) << (64 - 2 * s);
                    if (sp) pwd |= quad_bits(v2) << (64 - 2 * sp);
                    const u64 x = rw ^ pwd;
                    const u64 mmk = (x | (x >> 1)) & 0x5555555555555555ull;
                    const int nm = mmk ? ((__ffsll((i64)mmk) - 1) >> 1) : 32;
                    const int need = (k - 1 < m - 1 - i) ? (k - 1) : (m - 1 - i);
                    if (nm >= need) {
                        ev = PEV_FAIL;
                        burst_to = i + need;
                        c_brg++;
                    } else {                               // a safe step has no successor by another char: -1 without a gather
                        ev = PEV_EMIT1;
                        b = blo = i + k - 1;
                    }
                }
            } else if (M == P_EXT) {
                if (have) {
                    const int P = poff + i + k - 1, s = P & 31, sp = (int)((unsigned)r & 31u);
                    u64 rw = quad_bits(g0) >> (2 * s), pwd = quad_bits(v1) >> (2 * sp);
                    if (s) rw |= quad_bits(g1) << (64 - 2 * s);
                    if (sp) pwd |= quad_bits(v2) << (64 - 2 * sp);
                    const u64 rv = ((streaming == 2) ? (((u64)g1.w << 32) | (u64)g0.w) : (((u64)g1.z << 32) | (u64)g0.z)) >> s;
                    const u64 fA = (((u64)v2.z << 32) | (u64)v1.z) >> sp, fB = (((u64)v2.w << 32) | (u64)v1.w) >> sp;
                    const u64 pg = ~fA | fB;               // go = ~A | B, safe = A & B (k_path_reencode)
                    const u64 x = rw ^ pwd;
                    const u64 mmk = (x | (x >> 1)) & 0x5555555555555555ull;
                    const int nm = mmk ? ((__ffsll((i64)mmk) - 1) >> 1) : 32;
                    const u64 bad = ~(rv & pg) | (1ull << 32);
                    const int nv = __ffsll((i64)bad) - 1;
                    int nn = nm < nv ? nm : nv;
                    bool stopped = nn < 32;
                    if (nn > m - i) nn = m - i;
                    if (nn > 32 - cnt) { nn = 32 - cnt; stopped = false; }
                    seg_n = nn;
                    seg_src = (unsigned)r + 1u;
                    r += nn;
                    c_ext += (unsigned)nn;
                    bool sbit = false;
                    if (stopped && nm < nv) sbit = (((fA & fB) >> nm) & 1ull) != 0;
                    qshort = stopped && nn < 8;
                    if (i + nn == m) mode = P_IDLE;
                    else if (stopped) mode = sbit ? P_BRIDGE : P_TRANS;
                }
            } else if (M == P_INIT) {
                if (have) {
                    int wl = p;
                    bool again = false;
                    const bool viaf = (wk == 2) || (wk == 3 && pfon);
                    if (wk == 1 || wk == 5 || viaf) {
                        if (viaf) {
                            const unsigned b1 = (unsigned)hk & 127u, b2 = ((unsigned)hk >> 7) & 127u;
                            const unsigned w1 = (b1 < 64) ? (b1 < 32 ? v1.x : v1.y) : (b1 < 96 ? v1.z : v1.w);
                            const unsigned w2 = (b2 < 64) ? (b2 < 32 ? v1.x : v1.y) : (b2 < 96 ? v1.z : v1.w);
                            wl = L0;
                            if (((w1 >> (b1 & 31u)) & (w2 >> (b2 & 31u)) & 1u) != 0) {
                                if (wk == 3) {
                                    l = 0;
                                } else {
                                    again = true;
                                    wk = 0;
                                }
                            } else {
                                l = -1;
                            }
                        } else if (wk == 5) {
                            wl = k;
                            if ((v1.w & SBWT_SP2_USED) && quad_bits(v1) == hk && v1.z == (unsigned)l) {
                                l = (int)v2.x;
                                r = l;
                                tpos = (int)v2.y;
                            } else if (v1.w & SBWT_SP2_OVERFLOW) {
                                again = true;
                                j++;
                            } else {
                                l = -1;
                            }
                        } else {
                            const u64 key = hk;
                            const u64 w0 = quad_bits(v1), w1 = quad_bits(v2);
                            const bool m0 = (w0 & ~SBWT_SP_OVERFLOW) == key, m1 = w1 == key;
                            wl = ps;
                            if (m0 | m1) {
                                l = (int)(m0 ? v1.z : v2.z);
                                if (ix.stab_pos) {
                                    r = l;
                                    tpos = (int)(m0 ? v1.w : v2.w);
                                } else {
                                    r = l + (int)(m0 ? v1.w : v2.w);
                                }
                            } else if (w0 & SBWT_SP_OVERFLOW) {
                                again = true;
                                j++;
                            } else {
                                l = -1;
                            }
                        }
                    } else {
                        l = (int)(i64)quad_bits(v1);
                        r = (int)(i64)((u64)v1.z | ((u64)v1.w << 32));
                    }
                    if (!again) {
                        tabhit = (l != -1);
                        if (l == -1) {
                            ev = PEV_FAIL;
                            tfail = wstart + wl - 1;
                            imprecise = (wk != 2);
                        } else if (wk == 3) {
                            ev = PEV_PRES;
                        } else if (wk == 1 && ps < k && ix.stab2) {
                            wk = 5;
                            j = 0;
                        } else {
                            j = wl;
                            if (wstart + j == i + k) ev = PEV_END;
                            else mode = P_STEP;
                        }
                    }
                }
            } else {   // P_STEP
                if (have) {
                    l = (int)v1.z + (int)__popcll(quad_bits(v1) & low_mask(l & 63));
                    r = (int)v2.z + (int)__popcll(quad_bits(v2) & low_mask((r + 1) & 63)) - 1;
                    if (l > r) {
                        ev = PEV_FAIL;
                        tfail = wstart + j;
                    } else if (wstart + (++j) == i + k) {
                        ev = PEV_END;
                    }
                }
            }
            if (M == P_INIT) c_tab = uniform32(c_tab + (unsigned)__popcll(__ballot(tabhit)));
            if (M == P_TRANS) {
                c_stream = uniform32(c_stream + (unsigned)__popcll(__ballot(ev == PEV_EMIT1)));
                c_short = uniform32(c_short + (unsigned)__popcll(__ballot(qshort)));
            }
            if (M == P_EXT) c_short = uniform32(c_short + (unsigned)__popcll(__ballot(qshort)));

            if (M != P_FETCH && M != P_POS) {
                // ---- events: results, certificates, next state ----
                int burst_hi = -1;
                if (ev == PEV_END) {
                    if (wstart == i) {
                        res = l;
                        if (l != r) ws->status = SBWT_ERR_NOT_SINGLETON;   // SBWT.hh:410-413
                        if (tpos >= 0) { r = tpos; rknown = true; }
                        ev = PEV_EMIT1;
                        b = -1;
                    } else {
                        do_plan = true;
                        force = true;
                    }
                } else if (ev == PEV_FAIL) {
                    burst_hi = (wstart < m - 1) ? wstart : (m - 1);
                    if (burst_to >= 0) {
                        burst_hi = burst_to;
                        b = -1;
                    } else if (wk == 3) {
                        if (wstart >= b) b = -1;
                        else if (blo < wstart + 1) blo = wstart + 1;
                    } else if (imprecise && !(wstart == b && blo >= b)) {
                        blo = wstart;
                        b = tfail;
                    } else {
                        b = (wstart == b) ? -1 : tfail;
                        blo = b;
                    }
                    if (burst_hi == i) { ev = PEV_EMIT1; burst_hi = -1; }
                }
                if (ev == PEV_PRES) {
                    const int lo = blo > i ? blo : i;
                    if (wstart > lo) b = wstart - 1;
                    else blo = wstart + pw;
                    if (blo > b) b = -1;
                    do_plan = true;
                }
                if (ev == PEV_EMIT1) {
                    stage[cnt][slot] = (unsigned)res;
                    cnt++;
                    i++;
                }
                // ---- result writes: whole lines, written by groups of 16 lanes (see k_search_cert) ----
                {
                    int nleft = (burst_hi >= 0) ? (burst_hi - i + 1) : seg_n;
                    unsigned s2 = (burst_hi >= 0) ? 0xFFFFFFFFu : seg_src;
                    const int sub = lane & 15, grpl = lane >> 4;
                    for (;;) {
                        const i64 dst0 = obase + (i - cnt);
                        const int nn = nleft < 32 - cnt ? nleft : 32 - cnt;
                        const int total = cnt + nn;
                        const bool end = (i + nn == m);
                        const int over = (int)((unsigned)(dst0 + total) & lmask);
                        const bool post = act && (nn > 0 || (cnt > 0 && (end || over == 0)));
                        const int w = !post ? 0 : (end ? total : (over <= total ? total - over : 0));
                        const u64 pm = __ballot(post);
                        if (pm == 0) break;
                        const int ndesc = __popcll(pm);
                        if (post)
                            desc[__popcll(pm & lt)] = make_uint4((unsigned)dst0, (unsigned)((u64)dst0 >> 32),
                                                                 (unsigned)cnt | ((unsigned)total << 8) | ((unsigned)w << 16) | ((unsigned)slot << 24), s2);
                        for (int base = 0; base < ndesc; base += 4 * SBWT_POOL_PIPE) {
                            constexpr int PIPE = SBWT_POOL_PIPE;
                            i64 dd[PIPE];
                            int sa[PIPE], sb[PIPE], ca[PIPE], cb[PIPE], fa[PIPE], fb[PIPE], wl[PIPE], tl[PIPE], ht[PIPE];
#pragma unroll
                            for (int u = 0; u < PIPE; u++) {
                                const int idx = base + 4 * u + grpl;
                                const bool on = idx < ndesc && !(ix.debug & 1);
                                const uint4 ds = desc[idx < ndesc ? idx : 0];
                                const int dc = (int)(ds.z & 0xFFu), dt = (int)((ds.z >> 8) & 0xFFu);
                                const int j0 = 2 * sub;
                                ht[u] = (int)(ds.z >> 24);
                                wl[u] = on ? (int)((ds.z >> 16) & 0xFFu) - j0 : 0;
                                tl[u] = on ? dt - j0 : 0;
                                dd[u] = (i64)((u64)ds.x | ((u64)ds.y << 32)) + j0;
                                const bool isc = ds.w != 0xFFFFFFFFu && on;
                                const int r0 = j0 < SD - 1 ? j0 : SD - 2;
                                const unsigned c0 = (isc && j0 >= dc && tl[u] > 0) ? ds.w + (unsigned)(j0 - dc) : 0u;
                                const unsigned c1 = (isc && j0 + 1 >= dc && tl[u] > 1) ? ds.w + (unsigned)(j0 + 1 - dc) : 0u;
                                sa[u] = (int)stage[r0][ht[u]];
                                sb[u] = (int)stage[r0 + 1][ht[u]];
                                ca[u] = (int)ix.col[c0];
                                cb[u] = (int)ix.col[c1];
                                fa[u] = (j0 < dc) ? 0 : (isc ? 1 : 2);
                                fb[u] = (j0 + 1 < dc) ? 0 : (isc ? 1 : 2);
                            }
#pragma unroll
                            for (int u = 0; u < PIPE; u++) {
                                const int va = fa[u] == 0 ? sa[u] : (fa[u] == 1 ? ca[u] : -1);
                                const int vb = fb[u] == 0 ? sb[u] : (fb[u] == 1 ? cb[u] : -1);
                                if (wl[u] >= 2) st_stream2(out + dd[u], (i64)va, (i64)vb);
                                else if (wl[u] == 1) st_stream(out + dd[u], (i64)va);
                                if (tl[u] > 0 && wl[u] < 1) stage[-wl[u]][ht[u]] = (unsigned)va;
                                if (tl[u] > 1 && wl[u] < 2) stage[1 - wl[u]][ht[u]] = (unsigned)vb;
                            }
                        }
                        cnt = total - w;
                        i += nn;
                        nleft -= nn;
                        if (s2 != 0xFFFFFFFFu) s2 += (unsigned)nn;
                        if (__ballot(nleft > 0) == 0) break;
                    }
                }
                if (ev == PEV_EMIT1 || burst_hi >= 0) {
                    if (i == m) {
                        mode = P_IDLE;
                    } else if (ev == PEV_EMIT1 && res != -1 && streaming) {
                        mode = rknown ? tnext : P_POS;       // SBWT.hh:560-
                        l = res;
                    } else {
                        do_plan = true;                      // SBWT.hh:557-559 (with certificates)
                    }
                }
            }
            if (do_plan) {
                int s0 = i, nwk = (ps > 0) ? 1 : 0;
                if (!force && L0 > 0 && b >= i && b <= i + k - 1) {
                    const int lo = blo > i ? blo : i;
                    if (lo < b && p > 0 && k - pw >= 1) {
                        int x = lo + ((b - lo + 1) >> 1);
                        if (x > i + k - pw) x = i + k - pw;
                        if (x <= i) x = i + 1;
                        s0 = x;
                        nwk = 3;
                    } else {
                        s0 = (b - i >= L0 - 1) ? (b - L0 + 1) : b;
                        if (s0 + p - 1 > i + k - 1) s0 = i;
                        if (s0 != i) nwk = (pfon && s0 + L0 - 1 <= i + k - 1) ? 2 : 0;
                    }
                }
                wstart = s0;
                j = 0;
                wk = nwk;
                if (p > 0) mode = P_INIT;
                else { mode = P_STEP; l = 0; r = last_node; }
            }
        }

        // ---- the read goes back into the pool ----
        if (act) {
            st[0][slot] = make_uint4((unsigned)i, (unsigned)m, (unsigned)pgrp,
                                     (unsigned)poff | ((unsigned)j << 5) | ((unsigned)wk << 16) | ((g1ok ? 1u : 0u) << 19) | ((unsigned)cnt << 20));
            st[1][slot] = make_uint4((unsigned)l, (unsigned)r, (unsigned)b, (unsigned)blo);
            st[2][slot] = make_uint4((unsigned)wstart, (unsigned)tag, (unsigned)(u64)obase, (unsigned)((u64)obase >> 32));
            st[3][slot] = g0;
            st[4][slot] = g1;
            __hip_atomic_store(&modes[slot], (unsigned)mode, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }

    {
        u64 e = c_ext, eb = c_brg;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { e += __shfl_down(e, off); eb += __shfl_down(eb, off); }
        if (lane == 0) {
            atomicAdd(&ws->n_ext, e);
            atomicAdd(&ws->n_bridge, eb);
            atomicAdd(&ws->n_stream, (u64)c_stream);
            atomicAdd(&ws->n_search, (u64)c_search);
            atomicAdd(&ws->n_lf, (u64)c_lf);
            atomicAdd(&ws->n_tab_hit, (u64)c_tab);
        }
    }
}

// Four waves per workgroup share a pool of 256 reads (34 KB of LDS): 4 workgroups = 16 waves per CU.
void sbwt_launch_search_pool(const SbwtIndexView &ix, const uint4 *d_packed, const long long *d_read_off,
                             const long long *d_out_off, long long *d_out, long long n_reads, SbwtWorkHeader *ws,
                             int streaming, hipStream_t stream) {
    if (n_reads <= 0) return;
    i64 want = (n_reads + 255) / 256;
    unsigned cap = (ix.debug >> 8) ? (unsigned)(ix.debug >> 8) : 1024u;
    unsigned g = (unsigned)(want < (i64)cap ? want : (i64)cap);
    if (ix.debug & 32)
        hipLaunchKernelGGL((k_search_pool<256, 16>), dim3(g), dim3(256), 0, stream, ix, d_packed, d_read_off, d_out_off, d_out,
                           (i64)n_reads, ws, streaming);
    else
        hipLaunchKernelGGL((k_search_pool<256, 8>), dim3(g), dim3(256), 0, stream, ix, d_packed, d_read_off, d_out_off, d_out,
                           (i64)n_reads, ws, streaming);
}
