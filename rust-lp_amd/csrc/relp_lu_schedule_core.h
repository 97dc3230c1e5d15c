// relp_lu_schedule_core.h -- second half of the device-side refactorisation: from the rows of a triangular factor (as
// relp_lu_factor_core.h leaves them, together with the transposed pattern) to what the persistent pivot kernel solves with.
// [r4] Everything relp_lu.cpp does on the host for a schedule is done here, by ONE workgroup per schedule (the four schedules
// L, U, U', L' are independent: four workgroups):
//   * levels by Kahn's algorithm -- a row is ready when the rows it reads are solved; the frontier is a bitmap, enumerated in
//     ascending order, so a level lists its rows by pivot like relp_lu.cpp: finish_schedule (round 3 relaxed all m rows once
//     per level);
//   * fusion of consecutive levels into groups one pass solves, by local inversion (relp_lu.hpp: fuse_levels, same rules: a
//     group closes at 63 entries per row or `fuse_lanes` lanes; U and U' keep one entry per substitution path and the list of
//     pivots each path runs through, from which the per-pivot `via` lists of the Forrest-Tomlin update are built);
//   * the "ELL by pass" image (relp_lu.hpp: ell_pack): rows ordered by width inside a group (64, 32, .. 1 lanes: a row with n
//     entries takes the next power of two above n), so that every row starts at a multiple of its width and no row straddles a
//     pass of 256 lanes; overflow lists for rows beyond 63 entries; the rows without entries of U / U' as a list;
//     right-hand-side copies compacted for the wide layouts; reach arrays of the hyper-sparse start.
// Same conventions as relp_lu_factor_core.h: PAR_FOR loops that end in barriers, compiled for the device and, serially, for the
// host (tests/cpp/test_lu_device_model.cpp executes the image pass by pass, masked pivots included).
#pragma once
#include "relp_lu_factor_core.h"

namespace relp {

struct LufSchedIn {
    int32_t m;
    const int32_t* ptr; const int32_t* idx; const double* val;      // rows in pivot coordinates
    const int32_t* tptr; const int32_t* tidx;                       // the transposed pattern: who reads pivot k
    const double* diag;                // nullptr: unit diagonal (L, L')
    int32_t keep_trivial;              // rows without entries are kept (U, U': an update may mask them); also: maskable
    int32_t wide;                      // 32-bit slots (index | lg << 24), else 16-bit (index | lg << 13)
    int32_t triv_min;                  // rows without entries are LISTED from this many on (else packed as slots)
    int32_t fuse_lanes;                // lane budget of a fused group (0: a group per level)
    int32_t rhs_cap;                   // right-hand-side copies the layout has room for (wide: compacted; else one per pivot)
    int32_t want_bits;                 // layout 2: rhs_pos and triv_bits as well
};
struct LufSchedWork {                  // one set per schedule
    int32_t* indeg; int32_t* lev; int32_t* order;                   // m each; order: rows sorted by level
    int32_t* lvl_ptr; int32_t* lvl_grp; int32_t* grp_lvl0;          // nlev_cap + 2 each
    int32_t* grp_lane0; int32_t* grp_pass0; int32_t* grp_lanes;     // nlev_cap + 2 each
    uint32_t* bits0; uint32_t* bits1;                               // m / 32 + 2 words each
    int32_t* grp; int32_t* xbeg; int32_t* xlen;                     // m each: group of a packed row; its expanded entries
    int32_t* x_src; double* x_coef; int32_t* x_v0; int32_t* x_vn; int32_t x_cap;      // arena of expanded rows
    int32_t* pool; int32_t pool_cap;                                // pivots on the substitution paths
    int32_t* lg; int32_t* loff;                                     // m each
    int32_t* tmp; int32_t* tmp2;                                    // m + 2 each
    int32_t* ovf_off;                  // m + 1
    int32_t* rhs_id;                   // m
    int32_t* sc;                       // 32 scalars
    int32_t nlev_cap;
};
enum { LUF_D_PASSES = 0, LUF_D_LEVELS, LUF_D_LANES, LUF_D_OVF, LUF_D_BYTES, LUF_D_TRIV, LUF_D_STATUS, LUF_D_NRHS, LUF_D_USES_RHS, LUF_D_VIA,
       LUF_D_KAHN_LEVELS, LUF_D_WORDS = 12 };
struct LufSchedOut {
    char* image; int64_t image_cap;    // the contiguous image (passes | lvl_pass | rdiag | sval | oval | rovf | sidx | oidx)
    int32_t* desc;                     // LUF_D_*
    int32_t* triv; int32_t* reach; int32_t* level_of;               // m each; level_of[p]: a sweep whose right-hand side is zero on p and
                                                                    // on everything solved before p may start at group level_of[p] + 1
    int32_t* via_ptr; int32_t* via_pos; int32_t via_cap;            // m + 1 / via_cap (maskable schedules)
    int32_t* rhs_src;                  // m: the pivots whose right-hand side is copied, ascending (wide layouts)
    int32_t* rhs_pos; uint32_t* triv_bits;                          // layout 2: inverse of rhs_src; `triv` as a bitmap
};

LUF_FN int64_t luf_up16(int64_t b) { return (b + 15) / 16 * 16; }
// byte offsets of the arrays inside an image with the given counts
struct LufImageLayout { int64_t passes, lvl_pass, rdiag, sval, oval, rovf, sidx, oidx, total; };
LUF_FN LufImageLayout luf_image_layout(int32_t m, int32_t n_passes, int32_t n_levels, int32_t n_lanes, int32_t n_ovf, bool wide) {
    LufImageLayout L;
    int64_t o = 0;
    const int64_t isz = wide ? 4 : 2;
    L.passes = o; o += luf_up16(16LL * (n_passes + kEllPadHeaders));
    L.lvl_pass = o; o += luf_up16(4LL * (n_levels + 1));
    L.rdiag = o; o += luf_up16(8LL * (m + 1));
    L.sval = o; o += luf_up16(8LL * n_lanes);
    L.oval = o; o += luf_up16(8LL * n_ovf);
    L.rovf = o; o += luf_up16(n_ovf > 0 ? 8LL * m : 0);
    L.sidx = o; o += luf_up16(isz * n_lanes);
    L.oidx = o; o += luf_up16(isz * n_ovf);
    L.total = o;
    return L;
}

LUF_FN void luf_put_index(char* base, int64_t at, bool wide, int32_t index, int32_t lg) {
    if (wide) reinterpret_cast<uint32_t*>(base)[at] = (uint32_t)index | ((uint32_t)lg << kEllLgWide);
    else reinterpret_cast<uint16_t*>(base)[at] = (uint16_t)(index | (lg << kEllLg));
}
LUF_FN int32_t luf_lanes_of(int32_t n) { int32_t lg = 0; while ((1 << lg) < n + 1 && lg < 6) ++lg; return 1 << lg; }
LUF_FN int32_t luf_lg_of(int32_t n) { int32_t lg = 0; while ((1 << lg) < n + 1 && lg < 6) ++lg; return lg; }
LUF_FN int32_t luf_popc(uint32_t v) {
#if defined(RELP_LUF_DEVICE)
    return __popc(v);
#else
    return __builtin_popcount(v);
#endif
}
// the set bits of words[0 .. n_words), ascending -> out; returns their number (uniform).  A contiguous chunk of words per thread.
LUF_FN int32_t luf_select_bits(const uint32_t* words, int32_t n_words, int32_t* out) {
#if defined(RELP_LUF_DEVICE)
    const int32_t nt = LUF_NT, chunk = (n_words + nt - 1) / nt;
    const int32_t t = LUF_TID, lo = t * chunk < n_words ? t * chunk : n_words, hi = lo + chunk < n_words ? lo + chunk : n_words;
    int32_t cnt = 0;
    for (int32_t w = lo; w < hi; ++w) cnt += luf_popc((uint32_t)luf_ld(reinterpret_cast<const int32_t*>(words) + w));      // (set by atomics: through L2)
    int32_t total = 0;
    int32_t at = luf_block_exscan(cnt, &total);
    for (int32_t w = lo; w < hi; ++w) { uint32_t v = (uint32_t)luf_ld(reinterpret_cast<const int32_t*>(words) + w); while (v) { out[at++] = (w << 5) + __ffs((int)v) - 1; v &= v - 1; } }
    PAR_END
    return total;
#else
    int32_t at = 0;
    for (int32_t w = 0; w < n_words; ++w) { uint32_t v = words[w]; while (v) { out[at++] = (w << 5) + __builtin_ctz(v); v &= v - 1; } }
    return at;
#endif
}
LUF_FN void luf_or32(uint32_t* p, uint32_t v) {
#if defined(RELP_LUF_DEVICE)
    atomicOr(p, v);
#else
    *p |= v;
#endif
}

#if defined(LUF_COUNT)
static long long g_luf_count[4] = {0, 0, 0, 0};
#endif
// One schedule.  `W` of the factorisation supplies nothing any more (kept out of the signature); everything is in S.
LUF_FN void luf_build_schedule(const LufSchedIn& T, const LufSchedWork& S, const LufSchedOut& O) {
    const int32_t m = T.m;
    const bool maskable = T.keep_trivial != 0, wide = T.wide != 0;
    const int32_t rhs_base = m + 1;
    LUF_LAP_BEGIN
    LUF_SINGLE { for (int q = 0; q < LUF_D_WORDS; ++q) O.desc[q] = 0; } PAR_END
    auto n_of = [&](int32_t k) { return T.ptr[k + 1] - T.ptr[k]; };
    auto diag_of = [&](int32_t k) { return T.diag ? T.diag[k] : 1.0; };
    auto packed = [&](int32_t k) { return T.keep_trivial || n_of(k) > 0 || diag_of(k) != 1.0; };
    // ---- levels (Kahn): frontier bitmaps, ascending enumeration ----------------------------------------------------------
    const int32_t n_words = (m + 31) / 32;
    PAR_FOR(k, m) { luf_st(&S.indeg[k], n_of(k)); S.grp[k] = -1; S.xbeg[k] = -1; S.xlen[k] = n_of(k); O.reach[k] = 0x7fffffff; }
    PAR_FOR(w, n_words + 1) { S.bits0[w] = 0u; S.bits1[w] = 0u; } PAR_END
    PAR_FOR(k, m) { if (n_of(k) == 0) luf_or32(&S.bits0[k >> 5], 1u << (k & 31)); } PAR_END
    int32_t nlev = 0, done = 0;
    {
        uint32_t* cur = S.bits0; uint32_t* nxt = S.bits1;
        while (done < m) {
            const int32_t n = luf_select_bits(cur, n_words, S.order + done);
            if (n == 0 || nlev >= S.nlev_cap) { LUF_SINGLE { O.desc[LUF_D_STATUS] = n == 0 ? LUF_SINGULAR : LUF_NO_ROOM; } PAR_END return; }
            LUF_SINGLE { S.lvl_ptr[nlev] = done; }
            PAR_FOR(w, n_words) cur[w] = 0u; PAR_END
            PAR_FOR(q, n) {
                const int32_t k = S.order[done + q];
                S.lev[k] = nlev;
                for (int32_t e = T.tptr[k]; e < T.tptr[k + 1]; ++e) {
                    const int32_t d = T.tidx[e];
                    if (luf_fetch_add(&S.indeg[d], -1) == 1) luf_or32(&nxt[d >> 5], 1u << (d & 31));
                }
            } PAR_END
            done += n; ++nlev;
            uint32_t* sw = cur; cur = nxt; nxt = sw;
        }
    }
    LUF_SINGLE { S.lvl_ptr[nlev] = m; O.desc[LUF_D_KAHN_LEVELS] = nlev; } PAR_END
    LUF_LAP_AT(S.sc + 8, 0);
    // ---- groups of levels (relp_lu.hpp: fuse_levels) ---------------------------------------------------------------------------
    // Rows that join an open group are rewritten: an entry that reads a row j of the same group is replaced by j's
    // equation -- (rhs copy of j, v / d_j) and -v / d_j times j's own (expanded) entries -- so that the rows of a group
    // only read earlier groups and raw right-hand sides.
    const bool try_fuse = T.fuse_lanes > 0 && T.rhs_cap > 0 && (wide ? true : 2 * (int64_t)m + 2 <= (int64_t(1) << kEllLg));
    int32_t g = -1, lanes = 0, x_top = 0, p_top = 0;
    for (int32_t l = 0; l < nlev; ++l) {
        const int32_t r0 = S.lvl_ptr[l], nr = S.lvl_ptr[l + 1] - r0;
        LUF_SINGLE { luf_st(&S.sc[0], 0); luf_st(&S.sc[1], 0); luf_st(&S.sc[2], 0); } PAR_END
#if defined(LUF_COUNT)
        g_luf_count[1] += 1; g_luf_count[2] += nr;
#endif
        bool ok = try_fuse && g >= 0;
        // lanes the level takes as it is; upper bounds of the expanded rows
        PAR_FOR(q, nr) {
            const int32_t k = S.order[r0 + q], n = n_of(k);
            if (packed(k)) luf_add(&S.sc[0], luf_lanes_of(n));
            int32_t ub = 0, pub = 0;
            if (ok) {
                for (int32_t e = T.ptr[k]; e < T.ptr[k + 1]; ++e) {
                    const int32_t j = T.idx[e];
                    if (S.grp[j] != g) { ++ub; continue; }
                    ub += 1 + S.xlen[j]; pub += 1;
                    if (maskable) {
                        if (S.xbeg[j] < 0) pub += S.xlen[j];
                        else for (int32_t u = S.xbeg[j]; u < S.xbeg[j] + S.xlen[j]; ++u) pub += S.x_vn[u] + 1;
                    }
                }
                // (an upper bound before duplicates are combined; the combining below searches the row: 128^2 steps at worst --
                // a row of L was seen to spend 6.5 M clocks here at 448)
                if (ub > 128) luf_st(&S.sc[1], 1);
            }
            S.tmp[q + 1] = ub; S.tmp2[q + 1] = maskable ? pub : 0;
        } PAR_END
        const int32_t least = luf_ld(&S.sc[0]);
        if (ok && (lanes + least > T.fuse_lanes || luf_ld(&S.sc[1]))) ok = false;
        int32_t tot_x = 0, tot_p = 0, add = 0;
        if (ok) {
            tot_x = luf_offsets_from_counts(S.tmp, nr);
            tot_p = luf_offsets_from_counts(S.tmp2, nr);
            if ((int64_t)x_top + tot_x > S.x_cap || (int64_t)p_top + tot_p > S.pool_cap) ok = false;
        }
        if (ok) {
#if defined(RELP_LUF_DEVICE)
            if (!maskable) {
                // A wave per row, the expanded row in registers (one term per lane: a row beyond 63 terms ends the group anyway): a
                // term is combined with the one of the same index by `ballot(src == s)`, the rows it substitutes are read one entry
                // per lane and walked by broadcast.  The sums run in the order of the serial form below.  (One thread per row
                // searched its row in global memory for every term: 25,000 dependent steps and 4.8 M clocks on one row of L.)
                const int lane = LUF_TID & 63, wave = LUF_TID >> 6, nw = LUF_NT >> 6;
                for (int32_t q = wave; q < nr; q += nw) {
                    const int32_t k = S.order[r0 + q];
                    const int32_t xb = x_top + S.tmp[q];
                    int32_t src = -1, n = 0;
                    double coef = 0.0;
                    auto term = [&](int32_t s_, double c_) {
                        const unsigned long long mt = __ballot(src == s_);
                        if (mt) { if (lane == __ffsll(mt) - 1) coef += c_; }
                        else { if (lane == n) { src = s_; coef = c_; } ++n; }
                    };
                    for (int32_t e = T.ptr[k]; e < T.ptr[k + 1]; ++e) {
                        const int32_t j = T.idx[e];
                        const double v = T.val[e];
                        if (S.grp[j] != g) { term(j, v); continue; }
                        const double f = v / diag_of(j);
                        term(rhs_base + j, f);
                        const bool raw = S.xbeg[j] < 0;
                        const int32_t sb = raw ? T.ptr[j] : S.xbeg[j], sn = raw ? T.ptr[j + 1] - T.ptr[j] : S.xlen[j];
                        for (int32_t u0 = 0; u0 < sn; u0 += 64) {
                            const bool in = u0 + lane < sn;
                            const int32_t my_s = in ? (raw ? T.idx[sb + u0 + lane] : S.x_src[sb + u0 + lane]) : -1;
                            const double my_c = in ? (raw ? T.val[sb + u0 + lane] : S.x_coef[sb + u0 + lane]) : 0.0;
                            const int32_t cnt = sn - u0 < 64 ? sn - u0 : 64;
                            for (int32_t u = 0; u < cnt; ++u) term(__shfl(my_s, u, 64), -f * __shfl(my_c, u, 64));
                        }
                    }
                    if (lane < n && lane < 64) { S.x_src[xb + lane] = src; S.x_coef[xb + lane] = coef; }
                    if (lane == 0) {
                        if (n > 63) luf_st(&S.sc[1], 1);
                        S.loff[k] = xb; S.lg[k] = n;
                        if (packed(k)) luf_add(&S.sc[2], luf_lanes_of(n));
                    }
                }
                __syncthreads();
            } else
#endif
            {
            PAR_FOR(q, nr) {
                const int32_t k = S.order[r0 + q];
                const int32_t xb = x_top + S.tmp[q];
                int32_t n = 0, pp = p_top + S.tmp2[q];
                auto term = [&](int32_t src, double coef, int32_t v0, int32_t vn) {
                    if (!maskable) {                                 // one entry per index, summed in the order met
#if defined(LUF_COUNT)
                        g_luf_count[0] += n;
#endif
                        for (int32_t a = xb; a < xb + n; ++a) if (S.x_src[a] == src) { S.x_coef[a] += coef; return; }
                    }
                    S.x_src[xb + n] = src; S.x_coef[xb + n] = coef;
                    if (maskable) { S.x_v0[xb + n] = v0; S.x_vn[xb + n] = vn; }
                    ++n;
                };
                for (int32_t e = T.ptr[k]; e < T.ptr[k + 1]; ++e) {
                    const int32_t j = T.idx[e];
                    const double v = T.val[e];
                    if (S.grp[j] != g) { term(j, v, 0, 0); continue; }
                    const double f = v / diag_of(j);
                    int32_t v0 = pp;
                    if (maskable) S.pool[pp++] = j;
                    term(rhs_base + j, f, v0, 1);
                    if (S.xbeg[j] < 0) {
                        for (int32_t u = T.ptr[j]; u < T.ptr[j + 1]; ++u) {
                            v0 = pp;
                            if (maskable) S.pool[pp++] = j;
                            term(T.idx[u], -f * T.val[u], v0, maskable ? 1 : 0);
                        }
                    } else {
                        for (int32_t u = S.xbeg[j]; u < S.xbeg[j] + S.xlen[j]; ++u) {
                            v0 = pp;
                            if (maskable) { for (int32_t a = 0; a < S.x_vn[u]; ++a) S.pool[pp++] = S.pool[S.x_v0[u] + a]; S.pool[pp++] = j; }
                            term(S.x_src[u], -f * S.x_coef[u], v0, maskable ? S.x_vn[u] + 1 : 0);
                        }
                    }
                }
                if (n > 63) luf_st(&S.sc[1], 1);
                S.loff[k] = xb; S.lg[k] = n;                              // (kept aside until the level is accepted)
                if (packed(k)) luf_add(&S.sc[2], luf_lanes_of(n));
            } PAR_END
            }
            add = luf_ld(&S.sc[2]);
            if (luf_ld(&S.sc[1]) || lanes + add > T.fuse_lanes) ok = false;
        }
        if (ok) {
            PAR_FOR(q, nr) { const int32_t k = S.order[r0 + q]; S.xbeg[k] = S.loff[k]; S.xlen[k] = S.lg[k]; if (packed(k)) S.grp[k] = g; }
            lanes += add; x_top += tot_x; p_top += tot_p;
        } else {                                                         // the level opens a new group with its rows as they are
            ++g; lanes = least;
            LUF_SINGLE { S.grp_lvl0[g] = l; }
            PAR_FOR(q, nr) { const int32_t k = S.order[r0 + q]; if (packed(k)) S.grp[k] = g; }
        }
        LUF_SINGLE { S.lvl_grp[l] = g; } PAR_END
    }
    const int32_t ngroups = g + 1;
    LUF_LAP_AT(S.sc + 8, 1);
    LUF_SINGLE { S.grp_lvl0[ngroups] = nlev; } PAR_END
    // ---- what each row is: left out, listed, or a slot row of 2^lg lanes -----------------------------------------------------
    const int32_t n_trivial = T.keep_trivial ? luf_select(m, [&](int32_t k) { return n_of(k) == 0; }, S.tmp) : 0;
    const bool list_trivial = T.keep_trivial && n_trivial >= T.triv_min;
    // (of the listed rows only those with a diagonal other than 1 need the loop in front of the passes: ell_pack)
    const int32_t n_triv_listed = list_trivial ? luf_select(m, [&](int32_t k) { return n_of(k) == 0 && diag_of(k) != 1.0; }, O.triv) : 0;
    PAR_FOR(k, m) {
        const int32_t n = S.xlen[k];
        int32_t lg = luf_lg_of(n);
        if (n_of(k) == 0 && (list_trivial || !packed(k))) lg = -1;
        S.lg[k] = lg;
        S.ovf_off[k + 1] = (lg == 6 && n > 63) ? n - 63 : 0;
        // (start_after of fuse_levels: the last level of a group lets the sweep start behind the group, the others inside it)
        const int32_t l = S.lev[k], q = S.lvl_grp[l];
        O.level_of[k] = (l + 1 == nlev || S.lvl_grp[l + 1] != q) ? q : q - 1;
    } PAR_END
    const int32_t n_ovf = luf_offsets_from_counts(S.ovf_off, m);
    // ---- lane offsets inside the groups: widths descending, rows of one width in level order -----------------------------------
    PAR_FOR(q, ngroups) {
        const int32_t a0 = S.lvl_ptr[S.grp_lvl0[q]], a1 = S.lvl_ptr[S.grp_lvl0[q + 1]];
        if (a1 - a0 > 512) { S.grp_lanes[q] = -1; continue; }              // (a wide level: by the whole workgroup below)
        int32_t cnt[7] = {0, 0, 0, 0, 0, 0, 0}, base[7], total = 0;
        for (int32_t a = a0; a < a1; ++a) { const int32_t lg = S.lg[S.order[a]]; if (lg >= 0) ++cnt[lg]; }
        for (int w = 6; w >= 0; --w) { base[w] = total; total += cnt[w] << w; }
        for (int32_t a = a0; a < a1; ++a) { const int32_t k = S.order[a], lg = S.lg[k]; if (lg >= 0) { S.loff[k] = base[lg]; base[lg] += 1 << lg; } }
        S.grp_lanes[q] = total;
    } PAR_END
    for (int32_t q = 0; q < ngroups; ++q) {
        if (S.grp_lanes[q] >= 0) continue;
        const int32_t a0 = S.lvl_ptr[S.grp_lvl0[q]], a1 = S.lvl_ptr[S.grp_lvl0[q + 1]];
        int32_t total = 0;
        for (int w = 6; w >= 0; --w) {
            const int32_t nw = luf_select(a1 - a0, [&](int32_t a) { return S.lg[S.order[a0 + a]] == w; }, S.tmp);
            PAR_FOR(r, nw) { S.loff[S.order[a0 + S.tmp[r]]] = total + (r << w); } PAR_END
            total += nw << w;
        }
        LUF_SINGLE { S.grp_lanes[q] = total; } PAR_END
    }
    LUF_SINGLE {
        int32_t pass0 = 0, lane0 = 0;
        for (int32_t q = 0; q < ngroups; ++q) { S.grp_pass0[q] = pass0; S.grp_lane0[q] = lane0; pass0 += (S.grp_lanes[q] + 255) / 256; lane0 += S.grp_lanes[q]; }
        S.grp_pass0[ngroups] = pass0; S.grp_lane0[ngroups] = lane0;
    } PAR_END
    const int32_t n_passes = S.grp_pass0[ngroups], n_lanes = S.grp_lane0[ngroups];
    LUF_LAP_AT(S.sc + 8, 2);
    // ---- right-hand-side copies: compacted for the wide layouts (ell_pack: rhs_src) -----------------------------------------------
    int32_t n_rhs = 0, uses_rhs = 0;
    if (x_top > 0) {
        uses_rhs = 1;
        if (wide) {
            PAR_FOR(k, m) S.rhs_id[k] = 0; PAR_END
            PAR_FOR(k, m) { if (S.xbeg[k] >= 0) for (int32_t u = S.xbeg[k]; u < S.xbeg[k] + S.xlen[k]; ++u) if (S.x_src[u] >= rhs_base) S.rhs_id[S.x_src[u] - rhs_base] = 1; } PAR_END
            n_rhs = luf_select(m, [&](int32_t k) { return S.rhs_id[k] != 0; }, O.rhs_src);
            PAR_FOR(k, m) S.rhs_id[k] = -1; PAR_END
            PAR_FOR(i, n_rhs) S.rhs_id[O.rhs_src[i]] = i; PAR_END
            if (O.rhs_pos) { PAR_FOR(k, m) O.rhs_pos[k] = S.rhs_id[k]; PAR_END }
        }
    } else if (O.rhs_pos) { PAR_FOR(k, m) O.rhs_pos[k] = -1; PAR_END }
    const LufImageLayout L = luf_image_layout(m, n_passes, ngroups, n_lanes, n_ovf, wide);
    LUF_SINGLE {
        O.desc[LUF_D_PASSES] = n_passes; O.desc[LUF_D_LEVELS] = ngroups; O.desc[LUF_D_LANES] = n_lanes; O.desc[LUF_D_OVF] = n_ovf;
        O.desc[LUF_D_BYTES] = (int32_t)L.total; O.desc[LUF_D_TRIV] = n_triv_listed; O.desc[LUF_D_NRHS] = wide ? n_rhs : -1;
        O.desc[LUF_D_USES_RHS] = uses_rhs;
        if (L.total > O.image_cap) O.desc[LUF_D_STATUS] = LUF_NO_ROOM;
        if (wide && (n_rhs > T.rhs_cap || (int64_t)m + 1 + n_rhs > (int64_t(1) << kEllLgWide))) O.desc[LUF_D_STATUS] = LUF_NO_ROOM;
    } PAR_END
    if (O.desc[LUF_D_STATUS] != LUF_OK) return;
    // ---- the image ---------------------------------------------------------------------------------------------------------
    char* const img = O.image;
    EllPass* const passes = reinterpret_cast<EllPass*>(img + L.passes);
    int32_t* const lvl_pass = reinterpret_cast<int32_t*>(img + L.lvl_pass);
    double* const rdiag = reinterpret_cast<double*>(img + L.rdiag);
    double* const sval = reinterpret_cast<double*>(img + L.sval);
    double* const oval = reinterpret_cast<double*>(img + L.oval);
    int32_t* const rovf = reinterpret_cast<int32_t*>(img + L.rovf);
    PAR_FOR(q, ngroups + 1) lvl_pass[q] = S.grp_pass0[q];
    PAR_FOR(k, m + 1) rdiag[k] = (k < m && T.diag && (S.lg[k] >= 0 || list_trivial)) ? 1.0 / T.diag[k] : 1.0;
    // pass headers: lane0, lanes, info (widest row | last pass of its group << 8 | overflow << 9), group
    PAR_FOR(p, kEllPadHeaders) { EllPass& ps = passes[n_passes + p]; ps.lane0 = 0; ps.lanes = 0; ps.info = 0; ps.level = 0; }
    PAR_FOR(q, ngroups) {
        const int32_t np = S.grp_pass0[q + 1] - S.grp_pass0[q], tot = S.grp_lanes[q];
        for (int32_t a = 0; a < np; ++a) {
            EllPass& ps = passes[S.grp_pass0[q] + a];
            ps.lane0 = S.grp_lane0[q] + 256 * a;
            ps.lanes = tot - 256 * a < 256 ? tot - 256 * a : 256;
            ps.level = q;
            ps.info = a + 1 == np ? 1 << 8 : 0;
        }
    } PAR_END
    if (n_ovf > 0) { PAR_FOR(k, m) { rovf[2 * k] = S.ovf_off[k]; rovf[2 * k + 1] = S.ovf_off[k + 1]; } PAR_END }
    if (maskable) { PAR_FOR(k, m + 1) luf_st(&O.via_ptr[k], 0); PAR_END }
    auto slot_index = [&](int32_t src) { return (wide && src >= rhs_base) ? rhs_base + S.rhs_id[src - rhs_base] : src; };
    PAR_FOR(k, m) {
        const int32_t lgk = S.lg[k];
        if (lgk < 0) continue;
        const int32_t q = S.grp[k], w = 1 << lgk, n = S.xlen[k];
        const int32_t at = S.grp_lane0[q] + S.loff[k];
        const bool plain = S.xbeg[k] < 0;
        const int32_t e0 = plain ? T.ptr[k] : S.xbeg[k];
        luf_put_index(img + L.sidx, at, wide, k, lgk);                   // the row's own unknown: -(-1) x[k]
        sval[at] = -1.0;
        for (int32_t j = 1; j < w; ++j) {
            const bool has = j - 1 < n;
            const int32_t src = has ? (plain ? T.idx[e0 + j - 1] : S.x_src[e0 + j - 1]) : 0;
            luf_put_index(img + L.sidx, at + j, wide, has ? slot_index(src) : 0, lgk);
            sval[at + j] = has ? (plain ? T.val[e0 + j - 1] : S.x_coef[e0 + j - 1]) : 0.0;
            if (has && maskable && !plain) for (int32_t a = 0; a < S.x_vn[e0 + j - 1]; ++a) luf_add(&O.via_ptr[S.pool[S.x_v0[e0 + j - 1] + a] + 1], 1);
        }
        for (int32_t e = w - 1, o = S.ovf_off[k]; e < n; ++e, ++o) {       // (only rows as they are overflow: fused rows stop at 63)
            luf_put_index(img + L.oidx, o, wide, T.idx[e0 + e], 0);
            oval[o] = T.val[e0 + e];
        }
        // header bits of the row's pass: the widest row of a pass is its first one; an overflow row marks its pass
        const int32_t pass = S.grp_pass0[q] + S.loff[k] / 256;
        if (S.loff[k] % 256 == 0) luf_or(&passes[pass].info, lgk);
        if (S.ovf_off[k + 1] > S.ovf_off[k]) luf_or(&passes[pass].info, 1 << 9);
        // the reach of every pivot: the first group in which its value (or its right-hand-side copy) matters
        luf_min32(&O.reach[k], q);
        for (int32_t e = 0; e < n; ++e) {
            const int32_t src = plain ? T.idx[e0 + e] : S.x_src[e0 + e];
            luf_min32(&O.reach[src >= rhs_base ? src - rhs_base : src], q);
        }
    } PAR_END
    LUF_LAP_AT(S.sc + 8, 3);
    // ---- per pivot: the slots whose substitution path runs through it (what a Forrest-Tomlin update zeroes) -------------------------
    int32_t n_via = 0;
    if (maskable) {
        PAR_FOR(k, m + 1) { const int32_t v = luf_ld(&O.via_ptr[k]); O.via_ptr[k] = v; } PAR_END
        n_via = luf_offsets_from_counts(O.via_ptr, m);
        if (n_via > O.via_cap) { LUF_SINGLE { O.desc[LUF_D_STATUS] = LUF_NO_ROOM; } PAR_END return; }
        PAR_FOR(k, m) luf_st(&S.tmp[k], 0); PAR_END
        PAR_FOR(k, m) {
            if (S.lg[k] < 0 || S.xbeg[k] < 0) continue;
            const int32_t at = S.grp_lane0[S.grp[k]] + S.loff[k], e0 = S.xbeg[k];
            for (int32_t e = 0; e < S.xlen[k]; ++e)
                for (int32_t a = 0; a < S.x_vn[e0 + e]; ++a) {
                    const int32_t p = S.pool[S.x_v0[e0 + e] + a];
                    O.via_pos[O.via_ptr[p] + luf_fetch_add(&S.tmp[p], 1)] = at + 1 + e;
                }
        } PAR_END
    }
    if (O.triv_bits) {
        PAR_FOR(w, n_words + 1) O.triv_bits[w] = 0u; PAR_END
        PAR_FOR(i, n_triv_listed) luf_or32(&O.triv_bits[O.triv[i] >> 5], 1u << (O.triv[i] & 31)); PAR_END
    }
    LUF_SINGLE { O.desc[LUF_D_VIA] = n_via; } PAR_END
    LUF_LAP_AT(S.sc + 8, 4);
}

#if !defined(RELP_LUF_DEVICE)
// relp_kernels_luf.hip: the four schedules (L, U, U', L'), one workgroup each, on stream s behind launch_lu_factor
void launch_lu_schedules(const LufSchedIn in[4], const LufSchedWork work[4], const LufSchedOut out[4], const int32_t* status,
                         FtPivotInfo* pinfo, hipStream_t s);
#endif

}  // namespace relp
