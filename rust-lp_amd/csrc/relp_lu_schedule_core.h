// relp_lu_schedule_core.h -- second half of the device-side refactorisation: from the rows of a triangular factor (as
// relp_lu_factor_core.h leaves them) to what the persistent pivot kernel solves with -- levels, the "ELL by pass" image
// (relp_lu.hpp: ell_pack; one level = one group, i.e. unfused), the list of rows without entries, the reach array of the
// hyper-sparse start.  Same conventions as relp_lu_factor_core.h: one workgroup, PAR_FOR loops that end in barriers, compiled
// for the device and, serially, for the host (tests/cpp/test_lu_device_model.cpp executes the image pass by pass).
//
// Levels: lev[k] = 1 + max over the row's entries, by relaxation until nothing changes (as many rounds as there are levels).
// Inside a level rows are ordered by width (64, 32, .. 1 lanes: a row with n entries takes the next power of two above n) so
// that every row starts at a multiple of its width and no row straddles a pass of 256 lanes; rows of one width keep their
// pivot order.  Entries beyond the 63rd of a row go to the overflow lists.
#pragma once
#include "relp_lu_factor_core.h"

namespace relp {

struct LufSchedIn {
    int32_t m;
    const int32_t* ptr; const int32_t* idx; const double* val;      // rows in pivot coordinates
    const double* diag;                // nullptr: unit diagonal (L, L')
    int32_t ascending;                 // dependencies have smaller indices (L, U'), else larger (U, L')
    int32_t keep_trivial;              // rows without entries are kept (U, U': an update may mask them)
    int32_t wide;                      // 32-bit slots (index | lg << 24), else 16-bit (index | lg << 13)
    int32_t triv_min;                  // rows without entries are LISTED from this many on (else packed as slots)
};
struct LufSchedWork {
    int32_t* lev; int32_t* lg; int32_t* loff; int32_t* list;       // m each
    int32_t* lvl_lanes; int32_t* lvl_pass0; int32_t* lvl_lane0;     // nlev_cap + 1 each
    int32_t* ovf_off;                  // m + 1
    int32_t* hist;                     // 8
    int32_t* part;                     // threads + 2 (luf_select)
    int32_t* flag;                     // 4
    int32_t nlev_cap;
};
enum { LUF_D_PASSES = 0, LUF_D_LEVELS, LUF_D_LANES, LUF_D_OVF, LUF_D_BYTES, LUF_D_TRIV, LUF_D_STATUS, LUF_D_WORDS = 8 };
struct LufSchedOut {
    char* image; int64_t image_cap;    // the contiguous image (passes | lvl_pass | rdiag | sval | oval | rovf | sidx | oidx)
    int32_t* desc;                     // LUF_D_*
    int32_t* triv; int32_t* reach; int32_t* level_of;               // m each
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

// `W` of the factorisation supplies luf_select's scratch (part); everything else is S.
LUF_FN void luf_build_schedule(const LufSchedIn& T, const LufSchedWork& S, const LufSchedOut& O, const LufWork& W) {
    const int32_t m = T.m;
    LUF_SINGLE { for (int q = 0; q < LUF_D_WORDS; ++q) O.desc[q] = 0; } PAR_END
    // ---- levels ----------------------------------------------------------------------------------------------------------
    PAR_FOR(k, m) S.lev[k] = 0; PAR_END
    for (int32_t round = 0; round <= m; ++round) {
        LUF_SINGLE { S.flag[0] = 0; } PAR_END
        PAR_FOR(k, m) {
            int32_t l = 0;
            for (int32_t e = T.ptr[k]; e < T.ptr[k + 1]; ++e) { const int32_t d = S.lev[T.idx[e]] + 1; if (d > l) l = d; }
            if (l > S.lev[k]) { S.lev[k] = l; S.flag[0] = 1; }
        } PAR_END
        const int32_t changed = S.flag[0];
        PAR_END                                         // (everybody has read the flag before the next round clears it)
        if (!changed) break;
    }
    LUF_SINGLE { luf_st(&S.hist[7], 0); } PAR_END
    PAR_FOR(k, m) { if (S.lev[k] + 1 > luf_ld(&S.hist[7])) luf_max32(&S.hist[7], S.lev[k] + 1); } PAR_END
    const int32_t nlev = m > 0 ? luf_ld(&S.hist[7]) : 0;
    if (nlev > S.nlev_cap) { LUF_SINGLE { O.desc[LUF_D_STATUS] = LUF_NO_ROOM; } PAR_END return; }
    // ---- what each row is: left out, listed, or a slot row of 2^lg lanes -----------------------------------------------------
    const int32_t n_trivial = T.keep_trivial ? luf_select(m, [&](int32_t k) { return T.ptr[k + 1] == T.ptr[k]; }, O.triv, W) : 0;
    const bool list_trivial = T.keep_trivial && n_trivial >= T.triv_min;
    PAR_FOR(k, m) {
        const int32_t n = T.ptr[k + 1] - T.ptr[k];
        int32_t lg = 0;
        while ((1 << lg) < n + 1 && lg < 6) ++lg;
        if (n == 0 && (list_trivial || !T.keep_trivial)) lg = -1;          // (a unit row of L, or a row of the `triv` loop)
        S.lg[k] = lg;
        S.ovf_off[k + 1] = (lg == 6 && n > 63) ? n - 63 : 0;
        O.level_of[k] = S.lev[k];
        O.reach[k] = 0x7fffffff;
    } PAR_END
    const int32_t n_ovf = luf_offsets_from_counts(S.ovf_off, m);     // overflow ranges by pivot (few rows have more than 63 entries)
    // ---- level by level: the rows in pivot order, their lane offsets ----------------------------------------------------------
    int32_t pass0 = 0, lane0 = 0;
    for (int32_t l = 0; l < nlev; ++l) {
        const int32_t nl = luf_select(m, [&](int32_t k) { return S.lev[k] == l && S.lg[k] >= 0; }, S.list, W);
        LUF_SINGLE { for (int g = 0; g < 7; ++g) luf_st(&S.hist[g], 0); } PAR_END
        PAR_FOR(q, nl) luf_add(&S.hist[S.lg[S.list[q]]], 1); PAR_END
        int32_t base[7], total = 0, kinds = 0;
        for (int g = 6; g >= 0; --g) { const int32_t c = luf_ld(&S.hist[g]); base[g] = total; total += c << g; kinds += c > 0 ? 1 : 0; }
        PAR_FOR(q, nl) {
            const int32_t k = S.list[q], g = S.lg[k];
            int32_t before = q;                          // rows of the same width in front of this one
            if (kinds > 1) { before = 0; for (int32_t r = 0; r < q; ++r) before += S.lg[S.list[r]] == g ? 1 : 0; }
            S.loff[k] = base[g] + (before << g);
        } PAR_END
        LUF_SINGLE { S.lvl_lanes[l] = total; S.lvl_pass0[l] = pass0; S.lvl_lane0[l] = lane0; } PAR_END
        pass0 += (total + 255) / 256; lane0 += total;
    }
    const int32_t n_passes = pass0, n_lanes = lane0;
    const LufImageLayout L = luf_image_layout(m, n_passes, nlev, n_lanes, n_ovf, T.wide != 0);
    LUF_SINGLE {
        S.lvl_pass0[nlev] = n_passes; S.lvl_lane0[nlev] = n_lanes;
        O.desc[LUF_D_PASSES] = n_passes; O.desc[LUF_D_LEVELS] = nlev; O.desc[LUF_D_LANES] = n_lanes; O.desc[LUF_D_OVF] = n_ovf;
        O.desc[LUF_D_BYTES] = (int32_t)L.total; O.desc[LUF_D_TRIV] = list_trivial ? n_trivial : 0;
        if (L.total > O.image_cap) O.desc[LUF_D_STATUS] = LUF_NO_ROOM;
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
    PAR_FOR(l, nlev + 1) lvl_pass[l] = S.lvl_pass0[l]; PAR_END
    PAR_FOR(k, m + 1) rdiag[k] = (k < m && T.diag && (S.lg[k] >= 0 || list_trivial)) ? 1.0 / T.diag[k] : 1.0; PAR_END
    // pass headers: lane0, lanes, info (widest row | last pass of its level << 8 | overflow << 9), level
    PAR_FOR(p, n_passes + kEllPadHeaders) { passes[p].lane0 = 0; passes[p].lanes = 0; passes[p].info = 0; passes[p].level = 0; } PAR_END
    PAR_FOR(l, nlev) {
        const int32_t np = S.lvl_pass0[l + 1] - S.lvl_pass0[l], tot = S.lvl_lanes[l];
        for (int32_t a = 0; a < np; ++a) {
            EllPass& ps = passes[S.lvl_pass0[l] + a];
            ps.lane0 = S.lvl_lane0[l] + 256 * a;
            ps.lanes = tot - 256 * a < 256 ? tot - 256 * a : 256;
            ps.level = l;
            ps.info = a + 1 == np ? 1 << 8 : 0;
        }
    } PAR_END
    if (n_ovf > 0) { PAR_FOR(k, m) { rovf[2 * k] = S.ovf_off[k]; rovf[2 * k + 1] = S.ovf_off[k + 1]; } PAR_END }
    PAR_FOR(k, m) {
        const int32_t g = S.lg[k];
        if (g < 0) continue;
        const int32_t l = S.lev[k], w = 1 << g, n = T.ptr[k + 1] - T.ptr[k];
        const int32_t at = S.lvl_lane0[l] + S.loff[k];
        luf_put_index(img + L.sidx, at, T.wide != 0, k, g);               // the row's own unknown: -(-1) x[k]
        sval[at] = -1.0;
        for (int32_t j = 1; j < w; ++j) {
            const bool has = j - 1 < n;
            luf_put_index(img + L.sidx, at + j, T.wide != 0, has ? T.idx[T.ptr[k] + j - 1] : 0, g);
            sval[at + j] = has ? T.val[T.ptr[k] + j - 1] : 0.0;
        }
        for (int32_t e = T.ptr[k] + w - 1, o = S.ovf_off[k]; e < T.ptr[k + 1]; ++e, ++o) {
            luf_put_index(img + L.oidx, o, T.wide != 0, T.idx[e], 0);
            oval[o] = T.val[e];
        }
        // header bits of the row's pass: the widest row of a pass is its first one; an overflow row marks its pass
        const int32_t pass = S.lvl_pass0[l] + S.loff[k] / 256;
        if (S.loff[k] % 256 == 0) luf_or(&passes[pass].info, g);
        if (S.ovf_off[k + 1] > S.ovf_off[k]) luf_or(&passes[pass].info, 1 << 9);
    } PAR_END
    // ---- the reach of every pivot: the first level in which its value matters ------------------------------------------------
    PAR_FOR(k, m) {
        const int32_t n = T.ptr[k + 1] - T.ptr[k], l = S.lev[k];
        if (n == 0 && S.lg[k] < 0) continue;             // (a unit row that is left out, or a row of the `triv` loop)
        luf_min32(&O.reach[k], l);
        for (int32_t e = T.ptr[k]; e < T.ptr[k + 1]; ++e) luf_min32(&O.reach[T.idx[e]], l);
    } PAR_END
}

#if !defined(RELP_LUF_DEVICE)
// relp_kernels_luf.hip: the four schedules (L, U, U', L') by one workgroup on stream s, behind launch_lu_factor
void launch_lu_schedules(const LufSchedIn in[4], const LufSchedOut out[4], const LufSchedWork& S, const LufWork& W, const int32_t* status,
                         FtPivotInfo* pinfo, hipStream_t s);
#endif

}  // namespace relp
