// relp_engine_luf.cpp -- Engine: the refactorisation of the LU engine ON THE DEVICE (relp_lu_factor_core.h).
// What stays on the host is what is static: the row-major copy of the provider matrix the kernel enumerates basis rows
// from (built at create, after a row removal and at the phase switch), and the buffers.
#include "relp_engine_internal.hpp"
#include "relp_lu_factor_core.h"
#include "relp_lu_schedule_core.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace relp {

struct Engine::LufState {
    char* d_buf = nullptr;
    LufMatrix M{}; LufWork W{}; LufOut O{};
    int32_t cap = 0, nb_cap = 0;
    bool dirty = true;
    int32_t key[4] = {-1, -1, -1, -1};                         // (m, artificial columns, phase, wrapped artificials) the tables were built for
    LufSchedWork SW[4]; LufSchedIn sin[4]; LufSchedOut sout[4];   // the schedules (relp_lu_schedule_core.h), one work set each
    int32_t x_cap = 0, pool_cap = 0;
    FtPivotInfo* pinfo = nullptr;
    int64_t img_cap = 0;
    bool resident = false;                                       // hlu_ does not hold the rows of the factors in use (they are on the device)
    ~LufState() { if (d_buf) (void)hipFree(d_buf); }
};

void Engine::luf_release() { delete luf_; luf_ = nullptr; }
void Engine::luf_mark_dirty() { if (luf_) luf_->dirty = true; }

// static tables + workspace for the current shape of the problem (rows may have been removed, the phase may have changed)
relp_status_t Engine::luf_prepare() {
    if (!luf_) luf_ = new LufState();
    LufState& S = *luf_;
    if (S.d_buf) { HIP_TRY(hipFree(S.d_buf)); S.d_buf = nullptr; }
    const int32_t m = m_, nprov = nr_normal_ + nr_virtual_, na = nr_artificial_;
    // row-major copy of the provider columns (structural incl. bound rows, virtual), in column order
    std::vector<int32_t> rcount(m + 1, 0);
    auto each = [&](auto f) {
        for (int32_t p = 0; p < nr_normal_; ++p) {
            for (int64_t e = hc_ptr_[p]; e < hc_ptr_[p + 1]; ++e) f(hc_idx_[e], p, hc_val_[e]);
            if (bound_row_h_[p] >= 0) f(bound_row_h_[p], p, 1.0);
        }
        for (int32_t v = 0; v < nr_virtual_; ++v) {
            if (vrow0_h_[v] >= 0) f(vrow0_h_[v], nr_normal_ + v, (double)vsign_h_[v]);
            if (vrow1_h_[v] >= 0) f(vrow1_h_[v], nr_normal_ + v, 1.0);
        }
    };
    int64_t nnz = 0;
    each([&](int32_t i, int32_t, double) { ++rcount[i + 1]; ++nnz; });
    for (int32_t i = 0; i < m; ++i) rcount[i + 1] += rcount[i];
    std::vector<int32_t> rcol((size_t)nnz), fill(rcount.begin(), rcount.end() - 1), art_of_row(m, -1);
    std::vector<double> rval((size_t)nnz);
    each([&](int32_t i, int32_t p, double v) { const int32_t o = fill[i]++; rcol[o] = p; rval[o] = v; });
    for (int32_t a = 0; a < na; ++a) art_of_row[column_to_row_[a]] = a;
    // sizes: the bump is eliminated on sparse rows in an arena; RELP_LUF_BUMP_CAP bounds its rows (default: any), a bump or a
    // fill-in beyond the arrays falls back to the host
    const char* cap_env = std::getenv("RELP_LUF_BUMP_CAP");
    S.nb_cap = std::min<int32_t>(m, cap_env ? std::max(16, std::atoi(cap_env)) : m);
    // (a small dense bump may fill in completely)
    const int64_t arena_cap = std::min<int64_t>(INT32_MAX / 4, 3 * (nnz + (int64_t)wrapped_na_ + na + m) + 64 * (int64_t)S.nb_cap + 1024 +
                                                                  4 * std::min<int64_t>((int64_t)S.nb_cap * S.nb_cap, int64_t(1) << 21));
    S.cap = (int32_t)arena_cap;
    const int32_t nt = luf_threads();
    int64_t o = 0;
    auto take = [&](int64_t bytes) { const int64_t at = o; o += round_up(std::max<int64_t>(bytes, 16), 16); return at; };
    const int64_t o_rptr = take(4 * ((int64_t)m + 1)), o_rcol = take(4 * nnz), o_rval = take(8 * nnz), o_art = take(4 * (int64_t)m),
                  o_wr = take(4 * (int64_t)std::max<int32_t>(wrapped_na_, 1));
    const int64_t o_posp = take(4 * ((int64_t)nprov + 1)), o_posa = take(4 * ((int64_t)na + 1));
    int64_t o_m[13];
    for (auto& v : o_m) v = take(4 * (int64_t)m);       // wrow_pos rcount ccount claim claim2 list list2 piv brow bcol lrow lcol (+1 spare)
    const int64_t o_part = take(4 * ((int64_t)nt + 2));
    int64_t o_nb[11];
    for (auto& v : o_nb) v = take(4 * (int64_t)S.nb_cap);     // rbeg rlen rcap ract cact bcc bstep_row bstep_col cpiv prank acc
    int64_t o_nb8[5];
    for (auto& v : o_nb8) v = take(8 * (int64_t)std::max(S.nb_cap, 512));     // (>= 512: the dense finish keeps 8 x 64 partials there)    // pval cmax rowmark colbest cprio
    const int64_t o_ecol = take(4 * arena_cap), o_eval = take(8 * arena_cap);
    const int64_t o_ltr = take(4 * arena_cap), o_lts = take(4 * arena_cap), o_ltv = take(8 * arena_cap), o_ltp = take(4 * ((int64_t)S.nb_cap + 1)),
                  o_lto = take(4 * arena_cap), o_cnt = take(512);
    const int64_t o_red = take(8 * 8), o_sc = take(64);
    const char* dense_env = std::getenv("RELP_LUF_DENSE");                   // rows of the dense finish (<= 64; 0 = off)
    const int32_t dense_cap = dense_env ? std::max(0, std::min(64, std::atoi(dense_env))) : 64;
    const int64_t o_dense = take(8 * (int64_t)64 * 64), o_dint = take(4 * 8 * 64);
    const int64_t o_utr = take(4 * arena_cap), o_utc = take(4 * arena_cap), o_utv = take(8 * arena_cap), o_vw = take(4 * ((int64_t)m + 2)),
                  o_vtmp = take(4 * arena_cap);
    const int64_t o_status = take(32), o_rowperm = take(4 * (int64_t)m), o_colperm = take(4 * (int64_t)m), o_rstep = take(4 * (int64_t)m),
                  o_cstep = take(4 * (int64_t)m), o_diag = take(8 * (int64_t)m);
    int64_t o_tp[4], o_ti[4], o_tv[4];
    for (int q = 0; q < 4; ++q) { o_tp[q] = take(4 * ((int64_t)m + 1)); o_ti[q] = take(4 * (int64_t)S.cap); o_tv[q] = take(8 * (int64_t)S.cap); }
    // the schedules: one set of work arrays, one image arena, lists and descriptors per schedule (four workgroups build them side by side)
    const int32_t nlev_cap = m + 1;
    S.x_cap = (int32_t)std::min<int64_t>(INT32_MAX / 8, 4 * (int64_t)S.cap / 3 + 2 * (int64_t)m + 64);       // expanded rows of the fused groups
    S.pool_cap = (int32_t)std::min<int64_t>(INT32_MAX / 8, 2 * (int64_t)S.x_cap);
    S.img_cap = 64 + 40 * ((int64_t)S.x_cap + 2 * (int64_t)m + 64) + 16 * ((int64_t)m + 8);
    const int64_t n_words = m / 32 + 3;
    struct SchedOff { int64_t m4[10], lv[6], bits[2], xs, xc, xv0, xvn, pool, tmp, tmp2, ovf, sc, img, desc, triv, reach, lof, viap, viapos, rhs_src, rhs_pos, tbits; } so[4];
    for (int q = 0; q < 4; ++q) {
        SchedOff& f = so[q];
        for (auto& v : f.m4) v = take(4 * (int64_t)m);      // indeg lev order grp xbeg xlen lg loff rhs_id (+1 spare)
        for (auto& v : f.lv) v = take(4 * ((int64_t)nlev_cap + 2));
        for (auto& v : f.bits) v = take(4 * n_words);
        f.xs = take(4 * (int64_t)S.x_cap); f.xc = take(8 * (int64_t)S.x_cap); f.xv0 = take(4 * (int64_t)S.x_cap); f.xvn = take(4 * (int64_t)S.x_cap);
        f.pool = take(4 * (int64_t)S.pool_cap);
        f.tmp = take(4 * ((int64_t)m + 2)); f.tmp2 = take(4 * ((int64_t)m + 2)); f.ovf = take(4 * ((int64_t)m + 1)); f.sc = take(128);
        f.img = take(S.img_cap); f.desc = take(4 * LUF_D_WORDS); f.triv = take(4 * (int64_t)m); f.reach = take(4 * (int64_t)m); f.lof = take(4 * (int64_t)m);
        f.viap = take(4 * ((int64_t)m + 1)); f.viapos = take(4 * (int64_t)S.pool_cap); f.rhs_src = take(4 * (int64_t)m); f.rhs_pos = take(4 * (int64_t)m);
        f.tbits = take(4 * n_words);
    }
    const int64_t o_pinfo = take((int64_t)sizeof(FtPivotInfo) * m);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&S.d_buf), (size_t)o));
    HIP_TRY(hipMemset(S.d_buf, 0, (size_t)o));
    char* const B = S.d_buf;
    HIP_TRY(hipMemcpy(B + o_rptr, rcount.data(), 4 * ((size_t)m + 1), hipMemcpyHostToDevice));
    if (nnz) {
        HIP_TRY(hipMemcpy(B + o_rcol, rcol.data(), 4 * (size_t)nnz, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(B + o_rval, rval.data(), 8 * (size_t)nnz, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemcpy(B + o_art, art_of_row.data(), 4 * (size_t)m, hipMemcpyHostToDevice));
    if (wrapped_na_ > 0) HIP_TRY(hipMemcpy(B + o_wr, column_to_row_.data(), 4 * (size_t)wrapped_na_, hipMemcpyHostToDevice));
    auto I32 = [&](int64_t at) { return reinterpret_cast<int32_t*>(B + at); };
    auto F64 = [&](int64_t at) { return reinterpret_cast<double*>(B + at); };
    S.M = LufMatrix{};
    S.M.m = m; S.M.na = na; S.M.n_provider = nprov;
    S.M.csc = csc(); S.M.ct = table();
    S.M.rptr = I32(o_rptr); S.M.rcol = I32(o_rcol); S.M.rval = F64(o_rval); S.M.art_of_row = I32(o_art);
    S.M.wrapped_na = wrapped_na_; S.M.wrapped_row = I32(o_wr);
    LufWork& W = S.W;
    W.pos_p = I32(o_posp); W.pos_a = I32(o_posa);
    W.wrow_pos = I32(o_m[0]); W.rcount = I32(o_m[1]); W.ccount = I32(o_m[2]); W.claim = I32(o_m[3]); W.claim2 = I32(o_m[4]);
    W.list = I32(o_m[5]); W.list2 = I32(o_m[6]); W.piv = I32(o_m[7]); W.brow = I32(o_m[8]); W.bcol = I32(o_m[9]);
    W.lrow = I32(o_m[10]); W.lcol = I32(o_m[11]); W.part = I32(o_part);
    W.nb_cap = S.nb_cap;
    W.rbeg = I32(o_nb[0]); W.rlen = I32(o_nb[1]); W.rcap = I32(o_nb[2]); W.ract = I32(o_nb[3]); W.cact = I32(o_nb[4]); W.bcc = I32(o_nb[5]);
    W.bstep_row = I32(o_nb[6]); W.bstep_col = I32(o_nb[7]); W.cpiv = I32(o_nb[8]); W.prank = I32(o_nb[9]); W.acc = I32(o_nb[10]);
    auto U64 = [&](int64_t at) { return reinterpret_cast<unsigned long long*>(B + at); };
    W.pval = F64(o_nb8[0]); W.cmax = U64(o_nb8[1]); W.rowmark = U64(o_nb8[2]); W.colbest = U64(o_nb8[3]); W.cprio = U64(o_nb8[4]);
    W.ecol = I32(o_ecol); W.eval = F64(o_eval); W.arena_cap = (int32_t)arena_cap;
    W.lt_row = I32(o_ltr); W.lt_step = I32(o_lts); W.lt_val = F64(o_ltv); W.lt_cap = (int32_t)arena_cap; W.lt_ptr = I32(o_ltp); W.lt_ord = I32(o_lto);
    W.counters = I32(o_cnt); W.red = U64(o_red);
    W.scalars = I32(o_sc);
    W.dense = F64(o_dense); W.dint = I32(o_dint); W.dense_cap = dense_cap;
    W.ut_row = I32(o_utr); W.ut_col = I32(o_utc); W.ut_val = F64(o_utv); W.vw = I32(o_vw); W.vtmp = I32(o_vtmp); W.vtmp_cap = (int32_t)arena_cap;
    LufOut& O = S.O;
    O.status = I32(o_status); O.rowperm = I32(o_rowperm); O.colperm = I32(o_colperm); O.row_step = I32(o_rstep); O.col_step = I32(o_cstep);
    O.diag = F64(o_diag);
    LufTriangle* tri[4] = {&O.Lf, &O.Uf, &O.Ub, &O.Lb};
    for (int q = 0; q < 4; ++q) { tri[q]->ptr = I32(o_tp[q]); tri[q]->idx = I32(o_ti[q]); tri[q]->val = F64(o_tv[q]); }
    O.cap = S.cap;
    {
        const bool wide = ft_big_;
        const LufTriangle* tri4[4] = {&O.Lf, &O.Uf, &O.Ub, &O.Lb};
        const LufTriangle* trt4[4] = {&O.Lb, &O.Ub, &O.Uf, &O.Lf};     // the transposed pattern of each
        // (fused schedules read copies of some right-hand sides behind x: ft_rhs_cap_ words of the layout; RELP_FUSE_LANES, read at create)
        const int32_t fuse = ft_rhs_cap_ > 0 ? lu_fuse_lanes_env_ : 0;
        for (int q = 0; q < 4; ++q) {
            const SchedOff& f = so[q];
            const bool maskable = q == 1 || q == 2;
            S.sin[q] = LufSchedIn{m, tri4[q]->ptr, tri4[q]->idx, tri4[q]->val, trt4[q]->ptr, trt4[q]->idx, maskable ? O.diag : nullptr, maskable ? 1 : 0,
                                  wide ? 1 : 0, wide ? 512 : 0x7fffffff, fuse, ft_rhs_cap_, ft_tier_ >= 2 ? 1 : 0};
            LufSchedWork& w = S.SW[q];
            w = LufSchedWork{};
            w.indeg = I32(f.m4[0]); w.lev = I32(f.m4[1]); w.order = I32(f.m4[2]); w.grp = I32(f.m4[3]); w.xbeg = I32(f.m4[4]); w.xlen = I32(f.m4[5]);
            w.lg = I32(f.m4[6]); w.loff = I32(f.m4[7]); w.rhs_id = I32(f.m4[8]);
            w.lvl_ptr = I32(f.lv[0]); w.lvl_grp = I32(f.lv[1]); w.grp_lvl0 = I32(f.lv[2]); w.grp_lane0 = I32(f.lv[3]); w.grp_pass0 = I32(f.lv[4]); w.grp_lanes = I32(f.lv[5]);
            w.bits0 = reinterpret_cast<uint32_t*>(B + f.bits[0]); w.bits1 = reinterpret_cast<uint32_t*>(B + f.bits[1]);
            w.x_src = I32(f.xs); w.x_coef = F64(f.xc); w.x_v0 = I32(f.xv0); w.x_vn = I32(f.xvn); w.x_cap = S.x_cap; w.pool = I32(f.pool); w.pool_cap = S.pool_cap;
            w.tmp = I32(f.tmp); w.tmp2 = I32(f.tmp2); w.ovf_off = I32(f.ovf); w.sc = I32(f.sc); w.nlev_cap = nlev_cap;
            S.sout[q] = LufSchedOut{B + f.img, S.img_cap, I32(f.desc), I32(f.triv), I32(f.reach), I32(f.lof), I32(f.viap), I32(f.viapos), S.pool_cap,
                                    I32(f.rhs_src), ft_tier_ >= 2 ? I32(f.rhs_pos) : nullptr, ft_tier_ >= 2 ? reinterpret_cast<uint32_t*>(B + f.tbits) : nullptr};
        }
    }
    S.pinfo = reinterpret_cast<FtPivotInfo*>(B + o_pinfo);
    S.dirty = false;
    S.key[0] = m_; S.key[1] = nr_artificial_; S.key[2] = phase_; S.key[3] = wrapped_na_;
    return RELP_OK;
}

// P B Q = L U of the basis in d_basis_ by the device kernel.  `resident`: the second kernel also builds the four solve
// schedules on the device and they are installed where they are (persistent pivot kernel only) -- no factor leaves the device;
// otherwise the factors are downloaded and scheduled like lu_factor's (the product-form fallback of very large bases).
// RELP_E_UNSUPPORTED: the bump exceeds the dense working copy, or an arena is too small (the caller factorises on the host).
relp_status_t Engine::lu_factor_on_device(int32_t* device_status) {
    // (the device factorisation packs its images for layouts 0 and 1 and keeps its working sets in one workgroup's LDS:
    // beyond their row range the host factorises)
    if (ft_tier_ >= 2) return RELP_E_UNSUPPORTED;
    if (!luf_ || luf_->dirty || luf_->key[0] != m_ || luf_->key[1] != nr_artificial_ || luf_->key[2] != phase_ || luf_->key[3] != wrapped_na_) {
        const relp_status_t st = luf_prepare();            // (a row removal or the phase switch renumbers rows / columns)
        if (st) return st;
    }
    LufState& S = *luf_;
    const bool resident = ft_ && !luf_download_;
    const auto t0 = std::chrono::steady_clock::now();
    S.M.csc = csc(); S.M.ct = table();                  // (pointers may have been re-allocated)
    int32_t status[8] = {0}, desc[4][LUF_D_WORDS] = {};
    // (the bump's working set in LDS when it fits, RELP_LUF_LDS=0 keeps it in L2; an arena that overflows in LDS: once more from L2)
    static const bool lds_env = !(std::getenv("RELP_LUF_LDS") && std::atoi(std::getenv("RELP_LUF_LDS")) == 0);
    for (int attempt = lds_env ? 0 : 1; attempt < 2; ++attempt) {
        launch_lu_factor(S.M, d_basis_, S.W, S.O, stream_, attempt == 0);
        if (resident) launch_lu_schedules(S.sin, S.SW, S.sout, S.O.status, S.pinfo, stream_);
        HIP_TRY(hipMemcpyAsync(status, S.O.status, sizeof status, hipMemcpyDeviceToHost, stream_));
        if (resident) for (int q = 0; q < 4; ++q) HIP_TRY(hipMemcpyAsync(desc[q], S.sout[q].desc, sizeof desc[q], hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
        if (!(attempt == 0 && status[0] == LUF_NO_ROOM)) break;
        ++luf_lds_retries_;
    }
    luf_kernel_us_ += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    ++luf_runs_;
    if (std::getenv("RELP_DEBUG") && luf_runs_ % 200 == 0) {          // phase clocks of the factorisation kernel (relp_lu_factor_core.h: LUF_LAP)
        int32_t cnt[128] = {0};
        HIP_TRY(hipMemcpy(cnt, S.W.counters, sizeof cnt, hipMemcpyDeviceToHost));
        const unsigned long long* ph = reinterpret_cast<const unsigned long long*>(cnt + 8);
        static const char* nm[11] = {"maps+counts", "peel", "bump setup", "r:column max", "r:proposals", "r:independence+accept", "r:elimination", "r:leave + dense finish",
                                     "triplets", "row views", "column views"};
        std::fprintf(stderr, "[relp] device factorisation, %lld runs (%d with the bump in LDS, %lld of them again in global memory, dense finish on %.1f rows), %.0f us each (host clock, schedules included), last: %d rounds; clocks per run:",
                     (long long)luf_runs_, cnt[3], (long long)luf_lds_retries_, cnt[3] ? (double)cnt[4] / cnt[3] : 0.0, luf_kernel_us_ / luf_runs_, cnt[2]);
        for (int i = 0; i < 11; ++i) std::fprintf(stderr, " %s %.0f", nm[i], (double)ph[i] / luf_runs_);
        std::fprintf(stderr, "\n");
        if (ph[12 + 5]) std::fprintf(stderr, "[relp]   wave 0 of the elimination: rows %llu, hits %llu, pivot entries %llu; clocks: idle/loop %llu, row setup %llu, hit prologue %llu, entries %llu, write-back %llu\n",
                                     ph[17], ph[18], ph[19], ph[12], ph[13], ph[14], ph[15], ph[16]);
        static const char* sn[4] = {"L", "U", "U'", "L'"};
        for (int q = 0; q < 4 && resident; ++q) {
            int32_t sc[32] = {0};
            HIP_TRY(hipMemcpy(sc, S.SW[q].sc, sizeof sc, hipMemcpyDeviceToHost));
            const unsigned long long* sp = reinterpret_cast<const unsigned long long*>(sc + 8);
            std::fprintf(stderr, "[relp]   schedule %s clocks per run: levels %.0f, fusion %.0f, classify + offsets %.0f, image %.0f, via %.0f\n", sn[q],
                         (double)sp[0] / luf_runs_, (double)sp[1] / luf_runs_, (double)sp[2] / luf_runs_, (double)sp[3] / luf_runs_, (double)sp[4] / luf_runs_);
        }
    }
    luf_last_bump_ = status[1]; luf_last_peeled_ = status[2];
    if (device_status) *device_status = status[0];
    if (status[0] == LUF_SINGULAR) return fail(RELP_E_SINGULAR, "singular basis (device factorisation)");
    if (status[0] != LUF_OK) return RELP_E_UNSUPPORTED;      // bump too large / no room: not an error, the host takes over
    if (resident) for (int q = 0; q < 4; ++q) if (desc[q][LUF_D_STATUS] != LUF_OK) return RELP_E_UNSUPPORTED;
    const int32_t m = m_, nl = status[3], nu = status[4];
    hlu_ = LUFactors{};
    hlu_.m = m; hlu_.nnz_l = nl; hlu_.nnz_u = (int64_t)nu + m;
    S.resident = resident;
    if (!resident) return luf_download_factors();
    // install the factors where they are
    dlu_ = DeviceLU{};
    dlu_.m = m; dlu_.rowperm = S.O.rowperm; dlu_.colperm = S.O.colperm;
    dlu_.Uf.idx = S.O.Uf.idx; dlu_.Uf.val = S.O.Uf.val; dlu_.Uf.nnz = nu;
    TriangularSchedule* hs[4] = {&hlu_.Lf, &hlu_.Uf, &hlu_.Ub, &hlu_.Lb};
    DeviceSchedule* ds[4] = {&dlu_.Lf, &dlu_.Uf, &dlu_.Ub, &dlu_.Lb};
    fts_.m = m;
    fts_.inv_rowperm = S.O.row_step; fts_.inv_colperm = S.O.col_step; fts_.task_uf = S.O.row_step; fts_.task_ub = S.O.row_step;
    fts_.lev_ub = S.sout[2].level_of; fts_.pinfo = S.pinfo;
    const int64_t base = (int64_t)ft_lds_base_bytes(m, ft_tcap_, ft_eta_cap_, ft_tier_, ft_rhs_cap_);
    fts_.stage_bytes = (int32_t)std::max<int64_t>(0, kFtLdsBudget - base);
    int64_t need = 0;
    for (int q = 0; q < 4; ++q) {
        const int32_t* d = desc[q];
        const LufImageLayout L = luf_image_layout(m, d[LUF_D_PASSES], d[LUF_D_LEVELS], d[LUF_D_LANES], d[LUF_D_OVF], ft_big_);
        char* const img = S.sout[q].image;
        EllSchedule& e = fts_.ell[q];
        e = EllSchedule{};
        e.passes = reinterpret_cast<const EllPass*>(img + L.passes); e.lvl_pass = reinterpret_cast<const int32_t*>(img + L.lvl_pass);
        e.rdiag = reinterpret_cast<double*>(img + L.rdiag); e.sval = reinterpret_cast<double*>(img + L.sval);
        e.oval = reinterpret_cast<const double*>(img + L.oval); e.rovf = reinterpret_cast<const int32_t*>(img + L.rovf);
        e.sidx = reinterpret_cast<const uint16_t*>(img + L.sidx); e.oidx = reinterpret_cast<const uint16_t*>(img + L.oidx);
        const bool maskable = q == 1 || q == 2;
        e.via_ptr = maskable ? S.sout[q].via_ptr : nullptr; e.via_pos = maskable ? S.sout[q].via_pos : nullptr;
        e.n_passes = d[LUF_D_PASSES]; e.n_levels = d[LUF_D_LEVELS]; e.m = m; e.n_lanes = d[LUF_D_LANES]; e.n_ovf = d[LUF_D_OVF];
        e.bytes = d[LUF_D_BYTES]; e.rhs_base = d[LUF_D_USES_RHS] ? m + 1 : 0;
        e.n_triv = d[LUF_D_TRIV]; e.triv = S.sout[q].triv; e.reach = S.sout[q].reach; e.rhs_src = S.sout[q].rhs_src;
        e.n_rhs = ft_big_ ? std::max(d[LUF_D_NRHS], 0) : -1;
        e.rhs_pos = S.sout[q].rhs_pos; e.triv_bits = S.sout[q].triv_bits;
        fts_.stage[q] = e.bytes <= fts_.stage_bytes ? 1 : 0;
        if (fts_.stage[q]) need = std::max<int64_t>(need, e.bytes);
        ds[q]->n_levels = d[LUF_D_LEVELS];
        hs[q]->level_ptr.assign((size_t)d[LUF_D_KAHN_LEVELS] + 1, 0);     // (lu_stats reports the level counts)
    }
    fts_.lds_bytes = (int32_t)(base + need);
    return RELP_OK;
}

// the factors of the last device factorisation -> hlu_ with level schedules, like after lu_factor (relp_lu.cpp)
relp_status_t Engine::luf_download_factors() {
    LufState& S = *luf_;
    const int32_t m = m_, nl = (int32_t)hlu_.nnz_l, nu = (int32_t)(hlu_.nnz_u - m);
    hlu_.rowperm.resize(m); hlu_.colperm.resize(m);
    std::vector<double> diag(m), ones(m, 1.0);
    HIP_TRY(hipMemcpyAsync(hlu_.rowperm.data(), S.O.rowperm, 4 * (size_t)m, hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipMemcpyAsync(hlu_.colperm.data(), S.O.colperm, 4 * (size_t)m, hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipMemcpyAsync(diag.data(), S.O.diag, 8 * (size_t)m, hipMemcpyDeviceToHost, stream_));
    TriangularSchedule* sch[4] = {&hlu_.Lf, &hlu_.Uf, &hlu_.Ub, &hlu_.Lb};
    const LufTriangle* tri[4] = {&S.O.Lf, &S.O.Uf, &S.O.Ub, &S.O.Lb};
    const int32_t cnt[4] = {nl, nu, nu, nl};
    for (int q = 0; q < 4; ++q) {
        sch[q]->ptr.resize(m + 1); sch[q]->idx.resize(cnt[q]); sch[q]->val.resize(cnt[q]);
        HIP_TRY(hipMemcpyAsync(sch[q]->ptr.data(), tri[q]->ptr, 4 * ((size_t)m + 1), hipMemcpyDeviceToHost, stream_));
        if (cnt[q]) {
            HIP_TRY(hipMemcpyAsync(sch[q]->idx.data(), tri[q]->idx, 4 * (size_t)cnt[q], hipMemcpyDeviceToHost, stream_));
            HIP_TRY(hipMemcpyAsync(sch[q]->val.data(), tri[q]->val, 8 * (size_t)cnt[q], hipMemcpyDeviceToHost, stream_));
        }
    }
    HIP_TRY(hipStreamSynchronize(stream_));
    lu_levels_from_rows(m, ones, true, &hlu_.Lf);
    lu_levels_from_rows(m, diag, false, &hlu_.Uf);
    lu_levels_from_rows(m, diag, true, &hlu_.Ub);
    lu_levels_from_rows(m, ones, false, &hlu_.Lb);
    return RELP_OK;
}

// hlu_ as the host needs it for inspection (relp_lu_get_upper, relp_lu_factor_residual): after a device-resident
// factorisation the rows are still on the device
relp_status_t Engine::lu_host_factors() {
    if (luf_ && luf_->resident && hlu_.Lf.ptr.empty()) return luf_download_factors();
    return RELP_OK;
}

bool Engine::luf_is_resident() const { return luf_ && luf_->resident; }

relp_status_t Engine::lu_set_device_factorisation(bool on) {
    if (!lu_) return fail(RELP_E_UNSUPPORTED, "the device factorisation belongs to the LU engine");
    luf_enabled_ = on;
    return RELP_OK;
}

relp_status_t Engine::luf_stats(int64_t* out6) const {
    if (!lu_) return RELP_E_STATE;
    out6[0] = luf_enabled_ ? 1 : 0; out6[1] = luf_runs_; out6[2] = luf_fallbacks_; out6[3] = (int64_t)luf_kernel_us_;
    out6[4] = luf_last_bump_; out6[5] = luf_last_peeled_;
    return RELP_OK;
}

}  // namespace relp

namespace relp {

// max |P B Q - L U| over all entries, for the factors currently installed (whoever computed them) and the basis they were
// computed for; dense arithmetic on the host, so only for m <= 1,024 (tests: the reference's factorisation cases of
// decomposition/mod.rs:301-491 through the ABI).  *out = -1 when m is larger.
relp_status_t Engine::lu_factor_residual(double* out) {
    if (!lu_) return fail(RELP_E_UNSUPPORTED, "the LU engine's factors");
    *out = -1.0;
    const int32_t m = m_;
    if (m > 1024 || hlu_.m != m) return RELP_OK;
    if (ft_) {                                             // updates pending: the factors are those of an earlier basis
        relp_status_t hs = ft_read_hdr();
        if (hs) return hs;
    }
    if (since_flush_ > 0) return RELP_OK;
    relp_status_t st = lu_host_factors();
    if (st) return st;
    if ((st = lu_download_basis())) return st;
    std::vector<std::vector<std::pair<int32_t, double>>> cols;
    if ((st = lu_basis_columns(cols))) return st;
    std::vector<double> a((size_t)m * m, 0.0), L((size_t)m * m, 0.0), U((size_t)m * m, 0.0);
    for (int32_t c = 0; c < m; ++c) for (auto& e : cols[c]) a[(size_t)e.first * m + c] += e.second;
    for (int32_t k = 0; k < m; ++k) {
        L[(size_t)k * m + k] = 1.0; U[(size_t)k * m + k] = hlu_.Uf.diag[k];
        for (int32_t e = hlu_.Lf.ptr[k]; e < hlu_.Lf.ptr[k + 1]; ++e) L[(size_t)k * m + hlu_.Lf.idx[e]] = hlu_.Lf.val[e];
        for (int32_t e = hlu_.Uf.ptr[k]; e < hlu_.Uf.ptr[k + 1]; ++e) U[(size_t)k * m + hlu_.Uf.idx[e]] = hlu_.Uf.val[e];
    }
    double worst = 0.0;
    for (int32_t k = 0; k < m; ++k)
        for (int32_t l = 0; l < m; ++l) {
            double s = 0.0;
            for (int32_t q = 0; q <= std::min(k, l); ++q) s += L[(size_t)k * m + q] * U[(size_t)q * m + l];
            worst = std::max(worst, std::fabs(s - a[(size_t)hlu_.rowperm[k] * m + hlu_.colperm[l]]));
        }
    *out = worst;
    return RELP_OK;
}

}  // namespace relp
