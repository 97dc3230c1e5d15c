// relp_engine_ext.cpp -- Engine: the `BasisInverse` surface (carry/mod.rs:68-157) next to the tableau-level calls of
// relp_engine.cpp: basis_inverse_row, should_refactor, generate_column / cost_difference for a column the caller
// supplies, change_basis on the inverse alone, and the inspection of the LU engine's update file in the reference's own
// coordinates (lower_upper/mod.rs:40-57: `upper_triangular`, `updates`), which its known-answer tests compare.
#include "relp_engine_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace relp {

// BasisInverse::basis_inverse_row (carry/mod.rs:145-150; lower_upper/mod.rs:204-222; basis_inverse_rows.rs:181-183)
relp_status_t Engine::basis_inverse_row(int32_t row, double* out_m) {
    if (row < 0 || row >= m_) return fail(RELP_E_ARG, "row out of range");
    if (cfg_.shard_count > 1) return fail(RELP_E_UNSUPPORTED, "B^-1 is sharded");
    if (lu_) {
        if (ft_) launch_ft_btran(dlu_, fts_, ft_problem(0), row, nullptr, d_rho_, stream_);
        else launch_lu_btran(dlu_, deferred(), nullptr, row, d_rho_, d_lu_scratch_, nullptr, stream_);
        HIP_TRY(hipMemcpyAsync(out_m, d_rho_, sizeof(double) * m_, hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
        return RELP_OK;
    }
    if (tableau_) {
        // B^-1 = the tableau columns that were the identity originally: row r of T over those columns
        launch_tab_row(tview(), deferred(), row, d_aq_big(), d_rec_, stream_);
        std::vector<double> trow(n_store_);
        HIP_TRY(hipMemcpyAsync(trow.data(), d_aq_big(), sizeof(double) * n_store_, hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
        for (int32_t k = 0; k < m_; ++k) out_m[k] = trow[idcol_h_[k]];
        return RELP_OK;
    }
    enqueue_flush();
    HIP_TRY(hipMemcpyAsync(out_m, dBinv_ + (int64_t)row * ld_b_, sizeof(double) * m_, hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    return RELP_OK;
}

// BasisInverse::should_refactor: lower_upper/mod.rs:199-202 (`updates.len() > 10`; here: the configured number of
// pending updates, relp_config_t.update_block, or a full eta pool); basis_inverse_rows.rs:175-179 (never)
relp_status_t Engine::should_refactor(int32_t* out) {
    int32_t v = 0;
    if (lu_) {
        if (ft_) {
            relp_status_t st = ft_read_hdr();
            if (st) return st;
            v = (ft_need_refactor_ || h_ft_hdr_[0] >= fts_.max_updates) ? 1 : 0;
        } else {
            v = since_flush_ >= block_ ? 1 : 0;
        }
    }
    if (out) *out = v;
    return RELP_OK;
}

// BasisInverse::generate_column(original_column) (carry/mod.rs:108-118; lower_upper/mod.rs:157-190;
// basis_inverse_rows.rs:144-155) for a sparse column over the m tableau rows that the caller supplies
relp_status_t Engine::generate_column_of(const int32_t* idx, const double* val, int32_t nnz, double* out_m) {
    if (nnz < 0 || (nnz > 0 && (!idx || !val))) return fail(RELP_E_ARG, "bad column");
    if (cfg_.shard_count > 1) return fail(RELP_E_UNSUPPORTED, "generate_column_of in sharded mode");
    std::vector<double> a(ld_b_, 0.0);
    for (int32_t k = 0; k < nnz; ++k) {
        if (idx[k] < 0 || idx[k] >= m_) return fail(RELP_E_ARG, "row index out of range");
        a[idx[k]] = val[k];
    }
    if (tableau_) {
        // alpha = B^-1 a with B^-1 read off the tableau (host product: not a hot path of this engine)
        std::vector<double> binv((size_t)m_ * m_);
        relp_status_t st = get_basis_inverse(binv.data());
        if (st) return st;
        std::vector<double> alpha(m_, 0.0);
        for (int32_t i = 0; i < m_; ++i) {
            double s = 0.0;
            for (int32_t k = 0; k < nnz; ++k) s = std::fma(binv[(size_t)i * m_ + idx[k]], val[k], s);
            alpha[i] = s;
        }
        HIP_TRY(hipMemcpyAsync(d_alpha_, alpha.data(), sizeof(double) * m_, hipMemcpyHostToDevice, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
        if (out_m) std::memcpy(out_m, alpha.data(), sizeof(double) * m_);
        return RELP_OK;
    }
    relp_status_t st = download_rec();
    if (st) return st;
    h_rec_->outcome = DEV_RUNNING;
    if ((st = upload_rec())) return st;
    HIP_TRY(hipMemcpyAsync(d_aq_, a.data(), sizeof(double) * ld_b_, hipMemcpyHostToDevice, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));              // `a` is stack-owned
    if (lu_ && ft_) {
        launch_ft_ftran(dlu_, fts_, ft_problem(0), -1, d_aq_, d_alpha_, stream_);
    } else if (lu_) {
        launch_lu_ftran(dlu_, d_aq_, d_v_, d_lu_scratch_, d_rec_, stream_);
        launch_apply_w(deferred(), m_, d_v_, d_alpha_, d_rec_, stream_);
    } else {
        enqueue_flush();
        double* Binv = dBinv_ - (int64_t)row_lo_ * ld_b_;
        launch_ftran(Binv, ld_b_, m_, row_lo_, row_hi_, d_aq_, d_alpha_, 0, d_rec_, stream_);
    }
    if (out_m) HIP_TRY(hipMemcpyAsync(out_m, d_alpha_, sizeof(double) * m_, hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    return RELP_OK;
}

// InverseMaintener::cost_difference (carry/mod.rs:572-577): (-pi) . a for a column the caller supplies
relp_status_t Engine::cost_difference_of(const int32_t* idx, const double* val, int32_t nnz, double* out) {
    if (nnz < 0 || (nnz > 0 && (!idx || !val)) || !out) return fail(RELP_E_ARG, "bad column");
    std::vector<double> mp(m_);
    relp_status_t st = get_vector(1, mp.data());
    if (st) return st;
    double s = 0.0;
    for (int32_t k = 0; k < nnz; ++k) {
        if (idx[k] < 0 || idx[k] >= m_) return fail(RELP_E_ARG, "row index out of range");
        s = std::fma(mp[idx[k]], val[k], s);              // vector/dense.rs:81-92, in the column's order
    }
    *out = s;
    return RELP_OK;
}

// BasisInverse::change_basis(pivot_row_index, column) on the inverse alone (lower_upper/mod.rs:92-155): the column
// is the one of the last generate_column / generate_column_of, whose spike the engine kept (ColumnAndSpike, mod.rs:378-381)
relp_status_t Engine::lu_change_basis(int32_t row) {
    if (!lu_ || !ft_) return fail(RELP_E_UNSUPPORTED, "change_basis on the inverse alone is the LU engine's (Forrest-Tomlin mode)");
    if (row < 0 || row >= m_) return fail(RELP_E_ARG, "row out of range");
    relp_status_t st = ft_read_hdr();
    if (st) return st;
    if (h_ft_hdr_[0] >= ft_tcap_) return fail(RELP_E_STATE, "update file full: refactor first (relp_flush)");
    if ((st = download_rec())) return st;
    h_rec_->outcome = DEV_RUNNING;
    h_rec_->r = row;
    if ((st = upload_rec())) return st;
    launch_ft_update(dlu_, fts_, ft_problem(0), stream_);
    if ((st = ft_read_hdr())) return st;
    if (h_ft_hdr_[2] == 2) return fail(RELP_E_STATE, "the eta pool is full: refactor first (relp_flush), then generate the column again");
    return RELP_OK;
}

// LUDecomposition { lower_triangular, upper_triangular, .. } given literally, P = Q = I (lower_upper/mod.rs:33-57)
relp_status_t Engine::lu_set_factors(const int64_t* l_ptr, const int32_t* l_idx, const double* l_val, const int64_t* u_ptr,
                                     const int32_t* u_idx, const double* u_val) {
    if (!lu_) return fail(RELP_E_UNSUPPORTED, "relp_lu_set_factors is the LU engine's");
    if (!l_ptr || !u_ptr) return fail(RELP_E_ARG, "missing factors");
    std::vector<std::vector<std::pair<int32_t, double>>> lc(m_), uc(m_);
    for (int32_t j = 0; j < m_; ++j) {
        for (int64_t e = l_ptr[j]; e < l_ptr[j + 1]; ++e) lc[j].emplace_back(l_idx[e], l_val[e]);
        for (int64_t e = u_ptr[j]; e < u_ptr[j + 1]; ++e) uc[j].emplace_back(u_idx[e], u_val[e]);
    }
    std::string msg;
    if (!lu_from_triangles(m_, lc, uc, &hlu_, &msg)) return fail(RELP_E_ARG, msg);
    HIP_TRY(hipStreamSynchronize(stream_));
    relp_status_t st = lu_upload_factors();
    if (st) return st;
    if (ft_) { if ((st = ft_reset())) return st; }
    else launch_flush_reset(deferred(), d_rec_, stream_);
    HIP_TRY(hipStreamSynchronize(stream_));
    since_flush_ = 0;
    return RELP_OK;
}

namespace {
// everything the inspection calls need, downloaded once
struct FtHostView {
    int32_t t = 0, eta_used = 0, tcap = 0, ldt = 0, m = 0;
    std::vector<int32_t> slot_pivot, slot_prev, slot_live, tslot, eta_off, spk_off, eta_idx, spk_idx;
    std::vector<double> TC, eta_val, spk_val;
    // position of every pivot in the order BEFORE update k (k = t: the current order)
    std::vector<int32_t> positions_before(int32_t k) const {
        std::vector<int32_t> last(m, -1);
        for (int32_t s = 0; s < k; ++s) last[slot_pivot[s]] = s;
        std::vector<int32_t> pos(m, 0);
        int32_t at = 0;
        for (int32_t p = 0; p < m; ++p) if (last[p] < 0) pos[p] = at++;
        for (int32_t s = 0; s < k; ++s) if (last[slot_pivot[s]] == s) pos[slot_pivot[s]] = at++;
        return pos;
    }
};
}  // namespace

static relp_status_t ft_download(const FtState& st, hipStream_t stream, FtHostView* v, std::string* err) {
    auto get = [&](void* dst, const void* src, size_t bytes) { return bytes ? hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) : hipSuccess; };
    if (hipStreamSynchronize(stream) != hipSuccess) { *err = "sync failed"; return RELP_E_HIP; }
    int32_t hdr[4];
    if (get(hdr, st.hdr, sizeof(hdr)) != hipSuccess) { *err = "copy failed"; return RELP_E_HIP; }
    v->t = hdr[0]; v->eta_used = hdr[1]; v->tcap = st.tcap; v->ldt = st.ldt; v->m = st.m;
    const int nwp = kFtWaves + 1;
    v->slot_pivot.resize(st.tcap); v->slot_prev.resize(st.tcap); v->slot_live.resize(st.tcap); v->tslot.resize(st.m);
    v->eta_off.resize((size_t)st.tcap * nwp); v->spk_off.resize((size_t)st.tcap * nwp);
    v->TC.resize((size_t)st.tcap * st.ldt); v->eta_idx.resize(std::max(v->eta_used, 1)); v->eta_val.resize(std::max(v->eta_used, 1));
    v->spk_idx.resize((size_t)std::max(v->t, 1) * st.m); v->spk_val.resize((size_t)std::max(v->t, 1) * st.m);
    hipError_t e = hipSuccess;
    auto acc = [&](hipError_t x) { if (e == hipSuccess) e = x; };
    acc(get(v->slot_pivot.data(), st.slot_pivot, 4 * (size_t)st.tcap)); acc(get(v->slot_prev.data(), st.slot_prev, 4 * (size_t)st.tcap));
    acc(get(v->slot_live.data(), st.slot_live, 4 * (size_t)st.tcap)); acc(get(v->tslot.data(), st.tslot, 4 * (size_t)st.m));
    acc(get(v->eta_off.data(), st.eta_off, 4 * v->eta_off.size())); acc(get(v->spk_off.data(), st.spk_off, 4 * v->spk_off.size()));
    acc(get(v->TC.data(), st.TC, 8 * v->TC.size()));
    acc(get(v->eta_idx.data(), st.eta_idx, 4 * (size_t)v->eta_used)); acc(get(v->eta_val.data(), st.eta_val, 8 * (size_t)v->eta_used));
    acc(get(v->spk_idx.data(), st.spk_idx, 4 * (size_t)v->t * st.m)); acc(get(v->spk_val.data(), st.spk_val, 8 * (size_t)v->t * st.m));
    if (e != hipSuccess) { *err = std::string("copy failed: ") + hipGetErrorString(e); return RELP_E_HIP; }
    return RELP_OK;
}

relp_status_t Engine::lu_updates(int32_t* count) {
    if (!lu_) return fail(RELP_E_UNSUPPORTED, "the LU engine's update file");
    if (ft_) { relp_status_t st = ft_read_hdr(); if (st) return st; *count = h_ft_hdr_[0]; }
    else *count = (int32_t)since_flush_;
    return RELP_OK;
}

// `updates[k]` = (EtaFile { values, pivot, len }, RotateToBack { index: pivot }) (lower_upper/mod.rs:56, eta_file.rs:14-18):
// the pivot position and the (position, value) pairs of r, positions as they were BEFORE that update's rotation
relp_status_t Engine::lu_get_update(int32_t k, int32_t* pivot, int32_t* idx, double* val, int32_t cap, int32_t* nnz) {
    if (!lu_ || !ft_) return fail(RELP_E_UNSUPPORTED, "the eta file is the LU engine's (Forrest-Tomlin mode)");
    FtHostView v;
    relp_status_t st = ft_download(fts_, stream_, &v, &err_);
    if (st) return st;
    if (k < 0 || k >= v.t) return fail(RELP_E_ARG, "update index out of range");
    const std::vector<int32_t> pos = v.positions_before(k);
    std::vector<std::pair<int32_t, double>> out;
    const int nwp = kFtWaves + 1;
    for (int32_t e = v.eta_off[(size_t)k * nwp]; e < v.eta_off[(size_t)k * nwp + kFtWaves]; ++e)
        if (v.eta_val[e] != 0.0) out.emplace_back(pos[v.eta_idx[e]], v.eta_val[e]);
    for (int32_t s = 0; s < k; ++s) {
        const double c = v.TC[(size_t)k * v.ldt + s];
        if (c != 0.0) out.emplace_back(pos[v.slot_pivot[s]], c);
    }
    std::sort(out.begin(), out.end());
    if (pivot) *pivot = pos[v.slot_pivot[k]];
    if (nnz) *nnz = (int32_t)out.size();
    for (int32_t i = 0; i < (int32_t)out.size() && i < cap; ++i) { if (idx) idx[i] = out[i].first; if (val) val[i] = out[i].second; }
    return RELP_OK;
}

// `upper_triangular` (lower_upper/mod.rs:48-52): column-major, rows sorted, the diagonal last in its column, in the
// CURRENT positions (after every rotation so far)
relp_status_t Engine::lu_get_upper(int64_t* col_ptr, int32_t* row_idx, double* values, int64_t cap, int64_t* nnz) {
    if (!lu_ || !ft_) return fail(RELP_E_UNSUPPORTED, "the updated U is the LU engine's (Forrest-Tomlin mode)");
    FtHostView v;
    relp_status_t st = ft_download(fts_, stream_, &v, &err_);
    if (st) return st;
    if ((st = lu_host_factors())) return st;               // (after a device-resident factorisation the rows are fetched now)
    const std::vector<int32_t> pos = v.positions_before(v.t);
    std::vector<std::vector<std::pair<int32_t, double>>> cols(m_);
    const int nwp = kFtWaves + 1;
    for (int32_t l = 0; l < m_; ++l) {
        if (v.tslot[l] >= 0) continue;                      // its column is a spike now
        auto& c = cols[pos[l]];
        for (int32_t e = hlu_.Ub.ptr[l]; e < hlu_.Ub.ptr[l + 1]; ++e) {
            const int32_t k = hlu_.Ub.idx[e];
            if (v.tslot[k] < 0 && hlu_.Ub.val[e] != 0.0) c.emplace_back(pos[k], hlu_.Ub.val[e]);
        }
        c.emplace_back(pos[l], hlu_.Ub.diag[l]);
    }
    for (int32_t s = 0; s < v.t; ++s) {
        if (!v.slot_live[s]) continue;
        auto& c = cols[pos[v.slot_pivot[s]]];
        const size_t base = (size_t)s * m_;
        for (int32_t e = v.spk_off[(size_t)s * nwp]; e < v.spk_off[(size_t)s * nwp + kFtWaves]; ++e) {
            const int32_t k = v.spk_idx[base + e];
            if (v.tslot[k] < 0 && v.spk_val[base + e] != 0.0) c.emplace_back(pos[k], v.spk_val[base + e]);
        }
        for (int32_t s2 = 0; s2 < s; ++s2)
            if (v.slot_live[s2] && v.TC[(size_t)s2 * v.ldt + s] != 0.0) c.emplace_back(pos[v.slot_pivot[s2]], v.TC[(size_t)s2 * v.ldt + s]);
        c.emplace_back(pos[v.slot_pivot[s]], v.TC[(size_t)s * v.ldt + s]);
    }
    int64_t at = 0;
    for (int32_t j = 0; j < m_; ++j) {
        std::sort(cols[j].begin(), cols[j].end());
        if (col_ptr) col_ptr[j] = at;
        for (auto& e : cols[j]) { if (at < cap) { if (row_idx) row_idx[at] = e.first; if (values) values[at] = e.second; } ++at; }
    }
    if (col_ptr) col_ptr[m_] = at;
    if (nnz) *nnz = at;
    return RELP_OK;
}

}  // namespace relp
