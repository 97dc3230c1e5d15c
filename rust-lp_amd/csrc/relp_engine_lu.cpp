// relp_engine_lu.cpp -- Engine: the sparse LU engine (RELP_ENGINE_LU): CSC upload, host refactorisation,
// one pivot.  See relp_engine.hpp and relp_lu.hpp.
#include "relp_engine_internal.hpp"

#include <algorithm>
#include <climits>
#include <cstring>

namespace relp {

// ------------------------------------------------------------------------------------------------
// Sparse LU engine: matrix in CSC, factors from the host (relp_lu.cpp), solves on the device
// ------------------------------------------------------------------------------------------------
relp_status_t Engine::lu_load_matrix(const relp_matrix_data_t& md) {
    hc_ptr_.assign(nr_normal_ + 1, 0);
    hc_idx_.clear(); hc_val_.clear();
    if (md.format == RELP_FORMAT_CSC) {
        if (md.matrix_memory != RELP_MEM_HOST) return fail(RELP_E_UNSUPPORTED, "CSC input must be in host memory");
        if (!md.col_ptr) return fail(RELP_E_ARG, "col_ptr missing");
        for (int32_t j = 0; j < nr_normal_; ++j) {
            for (int64_t p = md.col_ptr[j]; p < md.col_ptr[j + 1]; ++p) {
                const int32_t i = md.row_idx[p];
                if (i < 0 || i >= mc_) return fail(RELP_E_ARG, "row index out of range");
                if (md.values[p] == 0.0) continue;
                hc_idx_.push_back(i); hc_val_.push_back(md.values[p]);
            }
            hc_ptr_[j + 1] = (int64_t)hc_idx_.size();
        }
    } else if (md.format == RELP_FORMAT_DENSE) {
        if (nr_normal_ > 0 && mc_ > 0 && !md.dense) return fail(RELP_E_ARG, "dense matrix missing");
        const int64_t src_ld = md.dense_ld > 0 ? md.dense_ld : mc_;
        if (src_ld < mc_) return fail(RELP_E_ARG, "dense_ld < nr_constraints");
        std::vector<double> col(std::max(mc_, 1));
        for (int32_t j = 0; j < nr_normal_; ++j) {
            const double* src = md.dense + (int64_t)j * src_ld;
            if (md.matrix_memory == RELP_MEM_DEVICE) {
                HIP_TRY(hipMemcpy(col.data(), src, sizeof(double) * mc_, hipMemcpyDeviceToHost));
                src = col.data();
            }
            for (int32_t i = 0; i < mc_; ++i)
                if (src[i] != 0.0) { hc_idx_.push_back(i); hc_val_.push_back(src[i]); }
            hc_ptr_[j + 1] = (int64_t)hc_idx_.size();
        }
    } else {
        return fail(RELP_E_ARG, "unknown matrix format");
    }
    HIP_TRY(dev_alloc(&d_cptr_, nr_normal_ + 1));
    HIP_TRY(dev_alloc(&d_cidx_, (int64_t)hc_idx_.size()));
    HIP_TRY(dev_alloc(&d_cval_, (int64_t)hc_val_.size()));
    HIP_TRY(hipMemcpy(d_cptr_, hc_ptr_.data(), sizeof(int64_t) * hc_ptr_.size(), hipMemcpyHostToDevice));
    if (!hc_idx_.empty()) {
        HIP_TRY(hipMemcpy(d_cidx_, hc_idx_.data(), sizeof(int32_t) * hc_idx_.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_cval_, hc_val_.data(), sizeof(double) * hc_val_.size(), hipMemcpyHostToDevice));
    }
    return RELP_OK;
}

// Refactorisation (lower_upper/mod.rs:199-202 + carry/mod.rs:602-614): B from the current basis
// columns, P B Q = L U on the host, schedules to the device, W := empty.  Synchronises the stream.
relp_status_t Engine::lu_refactor() {
    std::vector<int32_t> basis(m_);
    HIP_TRY(hipMemcpyAsync(basis.data(), d_basis_, sizeof(int32_t) * m_, hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    std::vector<std::vector<std::pair<int32_t, double>>> cols(m_);
    for (int32_t i = 0; i < m_; ++i) {
        const int32_t j = basis[i];
        auto& c = cols[i];
        if (j < nr_artificial_) { c.emplace_back(column_to_row_[j], 1.0); continue; }
        if (j >= kWrappedArtificialBase) {                      // artificial that survived phase 1: still e_row
            c.emplace_back(column_to_row_[wrapped_na_ - 1 - (INT32_MAX - j)], 1.0);
            continue;
        }
        const int32_t p = j - nr_artificial_;
        if (p < nr_normal_) {
            for (int64_t e = hc_ptr_[p]; e < hc_ptr_[p + 1]; ++e) c.emplace_back(hc_idx_[e], hc_val_[e]);
            if (bound_row_h_[p] >= 0) c.emplace_back(bound_row_h_[p], 1.0);
        } else {
            const int32_t v = p - nr_normal_;
            if (v >= nr_virtual_) return fail(RELP_E_STATE, "basis column out of range");
            if (vrow0_h_[v] >= 0) c.emplace_back(vrow0_h_[v], (double)vsign_h_[v]);   // -1: its row was removed
            if (vrow1_h_[v] >= 0) c.emplace_back(vrow1_h_[v], 1.0);
        }
    }
    std::string msg;
    if (!lu_factor(m_, cols, &hlu_, &msg)) return fail(RELP_E_SINGULAR, msg);
    relp_status_t st = lu_upload_factors();
    if (st) return st;
    launch_flush_reset(deferred(), d_rec_, stream_);
    HIP_TRY(hipStreamSynchronize(stream_));
    since_flush_ = 0;
    ++lu_refactors_;
    return RELP_OK;
}

// hlu_ (host factors + level schedules) -> one device buffer, dlu_ points into it.  Also used by the revised
// engine's warm start, which forms the rows of B^-1 with the device BTRAN.
relp_status_t Engine::lu_upload_factors() {
    // pack everything into one buffer (16-byte aligned pieces): rowperm, colperm, then per schedule the
    // rows in solve order, the entry indices / values and the level offsets
    const TriangularSchedule* sch[4] = {&hlu_.Lf, &hlu_.Uf, &hlu_.Ub, &hlu_.Lb};
    std::vector<char> buf;
    auto put = [&](const void* src, size_t bytes) {
        const size_t o = buf.size();
        buf.resize(o + (bytes + 15) / 16 * 16);
        if (bytes) std::memcpy(buf.data() + o, src, bytes);
        return o;
    };
    const size_t o_rp = put(hlu_.rowperm.data(), sizeof(int32_t) * m_), o_cp = put(hlu_.colperm.data(), sizeof(int32_t) * m_);
    size_t o_rows[4], o_idx[4], o_val[4], o_lp[4];
    std::vector<LuRow> rows(m_);
    for (int k = 0; k < 4; ++k) {
        const TriangularSchedule& t = *sch[k];
        for (int32_t i = 0; i < m_; ++i) {
            const int32_t r = t.level_rows[i];
            rows[i] = LuRow{r, t.ptr[r], t.ptr[r + 1], 0, 1.0 / t.diag[r]};
        }
        o_rows[k] = put(rows.data(), sizeof(LuRow) * m_);
        o_idx[k] = put(t.idx.data(), sizeof(int32_t) * t.idx.size());
        o_val[k] = put(t.val.data(), sizeof(double) * t.val.size());
        o_lp[k] = put(t.level_ptr.data(), sizeof(int32_t) * t.level_ptr.size());
    }
    if ((int64_t)buf.size() > lu_cap_) {
        if (d_lu_buf_) HIP_TRY(hipFree(d_lu_buf_));
        lu_cap_ = (int64_t)buf.size() * 3 / 2 + 256;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_lu_buf_), (size_t)lu_cap_));
    }
    HIP_TRY(hipMemcpyAsync(d_lu_buf_, buf.data(), buf.size(), hipMemcpyHostToDevice, stream_));
    dlu_.m = m_; dlu_.pad_ = 0;
    dlu_.rowperm = reinterpret_cast<const int32_t*>(d_lu_buf_ + o_rp);
    dlu_.colperm = reinterpret_cast<const int32_t*>(d_lu_buf_ + o_cp);
    DeviceSchedule* ds[4] = {&dlu_.Lf, &dlu_.Uf, &dlu_.Ub, &dlu_.Lb};
    for (int k = 0; k < 4; ++k) {
        ds[k]->rows = reinterpret_cast<const LuRow*>(d_lu_buf_ + o_rows[k]);
        ds[k]->idx = reinterpret_cast<const int32_t*>(d_lu_buf_ + o_idx[k]);
        ds[k]->val = reinterpret_cast<const double*>(d_lu_buf_ + o_val[k]);
        ds[k]->level_ptr = reinterpret_cast<const int32_t*>(d_lu_buf_ + o_lp[k]);
        ds[k]->n_levels = (int32_t)sch[k]->level_ptr.size() - 1;
        ds[k]->nnz = (int32_t)sch[k]->idx.size();
    }
    HIP_TRY(hipStreamSynchronize(stream_));             // buf is stack-owned
    return RELP_OK;
}

relp_status_t Engine::lu_stats(int64_t* out8) const {
    if (!lu_) return RELP_E_STATE;
    out8[0] = lu_refactors_; out8[1] = hlu_.m; out8[2] = hlu_.nnz_l; out8[3] = hlu_.nnz_u;
    out8[4] = (int64_t)hlu_.Lf.level_ptr.size() - 1; out8[5] = (int64_t)hlu_.Uf.level_ptr.size() - 1;
    out8[6] = (int64_t)hlu_.Ub.level_ptr.size() - 1; out8[7] = (int64_t)hlu_.Lb.level_ptr.size() - 1;
    return RELP_OK;
}

// One pivot of the LU engine: CSC PRICE -> select + scatter a_q -> FTRAN (L, U solves) -> W correction
// -> ratio test -> W update -> BTRAN for the pivot row -> b, -pi, basis.
void Engine::enqueue_iteration_lu(int rule) {
    const ColumnTable ct = table();
    const DeferredUpdate du = deferred();
    SelectPartials sp;
    sp.k1 = d_part_k1_; sp.j = d_part_j_; sp.in_basis = d_in_basis_; sp.tol_cost = cfg_.tol_cost; sp.rule = rule;
    const int nb_struct = price_csc_blocks(0, nr_normal_);
    sp.n = nr_columns(); sp.offset = 0; sp.nb_struct = nb_struct; sp.tol_tie = cfg_.tol_tie; sp.p_lo = 0; sp.cols_per_slot = 256;
    const int nb_virt = price_virtual_blocks(ct);
    prof_begin(RELP_K_PRICE);
    if (nb_struct > 0 && nb_virt > 0) {
        launch_price_csc_all(csc(), ct, d_minus_pi_, d_d_, nr_normal_, phase_, sp, nb_virt, d_rec_, stream_);
    } else {
        launch_price_csc(csc(), ct, d_minus_pi_, d_d_, 0, nr_normal_, phase_, sp, d_rec_, stream_);
        SelectPartials spv = sp;
        spv.offset = nb_struct;
        launch_price_virtual_sel(ct, d_minus_pi_, d_d_, phase_, spv, d_rec_, stream_);
    }
    prof_end();
    prof_begin(RELP_K_SELECT_COLUMN);
    launch_select_partials_csc(sp, nb_struct + nb_virt, d_d_, csc(), ct, m_, d_aq_, d_rec_, stream_);
    prof_end();
    prof_begin(RELP_K_FTRAN);
    launch_lu_ftran(dlu_, d_aq_, d_v_, d_lu_scratch_, d_rec_, stream_);
    prof_end();
    prof_begin(RELP_K_APPLY_W);
    launch_apply_w_rmin(du, m_, d_v_, d_alpha_, d_b_, tolerances(), d_rmin_, d_rec_, stream_);
    prof_end();
    prof_begin(RELP_K_RATIO);
    launch_ratio_rows(d_alpha_, d_b_, d_basis_, m_, tolerances(), du, d_rmin_, 256, d_rec_, stream_);
    prof_end();
    prof_begin(RELP_K_UPDATE_W);
    launch_update_w(du, m_, d_alpha_, d_rec_, stream_);
    prof_end();
    prof_begin(RELP_K_UPDATE_VECTORS);
    launch_lu_btran(dlu_, du, nullptr, -1, d_rho_, d_lu_scratch_, d_rec_, stream_);
    launch_update_vectors(m_, d_alpha_, d_rho_, d_b_, d_minus_pi_, d_basis_, d_in_basis_, d_trace_, trace_cap_, d_rec_,
                          stream_);
    prof_end();
    if (++since_flush_ >= block_) enqueue_flush();
}

}  // namespace relp
