// relp_engine.cpp -- host driver: problem upload, phase logic, launch sequencing.  See relp_engine.hpp.
#include "relp_engine_internal.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace relp {

bool Engine::hip_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    err_ = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}

void Engine::free_all() {
    luf_release();
    auto fr = [](auto*& p) { if (p) { (void)hipFree(p); p = nullptr; } };
    if (owns_A_) fr(dA_);
    dA_ = nullptr;
    fr(dBinv_); fr(d_minus_pi_); fr(d_b_); fr(d_alpha_); fr(d_aq_); fr(d_rho_); fr(d_d_); fr(d_w_); fr(d_cost_);
    fr(d_basis_); fr(d_column_to_row_); fr(d_bound_row_); fr(d_vrow0_); fr(d_vrow1_); fr(d_vsign_); fr(d_trace_);
    fr(d_in_basis_); fr(d_rec_);
    fr(d_part_k1_); fr(d_part_j_);
    fr(dT0_); fr(dR0_); fr(d_b_alt_); fr(d_basis_alt_); fr(d_shadow_); fr(d_shadow_meta_);
    if (h_lu_buf_) { (void)hipHostFree(h_lu_buf_); h_lu_buf_ = nullptr; h_lu_cap_ = 0; }
    if (h_basis_) { (void)hipHostFree(h_basis_); h_basis_ = nullptr; }
    if (h_mirror_) { (void)hipHostFree(h_mirror_); h_mirror_ = nullptr; d_mirror_ = nullptr; }
    fr(d_cost_store_); fr(d_idcol_); fr(d_rmin_);
    fr(d_msg_cand_); fr(d_msg_cands_); fr(d_msg_slice_); fr(d_msg_slices_); fr(d_msg_rho_); fr(d_msg_status_); fr(d_msg_statuses_);
    fr(d_v_); fr(d_W_); fr(d_wr_); fr(d_R_); fr(d_S_); fr(d_pos_of_row_);
    fr(d_cptr_); fr(d_cidx_); fr(d_cval_); fr(d_lu_buf_); fr(d_lu_buf_alt_); fr(d_lu_scratch_); fr(d_ft_buf_); fr(d_pe_buf_);
    if (h_ft_hdr_) { (void)hipHostFree(h_ft_hdr_); h_ft_hdr_ = nullptr; }
    if (h_rec_) { (void)hipHostFree(h_rec_); h_rec_ = nullptr; }
    for (auto e : prof_ev_) (void)hipEventDestroy(e);
    prof_ev_.clear();
    if (owns_stream_ && stream_) { (void)hipStreamDestroy(stream_); }
    stream_ = nullptr;
}

Engine::~Engine() { rccl_release(); free_all(); }

ColumnTable Engine::table() const {
    ColumnTable ct;
    ct.nr_artificial = nr_artificial_;
    ct.nr_normal = nr_normal_;
    ct.nr_virtual = nr_virtual_;
    ct.nr_constraints = mc_;
    ct.column_to_row = d_column_to_row_;
    ct.bound_row = d_bound_row_;
    ct.vrow0 = d_vrow0_;
    ct.vrow1 = d_vrow1_;
    ct.vsign = d_vsign_;
    ct.cost = d_cost_;
    return ct;
}

Tolerances Engine::tolerances() const { return Tolerances{cfg_.tol_cost, cfg_.tol_pivot, cfg_.tol_zero, cfg_.tol_tie, cfg_.ratio_rule, (pivot_guard_on_ && guard_rel_ > 0.0) ? 1 : 0, guard_rel_}; }

TableauView Engine::tview() const {
    TableauView tv;
    // only the owned storage columns [sc_lo, sc_hi) are stored; shift so kernels index globally
    tv.T0 = dT0_ - (int64_t)sc_lo_ * ld_t_; tv.ld_t = ld_t_;
    tv.R0 = dR0_ - (int64_t)sc_lo_; tv.ld_r = ld_r_;
    tv.d = d_d_; tv.m = m_; tv.n_store = n_store_;
    tv.col_off = phase_ == 1 ? 0 : tab_na_;
    tv.n = nr_columns();
    tv.c_lo = sc_lo_; tv.c_hi = sc_hi_;
    return tv;
}

SelectPartials Engine::tab_partials(int rule) const {
    SelectPartials sp;
    sp.k1 = d_part_k1_; sp.j = d_part_j_; sp.in_basis = d_in_basis_; sp.tol_cost = cfg_.tol_cost; sp.rule = rule;
    sp.n = nr_columns(); sp.offset = 0; sp.nb_struct = 0; sp.tol_tie = cfg_.tol_tie; sp.p_lo = 0; sp.cols_per_slot = 8;
    return sp;
}

DeferredUpdate Engine::deferred() const {
    DeferredUpdate du;
    du.W = d_W_; du.ld = ld_b_; du.kmax = block_; du.S = d_S_; du.pos_of_row = d_pos_of_row_; du.wr = d_wr_; du.R = d_R_;
    return du;
}

relp_status_t Engine::download_rec() {
    HIP_TRY(hipMemcpyAsync(h_rec_, d_rec_, sizeof(PivotRecord), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    return RELP_OK;
}

relp_status_t Engine::upload_rec() {
    HIP_TRY(hipMemcpyAsync(d_rec_, h_rec_, sizeof(PivotRecord), hipMemcpyHostToDevice, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    return RELP_OK;
}

relp_status_t Engine::set_stream(hipStream_t s) {
    if (stream_) HIP_TRY(hipStreamSynchronize(stream_));
    if (owns_stream_ && stream_) HIP_TRY(hipStreamDestroy(stream_));
    stream_ = s;
    owns_stream_ = false;
    return RELP_OK;
}

// ------------------------------------------------------------------------------------------------
// Construction: MatrixData layout (matrix_data.rs:198-268, 308-371, 432-452) and the partially
// artificial start (partially.rs:125-206, carry/mod.rs:381-426)
// ------------------------------------------------------------------------------------------------
relp_status_t Engine::create(const relp_matrix_data_t& md, const relp_config_t& cfg) {
    (void)hipGetLastError();                               // (the launch check at the end must only see this create's launches)
    cfg_ = cfg;
    if (md.nr_normal < 0 || md.nr_eq < 0 || md.nr_range < 0 || md.nr_le < 0 || md.nr_ge < 0)
        return fail(RELP_E_ARG, "negative size");
    if (cfg_.shard_count < 1) cfg_.shard_count = 1;
    if (cfg_.shard_rank < 0 || cfg_.shard_rank >= cfg_.shard_count) return fail(RELP_E_ARG, "bad shard rank");
    if (cfg_.poll_interval < 1) cfg_.poll_interval = 64;
    if (cfg_.device >= 0) HIP_TRY(hipSetDevice(cfg_.device));

    nr_normal_ = md.nr_normal; nr_eq_ = md.nr_eq; nr_range_ = md.nr_range; nr_le_ = md.nr_le; nr_ge_ = md.nr_ge;
    mc_ = nr_eq_ + nr_range_ + nr_le_ + nr_ge_;
    if ((mc_ > 0 && !md.b) || (nr_normal_ > 0 && (!md.cost || !md.upper_bound)) || (nr_range_ > 0 && !md.ranges))
        return fail(RELP_E_ARG, "missing b / cost / upper_bound / ranges");

    cost_h_.assign(md.cost, md.cost + nr_normal_);
    upper_h_.assign(md.upper_bound, md.upper_bound + nr_normal_);
    bound_row_h_.assign(nr_normal_, -1);
    std::vector<int32_t> bound_to_var;
    for (int32_t j = 0; j < nr_normal_; ++j)
        if (std::isfinite(upper_h_[j])) { bound_row_h_[j] = mc_ + (int32_t)bound_to_var.size(); bound_to_var.push_back(j); }
    nr_bounds_ = (int32_t)bound_to_var.size();
    m_ = mc_ + nr_bounds_ + nr_range_;
    if (cfg_.engine == RELP_ENGINE_AUTO) {
        // INTEGRATION.md "which engine for which LP": the dense tableau while it fits comfortably, the LU engine beyond
        const double n_all = (double)nr_normal_ + nr_range_ + nr_le_ + nr_ge_ + nr_bounds_ + nr_range_ + m_;       // (+ m: identity / artificial block)
        const bool fits = 8.0 * (double)m_ * n_all <= 64e9 && m_ <= 50000;
        cfg_.engine = (cfg_.shard_count > 1 || fits) ? RELP_ENGINE_TABLEAU : RELP_ENGINE_LU;
    }
    if (m_ < 1) return fail(RELP_E_ARG, "empty problem");
    const int32_t row_start[7] = {0, nr_eq_, nr_eq_ + nr_range_, nr_eq_ + nr_range_ + nr_le_, mc_, mc_ + nr_bounds_, m_};
    // virtual columns in provider order: range slack | <= slack | >= slack | bound slack | range-bound slack
    nr_virtual_ = nr_range_ + nr_le_ + nr_ge_ + nr_bounds_ + nr_range_;
    n_provider_ = nr_normal_ + nr_virtual_;
    vrow0_h_.clear(); vrow1_h_.clear(); vsign_h_.clear();
    for (int32_t k = 0; k < nr_range_; ++k) { vrow0_h_.push_back(row_start[1] + k); vrow1_h_.push_back(row_start[5] + k); vsign_h_.push_back(1); }
    for (int32_t k = 0; k < nr_le_; ++k) { vrow0_h_.push_back(row_start[2] + k); vrow1_h_.push_back(-1); vsign_h_.push_back(1); }
    for (int32_t k = 0; k < nr_ge_; ++k) { vrow0_h_.push_back(row_start[3] + k); vrow1_h_.push_back(-1); vsign_h_.push_back(-1); }
    for (int32_t k = 0; k < nr_bounds_; ++k) { vrow0_h_.push_back(row_start[4] + k); vrow1_h_.push_back(-1); vsign_h_.push_back(1); }
    for (int32_t k = 0; k < nr_range_; ++k) { vrow0_h_.push_back(row_start[5] + k); vrow1_h_.push_back(-1); vsign_h_.push_back(1); }
    // right_hand_side = (b, upper bounds, ranges), matrix_data.rs:359-371
    rhs_h_.assign(m_, 0.0);
    for (int32_t i = 0; i < mc_; ++i) rhs_h_[i] = md.b[i];
    for (int32_t k = 0; k < nr_bounds_; ++k) rhs_h_[mc_ + k] = upper_h_[bound_to_var[k]];
    for (int32_t k = 0; k < nr_range_; ++k) rhs_h_[mc_ + nr_bounds_ + k] = md.ranges[k];

    // shards: structural columns and rows of B^-1
    {
        const int32_t G = cfg_.shard_count, g = cfg_.shard_rank;
        const int32_t cper = (nr_normal_ + G - 1) / G;
        col_lo_ = std::min(nr_normal_, g * cper);
        col_hi_ = std::min(nr_normal_, col_lo_ + cper);
        row_stride_ = (int32_t)round_up((m_ + G - 1) / G, 2);
        row_lo_ = std::min(m_, g * row_stride_);
        row_hi_ = std::min(m_, row_lo_ + row_stride_);
        cand_len_ = candidate_len_for(m_);
    }
    const bool want_tableau_early = cfg_.engine == RELP_ENGINE_TABLEAU;

    // initial basis: <=-slacks, bound slacks, range-bound slacks are real pivots (matrix_data.rs:432-452);
    // every other row gets an artificial, numbered before all provider columns (partially.rs:72-80)
    std::vector<int32_t> real_row, real_col;
    const int32_t col_start2 = nr_normal_ + nr_range_;                    // <= slacks
    const int32_t col_start4 = nr_normal_ + nr_range_ + nr_le_ + nr_ge_;  // bound slacks
    const int32_t col_start5 = col_start4 + nr_bounds_;                   // range-bound slacks
    for (int32_t k = 0; k < nr_le_; ++k) { real_row.push_back(row_start[2] + k); real_col.push_back(col_start2 + k); }
    for (int32_t k = 0; k < nr_bounds_; ++k) { real_row.push_back(row_start[4] + k); real_col.push_back(col_start4 + k); }
    for (int32_t k = 0; k < nr_range_; ++k) { real_row.push_back(row_start[5] + k); real_col.push_back(col_start5 + k); }
    const int32_t nr_real = (int32_t)real_row.size();
    nr_artificial_ = m_ - nr_real;
    if (cfg_.shard_count > 1 && nr_artificial_ > 0 && !want_tableau_early)
        return fail(RELP_E_UNSUPPORTED, "the sharded revised engine needs a full slack basis (no artificial variables); "
                                        "the sharded tableau engine runs both phases");
    column_to_row_.assign(nr_artificial_, 0);
    {
        int32_t i = 0;
        for (int32_t ith = 0; ith < nr_artificial_; ++ith) {
            while (i < nr_real && ith + i == real_row[i]) ++i;
            column_to_row_[ith] = ith + i;
        }
    }
    std::vector<int32_t> basis(m_);
    {
        int32_t ac = 0;
        for (int32_t row = 0; row < m_; ++row) {
            const bool can_a = ac < nr_artificial_, can_r = (row - ac) < nr_real;
            if (can_a && can_r) {
                if (column_to_row_[ac] < real_row[row - ac]) basis[row] = ac++;
                else basis[row] = nr_artificial_ + real_col[row - ac];
            } else if (can_a) basis[row] = ac++;
            else basis[row] = nr_artificial_ + real_col[row - ac];
        }
    }
    phase_ = 1;
    n_alloc_ = nr_artificial_ + n_provider_;
    if (want_tableau_early) {
        // the tableau shards its STORED columns [artificial | structural | virtual] contiguously; the
        // structural part of the owned range is what the caller supplies in `dense`
        const int32_t G = cfg_.shard_count, g = cfg_.shard_rank;
        const int32_t per = (int32_t)round_up((n_alloc_ + G - 1) / G, 2);
        sc_lo_ = std::min(n_alloc_, g * per);
        sc_hi_ = std::min(n_alloc_, sc_lo_ + per);
        col_lo_ = std::min(nr_normal_, std::max(0, sc_lo_ - nr_artificial_));
        col_hi_ = std::min(nr_normal_, std::max(0, sc_hi_ - nr_artificial_));
        if (col_hi_ < col_lo_) col_hi_ = col_lo_;
    }

    // ---- device allocations ----
    HIP_TRY(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    owns_stream_ = true;
    const int32_t n_local = col_hi_ - col_lo_;
    lu_ = cfg_.engine == RELP_ENGINE_LU;
    if (lu_) {
        if (cfg_.shard_count > 1) return fail(RELP_E_UNSUPPORTED, "the LU engine is not sharded");
        relp_status_t lst = lu_load_matrix(md);
        if (lst) return lst;
    } else if (md.format == RELP_FORMAT_DENSE) {
        if (nr_normal_ > 0 && mc_ > 0 && !md.dense) return fail(RELP_E_ARG, "dense matrix missing");
        const int64_t src_ld = md.dense_ld > 0 ? md.dense_ld : mc_;
        if (src_ld < mc_) return fail(RELP_E_ARG, "dense_ld < nr_constraints");
        // In sharded mode `dense` holds only the owned columns [col_lo, col_hi).
        if (md.matrix_memory == RELP_MEM_DEVICE && (src_ld % 2) == 0) {
            dA_ = const_cast<double*>(md.dense); ld_a_ = src_ld; owns_A_ = false;          // zero-copy adoption
        } else {
            ld_a_ = round_up(std::max<int64_t>(mc_, 1), 2);
            HIP_TRY(dev_alloc(&dA_, ld_a_ * std::max(n_local, 1)));
            owns_A_ = true;
            if (n_local > 0 && mc_ > 0)
                HIP_TRY(hipMemcpy2D(dA_, ld_a_ * sizeof(double), md.dense, src_ld * sizeof(double), mc_ * sizeof(double),
                                    n_local, md.matrix_memory == RELP_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
        }
    } else if (md.format == RELP_FORMAT_CSC) {
        if (md.matrix_memory != RELP_MEM_HOST) return fail(RELP_E_UNSUPPORTED, "CSC input must be in host memory");
        if (!md.col_ptr) return fail(RELP_E_ARG, "col_ptr missing");
        ld_a_ = round_up(std::max<int64_t>(mc_, 1), 2);
        // the dense engines' copy of the owned columns, scattered on the device from the CSC arrays (never staged dense on the host)
        const int64_t e0 = n_local > 0 ? md.col_ptr[col_lo_] : 0, e1 = n_local > 0 ? md.col_ptr[col_hi_] : 0;
        for (int64_t p = e0; p < e1; ++p)
            if (md.row_idx[p] < 0 || md.row_idx[p] >= mc_) return fail(RELP_E_ARG, "row index out of range");
        const int64_t cells = ld_a_ * std::max(n_local, 1);
        HIP_TRY(dev_alloc(&dA_, cells));
        owns_A_ = true;
        HIP_TRY(hipMemset(dA_, 0, (size_t)cells * sizeof(double)));
        if (e1 > e0) {
            int64_t* t_ptr = nullptr; int32_t* t_idx = nullptr; double* t_val = nullptr;
            auto drop = [&]() { if (t_ptr) (void)hipFree(t_ptr); if (t_idx) (void)hipFree(t_idx); if (t_val) (void)hipFree(t_val); };
            if (hipMalloc(reinterpret_cast<void**>(&t_ptr), sizeof(int64_t) * (size_t)(n_local + 1)) != hipSuccess ||
                hipMalloc(reinterpret_cast<void**>(&t_idx), sizeof(int32_t) * (size_t)(e1 - e0)) != hipSuccess ||
                hipMalloc(reinterpret_cast<void**>(&t_val), sizeof(double) * (size_t)(e1 - e0)) != hipSuccess) {
                drop();
                return fail(RELP_E_ALLOC, "staging the CSC arrays");
            }
            hipError_t err = hipMemcpy(t_ptr, md.col_ptr + col_lo_, sizeof(int64_t) * (size_t)(n_local + 1), hipMemcpyHostToDevice);
            if (err == hipSuccess) err = hipMemcpy(t_idx, md.row_idx + e0, sizeof(int32_t) * (size_t)(e1 - e0), hipMemcpyHostToDevice);
            if (err == hipSuccess) err = hipMemcpy(t_val, md.values + e0, sizeof(double) * (size_t)(e1 - e0), hipMemcpyHostToDevice);
            if (err == hipSuccess) {
                launch_csc_to_dense(t_ptr, t_idx, t_val, n_local, dA_, ld_a_, nullptr);
                err = hipDeviceSynchronize();
            }
            drop();
            if (err != hipSuccess) return fail(RELP_E_HIP, "dense copy of the CSC input");
        }
    } else {
        return fail(RELP_E_ARG, "unknown matrix format");
    }

    ld_b_ = round_up(m_, 16);
    const int64_t rows_local = std::max(row_hi_ - row_lo_, 1);
    const bool want_tableau = cfg_.engine == RELP_ENGINE_TABLEAU;
    // the tableau engine reads B^-1 off the identity columns of T; the explicit inverse is not stored
    HIP_TRY(dev_alloc(&dBinv_, (want_tableau || lu_) ? 16 : rows_local * ld_b_));
    HIP_TRY(dev_alloc(&d_minus_pi_, ld_b_));
    HIP_TRY(dev_alloc(&d_b_, ld_b_));
    HIP_TRY(dev_alloc(&d_alpha_, ld_b_));
    HIP_TRY(dev_alloc(&d_aq_, ld_b_));
    HIP_TRY(dev_alloc(&d_rho_, ld_b_));
    HIP_TRY(dev_alloc(&d_w_, ld_b_));
    HIP_TRY(dev_alloc(&d_d_, n_alloc_));
    HIP_TRY(dev_alloc(&d_cost_, nr_normal_));
    HIP_TRY(dev_alloc(&d_basis_, m_));
    HIP_TRY(dev_alloc(&d_column_to_row_, nr_artificial_));
    HIP_TRY(dev_alloc(&d_bound_row_, nr_normal_));
    HIP_TRY(dev_alloc(&d_vrow0_, nr_virtual_));
    HIP_TRY(dev_alloc(&d_vrow1_, nr_virtual_));
    HIP_TRY(dev_alloc(&d_vsign_, nr_virtual_));
    HIP_TRY(dev_alloc(&d_in_basis_, n_alloc_));
    HIP_TRY(dev_alloc(&d_rec_, 1));
    {
        const int64_t slots = price_structural_blocks(col_lo_, col_hi_) + (nr_artificial_ + nr_virtual_ + 255) / 256 + 8 +
                              tab_scan_blocks(n_alloc_) + price_csc_blocks(0, nr_normal_);
        HIP_TRY(dev_alloc(&d_part_k1_, slots));
        HIP_TRY(dev_alloc(&d_part_j_, slots));
    }
    block_ = cfg_.update_block < 0 ? (m_ >= 4096 ? 64 : 0) : std::min(cfg_.update_block, 128);
    tableau_ = cfg_.engine == RELP_ENGINE_TABLEAU;
    if (lu_) {
        // pivots between refactorisations (the reference refactors after 10 updates, lower_upper/mod.rs:199;
        // here an update is one column of W, so longer blocks are cheap)
        block_ = cfg_.update_block < 0 ? 128 : std::max(1, std::min(cfg_.update_block, 128));
        HIP_TRY(dev_alloc(&d_lu_scratch_, ld_b_));
        {   // environment switches of the LU engine, read once per engine
            const char* la = std::getenv("RELP_LU_LOOKAHEAD");
            const char* fl = std::getenv("RELP_FUSE_LANES");
            const char* df = std::getenv("RELP_LU_DEVICE_FACTOR");
            luf_enabled_ = df && std::atoi(df) != 0;
            luf_download_ = df && std::atoi(df) == 2;
            lu_lookahead_env_ = la ? std::atoi(la) : 8;
            lu_lookahead_set_ = la != nullptr;
            lu_fuse_lanes_env_ = fl ? std::atoi(fl) : 256;
        }
        relp_status_t fst = ft_plan_and_alloc();           // Forrest-Tomlin on the device when the LDS budget allows
        if (fst) return fst;
    }
    if (tableau_) {
        if (block_ == 0) block_ = 64;                  // the tableau is always maintained in blocks
        // automatic choice for wide tableaus: 96 pivots per flush where the flush dominates the pivot.  Measured: 10,000 x
        // 50,000 (60,000 stored columns) 19,600 it/s against 18,600 with 64; at 10,000 x 10,000 (20,000 stored columns) the gain is
        // within the noise of the windows (+0.8 %) while the flush kernel leaves its best operating point (0.58 instead of 0.61
        // of HBM peak: more arithmetic per byte), so 64 stays there.
        if (cfg_.update_block < 0 && m_ >= 4096 && n_alloc_ >= 40000) block_ = 96;
        n_store_ = n_alloc_;
        tab_na_ = nr_artificial_;
        const int64_t n_owned = std::max(sc_hi_ - sc_lo_, 1);
        ld_t_ = round_up(m_, 2);
        ld_r_ = round_up(n_owned, 2);
        HIP_TRY(dev_alloc(&dT0_, ld_t_ * n_owned));
        HIP_TRY(dev_alloc(&dR0_, ld_r_ * (block_ + 1)));        // + one scratch row (d_aq_big)
        {   // two launches per pivot instead of three in the single-GPU loop (RELP_FUSED_UPDATE=0: k_ratio_blocks + k_tab_update_all)
            const char* e = std::getenv("RELP_FUSED_UPDATE");
            fused_update_ = !(e && std::atoi(e) == 0);      // (also the native sharded loop, relp_shard_run)
            if (cfg_.pivot_rescue && cfg_.shard_count == 1) fused_update_ = false;       // (the pivot guard lives in the shared ratio epilogue)
            if (fused_update_) {
                HIP_TRY(dev_alloc(&d_b_alt_, ld_b_));
                HIP_TRY(dev_alloc(&d_basis_alt_, m_));
                HIP_TRY(dev_alloc(&d_shadow_, std::max(block_, 1) + 1));
                HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_shadow_meta_), 2 * sizeof(int32_t)));
                const int32_t none[2] = {-1, 0};
                HIP_TRY(hipMemcpy(d_shadow_meta_, none, sizeof none, hipMemcpyHostToDevice));
            }
        }
        HIP_TRY(dev_alloc(&d_cost_store_, n_store_));
        HIP_TRY(dev_alloc(&d_idcol_, m_));
    }
    if (block_ > 0 && ft_) {
        HIP_TRY(dev_alloc(&d_v_, ld_b_));                  // (no W: the update file lives in FtState)
    } else if (block_ > 0) {
        HIP_TRY(dev_alloc(&d_v_, ld_b_));
        HIP_TRY(dev_alloc(&d_W_, ld_b_ * block_));
        if (!tableau_) HIP_TRY(dev_alloc(&d_R_, ld_b_ * block_));
        HIP_TRY(dev_alloc(&d_wr_, block_));
        HIP_TRY(dev_alloc(&d_S_, block_));
        HIP_TRY(dev_alloc(&d_pos_of_row_, m_));
        HIP_TRY(hipMemset(d_pos_of_row_, 0xFF, sizeof(int32_t) * m_));      // -1 everywhere
    }
    HIP_TRY(dev_alloc(&d_rmin_, m_ / 8 + 2));          // block minima of the ratio test (8 or 256 rows per block)
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h_rec_), sizeof(PivotRecord), hipHostMallocDefault));
    trace_cap_ = std::max(cfg_.trace_capacity, 0);
    if (trace_cap_ > 0) HIP_TRY(dev_alloc(&d_trace_, 4 * trace_cap_));

    auto up = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
        return bytes ? hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
    };
    HIP_TRY(up(d_cost_, cost_h_.data(), sizeof(double) * nr_normal_));
    HIP_TRY(up(d_bound_row_, bound_row_h_.data(), sizeof(int32_t) * nr_normal_));
    HIP_TRY(up(d_vrow0_, vrow0_h_.data(), sizeof(int32_t) * nr_virtual_));
    HIP_TRY(up(d_vrow1_, vrow1_h_.data(), sizeof(int32_t) * nr_virtual_));
    HIP_TRY(up(d_vsign_, vsign_h_.data(), sizeof(int32_t) * nr_virtual_));
    HIP_TRY(up(d_column_to_row_, column_to_row_.data(), sizeof(int32_t) * nr_artificial_));
    HIP_TRY(up(d_basis_, basis.data(), sizeof(int32_t) * m_));
    HIP_TRY(up(d_b_, rhs_h_.data(), sizeof(double) * m_));
    // Carry::create_for_partially_artificial, carry/mod.rs:381-426
    std::vector<double> minus_pi(m_, 0.0);
    double objective = 0.0;
    for (int32_t k = 0; k < nr_artificial_; ++k) { objective += rhs_h_[column_to_row_[k]]; minus_pi[column_to_row_[k]] = -1.0; }
    initial_phase1_objective_ = objective;
    HIP_TRY(up(d_minus_pi_, minus_pi.data(), sizeof(double) * m_));
    std::vector<uint8_t> flags(n_alloc_, 0);
    for (int32_t r = 0; r < m_; ++r) flags[basis[r]] = 1;
    HIP_TRY(up(d_in_basis_, flags.data(), flags.size()));
    // identity rows [row_lo, row_hi): local row i has its 1 in column row_lo + i (BasisInverse::identity)
    if (row_hi_ > row_lo_ && !tableau_ && !lu_) launch_set_identity(dBinv_, ld_b_, row_lo_, row_hi_, stream_);
    if (tableau_) {
        // T0 = the original matrix in row space (B = I), d = c - c_B' T0 with the phase-1 costs
        idcol_h_ = basis;                                    // the initial basis column of row k is e_k
        HIP_TRY(up(d_idcol_, idcol_h_.data(), sizeof(int32_t) * m_));
        const double* A = dA_ - (int64_t)col_lo_ * ld_a_;
        launch_tab_build(tview(), A, ld_a_, table(), stream_);
        cost_store_h_.assign(n_store_, 0.0);
        for (int32_t k = 0; k < nr_artificial_; ++k) cost_store_h_[k] = 1.0;
        std::vector<double> w(ld_b_, 0.0);
        for (int32_t k = 0; k < nr_artificial_; ++k) w[column_to_row_[k]] = 1.0;
        HIP_TRY(up(d_cost_store_, cost_store_h_.data(), sizeof(double) * n_store_));
        HIP_TRY(up(d_w_, w.data(), sizeof(double) * ld_b_));
        launch_tab_price_init(tview(), d_w_, d_cost_store_, stream_);
    }
    // f64 only: the explicit inverse / the tableau is updated thousands of times on long solves; for sparse problems
    // below 4,097 rows (where the host factorisation of a basis is cheap: CSC input with at most 10 % nonzeros) it is
    // rebuilt from the basis columns every 1,000 pivots (relp_set_reinversion_interval changes or enables it)
    {
        bool sparse_input = false;
        if (md.format == RELP_FORMAT_CSC && md.col_ptr && nr_normal_ > 0 && mc_ > 0)
            sparse_input = (double)md.col_ptr[nr_normal_] <= 0.10 * (double)nr_normal_ * (double)mc_;
        reinvert_interval_ = (!lu_ && cfg_.shard_count == 1 && m_ <= 4096 && sparse_input) ? 1000 : 0;
        // relp_config_t.auto_reinversion: any LP, starting at 256 pivots; every rebuild measures what it corrected and adapts
        if (cfg_.auto_reinversion && !lu_ && cfg_.shard_count == 1) reinvert_interval_ = 256;
    }
    std::memset(h_rec_, 0, sizeof(PivotRecord));
    h_rec_->outcome = DEV_RUNNING;
    h_rec_->minus_objective = -objective;
    h_rec_->last_selected = -1;
    h_rec_->phase = 1;
    relp_status_t st = upload_rec();
    if (st) return st;
    HIP_TRY(hipDeviceSynchronize());   // hipMemset on the null stream vs. our non-blocking stream
    // (a refused launch -- e.g. more than 2^32 - 1 threads -- returns nothing by itself: without this the engine would start
    // from a tableau that was never built)
    if (hipGetLastError() != hipSuccess) return fail(RELP_E_HIP, "a kernel launch of the set-up failed");
    if (lu_ && (st = lu_refactor())) return st;
    return RELP_OK;
}

// ------------------------------------------------------------------------------------------------
// Profiling
// ------------------------------------------------------------------------------------------------
relp_status_t Engine::profile_enable(bool enable, int64_t max_launches, int32_t sample_every) {
    prof_stride_ = sample_every > 0 ? sample_every : 1;
    prof_tick_ = 0;
    HIP_TRY(hipStreamSynchronize(stream_));
    for (auto e : prof_ev_) (void)hipEventDestroy(e);
    prof_ev_.clear(); prof_kid_.clear(); prof_open_ = false;
    prof_on_ = enable;
    if (enable) {
        prof_ev_.resize((size_t)max_launches * 2);
        for (auto& e : prof_ev_) HIP_TRY(hipEventCreate(&e));
        prof_kid_.reserve((size_t)max_launches);
    }
    return RELP_OK;
}

void Engine::prof_begin(int kid, hipStream_t on) {
    prof_open_ = false;
    if (!prof_on_ || (prof_kid_.size() + 1) * 2 > prof_ev_.size()) return;
    if (kid != RELP_K_FLUSH && (prof_tick_ % prof_stride_) != 0) return;   // sampled pivots only
    (void)hipEventRecord(prof_ev_[2 * prof_kid_.size()], on ? on : stream_);
    prof_kid_.push_back(kid);
    prof_open_ = true;
}

void Engine::prof_end(hipStream_t on) {
    if (!prof_open_) return;
    (void)hipEventRecord(prof_ev_[2 * (prof_kid_.size() - 1) + 1], on ? on : stream_);
    prof_open_ = false;
}

relp_status_t Engine::profile_read(int kernel_id, int64_t* launches, double* total_ms) {
    HIP_TRY(hipStreamSynchronize(stream_));
    int64_t n = 0; double ms = 0.0;
    for (size_t k = 0; k < prof_kid_.size(); ++k) {
        if (prof_kid_[k] != kernel_id) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, prof_ev_[2 * k], prof_ev_[2 * k + 1]) == hipSuccess) { ms += t; ++n; }
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = ms;
    return RELP_OK;
}

// ------------------------------------------------------------------------------------------------
// One pivot on the device
// ------------------------------------------------------------------------------------------------
// PRICE over the owned structural columns and every virtual column with vector `vec` (= -pi).
void Engine::enqueue_price(int cost_mode, const double* vec, const PivotRecord* rec, int32_t p_lo, int32_t p_hi) {
    const ColumnTable ct = table();
    if (lu_) {
        launch_price_csc(csc(), ct, vec, d_d_, 0, nr_normal_, cost_mode, SelectPartials{}, rec, stream_);
        launch_price_virtual(ct, vec, d_d_, cost_mode, rec, stream_);
        return;
    }
    // dA_ holds the owned columns only: shift the base so that global column p indexes correctly
    const double* A = dA_ - (int64_t)col_lo_ * ld_a_;
    launch_price_structural(A, ld_a_, ct, vec, d_d_, p_lo, p_hi, cost_mode, rec, stream_);
    if (p_lo > 0 || p_hi < nr_normal_) launch_price_mask_unowned(ct, d_d_, p_lo, p_hi, rec, stream_);
    launch_price_virtual(ct, vec, d_d_, cost_mode, rec, stream_);
}

// One pivot of the dense-tableau engine: 5 launches, O(K (m + n)) bytes.
void Engine::enqueue_iteration_tableau(int rule) {
    const TableauView tv = tview();
    const DeferredUpdate du = deferred();
    const SelectPartials sp = tab_partials(rule);
    // 3 launches: [PRICE's final reduction + tableau column] -> [ratio test + block bookkeeping] ->
    // [tableau row / reduced costs / next PRICE partials  ||  W, b, basis]
    if (fused_update_ && in_loop_) {
        // 2 launches: [PRICE's final reduction + tableau column + block minima of the ratios] -> [ratio test in every
        // workgroup + tableau row / reduced costs / next PRICE partials || W, b, basis]
        prof_begin(RELP_K_FTRAN);
        launch_tab_select_column_rmin(tv, du, sp, tab_scan_blocks(sc_hi_ - sc_lo_), d_alpha_, d_b_, tolerances(), d_rmin_,
                                      d_rec_, stream_, d_shadow_, d_shadow_meta_);
        prof_end();
        prof_begin(RELP_K_PRICE);
        launch_tab_ratio_update_all(tv, du, sp, m_, d_alpha_, d_b_, d_b_alt_, d_basis_, d_basis_alt_, d_in_basis_, d_trace_,
                                    trace_cap_, tolerances(), d_rmin_, d_shadow_, d_shadow_meta_, d_rec_, stream_);
        prof_end();
        std::swap(d_b_, d_b_alt_);
        std::swap(d_basis_, d_basis_alt_);
        shadow_pending_ = true;
        if (++since_flush_ >= block_) enqueue_flush();
        return;
    }
    tab_settle();
    prof_begin(RELP_K_FTRAN);
    launch_tab_select_column_rmin(tv, du, sp, tab_scan_blocks(sc_hi_ - sc_lo_), d_alpha_, d_b_, tolerances(), d_rmin_,
                                  d_rec_, stream_);
    prof_end();
    prof_begin(RELP_K_RATIO);
    launch_ratio_blocks(d_alpha_, d_b_, d_basis_, m_, tolerances(), du, d_rmin_, d_rec_, stream_);
    prof_end();
    prof_begin(RELP_K_PRICE);
    launch_tab_update_all(tv, du, sp, m_, d_alpha_, d_b_, d_basis_, d_in_basis_, d_trace_, trace_cap_, d_rec_, stream_);
    prof_end();
    if (++since_flush_ >= block_) enqueue_flush();
}

void Engine::enqueue_iteration(int rule) {
    struct Tick { int64_t& t; ~Tick() { ++t; } } tick{prof_tick_};
    if (tableau_) { enqueue_iteration_tableau(rule); return; }
    if (lu_) { enqueue_iteration_lu(rule); return; }
    const ColumnTable ct = table();
    const double* A = dA_ - (int64_t)col_lo_ * ld_a_;
    double* Binv = dBinv_ - (int64_t)row_lo_ * ld_b_;
    const int n = nr_columns();
    // PRICE with the partial argmin fused in, then one single-workgroup launch that picks the
    // entering column and builds it in row space
    SelectPartials sp;
    sp.k1 = d_part_k1_; sp.j = d_part_j_; sp.in_basis = d_in_basis_; sp.tol_cost = cfg_.tol_cost; sp.rule = rule;
    const int nb_struct = price_structural_blocks(col_lo_, col_hi_);
    sp.n = n; sp.offset = 0; sp.nb_struct = nb_struct; sp.tol_tie = cfg_.tol_tie; sp.p_lo = col_lo_; sp.cols_per_slot = 8;
    const int nb_virt = price_virtual_blocks(ct);
    prof_begin(RELP_K_PRICE);
    if (col_lo_ > 0 || col_hi_ < nr_normal_) {
        launch_price_structural_sel(A, ld_a_, ct, d_minus_pi_, d_d_, col_lo_, col_hi_, phase_, sp, d_rec_, stream_);
        launch_price_mask_unowned(ct, d_d_, col_lo_, col_hi_, d_rec_, stream_);
        SelectPartials spv = sp;
        spv.offset = nb_struct;
        launch_price_virtual_sel(ct, d_minus_pi_, d_d_, phase_, spv, d_rec_, stream_);
    } else {
        launch_price_all_sel(A, ld_a_, ct, d_minus_pi_, d_d_, col_lo_, col_hi_, phase_, sp, d_rec_, stream_);
    }
    prof_end();
    prof_begin(RELP_K_SELECT_COLUMN);
    launch_select_partials(sp, nb_struct + nb_virt, d_d_, A, ld_a_, ct, m_, d_aq_, d_rec_, stream_);
    prof_end();
    if (block_ == 0) {
        // explicit inverse, rank-1 update at every pivot (basis_inverse_rows.rs:131-142)
        // FTRAN leaves the minimum ratio of every 8 rows behind; the ratio test starts from those
        prof_begin(RELP_K_FTRAN);
        launch_ftran_rmin(Binv, ld_b_, m_, d_aq_, d_alpha_, d_b_, tolerances(), d_rmin_, d_rec_, stream_);
        prof_end();
        prof_begin(RELP_K_RATIO);
        launch_ratio_rows(d_alpha_, d_b_, d_basis_, m_, tolerances(), DeferredUpdate{}, d_rmin_, ftran_rows_per_block(), d_rec_,
                          stream_);
        prof_end();
        prof_begin(RELP_K_UPDATE_VECTORS);
        launch_compute_rho(Binv, ld_b_, m_, row_lo_, row_hi_, d_rho_, d_rec_, stream_);
        prof_end();
        // rank-1 update of B^-1 together with b, -pi, -obj, basis, flags, trace
        prof_begin(RELP_K_UPDATE_INVERSE);
        launch_update_inverse_vectors(Binv, ld_b_, m_, d_alpha_, d_rho_, d_b_, d_minus_pi_, d_basis_, d_in_basis_, d_trace_,
                                      trace_cap_, d_rec_, stream_);
        prof_end();
        return;
    }
    // deferred update: B^-1 = (I + W S') B0inv
    const DeferredUpdate du = deferred();
    prof_begin(RELP_K_FTRAN);
    launch_ftran(Binv, ld_b_, m_, row_lo_, row_hi_, d_aq_, d_v_, 0, d_rec_, stream_);
    prof_end();
    prof_begin(RELP_K_APPLY_W);
    launch_apply_w_rmin(du, m_, d_v_, d_alpha_, d_b_, tolerances(), d_rmin_, d_rec_, stream_);
    prof_end();
    prof_begin(RELP_K_RATIO);
    launch_ratio_rows(d_alpha_, d_b_, d_basis_, m_, tolerances(), du, d_rmin_, 256, d_rec_, stream_);
    prof_end();
    prof_begin(RELP_K_UPDATE_W);
    launch_update_w(du, m_, d_alpha_, d_rec_, stream_);
    launch_rho_deferred(du, Binv, ld_b_, m_, row_lo_, row_hi_, d_rho_, d_rec_, stream_);
    prof_end();
    prof_begin(RELP_K_UPDATE_VECTORS);
    launch_update_vectors(m_, d_alpha_, d_rho_, d_b_, d_minus_pi_, d_basis_, d_in_basis_, d_trace_, trace_cap_, d_rec_,
                          stream_);
    prof_end();
    if (++since_flush_ >= block_) enqueue_flush();
}

void Engine::tab_settle() {
    if (shadow_pending_) {                             // the fused update's new row r of W is still in its shadow row
        launch_tab_apply_shadow(deferred(), d_shadow_, d_shadow_meta_, stream_);
        shadow_pending_ = false;
    }
}

// Fold the pending pivots into the explicit inverse: B0inv += W (S' B0inv).  Valid in any state
// (also after the loop froze): (B0inv, W, S) is consistent after every completed pivot.
void Engine::enqueue_flush() {
    if (block_ == 0) return;
    if (lu_) {
        if (since_flush_ == 0 && !ft_need_refactor_) return;   // the factors already describe the current basis
        prof_begin(RELP_K_FLUSH);
        const relp_status_t st = lu_refactor();
        prof_end();
        if (st && !lu_status_) lu_status_ = st;
        return;
    }
    if (tableau_) {
        tab_settle();
        // T0 += W R0 on the f64 matrix cores
        const DeferredUpdate dut = deferred();
        prof_begin(RELP_K_FLUSH);
        launch_tab_flush(tview(), dut, d_rec_, stream_);
        launch_flush_reset(dut, d_rec_, stream_);
        prof_end();
        since_flush_ = 0;
        // The reduced costs are only ever updated (d -= theta * row); every few flushes they are recomputed
        // from the flushed tableau, d = c - c_B' T0, so that rounding does not pile up over thousands of
        // pivots (one extra pass over T0 per kRepriceEveryFlushes * K pivots).
        if (++flushes_since_reprice_ >= kRepriceEveryFlushes) {
            flushes_since_reprice_ = 0;
            const TableauView tv = tview();
            const int rule = phase_ == 1 ? cfg_.phase_one_rule : cfg_.phase_two_rule;
            launch_tab_basis_costs(tv, d_basis_, d_cost_store_, d_w_, stream_);
            launch_tab_price_init(tv, d_w_, d_cost_store_, stream_);
            launch_tab_scan(tv, tab_partials(rule), d_rec_, stream_);
        }
        return;
    }
    const DeferredUpdate du = deferred();
    double* Binv = dBinv_ - (int64_t)row_lo_ * ld_b_;
    prof_begin(RELP_K_FLUSH);
    launch_flush_snapshot(du, Binv, ld_b_, row_lo_, row_hi_, d_rec_, stream_);
    launch_flush_apply(du, Binv, ld_b_, m_, row_lo_, row_hi_, d_rec_, stream_);
    launch_flush_reset(du, d_rec_, stream_);
    prof_end();
    since_flush_ = 0;
}

relp_status_t Engine::flush() {
    tab_settle();
    if (cfg_.shard_count > 1 && block_ > 0 && !tableau_) {
        // the rows S' B0inv live on different ranks: with the collective hooks attached the library completes the
        // snapshot itself (every rank must call), otherwise the caller drives relp_shard_flush_begin / _end
        if (!coll_allreduce_) return fail(RELP_E_STATE, "sharded flush needs relp_shard_flush_begin/end (or the collective hooks)");
        if (since_flush_ == 0) return RELP_OK;
        double* snap = nullptr; int64_t len = 0;
        relp_status_t st = shard_flush_begin(&snap, &len);
        if (st) return st;
        if (len > 0) {
            if (coll_allreduce_(coll_ctx_, snap, len, stream_)) return fail(RELP_E_HIP, "all-reduce of the flush snapshot failed");
            if ((st = shard_flush_end())) return st;
        }
        since_flush_ = 0;
        return RELP_OK;
    }
    enqueue_flush();                                   // sharded tableau: the flush is local to the owned columns
    return RELP_OK;
}

// ---- step-wise API ------------------------------------------------------------------------------
relp_status_t Engine::select_primal_pivot_column(int rule, int32_t* found, int32_t* column, double* cost) {
    relp_status_t st = download_rec();
    if (st) return st;
    h_rec_->outcome = DEV_RUNNING;
    if ((st = upload_rec())) return st;
    if (tableau_) {
        SelectPartials sp;
        sp.k1 = d_part_k1_; sp.j = d_part_j_; sp.in_basis = d_in_basis_; sp.tol_cost = cfg_.tol_cost; sp.rule = rule;
        sp.n = nr_columns(); sp.offset = 0; sp.nb_struct = 0; sp.tol_tie = cfg_.tol_tie; sp.p_lo = 0; sp.cols_per_slot = 8;
        launch_tab_scan(tview(), sp, d_rec_, stream_);
        launch_tab_select(tview(), sp, tab_scan_blocks(sc_hi_ - sc_lo_), d_rec_, stream_);
    } else {
        enqueue_price(phase_, d_minus_pi_, d_rec_, col_lo_, col_hi_);
        launch_select_column(d_d_, d_in_basis_, nr_columns(), rule, cfg_.tol_cost, cfg_.tol_tie, d_rec_, stream_);
    }
    if ((st = download_rec())) return st;
    const bool ok = h_rec_->outcome == DEV_RUNNING;
    if (found) *found = ok ? 1 : 0;
    if (ok) { if (column) *column = h_rec_->q; if (cost) *cost = h_rec_->d_q; }
    h_rec_->outcome = DEV_RUNNING;
    return upload_rec();
}

relp_status_t Engine::relative_costs(double* out_n) {
    if (!tableau_) enqueue_price(phase_, d_minus_pi_, nullptr, col_lo_, col_hi_);
    const double* src = tableau_ ? d_d_ + (phase_ == 1 ? 0 : tab_na_) : d_d_;     // the tableau keeps d up to date
    HIP_TRY(hipMemcpyAsync(out_n, src, sizeof(double) * nr_columns(), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    return RELP_OK;
}

relp_status_t Engine::generate_column(int32_t column, double* out_m) {
    if (column < 0 || column >= nr_columns()) return fail(RELP_E_ARG, "column out of range");
    relp_status_t st = download_rec();
    if (st) return st;
    h_rec_->outcome = DEV_RUNNING;
    h_rec_->q = column;
    if ((st = upload_rec())) return st;
    if (tableau_) {
        launch_tab_column(tview(), deferred(), d_alpha_, d_rec_, stream_);
    } else if (lu_ && ft_) {
        launch_ft_ftran(dlu_, fts_, ft_problem(0), column, nullptr, d_alpha_, stream_);     // leaves the spike for change_basis
    } else if (lu_) {
        launch_build_column_csc(csc(), table(), m_, d_aq_, d_rec_, stream_);
        launch_lu_ftran(dlu_, d_aq_, d_v_, d_lu_scratch_, d_rec_, stream_);
        launch_apply_w(deferred(), m_, d_v_, d_alpha_, d_rec_, stream_);
    } else {
        enqueue_flush();                               // the step-wise calls work on the explicit inverse
        const double* A = dA_ - (int64_t)col_lo_ * ld_a_;
        double* Binv = dBinv_ - (int64_t)row_lo_ * ld_b_;
        launch_build_column(A, ld_a_, table(), m_, d_aq_, d_rec_, stream_);
        launch_ftran(Binv, ld_b_, m_, row_lo_, row_hi_, d_aq_, d_alpha_, 0, d_rec_, stream_);
    }
    if (out_m) HIP_TRY(hipMemcpyAsync(out_m, d_alpha_, sizeof(double) * m_, hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    return RELP_OK;
}

relp_status_t Engine::generate_element(int32_t row, int32_t column, double* out) {
    if (row < 0 || row >= m_) return fail(RELP_E_ARG, "row out of range");
    std::vector<double> col(m_);
    relp_status_t st = generate_column(column, col.data());
    if (st) return st;
    if (out) *out = col[row];
    return RELP_OK;
}

relp_status_t Engine::select_primal_pivot_row(int32_t* found, int32_t* row) {
    launch_ratio(d_alpha_, d_b_, d_basis_, m_, tolerances(), d_rec_, stream_);
    relp_status_t st = download_rec();
    if (st) return st;
    const bool ok = h_rec_->outcome == DEV_RUNNING;
    if (found) *found = ok ? 1 : 0;
    if (ok && row) *row = h_rec_->r;
    h_rec_->outcome = DEV_RUNNING;
    return upload_rec();
}

relp_status_t Engine::select_primal_pivot_row_of(const double* column, int32_t* found, int32_t* row) {
    HIP_TRY(hipMemcpyAsync(d_aq_, column, sizeof(double) * m_, hipMemcpyHostToDevice, stream_));
    launch_ratio(d_aq_, d_b_, d_basis_, m_, tolerances(), d_rec_, stream_);
    relp_status_t st = download_rec();
    if (st) return st;
    const bool ok = h_rec_->outcome == DEV_RUNNING;
    if (found) *found = ok ? 1 : 0;
    if (ok && row) *row = h_rec_->r;
    h_rec_->outcome = DEV_RUNNING;
    return upload_rec();
}

relp_status_t Engine::bring_into_basis(int32_t column, int32_t row, double cost, int32_t* leaving) {
    if (column < 0 || column >= nr_columns() || row < 0 || row >= m_) return fail(RELP_E_ARG, "index out of range");
    relp_status_t st = download_rec();
    if (st) return st;
    double alpha_r = 0.0, b_r = 0.0; int32_t lv = 0;
    HIP_TRY(hipMemcpy(&alpha_r, d_alpha_ + row, sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&b_r, d_b_ + row, sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&lv, d_basis_ + row, sizeof(int32_t), hipMemcpyDeviceToHost));
    if (alpha_r == 0.0) return fail(RELP_E_ZERO_PIVOT, "Pivot value can't be zero.");
    if (!tableau_ && !lu_) enqueue_flush();
    if ((st = download_rec())) return st;               // the flush may have reset the block counters
    h_rec_->outcome = DEV_RUNNING;
    h_rec_->q = column; h_rec_->d_q = cost; h_rec_->r = row; h_rec_->leaving = lv; h_rec_->alpha_r = alpha_r; h_rec_->b_r = b_r;
    if ((st = upload_rec())) return st;
    if (tableau_) {
        const TableauView tv = tview();
        const DeferredUpdate du = deferred();
        SelectPartials sp;
        sp.k1 = d_part_k1_; sp.j = d_part_j_; sp.in_basis = d_in_basis_; sp.tol_cost = cfg_.tol_cost;
        sp.rule = phase_ == 1 ? cfg_.phase_one_rule : cfg_.phase_two_rule;
        sp.n = tv.n; sp.offset = 0; sp.nb_struct = 0; sp.tol_tie = cfg_.tol_tie; sp.p_lo = 0; sp.cols_per_slot = 8;
        launch_eta_prepare(du, d_rec_, stream_);
        launch_tab_row_update(tv, du, sp, d_rec_, stream_);
        launch_update_w(du, m_, d_alpha_, d_rec_, stream_);
        launch_tab_update_vectors(m_, d_alpha_, d_b_, d_basis_, d_in_basis_, d_trace_, trace_cap_, d_rec_, stream_);
        if (++since_flush_ >= block_) enqueue_flush();
        HIP_TRY(hipStreamSynchronize(stream_));
        if (leaving) *leaving = lv;
        return RELP_OK;
    }
    if (lu_ && ft_) {
        // Carry::change_basis (carry/mod.rs:549-570): b, then the basis inverse (Forrest-Tomlin update with the spike
        // of the last generate_column), then -pi from row r of the NEW inverse
        const FtProblem pb = ft_problem(0);
        launch_ft_update(dlu_, fts_, pb, stream_);
        if ((st = ft_read_hdr())) return st;
        if (h_ft_hdr_[2] == 2) {
            // r did not fit the eta pool: refactorise the CURRENT basis, form the spike of the entering column again on
            // the fresh factors (alpha itself is unchanged) and update those
            std::vector<double> keep_alpha(m_);
            HIP_TRY(hipMemcpy(keep_alpha.data(), d_alpha_, sizeof(double) * m_, hipMemcpyDeviceToHost));
            if ((st = lu_refactor())) return st;
            launch_ft_ftran(dlu_, fts_, pb, column, nullptr, d_alpha_, stream_);
            HIP_TRY(hipMemcpyAsync(d_alpha_, keep_alpha.data(), sizeof(double) * m_, hipMemcpyHostToDevice, stream_));
            HIP_TRY(hipStreamSynchronize(stream_));
            launch_ft_update(dlu_, fts_, pb, stream_);
        }
        launch_ft_btran(dlu_, fts_, pb, -2, nullptr, d_rho_, stream_);
        launch_update_vectors(m_, d_alpha_, d_rho_, d_b_, d_minus_pi_, d_basis_, d_in_basis_, d_trace_, trace_cap_,
                              d_rec_, stream_);
        if ((st = ft_read_hdr())) return st;
        if (leaving) *leaving = lv;
        if (ft_need_refactor_) return lu_refactor();       // Carry::after_basis_change (carry/mod.rs:602-614)
        return RELP_OK;
    }
    if (lu_) {
        const DeferredUpdate du = deferred();
        launch_eta_prepare(du, d_rec_, stream_);
        launch_update_w(du, m_, d_alpha_, d_rec_, stream_);
        launch_lu_btran(dlu_, du, nullptr, -1, d_rho_, d_lu_scratch_, d_rec_, stream_);
        launch_update_vectors(m_, d_alpha_, d_rho_, d_b_, d_minus_pi_, d_basis_, d_in_basis_, d_trace_, trace_cap_,
                              d_rec_, stream_);
        if (++since_flush_ >= block_) enqueue_flush();
        HIP_TRY(hipStreamSynchronize(stream_));
        if (leaving) *leaving = lv;
        if (lu_status_) { const relp_status_t e = lu_status_; lu_status_ = RELP_OK; return e; }
        return RELP_OK;
    }
    double* Binv = dBinv_ - (int64_t)row_lo_ * ld_b_;
    launch_compute_rho(Binv, ld_b_, m_, row_lo_, row_hi_, d_rho_, d_rec_, stream_);
    launch_update_vectors(m_, d_alpha_, d_rho_, d_b_, d_minus_pi_, d_basis_, d_in_basis_, d_trace_, trace_cap_, d_rec_,
                          stream_);
    launch_update_inverse(Binv, ld_b_, m_, row_lo_, row_hi_, d_alpha_, d_rho_, d_rec_, stream_);
    HIP_TRY(hipStreamSynchronize(stream_));
    if (leaving) *leaving = lv;
    return RELP_OK;
}

// ------------------------------------------------------------------------------------------------
// Loops (phase_one.rs:125-170, phase_two.rs:22-51)
// ------------------------------------------------------------------------------------------------
static constexpr int32_t kHeldNoCandidate = 100;          // run_loop: no candidate left while columns are barred (pivot_rescue)

relp_status_t Engine::robust_stats(int64_t* out4) const {
    out4[0] = rescue_small_pivots_; out4[1] = rescue_barred_; out4[2] = rescue_confirmations_; out4[3] = reinvert_interval_;
    return RELP_OK;
}

relp_status_t Engine::rescue_unbar_all() {
    if (barred_.empty()) return RELP_OK;
    HIP_TRY(hipStreamSynchronize(stream_));
    const uint8_t zero = 0;
    for (int32_t j : barred_) HIP_TRY(hipMemcpy(d_in_basis_ + j, &zero, 1, hipMemcpyHostToDevice));
    barred_.clear();
    return RELP_OK;
}

// phase_one::primal / phase_two::primal with relp_config_t.pivot_rescue: the loop itself is run_loop(); an exit without a pivot
// row (phase_one.rs:143 panics there, phase_two.rs:44 returns Unbounded) is believed only when the entering column really has no
// entry worth pivoting on.
relp_status_t Engine::run(int64_t max_iters, int64_t* done, int32_t* outcome) {
    if (!cfg_.pivot_rescue || cfg_.shard_count > 1) return run_loop(max_iters, done, outcome);
    if (const char* e = std::getenv("RELP_PIVOT_GUARD")) guard_rel_ = std::atof(e);          // (measurement aid: the guard's relative threshold)
    struct GuardScope { bool& g; GuardScope(bool& x) : g(x) { g = true; } ~GuardScope() { g = false; } } guard_scope(pivot_guard_on_);
    int64_t total = 0;
    int32_t oc = RELP_RUNNING;
    relp_status_t st;
    bool confirming = false;                  // the outcome is being re-checked with every column priced again
    for (int guard = 0;; ++guard) {
        int64_t d = 0;
        hold_phase_end_ = !barred_.empty();
        st = run_loop(std::max<int64_t>(max_iters - total, 0), &d, &oc);
        hold_phase_end_ = false;
        if (st) return st;
        total += d;
        if (d > 0) confirming = false;
        const bool no_row = oc == RELP_NO_ROW_PHASE_ONE || oc == RELP_UNBOUNDED;
        if (no_row && guard < 100000) {
            if (d > 0 && (st = rescue_unbar_all())) return st;             // the basis has changed: the bars are out of date
            const int32_t q = h_rec_->q;
            const double d_q = h_rec_->d_q;
            std::vector<double> alpha(m_);
            HIP_TRY(hipStreamSynchronize(stream_));
            HIP_TRY(hipMemcpy(alpha.data(), d_alpha_, sizeof(double) * m_, hipMemcpyDeviceToHost));
            double amax = 0.0, apos = 0.0;
            for (double v : alpha) { amax = std::max(amax, std::fabs(v)); apos = std::max(apos, v); }
            const double rel = guard_rel_ > 0.0 ? guard_rel_ : cfg_.tol_pivot;       // the rescue's tolerance is relative to the column
            if (apos > 0.0 && apos >= rel * amax && apos > cfg_.tol_zero) {
                // a column with entries worth pivoting on relative to its own size (all of it small, or the loop's choice was
                // small beside larger entries): ONE pivot by the full-scan ratio test with the tolerance relative to the column
                const double keep = cfg_.tol_pivot;
                cfg_.tol_pivot = std::min(keep, std::max(rel * amax, 1e-300) * (1.0 - 1e-12));
                pivot_guard_on_ = false;
                int32_t found = 0, r = -1;
                st = generate_column(q, nullptr);
                if (!st) st = select_primal_pivot_row(&found, &r);
                if (!st && found) st = bring_into_basis(q, r, d_q, nullptr);
                cfg_.tol_pivot = keep;
                pivot_guard_on_ = true;
                if (st) return st;
                if (found) {
                    ++rescue_small_pivots_; ++total;
                    if ((st = download_rec())) return st;
                    h_rec_->outcome = DEV_RUNNING;                          // (the update kernels have counted and traced the pivot)
                    if ((st = upload_rec())) return st;
                    if (total >= max_iters) { oc = RELP_RUNNING; break; }
                    continue;
                }
            }
            if (apos <= 0.0 && barred_.empty() && !confirming) break;      // nothing positive at all: the outcome stands
            if (apos > 0.0 || !barred_.empty()) {
                if (apos <= 0.0) break;                                    // a genuinely unbounded column beside barred ones
                // positive entries that are noise beside the column's size (or below every tolerance): not a column to enter now
                const uint8_t two = 2;
                HIP_TRY(hipMemcpy(d_in_basis_ + q, &two, 1, hipMemcpyHostToDevice));
                barred_.push_back(q);
                ++rescue_barred_;
                if ((st = download_rec())) return st;
                h_rec_->outcome = DEV_RUNNING;
                if ((st = upload_rec())) return st;
                continue;
            }
            break;
        }
        if (oc == kHeldNoCandidate) {
            // no candidate left, but some columns were barred: price them again and see whether the outcome survives
            if (!confirming) {
                if ((st = rescue_unbar_all())) return st;
                confirming = true;
                ++rescue_confirmations_;
                if ((st = download_rec())) return st;
                h_rec_->outcome = DEV_RUNNING;
                if ((st = upload_rec())) return st;
                continue;
            }
            // barred again without a pivot in between: they stay out, the phase ends
            if ((st = download_rec())) return st;
            h_rec_->outcome = DEV_NO_CANDIDATE;
            if (phase_ == 2) oc = RELP_OPTIMAL;
            else if ((st = finish_phase_one(&oc))) return st;
            barred_.clear();                                               // (the phase switch rebuilt the flags)
            break;
        }
        break;
    }
    if (done) *done = total;
    if (outcome) *outcome = oc;
    return RELP_OK;
}

relp_status_t Engine::run_loop(int64_t max_iters, int64_t* done, int32_t* outcome) {
    if (ft_) return run_ft(max_iters, done, outcome);
    relp_status_t st = download_rec();
    if (st) return st;
    const long long start = h_rec_->iterations;
    const int rule = phase_ == 1 ? cfg_.phase_one_rule : cfg_.phase_two_rule;
    // Phase 1 often ends after very few pivots (none at all with a full slack basis): poll at 1, 2, 4, ...
    // there so that an early end does not leave a long tail of no-op launches queued.
    if (tableau_) {
        // (re)entering the loop: rebuild the PRICE partials from d (afterwards every pivot leaves them behind)
        launch_tab_scan(tview(), tab_partials(rule), d_rec_, stream_);
        tab_partials_valid_ = true;
    }
    int64_t next_poll = phase_ == 1 ? 1 : cfg_.poll_interval;
    const auto t_enq0 = std::chrono::steady_clock::now();
    int64_t enqueued = 0;
    {
    LoopScope loop(*this);
    for (int64_t it = 0; it < max_iters && h_rec_->outcome == DEV_RUNNING; ++it) {
        enqueue_iteration(rule);
        ++enqueued;
        if (lu_status_) { const relp_status_t e = lu_status_; lu_status_ = RELP_OK; return e; }
        if (reinvert_interval_ > 0 && ++since_reinvert_ >= reinvert_interval_) {
            if ((st = download_rec())) return st;
            if (h_rec_->outcome != DEV_RUNNING) break;
            if ((st = reinvert())) return st;
        }
        if (it + 1 == next_poll) {
            next_poll += phase_ == 1 ? std::min<int64_t>(next_poll, cfg_.poll_interval) : cfg_.poll_interval;
            if ((st = download_rec())) return st;
            if (h_rec_->outcome != DEV_RUNNING) break;
        }
    }
    }
    if (std::getenv("RELP_DEBUG") && enqueued >= 64)
        std::fprintf(stderr, "[relp] run: %lld pivots enqueued in %.1f us of host time each\n", (long long)enqueued,
                     std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_enq0).count() / enqueued);
    if ((st = download_rec())) return st;
    if (hipGetLastError() != hipSuccess) return fail(RELP_E_HIP, "kernel launch failed");
    int32_t oc = RELP_RUNNING;
    if (h_rec_->outcome == DEV_NO_CANDIDATE) {
        if (hold_phase_end_) oc = kHeldNoCandidate;
        else if (phase_ == 2) oc = RELP_OPTIMAL;
        else if ((st = finish_phase_one(&oc))) return st;
    } else if (h_rec_->outcome == DEV_NO_ROW) {
        oc = phase_ == 2 ? RELP_UNBOUNDED : RELP_NO_ROW_PHASE_ONE;
    }
    if (done) *done = h_rec_->iterations - start;          // includes the zero-level pivots of the phase boundary
    if (outcome) *outcome = oc;
    return RELP_OK;
}

relp_status_t Engine::solve_relaxation(int64_t max_iters, int32_t* outcome) {
    int32_t oc = RELP_RUNNING; int64_t done = 0, total = 0;
    relp_status_t st;
    if (phase_ == 1) {
        if ((st = run(max_iters, &done, &oc))) return st;
        total += done;
        if (oc != RELP_PHASE_ONE_DONE) { if (outcome) *outcome = oc; return RELP_OK; }
    }
    if ((st = run(std::max<int64_t>(max_iters - total, 0), &done, &oc))) return st;
    if (outcome) *outcome = oc;
    return RELP_OK;
}

// phase_one.rs:146-166: objective == 0 -> feasible (remove artificials, switch) else infeasible
relp_status_t Engine::finish_phase_one(int32_t* outcome) {
    tab_settle();
    const double obj = -h_rec_->minus_objective;
    if (std::fabs(obj) > cfg_.tol_feas * std::max(1.0, initial_phase1_objective_)) { *outcome = RELP_INFEASIBLE; return RELP_OK; }
    std::vector<int32_t> rows_to_remove;
    if (cfg_.shard_count == 1 || tableau_) enqueue_flush();        // the phase boundary works on the explicit inverse
    relp_status_t st = (cfg_.shard_count > 1 && tableau_) ? remove_artificial_basis_variables_sharded(rows_to_remove)
                                                          : remove_artificial_basis_variables(rows_to_remove);
    if (st) return st;
    if ((st = switch_to_phase_two(rows_to_remove))) return st;
    *outcome = RELP_PHASE_ONE_DONE;
    return RELP_OK;
}

// phase_one.rs:223-260 (pivots "at zero level"; pushes the artificial index like the reference)
relp_status_t Engine::remove_artificial_basis_variables(std::vector<int32_t>& rows_to_remove) {
    HIP_TRY(hipStreamSynchronize(stream_));            // null-stream copies below vs. kernels on stream_
    std::vector<int32_t> basis(m_);
    HIP_TRY(hipMemcpy(basis.data(), d_basis_, sizeof(int32_t) * m_, hipMemcpyDeviceToHost));
    std::vector<int32_t> arts;
    for (int32_t v : basis) if (v < nr_artificial_) arts.push_back(v);
    if (arts.empty()) return RELP_OK;
    std::sort(arts.begin(), arts.end());
    const int n = nr_columns();
    std::vector<double> d(n), tau(n);
    std::vector<uint8_t> inb(n);
    relp_status_t st;
    const bool textbook = cfg_.artificial_removal == RELP_ARTIFICIAL_TEXTBOOK;
    for (int32_t a : arts) {
        int32_t pivot_row = column_to_row_[a];             // phase_one.rs:236: the row the artificial STARTED in
        if (textbook) pivot_row = (int32_t)(std::find(basis.begin(), basis.end(), a) - basis.begin());   // the row it is basic in
        if ((st = relative_costs(d.data()))) return st;
        // tableau row pivot_row over every column: (row of B^-1) . a_j, no cost term
        if (tableau_) {
            launch_tab_row(tview(), deferred(), pivot_row, d_aq_big(), d_rec_, stream_);        // single GPU: all columns
            HIP_TRY(hipMemcpyAsync(tau.data(), d_aq_big(), sizeof(double) * n, hipMemcpyDeviceToHost, stream_));
        } else if (lu_) {
            if (ft_) launch_ft_btran(dlu_, fts_, ft_problem(0), pivot_row, nullptr, d_rho_, stream_);
            else launch_lu_btran(dlu_, deferred(), nullptr, pivot_row, d_rho_, d_lu_scratch_, nullptr, stream_);
            enqueue_price(0, d_rho_, nullptr, 0, nr_normal_);
            HIP_TRY(hipMemcpyAsync(tau.data(), d_d_, sizeof(double) * n, hipMemcpyDeviceToHost, stream_));
        } else {
            double* Binv = dBinv_ - (int64_t)row_lo_ * ld_b_;
            enqueue_price(0, Binv + (int64_t)pivot_row * ld_b_, nullptr, col_lo_, col_hi_);
            HIP_TRY(hipMemcpyAsync(tau.data(), d_d_, sizeof(double) * n, hipMemcpyDeviceToHost, stream_));
        }
        HIP_TRY(hipMemcpyAsync(inb.data(), d_in_basis_, n, hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
        int32_t q = -1;
        for (int32_t j = nr_artificial_; j < n; ++j) {
            if (inb[j]) continue;
            if (!textbook && std::fabs(d[j]) > cfg_.tol_cost) continue;        // phase_one.rs:241: cost.is_zero()
            if (std::fabs(tau[j]) > cfg_.tol_pivot) { q = j; break; }
        }
        // phase_one.rs:252 pushes the artificial's index; RELP_ARTIFICIAL_TEXTBOOK its own row, and remove_rows moves the
        // artificial into that position first (relp_engine.h)
        if (q < 0) {
            if (textbook) { stuck_artificials_.push_back(a); rows_to_remove.push_back(column_to_row_[a]); }
            else rows_to_remove.push_back(a);
            continue;
        }
        if ((st = generate_column(q, nullptr))) return st;
        if ((st = bring_into_basis(q, pivot_row, d[q], nullptr))) return st;
        basis[pivot_row] = q;
    }
    return RELP_OK;
}

// kind/non_artificial.rs:151-220, carry/mod.rs:484-510 (+ :650-689 when rows are removed)
relp_status_t Engine::switch_to_phase_two(const std::vector<int32_t>& rows_to_remove) {
    relp_status_t st;
    if (cfg_.shard_count == 1 || tableau_) enqueue_flush();     // zero-level pivots may have left updates pending
    // the host-side copies below go through the null stream, which does not order with stream_: everything
    // enqueued so far (the flush, a possible re-pricing that uses d_w_) must have finished first
    HIP_TRY(hipStreamSynchronize(stream_));
    if (!rows_to_remove.empty()) {
        std::vector<int32_t> rows(rows_to_remove);          // ascending and distinct: remove_rows walks the list beside the rows
        std::sort(rows.begin(), rows.end());
        rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
        st = remove_rows(rows);
        stuck_artificials_.clear();
        if (st) return st;
    }
    std::vector<int32_t> basis(m_);
    HIP_TRY(hipMemcpy(basis.data(), d_basis_, sizeof(int32_t) * m_, hipMemcpyDeviceToHost));
    const int32_t na = nr_artificial_;
    // An artificial variable can survive remove_artificial_basis_variables: when it re-entered the basis in
    // another row than its own, the zero-level pivot of phase_one.rs:236 is made in its ORIGINAL row.  In
    // the reference's release-built integration tests `basis_column -= nr_artificial` (carry/mod.rs:524, 663)
    // then wraps to a huge usize: a column without cost that sorts last in Bland's tie-break and stays basic
    // at value zero until the ratio test removes it (Netlib BOEING2 walks through this state).  Same here
    // with the wrapped index squeezed into int32, order preserved: INT32_MAX - (na - 1 - a).
    for (auto& v : basis) v = v < na ? INT32_MAX - (na - 1 - v) : v - na;
    wrapped_na_ = na;
    nr_artificial_ = 0;
    phase_ = 2;
    HIP_TRY(hipMemcpy(d_basis_, basis.data(), sizeof(int32_t) * m_, hipMemcpyHostToDevice));
    std::vector<uint8_t> flags(n_alloc_, 0);
    for (int32_t v : basis) {
        if (v >= kWrappedArtificialBase) continue;
        if (v < 0 || v >= n_provider_) return fail(RELP_E_STATE, "basis column out of range at the phase switch");
        flags[v] = 1;
    }
    HIP_TRY(hipMemcpy(d_in_basis_, flags.data(), flags.size(), hipMemcpyHostToDevice));
    // -pi = -(c_B' B^-1) (create_minus_pi_from_artificial, carry/mod.rs:214-248), accumulated over rows in order
    std::vector<double> w(m_, 0.0), b(m_);
    for (int32_t i = 0; i < m_; ++i) if (basis[i] < nr_normal_) w[i] = cost_h_[basis[i]];
    if (cfg_.shard_count > 1 && !tableau_)
        for (double v : w) if (v != 0.0) return fail(RELP_E_UNSUPPORTED, "sharded phase switch needs an all-slack basis");
    if (lu_) for (auto& v : w) v = -v;                 // BTRAN with rhs -c_B gives -pi directly
    HIP_TRY(hipMemcpy(d_w_, w.data(), sizeof(double) * m_, hipMemcpyHostToDevice));
    if (lu_) {
        if (lu_status_) { const relp_status_t e = lu_status_; lu_status_ = RELP_OK; return e; }
        if (ft_) launch_ft_btran(dlu_, fts_, ft_problem(0), -1, d_w_, d_minus_pi_, stream_);
        else launch_lu_btran(dlu_, deferred(), d_w_, -1, d_minus_pi_, d_lu_scratch_, nullptr, stream_);
    } else if (tableau_) {
        // phase-2 reduced costs of every stored column: d = c - c_B' T (the artificial block keeps cost 0)
        cost_store_h_.assign(n_store_, 0.0);
        for (int32_t p = 0; p < nr_normal_; ++p) cost_store_h_[tab_na_ + p] = cost_h_[p];
        HIP_TRY(hipMemcpy(d_cost_store_, cost_store_h_.data(), sizeof(double) * n_store_, hipMemcpyHostToDevice));
        launch_tab_price_init(tview(), d_w_, d_cost_store_, stream_);
        tab_partials_valid_ = false;
    } else {
        double* Binv = dBinv_ - (int64_t)row_lo_ * ld_b_;
        launch_weighted_column_sums(Binv, ld_b_, m_, d_w_, d_minus_pi_, stream_);
    }
    // -obj = -sum_i b_i c_B(i) (create_minus_obj_from_artificial, carry/mod.rs:258-271)
    HIP_TRY(hipMemcpyAsync(b.data(), d_b_, sizeof(double) * m_, hipMemcpyDeviceToHost, stream_));
    if ((st = download_rec())) return st;
    double objective = 0.0;
    for (int32_t i = 0; i < m_; ++i) if (basis[i] < nr_normal_) objective += b[i] * cost_h_[basis[i]];
    h_rec_->minus_objective = -objective;
    h_rec_->outcome = DEV_RUNNING;
    h_rec_->last_selected = -1;
    h_rec_->phase = 2;
    return upload_rec();
}

// Rank-deficient problems (filter/generic_wrapper.rs:51, carry/mod.rs:650-689, basis_inverse_rows.rs:190-204):
// delete the given rows (and the same columns of B^-1) everywhere.  Rare, host round trip.
relp_status_t Engine::remove_rows(const std::vector<int32_t>& rows) {
    if (cfg_.shard_count > 1 && !tableau_) return fail(RELP_E_UNSUPPORTED, "row removal in the sharded revised engine");
    for (size_t k = 0; k < rows.size(); ++k)
        if (rows[k] < 0 || rows[k] >= m_ || (k > 0 && rows[k] <= rows[k - 1])) return fail(RELP_E_STATE, "rows to remove must be ascending and distinct");
    std::vector<int32_t> map(m_, 0);   // old row -> new row, -1 = removed
    {
        size_t f = 0; int32_t out = 0;
        for (int32_t i = 0; i < m_; ++i) {
            if (f < rows.size() && rows[f] == i) { map[i] = -1; ++f; } else map[i] = out++;
        }
    }
    const int32_t m_new = m_ - (int32_t)rows.size();
    for (int32_t r : rows) if (r >= mc_) return fail(RELP_E_STATE, "only constraint rows can be redundant");
    // The reference marks the INDEX of a stuck artificial as the redundant row (phase_one.rs:252).  When that index
    // lands on a <= or >= row, a row that is not redundant is deleted and what remains is no longer the inverse of a
    // basis of the filtered problem: the literal state is kept from here on, never rebuilt from the columns.
    for (int32_t r : rows) if (r >= nr_eq_ + nr_range_) reinvert_interval_ = 0;
    const int32_t mc_new = mc_ - (int32_t)rows.size();
    // RELP_ARTIFICIAL_TEXTBOOK: an artificial variable that is stuck in basis position r but started in row o != r takes
    // its own constraint with it, i.e. the pair (constraint o, position r) goes -- always a basis of the filtered problem,
    // which (r, r) is only if (B^-1)[r][r] != 0.  Positions are labels: exchanging r and o (rows of B^-1 / of the tableau,
    // b, the basis array) first lets one index name both.  perm: position i of the new order holds the old position perm[i].
    std::vector<int32_t> perm(m_), basis_perm;
    for (int32_t i = 0; i < m_; ++i) perm[i] = i;
    bool identity_perm = true;
    if (!stuck_artificials_.empty()) {
        basis_perm.resize(m_);
        HIP_TRY(hipStreamSynchronize(stream_));
        HIP_TRY(hipMemcpy(basis_perm.data(), d_basis_, sizeof(int32_t) * m_, hipMemcpyDeviceToHost));
        for (int32_t a : stuck_artificials_) {
            const int32_t o = column_to_row_[a];
            const int32_t cur = (int32_t)(std::find(basis_perm.begin(), basis_perm.end(), a) - basis_perm.begin());
            if (cur >= m_) return fail(RELP_E_STATE, "a stuck artificial variable is not basic");
            if (cur == o) continue;
            std::swap(basis_perm[cur], basis_perm[o]); std::swap(perm[cur], perm[o]);
            identity_perm = false;
        }
    }
    // B^-1 (or the tableau), b, basis
    const bool no_inv = tableau_ || lu_;
    std::vector<double> Bh(no_inv ? 1 : (size_t)m_ * ld_b_), b(m_);
    std::vector<int32_t> basis(m_);
    if (tableau_) {
        // every stored column loses the rows; the columns that were the identity of those rows stay as
        // (never priced) artificial columns.  Column by column to bound the host buffer.
        HIP_TRY(hipStreamSynchronize(stream_));
        std::vector<double> col(ld_t_), coln(ld_t_);
        const int32_t n_owned = sc_hi_ - sc_lo_;           // the stored columns of this rank (all of them unsharded)
        for (int32_t c = 0; c < n_owned; ++c) {
            HIP_TRY(hipMemcpy(col.data(), dT0_ + (int64_t)c * ld_t_, sizeof(double) * m_, hipMemcpyDeviceToHost));
            std::fill(coln.begin(), coln.end(), 0.0);
            for (int32_t i = 0; i < m_; ++i) if (map[i] >= 0) coln[map[i]] = col[perm[i]];
            HIP_TRY(hipMemcpy(dT0_ + (int64_t)c * ld_t_, coln.data(), sizeof(double) * ld_t_, hipMemcpyHostToDevice));
        }
        std::vector<int32_t> idn;
        for (int32_t i = 0; i < m_; ++i) if (map[i] >= 0) idn.push_back(idcol_h_[i]);
        idcol_h_ = idn;
        HIP_TRY(hipMemcpy(d_idcol_, idcol_h_.data(), sizeof(int32_t) * idcol_h_.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemset(d_pos_of_row_, 0xFF, sizeof(int32_t) * m_));
    } else if (lu_) {
        // the CSC matrix loses the rows; the factors are rebuilt below
        HIP_TRY(hipStreamSynchronize(stream_));
        std::vector<int64_t> np(nr_normal_ + 1, 0);
        size_t o = 0;
        for (int32_t j = 0; j < nr_normal_; ++j) {
            for (int64_t e = hc_ptr_[j]; e < hc_ptr_[j + 1]; ++e)
                if (map[hc_idx_[e]] >= 0) { hc_idx_[o] = map[hc_idx_[e]]; hc_val_[o] = hc_val_[e]; ++o; }
            np[j + 1] = (int64_t)o;
        }
        hc_idx_.resize(o); hc_val_.resize(o); hc_ptr_ = np;
        HIP_TRY(hipMemcpy(d_cptr_, hc_ptr_.data(), sizeof(int64_t) * hc_ptr_.size(), hipMemcpyHostToDevice));
        if (o) {
            HIP_TRY(hipMemcpy(d_cidx_, hc_idx_.data(), sizeof(int32_t) * o, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(d_cval_, hc_val_.data(), sizeof(double) * o, hipMemcpyHostToDevice));
        }
        if (ft_) { const relp_status_t pst = ft_build_price_ell(); if (pst) return pst; }
    } else {
        HIP_TRY(hipMemcpy(Bh.data(), dBinv_, Bh.size() * sizeof(double), hipMemcpyDeviceToHost));
    }
    HIP_TRY(hipMemcpy(b.data(), d_b_, sizeof(double) * m_, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(basis.data(), d_basis_, sizeof(int32_t) * m_, hipMemcpyDeviceToHost));
    if (!identity_perm) {                              // the stuck artificial variables move into their own rows (see above)
        std::vector<double> b0(b), B0;
        for (int32_t i = 0; i < m_; ++i) b[i] = b0[perm[i]];
        basis = basis_perm;
        if (!no_inv) {
            B0 = Bh;
            for (int32_t i = 0; i < m_; ++i)
                if (perm[i] != i) std::copy(B0.begin() + (size_t)perm[i] * ld_b_, B0.begin() + (size_t)(perm[i] + 1) * ld_b_, Bh.begin() + (size_t)i * ld_b_);
        }
    }
    const int64_t ld_new = ld_b_;
    std::vector<double> Bn(no_inv ? 1 : (size_t)m_ * ld_new, 0.0), bn(m_, 0.0);
    std::vector<int32_t> basisn(m_, 0);
    for (int32_t i = 0; i < m_; ++i) {
        if (map[i] < 0) continue;
        if (!no_inv)
            for (int32_t j = 0; j < m_; ++j) if (map[j] >= 0) Bn[(size_t)map[i] * ld_new + map[j]] = Bh[(size_t)i * ld_b_ + j];
        bn[map[i]] = b[i];
        basisn[map[i]] = basis[i];
    }
    if (!no_inv) HIP_TRY(hipMemcpy(dBinv_, Bn.data(), Bn.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(d_b_, 0, sizeof(double) * ld_b_));
    HIP_TRY(hipMemcpy(d_b_, bn.data(), sizeof(double) * m_new, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_basis_, basisn.data(), sizeof(int32_t) * m_new, hipMemcpyHostToDevice));
    // A: drop the rows inside every structural column (the tableau engine no longer reads A)
    if (nr_normal_ > 0 && !lu_ && cfg_.shard_count == 1) {   // (the unsharded tableau engine re-tabulates from A)
        std::vector<double> Ah((size_t)ld_a_ * nr_normal_);
        HIP_TRY(hipMemcpy(Ah.data(), dA_, Ah.size() * sizeof(double), hipMemcpyDeviceToHost));
        std::vector<double> An((size_t)ld_a_ * nr_normal_, 0.0);
        for (int32_t j = 0; j < nr_normal_; ++j)
            for (int32_t i = 0; i < mc_; ++i) if (map[i] >= 0) An[(size_t)j * ld_a_ + map[i]] = Ah[(size_t)j * ld_a_ + i];
        if (!owns_A_) { HIP_TRY(dev_alloc(&dA_, (int64_t)An.size())); owns_A_ = true; }
        HIP_TRY(hipMemcpy(dA_, An.data(), An.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    // descriptors: every remaining row index shifts down (column/into_filtered, matrix_data.rs:592-614)
    auto remap = [&](std::vector<int32_t>& v) { for (auto& x : v) if (x >= 0) x = map[x]; };
    remap(bound_row_h_); remap(vrow0_h_); remap(vrow1_h_);
    // a slack whose row disappears keeps its column index and becomes an empty column (vrow0 = -1), as
    // Column::into_filtered does (matrix_data.rs:592-614)
    for (auto& x : column_to_row_) x = map[x] >= 0 ? map[x] : 0;
    HIP_TRY(hipMemcpy(d_bound_row_, bound_row_h_.data(), sizeof(int32_t) * nr_normal_, hipMemcpyHostToDevice));
    if (nr_virtual_) {
        HIP_TRY(hipMemcpy(d_vrow0_, vrow0_h_.data(), sizeof(int32_t) * nr_virtual_, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_vrow1_, vrow1_h_.data(), sizeof(int32_t) * nr_virtual_, hipMemcpyHostToDevice));
    }
    std::vector<double> rhs_new;
    for (int32_t i = 0; i < m_; ++i) if (map[i] >= 0) rhs_new.push_back(rhs_h_[i]);
    rhs_h_ = rhs_new;
    m_ = m_new; mc_ = mc_new;
    row_lo_ = 0; row_hi_ = m_;
    row_stride_ = (int32_t)round_up(m_, 2);
    cand_len_ = candidate_len_for(m_);
    // stale tails of the m-vectors must be zero for the 16-byte loads
    HIP_TRY(hipMemset(d_alpha_, 0, sizeof(double) * ld_b_));
    HIP_TRY(hipMemset(d_aq_, 0, sizeof(double) * ld_b_));
    HIP_TRY(hipMemset(d_rho_, 0, sizeof(double) * ld_b_));
    HIP_TRY(hipMemset(d_minus_pi_, 0, sizeof(double) * ld_b_));
    HIP_TRY(hipDeviceSynchronize());
    if (lu_) return lu_refactor();
    return RELP_OK;
}

relp_status_t Engine::set_reinversion_interval(int64_t pivots) {
    if (pivots < 0) return fail(RELP_E_ARG, "negative interval");
    if (pivots > 0 && (lu_ || cfg_.shard_count > 1))
        return fail(RELP_E_UNSUPPORTED, "re-inversion is for the unsharded revised and tableau engines (the LU engine refactorises anyway)");
    reinvert_interval_ = pivots;
    since_reinvert_ = 0;
    return RELP_OK;
}

// Sparse columns of the current basis in row space, for the host factorisation: artificial columns (phase 1,
// and the ones that survived it with a wrapped index) are unit columns of their rows, structural columns come
// from the device-resident A (+ their bound row), virtual columns from the descriptors.
relp_status_t Engine::build_basis_columns(const std::vector<int32_t>& basis,
                                          std::vector<std::vector<std::pair<int32_t, double>>>* cols) {
    cols->assign(m_, {});
    std::vector<double> colbuf(std::max(mc_, 1));
    for (int32_t i = 0; i < m_; ++i) {
        const int32_t j = basis[i];
        auto& c = (*cols)[i];
        if (j >= kWrappedArtificialBase) { c.emplace_back(column_to_row_[wrapped_na_ - 1 - (INT32_MAX - j)], 1.0); continue; }
        if (j < nr_artificial_) { c.emplace_back(column_to_row_[j], 1.0); continue; }
        const int32_t p = j - nr_artificial_;
        if (p < 0 || p >= n_provider_) return fail(RELP_E_STATE, "basis column out of range");
        if (p < nr_normal_) {
            HIP_TRY(hipMemcpy(colbuf.data(), dA_ + (int64_t)(p - col_lo_) * ld_a_, sizeof(double) * mc_, hipMemcpyDeviceToHost));
            for (int32_t r = 0; r < mc_; ++r) if (colbuf[r] != 0.0) c.emplace_back(r, colbuf[r]);
            if (bound_row_h_[p] >= 0) c.emplace_back(bound_row_h_[p], 1.0);
        } else {
            const int32_t v = p - nr_normal_;
            if (vrow0_h_[v] >= 0) c.emplace_back(vrow0_h_[v], (double)vsign_h_[v]);
            if (vrow1_h_[v] >= 0) c.emplace_back(vrow1_h_[v], 1.0);
        }
    }
    return RELP_OK;
}

// B^-1, b, -pi and -obj from scratch for the CURRENT basis (either phase): host LU of the basis columns, then on
// the device the m unit BTRANs, b = B^-1 rhs and -pi = -(c_B' B^-1) with the costs of the phase.  What
// `LUDecomposition` does every 11 updates (lower_upper/mod.rs:199-202) the f64 explicit inverse needs every now and
// then: it is only ever updated, and on ill-conditioned LPs its error reaches the pivot tolerance after a few
// thousand pivots.
// The dense tableau's counterpart: T0 = B^-1 [artificial | A + bound rows | virtual] recomputed column by column from
// a fresh factorisation of the basis (one launch: workgroup c solves stored column c), b = B^-1 rhs, d re-priced
// from the new T0, -obj from b.  T0 is otherwise only ever updated (every flush adds W R0 to it).
relp_status_t Engine::retabulate() {
    SettledScope settled(*this);
    since_reinvert_ = 0;
    retab_done_ = false;
    enqueue_flush();
    HIP_TRY(hipStreamSynchronize(stream_));
    std::vector<double> b_before;
    if (cfg_.auto_reinversion) { b_before.resize(m_); HIP_TRY(hipMemcpy(b_before.data(), d_b_, sizeof(double) * m_, hipMemcpyDeviceToHost)); }
    std::vector<int32_t> basis(m_);
    HIP_TRY(hipMemcpy(basis.data(), d_basis_, sizeof(int32_t) * m_, hipMemcpyDeviceToHost));
    std::vector<std::vector<std::pair<int32_t, double>>> cols;
    relp_status_t st = build_basis_columns(basis, &cols);
    if (st) return st;
    std::string msg;
    if (!lu_factor(m_, cols, &hlu_, &msg)) return RELP_OK;        // keep the updated tableau
    if ((st = lu_upload_factors())) return st;
    if (!d_lu_scratch_) HIP_TRY(dev_alloc(&d_lu_scratch_, ld_b_));
    const TableauView tv = tview();
    const double* A = dA_ - (int64_t)col_lo_ * ld_a_;
    // every stored column, the artificial block included: phase 2 never prices it, but B^-1 and -pi are read off the
    // columns that were the identity originally (relp_get_basis_inverse, relp_basis_inverse_row, relp_get_minus_pi)
    const int32_t c_first = sc_lo_;
    ColumnTable storage = table();
    storage.nr_artificial = tab_na_;                    // storage columns keep the artificial block in front
    if (!launch_lu_ftran_cols(dlu_, tv, A, ld_a_, storage, c_first, sc_hi_ - c_first, stream_)) return RELP_OK;
    HIP_TRY(hipMemcpyAsync(d_aq_, rhs_h_.data(), sizeof(double) * m_, hipMemcpyHostToDevice, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    launch_lu_ftran(dlu_, d_aq_, d_b_, d_lu_scratch_, nullptr, stream_);                 // b = B^-1 rhs
    // d = c - c_B' T0 and the PRICE partials, as after every few flushes
    const int rule = phase_ == 1 ? cfg_.phase_one_rule : cfg_.phase_two_rule;
    launch_tab_basis_costs(tv, d_basis_, d_cost_store_, d_w_, stream_);
    launch_tab_price_init(tv, d_w_, d_cost_store_, stream_);
    launch_tab_scan(tv, tab_partials(rule), d_rec_, stream_);
    tab_partials_valid_ = true;
    flushes_since_reprice_ = 0;
    std::vector<double> b(m_);
    HIP_TRY(hipMemcpyAsync(b.data(), d_b_, sizeof(double) * m_, hipMemcpyDeviceToHost, stream_));
    if ((st = download_rec())) return st;
    double objective = 0.0;
    for (int32_t i = 0; i < m_; ++i) {
        const int32_t j = basis[i];
        if (j >= kWrappedArtificialBase) continue;
        if (phase_ == 1) { if (j < nr_artificial_) objective += b[i]; }
        else if (j < nr_normal_) objective += cost_h_[j] * b[i];
    }
    h_rec_->minus_objective = -objective;
    ++reinversions_;
    retab_done_ = true;
    if (cfg_.auto_reinversion) auto_reinversion_adapt(b_before, b);
    return upload_rec();
}

relp_status_t Engine::reinvert() {
    SettledScope settled(*this);
    if (lu_ || cfg_.shard_count > 1) return fail(RELP_E_UNSUPPORTED, "re-inversion is for the unsharded revised and tableau engines");
    if (tableau_) return retabulate();
    since_reinvert_ = 0;
    enqueue_flush();
    HIP_TRY(hipStreamSynchronize(stream_));
    std::vector<double> b_before;
    if (cfg_.auto_reinversion) { b_before.resize(m_); HIP_TRY(hipMemcpy(b_before.data(), d_b_, sizeof(double) * m_, hipMemcpyDeviceToHost)); }
    std::vector<int32_t> basis(m_);
    HIP_TRY(hipMemcpy(basis.data(), d_basis_, sizeof(int32_t) * m_, hipMemcpyDeviceToHost));
    std::vector<std::vector<std::pair<int32_t, double>>> cols;
    relp_status_t st = build_basis_columns(basis, &cols);
    if (st) return st;
    std::string msg;
    if (!lu_factor(m_, cols, &hlu_, &msg)) return RELP_OK;        // numerically singular for the LU: keep the updated inverse
    if ((st = lu_upload_factors())) return st;
    if (!d_lu_scratch_) HIP_TRY(dev_alloc(&d_lu_scratch_, ld_b_));
    HIP_TRY(hipMemsetAsync(dBinv_, 0, sizeof(double) * (size_t)m_ * ld_b_, stream_));
    DeferredUpdate none = deferred();
    none.kmax = 0;
    launch_lu_btran_rows(dlu_, none, dBinv_, ld_b_, d_lu_scratch_, stream_);
    // costs of the phase: Cost::One on artificial columns in phase 1, the variable costs in phase 2
    std::vector<double> w(ld_b_, 0.0), b(m_);
    for (int32_t i = 0; i < m_; ++i) {
        const int32_t j = basis[i];
        if (j >= kWrappedArtificialBase) continue;
        if (phase_ == 1) w[i] = j < nr_artificial_ ? 1.0 : 0.0;
        else if (j < nr_normal_) w[i] = cost_h_[j];
    }
    HIP_TRY(hipMemcpyAsync(d_w_, w.data(), sizeof(double) * ld_b_, hipMemcpyHostToDevice, stream_));
    HIP_TRY(hipMemcpyAsync(d_aq_, rhs_h_.data(), sizeof(double) * m_, hipMemcpyHostToDevice, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    launch_weighted_column_sums(dBinv_, ld_b_, m_, d_w_, d_minus_pi_, stream_);
    launch_ftran(dBinv_, ld_b_, m_, 0, m_, d_aq_, d_b_, 0, d_rec_, stream_);
    HIP_TRY(hipMemcpyAsync(b.data(), d_b_, sizeof(double) * m_, hipMemcpyDeviceToHost, stream_));
    if ((st = download_rec())) return st;
    double objective = 0.0;
    for (int32_t i = 0; i < m_; ++i) objective += w[i] * b[i];
    h_rec_->minus_objective = -objective;
    ++reinversions_;
    if (cfg_.auto_reinversion) auto_reinversion_adapt(b_before, b);
    return upload_rec();
}

// relp_config_t.auto_reinversion: a rebuild that moved b by more than 1e-7 (relative to its largest entry) came too late -- the
// interval is halved, down to 32 pivots; one that moved it by less than 1e-10 could have waited -- doubled, up to 1,024 (a small
// drift of b does not vouch for the inverse: GREENBEA's tableau is lost at 4,096)
void Engine::auto_reinversion_adapt(const std::vector<double>& before, const std::vector<double>& after) {
    double diff = 0.0, scale = 1.0;
    for (size_t i = 0; i < after.size() && i < before.size(); ++i) { diff = std::max(diff, std::fabs(after[i] - before[i])); scale = std::max(scale, std::fabs(after[i])); }
    last_reinvert_drift_ = diff / scale;
    if (last_reinvert_drift_ > 1e-7) reinvert_interval_ = std::max<int64_t>(32, reinvert_interval_ / 2);
    else if (last_reinvert_drift_ < 1e-10) reinvert_interval_ = std::min<int64_t>(1024, reinvert_interval_ * 2);
    if (std::getenv("RELP_DEBUG")) std::fprintf(stderr, "[relp] rebuild %lld: b moved by %.2e (relative), interval now %lld\n", (long long)reinversions_, last_reinvert_drift_, (long long)reinvert_interval_);
}

relp_status_t Engine::from_basis(const int32_t* basis_columns) {
    // InverseMaintener::from_basis (carry/mod.rs:428-463): any basis on the LU and the revised engine (the
    // latter factorises on the host and runs the m unit solves on the device; slack bases are a signed permutation and take a shortcut).
    if (cfg_.shard_count > 1) return fail(RELP_E_UNSUPPORTED, "from_basis in sharded mode");
    if (tableau_) {
        // T = B^-1 [A | I] for the given basis: factorise it (host, like every refactorisation), then every stored column
        // by one FTRAN each in a single launch (re-tabulation), b = B^-1 rhs, d = c - c_B' T
        enqueue_flush();
        HIP_TRY(hipStreamSynchronize(stream_));
        std::vector<int32_t> basis(basis_columns, basis_columns + m_);
        std::vector<uint8_t> flags(n_alloc_, 0);
        for (int32_t v : basis) {
            if (v < 0 || v >= n_provider_) return fail(RELP_E_ARG, "from_basis: column out of range");
            if (flags[v]) return fail(RELP_E_SINGULAR, "from_basis: duplicate column");
            flags[v] = 1;
        }
        const int32_t keep_na = nr_artificial_, keep_phase = phase_;
        nr_artificial_ = 0; phase_ = 2;
        HIP_TRY(hipMemcpy(d_basis_, basis.data(), sizeof(int32_t) * m_, hipMemcpyHostToDevice));
        cost_store_h_.assign(n_store_, 0.0);
        for (int32_t p = 0; p < nr_normal_; ++p) cost_store_h_[tab_na_ + p] = cost_h_[p];
        HIP_TRY(hipMemcpy(d_cost_store_, cost_store_h_.data(), sizeof(double) * n_store_, hipMemcpyHostToDevice));
        relp_status_t st = retabulate();
        if (st || !retab_done_) {
            nr_artificial_ = keep_na; phase_ = keep_phase;
            return st ? st : fail(RELP_E_SINGULAR, "from_basis: the basis could not be factorised (or m is too large for the "
                                                    "LDS-resident solves of the re-tabulation)");
        }
        HIP_TRY(hipMemcpy(d_in_basis_, flags.data(), flags.size(), hipMemcpyHostToDevice));
        if ((st = download_rec())) return st;
        h_rec_->outcome = DEV_RUNNING; h_rec_->last_selected = -1; h_rec_->phase = 2;
        tab_partials_valid_ = false;
        return upload_rec();
    }
    if (lu_) {
        // any basis: factorise it, b = B^-1 rhs (FTRAN), -pi = -c_B' B^-1 (BTRAN), carry/mod.rs:428-463
        HIP_TRY(hipStreamSynchronize(stream_));
        std::vector<int32_t> basis(basis_columns, basis_columns + m_);
        std::vector<uint8_t> flags(n_alloc_, 0);
        for (int32_t v : basis) {
            if (v < 0 || v >= n_provider_) return fail(RELP_E_ARG, "from_basis: column out of range");
            if (flags[v]) return fail(RELP_E_SINGULAR, "from_basis: duplicate column");
            flags[v] = 1;
        }
        nr_artificial_ = 0; phase_ = 2;
        HIP_TRY(hipMemcpy(d_basis_, basis.data(), sizeof(int32_t) * m_, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_in_basis_, flags.data(), flags.size(), hipMemcpyHostToDevice));
        since_flush_ = 1;                              // force: the factors are stale
        relp_status_t st = lu_refactor();
        if (st) return st;
        std::vector<double> w(ld_b_, 0.0), b(m_);
        for (int32_t i = 0; i < m_; ++i) if (basis[i] < nr_normal_) w[i] = -cost_h_[basis[i]];
        HIP_TRY(hipMemcpy(d_w_, w.data(), sizeof(double) * m_, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_aq_, rhs_h_.data(), sizeof(double) * m_, hipMemcpyHostToDevice));
        if (ft_) {
            launch_ft_ftran(dlu_, fts_, ft_problem(0), -1, d_aq_, d_b_, stream_);
            launch_ft_btran(dlu_, fts_, ft_problem(0), -1, d_w_, d_minus_pi_, stream_);
        } else {
            launch_lu_ftran(dlu_, d_aq_, d_b_, d_lu_scratch_, nullptr, stream_);
            launch_lu_btran(dlu_, deferred(), d_w_, -1, d_minus_pi_, d_lu_scratch_, nullptr, stream_);
        }
        HIP_TRY(hipMemcpyAsync(b.data(), d_b_, sizeof(double) * m_, hipMemcpyDeviceToHost, stream_));
        if ((st = download_rec())) return st;
        double objective = 0.0;
        for (int32_t i = 0; i < m_; ++i) if (basis[i] < nr_normal_) objective += b[i] * cost_h_[basis[i]];
        h_rec_->minus_objective = -objective; h_rec_->outcome = DEV_RUNNING; h_rec_->last_selected = -1; h_rec_->phase = 2;
        return upload_rec();
    }
    enqueue_flush();                                   // leaves the deferred state empty
    HIP_TRY(hipStreamSynchronize(stream_));
    std::vector<int32_t> basis(basis_columns, basis_columns + m_);
    std::vector<double> Bn((size_t)m_ * ld_b_, 0.0), b(m_, 0.0), minus_pi(ld_b_, 0.0);
    double objective = 0.0;
    bool unit_basis = true, on_device = false;
    for (int32_t i = 0; i < m_; ++i) {
        const int32_t p = basis[i];
        if (p < 0 || p >= n_provider_) return fail(RELP_E_ARG, "from_basis: column out of range");
        if (p < nr_normal_ || vrow1_h_[p - nr_normal_] >= 0) unit_basis = false;
    }
    if (unit_basis) {
        // slack basis (two_phase/mod.rs:103-111 `FullInitialBasis`): the inverse is a signed permutation
        std::vector<uint8_t> seen(m_, 0);
        for (int32_t i = 0; i < m_; ++i) {
            const int32_t v = basis[i] - nr_normal_;
            const int32_t row = vrow0_h_[v];
            if (row < 0) return fail(RELP_E_SINGULAR, "from_basis: empty column in the basis");
            if (seen[row]) return fail(RELP_E_SINGULAR, "from_basis: duplicate pivot row");
            seen[row] = 1;
            // column i of B is sign * e_row  =>  row i of B^-1 is sign * e_row'
            Bn[(size_t)i * ld_b_ + row] = (double)vsign_h_[v];
            b[i] = (double)vsign_h_[v] * rhs_h_[row];
        }
    } else {
        // any basis: factorise it on the host (relp_lu.cpp) and form the rows of B^-1 by unit BTRANs,
        // O(m (m + nnz(L) + nnz(U))) - the one-off `BasisInverseRows::invert` of the reference
        // (basis_inverse_rows.rs:103-129: LU-invert, then m unit solves)
        std::vector<std::vector<std::pair<int32_t, double>>> cols(m_);
        std::vector<double> colbuf(std::max(mc_, 1));
        for (int32_t i = 0; i < m_; ++i) {
            const int32_t p = basis[i];
            auto& c = cols[i];
            if (p < nr_normal_) {
                HIP_TRY(hipMemcpy(colbuf.data(), dA_ + (int64_t)(p - col_lo_) * ld_a_, sizeof(double) * mc_, hipMemcpyDeviceToHost));
                for (int32_t r = 0; r < mc_; ++r) if (colbuf[r] != 0.0) c.emplace_back(r, colbuf[r]);
                if (bound_row_h_[p] >= 0) c.emplace_back(bound_row_h_[p], 1.0);
            } else {
                const int32_t v = p - nr_normal_;
                if (vrow0_h_[v] >= 0) c.emplace_back(vrow0_h_[v], (double)vsign_h_[v]);
                if (vrow1_h_[v] >= 0) c.emplace_back(vrow1_h_[v], 1.0);
            }
        }
        std::string msg;
        if (!lu_factor(m_, cols, &hlu_, &msg)) return fail(RELP_E_SINGULAR, "from_basis: " + msg);
        // the factorisation is host work (like every refactorisation); the m unit solves, b = B^-1 rhs and
        // -pi = -(c_B' B^-1) run on the device: row i of B^-1 is the BTRAN of e_i, written in place
        relp_status_t st = lu_upload_factors();
        if (st) return st;
        if (!d_lu_scratch_) HIP_TRY(dev_alloc(&d_lu_scratch_, ld_b_));
        HIP_TRY(hipMemsetAsync(dBinv_, 0, sizeof(double) * (size_t)m_ * ld_b_, stream_));
        DeferredUpdate none = deferred();
        none.kmax = 0;
        launch_lu_btran_rows(dlu_, none, dBinv_, ld_b_, d_lu_scratch_, stream_);
        std::vector<double> w(ld_b_, 0.0);
        for (int32_t i = 0; i < m_; ++i) if (basis[i] < nr_normal_) w[i] = cost_h_[basis[i]];
        HIP_TRY(hipMemcpyAsync(d_w_, w.data(), sizeof(double) * ld_b_, hipMemcpyHostToDevice, stream_));
        HIP_TRY(hipMemcpyAsync(d_aq_, rhs_h_.data(), sizeof(double) * m_, hipMemcpyHostToDevice, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));                     // w, rhs_h_ are host buffers
        launch_weighted_column_sums(dBinv_, ld_b_, m_, d_w_, d_minus_pi_, stream_);          // -pi, carry/mod.rs:214-248
        if ((st = download_rec())) return st;
        h_rec_->outcome = DEV_RUNNING;
        if ((st = upload_rec())) return st;
        launch_ftran(dBinv_, ld_b_, m_, 0, m_, d_aq_, d_b_, 0, d_rec_, stream_);            // b = B^-1 rhs
        HIP_TRY(hipMemcpyAsync(b.data(), d_b_, sizeof(double) * m_, hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
        for (int32_t i = 0; i < m_; ++i) objective += w[i] * b[i];
        on_device = true;
    }
    if (!on_device) {
        HIP_TRY(hipMemcpy(dBinv_, Bn.data(), Bn.size() * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_b_, b.data(), sizeof(double) * m_, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_minus_pi_, minus_pi.data(), sizeof(double) * ld_b_, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemcpy(d_basis_, basis.data(), sizeof(int32_t) * m_, hipMemcpyHostToDevice));
    std::vector<uint8_t> flags(n_alloc_, 0);
    for (int32_t v : basis) flags[v] = 1;
    HIP_TRY(hipMemcpy(d_in_basis_, flags.data(), flags.size(), hipMemcpyHostToDevice));
    nr_artificial_ = 0; phase_ = 2;
    relp_status_t st = download_rec();
    if (st) return st;
    h_rec_->minus_objective = -objective; h_rec_->outcome = DEV_RUNNING; h_rec_->last_selected = -1; h_rec_->phase = 2;
    return upload_rec();
}

// ------------------------------------------------------------------------------------------------
// Getters
// ------------------------------------------------------------------------------------------------
relp_status_t Engine::get_objective(double* out) {
    relp_status_t st = download_rec();
    if (st) return st;
    *out = -h_rec_->minus_objective;
    return RELP_OK;
}

relp_status_t Engine::get_vector(int which, double* out) {
    if (tableau_ && which == 1) {
        if (cfg_.shard_count > 1) return fail(RELP_E_UNSUPPORTED, "-pi of the sharded tableau: d is column-sharded");
        // -pi_k = d_j - c_j for the stored column j that was e_k originally
        std::vector<double> d(n_store_);
        HIP_TRY(hipMemcpyAsync(d.data(), d_d_, sizeof(double) * n_store_, hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
        for (int32_t k = 0; k < m_; ++k) out[k] = d[idcol_h_[k]] - cost_store_h_[idcol_h_[k]];
        return RELP_OK;
    }
    const double* src = which == 0 ? d_b_ : which == 1 ? d_minus_pi_ : d_alpha_;
    HIP_TRY(hipMemcpyAsync(out, src, sizeof(double) * m_, hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    return RELP_OK;
}

relp_status_t Engine::get_basis_indices(int32_t* out) {
    HIP_TRY(hipMemcpyAsync(out, d_basis_, sizeof(int32_t) * m_, hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    return RELP_OK;
}

relp_status_t Engine::get_basis_inverse(double* out) {
    if (cfg_.shard_count > 1) return fail(RELP_E_UNSUPPORTED, "B^-1 is row-sharded");
    if (lu_) {
        // row i of B^-1 = BTRAN of e_i (with the pending W); test / debugging path
        double* tmp = nullptr;
        HIP_TRY(dev_alloc(&tmp, (int64_t)m_ * m_));
        for (int32_t i = 0; i < m_; ++i) {
            if (ft_) launch_ft_btran(dlu_, fts_, ft_problem(0), i, nullptr, tmp + (int64_t)i * m_, stream_);
            else launch_lu_btran(dlu_, deferred(), nullptr, i, tmp + (int64_t)i * m_, d_lu_scratch_, nullptr, stream_);
        }
        HIP_TRY(hipMemcpyAsync(out, tmp, sizeof(double) * m_ * m_, hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
        HIP_TRY(hipFree(tmp));
        return RELP_OK;
    }
    enqueue_flush();
    if (tableau_) {
        // B^-1 = the tableau columns of the original identity columns
        double* tmp = nullptr;
        HIP_TRY(dev_alloc(&tmp, (int64_t)m_ * m_));
        launch_tab_gather_columns(tview(), d_idcol_, tmp, stream_);
        HIP_TRY(hipMemcpyAsync(out, tmp, sizeof(double) * m_ * m_, hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
        HIP_TRY(hipFree(tmp));
        return RELP_OK;
    }
    HIP_TRY(hipStreamSynchronize(stream_));
    HIP_TRY(hipMemcpy2D(out, sizeof(double) * m_, dBinv_, sizeof(double) * ld_b_, sizeof(double) * m_, m_, hipMemcpyDeviceToHost));
    return RELP_OK;
}

relp_status_t Engine::current_bfs(int32_t* cols, double* vals, int32_t cap, int32_t* count) {
    std::vector<double> b(m_); std::vector<int32_t> basis(m_);
    relp_status_t st = get_vector(0, b.data());
    if (st) return st;
    if ((st = get_basis_indices(basis.data()))) return st;
    std::vector<std::pair<int32_t, double>> t;
    for (int32_t i = 0; i < m_; ++i) if (b[i] != 0.0) t.emplace_back(basis[i], b[i]);
    std::sort(t.begin(), t.end(), [](auto& a, auto& c) { return a.first < c.first; });
    int32_t k = 0;
    for (auto& e : t) { if (k < cap) { cols[k] = e.first; vals[k] = e.second; } ++k; }
    if (count) *count = k;
    return RELP_OK;
}

relp_status_t Engine::get_iterations(int64_t* out) {
    relp_status_t st = download_rec();
    if (st) return st;
    *out = h_rec_->iterations;
    return RELP_OK;
}

relp_status_t Engine::get_degenerate_pivots(int64_t* out) {
    relp_status_t st = download_rec();
    if (st) return st;
    *out = h_rec_->degenerate;
    return RELP_OK;
}

relp_status_t Engine::get_trace(int32_t* phase, int32_t* entering, int32_t* row, int32_t* leaving, int64_t cap,
                                int64_t* count) {
    relp_status_t st = download_rec();
    if (st) return st;
    const int64_t n = std::min<int64_t>(h_rec_->iterations, trace_cap_);
    const int64_t k = std::min(n, cap);
    int32_t* outs[4] = {phase, entering, row, leaving};
    for (int f = 0; f < 4; ++f)
        if (outs[f] && k > 0) HIP_TRY(hipMemcpy(outs[f], d_trace_ + f * trace_cap_, sizeof(int32_t) * k, hipMemcpyDeviceToHost));
    if (count) *count = n;
    return RELP_OK;
}

// tableau/mod.rs:253-289: regenerate every basis column (must be e_i), basic reduced costs (0), b >= 0
relp_status_t Engine::check_basis(double* max_identity_error, double* max_basic_cost, double* min_b) {
    std::vector<int32_t> basis(m_);
    relp_status_t st = get_basis_indices(basis.data());
    if (st) return st;
    std::vector<double> col(m_), d(nr_columns()), b(m_);
    double e1 = 0.0, e2 = 0.0, mb = std::numeric_limits<double>::infinity();
    for (int32_t i = 0; i < m_; ++i) {
        if (basis[i] >= kWrappedArtificialBase) continue;       // no column to regenerate
        if ((st = generate_column(basis[i], col.data()))) return st;
        for (int32_t k = 0; k < m_; ++k) e1 = std::max(e1, std::fabs(col[k] - (k == i ? 1.0 : 0.0)));
    }
    if ((st = relative_costs(d.data()))) return st;
    for (int32_t i = 0; i < m_; ++i) if (basis[i] < kWrappedArtificialBase) e2 = std::max(e2, std::fabs(d[basis[i]]));
    if ((st = get_vector(0, b.data()))) return st;
    for (double v : b) mb = std::min(mb, v);
    if (max_identity_error) *max_identity_error = e1;
    if (max_basic_cost) *max_basic_cost = e2;
    if (min_b) *min_b = mb;
    return RELP_OK;
}

}  // namespace relp
