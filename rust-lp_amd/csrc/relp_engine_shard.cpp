// relp_engine_shard.cpp -- Engine: the entry points of the sharded (multi-GPU) engines.
#include "relp_engine_internal.hpp"

namespace relp {

// ------------------------------------------------------------------------------------------------
// Shards.  Everything is enqueued on stream_; the caller interleaves the
// RCCL collectives on the same stream, so there is no host sync inside a pivot.
// ------------------------------------------------------------------------------------------------
void Engine::shard_ranges(int32_t* col_lo, int32_t* col_hi, int32_t* row_lo, int32_t* row_hi, int32_t* stride) const {
    if (col_lo) *col_lo = col_lo_;
    if (col_hi) *col_hi = col_hi_;
    if (row_lo) *row_lo = row_lo_;
    if (row_hi) *row_hi = row_hi_;
    if (stride) *stride = row_stride_;
}

// Tableau engine, one pivot after the candidates were exchanged: ratio test (replicated), row update
// of the owned columns, W / b / basis update (replicated); the flush is local to the owned columns.
relp_status_t Engine::shard_pivot() {
    if (lu_) return fail(RELP_E_UNSUPPORTED, "the LU engine is not sharded");
    if (!tableau_) return fail(RELP_E_STATE, "relp_shard_pivot is the tableau engine's step");
    const int rule = phase_ == 1 ? cfg_.phase_one_rule : cfg_.phase_two_rule;
    const TableauView tv = tview();
    const DeferredUpdate du = deferred();
    const SelectPartials sp = tab_partials(rule);
    // the ratio test ran in relp_shard_select_column; tableau row / reduced costs / PRICE partials of the
    // owned columns and W, b, basis (replicated) in one launch
    prof_begin(RELP_K_PRICE);
    launch_tab_update_all(tv, du, sp, m_, d_alpha_, d_b_, d_basis_, d_in_basis_, d_trace_, trace_cap_, d_rec_, stream_);
    prof_end();
    tab_partials_valid_ = true;
    if (++since_flush_ >= block_) enqueue_flush();
    ++prof_tick_;
    return RELP_OK;
}

relp_status_t Engine::shard_price(double* dev_candidate) {
    if (lu_) return fail(RELP_E_UNSUPPORTED, "the LU engine is not sharded");
    const int rule = phase_ == 1 ? cfg_.phase_one_rule : cfg_.phase_two_rule;
    if (tableau_) {
        // local PRICE result = the partial argmin the last row update left behind (or a scan of d);
        // the candidate message carries the tableau column alpha itself
        const TableauView tv = tview();
        const SelectPartials sp = tab_partials(rule);
        if (!tab_partials_valid_) { launch_tab_scan(tv, sp, d_rec_, stream_); tab_partials_valid_ = true; }
        // PRICE's final reduction over the local partials + the local winner's tableau column, written
        // straight into the candidate message
        prof_begin(RELP_K_FTRAN);
        launch_tab_select_column_msg(tv, deferred(), sp, tab_scan_blocks(sc_hi_ - sc_lo_), dev_candidate, d_b_, tolerances(), d_rec_,
                                     stream_);
        prof_end();
        return RELP_OK;
    }
    const double* A = dA_ - (int64_t)col_lo_ * ld_a_;
    prof_begin(RELP_K_PRICE);
    enqueue_price(phase_, d_minus_pi_, d_rec_, col_lo_, col_hi_);
    prof_end();
    prof_begin(RELP_K_SELECT_COLUMN);
    launch_select_column(d_d_, d_in_basis_, nr_columns(), rule, cfg_.tol_cost, cfg_.tol_tie, d_rec_, stream_);
    prof_end();
    prof_begin(RELP_K_BUILD_COLUMN);
    launch_build_column(A, ld_a_, table(), m_, d_aq_, d_rec_, stream_);
    launch_pack_candidate(d_aq_, m_, dev_candidate, d_rec_, stream_);
    prof_end();
    return RELP_OK;
}

relp_status_t Engine::shard_select_column(const double* dev_candidates, int32_t count) {
    if (lu_) return fail(RELP_E_UNSUPPORTED, "the LU engine is not sharded");
    const int rule = phase_ == 1 ? cfg_.phase_one_rule : cfg_.phase_two_rule;
    if (tableau_) {
        // the winner's payload is the entering tableau column (alpha) itself: pick it and run the ratio test
        if (count > 64) return fail(RELP_E_UNSUPPORTED, "at most 64 shards");
        prof_begin(RELP_K_RATIO);
        launch_tab_select_candidate_ratio(dev_candidates, count, cand_len_, m_, d_alpha_, d_b_, d_basis_, rule, tolerances(),
                                          deferred(), -1, d_rec_, stream_);
        prof_end();
        return RELP_OK;
    }
    launch_select_candidate(dev_candidates, count, cand_len_, m_, d_aq_, rule, cfg_.tol_tie, d_rec_, stream_);
    return RELP_OK;
}

relp_status_t Engine::shard_ftran(double* dev_alpha_slice) {
    if (lu_) return fail(RELP_E_UNSUPPORTED, "the LU engine is not sharded");
    double* Binv = dBinv_ - (int64_t)row_lo_ * ld_b_;
    prof_begin(RELP_K_FTRAN);
    launch_ftran(Binv, ld_b_, m_, row_lo_, row_hi_, d_aq_, dev_alpha_slice, row_lo_, d_rec_, stream_);
    launch_pad_slice(dev_alpha_slice, row_hi_ - row_lo_, row_stride_, stream_);
    prof_end();
    return RELP_OK;
}

relp_status_t Engine::shard_ratio(const double* dev_alpha_slices, int32_t count, double* dev_rho) {
    if (lu_) return fail(RELP_E_UNSUPPORTED, "the LU engine is not sharded");
    double* Binv = dBinv_ - (int64_t)row_lo_ * ld_b_;
    if (block_ == 0) {
        prof_begin(RELP_K_RATIO);
        launch_gather_alpha(dev_alpha_slices, count, row_stride_, m_, d_alpha_, d_rec_, stream_);
        launch_ratio(d_alpha_, d_b_, d_basis_, m_, tolerances(), d_rec_, stream_);
        launch_compute_rho(Binv, ld_b_, m_, row_lo_, row_hi_, dev_rho, d_rec_, stream_);
        prof_end();
        return RELP_OK;
    }
    // deferred: the slices hold v = B0inv a_q; W is replicated, so every rank forms the full alpha,
    // updates its copy of W and contributes the rows of B0inv it owns to rho (SUM over ranks).
    const DeferredUpdate du = deferred();
    prof_begin(RELP_K_APPLY_W);
    launch_gather_alpha(dev_alpha_slices, count, row_stride_, m_, d_v_, d_rec_, stream_);
    launch_apply_w(du, m_, d_v_, d_alpha_, d_rec_, stream_);
    prof_end();
    prof_begin(RELP_K_RATIO);
    launch_ratio(d_alpha_, d_b_, d_basis_, m_, tolerances(), d_rec_, stream_);
    prof_end();
    prof_begin(RELP_K_UPDATE_W);
    launch_eta_prepare(du, d_rec_, stream_);
    launch_update_w(du, m_, d_alpha_, d_rec_, stream_);
    launch_rho_deferred(du, Binv, ld_b_, m_, row_lo_, row_hi_, dev_rho, d_rec_, stream_);
    prof_end();
    return RELP_OK;
}

relp_status_t Engine::shard_flush_begin(double** dev_snapshot, int64_t* len) {
    if (lu_) return fail(RELP_E_UNSUPPORTED, "the LU engine is not sharded");
    if (len) *len = 0;
    if (block_ == 0) return RELP_OK;
    double* Binv = dBinv_ - (int64_t)row_lo_ * ld_b_;
    prof_begin(RELP_K_FLUSH);
    launch_flush_snapshot(deferred(), Binv, ld_b_, row_lo_, row_hi_, d_rec_, stream_);
    prof_end();
    if (dev_snapshot) *dev_snapshot = d_R_;
    if (len) *len = ld_b_ * block_;
    return RELP_OK;
}

relp_status_t Engine::shard_flush_end() {
    if (lu_) return fail(RELP_E_UNSUPPORTED, "the LU engine is not sharded");
    if (block_ == 0) return RELP_OK;
    const DeferredUpdate du = deferred();
    double* Binv = dBinv_ - (int64_t)row_lo_ * ld_b_;
    prof_begin(RELP_K_FLUSH);
    launch_flush_apply(du, Binv, ld_b_, m_, row_lo_, row_hi_, d_rec_, stream_);
    launch_flush_reset(du, d_rec_, stream_);
    prof_end();
    since_flush_ = 0;
    return RELP_OK;
}

relp_status_t Engine::shard_update(const double* dev_rho) {
    if (lu_) return fail(RELP_E_UNSUPPORTED, "the LU engine is not sharded");
    double* Binv = dBinv_ - (int64_t)row_lo_ * ld_b_;
    prof_begin(RELP_K_UPDATE_VECTORS);
    launch_update_vectors(m_, d_alpha_, dev_rho, d_b_, d_minus_pi_, d_basis_, d_in_basis_, d_trace_, trace_cap_, d_rec_,
                          stream_);
    prof_end();
    if (block_ == 0) {
        prof_begin(RELP_K_UPDATE_INVERSE);
        launch_update_inverse(Binv, ld_b_, m_, row_lo_, row_hi_, d_alpha_, dev_rho, d_rec_, stream_);
        prof_end();
    } else {
        ++since_flush_;
    }
    return RELP_OK;
}

// ------------------------------------------------------------------------------------------------
// Native loop over the shard steps: the library enqueues kernels and collectives itself.
// ------------------------------------------------------------------------------------------------
// phase_one.rs:223-260 for the sharded tableau engine.  The basis is replicated, so every rank walks the same
// sorted list of basic artificial variables; for each one the ranks exchange their first eligible column of the
// artificial's row through the same candidate message and all-gather as a PRICE step, the lowest column wins on
// every rank and enters in that row at zero level.  No eligible column anywhere: the row is redundant.
relp_status_t Engine::remove_artificial_basis_variables_sharded(std::vector<int32_t>& rows_to_remove) {
    HIP_TRY(hipStreamSynchronize(stream_));
    std::vector<int32_t> basis(m_);
    HIP_TRY(hipMemcpy(basis.data(), d_basis_, sizeof(int32_t) * m_, hipMemcpyDeviceToHost));
    std::vector<int32_t> arts;
    for (int32_t v : basis) if (v < nr_artificial_) arts.push_back(v);
    if (arts.empty()) return RELP_OK;
    if (!coll_allgather_)
        return fail(RELP_E_STATE, "artificial variables are still basic after phase 1: the sharded engine pivots them out "
                                  "through the collective hooks (relp_shard_set_collectives / relp_rccl_attach)");
    std::sort(arts.begin(), arts.end());
    relp_status_t st;
    const int32_t g = std::max(cfg_.shard_count, 1);
    const bool textbook = cfg_.artificial_removal == RELP_ARTIFICIAL_TEXTBOOK;
    for (int32_t a : arts) {
        int32_t pivot_row = column_to_row_[a];             // phase_one.rs:236: the row the artificial STARTED in
        if (textbook) pivot_row = (int32_t)(std::find(basis.begin(), basis.end(), a) - basis.begin());   // the row it is basic in
        if ((st = download_rec())) return st;
        h_rec_->outcome = DEV_RUNNING;
        if ((st = upload_rec())) return st;
        const TableauView tv = tview();
        const DeferredUpdate du = deferred();
        const SelectPartials sp = tab_partials(RELP_RULE_FIRST_PROFITABLE);        // key = column index
        Tolerances zt = tolerances();
        if (textbook) zt.cost = INFINITY;               // any reduced cost will do: the pivot is at zero level
        launch_tab_zero_level_scan(tv, du, sp, pivot_row, nr_artificial_, zt, d_rec_, stream_);
        launch_tab_select_column_msg(tv, du, sp, tab_scan_blocks(sc_hi_ - sc_lo_), d_msg_cand_, d_b_, tolerances(), d_rec_, stream_);
        if (coll_allgather_(coll_ctx_, d_msg_cand_, d_msg_cands_, cand_len_ * (int64_t)sizeof(double), stream_))
            return fail(RELP_E_HIP, "all-gather of the zero-level candidates failed");
        launch_tab_select_candidate_ratio(d_msg_cands_, g, cand_len_, m_, d_alpha_, d_b_, d_basis_, RELP_RULE_FIRST_PROFITABLE,
                                          tolerances(), du, pivot_row, d_rec_, stream_);
        if ((st = download_rec())) return st;
        if (h_rec_->outcome == DEV_NO_CANDIDATE) {      // (textbook: the artificial's own row; remove_rows moves it there first)
            if (textbook) { stuck_artificials_.push_back(a); rows_to_remove.push_back(column_to_row_[a]); }
            else rows_to_remove.push_back(a);
            continue;
        }
        if (h_rec_->alpha_r == 0.0) return fail(RELP_E_ZERO_PIVOT, "Pivot value can't be zero.");
        launch_tab_update_all(tv, du, sp, m_, d_alpha_, d_b_, d_basis_, d_in_basis_, d_trace_, trace_cap_, d_rec_, stream_);
        tab_partials_valid_ = false;
        basis[pivot_row] = h_rec_->q;
        if (++since_flush_ >= block_) enqueue_flush();
    }
    return RELP_OK;
}

relp_status_t Engine::shard_set_collectives(relp_allgather_fn ag, relp_allreduce_sum_fn ar, void* ctx) {
    if (lu_) return fail(RELP_E_UNSUPPORTED, "the LU engine is not sharded");
    if (!ag || (!tableau_ && !ar)) return fail(RELP_E_ARG, "collective hooks missing");
    coll_allgather_ = ag; coll_allreduce_ = ar; coll_ctx_ = ctx;
    const int64_t g = std::max(cfg_.shard_count, 1);
    if (!d_msg_cand_) {
        HIP_TRY(dev_alloc(&d_msg_status_, 2));
        HIP_TRY(dev_alloc(&d_msg_statuses_, 2 * g));
        HIP_TRY(dev_alloc(&d_msg_cand_, cand_len_));
        HIP_TRY(dev_alloc(&d_msg_cands_, cand_len_ * g));
        if (!tableau_) {
            HIP_TRY(dev_alloc(&d_msg_slice_, row_stride_));
            HIP_TRY(dev_alloc(&d_msg_slices_, (int64_t)row_stride_ * g));
            HIP_TRY(dev_alloc(&d_msg_rho_, rho_len()));
        }
    }
    return RELP_OK;
}

// one pivot: the steps of rust-lp_amd/sharded.py `_iteration`, same order, same buffers
relp_status_t Engine::shard_iteration() {
    relp_status_t st;
    const int32_t g = std::max(cfg_.shard_count, 1);
    coll_step_ = 0;
    shadow_flush_ = since_flush_;
    if (inject_failure_after_ == 0) { inject_failure_after_ = -1; return fail(RELP_E_STATE, "injected failure (relp_shard_inject_failure)"); }
    if (inject_failure_after_ > 0) --inject_failure_after_;
    auto broken = [&](const char* what) { coll_broken_ = true; return fail(RELP_E_HIP, what); };
    if (tableau_ && fused_update_ && in_loop_) {
        // two launches around the all-gather: [local PRICE winner + its tableau column into the message] -> gather ->
        // [winner among the messages + ratio test in every workgroup + the update] (relp_kernels.h: launch_tab_ratio_update_all)
        if (g > 64) return fail(RELP_E_UNSUPPORTED, "at most 64 shards");
        const int rule = phase_ == 1 ? cfg_.phase_one_rule : cfg_.phase_two_rule;
        const TableauView tv = tview();
        const DeferredUpdate du = deferred();
        const SelectPartials sp = tab_partials(rule);
        if (!tab_partials_valid_) { launch_tab_scan(tv, sp, d_rec_, stream_); tab_partials_valid_ = true; }
        prof_begin(RELP_K_FTRAN);
        launch_tab_select_column_msg(tv, du, sp, tab_scan_blocks(sc_hi_ - sc_lo_), d_msg_cand_, d_b_, tolerances(), d_rec_, stream_,
                                     d_shadow_, d_shadow_meta_);
        prof_end();
        if (coll_allgather_(coll_ctx_, d_msg_cand_, d_msg_cands_, cand_len_ * (int64_t)sizeof(double), stream_))
            return broken("all-gather of the PRICE candidates failed");
        ++coll_step_;
        prof_begin(RELP_K_PRICE);
        launch_tab_ratio_update_all(tv, du, sp, m_, nullptr, d_b_, d_b_alt_, d_basis_, d_basis_alt_, d_in_basis_, d_trace_,
                                    trace_cap_, tolerances(), nullptr, d_shadow_, d_shadow_meta_, d_rec_, stream_, d_msg_cands_, g,
                                    cand_len_, rule);
        prof_end();
        std::swap(d_b_, d_b_alt_);
        std::swap(d_basis_, d_basis_alt_);
        shadow_pending_ = true;
        if (++since_flush_ >= block_) enqueue_flush();
        ++prof_tick_;
        return RELP_OK;
    }
    if ((st = shard_price(d_msg_cand_))) return st;
    if (coll_allgather_(coll_ctx_, d_msg_cand_, d_msg_cands_, cand_len_ * (int64_t)sizeof(double), stream_))
        return broken("all-gather of the PRICE candidates failed");
    ++coll_step_;
    if ((st = shard_select_column(d_msg_cands_, g))) return st;
    if (tableau_) return shard_pivot();
    if ((st = shard_ftran(d_msg_slice_))) return st;
    if (coll_allgather_(coll_ctx_, d_msg_slice_, d_msg_slices_, row_stride_ * (int64_t)sizeof(double), stream_))
        return broken("all-gather of the FTRAN slices failed");
    ++coll_step_;
    if ((st = shard_ratio(d_msg_slices_, g, d_msg_rho_))) return st;
    if (coll_allreduce_(coll_ctx_, d_msg_rho_, rho_len(), stream_)) return broken("all-reduce of the pivot row failed");
    ++coll_step_;
    if ((st = shard_update(d_msg_rho_))) return st;
    if (block_ > 0 && since_flush_ >= block_) {
        double* snap = nullptr; int64_t len = 0;
        if ((st = shard_flush_begin(&snap, &len))) return st;
        if (len > 0) {
            if (coll_allreduce_(coll_ctx_, snap, len, stream_)) return broken("all-reduce of the flush snapshot failed");
            ++coll_step_;
            if ((st = shard_flush_end())) return st;
        }
        since_flush_ = 0;
    }
    return RELP_OK;
}

// After a LOCAL failure (a kernel launch, an allocation: anything but a collective) this rank keeps the collective sequence
// of the pivots alive with whatever is in its buffers, so that the other ranks are not left waiting inside a collective; at
// the next poll every rank learns of the failure (shard_agree_on_status) and all leave relp_shard_run together.
// `from_step`: the collectives of this pivot with a lower index have happened already (the pivot that failed).
relp_status_t Engine::shard_iteration_comm_only(int from_step) {
    auto broken = [&](const char* what) { coll_broken_ = true; return fail(RELP_E_HIP, what); };
    if (from_step <= 0 && coll_allgather_(coll_ctx_, d_msg_cand_, d_msg_cands_, cand_len_ * (int64_t)sizeof(double), stream_)) return broken("all-gather failed");
    if (tableau_) return RELP_OK;                          // (its flush is local)
    if (from_step <= 1 && coll_allgather_(coll_ctx_, d_msg_slice_, d_msg_slices_, row_stride_ * (int64_t)sizeof(double), stream_)) return broken("all-gather failed");
    if (from_step <= 2 && coll_allreduce_(coll_ctx_, d_msg_rho_, rho_len(), stream_)) return broken("all-reduce failed");
    ++shadow_flush_;
    if (block_ > 0 && shadow_flush_ >= block_) {
        if (from_step <= 3 && coll_allreduce_(coll_ctx_, d_R_, ld_b_ * block_, stream_)) return broken("all-reduce failed");
        shadow_flush_ = 0;
    }
    return RELP_OK;
}

// One 16-byte all-gather of (status, rank): every rank returns the same verdict.
relp_status_t Engine::shard_agree_on_status(relp_status_t local) {
    if (coll_broken_) return local ? local : RELP_E_HIP;
    const int32_t g = std::max(cfg_.shard_count, 1);
    const double mine[2] = {(double)local, (double)cfg_.shard_rank};
    if (hipMemcpyAsync(d_msg_status_, mine, sizeof(mine), hipMemcpyHostToDevice, stream_) != hipSuccess ||
        hipStreamSynchronize(stream_) != hipSuccess)
        return local ? local : fail(RELP_E_HIP, "status upload failed");
    if (coll_allgather_(coll_ctx_, d_msg_status_, d_msg_statuses_, 2 * (int64_t)sizeof(double), stream_)) {
        coll_broken_ = true;
        return local ? local : fail(RELP_E_HIP, "all-gather of the rank statuses failed");
    }
    std::vector<double> all(2 * (size_t)g, 0.0);
    if (hipMemcpyAsync(all.data(), d_msg_statuses_, sizeof(double) * all.size(), hipMemcpyDeviceToHost, stream_) != hipSuccess ||
        hipStreamSynchronize(stream_) != hipSuccess)
        return local ? local : fail(RELP_E_HIP, "status download failed");
    for (int32_t r = 0; r < g; ++r)
        if (all[2 * r] != 0.0) {
            if (local) return local;                               // this rank's own error text stays
            return fail((relp_status_t)(int)all[2 * r], "rank " + std::to_string(r) + " of the sharded loop failed; all ranks stop");
        }
    return RELP_OK;
}

relp_status_t Engine::shard_run(int64_t max_iters, int64_t* done, int32_t* outcome) {
    if (!coll_allgather_) return fail(RELP_E_STATE, "relp_shard_run needs relp_shard_set_collectives / relp_rccl_attach first");
    int32_t oc = RELP_RUNNING; int64_t start = 0, it = 0;
    relp_status_t local = RELP_OK;
    relp_status_t st = poll(&oc, &start);
    if (st) return st;
    it = start;
    // every rank polls after the same pivot counts (the outcome is replicated), so the ranks leave the
    // loop together and the collectives stay matched; phase 1 polls at 1, 2, 4, ... like relp_run.  A rank that fails
    // locally keeps the collectives of the chunk going and reports at the poll: nobody is left inside a collective.
    int64_t chunk = phase_ == 1 ? 1 : cfg_.poll_interval;
    for (int64_t left = max_iters; left > 0 && oc == RELP_RUNNING;) {
        const int64_t n = std::min(left, chunk);
        LoopScope loop(*this);                             // (ends before the poll: a phase boundary reads the settled tableau)
        for (int64_t k = 0; k < n; ++k) {
            if (!local) {
                local = shard_iteration();
                if (local && !coll_broken_ && (st = shard_iteration_comm_only(coll_step_))) return st;   // finish the failed pivot's exchanges
            } else if ((st = shard_iteration_comm_only(0))) return st;
            if (coll_broken_) return local;
        }
        left -= n;
        tab_settle();
        in_loop_ = false;
        if ((st = shard_agree_on_status(local))) return st;       // before the poll: a phase boundary inside it exchanges again
        if ((st = poll(&oc, &it))) return st;
        chunk = std::min<int64_t>(chunk * 2, cfg_.poll_interval);
    }
    if (hipGetLastError() != hipSuccess) return fail(RELP_E_HIP, "kernel launch failed");
    if (done) *done = it - start;
    if (outcome) *outcome = oc;
    return RELP_OK;
}

relp_status_t Engine::poll(int32_t* outcome, int64_t* iterations) {
    relp_status_t st = download_rec();
    if (st) return st;
    int32_t oc = RELP_RUNNING;
    if (h_rec_->outcome == DEV_NO_CANDIDATE) {
        if (phase_ == 2) oc = RELP_OPTIMAL;
        else if ((st = finish_phase_one(&oc))) return st;
    } else if (h_rec_->outcome == DEV_NO_ROW) oc = phase_ == 2 ? RELP_UNBOUNDED : RELP_NO_ROW_PHASE_ONE;
    if (iterations) *iterations = h_rec_->iterations;      // after the phase boundary: it may have pivoted at zero level
    if (outcome) *outcome = oc;
    return RELP_OK;
}

}  // namespace relp
