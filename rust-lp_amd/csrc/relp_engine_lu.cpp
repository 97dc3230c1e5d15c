// relp_engine_lu.cpp -- Engine: the sparse LU engine (RELP_ENGINE_LU): CSC upload, host refactorisation,
// one pivot.  See relp_engine.hpp and relp_lu.hpp.
#include "relp_engine_internal.hpp"

#include <algorithm>
#include <climits>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>

namespace relp {

static_assert(sizeof(EllPassHost) == 16 && sizeof(EllPass) == 16, "the packed pass header is one 16-byte load");

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

// The basis as it is on the device -> pinned host memory.  Synchronises the stream.
relp_status_t Engine::lu_download_basis() {
    if (!h_basis_) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h_basis_), sizeof(int32_t) * (size_t)std::max(m_alloc_rows_, m_), hipHostMallocDefault));
    HIP_TRY(hipMemcpyAsync(h_basis_, d_basis_, sizeof(int32_t) * m_, hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    return RELP_OK;
}

// the columns of the basis in h_basis_ from the host copy of the matrix (the LU engine never holds A densely)
relp_status_t Engine::lu_basis_columns(std::vector<std::vector<std::pair<int32_t, double>>>& cols) {
    const int32_t* const basis = h_basis_;
    cols.resize(m_);
    for (int32_t i = 0; i < m_; ++i) {
        const int32_t j = basis[i];
        auto& c = cols[i];
        c.clear();
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
    return RELP_OK;
}

// The same columns as one flat copy (column i = entries [ptr[i], ptr[i + 1])): what lu_factor_csc wants -- at 64,000 rows the
// vector per column was a cache miss per column and pass.
relp_status_t Engine::lu_basis_flat(std::vector<int64_t>& ptr, std::vector<int32_t>& idx, std::vector<double>& val) {
    const int32_t* const basis = h_basis_;
    ptr.assign((size_t)m_ + 1, 0);
    idx.clear(); val.clear();
    auto put = [&](int32_t row, double v) { idx.push_back(row); val.push_back(v); };
    for (int32_t i = 0; i < m_; ++i) {
        const int32_t j = basis[i];
        if (j < nr_artificial_) put(column_to_row_[j], 1.0);
        else if (j >= kWrappedArtificialBase) put(column_to_row_[wrapped_na_ - 1 - (INT32_MAX - j)], 1.0);
        else {
            const int32_t p = j - nr_artificial_;
            if (p < nr_normal_) {
                for (int64_t e = hc_ptr_[p]; e < hc_ptr_[p + 1]; ++e) put(hc_idx_[e], hc_val_[e]);
                if (bound_row_h_[p] >= 0) put(bound_row_h_[p], 1.0);
            } else {
                const int32_t v = p - nr_normal_;
                if (v >= nr_virtual_) return fail(RELP_E_STATE, "basis column out of range");
                if (vrow0_h_[v] >= 0) put(vrow0_h_[v], (double)vsign_h_[v]);
                if (vrow1_h_[v] >= 0) put(vrow1_h_[v], 1.0);
            }
        }
        ptr[(size_t)i + 1] = (int64_t)idx.size();
    }
    return RELP_OK;
}

// P B Q = L U on the host for the basis in h_basis_ (hlu_ is overwritten)
relp_status_t Engine::lu_factor_downloaded_basis() {
    if (!std::getenv("RELP_DUMP_BASIS")) {
        const relp_status_t fst = lu_basis_flat(basis_ptr_, basis_idx_, basis_val_);
        if (fst) return fst;
        std::string msg;
        if (!lu_factor_csc(m_, basis_ptr_.data(), basis_idx_.data(), basis_val_.data(), &hlu_, &msg)) return fail(RELP_E_SINGULAR, msg);
        return RELP_OK;
    }
    std::vector<std::vector<std::pair<int32_t, double>>>& cols = basis_cols_;     // (kept: no 790 allocations per refactorisation)
    const relp_status_t cst = lu_basis_columns(cols);
    if (cst) return cst;
    if (const char* dump = std::getenv("RELP_DUMP_BASIS")) {
        if (lu_refactors_ == 100) {                        // one mid-solve basis as text: m, then per column "n i v i v ..."
            if (FILE* f = std::fopen(dump, "w")) {
                std::fprintf(f, "%d\n", m_);
                for (auto& c : cols) { std::fprintf(f, "%zu", c.size()); for (auto& e : c) std::fprintf(f, " %d %.17g", e.first, e.second); std::fprintf(f, "\n"); }
                std::fclose(f);
            }
        }
    }
    std::string msg;
    if (!lu_factor(m_, cols, &hlu_, &msg)) return fail(RELP_E_SINGULAR, msg);
    return RELP_OK;
}

void Engine::lu_refactor_clock(std::chrono::steady_clock::time_point tb, std::chrono::steady_clock::time_point t0,
                               std::chrono::steady_clock::time_point t1, std::chrono::steady_clock::time_point t2) {
    refactor_us_[0] += std::chrono::duration<double, std::micro>(t0 - tb).count();
    refactor_us_[1] += std::chrono::duration<double, std::micro>(t1 - t0).count();
    refactor_us_[2] += std::chrono::duration<double, std::micro>(t2 - t1).count();
    if (std::getenv("RELP_DEBUG") && lu_refactors_ % 60 == 59)
        std::fprintf(stderr, "[relp] refactorisations so far %lld: basis download + columns %.0f us, lu_factor %.0f us, schedules + upload %.0f us (averages)\n",
                     (long long)lu_refactors_ + 1, refactor_us_[0] / (lu_refactors_ + 1), refactor_us_[1] / (lu_refactors_ + 1),
                     refactor_us_[2] / (lu_refactors_ + 1));
}

// Refactorisation (lower_upper/mod.rs:199-202 + carry/mod.rs:602-614): B from the current basis
// columns, P B Q = L U on the host, schedules to the device, W := empty.  Synchronises the stream.
relp_status_t Engine::lu_refactor() {
    const auto tb = std::chrono::steady_clock::now();
    relp_status_t st = RELP_OK;
    bool on_device = false;
    if (luf_enabled_) {                                    // P B Q = L U by the device kernel: no basis download, no host search
        int32_t dev = 0;
        st = lu_factor_on_device(&dev);
        if (st == RELP_OK) on_device = true;
        else if (st != RELP_E_UNSUPPORTED) return st;
        else ++luf_fallbacks_;
    }
    if (!on_device && (st = lu_download_basis())) return st;
    const auto t0 = std::chrono::steady_clock::now();
    if (!on_device && (st = lu_factor_downloaded_basis())) return st;
    const auto t1 = std::chrono::steady_clock::now();
    // (a device-resident factorisation has installed its schedules where the kernels built them: nothing to pack or upload)
    if (!(on_device && luf_is_resident()) && (st = lu_upload_factors())) return st;
    if (ft_) { if ((st = ft_reset())) return st; }
    else launch_flush_reset(deferred(), d_rec_, stream_);
    HIP_TRY(hipStreamSynchronize(stream_));
    lu_refactor_clock(tb, t0, t1, std::chrono::steady_clock::now());
    since_flush_ = 0;
    ++lu_refactors_;
    return RELP_OK;
}

// Refactorisation without stopping the pivot kernel (persistent-kernel mode).  The kernel has just returned with
// `max_updates - lookahead` updates pending.  Its basis is downloaded, the kernel is launched again on the OLD factors (it
// may fill the rest of the update file), and while it pivots the host factorises the downloaded basis into the other
// device buffer.  When both are done the new factors are installed and the basis changes made meanwhile (the kernel's
// journal) are applied to them as Forrest-Tomlin updates by k_ft_replay: the device pivots for the ~0.6 ms it used to
// wait.  If the kernel ended the phase meanwhile, the new factors are dropped: the old ones with their update file
// describe the final basis.
relp_status_t Engine::lu_refactor_lookahead(int rule, int64_t budget, bool have_basis) {
    const auto tb = std::chrono::steady_clock::now();
    relp_status_t st = have_basis ? RELP_OK : lu_download_basis();       // (have_basis: the kernel's own report held it)
    if (st) return st;
    FtState go = fts_;
    go.max_updates = std::max(go.max_updates, std::min(cfg_.update_block < 0 ? ft_tcap_ : cfg_.update_block, ft_tcap_));
    if (lu_pipeline_cap_ > 0) go.max_updates = std::min(lu_pipeline_cap_, ft_tcap_);       // (run_ft: RELP_LU_PIPELINE_SHORT)
    prof_tick_ = 0;
    prof_begin(RELP_K_FT_RUN);
    ft_enqueue_pivots(go, rule, budget);
    prof_end();
    const auto t0 = std::chrono::steady_clock::now();
    // the current factors stay valid (and in use) until the new ones are installed
    LUFactors old_h = std::move(hlu_);
    const DeviceLU old_d = dlu_;
    const FtState old_f = fts_;
    char* const old_buf = d_lu_buf_;
    const int64_t old_cap = lu_cap_;
    d_lu_buf_ = d_lu_buf_alt_; lu_cap_ = lu_cap_alt_;
    hlu_ = LUFactors{};
    auto restore = [&]() {
        d_lu_buf_alt_ = d_lu_buf_; lu_cap_alt_ = d_lu_buf_ ? lu_cap_ : 0;      // (null after a failed re-allocation)
        hlu_ = std::move(old_h); dlu_ = old_d; fts_ = old_f; d_lu_buf_ = old_buf; lu_cap_ = old_cap;
    };
    st = lu_factor_downloaded_basis();
    const auto t1 = std::chrono::steady_clock::now();
    if (!st) st = lu_upload_factors();                     // (its copy queues behind the kernel; it returns when both are done)
    if (st) {                                              // keep what works; the kernel's result decides what happens next
        restore();
        relp_status_t st2 = ft_read_report(nullptr);
        ft_need_refactor_ = true;
        return st2 ? st2 : st;
    }
    if ((st = ft_read_report(nullptr))) { restore(); return st; }
    const int32_t changes = h_ft_hdr_[3];
    if (h_rec_->outcome != DEV_RUNNING || h_ft_hdr_[2] == 2 || changes > ft_tcap_) {
        restore();                                         // phase over (or something off): nothing to install
        if (h_ft_hdr_[2] == 2 || changes > ft_tcap_) ft_need_refactor_ = true;
        return RELP_OK;
    }
    d_lu_buf_alt_ = old_buf; lu_cap_alt_ = old_cap;
    if ((st = ft_reset())) return st;
    prof_begin(RELP_K_FLUSH);
    launch_ft_replay(dlu_, fts_, ft_problem(rule), changes, stream_);
    prof_end();
    lu_refactor_clock(tb, t0, t1, std::chrono::steady_clock::now());
    since_flush_ = changes;
    ft_need_refactor_ = false;                             // (a replay that fails marks the header; the next launch returns at once)
    ++lu_refactors_;
    ++lu_lookahead_installs_;
    lu_replayed_changes_ += changes;
    return RELP_OK;
}

// hlu_ (host factors + level schedules) -> one device buffer, dlu_ points into it.  Also used by the revised
// engine's warm start, which forms the rows of B^-1 with the device BTRAN.
relp_status_t Engine::lu_upload_factors() {
    // pack everything into one buffer (16-byte aligned pieces): rowperm, colperm, then per schedule the
    // rows in solve order, the entry indices / values and the level offsets
    const TriangularSchedule* sch[4] = {&hlu_.Lf, &hlu_.Uf, &hlu_.Ub, &hlu_.Lb};
    // (assembled in pinned memory that lives as long as the engine: the copy to the device is one DMA, not a staged one)
    size_t buf_size = 0;
    bool buf_failed = false;
    auto put = [&](const void* src, size_t bytes) {
        const size_t o = buf_size, need = o + (bytes + 15) / 16 * 16;
        if (need > h_lu_cap_) {
            const size_t cap = std::max<size_t>(need * 2, size_t(1) << 20);
            char* grown = nullptr;
            if (hipHostMalloc(reinterpret_cast<void**>(&grown), cap, hipHostMallocDefault) != hipSuccess) { buf_failed = true; return o; }
            if (buf_size) std::memcpy(grown, h_lu_buf_, buf_size);
            if (h_lu_buf_) (void)hipHostFree(h_lu_buf_);
            h_lu_buf_ = grown; h_lu_cap_ = cap;
        }
        if (bytes) std::memcpy(h_lu_buf_ + o, src, bytes);
        buf_size = need;
        return o;
    };
    const size_t o_rp = put(hlu_.rowperm.data(), sizeof(int32_t) * m_), o_cp = put(hlu_.colperm.data(), sizeof(int32_t) * m_);
    // Forrest-Tomlin kernels: original row -> pivot, basis position -> pivot, pivot -> its row in the U / U' schedules
    std::vector<int32_t> inv_rp(m_), inv_cp(m_), task_uf(m_), task_ub(m_);
    for (int32_t k = 0; k < m_; ++k) {
        inv_rp[hlu_.rowperm[k]] = k; inv_cp[hlu_.colperm[k]] = k;
        task_uf[hlu_.Uf.level_rows[k]] = k; task_ub[hlu_.Ub.level_rows[k]] = k;
    }
    const size_t o_irp = put(inv_rp.data(), sizeof(int32_t) * m_), o_icp = put(inv_cp.data(), sizeof(int32_t) * m_);
    const size_t o_tuf = put(task_uf.data(), sizeof(int32_t) * m_), o_tub = put(task_ub.data(), sizeof(int32_t) * m_);
    // the four schedules once more for the persistent pivot kernel: consecutive levels fused into groups one pass solves
    // (relp_lu.hpp: fuse_levels), packed "ELL by pass", one contiguous image each (headers | lvl_pass | rdiag | sval | oval |
    // rovf | sidx | oidx); rows of U and U' without entries are kept, an update may mask them
    EllPacked ell[4];
    size_t o_rpos[4] = {0, 0, 0, 0}, o_tbits[4] = {0, 0, 0, 0};
    size_t o_ell[4] = {0, 0, 0, 0}, o_via_ptr[4] = {0, 0, 0, 0}, o_via_pos[4] = {0, 0, 0, 0}, o_triv[4] = {0, 0, 0, 0},
           o_reach[4] = {0, 0, 0, 0}, o_rhs[4] = {0, 0, 0, 0};
    int32_t rhs_base[4] = {0, 0, 0, 0};
    std::vector<int32_t> lev_ub(m_, 0);
    for (int32_t l = 0; l + 1 < (int32_t)hlu_.Ub.level_ptr.size(); ++l)
        for (int32_t t = hlu_.Ub.level_ptr[l]; t < hlu_.Ub.level_ptr[l + 1]; ++t) lev_ub[hlu_.Ub.level_rows[t]] = l;
    static_assert(kEllLgShift == kEllLg, "host packing and device decoding of sidx");
    if (ft_) {
        const int32_t fuse_cap = lu_fuse_lanes_env_;       // (RELP_FUSE_LANES, read at create)
        // (fused schedules read copies of some right-hand sides behind x: ft_rhs_cap_ words of LDS, and of index space)
        const int32_t cap = ft_rhs_cap_ > 0 ? fuse_cap : 0;
        const int64_t index_room = ft_big_ ? (int64_t(1) << kEllLgShiftWide) : (int64_t(1) << kEllLgShift);
        bool uses_rhs[4] = {false, false, false, false};
        // fusion and packing of the four schedules are independent: U on this thread, U' and L + L' on two helpers (the
        // refactorisation runs beside the pivot kernel, and what the host takes longer than the kernel's look-ahead the
        // device waits)
        auto prepare = [&](int k) {
            const bool maskable = k == 1 || k == 2;
            FusedSchedule fs;
            // (big layout: right-hand-side copies compacted, rows without entries as a list when that saves two passes or more;
            // all-in-LDS layout: a copy per pivot by one LDS loop and every row a slot, as measured fastest on 25FV47)
            const int32_t triv_min = ft_big_ ? 512 : 0x7fffffff;
            fuse_levels(*sch[k], maskable, maskable, cap, &fs);
            ell_pack(fs, maskable, &ell[k], ft_big_, ft_big_, triv_min);
            if (ft_big_ && ((int64_t)ell[k].rhs_src.size() > ft_rhs_cap_ || (int64_t)m_ + 1 + (int64_t)ell[k].rhs_src.size() > index_room)) {
                fuse_levels(*sch[k], maskable, maskable, 0, &fs);      // more copies than the layout has room for: level by level
                ell_pack(fs, maskable, &ell[k], ft_big_, ft_big_, triv_min);
            }
            uses_rhs[k] = false;
            for (int32_t v : fs.s.idx) if (v >= fs.rhs_base) { uses_rhs[k] = true; break; }
            if (k == 2) lev_ub = fs.start_after;
            if (uses_rhs[k]) rhs_base[k] = fs.rhs_base;
        };
        if (m_ >= 256) {
            host_pool_.run(0, [&] { prepare(2); });
            host_pool_.run(1, [&] { prepare(0); prepare(3); });
            prepare(1);
            host_pool_.wait();
        } else {
            for (int k = 0; k < 4; ++k) prepare(k);
        }
        for (int k = 0; k < 4; ++k) {
            const EllPacked& e = ell[k];
            std::vector<EllPassHost> hdrs(e.passes);
            hdrs.resize(hdrs.size() + kEllPadHeaders, EllPassHost{0, 0, 0, 0});       // the kernel reads headers ahead
            o_ell[k] = put(hdrs.data(), sizeof(EllPassHost) * hdrs.size());
            put(e.lvl_pass.data(), sizeof(int32_t) * e.lvl_pass.size());
            put(e.rdiag.data(), sizeof(double) * e.rdiag.size());
            put(e.sval.data(), sizeof(double) * e.sval.size());
            put(e.oval.data(), sizeof(double) * e.oval.size());
            put(e.rovf.data(), sizeof(int32_t) * e.rovf.size());
            if (ft_big_) { put(e.sidx32.data(), sizeof(uint32_t) * e.sidx32.size()); put(e.oidx32.data(), sizeof(uint32_t) * e.oidx32.size()); }
            else { put(e.sidx.data(), sizeof(uint16_t) * e.sidx.size()); put(e.oidx.data(), sizeof(uint16_t) * e.oidx.size()); }
        }
        for (int k = 1; k <= 2; ++k) {
            o_via_ptr[k] = put(ell[k].via_ptr.data(), sizeof(int32_t) * ell[k].via_ptr.size());
            o_via_pos[k] = put(ell[k].via_pos.data(), sizeof(int32_t) * ell[k].via_pos.size());
        }
        for (int k = 0; k < 4; ++k) {
            o_rhs[k] = put(ell[k].rhs_src.data(), sizeof(int32_t) * ell[k].rhs_src.size());
            o_triv[k] = put(ell[k].triv.data(), sizeof(int32_t) * ell[k].triv.size());
            if (ft_tier_ >= 2) {                       // layout 2 walks the non-zeros of x: the inverse of rhs_src, `triv` as a bitmap
                std::vector<int32_t> pos(m_, -1);
                for (size_t i = 0; i < ell[k].rhs_src.size(); ++i) pos[ell[k].rhs_src[i]] = (int32_t)i;
                std::vector<uint32_t> tb((size_t)(m_ + 31) / 32 + 1, 0u);
                for (int32_t r : ell[k].triv) tb[(size_t)r >> 5] |= 1u << (r & 31);
                o_rpos[k] = put(pos.data(), sizeof(int32_t) * pos.size());
                o_tbits[k] = put(tb.data(), sizeof(uint32_t) * tb.size());
            }
            o_reach[k] = put(ell[k].reach.data(), sizeof(int32_t) * ell[k].reach.size());
        }
    }
    const size_t o_lub = put(lev_ub.data(), sizeof(int32_t) * m_);
    size_t o_pinfo = 0;
    if (ft_) {
        std::vector<FtPivotInfo> pinfo(m_);
        for (int32_t p = 0; p < m_; ++p) {
            FtPivotInfo& q = pinfo[p];
            q.u_e0 = hlu_.Uf.ptr[p]; q.u_e1 = hlu_.Uf.ptr[p + 1];
            const bool v1 = !ell[1].via_ptr.empty(), v2 = !ell[2].via_ptr.empty();
            q.via_u0 = v1 ? ell[1].via_ptr[p] : 0; q.via_u1 = v1 ? ell[1].via_ptr[p + 1] : 0;
            q.via_t0 = v2 ? ell[2].via_ptr[p] : 0; q.via_t1 = v2 ? ell[2].via_ptr[p + 1] : 0;
            q.lev_ub = lev_ub[p]; q.pad_ = 0;
        }
        o_pinfo = put(pinfo.data(), sizeof(FtPivotInfo) * pinfo.size());
    }
    size_t o_rows[4], o_idx[4], o_val[4], o_lp[4], o_seg[4];
    int32_t n_seg[4] = {0, 0, 0, 0};
    std::vector<LuRow> rows(m_);
    for (int k = 0; k < 4; ++k) {
        const TriangularSchedule& t = *sch[k];
        o_rows[k] = o_idx[k] = o_val[k] = o_lp[k] = o_seg[k] = 0;
        if (ft_) {
            // the Forrest-Tomlin kernels solve from the ELL images; of the row-wise schedules they read one thing, the
            // entries of a row of U (the u_bar of an update)
            if (k == 1) { o_idx[k] = put(t.idx.data(), sizeof(int32_t) * t.idx.size()); o_val[k] = put(t.val.data(), sizeof(double) * t.val.size()); }
            continue;
        }
        for (int32_t i = 0; i < m_; ++i) {
            const int32_t r = t.level_rows[i];
            rows[i] = LuRow{r, t.ptr[r], t.ptr[r + 1], 0, 1.0 / t.diag[r]};
        }
        o_rows[k] = put(rows.data(), sizeof(LuRow) * m_);
        o_idx[k] = put(t.idx.data(), sizeof(int32_t) * t.idx.size());
        o_val[k] = put(t.val.data(), sizeof(double) * t.val.size());
        o_lp[k] = put(t.level_ptr.data(), sizeof(int32_t) * t.level_ptr.size());
        // runs of levels for the pipelined solve: "solo" runs (every level <= 8 rows, at least 2 levels) are walked by one
        // wavefront without workgroup barriers.  Directly behind level_ptr: the kernel stages all five arrays in one copy.
        std::vector<int32_t> segs;
        constexpr int32_t kSoloRows = 8;                   // one pass of eight 8-lane groups
        const int32_t nlev = (int32_t)t.level_ptr.size() - 1;
        for (int32_t l = 1; l < nlev;) {
            const bool narrow = t.level_ptr[l + 1] - t.level_ptr[l] <= kSoloRows;
            int32_t e = l + 1;
            while (e < nlev && ((t.level_ptr[e + 1] - t.level_ptr[e] <= kSoloRows) == narrow)) ++e;
            const bool solo = narrow && e - l >= 2;
            if (!segs.empty() && !solo && !segs[segs.size() - 1]) segs[segs.size() - 2] = e;      // merge wide runs
            else { segs.push_back(l); segs.push_back(e); segs.push_back(solo ? 1 : 0); }
            l = e;
        }
        n_seg[k] = (int32_t)segs.size() / 3;
        o_seg[k] = put(segs.data(), sizeof(int32_t) * segs.size());
    }
    if (buf_failed) return fail(RELP_E_ALLOC, "pinned staging buffer for the factors");
    if ((int64_t)buf_size > lu_cap_) {
        // (pointer and capacity are cleared before the new allocation: if it fails nothing dangles, and the look-ahead's
        // restore() below skips an empty buffer)
        char* const stale = d_lu_buf_;
        d_lu_buf_ = nullptr; lu_cap_ = 0;
        if (stale) HIP_TRY(hipFree(stale));
        const int64_t want = (int64_t)buf_size * 3 / 2 + 256;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_lu_buf_), (size_t)want));
        lu_cap_ = want;
    }
    HIP_TRY(hipMemcpyAsync(d_lu_buf_, h_lu_buf_, buf_size, hipMemcpyHostToDevice, stream_));
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
        ds[k]->seg = reinterpret_cast<const int32_t*>(d_lu_buf_ + o_seg[k]);
        ds[k]->n_seg = n_seg[k]; ds[k]->pad_ = 0;
    }
    HIP_TRY(hipStreamSynchronize(stream_));             // (the pinned buffer is rewritten by the next refactorisation)
    if (ft_) {
        if (ft_tier_ >= 2 && fts_.m != m_) {           // rows were removed: the bitmaps saved between launches describe another m
            const int32_t reset[4] = {-1, -1, 0, 0};
            HIP_TRY(hipMemcpyAsync(fts_.nzc, reset, sizeof reset, hipMemcpyHostToDevice, stream_));
            HIP_TRY(hipStreamSynchronize(stream_));
        }
        fts_.m = m_;
        fts_.inv_rowperm = reinterpret_cast<const int32_t*>(d_lu_buf_ + o_irp);
        fts_.inv_colperm = reinterpret_cast<const int32_t*>(d_lu_buf_ + o_icp);
        fts_.task_uf = reinterpret_cast<const int32_t*>(d_lu_buf_ + o_tuf);
        fts_.task_ub = reinterpret_cast<const int32_t*>(d_lu_buf_ + o_tub);
        fts_.lev_ub = reinterpret_cast<const int32_t*>(d_lu_buf_ + o_lub);
        fts_.pinfo = reinterpret_cast<const FtPivotInfo*>(d_lu_buf_ + o_pinfo);
        // what is left of the CU's LDS after the work vectors stages one schedule image at a time
        const int64_t base = (int64_t)ft_lds_base_bytes(m_, ft_tcap_, ft_eta_cap_, ft_tier_, ft_rhs_cap_);
        const int64_t idx_bytes = ft_big_ ? 4 : 2;
        fts_.stage_bytes = (int32_t)std::max<int64_t>(0, kFtLdsBudget - base);
        int64_t need = 0;
        auto up16 = [](int64_t b) { return (b + 15) / 16 * 16; };
        for (int k = 0; k < 4; ++k) {
            const EllPacked& e = ell[k];
            EllSchedule& d = fts_.ell[k];
            const int64_t np = (int64_t)e.passes.size(), nlv = (int64_t)e.lvl_pass.size(), nln = (int64_t)e.lanes(),
                          nov = (int64_t)e.overflow();
            char* q = d_lu_buf_ + o_ell[k];
            char* const q0 = q;
            d.passes = reinterpret_cast<const EllPass*>(q); q += up16(16 * (np + kEllPadHeaders));
            d.lvl_pass = reinterpret_cast<const int32_t*>(q); q += up16(4 * nlv);
            d.rdiag = reinterpret_cast<double*>(q); q += up16(8 * ((int64_t)m_ + 1));
            d.sval = reinterpret_cast<double*>(q); q += up16(8 * nln);
            d.oval = reinterpret_cast<const double*>(q); q += up16(8 * nov);
            d.rovf = reinterpret_cast<const int32_t*>(q); q += up16(4 * (int64_t)e.rovf.size());
            d.sidx = reinterpret_cast<const uint16_t*>(q); q += up16(idx_bytes * nln);      // (uint32_t when FtState::big)
            d.oidx = reinterpret_cast<const uint16_t*>(q); q += up16(idx_bytes * nov);
            const int64_t total = q - q0;
            d.n_passes = (int32_t)np; d.n_levels = (int32_t)nlv - 1; d.m = m_; d.n_lanes = (int32_t)nln; d.n_ovf = (int32_t)nov;
            d.bytes = (int32_t)total;
            d.rhs_base = rhs_base[k];
            d.n_triv = (int32_t)e.triv.size();
            d.triv = reinterpret_cast<const int32_t*>(d_lu_buf_ + o_triv[k]);
            d.reach = reinterpret_cast<const int32_t*>(d_lu_buf_ + o_reach[k]);
            d.rhs_src = reinterpret_cast<const int32_t*>(d_lu_buf_ + o_rhs[k]);
            d.n_rhs = ft_big_ ? (int32_t)e.rhs_src.size() : -1; d.pad_ = 0;
            d.rhs_pos = ft_tier_ >= 2 ? reinterpret_cast<const int32_t*>(d_lu_buf_ + o_rpos[k]) : nullptr;
            d.triv_bits = ft_tier_ >= 2 ? reinterpret_cast<const uint32_t*>(d_lu_buf_ + o_tbits[k]) : nullptr;
            const bool has_via = !e.via_ptr.empty();
            d.via_ptr = has_via ? reinterpret_cast<const int32_t*>(d_lu_buf_ + o_via_ptr[k]) : nullptr;
            d.via_pos = has_via ? reinterpret_cast<const int32_t*>(d_lu_buf_ + o_via_pos[k]) : nullptr;
            fts_.stage[k] = total <= fts_.stage_bytes ? 1 : 0;
            if (fts_.stage[k]) need = std::max(need, total);
        }
        if (ft_tier_ >= 2)                                 // (layout 2 copies the pass headers of an image that is not staged: sweep())
            for (int k = 0; k < 4; ++k) {
                const int64_t hb = 16 * ((int64_t)fts_.ell[k].n_passes + kEllPadHeaders);
                if (!fts_.stage[k] && hb <= fts_.stage_bytes) need = std::max(need, hb);
            }
        fts_.lds_bytes = (int32_t)(base + need);
        if (std::getenv("RELP_DEBUG") && lu_refactors_ % 60 == 59)
            for (int k = 0; k < 4; ++k)
                std::fprintf(stderr, "[relp] schedule %d: %d levels, %d passes, %d lanes (%d entries), image %d bytes, staged %d (stage area %d, base %lld)\n",
                             k, fts_.ell[k].n_levels, fts_.ell[k].n_passes, fts_.ell[k].n_lanes, (int)sch[k]->idx.size(),
                             fts_.ell[k].bytes, fts_.stage[k], fts_.stage_bytes, (long long)base);
    }
    return RELP_OK;
}

// ------------------------------------------------------------------------------------------------
// Forrest-Tomlin mode (relp_kernels_ft.hip)
// ------------------------------------------------------------------------------------------------
relp_status_t Engine::ft_plan_and_alloc() {
    ft_ = false; ft_big_ = false; ft_tier_ = 0; ft_rhs_cap_ = 0;
    if (m_ > kFtMaxRows) return RELP_OK;
    // The dense tail of U (tcap x tcap in LDS) is as large as the refactorisation interval asks for, not larger: what it does
    // not take stages the triangular factors, and an image that does not fit is solved from L2 at several times the cost.
    // Default interval 48: with a refactorisation at ~0.7 ms and ~1,100 clocks per pending update and pivot, the optimum is
    // flat between 40 and 64, and 48 x 49 doubles leave 14 KB more for the images than 64 x 65.
    int32_t want = cfg_.update_block < 0 ? 48 : std::max(1, std::min(cfg_.update_block, kFtMaxSlots));
    // (RELP_LU_PIPELINE_SHORT, run_ft: a short interval pivots on into a tail twice as long while the host factorises)
    if (std::getenv("RELP_LU_PIPELINE_SHORT") && std::atoi(std::getenv("RELP_LU_PIPELINE_SHORT")) != 0 && want < 24) want = 2 * want;
    // Two layouts (relp_kernels_ft.hip: ft_layout).  "All in LDS": x with m right-hand-side copies, spike, -pi, permutations,
    // eta pool -- 63 bytes per row.  "big": x, -pi and the slot tables only (17 bytes per row + 8 per right-hand-side copy the
    // fused schedules may use: as many as fit, a schedule that needs more is packed level by level), the rest read from L2;
    // slot indices of the images 32 bits wide.  The first is taken while it leaves the dense
    // tail the interval asks for AND >= kFtMinStage bytes to stage the factor images (an image that is not staged is solved
    // from L2 at several times the cost); RELP_FT_BIG = 0 / 1 forces one of them.
    const int64_t eta_cap = std::max<int64_t>((int64_t)2 * m_ + 64, 1024);   // (one eta never exceeds m entries)
    constexpr int64_t kFtMinStage = 64 * 1024;
    (void)kFtMinStage;
    const char* big_env = std::getenv("RELP_FT_BIG");
    const int force_big = big_env ? std::atoi(big_env) : -1;
    auto plan = [&](int32_t tier, int32_t rhs_cap, int64_t min_stage, int32_t min_tcap = 16) {
        for (int32_t tcap : {64, 48, 32, 16}) {
            if (tcap != 16 && tcap - 16 >= want) continue;              // a smaller tail serves the interval
            if (tcap < std::min(want, min_tcap)) return false;          // (a refactorisation every 16 pivots is the last resort)
            if (tcap < want && min_stage > 4096) return false;          // (only the last resort shortens the interval)
            if ((int64_t)ft_lds_base_bytes(m_, tcap, (int32_t)eta_cap, tier, rhs_cap) + min_stage <= kFtLdsBudget) {
                ft_tcap_ = tcap; ft_eta_cap_ = (int32_t)eta_cap; ft_tier_ = tier; ft_big_ = tier >= 1; ft_rhs_cap_ = rhs_cap; ft_ = true;
                return true;
            }
        }
        return false;
    };
    // (16-bit slot indices: m + 1 + copies < 8,192)
    const int32_t small_rhs = (int32_t)std::max<int64_t>(0, std::min<int64_t>(m_, (int64_t(1) << kEllLgShift) - 2 - m_));
    // (measured on GREENBEB, m = 2,228: all-in-LDS with a 32-slot tail and nothing staged 209,000 clocks per pivot, big with a
    // 48-slot tail and 78 KB of staging 235,000 -- what the big layout reads from L2 costs more than staging saves; so the
    // all-in-LDS layout is taken whenever it fits at all)
    // Layout 2 (nothing per row in LDS, x and -pi in L2) takes whatever the other two cannot hold: any m, every right-hand-side
    // copy the fused schedules ask for, the whole LDS minus the dense tail as the staging area.  RELP_FT_BIG = 2 forces it.
    // (the 16-bit row indices of the PRICE copy end at 32,767; the slot indices of the images at 2^24)
    const bool fits_tier1 = m_ < kPriceLongFlag;
    const bool fits_tier2 = 2 * (int64_t)m_ + 2 <= (int64_t(1) << kEllLgShiftWide);
    if (force_big < 1 && m_ < kPriceLongFlag && small_rhs > 0 && plan(0, small_rhs, 4096, force_big == 0 ? 16 : 32)) {}
    else if (force_big != 0 && force_big != 2 && fits_tier1 &&
             (plan(1, m_, kFtMinStage) || plan(1, std::min(m_, 2048), 32 * 1024) || plan(1, std::min(m_, 1024), 8 * 1024) || plan(1, 0, 4096))) {}
    // (layout 2 by default refactorises every 64 updates and lets the kernel make 16 of them while the host factorises: at
    // 64,000 rows a refactorisation is 5 ms of host time against 0.35 ms per pivot, and a pending update costs a pivot whose
    // per-row loops dominate next to nothing -- measured 2,670 it/s at 48 / 8, 2,870 at 64 / 8, 2,975 at 64 / 16)
    else if (force_big != 0 && force_big != 1 && fits_tier2 && ((want = cfg_.update_block < 0 ? 64 : want), plan(2, m_, kFtMinStage))) {}
    if (!ft_) return RELP_OK;
    if (std::getenv("RELP_DEBUG"))
        std::fprintf(stderr, "[relp] persistent pivot kernel: m %d, layout %s, %d right-hand-side copies, dense tail %d, LDS base %zu bytes\n",
                     m_, ft_tier_ >= 2 ? "nothing per row in LDS" : ft_big_ ? "big" : "all-in-LDS", ft_rhs_cap_, ft_tcap_,
                     ft_lds_base_bytes(m_, ft_tcap_, ft_eta_cap_, ft_tier_, ft_rhs_cap_));
    const int64_t tc = ft_tcap_, ldt = tc + 1, m = m_, nwp = kFtWaves + 1;
    std::vector<char> dummy;
    int64_t o = 0;
    auto take = [&](int64_t bytes) { const int64_t at = o; o += round_up(bytes, 16); return at; };
    // (what a refactorisation clears to 0 first, then what it clears to -1, then the rest: two memsets per reset)
    const int64_t o_hdr = take(16), o_sp = take(4 * tc), o_lv = take(4 * tc), o_tc = take(8 * tc * ldt), o_eo = take(4 * tc * nwp),
                  o_so = take(4 * tc * nwp), o_pv = take(4 * tc), o_ts = take(4 * m), o_ei = take(4 * (int64_t)ft_eta_cap_),
                  o_ev = take(8 * (int64_t)ft_eta_cap_), o_si = take(4 * tc * m), o_sv = take(8 * tc * m), o_spike = take(8 * m), o_prof = take(8 * 32),
                  o_journal = take(8 * tc), o_spw = take(8 * m), o_xw = take(ft_tier_ >= 2 ? 8 * (m + 1 + ft_rhs_cap_) : 0),
                  o_cm = take(ft_tier_ >= 1 ? 8 * ((m + 63) / 64 + 1) : 0), o_nzi = take(ft_tier_ >= 1 ? 4 * m : 0),
                  o_nzv = take(ft_tier_ >= 1 ? 8 * m : 0), o_rhoi = take(ft_tier_ >= 2 ? 4 * m : 0), o_nzc = take(16),
                  o_bits = take(ft_tier_ >= 2 ? kFtBitmapBytes + 1024 : 0);
    ft_zero_bytes_ = o_pv - o_hdr; ft_ones_bytes_ = o_ei - o_pv;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_ft_buf_), (size_t)o));
    HIP_TRY(hipMemset(d_ft_buf_, 0, (size_t)o));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h_ft_hdr_), 4 * sizeof(int32_t), hipHostMallocDefault));
    std::memset(h_ft_hdr_, 0, 4 * sizeof(int32_t));
    {   // the kernel's report in mapped host memory; without it (allocation refused) the copies below do the same job
        void* hp = nullptr; void* dp = nullptr;
        if (hipHostMalloc(&hp, sizeof(FtMirror) + sizeof(int32_t) * (size_t)m_, hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
            h_mirror_ = static_cast<FtMirror*>(hp); d_mirror_ = static_cast<FtMirror*>(dp);
            std::memset(h_mirror_, 0, sizeof(FtMirror));
        } else {
            if (hp) (void)hipHostFree(hp);
            (void)hipGetLastError();
        }
    }
    fts_ = FtState{};
    fts_.m = m_; fts_.tcap = ft_tcap_; fts_.ldt = (int32_t)ldt; fts_.eta_cap = ft_eta_cap_;
    fts_.hdr = reinterpret_cast<int32_t*>(d_ft_buf_ + o_hdr);
    fts_.slot_pivot = reinterpret_cast<int32_t*>(d_ft_buf_ + o_sp);
    fts_.slot_prev = reinterpret_cast<int32_t*>(d_ft_buf_ + o_pv);
    fts_.slot_live = reinterpret_cast<int32_t*>(d_ft_buf_ + o_lv);
    fts_.tslot = reinterpret_cast<int32_t*>(d_ft_buf_ + o_ts);
    fts_.TC = reinterpret_cast<double*>(d_ft_buf_ + o_tc);
    fts_.eta_off = reinterpret_cast<int32_t*>(d_ft_buf_ + o_eo);
    fts_.spk_off = reinterpret_cast<int32_t*>(d_ft_buf_ + o_so);
    fts_.eta_idx = reinterpret_cast<int32_t*>(d_ft_buf_ + o_ei);
    fts_.eta_val = reinterpret_cast<double*>(d_ft_buf_ + o_ev);
    fts_.spk_idx = reinterpret_cast<int32_t*>(d_ft_buf_ + o_si);
    fts_.spk_val = reinterpret_cast<double*>(d_ft_buf_ + o_sv);
    fts_.spike = reinterpret_cast<double*>(d_ft_buf_ + o_spike);
    fts_.sp_work = reinterpret_cast<double*>(d_ft_buf_ + o_spw);
    fts_.x_work = ft_tier_ >= 2 ? reinterpret_cast<double*>(d_ft_buf_ + o_xw) : nullptr;
    fts_.chunk_mask = ft_tier_ >= 1 ? reinterpret_cast<unsigned long long*>(d_ft_buf_ + o_cm) : nullptr;
    fts_.nz_idx = ft_tier_ >= 1 ? reinterpret_cast<int32_t*>(d_ft_buf_ + o_nzi) : nullptr;
    fts_.nz_val = ft_tier_ >= 1 ? reinterpret_cast<double*>(d_ft_buf_ + o_nzv) : nullptr;
    fts_.rho_idx = ft_tier_ >= 2 ? reinterpret_cast<int32_t*>(d_ft_buf_ + o_rhoi) : nullptr;
    fts_.nzc = reinterpret_cast<int32_t*>(d_ft_buf_ + o_nzc);
    fts_.bits_save = ft_tier_ >= 2 ? reinterpret_cast<uint32_t*>(d_ft_buf_ + o_bits) : nullptr;
    {   // (pb.alpha / pb.rho unknown, no bitmaps saved yet)
        const int32_t reset[4] = {-1, -1, 0, 0};
        HIP_TRY(hipMemcpy(fts_.nzc, reset, sizeof reset, hipMemcpyHostToDevice));
    }
    fts_.big = ft_tier_; fts_.rhs_cap = ft_rhs_cap_;
    {   // hyper-sparse starts: L and L' by default (U' starts from the leaving pivot's level anyway; on U the spike reaches the first groups: measured 31.4 of 31.4 passes on 25FV47, not worth the reduction); RELP_FT_HYPER = bit mask
        const char* e = std::getenv("RELP_FT_HYPER");
        fts_.hyper = e ? std::atoi(e) : 0x9;
        hyper_forced_ = e != nullptr;
    }
    fts_.prof = reinterpret_cast<long long*>(d_ft_buf_ + o_prof);
    fts_.journal = reinterpret_cast<int32_t*>(d_ft_buf_ + o_journal);
    // refactor when this many updates are pending (lower_upper/mod.rs:199-202 refactors when updates.len() > 10, i.e.
    // relp_config_t.update_block = 11 reproduces the reference's cadence)
    fts_.max_updates = cfg_.update_block < 0 ? ft_tcap_ : std::max(1, std::min(cfg_.update_block, ft_tcap_));
    block_ = fts_.max_updates;
    {   // RELP_FT_GRID_PRICE = 0 / 1 forces the choice (any layout: -pi is current in global memory between launches)
        const char* e = std::getenv("RELP_FT_GRID_PRICE");
        ft_grid_price_ = e ? std::atoi(e) != 0 : (ft_tier_ >= 2 && nr_columns() >= 32768);
    }
    return ft_build_price_ell();
}

// k-major PRICE copy of the structural columns (relp_kernels.h: PriceEll; rebuilt when rows are removed)
relp_status_t Engine::ft_build_price_ell() {
    const int64_t ns = std::max(nr_normal_, 1);
    std::vector<uint16_t> idx((size_t)kPriceSlots * ns, 0);
    std::vector<double> val((size_t)kPriceSlots * ns, 0.0);
    std::vector<int32_t> longs, very_long;
    for (int32_t p = 0; p < nr_normal_; ++p) {
        const int64_t n = hc_ptr_[p + 1] - hc_ptr_[p];
        for (int64_t k = 0; k < std::min<int64_t>(n, kPriceSlots); ++k) {
            idx[(size_t)k * ns + p] = (uint16_t)hc_idx_[hc_ptr_[p] + k]; val[(size_t)k * ns + p] = hc_val_[hc_ptr_[p] + k];
        }
        if (n > kPriceSlots) {
            idx[p] |= kPriceLongFlag;
            (n > kPriceLongSlots ? very_long : longs).push_back(p);
        }
    }
    const int64_t nl = std::max<int64_t>((int64_t)longs.size(), 1);
    std::vector<uint16_t> lidx((size_t)kPriceLongSlots * nl, 0);
    std::vector<double> lval((size_t)kPriceLongSlots * nl, 0.0);
    for (size_t i = 0; i < longs.size(); ++i) {
        const int32_t p = longs[i];
        for (int64_t k = 0; k < hc_ptr_[p + 1] - hc_ptr_[p]; ++k) {
            lidx[(size_t)k * nl + i] = (uint16_t)hc_idx_[hc_ptr_[p] + k]; lval[(size_t)k * nl + i] = hc_val_[hc_ptr_[p] + k];
        }
    }
    std::vector<char> buf;
    auto put = [&](const void* src, size_t bytes) {
        const size_t o = buf.size();
        buf.resize(o + (std::max<size_t>(bytes, 1) + 15) / 16 * 16);
        if (bytes) std::memcpy(buf.data() + o, src, bytes);
        return o;
    };
    const size_t o_val = put(val.data(), val.size() * 8), o_lval = put(lval.data(), lval.size() * 8);
    const size_t o_idx = put(idx.data(), idx.size() * 2), o_lidx = put(lidx.data(), lidx.size() * 2);
    const size_t o_long = put(longs.data(), longs.size() * 4), o_vl = put(very_long.data(), very_long.size() * 4);
    std::vector<uint16_t> long_of((size_t)ns, 0xFFFF);
    if (longs.size() < 0xFFFF) for (size_t i = 0; i < longs.size(); ++i) long_of[longs[i]] = (uint16_t)i;
    const size_t o_lof = put(long_of.data(), long_of.size() * 2);
    size_t o_idx32 = 0, o_lidx32 = 0;
    if (ft_tier_ >= 2) {                                   // the same two tables with 32-bit row indices (bit 31 = long column)
        std::vector<uint32_t> w(idx.size(), 0), lw(lidx.size(), 0);
        for (int32_t p = 0; p < nr_normal_; ++p) {
            const int64_t n = hc_ptr_[p + 1] - hc_ptr_[p];
            for (int64_t k = 0; k < std::min<int64_t>(n, kPriceSlots); ++k) w[(size_t)k * ns + p] = (uint32_t)hc_idx_[hc_ptr_[p] + k];
            if (n > kPriceSlots) w[p] |= kPriceLongFlag32;
        }
        for (size_t i = 0; i < longs.size(); ++i) {
            const int32_t p = longs[i];
            for (int64_t k = 0; k < hc_ptr_[p + 1] - hc_ptr_[p]; ++k) lw[(size_t)k * nl + i] = (uint32_t)hc_idx_[hc_ptr_[p] + k];
        }
        o_idx32 = put(w.data(), w.size() * 4); o_lidx32 = put(lw.data(), lw.size() * 4);
    }
    if (d_pe_buf_) { HIP_TRY(hipFree(d_pe_buf_)); d_pe_buf_ = nullptr; }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_pe_buf_), buf.size()));
    HIP_TRY(hipMemcpy(d_pe_buf_, buf.data(), buf.size(), hipMemcpyHostToDevice));
    pe_.val = reinterpret_cast<const double*>(d_pe_buf_ + o_val);
    pe_.lval = reinterpret_cast<const double*>(d_pe_buf_ + o_lval);
    pe_.idx = reinterpret_cast<const uint16_t*>(d_pe_buf_ + o_idx);
    pe_.lidx = reinterpret_cast<const uint16_t*>(d_pe_buf_ + o_lidx);
    pe_.long_cols = reinterpret_cast<const int32_t*>(d_pe_buf_ + o_long);
    pe_.very_long = reinterpret_cast<const int32_t*>(d_pe_buf_ + o_vl);
    pe_.long_of = reinterpret_cast<const uint16_t*>(d_pe_buf_ + o_lof);
    pe_.idx32 = ft_tier_ >= 2 ? reinterpret_cast<const uint32_t*>(d_pe_buf_ + o_idx32) : nullptr;
    pe_.lidx32 = ft_tier_ >= 2 ? reinterpret_cast<const uint32_t*>(d_pe_buf_ + o_lidx32) : nullptr;
    pe_.n_long = (int32_t)longs.size(); pe_.n_very_long = (int32_t)very_long.size();
    return RELP_OK;
}

// empty update file: t = 0, every pivot "never updated", TC = 0 (after a refactorisation)
relp_status_t Engine::ft_reset() {
    HIP_TRY(hipMemsetAsync(fts_.hdr, 0, (size_t)ft_zero_bytes_, stream_));                 // hdr, slot_pivot, slot_live, TC, offsets
    HIP_TRY(hipMemsetAsync(fts_.slot_prev, 0xFF, (size_t)ft_ones_bytes_, stream_));        // slot_prev, tslot: -1
    ft_need_refactor_ = false;
    return RELP_OK;
}

relp_status_t Engine::ft_read_hdr() {
    HIP_TRY(hipMemcpyAsync(h_ft_hdr_, fts_.hdr, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    since_flush_ = h_ft_hdr_[0];
    ft_need_refactor_ = h_ft_hdr_[2] != 0;
    return RELP_OK;
}

// After a k_ft_run launch: record, header and (mirror only) the basis on the host.  Synchronises the stream.
relp_status_t Engine::ft_read_report(bool* have_basis) {
    if (have_basis) *have_basis = false;
    if (!h_mirror_) {
        HIP_TRY(hipMemcpyAsync(h_rec_, d_rec_, sizeof(PivotRecord), hipMemcpyDeviceToHost, stream_));
        return ft_read_hdr();
    }
    HIP_TRY(hipStreamSynchronize(stream_));
    *h_rec_ = h_mirror_->rec;
    std::memcpy(h_ft_hdr_, h_mirror_->hdr, 4 * sizeof(int32_t));
    // Hyper-sparse starts pay when they skip more passes than the scan for the first group costs (~1,500 clocks = 2 passes):
    // a schedule whose sweeps saved less in this launch starts at the front in the next ones, and is probed again later.
    if (!hyper_forced_ && ft_big_) {
        for (int k : {0, 3}) {
            const int32_t sw = h_mirror_->sweeps[k];
            if (sw <= 0) continue;
            if ((fts_.hyper >> k) & 1) {
                if ((double)(h_mirror_->whole[k] - h_mirror_->walked[k]) < 2.0 * sw) { fts_.hyper &= ~(1 << k); hyper_probe_in_[k] = 64; }
            } else if (--hyper_probe_in_[k] <= 0) {
                fts_.hyper |= 1 << k;
            }
        }
    }
    since_flush_ = h_ft_hdr_[0];
    ft_need_refactor_ = h_ft_hdr_[2] != 0;
    if (have_basis) {
        if (!h_basis_) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h_basis_), sizeof(int32_t) * (size_t)std::max(m_alloc_rows_, m_), hipHostMallocDefault));
        std::memcpy(h_basis_, h_mirror_->basis, sizeof(int32_t) * (size_t)m_);     // (the next launch rewrites the mirror)
        *have_basis = true;
    }
    return RELP_OK;
}

// Pivots of the persistent kernel on `go`: one launch that runs until something stops it, or -- Dantzig's rule over very many
// columns (ft_grid_price_) -- PRICE as a grid launch (every CU instead of one) followed by ONE pivot of the persistent kernel with
// the column the grid chose, as many times as the update file has room, enqueued without a synchronisation (a launch whose
// record says "decided" returns at once), plus one launch that only marks the refactorisation due.  The launches of a batch
// share the journal (hdr[3]) and honour each other's "refactorisation due" (hdr[2]); only the last one writes the host mirror.
void Engine::ft_enqueue_pivots(const FtState& go, int rule, int64_t left) {
    if (!(ft_grid_price_ && rule == RELP_RULE_STEEPEST_DESCENT)) {
        launch_ft_run(dlu_, go, ft_problem(rule), left, stream_);
        return;
    }
    const int64_t room = std::max<int64_t>((int64_t)go.max_updates - since_flush_, 0);
    const int64_t batch = std::min<int64_t>(left, room + 1);
    (void)hipMemsetAsync(fts_.hdr + 2, 0, 2 * sizeof(int32_t), stream_);
    FtProblem pb = ft_problem(rule);
    pb.external_price = 1;
    const ColumnTable ct = table();
    SelectPartials sp;
    sp.k1 = d_part_k1_; sp.j = d_part_j_; sp.in_basis = d_in_basis_; sp.tol_cost = cfg_.tol_cost; sp.rule = rule;
    const int nb_struct = price_csc_blocks(0, nr_normal_), nb_virt = price_virtual_blocks(ct);
    sp.n = nr_columns(); sp.offset = 0; sp.nb_struct = nb_struct; sp.tol_tie = cfg_.tol_tie; sp.p_lo = 0; sp.cols_per_slot = 256;
    for (int64_t k = 0; k < batch; ++k) {
        if (nb_struct > 0 && nb_virt > 0) {
            launch_price_csc_all(csc(), ct, d_minus_pi_, d_d_, nr_normal_, phase_, sp, nb_virt, d_rec_, stream_);
        } else {
            launch_price_csc(csc(), ct, d_minus_pi_, d_d_, 0, nr_normal_, phase_, sp, d_rec_, stream_);
            SelectPartials spv = sp;
            spv.offset = nb_struct;
            launch_price_virtual_sel(ct, d_minus_pi_, d_d_, phase_, spv, d_rec_, stream_);
        }
        launch_select_partials_csc(sp, nb_struct + nb_virt, d_d_, csc(), ct, m_, nullptr, d_rec_, stream_);
        pb.mirror = k + 1 == batch ? d_mirror_ : nullptr;
        launch_ft_run(dlu_, go, pb, 1, stream_);
    }
}

FtProblem Engine::ft_problem(int rule) const {
    FtProblem pb{};
    pb.csc = csc(); pb.ct = table(); pb.pe = pe_;
    pb.minus_pi = d_minus_pi_; pb.b = d_b_; pb.alpha = d_alpha_; pb.rho = d_rho_; pb.d = d_d_;
    pb.basis = d_basis_; pb.in_basis = d_in_basis_; pb.trace = d_trace_; pb.trace_cap = trace_cap_;
    pb.rec = d_rec_;
    pb.mirror = d_mirror_;
    pb.tol = tolerances();
    pb.rule = rule; pb.n = nr_columns(); pb.phase = phase_;
    return pb;
}

// phase_one::primal / phase_two::primal with the pivots themselves on the device: one launch runs until the outcome is
// decided, the iteration budget is spent or the update file is full (then: refactorise on the host, launch again)
relp_status_t Engine::run_ft(int64_t max_iters, int64_t* done, int32_t* outcome) {
    relp_status_t st = download_rec();
    if (st) return st;
    // (layout 2 keeps lists of where alpha and rho are not zero and rewrites those places only; whatever ran since the last call --
    // step-wise API, phase switch, warm start -- may have written the two vectors: the first pivot rewrites them densely)
    if (ft_tier_ >= 2) HIP_TRY(hipMemsetAsync(fts_.nzc, 0xFF, 2 * sizeof(int32_t), stream_));
    const long long start = h_rec_->iterations;
    const int rule = phase_ == 1 ? cfg_.phase_one_rule : cfg_.phase_two_rule;
    struct Tick { int64_t& t; ~Tick() { ++t; } };
    // Look-ahead refactorisation (lu_refactor_lookahead): the kernel returns `la` updates before the file is full, and fills
    // the rest while the host factorises.  On for refactorisation intervals from 24 on; RELP_LU_LOOKAHEAD = 0 switches it off.
    const int32_t la_env = luf_enabled_ ? 0 : (ft_tier_ >= 2 && !lu_lookahead_set_) ? 16 : lu_lookahead_env_;     // (RELP_LU_LOOKAHEAD, read at create; the device
                                                                     // factorisation is synchronous on the engine's stream)
    const int32_t la = (fts_.max_updates >= 24 && la_env > 0) ? std::min(la_env, fts_.max_updates / 3) : 0;
    // Short intervals (the reference's cadence of 11: too short for the look-ahead above, which needs its updates inside the interval):
    // RELP_LU_PIPELINE_SHORT=1 lets the kernel return at the interval, pivot on into the REST of the dense tail (up to twice the
    // interval) while the host factorises, and replays those pivots onto the new factors -- the factors lag one interval behind, the
    // pivots are the same, the device never waits for a whole factorisation.  Opt-in: a measurement (bench: reference cadence).
    const char* pipeline_s = std::getenv("RELP_LU_PIPELINE_SHORT");
    const bool pipeline_env = pipeline_s && std::atoi(pipeline_s) != 0;
    const bool pipeline_short = pipeline_env && la == 0 && !luf_enabled_ && la_env > 0 && fts_.max_updates < 24 && 2 * fts_.max_updates <= ft_tcap_;
    bool have_basis = false;                               // h_basis_ holds the basis as the last launch left it
    while (h_rec_->outcome == DEV_RUNNING && h_rec_->iterations - start < max_iters) {
        if (ft_need_refactor_) {
            const bool ahead = (la > 0 && h_ft_hdr_[2] == 1 && h_ft_hdr_[0] == fts_.max_updates - la) ||
                               (pipeline_short && h_ft_hdr_[2] == 1 && h_ft_hdr_[0] >= fts_.max_updates && h_ft_hdr_[0] < 2 * fts_.max_updates);
            prof_tick_ = 0;                                // refactorisations are always bracketed (like the flush)
            if (ahead) {
                lu_pipeline_cap_ = pipeline_short ? 2 * fts_.max_updates : 0;
                if ((st = lu_refactor_lookahead(rule, max_iters - (h_rec_->iterations - start), have_basis))) return st;
                have_basis = false;                        // (the relaunched kernel has changed the basis since)
                if (ft_need_refactor_ || h_rec_->outcome != DEV_RUNNING) continue;      // (re-examined at the loop head)
            } else {
                prof_begin(RELP_K_FLUSH);
                st = lu_refactor();
                prof_end();
                if (st) return st;
            }
            if (h_rec_->iterations - start >= max_iters) break;
        }
        prof_tick_ = 0;
        prof_begin(RELP_K_FT_RUN);
        FtState go = fts_;
        go.max_updates = fts_.max_updates - la;
        const int64_t left = max_iters - (h_rec_->iterations - start);
        ft_enqueue_pivots(go, rule, left);
        prof_end();
        if ((st = ft_read_report(&have_basis))) return st;
        if (!ft_need_refactor_ && h_rec_->outcome == DEV_RUNNING && h_rec_->iterations - start < max_iters)
            return fail(RELP_E_STATE, "the pivot kernel stopped without a reason");
    }
    if (hipGetLastError() != hipSuccess) return fail(RELP_E_HIP, "kernel launch failed");
    int32_t oc = RELP_RUNNING;
    if (h_rec_->outcome == DEV_NO_CANDIDATE) {
        if (hold_phase_end_) oc = 100;                     // (kHeldNoCandidate, relp_engine.cpp: columns are barred by the pivot rescue)
        else if (phase_ == 2) oc = RELP_OPTIMAL;
        else if ((st = finish_phase_one(&oc))) return st;
    } else if (h_rec_->outcome == DEV_NO_ROW) {
        oc = phase_ == 2 ? RELP_UNBOUNDED : RELP_NO_ROW_PHASE_ONE;
    }
    if (done) *done = h_rec_->iterations - start;
    if (outcome) *outcome = oc;
    return RELP_OK;
}

// shader clocks (thread 0 of the persistent kernel) per phase of the pivot, accumulated since create
relp_status_t Engine::lu_phase_cycles(int64_t* out16) {
    if (!lu_ || !ft_) return fail(RELP_E_UNSUPPORTED, "phase clocks are the persistent pivot kernel's");
    HIP_TRY(hipStreamSynchronize(stream_));
    HIP_TRY(hipMemcpy(out16, fts_.prof, 16 * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (std::getenv("RELP_DEBUG")) {                       // passes walked per sweep of L, U, U', L' against the whole schedule
        int64_t ex[16];
        HIP_TRY(hipMemcpy(ex, fts_.prof + 16, sizeof ex, hipMemcpyDeviceToHost));
        if (ex[15]) std::fprintf(stderr, "[relp] non-zeros per pivot (layout 2): entering column %.1f, eta row %.1f, spike %.1f\n",
                                 (double)ex[12] / ex[15], (double)ex[13] / ex[15], (double)ex[14] / ex[15]);
        static const char* nm[4] = {"L", "U", "U'", "L'"};
        for (int k = 0; k < 4; ++k)
            if (ex[4 + k]) std::fprintf(stderr, "[relp] %s: %lld sweeps, %.1f passes walked of %.1f on average\n", nm[k], (long long)ex[4 + k],
                                        (double)ex[k] / ex[4 + k], (double)ex[8 + k] / ex[4 + k]);
    }
    return RELP_OK;
}

relp_status_t Engine::lu_stats(int64_t* out8) const {
    if (!lu_) return RELP_E_STATE;
    out8[0] = lu_refactors_; out8[1] = hlu_.m; out8[2] = hlu_.nnz_l; out8[3] = hlu_.nnz_u;
    out8[4] = (int64_t)hlu_.Lf.level_ptr.size() - 1; out8[5] = (int64_t)hlu_.Uf.level_ptr.size() - 1;
    out8[6] = (int64_t)hlu_.Ub.level_ptr.size() - 1; out8[7] = (int64_t)hlu_.Lb.level_ptr.size() - 1;
    return RELP_OK;
}

relp_status_t Engine::lu_kernel_layout(int32_t* out4) const {
    if (!lu_) return RELP_E_STATE;
    out4[0] = ft_ ? 1 : 0; out4[1] = ft_ ? ft_tier_ : -1; out4[2] = ft_ ? ft_tcap_ : 0; out4[3] = (ft_ && ft_grid_price_) ? 1 : 0;
    return RELP_OK;
}

relp_status_t Engine::lu_lookahead_stats(int64_t* out4) const {
    if (!lu_) return RELP_E_STATE;
    out4[0] = lu_lookahead_installs_; out4[1] = lu_replayed_changes_; out4[2] = (ft_tier_ >= 2 && !lu_lookahead_set_) ? 16 : lu_lookahead_env_; out4[3] = lu_fuse_lanes_env_;
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
