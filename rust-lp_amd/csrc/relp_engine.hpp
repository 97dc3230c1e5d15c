// relp_engine.hpp -- host driver of the explicit-inverse pivot engine.
//
// `Engine` plays the role of the reference's `Tableau<Carry<f64, BasisInverseRows<f64>>, K>` plus
// the `PivotRule` state and the `MatrixData` provider (file:line under
// /root/reference/src/algorithm/two_phase/):
//   MatrixData columns/rows          matrix_provider/matrix_data.rs:198-268, 308-371, 432-452
//   Partially / NonArtificial kinds  tableau/kind/artificial/partially.rs:125-206, kind/non_artificial.rs:151-220
//   Carry constructors / updates     tableau/inverse_maintenance/carry/mod.rs:214-271, 283-333, 381-426, 484-570
//   phase loops                      phase_one.rs:125-170, 223-260; phase_two.rs:22-51; two_phase/mod.rs:30-76
// All numeric state lives in HBM; the host only sequences launches and handles the rare
// phase-boundary work.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/relp_engine.h"
#include "relp_kernels.h"
#include "relp_lu.hpp"

namespace relp {

class Engine {
  public:
    Engine() = default;
    ~Engine();

    relp_status_t create(const relp_matrix_data_t& md, const relp_config_t& cfg);
    relp_status_t set_stream(hipStream_t s);

    // step-wise pivot
    relp_status_t select_primal_pivot_column(int rule, int32_t* found, int32_t* column, double* cost);
    relp_status_t relative_costs(double* out_n);
    relp_status_t generate_column(int32_t column, double* out_m);
    relp_status_t generate_element(int32_t row, int32_t column, double* out);
    relp_status_t select_primal_pivot_row(int32_t* found, int32_t* row);
    relp_status_t select_primal_pivot_row_of(const double* column, int32_t* found, int32_t* row);
    relp_status_t bring_into_basis(int32_t column, int32_t row, double cost, int32_t* leaving);

    // loops
    relp_status_t run(int64_t max_iters, int64_t* done, int32_t* outcome);
    int32_t engine_kind() const { return cfg_.engine; }
    relp_status_t robust_stats(int64_t* out4) const;
    relp_status_t solve_relaxation(int64_t max_iters, int32_t* outcome);
    relp_status_t from_basis(const int32_t* basis_columns);
    relp_status_t set_reinversion_interval(int64_t pivots);
    int64_t reinversions() const { return reinversions_; }
    relp_status_t flush();
    int32_t update_block() const { return block_; }
    relp_status_t lu_stats(int64_t* out8) const;
    relp_status_t lu_lookahead_stats(int64_t* out4) const;
    relp_status_t lu_kernel_layout(int32_t* out4) const;
    relp_status_t luf_stats(int64_t* out6) const;
    relp_status_t lu_set_device_factorisation(bool on);
    relp_status_t lu_factor_residual(double* out);
    relp_status_t lu_basis_columns(std::vector<std::vector<std::pair<int32_t, double>>>& cols);
    relp_status_t lu_basis_flat(std::vector<int64_t>& ptr, std::vector<int32_t>& idx, std::vector<double>& val);
    relp_status_t lu_phase_cycles(int64_t* out16);
    relp_status_t lu_download_basis();
    relp_status_t lu_factor_downloaded_basis();
    relp_status_t lu_refactor_lookahead(int rule, int64_t budget, bool have_basis);
    relp_status_t ft_read_report(bool* have_basis);
    void lu_refactor_clock(std::chrono::steady_clock::time_point tb, std::chrono::steady_clock::time_point t0,
                           std::chrono::steady_clock::time_point t1, std::chrono::steady_clock::time_point t2);
    // BasisInverse surface of the LU engine (carry/mod.rs:68-157, lower_upper/mod.rs:199-222)
    relp_status_t basis_inverse_row(int32_t row, double* out_m);
    relp_status_t should_refactor(int32_t* out);
    relp_status_t generate_column_of(const int32_t* idx, const double* val, int32_t nnz, double* out_m);
    relp_status_t cost_difference_of(const int32_t* idx, const double* val, int32_t nnz, double* out);
    relp_status_t lu_change_basis(int32_t row);
    relp_status_t lu_updates(int32_t* count);
    relp_status_t lu_get_update(int32_t k, int32_t* pivot, int32_t* idx, double* val, int32_t cap, int32_t* nnz);
    relp_status_t lu_get_upper(int64_t* col_ptr, int32_t* row_idx, double* values, int64_t cap, int64_t* nnz);
    relp_status_t lu_set_factors(const int64_t* l_ptr, const int32_t* l_idx, const double* l_val, const int64_t* u_ptr,
                                 const int32_t* u_idx, const double* u_val);
    relp_status_t shard_flush_begin(double** dev_snapshot, int64_t* len);
    relp_status_t shard_flush_end();

    // getters
    int32_t nr_rows() const { return m_; }
    int32_t nr_columns() const { return nr_artificial_ + n_provider_; }
    int32_t phase() const { return phase_; }
    int32_t nr_artificial() const { return nr_artificial_; }
    relp_status_t get_objective(double* out);
    relp_status_t get_vector(int which, double* out);  // 0 b, 1 minus_pi, 2 alpha
    relp_status_t get_basis_indices(int32_t* out);
    relp_status_t get_basis_inverse(double* out);
    relp_status_t current_bfs(int32_t* cols, double* vals, int32_t cap, int32_t* count);
    relp_status_t get_iterations(int64_t* out);
    relp_status_t get_degenerate_pivots(int64_t* out);
    relp_status_t get_trace(int32_t* phase, int32_t* entering, int32_t* row, int32_t* leaving, int64_t cap,
                            int64_t* count);
    relp_status_t check_basis(double* max_identity_error, double* max_basic_cost, double* min_b);

    relp_status_t profile_enable(bool enable, int64_t max_launches, int32_t sample_every);
    relp_status_t profile_read(int kernel_id, int64_t* launches, double* total_ms);

    // shards
    void shard_ranges(int32_t* col_lo, int32_t* col_hi, int32_t* row_lo, int32_t* row_hi, int32_t* stride) const;
    int64_t candidate_len() const { return cand_len_; }
    int64_t rho_len() const { return ld_b_; }
    relp_status_t shard_price(double* dev_candidate);
    relp_status_t shard_select_column(const double* dev_candidates, int32_t count);
    relp_status_t shard_ftran(double* dev_alpha_slice);
    relp_status_t shard_ratio(const double* dev_alpha_slices, int32_t count, double* dev_rho);
    relp_status_t shard_update(const double* dev_rho);
    relp_status_t shard_pivot();
    relp_status_t shard_set_collectives(relp_allgather_fn ag, relp_allreduce_sum_fn ar, void* ctx);
    relp_status_t shard_run(int64_t max_iters, int64_t* done, int32_t* outcome);
    void shard_inject_failure(int64_t after_pivots) { inject_failure_after_ = after_pivots; }
    relp_status_t rccl_attach(const uint8_t* id);
    relp_status_t poll(int32_t* outcome, int64_t* iterations);

    const char* last_error() const { return err_.c_str(); }

  private:
    // ---- MatrixData (host mirror) ----
    int32_t nr_normal_ = 0, nr_eq_ = 0, nr_range_ = 0, nr_le_ = 0, nr_ge_ = 0;
    int32_t mc_ = 0;          // constraint rows of A
    int32_t nr_bounds_ = 0;   // variables with an upper bound
    int32_t m_ = 0;           // tableau rows
    int32_t n_provider_ = 0;  // provider columns (structural + virtual)
    int32_t nr_virtual_ = 0;
    std::vector<double> cost_h_, upper_h_, rhs_h_;
    std::vector<int32_t> bound_row_h_, vrow0_h_, vrow1_h_, vsign_h_;
    relp_config_t cfg_{};

    // ---- Kind ----
    int32_t phase_ = 1;
    int32_t nr_artificial_ = 0;
    std::vector<int32_t> column_to_row_;
    double initial_phase1_objective_ = 0.0;
    int32_t wrapped_na_ = 0;      // nr_artificial at the phase switch (decodes wrapped artificial indices)

    // ---- device state ----
    double* dA_ = nullptr; int64_t ld_a_ = 0; bool owns_A_ = false;
    double* dBinv_ = nullptr; int64_t ld_b_ = 0;
    double *d_minus_pi_ = nullptr, *d_b_ = nullptr, *d_alpha_ = nullptr, *d_aq_ = nullptr, *d_rho_ = nullptr,
           *d_d_ = nullptr, *d_w_ = nullptr, *d_cost_ = nullptr;
    int32_t *d_basis_ = nullptr, *d_column_to_row_ = nullptr, *d_bound_row_ = nullptr, *d_vrow0_ = nullptr,
            *d_vrow1_ = nullptr, *d_vsign_ = nullptr, *d_trace_ = nullptr;
    uint8_t* d_in_basis_ = nullptr;
    PivotRecord* d_rec_ = nullptr;
    PivotRecord* h_rec_ = nullptr;  // pinned
    hipStream_t stream_ = nullptr; bool owns_stream_ = false;
    int64_t trace_cap_ = 0;
    // deferred update (B^-1 = (I + W S') B0inv), see relp_kernels.h
    int32_t block_ = 0;            // K, 0 = explicit rank-1 updates
    int64_t since_flush_ = 0;      // pivots enqueued since the last flush
    double* d_v_ = nullptr;        // B0inv a_q before the W correction
    double *d_W_ = nullptr, *d_wr_ = nullptr, *d_R_ = nullptr;
    int32_t *d_S_ = nullptr, *d_pos_of_row_ = nullptr;
    double* d_part_k1_ = nullptr;  // PRICE workgroups' partial argmin (SelectPartials)
    int32_t* d_part_j_ = nullptr;
    // dense-tableau engine (cfg.engine == RELP_ENGINE_TABLEAU)
    bool tableau_ = false;
    double* dT0_ = nullptr; int64_t ld_t_ = 0;
    double* dR0_ = nullptr; int64_t ld_r_ = 0;
    double* d_cost_store_ = nullptr;     // cost per stored column in the current phase
    int32_t* d_idcol_ = nullptr;         // stored column that was e_k originally, per row k
    // Ratio test + update in one launch (single-GPU loop, relp_kernels.h: launch_tab_ratio_update_all): the second copies of b
    // and the basis array it writes (swapped with d_b_ / d_basis_ after every pivot), the shadow row of W and its {row, length}
    bool fused_update_ = false, shadow_pending_ = false;
    double *d_b_alt_ = nullptr, *d_shadow_ = nullptr;
    int32_t *d_basis_alt_ = nullptr, *d_shadow_meta_ = nullptr;
    double* d_rmin_ = nullptr;           // minimum ratio per block of 256 rows (k_tab_select_column -> k_ratio_blocks)
    int32_t n_store_ = 0;                // stored columns = original artificials + provider columns
    int32_t tab_na_ = 0;                 // original number of artificial columns (their block is kept)
    int32_t sc_lo_ = 0, sc_hi_ = 0;      // storage columns owned by this rank (sharded tableau)
    bool tab_partials_valid_ = false;    // the PRICE partials describe the current d
    static constexpr int kRepriceEveryFlushes = 8;
    int32_t flushes_since_reprice_ = 0;
    std::vector<int32_t> idcol_h_;
    std::vector<double> cost_store_h_;
    bool in_loop_ = false;               // inside relp_run / relp_shard_run (the two-launch pivot keeps a shadow row pending)
    void tab_settle();                   // fold the fused update's shadow row into W
    struct LoopScope {                   // marks the pivot loops; settles on every way out
        Engine& e;
        explicit LoopScope(Engine& en) : e(en) { e.in_loop_ = true; }
        ~LoopScope() { e.in_loop_ = false; e.tab_settle(); }
    };
    struct SettledScope {                // work that reads or rebuilds the tableau from inside a pivot loop (re-tabulation)
        Engine& e; bool was;
        explicit SettledScope(Engine& en) : e(en), was(en.in_loop_) { e.in_loop_ = false; e.tab_settle(); }
        ~SettledScope() { e.in_loop_ = was; }
    };
    TableauView tview() const;
    double* d_aq_big() { return dR0_ + (int64_t)block_ * ld_r_; }         // scratch row behind R0 (owned columns)
    SelectPartials tab_partials(int rule) const;
    void enqueue_iteration_tableau(int rule);
    relp_status_t tableau_reprice();
    // sparse LU engine (cfg.engine == RELP_ENGINE_LU): B^-1 = (I + W S') (L U)^-1, refactor every block_ pivots
    bool lu_ = false;
    std::vector<int64_t> hc_ptr_; std::vector<int32_t> hc_idx_; std::vector<double> hc_val_;   // host CSC of A
    int64_t* d_cptr_ = nullptr; int32_t* d_cidx_ = nullptr; double* d_cval_ = nullptr;         // device CSC of A
    LUFactors hlu_;
    char* d_lu_buf_ = nullptr; int64_t lu_cap_ = 0;       // packed factors (permutations, rows, entries, levels)
    std::vector<std::vector<std::pair<int32_t, double>>> basis_cols_;      // the basis columns handed to lu_factor
    std::vector<int64_t> basis_ptr_; std::vector<int32_t> basis_idx_; std::vector<double> basis_val_;   // ... as one flat copy (lu_factor_csc)
    // two helper threads for the host side of a refactorisation (created at first use, joined with the engine)
    struct HostPool {
        struct Slot { std::thread th; std::mutex mu; std::condition_variable cv; std::function<void()> job; bool busy = false, stop = false; };
        Slot slot[2];
        void run(int i, std::function<void()> f) {
            Slot& s = slot[i];
            if (!s.th.joinable())
                s.th = std::thread([&s] {
                    std::unique_lock<std::mutex> lk(s.mu);
                    for (;;) {
                        s.cv.wait(lk, [&s] { return s.busy || s.stop; });
                        if (s.stop) return;
                        lk.unlock(); s.job(); lk.lock();
                        s.busy = false;
                        s.cv.notify_all();
                    }
                });
            { std::lock_guard<std::mutex> lk(s.mu); s.job = std::move(f); s.busy = true; }
            s.cv.notify_all();
        }
        void wait() {
            for (Slot& s : slot) { std::unique_lock<std::mutex> lk(s.mu); s.cv.wait(lk, [&s] { return !s.busy; }); }
        }
        ~HostPool() {
            for (Slot& s : slot) {
                if (!s.th.joinable()) continue;
                { std::lock_guard<std::mutex> lk(s.mu); s.stop = true; }
                s.cv.notify_all();
                s.th.join();
            }
        }
    } host_pool_;
    char* h_lu_buf_ = nullptr; size_t h_lu_cap_ = 0;      // the same, assembled in pinned host memory
    char* d_lu_buf_alt_ = nullptr; int64_t lu_cap_alt_ = 0;   // second device buffer: the factors the host prepares while the kernel runs
    int32_t* h_basis_ = nullptr; int32_t m_alloc_rows_ = 0;   // pinned: the basis a refactorisation downloads
    FtMirror* h_mirror_ = nullptr; FtMirror* d_mirror_ = nullptr;   // pinned + mapped: what k_ft_run reports (relp_kernels.h)
    double* d_lu_scratch_ = nullptr;
    DeviceLU dlu_{};
    relp_status_t lu_status_ = RELP_OK;                   // a failed refactorisation inside the loop
    int64_t lu_refactors_ = 0;
    int64_t lu_lookahead_installs_ = 0, lu_replayed_changes_ = 0;     // look-ahead refactorisations installed, journal entries replayed
    int32_t lu_lookahead_env_ = 8, lu_fuse_lanes_env_ = 256;          // RELP_LU_LOOKAHEAD, RELP_FUSE_LANES (read at create)
    bool lu_lookahead_set_ = false;                      // RELP_LU_LOOKAHEAD given (else layout 2 takes 16)
    int32_t lu_pipeline_cap_ = 0;                        // RELP_LU_PIPELINE_SHORT: the update file's cap while the host factorises (run_ft)
    double refactor_us_[3] = {0.0, 0.0, 0.0};            // host time: basis + columns, factorisation, schedules + upload
    // revised engine: B^-1 is re-inverted from the basis columns every `reinvert_interval_` pivots (0 = never)
    int64_t reinvert_interval_ = 0, since_reinvert_ = 0, reinversions_ = 0;
    DeviceCSC csc() const { return DeviceCSC{d_cptr_, d_cidx_, d_cval_}; }
    relp_status_t lu_load_matrix(const relp_matrix_data_t& md);
    relp_status_t lu_refactor();
    relp_status_t lu_upload_factors();
    relp_status_t reinvert();
    relp_status_t retabulate();
    bool retab_done_ = false;                             // the last retabulate() rebuilt the tableau (it keeps the old one otherwise)
    relp_status_t build_basis_columns(const std::vector<int32_t>& basis, std::vector<std::vector<std::pair<int32_t, double>>>* cols);
    void enqueue_iteration_lu(int rule);
    // Forrest-Tomlin mode of the LU engine (relp_kernels_ft.hip): whole pivots in one persistent workgroup; chosen at
    // create when the work vectors, the eta pool and the dense tail of U fit one CU's LDS (m <= kFtMaxRows)
    bool ft_ = false;
    FtState fts_{};
    char* d_ft_buf_ = nullptr;
    int32_t* h_ft_hdr_ = nullptr;                         // pinned copy of fts_.hdr
    int32_t ft_tcap_ = 0, ft_eta_cap_ = 0;
    // the refactorisation on the device (relp_engine_luf.cpp, relp_lu_factor_core.h): RELP_LU_DEVICE_FACTOR=1 or
    // relp_lu_set_device_factorisation; a bump beyond the dense working copy falls back to lu_factor on the host
    struct LufState;
    LufState* luf_ = nullptr;
    bool luf_enabled_ = false;
    int64_t luf_runs_ = 0, luf_fallbacks_ = 0, luf_lds_retries_ = 0; double luf_kernel_us_ = 0.0; int32_t luf_last_bump_ = 0, luf_last_peeled_ = 0;
    relp_status_t luf_prepare();
    relp_status_t lu_factor_on_device(int32_t* device_status);
    relp_status_t luf_download_factors();
    relp_status_t lu_host_factors();
    bool luf_is_resident() const;
    bool luf_download_ = false;                          // RELP_LU_DEVICE_FACTOR=2: download the factors, schedule on the host
    void luf_release();
    void luf_mark_dirty();
    bool hyper_forced_ = false; int32_t hyper_probe_in_[4] = {0, 0, 0, 0};    // adaptive hyper-sparse starts (ft_read_report)
    bool ft_big_ = false; int32_t ft_rhs_cap_ = 0;        // layout of the persistent kernel (relp_kernels_ft.hip: ft_layout)
    bool ft_grid_price_ = false;                          // Dantzig PRICE as a grid launch per pivot (run_ft): layout 2 with very many columns
    int32_t ft_tier_ = 0;                                 // 0 all in LDS, 1 big (ft_big_), 2 no per-row array in LDS (ft_big_ too)
    int64_t ft_zero_bytes_ = 0, ft_ones_bytes_ = 0;       // the two regions of the state buffer a refactorisation resets
    bool ft_need_refactor_ = false;
    relp_status_t ft_plan_and_alloc();
    relp_status_t ft_reset();
    relp_status_t ft_read_hdr();
    FtProblem ft_problem(int rule) const;
    void ft_enqueue_pivots(const FtState& go, int rule, int64_t left);
    char* d_pe_buf_ = nullptr; PriceEll pe_{};            // PRICE copy of the structural columns (relp_kernels.h: PriceEll)
    relp_status_t ft_build_price_ell();
    relp_status_t run_ft(int64_t max_iters, int64_t* done, int32_t* outcome);
    DeferredUpdate deferred() const;
    void enqueue_flush();
    int32_t n_alloc_ = 0;     // allocated tableau columns (artificial + provider)

    // ---- shards ----
    int32_t col_lo_ = 0, col_hi_ = 0, row_lo_ = 0, row_hi_ = 0, row_stride_ = 0;
    // [key, j, d_j, column (m)] and, tableau engine, the minimum ratio of every block of 256 rows
    int64_t candidate_len_for(int32_t m) const {
        return round_up_even(3 + (int64_t)m + (cfg_.engine == RELP_ENGINE_TABLEAU ? (m + 255) / 256 : 0));
    }
    static int64_t round_up_even(int64_t v) { return (v + 1) & ~int64_t(1); }
    int64_t cand_len_ = 0;
    // native multi-GPU loop: collective hooks and the message buffers they exchange
    relp_allgather_fn coll_allgather_ = nullptr;
    relp_allreduce_sum_fn coll_allreduce_ = nullptr;
    void* coll_ctx_ = nullptr;
    void* rccl_comm_ = nullptr;          // ncclComm_t owned by this engine (relp_rccl_attach)
    double* d_msg_cand_ = nullptr;       // this rank's candidate / all ranks' candidates
    double* d_msg_cands_ = nullptr;
    double* d_msg_slice_ = nullptr;      // revised engine: alpha slice / all slices, rho
    double* d_msg_slices_ = nullptr;
    double* d_msg_rho_ = nullptr;
    relp_status_t shard_iteration();
    relp_status_t shard_iteration_comm_only(int from_step);
    int coll_step_ = 0;                  // collectives of the current pivot already done (a failed pivot is completed from here)
    int64_t shadow_flush_ = 0;           // pivots since the last flush as every rank counts them
    relp_status_t shard_agree_on_status(relp_status_t local);
    double* d_msg_status_ = nullptr;     // this rank's status / all ranks' statuses (agreed on at every poll of relp_shard_run)
    double* d_msg_statuses_ = nullptr;
    int64_t inject_failure_after_ = -1;  // test hook (relp_shard_inject_failure)
    bool coll_broken_ = false;           // a collective hook itself failed: nothing can be agreed on any more
    relp_status_t remove_artificial_basis_variables_sharded(std::vector<int32_t>& rows_to_remove);
    void rccl_release();

    // ---- profiling ----
    bool prof_on_ = false;
    std::vector<hipEvent_t> prof_ev_;
    std::vector<int> prof_kid_;
    bool prof_open_ = false;
    int32_t prof_stride_ = 1;      // bracket the kernels of every prof_stride_-th pivot only
    int64_t prof_tick_ = 0;

    std::string err_;

    // helpers
    ColumnTable table() const;
    Tolerances tolerances() const;
    bool hip_ok(hipError_t e, const char* what);
    relp_status_t fail(relp_status_t code, const std::string& msg) { err_ = msg; return code; }
    relp_status_t download_rec();
    relp_status_t upload_rec();
    void prof_begin(int kid, hipStream_t on = nullptr);
    void prof_end(hipStream_t on = nullptr);
    void enqueue_price(int cost_mode, const double* vec, const PivotRecord* rec, int32_t p_lo, int32_t p_hi);
    void enqueue_iteration(int rule);
    relp_status_t finish_phase_one(int32_t* outcome);
    relp_status_t remove_artificial_basis_variables(std::vector<int32_t>& rows_to_remove);
    relp_status_t switch_to_phase_two(const std::vector<int32_t>& rows_to_remove);
    // RELP_ARTIFICIAL_TEXTBOOK: artificial variables no zero-level pivot could remove; remove_rows exchanges the basis position
    // each one sits in with its own row before both go (the pair (own constraint, position) always leaves a basis)
    std::vector<int32_t> stuck_artificials_;
    // relp_config_t.pivot_rescue (relp_engine.h): the loop of run_loop() behind a look at every exit without a pivot row
    relp_status_t run_loop(int64_t max_iters, int64_t* done, int32_t* outcome);
    relp_status_t rescue_unbar_all();
    void auto_reinversion_adapt(const std::vector<double>& before, const std::vector<double>& after);
    std::vector<int32_t> barred_;                  // columns barred from pricing (in_basis flag 2) until the basis changes
    // RELP_PIVOT_GUARD: a pivot element below this fraction of the column's largest |entry| ends the loop for a rescue.  0 = off,
    // the default: measured on 16 files of the corpus (profiles/r04_guard_sweep.md), 1e-5 / 1e-7 / 1e-9 fire on most pivots of the
    // ill-conditioned files and the ratio test without the small rows overshoots them -- wrong `infeasible` outcomes on four files
    double guard_rel_ = 0.0;
    bool pivot_guard_on_ = false;                  // Tolerances::pivot_guard of the launches (on inside the rescued loop only)
    bool hold_phase_end_ = false;                  // run_loop returns kHeldNoCandidate instead of ending the phase
    int64_t rescue_small_pivots_ = 0, rescue_barred_ = 0, rescue_confirmations_ = 0;
    double last_reinvert_drift_ = -1.0;            // relp_config_t.auto_reinversion: what the last rebuild moved b by (relative)
    relp_status_t remove_rows(const std::vector<int32_t>& rows);
    void free_all();
};

}  // namespace relp
