// relp_kernels_ft.hip -- the sparse LU engine's persistent pivot kernel: PRICE (CSC) -> FTRAN -> RATIO -> Forrest-Tomlin
// update -> BTRAN -> b, -pi, basis for MANY pivots in one launch of one workgroup, all work vectors, the eta file and the
// dense tail of U in LDS.  Reference rows a4 (LU), a8: carry/lower_upper/mod.rs:92-222, eta_file.rs:49-133,
// permutation/rotate_to_back.rs:15-110; see the layout notes at `FtState` in relp_kernels.h and the numpy design
// check scripts/ft_prototype.py.
//
// Why one workgroup: at Netlib sizes (m ~ 10^3, factors ~ 10^5 bytes) a pivot is a chain of ~150 dependent sparse
// steps over data that fits one CU's LDS; kernel boundaries (~6.5 us each, 8 per pivot before) and grid barriers
// (4.8 us at 64 workgroups) cost more than the arithmetic.  Everything between two refactorisations therefore runs
// behind __syncthreads only.
#include "relp_lu_device.h"

#include <type_traits>
#include <vector>

namespace relp {

namespace {

constexpr int NT = kFtThreads;
constexpr int NW = kFtWaves;
static_assert(kFtMaxSlots * 8 <= kFtThreads && kFtMaxSlots <= 64, "eight lanes per slot; one lane per slot in the chains");

__host__ __device__ inline int64_t up16(int64_t b) { return (b + 15) / 16 * 16; }

// byte offsets of the dynamic LDS arrays
struct FtLayout {
    int64_t x, sp, pi, perm, tc, dots, zt, uv, ct, slot_pivot, slot_prev, slot_live, slot_next, eta_off, spk_off, tslot, eta_idx,
        eta_val, red_d, red_i, bx, bsp, stage, total;
    int bx_words, bsp_words, bit_shift;               // (layout 2) the bitmaps of x and of the spike, one bit per 2^bit_shift entries
};
// `big` (FtState::big, m beyond ~2,400 rows): only what a pass of a solve touches stays in LDS -- x (with `rhs_cap` words for
// the right-hand-side copies of the fused schedules), -pi (PRICE gathers from it), the slot tables and the dense tail; the spike, the
// permutations and the eta pool, each read a few times per pivot, are read from global memory (L2).
// tier 2 (m beyond what 17 bytes per row leave of the LDS, ~9,000 rows): x, -pi and the pivot -> slot table are global too;
// the LDS holds the dense tail, the slot tables and the staged factor image, nothing that grows with m.
__host__ __device__ inline FtLayout ft_layout(int m, int tcap, int eta_cap, int tier, int rhs_cap) {
    FtLayout L;
    int64_t o = 0;
    const int ldt = tcap + 1;
    const bool big = tier >= 1, huge = tier >= 2;
    auto take = [&](int64_t bytes) { const int64_t at = o; o += up16(bytes); return at; };
    L.x = take(huge ? 0 : 8LL * (m + 1 + rhs_cap));           // x[m]: scratch word of ell_solve; x[m + 1 ..]: right-hand-side copies
    L.sp = take(big ? 0 : 8LL * m); L.pi = take(huge ? 0 : 8LL * m);
    L.perm = take(big ? 0 : 2LL * 3 * m);            // inv_rowperm | inv_colperm | rowperm as 16-bit indices
    L.tc = take(8LL * tcap * ldt);
    L.dots = take(8LL * tcap); L.zt = take(8LL * tcap); L.uv = take(8LL * tcap); L.ct = take(8LL * tcap);
    L.slot_pivot = take(4LL * tcap); L.slot_prev = take(4LL * tcap); L.slot_live = take(4LL * tcap); L.slot_next = take(4LL * tcap);
    L.eta_off = take(4LL * tcap * (NW + 1)); L.spk_off = take(4LL * tcap * (NW + 1));
    L.tslot = take(huge ? 0 : m);
    L.eta_idx = take(big ? 0 : 4LL * eta_cap); L.eta_val = take(big ? 0 : 8LL * eta_cap);
    L.red_d = take(8LL * 2 * NW); L.red_i = take(4LL * 4 * NW + 64);
    L.bit_shift = huge ? ft_bitmap_shift(m, rhs_cap) : 0;
    L.bx_words = huge ? (int)(((((int64_t)m + 1 + rhs_cap) >> L.bit_shift) + 32) / 32) : 0;
    L.bsp_words = huge ? (int)(((((int64_t)m) >> L.bit_shift) + 32) / 32) : 0;
    L.bx = take(4LL * L.bx_words); L.bsp = take(4LL * L.bsp_words);
    L.stage = o; L.total = o;
    return L;
}

// Phase clock of the persistent kernel (thread 0, s_memtime): where a pivot's time goes (relp_lu_phase_cycles).
// Diagnostic builds (make CXXFLAGS="... -DPRICE_DIAG" / -DUT_DIAG) put extra laps inside PRICE / the U' solve and book
// them on the small phases FT_LOAD_STORE, FT_B, FT_SCATTER, FT_VECTORS (subtract those phases' normal shares): that is how
// the per-element branches of the chain coefficients and the dependent loads of the long columns were found.
enum FtPhase { FT_PRICE = 0, FT_SCATTER, FT_L, FT_ETA_FWD, FT_PUSH, FT_U, FT_RATIO, FT_B, FT_UBAR, FT_UT, FT_COMPACT, FT_BT_CHAIN,
               FT_LT, FT_VECTORS, FT_LOAD_STORE, FT_STAGE, FT_PHASES };
struct FtClock {
    long long acc[FT_PHASES]; long long last; bool on;
    long long passes[4], sweeps[4], total[4];        // per schedule: passes walked, sweeps, passes of the whole schedule
    long long nnz[4];                                // non-zeros met: entering column alpha, eta row, spike, (pivots counted)
    __device__ __forceinline__ void start(const long long* out) {
        on = out != nullptr && threadIdx.x == 0;
        for (int k = 0; k < FT_PHASES; ++k) acc[k] = 0;
        for (int k = 0; k < 4; ++k) { passes[k] = 0; sweeps[k] = 0; total[k] = 0; nnz[k] = 0; }
        last = on ? clock64() : 0;
    }
    __device__ __forceinline__ void lap(int phase) {            // phase is a constant at every call site: acc stays in registers
        if (on) { const long long now = clock64(); acc[phase] += now - last; last = now; }
    }
    __device__ __forceinline__ void flush(long long* out, bool with_nnz) {
        if (on) {
            for (int k = 0; k < FT_PHASES; ++k) out[k] += acc[k];
            for (int k = 0; k < 4; ++k) { out[16 + k] += passes[k]; out[20 + k] += sweeps[k]; out[24 + k] += total[k]; if (with_nnz) out[28 + k] += nnz[k]; }
        }
    }
};

template <int kTier>
struct FtCtxT {
    static constexpr int tier = kTier;
    static constexpr bool big = kTier >= 1, huge = kTier >= 2;
    typedef typename std::conditional<big, int32_t, unsigned short>::type perm_t;
    typedef typename std::conditional<huge, int32_t, signed char>::type tslot_t;
    // the wavefront that owns pivot k in the bucketed lists (eta pool, spike pool): see ft_compact
    static __device__ __forceinline__ int bucket_of(int k) { return big ? (k >> 6) % NW : k % NW; }
    unsigned long long* chunk_mask;                    // (layout 2) ceil(m / 64) ballots, scratch of ft_compact
    // (layout 2) x and the spike as sparse vectors: dense arrays in L2 plus "may be non-zero" bitmaps in LDS, a bit per 2^gs
    // entries.  Invariant: bit clear => the entries are 0.0 in memory.  Every store of a value that may be non-zero marks
    // (hs_mark); every per-row loop walks the set bits (hs_for_each) -- an entering column of a 120,000-row multi-commodity
    // LP has 13 non-zeros, its eta row 16, its spike 2.
    uint32_t *bx, *bsp;
    int gs, bx_words, bsp_words;
    double *x, *sp, *pi, *TC, *dots, *zt, *uv, *ct, *eta_val, *red_d;      // (big: sp, eta_val, eta_idx and the permutations are global)
    int *slot_pivot, *slot_prev, *slot_live, *slot_next, *eta_off, *spk_off, *eta_idx, *red_i;
    tslot_t* tslot;
    const perm_t *irp, *icp, *rp;                      // original row -> pivot, basis position -> pivot, pivot -> original row
    char* stage;
    int m, tcap, ldt, t, eta_used, eta_cap, journal_n;
    FtClock clk;
};

template <class Ctx>
__device__ __forceinline__ void ft_bind(Ctx& c, char* lds, const DeviceLU& lu, const FtState& st) {
    const FtLayout L = ft_layout(st.m, st.tcap, st.eta_cap, Ctx::tier, st.rhs_cap);
    if constexpr (Ctx::huge) {                         // (-pi: bound by ft_load to the engine's own vector)
        c.x = st.x_work; c.pi = nullptr; c.tslot = st.tslot; c.chunk_mask = st.chunk_mask;
        c.bx = (uint32_t*)(lds + L.bx); c.bsp = (uint32_t*)(lds + L.bsp); c.gs = L.bit_shift; c.bx_words = L.bx_words; c.bsp_words = L.bsp_words;
    }
    else { c.x = (double*)(lds + L.x); c.pi = (double*)(lds + L.pi); c.tslot = (signed char*)(lds + L.tslot); }
    c.TC = (double*)(lds + L.tc);
    if constexpr (Ctx::big) {
        c.chunk_mask = st.chunk_mask;
        c.sp = st.sp_work; c.irp = st.inv_rowperm; c.icp = st.inv_colperm; c.rp = lu.rowperm;
        c.eta_idx = st.eta_idx; c.eta_val = st.eta_val;
    } else {
        c.sp = (double*)(lds + L.sp);
        c.irp = (const unsigned short*)(lds + L.perm); c.icp = c.irp + st.m; c.rp = c.icp + st.m;
        c.eta_idx = (int*)(lds + L.eta_idx); c.eta_val = (double*)(lds + L.eta_val);
    }
    c.dots = (double*)(lds + L.dots); c.zt = (double*)(lds + L.zt); c.uv = (double*)(lds + L.uv); c.ct = (double*)(lds + L.ct);
    c.slot_pivot = (int*)(lds + L.slot_pivot); c.slot_prev = (int*)(lds + L.slot_prev); c.slot_live = (int*)(lds + L.slot_live);
    c.slot_next = (int*)(lds + L.slot_next); c.eta_off = (int*)(lds + L.eta_off); c.spk_off = (int*)(lds + L.spk_off);
    c.red_d = (double*)(lds + L.red_d); c.red_i = (int*)(lds + L.red_i); c.stage = lds + L.stage;
    c.m = st.m; c.tcap = st.tcap; c.ldt = st.ldt; c.eta_cap = st.eta_cap; c.journal_n = 0;
    c.clk.start(nullptr);
}

// ---- layout 2: sparse-vector helpers ---------------------------------------------------------------------------------------
__device__ __forceinline__ void hs_mark(uint32_t* bits, int gs, int k) { atomicOr(&bits[(k >> gs) >> 5], 1u << ((k >> gs) & 31)); }
__device__ __forceinline__ bool hs_test(const uint32_t* bits, int gs, int k) { return (bits[(k >> gs) >> 5] >> ((k >> gs) & 31)) & 1u; }
// f(k) for every entry k < n whose bit is set; a thread per word, its bits one after the other (the vectors hold a few dozen
// non-zeros among 10^5 entries: most threads find nothing).  No barrier inside; the caller decides.
template <class F>
__device__ __forceinline__ void hs_for_each(const uint32_t* bits, int gs, int first, int n, F f) {
    const int w0 = (first >> gs) >> 5, w1 = (((n - 1) >> gs) >> 5) + 1;
    for (int w = w0 + (int)threadIdx.x; w < w1; w += NT) {
        uint32_t word = bits[w];
        while (word) {
            const int b = __ffs((int)word) - 1;
            word &= word - 1;
            const int k0 = ((w << 5) + b) << gs;
            if (gs == 0) { if (k0 >= first && k0 < n) f(k0); }
            else for (int k = max(k0, first); k < min(k0 + (1 << gs), n); ++k) f(k);
        }
    }
}
// vec := 0 wherever its bitmap says it may not be, bitmap := empty.  Barriers on both sides.
__device__ __forceinline__ void hs_clear(double* vec, uint32_t* bits, int gs, int words, int n) {
    __syncthreads();
    hs_for_each(bits, gs, 0, n, [&](int k) { vec[k] = 0.0; });
    __syncthreads();
    for (int w = threadIdx.x; w < words; w += NT) bits[w] = 0u;
    __syncthreads();
}
// Kernel entry: the bitmaps as the previous launch left them (FtState::bits_save), or -- unknown contents -- both vectors zeroed
// densely.  Kernel exit: hs_leave(save = true) by the kernels that keep the invariant to the end.
template <class Ctx>
__device__ __forceinline__ void hs_enter(Ctx& c, const FtState& st) {
    if constexpr (Ctx::huge) {
        const int tid = threadIdx.x;
        if (st.nzc[2] == 1) {
            for (int w = tid; w < c.bx_words; w += NT) c.bx[w] = st.bits_save[w];
            for (int w = tid; w < c.bsp_words; w += NT) c.bsp[w] = st.bits_save[c.bx_words + w];
        } else {
            for (int k = tid; k < c.m + 1 + st.rhs_cap; k += NT) c.x[k] = 0.0;
            for (int k = tid; k < c.m; k += NT) c.sp[k] = 0.0;
            for (int w = tid; w < c.bx_words; w += NT) c.bx[w] = 0u;
            for (int w = tid; w < c.bsp_words; w += NT) c.bsp[w] = 0u;
        }
        __syncthreads();
    }
}
template <class Ctx>
__device__ __forceinline__ void hs_leave(Ctx& c, const FtState& st, bool save) {
    if constexpr (Ctx::huge) {
        __syncthreads();
        const int tid = threadIdx.x;
        if (save) {
            for (int w = tid; w < c.bx_words; w += NT) st.bits_save[w] = c.bx[w];
            for (int w = tid; w < c.bsp_words; w += NT) st.bits_save[c.bx_words + w] = c.bsp[w];
        }
        if (tid == 0) st.nzc[2] = save ? 1 : 0;
    }
}
// x[k] := v (a value that may be non-zero)
template <class Ctx>
__device__ __forceinline__ void xset(Ctx& c, int k, double v) {
    c.x[k] = v;
    if constexpr (Ctx::huge) hs_mark(c.bx, c.gs, k);
}
// x[k] := value(k) for every k < m (a dense right-hand side of the step-wise API).  Ends with a barrier.
template <class Ctx, class F>
__device__ __forceinline__ void xfill(Ctx& c, const FtState& st, F value) {
    if constexpr (Ctx::huge) {
        hs_clear(c.x, c.bx, c.gs, c.bx_words, c.m + 1 + st.rhs_cap);
        for (int k = threadIdx.x; k < c.m; k += NT) { const double v = value(k); if (v != 0.0) xset(c, k, v); }
    } else {
        for (int k = threadIdx.x; k < c.m; k += NT) c.x[k] = value(k);
    }
    __syncthreads();
}
// x := 0 (ends with a barrier)
template <class Ctx>
__device__ __forceinline__ void xzero(Ctx& c, const FtState& st) {
    if constexpr (Ctx::huge) hs_clear(c.x, c.bx, c.gs, c.bx_words, c.m + 1 + st.rhs_cap);
    else { for (int k = threadIdx.x; k < c.m; k += NT) c.x[k] = 0.0; __syncthreads(); }
}

// global state -> LDS (ends with a barrier)
template <class Ctx>
__device__ __forceinline__ void ft_load(Ctx& c, const DeviceLU& lu, const FtState& st, const double* minus_pi) {
    const int tid = threadIdx.x;
    c.t = st.hdr[0]; c.eta_used = st.hdr[1];
    if constexpr (!Ctx::big) {   // the permutations as 16-bit copies: every pivot needs a few entries of each, a global round trip apiece otherwise
        unsigned short* w = const_cast<unsigned short*>(c.irp);
        for (int k = tid; k < c.m; k += NT) {
            w[k] = (unsigned short)st.inv_rowperm[k]; w[c.m + k] = (unsigned short)st.inv_colperm[k];
            w[2 * c.m + k] = (unsigned short)lu.rowperm[k];
        }
    }
    for (int i = tid; i < c.tcap * c.ldt; i += NT) c.TC[i] = st.TC[i];
    for (int s = tid; s < c.tcap; s += NT) {
        c.slot_pivot[s] = st.slot_pivot[s]; c.slot_prev[s] = st.slot_prev[s]; c.slot_live[s] = st.slot_live[s];
        c.slot_next[s] = -1;
    }
    for (int i = tid; i < c.tcap * (NW + 1); i += NT) { c.eta_off[i] = st.eta_off[i]; c.spk_off[i] = st.spk_off[i]; }
    if constexpr (!Ctx::huge) for (int k = tid; k < c.m; k += NT) c.tslot[k] = (signed char)st.tslot[k];
    if constexpr (!Ctx::big) for (int e = tid; e < c.eta_used; e += NT) { c.eta_idx[e] = st.eta_idx[e]; c.eta_val[e] = st.eta_val[e]; }
    if constexpr (Ctx::huge) c.pi = const_cast<double*>(minus_pi);
    else if (minus_pi) for (int i = tid; i < c.m; i += NT) c.pi[i] = minus_pi[i];
    __syncthreads();
    for (int s = tid; s < c.t; s += NT) { const int pv = c.slot_prev[s]; if (pv >= 0) c.slot_next[pv] = s; }
    __syncthreads();
}

// LDS -> global state (what an update may have changed)
template <class Ctx>
__device__ __forceinline__ void ft_store(const Ctx& c, const FtState& st, double* minus_pi, int need_refactor) {
    const int tid = threadIdx.x;
    __syncthreads();
    for (int i = tid; i < c.tcap * c.ldt; i += NT) st.TC[i] = c.TC[i];
    for (int s = tid; s < c.tcap; s += NT) {
        st.slot_pivot[s] = c.slot_pivot[s]; st.slot_prev[s] = c.slot_prev[s]; st.slot_live[s] = c.slot_live[s];
    }
    for (int i = tid; i < c.tcap * (NW + 1); i += NT) { st.eta_off[i] = c.eta_off[i]; st.spk_off[i] = c.spk_off[i]; }
    if constexpr (!Ctx::huge) for (int k = tid; k < c.m; k += NT) st.tslot[k] = c.tslot[k];
    if constexpr (!Ctx::big) for (int e = tid; e < c.eta_used; e += NT) { st.eta_idx[e] = c.eta_idx[e]; st.eta_val[e] = c.eta_val[e]; }
    if constexpr (!Ctx::huge) if (minus_pi) for (int i = tid; i < c.m; i += NT) minus_pi[i] = c.pi[i];
    if (tid == 0) { st.hdr[0] = c.t; st.hdr[1] = c.eta_used; st.hdr[2] = need_refactor; st.hdr[3] = c.journal_n; }
}

// value of lane `src` (wave-uniform) in every lane
__device__ __forceinline__ double lane_bcast(double v, int src) {
    src = __builtin_amdgcn_readfirstlane(src);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// Triangular chain over the update slots by ONE wavefront (lane = slot), the sequential core of the eta file and of the
// dense tail of U: for s2 in order, lane s2 publishes w = (base - sub - acc) * scale, and every lane "after" s2 accumulates
// its coefficient times that value.  kRow: the coefficient of (lane, s2) is TC[lane][s2], else TC[s2][lane].  `link`: when the
// published slot is this lane's predecessor (the same pivot updated earlier / later), its value replaces `base`.
// Branch-free: w is kept up to date incrementally (w -= coef * scale * v), the link is folded into the coefficient
// (coef - 1 at s2 = link adds v to the base), coefficients are fetched eight steps ahead and are 0 wherever a step must not
// act, so a step is two v_readlane and one v_fma; steps beyond t (the last chunk) multiply by 0.  A lane's w no longer
// changes once its own step has passed (its later coefficients are exact zeros), so the value it published is simply its
// final w: no per-step capture.
// kLink / kScale: the chain uses `link` / a `scale` other than 1 (the eta chains link and do not scale, the solves with the
// dense tail scale and do not link: three to four instructions less per coefficient)
template <bool kAsc, bool kRow, bool kLink, bool kScale>
__device__ __forceinline__ double tc_chain(const double* TC, int ldt, int t, int lane, double base, double sub, double scale,
                                           int link) {
    constexpr int CH = 8;
    const bool in = lane < t;
    const int row = in ? lane : 0;
    double w = in ? (base - sub) * scale : 0.0;
    double cur[CH], nxt[CH];
    auto load = [&](int ch, double* buf) {
        double cf[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int o = ch * CH + j, s2 = kAsc ? o : t - 1 - o;
            const int sc = s2 < 0 ? 0 : (s2 >= t ? t - 1 : s2);                 // (clamped: the load itself is unconditional)
            cf[j] = kRow ? TC[row * ldt + sc] : TC[sc * ldt + row];
        }
        // (all eight loads are issued before any is used, and none is sunk into the branch below: one wait per chunk)
        asm volatile("" : "+v"(cf[0]), "+v"(cf[1]), "+v"(cf[2]), "+v"(cf[3]), "+v"(cf[4]), "+v"(cf[5]), "+v"(cf[6]), "+v"(cf[7]));
        // (straight-line selects: "after" is false by itself for the steps beyond t -- s2 >= t > lane ascending, s2 < 0 <= lane
        // descending -- and written as a branch per element this block cost more than the eight chain steps it feeds)
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int o = ch * CH + j, s2 = kAsc ? o : t - 1 - o;
            const bool after = in && (kAsc ? lane > s2 : lane < s2);
            const double v = kLink ? cf[j] - (link == s2 ? 1.0 : 0.0) : cf[j];
            buf[j] = kScale ? v * (after ? scale : 0.0) : (after ? v : 0.0);
        }
    };
    const int nch = (t + CH - 1) / CH;
    load(0, cur);
    for (int ch = 0; ch < nch; ++ch) {
        load(ch + 1 < nch ? ch + 1 : ch, nxt);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int o = ch * CH + j, s2 = kAsc ? o : t - 1 - o;
            const double v = lane_bcast(w, s2 < 0 ? 0 : s2);
            w = fma(-cur[j], v, w);
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) cur[j] = nxt[j];
    }
    return w;
}

// lanes per update slot when all t slots are worked on at once: 64, 32, 16 or 8 (as a shift)
__device__ __forceinline__ int slot_lanes_shift(int t) { return t <= NT / 64 ? 6 : t <= NT / 32 ? 5 : t <= NT / 16 ? 4 : 3; }

// `which`: 0 L, 1 U, 2 U', 3 L'.  first_level: levels below it hold zeros only (0 = everything).
template <class Ctx> __device__ __forceinline__ int block_min_int(Ctx& c, int v);

// Hyper-sparse start of a sweep (the reference's solves only touch what they reach: lower_upper/mod.rs:236-271, 359-374 walk a
// BTreeMap of non-zeros): the first group of the schedule in which a non-zero of x matters (EllSchedule::reach).  One
// coalesced pass over reach (its loads do not depend on x), one block reduction.  x must be complete (behind a barrier).
template <class Ctx>
__device__ __forceinline__ int ft_first_group(Ctx& c, const int32_t* __restrict__ reach) {
    int g = 0x7fffffff;
    if constexpr (Ctx::huge) {
        hs_for_each(c.bx, c.gs, 0, c.m, [&](int k) { if (c.x[k] != 0.0) g = min(g, reach[k]); });
        return block_min_int(c, g);
    }
    for (int k0 = threadIdx.x; k0 < c.m; k0 += 4 * NT) {
        int r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = reach[min(k0 + u * NT, c.m - 1)];
        asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));
#pragma unroll
        for (int u = 0; u < 4; ++u) if (k0 + u * NT < c.m && c.x[k0 + u * NT] != 0.0) g = min(g, r[u]);
    }
    return block_min_int(c, g);
}

// `hyper`: derive the first level from the non-zeros of x (st.hyper selects the schedules for which that pays)
template <class Ctx>
__device__ __forceinline__ void sweep(const FtState& st, int which, Ctx& c, int first_level = 0, bool hyper = false) {
    auto lap = [&]() { c.clk.lap(FT_STAGE); };
    if constexpr (Ctx::big) {                          // (big layout only, see ell_stage)
        if (hyper && ((st.hyper >> which) & 1)) first_level = max(first_level, ft_first_group(c, st.ell[which].reach));
    }
    int passes;
    uint32_t* bx = nullptr;
    int gs = 0;
    if constexpr (Ctx::huge) { bx = c.bx; gs = c.gs; }
    // (layout 2: the pass headers of an image that is not staged are copied into the staging area when they fit)
    int4* hdr_lds = 16LL * (st.ell[which].n_passes + kEllPadHeaders) <= st.stage_bytes ? reinterpret_cast<int4*>(c.stage) : nullptr;
    if (st.stage[which]) passes = ell_solve_pp<true, NT, Ctx::big, decltype(lap), Ctx::huge>(st.ell[which], c.stage, c.x, c.m, first_level, lap, bx, gs, st.rhs_cap);
    else passes = ell_solve_pp<false, NT, Ctx::big, decltype(lap), Ctx::huge>(st.ell[which], c.stage, c.x, c.m, first_level, lap, bx, gs, st.rhs_cap, hdr_lds);
    if (c.clk.on) { c.clk.passes[which] += passes; c.clk.sweeps[which] += 1; c.clk.total[which] += st.ell[which].n_passes; }
}

// ---- FTRAN: x = P a on entry (pivot-indexed); x = U^-1 R_t .. R_1 L^-1 (P a) on exit, the spike in sp -----------------
// lower_upper/mod.rs:157-190
// spike_only: stop once the spike (what an update needs) is formed, before the solve with U
template <class Ctx>
__device__ __forceinline__ void ft_ftran(const DeviceLU& lu, const FtState& st, Ctx& c, bool spike_only = false) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = c.t;
    sweep(st, 0, c, 0, true);
    c.clk.lap(FT_L);
    if (t > 0) {
        // eta.apply_right for every update (eta_file.rs:72-109): w[p_s] -= r_s . w.  The sparse parts read entries no
        // eta modifies (pivots never updated at the time), so all of them are formed first, a slot per wavefront ...
        {   // all slots at once, 8 to 64 lanes per slot (as many as the workgroup has for t slots)
            const int lg = slot_lanes_shift(t), G = 1 << lg;
            const int s = tid >> lg, l = tid & (G - 1);
            double sum = 0.0;
            if (s < t) {
                const int e0 = c.eta_off[s * (NW + 1)], e1 = c.eta_off[s * (NW + 1) + NW];
                if constexpr (Ctx::big) {              // (the pool in L2 -- in layout 2 x too: four entries requested before the first is used)
                    for (int e = e0 + l; e < e1; e += 4 * G) {
                        int k[4];
                        double v[4], xv[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) { const int ee = min(e + u * G, e1 - 1); k[u] = c.eta_idx[ee]; v[u] = c.eta_val[ee]; }
                        asm volatile("" : "+v"(k[0]), "+v"(k[1]), "+v"(k[2]), "+v"(k[3]));
#pragma unroll
                        for (int u = 0; u < 4; ++u) xv[u] = c.x[k[u]];
                        asm volatile("" : "+v"(xv[0]), "+v"(xv[1]), "+v"(xv[2]), "+v"(xv[3]));
#pragma unroll
                        for (int u = 0; u < 4; ++u) sum = fma(e + u * G < e1 ? v[u] : 0.0, xv[u], sum);
                    }
                } else
                for (int e = e0 + l; e < e1; e += G) sum = fma(c.eta_val[e], c.x[c.eta_idx[e]], sum);
            }
            sum = group_sum(sum, G);
            if (s < t && l == 0) c.dots[s] = sum;
        }
        __syncthreads();
        // ... then the chain over the updated pivots (lane = slot), followed by the solve with the dense tail of U
        if (wave == 0) {
            const int s = lane;
            const bool in = s < t;
            const int prev = in ? c.slot_prev[s] : -1, piv = in ? c.slot_pivot[s] : 0;
            const bool live = in && c.slot_live[s];
            const double val = tc_chain<true, true, true, false>(c.TC, c.ldt, t, s, (in && prev < 0) ? c.x[piv] : 0.0, in ? c.dots[s] : 0.0, 1.0, prev);
            if (live) xset(c, piv, val);
            // U z = w on the updated pivots: they are last in the order, so they are solved first (from the back)
            const double rdiag = in ? 1.0 / c.TC[s * c.ldt + s] : 1.0;
            const double z = tc_chain<false, true, false, true>(c.TC, c.ldt, t, s, live ? val : 0.0, 0.0, rdiag, -1);
            if (in) c.zt[s] = live ? z : 0.0;
        }
        __syncthreads();
    }
    c.clk.lap(FT_ETA_FWD);
    if constexpr (Ctx::huge) {                         // the spike (mod.rs:176): the old one cleared, the non-zeros of x copied
        hs_clear(c.sp, c.bsp, c.gs, c.bsp_words, c.m);
        hs_for_each(c.bx, c.gs, 0, c.m, [&](int k) { c.sp[k] = c.x[k]; });
        for (int w = tid; w < c.bsp_words; w += NT) c.bsp[w] = c.bx[w];
    } else
    for (int k = tid; k < c.m; k += NT) c.sp[k] = c.x[k];          // the spike (mod.rs:176)
    if (spike_only) { __syncthreads(); return; }
    if (t > 0) {
        __syncthreads();
        // the spike columns of the updated pivots act on the never-updated rows: x[k] -= U[k, p_s] z_s.  One wavefront per
        // bucket of rows (k % NW), slots in order: no two lanes ever touch one row, the summation order is fixed
        // (lane s keeps slot s's z and bucket bounds in registers; the pairs of the next slot are in flight while this
        // one is applied: the loop touches LDS only for the read-modify-write of x)
        {
            const bool on = lane < t && c.slot_live[lane] && c.zt[lane] != 0.0;
            const double zreg = lane < t ? c.zt[lane] : 0.0;
            const int o0 = lane < t ? c.spk_off[lane * (NW + 1) + wave] : 0, o1 = lane < t ? c.spk_off[lane * (NW + 1) + wave + 1] : 0;
            unsigned long long mask = __ballot(on);
            int pk = -1, pe0 = 0, pe1 = 0, ps = 0;
            double pv = 0.0;
            auto fetch = [&]() {
                ps = __builtin_amdgcn_readfirstlane(__ffsll((long long)mask) - 1);
                mask &= mask - 1;
                pe0 = __builtin_amdgcn_readlane(o0, ps); pe1 = __builtin_amdgcn_readlane(o1, ps);
                pk = -1;
                if (pe0 + lane < pe1) { pk = st.spk_idx[ps * c.m + pe0 + lane]; pv = st.spk_val[ps * c.m + pe0 + lane]; }
            };
            bool more = mask != 0;
            if (more) fetch();
            while (more) {
                const int cs = ps, ck = pk, ce0 = pe0, ce1 = pe1;
                const double cv = pv;
                more = mask != 0;
                if (more) fetch();
                const double z = lane_bcast(zreg, cs);
                if (ck >= 0 && c.tslot[ck] < 0) xset(c, ck, fma(-cv, z, c.x[ck]));
                for (int e = ce0 + lane + 64; e < ce1; e += 64) {
                    const int k = st.spk_idx[cs * c.m + e];
                    if (c.tslot[k] < 0) xset(c, k, fma(-st.spk_val[cs * c.m + e], z, c.x[k]));
                }
                wave_fence();
            }
        }
        __syncthreads();
        if (tid < t && c.slot_live[tid]) c.x[c.slot_pivot[tid]] = 0.0;      // masked in the U0 sweep
        __syncthreads();
    }
    c.clk.lap(FT_PUSH);
    sweep(st, 1, c);                                   // (no hyper-sparse start: the spike reaches the first groups of U, measured)
    if (t > 0) {
        if (tid < t && c.slot_live[tid]) xset(c, c.slot_pivot[tid], c.zt[tid]);
        __syncthreads();
    }
    c.clk.lap(FT_U);
}

// ---- y' U = c' for the current U: c in x (pivot-indexed) on entry; y over the never-updated pivots in x, over the
// slots in zt on exit (invert_upper_left, mod.rs:332-356, on the unrotated representation) ------------------------------
// first_level: the sweep over U0' may start at this level (everything below is known to be zero)
template <class Ctx>
__device__ __forceinline__ void ft_ut_solve(const DeviceLU& lu, const FtState& st, Ctx& c, bool do_sweep, int first_level = 0) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = c.t;
    if (t > 0) {
        if (tid < t) {
            const bool live = c.slot_live[tid];
            const int piv = c.slot_pivot[tid];
            c.ct[tid] = live ? c.x[piv] : 0.0;
            if (live) c.x[piv] = 0.0;
        }
        __syncthreads();
    }
#ifdef UT_DIAG
    c.clk.lap(FT_LOAD_STORE);
#endif
    if (do_sweep) sweep(st, 2, c, first_level);
#ifdef UT_DIAG
    c.clk.lap(FT_UT);
#endif
    if (t > 0) {
        {   // spike column of slot s against y over the never-updated pivots: all slots at once, 8 to 64 lanes per slot.  The
            // spike pool is global (L2): four entries per lane are requested before the first is used -- a spike has up to a
            // few hundred entries, and one entry per dependent round trip made this loop a tenth of the pivot.
            const int lg = slot_lanes_shift(t), G = 1 << lg;
            const int s = tid >> lg, l = tid & (G - 1);
            double sum = 0.0;
            if (s < t && do_sweep && c.slot_live[s]) {
                const int base = s * c.m;
                const int e0 = c.spk_off[s * (NW + 1)], e1 = c.spk_off[s * (NW + 1) + NW];
                for (int e = e0 + l; e < e1; e += 4 * G) {
                    int k[4];
                    double v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const int ee = min(e + u * G, e1 - 1); k[u] = st.spk_idx[base + ee]; v[u] = st.spk_val[base + ee]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool on = e + u * G < e1 && c.tslot[k[u]] < 0;
                        sum = fma(on ? v[u] : 0.0, c.x[k[u]], sum);
                    }
                }
            }
            sum = group_sum(sum, G);
            if (s < t && l == 0) c.dots[s] = sum;
        }
        __syncthreads();
#ifdef UT_DIAG
        c.clk.lap(FT_B);
#endif
        if (wave == 0) {
            const int s = lane;
            const bool in = s < t;
            const double rdiag = in ? 1.0 / c.TC[s * c.ldt + s] : 1.0;
            const double y = tc_chain<true, false, false, true>(c.TC, c.ldt, t, s, in ? c.ct[s] - c.dots[s] : 0.0, 0.0, rdiag, -1);
            if (in) c.zt[s] = c.slot_live[s] ? y : 0.0;
        }
        __syncthreads();
    }
#ifdef UT_DIAG
    c.clk.lap(FT_VECTORS);
#else
    c.clk.lap(FT_UT);
#endif
}

// ---- BTRAN: c (pivot-indexed, i.e. Q' c) in x on entry; w = P z with z' B = c' on exit (mod.rs:204-222) ------------------
template <class Ctx>
__device__ __forceinline__ void ft_btran(const DeviceLU& lu, const FtState& st, Ctx& c, bool do_sweep, int first_level = 0) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = c.t;
    ft_ut_solve(lu, st, c, do_sweep, first_level);
    if (t > 0) {
        // eta.apply_left in reverse order (eta_file.rs:49-65): v[j] -= r_s[j] v[p_s]; first the chain over the slots ...
        if (wave == 0) {
            const int s = lane;
            const bool in = s < t;
            const int nxt = in ? c.slot_next[s] : -1, prev = in ? c.slot_prev[s] : -1;
            // (a slot without successor is live: its value starts from y)
            const double u = tc_chain<false, false, true, false>(c.TC, c.ldt, t, s, (in && nxt < 0) ? c.zt[s] : 0.0, 0.0, 1.0, nxt);
            if (in) {
                c.uv[s] = u;
                if (prev < 0) xset(c, c.slot_pivot[s], u);       // the value every earlier eta (sparse part) and L' see
            }
        }
        __syncthreads();
        // ... then the sparse parts, one wavefront per bucket of pivots (j % NW), slots in reverse order; the pairs of the
        // next slot are fetched while this one is applied
        {
            const double ureg = lane < t ? c.uv[lane] : 0.0;
            const int o0 = lane < t ? c.eta_off[lane * (NW + 1) + wave] : 0, o1 = lane < t ? c.eta_off[lane * (NW + 1) + wave + 1] : 0;
            unsigned long long mask = __ballot(lane < t && ureg != 0.0 && o1 > o0);
            int pj = -1, pe0 = 0, pe1 = 0, ps = 0;
            double pv = 0.0;
            auto fetch = [&]() {
                ps = __builtin_amdgcn_readfirstlane(63 - __clzll((long long)mask));
                mask &= ~(1ull << ps);
                pe0 = __builtin_amdgcn_readlane(o0, ps); pe1 = __builtin_amdgcn_readlane(o1, ps);
                pj = -1;
                if (pe0 + lane < pe1) { pj = c.eta_idx[pe0 + lane]; pv = c.eta_val[pe0 + lane]; }
            };
            bool more = mask != 0;
            if (more) fetch();
            while (more) {
                const int cs = ps, cj = pj, ce0 = pe0, ce1 = pe1;
                const double cv = pv;
                more = mask != 0;
                if (more) fetch();
                const double u = lane_bcast(ureg, cs);
                if (cj >= 0) xset(c, cj, fma(-cv, u, c.x[cj]));
                for (int e = ce0 + lane + 64; e < ce1; e += 64) {
                    const int j = c.eta_idx[e];
                    xset(c, j, fma(-c.eta_val[e], u, c.x[j]));
                }
                wave_fence();
            }
        }
        __syncthreads();
    }
    c.clk.lap(FT_BT_CHAIN);
    sweep(st, 3, c, 0, true);
    c.clk.lap(FT_LT);
}

// Bucketed, order-preserving compaction of the non-zeros of vec over the never-updated pivots (pivot `skip` excluded):
// bucket w = pivots with k % NW == w, written by wavefront w at [base + off[w], base + off[w + 1]).  off (NW + 1 ints) is
// left in `off_out` relative to off_base.  Returns the total (all threads).  Ends with a barrier.
// `room`: entries the destination can still take; when the non-zeros do not fit nothing is written and -1 is returned.
// Layout 2 (vec and the slot table in L2; `bits` = the bitmap of vec): bucket w = the 64-pivot chunks w, w + NW, ..
// (Ctx::bucket_of).  A wavefront looks up 64 of its chunks in the bitmap at a time (a lane per chunk) and visits those that
// may hold anything: an eta row or a spike has a few dozen non-zeros among 10^5 pivots.  The first pass leaves every visited
// chunk's ballot in `chunk_mask` for the second.
template <class Ctx, class IdxPtr, class ValPtr>
__device__ __forceinline__ int ft_compact(Ctx& c, const double* vec, const uint32_t* bits, int skip, IdxPtr out_idx, ValPtr out_val,
                                          int out_base, int* off_out, int off_base, int room) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if constexpr (Ctx::huge) {
        const int n_chunks = (c.m + 63) >> 6;
        const int mine_chunks = (n_chunks - wave + NW - 1) / NW;       // chunks wave + NW * i, i < mine_chunks
        unsigned long long* masks = c.chunk_mask;
        const int gs = c.gs;
        auto chunk_may = [&](int ch) -> bool {                          // any bit of the chunk's 64 entries set?
            const int fb = (ch << 6) >> gs;
            if (gs == 0) return (bits[fb >> 5] | bits[(fb >> 5) + 1]) != 0u;
            const int nb = 64 >> gs;                                    // (gs <= 5: the field lies inside one word)
            return ((bits[fb >> 5] >> (fb & 31)) & ((nb >= 32) ? 0xffffffffu : ((1u << nb) - 1u))) != 0u;
        };
        int cnt = 0;
        for (int i0 = 0; i0 < mine_chunks; i0 += 64) {
            const int i = i0 + lane;
            unsigned long long todo = __ballot(i < mine_chunks && chunk_may(wave + NW * i));
            while (todo) {
                const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
                todo &= todo - 1;
                const int ch = wave + NW * (i0 + src), k = (ch << 6) + lane;
                bool on = k < c.m && k != skip && hs_test(bits, gs, k);
                if (on) on = c.tslot[k] < 0 && vec[k] != 0.0;
                const unsigned long long mask = __ballot(on);
                if (lane == 0) masks[ch] = mask;
                cnt += __popcll(mask);
            }
        }
        if (lane == 0) c.red_i[wave] = cnt;
        __syncthreads();
        int mine = 0, total = 0;
        for (int w = 0; w < NW; ++w) { const int v = c.red_i[w]; if (w < wave) mine += v; total += v; }
        if (total > room) { __syncthreads(); return -1; }
        if (tid <= NW) {
            int o = 0;
            for (int w = 0; w < tid; ++w) o += c.red_i[w];
            off_out[tid] = off_base + o;
        }
        int pos = out_base + mine;
        for (int i0 = 0; i0 < mine_chunks && cnt > 0; i0 += 64) {
            const int i = i0 + lane;
            unsigned long long todo = __ballot(i < mine_chunks && chunk_may(wave + NW * i));
            while (todo) {
                const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
                todo &= todo - 1;
                const int ch = wave + NW * (i0 + src), k = (ch << 6) + lane;
                const unsigned long long mask = masks[ch];
                if ((mask >> lane) & 1ull) {
                    const int at = pos + __popcll(mask & ((1ull << lane) - 1ull));
                    out_idx[at] = k;
                    out_val[at] = vec[k];
                }
                pos += __popcll(mask);
            }
        }
        __syncthreads();
        return total;
    }
    if constexpr (Ctx::big) {
        // Layout 1 (the slot table and x in LDS, the spike in L2): the same 64-pivot chunks as layout 2, so that the spike is
        // read in whole cache lines (pivots k = wave + 8 i were every eighth word of 64 lines per load); four chunks requested
        // before the first is looked at; the first pass leaves every chunk's ballot in `chunk_mask`, the second touches the
        // chunks that hold anything only.
        const int n_chunks = (c.m + 63) >> 6;
        const int mine_chunks = (n_chunks - wave + NW - 1) / NW;       // chunks wave + NW * i, i < mine_chunks
        unsigned long long* masks = c.chunk_mask;
        int cnt = 0;
        for (int i0 = 0; i0 < mine_chunks; i0 += 4) {
            int ts[4];
            double vv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = min(((wave + NW * min(i0 + u, mine_chunks - 1)) << 6) + lane, c.m - 1);
                ts[u] = c.tslot[k]; vv[u] = vec[k];
            }
            asm volatile("" : "+v"(ts[0]), "+v"(ts[1]), "+v"(ts[2]), "+v"(ts[3]), "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3]));
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ch = wave + NW * (i0 + u), k = (ch << 6) + lane;
                const bool on = i0 + u < mine_chunks && k < c.m && k != skip && ts[u] < 0 && vv[u] != 0.0;
                const unsigned long long mask = __ballot(on);
                if (i0 + u < mine_chunks && lane == 0) masks[ch] = mask;
                cnt += __popcll(mask);
            }
        }
        if (lane == 0) c.red_i[wave] = cnt;
        __syncthreads();
        int mine = 0, total = 0;
        for (int w = 0; w < NW; ++w) { const int v = c.red_i[w]; if (w < wave) mine += v; total += v; }
        if (total > room) { __syncthreads(); return -1; }
        if (tid <= NW) {
            int o = 0;
            for (int w = 0; w < tid; ++w) o += c.red_i[w];
            off_out[tid] = off_base + o;
        }
        int pos = out_base + mine;
        for (int i0 = 0; i0 < mine_chunks && cnt > 0; i0 += 64) {      // (a wavefront reads 64 of its own masks at a time)
            const int i = i0 + lane;
            const unsigned long long mk = i < mine_chunks ? masks[wave + NW * i] : 0ull;
            unsigned long long any = __ballot(mk != 0ull);
            while (any) {
                const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)any) - 1);
                any &= any - 1;
                const unsigned lo = __builtin_amdgcn_readlane((unsigned)mk, src), hi = __builtin_amdgcn_readlane((unsigned)(mk >> 32), src);
                const unsigned long long mask = ((unsigned long long)hi << 32) | lo;
                const int k = ((wave + NW * (i0 + src)) << 6) + lane;
                if ((mask >> lane) & 1ull) {
                    const int at = pos + __popcll(mask & ((1ull << lane) - 1ull));
                    out_idx[at] = k;
                    out_val[at] = vec[k];
                }
                pos += __popcll(mask);
            }
        }
        __syncthreads();
        return total;
    }
    const int per_wave = (c.m - wave + NW - 1) / NW;           // pivots k = wave + NW * i, i < per_wave
    int cnt = 0;
    for (int i0 = 0; i0 < per_wave; i0 += 64) {
        const int i = i0 + lane, k = wave + NW * i;
        const bool on = i < per_wave && k != skip && c.tslot[k] < 0 && vec[k] != 0.0;
        cnt += __popcll(__ballot(on));
    }
    if (lane == 0) c.red_i[wave] = cnt;
    __syncthreads();
    int mine = 0, total = 0;
    for (int w = 0; w < NW; ++w) { const int v = c.red_i[w]; if (w < wave) mine += v; total += v; }
    if (total > room) { __syncthreads(); return -1; }
    if (tid <= NW) {
        int o = 0;
        for (int w = 0; w < tid; ++w) o += c.red_i[w];
        off_out[tid] = off_base + o;
    }
    int pos = out_base + mine;
    for (int i0 = 0; i0 < per_wave; i0 += 64) {
        const int i = i0 + lane, k = wave + NW * i;
        const bool on = i < per_wave && k != skip && c.tslot[k] < 0 && vec[k] != 0.0;
        const unsigned long long mask = __ballot(on);
        if (on) {
            const int at = pos + __popcll(mask & ((1ull << lane) - 1ull));
            out_idx[at] = k;
            out_val[at] = vec[k];
        }
        pos += __popcll(mask);
    }
    __syncthreads();
    return total;
}

// ---- the Forrest-Tomlin update (mod.rs:92-155) for the basis change in basis position `r`; the spike is in sp -------------
// Returns false, with the update file untouched, when r does not fit the eta pool any more (refactorisation due).
template <class Ctx>
__device__ __forceinline__ bool ft_update(const DeviceLU& lu, const FtState& st, Ctx& c, int r) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = c.t;
    const int p = c.icp[r];                            // the pivot whose column leaves (mod.rs:99-107)
    const FtPivotInfo pin = st.pinfo[p];               // (one load; its entries and lists follow in a second round trip)
    const int s_old = c.tslot[p];
    // u_bar = row p of U right of the diagonal (mod.rs:110-121), scattered into x
    xzero(c, st);
    int any = 0;
    if (s_old < 0) {
        for (int e = pin.u_e0 + tid; e < pin.u_e1; e += NT) {
            const int l = lu.Uf.idx[e];
            if (c.tslot[l] < 0) { xset(c, l, lu.Uf.val[e]); any = 1; }
        }
        // entries of row p in the spike columns of the live slots: only the bucket p % NW of each list can hold them
        const int bucket = Ctx::bucket_of(p);
        for (int s = wave; s < t; s += NW) {
            if (!c.slot_live[s]) continue;
            const int base = s * c.m;
            const int e0 = c.spk_off[s * (NW + 1) + bucket], e1 = c.spk_off[s * (NW + 1) + bucket + 1];
            for (int e = e0 + lane; e < e1; e += 64)
                if (st.spk_idx[base + e] == p) xset(c, c.slot_pivot[s], st.spk_val[base + e]);
        }
    } else {
        if (tid < t && tid > s_old && c.slot_live[tid]) xset(c, c.slot_pivot[tid], c.TC[s_old * c.ldt + tid]);
    }
    const bool do_sweep = __syncthreads_or(any) != 0;
    c.clk.lap(FT_UBAR);
    // r' = u_bar' U^-1 (mod.rs:122); u_bar lives on pivots after p, i.e. in levels above p's own
    ft_ut_solve(lu, st, c, do_sweep, pin.lev_ub + 1);
    // R = I - e_p r' (eta_file.rs:10-18): sparse part over the never-updated pivots into the eta pool, bucketed ...
    const int tn = t;                                  // the new slot
    int eta_n = 0;
    if (do_sweep) {
        eta_n = ft_compact(c, c.x, c.bx, p, c.eta_idx, c.eta_val, c.eta_used, c.eta_off + tn * (NW + 1), c.eta_used, c.eta_cap - c.eta_used);
        if (eta_n < 0) return false;
        if constexpr (Ctx::huge) if (c.clk.on) c.clk.nnz[1] += eta_n;
    } else {
        if (tid <= NW) c.eta_off[tn * (NW + 1) + tid] = c.eta_used;
        __syncthreads();
    }
    // ... part over the updated pivots into row tn of TC; the diagonal of the spike (update_spike_pivot_value,
    // eta_file.rs:111-133): sp[p] - r . sp
    double part = 0.0;
    for (int e = c.eta_used + tid; e < c.eta_used + eta_n; e += NT) part = fma(c.eta_val[e], c.sp[c.eta_idx[e]], part);
    if (tid < t) {
        const double y = c.slot_live[tid] ? c.zt[tid] : 0.0;
        c.TC[tn * c.ldt + tid] = y;
        if (y != 0.0) part = fma(y, c.sp[c.slot_pivot[tid]], part);
    }
    part = group_sum(part, 64);
    if (lane == 0) c.red_d[wave] = part;
    __syncthreads();
    double rdot = 0.0;
    for (int w = 0; w < NW; ++w) rdot += c.red_d[w];
    const double diag = c.sp[p] - rdot;
    // delete row p and column p from U (mod.rs:127-129): mask the pivot in U0, or kill its slot
    if (s_old < 0) {
        if (tid == 0) {
            st.ell[1].rdiag[p] = 0.0;
            st.ell[2].rdiag[p] = 0.0;
        }
        // fused levels: the entries of p's group-mates that were substituted through row p go with it
        for (int e = pin.via_u0 + tid; e < pin.via_u1; e += NT) st.ell[1].sval[st.ell[1].via_pos[e]] = 0.0;
        for (int e = pin.via_t0 + tid; e < pin.via_t1; e += NT) st.ell[2].sval[st.ell[2].via_pos[e]] = 0.0;
    } else {
        if (tid >= s_old && tid < c.tcap) c.TC[s_old * c.ldt + tid] = 0.0;    // (left of the diagonal: eta coefficients, kept)
        __syncthreads();
        if (tid <= s_old) c.TC[tid * c.ldt + s_old] = tid == s_old ? 1.0 : 0.0;
        if (tid == 0) { c.slot_live[s_old] = 0; c.slot_next[s_old] = tn; }
    }
    __syncthreads();
    // the spike becomes the last column (mod.rs:139-148): entries in the rows of the live slots into column tn of TC ...
    if (tid < t && c.slot_live[tid]) c.TC[tid * c.ldt + tn] = c.sp[c.slot_pivot[tid]];
    if (tid == 0) {
        c.TC[tn * c.ldt + tn] = diag;
        c.slot_pivot[tn] = p; c.slot_prev[tn] = s_old; c.slot_live[tn] = 1; c.slot_next[tn] = -1;
    }
    // ... entries in the never-updated rows into the spike pool, bucketed
    const int spike_n = ft_compact(c, c.sp, c.bsp, p, st.spk_idx + (int64_t)tn * c.m, st.spk_val + (int64_t)tn * c.m, 0, c.spk_off + tn * (NW + 1), 0, c.m);
    if constexpr (Ctx::huge) { if (c.clk.on) c.clk.nnz[2] += spike_n; } else (void)spike_n;
    if (tid == 0) c.tslot[p] = (typename Ctx::tslot_t)tn;
    c.eta_used += eta_n;
    c.t = tn + 1;
    __syncthreads();
    c.clk.lap(FT_COMPACT);
    return true;
}

// x := P a for tableau column q (partially.rs:72-80, matrix_data.rs:308-348), pivot-indexed.  Ends with a barrier.
template <class Ctx>
__device__ __forceinline__ void ft_scatter_column(const FtState& st, const FtProblem& pb, Ctx& c, int q) {
    const int tid = threadIdx.x;
    const ColumnTable& ct = pb.ct;
    xzero(c, st);
    if (q < ct.nr_artificial) {
        if (tid == 0) xset(c, c.irp[ct.column_to_row[q]], 1.0);
    } else {
        const int p = q - ct.nr_artificial;
        if (p < ct.nr_normal) {
            const int64_t s0 = pb.csc.col_ptr[p], s1 = pb.csc.col_ptr[p + 1];
            for (int64_t e = s0 + tid; e < s1; e += NT) xset(c, c.irp[pb.csc.row_idx[e]], pb.csc.values[e]);
            const int br = ct.bound_row[p];
            if (tid == 0 && br >= 0) xset(c, c.irp[br], 1.0);
        } else if (tid == 0) {
            const int v = p - ct.nr_normal;
            const int r0 = ct.vrow0[v], r1 = ct.vrow1[v];
            if (r0 >= 0) xset(c, c.irp[r0], (double)ct.vsign[v]);
            if (r1 >= 0) xset(c, c.irp[r1], 1.0);
        }
    }
    __syncthreads();
}

// workgroup minimum of (key, j), lexicographic; result in every thread
// Minimum over the wavefront, valid in lane 0: lane swaps and DPP row shifts, register to register (the same tree as
// group_sum; __shfl_down goes through the LDS crossbar, ~100 clocks a step, and a pivot has six block reductions).
__device__ __forceinline__ double wave_min_f64(double v) {
    v = fmin(v, lane_plus_32(v));
    v = fmin(v, lane_plus_16(v));
    v = fmin(v, dpp_row_shl<0x108>(v));
    v = fmin(v, dpp_row_shl<0x104>(v));
    v = fmin(v, dpp_row_shl<0x102>(v));
    v = fmin(v, dpp_row_shl<0x101>(v));
    return v;
}
template <int kCtrl>
__device__ __forceinline__ int dpp_row_shl_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, kCtrl, 0xf, 0xf, true); }
__device__ __forceinline__ int wave_min_i32(int v) {
    { const auto a = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false); v = min(v, (int)a[1]); }
    { const auto a = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false); v = min(v, (int)a[1]); }
    v = min(v, dpp_row_shl_i32<0x108>(v));
    v = min(v, dpp_row_shl_i32<0x104>(v));
    v = min(v, dpp_row_shl_i32<0x102>(v));
    v = min(v, dpp_row_shl_i32<0x101>(v));
    return v;
}

template <class Ctx>
__device__ __forceinline__ int block_min_int(Ctx& c, int v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_min_i32(v);
    __syncthreads();
    if (lane == 0) c.red_i[wave] = v;
    __syncthreads();
    v = c.red_i[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) v = min(v, c.red_i[w]);
    return v;
}

template <class Ctx>
__device__ __forceinline__ double block_min_double(Ctx& c, double v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_min_f64(v);
    __syncthreads();
    if (lane == 0) c.red_d[wave] = v;
    __syncthreads();
    v = c.red_d[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) v = fmin(v, c.red_d[w]);
    return v;
}

// minimum of 64-bit tie keys (relp_device_common.h: tie_key) over the workgroup; only the RELP_RATIO_LARGEST_PIVOT rule
// comes here, the reference's rule reduces 32-bit leaving columns (block_min_int)
template <class Ctx>
__device__ __forceinline__ tie_key_t block_min_key64(Ctx& c, tie_key_t v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const tie_key_t o = __shfl_down(v, off, 64); v = o < v ? o : v; }
    tie_key_t* red = reinterpret_cast<tie_key_t*>(c.red_d);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    v = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) v = red[w] < v ? red[w] : v;
    return v;
}

// lexicographic minimum of (key, j) over the workgroup: the smallest key, then the lowest j among the threads that hold it
template <class Ctx>
__device__ __forceinline__ void block_min_key(Ctx& c, double& key, int& kj) {
    const double kmin = block_min_double(c, key);
    kj = block_min_int(c, key == kmin ? kj : 0x7fffffff);
    key = kmin;
}

// ------------------------------------------------------------------------------------------------------------------
// The persistent pivot kernel
// ------------------------------------------------------------------------------------------------------------------
template <int kTier>
__global__ __launch_bounds__(NT) void k_ft_run(DeviceLU lu, FtState st, FtProblem pb, long long max_pivots) {
    extern __shared__ __align__(16) char lds[];
    PivotRecord* rec = pb.rec;
    if (rec->outcome != DEV_RUNNING) {
        // nothing to do; the report is still this launch's own (no basis change in the journal), never the previous one's
        if (pb.mirror) {
            if (threadIdx.x == 0) {
                pb.mirror->rec = *rec;
                pb.mirror->hdr[0] = st.hdr[0]; pb.mirror->hdr[1] = st.hdr[1]; pb.mirror->hdr[2] = st.hdr[2]; pb.mirror->hdr[3] = 0;
                for (int k = 0; k < 4; ++k) { pb.mirror->walked[k] = 0; pb.mirror->whole[k] = 0; pb.mirror->sweeps[k] = 0; }
            }
            for (int i = threadIdx.x; i < st.m; i += NT) pb.mirror->basis[i] = pb.basis[i];
        }
        if (threadIdx.x == 0) st.hdr[3] = 0;
        return;
    }
    FtCtxT<kTier> c;
    ft_bind(c, lds, lu, st);
    c.clk.start(st.prof);
    ft_load(c, lu, st, pb.minus_pi);
    hs_enter(c, st);
    int alpha_nnz = 0, rho_nnz = 0;                    // (layout 2) entries of nz_idx / rho_idx that describe pb.alpha / pb.rho
    if constexpr (kTier >= 2) { alpha_nnz = st.nzc[0]; rho_nnz = st.nzc[1]; }
    c.clk.lap(FT_LOAD_STORE);
    const int tid = threadIdx.x;
    const ColumnTable& ct = pb.ct;
    const int m = c.m, n = pb.n, rule = pb.rule, cost_mode = pb.phase;
    int last_selected = rec->last_selected;
    double minus_objective = rec->minus_objective;
    long long iterations = rec->iterations;
    int degenerate = rec->degenerate;
    int outcome = DEV_RUNNING, need_refactor = st.hdr[2] >= 2 ? 2 : 0;       // (2: a replay failed, the factors are unusable)
    // (a launch of the one-pivot-per-launch loop must not pivot on factors an earlier launch of its batch declared due: the host
    // looks at the header after the batch only.  The persistent loop is relaunched on purpose with hdr[2] == 1: look-ahead.)
    if (pb.external_price && st.hdr[2]) need_refactor = st.hdr[2];
    if (pb.external_price) c.journal_n = st.hdr[3];                     // (one journal per batch)
    int q = rec->q, r = rec->r, leaving = rec->leaving;
    double d_q = rec->d_q, alpha_r = rec->alpha_r, b_r = rec->b_r, key1 = rec->key1;
    PivotRecord fake;                                  // select_key reads rule memory through a record
    fake.last_selected = last_selected;
    // (row indices of the PRICE copy: 16 bits, bit 15 = "long column"; 32 bits with bit 31 in layout 2)
    constexpr int kLongFlag = kTier >= 2 ? (int)kPriceLongFlag32 : kPriceLongFlag, kLongMask = kTier >= 2 ? 0x7fffffff : kPriceLongFlag - 1;
    auto pe_idx = [&](int i) -> int { if constexpr (kTier >= 2) return (int)pb.pe.idx32[i]; else return pb.pe.idx[i]; };
    auto pe_lidx = [&](int i) -> int { if constexpr (kTier >= 2) return (int)pb.pe.lidx32[i]; else return pb.pe.lidx[i]; };

    for (long long it = 0; it < max_pivots; ++it) {
        if (need_refactor || c.t >= st.max_updates || c.t >= c.tcap) { if (!need_refactor) need_refactor = 1; break; }
        // ---- PRICE (pivot_rule.rs:38-126 over tableau/mod.rs:102-108): d_j = c_j + (-pi) . a_j, thread per column ----------
        // (external_price: this launch makes one pivot with the column a grid-wide PRICE chose -- Engine::run_ft, Dantzig's rule
        // over very many columns, where one workgroup pricing all of them was 70 % of the pivot)
        if (!pb.external_price) {
        fake.last_selected = last_selected;
        double key = INFINITY, kv = 0.0;
        int kj = 0x7fffffff;
        // (structural columns from the k-major PRICE copy: every load of a column is issued before the first is used, incl.
        // its basis flag, bound row and cost; summation in the column's own order like k_price_csc)
        const int na = ct.nr_artificial, nstr = ct.nr_normal;
        auto consider = [&](int j, double v, int basic) {
            pb.d[j] = v;
            if (!basic && v < -pb.tol.cost) {
                const double k = select_key(rule, n, &fake, j, v);
                if (k < key || (k == key && j < kj)) { key = k; kj = j; kv = v; }
            }
        };
        const double* costs = cost_mode == 2 ? ct.cost : nullptr;
        int my_j = 0x7fffffff;                                      // this thread's own candidate and its reduced cost
        double my_v = 0.0;
        if (rule != 2) {
            // FirstProfitable / FirstProfitableWithMemory (pivot_rule.rs:38-95) take the first profitable column of a fixed
            // search order: the columns are priced NT at a time along that order and the scan stops at the first chunk that
            // holds one -- the same column as a full scan would choose, for a fraction of the loads.  (d is not refreshed
            // here; nothing reads it under these rules.)
            const int start = (rule == 1 && last_selected >= 0) ? last_selected : 0;
            for (int base = 0; base < n; base += NT) {
                const int k = base + tid;
                int my_k = 0x7fffffff;
                if (k < n) {
                    int j = start + k;
                    if (j >= n) j -= n;
                    const int basic = pb.in_basis[j];
                    double v;
                    if (j < na) {
                        v = (cost_mode == 1 ? 1.0 : 0.0) + c.pi[ct.column_to_row[j]];
                    } else if (j < na + nstr) {
                        const int p = j - na;
                        int ri[kPriceSlots];
                        double va[kPriceSlots], px[kPriceSlots];
#pragma unroll
                        for (int u = 0; u < kPriceSlots; ++u) { ri[u] = pe_idx(u * nstr + p); va[u] = pb.pe.val[u * nstr + p]; }
                        const int br = ct.bound_row[p];
                        const int li = pb.pe.long_of[p];
                        const double cp = costs ? costs[p] : 0.0;
                        const bool is_long = (ri[0] & kLongFlag) != 0;
                        ri[0] &= kLongMask;
#pragma unroll
                        for (int u = 0; u < kPriceSlots; ++u) px[u] = c.pi[ri[u]];
                        const double pb_r = c.pi[br >= 0 ? br : 0];
                        v = 0.0;
#pragma unroll
                        for (int u = 0; u < kPriceSlots; ++u) v = fma(px[u], va[u], v);
                        if (is_long && li != 0xFFFF) {              // its entries 8.. from the second tier, same order
                            const int nl = pb.pe.n_long;
                            constexpr int kRest = kPriceLongSlots - kPriceSlots;
                            int rj[kRest];
                            double vb[kRest], py[kRest];
#pragma unroll
                            for (int u = 0; u < kRest; ++u) { rj[u] = pe_lidx((kPriceSlots + u) * nl + li); vb[u] = pb.pe.lval[(kPriceSlots + u) * nl + li]; }
#pragma unroll
                            for (int u = 0; u < kRest; ++u) py[u] = c.pi[rj[u]];
#pragma unroll
                            for (int u = 0; u < kRest; ++u) v = fma(py[u], vb[u], v);
                        } else if (is_long) {                       // beyond the second tier: the whole column from the CSC arrays
                            v = 0.0;
                            for (int64_t e = pb.csc.col_ptr[p]; e < pb.csc.col_ptr[p + 1]; ++e) v = fma(c.pi[pb.csc.row_idx[e]], pb.csc.values[e], v);
                        }
                        if (br >= 0) v += pb_r;
                        if (costs) v += cp;
                    } else {
                        const int vv = j - na - nstr;
                        const int r0 = ct.vrow0[vv], r1 = ct.vrow1[vv];
                        v = r0 >= 0 ? (double)ct.vsign[vv] * c.pi[r0] : 0.0;
                        if (r1 >= 0) v += c.pi[r1];
                    }
                    if (!basic && v < -pb.tol.cost) { my_k = k; my_v = v; }
                }
                const int kmin = block_min_int(c, my_k);
                if (kmin != 0x7fffffff) {
                    key = (double)kmin;                             // (= select_key: the position in the search order)
                    kj = start + kmin;
                    if (kj >= n) kj -= n;
                    if (my_k == kmin) my_j = kj;
                    break;
                }
            }
        } else {
        for (int j = tid; j < na; j += NT) consider(j, (cost_mode == 1 ? 1.0 : 0.0) + c.pi[ct.column_to_row[j]], pb.in_basis[j]);
        for (int p = tid; p < nstr; p += NT) {
            int ri[kPriceSlots];
            double va[kPriceSlots], px[kPriceSlots];
#pragma unroll
            for (int u = 0; u < kPriceSlots; ++u) { ri[u] = pe_idx(u * nstr + p); va[u] = pb.pe.val[u * nstr + p]; }
            const int br = ct.bound_row[p];
            const int basic = pb.in_basis[na + p];
            const double cp = costs ? costs[p] : 0.0;
            const bool is_long = (ri[0] & kLongFlag) != 0;    // more than kPriceSlots entries: priced below
            ri[0] &= kLongMask;
#pragma unroll
            for (int u = 0; u < kPriceSlots; ++u) px[u] = c.pi[ri[u]];
            const double pb_r = c.pi[br >= 0 ? br : 0];
            double v = 0.0;
#pragma unroll
            for (int u = 0; u < kPriceSlots; ++u) v = fma(px[u], va[u], v);
            if (br >= 0) v += pb_r;
            if (costs) v += cp;
            if (!is_long) consider(na + p, v, basic);
        }
        for (int i = tid; i < pb.pe.n_long; i += NT) {             // columns of kPriceSlots + 1 .. kPriceLongSlots entries
            const int nl = pb.pe.n_long;
            const int p = pb.pe.long_cols[i];
            int ri[kPriceLongSlots];
            double va[kPriceLongSlots], px[kPriceLongSlots];
#pragma unroll
            for (int u = 0; u < kPriceLongSlots; ++u) { ri[u] = pe_lidx(u * nl + i); va[u] = pb.pe.lval[u * nl + i]; }
            const int br = ct.bound_row[p];
            const int basic = pb.in_basis[na + p];
            const double cp = costs ? costs[p] : 0.0;
#pragma unroll
            for (int u = 0; u < kPriceLongSlots; ++u) px[u] = c.pi[ri[u]];
            const double pb_r = c.pi[br >= 0 ? br : 0];
            double v = 0.0;
#pragma unroll
            for (int u = 0; u < kPriceLongSlots; ++u) v = fma(px[u], va[u], v);
            if (br >= 0) v += pb_r;
            if (costs) v += cp;
            consider(na + p, v, basic);
        }
        for (int i = tid; i < pb.pe.n_very_long; i += NT) {        // the few columns beyond that
            const int p = pb.pe.very_long[i];
            double v = 0.0;
            for (int64_t e = pb.csc.col_ptr[p]; e < pb.csc.col_ptr[p + 1]; ++e) v = fma(c.pi[pb.csc.row_idx[e]], pb.csc.values[e], v);
            const int br = ct.bound_row[p];
            if (br >= 0) v += c.pi[br];
            if (cost_mode == 2) v += ct.cost[p];
            consider(na + p, v, pb.in_basis[na + p]);
        }
        for (int vv = tid; vv < ct.nr_virtual; vv += NT) {
            const int r0 = ct.vrow0[vv], r1 = ct.vrow1[vv];
            double v = r0 >= 0 ? (double)ct.vsign[vv] * c.pi[r0] : 0.0;
            if (r1 >= 0) v += c.pi[r1];
            consider(na + nstr + vv, v, pb.in_basis[na + nstr + vv]);
        }
#ifdef PRICE_DIAG
        c.clk.lap(FT_PRICE);
#endif
        my_j = kj;
        my_v = kv;
        block_min_key(c, key, kj);
#ifdef PRICE_DIAG
        c.clk.lap(FT_LOAD_STORE);
#endif
        if (kj != 0x7fffffff && rule == 2 && pb.tol.tie > 0.0) {
            // Dantzig ties: lowest index within the tie band of the minimum (every thread re-reads a stripe of d, eight
            // columns per round so that the loads of a round are in flight together)
            const double bound = key + pb.tol.tie * fmax(1.0, fabs(key));
            my_j = 0x7fffffff;
            for (int j0 = tid; j0 < n; j0 += 8 * NT) {
                double dv[8];
                int ib[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int j = min(j0 + u * NT, n - 1); dv[u] = pb.d[j]; ib[u] = pb.in_basis[j]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = j0 + u * NT;
                    if (j < n && !ib[u] && dv[u] < -pb.tol.cost && dv[u] <= bound && j < my_j) { my_j = j; my_v = dv[u]; }
                }
            }
            kj = block_min_int(c, my_j);
        }
        }
        if (kj == 0x7fffffff) {
            outcome = DEV_NO_CANDIDATE;
            if (rule == 1) last_selected = -1;
            break;
        }
#ifdef PRICE_DIAG
        c.clk.lap(FT_B);
#endif
        if (my_j == kj) c.red_d[NW] = my_v;                         // the owner publishes d_q
        __syncthreads();
        q = kj; key1 = key; d_q = c.red_d[NW];
        if (rule == 1) last_selected = q;
        }
#ifdef PRICE_DIAG
        c.clk.lap(FT_SCATTER);
#else
        c.clk.lap(FT_PRICE);
#endif

        // ---- FTRAN (mod.rs:157-190) ------------------------------------------------------------------------------------
        ft_scatter_column(st, pb, c, q);
        c.clk.lap(FT_SCATTER);
        ft_ftran(lu, st, c);

        // ---- RATIO TEST (tableau/mod.rs:221-247; two passes as relp_device_common.h ratio_body) ------------------------
        double br;
        if constexpr (kTier >= 1) {
        // Layouts 1 and 2: the solve leaves x pivot-indexed and mostly zero.  One coalesced pass over x scatters alpha (dense, for
        // whoever reads it after the launch) and appends the non-zeros as (row, alpha) pairs to a list; the ratio test, the tie
        // band, the leaving row and the update of b then touch the list only (tableau/mod.rs:221-247 walks the stored
        // non-zeros of the column just the same).  Every reduction is order-free (minima), so the list's order does not matter.
        // pb.alpha (dense, for whoever reads it after the launch): zeroed where the previous column was not -- its list is
        // still there -- then the non-zeros of x, found through the bitmap, are scattered and listed
        if constexpr (kTier >= 2) {
            if (alpha_nnz < 0) { for (int i = tid; i < m; i += NT) pb.alpha[i] = 0.0; }
            else for (int e = tid; e < alpha_nnz; e += NT) pb.alpha[st.nz_idx[e]] = 0.0;
            if (tid == 0) c.red_i[2 * NW] = 0;
            __syncthreads();
            hs_for_each(c.bx, c.gs, 0, m, [&](int k) {
                const double v = c.x[k];
                if (v != 0.0) {
                    const int i = lu.colperm[k];
                    pb.alpha[i] = v;
                    const int e = atomicAdd(&c.red_i[2 * NW], 1);
                    st.nz_idx[e] = i; st.nz_val[e] = v;
                }
            });
        } else {
            // (layout 1: x in LDS, no bitmap -- one pass over x with the permutation coalesced from L2 beside it: alpha written
            // everywhere, the non-zeros listed.  The three passes this replaces each went x[icp[i]], b[i], basis[i] through L2
            // for every row.)
            if (tid == 0) c.red_i[2 * NW] = 0;
            __syncthreads();
            for (int k0 = tid; k0 < m; k0 += 4 * NT) {
                int cp[4];
                double xv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int k = min(k0 + u * NT, m - 1); xv[u] = c.x[k]; cp[u] = lu.colperm[k]; }
                asm volatile("" : "+v"(cp[0]), "+v"(cp[1]), "+v"(cp[2]), "+v"(cp[3]));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (k0 + u * NT < m) {
                        pb.alpha[cp[u]] = xv[u];
                        if (xv[u] != 0.0) { const int e = atomicAdd(&c.red_i[2 * NW], 1); st.nz_idx[e] = cp[u]; st.nz_val[e] = xv[u]; }
                    }
                }
            }
        }
        __syncthreads();
        const int nnz = c.red_i[2 * NW];
        alpha_nnz = nnz;
        if constexpr (kTier >= 2) if (c.clk.on) { c.clk.nnz[0] += nnz; c.clk.nnz[3] += 1; }
        // (the first entry of every thread stays in registers: a column rarely has more than NT non-zeros)
        const bool h0 = tid < nnz;
        const int i0 = h0 ? st.nz_idx[tid] : 0;
        const double a0 = h0 ? st.nz_val[tid] : 0.0;
        const double b0 = h0 ? pb.b[i0] : 0.0;
        double mn = h0 ? row_ratio(a0, b0, pb.tol) : INFINITY;
        for (int e = tid + NT; e < nnz; e += NT) mn = fmin(mn, row_ratio(st.nz_val[e], pb.b[st.nz_idx[e]], pb.tol));
        const double gmin = block_min_double(c, mn);
        if (gmin == INFINITY) { outcome = DEV_NO_ROW; break; }
        const double bound = gmin + pb.tol.tie * fmax(1.0, fabs(gmin));
        int best_leave = 0x7fffffff, rr = 0x7fffffff;
        tie_key_t best = kNoTieKey;
        const bool by_column = pb.tol.ratio_rule == 0;
        auto candidate = [&](int i, double a, double bi) {
            if (row_ratio(a, bi, pb.tol) <= bound) {
                const int sc = pb.basis[i];
                if (by_column) { if (sc < best_leave || (sc == best_leave && i < rr)) { best_leave = sc; rr = i; } }
                else { const tie_key_t k = tie_key(a, sc, 1); if (k < best || (k == best && i < rr)) { best = k; best_leave = sc; rr = i; } }
            }
        };
        if (h0) candidate(i0, a0, b0);
        for (int e = tid + NT; e < nnz; e += NT) { const int i = st.nz_idx[e]; candidate(i, st.nz_val[e], pb.b[i]); }
        if (by_column) leaving = block_min_int(c, best_leave);
        else {
            const tie_key_t kmin = block_min_key64(c, best);
            leaving = tie_key_leaving(kmin);
            if (best != kmin) best_leave = 0x7fffffff;
        }
        r = block_min_int(c, best_leave == leaving ? rr : 0x7fffffff);       // (a column is basic in one row)
        if (best_leave == leaving && rr == r) { c.red_d[NW] = pb.alpha[r]; c.red_d[NW + 1] = pb.b[r]; }
        __syncthreads();
        alpha_r = c.red_d[NW];
        b_r = c.red_d[NW + 1];
        if (pb.tol.pivot_guard) {                      // (Tolerances::pivot_guard: too small beside the column's largest entry)
            double mx = h0 ? fabs(a0) : 0.0;
            for (int e = tid + NT; e < nnz; e += NT) mx = fmax(mx, fabs(st.nz_val[e]));
            const double amax = -block_min_double(c, -mx);
            if (alpha_r < pb.tol.guard_rel * amax) { outcome = DEV_NO_ROW; break; }
        }
        c.clk.lap(FT_RATIO);
        br = b_r / alpha_r;
        if (h0) pb.b[i0] = i0 == r ? br : fma(-a0, br, b0);
        for (int e = tid + NT; e < nnz; e += NT) {
            const int i = st.nz_idx[e];
            pb.b[i] = i == r ? br : fma(-st.nz_val[e], br, pb.b[i]);
        }
        __syncthreads();
        } else {
        // Rows tid and tid + NT of this thread stay in registers through the three passes and the update of b (one round of
        // global loads for m <= 2 NT); rows beyond that take the generic loops.
        const int i0 = tid, i1 = tid + NT;
        const bool h0 = i0 < m, h1 = i1 < m;
        const int p0 = c.icp[h0 ? i0 : 0], p1 = c.icp[h1 ? i1 : 0];
        double b0 = pb.b[h0 ? i0 : 0], b1 = pb.b[h1 ? i1 : 0];
        const int s0 = pb.basis[h0 ? i0 : 0], s1 = pb.basis[h1 ? i1 : 0];
        const double a0 = c.x[p0], a1 = c.x[p1];
        if (h0) pb.alpha[i0] = a0;
        if (h1) pb.alpha[i1] = a1;
        const double t0 = h0 ? row_ratio(a0, b0, pb.tol) : INFINITY, t1 = h1 ? row_ratio(a1, b1, pb.tol) : INFINITY;
        double mn = fmin(t0, t1);
        for (int i = tid + 2 * NT; i < m; i += NT) {
            const double a = c.x[c.icp[i]];
            pb.alpha[i] = a;
            mn = fmin(mn, row_ratio(a, pb.b[i], pb.tol));
        }
        const double gmin = block_min_double(c, mn);
        if (gmin == INFINITY) { outcome = DEV_NO_ROW; break; }
        const double bound = gmin + pb.tol.tie * fmax(1.0, fabs(gmin));
        if (pb.tol.ratio_rule == 0) {                  // the reference: lowest leaving column inside the band
            int best_leave = 0x7fffffff;
            if (t0 <= bound) best_leave = s0;
            if (t1 <= bound) best_leave = min(best_leave, s1);
            for (int i = tid + 2 * NT; i < m; i += NT) {
                const double a = c.x[c.icp[i]];
                if (row_ratio(a, pb.b[i], pb.tol) <= bound) best_leave = min(best_leave, pb.basis[i]);
            }
            leaving = block_min_int(c, best_leave);
        } else {                                       // RELP_RATIO_LARGEST_PIVOT: pivot size first, then the column
            tie_key_t best = kNoTieKey;
            if (t0 <= bound) best = tie_key(a0, s0, 1);
            if (t1 <= bound) { const tie_key_t k1 = tie_key(a1, s1, 1); best = k1 < best ? k1 : best; }
            for (int i = tid + 2 * NT; i < m; i += NT) {
                const double a = c.x[c.icp[i]];
                if (row_ratio(a, pb.b[i], pb.tol) <= bound) { const tie_key_t k = tie_key(a, pb.basis[i], 1); best = k < best ? k : best; }
            }
            leaving = tie_key_leaving(block_min_key64(c, best));
        }
        // the row of the leaving column
        int rr = 0x7fffffff;
        if (h0 && s0 == leaving && t0 <= bound) rr = i0;
        if (h1 && s1 == leaving && t1 <= bound) rr = i1;             // (a column is basic in one row)
        for (int i = tid + 2 * NT; i < m; i += NT)
            if (pb.basis[i] == leaving && row_ratio(c.x[c.icp[i]], pb.b[i], pb.tol) <= bound) rr = i;
        r = block_min_int(c, rr);
        // alpha_r and b_r from their owner
        if (rr == r) { c.red_d[NW] = c.x[c.icp[r]]; c.red_d[NW + 1] = pb.b[r]; }
        __syncthreads();
        alpha_r = c.red_d[NW];
        b_r = c.red_d[NW + 1];
        if (pb.tol.pivot_guard) {                      // (Tolerances::pivot_guard: too small beside the column's largest entry)
            double mx = fmax(h0 ? fabs(a0) : 0.0, h1 ? fabs(a1) : 0.0);
            for (int i = tid + 2 * NT; i < m; i += NT) mx = fmax(mx, fabs(c.x[c.icp[i]]));
            const double amax = -block_min_double(c, -mx);
            if (alpha_r < pb.tol.guard_rel * amax) { outcome = DEV_NO_ROW; break; }
        }
        c.clk.lap(FT_RATIO);

        // ---- b (carry/mod.rs:283-313) while alpha is still in x -----------------------------------------------------
        br = b_r / alpha_r;
        if (h0) { if (i0 == r) pb.b[i0] = br; else if (a0 != 0.0) pb.b[i0] = fma(-a0, br, b0); }
        if (h1) { if (i1 == r) pb.b[i1] = br; else if (a1 != 0.0) pb.b[i1] = fma(-a1, br, b1); }
        for (int i = tid + 2 * NT; i < m; i += NT) {
            if (i == r) pb.b[i] = br;
            else {
                const double a = c.x[c.icp[i]];
                if (a != 0.0) pb.b[i] = fma(-a, br, pb.b[i]);
            }
        }
        __syncthreads();
        }

        c.clk.lap(FT_B);
        // ---- basis inverse: the Forrest-Tomlin update, then row r of the new inverse (mod.rs:92-155, 204-222) --------------
        const bool updated = ft_update(lu, st, c, r);
        if (tid == 0 && c.journal_n < c.tcap) { st.journal[2 * c.journal_n] = r; st.journal[2 * c.journal_n + 1] = q; }
        c.journal_n += 1;
        const int pl = c.icp[r];
        xzero(c, st);
        if (tid == 0) xset(c, pl, 1.0);
        __syncthreads();
        double rho_scale = 1.0;
        if (updated) {
            ft_btran(lu, st, c, false);                // the leaving pivot is last in U now: no sweep over U0'
        } else {
            // the eta pool is full: row r of the new inverse = row r of the old one / alpha_r (basis_inverse_rows.rs:42-51
            // states the same division); the factors are rebuilt before the next pivot
            ft_btran(lu, st, c, c.tslot[pl] < 0, st.lev_ub[pl]);
            rho_scale = 1.0 / alpha_r;
            need_refactor = 1;
        }
        // ---- -pi, -obj, basis (carry/mod.rs:326-333, 549-570) ----------------------------------------------------------
        if constexpr (kTier >= 2) {                     // (pb.rho as pb.alpha above; -pi changes where rho is not zero only)
            if (rho_nnz < 0) { for (int i = tid; i < m; i += NT) pb.rho[i] = 0.0; }
            else for (int e = tid; e < rho_nnz; e += NT) pb.rho[st.rho_idx[e]] = 0.0;
            if (tid == 0) c.red_i[2 * NW + 1] = 0;
            __syncthreads();
            hs_for_each(c.bx, c.gs, 0, m, [&](int k) {
                const double rho = c.x[k] * rho_scale;
                if (rho != 0.0) {
                    const int i = c.rp[k];
                    pb.rho[i] = rho;
                    c.pi[i] = fma(-d_q, rho, c.pi[i]);
                    st.rho_idx[atomicAdd(&c.red_i[2 * NW + 1], 1)] = i;
                }
            });
            __syncthreads();
            rho_nnz = c.red_i[2 * NW + 1];
        } else if constexpr (kTier == 1) {             // (the permutation comes from L2: four rows requested at a time)
            for (int k0 = tid; k0 < m; k0 += 4 * NT) {
                int ii[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) ii[u] = c.rp[min(k0 + u * NT, m - 1)];
                asm volatile("" : "+v"(ii[0]), "+v"(ii[1]), "+v"(ii[2]), "+v"(ii[3]));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (k0 + u * NT < m) {
                        const double rho = c.x[k0 + u * NT] * rho_scale;
                        pb.rho[ii[u]] = rho;
                        c.pi[ii[u]] = fma(-d_q, rho, c.pi[ii[u]]);
                    }
                }
            }
        } else
        for (int k = tid; k < m; k += NT) {
            const int i = c.rp[k];
            const double rho = c.x[k] * rho_scale;
            pb.rho[i] = rho;
            c.pi[i] = fma(-d_q, rho, c.pi[i]);
        }
        if (tid == 0) {
            pb.basis[r] = q;
            if (leaving < kWrappedArtificialBase) pb.in_basis[leaving] = 0;
            pb.in_basis[q] = 1;
            if (pb.trace && iterations < pb.trace_cap) {
                pb.trace[0 * pb.trace_cap + iterations] = pb.phase;
                pb.trace[1 * pb.trace_cap + iterations] = q;
                pb.trace[2 * pb.trace_cap + iterations] = r;
                pb.trace[3 * pb.trace_cap + iterations] = leaving;
            }
        }
        minus_objective = fma(-d_q, br, minus_objective);
        if (br == 0.0) degenerate += 1;
        iterations += 1;
        __syncthreads();
        c.clk.lap(FT_VECTORS);
    }
    ft_store(c, st, pb.minus_pi, need_refactor);
    hs_leave(c, st, true);
    if constexpr (kTier >= 2) if (tid == 0) { st.nzc[0] = alpha_nnz; st.nzc[1] = rho_nnz; }
    c.clk.lap(FT_LOAD_STORE);
    c.clk.flush(st.prof, kTier >= 2);
#ifdef PASS_DIAG
    if (tid == 0 && outcome != DEV_RUNNING)
        printf("pass_diag: own-pass clocks %llu / %llu over %llu / %llu passes; barrier clocks %llu / %llu over %llu / %llu; fetch clocks %llu / %llu; other set's passes %llu / %llu\n",
               pass_diag[0], pass_diag[1], pass_diag[2], pass_diag[3], pass_diag[4], pass_diag[5], pass_diag[6], pass_diag[7], pass_diag[8], pass_diag[9], pass_diag[10], pass_diag[11]);
#endif
    if (tid == 0) {
        rec->outcome = outcome; rec->q = q; rec->d_q = d_q; rec->r = r; rec->leaving = leaving; rec->alpha_r = alpha_r;
        rec->b_r = b_r; rec->minus_objective = minus_objective; rec->iterations = iterations;
        rec->last_selected = last_selected; rec->key1 = key1; rec->degenerate = degenerate;
    }
    if (pb.mirror) {                                   // (the host reads these after synchronising the stream)
        if (tid == 0) {
            pb.mirror->rec = *rec;
            pb.mirror->hdr[0] = c.t; pb.mirror->hdr[1] = c.eta_used; pb.mirror->hdr[2] = need_refactor; pb.mirror->hdr[3] = c.journal_n;
            if constexpr (kTier >= 1) {
                for (int k = 0; k < 4; ++k) {
                    pb.mirror->walked[k] = (int32_t)c.clk.passes[k]; pb.mirror->whole[k] = (int32_t)c.clk.total[k];
                    pb.mirror->sweeps[k] = (int32_t)c.clk.sweeps[k];
                }
            }
        }
        for (int i = tid; i < m; i += NT) pb.mirror->basis[i] = pb.basis[i];
    }
}

// ---- single steps (step-wise API, phase boundaries) ------------------------------------------------------------------
template <int kTier>
__global__ __launch_bounds__(NT) void k_ft_ftran(DeviceLU lu, FtState st, FtProblem pb, int column, const double* rhs,
                                                  double* alpha) {
    extern __shared__ __align__(16) char lds[];
    FtCtxT<kTier> c;
    ft_bind(c, lds, lu, st);
    ft_load(c, lu, st, nullptr);
    hs_enter(c, st);
    const int tid = threadIdx.x;
    if (rhs) xfill(c, st, [&](int k) { return rhs[lu.rowperm[k]]; });
    else ft_scatter_column(st, pb, c, column == -2 ? pb.rec->q : column);
    ft_ftran(lu, st, c);
    for (int k = tid; k < c.m; k += NT) { alpha[lu.colperm[k]] = c.x[k]; st.spike[k] = c.sp[k]; }
    hs_leave(c, st, true);
    if constexpr (kTier >= 2) if (tid == 0) st.nzc[0] = -1;             // (`alpha` may be the engine's own vector: its list is void)
}

template <int kTier>
__global__ __launch_bounds__(NT) void k_ft_btran(DeviceLU lu, FtState st, FtProblem pb, int row, const double* rhs, double* rho) {
    extern __shared__ __align__(16) char lds[];
    FtCtxT<kTier> c;
    ft_bind(c, lds, lu, st);
    ft_load(c, lu, st, nullptr);
    const int tid = threadIdx.x;
    bool sweep_u = true;
    int first = 0;
    hs_enter(c, st);
    if (rhs) {
        xfill(c, st, [&](int k) { return rhs[lu.colperm[k]]; });
    } else {
        const int r = row == -2 ? pb.rec->r : row;
        const int p = st.inv_colperm[r];
        xzero(c, st);
        if (tid == 0) xset(c, p, 1.0);
        sweep_u = c.tslot[p] < 0;                      // an updated pivot has no entry in U0 any more
        first = st.lev_ub[p];                          // e_p: nothing below the level of p's own row
    }
    __syncthreads();
    ft_btran(lu, st, c, sweep_u, first);
    for (int k = tid; k < c.m; k += NT) rho[lu.rowperm[k]] = c.x[k];
    hs_leave(c, st, true);
    if constexpr (kTier >= 2) if (tid == 0) st.nzc[1] = -1;             // (`rho` may be the engine's own vector)
}

template <int kTier>
__global__ __launch_bounds__(NT) void k_ft_update(DeviceLU lu, FtState st, FtProblem pb) {
    extern __shared__ __align__(16) char lds[];
    FtCtxT<kTier> c;
    ft_bind(c, lds, lu, st);
    ft_load(c, lu, st, nullptr);
    hs_enter(c, st);
    if constexpr (kTier >= 2) {
        hs_clear(c.sp, c.bsp, c.gs, c.bsp_words, c.m);
        for (int k = threadIdx.x; k < c.m; k += NT) { const double v = st.spike[k]; if (v != 0.0) { c.sp[k] = v; hs_mark(c.bsp, c.gs, k); } }
    } else {
        for (int k = threadIdx.x; k < c.m; k += NT) c.sp[k] = st.spike[k];
    }
    __syncthreads();
    const bool updated = ft_update(lu, st, c, pb.rec->r);
    ft_store(c, st, nullptr, !updated ? 2 : (c.t >= st.max_updates || c.t >= c.tcap) ? 1 : 0);
    hs_leave(c, st, true);
}

void ft_allow_lds(const void* fn, int bytes) {
    static std::vector<const void*> done;
    for (auto f : done) if (f == fn) return;
    (void)bytes;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kFtLdsBudget) != hipSuccess) (void)hipGetLastError();
    done.push_back(fn);
}

}  // namespace

size_t ft_lds_base_bytes(int32_t m, int32_t tcap, int32_t eta_cap, int32_t tier, int32_t rhs_cap) {
    return (size_t)ft_layout(m, tcap, eta_cap, tier, rhs_cap).total;
}
int64_t ft_schedule_stage_bytes(int32_t m, int64_t nnz, int32_t n_levels, int32_t n_seg) { return schedule_lds_bytes(m, nnz, n_levels, n_seg); }

// (st.big selects the instantiation: ft_layout, the index width of the images and where the spike, the permutations and the
// eta pool live all follow from it)
#define FT_LAUNCH(kernel, ...)                                                                                        \
    do {                                                                                                              \
        if (st.big >= 2) {                                                                                            \
            ft_allow_lds(reinterpret_cast<const void*>(kernel<2>), st.lds_bytes);                                     \
            hipLaunchKernelGGL(kernel<2>, dim3(1), dim3(NT), (size_t)st.lds_bytes, s, __VA_ARGS__);                   \
        } else if (st.big) {                                                                                          \
            ft_allow_lds(reinterpret_cast<const void*>(kernel<1>), st.lds_bytes);                                     \
            hipLaunchKernelGGL(kernel<1>, dim3(1), dim3(NT), (size_t)st.lds_bytes, s, __VA_ARGS__);                   \
        } else {                                                                                                      \
            ft_allow_lds(reinterpret_cast<const void*>(kernel<0>), st.lds_bytes);                                     \
            hipLaunchKernelGGL(kernel<0>, dim3(1), dim3(NT), (size_t)st.lds_bytes, s, __VA_ARGS__);                   \
        }                                                                                                             \
    } while (0)

void launch_ft_run(const DeviceLU& lu, const FtState& st, const FtProblem& pb, int64_t max_pivots, hipStream_t s) {
    FT_LAUNCH(k_ft_run, lu, st, pb, (long long)max_pivots);
}

void launch_ft_ftran(const DeviceLU& lu, const FtState& st, const FtProblem& pb, int32_t column, const double* rhs, double* alpha,
                     hipStream_t s) {
    FT_LAUNCH(k_ft_ftran, lu, st, pb, column, rhs, alpha);
}

void launch_ft_btran(const DeviceLU& lu, const FtState& st, const FtProblem& pb, int32_t row, const double* rhs, double* rho,
                     hipStream_t s) {
    FT_LAUNCH(k_ft_btran, lu, st, pb, row, rhs, rho);
}

// The basis changes the pivot kernel made while the host was factorising an earlier basis, applied to those fresh factors:
// per change the spike of the entering column (L solve + the etas replayed so far) and the Forrest-Tomlin update of the
// leaving position -- what Carry::bring_into_basis does minus everything that is not the factorisation (b, -pi, the basis
// array are already current).
template <int kTier>
__global__ __launch_bounds__(NT) void k_ft_replay(DeviceLU lu, FtState st, FtProblem pb, int count) {
    extern __shared__ __align__(16) char lds[];
    FtCtxT<kTier> c;
    ft_bind(c, lds, lu, st);
    ft_load(c, lu, st, nullptr);
    hs_enter(c, st);
    int need = 0;
    for (int i = 0; i < count; ++i) {
        if (c.t >= c.tcap) { need = 2; break; }
        const int r = st.journal[2 * i], q = st.journal[2 * i + 1];
        ft_scatter_column(st, pb, c, q);
        ft_ftran(lu, st, c, true);
        if (!ft_update(lu, st, c, r)) { need = 2; break; }
    }
    if (!need && (c.t >= st.max_updates || c.t >= c.tcap)) need = 1;
    ft_store(c, st, nullptr, need);
    hs_leave(c, st, true);
}

void launch_ft_replay(const DeviceLU& lu, const FtState& st, const FtProblem& pb, int32_t count, hipStream_t s) {
    FT_LAUNCH(k_ft_replay, lu, st, pb, (int)count);
}

void launch_ft_update(const DeviceLU& lu, const FtState& st, const FtProblem& pb, hipStream_t s) {
    FT_LAUNCH(k_ft_update, lu, st, pb);
}

}  // namespace relp
