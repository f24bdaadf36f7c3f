// cda_hip.hip - the C-ABI of include/cda.h and the kernels that do not own a market-wave (gfx950 only; no CPU fallback, no shims).
//
// File map of csrc/:
//   cda_dec.hpp      28-digit decimal ledger arithmetic (Python Decimal semantics)
//   cda_libm.hpp     glibc's log1p / exp restated (what numpy's normal sampler calls on the host)
//   cda_market.hpp   everything of a market-wave that does not depend on the book capacity: record header, accounts, numpy RNG
//   cda_book.inc     the pooled order book, matching, ledger functions, observation - compiled once per book capacity
//   cda_kernels.inc  the market-wave kernels (all: one wave64 per market, WPB markets per workgroup, no inter-wave
//                    communication), included below for CAP = 256 (cda::cap256) and CAP = 512 (cda::cap512):
//                      k_reset         reset(seed)            continuousDoubleAuction_env.py:175-231
//                      k_step          step(actions)          continuousDoubleAuction_env.py:265-309  <- the hot kernel
//                      k_run_random    CDA_rand.run_random    CDA_rand.py:40-85
//                      k_place_order   Trader.place_order     agent/trader.py:49-106   (test hook, 1 market)
//                      k_mark_to_mkt   Exchg_Helper.mark_to_mkt                         (test hook, 1 market)
//                      k_raw_snapshot  agg_LOB_raw            exchg/state_helper.py:159-160
//   here             k_init_arena, k_random_actions, k_nav_conservation, k_check_invariants, k_flags, k_book_peak, k_handback_unpack,
//                    self-tests, and the host side: arena, capacity dispatch, every extern "C" entry point
//   (the measuring probes - operation micro-benchmark, clock probe, PMC calibration - live in tools/csrc/cda_tools.hip, a
//    library of their own: nothing of them is in the product)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <dlfcn.h>

#include "../../include/cda.h"
#include "../../include/cda_random_agents.h"
#include "cda_dec.hpp"
#include "cda_market.hpp"
#include <type_traits>
#include <mutex>
#include "../../include/cda_mlp.h"          // the network's layout constants (n_hist = 4): k_policy_step evaluates the policy inside the step kernel
namespace cda { namespace mlpdev {
#include "cda_mlp_dev.inc"
} }
// ... and for the other compiled history depths (CDA_MLP_HIST_VARIANTS): the header's layout macros follow CDA_MLP_HIST
#undef CDA_MLP_HIST
#define CDA_MLP_HIST 1
namespace cda { namespace mlpdev_h1 {
#include "cda_mlp_dev.inc"
} }
#undef CDA_MLP_HIST
#define CDA_MLP_HIST 2
namespace cda { namespace mlpdev_h2 {
#include "cda_mlp_dev.inc"
} }
#undef CDA_MLP_HIST
#define CDA_MLP_HIST 3
namespace cda { namespace mlpdev_h3 {
#include "cda_mlp_dev.inc"
} }
#undef CDA_MLP_HIST
#define CDA_MLP_HIST 6
namespace cda { namespace mlpdev_h6 {
#include "cda_mlp_dev.inc"
} }
#undef CDA_MLP_HIST
#define CDA_MLP_HIST 7
namespace cda { namespace mlpdev_h7 {
#include "cda_mlp_dev.inc"
} }
#undef CDA_MLP_HIST
#define CDA_MLP_HIST 8
namespace cda { namespace mlpdev_h8 {
#include "cda_mlp_dev.inc"
} }
#undef CDA_MLP_HIST
#define CDA_MLP_HIST 4
#pragma clang fp contract(off)              // (the includes above switched contraction on for the network's arithmetic; everything below is the env's)

using namespace cda;

#ifndef CDA_WPB
#define CDA_WPB 4            // markets (waves) per workgroup
#endif
// The step kernel is register-budgeted for 4 waves per SIMD (128 VGPRs, no spills).  A second build for 6 waves per
// SIMD (80 VGPRs) used to serve batches beyond 4096 markets; as the ledger code grew it needed > 100 spilled VGPRs
// and measured 20-27 % SLOWER than this build at 8 k..64 k markets, so it is gone.
#ifndef CDA_MIN_WAVES
#define CDA_MIN_WAVES 4
#endif
// (experiments: -DCDA_MIN_WAVES=5 budgets the step kernels for FIVE waves per SIMD - 96 VGPRs -, CDA_VGPR_CAP_ATTR takes any other attribute; profiles/r06/five_waves_per_simd.txt)
#ifndef CDA_VGPR_CAP_ATTR
#define CDA_VGPR_CAP_ATTR
#endif

// ------------------------------------------------------------------------------------------
// device helpers shared by the kernels
// ------------------------------------------------------------------------------------------
struct MarketPtrs {
    uint32_t* hdr; uint32_t* acc; float* hist; int32_t* book;
    int32_t* spill;          // the market's HBM tier (cda_book.inc): int32 hdr[16], then the two sides' rings
};
// arena = [N market records][done_buf: N bytes, padded to 256][episode metrics: f64 [N][A][CDA_EM_AGENT_FIELDS], f64 [N][CDA_EM_ENV_FIELDS]][N spill regions]
__host__ __device__ static inline size_t spill_region_bytes(int32_t spill_cap) { return spill_cap > 0 ? 64 + (size_t)spill_cap * 2 * BOOK_FIELDS * 4 : 0; }
__host__ __device__ static inline size_t em_agent_off(const Params& P) { return (size_t)P.n_markets * (size_t)P.lay.stride + (((size_t)P.n_markets + 255) & ~(size_t)255); }
__host__ __device__ static inline size_t em_env_off(const Params& P) { return em_agent_off(P) + (size_t)P.n_markets * (size_t)P.cfg.num_agents * CDA_EM_AGENT_FIELDS * 8; }
__host__ __device__ static inline size_t spill_arena_off(const Params& P) { return em_env_off(P) + (size_t)P.n_markets * CDA_EM_ENV_FIELDS * 8; }
__device__ __forceinline__ MarketPtrs market_ptrs(uint8_t* arena, const Params& P, int mi) {
    uint8_t* rec = arena + (size_t)mi * (size_t)P.lay.stride;
    MarketPtrs r;
    r.hdr = (uint32_t*)rec; r.acc = (uint32_t*)(rec + P.lay.acc_off);
    r.hist = (float*)(rec + P.lay.hist_off); r.book = (int32_t*)(rec + P.lay.book_off);
    r.spill = (int32_t*)(arena + spill_arena_off(P) + (size_t)mi * spill_region_bytes(P.lay.spill_cap));
    return r;
}
// One 16-byte request per lane moves the accounts (36 lanes at 4 agents) and the history ring (42 lanes at n_hist 4):
// both are issued before anything is waited for, next to the header word and the book prefetch, so the whole record
// costs ONE HBM round trip.  (A word-by-word loop serialises a round trip per 64 words.)
struct VecPrefetch { uint4 v; bool has; };
__device__ __forceinline__ VecPrefetch vec_prefetch(const void* src, int nbytes, int lane) {
    VecPrefetch p; p.has = lane < (nbytes >> 4);
    p.v = reinterpret_cast<const uint4*>(src)[p.has ? lane : 0];     // every lane requests (nbytes >= 16): no branch, hence no wait, around the load
    return p;
}
__device__ __forceinline__ void vec_finish(void* dst, const void* src, int nbytes, const VecPrefetch& p, int lane) {   // 16-byte aligned, nbytes % 4 == 0
    const int n16 = nbytes >> 4;
    if (p.has) reinterpret_cast<uint4*>(dst)[lane] = p.v;
    for (int i = lane + WAVE; i < n16; i += WAVE) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
    for (int w = (n16 << 2) + lane; w < (nbytes >> 2); w += WAVE) reinterpret_cast<uint32_t*>(dst)[w] = reinterpret_cast<const uint32_t*>(src)[w];
}
// ------------------------------------------------------------------------------------------
// reset
// ------------------------------------------------------------------------------------------
// dynamic LDS of a workgroup: [decimal power-of-ten table (640 B)] [k_step only: ziggurat wi, ki (4 KB), PCG jump table (512 B)] [wave 0 image] ...
constexpr int ZIG_LDS_BYTES = (2 * 256 + PCG_JUMP_WORDS64) * 8;
__device__ __forceinline__ void zig_tables_init() {       // every thread of the workgroup, before the first __syncthreads
    unsigned long long* t = reinterpret_cast<unsigned long long*>(cda_smem + DEC_TABLE_BYTES);
    for (int i = (int)threadIdx.x; i < 256; i += (int)blockDim.x) { t[i] = cda_zig_wi_bits[i]; t[256 + i] = cda_zig_ki[i]; }
    for (int i = (int)threadIdx.x; i < PCG_JUMP_WORDS64; i += (int)blockDim.x) t[512 + i] = reinterpret_cast<const unsigned long long*>(&PCG_JUMP)[i];
}

// The same staging for a 64 * CDA_WPB = 256-thread workgroup, in two halves: stage_tables_issue() only REQUESTS this thread's
// table words (five independent loads in one basic block), stage_tables_commit() writes them to LDS and holds the
// workgroup's one __syncthreads.  k_step issues the table requests first, then the market record's and the actions', and
// commits in between: the wait for the (L2-resident) tables then leaves the record's HBM round trip in flight instead of
// paying for it once per table, which is what separate load / wait / store loops did.
struct TablePrefetch { unsigned long long wi, ki, jump; double rcp; uint32_t pow10; };
static_assert(PCG_JUMP_WORDS64 <= 64 * CDA_WPB && DEC_LDS_POW * 4 <= 64 * CDA_WPB && 64 * CDA_WPB == 256, "one table word of each kind per thread");
__device__ __forceinline__ TablePrefetch stage_tables_issue() {
    const int t = (int)threadIdx.x;
    TablePrefetch q;
    const int tj = t < PCG_JUMP_WORDS64 ? t : PCG_JUMP_WORDS64 - 1, tp = t < DEC_LDS_POW * 4 ? t : DEC_LDS_POW * 4 - 1;   // clamped: no branch around a load
    q.wi = cda_zig_wi_bits[t]; q.ki = cda_zig_ki[t];
    q.jump = reinterpret_cast<const unsigned long long*>(&PCG_JUMP)[tj];
    q.pow10 = POW10.v[tp >> 2][tp & 3];
    q.rcp = RCP10.v[t < 10 ? t : 9];
    return q;
}
__device__ __forceinline__ void stage_tables_commit(const TablePrefetch& q) {
    if ((uint32_t)(uintptr_t)cda_smem != 0u) __builtin_trap();          // see lds_pow10()
    const int t = (int)threadIdx.x;
    unsigned long long* z = reinterpret_cast<unsigned long long*>(cda_smem + DEC_TABLE_BYTES);
    z[t] = q.wi; z[256 + t] = q.ki;
    // clamped indices again: the surplus threads rewrite the last word with the same value - a store under a condition would
    // let the compiler sink its LOAD into the conditional block, behind the record's requests
    z[512 + (t < PCG_JUMP_WORDS64 ? t : PCG_JUMP_WORDS64 - 1)] = q.jump;
    reinterpret_cast<uint32_t*>(cda_smem)[t < DEC_LDS_POW * 4 ? t : DEC_LDS_POW * 4 - 1] = q.pow10;
    reinterpret_cast<double*>(cda_smem + DEC_RCP_OFF)[t < 10 ? t : 9] = q.rcp;           // RN(1 / 10^k), folded at compile time
    __syncthreads();
}

// ------------------------------------------------------------------------------------------
// step - the hot kernel
// ------------------------------------------------------------------------------------------

struct FinCapture { float* obs; int32_t* count; int32_t* index_out; int32_t cap; };      // episode-end capture: see capture_final_obs (cda_kernels.inc)
struct StepArgs {
    const int32_t* category; const float* size_mean; const float* size_sigma;
    const int32_t* price; const int32_t* price_offset; const uint8_t* present;
    float* obs_out; double* reward_out; uint8_t* terminated_out; uint8_t* truncated_out;
    uint8_t* done_out;                  // auto_reset only: terminated | truncated, the mask of the k_reset launch that follows
    cda_info_ptrs info; int has_info;
    int first_market, end_market;       // this launch steps the markets [first_market, end_market); every array argument is the full [N, ...] one
    uint8_t* handback; int32_t handback_stride;   // cda_set_handback: compact per-market records of what is new this step (NULL = off)
    float* fin_obs; int32_t* fin_count; int32_t* fin_index_out; int32_t fin_cap;   // episode-end capture (cda_step_range_capture; NULL = off): read in the cold reset paths only
    unsigned long long* phase_cycles;   // debug builds only (CDA_PHASE_TIMING): [N,40] cycle stamps
    int dbg_skip;                       // debug builds only (CDA_DEBUG_SKIP): phases to leave out (tools/fixed_cost_probe.py); results are then wrong
};
#ifdef CDA_DEBUG_SKIP
#define CDA_DBG_HAS(S, bit) (((S).dbg_skip & (bit)) != 0)
#define CDA_DBG_SKIP(S, bit, stmt) do { if (CDA_DBG_HAS(S, bit)) stmt; } while (0)
#else
#define CDA_DBG_HAS(S, bit) false
#define CDA_DBG_SKIP(S, bit, stmt) do {} while (0)
#endif

__device__ __forceinline__ float clampf(float v, float lo, float hi) { if (!(v >= lo)) return lo; if (!(v <= hi)) return hi; return v; }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---- the phases of one step, shared by k_step (one step per launch) and k_run_random (a whole episode per launch) ----
#ifdef CDA_PHASE_TIMING
#define PH_MARK(ph, i) do { unsigned long long _t = __builtin_readcyclecounter(); if ((ph) && lane == 0) (ph)[i] = _t; } while (0)
#else
#define PH_MARK(ph, i) do {} while (0)
#endif
struct LaneAction { int cat, level, off; float mean, sigma; bool pres; int ord; };     // lane a: agent a's action words (unclamped); ord: its place in the caller's dict
struct StepReward { double r, t0, t1, t2, t3, t4, drawdown, max_nav; bool bankrupt; };
__device__ __forceinline__ void clear_step_counters(Acc& a) {          // exchg_helper.py:116-120
    a.num_trades_step = 0; a.num_passive_fills_step = 0; a.order_step_placed = 0; a.num_rejected_step = 0;
}

// ------------------------------------------------------------------------------------------
// episode metrics (include/cda.h cda_episode_metrics_*): the reference callback's per-episode tallies and its end-of-episode check, on the device
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ EpStats* ep_stats(const MarketPtrs& mp, const Params& P) { return reinterpret_cast<EpStats*>(reinterpret_cast<uint8_t*>(mp.hdr) + P.lay.ep_off); }
// fire-and-forget: a hardware atomic whose result nobody reads is a store-like request (no return value, no wait).  Only the market's own wave ever touches its tallies,
// so the order of the additions to one address is the program's: the f64 sums are the sums in step order, bit for bit.
__device__ __forceinline__ void ep_add(double* p, double v) {
    (void)__builtin_amdgcn_global_atomic_fadd_f64((__attribute__((address_space(1))) double*)p, v);
}
__device__ __forceinline__ void ep_add(int32_t* p, int32_t v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// on_episode_step (league_based_self_play_callback.py:541-600) for agent `lane` of this market: the step's counters are read BEFORE they are zeroed (exchg_helper.py:116-120)
__device__ __forceinline__ void ep_tally(EpStats* es, const StepReward& rw, const Acc& a, bool passed) {
    ep_add(&es->term_sum[0], rw.t0); ep_add(&es->term_sum[1], rw.t1); ep_add(&es->term_sum[2], rw.t2); ep_add(&es->term_sum[3], rw.t3); ep_add(&es->term_sum[4], rw.t4);
    ep_add(&es->term_sq[0], rw.t0 * rw.t0); ep_add(&es->term_sq[1], rw.t1 * rw.t1); ep_add(&es->term_sq[2], rw.t2 * rw.t2); ep_add(&es->term_sq[3], rw.t3 * rw.t3);
    ep_add(&es->term_sq[4], rw.t4 * rw.t4);
    ep_add(&es->ret, rw.r);
    ep_add(&es->passes, passed ? 1 : 0); ep_add(&es->rejections, a.num_rejected_step); ep_add(&es->placed, a.order_step_placed);
    ep_add(&es->trades, a.num_trades_step); ep_add(&es->passive, a.num_passive_fills_step);
}
__device__ __forceinline__ double ep_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int32_t ep_load(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ep_clear(EpStats* es) {                  // lane a: discard agent a's running tallies
    uint4* q = reinterpret_cast<uint4*>(es);
    #pragma unroll
    for (int i = 0; i < (int)(sizeof(EpStats) / 16); i++) q[i] = make_uint4(0u, 0u, 0u, 0u);
}
// on_episode_end (league_based_self_play_callback.py:627-755) for ONE market whose episode is over, called by every lane of its wave with the decimal tables staged and
// the market's HEADER in memory as the episode's last step left it (done mask, step count).  `acc`: the accounts of that step - the wave's LDS image where the step ran in
// this launch (what this wave has just stored need not be what its L1 returns), NULL = the record in memory.  Cold by construction: once per market and episode; called
// from the kernels' own level, never from inside another out-of-line routine (a kernel's scratch is the deepest call chain's).
//   - sum of NAV in agent order, exactly (Decimal arithmetic, prec 28: the code of k_nav_conservation), against num_agents x init_cash;
//   - agent a's tallies and last-step account figures (_log_episode_account :418-470) -> accumulators of (market, a); the episode's -> accumulators of the market;
//   - the running tallies are cleared, the header in memory is marked ST_EP_SUMMARISED (+ CDA_FLAG_NAV_CONSERVATION on a violation).
// Returns whether conservation was violated (for a caller that holds the header in registers and is about to store it).
__attribute__((noinline, cold)) __device__ bool episode_summarise(uint8_t* arena, const Params* Pp, int mi, const Acc* acc) {
    const Params& P = *Pp;
    const int lane = lane_id(), A = P.cfg.num_agents;
    uint8_t* rec = arena + (size_t)mi * (size_t)P.lay.stride;
    uint32_t* hdr = reinterpret_cast<uint32_t*>(rec);
    if (!acc) acc = reinterpret_cast<const Acc*>(rec + P.lay.acc_off);
    EpStats* s = reinterpret_cast<EpStats*>(rec + P.lay.ep_off) + (lane < A ? lane : 0);
    double* f = reinterpret_cast<double*>(arena + em_agent_off(P)) + ((size_t)mi * (size_t)A + (size_t)(lane < A ? lane : 0)) * CDA_EM_AGENT_FIELDS;
    double* M = reinterpret_cast<double*>(arena + em_env_off(P)) + (size_t)mi * CDA_EM_ENV_FIELDS;
    // EVERY load first - the header words, the tallies (written by atomics: read where they live), the accumulator rows - and unconditionally (a lane beyond the agents
    // reads agent 0's): the seventy-odd requests are ONE round trip under the decimal arithmetic below.  Read and updated field by field, each store would fence the
    // next load behind it (and every lock-step episode end of a batch is N of these at once).
    const uint32_t done_mask = hdr[H_DONE_MASK], status = hdr[H_STATUS], flags = hdr[H_FLAGS];
    const int t_step = (int)hdr[H_T_STEP];
    double ts[CDA_NUM_REWARD_TERMS], tq[CDA_NUM_REWARD_TERMS], fv[CDA_EM_BANKRUPT + 1], mv[CDA_EM_ENV_FIELDS];
    #pragma unroll
    for (int j = 0; j < CDA_NUM_REWARD_TERMS; j++) { ts[j] = ep_load(&s->term_sum[j]); tq[j] = ep_load(&s->term_sq[j]); }
    const double ret = ep_load(&s->ret);
    const int passes = ep_load(&s->passes), rejections = ep_load(&s->rejections), placed = ep_load(&s->placed), trades = ep_load(&s->trades), passive = ep_load(&s->passive);
    #pragma unroll
    for (int j = 0; j <= CDA_EM_BANKRUPT; j++) fv[j] = f[j];
    #pragma unroll
    for (int j = 0; j < CDA_EM_ENV_FIELDS; j++) mv[j] = M[j];
    __builtin_amdgcn_sched_barrier(0);
    D total = d_zero();
    for (int a = 0; a < A; a++) total = d_add(total, ld_dec(acc[a].nav));
    D err = d_sub(total, d_mul_int(d_from_i64(P.cfg.init_cash), (uint32_t)A));
    err.sign = 0;
    uint32_t ferr = 0;
    const double e = d_to_double(err, &ferr);
    const bool viol = e > P.ep_tol || ferr != 0;
    double ratio = -1.0;                                                 // lane a: agent a's passive share of its own fills, if it has enough of them
    if (lane < A) {
        const Acc& ac = acc[lane];
        uint32_t f2 = 0;
        const D navd = ld_dec(ac.nav);
        const double nav = d_to_double(navd, &f2);
        const D ddd = d_sub(ld_dec(ac.max_nav), navd);
        const double dd = d_sgn(ddd) > 0 ? d_to_double(ddd, &f2) : 0.0;      // float(max(0, max_nav - nav)), reward_helper.py:60-62
        if (trades >= 5) ratio = (double)passive / (double)trades;
        const bool first = fv[CDA_EM_EPISODES] == 0.0;
        fv[CDA_EM_EPISODES] += 1.0;
        fv[CDA_EM_AGENT_STEPS] += (double)t_step;
        fv[CDA_EM_PASSES] += (double)passes; fv[CDA_EM_REJECTIONS] += (double)rejections; fv[CDA_EM_PLACED] += (double)placed;
        fv[CDA_EM_TRADES] += (double)trades; fv[CDA_EM_PASSIVE] += (double)passive;
        #pragma unroll
        for (int j = 0; j < CDA_NUM_REWARD_TERMS; j++) { fv[CDA_EM_TERM_SUM + j] += ts[j]; fv[CDA_EM_TERM_SQ + j] += tq[j]; }
        fv[CDA_EM_RETURN_SUM] += ret; fv[CDA_EM_RETURN_SQ] += ret * ret;
        fv[CDA_EM_NAV_SUM] += nav;
        fv[CDA_EM_NAV_MIN] = first ? nav : fmin(fv[CDA_EM_NAV_MIN], nav);
        fv[CDA_EM_NAV_MAX] = first ? nav : fmax(fv[CDA_EM_NAV_MAX], nav);
        fv[CDA_EM_DRAWDOWN_SUM] += dd;
        fv[CDA_EM_ABS_POSITION_SUM] += fabs((double)ac.net_position);
        fv[CDA_EM_NUM_TRADES_SUM] += (double)ac.num_trades;
        if (ratio >= 0.0) { fv[CDA_EM_MAKER_RATIO_SUM] += ratio; fv[CDA_EM_MAKER_RATIO_N] += 1.0; fv[CDA_EM_MAKER_RATIO_MAX] = fmax(fv[CDA_EM_MAKER_RATIO_MAX], ratio); }
        fv[CDA_EM_BANKRUPT] += (double)((done_mask >> lane) & 1u);
        #pragma unroll
        for (int j = 0; j <= CDA_EM_BANKRUPT; j++) f[j] = fv[j];
        ep_clear(s);
    }
    double best = -1.0;                                                  // maker_fill_ratio_max (:344-372): the episode's most maker-like agent
    for (int a = 0; a < A; a++) best = fmax(best, __shfl(ratio, a, WAVE));
    if (lane == 0) {
        mv[CDA_EM_ENV_EPISODES] += 1.0;
        mv[CDA_EM_ENV_NAV_VIOLATIONS] += viol ? 1.0 : 0.0;
        mv[CDA_EM_ENV_NAV_ERROR_SUM] += e;
        mv[CDA_EM_ENV_NAV_ERROR_MAX] = fmax(mv[CDA_EM_ENV_NAV_ERROR_MAX], e);
        if (best >= 0.0) { mv[CDA_EM_ENV_MAKER_MAX_SUM] += best; mv[CDA_EM_ENV_MAKER_MAX_N] += 1.0; }
        mv[CDA_EM_ENV_STEPS] += (double)t_step;
        mv[CDA_EM_ENV_TERMINATED] += __popc(done_mask) == A ? 1.0 : 0.0;
        #pragma unroll
        for (int j = 0; j < CDA_EM_ENV_FIELDS; j++) M[j] = mv[j];
        hdr[H_STATUS] = status | (uint32_t)ST_EP_SUMMARISED;
        if (viol) hdr[H_FLAGS] = flags | CDA_FLAG_NAV_CONSERVATION;
    }
    return viol;
}

// ------------------------------------------------------------------------------------------
// run_random - a whole random-agent episode per launch (CDA_rand.py:40-85)
// ------------------------------------------------------------------------------------------
struct RunArgs {
    int32_t n_steps; uint64_t seed, market_base;
    float* obs_out; double* return_out; uint8_t* terminated_out; uint8_t* truncated_out; int32_t* steps_out;
};

// ---- the market-wave kernels, one build per book capacity ------------------------------------------------
#define CDA_CAP 256
#define CDA_CAPNS cap256
#include "cda_kernels.inc"
#undef CDA_CAP
#undef CDA_CAPNS
#define CDA_CAP 512
#define CDA_CAPNS cap512
#include "cda_kernels.inc"
#undef CDA_CAP
#undef CDA_CAPNS

// Sum-of-NAV conservation check, one thread per market (callbk/league_based_self_play_callback.py:679-704)
__global__ void k_nav_conservation(const uint8_t* arena, Params P, double tol, double* err_out, uint8_t* viol_out) {
    dec_tables_init();
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= P.n_markets) return;
    const Acc* acc = reinterpret_cast<const Acc*>(arena + (size_t)i * (size_t)P.lay.stride + (size_t)P.lay.acc_off);
    D total = d_zero();
    for (int a = 0; a < P.cfg.num_agents; a++) total = d_add(total, ld_dec(acc[a].nav));
    D err = d_sub(total, d_mul_int(d_from_i64(P.cfg.init_cash), (uint32_t)P.cfg.num_agents));
    err.sign = 0;
    uint32_t ferr = 0;
    const double e = d_to_double(err, &ferr);
    err_out[i] = e;
    if (viol_out) viol_out[i] = (uint8_t)(e > tol || ferr != 0);
}
// Structural invariants of every market, one thread per market (size-independent property check for the full-size runs):
// the reference's OrderTree keeps each side sorted by price (ordertree.py:44-58), a book is never left crossed
// (orderbook.py:162-194 matches before it rests), resting quantities are positive, cash_on_hold is exactly the value of
// the trader's own resting orders (cash_processor.py:15-29 / :85-97 escrow and release) and every unit long is a unit
// short (account.py:196-213).
__host__ __device__ static inline int book_phys_rt(int cap, int s, int i) { return s == 0 ? i : cap - 1 - i; }     // cda_book.inc book_phys with a run-time capacity
// order i of side sd of a market, wherever it lives: the first `tile_n` orders in the record's tile, the rest in the spill ring
struct BookView {
    const int32_t* tile; const int32_t* spill; int cap, spill_cap; int tile_n[2], tail_n[2], tail_base[2];
    __host__ __device__ int total(int sd) const { return tile_n[sd] + tail_n[sd]; }
    __host__ __device__ int32_t get(int sd, int f, int i) const {
        if (i < tile_n[sd]) return tile[f * cap + book_phys_rt(cap, sd, i)];
        const uint32_t sl = (uint32_t)(tail_base[sd] + (i - tile_n[sd])) & (uint32_t)(spill_cap - 1);
        return spill[16 + (size_t)(sd * BOOK_FIELDS + f) * (size_t)spill_cap + sl];
    }
};
__host__ __device__ static inline BookView book_view(const uint32_t* h, const int32_t* tile, const int32_t* spill, int cap, int spill_cap) {
    BookView v;
    v.tile = tile; v.spill = spill; v.cap = cap; v.spill_cap = spill_cap;
    v.tile_n[0] = (int)h[H_N_BIDS]; v.tile_n[1] = (int)h[H_N_ASKS];
    for (int sd = 0; sd < 2; sd++) {
        const bool has = spill_cap > 0 && (h[H_STATUS] & (uint32_t)(ST_TAIL_BID << sd)) != 0;
        v.tail_n[sd] = has ? spill[sd] : 0; v.tail_base[sd] = has ? spill[2 + sd] : 0;
    }
    return v;
}
__global__ void k_check_invariants(const uint8_t* arena, Params P, int CAP, uint32_t* out) {
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= P.n_markets) return;
    const uint8_t* rec = arena + (size_t)i * (size_t)P.lay.stride;
    const uint32_t* h = (const uint32_t*)rec;
    const Acc* acc = (const Acc*)(rec + P.lay.acc_off);
    const int A = P.cfg.num_agents;
    const BookView bv = book_view(h, (const int32_t*)(rec + P.lay.book_off), (const int32_t*)(arena + spill_arena_off(P) + (size_t)i * spill_region_bytes(P.lay.spill_cap)), CAP, P.lay.spill_cap);
    uint32_t v = 0;
    if (bv.tile_n[0] < 0 || bv.tile_n[1] < 0 || bv.tile_n[0] + bv.tile_n[1] > CAP || bv.tail_n[0] < 0 || bv.tail_n[1] < 0 ||
        bv.tail_n[0] > P.lay.spill_cap || bv.tail_n[1] > P.lay.spill_cap) { out[i] = CDA_INV_BOOK_COUNT; return; }
    if ((bv.tail_n[0] > 0 && bv.tile_n[0] == 0) || (bv.tail_n[1] > 0 && bv.tile_n[1] == 0)) v |= CDA_INV_BOOK_COUNT;   // the top of a side is always in its tile
    long long held[CDA_MAX_AGENTS];
    for (int a = 0; a < CDA_MAX_AGENTS; a++) held[a] = 0;
    for (int sd = 0; sd < 2; sd++) {
        const int n = bv.total(sd);
        int32_t prev = 0;
        for (int k = 0; k < n; k++) {
            const int32_t p = bv.get(sd, 0, k), q = bv.get(sd, 1, k);
            const int owner = bv.get(sd, 2, k) & 15;
            if (q <= 0 || p <= 0) v |= CDA_INV_QTY;
            if (owner >= A) v |= CDA_INV_OWNER; else held[owner] += (long long)p * (long long)q;
            if (k > 0 && (sd == 0 ? prev < p : prev > p)) v |= sd == 0 ? CDA_INV_BIDS_SORTED : CDA_INV_ASKS_SORTED;
            prev = p;
        }
    }
    if (bv.total(0) > 0 && bv.total(1) > 0 && bv.get(0, 0, 0) >= bv.get(1, 0, 0)) v |= CDA_INV_CROSSED;
    long long net = 0;
    for (int a = 0; a < A; a++) {
        net += acc[a].net_position;
        const cda_dec& hd = acc[a].hold;                                   // exact: hold * 10^-exp == held
        u128 c = ((u128)hd.w[2] << 64) | ((u128)hd.w[1] << 32) | (u128)hd.w[0];
        u128 want = (u128)(unsigned long long)held[a];
        bool ok = !hd.sign || c == 0;
        if (hd.exp <= 0 && hd.exp >= -18) { for (int k = 0; k < -hd.exp; k++) want *= 10u; ok = ok && c == want; }
        else ok = false;
        if (!ok) v |= CDA_INV_ESCROW;
    }
    if (net != 0) v |= CDA_INV_NET_POSITION;
    out[i] = v;
}
// cda_episode_metrics_collect, stage 1: block b reduces its share of the (market, agent) accumulators by module - thread (group g of 8, field f of 32) walks the
// pairs g + 8 (b + EM_BLOCKS k) and keeps one accumulator per module in registers - and of the per-market accumulators; the groups of a block are combined in LDS
// in a fixed order.  Stage 2 adds the blocks' partials in block order: the result does not depend on timing.
constexpr int EM_BLOCKS = 256;
__device__ __forceinline__ double em_combine(int field, double x, double y) {
    return field == CDA_EM_NAV_MIN ? fmin(x, y) : ((field == CDA_EM_NAV_MAX || field == CDA_EM_MAKER_RATIO_MAX) ? fmax(x, y) : x + y);
}
__device__ __forceinline__ double em_identity(int field) {
    return field == CDA_EM_NAV_MIN ? __longlong_as_double(0x7ff0000000000000LL) : ((field == CDA_EM_NAV_MAX || field == CDA_EM_MAKER_RATIO_MAX) ? __longlong_as_double(0xfff0000000000000LL) : 0.0);
}
__global__ __launch_bounds__(256) void k_em_partial(uint8_t* arena, Params P, const int32_t* module_of, int n_modules, int clear, double* partials) {
    __shared__ double red[8][CDA_EM_AGENT_FIELDS];
    const int tid = (int)threadIdx.x, g = tid >> 5, f = tid & 31, b = (int)blockIdx.x;
    double* F = reinterpret_cast<double*>(arena + em_agent_off(P));
    double* M = reinterpret_cast<double*>(arena + em_env_off(P));
    const long long pairs = (long long)P.n_markets * P.cfg.num_agents;
    double acc[CDA_EM_MAX_MODULES];
    #pragma unroll
    for (int k = 0; k < CDA_EM_MAX_MODULES; k++) acc[k] = em_identity(f);
    for (long long p = (long long)b * 8 + g; p < pairs; p += (long long)EM_BLOCKS * 8) {
        double* row = F + p * CDA_EM_AGENT_FIELDS;
        const double n = row[CDA_EM_EPISODES], v = row[f];
        const int m = module_of ? module_of[p] : 0;
        if (n != 0.0 && m >= 0 && m < n_modules) {
            #pragma unroll
            for (int k = 0; k < CDA_EM_MAX_MODULES; k++) acc[k] = k == m ? em_combine(f, acc[k], v) : acc[k];
        }
        if (clear) { __syncwarp(); row[f] = 0.0; }            // (every lane of the 32 has read row[0] before anyone clears it)
    }
    double* out = partials + (size_t)b * (CDA_EM_MAX_MODULES * CDA_EM_AGENT_FIELDS + CDA_EM_ENV_FIELDS);
    #pragma unroll
    for (int k = 0; k < CDA_EM_MAX_MODULES; k++) {             // (unrolled: acc[] stays in registers)
        if (k < n_modules) {
            __syncthreads();
            red[g][f] = acc[k];
            __syncthreads();
            if (g == 0) { double r = red[0][f]; for (int q = 1; q < 8; q++) r = em_combine(f, r, red[q][f]); out[k * CDA_EM_AGENT_FIELDS + f] = r; }
        }
    }
    // the per-market accumulators: thread (group of 32, field of 8)
    __syncthreads();
    const int g2 = tid >> 3, f2 = tid & 7;
    double a2 = 0.0;
    for (int i = b * 32 + g2; i < P.n_markets; i += EM_BLOCKS * 32) {
        double* row = M + (size_t)i * CDA_EM_ENV_FIELDS;
        const double v = row[f2];
        a2 = f2 == CDA_EM_ENV_NAV_ERROR_MAX ? fmax(a2, v) : a2 + v;
        if (clear) row[f2] = 0.0;
    }
    double* red2 = &red[0][0];                                 // 256 doubles
    red2[tid] = a2;
    __syncthreads();
    if (tid < CDA_EM_ENV_FIELDS) {
        double r = red2[tid];
        for (int q = 1; q < 32; q++) { const double v = red2[q * 8 + tid]; r = tid == CDA_EM_ENV_NAV_ERROR_MAX ? fmax(r, v) : r + v; }
        out[CDA_EM_MAX_MODULES * CDA_EM_AGENT_FIELDS + tid] = r;
    }
}
// stage 2: one wave per output word; lane l adds the partials of blocks l, l + 64, ... in that order, the lanes are combined by a butterfly - a fixed order again
__global__ __launch_bounds__(64) void k_em_final(const double* partials, int n_modules, double* agent_out, double* env_out) {
    const int t = (int)blockIdx.x, lane = (int)threadIdx.x;
    const size_t per = CDA_EM_MAX_MODULES * CDA_EM_AGENT_FIELDS + CDA_EM_ENV_FIELDS;
    const bool agent = t < n_modules * CDA_EM_AGENT_FIELDS;
    const int f = agent ? t % CDA_EM_AGENT_FIELDS : -1 - (t - n_modules * CDA_EM_AGENT_FIELDS);        // env words: negative
    const size_t col = agent ? (size_t)t : (size_t)(CDA_EM_MAX_MODULES * CDA_EM_AGENT_FIELDS + (t - n_modules * CDA_EM_AGENT_FIELDS));
    const size_t ncol = agent ? (size_t)(t / CDA_EM_AGENT_FIELDS) * CDA_EM_AGENT_FIELDS + CDA_EM_EPISODES : col;
    auto comb = [f](double x, double y) { return f >= 0 ? em_combine(f, x, y) : (-1 - f == CDA_EM_ENV_NAV_ERROR_MAX ? fmax(x, y) : x + y); };
    double v[EM_BLOCKS / 64], nn[EM_BLOCKS / 64];
    #pragma unroll
    for (int k = 0; k < EM_BLOCKS / 64; k++) { v[k] = partials[(size_t)(lane + 64 * k) * per + col]; nn[k] = partials[(size_t)(lane + 64 * k) * per + ncol]; }
    double r = v[0], n = nn[0];
    #pragma unroll
    for (int k = 1; k < EM_BLOCKS / 64; k++) { r = comb(r, v[k]); n += nn[k]; }
    #pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { r = comb(r, __shfl_xor(r, d, 64)); n += __shfl_xor(n, d, 64); }
    if (lane == 0) {
        if (agent) agent_out[t] = n == 0.0 ? 0.0 : r;           // a module that played nothing reports zeros, not infinities
        else env_out[t - n_modules * CDA_EM_AGENT_FIELDS] = r;
    }
}
// cda_episode_metrics_enable: the switch lives in every market's header (ST_EP_ON); the running tallies start from zero
__global__ void k_ep_enable(uint8_t* arena, Params P, int on) {
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= P.n_markets) return;
    uint8_t* rec = arena + (size_t)i * (size_t)P.lay.stride;
    uint32_t* h = (uint32_t*)rec;
    h[H_STATUS] = on ? (h[H_STATUS] | (uint32_t)ST_EP_ON) : (h[H_STATUS] & ~(uint32_t)(ST_EP_ON | ST_EP_SUMMARISED));
    uint32_t* t = (uint32_t*)(rec + P.lay.ep_off);
    for (int k = 0; k < P.cfg.num_agents * (int)(sizeof(EpStats) / 4); k++) t[k] = 0;
}
__global__ void k_flags(uint8_t* arena, Params P, uint32_t* out) {
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < P.n_markets) out[i] = ((const uint32_t*)(arena + (size_t)i * (size_t)P.lay.stride))[H_FLAGS];
}
// cda_create: every market record is built the way the reference's __init__ leaves a fresh env (accounts at init_cash:
// agent/trader.py:21-23 -> account/account.py:13-53; empty book; zero history), so that a step / run_random / place_order
// on a never-reset env works on defined state.  One thread per market; word loops (runs once per env).
__global__ void k_init_arena(uint8_t* arena, Params P) {
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= P.n_markets) return;
    uint8_t* rec = arena + (size_t)i * (size_t)P.lay.stride;
    uint32_t* w = (uint32_t*)rec;
    for (int k = 0; k < P.lay.stride / 4; k++) w[k] = 0;
    Acc* acc = (Acc*)(rec + P.lay.acc_off);
    uint32_t f = 0;
    const D cash = d_from_i64(P.cfg.init_cash);
    for (int a = 0; a < P.cfg.num_agents; a++) { st_dec(acc[a].cash, cash, f); st_dec(acc[a].nav, cash, f); st_dec(acc[a].prev_nav, cash, f); st_dec(acc[a].max_nav, cash, f); }
}
// the random agents' actions of steps [step0, step0 + n_steps) for n_markets markets: one thread per (step, market, agent)
__global__ void k_random_actions(uint64_t seed, uint64_t market_base, int step0, int n_steps, int n_markets, int num_agents,
                                 int32_t* category, float* size_mean, float* size_sigma, int32_t* price, int32_t* price_offset) {
    const size_t per_step = (size_t)n_markets * (size_t)num_agents, total = per_step * (size_t)n_steps;
    for (size_t ix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; ix < total; ix += (size_t)gridDim.x * blockDim.x) {
        const size_t t = ix / per_step, r = ix - t * per_step;
        const size_t mk = r / (size_t)num_agents, a = r - mk * (size_t)num_agents;
        cda_random_action(seed, market_base + (uint64_t)mk, (uint32_t)(step0 + (int)t), (uint32_t)a, &category[ix], &size_mean[ix], &size_sigma[ix], &price[ix], &price_offset[ix]);
    }
}
// receiving side of the hand-back: one wave per record.  Lane l < 42 holds column l of every frame of its row in registers, so
// the shift by one frame has no read-after-write hazard; restarted rows are filled with the new frame.
// n_rows_total > 0: uneven shards - segment s owns seg_row_stride rows, the LAST one the remainder (n_rows_total - s * seg_row_stride); records
// beyond a segment's rows are padding (every rank sends the same count) and are skipped.
__global__ __launch_bounds__(256) void k_handback_unpack(const uint8_t* rec, int n_seg, int seg_records, long long seg_row_stride, long long row0, int A, int H, int stride,
                                                          long long n_rows_total, float* obs_full, double* reward_full, uint8_t* term_full, uint8_t* trunc_full) {
    const long long r = (long long)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = (int)(threadIdx.x & 63);
    if (r >= (long long)n_seg * seg_records) return;
    const uint8_t* p = rec + (size_t)r * (size_t)stride;
    const long long seg = r / seg_records, local = row0 + (r % seg_records);
    if (n_rows_total > 0 && local >= (seg == n_seg - 1 ? n_rows_total - seg * seg_row_stride : seg_row_stride)) return;
    const size_t row = (size_t)(seg * seg_row_stride + local);
    const uint8_t* fl = p + CDA_SNAPSHOT_DIM * 4 + A * 8;
    const bool restarted = fl[2] != 0;
    if (lane < CDA_SNAPSHOT_DIM) {
        float* o = obs_full + row * (size_t)(H * CDA_SNAPSHOT_DIM);
        const float nf = reinterpret_cast<const float*>(p)[lane];
        float keep[CDA_MAX_HIST];
        #pragma unroll
        for (int h = 1; h < CDA_MAX_HIST; h++) keep[h] = h < H ? o[h * CDA_SNAPSHOT_DIM + lane] : 0.0f;
        #pragma unroll
        for (int h = 1; h < CDA_MAX_HIST; h++) if (h < H) o[(h - 1) * CDA_SNAPSHOT_DIM + lane] = restarted ? nf : keep[h];
        o[(H - 1) * CDA_SNAPSHOT_DIM + lane] = nf;
    }
    if (lane < A) reward_full[row * (size_t)A + (size_t)lane] = reinterpret_cast<const double*>(p + CDA_SNAPSHOT_DIM * 4)[lane];
    if (lane == 0) { term_full[row] = fl[0]; trunc_full[row] = fl[1]; }
}
__global__ void k_book_peak(const uint8_t* arena, Params P, int32_t* out) {
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < P.n_markets) out[i] = (int32_t)((const uint32_t*)(arena + (size_t)i * (size_t)P.lay.stride))[H_PEAK_ORDERS];
}

__global__ void k_selftest_dec(int op, int n, const cda_dec* a, const cda_dec* b, cda_dec* out) {
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    dec_tables_init();
    if (i >= n) return;
    D x = ld_dec(a[i]);
    D y = b ? ld_dec(b[i]) : d_zero();
    cda_dec o; o.w[0] = o.w[1] = o.w[2] = 0; o.exp = 0; o.sign = 0; o.pad = 0;
    uint32_t f = 0;
    uint32_t yc = y.w0;   // integer-valued small operand for mul/div
    switch (op) {
        case 0: st_dec(o, d_add(x, y), f); break;
        case 1: st_dec(o, d_sub(x, y), f); break;
        case 2: { D r = d_mul_u32(x, yc, y.exp); r.sign ^= y.sign; st_dec(o, r, f); break; }
        case 3: { D r = d_div_u32(x, yc); r.sign ^= y.sign; st_dec(o, r, f); break; }
        case 4: o.w[0] = (uint32_t)(d_cmp(x, y) + 1); break;
        case 6: {                                       // the transfer leaf: x + y, pad = 1 when the leaf itself produced it
            D r = d_add_order_value(x, y);
            const bool handled = r.exp != D_NOT_HANDLED;
            if (!handled) r = d_add(x, y);
            st_dec(o, r, f); o.pad = handled ? 1u : 0u; break;
        }
        default: { double d = d_to_double(x, &f); unsigned long long bits = (unsigned long long)__double_as_longlong(d); o.w[0] = (uint32_t)bits; o.w[1] = (uint32_t)(bits >> 32); o.w[2] = f; break; }
    }
    out[i] = o;
}
__global__ void k_selftest_libm(int op, int n, const double* x, double* y) {
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < n) y[i] = op == 0 ? glibc_log1p(x[i]) : (op == 1 ? glibc_exp(x[i]) : sqrt(x[i]));     // op 2: the IEEE square root of the observation's size rows
}
__global__ void k_selftest_rng(uint64_t seed, int lo, int hi, int n_steps, int n_normals, int perm_n,
                               int32_t* first_int, double* normals, int32_t* perms, uint64_t* final_state) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Mkt m; rng_seed(m, seed);
    *first_int = rng_integers(m, lo, hi);
    for (int s = 0; s < n_steps; s++) {
        for (int k = 0; k < n_normals; k++) normals[(size_t)s * (size_t)n_normals + (size_t)k] = rng_std_normal(m);
        int32_t* p = perms + (size_t)s * (size_t)perm_n;
        for (int i = 0; i < perm_n; i++) p[i] = i;
        for (int i = perm_n - 1; i >= 1; i--) { int j = (int)rng_interval(m, (uint32_t)i); int32_t t = p[i]; p[i] = p[j]; p[j] = t; }
    }
    final_state[0] = (uint64_t)(m.rng_state >> 64); final_state[1] = (uint64_t)m.rng_state;
    final_state[2] = (uint64_t)(m.rng_inc >> 64); final_state[3] = (uint64_t)m.rng_inc;
    final_state[4] = m.has_u32; final_state[5] = m.uinteger;
}

// ==========================================================================================
// host side: the C-ABI
// ==========================================================================================
struct cda_env {
    Params P;
    int device;
    int cap;                 // book capacity of this env: 256 or 512 (selects the cap256:: / cap512:: kernels)
    uint8_t* arena;
    size_t arena_bytes;
    uint8_t* done_buf;       // auto_reset: u8[N] behind the market records - terminated | truncated of the last step
    uint8_t* handback;       // cda_set_handback: caller-owned [N, handback_stride(A)] bytes, or NULL
    int64_t hb_row_stride, hb_rows_total;   // cda_set_handback_geometry: rows between two ranks' first markets / global row count (0, 0 = equal shards of N)
    int32_t spill_wanted;    // orders per side the spill ring was asked to hold (automatic: num_agents * max_step rounded up); P.lay.spill_cap is what it got
    double* em_partials;     // cda_episode_metrics_collect: the blocks' partial sums (EM_BLOCKS rows)
};
static inline int32_t handback_stride_of(int32_t num_agents) { return (CDA_SNAPSHOT_DIM * 4 + num_agents * 8 + 3 + 7) & ~7; }

static thread_local char g_err[256] = "";
static int g_dbg_skip = 0;                          // debug (CDA_DEBUG_SKIP builds)
static unsigned long long* g_phase_cycles = NULL;   // debug (CDA_PHASE_TIMING builds): device buffer [N,40]
static int hip_fail(hipError_t e, const char* what) {
    snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    return CDA_ERR_HIP;
}
#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) return hip_fail(_e, #x); } while (0)

extern "C" {

const char* cda_strerror(int status) {
    switch (status) {
        case CDA_OK: return "ok";
        case CDA_ERR_INVALID: return "invalid argument or config outside the supported domain";
        case CDA_ERR_NO_DEVICE: return "no HIP device: this library has no CPU fallback";
        case CDA_ERR_HIP: return g_err[0] ? g_err : "HIP runtime error";
        case CDA_ERR_UNSUPPORTED: return "unsupported configuration (tick_size must be an integer in 1 .. 65536) or a launch this env does not qualify for";
        case CDA_ERR_NOMEM: return "out of memory";
        default: return "unknown status";
    }
}

int cda_default_config(cda_config* c) {
    if (!c) return CDA_ERR_INVALID;
    memset(c, 0, sizeof *c);
    c->book_capacity = 0;
    c->num_agents = 5; c->max_step = 64; c->n_hist = 4; c->tick_size = 1; c->init_cash = 1000000;
    c->initial_price_min = 10; c->initial_price_max = 100; c->min_size = 1; c->mkt_max_size = 100; c->limit_size_multiple = 10;
    c->order_penalty = 0.1; c->trade_penalty = 0.05; c->drawdown_penalty = 0.2; c->passive_bonus = 0.1; c->loss_multiplier = 1.5;
    return CDA_OK;
}

static int cfg_ok(const cda_config* c) {
    if (c->num_agents < 1 || c->num_agents > CDA_MAX_AGENTS) return CDA_ERR_INVALID;
    if (c->n_hist < 1 || c->n_hist > CDA_MAX_HIST) return CDA_ERR_INVALID;
    if (c->tick_size < 1 || c->tick_size > CDA_TICK_MAX) return CDA_ERR_UNSUPPORTED;
    if (c->initial_price_max < c->initial_price_min || c->initial_price_min < 0) return CDA_ERR_INVALID;
    if (c->initial_price_max >= (1 << 24)) return CDA_ERR_INVALID;      // prices live below 2^24 ticks (the clamp of step_market; the libm sweeps cover exactly that domain)
    if (c->min_size < 0 || c->mkt_max_size < c->min_size || c->limit_size_multiple < 1) return CDA_ERR_INVALID;
    // Sizes are int32 on the device and a price level sums up to CDA_BOOK_CAP_MAX of them: the largest decodable size is
    // (mkt_max_size * limit_size_multiple - min_size) / 2 * |mean| + sigma * z + min_size, clamped at 1e9 (flagged).  Keep
    // the configured scale itself well inside int32 so that neither the product nor a level sum can wrap silently.
    if ((int64_t)c->mkt_max_size * (int64_t)c->limit_size_multiple + (int64_t)c->min_size > (int64_t)(1 << 30) / CDA_BOOK_CAP_MAX) return CDA_ERR_INVALID;
    if (c->book_spill < -1 || c->book_spill > CDA_SPILL_MAX) return CDA_ERR_INVALID;
    if (c->max_step < 1) return CDA_ERR_INVALID;
    if (c->book_capacity != 0 && c->book_capacity != 256 && c->book_capacity != 512) return CDA_ERR_INVALID;
    if (c->init_cash > (1LL << 62) || c->init_cash < -(1LL << 62)) return CDA_ERR_INVALID;
    if (c->auto_reset != 0 && c->auto_reset != 1) return CDA_ERR_INVALID;
    return CDA_OK;
}

// launch the capacity variant of a market-wave kernel that matches the env
#define LAUNCH_CAP(e, kern, grid, block, smem, stream, ...) do { \
    if ((e)->cap == 512) hipLaunchKernelGGL(cda::cap512::kern, grid, block, smem, stream, __VA_ARGS__); \
    else hipLaunchKernelGGL(cda::cap256::kern, grid, block, smem, stream, __VA_ARGS__); } while (0)
static dim3 grid_for(int n) { return dim3((unsigned)((n + CDA_WPB - 1) / CDA_WPB)); }
static size_t smem_for(const cda_env* e, int waves) {
    const int per = e->cap == 512 ? cda::cap512::lds_bytes_per_wave(e->P.cfg.num_agents, e->P.cfg.n_hist) : cda::cap256::lds_bytes_per_wave(e->P.cfg.num_agents, e->P.cfg.n_hist);
    return (size_t)DEC_TABLE_BYTES + (size_t)waves * (size_t)per;
}

static int grant_policy_step_lds(const cda_env* e);
int cda_create(const cda_config* cfg, int32_t n_markets, int32_t device, cda_env** out) {
    if (!cfg || !out || n_markets < 1) return CDA_ERR_INVALID;
    int rc = cfg_ok(cfg); if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CDA_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return CDA_ERR_INVALID;
    HIPCHK(hipSetDevice(device));
    cda_env* e = (cda_env*)calloc(1, sizeof *e);
    if (!e) return CDA_ERR_NOMEM;
    e->device = device;
    // book capacity: as asked, else by the agent count (census, profiles/r02: random agents hold up to 83 / 129 / 235 resting
    // orders per market at 4 / 8 / 16 agents over 4096 steps; the flip-heavy law overflows 256 at 16 agents)
    e->cap = cfg->book_capacity ? cfg->book_capacity : (cfg->num_agents <= 8 ? 256 : 512);
    Params& P = e->P;
    P.cfg = *cfg; P.n_markets = n_markets;
    P.mkt_mul = (float)((double)(cfg->mkt_max_size - cfg->min_size) / 2.0);
    P.lim_mul = (float)((double)((int64_t)cfg->mkt_max_size * (int64_t)cfg->limit_size_multiple - (int64_t)cfg->min_size) / 2.0);
    P.lay = record_layout(cfg->num_agents, cfg->n_hist, e->cap);
    // HBM tier of the book: a ring of spill_cap orders per side behind every market's tile.  Automatic size: a side gains at
    // most one resting order per agent and step, so num_agents * max_step (+ the tile) can never be exceeded inside an episode -
    // the unbounded OrderTree of the reference (ordertree.py:5-58), priced in HBM: 32 B per order slot and market (hipMalloc'd
    // memory is physically backed whether a market ever touches it or not).  The automatic size is halved while the rings would
    // take more than a quarter of the device's free memory - down to CDA_SPILL_MIN if it must: an env that fitted without the tier
    // still fits with it.  cda_book_spill() reports what was granted, cda_book_spill_wanted() what "unbounded inside an episode"
    // needs: a caller learns at construction when the two differ (the Python env warns).
    P.lay.spill_cap = 0;
    e->spill_wanted = 0;
    if (cfg->book_spill >= 0) {
        int64_t want = cfg->book_spill > 0 ? (int64_t)cfg->book_spill : (int64_t)cfg->num_agents * (int64_t)cfg->max_step;
        if (cfg->book_spill == 0 && want < 1024) want = 1024;
        int64_t capv = CDA_SPILL_MIN;
        while (capv < want && capv < CDA_SPILL_MAX) capv <<= 1;
        e->spill_wanted = (int32_t)capv;
        size_t free_b = 0, total_b = 0;
        if (cfg->book_spill == 0 && hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            while (capv > CDA_SPILL_MIN && spill_region_bytes((int32_t)capv) * (size_t)n_markets > free_b / 4) capv >>= 1;
        // a price level's size is summed in int32 (the observation's raw snapshot): the largest decodable order size times the
        // orders one side can hold must stay below 2^31 - the same rule cfg_ok applies to the tile alone
        const int64_t scale = (int64_t)cfg->mkt_max_size * (int64_t)cfg->limit_size_multiple + (int64_t)cfg->min_size;
        while (capv > CDA_SPILL_MIN && scale * ((int64_t)e->cap + capv) > 0x7fffffffLL) {
            if (cfg->book_spill > 0) { free(e); return CDA_ERR_INVALID; }
            capv >>= 1;
        }
        P.lay.spill_cap = (int32_t)capv;
    }
    const size_t records = (size_t)P.lay.stride * (size_t)n_markets;
    e->arena_bytes = spill_arena_off(P) + spill_region_bytes(P.lay.spill_cap) * (size_t)n_markets;
    hipError_t he = hipMalloc((void**)&e->arena, e->arena_bytes);
    if (he != hipSuccess) { free(e); return he == hipErrorOutOfMemory ? CDA_ERR_NOMEM : hip_fail(he, "hipMalloc"); }
    e->done_buf = e->arena + records;
    P.ep_tol = 1e-6;
    hipLaunchKernelGGL(k_init_arena, dim3((unsigned)((n_markets + 255) / 256)), dim3(256), 0, 0, e->arena, P);
    he = hipMemsetAsync(e->arena + em_agent_off(P), 0, spill_arena_off(P) - em_agent_off(P), 0);          // the episode-metric accumulators
    if (he == hipSuccess) he = hipMalloc((void**)&e->em_partials, (size_t)EM_BLOCKS * (CDA_EM_MAX_MODULES * CDA_EM_AGENT_FIELDS + CDA_EM_ENV_FIELDS) * sizeof(double));
    if (he == hipSuccess) he = hipDeviceSynchronize();
    if (he != hipSuccess) { (void)hipFree(e->arena); if (e->em_partials) (void)hipFree(e->em_partials); free(e); return hip_fail(he, "k_init_arena"); }
    if (cda_policy_step_supported(e)) (void)grant_policy_step_lds(e);          // (a refusal resurfaces as CDA_ERR_HIP at the first cda_policy_step_range)
    *out = e;
    return CDA_OK;
}

int cda_destroy(cda_env* e) {
    if (!e) return CDA_OK;
    (void)hipSetDevice(e->device);
    (void)hipFree(e->arena);
    if (e->em_partials) (void)hipFree(e->em_partials);
    free(e);
    return CDA_OK;
}

static int range_ok(const cda_env* e, int32_t first, int32_t n) { return first >= 0 && n >= 1 && (int64_t)first + (int64_t)n <= (int64_t)e->P.n_markets; }

int cda_reset_range(cda_env* e, int32_t first_market, int32_t n_markets, const uint64_t* seeds, const uint8_t* mask, float* obs_out, void* stream) {
    if (!e || !range_ok(e, first_market, n_markets)) return CDA_ERR_INVALID;
    HIPCHK(hipSetDevice(e->device));
    LAUNCH_CAP(e, k_reset, grid_for(n_markets), dim3(64 * CDA_WPB), smem_for(e, CDA_WPB), (hipStream_t)stream, e->arena, e->P, seeds, mask, obs_out,
                       (int)first_market, (int)(first_market + n_markets), e->handback, handback_stride_of(e->P.cfg.num_agents), 0, FinCapture{NULL, NULL, NULL, 0});
    HIPCHK(hipGetLastError());
    return CDA_OK;
}
int cda_reset(cda_env* e, const uint64_t* seeds, const uint8_t* mask, float* obs_out, void* stream) {
    if (!e) return CDA_ERR_INVALID;
    return cda_reset_range(e, 0, e->P.n_markets, seeds, mask, obs_out, stream);
}

// one launch of k_step (+ the auto-reset pass) over [first, first + n) on `stream`; arguments validated by the callers
static int launch_step(cda_env* e, int32_t first, int32_t n, const StepArgs& S0, hipStream_t stream) {
    const size_t smem = smem_for(e, CDA_WPB) + ZIG_LDS_BYTES;
    const bool tally = e->P.lay.ep_on != 0;               // episode metrics on: the instances that tally (their code is not even in the others)
#define CDA_LAUNCH_STEP(ns) do { \
        cda::ns::StepKernArgs K; K.arena = e->arena; K.P = e->P; K.S = S0; K.S.first_market = first; K.S.end_market = first + n; \
        if (K.S.has_info) { if (tally) hipLaunchKernelGGL((cda::ns::k_step<true, true>), grid_for(n), dim3(64 * CDA_WPB), smem, stream, K); \
                            else hipLaunchKernelGGL((cda::ns::k_step<true, false>), grid_for(n), dim3(64 * CDA_WPB), smem, stream, K); } \
        else { if (tally) hipLaunchKernelGGL((cda::ns::k_step<false, true>), grid_for(n), dim3(64 * CDA_WPB), smem, stream, K); \
               else hipLaunchKernelGGL((cda::ns::k_step<false, false>), grid_for(n), dim3(64 * CDA_WPB), smem, stream, K); } } while (0)
    if (e->cap == 512) CDA_LAUNCH_STEP(cap512); else CDA_LAUNCH_STEP(cap256);
#undef CDA_LAUNCH_STEP
    HIPCHK(hipGetLastError());
    // auto_reset: in the info-less kernel a market whose episode ended resets itself as the kernel's last act (reset_after_step in
    // cda_kernels.inc); with info tensors the pass is a launch of its own behind the step (every market-wave of it exits at once unless its
    // episode just ended)
    if (e->P.cfg.auto_reset && S0.has_info) {
        LAUNCH_CAP(e, k_reset, grid_for(n), dim3(64 * CDA_WPB), smem_for(e, CDA_WPB), stream, e->arena, e->P,
                           (const uint64_t*)NULL, (const uint8_t*)e->done_buf, S0.obs_out, (int)first, (int)(first + n),
                           e->handback, handback_stride_of(e->P.cfg.num_agents), 1, FinCapture{S0.fin_obs, S0.fin_count, S0.fin_index_out, S0.fin_cap});
        HIPCHK(hipGetLastError());
    }
    return CDA_OK;
}
static int fill_step_args(cda_env* e, StepArgs& S, const int32_t* category, const float* size_mean, const float* size_sigma,
                          const int32_t* price, const int32_t* price_offset, const uint8_t* present,
                          float* obs_out, double* reward_out, uint8_t* terminated_out, uint8_t* truncated_out, const cda_info_ptrs* info_out) {
    if (!e || !category || !size_mean || !size_sigma || !price || !price_offset) return CDA_ERR_INVALID;
    if (!obs_out || !reward_out || !terminated_out || !truncated_out) return CDA_ERR_INVALID;
    S.category = category; S.size_mean = size_mean; S.size_sigma = size_sigma; S.price = price; S.price_offset = price_offset;
    S.present = present; S.obs_out = obs_out; S.reward_out = reward_out; S.terminated_out = terminated_out; S.truncated_out = truncated_out;
    if (info_out) { S.info = *info_out; S.has_info = 1; } else { memset(&S.info, 0, sizeof S.info); S.has_info = 0; }
    S.phase_cycles = g_phase_cycles;
    S.dbg_skip = g_dbg_skip;
    S.done_out = e->P.cfg.auto_reset ? e->done_buf : NULL;
    S.handback = e->handback; S.handback_stride = handback_stride_of(e->P.cfg.num_agents);
    S.first_market = 0; S.end_market = e->P.n_markets;
    S.fin_obs = NULL; S.fin_count = NULL; S.fin_index_out = NULL; S.fin_cap = 0;
    return CDA_OK;
}

int cda_step_range(cda_env* e, int32_t first_market, int32_t n_markets,
                   const int32_t* category, const float* size_mean, const float* size_sigma,
                   const int32_t* price, const int32_t* price_offset, const uint8_t* present,
                   float* obs_out, double* reward_out, uint8_t* terminated_out, uint8_t* truncated_out,
                   const cda_info_ptrs* info_out, void* stream) {
    StepArgs S;
    int rc = fill_step_args(e, S, category, size_mean, size_sigma, price, price_offset, present, obs_out, reward_out, terminated_out, truncated_out, info_out);
    if (rc) return rc;
    if (!range_ok(e, first_market, n_markets)) return CDA_ERR_INVALID;
    HIPCHK(hipSetDevice(e->device));
    return launch_step(e, first_market, n_markets, S, (hipStream_t)stream);
}

int cda_step_range_capture(cda_env* e, int32_t first_market, int32_t n_markets,
                           const int32_t* category, const float* size_mean, const float* size_sigma,
                           const int32_t* price, const int32_t* price_offset, const uint8_t* present,
                           float* obs_out, double* reward_out, uint8_t* terminated_out, uint8_t* truncated_out,
                           const cda_info_ptrs* info_out, float* fin_obs, int32_t fin_cap, int32_t* fin_count, int32_t* fin_index_out, void* stream) {
    StepArgs S;
    int rc = fill_step_args(e, S, category, size_mean, size_sigma, price, price_offset, present, obs_out, reward_out, terminated_out, truncated_out, info_out);
    if (rc) return rc;
    if (!range_ok(e, first_market, n_markets)) return CDA_ERR_INVALID;
    if (fin_index_out) {
        if (!e->P.cfg.auto_reset || !fin_obs || !fin_count || fin_cap < 1) return CDA_ERR_INVALID;
        S.fin_obs = fin_obs; S.fin_count = fin_count; S.fin_index_out = fin_index_out; S.fin_cap = fin_cap;
    }
    HIPCHK(hipSetDevice(e->device));
    return launch_step(e, first_market, n_markets, S, (hipStream_t)stream);
}

// The policy inside the step kernel (k_policy_step, cda_kernels.inc): what include/cda_mlp.h's rollout chains launch per step when the env qualifies.
static int policy_step_lds(const cda_env* e) {            // dynamic LDS of k_policy_step at this env's shape; 0 = no instance for its history depth
    const cda_config& c = e->P.cfg;
    switch (c.n_hist) {
        case 4: return cda::cap256::policy_step_lds_bytes(c.num_agents, c.n_hist);
        case 1: return cda::cap256::policy_step_lds_bytes_h1(c.num_agents, c.n_hist);
        case 2: return cda::cap256::policy_step_lds_bytes_h2(c.num_agents, c.n_hist);
        case 3: return cda::cap256::policy_step_lds_bytes_h3(c.num_agents, c.n_hist);
        case 6: return cda::cap256::policy_step_lds_bytes_h6(c.num_agents, c.n_hist);
        case 7: return cda::cap256::policy_step_lds_bytes_h7(c.num_agents, c.n_hist);
        case 8: return cda::cap256::policy_step_lds_bytes_h8(c.num_agents, c.n_hist);
        default: return 0;
    }
}
// the instance of k_policy_step for a history depth (NULL: none compiled), with / without the episode-metric tallies
typedef void (*policy_step_kern_t)(cda::cap256::PolicyStepKernArgs);
static policy_step_kern_t policy_step_kernel(int hist, bool tally) {
    switch (hist) {
        case 4: return tally ? cda::cap256::k_policy_step<true> : cda::cap256::k_policy_step<false>;
#define CDA_PS_CASE(h) case h: return tally ? cda::cap256::k_policy_step_h##h<true> : cda::cap256::k_policy_step_h##h<false>;
        CDA_PS_CASE(1) CDA_PS_CASE(2) CDA_PS_CASE(3) CDA_PS_CASE(6) CDA_PS_CASE(7) CDA_PS_CASE(8)
#undef CDA_PS_CASE
        default: return nullptr;
    }
}
// More than 64 KB of dynamic LDS has to be granted per device and kernel instance: once, under a lock (envs are created and stepped from several host threads - one per
// GPU in the guarded collectives of parallel.py), at cda_create - not lazily inside the launch path, whose first call sits inside a stream capture (round-5 ADVICE).
static int grant_policy_step_lds(const cda_env* e) {
    static std::mutex mu;
    static unsigned long long granted[CDA_MAX_HIST + 1] = {0};      // per history-depth instance: a bit per device (both tally variants together)
    const int hist = e->P.cfg.n_hist, slot = hist;
    std::lock_guard<std::mutex> lock(mu);
    if (granted[slot] >> (e->device & 63) & 1ull) return 0;
    const policy_step_kern_t ks[2] = {policy_step_kernel(hist, false), policy_step_kernel(hist, true)};
    if (!ks[0] || !ks[1]) return 1;
    for (int k = 0; k < 2; k++)
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(ks[k]), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 1;
    granted[slot] |= 1ull << (e->device & 63);
    return 0;
}
int cda_policy_step_supported(const cda_env* e) {
    if (!e) return 0;
    const int lds = e->cap == 256 ? policy_step_lds(e) : 0;
    return lds > 0 && lds <= 160 * 1024 && e->P.cfg.num_agents <= 8 && !e->handback;
}
// ... and whether it pays: every workgroup of k_policy_step streams the whole network for its sixteen rows, so the one launch wins while all of the env's
// workgroups are resident at once (N <= 16 x CUs: 4096 markets on an MI355X - policy in the loop 278 -> 301-308 M at 4096 x 4) and loses to the batched policy
// kernel beyond (16 384 x 4: 433 -> 408 M); below a quarter of that it is a wash (profiles/r05/policy_in_the_step_kernel.txt, C)
int cda_policy_step_advised(const cda_env* e) {
    if (!cda_policy_step_supported(e)) return 0;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device) != hipSuccess || cus < 1) return 0;
    return (long long)e->P.n_markets <= 16LL * cus;
}
int cda_policy_step_range(cda_env* e, int32_t first_market, int32_t n_markets, const void* wb, const float* theta, const float* obs_in,
                          uint64_t seed, const int64_t* counter_dev, int64_t draw,
                          int32_t* category, float* size_mean, float* size_sigma, int32_t* price, int32_t* price_offset,
                          float* a_cont, float* logp, float* value, float* rec, float* dist,
                          float* obs_out, double* reward_out, uint8_t* terminated_out, uint8_t* truncated_out,
                          float* fin_obs, int32_t fin_cap, int32_t* fin_count, int32_t* fin_index_out, void* stream) {
    if (!e || !wb || !theta || !obs_in || !counter_dev || !a_cont || !logp || !value) return CDA_ERR_INVALID;
    if (!cda_policy_step_supported(e)) return CDA_ERR_UNSUPPORTED;
    StepArgs S;
    int rc = fill_step_args(e, S, category, size_mean, size_sigma, price, price_offset, NULL, obs_out, reward_out, terminated_out, truncated_out, NULL);
    if (rc) return rc;
    if (!range_ok(e, first_market, n_markets)) return CDA_ERR_INVALID;
    if (fin_index_out) {
        if (!e->P.cfg.auto_reset || !fin_obs || !fin_count || fin_cap < 1) return CDA_ERR_INVALID;
        S.fin_obs = fin_obs; S.fin_count = fin_count; S.fin_index_out = fin_index_out; S.fin_cap = fin_cap;
    }
    HIPCHK(hipSetDevice(e->device));
    const size_t smem = (size_t)policy_step_lds(e);
    const policy_step_kern_t kern = policy_step_kernel(e->P.cfg.n_hist, e->P.lay.ep_on != 0);
    if (!kern) return CDA_ERR_UNSUPPORTED;
    if (grant_policy_step_lds(e)) return CDA_ERR_HIP;     // (normally a no-op: cda_create granted it - outside any stream capture)
    cda::cap256::PolicyStepKernArgs KA;
    KA.K.arena = e->arena; KA.K.P = e->P; KA.K.S = S; KA.K.S.first_market = first_market; KA.K.S.end_market = first_market + n_markets;
    KA.F.obs_in = obs_in; KA.F.wb = wb; KA.F.theta = theta; KA.F.seed = seed; KA.F.counter = (const long long*)counter_dev; KA.F.draw = draw;
    KA.F.category = category; KA.F.size_mean = size_mean; KA.F.size_sigma = size_sigma; KA.F.price = price; KA.F.price_offset = price_offset;
    KA.F.a_cont = a_cont; KA.F.logp = logp; KA.F.value = value; KA.F.rec = rec; KA.F.dist = dist;
    hipLaunchKernelGGL(kern, dim3((unsigned)((n_markets + cda::cap256::PS_WPB - 1) / cda::cap256::PS_WPB)), dim3(64 * cda::cap256::PS_WPB), smem, (hipStream_t)stream, KA);
    HIPCHK(hipGetLastError());
    return CDA_OK;
}

int cda_step(cda_env* e, const int32_t* category, const float* size_mean, const float* size_sigma,
             const int32_t* price, const int32_t* price_offset, const uint8_t* present,
             float* obs_out, double* reward_out, uint8_t* terminated_out, uint8_t* truncated_out,
             const cda_info_ptrs* info_out, void* stream) {
    if (!e) return CDA_ERR_INVALID;
    return cda_step_range(e, 0, e->P.n_markets, category, size_mean, size_sigma, price, price_offset, present,
                          obs_out, reward_out, terminated_out, truncated_out, info_out, stream);
}

void cda_group_range(int32_t n_markets, int32_t n_groups, int32_t group, int32_t* first_out, int32_t* count_out) {
    if (n_groups < 1 || group < 0 || group >= n_groups || n_markets < 0) { if (first_out) *first_out = 0; if (count_out) *count_out = 0; return; }
    const int64_t lo = (int64_t)n_markets * group / n_groups, hi = (int64_t)n_markets * (group + 1) / n_groups;
    if (first_out) *first_out = (int32_t)lo;
    if (count_out) *count_out = (int32_t)(hi - lo);
}

int cda_step_groups(cda_env* e, int32_t n_groups,
                    const int32_t* category, const float* size_mean, const float* size_sigma,
                    const int32_t* price, const int32_t* price_offset, const uint8_t* present,
                    float* obs_out, double* reward_out, uint8_t* terminated_out, uint8_t* truncated_out,
                    const cda_info_ptrs* info_out, void* const* streams) {
    StepArgs S;
    int rc = fill_step_args(e, S, category, size_mean, size_sigma, price, price_offset, present, obs_out, reward_out, terminated_out, truncated_out, info_out);
    if (rc) return rc;
    if (!streams || n_groups < 1 || n_groups > e->P.n_markets || n_groups > CDA_MAX_GROUPS) return CDA_ERR_INVALID;
    HIPCHK(hipSetDevice(e->device));
    for (int32_t g = 0; g < n_groups; g++) {
        int32_t first, n;
        cda_group_range(e->P.n_markets, n_groups, g, &first, &n);
        rc = launch_step(e, first, n, S, (hipStream_t)streams[g]);
        if (rc) return rc;
    }
    return CDA_OK;
}

// ---- the hand-back of every chain, natively: RCCL is called directly (resolved from the copy already loaded in the process) ----
typedef int (*nccl_allgather_fn)(const void* sendbuff, void* recvbuff, size_t sendcount, int datatype, void* comm, hipStream_t stream);
static nccl_allgather_fn resolve_allgather() {
    static nccl_allgather_fn fn = NULL;
    static int tried = 0;
    if (!tried) { tried = 1; fn = (nccl_allgather_fn)dlsym(RTLD_DEFAULT, "ncclAllGather"); }
    return fn;
}
static int handback_chain(cda_env* e, int32_t first, int32_t count, hipStream_t stream, void* comm, int32_t world, void* gathered,
                          float* obs_full, double* reward_full, uint8_t* term_full, uint8_t* trunc_full) {
    const int32_t A = e->P.cfg.num_agents, H = e->P.cfg.n_hist, stride = handback_stride_of(A);
    const uint8_t* mine = e->handback + (size_t)first * (size_t)stride;
    const uint8_t* src = mine;
    if (world > 1 || comm) {                                  // (a caller may hand a one-rank communicator over to exercise the collective)
        nccl_allgather_fn ag = resolve_allgather();
        if (!ag || !comm || !gathered) { snprintf(g_err, sizeof g_err, "ncclAllGather is not available in this process (RCCL not loaded) or no communicator was given"); return CDA_ERR_UNSUPPORTED; }
        const int rc = ag(mine, gathered, (size_t)count * (size_t)stride, /* ncclUint8 */ 1, comm, stream);
        if (rc != 0) { snprintf(g_err, sizeof g_err, "ncclAllGather failed with ncclResult %d", rc); return CDA_ERR_HIP; }
        src = (const uint8_t*)gathered;
    }
    hipLaunchKernelGGL(k_handback_unpack, dim3((unsigned)(((size_t)world * count + 3) / 4)), dim3(256), 0, stream, src, (int)world, (int)count,
                       (long long)(e->hb_row_stride > 0 ? e->hb_row_stride : e->P.n_markets), (long long)first, (int)A, (int)H, (int)stride, (long long)e->hb_rows_total,
                       obs_full, reward_full, term_full, trunc_full);
    HIPCHK(hipGetLastError());
    return CDA_OK;
}
int cda_handback_groups(cda_env* e, int32_t n_groups, void* const* streams, void* const* comms, int32_t world, void* const* gathered,
                        float* obs_full, double* reward_full, uint8_t* terminated_full, uint8_t* truncated_full) {
    if (!e || !e->handback || !streams || n_groups < 1 || n_groups > e->P.n_markets || n_groups > CDA_MAX_GROUPS || world < 1) return CDA_ERR_INVALID;
    if (!obs_full || !reward_full || !terminated_full || !truncated_full || (world > 1 && (!comms || !gathered))) return CDA_ERR_INVALID;
    HIPCHK(hipSetDevice(e->device));
    for (int32_t g = 0; g < n_groups; g++) {
        int32_t first, n;
        cda_group_range(e->P.n_markets, n_groups, g, &first, &n);
        int rc = handback_chain(e, first, n, (hipStream_t)streams[g], comms ? comms[g] : NULL, world, gathered ? gathered[g] : NULL,
                                obs_full, reward_full, terminated_full, truncated_full);
        if (rc) return rc;
    }
    return CDA_OK;
}
int cda_step_groups_handback(cda_env* e, int32_t n_groups,
                             const int32_t* category, const float* size_mean, const float* size_sigma,
                             const int32_t* price, const int32_t* price_offset, const uint8_t* present,
                             float* obs_out, double* reward_out, uint8_t* terminated_out, uint8_t* truncated_out,
                             const cda_info_ptrs* info_out, void* const* streams,
                             void* const* comms, int32_t world, void* const* gathered,
                             float* obs_full, double* reward_full, uint8_t* terminated_full, uint8_t* truncated_full) {
    StepArgs S;
    int rc = fill_step_args(e, S, category, size_mean, size_sigma, price, price_offset, present, obs_out, reward_out, terminated_out, truncated_out, info_out);
    if (rc) return rc;
    if (!e->handback || !streams || n_groups < 1 || n_groups > e->P.n_markets || n_groups > CDA_MAX_GROUPS || world < 1) return CDA_ERR_INVALID;
    if (!obs_full || !reward_full || !terminated_full || !truncated_full || (world > 1 && (!comms || !gathered))) return CDA_ERR_INVALID;
    HIPCHK(hipSetDevice(e->device));
    for (int32_t g = 0; g < n_groups; g++) {                 // every chain: its step, then ITS collective and rebuild, all on its own stream
        int32_t first, n;
        cda_group_range(e->P.n_markets, n_groups, g, &first, &n);
        rc = launch_step(e, first, n, S, (hipStream_t)streams[g]);
        if (rc) return rc;
        rc = handback_chain(e, first, n, (hipStream_t)streams[g], comms ? comms[g] : NULL, world, gathered ? gathered[g] : NULL,
                            obs_full, reward_full, terminated_full, truncated_full);
        if (rc) return rc;
    }
    return CDA_OK;
}

int cda_run_random(cda_env* e, int32_t n_steps, uint64_t action_seed, uint64_t market_index_base,
                   float* obs_out, double* episode_return_out, uint8_t* terminated_out, uint8_t* truncated_out,
                   int32_t* steps_taken_out, void* stream) {
    if (!e || n_steps < 0) return CDA_ERR_INVALID;
    HIPCHK(hipSetDevice(e->device));
    RunArgs R;
    R.n_steps = n_steps; R.seed = action_seed; R.market_base = market_index_base;
    R.obs_out = obs_out; R.return_out = episode_return_out; R.terminated_out = terminated_out; R.truncated_out = truncated_out;
    R.steps_out = steps_taken_out;
#define CDA_LAUNCH_RUN(ns) do { cda::ns::RunKernArgs K; K.arena = e->arena; K.P = e->P; K.R = R; \
        if (e->P.lay.ep_on) hipLaunchKernelGGL(cda::ns::k_run_random<true>, grid_for(e->P.n_markets), dim3(64 * CDA_WPB), smem_for(e, CDA_WPB) + ZIG_LDS_BYTES, (hipStream_t)stream, K); \
        else hipLaunchKernelGGL(cda::ns::k_run_random<false>, grid_for(e->P.n_markets), dim3(64 * CDA_WPB), smem_for(e, CDA_WPB) + ZIG_LDS_BYTES, (hipStream_t)stream, K); } while (0)
    if (e->cap == 512) CDA_LAUNCH_RUN(cap512); else CDA_LAUNCH_RUN(cap256);
#undef CDA_LAUNCH_RUN
    HIPCHK(hipGetLastError());
    return CDA_OK;
}

int cda_random_actions_host(uint64_t action_seed, uint64_t market_index_base, int32_t step, int32_t n_markets, int32_t num_agents,
                            int32_t* category, float* size_mean, float* size_sigma, int32_t* price, int32_t* price_offset) {
    if (step < 0 || n_markets < 0 || num_agents < 1 || !category || !size_mean || !size_sigma || !price || !price_offset) return CDA_ERR_INVALID;
    for (int32_t i = 0; i < n_markets; i++)
        for (int32_t a = 0; a < num_agents; a++) {
            const size_t ix = (size_t)i * (size_t)num_agents + (size_t)a;
            cda_random_action(action_seed, market_index_base + (uint64_t)i, (uint32_t)step, (uint32_t)a, &category[ix], &size_mean[ix], &size_sigma[ix],
                              &price[ix], &price_offset[ix]);
        }
    return CDA_OK;
}

int cda_random_actions(uint64_t action_seed, uint64_t market_index_base, int32_t step0, int32_t n_steps, int32_t n_markets, int32_t num_agents,
                       int32_t* category, float* size_mean, float* size_sigma, int32_t* price, int32_t* price_offset, void* stream) {
    if (step0 < 0 || n_steps < 0 || n_markets < 0 || num_agents < 1 || !category || !size_mean || !size_sigma || !price || !price_offset) return CDA_ERR_INVALID;
    const size_t total = (size_t)n_steps * (size_t)n_markets * (size_t)num_agents;
    if (total == 0) return CDA_OK;
    {   // the launch goes to the device that owns the output arrays (this entry point has no env to ask)
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, category) == hipSuccess) HIPCHK(hipSetDevice(at.device));
    }
    const size_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(k_random_actions, dim3((unsigned)(blocks > 65535 ? 65535 : blocks)), dim3(256), 0, (hipStream_t)stream, action_seed, market_index_base,
                       (int)step0, (int)n_steps, (int)n_markets, (int)num_agents, category, size_mean, size_sigma, price, price_offset);
    HIPCHK(hipGetLastError());
    return CDA_OK;
}

int32_t cda_handback_stride(int32_t num_agents) { return num_agents >= 1 && num_agents <= CDA_MAX_AGENTS ? handback_stride_of(num_agents) : 0; }
int cda_set_handback(cda_env* e, void* records_dev) {
    if (!e) return CDA_ERR_INVALID;
    e->handback = (uint8_t*)records_dev;
    return CDA_OK;
}
int cda_set_handback_geometry(cda_env* e, int64_t shard_row_stride, int64_t n_rows_total) {
    if (!e || shard_row_stride < 0 || n_rows_total < 0 || (n_rows_total > 0 && shard_row_stride < 1)) return CDA_ERR_INVALID;
    e->hb_row_stride = shard_row_stride; e->hb_rows_total = n_rows_total;
    return CDA_OK;
}
int cda_handback_unpack(const void* records_dev, int32_t n_segments, int32_t seg_records, int64_t seg_row_stride, int64_t row0,
                        int32_t num_agents, int32_t n_hist, int64_t n_rows_total,
                        float* obs_full, double* reward_full, uint8_t* terminated_full, uint8_t* truncated_full, void* stream) {
    if (!records_dev || n_segments < 0 || seg_records < 0 || seg_row_stride < 0 || row0 < 0 || n_rows_total < 0) return CDA_ERR_INVALID;
    if (num_agents < 1 || num_agents > CDA_MAX_AGENTS || n_hist < 1 || n_hist > CDA_MAX_HIST) return CDA_ERR_INVALID;
    if (!obs_full || !reward_full || !terminated_full || !truncated_full) return CDA_ERR_INVALID;
    const int64_t count = (int64_t)n_segments * seg_records;
    if (count == 0) return CDA_OK;
    if (count > (int64_t)1 << 30) return CDA_ERR_INVALID;
    int dev = 0;                                          // the launch goes to the device that owns the destination (a C caller may sit on another one)
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, obs_full) == hipSuccess) dev = at.device;
    HIPCHK(hipSetDevice(dev));
    hipLaunchKernelGGL(k_handback_unpack, dim3((unsigned)((count + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)records_dev, (int)n_segments,
                       (int)seg_records, (long long)seg_row_stride, (long long)row0, (int)num_agents, (int)n_hist, (int)handback_stride_of(num_agents),
                       (long long)n_rows_total, obs_full, reward_full, terminated_full, truncated_full);
    HIPCHK(hipGetLastError());
    return CDA_OK;
}

int cda_book_peak(cda_env* e, int32_t* peak_out, void* stream) {
    if (!e || !peak_out) return CDA_ERR_INVALID;
    HIPCHK(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_book_peak, dim3((unsigned)((e->P.n_markets + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)e->arena, e->P, peak_out);
    HIPCHK(hipGetLastError());
    return CDA_OK;
}

int cda_nav_conservation(cda_env* e, double tolerance, double* abs_error_out, uint8_t* violated_out, void* stream) {
    if (!e || !abs_error_out) return CDA_ERR_INVALID;
    HIPCHK(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_nav_conservation, dim3((unsigned)((e->P.n_markets + 63) / 64)), dim3(64), DEC_TABLE_BYTES, (hipStream_t)stream,
                       (const uint8_t*)e->arena, e->P, tolerance, abs_error_out, violated_out);
    HIPCHK(hipGetLastError());
    return CDA_OK;
}

int cda_episode_metrics_enable(cda_env* e, int32_t on, double nav_tolerance) {
    if (!e || (on != 0 && on != 1) || !(nav_tolerance >= 0.0)) return CDA_ERR_INVALID;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipDeviceSynchronize());
    e->P.lay.ep_on = on;
    e->P.ep_tol = nav_tolerance;
    hipLaunchKernelGGL(k_ep_enable, dim3((unsigned)((e->P.n_markets + 255) / 256)), dim3(256), 0, 0, e->arena, e->P, (int)on);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    return CDA_OK;
}
int cda_episode_metrics_collect(cda_env* e, const int32_t* module_of, int32_t n_modules, double* agent_out, double* env_out, int32_t clear, void* stream) {
    if (!e || !agent_out || !env_out || n_modules < 1 || n_modules > CDA_EM_MAX_MODULES) return CDA_ERR_INVALID;
    HIPCHK(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_em_partial, dim3(EM_BLOCKS), dim3(256), 0, (hipStream_t)stream, e->arena, e->P, module_of, (int)n_modules, (int)(clear != 0), e->em_partials);
    const int n_out = n_modules * CDA_EM_AGENT_FIELDS + CDA_EM_ENV_FIELDS;
    hipLaunchKernelGGL(k_em_final, dim3((unsigned)n_out), dim3(64), 0, (hipStream_t)stream, (const double*)e->em_partials, (int)n_modules, agent_out, env_out);
    HIPCHK(hipGetLastError());
    return CDA_OK;
}

int cda_place_order(cda_env* e, int32_t market, int32_t trader, int32_t type, int32_t side, int32_t size, int32_t price) {
    if (!e || market < 0 || market >= e->P.n_markets || trader < 0 || trader >= e->P.cfg.num_agents) return CDA_ERR_INVALID;
    if (type < 0 || type > 3 || side < 0 || side > 1 || size < 1) return CDA_ERR_INVALID;
    if (type != 0 && price < 1) return CDA_ERR_INVALID;
    HIPCHK(hipSetDevice(e->device));
    LAUNCH_CAP(e, k_place_order, dim3(1), dim3(64), smem_for(e, 1), 0, e->arena, e->P, market, trader, type, side, size, price);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    return CDA_OK;
}
int cda_mark_to_mkt(cda_env* e, int32_t market) {
    if (!e || market < 0 || market >= e->P.n_markets) return CDA_ERR_INVALID;
    HIPCHK(hipSetDevice(e->device));
    LAUNCH_CAP(e, k_mark_to_mkt, dim3(1), dim3(64), smem_for(e, 1), 0, e->arena, e->P, market);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    return CDA_OK;
}

// one side of a market's book in queue order -> out[0 .. min(n, max_orders)); `rec` = host copy of the market record
static int read_book_side(cda_env* e, int32_t market, const uint8_t* rec, int sd, cda_order* out, int32_t max_orders, int32_t* n_out) {
    const Params& P = e->P;
    const uint32_t* h = (const uint32_t*)rec;
    const int32_t* tile = (const int32_t*)(rec + P.lay.book_off);
    int32_t shdr[16] = {0};
    const bool has_tail = P.lay.spill_cap > 0 && (h[H_STATUS] & (uint32_t)(ST_TAIL_BID << sd)) != 0;
    const uint8_t* sp = e->arena + spill_arena_off(P) + (size_t)market * spill_region_bytes(P.lay.spill_cap);
    if (has_tail) HIPCHK(hipMemcpy(shdr, sp, sizeof shdr, hipMemcpyDeviceToHost));
    const int tile_n = (int)(sd == 0 ? h[H_N_BIDS] : h[H_N_ASKS]), tail_n = has_tail ? shdr[sd] : 0, base = shdr[2 + sd];
    if (tile_n < 0 || tile_n > e->cap || tail_n < 0 || tail_n > P.lay.spill_cap) return CDA_ERR_INVALID;
    *n_out = tile_n + tail_n;
    int32_t* ring = NULL;
    const int want_tail = max_orders > tile_n ? (tail_n < max_orders - tile_n ? tail_n : max_orders - tile_n) : 0;
    if (want_tail > 0) {                                 // the side's four ring arrays, whole (a dump is not a hot path)
        const size_t bytes = (size_t)BOOK_FIELDS * (size_t)P.lay.spill_cap * 4;
        ring = (int32_t*)malloc(bytes);
        if (!ring) return CDA_ERR_NOMEM;
        hipError_t he = hipMemcpy(ring, sp + 64 + (size_t)sd * bytes, bytes, hipMemcpyDeviceToHost);
        if (he != hipSuccess) { free(ring); return hip_fail(he, "hipMemcpy D2H (spill ring)"); }
    }
    for (int i = 0; i < tile_n + want_tail && i < max_orders; i++) {
        int32_t f[BOOK_FIELDS];
        for (int k = 0; k < BOOK_FIELDS; k++) {
            if (i < tile_n) f[k] = tile[k * e->cap + book_phys_rt(e->cap, sd, i)];
            else f[k] = ring[(size_t)k * (size_t)P.lay.spill_cap + ((uint32_t)(base + (i - tile_n)) & (uint32_t)(P.lay.spill_cap - 1))];
        }
        out[i].price = f[0]; out[i].qty = f[1]; out[i].owner = f[2] & 15; out[i].order_id = (int32_t)((uint32_t)f[2] >> 4); out[i].timestamp = f[3];
    }
    free(ring);
    return CDA_OK;
}
int cda_get_book(cda_env* e, int32_t market, int32_t side, cda_order* orders_out_host, int32_t max_orders, int32_t* n_out_host) {
    if (!e || market < 0 || market >= e->P.n_markets || side < 0 || side > 1 || max_orders < 0 || (max_orders > 0 && !orders_out_host) || !n_out_host) return CDA_ERR_INVALID;
    HIPCHK(hipSetDevice(e->device));
    const Params& P = e->P;
    HIPCHK(hipDeviceSynchronize());
    uint8_t* rec = (uint8_t*)malloc((size_t)P.lay.stride);
    if (!rec) return CDA_ERR_NOMEM;
    hipError_t he = hipMemcpy(rec, e->arena + (size_t)market * (size_t)P.lay.stride, (size_t)P.lay.stride, hipMemcpyDeviceToHost);
    if (he != hipSuccess) { free(rec); return hip_fail(he, "hipMemcpy D2H"); }
    int rc = read_book_side(e, market, rec, side, orders_out_host, max_orders, n_out_host);
    free(rec);
    return rc;
}

int cda_get_state(cda_env* e, int32_t market, cda_market_state* s) {
    if (!e || !s || market < 0 || market >= e->P.n_markets) return CDA_ERR_INVALID;
    HIPCHK(hipSetDevice(e->device));
    const Params& P = e->P;
    HIPCHK(hipDeviceSynchronize());
    uint8_t* rec = (uint8_t*)malloc((size_t)P.lay.stride);
    if (!rec) return CDA_ERR_NOMEM;
    hipError_t he = hipMemcpy(rec, e->arena + (size_t)market * (size_t)P.lay.stride, (size_t)P.lay.stride, hipMemcpyDeviceToHost);
    if (he != hipSuccess) { free(rec); return hip_fail(he, "hipMemcpy D2H"); }
    memset(s, 0, sizeof *s);
    const uint32_t* h = (const uint32_t*)rec;
    s->rng_state_lo = (uint64_t)h[H_RNG_STATE_LO] | ((uint64_t)h[H_RNG_STATE_LO + 1] << 32);
    s->rng_state_hi = (uint64_t)h[H_RNG_STATE_HI] | ((uint64_t)h[H_RNG_STATE_HI + 1] << 32);
    s->rng_inc_lo = (uint64_t)h[H_RNG_INC_LO] | ((uint64_t)h[H_RNG_INC_LO + 1] << 32);
    s->rng_inc_hi = (uint64_t)h[H_RNG_INC_HI] | ((uint64_t)h[H_RNG_INC_HI + 1] << 32);
    s->rng_has_uint32 = h[H_HAS_U32]; s->rng_uinteger = h[H_UINTEGER];
    s->t_step = (int32_t)h[H_T_STEP]; s->lob_time = (int32_t)h[H_LOB_TIME]; s->next_order_id = (int32_t)h[H_NEXT_OID];
    s->last_price = (int32_t)h[H_LAST_PRICE]; s->has_trade = (int32_t)h[H_HAS_TRADE]; s->last_trade_price = (int32_t)h[H_LAST_TRADE_PRICE];
    s->done_mask = h[H_DONE_MASK]; s->flags = h[H_FLAGS];
    {
        int32_t nn[2] = {0, 0};
        for (int sd = 0; sd < 2; sd++) {
            int rc = read_book_side(e, market, rec, sd, sd == 0 ? s->bids : s->asks, CDA_BOOK_CAP_MAX, &nn[sd]);
            if (rc) { free(rec); return rc; }
        }
        s->n_bids = nn[0]; s->n_asks = nn[1];              // the true lengths; the arrays hold the first CDA_BOOK_CAP_MAX of each side
    }
    const Acc* ap = (const Acc*)(rec + P.lay.acc_off);
    for (int a = 0; a < P.cfg.num_agents; a++) {
        cda_account_state* o = &s->acc[a];
        o->cash = ap[a].cash; o->cash_on_hold = ap[a].hold; o->position_val = ap[a].posval; o->vwap = ap[a].vwap;
        o->nav = ap[a].nav; o->prev_nav = ap[a].prev_nav; o->max_nav = ap[a].max_nav;
        o->net_position = ap[a].net_position; o->num_trades = ap[a].num_trades;
        o->num_trades_step = ap[a].num_trades_step; o->num_passive_fills_step = ap[a].num_passive_fills_step;
        o->order_step_placed = ap[a].order_step_placed; o->num_rejected_step = ap[a].num_rejected_step;
    }
    const float* hp = (const float*)(rec + P.lay.hist_off);
    int H = P.cfg.n_hist, head = (int)h[H_HIST_HEAD];
    for (int j = 0; j < H; j++) {
        int slot = (head + j) % H;
        memcpy(s->hist + j * CDA_SNAPSHOT_DIM, hp + slot * CDA_SNAPSHOT_DIM, sizeof(float) * CDA_SNAPSHOT_DIM);
    }
    free(rec);
    return CDA_OK;
}

/* The book is restored from the struct's arrays (up to CDA_BOOK_CAP_MAX orders per side: what does not fit the LDS tile goes
 * to the HBM spill ring).  A state whose side is LONGER than the arrays (a dump of a big book) can only be put back onto the
 * market it came from: if both lengths equal the market's current ones its book is left in place and everything else is
 * restored (get_state -> edit accounts -> set_state round trips on any book); otherwise CDA_ERR_INVALID. */
int cda_set_state(cda_env* e, int32_t market, const cda_market_state* s) {
    if (!e || !s || market < 0 || market >= e->P.n_markets) return CDA_ERR_INVALID;
    if (s->n_bids < 0 || s->n_asks < 0) return CDA_ERR_INVALID;
    HIPCHK(hipSetDevice(e->device));
    const Params& P = e->P;
    const int CAP = e->cap;
    HIPCHK(hipDeviceSynchronize());
    uint8_t* rec = (uint8_t*)calloc(1, (size_t)P.lay.stride);
    if (!rec) return CDA_ERR_NOMEM;
    uint32_t* h = (uint32_t*)rec;
    uint8_t* dev_rec = e->arena + (size_t)market * (size_t)P.lay.stride;
    uint8_t* dev_spill = e->arena + spill_arena_off(P) + (size_t)market * spill_region_bytes(P.lay.spill_cap);
    const bool keep_book = s->n_bids > CDA_BOOK_CAP_MAX || s->n_asks > CDA_BOOK_CAP_MAX;
    int32_t* bp = (int32_t*)(rec + P.lay.book_off);
    if (keep_book) {
        hipError_t he = hipMemcpy(rec, dev_rec, (size_t)P.lay.stride, hipMemcpyDeviceToHost);
        if (he != hipSuccess) { free(rec); return hip_fail(he, "hipMemcpy D2H"); }
        int32_t shdr[16] = {0};
        if (P.lay.spill_cap > 0 && (h[H_STATUS] & ST_TAIL_ANY)) {
            he = hipMemcpy(shdr, dev_spill, sizeof shdr, hipMemcpyDeviceToHost);
            if (he != hipSuccess) { free(rec); return hip_fail(he, "hipMemcpy D2H"); }
        }
        const int cur_b = (int)h[H_N_BIDS] + ((h[H_STATUS] & ST_TAIL_BID) ? shdr[0] : 0), cur_a = (int)h[H_N_ASKS] + ((h[H_STATUS] & ST_TAIL_ASK) ? shdr[1] : 0);
        if (cur_b != s->n_bids || cur_a != s->n_asks) { free(rec); return CDA_ERR_INVALID; }
        h[H_STATUS] &= ~(uint32_t)ST_LEVELS_VALID;
    } else {
        // tile / tail split of a restored book: everything in the tile when it fits, else the tile is shared evenly
        int tb = s->n_bids, ta = s->n_asks;
        if (tb + ta > CAP) {
            tb = s->n_bids < CAP / 2 ? s->n_bids : CAP / 2;
            ta = s->n_asks < CAP - tb ? s->n_asks : CAP - tb;
            tb = s->n_bids < CAP - ta ? s->n_bids : CAP - ta;
        }
        const int tail_n[2] = {s->n_bids - tb, s->n_asks - ta}, tile_n[2] = {tb, ta};
        if (tail_n[0] > P.lay.spill_cap || tail_n[1] > P.lay.spill_cap) { free(rec); return CDA_ERR_INVALID; }
        h[H_N_BIDS] = (uint32_t)tb; h[H_N_ASKS] = (uint32_t)ta;
        h[H_STATUS] = (tail_n[0] > 0 ? ST_TAIL_BID : 0) | (tail_n[1] > 0 ? ST_TAIL_ASK : 0) | (P.lay.ep_on ? ST_EP_ON : 0);
        int32_t* ring = NULL;
        const size_t side_bytes = (size_t)BOOK_FIELDS * (size_t)P.lay.spill_cap * 4;
        if (tail_n[0] > 0 || tail_n[1] > 0) { ring = (int32_t*)calloc(1, 2 * side_bytes); if (!ring) { free(rec); return CDA_ERR_NOMEM; } }
        for (int sd = 0; sd < 2; sd++) {
            const int n = sd == 0 ? s->n_bids : s->n_asks;
            for (int i = 0; i < n; i++) {
                const cda_order* o = sd == 0 ? &s->bids[i] : &s->asks[i];
                if (o->owner < 0 || o->owner >= P.cfg.num_agents || o->order_id < 0 || o->order_id >= (1 << 27)) { free(rec); free(ring); return CDA_ERR_INVALID; }
                const int32_t f[BOOK_FIELDS] = {o->price, o->qty, (int32_t)(((uint32_t)o->order_id << 4) | (uint32_t)o->owner), o->timestamp};
                for (int k = 0; k < BOOK_FIELDS; k++) {
                    if (i < tile_n[sd]) bp[k * CAP + book_phys_rt(CAP, sd, i)] = f[k];
                    else ring[((size_t)sd * BOOK_FIELDS + (size_t)k) * (size_t)P.lay.spill_cap + (size_t)(i - tile_n[sd])] = f[k];
                }
            }
        }
        if (ring) {
            const int32_t shdr[16] = {tail_n[0], tail_n[1], 0, 0};
            hipError_t he = hipMemcpy(dev_spill, shdr, sizeof shdr, hipMemcpyHostToDevice);
            if (he == hipSuccess) he = hipMemcpy(dev_spill + 64, ring, 2 * side_bytes, hipMemcpyHostToDevice);
            free(ring);
            if (he != hipSuccess) { free(rec); return hip_fail(he, "hipMemcpy H2D (spill ring)"); }
        }
        h[H_PEAK_ORDERS] = (uint32_t)(s->n_bids + s->n_asks);
    }
    h[H_RNG_STATE_LO] = (uint32_t)s->rng_state_lo; h[H_RNG_STATE_LO + 1] = (uint32_t)(s->rng_state_lo >> 32);
    h[H_RNG_STATE_HI] = (uint32_t)s->rng_state_hi; h[H_RNG_STATE_HI + 1] = (uint32_t)(s->rng_state_hi >> 32);
    h[H_RNG_INC_LO] = (uint32_t)s->rng_inc_lo; h[H_RNG_INC_LO + 1] = (uint32_t)(s->rng_inc_lo >> 32);
    h[H_RNG_INC_HI] = (uint32_t)s->rng_inc_hi; h[H_RNG_INC_HI + 1] = (uint32_t)(s->rng_inc_hi >> 32);
    h[H_HAS_U32] = s->rng_has_uint32; h[H_UINTEGER] = s->rng_uinteger;
    h[H_T_STEP] = (uint32_t)s->t_step; h[H_LOB_TIME] = (uint32_t)s->lob_time; h[H_NEXT_OID] = (uint32_t)s->next_order_id;
    h[H_LAST_PRICE] = (uint32_t)s->last_price; h[H_HAS_TRADE] = (uint32_t)s->has_trade; h[H_LAST_TRADE_PRICE] = (uint32_t)s->last_trade_price;
    h[H_DONE_MASK] = s->done_mask; h[H_FLAGS] = s->flags;
    h[H_SEEDED] = 1; h[H_HIST_HEAD] = 0;
    Acc* ap = (Acc*)(rec + P.lay.acc_off);
    for (int a = 0; a < P.cfg.num_agents; a++) {
        const cda_account_state* o = &s->acc[a];
        ap[a].cash = o->cash; ap[a].hold = o->cash_on_hold; ap[a].posval = o->position_val; ap[a].vwap = o->vwap;
        ap[a].nav = o->nav; ap[a].prev_nav = o->prev_nav; ap[a].max_nav = o->max_nav;
        ap[a].net_position = o->net_position; ap[a].num_trades = o->num_trades;
        ap[a].num_trades_step = o->num_trades_step; ap[a].num_passive_fills_step = o->num_passive_fills_step;
        ap[a].order_step_placed = o->order_step_placed; ap[a].num_rejected_step = o->num_rejected_step;
    }
    memcpy(rec + P.lay.hist_off, s->hist, sizeof(float) * (size_t)P.cfg.n_hist * CDA_SNAPSHOT_DIM);
    // (the running episode tallies behind the book tile are not part of the dump: they stay as they are)
    hipError_t he = hipMemcpy(rec + P.lay.ep_off, dev_rec + P.lay.ep_off, (size_t)P.cfg.num_agents * sizeof(EpStats), hipMemcpyDeviceToHost);
    if (he == hipSuccess) he = hipMemcpy(dev_rec, rec, (size_t)P.lay.stride, hipMemcpyHostToDevice);
    free(rec);
    if (he != hipSuccess) return hip_fail(he, "hipMemcpy H2D");
    return CDA_OK;
}

int cda_get_raw_snapshot(cda_env* e, float* raw_out, void* stream) {
    if (!e || !raw_out) return CDA_ERR_INVALID;
    HIPCHK(hipSetDevice(e->device));
    LAUNCH_CAP(e, k_raw_snapshot, grid_for(e->P.n_markets), dim3(64 * CDA_WPB), smem_for(e, CDA_WPB), (hipStream_t)stream, e->arena, e->P, raw_out);
    HIPCHK(hipGetLastError());
    return CDA_OK;
}
int cda_last_flags(cda_env* e, uint32_t* flags_out, void* stream) {
    if (!e || !flags_out) return CDA_ERR_INVALID;
    HIPCHK(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_flags, dim3((unsigned)((e->P.n_markets + 255) / 256)), dim3(256), 0, (hipStream_t)stream, e->arena, e->P, flags_out);
    HIPCHK(hipGetLastError());
    return CDA_OK;
}

int cda_check_invariants(cda_env* e, uint32_t* violations_out, void* stream) {
    if (!e || !violations_out) return CDA_ERR_INVALID;
    HIPCHK(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_check_invariants, dim3((unsigned)((e->P.n_markets + 63) / 64)), dim3(64), 0, (hipStream_t)stream, (const uint8_t*)e->arena, e->P, e->cap, violations_out);
    HIPCHK(hipGetLastError());
    return CDA_OK;
}

int cda_selftest_dec(int32_t device, int32_t op, int32_t n, const cda_dec* a_host, const cda_dec* b_host, cda_dec* out_host) {
    if (n < 0 || !a_host || !out_host || op < 0 || op > 6) return CDA_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CDA_ERR_NO_DEVICE;
    HIPCHK(hipSetDevice(device));
    if (n == 0) return CDA_OK;
    cda_dec *da = NULL, *db = NULL, *dout = NULL;
    size_t bytes = (size_t)n * sizeof(cda_dec);
    HIPCHK(hipMalloc((void**)&da, bytes)); HIPCHK(hipMalloc((void**)&dout, bytes));
    HIPCHK(hipMemcpy(da, a_host, bytes, hipMemcpyHostToDevice));
    if (b_host) { HIPCHK(hipMalloc((void**)&db, bytes)); HIPCHK(hipMemcpy(db, b_host, bytes, hipMemcpyHostToDevice)); }
    hipLaunchKernelGGL(k_selftest_dec, dim3((unsigned)((n + 63) / 64)), dim3(64), DEC_TABLE_BYTES, 0, op, n, da, db, dout);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out_host, dout, bytes, hipMemcpyDeviceToHost));
    (void)hipFree(da); (void)hipFree(dout); if (db) (void)hipFree(db);
    return CDA_OK;
}

int cda_selftest_libm(int32_t device, int32_t op, int32_t n, const double* x_host, double* y_host) {
    if (n < 0 || !x_host || !y_host || op < 0 || op > 2) return CDA_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CDA_ERR_NO_DEVICE;
    HIPCHK(hipSetDevice(device));
    if (n == 0) return CDA_OK;
    double *dx = NULL, *dy = NULL;
    const size_t bytes = (size_t)n * sizeof(double);
    HIPCHK(hipMalloc((void**)&dx, bytes)); HIPCHK(hipMalloc((void**)&dy, bytes));
    HIPCHK(hipMemcpy(dx, x_host, bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_selftest_libm, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (int)op, (int)n, dx, dy);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(y_host, dy, bytes, hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(dy);
    return CDA_OK;
}
/* the SAME source compiled for the host: needs no GPU (the CPU suite checks the restatement against the machine's libm) */
int cda_selftest_libm_host(int32_t op, int32_t n, const double* x_host, double* y_host) {
    if (n < 0 || !x_host || !y_host || op < 0 || op > 1) return CDA_ERR_INVALID;
    for (int32_t i = 0; i < n; i++) y_host[i] = op == 0 ? glibc_log1p(x_host[i]) : glibc_exp(x_host[i]);
    return CDA_OK;
}

int cda_selftest_rng(int32_t device, uint64_t seed, int32_t lo, int32_t hi, int32_t n_steps, int32_t n_normals, int32_t perm_n,
                     int32_t* first_int_host, double* normals_host, int32_t* perms_host, uint64_t* final_state_host) {
    if (n_steps < 0 || n_normals < 0 || perm_n < 0 || !first_int_host || !final_state_host) return CDA_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CDA_ERR_NO_DEVICE;
    HIPCHK(hipSetDevice(device));
    int32_t* dfi = NULL; double* dn = NULL; int32_t* dp = NULL; uint64_t* dfs = NULL;
    size_t nb = sizeof(double) * (size_t)n_steps * (size_t)n_normals + 8, pb = sizeof(int32_t) * (size_t)n_steps * (size_t)perm_n + 8;
    HIPCHK(hipMalloc((void**)&dfi, 8)); HIPCHK(hipMalloc((void**)&dn, nb)); HIPCHK(hipMalloc((void**)&dp, pb)); HIPCHK(hipMalloc((void**)&dfs, 64));
    hipLaunchKernelGGL(k_selftest_rng, dim3(1), dim3(64), 0, 0, seed, lo, hi, n_steps, n_normals, perm_n, dfi, dn, dp, dfs);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(first_int_host, dfi, 4, hipMemcpyDeviceToHost));
    if (normals_host && n_steps * n_normals) HIPCHK(hipMemcpy(normals_host, dn, nb - 8, hipMemcpyDeviceToHost));
    if (perms_host && n_steps * perm_n) HIPCHK(hipMemcpy(perms_host, dp, pb - 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(final_state_host, dfs, 48, hipMemcpyDeviceToHost));
    (void)hipFree(dfi); (void)hipFree(dn); (void)hipFree(dp); (void)hipFree(dfs);
    return CDA_OK;
}

#ifdef CDA_PHASE_TIMING
/* debug builds only (not in include/cda.h): device buffer [N,40] of cycle stamps, used by tools/phase_timing.py */
void cda_debug_set_phase_buffer(unsigned long long* dev_buf) { g_phase_cycles = dev_buf; }
#endif
#ifdef CDA_DEBUG_SKIP
/* debug builds only (not in include/cda.h): leave phases of k_step out to price them (wrong results) */
void cda_debug_set_skip(int mask) { g_dbg_skip = mask; }
#endif
#ifdef CDA_DEC_COUNTERS
int cda_debug_dec_calls(unsigned long long* host8, int reset) {
    if (hipMemcpyFromSymbol(host8, HIP_SYMBOL(cda::g_dec_calls), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(cda::g_dec_calls), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
#endif

int32_t cda_book_capacity(const cda_env* e) { return e ? e->cap : 0; }
int32_t cda_book_spill(const cda_env* e) { return e ? e->P.lay.spill_cap : 0; }
int32_t cda_num_markets(const cda_env* e) { return e ? e->P.n_markets : 0; }
int32_t cda_num_agents(const cda_env* e) { return e ? e->P.cfg.num_agents : 0; }
int32_t cda_book_spill_wanted(const cda_env* e) { return e ? e->spill_wanted : 0; }
int32_t cda_obs_dim(const cda_env* e) { return e ? e->P.cfg.n_hist * CDA_SNAPSHOT_DIM : 0; }
int64_t cda_state_bytes_per_market(const cda_env* e) { return e ? (int64_t)e->P.lay.stride : 0; }

}  // extern "C"
