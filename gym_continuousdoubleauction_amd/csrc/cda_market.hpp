// cda_market.hpp - one wavefront steps one market (gfx950 / CDNA4, wave64).
//
// Execution model: a 64-lane wave owns one independent market for the whole step.  The market's
// order book (two sides, queue-ordered SoA arrays) and its accounts are staged HBM -> LDS at kernel
// entry and written back at exit.  Lanes are used for
//   * the order pool: searches (`_get_order_ID`), insert position, shift on insert / removal and the
//     top-10 level aggregation are 64-wide scans over the queue-ordered arrays (ballot + popcount);
//   * the ledger: lane a owns account a; a fill settles its two parties in two lanes at once and
//     mark-to-market / reward / info run on A lanes in parallel (28-digit decimal, cda_dec.hpp);
//   * the observation: 42 lanes compute one snapshot frame.
// The A order operations inside a step are inherently sequential (price-time priority); everything
// wave-uniform (RNG, decode, matching loop control) is executed redundantly by all lanes.
//
// Reference citations are relative to /root/reference/gym_continuousDoubleAuction/envs/.
#ifndef CDA_MARKET_COMMON_HPP
#define CDA_MARKET_COMMON_HPP
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cda.h"
#include "cda_dec.hpp"
#include "cda_libm.hpp"

#define CDA_ZIG_QUAL __device__ const
#include "ziggurat_tables.h"

namespace cda {

constexpr int WAVE = 64;
enum { T_MARKET = 0, T_LIMIT = 1, T_MODIFY = 2, T_CANCEL = 3 };
enum { S_BID = 0, S_ASK = 1, S_NONE = 2 };

// Cross-lane visibility of LDS written by other lanes of the same wave: the hardware executes a
// wave's LDS operations in order, so only the compiler has to be stopped from forwarding values.
#ifdef CDA_WSYNC_FENCE
#define CDA_WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                         __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#else
// A compiler-only barrier is enough: LDS instructions of one wave are issued and executed in program order, so a
// later ds_read always sees an earlier ds_write of ANY lane; nothing has to be drained (a fence would also wait for
// the unrelated global loads/stores in flight).
#define CDA_WSYNC() do { __asm__ volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); __asm__ volatile("" ::: "memory"); } while (0)
#endif

// ---- HBM record of one market (see DESIGN.md "Data layout") --------------------------------
// header words (u32)
enum {
    H_RNG_STATE_LO = 0, H_RNG_STATE_HI = 2, H_RNG_INC_LO = 4, H_RNG_INC_HI = 6, H_HAS_U32 = 8, H_UINTEGER = 9,
    H_T_STEP = 10, H_LOB_TIME = 11, H_NEXT_OID = 12, H_LAST_PRICE = 13, H_HAS_TRADE = 14, H_LAST_TRADE_PRICE = 15,
    H_DONE_MASK = 16, H_FLAGS = 17, H_N_BIDS = 18, H_N_ASKS = 19, H_SEEDED = 20, H_HIST_HEAD = 21,
    H_STATUS = 22,            // ST_* bits below (bit 0 used to be the whole word: "levels valid")
    H_PEAK_ORDERS = 23,       // most resting orders (both sides together) the market has held since its last reset (cda_book_peak)
    H_LEVELS = 24,            // 40 words: the top-10 aggregation (lvl_px[2][10], lvl_sz[2][10]) of the book as stored,
                              // so that the next step's pre-step snapshot is a copy instead of a scan (valid flag above)
    H_WORDS = 64
};
constexpr int HEADER_BYTES = H_WORDS * 4;   // 256: one coalesced load per wave
// H_STATUS bits.  H_N_BIDS / H_N_ASKS count the orders of a side that live in the record's book TILE (the LDS-staged top of
// the book); a side whose ST_TAIL bit is set continues in the market's HBM spill ring (cda_book.inc), otherwise the tile is
// the whole side - the only case the hot path ever sees.
enum { ST_LEVELS_VALID = 1, ST_TAIL_BID = 2, ST_TAIL_ASK = 4, ST_TAIL_ANY = 6,
       ST_EP_SUMMARISED = 8,     // episode metrics: this market's finished episode has been checked and credited (a later reset must not do it again)
       ST_EP_ON = 16 };          // episode metrics are on (cda_episode_metrics_enable sets it in every market's header): the step learns it from the record it loads
                                 // anyway - as a kernel argument it would be one more scalar the hot kernel carries through every phase

struct Acc {                     // 144 B, 16-byte aligned; lane a owns account a
    cda_dec cash, hold, posval, vwap, nav, prev_nav, max_nav;     // 7 x 16 B
    int32_t net_position, num_trades;
    int32_t num_trades_step, num_passive_fills_step, order_step_placed, num_rejected_step;
    int32_t pad[2];
};
static_assert(sizeof(Acc) == 144, "Acc layout");

// Episode metrics (include/cda.h cda_episode_metrics_*): the RUNNING episode's tallies of agent a - what the reference's callback keeps per episode
// (train/callbk/league_based_self_play_callback.py:55-90 _new_tally, :541-600 on_episode_step) - live in the market record behind the book tile.  The
// step adds to them with fire-and-forget atomics (nothing is loaded, no register is held across the step); the cold episode-end path reads and clears them.
struct EpStats {                 // 112 B
    double term_sum[CDA_NUM_REWARD_TERMS], term_sq[CDA_NUM_REWARD_TERMS];   // sum / sum of squares of each reward term over the episode's steps
    double ret;                  // the episode's return so far
    int32_t passes, rejections, placed, trades, passive, pad;
};
static_assert(sizeof(EpStats) == 112, "EpStats layout");

constexpr int BOOK_FIELDS = 4;
__device__ __forceinline__ int oo_owner(int32_t oo) { return oo & 15; }
__device__ __forceinline__ int oo_pack(int32_t oid, int owner) { return (int32_t)(((uint32_t)oid << 4) | (uint32_t)owner); }

struct Layout {                  // byte offsets inside a market record
    int32_t acc_off, hist_off, book_off, stride;
    int32_t spill_cap;           // orders per side the market's HBM spill ring holds (a power of two; 0 = no HBM tier)
    int32_t ep_off;              // EpStats[num_agents] (always laid out; touched only while ep_on)
    int32_t ep_on;               // cda_episode_metrics_enable (host copy of the headers' ST_EP_ON: k_reset stages the decimal tables for the episode-end check)
};

// the record layout of an env with `num_agents` agents, `n_hist` frames and a book tile of `cap` orders
__host__ __device__ constexpr Layout record_layout(int num_agents, int n_hist, int cap) {
    Layout l{};
    int off = HEADER_BYTES;
    l.acc_off = off; off += num_agents * (int)sizeof(Acc);
    l.hist_off = off; off += n_hist * CDA_SNAPSHOT_DIM * 4; off = (off + 15) & ~15;
    l.book_off = off; off += cap * BOOK_FIELDS * 4;
    l.ep_off = off; off += num_agents * (int)sizeof(EpStats);
    l.stride = (off + 255) & ~255;
    l.spill_cap = 0;
    l.ep_on = 0;
    return l;
}

struct Params {
    cda_config cfg;
    Layout lay;
    int32_t n_markets;
    float mkt_mul, lim_mul;
    double ep_tol;               // episode metrics: the callback's nav_tolerance (1e-6)
};

// ---- uniform (per-wave) market scalars kept in registers -----------------------------------
struct Mkt {
    u128 rng_state, rng_inc;
    uint32_t has_u32, uinteger;
    int32_t t_step, lob_time, next_oid, last_price, has_trade, last_trade_price;
    uint32_t done_mask, flags;
    int32_t nb, na;              // resting orders per side (two scalars: a dynamically indexed array would force Mkt into scratch)
    int32_t seeded, hist_head, status;   // status: ST_* bits (cached level aggregation valid; which sides continue in the HBM spill ring)
    int32_t peak_orders;         // census: see H_PEAK_ORDERS
#ifdef CDA_DEBUG_SKIP
    int32_t dbg;                 // debug: pieces of the order phase to leave out (tools/inst_count.sh); results are then wrong
#endif
    int32_t fills;               // fills settled in the current step (issue priority of this wave grows with it)
#ifdef CDA_PHASE_TIMING
    unsigned long long tacc[24];    // debug: cycles in approval / find / match+settle / insert+remove / escrow+cancel, fills, ...; 14..20: inside a fill
#endif
};
#ifdef CDA_PHASE_TIMING
#define TACC_BEGIN() unsigned long long _tb = __builtin_readcyclecounter()
#define TACC_END(m, i) do { unsigned long long _te = __builtin_readcyclecounter(); (m).tacc[i] += _te - _tb; _tb = _te; } while (0)
#define TACC_COUNT(m, i, n) do { (m).tacc[i] += (n); } while (0)
#define CDA_TF_PARAM , unsigned long long* tf
#define CDA_TF_ARG(m) , (m).tacc
#define TF_BEGIN() unsigned long long _fb = __builtin_readcyclecounter()
#define TF_END(i) do { unsigned long long _fe = __builtin_readcyclecounter(); tf[i] += _fe - _fb; _fb = _fe; } while (0)
#define TF_COUNT(i) do { tf[i] += 1; } while (0)
#define TF_RESYNC() do { _fb = __builtin_readcyclecounter(); } while (0)
#else
#define CDA_TF_PARAM
#define CDA_TF_ARG(m)
#define TF_BEGIN() do {} while (0)
#define TF_END(i) do {} while (0)
#define TF_COUNT(i) do {} while (0)
#define TF_RESYNC() do {} while (0)
#define TACC_BEGIN() do {} while (0)
#define TACC_END(m, i) do {} while (0)
#define TACC_COUNT(m, i, n) do {} while (0)
#endif
#ifdef CDA_DEBUG_SKIP
#define CDA_MKT_DBG(m, bit) (((m).dbg & (bit)) != 0)
#else
#define CDA_MKT_DBG(m, bit) false
#endif
__device__ __forceinline__ int mkt_n(const Mkt& m, int s) { return s == 0 ? m.nb : m.na; }
__device__ __forceinline__ void mkt_set_n(Mkt& m, int s, int v) { if (s == 0) m.nb = v; else m.na = v; }
__device__ __forceinline__ bool mkt_has_tail(const Mkt& m, int s) { return (m.status & (ST_TAIL_BID << s)) != 0; }

__device__ __forceinline__ D ld_dec(const cda_dec& p) { return d_make(p.w[0], p.w[1], p.w[2], (int)p.exp, (int)p.sign); }
__device__ __forceinline__ void st_dec(cda_dec& p, const D& d, uint32_t& flags) {
    p.w[0] = d.w0; p.w[1] = d.w1; p.w[2] = d.w2;
    if (d.exp < -32768 || d.exp > 32767) flags |= CDA_FLAG_DEC_DOMAIN;
    p.exp = (int16_t)d.exp; p.sign = (uint8_t)d.sign; p.pad = 0;
}

// ======================================================================================
// numpy RNG: PCG64 XSL-RR, half-word buffered 32-bit draws, Lemire bounded ints,
// masked-rejection intervals, ziggurat normal.  Wave-uniform: every lane computes the same stream.
// ======================================================================================
__device__ __forceinline__ u128 pcg_mult() { return ((u128)0x2360ed051fc65da4ULL << 64) | 0x4385df649fccf645ULL; }
__device__ __forceinline__ uint64_t rng_next64(Mkt& m) {
    m.rng_state = m.rng_state * pcg_mult() + m.rng_inc;
    uint64_t hi = (uint64_t)(m.rng_state >> 64), lo = (uint64_t)m.rng_state, x = hi ^ lo;
    unsigned rot = (unsigned)(hi >> 58);
    return (x >> rot) | (x << ((64u - rot) & 63u));
}
__device__ __forceinline__ uint64_t pcg_output(u128 state) {
    uint64_t hi = (uint64_t)(state >> 64), lo = (uint64_t)state, x = hi ^ lo;
    unsigned rot = (unsigned)(hi >> 58);
    return (x >> rot) | (x << ((64u - rot) & 63u));
}
// Jump-ahead of the LCG: after k + 1 advances state = M^(k+1) * s + G_(k+1) * inc with G_n = 1 + M + ... + M^(n-1).
// Lets the k-th present agent of a market compute ITS draw without waiting for the k draws before it.
struct PcgJump { uint64_t mlo[CDA_MAX_AGENTS], mhi[CDA_MAX_AGENTS], glo[CDA_MAX_AGENTS], ghi[CDA_MAX_AGENTS]; };
constexpr PcgJump make_pcg_jump() {
    PcgJump t{};
    const u128 M = ((u128)0x2360ed051fc65da4ULL << 64) | 0x4385df649fccf645ULL;
    u128 mk = 1, gk = 0;
    for (int k = 0; k < CDA_MAX_AGENTS; k++) {
        gk = gk * M + 1; mk = mk * M;
        t.mlo[k] = (uint64_t)mk; t.mhi[k] = (uint64_t)(mk >> 64); t.glo[k] = (uint64_t)gk; t.ghi[k] = (uint64_t)(gk >> 64);
    }
    return t;
}
__device__ const PcgJump PCG_JUMP = make_pcg_jump();
constexpr int PCG_JUMP_WORDS64 = 4 * CDA_MAX_AGENTS;
__device__ __forceinline__ u128 pcg_jump_state(const unsigned long long* jt, int k, u128 state, u128 inc) {   // jt: LDS copy of PCG_JUMP
    u128 mk = ((u128)jt[CDA_MAX_AGENTS + k] << 64) | jt[k], gk = ((u128)jt[3 * CDA_MAX_AGENTS + k] << 64) | jt[2 * CDA_MAX_AGENTS + k];
    return mk * state + gk * inc;
}
__device__ __forceinline__ uint32_t rng_next32(Mkt& m) {
    if (m.has_u32) { m.has_u32 = 0; return m.uinteger; }
    uint64_t v = rng_next64(m);
    m.has_u32 = 1; m.uinteger = (uint32_t)(v >> 32);
    return (uint32_t)v;
}
__device__ __forceinline__ double rng_double(Mkt& m) { return (double)(rng_next64(m) >> 11) * (1.0 / 9007199254740992.0); }

__device__ __forceinline__ uint32_t ss_hashmix(uint32_t value, uint32_t& hc) { value ^= hc; hc *= 0x931e8875u; value *= hc; value ^= value >> 16; return value; }
__device__ __forceinline__ uint32_t ss_mix(uint32_t x, uint32_t y) { uint32_t r = 0xca01f9ddu * x - 0x4973f715u * y; r ^= r >> 16; return r; }
// Generator(PCG64(SeedSequence(seed))) for an integer seed < 2^64
__device__ __forceinline__ void rng_seed(Mkt& m, uint64_t seed) {
    uint32_t e0 = (uint32_t)seed, e1 = (uint32_t)(seed >> 32);
    uint32_t pool[4], hc = 0x43b0d7e5u;
    pool[0] = ss_hashmix(e0, hc);
    pool[1] = ss_hashmix(e1, hc);      // entropy word 1 (0 when the seed fits 32 bits: same as padding)
    pool[2] = ss_hashmix(0u, hc);
    pool[3] = ss_hashmix(0u, hc);
    #pragma unroll
    for (int s = 0; s < 4; s++) {
        #pragma unroll
        for (int d = 0; d < 4; d++) if (s != d) pool[d] = ss_mix(pool[d], ss_hashmix(pool[s], hc));
    }
    uint32_t hb = 0x8b51f9ddu, st[8];
    #pragma unroll
    for (int i = 0; i < 8; i++) { uint32_t v = pool[i & 3]; v ^= hb; hb *= 0x58f38dedu; v *= hb; v ^= v >> 16; st[i] = v; }
    uint64_t v0 = (uint64_t)st[0] | ((uint64_t)st[1] << 32), v1 = (uint64_t)st[2] | ((uint64_t)st[3] << 32);
    uint64_t v2 = (uint64_t)st[4] | ((uint64_t)st[5] << 32), v3 = (uint64_t)st[6] | ((uint64_t)st[7] << 32);
    u128 initstate = ((u128)v0 << 64) | v1, initseq = ((u128)v2 << 64) | v3;
    m.rng_inc = (initseq << 1) | 1;
    m.rng_state = m.rng_inc;                         // 0 * mult + inc
    m.rng_state += initstate;
    m.rng_state = m.rng_state * pcg_mult() + m.rng_inc;
    m.has_u32 = 0; m.uinteger = 0;
}
__device__ __forceinline__ int32_t rng_integers(Mkt& m, int32_t lo, int32_t hi_incl) {
    uint32_t rng = (uint32_t)(hi_incl - lo);
    if (rng == 0) return lo;
    uint32_t rng_excl = rng + 1u;
    uint64_t mm = (uint64_t)rng_next32(m) * rng_excl;
    uint32_t leftover = (uint32_t)mm;
    if (leftover < rng_excl) {
        uint32_t threshold = (0xFFFFFFFFu - rng) % rng_excl;
        while (leftover < threshold) { mm = (uint64_t)rng_next32(m) * rng_excl; leftover = (uint32_t)mm; }
    }
    return lo + (int32_t)(mm >> 32);
}
__device__ __forceinline__ uint32_t rng_interval(Mkt& m, uint32_t max) {
    uint32_t mask = max, v;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    do { v = rng_next32(m) & mask; } while (v > max);
    return v;
}
// wi / ki: the two tables every draw reads (numpy's wi_double, ki_double).  k_step passes LDS copies (the index is data
// dependent, so each draw pays the table latency in full: ~100 cycles from LDS instead of a scalar-cache / L2 round
// trip); fi is only read on the rare wedge path and stays in global memory.
// first candidate of the ziggurat from one 64-bit draw; true = accepted (98.8 % of the draws)
__device__ __forceinline__ bool zig_first_candidate(uint64_t u, const unsigned long long* wi, const unsigned long long* ki, double* x_out) {
    int idx = (int)(u & 0xff); u >>= 8;
    int sign = (int)(u & 1);
    uint64_t rabs = (u >> 1) & 0x000fffffffffffffULL;
    double x = (double)rabs * __longlong_as_double((long long)wi[idx]);
    *x_out = sign ? -x : x;
    return rabs < ki[idx];
}
__device__ __forceinline__ double rng_std_normal(Mkt& m, const unsigned long long* wi = cda_zig_wi_bits, const unsigned long long* ki = cda_zig_ki) {
    const double zr = 3.6541528853610087963519472518, inv_r = 0.27366123732975827203338247596;
    for (;;) {
        uint64_t u = rng_next64(m);
        int idx = (int)(u & 0xff); u >>= 8;
        int sign = (int)(u & 1);
        uint64_t rabs = (u >> 1) & 0x000fffffffffffffULL;
        double x = (double)rabs * __longlong_as_double((long long)wi[idx]);
        if (sign) x = -x;
        if (rabs < ki[idx]) return x;
        if (idx == 0) {
            for (;;) {
                double xx = -inv_r * glibc_log1p(-rng_double(m));
                double yy = -glibc_log1p(-rng_double(m));
                if (yy + yy > xx * xx) return ((rabs >> 8) & 1) ? -(zr + xx) : zr + xx;
            }
        } else {
            double f1 = __longlong_as_double((long long)cda_zig_fi_bits[idx - 1]);
            double f0 = __longlong_as_double((long long)cda_zig_fi_bits[idx]);
            if ((f1 - f0) * rng_double(m) + f0 < glibc_exp(-0.5 * x * x)) return x;
        }
    }
}

// ======================================================================================
// HBM <-> LDS staging
// ======================================================================================
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (WAVE - 1)); }

__device__ __forceinline__ uint32_t load_header_word(const uint32_t* hp, int lane) { return lane < H_WORDS ? hp[lane] : 0u; }
__device__ __forceinline__ void decode_header(uint32_t v, Mkt& m);
__device__ __forceinline__ void load_header(const uint32_t* hp, Mkt& m, int lane) { decode_header(load_header_word(hp, lane), m); }
__device__ __forceinline__ void decode_header(uint32_t v, Mkt& m) {
    #define RL(i) ((uint32_t)__builtin_amdgcn_readlane((int)v, (i)))
    uint64_t slo = (uint64_t)RL(H_RNG_STATE_LO) | ((uint64_t)RL(H_RNG_STATE_LO + 1) << 32);
    uint64_t shi = (uint64_t)RL(H_RNG_STATE_HI) | ((uint64_t)RL(H_RNG_STATE_HI + 1) << 32);
    uint64_t ilo = (uint64_t)RL(H_RNG_INC_LO) | ((uint64_t)RL(H_RNG_INC_LO + 1) << 32);
    uint64_t ihi = (uint64_t)RL(H_RNG_INC_HI) | ((uint64_t)RL(H_RNG_INC_HI + 1) << 32);
    m.rng_state = ((u128)shi << 64) | slo; m.rng_inc = ((u128)ihi << 64) | ilo;
    m.has_u32 = RL(H_HAS_U32); m.uinteger = RL(H_UINTEGER);
    m.t_step = (int32_t)RL(H_T_STEP); m.lob_time = (int32_t)RL(H_LOB_TIME); m.next_oid = (int32_t)RL(H_NEXT_OID);
    m.last_price = (int32_t)RL(H_LAST_PRICE); m.has_trade = (int32_t)RL(H_HAS_TRADE); m.last_trade_price = (int32_t)RL(H_LAST_TRADE_PRICE);
    m.done_mask = RL(H_DONE_MASK); m.flags = RL(H_FLAGS);
    m.nb = (int32_t)RL(H_N_BIDS); m.na = (int32_t)RL(H_N_ASKS);
    m.seeded = (int32_t)RL(H_SEEDED); m.hist_head = (int32_t)RL(H_HIST_HEAD); m.status = (int32_t)RL(H_STATUS);
    m.peak_orders = (int32_t)RL(H_PEAK_ORDERS);
#ifdef CDA_DEBUG_SKIP
    m.dbg = 0;
#endif
    m.fills = 0;
    #undef RL
}
__device__ __forceinline__ void store_header(uint32_t* hp, const Mkt& m, int lane) {
    if (lane == 0) {
        hp[H_RNG_STATE_LO] = (uint32_t)m.rng_state; hp[H_RNG_STATE_LO + 1] = (uint32_t)(m.rng_state >> 32);
        hp[H_RNG_STATE_HI] = (uint32_t)(m.rng_state >> 64); hp[H_RNG_STATE_HI + 1] = (uint32_t)(m.rng_state >> 96);
        hp[H_RNG_INC_LO] = (uint32_t)m.rng_inc; hp[H_RNG_INC_LO + 1] = (uint32_t)(m.rng_inc >> 32);
        hp[H_RNG_INC_HI] = (uint32_t)(m.rng_inc >> 64); hp[H_RNG_INC_HI + 1] = (uint32_t)(m.rng_inc >> 96);
        hp[H_HAS_U32] = m.has_u32; hp[H_UINTEGER] = m.uinteger;
        hp[H_T_STEP] = (uint32_t)m.t_step; hp[H_LOB_TIME] = (uint32_t)m.lob_time; hp[H_NEXT_OID] = (uint32_t)m.next_oid;
        hp[H_LAST_PRICE] = (uint32_t)m.last_price; hp[H_HAS_TRADE] = (uint32_t)m.has_trade; hp[H_LAST_TRADE_PRICE] = (uint32_t)m.last_trade_price;
        hp[H_DONE_MASK] = m.done_mask; hp[H_FLAGS] = m.flags;
        hp[H_N_BIDS] = (uint32_t)m.nb; hp[H_N_ASKS] = (uint32_t)m.na;
        hp[H_SEEDED] = (uint32_t)m.seeded; hp[H_HIST_HEAD] = (uint32_t)m.hist_head; hp[H_STATUS] = (uint32_t)m.status;
        hp[H_PEAK_ORDERS] = (uint32_t)m.peak_orders;
    }
}
__device__ __forceinline__ void copy_words(uint32_t* dst, const uint32_t* src, int nwords, int lane) {
    for (int i = lane; i < nwords; i += WAVE) dst[i] = src[i];
}

// minimum of a 32-bit value over the wave with data-parallel-primitive moves (no LDS crossbar round trips): an inclusive
// min-scan inside each row of 16 lanes (row_shr 1, 2, 4, 8), then row 0 -> 1 and 2 -> 3 (row_bcast:15), then rows 0-1 -> 2-3
// (row_bcast:31); lane 63 holds the result.  Lanes without a source keep the identity.
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    const int ident = -1;                                   // 0xffffffff
    #define CDA_DPP_MIN(ctrl, rmask) { uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(ident, (int)v, (ctrl), (rmask), 0xf, false); v = o < v ? o : v; }
    CDA_DPP_MIN(0x111, 0xf) CDA_DPP_MIN(0x112, 0xf) CDA_DPP_MIN(0x114, 0xf) CDA_DPP_MIN(0x118, 0xf)
    CDA_DPP_MIN(0x142, 0xa) CDA_DPP_MIN(0x143, 0xc)
    #undef CDA_DPP_MIN
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// ======================================================================================
// Ledger (account/account.py, cash_processor.py, calculate.py) - executed by the owning lane
// ======================================================================================
// Account a is owned by lane a; lanes a+16 and a+32 are its HELPERS: independent decimal operations on one account are
// issued on the owner and its helpers in the same instruction stream (CDA_MAX_AGENTS = 16 <= 16 lanes per group), e.g.
// cash -= v on lane a while hold += v on lane a+16.  All addressing is by (lane & 15); every helper sees the same LDS.
__device__ __forceinline__ int lane_acc(int lane) { return lane & 15; }
__device__ __forceinline__ int lane_grp(int lane) { return lane >> 4; }
__device__ __forceinline__ D d_shfl(const D& v, int src) {
    D r;
    r.w0 = (uint32_t)__shfl((int)v.w0, src, WAVE); r.w1 = (uint32_t)__shfl((int)v.w1, src, WAVE); r.w2 = (uint32_t)__shfl((int)v.w2, src, WAVE);
    int es = __shfl((v.exp << 1) | (v.sign & 1), src, WAVE);
    r.exp = es >> 1; r.sign = es & 1;
    return r;
}
__device__ __forceinline__ D cal_profit(bool is_long, D mkt, D raw) { return is_long ? d_sub(mkt, raw) : d_sub(raw, mkt); }

}  // namespace cda
#endif  // CDA_MARKET_COMMON_HPP
