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
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cda.h"
#include "cda_dec.hpp"
#include "cda_libm.hpp"

#define CDA_ZIG_QUAL __device__ const
#include "ziggurat_tables.h"

namespace cda {

constexpr int WAVE = 64;
constexpr int CAP = CDA_BOOK_CAP;
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
    H_DONE_MASK = 16, H_FLAGS = 17, H_N_BIDS = 18, H_N_ASKS = 19, H_SEEDED = 20, H_HIST_HEAD = 21, H_LEVELS_VALID = 22,
    H_PEAK_ORDERS = 23,       // most resting orders (both sides together) the market has held since its last reset (cda_book_peak)
    H_LEVELS = 24,            // 40 words: the top-10 aggregation (lvl_px[2][10], lvl_sz[2][10]) of the book as stored,
                              // so that the next step's pre-step snapshot is a copy instead of a scan (valid flag above)
    H_WORDS = 64
};
constexpr int HEADER_BYTES = H_WORDS * 4;   // 256: one coalesced load per wave

struct Acc {                     // 144 B, 16-byte aligned; lane a owns account a
    cda_dec cash, hold, posval, vwap, nav, prev_nav, max_nav;     // 7 x 16 B
    int32_t net_position, num_trades;
    int32_t num_trades_step, num_passive_fills_step, order_step_placed, num_rejected_step;
    int32_t pad[2];
};
static_assert(sizeof(Acc) == 144, "Acc layout");

// The two sides share ONE pool of CAP slots per field: bids grow up from slot 0, asks grow down from slot CAP-1
// (logical index i of the ask side lives in physical slot CAP-1-i).  Halves the LDS footprint of a market
// (4 KB), which is what lets more than 4 waves per SIMD be resident; capacity is CAP resting orders per market.
__host__ __device__ constexpr int book_phys(int s, int i) { return s == 0 ? i : CAP - 1 - i; }
struct BookField {
    int32_t v[CAP];
    struct Row { int32_t* base; int s; __device__ __forceinline__ int32_t& operator[](int i) const { return base[book_phys(s, i)]; } };
    struct CRow { const int32_t* base; int s; __device__ __forceinline__ const int32_t& operator[](int i) const { return base[book_phys(s, i)]; } };
    __device__ __forceinline__ Row operator[](int s) { return Row{v, s}; }
    __device__ __forceinline__ CRow operator[](int s) const { return CRow{v, s}; }
};
struct Book {                    // LDS image of the book, queue order per side (best first, FIFO in a level)
    BookField price, qty;
    BookField oo;                // (order_id << 4) | owner   (owner < 16, order_id < 2^27)
    BookField ts;
};
constexpr int BOOK_BYTES = sizeof(Book);   // 4096
constexpr int BOOK_FIELDS = 4;
__device__ __forceinline__ int oo_owner(int32_t oo) { return oo & 15; }
__device__ __forceinline__ int oo_pack(int32_t oid, int owner) { return (int32_t)(((uint32_t)oid << 4) | (uint32_t)owner); }

struct Lds {                     // per-wave LDS image; `acc` is sized for the env's agent count at launch
    Book book;
    int32_t lvl_px[2][CDA_K_ROWS];
    int32_t lvl_sz[2][CDA_K_ROWS];
    int32_t act_tsp[CDA_MAX_AGENTS];   // decoded order of agent a: type | side << 2 | (price + 1) << 4   (price < 2^24)
    int32_t act_size[CDA_MAX_AGENTS];
    Acc acc[CDA_MAX_AGENTS];     // only the first num_agents records are backed by LDS
};
// bytes of LDS one wave needs for `agents` accounts
// (the history ring of the observation, n_hist x 42 floats, is staged behind the accounts: lds_hist())
__host__ __device__ constexpr int lds_bytes_per_wave(int agents, int n_hist) {
    return (int)(sizeof(Book) + 2 * 2 * CDA_K_ROWS * 4 + 2 * CDA_MAX_AGENTS * 4) + agents * (int)sizeof(Acc) + ((n_hist * CDA_SNAPSHOT_DIM * 4 + 15) & ~15);
}

__device__ __forceinline__ float* lds_hist(Lds& L, int agents) { return reinterpret_cast<float*>(&L.acc[agents]); }

struct Layout {                  // byte offsets inside a market record
    int32_t acc_off, hist_off, book_off, stride;
};

struct Params {
    cda_config cfg;
    Layout lay;
    int32_t n_markets;
    float mkt_mul, lim_mul;
};

// ---- uniform (per-wave) market scalars kept in registers -----------------------------------
struct Mkt {
    u128 rng_state, rng_inc;
    uint32_t has_u32, uinteger;
    int32_t t_step, lob_time, next_oid, last_price, has_trade, last_trade_price;
    uint32_t done_mask, flags;
    int32_t nb, na;              // resting orders per side (two scalars: a dynamically indexed array would force Mkt into scratch)
    int32_t seeded, hist_head, levels_valid;
    int32_t peak_orders;         // census: see H_PEAK_ORDERS
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
__device__ __forceinline__ int mkt_n(const Mkt& m, int s) { return s == 0 ? m.nb : m.na; }
__device__ __forceinline__ void mkt_set_n(Mkt& m, int s, int v) { if (s == 0) m.nb = v; else m.na = v; }

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
    m.seeded = (int32_t)RL(H_SEEDED); m.hist_head = (int32_t)RL(H_HIST_HEAD); m.levels_valid = (int32_t)RL(H_LEVELS_VALID);
    m.peak_orders = (int32_t)RL(H_PEAK_ORDERS);
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
        hp[H_SEEDED] = (uint32_t)m.seeded; hp[H_HIST_HEAD] = (uint32_t)m.hist_head; hp[H_LEVELS_VALID] = (uint32_t)m.levels_valid;
        hp[H_PEAK_ORDERS] = (uint32_t)m.peak_orders;
    }
}
// book record in HBM: [field][CAP] int32, pooled like the LDS image; only the live prefix of each side moves.
// Prefetch form: the first 64 entries of every array are requested BEFORE the header (which holds the
// counts) has arrived, so the two HBM round trips of a naive load overlap into one.
struct BookPrefetch { int32_t v[2][BOOK_FIELDS]; };
__device__ __forceinline__ BookPrefetch prefetch_book(const int32_t* bp, int lane) {
    BookPrefetch r;
    #pragma unroll
    for (int s = 0; s < 2; s++) {
        #pragma unroll
        for (int f = 0; f < BOOK_FIELDS; f++) r.v[s][f] = bp[f * CAP + book_phys(s, lane)];
    }
    return r;
}
__device__ __forceinline__ void finish_book_load(const int32_t* bp, const BookPrefetch& pre, Book& bk, const Mkt& m, int lane) {
    #pragma unroll
    for (int s = 0; s < 2; s++) {
        const int ns = mkt_n(m, s);
        if (lane < ns) { bk.price[s][lane] = pre.v[s][0]; bk.qty[s][lane] = pre.v[s][1]; bk.oo[s][lane] = pre.v[s][2]; bk.ts[s][lane] = pre.v[s][3]; }
        for (int i = lane + WAVE; i < ns; i += WAVE) {
            const int ph = book_phys(s, i);
            bk.price[s][i] = bp[0 * CAP + ph]; bk.qty[s][i] = bp[1 * CAP + ph]; bk.oo[s][i] = bp[2 * CAP + ph]; bk.ts[s][i] = bp[3 * CAP + ph];
        }
    }
}
__device__ __forceinline__ void load_book(const int32_t* bp, Book& bk, const Mkt& m, int lane) {
    #pragma unroll
    for (int s = 0; s < 2; s++) {
        for (int i = lane; i < mkt_n(m, s); i += WAVE) {
            const int ph = book_phys(s, i);
            bk.price[s][i] = bp[0 * CAP + ph]; bk.qty[s][i] = bp[1 * CAP + ph]; bk.oo[s][i] = bp[2 * CAP + ph]; bk.ts[s][i] = bp[3 * CAP + ph];
        }
    }
}
__device__ __forceinline__ void store_book(int32_t* bp, const Book& bk, const Mkt& m, int lane) {
    #pragma unroll
    for (int s = 0; s < 2; s++) {
        for (int i = lane; i < mkt_n(m, s); i += WAVE) {
            const int ph = book_phys(s, i);
            bp[0 * CAP + ph] = bk.price[s][i]; bp[1 * CAP + ph] = bk.qty[s][i]; bp[2 * CAP + ph] = bk.oo[s][i]; bp[3 * CAP + ph] = bk.ts[s][i];
        }
    }
}
__device__ __forceinline__ void copy_words(uint32_t* dst, const uint32_t* src, int nwords, int lane) {
    for (int i = lane; i < nwords; i += WAVE) dst[i] = src[i];
}

// ======================================================================================
// Order book primitives (OrderTree / OrderList, orderbook/ordertree.py, orderlist.py)
// ======================================================================================
// remove `cnt` entries starting at `idx` (shift the tail down)
__device__ __forceinline__ void book_remove(Book& bk, int s, int n, int idx, int cnt, int lane) {
    for (int base = idx - (idx % WAVE); base < n - cnt; base += WAVE) {
        int i = base + lane;
        bool mv = i >= idx && i < n - cnt;
        int p = 0, q = 0, o = 0, t = 0;
        if (mv) { p = bk.price[s][i + cnt]; q = bk.qty[s][i + cnt]; o = bk.oo[s][i + cnt]; t = bk.ts[s][i + cnt]; }
        CDA_WSYNC();
        if (mv) { bk.price[s][i] = p; bk.qty[s][i] = q; bk.oo[s][i] = o; bk.ts[s][i] = t; }
        CDA_WSYNC();
    }
}
// OrderTree.insert_order (ordertree.py:44-58): tail of its price level.  false = side full.
__device__ __forceinline__ bool book_insert(Book& bk, int s, int n, int n_other, int price, int qty, int owner, int oid, int ts, int lane) {
    if (n + n_other >= CAP) return false;             // the pool is shared by both sides
    int pos = 0;
    for (int base = 0; base < n; base += WAVE) {
        int i = base + lane;
        bool c = false;
        if (i < n) { int rp = bk.price[s][i]; c = (s == S_BID) ? (rp >= price) : (rp <= price); }
        pos += __popcll(__ballot(c));
    }
    for (int base = n - (n % WAVE); base >= 0; base -= WAVE) {
        int i = base + lane;
        bool mv = i > pos && i <= n;
        int p = 0, q = 0, o = 0, t = 0;
        if (mv) { p = bk.price[s][i - 1]; q = bk.qty[s][i - 1]; o = bk.oo[s][i - 1]; t = bk.ts[s][i - 1]; }
        CDA_WSYNC();
        if (mv) { bk.price[s][i] = p; bk.qty[s][i] = q; bk.oo[s][i] = o; bk.ts[s][i] = t; }
        CDA_WSYNC();
    }
    // every lane writes the same values to the same slot (keeps each lane's view coherent)
    bk.price[s][pos] = price; bk.qty[s][pos] = qty; bk.oo[s][pos] = oo_pack(oid, owner); bk.ts[s][pos] = ts;
    CDA_WSYNC();
    return true;
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
// Trader._get_order_ID (agent/trader.py:254-287): index on the side or -1
__device__ __forceinline__ int find_own_order(const Book& bk, int s, int n, int tr, int type, int price, int lane) {
    if (type == T_MODIFY) {                 // oldest own order: minimum timestamp (LOB.time is unique per resting order)
        uint32_t best = 0xffffffffu;
        for (int i = lane; i < n; i += WAVE)
            if (oo_owner(bk.oo[s][i]) == tr) { uint32_t ts = (uint32_t)bk.ts[s][i]; best = ts < best ? ts : best; }
        const uint32_t mn = wave_min_u32(best);
        if (mn == 0xffffffffu) return -1;
        for (int base = 0; base < n; base += WAVE) {
            int i = base + lane;
            uint64_t mk = __ballot(i < n && oo_owner(bk.oo[s][i]) == tr && (uint32_t)bk.ts[s][i] == mn);
            if (mk) return base + (__ffsll((long long)mk) - 1);
        }
        return -1;
    }
    // limit / cancel: first own order at that price in order_map insertion order == first in the
    // level's FIFO, i.e. first hit in queue order
    for (int base = 0; base < n; base += WAVE) {
        int i = base + lane;
        bool c = i < n && oo_owner(bk.oo[s][i]) == tr && bk.price[s][i] == price;
        uint64_t mk = __ballot(c);
        if (mk) return base + (__ffsll((long long)mk) - 1);
    }
    return -1;
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

// Memory-to-memory style on purpose: every statement loads its operands from the account record in LDS and
// stores its result back, so that almost nothing is live across the (rare) calls into the rounding helpers.
// Keeping the whole account in registers made this function need 165 VGPRs (callee-saved registers are sparse
// in the AMDGPU calling convention) and capped the kernel at 3-4 waves per SIMD.
#define ACC_UPD(field, expr) do { D _r = (expr); st_dec(a.field, _r, flags); } while (0)
__device__ __forceinline__ void xfer_inc(Acc& a, bool counter, D v, uint32_t& flags) {   // size_increase_cash_transfer (cash_processor.py:31-36)
    cda_dec& fld = counter ? a.hold : a.cash;
    D r = d_sub(ld_dec(fld), v);
    st_dec(fld, r, flags);
}
__device__ __forceinline__ void xfer_dec(Acc& a, bool counter, D v, uint32_t& flags) {   // size_decrease_cash_transfer (cash_processor.py:38-45)
    ACC_UPD(cash, d_add(ld_dec(a.cash), v));
    if (counter) { ACC_UPD(hold, d_sub(ld_dec(a.hold), v)); ACC_UPD(cash, d_add(ld_dec(a.cash), v)); }
}
// position_val = raw + profit with raw = n * VWAP, mkt = n * price (account.py:128-131, :141-143, :155-157)
__device__ __forceinline__ D posval_from(Acc& a, uint32_t n, int32_t price, bool is_long) {
    D raw = d_mul_int(ld_dec(a.vwap), n), mkt = d_mul_u32(d_price(price), n, 0);
    return d_add(raw, cal_profit(is_long, mkt, raw));
}
// One fill (trader.py:303-345 _process_trades / _process_counter_party, account.py:215-231 process_acc): the passive
// party and the initiator settle at the same time, each on its owner lane plus helper lanes (+16, +32).  The ledger
// update of every mode is laid out as the SAME sequence of generic operations so that both parties - whatever their
// modes - run them in one instruction stream:
//   stage 1  one multiplication   owner: VWAP * |pos|  (numerator term of _size_increase/_size_decrease, `raw` of _covered)
//   stage 2  one addition         owner: numerator +- trade value | profit = +-(mkt - raw) | position_val += trade value (_neutral)
//                                 helpers: the cash / cash_on_hold transfers that do not depend on the owner's chain
//   stage 3  owner only           VWAP = numerator / n   or the rest of _covered / _covered_side_chg
// Operations on ONE field keep the reference's order; different fields of an account are independent.
//
// LAZY_POSVAL: inside k_step every fill is followed, before anything can read it, by Calculate.mark_to_mkt (it runs
// whenever the tape is non-empty), which recomputes position_val from (net_position, VWAP, price) alone.  The value
// _size_increase/_size_decrease store is therefore dead there and its four decimal operations are skipped; the paths
// that READ position_val (_neutral, _covered) are untouched.  The one-order test hook keeps the full semantics.
template <bool LAZY_POSVAL>
__device__ __forceinline__ void settle_fill(Lds& L, int tr, int counter, int32_t q, int32_t price, int init_side, uint32_t& flags, int lane CDA_TF_PARAM) {
    uint32_t f = 0;
    TF_BEGIN();
    const int al = lane_acc(lane), g = lane_grp(lane);
    if (counter != tr) {
        if ((al == tr || al == counter) && g < 3) {
            Acc& a = L.acc[al];
            const bool is_counter = al == counter;
            const int own_side = is_counter ? init_side ^ 1 : init_side;
            const int32_t pos = a.net_position;
            const uint32_t ap = (uint32_t)(pos < 0 ? -pos : pos);
            const bool is_long = pos > 0;
            int mode;                                       // 0 _neutral, 1 _size_increase, 2 _size_decrease, 3 _covered, 4 _covered_side_chg
            if (pos == 0) mode = 0;
            else if (is_long == (own_side == S_BID)) mode = 1;
            else mode = ap > (uint32_t)q ? 2 : (ap == (uint32_t)q ? 3 : 4);
            const D tv = d_mul_u32(d_price(price), (uint32_t)q, 0);             // trade value
            D X = d_zero(), mkt = d_zero();
            TF_END(14);
            if (g == 0 && mode != 0) X = d_mul_int(ld_dec(a.vwap), ap);         // stage 1
            TF_END(15);
            D x2 = d_zero(), y2 = d_zero();
            cda_dec* dst = nullptr;
            bool s2 = false;
            if (g == 0) {
                s2 = true;
                if (mode == 0) { x2 = ld_dec(a.posval); y2 = tv; dst = &a.posval; }                 // account.py:173-176
                else if (mode <= 2) { x2 = X; y2 = tv; y2.sign = mode == 2 ? 1 : 0; }               // account.py:124-133, :151-161
                else { mkt = d_mul_u32(d_price(price), ap, 0); x2 = is_long ? mkt : X; y2 = d_neg(is_long ? X : mkt); }   // calculate.py cal_profit
            } else if (mode <= 2) {
                if (g == 1) {                               // first statement of size_increase / size_decrease_cash_transfer (cash_processor.py:31-45)
                    s2 = true;
                    dst = (mode != 2 && is_counter) ? &a.hold : &a.cash;
                    x2 = ld_dec(*dst); y2 = tv; y2.sign = mode != 2 ? 1 : 0;
                } else if (mode == 2 && is_counter) {       // the passive party's escrow release
                    s2 = true;
                    dst = &a.hold; x2 = ld_dec(a.hold); y2 = d_neg(tv);
                }
            }
            D Y = d_zero();
            TF_END(16);
            if (s2) { Y = d_add(x2, y2); if (dst) st_dec(*dst, Y, f); }        // stage 2
            TF_END(17);
            if (g == 0) {                                    // stage 3
                a.num_trades += 1; a.num_trades_step += 1; if (is_counter) a.num_passive_fills_step += 1;
                if (mode == 0) st_dec(a.vwap, d_price(price), f);
                else if (mode <= 2) {
                    const uint32_t n = mode == 1 ? ap + (uint32_t)q : ap - (uint32_t)q;
                    ACC_UPD(vwap, d_div_u32(Y, n));
                    if (!LAZY_POSVAL) ACC_UPD(posval, posval_from(a, n, price, is_long));
                    if (mode == 2 && is_counter) ACC_UPD(cash, d_add(ld_dec(a.cash), tv));       // third statement of size_decrease_cash_transfer
                } else {                                     // _covered (account.py:135-149)
                    D pv = d_add(X, Y);
                    ACC_UPD(cash, d_add(ld_dec(a.cash), d_sub(pv, mkt)));                        // size_zero_cash_transfer (cash_processor.py:47-53)
                    st_dec(a.posval, d_zero(), f); st_dec(a.vwap, d_zero(), f);
                    if (mode == 3) xfer_dec(a, is_counter, tv, f);
                    else {                                   // _covered_side_chg (account.py:163-171)
                        xfer_dec(a, is_counter, mkt, f);
                        D npv = d_mul_u32(d_price(price), (uint32_t)q - ap, 0);
                        st_dec(a.posval, npv, f); st_dec(a.vwap, d_price(price), f);
                        xfer_inc(a, is_counter, npv, f);
                    }
                }
                int64_t np = (int64_t)pos + (own_side == S_BID ? (int64_t)q : -(int64_t)q);
                if (np > 2147483647LL || np < -2147483647LL) f |= CDA_FLAG_INT_OVERFLOW << 8;
                a.net_position = (int32_t)np;
            }
            TF_END(18); TF_COUNT(20);
        }
        CDA_WSYNC();
        TF_RESYNC();
    } else if (lane == tr) {                             // init_is_counter_cash_transfer (cash_processor.py:55-62)
        Acc& a = L.acc[lane];
        D tv = d_mul_u32(d_price(price), (uint32_t)q, 0);
        D hold = d_sub(ld_dec(a.hold), tv), cash = d_add(ld_dec(a.cash), tv);
        st_dec(a.hold, hold, f); st_dec(a.cash, cash, f);
    }
    if (__ballot((f & 0xffu) != 0)) flags |= CDA_FLAG_DEC_DOMAIN;
    if (__ballot((f >> 8) != 0)) flags |= CDA_FLAG_INT_OVERFLOW;
    TF_END(19);
}

// matching loops of OrderBook.process_order_list / process_market_order / process_limit_order
// (orderbook.py:61-194).  limit < 0 = market order.  Returns the unfilled quantity.
template <bool LAZY_POSVAL>
__device__ __forceinline__ int32_t match(Lds& L, Mkt& m, int tr, int own_side, int32_t qty, int32_t limit, int lane) {
    int opp = own_side ^ 1;
    int h = 0, nopp = mkt_n(m, opp);
    Book& bk = L.book;
    while (qty > 0 && h < nopp) {
        int32_t p = bk.price[opp][h];
        if (limit >= 0) { if (own_side == S_BID ? !(limit >= p) : !(limit <= p)) break; }
        int32_t rq = bk.qty[opp][h], c = oo_owner(bk.oo[opp][h]), f;
        if (qty < rq) { f = qty; bk.qty[opp][h] = rq - qty; qty = 0; CDA_WSYNC(); }    // all lanes store the same value
        else { f = rq; qty -= rq; h++; }
        m.has_trade = 1; m.last_trade_price = p;
        TACC_COUNT(m, 5, 1);
        // The launch ends with its slowest market-wave, and the slow ones are those that settle many fills: from the first
        // fill on, this wave asks its SIMD for instruction-issue priority over the (lighter) waves it shares the SIMD with.
        m.fills += 1;
        if (m.fills == 1) __builtin_amdgcn_s_setprio(1);
        else if (m.fills == 2) __builtin_amdgcn_s_setprio(2);
        else if (m.fills == 3) __builtin_amdgcn_s_setprio(3);
        settle_fill<LAZY_POSVAL>(L, tr, c, f, p, own_side, m.flags, lane CDA_TF_ARG(m));
    }
    if (h) { book_remove(bk, opp, nopp, 0, h, lane); mkt_set_n(m, opp, nopp - h); }
    return qty;
}

// cash -+= v and hold +-= v of account `tr` (cash_processor.py:15-29 order_in_book_passive_party with dir = -1: escrow a
// resting order; :85-97 cancel_cash_transfer with dir = +1: release it).  Group 0 updates cash, group 1 cash_on_hold.
__device__ __forceinline__ void cash_hold_transfer(Lds& L, int tr, int32_t price, int32_t qty, int cash_dir, uint32_t& flags, int lane) {
    const int g = lane_grp(lane);
    if (lane_acc(lane) == tr && g < 2) {
        Acc& a = L.acc[tr];
        D v = d_mul_u32(d_price(price), (uint32_t)qty, 0);
        cda_dec& fld = g == 0 ? a.cash : a.hold;
        v.sign = (g == 0) == (cash_dir > 0) ? 0 : 1;
        const D cur = ld_dec(fld);
        D r = d_add_order_value(cur, v);
        if (r.exp == D_NOT_HANDLED) r = d_add(cur, v);
        st_dec(fld, r, flags);
    }
    CDA_WSYNC();
}

// Trader._order_approved (agent/trader.py:108-151), evaluated by lane `tr`, result broadcast
__device__ __forceinline__ bool order_approved(Lds& L, const Mkt& m, int tr, int side, int32_t size, int32_t price, int lane) {
    int ok = 0;
    if (lane == tr) {
        const Acc& a = L.acc[lane];
        // every LDS operand is requested up front (one round trip instead of four dependent ones)
        const D nav = ld_dec(a.nav), cash = ld_dec(a.cash);
        const int32_t pos32 = a.net_position;
        const int32_t opp_best = L.book.price[side ^ 1][0];              // read even if that side is empty (slot 0 / CAP-1 exists)
        if (d_sgn(nav) > 0) {
            int64_t pos = pos32, opening;
            int64_t apos = pos < 0 ? -pos : pos;
            if ((side == S_BID && pos >= 0) || (side == S_ASK && pos <= 0)) opening = size;
            else { opening = (int64_t)size - apos; if (opening < 0) opening = 0; }
            if (opening <= 0) ok = 1;
            else {
                D est;
                if (price < 0) {
                    int opp = side ^ 1;
                    if (mkt_n(m, opp) > 0) est = d_price(opp_best);
                    else if (m.has_trade) est = d_price(m.last_trade_price);
                    else est = d_from_u32(1);
                } else est = d_price(price);
                // order_val = opening * est = ov * 10^-1 with ov < 2^63 (est is a tick price or 1: coefficient < 2^28).  When the
                // bit lengths alone prove cash > order_val the exact decimal compare is skipped (cash of 1e6 vs orders of 1e2..1e5).
                uint64_t ov = (uint64_t)((uint32_t)est.w0) * (uint64_t)opening;          // exact: est.w1 == est.w2 == 0
                int kdig = est.exp - cash.exp;                                            // cash coefficient is compared with ov * 10^kdig
                int bc = bits128(d_c128(cash)), bo = 64 - __clzll(ov | 1ull);
                if (!cash.sign && kdig >= 0 && kdig <= 30 && opening < (1LL << 31) && bc - 1 >= bo + ((kdig * 3402) >> 10) + 1) ok = 1;
                else ok = d_cmp(cash, d_mul_u32(est, (uint32_t)opening, 0)) >= 0;
            }
        }
    }
    return __ballot(ok != 0) != 0;                       // only lane `tr` can have set it
}

// Trader.place_order (agent/trader.py:49-106) with Trader._place_limit_order / _modify_limit_order /
// __modify_limit_order / _cancel_limit_order (:189-252) and OrderBook.process_order / modify_order /
// cancel_order (orderbook/orderbook.py:33-59, :196-266).  Structured so that the matching loop has ONE
// call site: the type-specific part only decides what (if anything) is matched and what may rest.
template <bool LAZY_POSVAL>
__device__ __forceinline__ void place_order(Lds& L, Mkt& m, int tr, int type, int side, int32_t size, int32_t price, int lane) {
    if (side == S_NONE) return;
    TACC_BEGIN();
    bool approved_ = order_approved(L, m, tr, side, size, type == T_MARKET ? -1 : price, lane);
    TACC_END(m, 0);
    if (!approved_) {
        if (lane == tr) L.acc[lane].num_rejected_step += 1;
        return;
    }
    if ((type == T_MARKET || type == T_LIMIT) && lane == tr) L.acc[lane].order_step_placed = 1;
    Book& bk = L.book;
    int32_t rest_price = 0, rest_qty = 0;
    bool do_match = false, can_rest = false;
    int32_t m_limit = -1, rest_oid = 0;
    uint32_t f = 0;
    if (type == T_MARKET) {
        m.lob_time += 1; m.next_oid += 1;                     // orderbook.py:39-44
        do_match = true;
    } else {
        const int nside = mkt_n(m, side);
        int idx = find_own_order(bk, side, nside, tr, type, price, lane);
        TACC_END(m, 1);
        if (type == T_LIMIT && idx < 0) {                     // a new order
            m.lob_time += 1; m.next_oid += 1;
            do_match = true; can_rest = true; m_limit = price; rest_oid = m.next_oid;
        } else if (idx >= 0) {
            int32_t op = bk.price[side][idx], oq = bk.qty[side][idx];
            int32_t ooid = (int32_t)((uint32_t)bk.oo[side][idx] >> 4);
            m.lob_time += 1;
            TACC_END(m, 4);
            if (type == T_CANCEL) {                           // trader.py:237-252: cancel, then release the escrow
                book_remove(bk, side, nside, idx, 1, lane); mkt_set_n(m, side, nside - 1);
                TACC_END(m, 6);
                cash_hold_transfer(L, tr, op, oq, +1, f, lane);
                TACC_END(m, 7);
            } else {                                          // upsert / modify: release, then modify_order
                cash_hold_transfer(L, tr, op, oq, +1, f, lane);
                TACC_END(m, 7);
                TACC_COUNT(m, 10, 1);
                if (price == op && size <= oq) {              // in place: priority kept, timestamp := now
                    bk.qty[side][idx] = size; bk.ts[side][idx] = m.lob_time;
                    CDA_WSYNC();
                    rest_price = price; rest_qty = size;
                } else {                                      // remove and re-process with the same order id
                    book_remove(bk, side, nside, idx, 1, lane); mkt_set_n(m, side, nside - 1);
                    do_match = true; can_rest = true; m_limit = price; rest_oid = ooid;
                }
                TACC_END(m, 6);
            }
        }
    }
    if (m.next_oid >= (1 << 27)) m.flags |= CDA_FLAG_INT_OVERFLOW;
    TACC_END(m, 4);
    if (do_match) {
        int32_t left = match<LAZY_POSVAL>(L, m, tr, side, size, m_limit, lane);
        TACC_END(m, 2);
        if (left > 0 && can_rest) {
            const int nown = mkt_n(m, side);
            if (book_insert(bk, side, nown, mkt_n(m, side ^ 1), price, left, tr, rest_oid, m.lob_time, lane)) { mkt_set_n(m, side, nown + 1); rest_price = price; rest_qty = left; m.peak_orders = max(m.peak_orders, m.nb + m.na); }
            else m.flags |= CDA_FLAG_BOOK_OVERFLOW;
        }
        TACC_END(m, 3);
    }
    TACC_END(m, 4);
    if (rest_qty > 0) { cash_hold_transfer(L, tr, rest_price, rest_qty, -1, f, lane); TACC_COUNT(m, 11, 1); }
    TACC_END(m, 8);
    if (__ballot(f != 0)) m.flags |= CDA_FLAG_DEC_DOMAIN;
}

// Exchg_Helper.mark_to_mkt + Calculate.mark_to_mkt (exchg_helper.py:56-66, calculate.py:35-55)
__device__ __forceinline__ void mark_to_mkt(Lds& L, Mkt& m, int A, int lane) {
    if (!m.has_trade) return;
    m.last_price = m.last_trade_price;
    uint32_t f = 0;
    // Three independent chains per account run on the owner lane and its two helpers in the same instruction stream:
    //   group 0: diff = +-(p - VWAP), profit = diff * |pos|      group 1: raw = VWAP * |pos|      group 2: cash + cash_on_hold
    // then the owner finishes position_val = raw + profit, nav = (cash + hold) + position_val: 4 dependent operations, not 6.
    const int g = lane_grp(lane), al = lane_acc(lane);
    const bool act = al < A && g < 3;
    D r1 = d_zero(), r2 = d_zero();
    uint32_t ap = 0;
    if (act) {
        const Acc& a = L.acc[al];
        int32_t pos = a.net_position;
        ap = (uint32_t)(pos < 0 ? -pos : pos);
        D vwap = ld_dec(a.vwap);
        if (g != 1) {                                         // stage 1, one addition: g0 p - VWAP (or VWAP - p), g2 cash + hold
            D x, y;
            if (g == 0) { D p = d_price(m.last_trade_price); x = pos >= 0 ? p : vwap; y = d_neg(pos >= 0 ? vwap : p); }
            else { x = ld_dec(a.cash); y = ld_dec(a.hold); }
            r1 = d_add(x, y);
        }
        if (g != 2) r2 = d_mul_int(g == 0 ? r1 : vwap, ap);   // stage 2, one multiplication: g0 profit, g1 raw
    }
    D raw = d_shfl(r2, (lane + 16) & 63), ssum = d_shfl(r1, (lane + 32) & 63);
    if (act && g == 0) {
        Acc& a = L.acc[al];
        D posval = d_add(raw, r2);
        D nav = d_add(ssum, posval);
        a.prev_nav = a.nav;
        st_dec(a.posval, posval, f); st_dec(a.nav, nav, f);
        if (d_cmp(nav, ld_dec(a.max_nav)) > 0) st_dec(a.max_nav, nav, f);
    }
    CDA_WSYNC();
    if (__ballot(f != 0)) m.flags |= CDA_FLAG_DEC_DOMAIN;
}

// ======================================================================================
// Observation (State_Helper.set_agg_LOB, exchg/state_helper.py:113-214)
// ======================================================================================
// top-K aggregation per side -> L.lvl_px / L.lvl_sz (integers; 0 = empty level)
__device__ __forceinline__ void aggregate_levels(Lds& L, const Mkt& m, int lane) {
    if (lane < 2 * CDA_K_ROWS) { (&L.lvl_px[0][0])[lane] = 0; (&L.lvl_sz[0][0])[lane] = 0; }
    CDA_WSYNC();
    #pragma unroll
    for (int s = 0; s < 2; s++) {
        int carry = 0, n = mkt_n(m, s);
        for (int base = 0; base < n && carry <= CDA_K_ROWS; base += WAVE) {
            int i = base + lane;
            bool valid = i < n;
            int p = valid ? L.book.price[s][i] : 0;
            int pp = (valid && i > 0) ? L.book.price[s][i - 1] : -1;
            bool head = valid && (i == 0 || p != pp);
            uint64_t mk = __ballot(head);
            uint64_t le = lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull);
            int lvl = carry + __popcll(mk & le) - 1;
            if (valid && lvl < CDA_K_ROWS) {
                atomicAdd(&L.lvl_sz[s][lvl], L.book.qty[s][i]);
                if (head) L.lvl_px[s][lvl] = p;
            }
            carry += __popcll(mk);
        }
    }
    CDA_WSYNC();
}
// raw f32[40] = agg_LOB_raw (state_helper.py:159-160); empty ask levels stay +0.0
__device__ __forceinline__ float raw_value(const Lds& L, int j) {
    int row = j / CDA_K_ROWS, k = j % CDA_K_ROWS;
    int v = row == 0 ? L.lvl_px[0][k] : row == 1 ? L.lvl_sz[0][k] : row == 2 ? L.lvl_px[1][k] : L.lvl_sz[1][k];
    float f = (float)v;
    return (row >= 2 && v != 0) ? -f : f;
}
// one normalised snapshot value for lane j in [0,42)
__device__ __forceinline__ float snapshot_value(const Lds& L, const Mkt& m, int tick, int j) {
    double l1_bid = (double)L.lvl_px[0][0], l1_ask = (double)L.lvl_px[1][0], M;
    bool two = l1_bid > 0 && l1_ask > 0;
    if (two) M = (l1_bid + l1_ask) / 2.0;
    else if (l1_bid > 0) M = l1_bid;
    else if (l1_ask > 0) M = l1_ask;
    else { M = (double)m.last_price; if (M <= 0) M = 100.0; }
    double out;
    if (j < 40) {
        // rows 0 / 2 (bid / ask price distance) share one division, rows 1 / 3 (sizes) one square root: the 20 + 20 lanes of
        // a row pair run ONE instruction sequence instead of two
        const int row = j / CDA_K_ROWS, k = j % CDA_K_ROWS, side = row >> 1;
        const double raw = (double)((row & 1) ? L.lvl_sz[side][k] : L.lvl_px[side][k]);
        if (raw == 0.0) out = 0.0;                                     // empty level (raw values are never negative)
        else if (row & 1) { const double r = sqrt(raw); out = side ? -r : r; }
        else { const double q = (side ? raw - M : M - raw) / M; out = side ? -q : q; }
    } else {
        // lanes 40 (log M) and 41 (log1p of the spread in ticks) share ONE evaluation: M is a half-integer, M - 1 is exact, and
        // float32(log1p(M - 1)) == float32(numpy.log(M)) == float32(libm log(M)) for EVERY M = k/2 with k <= 2^25, i.e. for every
        // mid of two prices below 2^24 ticks (the price clamp of step_market) - swept exhaustively by tools/sweep_libm.py, so
        // there is no separate log().  log1p(+0) = +0 covers the one-sided book.
        double arg = 0.0;
        if (j == 40) arg = M - 1.0;
        else if (two) { double st = (l1_ask - l1_bid) / (double)tick; arg = st > 0.0 ? st : 0.0; }
        out = glibc_log1p(arg);
    }
    return (float)out;
}

}  // namespace cda
