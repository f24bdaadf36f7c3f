// cda_dec.hpp - device-side 28-digit decimal ledger arithmetic for gfx950 (HIP).
//
// The reference keeps every account in Python `decimal.Decimal` (default context: prec 28,
// ROUND_HALF_EVEN; account/account.py:12-53).  Rewards, NAV strings and observations' downstream
// consumers see the 28-digit rounding noise, so the device ledger reproduces the General Decimal
// Arithmetic operations exactly: a value is (sign, coefficient < 10^28 < 2^94, exponent).
// There is no MFMA-shaped work here: it is carry chains, compares and constant-divisor divisions.
//
// Three tiers, chosen per operation from the operands' bit lengths:
//   1. inline, 128-bit, exact result already < 10^28: a handful of VALU instructions, no memory;
//   2. `*_mid` leaf helpers on 4 x u32 limbs (intermediates < 2^126): rounding to 28 digits with
//      compile-time-constant divisors and a power-of-ten table held in LDS;
//   3. `*_wide` helpers on 8 x u32 limbs (<= 58 digits): the fully general path, rarely taken.
//
// Call-site facts this file exploits: every multiplication has one operand that is a
// plain integer (< 2^32) or a tick price, and every division is by an integer position size.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned __int128 u128;

namespace cda {

#ifdef CDA_DEC_COUNTERS
// debug builds: how often each out-of-line decimal routine runs (tools/phase_timing.py --counters prints them per market-step)
__device__ unsigned long long g_dec_calls[8];
#define DEC_COUNT(i) atomicAdd(&g_dec_calls[i], 1ull)
#else
#define DEC_COUNT(i) do {} while (0)
#endif

template <int NL> struct WN { uint32_t w[NL]; };   // unsigned, NL x 32 bits, little endian
typedef WN<8> W;
typedef WN<4> W4;

struct D {                            // a Decimal in registers
    uint32_t w0, w1, w2;
    int32_t exp;
    int32_t sign;
};

// ---- compile-time tables ----------------------------------------------------------------
struct Pow10Tab { uint32_t v[60][8]; };
constexpr Pow10Tab make_pow10() {
    Pow10Tab t{};
    t.v[0][0] = 1;
    for (int k = 1; k < 60; k++) {
        uint64_t c = 0;
        for (int i = 0; i < 8; i++) { c += (uint64_t)t.v[k - 1][i] * 10u; t.v[k][i] = (uint32_t)c; c >>= 32; }
    }
    return t;
}
struct Pow5Tab { uint64_t lo[56], hi[56]; };
constexpr Pow5Tab make_pow5() {
    Pow5Tab t{};
    u128 v = 1;
    for (int k = 0; k < 56; k++) { t.lo[k] = (uint64_t)v; t.hi[k] = (uint64_t)(v >> 64); v *= 5; }
    return t;
}
struct Rcp10Tab { double v[10]; };
constexpr Rcp10Tab make_rcp10() {
    Rcp10Tab t{};
    double p = 1.0;
    for (int k = 0; k < 10; k++) { t.v[k] = 1.0 / p; p *= 10.0; }          // 10^k exact in binary64; one correctly rounded division each
    return t;
}
__device__ const Rcp10Tab RCP10 = make_rcp10();
__device__ const Pow10Tab POW10 = make_pow10();     // global memory: only the wide path reads it
__device__ const Pow5Tab POW5 = make_pow5();

// LDS copy of 10^0 .. 10^38 as 4 limbs, at the start of the workgroup's dynamic LDS; filled by
// dec_tables_init() at kernel entry.
constexpr int DEC_LDS_POW = 39;
constexpr int DEC_RCP_OFF = 640;                    // 39 * 16 = 624, padded; then RN(1 / 10^k), k = 0..9, as doubles
constexpr int DEC_TABLE_BYTES = 736;                // 640 + 10 * 8 = 720, padded
extern __shared__ __attribute__((aligned(16))) unsigned char cda_smem[];
// The tables are read through ABSOLUTE LDS addresses.  None of these kernels has static __shared__ data, so the dynamic
// segment - and with it the table - starts at LDS address 0 (dec_tables_init() traps if that ever stops being true).
// Going through the `cda_smem` symbol instead costs every out-of-line routine a lookup of the kernel's dynamic-LDS
// offset in a table in memory (s_getpc + s_load_dword + s_waitcnt lgkmcnt(0)) on each call: two per d_round_mid.
typedef const __attribute__((address_space(3))) uint32_t* lds_u32p;
typedef const __attribute__((address_space(3))) double* lds_f64p;
__device__ __forceinline__ lds_u32p lds_pow10(int k) { return reinterpret_cast<lds_u32p>((uintptr_t)(16u * (uint32_t)k)); }
__device__ __forceinline__ double lds_rcp10(int k) { return *reinterpret_cast<lds_f64p>((uintptr_t)((uint32_t)DEC_RCP_OFF + 8u * (uint32_t)k)); }   // k <= 9
// every thread of the workgroup must call this once, before any decimal operation
__device__ __forceinline__ void dec_tables_init() {
    if ((uint32_t)(uintptr_t)cda_smem != 0u) __builtin_trap();          // see lds_pow10()
    uint32_t* t = reinterpret_cast<uint32_t*>(cda_smem);
    for (int i = (int)threadIdx.x; i < DEC_LDS_POW * 4; i += (int)blockDim.x) t[i] = POW10.v[i >> 2][i & 3];
    if (threadIdx.x < 10) reinterpret_cast<double*>(cda_smem + DEC_RCP_OFF)[threadIdx.x] = RCP10.v[threadIdx.x];      // RN(1 / 10^k), folded at compile time
    __syncthreads();
}

// ---- wide integer helpers (templated on the limb count) ------------------------------------
template <int NL> __device__ __forceinline__ WN<NL> w_from3(uint32_t a, uint32_t b, uint32_t c) {
    WN<NL> r;
    #pragma unroll
    for (int i = 0; i < NL; i++) r.w[i] = 0;
    r.w[0] = a; r.w[1] = b; r.w[2] = c;
    return r;
}
template <int NL> __device__ __forceinline__ bool w_is_zero(const WN<NL>& a) {
    uint32_t o = 0;
    #pragma unroll
    for (int i = 0; i < NL; i++) o |= a.w[i];
    return o == 0;
}
template <int NL> __device__ __forceinline__ int w_cmp(const WN<NL>& a, const WN<NL>& b) {
    int r = 0;
    #pragma unroll
    for (int i = 0; i < NL; i++) { if (a.w[i] != b.w[i]) r = a.w[i] < b.w[i] ? -1 : 1; }   // highest differing limb wins
    return r;
}
template <int NL> __device__ __forceinline__ WN<NL> w_add(const WN<NL>& a, const WN<NL>& b) {
    WN<NL> r; uint64_t c = 0;
    #pragma unroll
    for (int i = 0; i < NL; i++) { c += (uint64_t)a.w[i] + b.w[i]; r.w[i] = (uint32_t)c; c >>= 32; }
    return r;
}
template <int NL> __device__ __forceinline__ WN<NL> w_sub(const WN<NL>& a, const WN<NL>& b) {   // a >= b
    WN<NL> r; int64_t c = 0;
    #pragma unroll
    for (int i = 0; i < NL; i++) { c += (int64_t)a.w[i] - (int64_t)b.w[i]; r.w[i] = (uint32_t)c; c >>= 32; }
    return r;
}
template <int NL> __device__ __forceinline__ void w_mul_small(WN<NL>& x, uint32_t m) {
    uint64_t c = 0;
    #pragma unroll
    for (int i = 0; i < NL; i++) { c += (uint64_t)x.w[i] * m; x.w[i] = (uint32_t)c; c >>= 32; }
}
__device__ __forceinline__ uint32_t pow10_sel(int r) {   // 10^r, r in 0..9, as a select chain (no memory)
    uint32_t v = 1u;
    v = r == 1 ? 10u : v; v = r == 2 ? 100u : v; v = r == 3 ? 1000u : v; v = r == 4 ? 10000u : v; v = r == 5 ? 100000u : v;
    v = r == 6 ? 1000000u : v; v = r == 7 ? 10000000u : v; v = r == 8 ? 100000000u : v; v = r == 9 ? 1000000000u : v;
    return v;
}
template <int NL> __device__ __forceinline__ void w_mul_pow10(WN<NL>& x, int k) {
    while (k >= 9) { w_mul_small(x, 1000000000u); k -= 9; }
    if (k > 0) w_mul_small(x, pow10_sel(k));
}
template <int NL> __device__ __forceinline__ void w_inc(WN<NL>& x) {
    uint64_t c = 1;
    #pragma unroll
    for (int i = 0; i < NL; i++) { c += x.w[i]; x.w[i] = (uint32_t)c; c >>= 32; }
}
// x /= DV (compile-time divisor -> multiply-high sequences), returns the remainder
template <uint32_t DV, int NL> __device__ __forceinline__ uint32_t w_divc(WN<NL>& x) {
    uint64_t rem = 0;
    #pragma unroll
    for (int i = NL - 1; i >= 0; i--) { uint64_t cur = (rem << 32) | x.w[i]; uint64_t q = cur / DV; x.w[i] = (uint32_t)q; rem = cur - q * DV; }
    return (uint32_t)rem;
}
template <int NL> __device__ __forceinline__ uint32_t w_div_pow10_small(WN<NL>& x, int r) {   // r in 1..9
    switch (r) {
        case 1: return w_divc<10u>(x);
        case 2: return w_divc<100u>(x);
        case 3: return w_divc<1000u>(x);
        case 4: return w_divc<10000u>(x);
        case 5: return w_divc<100000u>(x);
        case 6: return w_divc<1000000u>(x);
        case 7: return w_divc<10000000u>(x);
        case 8: return w_divc<100000000u>(x);
        default: return w_divc<1000000000u>(x);
    }
}
// x /= d for a run-time 32-bit divisor, returns the remainder.  One f64 reciprocal per call, then per limb a
// floating estimate of the 64/32 quotient corrected by an exact remainder test (the estimate is off by at most
// one: cur < d * 2^32 < 2^64 converts to f64 with relative error 2^-53, and so does the reciprocal).  The
// generic 64-bit integer division the compiler would emit costs ~70 instructions per limb.
template <int NL> __device__ __forceinline__ uint32_t w_div_u32(WN<NL>& x, uint32_t d) {
    const double rd = 1.0 / (double)d;
    uint64_t rem = 0;
    #pragma unroll
    for (int i = NL - 1; i >= 0; i--) {
        uint64_t cur = (rem << 32) | x.w[i];              // rem < d  =>  cur / d < 2^32
        uint64_t q = (uint64_t)((double)cur * rd);
        q = q > 0xffffffffull ? 0xffffffffull : q;
        int64_t r = (int64_t)(cur - q * d);
        if (r < 0) { q -= 1; r += d; }
        if (r >= (int64_t)d) { q += 1; r -= d; }
        x.w[i] = (uint32_t)q; rem = (uint64_t)r;
    }
    return (uint32_t)rem;
}
// The same for a divisor below 2^30 with its reciprocal rd = RN(1/d) at hand: the quotient digit is one saturating
// f64 -> u32 conversion and the remainder test fits signed 32-bit arithmetic (the true remainder of the estimate lies
// in (-d, 2d)), so a limb costs about a dozen instructions.
template <int NL> __device__ __forceinline__ uint32_t w_div_small(WN<NL>& x, uint32_t d, double rd) {
    uint32_t rem = 0;
    #pragma unroll
    for (int i = NL - 1; i >= 0; i--) {
        const double cur = __builtin_fma((double)rem, 4294967296.0, (double)x.w[i]);     // exact: < 2^62 with 30 + 32 bits... rounded to 53
        uint32_t q = (uint32_t)(cur * rd);                      // v_cvt_u32_f64 saturates at 2^32 - 1; off by at most one
        int32_t r = (int32_t)(x.w[i] - q * d);                  // (rem * 2^32 + limb - q * d) mod 2^32, as a signed value
        const bool lo = r < 0;
        q -= lo ? 1u : 0u; r += lo ? (int32_t)d : 0;
        const bool hi = r >= (int32_t)d;
        q += hi ? 1u : 0u; r -= hi ? (int32_t)d : 0;
        x.w[i] = q; rem = (uint32_t)r;
    }
    return rem;
}
// the same with the running remainder seeded by the caller (rem0 < d): continues a long division below its top limb
template <int NL> __device__ __forceinline__ uint32_t w_div_small_seeded(WN<NL>& x, uint32_t d, double rd, uint32_t rem0) {
    uint32_t rem = rem0;
    #pragma unroll
    for (int i = NL - 1; i >= 0; i--) {
        const double cur = __builtin_fma((double)rem, 4294967296.0, (double)x.w[i]);
        uint32_t q = (uint32_t)(cur * rd);
        int32_t r = (int32_t)(x.w[i] - q * d);
        const bool lo = r < 0;
        q -= lo ? 1u : 0u; r += lo ? (int32_t)d : 0;
        const bool hi = r >= (int32_t)d;
        q += hi ? 1u : 0u; r -= hi ? (int32_t)d : 0;
        x.w[i] = q; rem = (uint32_t)r;
    }
    return rem;
}
template <int NL> __device__ __forceinline__ int w_bits(const WN<NL>& a) {
    int b = 0;
    #pragma unroll
    for (int i = 0; i < NL; i++) if (a.w[i]) b = 32 * i + 32 - __clz(a.w[i]);
    return b;
}
template <int NL> __device__ __forceinline__ WN<NL> w_pow10(int k);
template <> __device__ __forceinline__ W w_pow10<8>(int k) {        // global table (wide path only)
    W r;
    #pragma unroll
    for (int i = 0; i < 8; i++) r.w[i] = POW10.v[k][i];
    return r;
}
template <> __device__ __forceinline__ W4 w_pow10<4>(int k) {       // LDS table, k <= 38
    const lds_u32p p = lds_pow10(k);
    W4 r; r.w[0] = p[0]; r.w[1] = p[1]; r.w[2] = p[2]; r.w[3] = p[3];
    return r;
}
// len(str(x)) (1 for zero): floor(bits*log10(2)) then one table compare
template <int NL> __device__ __forceinline__ int w_ndigits(const WN<NL>& x) {
    int b = w_bits(x);
    if (b == 0) return 1;
    int t = (b * 1233) >> 12;
    WN<NL> p = w_pow10<NL>(t);
    return t + (w_cmp(x, p) >= 0 ? 1 : 0);
}

// ---- Decimal ------------------------------------------------------------------------------
__device__ __forceinline__ D d_make(uint32_t w0, uint32_t w1, uint32_t w2, int exp, int sign) { D d; d.w0 = w0; d.w1 = w1; d.w2 = w2; d.exp = exp; d.sign = sign; return d; }
__device__ __forceinline__ D d_zero() { return d_make(0, 0, 0, 0, 0); }
__device__ __forceinline__ D d_from_i64(int64_t v) {
    uint64_t a = v < 0 ? (uint64_t)(-(v + 1)) + 1u : (uint64_t)v;
    return d_make((uint32_t)a, (uint32_t)(a >> 32), 0, 0, v < 0);
}
__device__ __forceinline__ D d_from_u32(uint32_t v) { return d_make(v, 0, 0, 0, 0); }
// a tick price as the book holds it: Decimal(str(float(p))) = Decimal('p.0') (orderbook.py:52,239)
__device__ __forceinline__ D d_price(int32_t p) { uint64_t a = (uint64_t)(uint32_t)p * 10u; return d_make((uint32_t)a, (uint32_t)(a >> 32), 0, -1, 0); }
// Decimal('p.0') * n for a tick price p < 2^24 and an integer n < 2^31 (order and trade values, orderbook.py:52,239 with
// trader.py:312): always exact - the coefficient 10 * p * n stays below 2^59 - so no generic 128-bit product, no 10^28 test
__device__ __forceinline__ D d_price_times(int32_t p, uint32_t n) {
    const uint64_t c = (uint64_t)(uint32_t)p * 10u * (uint64_t)n;
    return d_make((uint32_t)c, (uint32_t)(c >> 32), 0, -1, 0);
}
__device__ __forceinline__ bool d_is_zero(const D& a) { return (a.w0 | a.w1 | a.w2) == 0; }
template <int NL> __device__ __forceinline__ WN<NL> d_wide(const D& a) { return w_from3<NL>(a.w0, a.w1, a.w2); }
__device__ __forceinline__ D d_neg(D a) { a.sign ^= 1; return a; }
// sign of a compared with 0: -1, 0, 1
__device__ __forceinline__ int d_sgn(const D& a) { return d_is_zero(a) ? 0 : (a.sign ? -1 : 1); }

// Decimal._fix (_pydecimal.py:1661): round a wide coefficient to 28 digits, half-even
template <int NL> __device__ __forceinline__ D d_fix_impl(int sign, WN<NL> x, int exp) {
    if (w_is_zero(x)) return d_make(0, 0, 0, exp, sign);
    int nd = w_ndigits(x);
    if (nd > 28) {
        int drop = nd - 28;
        // one division by 10^drop when drop <= 9 (the usual case): the remainder decides half-even directly;
        // larger drops first peel 10^9 chunks whose remainders only feed the sticky bit
        bool sticky = false;
        int k = drop;
        while (k > 9) { sticky |= (w_divc<1000000000u>(x) != 0); k -= 9; }
        uint32_t rem = w_div_pow10_small(x, k);
        uint32_t half = 5u * pow10_sel(k - 1);
        bool up = rem > half || (rem == half && (sticky || (x.w[0] & 1u)));
        if (up) {
            w_inc(x);
            // 10^28 = 0x204fce5e_3e250261_10000000
            if (x.w[0] == 0x10000000u && x.w[1] == 0x3e250261u && x.w[2] == 0x204fce5eu) {
                x.w[0] = 0xe8000000u; x.w[1] = 0x9fd0803cu; x.w[2] = 0x033b2e3cu; drop += 1;   // 10^27
            }
        }
        exp += drop;
    }
    return d_make(x.w[0], x.w[1], x.w[2], exp, sign);
}
__device__ __noinline__ D d_fix_mid(int sign, W4 x, int exp) { DEC_COUNT(0); return d_fix_impl<4>(sign, x, exp); }
__device__ __noinline__ D d_fix_wide(int sign, W x, int exp) { DEC_COUNT(1); return d_fix_impl<8>(sign, x, exp); }

// ---- 128-bit helpers -----------------------------------------------------------------------
__device__ __forceinline__ u128 d_c128(const D& a) { return ((u128)a.w2 << 64) | ((u128)a.w1 << 32) | (u128)a.w0; }
__device__ __forceinline__ D d_from128(u128 c, int exp, int sign) { return d_make((uint32_t)c, (uint32_t)(c >> 32), (uint32_t)(c >> 64), exp, sign); }
__device__ __forceinline__ W4 w4_from128(u128 c) { W4 r; r.w[0] = (uint32_t)c; r.w[1] = (uint32_t)(c >> 32); r.w[2] = (uint32_t)(c >> 64); r.w[3] = (uint32_t)(c >> 96); return r; }
__device__ __forceinline__ u128 p28_128() { return ((u128)0x204fce5eULL << 64) | 0x3e25026110000000ULL; }   // 10^28
__device__ __forceinline__ int bits128(u128 c) {
    uint64_t hi = (uint64_t)(c >> 64), lo = (uint64_t)c;
    return hi ? 128 - __clzll(hi) : (lo ? 64 - __clzll(lo) : 0);
}
__device__ __forceinline__ u128 mul_pow10_128(u128 x, int k) {   // caller guarantees the result fits
    while (k >= 9) { x *= (u128)1000000000u; k -= 9; }
    return x * (u128)pow10_sel(k);
}
// does c * 10^k stay below 2^126 ?  (3402/1024 > log2(10))
__device__ __forceinline__ bool scale_fits128(u128 c, int k) { return bits128(c) + ((k * 3402) >> 10) + 1 <= 126; }

// Decimal._fix for a 128-bit coefficient r >= 10^28 (the caller has checked), r < 2^127: the digit count comes from the bit length
// and ONE table compare (len(str(r)) = t or t + 1 with t = floor(bits * log10 2)), so the number of digits to drop is exact and the
// quotient of the one long division by 10^drop has exactly 28 digits.  Everything the division needs for BOTH candidate counts is
// requested from the LDS table up front (one round trip, next to the compare's operand) and selected afterwards; the rounding is
// straight-line (selects, no branches): the lanes of a call - an owner and its helper lanes, each with its own operands - would
// take different arms anyway.
__device__ __noinline__ D d_round_mid(int sign, u128 r, int exp) {      // a LEAF: no calls, so no return-address spill
    DEC_COUNT(2);
    const int t = (bits128(r) * 1233) >> 12;                    // 28 .. 38
    const int d0 = t - 28, d0c = d0 > 9 ? 9 : d0, d1c = d0 + 1 > 9 ? 9 : d0 + 1;    // drop is d0 or d0 + 1 (clamped: valid table indices whatever happens)
    const lds_u32p pt = lds_pow10(t);
    const uint32_t t0 = pt[0], t1 = pt[1], t2 = pt[2], t3 = pt[3];
    const uint32_t pw_a = lds_pow10(d0c)[0], pw_b = lds_pow10(d1c)[0];
    const double rd_a = lds_rcp10(d0c), rd_b = lds_rcp10(d1c);
    const u128 pw_t = ((u128)t3 << 96) | ((u128)t2 << 64) | ((u128)t1 << 32) | (u128)t0;
    const bool more = r >= pw_t;                                // t + 1 digits
    int drop = d0 + (more ? 1 : 0);                             // 1 .. 11, exact
    exp += drop;
    W4 x = w4_from128(r);
    bool sticky = false;
    uint32_t pw = more ? pw_b : pw_a;
    double rd = more ? rd_b : rd_a;
    if (drop > 9) {                                             // rare (r >= 10^37): peel the excess over nine digits first
        const int pre = drop - 9;
        sticky = w_div_small(x, lds_pow10(pre)[0], lds_rcp10(pre)) != 0;
        drop = 9; pw = 1000000000u; rd = lds_rcp10(9);
    }
    // r < 10^(28 + drop), so the quotient is below 10^28 < 2^94: its top limb is zero and the top limb of r (< 10^drop, or the
    // division could not come out below 2^96) is simply the first partial remainder - three limb steps instead of four
    WN<3> q; q.w[0] = x.w[0]; q.w[1] = x.w[1]; q.w[2] = x.w[2];
    const uint32_t rem = w_div_small_seeded(q, pw, rd, x.w[3]);
    const uint32_t half = pw >> 1;                              // 5 * 10^(drop-1)
    const bool up = rem > half || (rem == half && (sticky || (q.w[0] & 1u) != 0));
    // q + up, and 10^28 (= 0x204fce5e_3e250261_10000000) becomes 10^27 (= 0x033b2e3c_9fd0803c_e8000000) one exponent up
    const uint64_t s0 = (uint64_t)q.w[0] + (up ? 1u : 0u);
    const uint64_t s1 = (uint64_t)q.w[1] + (s0 >> 32);
    uint32_t w0 = (uint32_t)s0, w1 = (uint32_t)s1, w2 = q.w[2] + (uint32_t)(s1 >> 32);
    const bool ten28 = w0 == 0x10000000u && w1 == 0x3e250261u && w2 == 0x204fce5eu;
    w0 = ten28 ? 0xe8000000u : w0; w1 = ten28 ? 0x9fd0803cu : w1; w2 = ten28 ? 0x033b2e3cu : w2;
    exp += ten28 ? 1 : 0;
    return d_make(w0, w1, w2, exp, sign);
}

// ---- addition: Decimal.__add__ (_pydecimal.py:1157) with _normalize (:5640) and _rescale (:2612) ----
__device__ __noinline__ D d_add_wide(D a, D b) {
    DEC_COUNT(3);
    int exp = a.exp < b.exp ? a.exp : b.exp;
    bool az = d_is_zero(a), bz = d_is_zero(b);
    if (az && bz) return d_make(0, 0, 0, exp, a.sign < b.sign ? a.sign : b.sign);
    if (az || bz) {
        D o = az ? b : a;
        int e = exp > o.exp - 29 ? exp : o.exp - 29;
        W x = d_wide<8>(o);
        w_mul_pow10(x, o.exp - e);
        return d_fix_wide(o.sign, x, e);
    }
    bool swp = a.exp < b.exp;
    D dt = swp ? b : a, dO = swp ? a : b;            // dt: larger exponent (ties: a)
    W xt = d_wide<8>(dt), xo = d_wide<8>(dO);
    int et = dt.exp, eo = dO.exp;
    if (et != eo) {
        int tmp_len = w_ndigits(xt), oth_len = w_ndigits(xo);
        int m = tmp_len - 30; if (m > -1) m = -1;
        int e = et + m;
        if (oth_len + eo - 1 < e) { xo = w_from3<8>(1, 0, 0); eo = e; }
        w_mul_pow10(xt, et - eo);
    }
    W r; int rs;
    if (dt.sign != dO.sign) {
        int c = w_cmp(xt, xo);
        if (c == 0) return d_make(0, 0, 0, exp, 0);
        if (c > 0) { r = w_sub(xt, xo); rs = dt.sign; } else { r = w_sub(xo, xt); rs = dO.sign; }
    } else { r = w_add(xt, xo); rs = dt.sign; }
    return d_fix_wide(rs, r, eo);
}
// 128-bit tier, out of line: any exponent gap whose scaled operand still fits 128 bits, with rounding
__device__ __noinline__ D d_add_mid(D a, D b) {
    DEC_COUNT(4);
    bool swp = a.exp < b.exp;
    D t = swp ? b : a, o = swp ? a : b;                     // t: larger exponent (ties: a); both non-zero
    int diff = t.exp - o.exp;
    u128 ct = d_c128(t), co = d_c128(o);
    bool ok = true;
    if (diff) {
        // _normalize replaces `other` by 10^e only when other.adjusted() < e = t.exp + min(-1, len-30) <= t.exp-1;
        // other.adjusted() >= o.exp + floor((bits-1)*log10 2) decides that from bit lengths alone.
        int lb_adj = o.exp + (((bits128(co) - 1) * 1233) >> 12);
        ok = scale_fits128(ct, diff) && lb_adj >= t.exp - 1;
    }
    if (!ok) return d_add_wide(a, b);
    if (diff) ct = mul_pow10_128(ct, diff);
    u128 r; int rs;
    if (t.sign != o.sign) {
        if (ct == co) return d_make(0, 0, 0, o.exp, 0);
        if (ct > co) { r = ct - co; rs = t.sign; } else { r = co - ct; rs = o.sign; }
    } else { r = ct + co; rs = t.sign; }
    if (r < p28_128()) return d_from128(r, o.exp, rs);
    return d_round_mid(rs, r, o.exp);                       // < 2^127
}
// u32 * 10^k (k <= 28) from the LDS table: < 2^32 * 10^28 < 2^126
__device__ __forceinline__ u128 mul_u32_pow10_lds(uint32_t c, int k) {
    const lds_u32p p = lds_pow10(k);
    u128 pw = ((u128)p[2] << 64) | ((u128)p[1] << 32) | (u128)p[0];      // 10^28 < 2^94: three limbs
    return pw * (u128)c;
}
// Inline tier: the shapes the ledger produces almost always - equal exponents; a short coefficient (an order value,
// a price: < 2^32) on the larger-exponent side that is scaled with ONE table multiply; or two long coefficients a few
// digits apart (position_val = raw + profit, nav = cash + position_val) scaled by a power of ten below 2^32.
__device__ __forceinline__ D d_add(D a, D b) {
    const bool az = d_is_zero(a), bz = d_is_zero(b);
    if (!az && !bz) {
        const bool swp = a.exp < b.exp;
        const D t = swp ? b : a, o = swp ? a : b;           // t: larger exponent (ties: a)
        const int diff = t.exp - o.exp;
        const u128 co = d_c128(o);
        u128 ct;
        bool ok;
        if (diff == 0) { ct = d_c128(t); ok = true; }
        else {
            ok = (((bits128(co) - 1) * 1233) >> 12) >= diff - 1;   // no _normalize replacement (see d_add_mid)
            if ((t.w1 | t.w2) == 0 && diff <= 28) ct = mul_u32_pow10_lds(t.w0, diff);
            else if (diff <= 9) ct = d_c128(t) * (u128)lds_pow10(diff)[0];       // < 2^94 * 2^30
            else { ok = false; ct = 0; }
        }
        if (ok) {
            u128 r; int rs;
            if (t.sign != o.sign) {
                if (ct == co) return d_make(0, 0, 0, o.exp, 0);
                if (ct > co) { r = ct - co; rs = t.sign; } else { r = co - ct; rs = o.sign; }
            } else { r = ct + co; rs = t.sign; }
            if (r < p28_128()) return d_from128(r, o.exp, rs);
            return d_round_mid(rs, r, o.exp);
        }
        return d_add_mid(a, b);
    }
    if (az != bz) {
        const D o = az ? b : a, z = az ? a : b;
        if (z.exp >= o.exp) return o;                       // rescale by 10^0: the non-zero operand unchanged
    }
    return d_add_wide(a, b);
}
__device__ __forceinline__ D d_sub(D a, D b) { return d_add(a, d_neg(b)); }
constexpr int D_NOT_HANDLED = -0x40000000;                      // exponent sentinel of the leaf fast paths
// field + v for a cash / cash_on_hold transfer, as a LEAF: `v` is an order value (coefficient below 2^32, exponent -1) and
// does not lie below `field`; one table multiply scales it (10^0 when the exponents are equal), so the two lanes of a
// transfer - a 28-digit cash, a short cash_on_hold - run the same instructions.  Anything that does not fit the shape, or
// would need rounding, is handed back (exp == D_NOT_HANDLED) to the general addition.
__device__ __noinline__ D d_add_order_value(D field, D v) {
    // Straight-line on purpose: sum and both differences are formed and the result SELECTED - every `if` on per-lane data
    // costs an exec-mask save / branch / restore, and the two lanes of a transfer (cash: 28 digits, hold: short) would take
    // different arms anyway.
    const int diff = v.exp - field.exp;
    const u128 co = d_c128(field);
    const bool fz = co == 0;
    const int dcl = diff < 0 ? 0 : (diff > 28 ? 28 : diff);                            // a valid table index whatever the shape
    const bool shape = (v.w1 | v.w2) == 0 && v.w0 != 0 && diff == dcl &&
                       (fz || (((bits128(co) - 1) * 1233) >> 12) >= diff - 1);           // no _normalize replacement (see d_add_mid)
    const u128 ct = mul_u32_pow10_lds(v.w0, dcl);
    const bool same = field.sign == v.sign || fz;                                       // a zero field only lends its exponent (Decimal.__add__, `if not self`)
    const bool vg = ct > co;
    const u128 sum = ct + co, dv = ct - co, df = co - ct;
    const u128 r = same ? sum : (vg ? dv : df);
    int rs = same ? (fz ? v.sign : field.sign) : (vg ? v.sign : field.sign);
    rs = r == 0 ? 0 : rs;                                                               // equal magnitudes, opposite signs: +0 at the field's exponent
    const bool ok = shape && r < p28_128();
    return d_make(ok ? (uint32_t)r : 0u, ok ? (uint32_t)(r >> 32) : 0u, ok ? (uint32_t)(r >> 64) : 0u, ok ? field.exp : D_NOT_HANDLED, ok ? rs : 0);
}

// ---- multiplication: Decimal.__mul__ (_pydecimal.py:1267) for b = (+) m * 10^mexp with m < 2^32 ----
__device__ __forceinline__ D d_mul_u32(D a, uint32_t m, int mexp) {
    int exp = a.exp + mexp;
    if (d_is_zero(a) || m == 0) return d_make(0, 0, 0, exp, a.sign);
    u128 p = d_c128(a) * (u128)m;                           // < 2^94 * 2^32 = 2^126
    if (p < p28_128()) return d_from128(p, exp, a.sign);
    return d_round_mid(a.sign, p, exp);
}
__device__ __forceinline__ D d_mul_int(D a, uint32_t n) { return d_mul_u32(a, n, 0); }   // int * Decimal

// ---- division: Decimal.__truediv__ (_pydecimal.py:1324) for b = Decimal(n), n > 0 an integer < 2^32 ----
template <int NL> __device__ __forceinline__ uint32_t w_mod5(const WN<NL>& x) {     // 2^32 = 1 (mod 5): the limb sum decides
    uint64_t sacc = 0;
    #pragma unroll
    for (int i = 0; i < NL; i++) sacc += x.w[i];
    return (uint32_t)(sacc % 5u);
}
template <int NL> __device__ __forceinline__ D d_div_impl(D a, uint32_t n, int shift) {
    WN<NL> x = d_wide<NL>(a);
    int exp = a.exp - shift;
    w_mul_pow10(x, shift);
    uint32_t rem = w_div_u32(x, n);
    if (rem != 0) {
        // inexact: the quotient has 29 or 30 digits (the dividend was scaled to len(n) + 29 digits); `coeff % 5 == 0 -> += 1`
        // folds the lost remainder into the last digit, then _fix drops one or two digits half-even (_pydecimal.py:1362-1376)
        if (w_mod5(x) == 0) w_inc(x);
        WN<NL> p29 = w_pow10<NL>(29);
        const bool d30 = w_cmp(x, p29) >= 0;
        uint32_t dg, half;
        if (d30) { dg = w_divc<100u>(x); half = 50u; exp += 2; } else { dg = w_divc<10u>(x); half = 5u; exp += 1; }
        if (dg > half || (dg == half && (x.w[0] & 1u))) {
            w_inc(x);
            if (x.w[0] == 0x10000000u && x.w[1] == 0x3e250261u && x.w[2] == 0x204fce5eu) {    // reached 10^28
                x.w[0] = 0xe8000000u; x.w[1] = 0x9fd0803cu; x.w[2] = 0x033b2e3cu; exp += 1;   // 10^27
            }
        }
        return d_make(x.w[0], x.w[1], x.w[2], exp, a.sign);
    }
    int ideal = a.exp;
    while (exp + 9 <= ideal) { WN<NL> t = x; if (w_divc<1000000000u>(t) != 0) break; x = t; exp += 9; }
    while (exp < ideal) { WN<NL> t = x; if (w_divc<10u>(t) != 0) break; x = t; exp += 1; }
    return d_fix_impl<NL>(a.sign, x, exp);
}
__device__ __noinline__ D d_div_general(D a, uint32_t n) {
    DEC_COUNT(5);
    if (d_is_zero(a)) return d_make(0, 0, 0, a.exp, a.sign);
    {   // exact integer quotient (e.g. adding to a position at its own VWAP): the result is coefficient / n at the
        // ideal exponent a.exp - no scaling, no trailing-zero stripping, and it already has <= 28 digits
        WN<3> x; x.w[0] = a.w0; x.w[1] = a.w1; x.w[2] = a.w2;
        if (w_div_u32(x, n) == 0) return d_make(x.w[0], x.w[1], x.w[2], a.exp, a.sign);
    }
    W4 xa = d_wide<4>(a), xn = w_from3<4>(n, 0, 0);
    int ln = w_ndigits(xn);
    int shift = ln - w_ndigits(xa) + 29;                    // >= 2; a * 10^shift has ln + 28 or ln + 29 digits
    if (ln <= 9) return d_div_impl<4>(a, n, shift);         // <= 38 digits < 2^127
    return d_div_impl<8>(a, n, shift);
}
__device__ __forceinline__ int ndigits_u32(uint32_t n) {     // len(str(n)), n >= 1
    int d = 1;
    d = n >= 10u ? 2 : d; d = n >= 100u ? 3 : d; d = n >= 1000u ? 4 : d; d = n >= 10000u ? 5 : d; d = n >= 100000u ? 6 : d;
    d = n >= 1000000u ? 7 : d; d = n >= 10000000u ? 8 : d; d = n >= 100000000u ? 9 : d; d = n >= 1000000000u ? 10 : d;
    return d;
}
// Inexact quotients (a VWAP after almost every fill) are rounded DIRECTLY to 28 digits: the coefficient is scaled so that
// floor(c * 10^s / n) has 26..28 digits (s from the bit length of c and the digit count of n), the long division is
// continued digit by digit from the 32-bit remainder until there are 28, and the final remainder against n/2 decides
// half-even - the correctly rounded quotient, which is what Decimal.__truediv__ followed by _fix produces.  A zero
// remainder anywhere means the quotient is exact at that scale; that case (ideal exponent, trailing zeros) is left to
// d_div_general.
__device__ __noinline__ D d_div_inexact_leaf(D a, uint32_t n) { // a LEAF (no calls: no return-address spill to scratch)
    DEC_COUNT(6);
    const u128 c = d_c128(a);
    const int ln = ndigits_u32(n);
    if (c != 0 && ln <= 9) {
        const int dn_hi = ((bits128(c) * 1233) >> 12) + 1;        // len(str(c)) or one more
        int s = 27 + ln - dn_hi;
        s = s < 0 ? 0 : s;                                        // c * 10^s < 10^(27 + ln) <= 10^36 < 2^120
        W4 x = w4_from128(c);
        w_mul_pow10(x, s);
        const double rn = 1.0 / (double)n;
        uint32_t r = n < (1u << 30) ? w_div_small(x, n, rn) : w_div_u32(x, n);
        #pragma unroll 1
        while (r != 0 && !(x.w[2] > 0x033b2e3cu || (x.w[2] == 0x033b2e3cu && (x.w[1] > 0x9fd0803cu || (x.w[1] == 0x9fd0803cu && x.w[0] >= 0xe8000000u))))) {
            const uint64_t r10 = (uint64_t)r * 10u;               // next quotient digit from the remainder (x < 10^27 so far)
            uint32_t dg = (uint32_t)((double)r10 * rn);
            int64_t rr = (int64_t)(r10 - (uint64_t)dg * n);
            if (rr < 0) { dg -= 1; rr += n; }
            if (rr >= (int64_t)n) { dg += 1; rr -= n; }
            r = (uint32_t)rr;
            uint64_t cy = dg;                                     // x = x * 10 + dg
            #pragma unroll
            for (int i = 0; i < 4; i++) { cy += (uint64_t)x.w[i] * 10u; x.w[i] = (uint32_t)cy; cy >>= 32; }
            s += 1;
        }
        if (r != 0) {
            const uint64_t twice = (uint64_t)r * 2u;
            if (twice > n || (twice == n && (x.w[0] & 1u))) {
                w_inc(x);
                if (x.w[0] == 0x10000000u && x.w[1] == 0x3e250261u && x.w[2] == 0x204fce5eu) {    // reached 10^28
                    x.w[0] = 0xe8000000u; x.w[1] = 0x9fd0803cu; x.w[2] = 0x033b2e3cu; s -= 1;     // 10^27
                }
            }
            return d_make(x.w[0], x.w[1], x.w[2], a.exp - s, a.sign);
        }
    }
    return d_make(0, 0, 0, D_NOT_HANDLED, 0);
}
__device__ __forceinline__ D d_div_u32(D a, uint32_t n) {
    if (a.w2 == 0) {        // a short dividend - a position built at ONE price, the usual case early in an episode: mostly an exact
        WN<2> x; x.w[0] = a.w0; x.w[1] = a.w1;          // quotient, which the inexact leaf would only hand on to the general routine
        if (w_div_u32(x, n) == 0) return d_make(x.w[0], x.w[1], 0, a.exp, a.sign);     // coefficient / n at the ideal exponent (also 0 / n)
    }
    D r = d_div_inexact_leaf(a, n);
    if (r.exp == D_NOT_HANDLED) r = d_div_general(a, n);
    return r;
}

// ---- comparison: Decimal._cmp (_pydecimal.py:817): -1, 0, 1 ----
__device__ __noinline__ int d_cmp_wide(D a, D b) {          // both non-zero, same sign
    int s = a.sign ? -1 : 1;
    // a coefficient is < 10^28, so an exponent gap >= 28 decides on its own; otherwise scale and compare
    int diff = a.exp - b.exp;
    if (diff >= 28) return s;
    if (diff <= -28) return -s;
    W xa = d_wide<8>(a), xb = d_wide<8>(b);
    if (diff > 0) w_mul_pow10(xa, diff); else if (diff < 0) w_mul_pow10(xb, -diff);
    int c = w_cmp(xa, xb);
    return c == 0 ? 0 : (c > 0 ? s : -s);
}
__device__ __noinline__ int d_cmp_mid(D a, D b) {           // both non-zero, same sign, any exponents
    int s = a.sign ? -1 : 1;
    int diff = a.exp - b.exp;
    u128 ca = d_c128(a), cb = d_c128(b);
    if (diff > 0) { if (!scale_fits128(ca, diff)) return d_cmp_wide(a, b); ca = mul_pow10_128(ca, diff); }
    else if (diff < 0) { if (!scale_fits128(cb, -diff)) return d_cmp_wide(a, b); cb = mul_pow10_128(cb, -diff); }
    return ca == cb ? 0 : (ca > cb ? s : -s);
}
__device__ __forceinline__ int d_cmp(D a, D b) {
    const bool az = d_is_zero(a), bz = d_is_zero(b);
    if (az) return bz ? 0 : (b.sign ? 1 : -1);
    if (bz) return a.sign ? -1 : 1;
    if (a.sign != b.sign) return a.sign ? -1 : 1;
    const int s = a.sign ? -1 : 1;
    // the operand with the larger exponent is scaled: the same one multiply as in d_add (short coefficient x table entry, or a long
    // coefficient x a power of ten below 2^32 - nav against max_nav: two long coefficients a few digits apart)
    const bool swp = a.exp < b.exp;
    const D t = swp ? b : a, o = swp ? a : b;
    const int diff = t.exp - o.exp;
    const bool short_t = (t.w1 | t.w2) == 0;
    if (!(short_t ? diff <= 28 : diff <= 9)) return d_cmp_mid(a, b);
    const lds_u32p p = lds_pow10(diff);
    const uint32_t p0 = p[0], p1 = p[1], p2 = p[2];
    const u128 big = short_t ? (((u128)p2 << 64) | ((u128)p1 << 32) | (u128)p0) : d_c128(t);
    const u128 ct = big * (u128)(short_t ? t.w0 : p0), co = d_c128(o);
    const int c = ct == co ? 0 : (ct > co ? s : -s);        // t against o
    return swp ? -c : c;
}

// ---- Decimal.__float__ (_pydecimal.py:1610) = correctly rounded nearest double of coeff * 10^exp ----
// Domain of the exact paths: exp in [-109, 0] (anything else sets *domain_err).
// round a quotient q (top set bit at position 55 or 56, weights 2^t..2^(t-56)) + sticky to a double * 2^-k
__device__ __forceinline__ double round_quotient(uint64_t q, bool sticky, int t, int k) {
    int nb = 64 - __clzll(q);
    int sh = nb - 53;
    uint64_t mant = q >> sh;
    uint64_t low = q & (((uint64_t)1 << sh) - 1);
    uint64_t half = (uint64_t)1 << (sh - 1);
    if (low > half || (low == half && (sticky || (mant & 1)))) mant += 1;
    return ldexp((double)mant, t - 56 + sh - k);
}
__device__ __forceinline__ void w_shl(W& x, int n) {          // x <<= n, 0 <= n < 256
    int ws = n >> 5, bs = n & 31;
    W r;
    #pragma unroll
    for (int i = 7; i >= 0; i--) {
        uint32_t lo = (i - ws >= 0) ? x.w[(i - ws) & 7] : 0u, lo2 = (i - ws - 1 >= 0) ? x.w[(i - ws - 1) & 7] : 0u;
        r.w[i] = bs ? ((lo << bs) | (lo2 >> (32 - bs))) : lo;
    }
    x = r;
}
// deep exponents (55 <= k <= 109): the same restoring division on 256-bit integers, 5^k built on the fly
__device__ __noinline__ double d_to_double_deep(u128 c, int k) {
    W dv = w_from3<8>(1, 0, 0);
    for (int i = k; i > 0;) { int st = i >= 13 ? 13 : i; uint32_t f = 1; for (int j = 0; j < st; j++) f *= 5u; w_mul_small(dv, f); i -= st; }
    W rr = w_from3<8>((uint32_t)c, (uint32_t)(c >> 32), (uint32_t)(c >> 64)); rr.w[3] = (uint32_t)(c >> 96);
    int t = w_bits(rr) - w_bits(dv);
    if (t >= 0) w_shl(dv, t); else w_shl(rr, -t);
    uint64_t q = 0;
    for (int i = 0; i < 57; i++) {
        q <<= 1;
        if (w_cmp(rr, dv) >= 0) { rr = w_sub(rr, dv); q |= 1; }
        w_shl(rr, 1);
    }
    return round_quotient(q, !w_is_zero(rr), t, k);
}
__device__ __forceinline__ double pow10_exact(int k) {       // 10^k, 0 <= k <= 22: every partial product is itself an exact double
    double p = ((k & 1) ? 10.0 : 1.0) * ((k & 2) ? 100.0 : 1.0);
    p *= (k & 4) ? 1.0e4 : 1.0; p *= (k & 8) ? 1.0e8 : 1.0; p *= (k & 16) ? 1.0e16 : 1.0;
    return p;
}
// Certified floating path for c * 10^-k, c >= 2^53, 0 <= k <= 44: the quotient is formed in double-double
// arithmetic (c = ch + cl exactly with a 53-bit ch; q1 = RN(ch / p); the remainder ch - q1 * p is exact in one fma;
// q2 = RN((r1 + cl) / p); a second such division by 10^(k-22) when k > 22), which approximates the true value to
// better than 2^-48 ulp.  s = RN(q1 + q2) with the exact error term t (Fast2Sum) is the correctly rounded double
// unless the true value lies within that bound of a rounding boundary; the test below rejects everything within
// 2^-30 of half an ulp (and exact powers of two, whose lower neighbour is half as far), and the caller then takes
// the exact integer path.  Built with -ffp-contract=off; `/` on doubles is the correctly rounded IEEE division.
__device__ __noinline__ double d_to_double_dd(u128 c, int k) {     // a LEAF, by value; NaN = not certified
    const int sft = bits128(c) - 53;                          // 0 <= sft <= 41
    const uint64_t hi = (uint64_t)(c >> sft);
    const uint64_t lo = (uint64_t)(c & ((((u128)1) << sft) - 1));
    const double ch = __builtin_ldexp((double)hi, sft), cl = (double)lo;     // both exact
    const double p = pow10_exact(k > 22 ? 22 : k);
    double q1 = ch / p;
    double r1 = __builtin_fma(-q1, p, ch);
    double q2 = (r1 + cl) / p;
    if (k > 22) {
        const double p2 = pow10_exact(k - 22);
        const double Q1 = q1 / p2;
        const double R1 = __builtin_fma(-Q1, p2, q1);
        q2 = (R1 + q2) / p2;
        q1 = Q1;
    }
    const double s = q1 + q2;
    const double t = (q1 - s) + q2;                           // s + t == q1 + q2 exactly (|q1| >= |q2|)
    const uint64_t sb = (uint64_t)__double_as_longlong(s);
    const uint32_t e = (uint32_t)(sb >> 52) & 0x7ffu;
    const double not_certified = __longlong_as_double(0x7ff8000000000000LL);
    if ((sb & 0xfffffffffffffull) == 0 || e < 64u) return not_certified;
    const double hu = __longlong_as_double((long long)((uint64_t)(e - 53u) << 52));   // half an ulp of s
    if (!(__builtin_fabs(t) < hu * (1.0 - 9.313225746154785e-10))) return not_certified;   // 2^-30 short of the boundary
    return s;
}
__device__ __noinline__ double d_to_double_slow(D a) {      // by value only: no caller state is forced to memory
    DEC_COUNT(7);
    if (d_is_zero(a)) return a.sign ? -0.0 : 0.0;
    int k = -a.exp;
    double r;

    if (k > 54 && k <= 109) { r = d_to_double_deep(d_c128(a), k); return a.sign ? -r : r; }
    if (k < 0 || k > 109) {                                  // outside the exact domain (flagged by the inline wrapper)
        u128 c0 = d_c128(a);
        r = (double)(uint64_t)(c0 >> 64) * 18446744073709551616.0 + (double)(uint64_t)c0;
        r = r * pow(10.0, (double)a.exp);
        return a.sign ? -r : r;
    }
    // 999525.0000000000000000000000 -> 9995250 * 10^-1: strip factors of 10 (exact), 9 digits at a time
    WN<3> x; x.w[0] = a.w0; x.w[1] = a.w1; x.w[2] = a.w2;
    if (a.w2 != 0 || a.w1 >= (1u << 21)) {
        while (k >= 9) { WN<3> t = x; if (w_divc<1000000000u>(t) != 0) break; x = t; k -= 9; }
        // what is left has fewer than 9 strippable zeros: peel 4, 2, 1 (then 4, 2, 1 once more covers up to 8... a second
        // pass of the three tests handles 8 = 4+2+1+1) instead of one digit at a time
        #pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            if (k >= 4) { WN<3> t = x; if (w_divc<10000u>(t) == 0) { x = t; k -= 4; } }
            if (k >= 2) { WN<3> t = x; if (w_divc<100u>(t) == 0) { x = t; k -= 2; } }
            if (k >= 1) { WN<3> t = x; if (w_divc<10u>(t) == 0) { x = t; k -= 1; } }
        }
    }
    u128 c = ((u128)x.w[2] << 64) | ((u128)x.w[1] << 32) | x.w[0];
    if ((c >> 53) == 0 && k <= 22) {
        double p = ((k & 1) ? 10.0 : 1.0) * ((k & 2) ? 100.0 : 1.0);
        p *= (k & 4) ? 1.0e4 : 1.0; p *= (k & 8) ? 1.0e8 : 1.0; p *= (k & 16) ? 1.0e16 : 1.0;
        r = (double)(uint64_t)c / p;                     // both exact -> one correctly rounded division
        return a.sign ? -r : r;
    }
    // exact path: value = c / (5^k * 2^k)
    if (k <= 27) {
        // 5^k < 2^63: base-2^32 long division of the normalised coefficient (shifted up one limb) by the 64-bit divisor.
        // Each quotient digit is estimated in f64 (relative error 2^-52) and corrected with an exact 128-bit remainder test.
        const uint64_t dv = POW5.lo[k];
        const double rdv = 1.0 / (double)dv;
        const int nshift = 96 - bits128(c);                              // normalise: top bit of the coefficient to bit 95
        const u128 cn = c << nshift;                                     // >= 2^95 > dv  =>  quotient >= 2^96
        uint32_t num[4] = {0u, (uint32_t)cn, (uint32_t)(cn >> 32), (uint32_t)(cn >> 64)};   // (cn << 32), little endian
        uint32_t ql[4];
        uint64_t rem = 0;                                                // < dv
        #pragma unroll
        for (int i = 3; i >= 0; i--) {
            u128 cur = ((u128)rem << 32) | num[i];                       // < dv * 2^32
            double curd = (double)(uint64_t)(cur >> 32) * 4294967296.0 + (double)(uint32_t)cur;
            uint64_t q = (uint64_t)(curd * rdv);
            q = q > 0xffffffffull ? 0xffffffffull : q;
            u128 prod = (u128)q * dv;
            while (prod > cur) { q -= 1; prod -= dv; }
            u128 r128 = cur - prod;
            while (r128 >= (u128)dv) { q += 1; r128 -= dv; }
            ql[i] = (uint32_t)q; rem = (uint64_t)r128;
        }
        // Q = floor(cn * 2^32 / dv) >= 2^(127-63): at least 65 bits, so its top 57 bits + sticky decide the rounding
        u128 Q = ((u128)ql[3] << 96) | ((u128)ql[2] << 64) | ((u128)ql[1] << 32) | ql[0];
        int sh = bits128(Q) - 57;                                        // >= 8
        uint64_t q57 = (uint64_t)(Q >> sh);
        bool sticky = rem != 0 || (Q & (((u128)1 << sh) - 1)) != 0;
        // value = Q * 2^-32 * 2^-nshift * 2^-k, so one unit of q57 weighs 2^(sh - 32 - nshift - k); round_quotient wants the
        // unit weight as 2^(t - 56 - k)
        r = round_quotient(q57, sticky, sh + 24 - nshift, k);
        return a.sign ? -r : r;
    }
    // 28 <= k <= 54: restoring division for 57 quotient bits + sticky
    u128 dv = ((u128)POW5.hi[k] << 64) | POW5.lo[k];
    int bn = bits128(c), bd = bits128(dv);
    int t = bn - bd;
    u128 rr = c, dn = dv;
    if (t >= 0) dn <<= t; else rr <<= (-t);
    uint64_t q = 0;
    for (int i = 0; i < 57; i++) {
        q <<= 1;
        if (rr >= dn) { rr -= dn; q |= 1; }
        rr <<= 1;
    }
    r = round_quotient(q, rr != 0, t, k);
    return a.sign ? -r : r;
}
__device__ __forceinline__ double d_to_double(D a, uint32_t* domain_err) {
    int k = -a.exp;
    if (a.w2 == 0 && a.w1 < (1u << 21) && k >= 0 && k <= 22) {   // coefficient < 2^53: one exact division
        uint64_t c = ((uint64_t)a.w1 << 32) | a.w0;
        // 10^k (k <= 22) built from exact powers of ten: every partial product is itself an exact double
        double p = ((k & 1) ? 10.0 : 1.0) * ((k & 2) ? 100.0 : 1.0);
        p *= (k & 4) ? 1.0e4 : 1.0; p *= (k & 8) ? 1.0e8 : 1.0; p *= (k & 16) ? 1.0e16 : 1.0;
        double r = (double)c / p;
        return a.sign ? -r : r;
    }
    if (k >= 0 && k <= 44 && (a.w2 != 0 || a.w1 >= (1u << 21))) {           // >= 2^53: certified double-double quotient
        const double r = d_to_double_dd(d_c128(a), k);
        if (r == r) return a.sign ? -r : r;
    }
    if ((k < 0 || k > 109) && !d_is_zero(a) && domain_err) *domain_err |= 0x4u;
    return d_to_double_slow(a);
}

}  // namespace cda
