/*
 * cda_oracle.c - CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the reference's reset()/step() hot path for ONE market at a time,
 * written from the parity rulebook so that the HIP path has something to
 * be compared with on the GPU box, where the Python reference cannot travel.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (gym_continuousdoubleauction_amd/) never links, imports or calls it.
 *
 * PINNING: this oracle is pinned against golden vectors cut from the reference itself, imported
 * in the build container (tests/golden/make_goldens.py -> tests/golden/ fixtures), by
 * tests/test_oracle_golden.py (35 traces), and was cross-checked against the reference on 2300 fresh random episodes
 * over the whole config space (tests/golden/crosscheck_oracle.py); its decimal arithmetic is pinned against CPython's `decimal`
 * and its RNG against numpy, both live (tests/test_oracle_arith.py).
 *
 * Third-party arithmetic restated here (not under /root/reference):
 *   - numpy.random.Generator(PCG64(SeedSequence(seed))) - numpy==2.5.2 pinned by the reference
 *     (requirements-lock.txt:45; floor >=2.2): SeedSequence, PCG64 XSL-RR, next_uint32 half-word
 *     buffering, Lemire bounded ints, masked-rejection random_interval, Fisher-Yates permutation,
 *     ziggurat standard normal;
 *   - Python `decimal` (General Decimal Arithmetic, prec 28, ROUND_HALF_EVEN): add/sub/mul/div/
 *     compare/str/float as stated by /usr/lib/python3.10/_pydecimal.py:1157,1267,1324,1661,817.
 *
 * Reference citations are relative to /root/reference/gym_continuousDoubleAuction/envs/.
 */
#include "cda_oracle.h"
#include "../include/cda_random_agents.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CDA_ZIG_QUAL static const
#include "ziggurat_tables.h"

typedef unsigned __int128 u128;

/* ======================================================================================
 * 256-bit unsigned integers (8 x u32 limbs, little endian) - wide enough for every
 * intermediate of a 28-digit decimal operation (<= 58 digits).
 * ==================================================================================== */
#define BL 8
typedef struct { uint32_t w[BL]; } big;

static big big_from_u64(uint64_t v) { big r; memset(&r, 0, sizeof r); r.w[0] = (uint32_t)v; r.w[1] = (uint32_t)(v >> 32); return r; }
static int big_is_zero(const big* a) { for (int i = 0; i < BL; i++) if (a->w[i]) return 0; return 1; }
static int big_cmp(const big* a, const big* b) {
    for (int i = BL - 1; i >= 0; i--) { if (a->w[i] != b->w[i]) return a->w[i] < b->w[i] ? -1 : 1; }
    return 0;
}
static big big_add(const big* a, const big* b) {
    big r; uint64_t c = 0;
    for (int i = 0; i < BL; i++) { c += (uint64_t)a->w[i] + b->w[i]; r.w[i] = (uint32_t)c; c >>= 32; }
    return r;
}
static big big_sub(const big* a, const big* b) { /* a >= b */
    big r; int64_t c = 0;
    for (int i = 0; i < BL; i++) { c += (int64_t)a->w[i] - b->w[i]; r.w[i] = (uint32_t)c; c >>= 32; }
    return r;
}
static big big_mul(const big* a, const big* b) { /* truncated to 256 bits */
    big r; memset(&r, 0, sizeof r);
    for (int i = 0; i < BL; i++) {
        uint64_t c = 0;
        for (int j = 0; i + j < BL; j++) {
            c += (uint64_t)a->w[i] * b->w[j] + r.w[i + j];
            r.w[i + j] = (uint32_t)c; c >>= 32;
        }
    }
    return r;
}
static int big_bits(const big* a) {
    for (int i = BL - 1; i >= 0; i--) if (a->w[i]) return 32 * i + 32 - __builtin_clz(a->w[i]);
    return 0;
}
static int big_bit(const big* a, int i) { return (a->w[i >> 5] >> (i & 31)) & 1; }
static int big_fits_u64(const big* a);
static void big_divmod_u64(const big* a, uint64_t d, big* q, big* r);
/* schoolbook bit-serial division: q = a / b, r = a % b (b != 0) */
static void big_divmod(const big* a, const big* b, big* q, big* r) {
    if (big_fits_u64(b)) { big_divmod_u64(a, (uint64_t)b->w[0] | ((uint64_t)b->w[1] << 32), q, r); return; }
    big quo, rem; memset(&quo, 0, sizeof quo); memset(&rem, 0, sizeof rem);
    for (int i = big_bits(a) - 1; i >= 0; i--) {
        for (int k = BL - 1; k > 0; k--) rem.w[k] = (rem.w[k] << 1) | (rem.w[k - 1] >> 31);
        rem.w[0] = (rem.w[0] << 1) | (uint32_t)big_bit(a, i);
        if (big_cmp(&rem, b) >= 0) { rem = big_sub(&rem, b); quo.w[i >> 5] |= 1u << (i & 31); }
    }
    *q = quo; *r = rem;
}
/* a /= d (32-bit divisor), returns the remainder */
static uint32_t big_div_u32(big* a, uint32_t d) {
    uint64_t rem = 0;
    for (int i = BL - 1; i >= 0; i--) { uint64_t cur = (rem << 32) | a->w[i]; a->w[i] = (uint32_t)(cur / d); rem = cur % d; }
    return (uint32_t)rem;
}
static int big_fits_u64(const big* a) { for (int i = 2; i < BL; i++) if (a->w[i]) return 0; return 1; }
/* q = a / d, r = a % d for a divisor below 2^64 (limb-wise with a 128-bit intermediate) */
static void big_divmod_u64(const big* a, uint64_t d, big* q, big* r) {
    u128 rem = 0; big quo;
    for (int i = BL - 1; i >= 0; i--) { u128 cur = (rem << 32) | a->w[i]; quo.w[i] = (uint32_t)(cur / d); rem = cur % d; }
    *q = quo; *r = big_from_u64((uint64_t)rem);
}
static uint32_t big_mod_u32(const big* a, uint32_t d) {
    uint64_t rem = 0;
    for (int i = BL - 1; i >= 0; i--) rem = ((rem << 32) | a->w[i]) % d;
    return (uint32_t)rem;
}

#define NPOW 78
static big g_pow10[NPOW];
static int g_init_done = 0;
static void oracle_init(void) {
    if (g_init_done) return;
    g_pow10[0] = big_from_u64(1);
    big ten = big_from_u64(10);
    for (int i = 1; i < NPOW; i++) g_pow10[i] = big_mul(&g_pow10[i - 1], &ten);
    g_init_done = 1;
}
static int big_ndigits(const big* a) { /* len(str(a)); 1 for zero */
    int b = big_bits(a);
    if (b == 0) return 1;
    int t = (b * 1233) >> 12;                 /* floor(b*log10(2)) for b <= 256 */
    return t + (big_cmp(a, &g_pow10[t]) >= 0 ? 1 : 0);
}

/* ======================================================================================
 * Decimal (prec 28, ROUND_HALF_EVEN)
 * ==================================================================================== */
#define PREC 28
typedef struct { big c; int32_t exp; int sign; } dec;

static dec dec_from_i64(int64_t v) {
    dec d; d.sign = v < 0; d.exp = 0; d.c = big_from_u64(v < 0 ? (uint64_t)(-(v + 1)) + 1u : (uint64_t)v); return d;
}
static dec dec_zero(void) { return dec_from_i64(0); }
/* an integer tick price p as the book holds it: Decimal(str(float(p))) = Decimal('p.0')
 * (orderbook/orderbook.py:52,239) -> coefficient 10*p, exponent -1 */
static dec dec_price(int32_t p) { dec d = dec_from_i64((int64_t)p * 10); d.exp = -1; return d; }

/* Decimal._fix (_pydecimal.py:1661), without Emax/Emin handling */
static dec dec_fix(dec d) {
    if (big_is_zero(&d.c)) return d;
    int nd = big_ndigits(&d.c);
    if (nd <= PREC) return d;
    int drop = nd - PREC;
    big q = d.c; int sticky = 0, k = drop - 1;
    while (k >= 9) { sticky |= big_div_u32(&q, 1000000000u) != 0; k -= 9; }
    if (k > 0) { static const uint32_t p10[9] = {1u, 10u, 100u, 1000u, 10000u, 100000u, 1000000u, 10000000u, 100000000u}; sticky |= big_div_u32(&q, p10[k]) != 0; }
    uint32_t dg = big_div_u32(&q, 10u);
    if (dg > 5 || (dg == 5 && (sticky || (q.w[0] & 1)))) {
        big one = big_from_u64(1); q = big_add(&q, &one);
        if (big_cmp(&q, &g_pow10[PREC]) == 0) { q = g_pow10[PREC - 1]; drop += 1; }
    }
    d.c = q; d.exp += drop;
    return d;
}

/* Decimal.__add__ (_pydecimal.py:1157) + _normalize (:5640) + _rescale (:2612) */
static dec dec_add(dec a, dec b) {
    int32_t exp = a.exp < b.exp ? a.exp : b.exp;
    int az = big_is_zero(&a.c), bz = big_is_zero(&b.c);
    dec r;
    if (az && bz) { r.sign = a.sign < b.sign ? a.sign : b.sign; r.c = big_from_u64(0); r.exp = exp; return r; }
    if (az || bz) {
        dec o = az ? b : a;
        int32_t e = exp > o.exp - PREC - 1 ? exp : o.exp - PREC - 1;
        /* o.exp >= e always here (e <= min(exp) <= o.exp or e = o.exp-29) -> pad with zeros */
        r.sign = o.sign; r.c = big_mul(&o.c, &g_pow10[o.exp - e]); r.exp = e;
        return dec_fix(r);
    }
    dec *tmp, *oth;
    if (a.exp < b.exp) { tmp = &b; oth = &a; } else { tmp = &a; oth = &b; }
    int tmp_len = big_ndigits(&tmp->c), oth_len = big_ndigits(&oth->c);
    int32_t m = tmp_len - PREC - 2; if (m > -1) m = -1;
    int32_t e = tmp->exp + m;
    if (oth_len + oth->exp - 1 < e) { oth->c = big_from_u64(1); oth->exp = e; }
    tmp->c = big_mul(&tmp->c, &g_pow10[tmp->exp - oth->exp]); tmp->exp = oth->exp;
    /* a, b now share an exponent */
    if (a.sign != b.sign) {
        int c = big_cmp(&a.c, &b.c);
        if (c == 0) { r.sign = 0; r.c = big_from_u64(0); r.exp = exp; return r; }
        if (c > 0) { r.sign = a.sign; r.c = big_sub(&a.c, &b.c); }
        else       { r.sign = b.sign; r.c = big_sub(&b.c, &a.c); }
    } else { r.sign = a.sign; r.c = big_add(&a.c, &b.c); }
    r.exp = a.exp;
    return dec_fix(r);
}
static dec dec_neg(dec a) { a.sign ^= 1; return a; }
static dec dec_sub(dec a, dec b) { return dec_add(a, dec_neg(b)); }
/* Decimal.__mul__ (_pydecimal.py:1267) */
static dec dec_mul(dec a, dec b) {
    dec r; r.sign = a.sign ^ b.sign; r.exp = a.exp + b.exp;
    if (big_is_zero(&a.c) || big_is_zero(&b.c)) { r.c = big_from_u64(0); return r; }
    r.c = big_mul(&a.c, &b.c);
    return dec_fix(r);
}
/* Decimal.__truediv__ (_pydecimal.py:1324), b != 0 */
static dec dec_div(dec a, dec b) {
    dec r; r.sign = a.sign ^ b.sign;
    if (big_is_zero(&a.c)) { r.c = big_from_u64(0); r.exp = a.exp - b.exp; return r; }
    int shift = big_ndigits(&b.c) - big_ndigits(&a.c) + PREC + 1;
    int32_t exp = a.exp - b.exp - shift;
    big q, rem;
    if (shift >= 0) { big n = big_mul(&a.c, &g_pow10[shift]); big_divmod(&n, &b.c, &q, &rem); }
    else { big dd = big_mul(&b.c, &g_pow10[-shift]); big_divmod(&a.c, &dd, &q, &rem); }
    if (!big_is_zero(&rem)) {
        if (big_mod_u32(&q, 5) == 0) { big one = big_from_u64(1); q = big_add(&q, &one); }
    } else {
        int32_t ideal = a.exp - b.exp;
        while (exp < ideal && big_mod_u32(&q, 10) == 0) { big_div_u32(&q, 10u); exp++; }
    }
    r.c = q; r.exp = exp;
    return dec_fix(r);
}
/* Decimal._cmp (_pydecimal.py:817): -1, 0, 1 */
static int dec_cmp(dec a, dec b) {
    int az = big_is_zero(&a.c), bz = big_is_zero(&b.c);
    if (az) { if (bz) return 0; return b.sign ? 1 : -1; }
    if (bz) return a.sign ? -1 : 1;
    if (a.sign != b.sign) return a.sign ? -1 : 1;
    int s = a.sign ? -1 : 1;
    int aa = big_ndigits(&a.c) + a.exp, ba = big_ndigits(&b.c) + b.exp; /* adjusted()+1 */
    if (aa != ba) return aa > ba ? s : -s;
    if (a.exp > b.exp) a.c = big_mul(&a.c, &g_pow10[a.exp - b.exp]);
    else if (b.exp > a.exp) b.c = big_mul(&b.c, &g_pow10[b.exp - a.exp]);
    int c = big_cmp(&a.c, &b.c);
    return c == 0 ? 0 : (c > 0 ? s : -s);
}
static int dec_sign_cmp0(dec a) { if (big_is_zero(&a.c)) return 0; return a.sign ? -1 : 1; }

static void big_to_digits(const big* a, char* out /* >= 80 */) {
    char tmp[80]; int n = 0; big v = *a;
    if (big_is_zero(&v)) { out[0] = '0'; out[1] = 0; return; }
    while (!big_is_zero(&v)) {
        uint32_t r = big_div_u32(&v, 1000000000u);
        int last = big_is_zero(&v);
        for (int j = 0; j < 9 && (!last || r); j++) { tmp[n++] = (char)('0' + r % 10); r /= 10; }
    }
    for (int i = 0; i < n; i++) out[i] = tmp[n - 1 - i];
    out[n] = 0;
}
/* Decimal.__str__ (_pydecimal.py:1031) */
static void dec_to_str(dec d, char* out /* >= 128 */) {
    char digs[80]; big_to_digits(&d.c, digs);
    int len = (int)strlen(digs);
    int leftdigits = d.exp + len, dotplace;
    char* p = out;
    if (d.sign) *p++ = '-';
    if (d.exp <= 0 && leftdigits > -6) dotplace = leftdigits; else dotplace = 1;
    if (dotplace <= 0) { *p++ = '0'; *p++ = '.'; for (int i = 0; i < -dotplace; i++) *p++ = '0'; memcpy(p, digs, (size_t)len); p += len; }
    else if (dotplace >= len) { memcpy(p, digs, (size_t)len); p += len; for (int i = 0; i < dotplace - len; i++) *p++ = '0'; }
    else { memcpy(p, digs, (size_t)dotplace); p += dotplace; *p++ = '.'; memcpy(p, digs + dotplace, (size_t)(len - dotplace)); p += len - dotplace; }
    if (leftdigits != dotplace) p += sprintf(p, "E%+d", leftdigits - dotplace);
    *p = 0;
}
/* Decimal.__float__ (_pydecimal.py:1610): float(str(d)) - strtod is correctly rounded */
static double dec_to_double(dec d) { char s[128]; dec_to_str(d, s); return strtod(s, NULL); }

static cda_dec dec_pack(dec d, uint32_t* flags) {
    cda_dec o; memset(&o, 0, sizeof o);
    o.w[0] = d.c.w[0]; o.w[1] = d.c.w[1]; o.w[2] = d.c.w[2];
    for (int i = 3; i < BL; i++) if (d.c.w[i] && flags) *flags |= CDA_FLAG_DEC_DOMAIN;
    if ((d.exp < -32768 || d.exp > 32767) && flags) *flags |= CDA_FLAG_DEC_DOMAIN;
    o.exp = (int16_t)d.exp; o.sign = (uint8_t)d.sign;
    return o;
}
static dec dec_unpack(cda_dec p) {
    dec d; memset(&d, 0, sizeof d);
    d.c.w[0] = p.w[0]; d.c.w[1] = p.w[1]; d.c.w[2] = p.w[2]; d.exp = p.exp; d.sign = p.sign;
    return d;
}

/* ======================================================================================
 * numpy RNG: SeedSequence -> PCG64 -> integers / normal / permutation   
 * ==================================================================================== */
typedef struct { u128 state, inc; uint32_t has_uint32, uinteger; } rng_t;

static const u128 PCG_MULT = ((u128)0x2360ed051fc65da4ULL << 64) | 0x4385df649fccf645ULL;

static uint32_t ss_hashmix(uint32_t value, uint32_t* hash_const) {
    value ^= *hash_const; *hash_const *= 0x931e8875u; value *= *hash_const; value ^= value >> 16; return value;
}
static uint32_t ss_mix(uint32_t x, uint32_t y) {
    uint32_t r = 0xca01f9ddu * x - 0x4973f715u * y; r ^= r >> 16; return r;
}
/* numpy.random.SeedSequence(seed).generate_state(4, uint64) for an integer seed < 2^64 */
static void seed_sequence_u64x4(uint64_t seed, uint64_t out[4]) {
    uint32_t ent[2]; int n_ent;
    ent[0] = (uint32_t)seed; ent[1] = (uint32_t)(seed >> 32);
    n_ent = ent[1] ? 2 : 1;
    uint32_t pool[4], hc = 0x43b0d7e5u;
    for (int i = 0; i < 4; i++) pool[i] = ss_hashmix(i < n_ent ? ent[i] : 0u, &hc);
    for (int s = 0; s < 4; s++) for (int d = 0; d < 4; d++) if (s != d) pool[d] = ss_mix(pool[d], ss_hashmix(pool[s], &hc));
    uint32_t hb = 0x8b51f9ddu, st[8];
    for (int i = 0; i < 8; i++) {
        uint32_t v = pool[i & 3]; v ^= hb; hb *= 0x58f38dedu; v *= hb; v ^= v >> 16; st[i] = v;
    }
    for (int i = 0; i < 4; i++) out[i] = (uint64_t)st[2 * i] | ((uint64_t)st[2 * i + 1] << 32);
}
static void rng_seed(rng_t* r, uint64_t seed) {
    uint64_t v[4]; seed_sequence_u64x4(seed, v);
    u128 initstate = ((u128)v[0] << 64) | v[1], initseq = ((u128)v[2] << 64) | v[3];
    r->state = 0; r->inc = (initseq << 1) | 1;
    r->state = r->state * PCG_MULT + r->inc;
    r->state += initstate;
    r->state = r->state * PCG_MULT + r->inc;
    r->has_uint32 = 0; r->uinteger = 0;
}
static uint64_t rng_next64(rng_t* r) {
    r->state = r->state * PCG_MULT + r->inc;
    uint64_t hi = (uint64_t)(r->state >> 64), lo = (uint64_t)r->state, x = hi ^ lo;
    unsigned rot = (unsigned)(r->state >> 122);
    return (x >> rot) | (x << ((-rot) & 63));
}
static uint32_t rng_next32(rng_t* r) {
    if (r->has_uint32) { r->has_uint32 = 0; return r->uinteger; }
    uint64_t n = rng_next64(r);
    r->has_uint32 = 1; r->uinteger = (uint32_t)(n >> 32);
    return (uint32_t)n;
}
static double rng_double(rng_t* r) { return (double)(rng_next64(r) >> 11) * (1.0 / 9007199254740992.0); }
/* Generator.integers(lo, hi+1) for hi-lo < 2^32-1: Lemire on 32-bit draws */
static int64_t rng_integers(rng_t* r, int64_t lo, int64_t hi_incl) {
    uint32_t rng = (uint32_t)(hi_incl - lo);
    if (rng == 0) return lo;
    uint32_t rng_excl = rng + 1;
    uint64_t m = (uint64_t)rng_next32(r) * rng_excl;
    uint32_t leftover = (uint32_t)m;
    if (leftover < rng_excl) {
        uint32_t threshold = (0xFFFFFFFFu - rng) % rng_excl;
        while (leftover < threshold) { m = (uint64_t)rng_next32(r) * rng_excl; leftover = (uint32_t)m; }
    }
    return lo + (int64_t)(m >> 32);
}
static uint32_t rng_interval(rng_t* r, uint32_t max) {
    if (max == 0) return 0;
    uint32_t mask = max, v;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    while ((v = (rng_next32(r) & mask)) > max) {}
    return v;
}
static void rng_permutation(rng_t* r, int n, int* perm) {
    for (int i = 0; i < n; i++) perm[i] = i;
    for (int i = n - 1; i >= 1; i--) { int j = (int)rng_interval(r, (uint32_t)i); int t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
}
static double bits2d(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
static double rng_std_normal(rng_t* r) {
    static const double zr = 3.6541528853610087963519472518, inv_r = 0.27366123732975827203338247596;
    for (;;) {
        uint64_t u = rng_next64(r);
        int idx = (int)(u & 0xff); u >>= 8;
        int sign = (int)(u & 1);
        uint64_t rabs = (u >> 1) & 0x000fffffffffffffULL;
        double x = (double)rabs * bits2d(cda_zig_wi_bits[idx]);
        if (sign) x = -x;
        if (rabs < cda_zig_ki[idx]) return x;
        if (idx == 0) {
            for (;;) {
                double xx = -inv_r * log1p(-rng_double(r));
                double yy = -log1p(-rng_double(r));
                if (yy + yy > xx * xx) return ((rabs >> 8) & 1) ? -(zr + xx) : zr + xx;
            }
        } else {
            double f1 = bits2d(cda_zig_fi_bits[idx - 1]), f0 = bits2d(cda_zig_fi_bits[idx]);
            if ((f1 - f0) * rng_double(r) + f0 < exp(-0.5 * x * x)) return x;
        }
    }
}

/* ======================================================================================
 * One market
 * ==================================================================================== */
typedef struct { int32_t price, qty, owner, order_id, timestamp; } order_t;
typedef struct {
    int n, alloc;
    order_t* o;                /* queue order: best price first, FIFO inside a level; grown on demand (the reference's
                                  OrderTree is unbounded, orderbook/ordertree.py:5-58) */
} side_t;

typedef struct {
    dec cash, hold, posval, vwap, nav, prev_nav, max_nav;
    int32_t net_position, num_trades;
    int32_t num_trades_step, num_passive_fills_step, order_step_placed, num_rejected_step;
} acc_t;

typedef struct {
    rng_t rng; int seeded;
    int32_t t_step, lob_time, next_order_id, last_price, has_trade, last_trade_price;
    uint32_t done_mask, flags;
    side_t side[2];            /* 0 bids, 1 asks */
    int32_t book_cap;          /* resting orders the market may hold, both sides together; 0 = unbounded like the reference */
    int32_t peak_orders;       /* census: most resting orders held since the last reset */
    acc_t acc[CDA_MAX_AGENTS];
    float hist[CDA_MAX_HIST * CDA_SNAPSHOT_DIM];
    float raw[CDA_RAW_DIM];    /* agg_LOB_raw of the last set_agg_LOB */
} market_t;

struct oracle_env {
    cda_config cfg;
    int32_t n;
    int32_t book_cap;          /* the product's capacity (cda_create's rule) unless oracle_set_book_cap changed it; 0 = unbounded */
    float mkt_mul, lim_mul;
    market_t* m;
};

enum { T_MARKET = 0, T_LIMIT = 1, T_MODIFY = 2, T_CANCEL = 3 };
enum { S_BID = 0, S_ASK = 1, S_NONE = 2 };

typedef struct { int32_t price, qty, counter, counter_oid, init_side; } fill_t;
typedef struct { fill_t* f; int n, alloc; } fills_t;       /* the trades of one order, grown on demand */
static void fills_push(fills_t* fl, fill_t f) {
    if (fl->n == fl->alloc) { fl->alloc = fl->alloc ? 2 * fl->alloc : 16; fl->f = (fill_t*)realloc(fl->f, (size_t)fl->alloc * sizeof(fill_t)); if (!fl->f) abort(); }
    fl->f[fl->n++] = f;
}

static int better_or_equal(int s, int32_t resting, int32_t p) { return s == S_BID ? resting >= p : resting <= p; }

static void side_remove(side_t* sd, int idx) {
    for (int i = idx; i + 1 < sd->n; i++) sd->o[i] = sd->o[i + 1];
    sd->n--;
}
/* OrderTree.insert_order (orderbook/ordertree.py:44-58): tail of its price level */
static int side_insert(market_t* m, int s, order_t o) {
    side_t* sd = &m->side[s]; const side_t* other = &m->side[s ^ 1];
    if (m->book_cap > 0 && sd->n + other->n >= m->book_cap) return 0;       /* the product's pool (256 or 512 resting orders per market) when mirrored */
    if (sd->n == sd->alloc) { sd->alloc = sd->alloc ? 2 * sd->alloc : 64; sd->o = (order_t*)realloc(sd->o, (size_t)sd->alloc * sizeof(order_t)); if (!sd->o) abort(); }
    int pos = 0;
    while (pos < sd->n && better_or_equal(s, sd->o[pos].price, o.price)) pos++;
    for (int i = sd->n; i > pos; i--) sd->o[i] = sd->o[i - 1];
    sd->o[pos] = o; sd->n++;
    if (sd->n + other->n > m->peak_orders) m->peak_orders = sd->n + other->n;
    return 1;
}

/* OrderBook.process_order_list / process_market_order / process_limit_order matching loops
 * (orderbook/orderbook.py:61-194): fills at the resting order's price, head first. `limit` < 0
 * means a market order.  Returns the unfilled quantity. */
static int32_t match(market_t* m, int own_side, int32_t qty, int32_t limit, fills_t* fills) {
    side_t* opp = &m->side[own_side ^ 1];
    while (qty > 0 && opp->n > 0) {
        order_t* h = &opp->o[0];
        if (limit >= 0) { if (own_side == S_BID ? !(limit >= h->price) : !(limit <= h->price)) break; }
        fill_t f; f.price = h->price; f.counter = h->owner; f.counter_oid = h->order_id; f.init_side = own_side;
        if (qty < h->qty) { f.qty = qty; h->qty -= qty; qty = 0; }
        else { f.qty = h->qty; qty -= h->qty; side_remove(opp, 0); }
        fills_push(fills, f);
        m->has_trade = 1; m->last_trade_price = f.price;
    }
    return qty;
}

/* ---- Account (account/account.py, account/cash_processor.py) ---- */
static dec cal_profit(int is_long, dec mkt, dec raw) { return is_long ? dec_sub(mkt, raw) : dec_sub(raw, mkt); }
static void xfer_inc(acc_t* a, int counter, dec v) { if (!counter) a->cash = dec_sub(a->cash, v); else a->hold = dec_sub(a->hold, v); }
static void xfer_dec(acc_t* a, int counter, dec v) {
    if (!counter) a->cash = dec_add(a->cash, v);
    else { a->cash = dec_add(a->cash, v); a->hold = dec_sub(a->hold, v); a->cash = dec_add(a->cash, v); }
}
static dec acc_covered(acc_t* a, int is_long, dec p) { /* account.py:135-149 */
    int64_t ap = llabs((long long)a->net_position);
    dec raw = dec_mul(dec_from_i64(ap), a->vwap), mkt = dec_mul(dec_from_i64(ap), p);
    a->posval = dec_add(raw, cal_profit(is_long, mkt, raw));
    a->cash = dec_add(a->cash, dec_sub(a->posval, mkt));   /* size_zero_cash_transfer */
    a->posval = dec_zero(); a->vwap = dec_zero();
    return mkt;
}
/* Account.process_acc (account.py:215-231) */
static void process_acc(acc_t* a, int32_t q, int32_t price, int own_side, int counter) {
    a->num_trades++; a->num_trades_step++; if (counter) a->num_passive_fills_step++;
    dec p = dec_price(price), tv = dec_mul(dec_from_i64(q), p);
    int32_t pos = a->net_position; int64_t ap = llabs((long long)pos);
    int mode; /* 0 neutral, 1 inc, 2 dec, 3 flip */
    int is_long = pos > 0;
    if (pos > 0) mode = own_side == S_BID ? 1 : (pos >= q ? 2 : 3);
    else if (pos < 0) mode = own_side == S_ASK ? 1 : (ap >= q ? 2 : 3);
    else mode = 0;
    if (mode == 0) { a->posval = dec_add(a->posval, tv); a->vwap = p; xfer_inc(a, counter, tv); }
    else if (mode == 1) {
        int64_t n = ap + q;
        a->vwap = dec_div(dec_add(dec_mul(dec_from_i64(ap), a->vwap), tv), dec_from_i64(n));
        dec raw = dec_mul(dec_from_i64(n), a->vwap), mkt = dec_mul(dec_from_i64(n), p);
        a->posval = dec_add(raw, cal_profit(is_long, mkt, raw));
        xfer_inc(a, counter, tv);
    } else if (mode == 2) {
        int64_t left = ap - q;
        if (left > 0) {
            a->vwap = dec_div(dec_sub(dec_mul(dec_from_i64(ap), a->vwap), tv), dec_from_i64(left));
            dec raw = dec_mul(dec_from_i64(left), a->vwap), mkt = dec_mul(dec_from_i64(left), p);
            a->posval = dec_add(raw, cal_profit(is_long, mkt, raw));
        } else acc_covered(a, is_long, p);
        xfer_dec(a, counter, tv);
    } else {
        dec mkt = acc_covered(a, is_long, p);
        xfer_dec(a, counter, mkt);
        int64_t nw = q - ap;
        a->posval = dec_mul(dec_from_i64(nw), p); a->vwap = p;
        xfer_inc(a, counter, a->posval);
    }
    a->net_position += own_side == S_BID ? q : -q;   /* _update_net_position (account.py:196-213) */
}

static void settle(market_t* m, int tr, const fill_t* fills, int nf) {  /* trader.py:303-345 */
    for (int i = 0; i < nf; i++) {
        const fill_t* f = &fills[i];
        if (f->counter != tr) {
            process_acc(&m->acc[f->counter], f->qty, f->price, f->init_side ^ 1, 1);
            process_acc(&m->acc[tr], f->qty, f->price, f->init_side, 0);
        } else {
            dec tv = dec_mul(dec_from_i64(f->qty), dec_price(f->price));
            m->acc[tr].hold = dec_sub(m->acc[tr].hold, tv);
            m->acc[tr].cash = dec_add(m->acc[tr].cash, tv);
        }
    }
}
static void escrow_rest(acc_t* a, int32_t price, int32_t qty) {   /* cash_processor.py:15-29 */
    dec v = dec_mul(dec_price(price), dec_from_i64(qty));
    a->cash = dec_sub(a->cash, v); a->hold = dec_add(a->hold, v);
}
static void cancel_cash_transfer(acc_t* a, int32_t price, int32_t qty) {  /* cash_processor.py:85-97 */
    dec v = dec_mul(dec_price(price), dec_from_i64(qty));
    a->hold = dec_sub(a->hold, v); a->cash = dec_add(a->cash, v);
}

/* Trader._order_approved (agent/trader.py:108-151) */
static int order_approved(market_t* m, int tr, int side, int32_t size, int32_t price /* <0 market */) {
    acc_t* a = &m->acc[tr];
    if (dec_sign_cmp0(a->nav) <= 0) return 0;
    int64_t pos = a->net_position, opening;
    if ((side == S_BID && pos >= 0) || (side == S_ASK && pos <= 0)) opening = size;
    else { opening = (int64_t)size - llabs((long long)pos); if (opening < 0) opening = 0; }
    if (opening <= 0) return 1;
    dec est;
    if (price < 0) {
        side_t* opp = &m->side[side ^ 1];
        if (opp->n > 0) est = dec_price(opp->o[0].price);
        else if (m->has_trade) est = dec_price(m->last_trade_price);
        else est = dec_from_i64(1);
    } else est = dec_price(price);
    dec order_val = dec_mul(dec_from_i64(opening), est);
    return dec_cmp(a->cash, order_val) >= 0;
}

/* Trader._get_order_ID (agent/trader.py:254-287). Returns index on the side or -1. */
static int find_own_order(market_t* m, int tr, int side, int type, int32_t price) {
    side_t* sd = &m->side[side];
    int best = -1;
    if (type == T_MODIFY) {
        for (int i = 0; i < sd->n; i++)
            if (sd->o[i].owner == tr && (best < 0 || sd->o[i].timestamp < sd->o[best].timestamp)) best = i;
        return best;
    }
    /* limit / cancel: first own order in order_map insertion order with that price.  All orders
     * of one price sit contiguously in FIFO (= insertion) order, so the first hit in queue order
     * is the first in dict order. */
    for (int i = 0; i < sd->n; i++) if (sd->o[i].owner == tr && sd->o[i].price == price) return i;
    return -1;
}

/* Trader.__modify_limit_order + OrderBook.modify_order (trader.py:219-235, orderbook.py:210-266) */
static void modify_order(market_t* m, int tr, int side, int idx, int32_t new_price, int32_t new_qty,
                         fills_t* fills, int32_t* rest_price, int32_t* rest_qty) {
    side_t* sd = &m->side[side];
    order_t old = sd->o[idx];
    cancel_cash_transfer(&m->acc[tr], old.price, old.qty);
    m->lob_time += 1;
    if (new_price == old.price && new_qty <= old.qty) {
        sd->o[idx].qty = new_qty; sd->o[idx].timestamp = m->lob_time;
        *rest_price = new_price; *rest_qty = new_qty;
        return;
    }
    side_remove(sd, idx);
    int32_t left = match(m, side, new_qty, new_price, fills);
    if (left > 0) {
        order_t o; o.price = new_price; o.qty = left; o.owner = old.owner; o.order_id = old.order_id; o.timestamp = m->lob_time;
        if (side_insert(m, side, o)) { *rest_price = new_price; *rest_qty = left; }
        else m->flags |= CDA_FLAG_BOOK_OVERFLOW;
    }
}

/* Trader.place_order (agent/trader.py:49-106) */
static void place_order(market_t* m, int tr, int type, int side, int32_t size, int32_t price) {
    acc_t* a = &m->acc[tr];
    if (side == S_NONE) return;
    if (!order_approved(m, tr, side, size, type == T_MARKET ? -1 : price)) { a->num_rejected_step += 1; return; }
    if (type == T_MARKET || type == T_LIMIT) a->order_step_placed = 1;
    fills_t fills = {NULL, 0, 0};
    int32_t rest_price = 0, rest_qty = 0;
    if (type == T_MARKET) {
        m->lob_time += 1; m->next_order_id += 1;            /* orderbook.py:39-44 */
        match(m, side, size, -1, &fills);
    } else if (type == T_LIMIT) {
        int idx = find_own_order(m, tr, side, T_LIMIT, price);
        if (idx < 0) {
            m->lob_time += 1; m->next_order_id += 1;
            int32_t left = match(m, side, size, price, &fills);
            if (left > 0) {
                order_t o; o.price = price; o.qty = left; o.owner = tr; o.order_id = m->next_order_id; o.timestamp = m->lob_time;
                if (side_insert(m, side, o)) { rest_price = price; rest_qty = left; }
                else m->flags |= CDA_FLAG_BOOK_OVERFLOW;
            }
        } else modify_order(m, tr, side, idx, price, size, &fills, &rest_price, &rest_qty);
    } else if (type == T_MODIFY) {
        int idx = find_own_order(m, tr, side, T_MODIFY, price);
        if (idx >= 0) modify_order(m, tr, side, idx, price, size, &fills, &rest_price, &rest_qty);
    } else { /* cancel: trader.py:237-252, orderbook.py:196-208 */
        int idx = find_own_order(m, tr, side, T_CANCEL, price);
        if (idx >= 0) {
            order_t old = m->side[side].o[idx];
            m->lob_time += 1;
            side_remove(&m->side[side], idx);
            cancel_cash_transfer(a, old.price, old.qty);
        }
    }
    if (fills.n) settle(m, tr, fills.f, fills.n);
    free(fills.f);
    if (rest_qty > 0) escrow_rest(a, rest_price, rest_qty);
}

/* Exchg_Helper.mark_to_mkt + Calculate.mark_to_mkt (exchg_helper.py:56-66, calculate.py:35-55) */
static void mark_to_mkt(const cda_config* cfg, market_t* m) {
    if (!m->has_trade) return;
    dec p = dec_price(m->last_trade_price);
    m->last_price = m->last_trade_price;
    for (int i = 0; i < cfg->num_agents; i++) {
        acc_t* a = &m->acc[i];
        int64_t ap = llabs((long long)a->net_position);
        dec diff = a->net_position >= 0 ? dec_sub(p, a->vwap) : dec_sub(a->vwap, p);
        dec profit = dec_mul(dec_from_i64(ap), diff);
        dec raw = dec_mul(dec_from_i64(ap), a->vwap);
        a->posval = dec_add(raw, profit);
        a->prev_nav = a->nav;
        a->nav = dec_add(dec_add(a->cash, a->hold), a->posval);
        if (dec_cmp(a->nav, a->max_nav) > 0) a->max_nav = a->nav;
    }
}

/* State_Helper.set_agg_LOB (exchg/state_helper.py:113-214): raw f32[40] and normalised f32[42] */
static void set_agg_lob(const cda_config* cfg, market_t* m, float* snap /* [42] */) {
    double px[2][CDA_K_ROWS], sz[2][CDA_K_ROWS];
    for (int s = 0; s < 2; s++) {
        for (int k = 0; k < CDA_K_ROWS; k++) { px[s][k] = 0.0; sz[s][k] = 0.0; }
        int k = -1; int32_t cur = -1;
        for (int i = 0; i < m->side[s].n; i++) {
            const order_t* o = &m->side[s].o[i];
            if (k < 0 || o->price != cur) { k++; if (k >= CDA_K_ROWS) break; cur = o->price; px[s][k] = (double)o->price; }
            sz[s][k] += (double)o->qty;
        }
    }
    for (int k = 0; k < CDA_K_ROWS; k++) {
        m->raw[k] = (float)px[0][k]; m->raw[10 + k] = (float)sz[0][k];
        /* empty ask levels stay +0.0: the list is only written where a level exists (state_helper.py:141-143) */
        m->raw[20 + k] = px[1][k] != 0 ? (float)(-px[1][k]) : 0.0f; m->raw[30 + k] = sz[1][k] != 0 ? (float)(-sz[1][k]) : 0.0f;
    }
    double l1_bid = px[0][0] > 0 ? px[0][0] : 0.0, l1_ask = px[1][0] != 0 ? px[1][0] : 0.0, M;
    if (l1_bid > 0 && l1_ask > 0) M = (l1_bid + l1_ask) / 2.0;
    else if (l1_bid > 0) M = l1_bid;
    else if (l1_ask > 0) M = l1_ask;
    else { M = (double)m->last_price; if (M <= 0) M = 100.0; }
    for (int k = 0; k < CDA_K_ROWS; k++) {
        double bp = px[0][k], ap = px[1][k];   /* ap = |ask_price_list[k]| */
        snap[k]      = (float)(bp > 0 ? (M - bp) / M : 0.0);
        snap[10 + k] = (float)(sz[0][k] > 0 ? sqrt(sz[0][k]) : 0.0);
        snap[20 + k] = (float)(ap != 0 ? -((ap - M) / M) : 0.0);
        snap[30 + k] = (float)(sz[1][k] != 0 ? -sqrt(sz[1][k]) : 0.0);
    }
    snap[40] = (float)log(M);
    if (l1_bid > 0 && l1_ask > 0) {
        double st = (l1_ask - l1_bid) / (double)cfg->tick_size;
        snap[41] = (float)log1p(st > 0.0 ? st : 0.0);
    } else snap[41] = 0.0f;
}

static void emit_obs(const cda_config* cfg, const market_t* m, float* obs) {
    memcpy(obs, m->hist, sizeof(float) * (size_t)cfg->n_hist * CDA_SNAPSHOT_DIM);
}

static void market_reset(const cda_config* cfg, market_t* m, int have_seed, uint64_t seed, int index, float* obs) {
    if (have_seed) { rng_seed(&m->rng, seed); m->seeded = 1; }
    else if (!m->seeded) { rng_seed(&m->rng, (uint64_t)index); m->seeded = 1; }
    m->side[0].n = m->side[1].n = 0; m->peak_orders = 0;
    m->t_step = 0; m->lob_time = 0; m->next_order_id = 0; m->has_trade = 0; m->last_trade_price = 0;
    m->done_mask = 0; m->flags = 0;
    m->last_price = (int32_t)rng_integers(&m->rng, cfg->initial_price_min, cfg->initial_price_max);
    for (int i = 0; i < cfg->num_agents; i++) {      /* Account.reset_acc (account.py:55-82) */
        acc_t* a = &m->acc[i]; memset(a, 0, sizeof *a);
        a->cash = dec_from_i64(cfg->init_cash); a->hold = dec_zero(); a->posval = dec_zero(); a->vwap = dec_zero();
        a->nav = a->prev_nav = a->max_nav = a->cash;
    }
    float snap[CDA_SNAPSHOT_DIM];
    set_agg_lob(cfg, m, snap);
    for (int h = 0; h < cfg->n_hist; h++) memcpy(m->hist + h * CDA_SNAPSHOT_DIM, snap, sizeof snap);
    if (obs) emit_obs(cfg, m, obs);
}

static float clampf(float v, float lo, float hi) { if (!(v >= lo)) return lo; if (!(v <= hi)) return hi; return v; }
static int32_t clampi(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

typedef struct { int tr, type, side; int32_t size, price; } act_t;

static void market_step(const struct oracle_env* e, market_t* m, int mi,
                        const int32_t* category, const float* size_mean, const float* size_sigma,
                        const int32_t* price, const int32_t* price_offset, const uint8_t* present,
                        float* obs, double* reward, uint8_t* term, uint8_t* trunc,
                        const cda_info_ptrs* info, oracle_trace* trace) {
    const cda_config* cfg = &e->cfg;
    int A = cfg->num_agents;
    float snap[CDA_SNAPSHOT_DIM];
    /* 1. pre-step snapshot (continuousDoubleAuction_env.py:274) */
    set_agg_lob(cfg, m, snap);
    /* 2. set_actions (action_helper.py:145-172, :241-283) */
    act_t acts[CDA_MAX_AGENTS]; int na = 0; uint32_t pass_mask = 0;
    if (info && info->lob_actions) for (int k = 0; k < 4 * A; k++) info->lob_actions[(size_t)mi * (size_t)A * 4 + (size_t)k] = -1;
    /* The reference walks the caller's action dict in ITS iteration order (action_helper.py:164-170): one normal per key in that
     * order, and the arrival list handed to the shuffle is built in that order too.  `present[a]` carries it: 0 = agent a is
     * not in the dict, otherwise agents are visited by ascending present[a] (1 + position in the dict), ties - e.g. a plain
     * 0 / 1 mask - by ascending agent index. */
    int visit[CDA_MAX_AGENTS], nv = 0;
    for (int a = 0; a < A; a++) if (!present || present[a]) visit[nv++] = a;
    if (present) for (int i = 1; i < nv; i++) {                       /* stable insertion sort by present[] */
        int v = visit[i], j = i;
        while (j > 0 && present[visit[j - 1]] > present[v]) { visit[j] = visit[j - 1]; j--; }
        visit[j] = v;
    }
    for (int vi = 0; vi < nv; vi++) {
        const int a = visit[vi];
        int cat = clampi(category[a], 0, 8);
        int side = cat == 0 ? S_NONE : (cat <= 4 ? S_BID : S_ASK);
        int type = cat == 0 ? T_MARKET : (cat - 1) & 3;
        float mean = clampf(size_mean[a], -1.0f, 1.0f), sigma = clampf(size_sigma[a], 0.0f, 1.0f);
        float locf = (type == T_MARKET ? e->mkt_mul : e->lim_mul) * mean;     /* float32 product (NEP 50) */
        double z = rng_std_normal(&m->rng);
        double prod = (double)sigma * z;
        double sample = (double)locf + prod;                                   /* no FMA: -ffp-contract=off */
        double rs = rint(fabs(sample));
        int32_t size = (int32_t)rs + cfg->min_size;
        int32_t pr = -1;
        if (type != T_MARKET) {
            int level = clampi(price[a], 0, CDA_K_ROWS - 1), off = clampi(price_offset[a], 0, 2) - 1;
            if (side == S_BID) {
                int32_t p = (int32_t)m->raw[level];
                int32_t base = p == 0 ? m->last_price - (level + 1) * cfg->tick_size : (p < 0 ? -p : p);
                pr = base + off * cfg->tick_size;
            } else {
                int32_t p = (int32_t)fabsf(m->raw[20 + level]);
                int32_t base = p == 0 ? m->last_price + (level + 1) * cfg->tick_size : p;
                pr = base - off * cfg->tick_size;
            }
            if (pr < cfg->tick_size) pr = cfg->tick_size;
        }
        if (trace) { trace->z[a] = z; trace->dec_type[a] = type; trace->dec_side[a] = side; trace->dec_size[a] = size; trace->dec_price[a] = pr; }
        if (info && info->lob_actions && side != S_NONE) {                     /* env.LOB_actions (continuousDoubleAuction_env.py:284-285) */
            int32_t* la = info->lob_actions + ((size_t)mi * (size_t)A + (size_t)a) * 4;
            la[0] = side; la[1] = type; la[2] = size; la[3] = pr;
        }
        if (side != S_NONE) { acts[na].tr = a; acts[na].type = type; acts[na].side = side; acts[na].size = size; acts[na].price = pr; na++; }
        else pass_mask |= 1u << a;
    }
    /* 3. rand_exec_seq (action_helper.py:174-199) */
    int perm[CDA_MAX_AGENTS]; rng_permutation(&m->rng, na, perm);
    if (trace) { trace->n_acts = na; for (int i = 0; i < na; i++) trace->exec_order[i] = acts[perm[i]].tr; }
    /* 4. do_actions (action_helper.py:201-239) */
    for (int i = 0; i < na; i++) { const act_t* c = &acts[perm[i]]; place_order(m, c->tr, c->type, c->side, c->size, c->price); }
    /* 5. mark_to_mkt */
    mark_to_mkt(cfg, m);
    /* 6. prep_next_state (state_helper.py:80-92) */
    set_agg_lob(cfg, m, snap);
    memmove(m->hist, m->hist + CDA_SNAPSHOT_DIM, sizeof(float) * (size_t)(cfg->n_hist - 1) * CDA_SNAPSHOT_DIM);
    memcpy(m->hist + (cfg->n_hist - 1) * CDA_SNAPSHOT_DIM, snap, sizeof snap);
    if (obs) emit_obs(cfg, m, obs);
    /* 7. set_step_outputs (exchg_helper.py:93-124) */
    double best_bid = m->side[0].n ? (double)m->side[0].o[0].price : NAN;
    double best_ask = m->side[1].n ? (double)m->side[1].o[0].price : NAN;
    if (info) {
        if (info->last_price) info->last_price[mi] = (double)m->last_price;
        if (info->best_bid) info->best_bid[mi] = best_bid;
        if (info->best_ask) info->best_ask[mi] = best_ask;
        if (info->spread) info->spread[mi] = (m->side[0].n && m->side[1].n) ? best_ask - best_bid : NAN;
    }
    for (int a = 0; a < A; a++) {
        acc_t* ac = &m->acc[a];
        /* Reward_Helper.set_reward (reward_helper.py:35-102) */
        double nav_change = dec_to_double(dec_sub(ac->nav, ac->prev_nav));
        double nav_term = nav_change * (nav_change < 0 ? cfg->loss_multiplier : 1.0);
        dec dd = dec_sub(ac->max_nav, ac->nav);
        double drawdown = dec_sign_cmp0(dd) > 0 ? dec_to_double(dd) : 0.0;
        double t[5];
        t[0] = nav_term;
        t[1] = -(cfg->order_penalty * (double)ac->order_step_placed);
        t[2] = -(cfg->trade_penalty * (double)ac->num_trades_step);
        t[3] = -(cfg->drawdown_penalty * drawdown);
        t[4] = cfg->passive_bonus * (double)ac->num_passive_fills_step;
        double r = 0.0; for (int k = 0; k < 5; k++) r += t[k];
        if (reward) reward[a] = r;
        /* Done_Helper.set_done (done_helper.py:3-18) */
        if (dec_sign_cmp0(ac->nav) <= 0) m->done_mask |= 1u << a;
        if (info) {
            size_t ix = (size_t)mi * (size_t)A + (size_t)a;
            if (info->nav) info->nav[ix] = dec_pack(ac->nav, &m->flags);
            if (info->num_trades) info->num_trades[ix] = ac->num_trades;
            if (info->net_position) info->net_position[ix] = ac->net_position;
            if (info->vwap) info->vwap[ix] = dec_to_double(ac->vwap);
            if (info->cash) info->cash[ix] = dec_to_double(ac->cash);
            if (info->cash_on_hold) info->cash_on_hold[ix] = dec_to_double(ac->hold);
            if (info->position_val) info->position_val[ix] = dec_to_double(ac->posval);
            if (info->drawdown) info->drawdown[ix] = drawdown;
            if (info->max_nav) info->max_nav[ix] = dec_to_double(ac->max_nav);
            if (info->num_trades_step) info->num_trades_step[ix] = ac->num_trades_step;
            if (info->num_passive_fills_step) info->num_passive_fills_step[ix] = ac->num_passive_fills_step;
            if (info->order_step_placed) info->order_step_placed[ix] = ac->order_step_placed;
            if (info->num_rejected_step) info->num_rejected_step[ix] = ac->num_rejected_step;
            if (info->is_pass_action) info->is_pass_action[ix] = (uint8_t)((pass_mask >> a) & 1u);
            if (info->reward_terms) for (int k = 0; k < 5; k++) info->reward_terms[ix * 5 + (size_t)k] = t[k];
        }
        ac->num_trades_step = 0; ac->num_passive_fills_step = 0; ac->order_step_placed = 0; ac->num_rejected_step = 0;
    }
    /* Done_Helper.set_all_done (done_helper.py:20-54) */
    int ndone = __builtin_popcount(m->done_mask);
    if (term) *term = (uint8_t)(ndone == A);
    if (trunc) *trunc = (uint8_t)(m->t_step + 1 >= cfg->max_step);
    m->t_step += 1;
}

/* ======================================================================================
 * exported API (host pointers everywhere)
 * ==================================================================================== */
static int cfg_ok(const cda_config* c) {
    if (c->num_agents < 1 || c->num_agents > CDA_MAX_AGENTS) return CDA_ERR_INVALID;
    if (c->n_hist < 1 || c->n_hist > CDA_MAX_HIST) return CDA_ERR_INVALID;
    if (c->tick_size < 1 || c->tick_size > CDA_TICK_MAX) return CDA_ERR_UNSUPPORTED;
    if (c->initial_price_max < c->initial_price_min) return CDA_ERR_INVALID;
    if (c->min_size < 0 || c->mkt_max_size < c->min_size || c->limit_size_multiple < 1) return CDA_ERR_INVALID;
    if (c->book_capacity != 0 && c->book_capacity != CDA_BOOK_CAP && c->book_capacity != CDA_BOOK_CAP_MAX) return CDA_ERR_INVALID;
    if (c->book_spill < -1 || c->book_spill > CDA_SPILL_MAX) return CDA_ERR_INVALID;
    return CDA_OK;
}

int oracle_create(const cda_config* cfg, int32_t n_markets, oracle_env** out) {
    oracle_init();
    if (!cfg || !out || n_markets < 1) return CDA_ERR_INVALID;
    int rc = cfg_ok(cfg); if (rc) return rc;
    oracle_env* e = (oracle_env*)calloc(1, sizeof *e);
    if (!e) return CDA_ERR_NOMEM;
    e->cfg = *cfg; e->n = n_markets;
    /* the reference's book is unbounded (ordertree.py:5-58) and so is the product's with its HBM tier; only an env built
     * WITHOUT that tier (book_spill = -1) holds just its tile, which the oracle then mirrors together with the overflow flag */
    e->book_cap = cfg->book_spill >= 0 ? 0 : (cfg->book_capacity ? cfg->book_capacity : (cfg->num_agents <= 8 ? CDA_BOOK_CAP : CDA_BOOK_CAP_MAX));
    e->mkt_mul = (float)((double)(cfg->mkt_max_size - cfg->min_size) / 2.0);
    e->lim_mul = (float)((double)((int64_t)cfg->mkt_max_size * (int64_t)cfg->limit_size_multiple - (int64_t)cfg->min_size) / 2.0);
    e->m = (market_t*)calloc((size_t)n_markets, sizeof(market_t));
    if (!e->m) { free(e); return CDA_ERR_NOMEM; }
    for (int i = 0; i < n_markets; i++) { e->m[i].book_cap = e->book_cap; market_reset(cfg, &e->m[i], 0, 0, i, NULL); }
    for (int i = 0; i < n_markets; i++) e->m[i].seeded = 0;   /* construction does not count as seeding */
    *out = e; return CDA_OK;
}
int oracle_destroy(oracle_env* e) {
    if (e) { for (int i = 0; i < e->n; i++) { free(e->m[i].side[0].o); free(e->m[i].side[1].o); } free(e->m); free(e); }
    return CDA_OK;
}
/* 0 = unbounded book (the reference); 256 / 512 mirror the product's pools and their overflow flag */
int oracle_set_book_cap(oracle_env* e, int32_t cap) {
    if (!e || cap < 0) return CDA_ERR_INVALID;
    e->book_cap = cap;
    for (int i = 0; i < e->n; i++) e->m[i].book_cap = cap;
    return CDA_OK;
}
int oracle_book_peak(oracle_env* e, int32_t* peak_out) {
    if (!e || !peak_out) return CDA_ERR_INVALID;
    for (int i = 0; i < e->n; i++) peak_out[i] = e->m[i].peak_orders;
    return CDA_OK;
}
int oracle_book_size(oracle_env* e, int32_t market, int32_t* n_bids, int32_t* n_asks) {
    if (!e || market < 0 || market >= e->n) return CDA_ERR_INVALID;
    if (n_bids) *n_bids = e->m[market].side[0].n;
    if (n_asks) *n_asks = e->m[market].side[1].n;
    return CDA_OK;
}

int oracle_reset(oracle_env* e, const uint64_t* seeds, const uint8_t* mask, float* obs_out) {
    if (!e) return CDA_ERR_INVALID;
    size_t od = (size_t)e->cfg.n_hist * CDA_SNAPSHOT_DIM;
    for (int i = 0; i < e->n; i++) {
        if (mask && !mask[i]) continue;
        market_reset(&e->cfg, &e->m[i], seeds != NULL, seeds ? seeds[i] : 0, i, obs_out ? obs_out + od * (size_t)i : NULL);
    }
    return CDA_OK;
}

int oracle_step_range(oracle_env* e, int32_t first, int32_t count,
                      const int32_t* category, const float* size_mean, const float* size_sigma,
                      const int32_t* price, const int32_t* price_offset, const uint8_t* present,
                      float* obs_out, double* reward_out, uint8_t* terminated_out, uint8_t* truncated_out,
                      const cda_info_ptrs* info_out, oracle_trace* trace) {
    if (!e || first < 0 || count < 0 || first + count > e->n) return CDA_ERR_INVALID;
    int A = e->cfg.num_agents; size_t od = (size_t)e->cfg.n_hist * CDA_SNAPSHOT_DIM;
    for (int i = first; i < first + count; i++) {
        size_t o = (size_t)i * (size_t)A;
        market_step(e, &e->m[i], i, category + o, size_mean + o, size_sigma + o, price + o, price_offset + o,
                    present ? present + o : NULL,
                    obs_out ? obs_out + od * (size_t)i : NULL, reward_out ? reward_out + o : NULL,
                    terminated_out ? terminated_out + i : NULL, truncated_out ? truncated_out + i : NULL,
                    info_out, trace ? trace + i : NULL);
    }
    return CDA_OK;
}
int oracle_step(oracle_env* e,
                const int32_t* category, const float* size_mean, const float* size_sigma,
                const int32_t* price, const int32_t* price_offset, const uint8_t* present,
                float* obs_out, double* reward_out, uint8_t* terminated_out, uint8_t* truncated_out,
                const cda_info_ptrs* info_out, oracle_trace* trace) {
    if (!e) return CDA_ERR_INVALID;
    return oracle_step_range(e, 0, e->n, category, size_mean, size_sigma, price, price_offset, present,
                             obs_out, reward_out, terminated_out, truncated_out, info_out, trace);
}

/* cpu_baseline helper: n_steps steps of markets [first, first+count) cycling through T pre-generated
 * action steps laid out [T,N,A]; one call per worker thread (no per-step FFI overhead). */
int oracle_run_range(oracle_env* e, int32_t first, int32_t count, int32_t n_steps, int32_t T,
                     const int32_t* category, const float* size_mean, const float* size_sigma,
                     const int32_t* price, const int32_t* price_offset,
                     float* obs_out, double* reward_out, uint8_t* terminated_out, uint8_t* truncated_out) {
    if (!e || T < 1) return CDA_ERR_INVALID;
    size_t per = (size_t)e->n * (size_t)e->cfg.num_agents;
    for (int s = 0; s < n_steps; s++) {
        size_t o = per * (size_t)(s % T);
        int rc = oracle_step_range(e, first, count, category + o, size_mean + o, size_sigma + o, price + o, price_offset + o, NULL,
                                   obs_out, reward_out, terminated_out, truncated_out, NULL, NULL);
        if (rc) return rc;
    }
    return CDA_OK;
}

/* The random-agent driver on the CPU (CDA_rand.py:40-85 with the counter-based sampler of include/cda_random_agents.h):
 * markets [first, first+count) play n_steps steps each; the action of (market i, step step0 + s, agent a) is
 * cda_random_action(action_seed, market_index_base + i, step0 + s, a) - the stream cda_random_actions() puts in HBM for
 * the GPU leg of bench.py, so both legs consume identical inputs.  Outputs are the last step's (global indexing). */
int oracle_run_random_range_info(oracle_env* e, int32_t first, int32_t count, int32_t step0, int32_t n_steps,
                                 uint64_t action_seed, uint64_t market_index_base,
                                 float* obs_out, double* reward_out, uint8_t* terminated_out, uint8_t* truncated_out, const cda_info_ptrs* info);
int oracle_run_random_range(oracle_env* e, int32_t first, int32_t count, int32_t step0, int32_t n_steps,
                            uint64_t action_seed, uint64_t market_index_base,
                            float* obs_out, double* reward_out, uint8_t* terminated_out, uint8_t* truncated_out) {
    return oracle_run_random_range_info(e, first, count, step0, n_steps, action_seed, market_index_base, obs_out, reward_out, terminated_out, truncated_out, NULL);
}
/* ... with every info tensor of Info_Helper.set_info built each step (info_helper.py:30-116) - what the GPU headline leg of bench.py builds */
int oracle_run_random_range_info(oracle_env* e, int32_t first, int32_t count, int32_t step0, int32_t n_steps,
                                 uint64_t action_seed, uint64_t market_index_base,
                                 float* obs_out, double* reward_out, uint8_t* terminated_out, uint8_t* truncated_out, const cda_info_ptrs* info) {
    if (!e || first < 0 || count < 0 || first + count > e->n || step0 < 0 || n_steps < 0) return CDA_ERR_INVALID;
    const int A = e->cfg.num_agents; const size_t od = (size_t)e->cfg.n_hist * CDA_SNAPSHOT_DIM;
    for (int i = first; i < first + count; i++) {
        int32_t cat[CDA_MAX_AGENTS], pr[CDA_MAX_AGENTS], po[CDA_MAX_AGENTS]; float sm[CDA_MAX_AGENTS], ss[CDA_MAX_AGENTS];
        const size_t o = (size_t)i * (size_t)A;
        for (int s = 0; s < n_steps; s++) {
            for (int a = 0; a < A; a++) cda_random_action(action_seed, market_index_base + (uint64_t)i, (uint32_t)(step0 + s), (uint32_t)a, &cat[a], &sm[a], &ss[a], &pr[a], &po[a]);
            market_step(e, &e->m[i], i, cat, sm, ss, pr, po, NULL, obs_out ? obs_out + od * (size_t)i : NULL, reward_out ? reward_out + o : NULL,
                        terminated_out ? terminated_out + i : NULL, truncated_out ? truncated_out + i : NULL, info, NULL);
        }
    }
    return CDA_OK;
}

int oracle_place_order(oracle_env* e, int32_t market, int32_t trader, int32_t type, int32_t side, int32_t size, int32_t price) {
    if (!e || market < 0 || market >= e->n || trader < 0 || trader >= e->cfg.num_agents) return CDA_ERR_INVALID;
    if (type < 0 || type > 3 || side < 0 || side > 1 || size < 1) return CDA_ERR_INVALID;
    place_order(&e->m[market], trader, type, side, size, price);
    return CDA_OK;
}
int oracle_mark_to_mkt(oracle_env* e, int32_t market) {
    if (!e || market < 0 || market >= e->n) return CDA_ERR_INVALID;
    mark_to_mkt(&e->cfg, &e->m[market]); return CDA_OK;
}

int oracle_get_state(oracle_env* e, int32_t market, cda_market_state* s) {
    if (!e || !s || market < 0 || market >= e->n) return CDA_ERR_INVALID;
    market_t* m = &e->m[market];
    memset(s, 0, sizeof *s);
    s->rng_state_hi = (uint64_t)(m->rng.state >> 64); s->rng_state_lo = (uint64_t)m->rng.state;
    s->rng_inc_hi = (uint64_t)(m->rng.inc >> 64); s->rng_inc_lo = (uint64_t)m->rng.inc;
    s->rng_has_uint32 = m->rng.has_uint32; s->rng_uinteger = m->rng.uinteger;
    s->t_step = m->t_step; s->lob_time = m->lob_time; s->next_order_id = m->next_order_id;
    s->last_price = m->last_price; s->has_trade = m->has_trade; s->last_trade_price = m->last_trade_price;
    s->done_mask = m->done_mask; s->flags = m->flags;
    s->n_bids = m->side[0].n; s->n_asks = m->side[1].n;
    for (int sd = 0; sd < 2; sd++) for (int i = 0; i < m->side[sd].n && i < CDA_BOOK_CAP_MAX; i++) {   /* (an unbounded book's tail beyond the struct's capacity is not dumped) */
        cda_order* o = sd == 0 ? &s->bids[i] : &s->asks[i]; const order_t* q = &m->side[sd].o[i];
        o->price = q->price; o->qty = q->qty; o->owner = q->owner; o->order_id = q->order_id; o->timestamp = q->timestamp;
    }
    for (int a = 0; a < e->cfg.num_agents; a++) {
        cda_account_state* o = &s->acc[a]; const acc_t* q = &m->acc[a];
        o->cash = dec_pack(q->cash, &s->flags); o->cash_on_hold = dec_pack(q->hold, &s->flags);
        o->position_val = dec_pack(q->posval, &s->flags); o->vwap = dec_pack(q->vwap, &s->flags);
        o->nav = dec_pack(q->nav, &s->flags); o->prev_nav = dec_pack(q->prev_nav, &s->flags); o->max_nav = dec_pack(q->max_nav, &s->flags);
        o->net_position = q->net_position; o->num_trades = q->num_trades;
        o->num_trades_step = q->num_trades_step; o->num_passive_fills_step = q->num_passive_fills_step;
        o->order_step_placed = q->order_step_placed; o->num_rejected_step = q->num_rejected_step;
    }
    memcpy(s->hist, m->hist, sizeof(float) * (size_t)e->cfg.n_hist * CDA_SNAPSHOT_DIM);
    return CDA_OK;
}
/* same rule as cda_set_state: a side longer than the struct's arrays can only go back onto the market it was dumped from
 * (both lengths equal the market's: the book stays in place) */
int oracle_set_state(oracle_env* e, int32_t market, const cda_market_state* s) {
    if (!e || !s || market < 0 || market >= e->n) return CDA_ERR_INVALID;
    if (s->n_bids < 0 || s->n_asks < 0) return CDA_ERR_INVALID;
    market_t* m = &e->m[market];
    const int keep_book = s->n_bids > CDA_BOOK_CAP_MAX || s->n_asks > CDA_BOOK_CAP_MAX;
    if (keep_book && (s->n_bids != m->side[0].n || s->n_asks != m->side[1].n)) return CDA_ERR_INVALID;
    if (m->book_cap > 0 && s->n_bids + s->n_asks > m->book_cap) return CDA_ERR_INVALID;
    m->rng.state = ((u128)s->rng_state_hi << 64) | s->rng_state_lo; m->rng.inc = ((u128)s->rng_inc_hi << 64) | s->rng_inc_lo;
    m->rng.has_uint32 = s->rng_has_uint32; m->rng.uinteger = s->rng_uinteger; m->seeded = 1;
    m->t_step = s->t_step; m->lob_time = s->lob_time; m->next_order_id = s->next_order_id;
    m->last_price = s->last_price; m->has_trade = s->has_trade; m->last_trade_price = s->last_trade_price;
    m->done_mask = s->done_mask; m->flags = s->flags;
    for (int sd = 0; sd < 2 && !keep_book; sd++) {
        const int need = sd == 0 ? s->n_bids : s->n_asks; side_t* q = &m->side[sd];
        if (q->alloc < need) { q->alloc = need; q->o = (order_t*)realloc(q->o, (size_t)need * sizeof(order_t)); if (!q->o) abort(); }
        q->n = need;
    }
    if (!keep_book && m->side[0].n + m->side[1].n > m->peak_orders) m->peak_orders = m->side[0].n + m->side[1].n;
    for (int sd = 0; sd < 2 && !keep_book; sd++) for (int i = 0; i < m->side[sd].n; i++) {
        const cda_order* o = sd == 0 ? &s->bids[i] : &s->asks[i]; order_t* q = &m->side[sd].o[i];
        q->price = o->price; q->qty = o->qty; q->owner = o->owner; q->order_id = o->order_id; q->timestamp = o->timestamp;
    }
    for (int a = 0; a < e->cfg.num_agents; a++) {
        const cda_account_state* o = &s->acc[a]; acc_t* q = &m->acc[a];
        q->cash = dec_unpack(o->cash); q->hold = dec_unpack(o->cash_on_hold); q->posval = dec_unpack(o->position_val);
        q->vwap = dec_unpack(o->vwap); q->nav = dec_unpack(o->nav); q->prev_nav = dec_unpack(o->prev_nav); q->max_nav = dec_unpack(o->max_nav);
        q->net_position = o->net_position; q->num_trades = o->num_trades;
        q->num_trades_step = o->num_trades_step; q->num_passive_fills_step = o->num_passive_fills_step;
        q->order_step_placed = o->order_step_placed; q->num_rejected_step = o->num_rejected_step;
    }
    memcpy(m->hist, s->hist, sizeof(float) * (size_t)e->cfg.n_hist * CDA_SNAPSHOT_DIM);
    return CDA_OK;
}
/* one whole side in queue order (cda_get_book) */
int oracle_get_book(oracle_env* e, int32_t market, int32_t side, cda_order* out, int32_t max_orders, int32_t* n_out) {
    if (!e || market < 0 || market >= e->n || side < 0 || side > 1 || max_orders < 0 || !n_out) return CDA_ERR_INVALID;
    const side_t* sd = &e->m[market].side[side];
    *n_out = sd->n;
    for (int i = 0; i < sd->n && i < max_orders; i++) {
        const order_t* q = &sd->o[i];
        out[i].price = q->price; out[i].qty = q->qty; out[i].owner = q->owner; out[i].order_id = q->order_id; out[i].timestamp = q->timestamp;
    }
    return CDA_OK;
}
int oracle_get_raw_snapshot(oracle_env* e, float* raw_out) {
    if (!e || !raw_out) return CDA_ERR_INVALID;
    float snap[CDA_SNAPSHOT_DIM];
    for (int i = 0; i < e->n; i++) { set_agg_lob(&e->cfg, &e->m[i], snap); memcpy(raw_out + (size_t)i * CDA_RAW_DIM, e->m[i].raw, sizeof(float) * CDA_RAW_DIM); }
    return CDA_OK;
}
int oracle_last_flags(oracle_env* e, uint32_t* flags_out) {
    if (!e || !flags_out) return CDA_ERR_INVALID;
    for (int i = 0; i < e->n; i++) flags_out[i] = e->m[i].flags;
    return CDA_OK;
}

int oracle_dec_op(int32_t op, int32_t n, const cda_dec* a, const cda_dec* b, cda_dec* out) {
    oracle_init();
    for (int i = 0; i < n; i++) {
        dec x = dec_unpack(a[i]), y = b ? dec_unpack(b[i]) : dec_zero(), r;
        cda_dec o; memset(&o, 0, sizeof o);
        switch (op) {
            case 0: r = dec_add(x, y); o = dec_pack(r, NULL); break;
            case 1: r = dec_sub(x, y); o = dec_pack(r, NULL); break;
            case 2: r = dec_mul(x, y); o = dec_pack(r, NULL); break;
            case 3: r = dec_div(x, y); o = dec_pack(r, NULL); break;
            case 4: o.w[0] = (uint32_t)(dec_cmp(x, y) + 1); break;
            case 5: { double d = dec_to_double(x); uint64_t bits; memcpy(&bits, &d, 8); o.w[0] = (uint32_t)bits; o.w[1] = (uint32_t)(bits >> 32); break; }
            default: return CDA_ERR_INVALID;
        }
        out[i] = o;
    }
    return CDA_OK;
}
/* the host's own libm (what numpy calls): op 0 log1p, 1 exp, 2 log - the yardstick of the restated functions in csrc/cda_libm.hpp */
int oracle_libm(int32_t op, int64_t n, const double* x, double* y) {
    if (!x || !y || op < 0 || op > 2) return CDA_ERR_INVALID;
    for (int64_t i = 0; i < n; i++) y[i] = op == 0 ? log1p(x[i]) : (op == 1 ? exp(x[i]) : log(x[i]));
    return CDA_OK;
}
int oracle_dec_str(const cda_dec* a, char* out, int32_t cap) {
    oracle_init();
    char s[128]; dec_to_str(dec_unpack(*a), s);
    if ((int32_t)strlen(s) + 1 > cap) return CDA_ERR_INVALID;
    strcpy(out, s); return CDA_OK;
}

int oracle_rng(uint64_t seed, int32_t lo, int32_t hi, int32_t n_steps, int32_t n_normals, int32_t perm_n,
               int32_t* first_int, double* normals, int32_t* perms, uint64_t* final_state) {
    if (perm_n > 64 || perm_n < 0) return CDA_ERR_INVALID;
    rng_t r; rng_seed(&r, seed);
    if (final_state) {   /* state right after seeding is reported in slots 6..9 */
        final_state[6] = (uint64_t)(r.state >> 64); final_state[7] = (uint64_t)r.state;
        final_state[8] = (uint64_t)(r.inc >> 64); final_state[9] = (uint64_t)r.inc;
    }
    *first_int = (int32_t)rng_integers(&r, lo, hi);
    int perm[64];
    for (int s = 0; s < n_steps; s++) {
        for (int k = 0; k < n_normals; k++) normals[(size_t)s * (size_t)n_normals + (size_t)k] = rng_std_normal(&r);
        rng_permutation(&r, perm_n, perm);
        for (int k = 0; k < perm_n; k++) perms[(size_t)s * (size_t)perm_n + (size_t)k] = perm[k];
    }
    if (final_state) {
        final_state[0] = (uint64_t)(r.state >> 64); final_state[1] = (uint64_t)r.state;
        final_state[2] = (uint64_t)(r.inc >> 64); final_state[3] = (uint64_t)r.inc;
        final_state[4] = r.has_uint32; final_state[5] = r.uinteger;
    }
    return CDA_OK;
}
