/*
 * cda_oracle.h - CPU ORACLE (test infrastructure, NOT the product). See cda_oracle.c.
 * Same shapes as include/cda.h, `oracle_` prefix, HOST pointers everywhere.
 */
#ifndef CDA_ORACLE_H
#define CDA_ORACLE_H
#include "../include/cda.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_env oracle_env;

/* Per-market intermediate values of one step (for localising a mismatch against the goldens). */
typedef struct oracle_trace {
    double  z[CDA_MAX_AGENTS];          /* the standard normal drawn for each agent        */
    int32_t dec_type[CDA_MAX_AGENTS];   /* decoded order: 0 market 1 limit 2 modify 3 cancel */
    int32_t dec_side[CDA_MAX_AGENTS];   /* 0 bid, 1 ask, 2 none                            */
    int32_t dec_size[CDA_MAX_AGENTS];
    int32_t dec_price[CDA_MAX_AGENTS];  /* ticks, -1 = market                              */
    int32_t n_acts;
    int32_t exec_order[CDA_MAX_AGENTS]; /* trader index in execution (shuffled) order      */
} oracle_trace;

int oracle_create(const cda_config* cfg, int32_t n_markets, oracle_env** out);
int oracle_destroy(oracle_env* env);
int oracle_reset(oracle_env* env, const uint64_t* seeds, const uint8_t* mask, float* obs_out);
int oracle_step(oracle_env* env,
                const int32_t* category, const float* size_mean, const float* size_sigma,
                const int32_t* price, const int32_t* price_offset, const uint8_t* present,
                float* obs_out, double* reward_out, uint8_t* terminated_out, uint8_t* truncated_out,
                const cda_info_ptrs* info_out, oracle_trace* trace /* [N] or NULL */);
/* Same, for markets [first, first+count) only (lets the cpu_baseline partition markets over threads).
 * All array arguments are still indexed by the GLOBAL market index. */
int oracle_step_range(oracle_env* env, int32_t first, int32_t count,
                      const int32_t* category, const float* size_mean, const float* size_sigma,
                      const int32_t* price, const int32_t* price_offset, const uint8_t* present,
                      float* obs_out, double* reward_out, uint8_t* terminated_out, uint8_t* truncated_out,
                      const cda_info_ptrs* info_out, oracle_trace* trace);
int oracle_run_range(oracle_env* env, int32_t first, int32_t count, int32_t n_steps, int32_t T,
                     const int32_t* category, const float* size_mean, const float* size_sigma,
                     const int32_t* price, const int32_t* price_offset,
                     float* obs_out, double* reward_out, uint8_t* terminated_out, uint8_t* truncated_out);
/* Random agents of include/cda_random_agents.h for markets [first, first+count), steps step0 .. step0+n_steps-1: the
 * CPU leg of bench.py (the same action stream the GPU leg reads from HBM) and the replay side of the parity tests. */
int oracle_run_random_range(oracle_env* env, int32_t first, int32_t count, int32_t step0, int32_t n_steps,
                            uint64_t action_seed, uint64_t market_index_base,
                            float* obs_out, double* reward_out, uint8_t* terminated_out, uint8_t* truncated_out);
/* the same with every info tensor built each step (nullable info = oracle_run_random_range) */
int oracle_run_random_range_info(oracle_env* env, int32_t first, int32_t count, int32_t step0, int32_t n_steps,
                                 uint64_t action_seed, uint64_t market_index_base,
                                 float* obs_out, double* reward_out, uint8_t* terminated_out, uint8_t* truncated_out, const cda_info_ptrs* info);
/* Book capacity of the oracle: 0 = unbounded, as the reference's OrderTree (ordertree.py:5-58) - the default, and what the
 * product is with its HBM tier; an env created with cda_config.book_spill = -1 (the product without that tier) mirrors the
 * tile pool (cda_config.book_capacity; 256 up to 8 agents, 512 above) with its overflow flag.  oracle_book_peak: most resting orders each market
 * has held since its last reset; oracle_book_size: current orders per side (get_state dumps at most CDA_BOOK_CAP_MAX per side). */
int oracle_set_book_cap(oracle_env* env, int32_t cap);
int oracle_book_peak(oracle_env* env, int32_t* peak_out /* [N] */);
int oracle_book_size(oracle_env* env, int32_t market, int32_t* n_bids, int32_t* n_asks);
int oracle_get_book(oracle_env* env, int32_t market, int32_t side, cda_order* out, int32_t max_orders, int32_t* n_out);
int oracle_place_order(oracle_env* env, int32_t market, int32_t trader, int32_t type, int32_t side,
                       int32_t size, int32_t price);
int oracle_mark_to_mkt(oracle_env* env, int32_t market);
int oracle_get_state(oracle_env* env, int32_t market, cda_market_state* out);
int oracle_set_state(oracle_env* env, int32_t market, const cda_market_state* in);
int oracle_get_raw_snapshot(oracle_env* env, float* raw_out /* [N,40] */);
int oracle_last_flags(oracle_env* env, uint32_t* flags_out);

/* op: 0 add, 1 sub, 2 mul, 3 div, 4 cmp (out.w[0] = 0 lt / 1 eq / 2 gt), 5 to-double (bits in w[0..1]) */
int oracle_dec_op(int32_t op, int32_t n, const cda_dec* a, const cda_dec* b, cda_dec* out);
int oracle_dec_str(const cda_dec* a, char* out, int32_t cap);
/* the host's libm: op 0 log1p, 1 exp, 2 log */
int oracle_libm(int32_t op, int64_t n, const double* x, double* y);
/* final_state: [0..5] = state_hi, state_lo, inc_hi, inc_lo, has_uint32, uinteger after the draws;
 * [6..9] = state_hi, state_lo, inc_hi, inc_lo right after seeding. */
int oracle_rng(uint64_t seed, int32_t lo, int32_t hi, int32_t n_steps, int32_t n_normals, int32_t perm_n,
               int32_t* first_int, double* normals, int32_t* perms, uint64_t* final_state /* [10] */);

#ifdef __cplusplus
}
#endif
#endif
