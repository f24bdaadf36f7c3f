/*
 * cda.h - C-ABI of the MI355X-native vectorised continuous-double-auction environment.
 *
 * This is the drop-in boundary for ONE hot path of ChuaCheowHuan/gym-continuousDoubleAuction:
 * reset()/step() of the multi-agent limit-order-book env, batched over N independent markets.
 * The reference has no FFI; its boundary is the Python class
 *   gym_continuousDoubleAuction/envs/continuousDoubleAuction_env.py:21  (continuousDoubleAuctionEnv)
 * bound at gym_continuousDoubleAuction/train/train.py:441-443 (tune.register_env) and
 * gym_continuousDoubleAuction/__init__.py:18-21 (gymnasium.register).  Each entry point below
 * names the reference method it replaces.  A maintainer binds it with ctypes;
 * the in-tree Python host side (gym_continuousdoubleauction_amd/vec_env.py, env.py) is that binding.
 *
 * Conventions
 *  - every function returns CDA_OK (0) or a negative cda_status; nothing throws or exits;
 *  - array arguments are caller-owned DEVICE pointers (e.g. torch tensors' data_ptr()) unless the
 *    name ends in _host; the library owns only the opaque cda_env arena;
 *  - kernels are enqueued on the caller's HIP stream (`stream` is a hipStream_t, NULL = default
 *    stream) and are asynchronous; calls on one cda_env are not re-entrant;
 *  - there is NO CPU fallback: cda_create fails with CDA_ERR_NO_DEVICE when no gfx950 GPU is present.
 */
#ifndef CDA_H
#define CDA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Observation layout: config/tunable_constants.json:8-10 (k_rows 10, book_rows 4, extra_dim 2). */
#define CDA_K_ROWS        10
#define CDA_BOOK_ROWS     4
#define CDA_EXTRA_DIM     2
#define CDA_SNAPSHOT_DIM  42            /* book_rows*k_rows + extra_dim */
#define CDA_RAW_DIM       40            /* agg_LOB_raw: state_helper.py:159-160 */
#define CDA_MAX_HIST      16            /* n_hist upper bound of this build (reference default 4) */
#define CDA_TICK_MAX      65536         /* tick_size upper bound: prices live below 2^24, a ladder of ten levels of the largest tick still fits */
#define CDA_MAX_AGENTS    16            /* agents per market upper bound of this build */
#define CDA_BOOK_CAP      256           /* default LDS book tile: the top of a market's book, both sides together (cda_config.book_capacity) */
#define CDA_BOOK_CAP_MAX  512           /* the larger compiled tile; also sizes the arrays of the fixed-size parity dump (cda_market_state) */
#define CDA_SPILL_MIN     64            /* smallest / largest HBM spill ring per side (cda_config.book_spill, a power of two) */
#define CDA_SPILL_MAX     (1 << 20)
#define CDA_NUM_REWARD_TERMS 5          /* reward_helper.py:75-81 */
#define CDA_MAX_GROUPS    16            /* cda_step_groups: concurrent market groups per env */

typedef enum cda_status {
    CDA_OK = 0,
    CDA_ERR_INVALID = -1,       /* bad argument / config outside the supported domain */
    CDA_ERR_NO_DEVICE = -2,     /* no HIP device (the product path has no CPU fallback) */
    CDA_ERR_HIP = -3,           /* a HIP runtime call failed; see cda_strerror */
    CDA_ERR_UNSUPPORTED = -4,   /* a tick_size outside 1 .. CDA_TICK_MAX (the config carries integer ticks only: SURVEY App. A.10); a launch this env does not qualify for */
    CDA_ERR_NOMEM = -5
} cda_status;

/* Per-market sticky flag bits reported by cda_last_flags. */
#define CDA_FLAG_BOOK_OVERFLOW   0x1u   /* a rest was dropped: the market's book (LDS tile + HBM spill ring) was full */
#define CDA_FLAG_INT_OVERFLOW    0x2u   /* a size/position/price left the int32 / 2^24 domain */
#define CDA_FLAG_DEC_DOMAIN      0x4u   /* a ledger value left the 28-digit / exponent domain */
#define CDA_FLAG_NAV_CONSERVATION 0x8u  /* episode metrics (cda_episode_metrics_enable): an episode of this market ended with sum(NAV) != num_agents * init_cash
                                           beyond the tolerance; unlike the others it survives the device-side auto reset */

/*
 * Env config: the 17 keys of continuousDoubleAuction_env.py:27-55 with the defaults of
 * config/env_defaults.json:8-27 (is_render and tape_display_length do not reach the numeric path).
 */
typedef struct cda_config {
    int32_t num_agents;          /* num_of_agents        (5)       */
    int32_t max_step;            /* max_step             (64)      */
    int32_t n_hist;              /* n_hist               (4)       */
    int32_t tick_size;           /* tick_size            (1) - an INTEGER tick, 1 .. CDA_TICK_MAX: the price ladder's step (action_helper.py:341-397) and the unit of the
                                    observation's spread (state_helper.py:202-206); off the integer grid the reference's float price arithmetic is platform dependent */
    int64_t init_cash;           /* init_cash            (1000000) - integer, > -2^62 */
    int32_t initial_price_min;   /* initial_price_min    (10)      */
    int32_t initial_price_max;   /* initial_price_max    (100)     */
    int32_t min_size;            /* min_size             (1)       */
    int32_t mkt_max_size;        /* mkt_max_size         (100)     */
    int32_t limit_size_multiple; /* limit_size_multiple  (10)      */
    int32_t auto_reset;          /* extension, 0 = off (the reference has no such key; parity runs leave it 0). 1: a market whose
                                    step ended its episode (terminated or truncated) is reset in place right after that step, with
                                    reset(seed=None) semantics (the RNG stream continues, continuousDoubleAuction_env.py:186-188);
                                    its obs row then holds the NEW episode's first observation, reward / flags / info are the
                                    finished step's */
    int32_t book_capacity;       /* extension: size of the LDS-staged book TILE, the top of a market's book, both sides together:
                                    256 or 512 resting orders (two compiled builds of the market kernels); 0 = by agent count: 256 up
                                    to 8 agents, 512 above.  What does not fit the tile lives in the HBM spill ring below */
    int32_t book_spill;          /* extension: resting orders PER SIDE a market can hold in HBM behind its tile (the reference's
                                    OrderTree is unbounded, ordertree.py:5-58).  0 = automatic: num_agents * max_step rounded up to a
                                    power of two (at least 1024) - a side cannot grow faster than one order per agent and step, so
                                    no order is ever dropped inside an episode (halved, down to CDA_SPILL_MIN, while the rings of all markets would take more than a
                                    quarter of the free device memory: compare cda_book_spill / cda_book_spill_wanted); n > 0: rounded up to a power of two in
                                    [CDA_SPILL_MIN, CDA_SPILL_MAX]; -1: no HBM tier (the round-1/2 behaviour: the tile is the whole
                                    book).  A rest that fits neither is dropped and flagged (CDA_FLAG_BOOK_OVERFLOW), never silently */
    double  order_penalty;       /* 0.1  */
    double  trade_penalty;       /* 0.05 */
    double  drawdown_penalty;    /* 0.2  */
    double  passive_bonus;       /* 0.1  */
    double  loss_multiplier;     /* 1.5  */
} cda_config;

/*
 * A ledger value: Python `Decimal` (prec 28, ROUND_HALF_EVEN) as sign / coefficient / exponent.
 * value = (-1)^sign * (w[0] + w[1]*2^32 + w[2]*2^64) * 10^exp, coefficient < 10^28.
 * `str(Decimal)` is reproduced on the host from this triple (info["NAV"], info_helper.py:54).
 */
typedef struct cda_dec {
    uint32_t w[3];
    int16_t  exp;
    uint8_t  sign;
    uint8_t  pad;
} cda_dec;

/* Optional SoA outputs mirroring Info_Helper.set_info (info_helper.py:30-116). Any pointer may be NULL. */
typedef struct cda_info_ptrs {
    cda_dec* nav;                    /* [N,A]  exact NAV (info["NAV"] = str of this)        */
    int32_t* num_trades;             /* [N,A]                                                */
    int32_t* net_position;           /* [N,A]                                                */
    double*  vwap;                   /* [N,A]  float(acc.VWAP)                               */
    double*  cash;                   /* [N,A]                                                */
    double*  cash_on_hold;           /* [N,A]                                                */
    double*  position_val;           /* [N,A]                                                */
    double*  drawdown;               /* [N,A]                                                */
    double*  max_nav;                /* [N,A]                                                */
    int32_t* num_trades_step;        /* [N,A]  read before zeroing (exchg_helper.py:116-120)  */
    int32_t* num_passive_fills_step; /* [N,A]                                                */
    int32_t* order_step_placed;      /* [N,A]                                                */
    int32_t* num_rejected_step;      /* [N,A]                                                */
    uint8_t* is_pass_action;         /* [N,A]                                                */
    double*  reward_terms;           /* [N,A,5] nav_term, order, trade, drawdown, passive    */
    double*  last_price;             /* [N]                                                  */
    double*  best_bid;               /* [N]  NaN encodes None                                */
    double*  best_ask;               /* [N]  NaN encodes None                                */
    double*  spread;                 /* [N]  NaN encodes None                                */
    int32_t* lob_actions;            /* [N,A,4] env.LOB_actions (continuousDoubleAuction_env.py:284-285): agent a's decoded order
                                        {side 0 bid / 1 ask, type 0 market / 1 limit / 2 modify / 3 cancel, size, price in ticks or
                                        -1 for a market order}; side = -1 (row all -1) when the agent passed or was absent  */
} cda_info_ptrs;

/* ---- parity dump of one market (host struct) ------------------------------------------- */
typedef struct cda_order {
    int32_t price;      /* integer ticks (the book's Decimal('<price>.0')) */
    int32_t qty;
    int32_t owner;      /* trader index = Order.trade_id (order.py:17)      */
    int32_t order_id;
    int32_t timestamp;
} cda_order;

typedef struct cda_account_state {
    cda_dec cash, cash_on_hold, position_val, vwap, nav, prev_nav, max_nav;
    int32_t net_position, num_trades;
    int32_t num_trades_step, num_passive_fills_step, order_step_placed, num_rejected_step;
} cda_account_state;

/* The dump holds the first CDA_BOOK_CAP_MAX orders of each side; n_bids / n_asks are the TRUE counts and MAY EXCEED the arrays (since round 3:
 * the book is unbounded; cda_get_book reads a side of any length).  A C caller walking the arrays must stop at
 * min(n_bids, CDA_BOOK_CAP_MAX) / min(n_asks, CDA_BOOK_CAP_MAX).  cda_set_state with a count above CDA_BOOK_CAP_MAX means "keep the book as it is"
 * (the counts must equal the market's current ones) and restores everything else of the dump. */
typedef struct cda_market_state {
    uint64_t rng_state_hi, rng_state_lo, rng_inc_hi, rng_inc_lo;  /* numpy PCG64 */
    uint32_t rng_has_uint32, rng_uinteger;
    int32_t  t_step, lob_time, next_order_id;
    int32_t  last_price;          /* env.last_price (integer valued at tick_size 1) */
    int32_t  has_trade;           /* len(LOB.tape) > 0 */
    int32_t  last_trade_price;    /* LOB.tape[-1]['price'] */
    uint32_t done_mask;           /* env.done_set as a bit mask */
    uint32_t flags;
    int32_t  n_bids, n_asks;
    cda_order bids[CDA_BOOK_CAP_MAX]; /* queue order: best price first, FIFO inside a level */
    cda_order asks[CDA_BOOK_CAP_MAX];
    cda_account_state acc[CDA_MAX_AGENTS];
    float    hist[CDA_MAX_HIST * CDA_SNAPSHOT_DIM];  /* oldest frame first */
} cda_market_state;

typedef struct cda_env cda_env;

/* Fill `cfg` with config/env_defaults.json:8-27. */
int cda_default_config(cda_config* cfg);

/* Replaces continuousDoubleAuctionEnv.__init__ (continuousDoubleAuction_env.py:27-119) for
 * n_markets independent envs on HIP device `device`. */
int cda_create(const cda_config* cfg, int32_t n_markets, int32_t device, cda_env** out);
int cda_destroy(cda_env* env);

/* Replaces reset(seed=...) (continuousDoubleAuction_env.py:175-231).
 *  seeds : u64[N] device, or NULL = seed=None (every selected market keeps its RNG stream;
 *          a market that was never seeded is seeded with its index);
 *  mask  : u8[N] device, or NULL = all markets;   obs_out : f32[N, n_hist*42] (rows of unselected
 *          markets are left untouched). */
int cda_reset(cda_env* env, const uint64_t* seeds, const uint8_t* mask, float* obs_out, void* stream);

/* Replaces step(action_dict) (continuousDoubleAuction_env.py:265-309).
 *  category i32[N,A], size_mean f32[N,A], size_sigma f32[N,A], price i32[N,A], price_offset i32[N,A]:
 *  the Dict action of action_helper.py:126-138; `present` u8[N,A] or NULL (= every agent acts)
 *  encodes an action dict holding a subset of the agents, iterated in ascending agent order.
 *  Out: obs f32[N,n_hist*42] (one shared vector per market, state_helper.py:76,109),
 *  reward f64[N,A], terminated u8[N] / truncated u8[N] (the "__all__" flags, done_helper.py:20-54),
 *  info (nullable). */
int cda_step(cda_env* env,
             const int32_t* category, const float* size_mean, const float* size_sigma,
             const int32_t* price, const int32_t* price_offset, const uint8_t* present,
             float* obs_out, double* reward_out, uint8_t* terminated_out, uint8_t* truncated_out,
             const cda_info_ptrs* info_out, void* stream);

/* The same step for the markets [first_market, first_market + n_markets) only.  Every array argument is still the
 * FULL [N, ...] array (the kernel indexes it by global market); only the launch covers fewer markets.  Markets never
 * interact, so disjoint sub-ranges of one env may be stepped CONCURRENTLY on different streams: the reference's step()
 * has no cross-env barrier either (one env object per market, continuousDoubleAuction_env.py:86,197), the batch-wide
 * barrier of cda_step is an artefact of launching all markets as one grid. */
int cda_step_range(cda_env* env, int32_t first_market, int32_t n_markets,
                   const int32_t* category, const float* size_mean, const float* size_sigma,
                   const int32_t* price, const int32_t* price_offset, const uint8_t* present,
                   float* obs_out, double* reward_out, uint8_t* terminated_out, uint8_t* truncated_out,
                   const cda_info_ptrs* info_out, void* stream);
int cda_reset_range(cda_env* env, int32_t first_market, int32_t n_markets, const uint64_t* seeds, const uint8_t* mask,
                    float* obs_out, void* stream);
/* cda_step_range with EPISODE-END CAPTURE (auto_reset envs only; fin_index_out NULL = plain cda_step_range).  The device-side auto reset overwrites the
 * finished episode's last observation in obs_out with the new episode's first one; the reference's consumers still need the old one - RLlib bootstraps a
 * time-limit truncation with V(last observation) (the env's `truncateds`, continuousDoubleAuction_env.py:303), its episode record stores it with the
 * last step (train/episode_record.py:431-470).  A market whose episode ended with this step therefore appends that observation to a compact list first:
 * slot = atomicAdd(fin_count, 1); if slot < fin_cap: fin_obs[slot][n_hist * 42] = the observation, fin_index_out[market] = slot.  fin_index_out i32[N]
 * is pre-set to -1 and fin_count i32[1] zeroed by the caller.  Cold paths only: the step's hot code is unchanged. */
int cda_step_range_capture(cda_env* env, int32_t first_market, int32_t n_markets,
                           const int32_t* category, const float* size_mean, const float* size_sigma,
                           const int32_t* price, const int32_t* price_offset, const uint8_t* present,
                           float* obs_out, double* reward_out, uint8_t* terminated_out, uint8_t* truncated_out,
                           const cda_info_ptrs* info_out, float* fin_obs, int32_t fin_cap, int32_t* fin_count, int32_t* fin_index_out, void* stream);
/* The rollout's POLICY INSIDE THE STEP KERNEL (one launch per chain and step instead of a policy launch and a step launch): for the markets of the range, the
 * policy / value network of include/cda_mlp.h (wb, theta: cda_mlp_pack's images for the env's history depth) is evaluated on obs_in f32 [N][42 n_hist] by the step kernel's own
 * workgroups (sixteen market-waves each: bf16 MFMA on their sixteen rows), every agent's action is sampled exactly as cda_mlp_policy_step samples it (same key:
 * seed, *counter_dev, draw = the step's number; bit for bit the same actions, log-probabilities, values, records), WRITTEN to the five action arrays / a_cont /
 * logp / value / rec (may be NULL) / dist (may be NULL), and the markets are stepped with them exactly as cda_step_range_capture steps them (same outputs, same
 * auto reset and episode-end capture).  What the reference does between two env.step calls - RLlib's policy forward + action sampling, train/train.py:453-541 -
 * moved into the step itself.  cda_policy_step_supported: 1 when this env qualifies (256-order book tile, n_hist 1 / 2 / 4 / 8 - the depths the network is
 * compiled for -, <= 8 agents, no hand-back records);
 * otherwise cda_policy_step_range returns CDA_ERR_UNSUPPORTED and the caller launches the two kernels (cda_mlp_rollout_chain does). */
int cda_policy_step_supported(const cda_env* env);
int cda_policy_step_advised(const cda_env* env);      /* supported AND the env's markets fit the device in one round of sixteen-market workgroups (N <= 16 x CUs): where the one launch wins */
int cda_policy_step_range(cda_env* env, int32_t first_market, int32_t n_markets, const void* wb, const float* theta, const float* obs_in,
                          uint64_t seed, const int64_t* counter_dev, int64_t draw,
                          int32_t* category, float* size_mean, float* size_sigma, int32_t* price, int32_t* price_offset,
                          float* a_cont, float* logp, float* value, float* rec, float* dist,
                          float* obs_out, double* reward_out, uint8_t* terminated_out, uint8_t* truncated_out,
                          float* fin_obs, int32_t fin_cap, int32_t* fin_count, int32_t* fin_index_out, void* stream);
/* cda_step as n_groups (<= CDA_MAX_GROUPS) launches, group g = the markets cda_group_range() names, on streams[g].
 * One host call; each group is an independent chain of launches on its own stream, so one group's slowest market
 * overlaps the other groups' work instead of stalling the whole batch.  The outputs of group g are complete when
 * streams[g] reaches this point; ordering against other streams is the caller's business (events). */
int cda_step_groups(cda_env* env, int32_t n_groups,
                    const int32_t* category, const float* size_mean, const float* size_sigma,
                    const int32_t* price, const int32_t* price_offset, const uint8_t* present,
                    float* obs_out, double* reward_out, uint8_t* terminated_out, uint8_t* truncated_out,
                    const cda_info_ptrs* info_out, void* const* streams);
/* markets of group `group` of `n_groups`: [first, first + count) with first = N*group/n_groups (integer division) */
void cda_group_range(int32_t n_markets, int32_t n_groups, int32_t group, int32_t* first_out, int32_t* count_out);

/* The hand-back to a central learner (north_star: "RCCL all-gather over xGMI only for the observation/reward tensors handed
 * back to the learner"; the reference hands step()'s return value to RLlib in-process, train/train.py:509-518).  Of a market's
 * n_hist x 42 observation only the newest frame is new each step, so what has to travel is ONE compact record per market:
 *
 *     f32 frame[42]   the newest observation frame (state_helper.py:94-111 appends it to the history)
 *     f64 reward[A]
 *     u8  terminated, truncated      the "__all__" flags
 *     u8  restarted                   1: the observation restarts from `frame` (reset / auto-reset: all n_hist frames equal it)
 *     padding to cda_handback_stride(A) bytes (a multiple of 8)
 *
 * cda_set_handback(env, records): from now on every step (and reset) ALSO writes records[market] (device, [N, stride] bytes;
 * NULL switches it off).  The records of a contiguous market range are contiguous, so a group chain hands its own range to
 * the collective where the kernel left it.  cda_handback_unpack is the receiving side, for ANY process (it needs no env):
 * n_segments x seg_records records (an all-gathered buffer: one segment per rank), record k of segment s belonging to row
 * row0 + s * seg_row_stride + k of the learner's full arrays - the observation row is shifted by one frame and the new frame
 * appended (or, restarted, all frames set), reward / flags rows are overwritten.  One launch.  (n_rows_total: see cda_set_handback_geometry.) */
int32_t cda_handback_stride(int32_t num_agents);
int cda_set_handback(cda_env* env, void* records_dev);
int cda_handback_unpack(const void* records_dev, int32_t n_segments, int32_t seg_records, int64_t seg_row_stride, int64_t row0,
                        int32_t num_agents, int32_t n_hist, int64_t n_rows_total,
                        float* obs_full, double* reward_full, uint8_t* terminated_full, uint8_t* truncated_full, void* stream);
/* Uneven shards (n_rows_total not divisible by the rank count): every rank steps and sends the SAME number of records - the largest shard,
 * the smaller ones padded with markets nobody reads - and the receiving side skips the padding: with n_rows_total > 0, segment s owns
 * seg_row_stride rows (the last one n_rows_total - s * seg_row_stride) and record k of it is used only while row0 + k is below that.
 * n_rows_total = 0: every record is used (equal shards).  cda_set_handback_geometry gives the native chains (cda_step_groups_handback,
 * cda_handback_groups) the same two numbers: rows between two ranks' first markets, global row count (0, 0 = equal shards of N). */
int cda_set_handback_geometry(cda_env* env, int64_t shard_row_stride, int64_t n_rows_total);

/* cda_step_groups AND the hand-back of every chain in ONE host call (a Python loop over chains - stream context, collective, unpack -
 * costs ~25 us of host time per chain and step, more than the step itself): on streams[g], for the markets of group g,
 *     k_step  ->  ncclAllGather(this rank's records of group g -> gathered[g], world x count x stride bytes)  ->  cda_handback_unpack
 * into the learner-side arrays (row of global market = rank * shard_rows + local market).  comms[g] is an RCCL communicator
 * (ncclComm_t) the caller created for chain g - one per chain, so that the chains' collectives may be in flight together; the library
 * resolves ncclAllGather from the RCCL already loaded in the process (no link-time dependency).  world == 1: comms may be NULL, the
 * records are unpacked where the kernel left them.  Needs cda_set_handback. */
int cda_step_groups_handback(cda_env* env, int32_t n_groups,
                             const int32_t* category, const float* size_mean, const float* size_sigma,
                             const int32_t* price, const int32_t* price_offset, const uint8_t* present,
                             float* obs_out, double* reward_out, uint8_t* terminated_out, uint8_t* truncated_out,
                             const cda_info_ptrs* info_out, void* const* streams,
                             void* const* comms, int32_t world, void* const* gathered,
                             float* obs_full, double* reward_full, uint8_t* terminated_full, uint8_t* truncated_full);
/* the hand-back alone (after a reset: its records carry `restarted`), same arguments */
int cda_handback_groups(cda_env* env, int32_t n_groups, void* const* streams, void* const* comms, int32_t world, void* const* gathered,
                        float* obs_full, double* reward_full, uint8_t* terminated_full, uint8_t* truncated_full);

/* Replaces CDA_rand.run_random (CDA_rand.py:40-85): every market plays uniform random agents (the law and the
 * counter-based sampler of include/cda_random_agents.h, keyed by action_seed, market_index_base + market, the
 * market's own step counter and the agent) for up to n_steps steps, stopping early at its episode's end
 * (terminated or truncated), ALL INSIDE ONE LAUNCH: a market-wave keeps its state in LDS from step to step and
 * never waits for the slower markets of the batch.  Bit-identical to n_steps calls of cda_step on the same actions.
 *  Out (each nullable): obs f32[N,n_hist*42] after the last step taken, episode_return f64[N,A] = the sum, in step
 *  order, of each agent's rewards over the steps taken, terminated / truncated u8[N] of the last step taken (0 if
 *  none), steps_taken i32[N]. */
int cda_run_random(cda_env* env, int32_t n_steps, uint64_t action_seed, uint64_t market_index_base,
                   float* obs_out, double* episode_return_out, uint8_t* terminated_out, uint8_t* truncated_out,
                   int32_t* steps_taken_out, void* stream);
/* The same sampler on the host (host pointers, [n_markets, num_agents] each): the actions cda_run_random plays at
 * step `step`, for callers and tests that want to replay them through cda_step. */
int cda_random_actions_host(uint64_t action_seed, uint64_t market_index_base, int32_t step, int32_t n_markets,
                            int32_t num_agents, int32_t* category, float* size_mean, float* size_sigma,
                            int32_t* price, int32_t* price_offset);

/* The same sampler on the device: the actions of steps [step0, step0 + n_steps), five [n_steps, n_markets, num_agents]
 * device arrays - the resident synthetic input of bench.py (generator keyed (seed, step, market, agent)). */
int cda_random_actions(uint64_t action_seed, uint64_t market_index_base, int32_t step0, int32_t n_steps, int32_t n_markets,
                       int32_t num_agents, int32_t* category, float* size_mean, float* size_sigma, int32_t* price,
                       int32_t* price_offset, void* stream);

/* The reference's end-of-episode invariant (train/callbk/league_based_self_play_callback.py:679-704) for every market, on
 * the device: total = Decimal(0); total += NAV_a for a = 0..A-1 (the accounts' current NAV, in that order, prec-28
 * arithmetic); error = total - Decimal(init_cash) * A.  abs_error_out f64[N] (device) receives float(abs(error)),
 * violated_out u8[N] (device, nullable) whether it exceeds `tolerance` (the reference's nav_tolerance, 1e-6). */
int cda_nav_conservation(cda_env* env, double tolerance, double* abs_error_out, uint8_t* violated_out, void* stream);

/* ---- every episode checked and summarised on the device --------------------------------------------------------------------
 * What the reference's callback does around an episode (train/callbk/league_based_self_play_callback.py):
 *   on_episode_step (:541-600)   tallies per episode: passes, rejections, per-agent trades / passive fills, sum and sum of squares of the five reward terms;
 *   on_episode_end  (:627-755)   sum of the agents' NAV == num_agents x init_cash in Decimal (nav_tolerance 1e-6), `nav_conservation_violations` += 1 otherwise;
 *                                per-agent NAV / drawdown / |net_position| / num_trades of the last step (_log_episode_account :418-470); pass and rejection
 *                                fractions, reward-term means and variance shares, the most maker-like agent's passive share (_log_activity :295-416);
 *   train.py:1109-1164           the driver stops the run on a violation (strict_nav_check).
 * With auto_reset the finished episode's ledger is wiped by the in-kernel reset before any host code could look at it, so all of it happens on the
 * device.  cda_episode_metrics_enable(env, 1, tolerance): from now on
 *   - every step adds agent a's step counters and reward terms to the running tallies of (market, a) in the market record (fire-and-forget atomics:
 *     nothing is loaded and no register is held across the step);
 *   - a market whose episode ended (all agents done, or max_step reached) is CHECKED AND CREDITED exactly once, in the cold paths that handle an episode
 *     end - the in-kernel auto reset of cda_step / cda_policy_step_range, the auto-reset pass behind a step with info tensors, cda_reset of a market
 *     whose episode is over, the end of cda_run_random: the exact decimal sum of NAV (the arithmetic of cda_nav_conservation) sets the sticky
 *     CDA_FLAG_NAV_CONSERVATION bit (it survives the auto reset) and counts a violation; the tallies and the last step's account figures are added
 *     to per-(market, agent) and per-market accumulators;  a reset of a market whose episode is NOT over discards its running tallies (an episode
 *     the reference's callback never sees end either).
 * cda_episode_metrics_collect reduces the accumulators over the markets, by module: module_of i32[N,A] (device; NULL = every slot is module 0) names
 * the module that played slot a of market i during the episodes collected (league self-play: the draw of cda_league_assign; values outside
 * [0, n_modules) are skipped) -> agent_out f64[n_modules, CDA_EM_AGENT_FIELDS], env_out f64[CDA_EM_ENV_FIELDS] (device), in a FIXED order (two launches;
 * the same inputs give the same bits).  clear != 0 zeroes the accumulators behind the read.  Call it before slot assignments change.
 * Sums of integers are exact in f64 (< 2^53). */
#define CDA_EM_AGENT_FIELDS 32
enum {                                   /* per module: sums over the (episode, agent) pairs the module played */
    CDA_EM_EPISODES = 0,                 /* (episode, agent) pairs credited */
    CDA_EM_AGENT_STEPS = 1,              /* tally["agent_steps"] */
    CDA_EM_PASSES = 2,                   /* is_pass_action */
    CDA_EM_REJECTIONS = 3,               /* num_rejected_step */
    CDA_EM_PLACED = 4,                   /* order_step_placed */
    CDA_EM_TRADES = 5,                   /* num_trades_step */
    CDA_EM_PASSIVE = 6,                  /* num_passive_fills_step */
    CDA_EM_TERM_SUM = 7,                 /* 7..11: sum of reward term j over the agent-steps */
    CDA_EM_TERM_SQ = 12,                 /* 12..16: sum of its square */
    CDA_EM_RETURN_SUM = 17,              /* sum of episode returns (f64 sum of the step rewards, in step order) */
    CDA_EM_RETURN_SQ = 18,
    CDA_EM_NAV_SUM = 19,                 /* float(NAV) at the episode's last step */
    CDA_EM_NAV_MIN = 20,
    CDA_EM_NAV_MAX = 21,
    CDA_EM_DRAWDOWN_SUM = 22,            /* float(max(0, max_nav - nav)) */
    CDA_EM_ABS_POSITION_SUM = 23,
    CDA_EM_NUM_TRADES_SUM = 24,          /* acc.num_trades (the episode total) */
    CDA_EM_MAKER_RATIO_SUM = 25,         /* passive / trades of the pairs with trades >= 5 (_MIN_TRADES_FOR_MAKER_RATIO) */
    CDA_EM_MAKER_RATIO_N = 26,
    CDA_EM_MAKER_RATIO_MAX = 27,
    CDA_EM_BANKRUPT = 28                 /* pairs that ended in done_set (nav <= 0) */
};
#define CDA_EM_ENV_FIELDS 8
enum {                                   /* per env: sums over the episodes that ended */
    CDA_EM_ENV_EPISODES = 0,
    CDA_EM_ENV_NAV_VIOLATIONS = 1,       /* nav_conservation_violations */
    CDA_EM_ENV_NAV_ERROR_SUM = 2,        /* float(abs(sum(NAV) - A * init_cash)) */
    CDA_EM_ENV_NAV_ERROR_MAX = 3,
    CDA_EM_ENV_MAKER_MAX_SUM = 4,        /* maker_fill_ratio_max: the episode's max over its qualifying agents */
    CDA_EM_ENV_MAKER_MAX_N = 5,          /* episodes with a qualifying agent */
    CDA_EM_ENV_STEPS = 6,                /* env steps of the episodes */
    CDA_EM_ENV_TERMINATED = 7            /* episodes that ended with every agent done (the others were truncated at max_step) */
};
int cda_episode_metrics_enable(cda_env* env, int32_t on, double nav_tolerance);
int cda_episode_metrics_collect(cda_env* env, const int32_t* module_of, int32_t n_modules, double* agent_out, double* env_out, int32_t clear, void* stream);
#define CDA_EM_MAX_MODULES 32

/* Test/diagnostic hook: Trader.place_order (agent/trader.py:49-106) for ONE decoded order on one
 * market, bypassing decode and the RNG. type: 0 market, 1 limit, 2 modify, 3 cancel; side: 0 bid,
 * 1 ask; price in ticks (ignored for market). Synchronous. */
int cda_place_order(cda_env* env, int32_t market, int32_t trader, int32_t type, int32_t side,
                    int32_t size, int32_t price);
/* Test/diagnostic hook: Exchg_Helper.mark_to_mkt (exchg_helper.py:56-66) on one market. Synchronous. */
int cda_mark_to_mkt(cda_env* env, int32_t market);

/* Parity dump / restore of one market (synchronous; host struct). */
int cda_get_state(cda_env* env, int32_t market, cda_market_state* out_host);
int cda_set_state(cda_env* env, int32_t market, const cda_market_state* in_host);

/* One side of one market's book, whole, in queue order (best price first, FIFO inside a level; ordertree.py / orderlist.py):
 * up to max_orders orders -> orders_out_host (host), the side's true length -> n_out_host.  side: 0 bids, 1 asks.  Synchronous. */
int cda_get_book(cda_env* env, int32_t market, int32_t side, cda_order* orders_out_host, int32_t max_orders, int32_t* n_out_host);
/* Orders per side the HBM spill ring of this env holds (0 = no HBM tier) - what cda_create GRANTED - and what was asked for (automatic:
 * num_agents * max_step rounded up to a power of two, the size with which no order is ever dropped inside an episode).  The automatic size is
 * halved while the rings would take more than a quarter of the free device memory; when granted < wanted the book is bounded by tile + ring and
 * a rest beyond it is dropped and flagged (CDA_FLAG_BOOK_OVERFLOW) - a caller can tell at construction. */
int32_t cda_book_spill(const cda_env* env);
int32_t cda_book_spill_wanted(const cda_env* env);

/* Pre-step raw top-10 snapshot agg_LOB_raw f32[N,40] (state_helper.py:159-160) -> device buffer. */
int cda_get_raw_snapshot(cda_env* env, float* raw_out, void* stream);

/* Per-market sticky flags u32[N] -> device buffer. */
int cda_last_flags(cda_env* env, uint32_t* flags_out, void* stream);
/* Book census i32[N] -> device buffer: the most resting orders (both sides together) each market has held since its
 * last reset.  The reference's OrderTree is unbounded (ordertree.py:5-58); this build holds cda_book_capacity(env) in the
 * LDS tile and cda_book_spill(env) per side behind it in HBM. */
int cda_book_peak(cda_env* env, int32_t* peak_out, void* stream);

/* Structural invariants of every market -> u32[N] device buffer of CDA_INV_* bits (0 = all hold): sides sorted best
 * price first (ordertree.py:44-58), book not crossed (orderbook.py:162-194), quantities positive,
 * cash_on_hold == value of the trader's own resting orders, exactly (cash_processor.py:15-29, :85-97), positions net
 * to zero (account.py:196-213).  A size-independent property check for runs too large to replay on the CPU. */
#define CDA_INV_BIDS_SORTED   0x01u
#define CDA_INV_ASKS_SORTED   0x02u
#define CDA_INV_CROSSED       0x04u
#define CDA_INV_QTY           0x08u
#define CDA_INV_ESCROW        0x10u
#define CDA_INV_NET_POSITION  0x20u
#define CDA_INV_OWNER         0x80u
#define CDA_INV_BOOK_COUNT    0x100u
int cda_check_invariants(cda_env* env, uint32_t* violations_out, void* stream);

/* Device self-tests of the ledger arithmetic and the RNG (host pointers; synchronous).
 * op: 0 add, 1 sub, 2 mul, 3 div, 4 cmp (result in out[i].w[0]: 0 lt, 1 eq, 2 gt), 5 to-double
 * (bits in out[i].w[0..1]). For mul/div `b` must be integer valued with coefficient < 2^32.
 * 6 = a + b through the cash / cash_on_hold transfer leaf with the general addition as its fallback
 * (out[i].pad = 1 when the leaf itself produced the result). */
int cda_selftest_dec(int32_t device, int32_t op, int32_t n, const cda_dec* a_host, const cda_dec* b_host,
                     cda_dec* out_host);
/* Draw, on the device, the env's RNG schedule for one seed: one integers(lo,hi+1), then `n_steps`
 * times {n_normals standard normals, one permutation(perm_n)}. Outputs host arrays. */
int cda_selftest_rng(int32_t device, uint64_t seed, int32_t lo, int32_t hi, int32_t n_steps,
                     int32_t n_normals, int32_t perm_n, int32_t* first_int_host,
                     double* normals_host /*[n_steps*n_normals]*/, int32_t* perms_host /*[n_steps*perm_n]*/,
                     uint64_t* final_state_host /*[6]: state_hi, state_lo, inc_hi, inc_lo, has_uint32, uinteger*/);

/* The restated libm functions numpy's generator calls on the host (csrc/cda_libm.hpp; op 0: log1p, glibc 2.35
 * s_log1p.c; op 1: exp, glibc 2.35 e_exp.c as built with FMA; device only, op 2: the f64 square root the observation
 * uses, which must be the correctly rounded IEEE one) evaluated on the device / by the same source compiled
 * for the host (no GPU needed).  Host pointers; synchronous. */
int cda_selftest_libm(int32_t device, int32_t op, int32_t n, const double* x_host, double* y_host);
int cda_selftest_libm_host(int32_t op, int32_t n, const double* x_host, double* y_host);

/* Learner side of the PPO loop on the batched env (the reference trains through RLlib's PPO, train/train.py:453-541,
 * config/train_config.json `ppo`): clipped-surrogate + value + entropy loss of one minibatch and its gradient with respect to the
 * network outputs in ONE launch (csrc/cda_ppo.hip).  A ROW of network outputs serves `agents_per_row` consecutive samples: every agent
 * of a market is handed the same observation (state_helper.py:76,109), so a shared policy's outputs are the same for all of them and
 * the network runs once per market-step; agents_per_row = 1 is the plain one-row-per-sample op.  With B = rows * agents_per_row:
 * logits f32[rows,24] = category 9 | price 10 | price_offset 3 | two Gaussian means, value f32[rows], log_std f32[2], actions
 * a_cat / a_price / a_off i64[B] and a_cont f32[B,2], logp_old / adv / ret f32[B] (sample r * agents_per_row + a belongs to row r).
 * Out: d_logits f32[rows,24], d_value f32[rows] (gradients of the loss, summed over the row's samples), sums5 f64[5] (scratch), out6
 * f32[6] = mean policy loss, mean value loss, mean entropy, loss, d loss / d log_std[0..1] (means over the B samples).  Device pointers.
 * row_index i64[rows] (optional, NULL = identity): the rows are a shuffled minibatch and the samples of row r live at
 * row_index[r] * agents_per_row + a in the per-sample arrays (which then hold the WHOLE batch, unshuffled) - an epoch's shuffle moves
 * the observations only.
 * out_stride != 0 (a multiple of 4 above 24): `logits` is the network's padded output matrix f32[rows, out_stride] as it stands - logits
 * in columns 0..23, the value in column 24 - and `d_logits` its gradient in the same shape (d value in column 24, zeros behind);
 * `value` / `d_value` are not used (may be NULL).  out_stride == 0: the separate arrays described above. */
int cda_ppo_loss(const float* logits, const float* value, const float* log_std, const int64_t* a_cat, const int64_t* a_price,
                 const int64_t* a_off, const float* a_cont, const float* logp_old, const float* adv, const float* ret,
                 const int64_t* row_index, int64_t rows, int32_t agents_per_row, int32_t out_stride, float clip, float vf_coef, float ent_coef,
                 float* d_logits, float* d_value, double* sums5, float* out6, void* stream);

/* The rollout's policy step (csrc/cda_ppo.hip): sample the Dict action of B = rows * agents_per_row (market, agent) pairs from the
 * network outputs (logits f32[rows, logits_stride]: 24 for the bare logits, or the padded output matrix of cda_ppo_loss's out_stride,
 * whose column 24 is then copied to value_out f32[rows] if that is not NULL; row sharing as in cda_ppo_loss) in ONE launch: a_cat / a_price / a_off i64[B],
 * a_cont f32[B,2] (the raw Gaussian samples), logp f32[B], and the env's five action arrays i32 / f32 [B] (size_mean = tanh,
 * size_sigma = sigmoid of the Gaussian samples: the Box bounds of action_helper.py:126-138).  Randomness is counter based, keyed
 * (seed, *counter_dev, sample); counter_dev (i64[1], device) is incremented on the stream after every call, so a captured HIP graph
 * draws fresh numbers on every replay. */
int cda_policy_sample(const float* logits, int32_t logits_stride, float* value_out, const float* log_std, int64_t rows, int32_t agents_per_row,
                      uint64_t seed, int64_t* counter_dev,
                      int64_t* a_cat, int64_t* a_price, int64_t* a_off, float* a_cont, float* logp,
                      int32_t* env_category, float* env_size_mean, float* env_size_sigma, int32_t* env_price, int32_t* env_price_offset, void* stream);

/* Generalised advantage estimation over a rollout (ppo.gae's recursion, one launch): rew / val / done f32[n_steps, batch] (done = 1 where
 * the episode ended with that step), last_val f32[batch] -> adv, ret f32[n_steps, batch].  Device pointers. */
int cda_gae(const float* rew, const float* val, const float* last_val, const float* done, int32_t n_steps, int64_t batch,
            float gamma, float lam, float* adv, float* ret, void* stream);

/* Rollout buffers in one launch: for k < n_items copy bytes[k] bytes from src[k] to dst_base[k] + *slot_dev * bytes[k] (slot t of a
 * [T, bytes[k]] buffer; the caller keeps *slot_dev below T), then, if bump, *slot_dev += 1 on the stream.  The slot index is read on
 * the device, so a captured HIP graph replays the call unchanged for every step of a rollout.  src / dst_base / bytes are HOST arrays
 * of device pointers / sizes (read at call time); n_items <= CDA_SLOT_ITEMS_MAX. */
#define CDA_SLOT_ITEMS_MAX 12
int cda_store_slots(int32_t n_items, const void* const* src, void* const* dst_base, const int64_t* bytes, int64_t* slot_dev, int32_t bump, void* stream);

const char* cda_strerror(int status);
int32_t cda_num_markets(const cda_env* env);
int32_t cda_num_agents(const cda_env* env);
int32_t cda_book_capacity(const cda_env* env);   /* 256 or 512: the LDS tile this env was built with */
int32_t cda_obs_dim(const cda_env* env);
/* Bytes the arena keeps per market in HBM. */
int64_t cda_state_bytes_per_market(const cda_env* env);

#ifdef __cplusplus
}
#endif
#endif /* CDA_H */
