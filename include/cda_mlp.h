/*
 * cda_mlp.h - C-ABI of the policy / value network on the consumer side of the env path (BASELINE configs[4]).
 *
 * The reference trains through RLlib's PPO (gym_continuousDoubleAuction/train/train.py:453-541) with the network of
 * config/train_config.json:45-53: separate policy and value MLPs, 256 x 256, tanh.  Every agent of a market is handed the same
 * 168-float observation (envs/exchg/state_helper.py:76,109), so the network runs once per market-step.  This header is the whole
 * network as hand-written bf16 MFMA kernels for gfx950 (csrc/cda_mlp.hip) - the one dense contraction next to the env path:
 *
 *     obs f32[168] -> [policy | value] first layers 168 -> 2 x 256, tanh -> two independent 256 x 256 layers, tanh
 *                  -> heads: 24 policy outputs (category 9 | price 10 | price_offset 3 | two Gaussian means) from the policy half,
 *                     1 value from the value half: a padded row of 32 floats (columns 0..23 | 24 | zeros)
 *
 * Arithmetic: bf16 operands, f32 accumulation (v_mfma_f32_32x32x16_bf16), f32 biases, tanh in f32, activations rounded to bf16
 * between layers.  The SAME forward code serves the rollout (cda_mlp_policy_step) and the update (cda_mlp_forward_train), so the
 * log-probabilities a rollout records are the ones the first minibatch step of its update recomputes.
 *
 * Parameters: ONE f32 vector `theta` of CDA_MLP_PARAMS floats (the optimiser's master copy)
 *     W1 [512][168] | b1 [512] | W2 [2][256][256] | b2 [512] | Wo [32][256] | bo [32] | log_std [2]
 * rows 0..255 of W1 / b1 / b2 and block 0 of W2 are the policy network, the rest the value network; Wo rows 0..23 are the policy
 * heads (reading the policy half), row 24 the value head (reading the value half), rows 25..31 unused (zero, never updated).
 * cda_mlp_pack derives the bf16 operand copies `wb` (CDA_MLP_WB_ELEMS bf16) the kernels multiply with - five matrices of the sizes
 *     W1 [512][176] | W2 [2][256][256] | Wo [2][32][256] | W2^T [2][256][256] | Wo^T [2][256][32]
 * each stored in MFMA OPERAND ORDER: per group of 64 (heads: 32) output rows, per column tile and k-step, the 64 lanes' 16-byte pieces are 1 KB of
 * contiguous memory (csrc/cda_mlp.hip WRing / pack_one), so that a wave's weight request is eight whole cache lines.
 *
 * Tensors of the update live in HBM in the MFMA's own register layout ("packed": [rows/32][feature tiles of 32][2][64 lanes][8 bf16],
 * lane = (feature, row-half), the 8 slots = 8 rows) so that the weight-gradient kernel reads its operands with no transposition.
 *
 * Conventions as in cda.h: device pointers, caller-owned, plain sizes; CDA_OK or a negative cda_status; kernels are enqueued on
 * `stream`.  No CPU fallback.
 */
#ifndef CDA_MLP_H
#define CDA_MLP_H

#include <stdint.h>
#include "cda.h"

#ifdef __cplusplus
extern "C" {
#endif

/* History depth the entry points are compiled for.  The unsuffixed names are the reference's n_hist = 4 (config/train_config.json:17: 168-float observations);
 * the library also holds the SAME entry points for the depths listed in CDA_MLP_HIST_VARIANTS under the names <name>_h<H> (cda_mlp_policy_step_h8, ...: same
 * signatures, the constants below evaluated at that depth - one object file per depth, csrc/cda_mlp_variant.h).  Python: mlp.layout(n_hist). */
#ifndef CDA_MLP_HIST
#define CDA_MLP_HIST 4
#endif
#define CDA_MLP_HIST_VARIANTS "1 2 3 6 7 8"
#define CDA_MLP_OBS       (42 * CDA_MLP_HIST)                       /* 168 */
#define CDA_MLP_KX        ((CDA_MLP_OBS + 15) / 16 * 16)             /* 176: the observation padded to MFMA k-steps of 16 */
#define CDA_MLP_XTILES    ((CDA_MLP_KX + 31) / 32)                   /* 6: ... and to feature tiles of 32 in the packed layout */
#define CDA_MLP_HID       256
#define CDA_MLP_FEAT      512
#define CDA_MLP_NOUT      32
#define CDA_MLP_OFF_W1    0
#define CDA_MLP_OFF_B1    (CDA_MLP_FEAT * CDA_MLP_OBS)              /* 86016 */
#define CDA_MLP_OFF_W2    (CDA_MLP_OFF_B1 + CDA_MLP_FEAT)            /* 86528 */
#define CDA_MLP_OFF_B2    (CDA_MLP_OFF_W2 + 2 * CDA_MLP_HID * CDA_MLP_HID)   /* 217600 */
#define CDA_MLP_OFF_WO    (CDA_MLP_OFF_B2 + CDA_MLP_FEAT)            /* 218112 */
#define CDA_MLP_OFF_BO    (CDA_MLP_OFF_WO + CDA_MLP_NOUT * CDA_MLP_HID)      /* 226304 */
#define CDA_MLP_OFF_LS    (CDA_MLP_OFF_BO + CDA_MLP_NOUT)            /* 226336 */
#define CDA_MLP_PARAMS    (CDA_MLP_OFF_LS + 2)                       /* 226338 */
#define CDA_MLP_WB_W1     0                                          /* offsets (bf16 elements) inside the operand blob */
#define CDA_MLP_WB_W2     (CDA_MLP_FEAT * CDA_MLP_KX)                /* 90112 */
#define CDA_MLP_WB_WO     (CDA_MLP_WB_W2 + 2 * CDA_MLP_HID * CDA_MLP_HID)    /* 221184 */
#define CDA_MLP_WB_W2T    (CDA_MLP_WB_WO + 2 * CDA_MLP_NOUT * CDA_MLP_HID)   /* 237568 */
#define CDA_MLP_WB_WOT    (CDA_MLP_WB_W2T + 2 * CDA_MLP_HID * CDA_MLP_HID)   /* 368640 */
#define CDA_MLP_WB_ELEMS  (CDA_MLP_WB_WOT + 2 * CDA_MLP_HID * CDA_MLP_NOUT)   /* 385024 */
/* dense gradient slab of one row chunk: dW1 [512][32 XTILES] | dW2 [2][256][256] | dWo [32][512]  (f32) */
#define CDA_MLP_SLAB_W1   0
#define CDA_MLP_SLAB_W2   (CDA_MLP_FEAT * 32 * CDA_MLP_XTILES)       /* 98304 */
#define CDA_MLP_SLAB_WO   (CDA_MLP_SLAB_W2 + 2 * CDA_MLP_HID * CDA_MLP_HID)  /* 229376 */
#define CDA_MLP_SLAB      (CDA_MLP_SLAB_WO + CDA_MLP_NOUT * CDA_MLP_FEAT)    /* 245760 */
/* bias partial sums of one row tile of the backward kernel: db1 [512] | db2 [512] | dbo [32]  (f32) */
#define CDA_MLP_BSLAB     1056
#define CDA_MLP_SCRATCH   512                    /* f64 words of cda_mlp_adam's scratch */

/* rows per workgroup of the update's forward / backward kernels (32, 64 or 128: CDA_MLP_MT in the environment, default 128) = rows per
 * bias partial of cda_mlp_backward */
int32_t cda_mlp_tile_rows(void);
/* output panels ("jobs") of cda_mlp_wgrad: 2 (dW2) + 2 x groups of the x panel (dW1) + 1 (dWo) = 5 at n_hist 4; a caller sizes n_chunks so that jobs x chunks fills the 256 CUs */
int32_t cda_mlp_wgrad_jobs(void);

/* theta -> wb (one launch).  Called after every optimiser step (cda_mlp_adam does it itself). */
int cda_mlp_pack(const float* theta, void* wb, void* stream);

/* Rollout: network forward on the observations of the markets [first_market, first_market + n_markets) and, from its outputs, the
 * Dict action of every (market, agent) pair (action_helper.py:126-138), in ONE launch:
 *   obs f32[N,168] (full array, indexed by global market) ->
 *   env_* [N,A]: category i32, size_mean f32 (tanh of the Gaussian sample), size_sigma f32 (sigmoid), price i32, price_offset i32 -
 *   what cda_step consumes; a_cont f32[N,A,2] the raw Gaussian samples; logp f32[N,A]; value f32[N].
 * Randomness: counter based (include/cda_random_agents.h's splitmix64), keyed (seed, *counter_dev, draw, global sample index); nothing
 * is bumped - the caller varies `draw` (the step index inside a rollout) and *counter_dev (once per rollout). */
int cda_mlp_policy_step(const void* wb, const float* theta, const float* obs, int32_t first_market, int32_t n_markets, int32_t num_agents,
                        uint64_t seed, const int64_t* counter_dev, int64_t draw,
                        int32_t* env_category, float* env_size_mean, float* env_size_sigma, int32_t* env_price, int32_t* env_price_offset,
                        float* a_cont, float* logp, float* value, void* stream);
/* The network outputs alone for rows [first_row, first_row + n_rows) of obs f32[*,168] -> out f32[*,32] (same row indexing). */
int cda_mlp_forward(const void* wb, const float* theta, const float* obs, int64_t first_row, int64_t n_rows, float* out, void* stream);

/* An epoch's shuffle: perm i64[n] = a pseudo-random PERMUTATION of 0 .. n-1 determined by `key` (a keyed bijective mixer + cycle walking, one
 * launch; the sort behind torch.randperm is ~10 launches). */
int cda_mlp_permutation(uint64_t key, int64_t n, int64_t* perm, void* stream);

/* Update, step 0: rows of obs f32[*,168] selected by perm i64[n_rows] (NULL = identity) -> x_rm bf16[n_rows][176] (row-major, zero
 * padded) and x_pk (packed, 6 feature tiles).  n_rows % 32 == 0. */
int cda_mlp_prep_rows(const float* obs, const int64_t* perm, int64_t n_rows, void* x_rm, void* x_pk, void* stream);
/* step 1: forward on n_rows (% 32 == 0) prepared rows: h1p / h2p packed bf16 [n_rows/32][16][2][64][8], out f32[n_rows][32] (all three padded
 * to whole workgroup tiles, see cda_mlp_backward). */
int cda_mlp_forward_train(const void* wb, const float* theta, const void* x_rm, int64_t n_rows, void* h1p, void* h2p, float* out, void* stream);
/* step 2 (after the loss: cda_ppo_loss with out_stride 32 gives d_out f32[n_rows][32]): back-propagation through the heads and both
 * hidden layers: dz1p / dz2p packed bf16 (gradients at the pre-activations), doutp packed bf16 [n_rows/32][1][2][64][8], and the bias
 * partial sums bias_slab f32[ceil(n_rows / cda_mlp_tile_rows())][CDA_MLP_BSLAB].
 * h1p / h2p / dz1p / dz2p / doutp / out must hold ceil(n_rows / cda_mlp_tile_rows()) * cda_mlp_tile_rows() rows: the kernels write whole workgroup
 * tiles (rows past n_rows are scratch); d_out holds n_rows rows. */
int cda_mlp_backward(const void* wb, const float* d_out, const void* h1p, const void* h2p, int64_t n_rows,
                     void* dz1p, void* dz2p, void* doutp, float* bias_slab, void* stream);
/* step 3: weight gradients as n_chunks partial sums over row chunks: slab f32[n_chunks][CDA_MLP_SLAB]. */
int cda_mlp_wgrad(const void* x_pk, const void* h1p, const void* h2p, const void* dz1p, const void* dz2p, const void* doutp,
                  int64_t n_rows, int32_t n_chunks, float* slab, void* stream);
/* step 4: reduce the partials to the gradient of theta, clip its global norm to max_norm (torch.nn.utils.clip_grad_norm_), one Adam step
 * (torch.optim.Adam: no weight decay, bias-corrected; *step_dev f32[1] is incremented on the device), and refresh wb: two launches.
 * loss_sums5 (f64[CDA_MLP_LOSS_SLOTS][8], may be NULL): the sums the loss accumulated for this minibatch (clear = finish = 0) - cda_ppo_loss32 /
 * cda_ppo_loss_records add into words 0..4 of slot 0, cda_mlp_forward_backward into every slot; words [3..4] are d loss / d log_std;
 * the means go to loss_out6 (f32[6], may be NULL: what finish = 1 would have written, loss_samples = rows * agents_per_row) and the sums are
 * CLEARED for the next minibatch - no memset, no finishing launch between the steps of an update.
 * grad f32[CDA_MLP_PARAMS] receives the gradient before clipping (its never-written entries - the heads' rows 25..31 - must be zero: allocate
 * it zeroed); scratch f64[CDA_MLP_SCRATCH]: [2] = the squared norm of this call's gradient (output), the rest is the kernels' own. */
#define CDA_MLP_LOSS_SLOTS 64
/* loss_out6 is f32[8] for this call and for cda_mlp_forward_backward: [0] policy loss, [1] value loss (after the clamp), [2] entropy, [3] total
 * (policy + vf_coef value - ent_coef entropy + kl_coef KL), [4..5] d loss / d log_std, [6] mean KL(rollout policy || current policy) (0 without a KL term), [7] 0.
 * max_norm = INFINITY: no clipping (RLlib's default). */
#define CDA_LOSS_OUT_WORDS 8
int cda_mlp_adam(float* theta, float* adam_m, float* adam_v, float* step_dev, void* wb,
                 const float* slab, int32_t n_chunks, const float* bias_slab, int32_t n_bias_tiles,
                 double* loss_sums5, int64_t loss_samples, float vf_coef, float ent_coef, float kl_coef, float* loss_out6,
                 float lr, float beta1, float beta2, float eps, float max_norm, float* grad, double* scratch, void* stream);
/* The two halves of cda_mlp_adam as calls of their own, for a data-parallel learner (one process per GPU, each with its own shard of markets): between them
 * the caller all-reduces (sums) `grad` f32[CDA_MLP_PARAMS] over the ranks (one ncclAllReduce of 0.9 MB per minibatch step - the only collective of the
 * loop; every rank's loss was normalised with the GLOBAL minibatch size through norm_rows, so the sum IS the global gradient) and cda_mlp_apply recomputes
 * the norm of what arrived. */
int cda_mlp_reduce(const float* slab, int32_t n_chunks, const float* bias_slab, int32_t n_bias_tiles,
                   double* loss_sums5, int64_t loss_samples, float vf_coef, float ent_coef, float kl_coef, float* loss_out6, float* step_dev, float* grad, double* scratch, void* stream);
int cda_mlp_apply(float* theta, float* adam_m, float* adam_v, const float* step_dev, void* wb, const float* grad, int32_t recompute_norm,
                  float lr, float beta1, float beta2, float eps, float max_norm, double* scratch, void* stream);

/* cda_ppo_loss (cda.h) for int32 action arrays - the env's own action tensors as the rollout kernel wrote them.  norm_rows > 0:
 * the means (and the gradient's 1/B) are over norm_rows * agents_per_row samples instead of rows * agents_per_row (a minibatch
 * processed in several sub-batches); sums5 is then NOT cleared and out6 not written unless finish != 0. */
int cda_ppo_loss32(const float* outputs, const float* log_std, const int32_t* a_cat, const int32_t* a_price, const int32_t* a_off,
                   const float* a_cont, const float* logp_old, const float* adv, const float* ret, const int64_t* row_index,
                   int64_t rows, int32_t agents_per_row, int32_t out_stride, float clip, float vf_coef, float ent_coef,
                   float* d_outputs, double* sums5, float* out6, int64_t norm_rows, int32_t clear, int32_t finish, void* stream);

/* One whole rollout of one market chain on `stream` (a loop of launches, no host work in between): for t = 0 .. n_steps - 1
 *     cda_mlp_policy_step(obs[t]) -> env actions [t] | a_cont[t] | logp[t] | value[t]
 *     cda_step_range(env, first_market, n_markets, actions [t]) -> obs[t + 1], reward[t], terminated[t], truncated[t]  (+ auto reset)
 * then cda_mlp_forward's value column for obs[n_steps] -> value[n_steps].  Every buffer is [n_steps (+ 1 for obs, value)][N, ...] of
 * the env's full market count N; chains of disjoint market ranges may run concurrently on different streams.  obs[0] must hold the
 * current observation of the range (copy_first_obs != 0: it is first copied from obs[n_steps], the previous rollout's last). */
typedef struct cda_rollout_bufs {
    float*   obs;            /* [T+1][N][168] */
    int32_t* category;       /* [T][N][A] */
    float*   size_mean;      /* [T][N][A] */
    float*   size_sigma;     /* [T][N][A] */
    int32_t* price;          /* [T][N][A] */
    int32_t* price_offset;   /* [T][N][A] */
    float*   a_cont;         /* [T][N][A][2] */
    float*   logp;           /* [T][N][A] */
    float*   value;          /* [T+1][N] */
    double*  reward;         /* [T][N][A] */
    uint8_t* terminated;     /* [T][N] */
    uint8_t* truncated;      /* [T][N] */
    float*   record;         /* [T][N][A][8] or NULL: the sample records below (words 0..5 written by the policy step) */
    float*   dist;           /* [T][N][CDA_MLP_DIST_LD = 28] or NULL: the rollout policy's distribution per market-step (22 normalised log-probabilities of the categorical heads | the 2 Gaussian
                                means) - what the update's KL term needs (cda_ppo_extra) */
    const cda_info_ptrs* info_steps;   /* [T] (host array) or NULL: step t of this chain also writes the info tensors info_steps[t] (Info_Helper.set_info; the chain then runs
                                the step kernel with info outputs - meant for a small chain of SAMPLED markets beside the info-less ones: train/episode_record.py:197
                                records one episode in N).  Every pointer is indexed by GLOBAL market like all arrays here */
    int32_t* fin_index;      /* [T][N] or NULL: episode-end capture (cda_step_range_capture): the slot in fin_obs of the last observation of an episode that ended at */
    float*   fin_obs;        /*   (t, market), -1 = none (pre-set by the caller); fin_obs [fin_cap][168], fin_count i32[1] (zeroed by the caller before a rollout) */
    int32_t* fin_count;
    int32_t  fin_cap;
    int64_t* counter_bump;   /* NULL, or the chain's OWN rollout counter (normally == counter_dev): incremented by one as the chain's last launch - a caller with one
                                counter per chain then never launches anything ahead of a rollout (a shared counter may only be bumped when no chain is in flight) */
} cda_rollout_bufs;
int cda_mlp_rollout_chain(cda_env* env, const void* wb, const float* theta, int32_t first_market, int32_t n_markets, int32_t n_steps,
                          uint64_t seed, const int64_t* counter_dev, const cda_rollout_bufs* bufs, int32_t copy_first_obs, void* stream);

/* Sample records: what the update's loss reads of a sample, as ONE 32-byte record - a row's A samples are then one contiguous piece (the seven
 * separate per-sample arrays cost seven scattered gathers per row).  Words: */
#define CDA_REC_CATEGORY  0   /* i32 */
#define CDA_REC_PRICE     1   /* i32 */
#define CDA_REC_OFFSET    2   /* i32 */
#define CDA_REC_CONT0     3   /* f32: the raw Gaussian samples */
#define CDA_REC_CONT1     4
#define CDA_REC_LOGP      5   /* f32: log-probability under the rollout's policy */
#define CDA_REC_ADV       6   /* f32: advantage (unnormalised) */
#define CDA_REC_RET       7   /* f32: return */
/* Generalised advantage estimation of a whole rollout, straight from its buffers into the records (ppo.gae's recursion, one launch): reward f64
 * [T][N][A] (scaled by reward_scale here), value f32 [T+1][N] (slot T = the bootstrap value), terminated / truncated u8 [T][N] -> words ADV, RET
 * of rec [T][N][A][8]; stats2 f64[2] receives the sum of the advantages and of their squares. */
int cda_gae_records(const double* reward, const float* value, const uint8_t* terminated, const uint8_t* truncated, int32_t n_steps, int64_t n_markets,
                    int32_t num_agents, float reward_scale, float gamma, float lam, float* rec, double* stats2, void* stream);
/* The same with the time-limit bootstrap RLlib applies (a TRUNCATED, not terminated, step's target continues with V(last observation of the cut episode) instead of 0;
 * the device-side auto reset overwrites that observation, so the rollout captures it: cda_rollout_bufs.fin_*): fin_index i32 [T][N], fin_value f32
 * [max(n_trainable, 1)][fin_value_stride] = cda_mlp_values on the captured list.  n_trainable = 0: one shared policy; > 0: the league layout below. */
int cda_gae_records_bootstrap(const double* reward, const float* value, const uint8_t* terminated, const uint8_t* truncated, int32_t n_steps, int64_t n_markets,
                              int32_t num_agents, int32_t n_trainable, float reward_scale, float gamma, float lam,
                              const int32_t* fin_index, const float* fin_value, int64_t fin_value_stride, float* rec, double* stats, void* stream);
/* value f32 [n_rows] (net p: + p * value_stride) <- the value network of each of n_nets banked networks on obs f32 [n_rows][168] (one launch). */
int cda_mlp_values(const void* wb_bank, const float* theta_bank, int32_t n_nets, const float* obs, int64_t n_rows, float* value, int64_t value_stride, void* stream);
/* ... of the first min(n_rows, *n_rows_dev) rows only (n_rows_dev i32[1] on the device, NULL = all): the captured list's length lives on the device (fin_count) -
 * the grid covers the list's capacity, whole row tiles beyond the count leave at entry (an iteration in which no episode ended costs an empty launch, not a
 * forward pass over `capacity` rows). */
int cda_mlp_values_counted(const void* wb_bank, const float* theta_bank, int32_t n_nets, const float* obs, int64_t n_rows, const int32_t* n_rows_dev,
                           float* value, int64_t value_stride, void* stream);
/* Returns of COMPLETED episodes out of a rollout's buffers (what a learning curve is drawn from when the horizon is shorter than an episode): running f64 [N][A]
 * carries each (market, agent)'s return so far from rollout to rollout; a step that ends the market's episode adds the total to done_sum f64 [A] and 1 to
 * done_count f64 [A] (both accumulate: the caller clears them) and restarts it.  per_slot: what a league needs to credit returns to the MODULE that played a slot. */
int cda_episode_returns(const double* reward, const uint8_t* terminated, const uint8_t* truncated, int32_t n_steps, int64_t n_markets, int32_t num_agents,
                        double* running, double* done_sum, double* done_count, double* per_slot /* f64 [N][A][2] or NULL: this rollout's (sum, number) per (market, agent) */, void* stream);
/* cda_ppo_loss32 reading sample records (rec = [all rows][A][8]; row_index as there).  adv_stats2 (may be NULL) + adv_count: the advantages are
 * normalised on the fly, (adv - mean) / (std + 1e-8) with the unbiased std over the adv_count samples the sums were taken over. */
int cda_ppo_loss_records(const float* outputs, const float* log_std, const float* rec, const double* adv_stats2, int64_t adv_count, const int64_t* row_index,
                         int64_t rows, int32_t agents_per_row, int32_t out_stride, float clip, float vf_coef, float ent_coef,
                         float* d_outputs, double* sums5, float* out6, int64_t norm_rows, int32_t clear, int32_t finish, void* stream);

/* The update's per-row work in ONE launch: the n_rows observation rows obs[perm[i]] (perm NULL: obs[i]) are gathered from the rollout's f32 buffer,
 * run forward, their loss taken from the sample records rec[perm[i]] (cda_ppo_loss_records' rule, advantages normalised on the fly when adv_stats2
 * is given) and back-propagated to the pre-activations - what cda_mlp_prep_rows, cda_mlp_forward_train, cda_ppo_loss_records and cda_mlp_backward do
 * in four launches, without their round trips through HBM (outputs and their gradients stay in LDS, h1 / h2 in registers for tanh').  Written: the
 * packed images x_pk / h1p / h2p / dz1p / dz2p / doutp cda_mlp_wgrad reads, the bias sums (one CDA_MLP_BSLAB row per 64-row tile) and the five loss
 * sums; out / d_out (may be NULL): f32 [n_rows][32] copies of the outputs and their gradients.  Buffers are padded to whole 64-row tiles.
 * clear / finish / out6 / norm_rows: as cda_ppo_loss32, except that the loss sums are f64[CDA_MLP_LOSS_SLOTS][8] (a tile adds into slot `tile mod SLOTS`, words
 * 0..4: one hot cache line would stall every CU's memory pipeline behind its atomics). */
/* extra (may be NULL): what RLlib's PPO objective has beyond clip / vf_coef / ent_coef, and the record stride of a league update.
 *   rec_stride   floats between two rows' records; 0 = agents_per_row * 8 (a row's samples are all the row's agents).  League: a row is a market-step of A slots of which
 *                ONE (slot p of trainable net p) feeds net p's update: rec = records + 8 p, rec_stride = 8 A, agents_per_row = 1.
 *   kl_coef      adds kl_coef * mean KL(rollout policy || current policy) to the loss, the KL exact per row from dist_old f32 [rows of obs][CDA_MLP_DIST_LD]
 *                (cda_rollout_bufs.dist: the row's 22 normalised log-probabilities | 2 means | the 2 log-stds it was sampled with | 2 zeros); the mean KL comes back in
 *                loss_out6[6] (the caller adapts the coefficient, RLlib: x 1.5 above 2 kl_target, x 0.5 below 0.5 kl_target).  0 = no KL term.
 *                log_std_old: not read any more (the rows carry the log-stds since round 6); may be NULL.
 *   vf_clip      the squared value error is clamped to [0, vf_clip] (RLlib's vf_clip_param: clamped samples carry no gradient); <= 0 = off.
 *   sd_log_std   != 0: the STATE-DEPENDENT log-std head (RLlib's default module for Box actions: the policy network emits 2 means AND 2 log-stds per row; the reference's
 *                modules are that default, train/policy/policy_handler.py:69-76).  Output columns 25, 26 of the policy half are log-std OFFSETS: every kernel samples and
 *                scores with log_std[i] = theta[CDA_MLP_OFF_LS + i] + out[25 + i] (rows 25, 26 of Wo / bo are zero in a network built without the head, so this
 *                is the free log_std vector bit for bit).  With the flag the loss sends d loss / d log_std of every row back through columns 25, 26 (rows 25, 26
 *                of Wo and bo train) and the free vector stays where it is (its gradient is reported as zero); without it those columns' gradient is zero. */
#define CDA_MLP_DIST_LD 28
typedef struct cda_ppo_extra {
    int32_t rec_stride;
    float   kl_coef, vf_clip;
    const float* dist_old;
    const float* log_std_old;
    int32_t sd_log_std;
} cda_ppo_extra;
int cda_mlp_forward_backward(const void* wb, const float* theta, const float* obs, const int64_t* perm, int64_t n_rows, int64_t norm_rows,
                             const float* rec, const double* adv_stats2, int64_t adv_count, int32_t agents_per_row, float clip, float vf_coef, float ent_coef,
                             const cda_ppo_extra* extra,
                             void* x_pk, void* h1p, void* h2p, void* dz1p, void* dz2p, void* doutp, float* bias_slab,
                             double* sums5, float* out6, int32_t clear, int32_t finish, float* out, float* d_out, void* stream);

/* ---- league self-play on the same kernels (SURVEY 8(f) rows 1 + 3; the reference: train/train.py:466-503, train/callbk/league_based_self_play_callback.py:1286-1344) ----
 * The reference trains num_trained_agents SEPARATE policies (policy_0 plays slot 0, policy_1 slot 1, ...) against modules drawn per episode and slot from a pool of
 * uniform random modules (RandomRLModule, train/model/model_handler.py:38-53) and frozen champion snapshots.  Here the networks live in BANKS - theta f32
 * [n_nets][CDA_MLP_PARAMS], wb bf16 [n_nets][CDA_MLP_WB_ELEMS], nets 0 .. n_trainable - 1 the trainable ones, the rest frozen snapshots - and slot_net i32 [N][A]
 * (resident in HBM, rewritten by the host between episodes from LeagueSlotMapper.assign) names the net that plays (market, slot): >= 0 a bank row,
 * CDA_LEAGUE_RANDOM the uniform random module.  One launch per step: a (tile, net, half) job per workgroup - both halves of the trainable nets (policy +
 * value), the policy half of a frozen net, and only where one of the tile's slots is played by it; every workgroup samples the slots of ITS net from its
 * own logits (same key / index as cda_mlp_policy_step: a slot played by net n gets bit for bit the action a policy step with n's parameters gives it), the random
 * slots are drawn by net 0's workgroup from include/cda_random_agents.h's stream keyed (random_seed + rollout counter, market, step, slot). */
#define CDA_LEAGUE_RANDOM   (-1)
#define CDA_LEAGUE_MAX_NETS 16
typedef struct cda_league {
    const void*    wb_bank;
    const float*   theta_bank;
    const int32_t* slot_net;
    int32_t        n_nets, n_trainable;
    uint64_t       random_seed;
} cda_league;
/* cda_mlp_policy_step for a league: value f32 [N] of trainable net p at value + p * value_stride; dist (may be NULL) likewise at dist + p * dist_stride; rec (may be
 * NULL) f32 [N][A][8]: words 0..5 of every slot's record (non-network slots: the action words, zeros for the rest). */
int cda_mlp_league_step(const cda_league* league, const float* obs, int32_t first_market, int32_t n_markets, int32_t num_agents,
                        uint64_t seed, const int64_t* counter_dev, int64_t draw,
                        int32_t* env_category, float* env_size_mean, float* env_size_sigma, int32_t* env_price, int32_t* env_price_offset,
                        float* a_cont, float* logp, float* value, int64_t value_stride, float* rec, float* dist, int64_t dist_stride, void* stream);
/* cda_mlp_rollout_chain for a league: bufs as there except value f32 [n_trainable][T+1][N] and dist f32 [n_trainable][T][N][CDA_MLP_DIST_LD]. */
int cda_mlp_league_rollout_chain(cda_env* env, const cda_league* league, int32_t first_market, int32_t n_markets, int32_t n_steps,
                                 uint64_t seed, const int64_t* counter_dev, const cda_rollout_bufs* bufs, int32_t copy_first_obs, void* stream);
/* cda_gae_records for a league: value f32 [n_trainable][T+1][N]; slot p < n_trainable gets advantage / return from net p's values, the other slots' records are left
 * alone; stats2k f64 [n_trainable][2]: per net, the sums its update normalises the advantages with. */
int cda_gae_records_league(const double* reward, const float* value, const uint8_t* terminated, const uint8_t* truncated, int32_t n_steps, int64_t n_markets,
                           int32_t num_agents, int32_t n_trainable, float reward_scale, float gamma, float lam, float* rec, double* stats2k, void* stream);

/* The reference's agent-to-module mapping fn (league_based_self_play_callback.py:1286-1344) for all markets at once, on the device: slot s < n_trainable -> net s;
 * every other slot draws np.random.RandomState((episode_crc[market] + s) mod 2^32).choice(pool, p = probs) - bit for bit: one freshly seeded MT19937's first
 * random_sample(), searchsorted(cumsum(probs) / cumsum(probs)[-1], u, side = "right") - and receives pool_net[draw] (a bank row or CDA_LEAGUE_RANDOM).
 * episode_crc u32 [N] = zlib.crc32(str(episode id)) (host), pool_cdf f64 [pool_size] the normalised cumulative weights, slot_pool (may be NULL) i32 [N][A]: the draw
 * itself (index into the pool; -1 for the trainable slots) - what names the module in an episode record. */
int cda_league_assign(const uint32_t* episode_crc, int32_t n_markets, int32_t num_agents, int32_t n_trainable, const double* pool_cdf, const int32_t* pool_net,
                      int32_t pool_size, int32_t* slot_net, int32_t* slot_pool, void* stream);

/* Device self-test of the operand / accumulator conventions this file is built on: D f32[32][32] = A bf16-rounded f32[32][16] x
 * B f32[16][32] through one v_mfma_f32_32x32x16_bf16 (host pointers; synchronous). */
int cda_mlp_selftest_mfma(int32_t device, const float* a_host, const float* b_host, float* d_host);

#ifdef __cplusplus
}
#endif
#endif /* CDA_MLP_H */
