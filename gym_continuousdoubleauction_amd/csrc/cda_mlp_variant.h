/* cda_mlp_variant.h - one object file of csrc/cda_mlp.hip per history depth.
 *
 * The network kernels are compiled for ONE observation width (CDA_MLP_HIST frames of 42 floats: layer 1's k-steps, W1's operand layout, the LDS tiles, the dW1 slab
 * are compile-time shapes).  The default build (CDA_MLP_HIST = 4, the reference's n_hist) defines the entry points of include/cda_mlp.h under their own names; a
 * build with -DCDA_MLP_HIST=<H> renames every one of them to <name>_h<H> BEFORE the header is read, so declarations and definitions agree and the objects link into
 * one library (__graft_entry__.build_hip compiles CDA_MLP_HIST_VARIANTS).  Entry points that do not depend on the width (GAE, episode returns, league assignment)
 * are renamed too: a few duplicate kilobytes instead of a second source file.  Generated list: keep in step with include/cda_mlp.h (tests/test_capi_load.py checks it). */
#ifndef CDA_MLP_VARIANT_H
#define CDA_MLP_VARIANT_H
#if defined(CDA_MLP_HIST) && CDA_MLP_HIST != 4
#define CDA_MLP_SFX2(n, h) n##_h##h
#define CDA_MLP_SFX1(n, h) CDA_MLP_SFX2(n, h)
#define CDA_MLP_SFX(n) CDA_MLP_SFX1(n, CDA_MLP_HIST)
#define cda_mlp_tile_rows CDA_MLP_SFX(cda_mlp_tile_rows)
#define cda_mlp_wgrad_jobs CDA_MLP_SFX(cda_mlp_wgrad_jobs)
#define cda_mlp_pack CDA_MLP_SFX(cda_mlp_pack)
#define cda_mlp_policy_step CDA_MLP_SFX(cda_mlp_policy_step)
#define cda_mlp_forward CDA_MLP_SFX(cda_mlp_forward)
#define cda_mlp_permutation CDA_MLP_SFX(cda_mlp_permutation)
#define cda_mlp_prep_rows CDA_MLP_SFX(cda_mlp_prep_rows)
#define cda_mlp_forward_train CDA_MLP_SFX(cda_mlp_forward_train)
#define cda_mlp_backward CDA_MLP_SFX(cda_mlp_backward)
#define cda_mlp_wgrad CDA_MLP_SFX(cda_mlp_wgrad)
#define cda_mlp_adam CDA_MLP_SFX(cda_mlp_adam)
#define cda_mlp_reduce CDA_MLP_SFX(cda_mlp_reduce)
#define cda_mlp_apply CDA_MLP_SFX(cda_mlp_apply)
#define cda_ppo_loss32 CDA_MLP_SFX(cda_ppo_loss32)
#define cda_mlp_rollout_chain CDA_MLP_SFX(cda_mlp_rollout_chain)
#define cda_gae_records CDA_MLP_SFX(cda_gae_records)
#define cda_gae_records_bootstrap CDA_MLP_SFX(cda_gae_records_bootstrap)
#define cda_mlp_values CDA_MLP_SFX(cda_mlp_values)
#define cda_mlp_values_counted CDA_MLP_SFX(cda_mlp_values_counted)
#define cda_episode_returns CDA_MLP_SFX(cda_episode_returns)
#define cda_ppo_loss_records CDA_MLP_SFX(cda_ppo_loss_records)
#define cda_mlp_forward_backward CDA_MLP_SFX(cda_mlp_forward_backward)
#define cda_mlp_league_step CDA_MLP_SFX(cda_mlp_league_step)
#define cda_mlp_league_rollout_chain CDA_MLP_SFX(cda_mlp_league_rollout_chain)
#define cda_gae_records_league CDA_MLP_SFX(cda_gae_records_league)
#define cda_league_assign CDA_MLP_SFX(cda_league_assign)
#define cda_mlp_selftest_mfma CDA_MLP_SFX(cda_mlp_selftest_mfma)
#endif
#endif
