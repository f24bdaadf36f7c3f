// cda_mlp.hip - the policy / value network of the PPO loop on the batched env (include/cda_mlp.h; the reference's
// network: config/train_config.json:45-53, trained through RLlib at train/train.py:453-541) as hand-written bf16 MFMA kernels for
// gfx950.  This is the one dense contraction next to the env path, so it is the one place where the matrix cores are used.
//
// Conventions (checked on the device by cda_mlp_selftest_mfma):
//   v_mfma_f32_32x32x16_bf16  D[i][j] += sum_k A[i][k] B[k][j]
//     A operand, lane l: row i = l & 31, the 8 values k = 8 (l >> 5) .. + 7            (8 bf16 = 4 VGPRs)
//     B operand, lane l: column j = l & 31, the same 8 k
//     D, lane l, register r (16 f32): column j = l & 31, row i = (r & 3) + 8 (r >> 2) + 4 (l >> 5)
//   Every product here is "activations x weights^T": i = batch row, j = output feature, so A comes from a row-major bf16 image of the
//   activations in LDS ([row][k], 16-B reads, rows padded by 16 B: conflict free) and B from global memory, where the weights are kept in
//   OPERAND ORDER (cda_mlp_pack / k_adam: a wave's request for one k-step of one feature tile is 1 KB of contiguous memory - see WRing).
//   The accumulator then holds, per lane, ONE feature and 16 rows -
//   which is exactly an operand of the weight-gradient product (dW[i][j] = sum over ROWS of dz[row][i] h[row][j]: lane = feature,
//   the 8 slots = 8 rows; the pairing of slots between A and B is all that matters, not which rows they are).  So activations and
//   pre-activation gradients are written to HBM as the accumulators stand ("packed": 16 B per lane, 1 KB per wave store) and the
//   weight-gradient kernel loads them as MFMA operands with no transposition at all.
//
// The kernels:
//   k_mlp_fwd<MT, MODE>   4 waves = 32 MT rows x one network half (256 features: wave w owns 64, two paired tiles); the rollout's policy step
//                         (MODE_SAMPLE: forward + sampling + records, halves in separate workgroups), the bootstrap value, plain outputs
//   k_mlp_fb              the update: gather + forward + loss + back-propagation of a 64-row tile half, one launch (the default path)
//   k_mlp_fwd8 / k_mlp_bwd8 / k_mlp_bwd / k_ppo_loss*   the same steps as separate kernels (FusedUpdate(fused=False); the tests hold both equal)
//   k_mlp_wgrad, k_grad_reduce, k_adam, k_make_perm, k_prep_rows, k_gae_records
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "cda_mlp_variant.h"        // CDA_MLP_HIST != 4: every entry point of cda_mlp.h gets the suffix _h<H> (one object file per history depth)
#include "../../include/cda_mlp.h"
#include "../../include/cda_random_agents.h"

// The env kernels are built with -ffp-contract=off (their f64 sums must match CPython's); nothing here has such a constraint.
#pragma clang fp contract(fast)

#ifndef CDA_MLP_PF8
#define CDA_MLP_PF8 4
#endif
#ifndef CDA_MLP_PF4
#define CDA_MLP_PF4 4
#endif
#ifndef CDA_MLP_AD
#define CDA_MLP_AD 2
#endif

namespace {
#include "cda_mlp_dev.inc"      // types, tile constants, tanh / MFMA / weight-ring helpers, the sampling arithmetic: shared with csrc/cda_hip.hip (k_policy_step)
// k_mlp_fb's first LDS region: the 64-row observation tile, whose bytes are reused for the output tiles (f32 [64][33] + bf16 [64][40]) once layer 1 has read it -
// the larger of the two (the observation tile from n_hist = 3 on)
constexpr int FB_XS_BYTES = (64 * XS_LD * 2 > 64 * OUTS_LD * 4 + 64 * DO_LD * 2) ? 64 * XS_LD * 2 : 64 * OUTS_LD * 4 + 64 * DO_LD * 2;


// ---- the observation tile: M rows -> LDS image [row][XS_LD] bf16, columns 168 .. 175 zero --------------------------------------
template <int M, int NT = 256>
__device__ __forceinline__ void load_x_bf16(const __bf16* __restrict__ x_rm, long long row0, long long rows_end, __bf16* xs) {
    constexpr int CH = KX / 8, N = M * CH, PER = (N + NT - 1) / NT;              // 22 chunks of 16 B per row; PER chunks per thread
    bf16x8 v[PER];
    #pragma unroll
    for (int u = 0; u < PER; u++) {                                             // every request first ...
        const int c = (int)threadIdx.x + NT * u, cc = c < N ? c : N - 1, r = cc / CH, q = cc - r * CH;
        long long gr = row0 + r; if (gr >= rows_end) gr = rows_end - 1;         // clamp: rows past the end repeat the last one (never used)
        v[u] = *reinterpret_cast<const bf16x8*>(x_rm + gr * KX + q * 8);
    }
    #pragma unroll
    for (int u = 0; u < PER; u++) {                                             // ... then the LDS image
        const int c = (int)threadIdx.x + NT * u, r = c / CH, q = c - r * CH;
        if (c < N) *reinterpret_cast<bf16x8*>(xs + r * XS_LD + q * 8) = v[u];
    }
}
template <int M>
__device__ __forceinline__ void load_x_f32(const float* __restrict__ obs, long long row0, long long rows_end, __bf16* xs) {
    constexpr int CH = KX / VW, N = M * CH, PER = (N + 255) / 256;               // n_hist 4: 44 chunks of 4 values per row (42 real + 2 of zeros)
    obsvec v[PER];
    #pragma unroll
    for (int u = 0; u < PER; u++) {
        const int c = (int)threadIdx.x + 256 * u, cc = c < N ? c : N - 1, r = cc / CH, q = cc - r * CH;
        long long gr = row0 + r; if (gr >= rows_end) gr = rows_end - 1;
        v[u] = obs_zero();
        if (q < OBS / VW) v[u] = *reinterpret_cast<const obsvec*>(obs + gr * OBS + q * VW);
    }
    #pragma unroll
    for (int u = 0; u < PER; u++) {
        const int c = (int)threadIdx.x + 256 * u, r = c / CH, q = c - r * CH;
        if (c < N) obs_to_bf16(xs + r * XS_LD + q * VW, v[u], false);
    }
}

// Feature tiles of the hidden activations come in pairs (one wave's 64 features): position q of tile ft is feature
// 64 (ft / 2) + 2 q + (ft & 1) - in the packed HBM images of h1 / h2 / dz1 / dz2 and wherever they are consumed.
__host__ __device__ __forceinline__ constexpr int feature_of(int ft, int q) { return 64 * (ft >> 1) + 2 * q + (ft & 1); }
template <int MT, int JT, int KSTEPS, int PF, bool PAIRED>
__device__ __forceinline__ void layer_mma(const __bf16* a_lds, int a_ld, WRing<JT, KSTEPS, PF, PAIRED>& W, int lane, f32x16 (&acc)[MT][JT]) {
    constexpr int RING = WRing<JT, KSTEPS, PF, PAIRED>::RING;
    const __bf16* a_base = a_lds + (lane & 31) * a_ld + 8 * (lane >> 5);
    // the A operand (LDS) runs AD steps ahead as well: with one step the read was issued a few dozen cycles before its use (a wave issues its
    // step's MFMAs back to back) and every k-step ended in an exposed LDS round trip, ~200 cycles
    constexpr int AD = CDA_MLP_AD, AR = AD + 1;
    bf16x8 a[AR][MT];
    #pragma unroll
    for (int s0 = 0; s0 < AD && s0 < KSTEPS; s0++)
        #pragma unroll
        for (int it = 0; it < MT; it++) a[s0][it] = *reinterpret_cast<const bf16x8*>(a_base + 32 * it * a_ld + 16 * s0);
    __builtin_amdgcn_s_setprio(1);                                              // a wave in its MFMA loop goes first: its SIMD may host a wave in a VALU epilogue
    #pragma unroll
    for (int ks = 0; ks < KSTEPS; ks++) {
        if (ks + AD < KSTEPS) {
            #pragma unroll
            for (int it = 0; it < MT; it++) a[(ks + AD) % AR][it] = *reinterpret_cast<const bf16x8*>(a_base + 32 * it * a_ld + 16 * (ks + AD));
        }
        #pragma unroll
        for (int it = 0; it < MT; it++)
            #pragma unroll
            for (int jt = 0; jt < JT; jt++) acc[it][jt] = mfma(a[ks % AR][it], W.b[ks % RING][jt], acc[it][jt]);
        if (ks + RING < KSTEPS) {
            #pragma unroll
            for (int jt = 0; jt < JT; jt++) W.b[ks % RING][jt] = *W.piece(jt, ks + RING);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
}

// One accumulator tile -> its packed HBM image: two 16-B stores per lane, 1 KB per wave and store.
__device__ __forceinline__ void store_packed(__bf16* __restrict__ base, long long rt, int nft, int ft, int lane, const float (&v)[16]) {
    typedef __attribute__((ext_vector_type(8))) float f32x8;
    f32x8 f0, f1;
    #pragma unroll
    for (int r = 0; r < 8; r++) { f0[r] = v[r]; f1[r] = v[8 + r]; }
    bf16x8* dst = reinterpret_cast<bf16x8*>(base) + ((rt * nft + ft) * 2) * 64 + lane;
    dst[0] = __builtin_convertvector(f0, bf16x8); dst[64] = __builtin_convertvector(f1, bf16x8);      // 4 + 4 v_cvt_pk_bf16_f32
}
__device__ __forceinline__ void store_packed_bf(__bf16* __restrict__ base, long long rt, int nft, int ft, int lane, const __bf16 (&v)[16]) {
    bf16x8 p0, p1;
    #pragma unroll
    for (int r = 0; r < 8; r++) { p0[r] = v[r]; p1[r] = v[8 + r]; }
    bf16x8* dst = reinterpret_cast<bf16x8*>(base) + ((rt * nft + ft) * 2) * 64 + lane;
    dst[0] = p0; dst[64] = p1;
}
__device__ __forceinline__ void load_packed(const __bf16* __restrict__ base, long long rt, int nft, int ft, int lane, bf16x8 (&p)[2]) {
    const bf16x8* src = reinterpret_cast<const bf16x8*>(base) + ((rt * nft + ft) * 2) * 64 + lane;
    p[0] = src[0]; p[1] = src[64];
}
// Two paired accumulator tiles -> the row-major LDS image: lane (q, h) holds features 2 q and 2 q + 1 of rows rowmap(r, h): one dword per
// row, 32 lanes = 128 contiguous bytes (conflict free), one v_cvt_pk_bf16_f32 + one ds_write_b32 per two values
__device__ __forceinline__ void store_lds_pair(__bf16* act, int ld, int row_base, int col_base, int lane, const float (&v0)[16], const float (&v1)[16]) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const int q = lane & 31, h = lane >> 5;
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    #pragma unroll
    for (int r = 0; r < 16; r++) {
        f32x2 f; f[0] = v0[r]; f[1] = v1[r];
        *reinterpret_cast<bf16x2*>(act + (row_base + rowmap(r, h)) * ld + col_base + 2 * q) = __builtin_convertvector(f, bf16x2);
    }
}

// ---- forward ---------------------------------------------------------------------------------------------------------------
enum { MODE_TRAIN = 0, MODE_OUT = 1, MODE_SAMPLE = 2, MODE_VALUE = 3, MODE_LEAGUE = 4 };
struct FwdArgs {
    const __bf16* x_rm;          // MODE_TRAIN: [n_rows][176] bf16
    const float* obs;            // otherwise: f32 [*][168]
    long long first_row, n_rows; // rows [first_row, first_row + n_rows) of obs / out (MODE_TRAIN: first_row = 0)
    const __bf16* wb; const float* theta;
    __bf16* h1p; __bf16* h2p;    // MODE_TRAIN
    float* out;                  // MODE_TRAIN / MODE_OUT: f32 [*][32] (same row indexing as the input); MODE_SAMPLE / MODE_VALUE: unused
    // MODE_SAMPLE
    int agents; unsigned long long seed; const long long* counter; long long draw;
    int* env_cat; float* env_mean; float* env_sigma; int* env_price; int* env_off; float* a_cont; float* logp; float* value;
    float* rec;                  // optional sample records [*, A][8 words] (include/cda_mlp.h CDA_REC_*): the words the update's loss reads, one line per row
    int split_halves;            // 1: gridDim.y = 2, workgroup (x, y) runs network half y only (policy | value: independent networks; the rollout's launches);
                                 // 2: the value half only (the bootstrap value); 0: both halves, one after the other
    float* dist;                 // optional, MODE_SAMPLE / MODE_LEAGUE: the policy's distribution per ROW, f32 [*][24] = the 22 normalised log-probabilities of the three
                                 // categorical heads | the two Gaussian means (what the KL term of the update needs of the rollout's policy)
    // MODE_LEAGUE (and MODE_VALUE with n_train > 0): wb / theta are BANKS of n_nets networks (nets 0 .. n_train - 1 trainable); gridDim.y enumerates
    // (net, half) jobs: y < 2 n_train: net y / 2, half y & 1; above: the policy half of net n_train + (y - 2 n_train).  slot_net i32 [*, A]: the net that
    // plays (market, slot), < 0 = the uniform random module (drawn by net 0's policy workgroup).  value / dist of net p live p * value_stride / p * dist_stride further on.
    int n_train; const int* slot_net; long long value_stride, dist_stride; unsigned long long random_seed;
    const int* rows_limit;       // MODE_VALUE, optional (device): only rows [first_row, first_row + min(n_rows, *rows_limit)) are evaluated - whole tiles beyond leave at entry
    unsigned long long* dbg; int dbg_block;       // CDA_MLP_TIMING builds (tools/libcda_tools.so) only: cycle stamps of one workgroup, [4 waves][32]
};
#ifdef CDA_MLP_TIMING
#define MLP_MARK(i) do { if (A.dbg && (int)blockIdx.x == A.dbg_block && blockIdx.y == 0 && lane == 0) A.dbg[w * 32 + (i)] = __builtin_readcyclecounter(); } while (0)
#define MLP_MARK8(i) do { if (A.dbg && (int)blockIdx.x == A.dbg_block && lane == 0) A.dbg[w8 * 32 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define MLP_MARK(i) do {} while (0)
#define MLP_MARK8(i) do {} while (0)
#endif


template <int MT, int MODE>
__global__ __launch_bounds__(256) void k_mlp_fwd(FwdArgs A) {
    constexpr int M = 32 * MT, PF = MT == 1 ? 16 : (MT == 2 ? 3 : CDA_MLP_PF4);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16* xs = reinterpret_cast<__bf16*>(smem);                               // [M][XS_LD]
    __bf16* act = xs + M * XS_LD;                                               // [M][ACT_LD]
    float* outs = reinterpret_cast<float*>(act + M * ACT_LD);                   // MODE_SAMPLE: [M][OUTS_LD]
    const int lane = (int)threadIdx.x & 63, w = (int)threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    const long long row0 = A.first_row + (long long)blockIdx.x * M, rows_end = A.first_row + A.n_rows;
    if (MODE == MODE_VALUE && A.rows_limit != nullptr && (long long)blockIdx.x * M >= (long long)A.rows_limit[0]) return;     // (uniform: before any barrier)
    MLP_MARK(0);
    // A league launch (MODE_LEAGUE; the bootstrap values of its trainable nets: MODE_VALUE with n_train > 0): blockIdx.y names the (net, half) job
    // and the net's parameters are its row of the banks.  Uniform per workgroup: scalar registers.
    int net = 0, job_half = 0;
    if (MODE == MODE_LEAGUE) { const int y = (int)blockIdx.y; net = y < 2 * A.n_train ? y >> 1 : A.n_train + (y - 2 * A.n_train); job_half = y < 2 * A.n_train ? (y & 1) : 0; }
    if (MODE == MODE_VALUE && A.n_train > 0) net = (int)blockIdx.y;
    const float* theta = A.theta + (size_t)net * CDA_MLP_PARAMS;
    const __bf16* wb = A.wb + (size_t)net * CDA_MLP_WB_ELEMS;
    if (MODE == MODE_LEAGUE && net >= A.n_train) {
        // a frozen snapshot (champion) is only needed where one of the tile's (market, slot) pairs is played by it: whole workgroups leave otherwise
        int mine = 0;
        for (int s = (int)threadIdx.x; s < M * A.agents; s += 256) {
            const long long grow = row0 + s / A.agents;
            if (grow < rows_end) mine |= A.slot_net[grow * A.agents + (s % A.agents)] == net;
        }
        if (!__syncthreads_or(mine)) return;
    }
    // every bias this lane will add, requested before anything else (a request placed later retires behind a whole ring of weight
    // requests: vmcnt counts in order)
    float b1s[2][2], b2s[2][2];
    #pragma unroll
    for (int hf = 0; hf < 2; hf++)
        #pragma unroll
        for (int jt = 0; jt < 2; jt++) {
            b1s[hf][jt] = theta[CDA_MLP_OFF_B1 + 256 * hf + 64 * w + 2 * j + jt] * TWO_LOG2E;
            b2s[hf][jt] = theta[CDA_MLP_OFF_B2 + 256 * hf + 64 * w + 2 * j + jt] * TWO_LOG2E;
        }
    const float bo = theta[CDA_MLP_OFF_BO + j];
    const __bf16* W1b = wb + CDA_MLP_WB_W1; const __bf16* W2b = wb + CDA_MLP_WB_W2; const __bf16* Wob = wb + CDA_MLP_WB_WO;
    f32x16 acc3[1][1]; acc3[0][0] = zero16();                                   // heads: wave w owns row tile w (waves >= MT idle there)
    WRing<2, KX / 16, PF, true> R1; WRing<2, HID / 16, PF, true> R2; WRing<1, HID / 16, PF> RO;
    // The two halves are independent networks: a rollout launch gives each its own workgroup (half the serial chain, half the weight bytes
    // through one CU's L1); the update's launches run both in one workgroup (the observation tile is staged once).
    const int half_begin = MODE == MODE_LEAGUE ? job_half : (A.split_halves == 1 ? (int)blockIdx.y : (A.split_halves == 2 ? 1 : 0));
    const int half_end = (MODE == MODE_LEAGUE || A.split_halves) ? half_begin + 1 : 2;
    float* const value_out = A.value ? A.value + (size_t)net * A.value_stride : nullptr;
    R1.prime(W1b + (size_t)(256 * half_begin + 64 * w) * KX, KX, lane);         // (layer 1's weights fly while the observation tile is staged)
    if (MODE == MODE_TRAIN) load_x_bf16<M>(A.x_rm, row0, rows_end, xs); else load_x_f32<M>(A.obs, row0, rows_end, xs);
    __syncthreads();
    MLP_MARK(1);
    #pragma unroll 1
    for (int half = half_begin; half < half_end; half++) {
        const int f0 = 256 * half + 64 * w;                                     // this wave's first feature (of 512)
        {   // layer 1: [M, 176] x W1[f0 .. f0 + 63]^T
            f32x16 acc[MT][2];
            #pragma unroll
            for (int it = 0; it < MT; it++) { acc[it][0] = zero16(); acc[it][1] = zero16(); }
            layer_mma(xs, XS_LD, R1, lane, acc);
            MLP_MARK(2 + 8 * half);
            R2.prime(W2b + ((size_t)half * HID + 64 * w) * HID, HID, lane);     // (in flight across the epilogue and its barrier)
            if (half != half_begin) __syncthreads();                            // the heads of the previous half still read `act`
            const float bias0 = half ? b1s[1][0] : b1s[0][0], bias1 = half ? b1s[1][1] : b1s[0][1];
            #pragma unroll
            for (int it = 0; it < MT; it++) {
                float v0[16], v1[16];
                #pragma unroll
                for (int r = 0; r < 16; r++) { v0[r] = tanh_biased(acc[it][0][r], bias0); v1[r] = tanh_biased(acc[it][1][r], bias1); }
                if (MODE == MODE_TRAIN) {                                       // (buffers are padded to whole workgroup tiles: no tail predicate)
                    store_packed(A.h1p, row0 / 32 + it, 16, (f0 >> 5), lane, v0);
                    store_packed(A.h1p, row0 / 32 + it, 16, (f0 >> 5) + 1, lane, v1);
                }
                store_lds_pair(act, ACT_LD, 32 * it, 64 * w, lane, v0, v1);
                __builtin_amdgcn_sched_barrier(0);                              // (one row tile's tanh chains at a time: register pressure)
            }
        }
        MLP_MARK(3 + 8 * half);
        __syncthreads();
        MLP_MARK(4 + 8 * half);
        {   // layer 2: [M, 256] x W2[half][64 w .. + 63]^T
            f32x16 acc[MT][2];
            #pragma unroll
            for (int it = 0; it < MT; it++) { acc[it][0] = zero16(); acc[it][1] = zero16(); }
            layer_mma(act, ACT_LD, R2, lane, acc);
            MLP_MARK(5 + 8 * half);
            RO.prime(Wob + (size_t)half * NOUT * HID, HID, lane);
            __syncthreads();                                                    // every wave has read h1: h2 takes its place
            const float bias0 = half ? b2s[1][0] : b2s[0][0], bias1 = half ? b2s[1][1] : b2s[0][1];
            #pragma unroll
            for (int it = 0; it < MT; it++) {
                float v0[16], v1[16];
                #pragma unroll
                for (int r = 0; r < 16; r++) { v0[r] = tanh_biased(acc[it][0][r], bias0); v1[r] = tanh_biased(acc[it][1][r], bias1); }
                if (MODE == MODE_TRAIN) {
                    store_packed(A.h2p, row0 / 32 + it, 16, (f0 >> 5), lane, v0);
                    store_packed(A.h2p, row0 / 32 + it, 16, (f0 >> 5) + 1, lane, v1);
                }
                store_lds_pair(act, ACT_LD, 32 * it, 64 * w, lane, v0, v1);
                __builtin_amdgcn_sched_barrier(0);                              // (one row tile's tanh chains at a time: register pressure)
            }
        }
        MLP_MARK(6 + 8 * half);
        if (half + 1 < half_end) R1.prime(W1b + (size_t)(256 + 64 * w) * KX, KX, lane);   // the value half's first layer, requested across the heads
        __syncthreads();
        MLP_MARK(7 + 8 * half);
        // heads: [32 rows of tile w, 256] x Wob[half]^T (the other half's rows of Wob are zero); waves >= MT multiply a tile nobody reads
        layer_mma(act + (w < MT ? 32 * w : 0) * ACT_LD, ACT_LD, RO, lane, acc3);
        MLP_MARK(8 + 8 * half);
    }
    // outputs: column j of rows rowmap(r, h) of row tile w.  A workgroup that ran one half only owns that half's columns (policy: 0 .. 23 and the
    // zero padding; value: 24)
    const bool own_col = !A.split_halves || (half_begin == 0 ? j != N_LOGITS : j == N_LOGITS);
    if (w < MT) {
        #pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = 32 * w + rowmap(r, h);
            const float o = acc3[0][0][r] + bo;
            if (MODE == MODE_SAMPLE || MODE == MODE_LEAGUE) { if (half_begin == 0) outs[row * OUTS_LD + j] = o; else if (j == N_LOGITS && row0 + row < rows_end) value_out[row0 + row] = o; }
            else if (MODE == MODE_VALUE) { if (j == N_LOGITS && row0 + row < rows_end) value_out[row0 + row] = o; }
            else if (own_col && (MODE == MODE_TRAIN || row0 + row < rows_end)) A.out[(row0 + row) * NOUT + j] = o;
        }
    }
    MLP_MARK(18);
    if ((MODE == MODE_SAMPLE || MODE == MODE_LEAGUE) && half_begin == 0) {
        __syncthreads();
        // one thread per (row, agent) sample: three categorical heads by inverse CDF, two Gaussian heads by Box-Muller, the action's
        // log-probability, and the env's five action words (size_mean = tanh, size_sigma = sigmoid: the Box bounds of
        // action_helper.py:126-138)
        const int ag = A.agents;
        const unsigned long long key = rollout_key(A.seed, A.counter[0], A.draw);
        const float ls0 = theta[CDA_MLP_OFF_LS], ls1 = theta[CDA_MLP_OFF_LS + 1];
        for (int s = (int)threadIdx.x; s < M * ag; s += 256) {
            const int row = s / ag, a = s - row * ag;
            const long long grow = row0 + row;
            if (grow >= rows_end) continue;
            const long long i = grow * ag + a;
            if (MODE == MODE_LEAGUE) {
                // the slot's module: this net -> sampled below; the uniform random module (RandomRLModule's law, train/model/model_handler.py:38-53;
                // the counter-based stream of include/cda_random_agents.h keyed (random_seed + rollout counter, market, step, slot)) -> drawn by net 0's
                // workgroup; any other net -> that net's workgroup writes the slot
                const int mod = A.slot_net[i];
                if (mod != net) {
                    if (net == 0 && mod < 0) {
                        int c, p, o; float sm, ss;
                        cda_random_action(A.random_seed + (unsigned long long)A.counter[0] * 0x9e3779b97f4a7c15ull, (unsigned long long)grow, (unsigned int)A.draw, (unsigned int)a, &c, &sm, &ss, &p, &o);
                        A.env_cat[i] = c; A.env_price[i] = p; A.env_off[i] = o; A.env_mean[i] = sm; A.env_sigma[i] = ss;
                        A.a_cont[2 * i] = 0.0f; A.a_cont[2 * i + 1] = 0.0f; A.logp[i] = 0.0f;
                        if (A.rec) {
                            float4* rp = reinterpret_cast<float4*>(A.rec + 8 * i);
                            rp[0] = make_float4(__int_as_float(c), __int_as_float(p), __int_as_float(o), 0.0f);
                            *reinterpret_cast<float2*>(A.rec + 8 * i + 4) = make_float2(0.0f, 0.0f);
                        }
                    }
                    continue;
                }
            }
            float l[N_LOGITS];
            #pragma unroll
            for (int q = 0; q < N_LOGITS; q++) l[q] = outs[row * OUTS_LD + q];
            if (MODE == MODE_SAMPLE && a == 0 && !A.split_halves) A.value[grow] = outs[row * OUTS_LD + N_LOGITS];
            // (log-std = the free vector + the row's outputs 25, 26: the state-dependent head's offsets, zero in a network built without it - include/cda_mlp.h sd_log_std)
            const SampledAction sa = sample_action(l, key, i, ls0 + outs[row * OUTS_LD + N_LOGITS + 1], ls1 + outs[row * OUTS_LD + N_LOGITS + 2]);         // (cda_mlp_dev.inc: the arithmetic k_policy_step shares)
            const int c = sa.cat, p = sa.price, o = sa.off;
            const float x0 = sa.x0, x1 = sa.x1, lp = sa.logp;
            A.env_cat[i] = c; A.env_price[i] = p; A.env_off[i] = o;
            A.env_mean[i] = sa.size_mean;
            A.env_sigma[i] = sa.size_sigma;
            A.a_cont[2 * i] = x0; A.a_cont[2 * i + 1] = x1;
            A.logp[i] = lp;
            if (A.rec) {
                float4* rp = reinterpret_cast<float4*>(A.rec + 8 * i);
                rp[0] = make_float4(__int_as_float(c), __int_as_float(p), __int_as_float(o), x0);
                *reinterpret_cast<float2*>(A.rec + 8 * i + 4) = make_float2(x1, lp);
            }
        }
        if (A.dist && (MODE == MODE_SAMPLE || net < A.n_train)) {
            // the rollout policy's distribution of every row, for the update's KL term: 22 normalised log-probabilities | 2 means | 2 log-stds | 2 zeros (a thread per row)
            float* dist = A.dist + (size_t)net * A.dist_stride;
            for (int row = (int)threadIdx.x; row < M; row += 256) {
                const long long grow = row0 + row;
                if (grow >= rows_end) continue;
                float l[N_LOGITS], o24[N_LOGITS];
                #pragma unroll
                for (int q = 0; q < N_LOGITS; q++) l[q] = outs[row * OUTS_LD + q];
                dist_row(l, o24);
                float4* dp = reinterpret_cast<float4*>(dist + grow * CDA_MLP_DIST_LD);
                #pragma unroll
                for (int q = 0; q < N_LOGITS / 4; q++) dp[q] = make_float4(o24[4 * q], o24[4 * q + 1], o24[4 * q + 2], o24[4 * q + 3]);
                dp[N_LOGITS / 4] = make_float4(ls0 + outs[row * OUTS_LD + N_LOGITS + 1], ls1 + outs[row * OUTS_LD + N_LOGITS + 2], 0.0f, 0.0f);   // the log-stds the row was sampled with
            }
        }
    }
    MLP_MARK(19);
}

// ---- the update's forward with BOTH halves in flight: 8 waves, the value half one stage behind the policy half ---------------------------
// A tile's time in k_mlp_fwd is MFMA loops + tanh epilogues (VALU: two transcendentals per value, ~56 cycles) one after the other, with one wave
// per SIMD: the matrix pipe idles during every epilogue.  Here waves 0-3 run the policy network and waves 4-7 the value network of the same rows,
// each half through its own LDS buffer, and the value half simply starts ONE STAGE LATER (an extra barrier at its start, one at the policy
// half's end): while one half multiplies, the other one's epilogue runs on the same SIMDs' vector units.
//     stage        0     1     2     3     4     5
//     policy      M1    E1    M2    E2    MH     -
//     value        -    M1    E1    M2    E2    MH
template <int MT>
__global__ __launch_bounds__(512) void k_mlp_fwd8(FwdArgs A) {
    constexpr int M = 32 * MT, PF = CDA_MLP_PF8;         // weight requests in flight: L2 answers in ~1000 cycles under load, a k-step multiplies for 128
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16* xs = reinterpret_cast<__bf16*>(smem);                               // [M][XS_LD]
    const int lane = (int)threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int w8 = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), half = w8 >> 2, w = w8 & 3;
    __bf16* act = xs + M * XS_LD + half * (M * ACT_LD);                         // this half's [M][ACT_LD]
    const long long row0 = (long long)blockIdx.x * M, rows_end = A.n_rows;
    const int f0 = 256 * half + 64 * w;
    const float b1_0 = A.theta[CDA_MLP_OFF_B1 + f0 + 2 * j] * TWO_LOG2E, b1_1 = A.theta[CDA_MLP_OFF_B1 + f0 + 2 * j + 1] * TWO_LOG2E;
    const float b2_0 = A.theta[CDA_MLP_OFF_B2 + f0 + 2 * j] * TWO_LOG2E, b2_1 = A.theta[CDA_MLP_OFF_B2 + f0 + 2 * j + 1] * TWO_LOG2E;
    const float bo = A.theta[CDA_MLP_OFF_BO + j];
    MLP_MARK8(0);
    load_x_bf16<M, 512>(A.x_rm, row0, rows_end, xs);
    const __bf16* W1b = A.wb + CDA_MLP_WB_W1; const __bf16* W2b = A.wb + CDA_MLP_WB_W2; const __bf16* Wob = A.wb + CDA_MLP_WB_WO;
    WRing<2, KX / 16, PF, true> R1; WRing<2, HID / 16, PF, true> R2; WRing<1, HID / 16, PF> RO;
    R1.prime(W1b + (size_t)f0 * KX, KX, lane);
    __syncthreads();                                                            // the observation tile is in LDS
    MLP_MARK8(1);
    if (half == 1) __syncthreads();                                             // stage 0: the value half waits
    MLP_MARK8(2);
    f32x16 acc[MT][2];
    #pragma unroll
    for (int it = 0; it < MT; it++) { acc[it][0] = zero16(); acc[it][1] = zero16(); }
    layer_mma(xs, XS_LD, R1, lane, acc);                                        // M1
    MLP_MARK8(3);
    R2.prime(W2b + ((size_t)half * HID + 64 * w) * HID, HID, lane);
    __syncthreads();
    MLP_MARK8(4);
    #pragma unroll
    for (int it = 0; it < MT; it++) {                                           // E1
        float v0[16], v1[16];
        #pragma unroll
        for (int r = 0; r < 16; r++) { v0[r] = tanh_biased(acc[it][0][r], b1_0); v1[r] = tanh_biased(acc[it][1][r], b1_1); }
        store_packed(A.h1p, row0 / 32 + it, 16, (f0 >> 5), lane, v0);
        store_packed(A.h1p, row0 / 32 + it, 16, (f0 >> 5) + 1, lane, v1);
        store_lds_pair(act, ACT_LD, 32 * it, 64 * w, lane, v0, v1);
        __builtin_amdgcn_sched_barrier(0);
    }
    MLP_MARK8(5);
    __syncthreads();
    MLP_MARK8(6);
    #pragma unroll
    for (int it = 0; it < MT; it++) { acc[it][0] = zero16(); acc[it][1] = zero16(); }
    layer_mma(act, ACT_LD, R2, lane, acc);                                      // M2
    MLP_MARK8(7);
    RO.prime(Wob + (size_t)half * NOUT * HID, HID, lane);
    __syncthreads();                                                            // every wave of the half has read h1: h2 takes its place
    MLP_MARK8(8);
    #pragma unroll
    for (int it = 0; it < MT; it++) {                                           // E2
        float v0[16], v1[16];
        #pragma unroll
        for (int r = 0; r < 16; r++) { v0[r] = tanh_biased(acc[it][0][r], b2_0); v1[r] = tanh_biased(acc[it][1][r], b2_1); }
        store_packed(A.h2p, row0 / 32 + it, 16, (f0 >> 5), lane, v0);
        store_packed(A.h2p, row0 / 32 + it, 16, (f0 >> 5) + 1, lane, v1);
        store_lds_pair(act, ACT_LD, 32 * it, 64 * w, lane, v0, v1);
        __builtin_amdgcn_sched_barrier(0);
    }
    MLP_MARK8(9);
    __syncthreads();
    MLP_MARK8(10);
    f32x16 acc3[1][1]; acc3[0][0] = zero16();                                   // MH: wave w of the half owns row tile w (waves >= MT multiply a tile nobody reads)
    layer_mma(act + (w < MT ? 32 * w : 0) * ACT_LD, ACT_LD, RO, lane, acc3);
    MLP_MARK8(11);
    const bool own_col = half == 0 ? j != N_LOGITS : j == N_LOGITS;             // policy: columns 0 .. 23 and the zero padding; value: column 24
    if (w < MT && own_col) {
        #pragma unroll
        for (int r = 0; r < 16; r++) A.out[(row0 + 32 * w + rowmap(r, h)) * NOUT + j] = acc3[0][0][r] + bo;
    }
    MLP_MARK8(12);
    if (half == 0) __syncthreads();                                             // stage 5: the policy half's matching barrier
    MLP_MARK8(13);
}

// ---- update: the epoch's shuffle as a keyed bijection (no sort) -----------------------------------------------------------------------
// perm[i] = walk(i): a bijective mixer on [0, 2^bits) (add, odd multiply, xor-shift: each step invertible), iterated until the value falls
// below n (cycle walking: < 2 rounds on average, since 2^bits < 2 n).  torch.randperm is a device sort: ~10 launches, 130 us per epoch.
__device__ __forceinline__ unsigned int perm_mix(unsigned int x, int bits, unsigned long long key) {
    const unsigned int mask = bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u);
    const int s1 = (bits + 1) / 2, s2 = (2 * bits + 2) / 3;
    #pragma unroll
    for (int r = 0; r < 4; r++) {
        x = (x + (unsigned int)(key >> (16 * r))) & mask;
        x = (x * 0x9E3779B1u) & mask;
        x ^= x >> s1;
        x = (x * 0x85EBCA6Bu) & mask;
        x ^= x >> s2;
    }
    return x;
}
__global__ void k_make_perm(unsigned long long key, long long n, int bits, long long* __restrict__ perm) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned int x = (unsigned int)i;
    do { x = perm_mix(x, bits, key); } while ((long long)x >= n);
    perm[i] = (long long)x;
}

// ---- update, step 0: gather + convert + both images of the observation rows ---------------------------------------------------
// one workgroup per 32-row tile: the tile goes through LDS as f32 [32][32 XT + 1]; row-major bf16 rows out of it, and the packed image
// (lane = (feature, row half), slots = rows)
__global__ __launch_bounds__(256) void k_prep_rows(const float* __restrict__ obs, const long long* __restrict__ perm, long long n_rows,
                                                   __bf16* __restrict__ x_rm, __bf16* __restrict__ x_pk) {
    __shared__ float t[32][32 * XT + 1];
    const long long rt = blockIdx.x;
    constexpr int CPR = 32 * XT / VW;                                           // chunks of VW floats per row (n_hist 4: 48 of 4: 42 real, 6 of zeros)
    for (int c = (int)threadIdx.x; c < 32 * CPR; c += 256) {
        const int r = c / CPR, q = c - r * CPR;
        const long long src = perm ? perm[rt * 32 + r] : rt * 32 + r;
        obsvec v = obs_zero();
        if (q < OBS / VW) v = *reinterpret_cast<const obsvec*>(obs + src * OBS + q * VW);
        obs_spread(&t[r][VW * q], v);
    }
    __syncthreads();
    for (int c = (int)threadIdx.x; c < 32 * (KX / 8); c += 256) {               // row-major: 22 chunks of 8 per row
        const int r = c / (KX / 8), q = c - r * (KX / 8);
        bf16x8 v;
        #pragma unroll
        for (int e = 0; e < 8; e++) v[e] = (__bf16)t[r][8 * q + e];
        *reinterpret_cast<bf16x8*>(x_rm + (rt * 32 + r) * KX + 8 * q) = v;
    }
    for (int c = (int)threadIdx.x; c < XT * 2 * 64; c += 256) {                 // packed: [ft][ks][lane] 16-B pieces
        const int lane = c & 63, ks = (c >> 6) & 1, ft = c >> 7;
        const int j = lane & 31, h = lane >> 5;
        bf16x8 v;
        #pragma unroll
        for (int e = 0; e < 8; e++) v[e] = (__bf16)t[rowmap(8 * ks + e, h)][32 * ft + j];
        reinterpret_cast<bf16x8*>(x_pk)[((rt * XT + ft) * 2 + ks) * 64 + lane] = v;
    }
}

// ---- update, step 2: back-propagation to the pre-activations -------------------------------------------------------------------
struct BwdArgs {
    const __bf16* wb; const float* d_out; const __bf16* h1p; const __bf16* h2p; long long n_rows;
    __bf16* dz1p; __bf16* dz2p; __bf16* doutp; float* bias_slab;
};
template <int MT>
__global__ __launch_bounds__(256) void k_mlp_bwd(BwdArgs A) {
    constexpr int M = 32 * MT, PF = MT == 1 ? 16 : (MT == 2 ? 3 : CDA_MLP_PF4);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16* dos = reinterpret_cast<__bf16*>(smem);                              // [M][DO_LD]   d_out, bf16 (the A operand)
    __bf16* dact = dos + M * DO_LD;                                             // [M][ACT_LD]  dz2 of the current half
    float* dof = reinterpret_cast<float*>(dact + M * ACT_LD);                   // [M][OUTS_LD] d_out, f32 (bias sums)
    const int lane = (int)threadIdx.x & 63, w = (int)threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    const long long row0 = (long long)blockIdx.x * M;
    float* bs = A.bias_slab + (size_t)blockIdx.x * CDA_MLP_BSLAB;
    for (int c = (int)threadIdx.x; c < M * (NOUT / 4); c += 256) {
        const int r = c / (NOUT / 4), q = c - r * (NOUT / 4);
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (row0 + r < A.n_rows) v = *reinterpret_cast<const float4*>(A.d_out + (row0 + r) * NOUT + 4 * q);
        bf16x4 b; b[0] = (__bf16)v.x; b[1] = (__bf16)v.y; b[2] = (__bf16)v.z; b[3] = (__bf16)v.w;
        *reinterpret_cast<bf16x4*>(dos + r * DO_LD + 4 * q) = b;
        dof[r * OUTS_LD + 4 * q] = v.x; dof[r * OUTS_LD + 4 * q + 1] = v.y; dof[r * OUTS_LD + 4 * q + 2] = v.z; dof[r * OUTS_LD + 4 * q + 3] = v.w;
    }
    __syncthreads();
    // d_out in the packed layout (lane = output column, slots = rows) for the heads' weight gradient, and its column sums
    if (w < MT) {
        __bf16 v[16];
        #pragma unroll
        for (int r = 0; r < 16; r++) v[r] = dos[(32 * w + rowmap(r, h)) * DO_LD + j];
        store_packed_bf(A.doutp, row0 / 32 + w, 1, 0, lane, v);
    }
    if (threadIdx.x < NOUT) {
        float s = 0.0f;
        for (int r = 0; r < M; r++) s += dof[r * OUTS_LD + threadIdx.x];
        bs[2 * CDA_MLP_FEAT + threadIdx.x] = s;
    }
    const __bf16* W2T = A.wb + CDA_MLP_WB_W2T; const __bf16* WoT = A.wb + CDA_MLP_WB_WOT;
    WRing<2, NOUT / 16, PF, true> RO; WRing<2, HID / 16, PF, true> R2;
    RO.prime(WoT + (size_t)(64 * w) * NOUT, NOUT, lane);
    #pragma unroll 1
    for (int half = 0; half < 2; half++) {
        const int f0 = 256 * half + 64 * w, ft0 = f0 >> 5;
        {   // dH2 = d_out x Wo (K = 32), times tanh'
            // h2 in the accumulators' own layout (the packed image the forward wrote), requested before the product
            bf16x8 hp[MT][2][2];
            #pragma unroll
            for (int it = 0; it < MT; it++) {
                const long long rt = row0 / 32 + it;                            // (buffers are padded to whole workgroup tiles; d_out is zero past the end)
                load_packed(A.h2p, rt, 16, ft0, lane, hp[it][0]); load_packed(A.h2p, rt, 16, ft0 + 1, lane, hp[it][1]);
            }
            f32x16 acc[MT][2];
            #pragma unroll
            for (int it = 0; it < MT; it++) { acc[it][0] = zero16(); acc[it][1] = zero16(); }
            layer_mma(dos, DO_LD, RO, lane, acc);
            R2.prime(W2T + ((size_t)half * HID + 64 * w) * HID, HID, lane);
            if (half == 1) __syncthreads();                                     // dH1 of half 0 still reads `dact`
            float colsum0 = 0.0f, colsum1 = 0.0f;
            #pragma unroll
            for (int it = 0; it < MT; it++) {
                float v0[16], v1[16];
                #pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float h0 = (float)hp[it][0][r >> 3][r & 7], h1 = (float)hp[it][1][r >> 3][r & 7];
                    v0[r] = (float)(__bf16)(acc[it][0][r] * (1.0f - h0 * h0)); v1[r] = (float)(__bf16)(acc[it][1][r] * (1.0f - h1 * h1));
                    colsum0 += v0[r]; colsum1 += v1[r];
                }
                store_packed(A.dz2p, row0 / 32 + it, 16, ft0, lane, v0); store_packed(A.dz2p, row0 / 32 + it, 16, ft0 + 1, lane, v1);
                store_lds_pair(dact, ACT_LD, 32 * it, 64 * w, lane, v0, v1);
                __builtin_amdgcn_sched_barrier(0);
            }
            colsum0 += __shfl_xor(colsum0, 32, 64); colsum1 += __shfl_xor(colsum1, 32, 64);
            if (h == 0) { bs[CDA_MLP_FEAT + f0 + 2 * j] = colsum0; bs[CDA_MLP_FEAT + f0 + 2 * j + 1] = colsum1; }
        }
        __syncthreads();
        {   // dH1 = dz2 x W2[half] (K = 256), times tanh'
            bf16x8 hp[MT][2][2];
            #pragma unroll
            for (int it = 0; it < MT; it++) {
                const long long rt = row0 / 32 + it;
                load_packed(A.h1p, rt, 16, ft0, lane, hp[it][0]); load_packed(A.h1p, rt, 16, ft0 + 1, lane, hp[it][1]);
            }
            f32x16 acc[MT][2];
            #pragma unroll
            for (int it = 0; it < MT; it++) { acc[it][0] = zero16(); acc[it][1] = zero16(); }
            layer_mma(dact, ACT_LD, R2, lane, acc);
            if (half == 0) RO.prime(WoT + (size_t)(256 + 64 * w) * NOUT, NOUT, lane);
            float colsum0 = 0.0f, colsum1 = 0.0f;
            #pragma unroll
            for (int it = 0; it < MT; it++) {
                float v0[16], v1[16];
                #pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float h0 = (float)hp[it][0][r >> 3][r & 7], h1 = (float)hp[it][1][r >> 3][r & 7];
                    v0[r] = (float)(__bf16)(acc[it][0][r] * (1.0f - h0 * h0)); v1[r] = (float)(__bf16)(acc[it][1][r] * (1.0f - h1 * h1));
                    colsum0 += v0[r]; colsum1 += v1[r];
                }
                store_packed(A.dz1p, row0 / 32 + it, 16, ft0, lane, v0); store_packed(A.dz1p, row0 / 32 + it, 16, ft0 + 1, lane, v1);
                __builtin_amdgcn_sched_barrier(0);
            }
            colsum0 += __shfl_xor(colsum0, 32, 64); colsum1 += __shfl_xor(colsum1, 32, 64);
            if (h == 0) { bs[f0 + 2 * j] = colsum0; bs[f0 + 2 * j + 1] = colsum1; }
        }
    }
}

// The backward kernel in the same arrangement: 8 waves, the value half one stage behind.
//     stage        0          1          2       3
//     policy   MdH2 + E2     MdH1        E1      -
//     value        -       MdH2 + E2    MdH1     E1
template <int MT>
__global__ __launch_bounds__(512) void k_mlp_bwd8(BwdArgs A) {
    constexpr int M = 32 * MT, PF = CDA_MLP_PF8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16* dos = reinterpret_cast<__bf16*>(smem);                              // [M][DO_LD]   d_out, bf16 (the A operand)
    float* dof = reinterpret_cast<float*>(dos + M * DO_LD);                     // [M][OUTS_LD] d_out, f32 (bias sums)
    const int lane = (int)threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int w8 = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), half = w8 >> 2, w = w8 & 3;
    __bf16* dact = reinterpret_cast<__bf16*>(dof + M * OUTS_LD) + half * (M * ACT_LD);   // this half's [M][ACT_LD]
    const long long row0 = (long long)blockIdx.x * M;
    float* bs = A.bias_slab + (size_t)blockIdx.x * CDA_MLP_BSLAB;
    for (int c = (int)threadIdx.x; c < M * (NOUT / 4); c += 512) {
        const int r = c / (NOUT / 4), q = c - r * (NOUT / 4);
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (row0 + r < A.n_rows) v = *reinterpret_cast<const float4*>(A.d_out + (row0 + r) * NOUT + 4 * q);
        bf16x4 b; b[0] = (__bf16)v.x; b[1] = (__bf16)v.y; b[2] = (__bf16)v.z; b[3] = (__bf16)v.w;
        *reinterpret_cast<bf16x4*>(dos + r * DO_LD + 4 * q) = b;
        dof[r * OUTS_LD + 4 * q] = v.x; dof[r * OUTS_LD + 4 * q + 1] = v.y; dof[r * OUTS_LD + 4 * q + 2] = v.z; dof[r * OUTS_LD + 4 * q + 3] = v.w;
    }
    const int f0 = 256 * half + 64 * w, ft0 = f0 >> 5;
    const __bf16* W2T = A.wb + CDA_MLP_WB_W2T; const __bf16* WoT = A.wb + CDA_MLP_WB_WOT;
    WRing<2, NOUT / 16, PF, true> RO; WRing<2, HID / 16, PF, true> R2;
    RO.prime(WoT + (size_t)f0 * NOUT, NOUT, lane);
    __syncthreads();
    if (w8 < MT) {                                                              // d_out in the packed layout (lane = output column, slots = rows)
        __bf16 v[16];
        #pragma unroll
        for (int r = 0; r < 16; r++) v[r] = dos[(32 * w8 + rowmap(r, h)) * DO_LD + j];
        store_packed_bf(A.doutp, row0 / 32 + w8, 1, 0, lane, v);
    }
    if (threadIdx.x >= 448 && threadIdx.x < 448 + NOUT) {                        // (the last wave: not one that stores doutp)
        const int o = (int)threadIdx.x - 448;
        float sum = 0.0f;
        for (int r = 0; r < M; r++) sum += dof[r * OUTS_LD + o];
        bs[2 * CDA_MLP_FEAT + o] = sum;
    }
    if (half == 1) __syncthreads();                                             // stage 0: the value half waits
    {   // MdH2 + E2
        bf16x8 hp[MT][2][2];
        #pragma unroll
        for (int it = 0; it < MT; it++) { load_packed(A.h2p, row0 / 32 + it, 16, ft0, lane, hp[it][0]); load_packed(A.h2p, row0 / 32 + it, 16, ft0 + 1, lane, hp[it][1]); }
        f32x16 acc[MT][2];
        #pragma unroll
        for (int it = 0; it < MT; it++) { acc[it][0] = zero16(); acc[it][1] = zero16(); }
        layer_mma(dos, DO_LD, RO, lane, acc);
        R2.prime(W2T + ((size_t)half * HID + 64 * w) * HID, HID, lane);
        float colsum0 = 0.0f, colsum1 = 0.0f;
        #pragma unroll
        for (int it = 0; it < MT; it++) {
            float v0[16], v1[16];
            #pragma unroll
            for (int r = 0; r < 16; r++) {
                const float h0 = (float)hp[it][0][r >> 3][r & 7], h1 = (float)hp[it][1][r >> 3][r & 7];
                v0[r] = (float)(__bf16)(acc[it][0][r] * (1.0f - h0 * h0)); v1[r] = (float)(__bf16)(acc[it][1][r] * (1.0f - h1 * h1));
                colsum0 += v0[r]; colsum1 += v1[r];
            }
            store_packed(A.dz2p, row0 / 32 + it, 16, ft0, lane, v0); store_packed(A.dz2p, row0 / 32 + it, 16, ft0 + 1, lane, v1);
            store_lds_pair(dact, ACT_LD, 32 * it, 64 * w, lane, v0, v1);
            __builtin_amdgcn_sched_barrier(0);
        }
        colsum0 += __shfl_xor(colsum0, 32, 64); colsum1 += __shfl_xor(colsum1, 32, 64);
        if (h == 0) { bs[CDA_MLP_FEAT + f0 + 2 * j] = colsum0; bs[CDA_MLP_FEAT + f0 + 2 * j + 1] = colsum1; }
    }
    __syncthreads();
    {   // MdH1, then E1
        bf16x8 hp[MT][2][2];
        #pragma unroll
        for (int it = 0; it < MT; it++) { load_packed(A.h1p, row0 / 32 + it, 16, ft0, lane, hp[it][0]); load_packed(A.h1p, row0 / 32 + it, 16, ft0 + 1, lane, hp[it][1]); }
        f32x16 acc[MT][2];
        #pragma unroll
        for (int it = 0; it < MT; it++) { acc[it][0] = zero16(); acc[it][1] = zero16(); }
        layer_mma(dact, ACT_LD, R2, lane, acc);
        __syncthreads();
        float colsum0 = 0.0f, colsum1 = 0.0f;
        #pragma unroll
        for (int it = 0; it < MT; it++) {
            float v0[16], v1[16];
            #pragma unroll
            for (int r = 0; r < 16; r++) {
                const float h0 = (float)hp[it][0][r >> 3][r & 7], h1 = (float)hp[it][1][r >> 3][r & 7];
                v0[r] = (float)(__bf16)(acc[it][0][r] * (1.0f - h0 * h0)); v1[r] = (float)(__bf16)(acc[it][1][r] * (1.0f - h1 * h1));
                colsum0 += v0[r]; colsum1 += v1[r];
            }
            store_packed(A.dz1p, row0 / 32 + it, 16, ft0, lane, v0); store_packed(A.dz1p, row0 / 32 + it, 16, ft0 + 1, lane, v1);
            __builtin_amdgcn_sched_barrier(0);
        }
        colsum0 += __shfl_xor(colsum0, 32, 64); colsum1 += __shfl_xor(colsum1, 32, 64);
        if (h == 0) { bs[f0 + 2 * j] = colsum0; bs[f0 + 2 * j + 1] = colsum1; }
    }
    if (half == 0) __syncthreads();                                             // stage 3: the policy half's matching barrier
}

// ---- update, steps 0 - 2 in ONE kernel: gather, forward, loss, back-propagation of a 64-row tile ---------------------------------------------
// What the separate kernels pass through HBM stays on the chip: the observation rows are gathered (by the epoch's permutation) straight from the
// rollout's f32 buffer and their packed image for the weight gradients is written from the LDS tile (k_prep_rows and its two images: gone);
// the outputs and their gradients live in LDS (k_ppo_loss_rec's launch and four 8-MB round trips: gone); h1 / h2 of a wave's own 64 features
// stay in its registers for tanh' (the backward's re-read of 134 MB: gone).  HBM sees: x (f32), the records, h1p / h2p / dz1p / dz2p / doutp /
// x_pk (what the weight gradients read) and the bias sums.
// The two networks are independent down to the loss (the policy's loss terms need the 24 policy outputs only, the value loss the value only):
// a workgroup = 4 waves = ONE network half of one tile (blockIdx.y), 71 KB of LDS - two workgroups share a CU, any two: one's MFMA phases run
// under the other's tanh / loss phases and its gather (two dependent trips to HBM for random 672-B rows, ~14 k cycles) under the other's
// arithmetic, with no barrier between them.  (Both halves in one 8-wave workgroup, the value half one stage behind: 121 us, every stage as long
// as its slower half, the gather exposed; persistent workgroups prefetching the next tile: 131 us - 40 more live registers or 89 spilled SGPRs.)
//     per workgroup:  gather  [x_pk]  M1  E1 | M2 | E2 | MH | L | MdH2+E | MdH1  E          ( | = workgroup barrier)
// L: one lane per row (wave 0), logits / value read from the LDS output tile, d_out written back in place (f32, for the bias sums) and as the
// bf16 A operand of MdH2.  A row's d_out has two owners (columns != 24: policy, 24: value): each workgroup writes its own lanes of doutp.
struct FbArgs {
    const float* obs; const long long* perm; long long n_rows, norm_rows;
    const __bf16* wb; const float* theta;
    const float* rec; const double* adv_stats; long long adv_count; int agents; float clip, vf_coef, ent_coef;
    int rec_stride;                  // floats between two rows' records (agents * 8 when a row's samples are all the row's agents; league: A * 8 with rec pointing at the trainable slot)
    float kl_coef, vf_clip;          // the KL penalty (coefficient x mean KL(rollout policy || current policy), exact per row) and the clamp of the squared value error (<= 0: off)
    const float* dist_old; const float* log_std_old;      // kl_coef != 0: the rollout policy's distribution per row, f32 [*][CDA_MLP_DIST_LD] (FwdArgs::dist; its log-stds ride in the row: log_std_old is not read)
    int sd_log_std;                  // the state-dependent log-std head trains (include/cda_mlp.h cda_ppo_extra)
    __bf16* x_pk; __bf16* h1p; __bf16* h2p; __bf16* dz1p; __bf16* dz2p; __bf16* doutp; float* bias_slab;
    float* out; float* d_out;        // optional f32 [rows][32] copies of the outputs and their gradients (tests, diagnostics)
    double* sums5;
    unsigned long long* dbg; int dbg_block;
    int exper;                       // CDA_MLP_TIMING builds only (tools/fb_wgrad_fusion_probe.py): bit 0 = leave out the h1p / dz2p stores (what a fused dW2 would not need),
                                     // bit 1 = 64 extra MFMAs per wave and tile fed from LDS after MdH2 (the arithmetic a fused dW2 = dz2^T h1 would add); results are then wrong
};
__device__ __forceinline__ void keep_packed(const float (&v)[16], bf16x8 (&k)[2]) {
    typedef __attribute__((ext_vector_type(8))) float f32x8;
    f32x8 f0, f1;
    #pragma unroll
    for (int r = 0; r < 8; r++) { f0[r] = v[r]; f1[r] = v[8 + r]; }
    k[0] = __builtin_convertvector(f0, bf16x8); k[1] = __builtin_convertvector(f1, bf16x8);
}
__device__ __forceinline__ void store_packed_keep(__bf16* __restrict__ base, long long rt, int nft, int ft, int lane, const float (&v)[16], bf16x8 (&k)[2]) {
    typedef __attribute__((ext_vector_type(8))) float f32x8;
    f32x8 f0, f1;
    #pragma unroll
    for (int r = 0; r < 8; r++) { f0[r] = v[r]; f1[r] = v[8 + r]; }
    k[0] = __builtin_convertvector(f0, bf16x8); k[1] = __builtin_convertvector(f1, bf16x8);
    bf16x8* dst = reinterpret_cast<bf16x8*>(base) + ((rt * nft + ft) * 2) * 64 + lane;
    dst[0] = k[0]; dst[64] = k[1];
}
// quad (four neighbouring lanes) exchanges: DPP quad_perm, one instruction each, no LDS
template <int CTRL> __device__ __forceinline__ float dpp_f(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false)); }
__device__ __forceinline__ float quad_sum(float x) { x += dpp_f<0xB1>(x); x += dpp_f<0x4E>(x); return x; }          // [1,0,3,2] then [2,3,0,1]
__device__ __forceinline__ float quad_max(float x) { x = fmaxf(x, dpp_f<0xB1>(x)); x = fmaxf(x, dpp_f<0x4E>(x)); return x; }
template <int Q> __device__ __forceinline__ float quad_bcast(float x) { return dpp_f<Q * 0x55>(x); }
template <int Q> __device__ __forceinline__ int quad_bcast_i(int x) { return __builtin_amdgcn_update_dpp(0, x, Q * 0x55, 0xf, 0xf, false); }
#ifdef CDA_MLP_TIMING
#define MLP_MARKH(i) do { if (A.dbg && tile_id == A.dbg_block && lane == 0) A.dbg[(4 * half + w) * 32 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define MLP_MARKH(i) do {} while (0)
#endif

__global__ __launch_bounds__(256, 2) void k_mlp_fb(FbArgs A) {     // (two workgroups per CU: 256 registers per wave)
    constexpr int MT = 2, M = 64, PF = CDA_MLP_PF8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16* xs = reinterpret_cast<__bf16*>(smem);                               // [M][XS_LD]; once layer 1 has read it, its bytes hold:
    float* outs = reinterpret_cast<float*>(smem);                               //   [M][OUTS_LD] f32: the outputs, then (in place) their gradients
    __bf16* dos = reinterpret_cast<__bf16*>(smem + M * OUTS_LD * 4);            //   [M][DO_LD] bf16: the gradients as MdH2's A operand
    static_assert(M * OUTS_LD * 4 + M * DO_LD * 2 <= FB_XS_BYTES && M * XS_LD * 2 <= FB_XS_BYTES, "the region holds the observation tile, then the output tiles");
    const int lane = (int)threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    // Workgroup -> (tile, half), XCD-aware: workgroup ids go round the eight XCDs, so ids 8 apart share an L2.  The two halves of a tile gather
    // the same rows and records: id = 16 g + 8 half + k is tile 8 g + k - its sibling is dispatched 8 ids later, on the same XCD, and finds them
    // in that L2 (half, tile as the grid's y, x: siblings a thousand ids apart, every row fetched twice - 132 MB instead of 82 per launch).
    const int wg = (int)blockIdx.x, half = (wg >> 3) & 1, tile_id = 8 * (wg >> 4) + (wg & 7);
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if ((long long)tile_id * M >= A.n_rows) return;                             // (the last group of eight may be short; whole workgroups leave)
    __bf16* act = reinterpret_cast<__bf16*>(smem + FB_XS_BYTES);                // [M][ACT_LD]: h1, h2, then dz2
    float* lps = reinterpret_cast<float*>(act + M * ACT_LD);                    // [M][LPS_LD] f32: a row's log-probabilities, indexed by the agents' actions
    float* recs = lps + M * LPS_LD;                                             // [M][agents][8]: the tile's sample records
    const long long row0 = (long long)tile_id * M, rows_end = A.n_rows;
    const int f0 = 256 * half + 64 * w, ft0 = f0 >> 5;
    const float b1_0 = A.theta[CDA_MLP_OFF_B1 + f0 + 2 * j] * TWO_LOG2E, b1_1 = A.theta[CDA_MLP_OFF_B1 + f0 + 2 * j + 1] * TWO_LOG2E;
    const float b2_0 = A.theta[CDA_MLP_OFF_B2 + f0 + 2 * j] * TWO_LOG2E, b2_1 = A.theta[CDA_MLP_OFF_B2 + f0 + 2 * j + 1] * TWO_LOG2E;
    const float bo = A.theta[CDA_MLP_OFF_BO + j];
    float* bs = A.bias_slab + (size_t)tile_id * CDA_MLP_BSLAB;
    const __bf16* W1b = A.wb + CDA_MLP_WB_W1; const __bf16* W2b = A.wb + CDA_MLP_WB_W2; const __bf16* Wob = A.wb + CDA_MLP_WB_WO;
    const __bf16* W2T = A.wb + CDA_MLP_WB_W2T; const __bf16* WoT = A.wb + CDA_MLP_WB_WOT;
    WRing<2, KX / 16, PF, true> R1; WRing<2, HID / 16, PF, true> R2; WRing<1, HID / 16, PF> RO;
    MLP_MARKH(0);
    {   // the tile's rows, gathered.  Their source rows first (one lane each, through LDS): with the index known every request below is
        // unconditional and independent - the compiler issues them back to back (behind a per-request `perm ? perm[i] : i` it waited for each
        // index, then for each row: 14 round trips in a row)
        long long* srow = reinterpret_cast<long long*>(lps);                    // (the log-probabilities' bytes: used by the loss, long after)
        if (threadIdx.x < M) {
            long long gr = row0 + (int)threadIdx.x; if (gr >= rows_end) gr = rows_end - 1;   // rows past the end repeat the last one (their loss terms are masked)
            srow[threadIdx.x] = A.perm ? A.perm[gr] : gr;
        }
        R1.prime(W1b + (size_t)f0 * KX, KX, lane);
        __syncthreads();
        constexpr int CH = KX / VW, N = M * CH, PER = (N + 255) / 256;          // n_hist 4: 44 chunks of 4 values per row (42 real + 2 of zeros): 11 per thread
        obsvec v[PER];
        #pragma unroll
        for (int u = 0; u < PER; u++) {
            const int c = (int)threadIdx.x + 256 * u, cc = c < N ? c : N - 1, r = cc / CH, q = cc - r * CH;
            v[u] = *reinterpret_cast<const obsvec*>(A.obs + srow[r] * OBS + (q < OBS / VW ? q : OBS / VW - 1) * VW);
        }
        const int rp = 2 * A.agents, rpd = rp + (A.dist_old ? CDA_MLP_DIST_LD / 4 : 0);    // the records: [agents][8] f32 per row = 2 16-B pieces per agent; then the row's old distribution (7 pieces)
        for (int c = (int)threadIdx.x; c < M * rpd; c += 256) {
            const int r = c / rpd, q = c - r * rpd;
            reinterpret_cast<float4*>(recs)[c] = q < rp ? reinterpret_cast<const float4*>(A.rec + srow[r] * A.rec_stride)[q]
                                                        : reinterpret_cast<const float4*>(A.dist_old + srow[r] * CDA_MLP_DIST_LD)[q - rp];
        }
        #pragma unroll
        for (int u = 0; u < PER; u++) {
            const int c = (int)threadIdx.x + 256 * u, r = c / CH, q = c - r * CH;
            if (c < N) obs_to_bf16(xs + r * XS_LD + q * VW, v[u], q >= OBS / VW);
        }
    }
    __syncthreads();                                                            // the observation tile is in LDS
    MLP_MARKH(1);
    if (half == 1) {                                                            // value half: the packed image of x for the weight gradients
        #pragma unroll
        for (int u = 0; u < MT * XT * 2 / 4; u++) {                             // 24 pieces (row tile, feature tile, k-step), 6 per wave
            const int pc = w + 4 * u, it = pc / (XT * 2), rem = pc - it * (XT * 2), ft = rem >> 1, ks = rem & 1;
            bf16x8 v;
            #pragma unroll
            for (int e = 0; e < 8; e++) v[e] = (32 * ft + j < KX) ? xs[(32 * it + rowmap(8 * ks + e, h)) * XS_LD + 32 * ft + j] : (__bf16)0.0f;
            reinterpret_cast<bf16x8*>(A.x_pk)[(((row0 / 32 + it) * XT + ft) * 2 + ks) * 64 + lane] = v;
        }
    }
    MLP_MARKH(2);
    bf16x8 k1[MT][2][2], k2[MT][2][2];                                          // h1, h2 of this wave's 64 features, as stored: tanh' comes from them
    f32x16 acc[MT][2];
    #pragma unroll
    for (int it = 0; it < MT; it++) { acc[it][0] = zero16(); acc[it][1] = zero16(); }
    layer_mma(xs, XS_LD, R1, lane, acc);                                        // M1
    MLP_MARKH(3);
    R2.prime(W2b + ((size_t)half * HID + 64 * w) * HID, HID, lane);
    #pragma unroll
    for (int it = 0; it < MT; it++) {                                           // E1
        float v0[16], v1[16];
        #pragma unroll
        for (int r = 0; r < 16; r++) { v0[r] = tanh_biased(acc[it][0][r], b1_0); v1[r] = tanh_biased(acc[it][1][r], b1_1); }
#ifdef CDA_MLP_TIMING
        if (A.exper & 1) { keep_packed(v0, k1[it][0]); keep_packed(v1, k1[it][1]); } else
#endif
        {
        store_packed_keep(A.h1p, row0 / 32 + it, 16, ft0, lane, v0, k1[it][0]);
        store_packed_keep(A.h1p, row0 / 32 + it, 16, ft0 + 1, lane, v1, k1[it][1]);
        }
        store_lds_pair(act, ACT_LD, 32 * it, 64 * w, lane, v0, v1);
        __builtin_amdgcn_sched_barrier(0);
    }
    MLP_MARKH(4);
    __syncthreads();
    MLP_MARKH(5);
    #pragma unroll
    for (int it = 0; it < MT; it++) { acc[it][0] = zero16(); acc[it][1] = zero16(); }
    layer_mma(act, ACT_LD, R2, lane, acc);                                      // M2
    MLP_MARKH(6);
    RO.prime(Wob + (size_t)half * NOUT * HID, HID, lane);
    __syncthreads();                                                            // every wave has read h1: h2 takes its place
    MLP_MARKH(7);
    #pragma unroll
    for (int it = 0; it < MT; it++) {                                           // E2
        float v0[16], v1[16];
        #pragma unroll
        for (int r = 0; r < 16; r++) { v0[r] = tanh_biased(acc[it][0][r], b2_0); v1[r] = tanh_biased(acc[it][1][r], b2_1); }
        store_packed_keep(A.h2p, row0 / 32 + it, 16, ft0, lane, v0, k2[it][0]);
        store_packed_keep(A.h2p, row0 / 32 + it, 16, ft0 + 1, lane, v1, k2[it][1]);
        store_lds_pair(act, ACT_LD, 32 * it, 64 * w, lane, v0, v1);
        __builtin_amdgcn_sched_barrier(0);
    }
    MLP_MARKH(8);
    __syncthreads();
    MLP_MARKH(9);
    const bool own_col = half == 0 ? j != N_LOGITS : j == N_LOGITS;             // policy: columns 0 .. 23 and the zero padding; value: column 24
    {   // MH: wave w owns row tile w (waves >= MT multiply a tile nobody reads); the outputs go to LDS (the observation tile is dead)
        f32x16 acc3[1][1]; acc3[0][0] = zero16();
        layer_mma(act + (w < MT ? 32 * w : 0) * ACT_LD, ACT_LD, RO, lane, acc3);
        if (w < MT) {
            #pragma unroll
            for (int r = 0; r < 16; r++) outs[(32 * w + rowmap(r, h)) * OUTS_LD + j] = own_col ? acc3[0][0][r] + bo : 0.0f;
        }
    }
    MLP_MARKH(10);
    WRing<2, NOUT / 16, PF, true> RT; WRing<2, HID / 16, PF, true> R2T;
    RT.prime(WoT + (size_t)f0 * NOUT, NOUT, lane);
    __syncthreads();                                                            // the outputs are in LDS
    MLP_MARKH(11);
    // ---- the loss of this half's outputs: lane = row, wave 0 --------------------------------------------------------------------------------
    const float invB = 1.0f / ((float)A.norm_rows * (float)A.agents);
    if (half == 0) {
        // the policy's loss terms, FOUR lanes per row (every wave: rows 16 w .. + 15): lane s of a row's quad holds the six outputs 6 s .. 6 s + 5
        // (heads: category 0 .. 8, price 9 .. 18, offset 19 .. 21, the two means 22, 23), the softmax reductions run over the quad (DPP), the
        // row's agents are dealt round the quad and each agent's terms broadcast back.  (One lane per row on one wave: ~1000 instructions,
        // 7.5 k cycles with the other three waves waiting.)
        const int row = 16 * w + (lane >> 2), sq = lane & 3, i0 = 6 * sq;
        const bool live = row0 + row < rows_end;
        const int rec_ld = A.agents * 8 + (A.dist_old ? CDA_MLP_DIST_LD : 0);
        const float* rr = recs + (size_t)row * rec_ld;
        float* orow = outs + row * OUTS_LD;
        __bf16* drow = dos + row * DO_LD;
        float adv_mean = 0.0f, adv_rstd = 1.0f;
        if (A.adv_stats) {
            const double m = A.adv_stats[0] / (double)A.adv_count, var = (A.adv_stats[1] - (double)A.adv_count * m * m) / (double)(A.adv_count - 1);
            adv_mean = (float)m; adv_rstd = 1.0f / ((float)sqrt(var > 0.0 ? var : 0.0) + 1e-8f);
        }
        float l[6], d[6], pr[6], lp[6]; int hid[6];
        #pragma unroll
        for (int k = 0; k < 6; k++) { l[k] = orow[i0 + k]; d[k] = 0.0f; const int i = i0 + k; hid[k] = i < N_CAT ? 0 : (i < N_CAT + N_PRICE ? 1 : (i < N_CAT + N_PRICE + N_OFF ? 2 : 3)); }
        const float NEG = -3.0e38f;
        float mx[3], sm[3], cc[3], hh[3];
        #pragma unroll
        for (int t = 0; t < 3; t++) {
            float m = NEG;
            #pragma unroll
            for (int k = 0; k < 6; k++) m = fmaxf(m, hid[k] == t ? l[k] : NEG);
            mx[t] = quad_max(m);
        }
        #pragma unroll
        for (int k = 0; k < 6; k++) pr[k] = hid[k] < 3 ? __expf(l[k] - (hid[k] == 0 ? mx[0] : (hid[k] == 1 ? mx[1] : mx[2]))) : 0.0f;
        #pragma unroll
        for (int t = 0; t < 3; t++) {
            float x = 0.0f;
            #pragma unroll
            for (int k = 0; k < 6; k++) x += hid[k] == t ? pr[k] : 0.0f;
            sm[t] = quad_sum(x);
        }
        float inv[3];
        #pragma unroll
        for (int t = 0; t < 3; t++) { cc[t] = mx[t] + __logf(sm[t]); inv[t] = 1.0f / sm[t]; }
        #pragma unroll
        for (int k = 0; k < 6; k++) {
            pr[k] *= hid[k] == 0 ? inv[0] : (hid[k] == 1 ? inv[1] : inv[2]);
            lp[k] = hid[k] < 3 ? l[k] - (hid[k] == 0 ? cc[0] : (hid[k] == 1 ? cc[1] : cc[2])) : 0.0f;
        }
        #pragma unroll
        for (int t = 0; t < 3; t++) {
            float x = 0.0f;
            #pragma unroll
            for (int k = 0; k < 6; k++) x -= hid[k] == t ? pr[k] * lp[k] : 0.0f;
            hh[t] = quad_sum(x);
        }
        const float mean0 = quad_bcast<3>(l[4]), mean1 = quad_bcast<3>(l[5]);   // outputs 22, 23: lane 3's last two
        // the row's two log-stds: the free vector + outputs 25, 26 (the state-dependent head's offsets; zero in a network without the head) - read before the
        // tile's columns 24 .. 31 are overwritten with their gradients below (same wave, program order)
        const float lo0 = orow[N_LOGITS + 1], lo1 = orow[N_LOGITS + 2];
        const float ls0 = A.theta[CDA_MLP_OFF_LS] + lo0, ls1 = A.theta[CDA_MLP_OFF_LS + 1] + lo1;
        const float is0 = __expf(-ls0), is1 = __expf(-ls1);
        const float HALF_LOG_2PI = 0.918938533204672742f;
        const float ent = hh[0] + hh[1] + hh[2] + 1.0f + 2.0f * HALF_LOG_2PI + ls0 + ls1;
        const float es = A.ent_coef * invB;
        float* lrow = lps + row * LPS_LD;                                       // the row's log-probabilities, indexed by the agents' actions
        #pragma unroll
        for (int k = 0; k < 6; k++) lrow[i0 + k < LPS_LD ? i0 + k : LPS_LD - 1] = lp[k];    // (22 real entries; the means' two slots land on the row's spare word)
        float pg = 0.0f, en = 0.0f, dls0 = 0.0f, dls1 = 0.0f, G = 0.0f, dm0 = 0.0f, dm1 = 0.0f;
        for (int a4 = 0; a4 < A.agents; a4 += 4) {                              // agent a4 + s on lane s of the quad
            const int a = a4 + sq;
            float g = 0.0f, gz0 = 0.0f, gz1 = 0.0f; int ac = -1, ap = -1, ao = -1;
            if (a < A.agents) {
                const float4 w0 = reinterpret_cast<const float4*>(rr)[2 * a], w1 = reinterpret_cast<const float4*>(rr)[2 * a + 1];
                ac = __float_as_int(w0.x); ap = __float_as_int(w0.y); ao = __float_as_int(w0.z);
                const float z0 = (w0.w - mean0) * is0, z1 = (w1.x - mean1) * is1;
                const float logp = -0.5f * z0 * z0 - ls0 - HALF_LOG_2PI - 0.5f * z1 * z1 - ls1 - HALF_LOG_2PI +
                                   lrow[min(max(ac, 0), N_CAT - 1)] + lrow[N_CAT + min(max(ap, 0), N_PRICE - 1)] + lrow[N_CAT + N_PRICE + min(max(ao, 0), N_OFF - 1)];
                const float Av = (w1.z - adv_mean) * adv_rstd, ratio = __expf(logp - w1.y);
                const float un = ratio * Av, cl = fminf(fmaxf(ratio, 1.0f - A.clip), 1.0f + A.clip) * Av;
                pg -= fminf(un, cl);
                g = (un <= cl) ? -un * invB : 0.0f;
                en += ent;
                gz0 = g * z0 * is0; gz1 = g * z1 * is1;
                dls0 += g * (z0 * z0 - 1.0f) - es;
                dls1 += g * (z1 * z1 - 1.0f) - es;
            }
            // agent Q of this group, broadcast: its three actions as slot numbers of THIS lane (head ranges are disjoint: at most one of the
            // three can name a slot; an absent agent adds g = 0)
            #define CDA_FB_AGENT(Q) { \
                const float gq = quad_bcast<Q>(g); \
                const int r0 = quad_bcast_i<Q>(ac) - i0, r1 = N_CAT + quad_bcast_i<Q>(ap) - i0, r2 = N_CAT + N_PRICE + quad_bcast_i<Q>(ao) - i0; \
                G += gq; dm0 += quad_bcast<Q>(gz0); dm1 += quad_bcast<Q>(gz1); \
                _Pragma("unroll") for (int k = 0; k < 6; k++) d[k] += ((r0 == k) | (r1 == k) | (r2 == k)) ? gq : 0.0f; }
            CDA_FB_AGENT(0) CDA_FB_AGENT(1) CDA_FB_AGENT(2) CDA_FB_AGENT(3)
            #undef CDA_FB_AGENT
        }
        const float esA = es * (float)A.agents;
        #pragma unroll
        for (int k = 0; k < 6; k++) {
            const float hk = hid[k] == 0 ? hh[0] : (hid[k] == 1 ? hh[1] : hh[2]);
            d[k] += -G * pr[k] + esA * pr[k] * (lp[k] + hk);                    // (the means' slots: pr = 0)
        }
        float klsum = 0.0f;
        if (A.dist_old) {
            // KL(rollout policy || current policy) of the row, exact: the categorical heads from the old normalised log-probabilities, the Gaussian
            // heads in closed form; every sample of the row carries it (the row's agents share the distribution), so the row's weight is `agents`
            const float* od = rr + A.agents * 8 + i0;
            const float klw = A.kl_coef * (float)A.agents * invB;
            float kc = 0.0f, om[6];
            #pragma unroll
            for (int k = 0; k < 6; k++) {
                om[k] = od[k];
                const float po = hid[k] < 3 ? __expf(om[k]) : 0.0f;
                kc += hid[k] < 3 ? po * (om[k] - lp[k]) : 0.0f;
                d[k] += hid[k] < 3 ? klw * (pr[k] - po) : 0.0f;
            }
            kc = quad_sum(kc);
            const float mo0 = quad_bcast<3>(om[4]), mo1 = quad_bcast<3>(om[5]);
            const float lso0 = rr[A.agents * 8 + N_LOGITS], lso1 = rr[A.agents * 8 + N_LOGITS + 1];     // the log-stds the row was sampled with
            const float q0 = (__expf(2.0f * lso0) + (mo0 - mean0) * (mo0 - mean0)) * is0 * is0, q1 = (__expf(2.0f * lso1) + (mo1 - mean1) * (mo1 - mean1)) * is1 * is1;
            const float kg = (ls0 - lso0) + 0.5f * q0 - 0.5f + (ls1 - lso1) + 0.5f * q1 - 0.5f;
            dm0 += klw * (mean0 - mo0) * is0 * is0; dm1 += klw * (mean1 - mo1) * is1 * is1;
            if (sq == 0) { dls0 += klw * (1.0f - q0); dls1 += klw * (1.0f - q1); klsum = (kc + kg) * (float)A.agents; }
        }
        if (sq == 3) { d[4] = dm0; d[5] = dm1; }
        // columns 24 .. 31 of the row, two per lane of the quad: 24 is the value workgroup's (zero here), 25 / 26 carry d loss / d log_std of THIS row when the
        // state-dependent head trains (the row's agents' terms summed over the quad; the free vector's sums are then zero), 27 .. 31 zero
        float e0 = 0.0f, e1 = 0.0f;
        if (A.sd_log_std) {
            const float D0 = quad_sum(dls0), D1 = quad_sum(dls1);
            if (sq == 0) e1 = D0;
            if (sq == 1) e0 = D1;
            dls0 = dls1 = 0.0f;
        }
        if (A.out && live) {
            #pragma unroll
            for (int k = 0; k < 6; k++) A.out[(row0 + row) * NOUT + i0 + k] = l[k];
            if (sq != 0) A.out[(row0 + row) * NOUT + N_LOGITS + 2 * sq] = sq == 1 ? lo1 : 0.0f;
            A.out[(row0 + row) * NOUT + N_LOGITS + 2 * sq + 1] = sq == 0 ? lo0 : 0.0f;
        }
        if (!live) {
            pg = en = dls0 = dls1 = klsum = e0 = e1 = 0.0f;
            #pragma unroll
            for (int k = 0; k < 6; k++) d[k] = 0.0f;
        }
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        typedef __attribute__((ext_vector_type(2))) float f32x2;
        #pragma unroll
        for (int k = 0; k < 6; k += 2) {
            orow[i0 + k] = d[k]; orow[i0 + k + 1] = d[k + 1];
            f32x2 f; f[0] = d[k]; f[1] = d[k + 1];
            *reinterpret_cast<bf16x2*>(drow + i0 + k) = __builtin_convertvector(f, bf16x2);
        }
        orow[N_LOGITS + 2 * sq] = e0; orow[N_LOGITS + 2 * sq + 1] = e1;         // columns 24 .. 31 (24 is the value workgroup's, in its own tile)
        { f32x2 f; f[0] = e0; f[1] = e1; *reinterpret_cast<bf16x2*>(drow + N_LOGITS + 2 * sq) = __builtin_convertvector(f, bf16x2); }
        if (A.d_out && live) {
            #pragma unroll
            for (int k = 0; k < 6; k++) A.d_out[(row0 + row) * NOUT + i0 + k] = d[k];
            if (sq != 0) A.d_out[(row0 + row) * NOUT + N_LOGITS + 2 * sq] = e0;
            A.d_out[(row0 + row) * NOUT + N_LOGITS + 2 * sq + 1] = e1;
        }
        float v4[5] = {pg, en, dls0, dls1, klsum};
        #pragma unroll
        for (int q = 0; q < 5; q++) {
            float x = v4[q];
            #pragma unroll
            for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
            v4[q] = x;
        }
        double* slot = A.sums5 + 8 * ((4 * tile_id + w) & (CDA_MLP_LOSS_SLOTS - 1));   // a slot (one cache line) per wave mod 64: one hot line stalls every CU's memory pipeline behind its atomics
        if (lane == 0) {
            atomicAdd(&slot[0], (double)v4[0]); atomicAdd(&slot[2], (double)v4[1]); atomicAdd(&slot[3], (double)v4[2]); atomicAdd(&slot[4], (double)v4[3]);
            if (A.dist_old) atomicAdd(&slot[5], (double)v4[4]);
        }
    } else if (w == 0) {                                                        // the value loss: a lane per row
        const int row = lane;
        const bool live = row0 + row < rows_end;
        const float* rr = recs + (size_t)row * (A.agents * 8 + (A.dist_old ? CDA_MLP_DIST_LD : 0));
        float* orow = outs + row * OUTS_LD;
        __bf16* drow = dos + row * DO_LD;
        const float val = orow[N_LOGITS];
        float vl = 0.0f, dval = 0.0f;
        for (int a = 0; a < A.agents; a++) {
            const float dv = val - rr[8 * a + 7], sqe = dv * dv;
            const bool clamped = A.vf_clip > 0.0f && sqe > A.vf_clip;          // clamp(error^2, 0, vf_clip): the clamped samples carry no gradient
            vl += clamped ? A.vf_clip : sqe;
            dval += clamped ? 0.0f : 2.0f * A.vf_coef * dv * invB;
        }
        if (A.out && live) A.out[(row0 + row) * NOUT + N_LOGITS] = val;
        if (!live) { vl = 0.0f; dval = 0.0f; }
        #pragma unroll
        for (int q = 0; q < NOUT; q++) { orow[q] = q == N_LOGITS ? dval : 0.0f; drow[q] = (__bf16)(q == N_LOGITS ? dval : 0.0f); }
        if (A.d_out && live) A.d_out[(row0 + row) * NOUT + N_LOGITS] = dval;
        #pragma unroll
        for (int o = 32; o > 0; o >>= 1) vl += __shfl_down(vl, o, 64);
        if (lane == 0) atomicAdd(&A.sums5[8 * (tile_id & (CDA_MLP_LOSS_SLOTS - 1)) + 1], (double)vl);
    }
    MLP_MARKH(12);
    __syncthreads();                                                            // d_out is in LDS
    MLP_MARKH(13);
    if (w >= MT) {                                                              // d_out's packed image and column sums: this half's columns (waves 2, 3: a row tile each)
        const int it = w - MT;
        if (own_col) {
            bf16x8 v0, v1;
            #pragma unroll
            for (int r = 0; r < 8; r++) { v0[r] = dos[(32 * it + rowmap(r, h)) * DO_LD + j]; v1[r] = dos[(32 * it + rowmap(8 + r, h)) * DO_LD + j]; }
            bf16x8* dst = reinterpret_cast<bf16x8*>(A.doutp) + ((row0 / 32 + it) * 2) * 64 + lane;
            dst[0] = v0; dst[64] = v1;
        }
        float s4[4] = {0.0f, 0.0f, 0.0f, 0.0f};                                 // lane (column j, row half h) of wave 2 + it: 16 rows each, four running sums
        #pragma unroll
        for (int r = 0; r < 16; r++) s4[r & 3] += outs[(32 * it + 16 * h + r) * OUTS_LD + j];
        float sum = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        sum += __shfl_xor(sum, 32, 64);
        float* part = reinterpret_cast<float*>(lps);                            // (the log-probabilities are dead)
        if (h == 0) part[32 * it + j] = sum;
    }
    {   // MdH2 + E: dz2 = (d_out x Wo) tanh'(h2)
        #pragma unroll
        for (int it = 0; it < MT; it++) { acc[it][0] = zero16(); acc[it][1] = zero16(); }
        layer_mma(dos, DO_LD, RT, lane, acc);
        MLP_MARKH(21);
        R2T.prime(W2T + ((size_t)half * HID + 64 * w) * HID, HID, lane);
        float colsum0 = 0.0f, colsum1 = 0.0f;
        #pragma unroll
        for (int it = 0; it < MT; it++) {
            float v0[16], v1[16];
            #pragma unroll
            for (int r = 0; r < 16; r++) {
                const float h0 = (float)k2[it][0][r >> 3][r & 7], h1 = (float)k2[it][1][r >> 3][r & 7];
                v0[r] = (float)(__bf16)(acc[it][0][r] * (1.0f - h0 * h0)); v1[r] = (float)(__bf16)(acc[it][1][r] * (1.0f - h1 * h1));
                colsum0 += v0[r]; colsum1 += v1[r];
            }
#ifdef CDA_MLP_TIMING
            if (!(A.exper & 1))
#endif
            { store_packed(A.dz2p, row0 / 32 + it, 16, ft0, lane, v0); store_packed(A.dz2p, row0 / 32 + it, 16, ft0 + 1, lane, v1); }
            store_lds_pair(act, ACT_LD, 32 * it, 64 * w, lane, v0, v1);         // (h2's image was last read by MH, two barriers ago)
            __builtin_amdgcn_sched_barrier(0);
        }
        colsum0 += __shfl_xor(colsum0, 32, 64); colsum1 += __shfl_xor(colsum1, 32, 64);
        if (h == 0) { bs[CDA_MLP_FEAT + f0 + 2 * j] = colsum0; bs[CDA_MLP_FEAT + f0 + 2 * j + 1] = colsum1; }
#ifdef CDA_MLP_TIMING
        if (A.exper & 2) {
            // the arithmetic of a fused dW2: this wave's 64 dz2 features (A operand: the accumulators' own layout, here k2's registers stand in) against all 256
            // features of h1 read from LDS in operand form (16 B per lane, contiguous: the xs / act bytes stand in): 2 x 8 tiles x 4 k-steps = 64 MFMAs
            f32x16 extra[4];
            #pragma unroll
            for (int q = 0; q < 4; q++) extra[q] = zero16();
            const bf16x8* lb = reinterpret_cast<const bf16x8*>(smem) + lane;
            #pragma unroll
            for (int u = 0; u < 16; u++) {
                const bf16x8 b0 = lb[64 * (2 * u)], b1 = lb[64 * (2 * u + 1)];
                #pragma unroll
                for (int q = 0; q < 4; q++) extra[q] = mfma(k2[q >> 1][q & 1][u & 1], (q & 1) ? b1 : b0, extra[q]);
            }
            float keep = 0.0f;
            #pragma unroll
            for (int q = 0; q < 4; q++) keep += extra[q][0] + extra[q][15];
            if (keep == 123.456f) bs[0] = keep;                                  // (keeps the products alive)
        }
#endif
    }
    MLP_MARKH(14);
    __syncthreads();
    MLP_MARKH(15);
    if (w == 0 && h == 0 && own_col) {                                          // d_out's column sums: the two row tiles' shares
        const float* part = reinterpret_cast<const float*>(lps);
        bs[2 * CDA_MLP_FEAT + j] = part[j] + part[32 + j];
    }
    {   // MdH1, then E: dz1 = (dz2 x W2) tanh'(h1)
        #pragma unroll
        for (int it = 0; it < MT; it++) { acc[it][0] = zero16(); acc[it][1] = zero16(); }
        layer_mma(act, ACT_LD, R2T, lane, acc);
        MLP_MARKH(16);
        float colsum0 = 0.0f, colsum1 = 0.0f;
        #pragma unroll
        for (int it = 0; it < MT; it++) {
            float v0[16], v1[16];
            #pragma unroll
            for (int r = 0; r < 16; r++) {
                const float h0 = (float)k1[it][0][r >> 3][r & 7], h1 = (float)k1[it][1][r >> 3][r & 7];
                v0[r] = (float)(__bf16)(acc[it][0][r] * (1.0f - h0 * h0)); v1[r] = (float)(__bf16)(acc[it][1][r] * (1.0f - h1 * h1));
                colsum0 += v0[r]; colsum1 += v1[r];
            }
            store_packed(A.dz1p, row0 / 32 + it, 16, ft0, lane, v0); store_packed(A.dz1p, row0 / 32 + it, 16, ft0 + 1, lane, v1);
            __builtin_amdgcn_sched_barrier(0);
        }
        colsum0 += __shfl_xor(colsum0, 32, 64); colsum1 += __shfl_xor(colsum1, 32, 64);
        if (h == 0) { bs[f0 + 2 * j] = colsum0; bs[f0 + 2 * j + 1] = colsum1; }
    }
    MLP_MARKH(17);
}

// ---- update, step 3: weight gradients ------------------------------------------------------------------------------------------
// dW[i][j] = sum over rows of dz[row][i] h[row][j]: both operands come out of HBM in the packed layout, 16 B per lane, straight into
// the MFMA (no LDS).  One workgroup = one JOB (an output panel) x one row chunk; the chunk's partial sum goes to the slab.
//   job 0, 1: dW2[b]   256 x 256  = dz2[:, b] ^T h1[:, b]     wave (wi, wj): 128 x 128
//   then, per network half b, WG_XG jobs of dW1[256 b ..]  256 x 32 XT = dz1[:, b]^T x:  wave (wi, wj): 128 x 32 WG_TJ   (n_hist 4: jobs 2, 3: 256 x 192, 128 x 96 per wave)
//   last job: dWo      32 x 512   = d_out^T h2                 wave w: 32 x 128
template <int TI, int TJ>
__device__ __forceinline__ void wgrad_wave(const bf16x8* __restrict__ Ap, int a_nft, int a_ft0, const bf16x8* __restrict__ Bp, int b_nft, int b_ft0,
                                           long long rt0, long long rt1, int lane, f32x16 (&acc)[TI][TJ]) {
    // (a B tile past the operand's last one - the x panel's tile count need not be a multiple of a wave's share - repeats the last real tile: loaded, multiplied, never stored)
    // a step s = (row tile, k-step of 16 rows); operands of step s + D - 1 are requested before step s multiplies (ring of D register
    // sets, statically indexed; requests past the end repeat the last step and are never multiplied)
    constexpr int D = 4;
    bf16x8 a[D][TI], b[D][TJ];
    const long long s0 = rt0 * 2, s1 = rt1 * 2;
    auto request = [&](auto slot_c, long long s) {
        constexpr int slot = decltype(slot_c)::value;
        const long long sc = s < s1 ? s : s1 - 1, rt = sc >> 1; const int ks = (int)(sc & 1);
#ifdef CDA_WGRAD_NT              /* experiment (tools/wgrad_nt_probe.sh): the operands are read once - non-temporal requests */
        #pragma unroll
        for (int ti = 0; ti < TI; ti++) a[slot][ti] = __builtin_nontemporal_load(&Ap[((rt * a_nft + a_ft0 + ti) * 2 + ks) * 64 + lane]);
        #pragma unroll
        for (int tj = 0; tj < TJ; tj++) { const int ft = b_ft0 + tj < b_nft ? b_ft0 + tj : b_nft - 1; b[slot][tj] = __builtin_nontemporal_load(&Bp[((rt * b_nft + ft) * 2 + ks) * 64 + lane]); }
#else
        #pragma unroll
        for (int ti = 0; ti < TI; ti++) a[slot][ti] = Ap[((rt * a_nft + a_ft0 + ti) * 2 + ks) * 64 + lane];
        #pragma unroll
        for (int tj = 0; tj < TJ; tj++) { const int ft = b_ft0 + tj < b_nft ? b_ft0 + tj : b_nft - 1; b[slot][tj] = Bp[((rt * b_nft + ft) * 2 + ks) * 64 + lane]; }
#endif
    };
    request(std::integral_constant<int, 0>{}, s0); request(std::integral_constant<int, 1>{}, s0 + 1); request(std::integral_constant<int, 2>{}, s0 + 2);
    __builtin_amdgcn_sched_barrier(0);
    #define CDA_WG_STEP(d, nxt) \
        request(std::integral_constant<int, nxt>{}, s + d + D - 1); \
        __builtin_amdgcn_sched_barrier(0); \
        if (s + d < s1) { \
            _Pragma("unroll") for (int ti = 0; ti < TI; ti++) \
                _Pragma("unroll") for (int tj = 0; tj < TJ; tj++) acc[ti][tj] = mfma(a[d][ti], b[d][tj], acc[ti][tj]); \
        }
    #pragma unroll 1
    for (long long s = s0; s < s1; s += D) {
        CDA_WG_STEP(0, 3) CDA_WG_STEP(1, 0) CDA_WG_STEP(2, 1) CDA_WG_STEP(3, 2)
    }
    #undef CDA_WG_STEP
}
// A_PAIRED / B_PAIRED: the operand's feature tiles are the paired ones of the hidden activations (feature_of); a_ft0 / b_ft0: first tile of
// this wave inside the panel (i0 / j0 of the panel itself are folded into dst)
template <int TI, int TJ, bool A_PAIRED, bool B_PAIRED>
__device__ __forceinline__ void wgrad_store(float* __restrict__ dst, int ld, int a_ft0, int b_ft0, int lane, const f32x16 (&acc)[TI][TJ], int b_nft = 1 << 30) {
    const int j = lane & 31, h = lane >> 5;
    #pragma unroll
    for (int ti = 0; ti < TI; ti++)
        #pragma unroll
        for (int tj = 0; tj < TJ; tj++) {
            if (b_ft0 + tj >= b_nft) continue;                                   // a dummy tile (see wgrad_wave)
            const int col = B_PAIRED ? feature_of(b_ft0 + tj, j) : 32 * (b_ft0 + tj) + j;
            #pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = A_PAIRED ? feature_of(a_ft0 + ti, rowmap(r, h)) : 32 * (a_ft0 + ti) + rowmap(r, h);
                dst[(size_t)row * ld + col] = acc[ti][tj][r];
            }
        }
}
constexpr int WG_TJ = XT <= 2 ? 1 : (XT <= 4 ? 2 : 3), WG_XG = (XT + 2 * WG_TJ - 1) / (2 * WG_TJ), WG_JOBS = 3 + 2 * WG_XG;     // x tiles per wave, groups per half, jobs (n_hist 4: 3, 1, 5)
struct WgradArgs { const bf16x8* x_pk; const bf16x8* h1p; const bf16x8* h2p; const bf16x8* dz1p; const bf16x8* dz2p; const bf16x8* doutp;
                   long long n_rt; int n_chunks; float* slab; int first_job; };
__global__ __launch_bounds__(256) void k_mlp_wgrad(WgradArgs A) {
    const int lane = (int)threadIdx.x & 63, w = (int)threadIdx.x >> 6, wi = w >> 1, wj = w & 1;
    const int job = (int)blockIdx.y + A.first_job, chunk = (int)blockIdx.x;
    const long long rt0 = A.n_rt * chunk / A.n_chunks, rt1 = A.n_rt * (chunk + 1) / A.n_chunks;
    float* slab = A.slab + (size_t)chunk * CDA_MLP_SLAB;
    if (job < 2) {
        f32x16 acc[4][4];
        #pragma unroll
        for (int a = 0; a < 4; a++)
            #pragma unroll
            for (int b = 0; b < 4; b++) acc[a][b] = zero16();
        wgrad_wave<4, 4>(A.dz2p, 16, 8 * job + 4 * wi, A.h1p, 16, 8 * job + 4 * wj, rt0, rt1, lane, acc);
        wgrad_store<4, 4, true, true>(slab + CDA_MLP_SLAB_W2 + (size_t)job * HID * HID, HID, 4 * wi, 4 * wj, lane, acc);
    } else if (job < 2 + 2 * WG_XG) {
        // dW1: per network half, the x panel's XT tiles in WG_XG groups of 2 WG_TJ (a wave: 128 features x WG_TJ tiles; n_hist 4: one group, 128 x 96 per wave)
        const int b = (job - 2) / WG_XG, xg = (job - 2) - b * WG_XG, x0 = 2 * WG_TJ * xg + WG_TJ * wj;
        f32x16 acc[4][WG_TJ];
        #pragma unroll
        for (int a = 0; a < 4; a++)
            #pragma unroll
            for (int c = 0; c < WG_TJ; c++) acc[a][c] = zero16();
        wgrad_wave<4, WG_TJ>(A.dz1p, 16, 8 * b + 4 * wi, A.x_pk, XT, x0, rt0, rt1, lane, acc);
        wgrad_store<4, WG_TJ, true, false>(slab + CDA_MLP_SLAB_W1 + (size_t)(256 * b) * (32 * XT), 32 * XT, 4 * wi, x0, lane, acc, XT);
    } else {
        f32x16 acc[1][4];
        #pragma unroll
        for (int c = 0; c < 4; c++) acc[0][c] = zero16();
        wgrad_wave<1, 4>(A.doutp, 1, 0, A.h2p, 16, 4 * w, rt0, rt1, lane, acc);
        wgrad_store<1, 4, false, true>(slab + CDA_MLP_SLAB_WO, CDA_MLP_FEAT, 0, 4 * w, lane, acc);
    }
}

// ---- update, step 4: reduce, clip, Adam, repack --------------------------------------------------------------------------------
// The partial sums -> the gradient of theta and its squared norm.  Weights: one thread per FOUR consecutive entries of the dense slab
// (16-B loads, a wave reads 1 KB per chunk, eight chunks in flight), mapped back to the parameter index (the slab's paddings and the heads'
// masked blocks have none).  Biases (+ log_std): 64 entries per block, the row tiles split over 16 threads each.
constexpr int RED_DENSE_BLOCKS = CDA_MLP_SLAB / 4 / 256;                          // 240
constexpr int NORM_PARTIALS = 8;                                                 // scratch[8 + block]: the blocks' shares of the squared gradient norm
constexpr int RED_BIAS_BLOCKS = (CDA_MLP_BSLAB + 1 + 15) / 16;                    // 67 (entry 1056 = the log_std pair)
__device__ __forceinline__ int param_of_dense(int d) {
    if (d < CDA_MLP_SLAB_W2) { const int o = d / (32 * XT), i = d - o * (32 * XT); return i < OBS ? CDA_MLP_OFF_W1 + o * OBS + i : -1; }
    if (d < CDA_MLP_SLAB_WO) return CDA_MLP_OFF_W2 + (d - CDA_MLP_SLAB_W2);
    const int q = d - CDA_MLP_SLAB_WO, o = q / CDA_MLP_FEAT, c = q - o * CDA_MLP_FEAT;
    if (o < N_LOGITS || o == N_LOGITS + 1 || o == N_LOGITS + 2) return c < HID ? CDA_MLP_OFF_WO + o * HID + c : -1;     // (25, 26: the log-std head's rows - zero gradient unless it trains)
    if (o == N_LOGITS) return c >= HID ? CDA_MLP_OFF_WO + N_LOGITS * HID + (c - HID) : -1;
    return -1;
}
struct LossFinish { double* sums5; long long samples; float vf_coef, ent_coef, kl_coef; float* out6; };      // out6: f32[8] (CDA_LOSS_OUT_*)
__global__ __launch_bounds__(256) void k_grad_reduce(const float* __restrict__ slab, int n_chunks, const float* __restrict__ bslab, int n_tiles,
                                                     LossFinish LF, float* __restrict__ grad, double* __restrict__ norm2, float* __restrict__ step) {
    __shared__ float red[16][64];
    float sq = 0.0f;
    if ((int)blockIdx.x < RED_DENSE_BLOCKS) {
        const int d = 4 * ((int)blockIdx.x * 256 + (int)threadIdx.x);
        const float4* src = reinterpret_cast<const float4*>(slab + d);
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
        int c = 0;
        for (; c + 8 <= n_chunks; c += 8) {
            float4 v[8];
            #pragma unroll
            for (int u = 0; u < 8; u++) v[u] = src[(size_t)(c + u) * (CDA_MLP_SLAB / 4)];
            #pragma unroll
            for (int u = 0; u < 8; u += 4) {
                s0.x += v[u].x; s0.y += v[u].y; s0.z += v[u].z; s0.w += v[u].w;
                s1.x += v[u + 1].x; s1.y += v[u + 1].y; s1.z += v[u + 1].z; s1.w += v[u + 1].w;
                s2.x += v[u + 2].x; s2.y += v[u + 2].y; s2.z += v[u + 2].z; s2.w += v[u + 2].w;
                s3.x += v[u + 3].x; s3.y += v[u + 3].y; s3.z += v[u + 3].z; s3.w += v[u + 3].w;
            }
        }
        for (; c < n_chunks; c++) { const float4 v = src[(size_t)c * (CDA_MLP_SLAB / 4)]; s0.x += v.x; s0.y += v.y; s0.z += v.z; s0.w += v.w; }
        const float g[4] = {(s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w)};
        if (OBS % 4 == 0) {
            const int p = param_of_dense(d);                                      // (groups of four never straddle a row's end: 84, 168, 336 and 192, 256, 512 are multiples of 4)
            if (p >= 0) {
                #pragma unroll
                for (int e = 0; e < 4; e++) { grad[p + e] = g[e]; sq += g[e] * g[e]; }
            }
        } else {                                                                  // an odd history depth (42 H inputs): a group may end in W1's zero padding - entry by entry
            #pragma unroll
            for (int e = 0; e < 4; e++) { const int p = param_of_dense(d + e); if (p >= 0) { grad[p] = g[e]; sq += g[e] * g[e]; } }
        }
    } else {
        // 16 entries of the bias slab per block, the row tiles split 16 ways (a serial walk over hundreds of 4-KB strided partials by a
        // handful of threads was latency bound: 60 us)
        const int e = ((int)blockIdx.x - RED_DENSE_BLOCKS) * 16 + ((int)threadIdx.x & 15), part = (int)threadIdx.x >> 4;
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
        if (e < CDA_MLP_BSLAB) {
            const float* src = bslab + e;
            int t = part;
            for (; t + 48 < n_tiles; t += 64) {
                s0 += src[(size_t)t * CDA_MLP_BSLAB]; s1 += src[(size_t)(t + 16) * CDA_MLP_BSLAB]; s2 += src[(size_t)(t + 32) * CDA_MLP_BSLAB]; s3 += src[(size_t)(t + 48) * CDA_MLP_BSLAB];
            }
            for (; t < n_tiles; t += 16) s0 += src[(size_t)t * CDA_MLP_BSLAB];
        }
        red[part][threadIdx.x & 15] = (s0 + s1) + (s2 + s3);
        __shared__ double lred[16][6];                                           // the loss sums' slots, split the same 16 ways (one thread walking all 64: +5 us on the kernel)
        if (e == CDA_MLP_BSLAB && LF.sums5) {
            double t[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            for (int sl = part; sl < CDA_MLP_LOSS_SLOTS; sl += 16)
                #pragma unroll
                for (int q = 0; q < 6; q++) { t[q] += LF.sums5[8 * sl + q]; LF.sums5[8 * sl + q] = 0.0; }
            #pragma unroll
            for (int q = 0; q < 6; q++) lred[part][q] = t[q];
        }
        __syncthreads();
        if (part == 0) {
            float g = 0.0f;
            #pragma unroll
            for (int q = 0; q < 16; q++) g += red[q][threadIdx.x];
            int p = -1;
            if (e < CDA_MLP_FEAT) p = CDA_MLP_OFF_B1 + e;
            else if (e < 2 * CDA_MLP_FEAT) p = CDA_MLP_OFF_B2 + (e - CDA_MLP_FEAT);
            else if (e < CDA_MLP_BSLAB) { const int o = e - 2 * CDA_MLP_FEAT; p = CDA_MLP_OFF_BO + o; if (o > N_LOGITS + 2) g = 0.0f; }
            if (p >= 0) { grad[p] = g; sq = g * g; }
            if (e == CDA_MLP_BSLAB) {
                // log_std: its gradient comes with the loss sums (words 3, 4); this one thread also finishes the loss statistics (out6, what
                // k_ppo_finish32 would write); the sums are cleared for the next minibatch above - no memset, no extra launch
                float g0 = 0.0f, g1 = 0.0f;
                if (LF.sums5) {
                    double t5[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
                    for (int k = 0; k < 16; k++)
                        #pragma unroll
                        for (int q = 0; q < 6; q++) t5[q] += lred[k][q];
                    const double pg = t5[0] / (double)LF.samples, vl = t5[1] / (double)LF.samples, en = t5[2] / (double)LF.samples, kl = t5[5] / (double)LF.samples;
                    g0 = (float)t5[3]; g1 = (float)t5[4];
                    if (LF.out6) { LF.out6[0] = (float)pg; LF.out6[1] = (float)vl; LF.out6[2] = (float)en;
                                   LF.out6[3] = (float)(pg + (double)LF.vf_coef * vl - (double)LF.ent_coef * en + (double)LF.kl_coef * kl);
                                   LF.out6[4] = g0; LF.out6[5] = g1; LF.out6[6] = (float)kl; LF.out6[7] = 0.0f; }
                }
                grad[CDA_MLP_OFF_LS] = g0; grad[CDA_MLP_OFF_LS + 1] = g1; sq = g0 * g0 + g1 * g1;
            }
        }
    }
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_down(sq, o, 64);
    __shared__ float wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = sq;
    __syncthreads();
    // this block's share of the squared norm: a plain store (no atomic accumulator to clear between calls); k_adam sums the shares
    if (threadIdx.x == 0) norm2[NORM_PARTIALS + blockIdx.x] = (double)wsum[0] + (double)wsum[1] + (double)wsum[2] + (double)wsum[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) step[0] += 1.0f;                   // nobody reads it in this launch; k_adam (next on the stream) sees t
}
// element index inside a group's operand image: column tile jt, k-step ks, lane (position q in the tile, k-half hh), slot e
__device__ __forceinline__ size_t op_index(int ksteps, int jt, int k, int q) { return ((size_t)(jt * ksteps + (k >> 4)) * 64 + q + 32 * ((k & 15) >> 3)) * 8 + (k & 7); }
// one parameter -> its places in the bf16 operand blob (operand order, see WRing).  A PAIRED group = 64 consecutive output features of one half:
// feature 64 w + 2 q + jt sits in column tile jt at position q.
__device__ __forceinline__ void pack_one(int p, float v, __bf16* __restrict__ wb) {
    const __bf16 b = (__bf16)v;
    if (p < CDA_MLP_OFF_B1) {                                                    // W1[o][i] -> W1p: group o / 64 (8 groups of 64 x 176)
        const int o = p / OBS, i = p - o * OBS, g = o >> 6, f = o & 63;
        wb[CDA_MLP_WB_W1 + (size_t)g * 64 * KX + op_index(KX / 16, f & 1, i, f >> 1)] = b;
    } else if (p >= CDA_MLP_OFF_W2 && p < CDA_MLP_OFF_B2) {                      // W2[blk][o][i] -> W2p (rows = outputs) and W2Tp (rows = inputs, k = outputs)
        const int q = p - CDA_MLP_OFF_W2, blk = q / (HID * HID), o = (q / HID) % HID, i = q % HID;
        wb[CDA_MLP_WB_W2 + ((size_t)blk * 4 + (o >> 6)) * 64 * HID + op_index(HID / 16, o & 1, i, (o & 63) >> 1)] = b;
        wb[CDA_MLP_WB_W2T + ((size_t)blk * 4 + (i >> 6)) * 64 * HID + op_index(HID / 16, i & 1, o, (i & 63) >> 1)] = b;
    } else if (p >= CDA_MLP_OFF_WO && p < CDA_MLP_OFF_BO) {                      // Wo[o][i] -> Wop (one tile of 32 outputs per half) and WoTp (rows = features, k = outputs)
        const int q = p - CDA_MLP_OFF_WO, o = q / HID, i = q - o * HID;
        if (o <= N_LOGITS + 2) {                                                 // (25, 26: the log-std head's rows of the policy half)
            const int hf = o == N_LOGITS ? 1 : 0;
            wb[CDA_MLP_WB_WO + (size_t)hf * NOUT * HID + op_index(HID / 16, 0, i, o)] = b;
            wb[CDA_MLP_WB_WOT + ((size_t)hf * 4 + (i >> 6)) * 64 * NOUT + op_index(NOUT / 16, i & 1, o, (i & 63) >> 1)] = b;
        }
    }
}
__global__ void k_pack(const float* __restrict__ theta, __bf16* __restrict__ wb) {
    const int p = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (p < CDA_MLP_PARAMS) pack_one(p, theta[p], wb);
}
// scratch f64[CDA_MLP_SCRATCH]: [2] = the squared norm of this call's gradient (output), [8 + b] = block b's share of it (k_grad_reduce)
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ theta, float* __restrict__ m, float* __restrict__ v, const float* __restrict__ step, __bf16* __restrict__ wb,
                                              const float* __restrict__ grad, double* __restrict__ scratch, float lr, float b1, float b2, float eps, float max_norm) {
    const int p = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    // every block sums the shares itself (307 doubles from L2): no grid-wide accumulator, nothing to clear, no fence
    __shared__ double part[4];
    double acc = 0.0;
    for (int b = (int)threadIdx.x; b < RED_DENSE_BLOCKS + RED_BIAS_BLOCKS; b += 256) acc += scratch[NORM_PARTIALS + b];
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    const double n2 = (part[0] + part[1]) + (part[2] + part[3]);
    if (p == 0) scratch[2] = n2;
    const float t = step[0];                                                    // (already counts this step: k_grad_reduce raised it)
    if (p < CDA_MLP_PARAMS) {
        const float coef = fminf(1.0f, max_norm / ((float)sqrt(n2) + 1e-6f));    // torch.nn.utils.clip_grad_norm_
        const float g = grad[p] * coef;
        const float mm = b1 * m[p] + (1.0f - b1) * g, vv = b2 * v[p] + (1.0f - b2) * g * g;
        m[p] = mm; v[p] = vv;
        const float c1 = 1.0f - __powf(b1, t), c2 = 1.0f - __powf(b2, t);
        const float th = theta[p] - (lr / c1) * mm / (sqrtf(vv) / sqrtf(c2) + eps);
        theta[p] = th;
        pack_one(p, th, wb);
    }
}

// ---- the loss for int32 actions (cda_ppo.hip's k_ppo_loss, same arithmetic; the env's own action tensors) ----------------------
template <int N>
__device__ __forceinline__ void head_probs(const float* l, float* p, float* lp, float& ent) {
    float mx = l[0];
    #pragma unroll
    for (int q = 1; q < N; q++) mx = fmaxf(mx, l[q]);
    float s = 0.0f;
    #pragma unroll
    for (int q = 0; q < N; q++) { p[q] = __expf(l[q] - mx); s += p[q]; }
    const float ls = __logf(s), inv = 1.0f / s;
    float hh = 0.0f;
    #pragma unroll
    for (int q = 0; q < N; q++) { p[q] *= inv; lp[q] = l[q] - mx - ls; hh -= p[q] * lp[q]; }
    ent = hh;
}
template <int N>
__device__ __forceinline__ float pick(const float* v, int a) {
    float r = v[0];
    #pragma unroll
    for (int q = 1; q < N; q++) r = (a == q || (q == N - 1 && a > q)) ? v[q] : r;
    return r;
}
__global__ __launch_bounds__(256) void k_ppo_loss32(const float* __restrict__ outputs, const float* __restrict__ log_std,
                                                    const int* __restrict__ a_cat, const int* __restrict__ a_price, const int* __restrict__ a_off,
                                                    const float* __restrict__ a_cont, const float* __restrict__ logp_old, const float* __restrict__ adv,
                                                    const float* __restrict__ ret, const long long* __restrict__ row_index, long long R, long long Rnorm, int agents,
                                                    int stride, float clip, float vf_coef, float ent_coef, float* __restrict__ d_out, double* __restrict__ sums) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const float invB = 1.0f / ((float)Rnorm * (float)agents);
    float pg = 0.0f, vl = 0.0f, en = 0.0f, dls0 = 0.0f, dls1 = 0.0f;
    if (r < R) {
        float l[N_LOGITS], d[N_LOGITS], p[N_CAT + N_PRICE + N_OFF], lp[N_CAT + N_PRICE + N_OFF];
        const float4* lp4 = reinterpret_cast<const float4*>(outputs + r * stride);
        #pragma unroll
        for (int q = 0; q < N_LOGITS / 4; q++) { const float4 v = lp4[q]; l[4 * q] = v.x; l[4 * q + 1] = v.y; l[4 * q + 2] = v.z; l[4 * q + 3] = v.w; }
        const float ls0 = log_std[0], ls1 = log_std[1];
        const float is0 = __expf(-ls0), is1 = __expf(-ls1);
        const float HALF_LOG_2PI = 0.918938533204672742f;
        float h0, h1, h2;
        head_probs<N_CAT>(l, p, lp, h0);
        head_probs<N_PRICE>(l + N_CAT, p + N_CAT, lp + N_CAT, h1);
        head_probs<N_OFF>(l + N_CAT + N_PRICE, p + N_CAT + N_PRICE, lp + N_CAT + N_PRICE, h2);
        const float ent = h0 + h1 + h2 + 1.0f + 2.0f * HALF_LOG_2PI + ls0 + ls1;
        const float es = ent_coef * invB;
        #pragma unroll
        for (int q = 0; q < N_LOGITS; q++) d[q] = 0.0f;
        const float val = outputs[r * stride + N_LOGITS];
        float G = 0.0f, dval = 0.0f;
        const long long src_row = row_index ? row_index[r] : r;
        for (int a = 0; a < agents; a++) {
            const long long i = src_row * agents + a;
            const int ac = a_cat[i], ap = a_price[i], ao = a_off[i];
            const float z0 = (a_cont[2 * i] - l[22]) * is0, z1 = (a_cont[2 * i + 1] - l[23]) * is1;
            const float logp = -0.5f * z0 * z0 - ls0 - HALF_LOG_2PI - 0.5f * z1 * z1 - ls1 - HALF_LOG_2PI +
                               pick<N_CAT>(lp, ac) + pick<N_PRICE>(lp + N_CAT, ap) + pick<N_OFF>(lp + N_CAT + N_PRICE, ao);
            const float Av = adv[i], ratio = __expf(logp - logp_old[i]);
            const float un = ratio * Av, cl = fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip) * Av;
            pg -= fminf(un, cl);
            const float g_logp = (un <= cl) ? -un * invB : 0.0f;
            const float dv = val - ret[i];
            vl += dv * dv;
            dval += 2.0f * vf_coef * dv * invB;
            en += ent;
            G += g_logp;
            #pragma unroll
            for (int q = 0; q < N_CAT; q++) d[q] += (q == ac) ? g_logp : 0.0f;
            #pragma unroll
            for (int q = 0; q < N_PRICE; q++) d[N_CAT + q] += (q == ap) ? g_logp : 0.0f;
            #pragma unroll
            for (int q = 0; q < N_OFF; q++) d[N_CAT + N_PRICE + q] += (q == ao) ? g_logp : 0.0f;
            d[22] += g_logp * z0 * is0;
            d[23] += g_logp * z1 * is1;
            dls0 += g_logp * (z0 * z0 - 1.0f) - es;
            dls1 += g_logp * (z1 * z1 - 1.0f) - es;
        }
        const float esA = es * (float)agents;
        #pragma unroll
        for (int q = 0; q < N_CAT; q++) d[q] += -G * p[q] + esA * p[q] * (lp[q] + h0);
        #pragma unroll
        for (int q = 0; q < N_PRICE; q++) d[N_CAT + q] += -G * p[N_CAT + q] + esA * p[N_CAT + q] * (lp[N_CAT + q] + h1);
        #pragma unroll
        for (int q = 0; q < N_OFF; q++) d[N_CAT + N_PRICE + q] += -G * p[N_CAT + N_PRICE + q] + esA * p[N_CAT + N_PRICE + q] * (lp[N_CAT + N_PRICE + q] + h2);
        float4* dp4 = reinterpret_cast<float4*>(d_out + r * stride);
        #pragma unroll
        for (int q = 0; q < N_LOGITS / 4; q++) dp4[q] = make_float4(d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]);
        dp4[N_LOGITS / 4] = make_float4(dval, 0.0f, 0.0f, 0.0f);
        for (int q = N_LOGITS / 4 + 1; q < stride / 4; q++) dp4[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    float v5[5] = {pg, vl, en, dls0, dls1};
    __shared__ float part[5][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    #pragma unroll
    for (int q = 0; q < 5; q++) {
        float x = v5[q];
        #pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
        if (lane == 0) part[q][wave] = x;
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        const double t = (double)part[threadIdx.x][0] + (double)part[threadIdx.x][1] + (double)part[threadIdx.x][2] + (double)part[threadIdx.x][3];
        atomicAdd(&sums[threadIdx.x], t);
    }
}
__global__ void k_ppo_finish32(const double* sums, long long B, float vf_coef, float ent_coef, float* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const double pg = sums[0] / (double)B, vl = sums[1] / (double)B, en = sums[2] / (double)B;
        out[0] = (float)pg; out[1] = (float)vl; out[2] = (float)en; out[3] = (float)(pg + (double)vf_coef * vl - (double)ent_coef * en);
        out[4] = (float)sums[3]; out[5] = (float)sums[4];
    }
}

__global__ void k_ppo_finish_slots(const double* sums, long long B, float vf_coef, float ent_coef, float kl_coef, float* out) {      // the same over CDA_MLP_LOSS_SLOTS slots (out f32[8])
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double t5[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        for (int sl = 0; sl < CDA_MLP_LOSS_SLOTS; sl++)
            for (int q = 0; q < 6; q++) t5[q] += sums[8 * sl + q];
        const double pg = t5[0] / (double)B, vl = t5[1] / (double)B, en = t5[2] / (double)B, kl = t5[5] / (double)B;
        out[0] = (float)pg; out[1] = (float)vl; out[2] = (float)en; out[3] = (float)(pg + (double)vf_coef * vl - (double)ent_coef * en + (double)kl_coef * kl);
        out[4] = (float)t5[3]; out[5] = (float)t5[4]; out[6] = (float)kl; out[7] = 0.0f;
    }
}

// ---- the rollout's sample records: GAE straight into them, and the loss reading them -----------------------------------------------------
// One thread per (market, agent) column walks its T steps backwards (ppo.gae's recursion) on the rollout's own buffers - reward f64 [T][N][A]
// (scaled here), value f32 [T + 1][N] (slot T = the bootstrap value), terminated / truncated u8 [T][N] - and writes advantage and return into
// words 6, 7 of the step's sample record.  The sums of the advantages and of their squares go to stats f64[2] (cleared by the caller): the
// update normalises on the fly, (adv - mean) / (std + 1e-8) with the unbiased std, as ppo_update does with torch ops.
// n_train > 0 (league self-play): slot p < n_train is played by trainable net p, whose values are value[p][T + 1][N] and whose sums go to stats[2 p ..];
// the other slots' samples feed no update and are skipped.
__global__ __launch_bounds__(256) void k_gae_records(const double* __restrict__ reward, const float* __restrict__ value, const unsigned char* __restrict__ term,
                                                     const unsigned char* __restrict__ trunc, int T, long long N, int Ag, int n_train, float reward_scale, float gamma, float lam,
                                                     const int* __restrict__ fin_index, const float* __restrict__ fin_value, long long fin_value_stride,
                                                     float* __restrict__ rec, double* __restrict__ stats) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x, B = N * Ag;
    double s1 = 0.0, s2 = 0.0;
    const int slot = (int)(i % Ag);
    if (i < B && (n_train <= 0 || slot < n_train)) {
        const long long n = i / Ag;
        if (n_train > 0) { value += (long long)slot * (T + 1) * N; fin_value += (long long)slot * fin_value_stride; }
        float nxt = value[(long long)T * N + n], run = 0.0f;
        // eight steps' operands requested together, then the recursion over them (one step at a time, every iteration paid a memory round trip:
        // 39 us for 64 steps)
        for (int t0 = T - 1; t0 >= 0; t0 -= 8) {
            float rw[8], vv[8], nd[8], bv[8];
            #pragma unroll
            for (int u = 0; u < 8; u++) {
                const int t = t0 - u >= 0 ? t0 - u : 0;
                const long long k = (long long)t * B + i, kn = (long long)t * N + n;
                const bool tm = term[kn] != 0, tr = trunc[kn] != 0;
                rw[u] = (float)reward[k] * reward_scale; vv[u] = value[kn]; nd[u] = (tm | tr) ? 0.0f : 1.0f;
                // a time-limit truncation (not a termination) whose last observation was captured: the step bootstraps with V(that observation) - the value
                // of the state the episode was cut in - instead of 0; nothing propagates across the episode boundary either way (nd = 0)
                bv[u] = 0.0f;
                if (fin_index && tr && !tm) { const int fi = fin_index[kn]; if (fi >= 0) bv[u] = fin_value[fi]; }
            }
            #pragma unroll
            for (int u = 0; u < 8; u++) {
                const int t = t0 - u;
                if (t >= 0) {
                    const long long k = (long long)t * B + i;
                    const float delta = rw[u] + gamma * (nxt * nd[u] + bv[u]) - vv[u];
                    run = delta + gamma * lam * nd[u] * run;
                    *reinterpret_cast<float2*>(rec + 8 * k + 6) = make_float2(run, run + vv[u]);
                    s1 += (double)run; s2 += (double)run * (double)run;
                    nxt = vv[u];
                }
            }
        }
    }
    if (n_train > 0) {                                                          // per net: a wave reduction and two atomics per (wave, net)
        for (int p = 0; p < n_train; p++) {
            double a1 = slot == p ? s1 : 0.0, a2 = slot == p ? s2 : 0.0;
            #pragma unroll
            for (int o = 32; o > 0; o >>= 1) { a1 += __shfl_down(a1, o, 64); a2 += __shfl_down(a2, o, 64); }
            if ((threadIdx.x & 63) == 0) { atomicAdd(&stats[2 * p], a1); atomicAdd(&stats[2 * p + 1], a2); }
        }
        return;
    }
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_down(s1, o, 64); s2 += __shfl_down(s2, o, 64); }
    __shared__ double part[2][4];
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = s1; part[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x < 2) atomicAdd(&stats[threadIdx.x], (part[threadIdx.x][0] + part[threadIdx.x][1]) + (part[threadIdx.x][2] + part[threadIdx.x][3]));
}
// k_ppo_loss32 on sample records: a row's A samples are ONE contiguous piece of A x 32 bytes (the seven separate per-sample arrays cost seven
// scattered 16-byte gathers per row: 1.3 KB fetched per row for 128 B used)
__global__ __launch_bounds__(256) void k_ppo_loss_rec(const float* __restrict__ outputs, const float* __restrict__ log_std, const float* __restrict__ rec,
                                                      const double* __restrict__ adv_stats, long long n_stat, const long long* __restrict__ row_index,
                                                      long long R, long long Rnorm, int agents, int stride, float clip, float vf_coef, float ent_coef,
                                                      float* __restrict__ d_out, double* __restrict__ sums) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const float invB = 1.0f / ((float)Rnorm * (float)agents);
    float adv_mean = 0.0f, adv_rstd = 1.0f;
    if (adv_stats) {
        const double m = adv_stats[0] / (double)n_stat, var = (adv_stats[1] - (double)n_stat * m * m) / (double)(n_stat - 1);
        adv_mean = (float)m; adv_rstd = 1.0f / ((float)sqrt(var > 0.0 ? var : 0.0) + 1e-8f);
    }
    float pg = 0.0f, vl = 0.0f, en = 0.0f, dls0 = 0.0f, dls1 = 0.0f;
    if (r < R) {
        float l[N_LOGITS], d[N_LOGITS], p[N_CAT + N_PRICE + N_OFF], lp[N_CAT + N_PRICE + N_OFF];
        const float4* lp4 = reinterpret_cast<const float4*>(outputs + r * stride);
        #pragma unroll
        for (int q = 0; q < N_LOGITS / 4; q++) { const float4 v = lp4[q]; l[4 * q] = v.x; l[4 * q + 1] = v.y; l[4 * q + 2] = v.z; l[4 * q + 3] = v.w; }
        const float ls0 = log_std[0], ls1 = log_std[1];
        const float is0 = __expf(-ls0), is1 = __expf(-ls1);
        const float HALF_LOG_2PI = 0.918938533204672742f;
        float h0, h1, h2;
        head_probs<N_CAT>(l, p, lp, h0);
        head_probs<N_PRICE>(l + N_CAT, p + N_CAT, lp + N_CAT, h1);
        head_probs<N_OFF>(l + N_CAT + N_PRICE, p + N_CAT + N_PRICE, lp + N_CAT + N_PRICE, h2);
        const float ent = h0 + h1 + h2 + 1.0f + 2.0f * HALF_LOG_2PI + ls0 + ls1;
        const float es = ent_coef * invB;
        #pragma unroll
        for (int q = 0; q < N_LOGITS; q++) d[q] = 0.0f;
        const float val = outputs[r * stride + N_LOGITS];
        float G = 0.0f, dval = 0.0f;
        const long long src_row = row_index ? row_index[r] : r;
        const float4* rp = reinterpret_cast<const float4*>(rec + src_row * agents * 8);
        for (int a = 0; a < agents; a++) {
            const float4 w0 = rp[2 * a], w1 = rp[2 * a + 1];
            const int ac = __float_as_int(w0.x), ap = __float_as_int(w0.y), ao = __float_as_int(w0.z);
            const float z0 = (w0.w - l[22]) * is0, z1 = (w1.x - l[23]) * is1;
            const float logp = -0.5f * z0 * z0 - ls0 - HALF_LOG_2PI - 0.5f * z1 * z1 - ls1 - HALF_LOG_2PI +
                               pick<N_CAT>(lp, ac) + pick<N_PRICE>(lp + N_CAT, ap) + pick<N_OFF>(lp + N_CAT + N_PRICE, ao);
            const float Av = (w1.z - adv_mean) * adv_rstd, ratio = __expf(logp - w1.y);
            const float un = ratio * Av, cl = fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip) * Av;
            pg -= fminf(un, cl);
            const float g_logp = (un <= cl) ? -un * invB : 0.0f;
            const float dv = val - w1.w;
            vl += dv * dv;
            dval += 2.0f * vf_coef * dv * invB;
            en += ent;
            G += g_logp;
            #pragma unroll
            for (int q = 0; q < N_CAT; q++) d[q] += (q == ac) ? g_logp : 0.0f;
            #pragma unroll
            for (int q = 0; q < N_PRICE; q++) d[N_CAT + q] += (q == ap) ? g_logp : 0.0f;
            #pragma unroll
            for (int q = 0; q < N_OFF; q++) d[N_CAT + N_PRICE + q] += (q == ao) ? g_logp : 0.0f;
            d[22] += g_logp * z0 * is0;
            d[23] += g_logp * z1 * is1;
            dls0 += g_logp * (z0 * z0 - 1.0f) - es;
            dls1 += g_logp * (z1 * z1 - 1.0f) - es;
        }
        const float esA = es * (float)agents;
        #pragma unroll
        for (int q = 0; q < N_CAT; q++) d[q] += -G * p[q] + esA * p[q] * (lp[q] + h0);
        #pragma unroll
        for (int q = 0; q < N_PRICE; q++) d[N_CAT + q] += -G * p[N_CAT + q] + esA * p[N_CAT + q] * (lp[N_CAT + q] + h1);
        #pragma unroll
        for (int q = 0; q < N_OFF; q++) d[N_CAT + N_PRICE + q] += -G * p[N_CAT + N_PRICE + q] + esA * p[N_CAT + N_PRICE + q] * (lp[N_CAT + N_PRICE + q] + h2);
        float4* dp4 = reinterpret_cast<float4*>(d_out + r * stride);
        #pragma unroll
        for (int q = 0; q < N_LOGITS / 4; q++) dp4[q] = make_float4(d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]);
        dp4[N_LOGITS / 4] = make_float4(dval, 0.0f, 0.0f, 0.0f);
        for (int q = N_LOGITS / 4 + 1; q < stride / 4; q++) dp4[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    float v5[5] = {pg, vl, en, dls0, dls1};
    __shared__ float part[5][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    #pragma unroll
    for (int q = 0; q < 5; q++) {
        float x = v5[q];
        #pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
        if (lane == 0) part[q][wave] = x;
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        const double t = (double)part[threadIdx.x][0] + (double)part[threadIdx.x][1] + (double)part[threadIdx.x][2] + (double)part[threadIdx.x][3];
        atomicAdd(&sums[threadIdx.x], t);
    }
}

// Returns of COMPLETED episodes from a rollout's buffers: running f64 [N][A] carries every (market, agent)'s return so far across rollouts; a step that ends
// the market's episode adds the agent's total to done_sum f64 [A] (and 1 to done_count f64 [A]) and restarts it.  One thread per (market, agent), forwards in time.
__global__ __launch_bounds__(256) void k_episode_returns(const double* __restrict__ reward, const unsigned char* __restrict__ term, const unsigned char* __restrict__ trunc,
                                                         int T, long long N, int Ag, double* __restrict__ running, double* __restrict__ done_sum, double* __restrict__ done_count,
                                                         double* __restrict__ per_slot) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x, B = N * Ag;
    if (i >= B) return;
    const long long n = i / Ag; const int a = (int)(i - n * Ag);
    double run = running[i], s = 0.0, c = 0.0;
    for (int t0 = 0; t0 < T; t0 += 8) {                                         // eight steps' operands requested together (one step at a time: a round trip per step, 72 us for 64)
        double rw[8]; bool dn[8];
        #pragma unroll
        for (int u = 0; u < 8; u++) {
            const int t = t0 + u < T ? t0 + u : T - 1;
            rw[u] = reward[(long long)t * B + i]; dn[u] = (term[(long long)t * N + n] | trunc[(long long)t * N + n]) != 0;
        }
        #pragma unroll
        for (int u = 0; u < 8; u++)
            if (t0 + u < T) { run += rw[u]; if (dn[u]) { s += run; c += 1.0; run = 0.0; } }
    }
    running[i] = run;
    if (c > 0.0) { atomicAdd(&done_sum[a], s); atomicAdd(&done_count[a], c); }
    if (per_slot) { per_slot[2 * i] = s; per_slot[2 * i + 1] = c; }              // this rollout's completed episodes of (market, agent): sum of returns, number
}
__global__ void k_copy_rows(const float* __restrict__ src, float* __restrict__ dst, long long n4) {         // n4 pieces of VW floats
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) reinterpret_cast<obsvec*>(dst)[i] = reinterpret_cast<const obsvec*>(src)[i];
}

__global__ void k_bump_counter(long long* counter) { if (threadIdx.x == 0) counter[0] += 1; }       // a chain's own rollout counter, behind its last launch

__global__ void k_selftest_mfma(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ d) {
    const int lane = (int)threadIdx.x, j = lane & 31, h = lane >> 5;
    bf16x8 av, bv;
    #pragma unroll
    for (int e = 0; e < 8; e++) { av[e] = (__bf16)a[j * 16 + 8 * h + e]; bv[e] = (__bf16)b[(8 * h + e) * 32 + j]; }
    f32x16 acc = zero16();
    acc = mfma(av, bv, acc);
    #pragma unroll
    for (int r = 0; r < 16; r++) d[rowmap(r, h) * 32 + j] = acc[r];
}

// The update's forward / backward kernels: CDA_MLP_WAVES = 8 (default): the 8-wave kernels with both network halves in flight, 64 rows per
// workgroup; CDA_MLP_WAVES = 4: the 4-wave kernels, 32 * CDA_MLP_MT rows per workgroup (CDA_MLP_MT = 1, 2 or 4; default 4).
int train_waves() {
    static int wv = 0;
    if (!wv) { const char* e = getenv("CDA_MLP_WAVES"); wv = e ? atoi(e) : 8; if (wv != 4 && wv != 8) wv = 8; }
    return wv;
}
int train_mt() {
    static int mt = 0;
    if (!mt) {
        const char* e = getenv("CDA_MLP_MT"); mt = e ? atoi(e) : (train_waves() == 8 ? 2 : 4);
        if (mt != 1 && mt != 2 && mt != 4) mt = train_waves() == 8 ? 2 : 4;
        if (train_waves() == 8 && mt == 4) mt = 2;                               // (two activation buffers of 128 rows do not fit the LDS)
    }
    return mt;
}
size_t fwd8_lds(int mt) { const size_t M = 32 * (size_t)mt; return M * XS_LD * 2 + 2 * M * ACT_LD * 2; }
size_t bwd8_lds(int mt) { const size_t M = 32 * (size_t)mt; return M * DO_LD * 2 + M * OUTS_LD * 4 + 2 * M * ACT_LD * 2; }
size_t fb_lds(int agents, bool with_dist) { return (size_t)FB_XS_BYTES + (size_t)64 * ACT_LD * 2 + (size_t)64 * LPS_LD * 4 + (size_t)64 * (agents * 32 + (with_dist ? CDA_MLP_DIST_LD * 4 : 0)); }
int rollout_mt() {
    static int mt = 0;
    if (!mt) { const char* e = getenv("CDA_MLP_ROLLOUT_MT"); mt = e ? atoi(e) : 1; if (mt != 1 && mt != 2 && mt != 4) mt = 1; }
    return mt;
}
size_t fwd_lds(int mt, int mode) { const size_t M = 32 * (size_t)mt; return M * XS_LD * 2 + M * ACT_LD * 2 + ((mode == MODE_SAMPLE || mode == MODE_LEAGUE) ? M * OUTS_LD * 4 : 0); }
size_t bwd_lds(int mt) { const size_t M = 32 * (size_t)mt; return M * DO_LD * 2 + M * ACT_LD * 2 + M * OUTS_LD * 4; }

// more than 64 KB of dynamic LDS needs the function attribute raised, once per (kernel, device): remembered here (a launch path, not a setup path)
template <typename K>
int allow_lds(K kern, size_t bytes) {
    if (bytes <= 64 * 1024) return CDA_OK;
    struct Granted { const void* fn; int dev; size_t bytes; };
    static Granted granted[64]; static int n_granted = 0;
    const void* fn = reinterpret_cast<const void*>(kern);
    int dev = 0; if (hipGetDevice(&dev) != hipSuccess) return CDA_ERR_HIP;
    for (int i = 0; i < n_granted; i++) if (granted[i].fn == fn && granted[i].dev == dev && granted[i].bytes >= bytes) return CDA_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return CDA_ERR_HIP;
    if (n_granted < 64) { granted[n_granted].fn = fn; granted[n_granted].dev = dev; granted[n_granted].bytes = bytes; n_granted++; }
    return CDA_OK;
}
template <int MODE>
int launch_fwd(const FwdArgs& A, int mt, hipStream_t st, unsigned grid_y = 0) {
    const size_t lds = fwd_lds(mt, MODE);
    const dim3 grid((unsigned)((A.n_rows + 32 * mt - 1) / (32 * mt)), grid_y ? grid_y : (A.split_halves == 1 ? 2u : 1u));
    int rc = CDA_OK;
    if (mt == 4) { rc = allow_lds(k_mlp_fwd<4, MODE>, lds); if (!rc) hipLaunchKernelGGL((k_mlp_fwd<4, MODE>), grid, dim3(256), lds, st, A); }
    else if (mt == 2) { rc = allow_lds(k_mlp_fwd<2, MODE>, lds); if (!rc) hipLaunchKernelGGL((k_mlp_fwd<2, MODE>), grid, dim3(256), lds, st, A); }
    else { rc = allow_lds(k_mlp_fwd<1, MODE>, lds); if (!rc) hipLaunchKernelGGL((k_mlp_fwd<1, MODE>), grid, dim3(256), lds, st, A); }
    if (rc) return rc;
    return hipGetLastError() == hipSuccess ? CDA_OK : CDA_ERR_HIP;
}
}  // namespace

#ifdef CDA_MLP_TIMING
// tools/mlp_timing.py: one launch of the training / sampling forward with cycle stamps of workgroup `block` -> dbg u64[4][32] (device)
extern "C" int cda_tools_mlp_fwd_timing(const void* wb, const float* theta, const void* x_rm, const float* obs, int64_t n_rows, int32_t agents, void* h1p, void* h2p, float* out,
                                        void* scratch_i32x3_f32x5, const int64_t* counter, int32_t mt, int32_t sample, void* dbg, int32_t block, void* stream) {
    FwdArgs A; memset(&A, 0, sizeof A);
    A.x_rm = (const __bf16*)x_rm; A.obs = obs; A.first_row = 0; A.n_rows = n_rows; A.wb = (const __bf16*)wb; A.theta = theta;
    A.h1p = (__bf16*)h1p; A.h2p = (__bf16*)h2p; A.out = out; A.dbg = (unsigned long long*)dbg; A.dbg_block = block;
    if (sample == 2) {                                   // the 8-wave training forward
        const size_t lds = fwd8_lds(2);
        int rc = allow_lds(k_mlp_fwd8<2>, lds); if (rc) return rc;
        hipLaunchKernelGGL(k_mlp_fwd8<2>, dim3((unsigned)((n_rows + 63) / 64)), dim3(512), lds, (hipStream_t)stream, A);
        return hipGetLastError() == hipSuccess ? CDA_OK : CDA_ERR_HIP;
    }
    if (sample) {
        const size_t NA = (size_t)n_rows * agents; char* s = (char*)scratch_i32x3_f32x5;
        A.agents = agents; A.seed = 1; A.counter = (const long long*)counter; A.draw = 0;
        A.env_cat = (int*)s; A.env_price = (int*)(s + 4 * NA); A.env_off = (int*)(s + 8 * NA); A.env_mean = (float*)(s + 12 * NA); A.env_sigma = (float*)(s + 16 * NA);
        A.a_cont = (float*)(s + 20 * NA); A.logp = (float*)(s + 28 * NA); A.value = (float*)(s + 32 * NA);
        A.split_halves = 1;                              // as the rollout launches it: the stamps are the policy half's workgroup
        return launch_fwd<MODE_SAMPLE>(A, mt, (hipStream_t)stream);
    }
    return launch_fwd<MODE_TRAIN>(A, mt, (hipStream_t)stream);
}
#endif

extern "C" int32_t cda_mlp_tile_rows(void) { return 32 * train_mt(); }
extern "C" int32_t cda_mlp_wgrad_jobs(void) { return WG_JOBS; }

extern "C" int cda_mlp_pack(const float* theta, void* wb, void* stream) {
    if (!theta || !wb) return CDA_ERR_INVALID;
    if (hipMemsetAsync(wb, 0, (size_t)CDA_MLP_WB_ELEMS * 2, (hipStream_t)stream) != hipSuccess) return CDA_ERR_HIP;   // paddings and the masked blocks
    hipLaunchKernelGGL(k_pack, dim3((CDA_MLP_PARAMS + 255) / 256), dim3(256), 0, (hipStream_t)stream, theta, (__bf16*)wb);
    return hipGetLastError() == hipSuccess ? CDA_OK : CDA_ERR_HIP;
}

extern "C" int cda_mlp_policy_step(const void* wb, const float* theta, const float* obs, int32_t first_market, int32_t n_markets, int32_t num_agents,
                                   uint64_t seed, const int64_t* counter_dev, int64_t draw,
                                   int32_t* env_category, float* env_size_mean, float* env_size_sigma, int32_t* env_price, int32_t* env_price_offset,
                                   float* a_cont, float* logp, float* value, void* stream) {
    if (!wb || !theta || !obs || !counter_dev || !env_category || !env_size_mean || !env_size_sigma || !env_price || !env_price_offset || !a_cont || !logp || !value ||
        first_market < 0 || n_markets < 1 || num_agents < 1 || num_agents > CDA_MAX_AGENTS) return CDA_ERR_INVALID;
    FwdArgs A; memset(&A, 0, sizeof A);
    A.obs = obs; A.first_row = first_market; A.n_rows = n_markets; A.wb = (const __bf16*)wb; A.theta = theta;
    A.agents = num_agents; A.seed = seed; A.counter = (const long long*)counter_dev; A.draw = draw;
    A.env_cat = env_category; A.env_mean = env_size_mean; A.env_sigma = env_size_sigma; A.env_price = env_price; A.env_off = env_price_offset;
    A.a_cont = a_cont; A.logp = logp; A.value = value;
    A.split_halves = 1;
    return launch_fwd<MODE_SAMPLE>(A, rollout_mt(), (hipStream_t)stream);
}

extern "C" int cda_mlp_forward(const void* wb, const float* theta, const float* obs, int64_t first_row, int64_t n_rows, float* out, void* stream) {
    if (!wb || !theta || !obs || !out || first_row < 0 || n_rows < 1) return CDA_ERR_INVALID;
    FwdArgs A; memset(&A, 0, sizeof A);
    A.obs = obs; A.first_row = first_row; A.n_rows = n_rows; A.wb = (const __bf16*)wb; A.theta = theta; A.out = out;
    A.split_halves = 1;
    return launch_fwd<MODE_OUT>(A, n_rows >= 32768 ? 4 : rollout_mt(), (hipStream_t)stream);
}

extern "C" int cda_mlp_permutation(uint64_t key, int64_t n, int64_t* perm, void* stream) {
    if (!perm || n < 1 || n > ((int64_t)1 << 31)) return CDA_ERR_INVALID;
    int bits = 1;
    while (((int64_t)1 << bits) < n) bits++;
    hipLaunchKernelGGL(k_make_perm, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (unsigned long long)key, (long long)n, bits, (long long*)perm);
    return hipGetLastError() == hipSuccess ? CDA_OK : CDA_ERR_HIP;
}

extern "C" int cda_mlp_prep_rows(const float* obs, const int64_t* perm, int64_t n_rows, void* x_rm, void* x_pk, void* stream) {
    if (!obs || !x_rm || !x_pk || n_rows < 32 || (n_rows & 31)) return CDA_ERR_INVALID;
    hipLaunchKernelGGL(k_prep_rows, dim3((unsigned)(n_rows / 32)), dim3(256), 0, (hipStream_t)stream, obs, (const long long*)perm, (long long)n_rows, (__bf16*)x_rm, (__bf16*)x_pk);
    return hipGetLastError() == hipSuccess ? CDA_OK : CDA_ERR_HIP;
}

extern "C" int cda_mlp_forward_train(const void* wb, const float* theta, const void* x_rm, int64_t n_rows, void* h1p, void* h2p, float* out, void* stream) {
    if (!wb || !theta || !x_rm || !h1p || !h2p || !out || n_rows < 32 || (n_rows & 31)) return CDA_ERR_INVALID;
    FwdArgs A; memset(&A, 0, sizeof A);
    A.x_rm = (const __bf16*)x_rm; A.first_row = 0; A.n_rows = n_rows; A.wb = (const __bf16*)wb; A.theta = theta;
    A.h1p = (__bf16*)h1p; A.h2p = (__bf16*)h2p; A.out = out;
    if (train_waves() == 8) {
        const int mt = train_mt();
        const size_t lds = fwd8_lds(mt);
        const unsigned grid = (unsigned)((n_rows + 32 * mt - 1) / (32 * mt));
        int rc;
        if (mt == 2) { rc = allow_lds(k_mlp_fwd8<2>, lds); if (!rc) hipLaunchKernelGGL(k_mlp_fwd8<2>, dim3(grid), dim3(512), lds, (hipStream_t)stream, A); }
        else { rc = allow_lds(k_mlp_fwd8<1>, lds); if (!rc) hipLaunchKernelGGL(k_mlp_fwd8<1>, dim3(grid), dim3(512), lds, (hipStream_t)stream, A); }
        if (rc) return rc;
        return hipGetLastError() == hipSuccess ? CDA_OK : CDA_ERR_HIP;
    }
    return launch_fwd<MODE_TRAIN>(A, train_mt(), (hipStream_t)stream);
}

extern "C" int cda_mlp_backward(const void* wb, const float* d_out, const void* h1p, const void* h2p, int64_t n_rows,
                                void* dz1p, void* dz2p, void* doutp, float* bias_slab, void* stream) {
    if (!wb || !d_out || !h1p || !h2p || !dz1p || !dz2p || !doutp || !bias_slab || n_rows < 32 || (n_rows & 31)) return CDA_ERR_INVALID;
    BwdArgs A; A.wb = (const __bf16*)wb; A.d_out = d_out; A.h1p = (const __bf16*)h1p; A.h2p = (const __bf16*)h2p; A.n_rows = n_rows;
    A.dz1p = (__bf16*)dz1p; A.dz2p = (__bf16*)dz2p; A.doutp = (__bf16*)doutp; A.bias_slab = bias_slab;
    const int mt = train_mt();
    const unsigned grid = (unsigned)((n_rows + 32 * mt - 1) / (32 * mt));
    int rc;
    if (train_waves() == 8) {
        const size_t lds8 = bwd8_lds(mt);
        if (mt == 2) { rc = allow_lds(k_mlp_bwd8<2>, lds8); if (!rc) hipLaunchKernelGGL(k_mlp_bwd8<2>, dim3(grid), dim3(512), lds8, (hipStream_t)stream, A); }
        else { rc = allow_lds(k_mlp_bwd8<1>, lds8); if (!rc) hipLaunchKernelGGL(k_mlp_bwd8<1>, dim3(grid), dim3(512), lds8, (hipStream_t)stream, A); }
        if (rc) return rc;
        return hipGetLastError() == hipSuccess ? CDA_OK : CDA_ERR_HIP;
    }
    const size_t lds = bwd_lds(mt);
    if (mt == 4) { rc = allow_lds(k_mlp_bwd<4>, lds); if (!rc) hipLaunchKernelGGL(k_mlp_bwd<4>, dim3(grid), dim3(256), lds, (hipStream_t)stream, A); }
    else if (mt == 2) { rc = allow_lds(k_mlp_bwd<2>, lds); if (!rc) hipLaunchKernelGGL(k_mlp_bwd<2>, dim3(grid), dim3(256), lds, (hipStream_t)stream, A); }
    else { rc = allow_lds(k_mlp_bwd<1>, lds); if (!rc) hipLaunchKernelGGL(k_mlp_bwd<1>, dim3(grid), dim3(256), lds, (hipStream_t)stream, A); }
    if (rc) return rc;
    return hipGetLastError() == hipSuccess ? CDA_OK : CDA_ERR_HIP;
}

#ifdef CDA_MLP_TIMING
static unsigned long long* g_fb_dbg = NULL; static int g_fb_dbg_block = 0;
extern "C" void cda_tools_mlp_fb_dbg(void* dbg_u64x8x32, int32_t block) { g_fb_dbg = (unsigned long long*)dbg_u64x8x32; g_fb_dbg_block = block; }
// tools/fb_wgrad_fusion_probe.py: FbArgs::exper, extra dynamic LDS for k_mlp_fb (above half of the CU's 160 KB: ONE workgroup per CU), weight-gradient jobs to leave out
static int g_fb_exper = 0, g_fb_lds_pad = 0, g_wgrad_first_job = 0;
extern "C" void cda_tools_mlp_experiment(int32_t fb_flags, int32_t fb_lds_pad_bytes, int32_t wgrad_first_job) { g_fb_exper = fb_flags; g_fb_lds_pad = fb_lds_pad_bytes; g_wgrad_first_job = wgrad_first_job; }
#endif
extern "C" int cda_mlp_forward_backward(const void* wb, const float* theta, const float* obs, const int64_t* perm, int64_t n_rows, int64_t norm_rows,
                                        const float* rec, const double* adv_stats2, int64_t adv_count, int32_t agents_per_row, float clip, float vf_coef, float ent_coef,
                                        const cda_ppo_extra* extra,
                                        void* x_pk, void* h1p, void* h2p, void* dz1p, void* dz2p, void* doutp, float* bias_slab,
                                        double* sums5, float* out6, int32_t clear, int32_t finish, float* out, float* d_out, void* stream) {
    if (!wb || !theta || !obs || !rec || !x_pk || !h1p || !h2p || !dz1p || !dz2p || !doutp || !bias_slab || !sums5 || n_rows < 32 || (n_rows & 31) || norm_rows < 0 ||
        agents_per_row < 1 || agents_per_row > CDA_MAX_AGENTS || (adv_stats2 && adv_count < 2) || (finish && !out6)) return CDA_ERR_INVALID;
    if (extra && ((extra->rec_stride != 0 && extra->rec_stride < agents_per_row * 8) || (extra->rec_stride & 3) || (extra->kl_coef != 0.0f && !extra->dist_old)))
        return CDA_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    FbArgs A; memset(&A, 0, sizeof A);
    A.obs = obs; A.perm = (const long long*)perm; A.n_rows = n_rows; A.norm_rows = norm_rows > 0 ? norm_rows : n_rows; A.wb = (const __bf16*)wb; A.theta = theta;
    A.rec = rec; A.adv_stats = adv_stats2; A.adv_count = adv_count; A.agents = agents_per_row; A.clip = clip; A.vf_coef = vf_coef; A.ent_coef = ent_coef;
    A.rec_stride = extra && extra->rec_stride ? extra->rec_stride : agents_per_row * 8;
    A.kl_coef = extra ? extra->kl_coef : 0.0f; A.vf_clip = extra ? extra->vf_clip : 0.0f;
    A.dist_old = extra && extra->kl_coef != 0.0f ? extra->dist_old : NULL; A.log_std_old = extra ? extra->log_std_old : NULL; A.sd_log_std = extra ? extra->sd_log_std : 0;
    A.x_pk = (__bf16*)x_pk; A.h1p = (__bf16*)h1p; A.h2p = (__bf16*)h2p; A.dz1p = (__bf16*)dz1p; A.dz2p = (__bf16*)dz2p; A.doutp = (__bf16*)doutp; A.bias_slab = bias_slab;
    A.out = out; A.d_out = d_out; A.sums5 = sums5;
    size_t lds = fb_lds(agents_per_row, A.dist_old != NULL);
    {   // CDA_MLP_FB_ONE_WG=1 in the environment: extra dynamic LDS so that ONE workgroup fits a CU instead of two (a launch parameter, not a code path: measuring knob)
        static int one_wg = -1;
        if (one_wg < 0) { const char* e = getenv("CDA_MLP_FB_ONE_WG"); one_wg = e ? atoi(e) : 0; }
        if (one_wg && lds <= 80 * 1024) lds = 81 * 1024;
    }
#ifdef CDA_MLP_TIMING
    A.dbg = g_fb_dbg; A.dbg_block = g_fb_dbg_block; A.exper = g_fb_exper; lds += (size_t)g_fb_lds_pad;
#endif
    if (clear && hipMemsetAsync(sums5, 0, (size_t)CDA_MLP_LOSS_SLOTS * 8 * sizeof(double), st) != hipSuccess) return CDA_ERR_HIP;
    int rc = allow_lds(k_mlp_fb, lds); if (rc) return rc;
    const long long tiles = (n_rows + 63) / 64;
    hipLaunchKernelGGL(k_mlp_fb, dim3((unsigned)(16 * ((tiles + 7) / 8))), dim3(256), lds, st, A);
    if (finish) hipLaunchKernelGGL(k_ppo_finish_slots, dim3(1), dim3(64), 0, st, (const double*)sums5, A.norm_rows * agents_per_row, vf_coef, ent_coef, A.kl_coef, out6);
    return hipGetLastError() == hipSuccess ? CDA_OK : CDA_ERR_HIP;
}

extern "C" int cda_mlp_wgrad(const void* x_pk, const void* h1p, const void* h2p, const void* dz1p, const void* dz2p, const void* doutp,
                             int64_t n_rows, int32_t n_chunks, float* slab, void* stream) {
    if (!x_pk || !h1p || !h2p || !dz1p || !dz2p || !doutp || !slab || n_rows < 32 || (n_rows & 31) || n_chunks < 1 || n_chunks > n_rows / 32) return CDA_ERR_INVALID;
    WgradArgs A; A.x_pk = (const bf16x8*)x_pk; A.h1p = (const bf16x8*)h1p; A.h2p = (const bf16x8*)h2p; A.dz1p = (const bf16x8*)dz1p; A.dz2p = (const bf16x8*)dz2p;
    A.doutp = (const bf16x8*)doutp; A.n_rt = n_rows / 32; A.n_chunks = n_chunks; A.slab = slab; A.first_job = 0;
    unsigned jobs = (unsigned)WG_JOBS;
#ifdef CDA_MLP_TIMING
    A.first_job = g_wgrad_first_job; jobs = (unsigned)WG_JOBS - (unsigned)g_wgrad_first_job;      // (timing only: jobs 0, 1 = dW2's two blocks)
#endif
    hipLaunchKernelGGL(k_mlp_wgrad, dim3((unsigned)n_chunks, jobs), dim3(256), 0, (hipStream_t)stream, A);
    return hipGetLastError() == hipSuccess ? CDA_OK : CDA_ERR_HIP;
}

// squared norm of an (all-reduced) gradient as the per-block shares k_adam sums: block b's share to scratch[NORM_PARTIALS + b], the other shares zeroed
namespace { __global__ __launch_bounds__(256) void k_grad_norm(const float* __restrict__ grad, double* __restrict__ scratch) {
    double acc = 0.0;
    for (int p = (int)(blockIdx.x * 256 + threadIdx.x); p < CDA_MLP_PARAMS; p += 256 * (RED_DENSE_BLOCKS + RED_BIAS_BLOCKS)) acc += (double)grad[p] * (double)grad[p];
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) scratch[NORM_PARTIALS + blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
} }
extern "C" int cda_mlp_reduce(const float* slab, int32_t n_chunks, const float* bias_slab, int32_t n_bias_tiles,
                              double* loss_sums5, int64_t loss_samples, float vf_coef, float ent_coef, float kl_coef, float* loss_out6, float* step_dev, float* grad, double* scratch3, void* stream) {
    if (!step_dev || !slab || !bias_slab || !grad || !scratch3 || n_chunks < 1 || n_bias_tiles < 1 || (loss_sums5 && loss_samples < 1)) return CDA_ERR_INVALID;
    LossFinish LF; LF.sums5 = loss_sums5; LF.samples = loss_samples; LF.vf_coef = vf_coef; LF.ent_coef = ent_coef; LF.kl_coef = kl_coef; LF.out6 = loss_out6;
    hipLaunchKernelGGL(k_grad_reduce, dim3(RED_DENSE_BLOCKS + RED_BIAS_BLOCKS), dim3(256), 0, (hipStream_t)stream, slab, (int)n_chunks, bias_slab, (int)n_bias_tiles, LF, grad, scratch3, step_dev);
    return hipGetLastError() == hipSuccess ? CDA_OK : CDA_ERR_HIP;
}
extern "C" int cda_mlp_apply(float* theta, float* adam_m, float* adam_v, const float* step_dev, void* wb, const float* grad, int32_t recompute_norm,
                             float lr, float beta1, float beta2, float eps, float max_norm, double* scratch3, void* stream) {
    if (!theta || !adam_m || !adam_v || !step_dev || !wb || !grad || !scratch3) return CDA_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (recompute_norm) hipLaunchKernelGGL(k_grad_norm, dim3(RED_DENSE_BLOCKS + RED_BIAS_BLOCKS), dim3(256), 0, st, grad, scratch3);
    hipLaunchKernelGGL(k_adam, dim3((CDA_MLP_PARAMS + 255) / 256), dim3(256), 0, st, theta, adam_m, adam_v, step_dev, (__bf16*)wb, grad, scratch3, lr, beta1, beta2, eps, max_norm);
    return hipGetLastError() == hipSuccess ? CDA_OK : CDA_ERR_HIP;
}
extern "C" int cda_mlp_adam(float* theta, float* adam_m, float* adam_v, float* step_dev, void* wb,
                            const float* slab, int32_t n_chunks, const float* bias_slab, int32_t n_bias_tiles,
                            double* loss_sums5, int64_t loss_samples, float vf_coef, float ent_coef, float kl_coef, float* loss_out6,
                            float lr, float beta1, float beta2, float eps, float max_norm, float* grad, double* scratch3, void* stream) {
    if (!theta || !adam_m || !adam_v || !step_dev || !wb || !slab || !bias_slab || !grad || !scratch3 || n_chunks < 1 || n_bias_tiles < 1 || (loss_sums5 && loss_samples < 1)) return CDA_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    LossFinish LF; LF.sums5 = loss_sums5; LF.samples = loss_samples; LF.vf_coef = vf_coef; LF.ent_coef = ent_coef; LF.kl_coef = kl_coef; LF.out6 = loss_out6;
    hipLaunchKernelGGL(k_grad_reduce, dim3(RED_DENSE_BLOCKS + RED_BIAS_BLOCKS), dim3(256), 0, st, slab, (int)n_chunks, bias_slab, (int)n_bias_tiles, LF, grad, scratch3, step_dev);
    hipLaunchKernelGGL(k_adam, dim3((CDA_MLP_PARAMS + 255) / 256), dim3(256), 0, st, theta, adam_m, adam_v, (const float*)step_dev, (__bf16*)wb, (const float*)grad, scratch3,
                       lr, beta1, beta2, eps, max_norm);
    return hipGetLastError() == hipSuccess ? CDA_OK : CDA_ERR_HIP;
}

extern "C" int cda_ppo_loss32(const float* outputs, const float* log_std, const int32_t* a_cat, const int32_t* a_price, const int32_t* a_off,
                              const float* a_cont, const float* logp_old, const float* adv, const float* ret, const int64_t* row_index,
                              int64_t rows, int32_t agents_per_row, int32_t out_stride, float clip, float vf_coef, float ent_coef,
                              float* d_outputs, double* sums5, float* out6, int64_t norm_rows, int32_t clear, int32_t finish, void* stream) {
    if (!outputs || !log_std || !a_cat || !a_price || !a_off || !a_cont || !logp_old || !adv || !ret || !d_outputs || !sums5 || !out6 || rows < 1 ||
        agents_per_row < 1 || agents_per_row > CDA_MAX_AGENTS || out_stride <= N_LOGITS || (out_stride & 3) || norm_rows < 0) return CDA_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const long long rn = norm_rows > 0 ? norm_rows : rows;
    if (clear && hipMemsetAsync(sums5, 0, 5 * sizeof(double), st) != hipSuccess) return CDA_ERR_HIP;
    hipLaunchKernelGGL(k_ppo_loss32, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, outputs, log_std, a_cat, a_price, a_off, a_cont, logp_old, adv, ret,
                       (const long long*)row_index, (long long)rows, rn, (int)agents_per_row, (int)out_stride, clip, vf_coef, ent_coef, d_outputs, sums5);
    if (finish) hipLaunchKernelGGL(k_ppo_finish32, dim3(1), dim3(64), 0, st, (const double*)sums5, rn * agents_per_row, vf_coef, ent_coef, out6);
    return hipGetLastError() == hipSuccess ? CDA_OK : CDA_ERR_HIP;
}

static int gae_records(const double* reward, const float* value, const uint8_t* terminated, const uint8_t* truncated, int32_t n_steps, int64_t n_markets,
                       int32_t num_agents, int32_t n_train, float reward_scale, float gamma, float lam, float* rec, double* stats, void* stream,
                       const int32_t* fin_index = NULL, const float* fin_value = NULL, int64_t fin_value_stride = 0) {
    if (!reward || !value || !terminated || !truncated || !rec || !stats || n_steps < 1 || n_markets < 1 || num_agents < 1 || num_agents > CDA_MAX_AGENTS ||
        n_train < 0 || n_train > num_agents || (fin_index && !fin_value)) return CDA_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(stats, 0, 2 * sizeof(double) * (n_train > 0 ? n_train : 1), st) != hipSuccess) return CDA_ERR_HIP;
    const long long B = (long long)n_markets * num_agents;
    hipLaunchKernelGGL(k_gae_records, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, reward, value, (const unsigned char*)terminated, (const unsigned char*)truncated,
                       (int)n_steps, (long long)n_markets, (int)num_agents, (int)n_train, reward_scale, gamma, lam, (const int*)fin_index, fin_value, (long long)fin_value_stride, rec, stats);
    return hipGetLastError() == hipSuccess ? CDA_OK : CDA_ERR_HIP;
}
extern "C" int cda_gae_records(const double* reward, const float* value, const uint8_t* terminated, const uint8_t* truncated, int32_t n_steps, int64_t n_markets,
                               int32_t num_agents, float reward_scale, float gamma, float lam, float* rec, double* stats2, void* stream) {
    return gae_records(reward, value, terminated, truncated, n_steps, n_markets, num_agents, 0, reward_scale, gamma, lam, rec, stats2, stream);
}
// ... with the time-limit bootstrap: fin_index i32 [T][N] (slot of the step's captured last observation, -1 = none), fin_value f32 [max(n_trainable, 1)][fin_value_stride]
// (cda_mlp_values on the captured list).  n_trainable = 0: one shared policy (cda_gae_records' layout), > 0: the league's (cda_gae_records_league's).
extern "C" int cda_gae_records_bootstrap(const double* reward, const float* value, const uint8_t* terminated, const uint8_t* truncated, int32_t n_steps, int64_t n_markets,
                                         int32_t num_agents, int32_t n_trainable, float reward_scale, float gamma, float lam,
                                         const int32_t* fin_index, const float* fin_value, int64_t fin_value_stride, float* rec, double* stats, void* stream) {
    return gae_records(reward, value, terminated, truncated, n_steps, n_markets, num_agents, n_trainable, reward_scale, gamma, lam, rec, stats, stream, fin_index, fin_value, fin_value_stride);
}
extern "C" int cda_gae_records_league(const double* reward, const float* value, const uint8_t* terminated, const uint8_t* truncated, int32_t n_steps, int64_t n_markets,
                                      int32_t num_agents, int32_t n_trainable, float reward_scale, float gamma, float lam, float* rec, double* stats2k, void* stream) {
    if (n_trainable < 1) return CDA_ERR_INVALID;
    return gae_records(reward, value, terminated, truncated, n_steps, n_markets, num_agents, n_trainable, reward_scale, gamma, lam, rec, stats2k, stream);
}

extern "C" int cda_episode_returns(const double* reward, const uint8_t* terminated, const uint8_t* truncated, int32_t n_steps, int64_t n_markets, int32_t num_agents,
                                   double* running, double* done_sum, double* done_count, double* per_slot, void* stream) {
    if (!reward || !terminated || !truncated || !running || !done_sum || !done_count || n_steps < 1 || n_markets < 1 || num_agents < 1 || num_agents > CDA_MAX_AGENTS) return CDA_ERR_INVALID;
    const long long B = (long long)n_markets * num_agents;
    hipLaunchKernelGGL(k_episode_returns, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, reward, (const unsigned char*)terminated, (const unsigned char*)truncated,
                       (int)n_steps, (long long)n_markets, (int)num_agents, running, done_sum, done_count, per_slot);
    return hipGetLastError() == hipSuccess ? CDA_OK : CDA_ERR_HIP;
}

extern "C" int cda_ppo_loss_records(const float* outputs, const float* log_std, const float* rec, const double* adv_stats2, int64_t adv_count, const int64_t* row_index,
                                    int64_t rows, int32_t agents_per_row, int32_t out_stride, float clip, float vf_coef, float ent_coef,
                                    float* d_outputs, double* sums5, float* out6, int64_t norm_rows, int32_t clear, int32_t finish, void* stream) {
    if (!outputs || !log_std || !rec || !d_outputs || !sums5 || !out6 || rows < 1 || agents_per_row < 1 || agents_per_row > CDA_MAX_AGENTS ||
        out_stride <= N_LOGITS || (out_stride & 3) || norm_rows < 0 || (adv_stats2 && adv_count < 2)) return CDA_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const long long rn = norm_rows > 0 ? norm_rows : rows;
    if (clear && hipMemsetAsync(sums5, 0, 5 * sizeof(double), st) != hipSuccess) return CDA_ERR_HIP;
    hipLaunchKernelGGL(k_ppo_loss_rec, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, outputs, log_std, rec, adv_stats2, (long long)adv_count,
                       (const long long*)row_index, (long long)rows, rn, (int)agents_per_row, (int)out_stride, clip, vf_coef, ent_coef, d_outputs, sums5);
    if (finish) hipLaunchKernelGGL(k_ppo_finish32, dim3(1), dim3(64), 0, st, (const double*)sums5, rn * agents_per_row, vf_coef, ent_coef, out6);
    return hipGetLastError() == hipSuccess ? CDA_OK : CDA_ERR_HIP;
}

extern "C" int cda_mlp_league_step(const cda_league* L, const float* obs, int32_t first_market, int32_t n_markets, int32_t num_agents,
                                   uint64_t seed, const int64_t* counter_dev, int64_t draw,
                                   int32_t* env_category, float* env_size_mean, float* env_size_sigma, int32_t* env_price, int32_t* env_price_offset,
                                   float* a_cont, float* logp, float* value, int64_t value_stride, float* rec, float* dist, int64_t dist_stride, void* stream) {
    if (!L || !L->wb_bank || !L->theta_bank || !L->slot_net || L->n_trainable < 1 || L->n_nets < L->n_trainable || L->n_nets > CDA_LEAGUE_MAX_NETS ||
        !obs || !counter_dev || !env_category || !env_size_mean || !env_size_sigma || !env_price || !env_price_offset || !a_cont || !logp || !value ||
        first_market < 0 || n_markets < 1 || num_agents < 1 || num_agents > CDA_MAX_AGENTS) return CDA_ERR_INVALID;
    FwdArgs A; memset(&A, 0, sizeof A);
    A.obs = obs; A.first_row = first_market; A.n_rows = n_markets; A.wb = (const __bf16*)L->wb_bank; A.theta = L->theta_bank;
    A.agents = num_agents; A.seed = seed; A.counter = (const long long*)counter_dev; A.draw = draw;
    A.env_cat = env_category; A.env_mean = env_size_mean; A.env_sigma = env_size_sigma; A.env_price = env_price; A.env_off = env_price_offset;
    A.a_cont = a_cont; A.logp = logp; A.value = value; A.rec = rec; A.dist = dist;
    A.n_train = L->n_trainable; A.slot_net = L->slot_net; A.value_stride = value_stride; A.dist_stride = dist_stride; A.random_seed = L->random_seed;
    A.split_halves = 1;
    return launch_fwd<MODE_LEAGUE>(A, rollout_mt(), (hipStream_t)stream, (unsigned)(L->n_trainable + L->n_nets));     // 2 jobs per trainable net, 1 per frozen one
}

// The reference's agent-to-module mapping (train/callbk/league_based_self_play_callback.py:1286-1344) for every (market, pool slot) at once: slot s >= n_trainable
// of a market draws np.random.RandomState((crc32(str(episode id)) + s) mod 2^32).choice(pool, p) - ONE random_sample() of a freshly seeded MT19937: the
// init_genrand recurrence up to word 398, the twist + tempering of outputs 0 and 1, a 53-bit double, searchsorted(cdf, u, side = "right").  A thread per
// (market, slot): 400 dependent integer steps (the host-side numpy restatement, league.mt19937_first_double, walks the same recurrence over all seeds at
// once: tens of milliseconds at 2048 x 6 - longer than the episode it assigns).
namespace { __global__ __launch_bounds__(256) void k_league_assign(const unsigned int* __restrict__ episode_crc, int N, int Ag, int n_train, const double* __restrict__ cdf,
                                                       const int* __restrict__ pool_net, int P, int* __restrict__ slot_net, int* __restrict__ slot_pool) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= N * Ag) return;
    const int n = i / Ag, s = i - n * Ag;
    if (s < n_train) { slot_net[i] = s; if (slot_pool) slot_pool[i] = -1; return; }
    unsigned int x = episode_crc[n] + (unsigned int)s;                          // (crc + slot) mod 2^32
    unsigned int m0 = x, m1 = 0, m2 = 0, m397 = 0, m398 = 0;
    for (unsigned int k = 1; k <= 398; k++) {
        x = 1812433253u * (x ^ (x >> 30)) + k;
        if (k == 1) m1 = x; else if (k == 2) m2 = x; else if (k == 397) m397 = x; else if (k == 398) m398 = x;
    }
    auto word = [](unsigned int a, unsigned int b, unsigned int c) {             // output k: twist of (mt[k], mt[k + 1], mt[k + 397]), tempered
        const unsigned int y = (a & 0x80000000u) | (b & 0x7fffffffu);
        unsigned int v = c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        v ^= v >> 11; v ^= (v << 7) & 0x9d2c5680u; v ^= (v << 15) & 0xefc60000u; v ^= v >> 18;
        return v;
    };
    const unsigned int w0 = word(m0, m1, m397) >> 5, w1 = word(m1, m2, m398) >> 6;
    const double u = ((double)w0 * 67108864.0 + (double)w1) / 9007199254740992.0;
    int idx = 0;
    for (int q = 0; q < P; q++) idx += cdf[q] <= u ? 1 : 0;                      // searchsorted(..., side = "right")
    if (idx >= P) idx = P - 1;
    slot_net[i] = pool_net[idx];
    if (slot_pool) slot_pool[i] = idx;
} }
extern "C" int cda_league_assign(const uint32_t* episode_crc, int32_t n_markets, int32_t num_agents, int32_t n_trainable, const double* pool_cdf, const int32_t* pool_net,
                                 int32_t pool_size, int32_t* slot_net, int32_t* slot_pool, void* stream) {
    if (!episode_crc || !pool_cdf || !pool_net || !slot_net || n_markets < 1 || num_agents < 1 || num_agents > CDA_MAX_AGENTS || n_trainable < 0 || n_trainable > num_agents ||
        pool_size < 1) return CDA_ERR_INVALID;
    const int n = n_markets * num_agents;
    hipLaunchKernelGGL(k_league_assign, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const unsigned int*)episode_crc, (int)n_markets, (int)num_agents,
                       (int)n_trainable, pool_cdf, (const int*)pool_net, (int)pool_size, (int*)slot_net, (int*)slot_pool);
    return hipGetLastError() == hipSuccess ? CDA_OK : CDA_ERR_HIP;
}

#ifndef CDA_MLP_TIMING          /* (the tools build holds the network kernels only, not the env) */
// One chain's rollout: the loop of {policy step, env step} launches; L != NULL: the league's policy step (banks of nets, per-slot modules)
static int rollout_chain(cda_env* env, const cda_league* L, const void* wb, const float* theta, int32_t first_market, int32_t n_markets, int32_t n_steps,
                         uint64_t seed, const int64_t* counter_dev, const cda_rollout_bufs* B, int32_t copy_first_obs, void* stream) {
    if (!env || !counter_dev || !B || n_steps < 1 || first_market < 0 || n_markets < 1) return CDA_ERR_INVALID;
    if (L ? (!L->wb_bank || !L->theta_bank || !L->slot_net || L->n_trainable < 1 || L->n_nets < L->n_trainable || L->n_nets > CDA_LEAGUE_MAX_NETS) : (!wb || !theta)) return CDA_ERR_INVALID;
    if (!B->obs || !B->category || !B->size_mean || !B->size_sigma || !B->price || !B->price_offset || !B->a_cont || !B->logp || !B->value || !B->reward ||
        !B->terminated || !B->truncated) return CDA_ERR_INVALID;
    if (B->fin_index && (!B->fin_obs || !B->fin_count || B->fin_cap < 1)) return CDA_ERR_INVALID;
    const int64_t N = cda_num_markets(env);
    if (cda_obs_dim(env) != OBS || (int64_t)first_market + n_markets > N) return CDA_ERR_INVALID;
    const int32_t A = cda_num_agents(env);
    hipStream_t st = (hipStream_t)stream;
    const size_t NA = (size_t)N * A;
    const int n_train = L ? L->n_trainable : 0;
    if (copy_first_obs) {
        const long long n4 = (long long)n_markets * OBS / VW;
        hipLaunchKernelGGL(k_copy_rows, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, B->obs + ((size_t)n_steps * N + first_market) * OBS,
                           B->obs + (size_t)first_market * OBS, n4);
    }
    // One launch per step where the env qualifies (include/cda.h cda_policy_step_range: the policy evaluated inside the step kernel) - one shared policy,
    // no info chain; CDA_POLICY_STEP=0 in the environment keeps the two launches (A / B runs, tests of both paths).
    bool one_launch = false;
    {
        const char* ev = getenv("CDA_POLICY_STEP");              // read per call: an A / B run toggles it between two rollouts of one process
        const int want = ev ? atoi(ev) : 1;
        // (the env's history depth is this build's: checked above.  CDA_POLICY_STEP=2: wherever supported, also where the batched policy kernel is the faster one)
        one_launch = want != 0 && !L && !B->info_steps && (want == 2 ? cda_policy_step_supported(env) : cda_policy_step_advised(env));
    }
    for (int32_t t = 0; t < n_steps; t++) {
        const size_t o = (size_t)t * NA;
        if (one_launch) {
            const int rc1 = cda_policy_step_range(env, first_market, n_markets, wb, theta, B->obs + (size_t)t * N * OBS, seed, counter_dev, t,
                                                  B->category + o, B->size_mean + o, B->size_sigma + o, B->price + o, B->price_offset + o,
                                                  B->a_cont + 2 * o, B->logp + o, B->value + (size_t)t * N, B->record ? B->record + 8 * o : NULL,
                                                  B->dist ? B->dist + (size_t)t * N * CDA_MLP_DIST_LD : NULL,
                                                  B->obs + (size_t)(t + 1) * N * OBS, B->reward + o, B->terminated + (size_t)t * N, B->truncated + (size_t)t * N,
                                                  B->fin_obs, B->fin_cap, B->fin_count, B->fin_index ? B->fin_index + (size_t)t * N : NULL, stream);
            if (rc1 == CDA_OK) continue;
            if (t != 0) return rc1;
            one_launch = false;                                  // refused at the first step (nothing has run yet): the two launches take over
            (void)hipGetLastError();
        }
        FwdArgs P; memset(&P, 0, sizeof P);
        P.obs = B->obs + (size_t)t * N * OBS; P.first_row = first_market; P.n_rows = n_markets;
        P.wb = (const __bf16*)(L ? L->wb_bank : wb); P.theta = L ? L->theta_bank : theta;
        P.agents = A; P.seed = seed; P.counter = (const long long*)counter_dev; P.draw = t;
        P.env_cat = B->category + o; P.env_mean = B->size_mean + o; P.env_sigma = B->size_sigma + o; P.env_price = B->price + o; P.env_off = B->price_offset + o;
        P.a_cont = B->a_cont + 2 * o; P.logp = B->logp + o; P.value = B->value + (size_t)t * N;
        P.rec = B->record ? B->record + 8 * o : NULL;
        P.dist = B->dist ? B->dist + (size_t)t * N * CDA_MLP_DIST_LD : NULL;
        P.split_halves = 1;
        int rc;
        if (L) {
            P.n_train = n_train; P.slot_net = L->slot_net; P.random_seed = L->random_seed;
            P.value_stride = (long long)(n_steps + 1) * N; P.dist_stride = (long long)n_steps * N * CDA_MLP_DIST_LD;
            rc = launch_fwd<MODE_LEAGUE>(P, rollout_mt(), st, (unsigned)(L->n_trainable + L->n_nets));
        } else rc = launch_fwd<MODE_SAMPLE>(P, rollout_mt(), st);
        if (rc) return rc;
        rc = cda_step_range_capture(env, first_market, n_markets, B->category + o, B->size_mean + o, B->size_sigma + o, B->price + o, B->price_offset + o, NULL,
                                    B->obs + (size_t)(t + 1) * N * OBS, B->reward + o, B->terminated + (size_t)t * N, B->truncated + (size_t)t * N,
                                    B->info_steps ? &B->info_steps[t] : NULL,
                                    B->fin_obs, B->fin_cap, B->fin_count, B->fin_index ? B->fin_index + (size_t)t * N : NULL, stream);
        if (rc) return rc;
    }
    // the bootstrap value of the last observation (league: of every trainable net)
    FwdArgs V; memset(&V, 0, sizeof V);
    V.obs = B->obs + (size_t)n_steps * N * OBS; V.first_row = first_market; V.n_rows = n_markets;
    V.wb = (const __bf16*)(L ? L->wb_bank : wb); V.theta = L ? L->theta_bank : theta;
    V.value = B->value + (size_t)n_steps * N;
    V.n_train = n_train; V.value_stride = (long long)(n_steps + 1) * N;
    V.split_halves = 2;                                  // the value network alone
    const int rcv = launch_fwd<MODE_VALUE>(V, rollout_mt(), st, L ? (unsigned)n_train : 0u);
    if (rcv) return rcv;
    if (B->counter_bump) {                               // fresh draws for this chain's next rollout: no launch ahead of a rollout, none on anybody's critical path
        hipLaunchKernelGGL(k_bump_counter, dim3(1), dim3(64), 0, st, (long long*)B->counter_bump);
        if (hipGetLastError() != hipSuccess) return CDA_ERR_HIP;
    }
    return CDA_OK;
}
extern "C" int cda_mlp_rollout_chain(cda_env* env, const void* wb, const float* theta, int32_t first_market, int32_t n_markets, int32_t n_steps,
                                     uint64_t seed, const int64_t* counter_dev, const cda_rollout_bufs* B, int32_t copy_first_obs, void* stream) {
    return rollout_chain(env, NULL, wb, theta, first_market, n_markets, n_steps, seed, counter_dev, B, copy_first_obs, stream);
}
extern "C" int cda_mlp_league_rollout_chain(cda_env* env, const cda_league* L, int32_t first_market, int32_t n_markets, int32_t n_steps,
                                            uint64_t seed, const int64_t* counter_dev, const cda_rollout_bufs* B, int32_t copy_first_obs, void* stream) {
    if (!L) return CDA_ERR_INVALID;
    return rollout_chain(env, L, NULL, NULL, first_market, n_markets, n_steps, seed, counter_dev, B, copy_first_obs, stream);
}
#endif

// The values of captured last observations (cda_step_range_capture's list): value f32 [n_nets_trainable][cap] <- the value network(s) on fin_obs [cap][168].  Every
// row of the list is evaluated (the count lives on the device; rows never written hold finite garbage nobody reads).
extern "C" int cda_mlp_values_counted(const void* wb_bank, const float* theta_bank, int32_t n_nets, const float* obs, int64_t n_rows, const int32_t* n_rows_dev,
                                      float* value, int64_t value_stride, void* stream) {
    if (!wb_bank || !theta_bank || n_nets < 1 || n_nets > CDA_LEAGUE_MAX_NETS || !obs || !value || n_rows < 1) return CDA_ERR_INVALID;
    FwdArgs V; memset(&V, 0, sizeof V);
    V.obs = obs; V.first_row = 0; V.n_rows = n_rows; V.wb = (const __bf16*)wb_bank; V.theta = theta_bank; V.value = value;
    V.n_train = n_nets; V.value_stride = value_stride; V.split_halves = 2; V.rows_limit = (const int*)n_rows_dev;
    return launch_fwd<MODE_VALUE>(V, n_rows >= 32768 ? 4 : rollout_mt(), (hipStream_t)stream, (unsigned)n_nets);
}
extern "C" int cda_mlp_values(const void* wb_bank, const float* theta_bank, int32_t n_nets, const float* obs, int64_t n_rows, float* value, int64_t value_stride, void* stream) {
    return cda_mlp_values_counted(wb_bank, theta_bank, n_nets, obs, n_rows, NULL, value, value_stride, stream);
}

extern "C" int cda_mlp_selftest_mfma(int32_t device, const float* a_host, const float* b_host, float* d_host) {
    if (!a_host || !b_host || !d_host) return CDA_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return CDA_ERR_NO_DEVICE;
    float *a = NULL, *b = NULL, *d = NULL;
    if (hipMalloc((void**)&a, 32 * 16 * 4) != hipSuccess || hipMalloc((void**)&b, 16 * 32 * 4) != hipSuccess || hipMalloc((void**)&d, 32 * 32 * 4) != hipSuccess) return CDA_ERR_NOMEM;
    int rc = CDA_OK;
    if (hipMemcpy(a, a_host, 32 * 16 * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(b, b_host, 16 * 32 * 4, hipMemcpyHostToDevice) != hipSuccess) rc = CDA_ERR_HIP;
    if (!rc) {
        hipLaunchKernelGGL(k_selftest_mfma, dim3(1), dim3(64), 0, 0, (const float*)a, (const float*)b, d);
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(d_host, d, 32 * 32 * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = CDA_ERR_HIP;
    }
    (void)hipFree(a); (void)hipFree(b); (void)hipFree(d);
    return rc;
}
