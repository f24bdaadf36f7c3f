// cda_ppo.hip - the learner-side hot op of the PPO loop on the batched env (the reference trains through RLlib's
// PPO, train/train.py:453-541): the clipped-surrogate / value / entropy loss of one minibatch AND its gradient with respect to
// the network outputs, in ONE pass over the minibatch.  In PyTorch ops this is ~100 elementwise / reduction launches per
// minibatch over [B, 24] tensors (three log-softmaxes, gathers, exp, clamp, min, means ... and their backward), which is what
// the round-2 update spent its time launching; here one thread owns one sample: 24 logits in registers, everything else
// follows from them.  Plain pointers in, plain pointers out (include/cda.h cda_ppo_loss); no torch types.
//
//   policy heads (ppo.py): category 9 | price 10 | price_offset 3 logits, then the means of the two Gaussian heads (size_mean,
//   size_sigma before squashing), shared log_std[2].
//   logp   = sum_h log_softmax(l_h)[a_h] + sum_d ( -z_d^2 / 2 - log_std_d - log(2 pi) / 2 ),  z_d = (a_d - mu_d) exp(-log_std_d)
//   ratio  = exp(logp - logp_old);  surrogate = min(ratio A, clamp(ratio, 1 - c, 1 + c) A)
//   loss   = mean(-surrogate) + vf mean((v - ret)^2) - ent_coef mean(entropy)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cda.h"

namespace {
constexpr int N_CAT = 9, N_PRICE = 10, N_OFF = 3, N_LOGITS = N_CAT + N_PRICE + N_OFF + 2;     // 24

// softmax of one categorical head, kept per row: p_j, log p_j and the entropy
template <int N>
__device__ __forceinline__ void head_probs(const float* l, float* p, float* lp, float& ent) {
    float mx = l[0];
    #pragma unroll
    for (int j = 1; j < N; j++) mx = fmaxf(mx, l[j]);
    float s = 0.0f;
    #pragma unroll
    for (int j = 0; j < N; j++) { p[j] = __expf(l[j] - mx); s += p[j]; }
    const float ls = __logf(s), inv = 1.0f / s;
    float h = 0.0f;
    #pragma unroll
    for (int j = 0; j < N; j++) { p[j] *= inv; lp[j] = l[j] - mx - ls; h -= p[j] * lp[j]; }
    ent = h;
}
template <int N>
__device__ __forceinline__ float pick(const float* v, int a) {           // v[clamp(a)] without dynamic register indexing
    float r = v[0];
    #pragma unroll
    for (int j = 1; j < N; j++) r = (a == j || (j == N - 1 && a > j)) ? v[j] : r;
    return r;
}

// One thread per ROW of network outputs.  A row serves `agents` consecutive samples (the agents of one market see the same
// observation - state_helper.py:76,109 - so the shared policy's logits and value are the same for all of them and the network
// runs once per market-step, not once per agent; agents == 1 is the plain per-sample op).  The softmaxes are formed once per
// row; per sample only the gathers, the ratio and the clipping remain; the row's gradient is the sum over its samples.
__global__ __launch_bounds__(256) void k_ppo_loss(const float* __restrict__ logits, const float* __restrict__ value, const float* __restrict__ log_std,
                                                  const long long* __restrict__ a_cat, const long long* __restrict__ a_price, const long long* __restrict__ a_off,
                                                  const float* __restrict__ a_cont, const float* __restrict__ logp_old, const float* __restrict__ adv,
                                                  const float* __restrict__ ret, const long long* __restrict__ row_index, long long R, int agents,
                                                  int stride, int packed, float clip, float vf_coef, float ent_coef,
                                                  float* __restrict__ d_logits, float* __restrict__ d_value, double* __restrict__ sums) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const float invB = 1.0f / ((float)R * (float)agents);
    float pg = 0.0f, vl = 0.0f, en = 0.0f, dls0 = 0.0f, dls1 = 0.0f;
    if (r < R) {
        float l[N_LOGITS], d[N_LOGITS], p[N_CAT + N_PRICE + N_OFF], lp[N_CAT + N_PRICE + N_OFF];
        const float4* lp4 = reinterpret_cast<const float4*>(logits + r * stride);            // 96 B per row, 16-byte aligned (stride % 4 == 0)
        #pragma unroll
        for (int q = 0; q < N_LOGITS / 4; q++) { const float4 v = lp4[q]; l[4 * q] = v.x; l[4 * q + 1] = v.y; l[4 * q + 2] = v.z; l[4 * q + 3] = v.w; }
        const float ls0 = log_std[0], ls1 = log_std[1];
        const float is0 = __expf(-ls0), is1 = __expf(-ls1);
        const float HALF_LOG_2PI = 0.918938533204672742f;
        float h0, h1, h2;
        head_probs<N_CAT>(l, p, lp, h0);
        head_probs<N_PRICE>(l + N_CAT, p + N_CAT, lp + N_CAT, h1);
        head_probs<N_OFF>(l + N_CAT + N_PRICE, p + N_CAT + N_PRICE, lp + N_CAT + N_PRICE, h2);
        const float ent = h0 + h1 + h2 + 1.0f + 2.0f * HALF_LOG_2PI + ls0 + ls1;             // two Gaussian heads: 1/2 + log(2 pi)/2 + log_std each
        const float es = ent_coef * invB;                                                     // loss has -ent_coef * mean(ent)
        #pragma unroll
        for (int j = 0; j < N_LOGITS; j++) d[j] = 0.0f;
        const float val = packed ? logits[r * stride + N_LOGITS] : value[r];                 // packed: the value is column 24 of the same row
        float G = 0.0f, dval = 0.0f;                                                          // sum of d loss / d logp over the row's samples
        // the row's samples: rows r of the network outputs are a shuffled minibatch, row_index[r] is where its samples live in the
        // (unshuffled) per-sample arrays - the epoch's shuffle then moves the observations only
        const long long src_row = row_index ? row_index[r] : r;
        for (int a = 0; a < agents; a++) {
            const long long i = src_row * agents + a;
            const int ac = (int)a_cat[i], ap = (int)a_price[i], ao = (int)a_off[i];
            const float z0 = (a_cont[2 * i] - l[22]) * is0, z1 = (a_cont[2 * i + 1] - l[23]) * is1;
            const float logp = -0.5f * z0 * z0 - ls0 - HALF_LOG_2PI - 0.5f * z1 * z1 - ls1 - HALF_LOG_2PI +
                               pick<N_CAT>(lp, ac) + pick<N_PRICE>(lp + N_CAT, ap) + pick<N_OFF>(lp + N_CAT + N_PRICE, ao);
            const float A = adv[i], ratio = __expf(logp - logp_old[i]);
            const float un = ratio * A, cl = fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip) * A;
            pg -= fminf(un, cl);
            const float g_logp = (un <= cl) ? -un * invB : 0.0f;             // d(-min) / d logp: the unclipped branch carries the gradient
            const float dv = val - ret[i];
            vl += dv * dv;
            dval += 2.0f * vf_coef * dv * invB;
            en += ent;
            G += g_logp;
            // d logp / d l_j = [j == a] - p_j : the one-hot part here, the -p_j part once per row below
            #pragma unroll
            for (int j = 0; j < N_CAT; j++) d[j] += (j == ac) ? g_logp : 0.0f;
            #pragma unroll
            for (int j = 0; j < N_PRICE; j++) d[N_CAT + j] += (j == ap) ? g_logp : 0.0f;
            #pragma unroll
            for (int j = 0; j < N_OFF; j++) d[N_CAT + N_PRICE + j] += (j == ao) ? g_logp : 0.0f;
            d[22] += g_logp * z0 * is0;                                      // d logp / d mu = z exp(-log_std)
            d[23] += g_logp * z1 * is1;
            dls0 += g_logp * (z0 * z0 - 1.0f) - es;                          // d logp / d log_std = z^2 - 1 ; d ent / d log_std = 1
            dls1 += g_logp * (z1 * z1 - 1.0f) - es;
        }
        // d ent / d l_j = -p_j (log p_j + ent_head), once per sample of the row
        const float esA = es * (float)agents;
        #pragma unroll
        for (int j = 0; j < N_CAT; j++) d[j] += -G * p[j] + esA * p[j] * (lp[j] + h0);
        #pragma unroll
        for (int j = 0; j < N_PRICE; j++) d[N_CAT + j] += -G * p[N_CAT + j] + esA * p[N_CAT + j] * (lp[N_CAT + j] + h1);
        #pragma unroll
        for (int j = 0; j < N_OFF; j++) d[N_CAT + N_PRICE + j] += -G * p[N_CAT + N_PRICE + j] + esA * p[N_CAT + N_PRICE + j] * (lp[N_CAT + N_PRICE + j] + h2);
        float4* dp4 = reinterpret_cast<float4*>(d_logits + r * stride);
        #pragma unroll
        for (int q = 0; q < N_LOGITS / 4; q++) dp4[q] = make_float4(d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]);
        if (packed) {                                                                        // the gradient of the whole padded row: value in column 24, zeros behind it
            dp4[N_LOGITS / 4] = make_float4(dval, 0.0f, 0.0f, 0.0f);
            for (int q = N_LOGITS / 4 + 1; q < stride / 4; q++) dp4[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        } else d_value[r] = dval;
    }
    // block reduction (wave shuffles, then one LDS round), one double atomic per block and quantity
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
// sums[0..4] = sum(-surrogate), sum((v - ret)^2), sum(entropy), d loss / d log_std[0..1] -> out[0..5] = pg, v, ent means, loss, d log_std
__global__ void k_ppo_finish(const double* sums, long long B, float vf_coef, float ent_coef, float* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const double pg = sums[0] / (double)B, vl = sums[1] / (double)B, en = sums[2] / (double)B;
        out[0] = (float)pg; out[1] = (float)vl; out[2] = (float)en; out[3] = (float)(pg + (double)vf_coef * vl - (double)ent_coef * en);
        out[4] = (float)sums[3]; out[5] = (float)sums[4];
    }
}
// ---- the rollout's policy step: sample the Dict action of every (market, agent) row from the network outputs ----------------
// One thread per row: three categorical heads by inverse CDF on the softmax, two Gaussian heads by Box-Muller, the action's
// log-probability, and the env's five action tensors (size_mean = tanh, size_sigma = sigmoid of the Gaussian samples, the Box
// bounds of action_helper.py:126-138) - instead of the ~50 launches torch.distributions needs for the same (Categorical.sample's
// multinomial with its device-side asserts, log_prob gathers, Normal, the squashing and casts).  Randomness: the counter-based
// splitmix64 generator of include/cda_random_agents.h keyed (seed, draw counter, row): reproducible, no generator state.
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float u01(unsigned long long w, int half) {               // (0, 1): 24 bits of one 32-bit half
    const unsigned int x = half ? (unsigned int)(w >> 32) : (unsigned int)w;
    return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f);
}
template <int N>
__device__ __forceinline__ int sample_head(const float* l, float u, float& logp) {
    float mx = l[0];
    #pragma unroll
    for (int j = 1; j < N; j++) mx = fmaxf(mx, l[j]);
    float e[N], s = 0.0f;
    #pragma unroll
    for (int j = 0; j < N; j++) { e[j] = __expf(l[j] - mx); s += e[j]; }
    const float t = u * s;
    float c = 0.0f;
    int a = N - 1;
    #pragma unroll
    for (int j = N - 1; j >= 0; j--) { }                                               // (keeps the unroller honest about N)
    bool found = false;
    #pragma unroll
    for (int j = 0; j < N; j++) { c += e[j]; if (!found && t < c) { a = j; found = true; } }
    logp += l[a] - mx - __logf(s);
    return a;
}
__global__ __launch_bounds__(256) void k_policy_sample(const float* __restrict__ logits, const float* __restrict__ log_std, long long B, int agents, int stride,
                                                       float* __restrict__ value_out, unsigned long long seed, const long long* __restrict__ counter,
                                                       long long* __restrict__ a_cat, long long* __restrict__ a_price, long long* __restrict__ a_off,
                                                       float* __restrict__ a_cont, float* __restrict__ logp_out,
                                                       int* __restrict__ env_cat, float* __restrict__ env_mean, float* __restrict__ env_sigma,
                                                       int* __restrict__ env_price, int* __restrict__ env_off) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    float l[N_LOGITS];
    const long long row = i / agents;
    const float4* lp4 = reinterpret_cast<const float4*>(logits + row * stride);                 // the row's samples share its logits
    if (value_out && i == row * agents) value_out[row] = logits[row * stride + N_LOGITS];       // packed rows: the value, made contiguous on the way
    #pragma unroll
    for (int q = 0; q < N_LOGITS / 4; q++) { const float4 v = lp4[q]; l[4 * q] = v.x; l[4 * q + 1] = v.y; l[4 * q + 2] = v.z; l[4 * q + 3] = v.w; }
    const unsigned long long key = mix64(seed + (unsigned long long)counter[0] * 0xd1342543de82ef95ull);
    const unsigned long long w0 = mix64(key + (unsigned long long)i), w1 = mix64(w0), w2 = mix64(w1);
    float logp = 0.0f;
    const int c = sample_head<N_CAT>(l, u01(w0, 0), logp);
    const int p = sample_head<N_PRICE>(l + N_CAT, u01(w0, 1), logp);
    const int o = sample_head<N_OFF>(l + N_CAT + N_PRICE, u01(w1, 0), logp);
    // Box-Muller: two independent standard normals from two uniforms
    const float r = sqrtf(-2.0f * __logf(u01(w1, 1))), th = 6.283185307179586f * u01(w2, 0);
    const float n0 = r * __cosf(th), n1 = r * __sinf(th);
    const float ls0 = log_std[0], ls1 = log_std[1];
    const float x0 = l[22] + __expf(ls0) * n0, x1 = l[23] + __expf(ls1) * n1;
    const float HALF_LOG_2PI = 0.918938533204672742f;
    logp += -0.5f * n0 * n0 - ls0 - HALF_LOG_2PI - 0.5f * n1 * n1 - ls1 - HALF_LOG_2PI;
    a_cat[i] = c; a_price[i] = p; a_off[i] = o;
    a_cont[2 * i] = x0; a_cont[2 * i + 1] = x1;
    logp_out[i] = logp;
    env_cat[i] = c; env_price[i] = p; env_off[i] = o;
    env_mean[i] = tanhf(x0);
    env_sigma[i] = 1.0f / (1.0f + __expf(-x1));
}
// Generalised advantage estimation (the recursion of ppo.gae): one thread per sample column walks its T steps backwards.
// rew / val / done f32[T, B] (done = 1 where the episode ended WITH that step), last_val f32[B] -> adv, ret f32[T, B].
__global__ __launch_bounds__(256) void k_gae(const float* __restrict__ rew, const float* __restrict__ val, const float* __restrict__ last_val,
                                             const float* __restrict__ done, int T, long long B, float gamma, float lam,
                                             float* __restrict__ adv, float* __restrict__ ret) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    float nxt = last_val[i], run = 0.0f;
    for (int t = T - 1; t >= 0; t--) {
        const long long k = (long long)t * B + i;
        const float nd = 1.0f - done[k], v = val[k];
        const float delta = rew[k] + gamma * nxt * nd - v;
        run = delta + gamma * lam * nd * run;
        adv[k] = run; ret[k] = run + v;
        nxt = v;
    }
}
// Rollout buffers: item k's `bytes` bytes go to slot *slot of its [T, bytes] buffer - every per-step tensor of a rollout in ONE launch,
// the step index read on the device (a captured HIP graph replays it unchanged).  blockIdx.y = item.
struct SlotItems { const unsigned char* src[CDA_SLOT_ITEMS_MAX]; unsigned char* dst[CDA_SLOT_ITEMS_MAX]; long long bytes[CDA_SLOT_ITEMS_MAX]; int n; };
__global__ __launch_bounds__(256) void k_store_slots(SlotItems it, const long long* __restrict__ slot) {
    const int k = (int)blockIdx.y;
    if (k >= it.n) return;
    const long long nb = it.bytes[k], t = slot[0];
    const unsigned char* s = it.src[k];
    unsigned char* d = it.dst[k] + t * nb;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long long)gridDim.x * blockDim.x;
    if ((((unsigned long long)s | (unsigned long long)d | (unsigned long long)nb) & 15ull) == 0) {
        const uint4* s4 = reinterpret_cast<const uint4*>(s);
        uint4* d4 = reinterpret_cast<uint4*>(d);
        for (long long i = tid; i < (nb >> 4); i += nth) d4[i] = s4[i];
    } else {
        for (long long i = tid; i < nb; i += nth) d[i] = s[i];
    }
}
__global__ void k_bump(long long* counter) { if (threadIdx.x == 0 && blockIdx.x == 0) counter[0] += 1; }
}  // namespace

extern "C" int cda_policy_sample(const float* logits, int32_t logits_stride, float* value_out, const float* log_std, int64_t rows, int32_t agents_per_row,
                                 uint64_t seed, int64_t* counter_dev,
                                 int64_t* a_cat, int64_t* a_price, int64_t* a_off, float* a_cont, float* logp,
                                 int32_t* env_category, float* env_size_mean, float* env_size_sigma, int32_t* env_price, int32_t* env_price_offset, void* stream) {
    if (!logits || !log_std || !counter_dev || !a_cat || !a_price || !a_off || !a_cont || !logp || !env_category || !env_size_mean || !env_size_sigma ||
        !env_price || !env_price_offset || rows < 1 || agents_per_row < 1 || agents_per_row > CDA_MAX_AGENTS) return CDA_ERR_INVALID;
    if (logits_stride < N_LOGITS || (logits_stride & 3) || (value_out && logits_stride <= N_LOGITS)) return CDA_ERR_INVALID;
    const long long samples = (long long)rows * agents_per_row;
    hipLaunchKernelGGL(k_policy_sample, dim3((unsigned)((samples + 255) / 256)), dim3(256), 0, (hipStream_t)stream, logits, log_std, samples, (int)agents_per_row, (int)logits_stride,
                       value_out, (unsigned long long)seed, (const long long*)counter_dev, (long long*)a_cat, (long long*)a_price, (long long*)a_off, a_cont, logp,
                       env_category, env_size_mean, env_size_sigma, env_price, env_price_offset);
    hipLaunchKernelGGL(k_bump, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long*)counter_dev);       // the next call draws fresh numbers (graph replays too)
    return hipGetLastError() == hipSuccess ? CDA_OK : CDA_ERR_HIP;
}

extern "C" int cda_ppo_loss(const float* logits, const float* value, const float* log_std, const int64_t* a_cat, const int64_t* a_price,
                            const int64_t* a_off, const float* a_cont, const float* logp_old, const float* adv, const float* ret, const int64_t* row_index,
                            int64_t rows, int32_t agents_per_row, int32_t out_stride, float clip, float vf_coef, float ent_coef, float* d_logits, float* d_value, double* sums5, float* out6, void* stream) {
    if (!logits || !log_std || !a_cat || !a_price || !a_off || !a_cont || !logp_old || !adv || !ret || !d_logits || !sums5 || !out6 || rows < 1 ||
        agents_per_row < 1 || agents_per_row > CDA_MAX_AGENTS) return CDA_ERR_INVALID;
    const int packed = out_stride != 0;                       // one padded [rows, out_stride] matrix holds logits | value | zeros, and so does its gradient
    if (packed ? (out_stride <= N_LOGITS || (out_stride & 3)) : (!value || !d_value)) return CDA_ERR_INVALID;
    if (hipMemsetAsync(sums5, 0, 5 * sizeof(double), (hipStream_t)stream) != hipSuccess) return CDA_ERR_HIP;
    hipLaunchKernelGGL(k_ppo_loss, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, logits, value, log_std, (const long long*)a_cat,
                       (const long long*)a_price, (const long long*)a_off, a_cont, logp_old, adv, ret, (const long long*)row_index, (long long)rows, (int)agents_per_row,
                       packed ? (int)out_stride : N_LOGITS, packed, clip, vf_coef, ent_coef, d_logits, d_value, sums5);
    hipLaunchKernelGGL(k_ppo_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)sums5, (long long)rows * agents_per_row, vf_coef, ent_coef, out6);
    return hipGetLastError() == hipSuccess ? CDA_OK : CDA_ERR_HIP;
}

extern "C" int cda_gae(const float* rew, const float* val, const float* last_val, const float* done, int32_t n_steps, int64_t batch,
                       float gamma, float lam, float* adv, float* ret, void* stream) {
    if (!rew || !val || !last_val || !done || !adv || !ret || n_steps < 1 || batch < 1) return CDA_ERR_INVALID;
    hipLaunchKernelGGL(k_gae, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rew, val, last_val, done, (int)n_steps, (long long)batch,
                       gamma, lam, adv, ret);
    return hipGetLastError() == hipSuccess ? CDA_OK : CDA_ERR_HIP;
}

extern "C" int cda_store_slots(int32_t n_items, const void* const* src, void* const* dst_base, const int64_t* bytes, int64_t* slot_dev, int32_t bump, void* stream) {
    if (n_items < 1 || n_items > CDA_SLOT_ITEMS_MAX || !src || !dst_base || !bytes || !slot_dev) return CDA_ERR_INVALID;
    SlotItems it;
    long long largest = 0;
    for (int k = 0; k < n_items; k++) {
        if (!src[k] || !dst_base[k] || bytes[k] < 1) return CDA_ERR_INVALID;
        it.src[k] = (const unsigned char*)src[k]; it.dst[k] = (unsigned char*)dst_base[k]; it.bytes[k] = (long long)bytes[k];
        largest = bytes[k] > largest ? bytes[k] : largest;
    }
    it.n = n_items;
    long long blocks = (largest / 16 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks);
    hipLaunchKernelGGL(k_store_slots, dim3((unsigned)blocks, (unsigned)n_items), dim3(256), 0, (hipStream_t)stream, it, (const long long*)slot_dev);
    if (bump) hipLaunchKernelGGL(k_bump, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long*)slot_dev);
    return hipGetLastError() == hipSuccess ? CDA_OK : CDA_ERR_HIP;
}
