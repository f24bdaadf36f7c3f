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

template <int N>
__device__ __forceinline__ void head(const float* l, int a, float g_logp, float ent_scale, float* d, float& logp, float& ent) {
    float mx = l[0];
    #pragma unroll
    for (int j = 1; j < N; j++) mx = fmaxf(mx, l[j]);
    float e[N], s = 0.0f;
    #pragma unroll
    for (int j = 0; j < N; j++) { e[j] = __expf(l[j] - mx); s += e[j]; }
    const float ls = __logf(s), inv = 1.0f / s;
    float h = 0.0f;
    #pragma unroll
    for (int j = 0; j < N; j++) { const float p = e[j] * inv, lp = l[j] - mx - ls; h -= p * lp; }
    #pragma unroll
    for (int j = 0; j < N; j++) {
        const float p = e[j] * inv, lp = l[j] - mx - ls;
        // d logp / d l_j = [j == a] - p_j ;  d ent / d l_j = -p_j (log p_j + ent)
        d[j] = g_logp * ((j == a ? 1.0f : 0.0f) - p) + ent_scale * p * (lp + h);
    }
    a = a < 0 ? 0 : (a >= N ? N - 1 : a);
    logp += l[a] - mx - ls;
    ent += h;
}

__global__ __launch_bounds__(256) void k_ppo_loss(const float* __restrict__ logits, const float* __restrict__ value, const float* __restrict__ log_std,
                                                  const long long* __restrict__ a_cat, const long long* __restrict__ a_price, const long long* __restrict__ a_off,
                                                  const float* __restrict__ a_cont, const float* __restrict__ logp_old, const float* __restrict__ adv,
                                                  const float* __restrict__ ret, long long B, float clip, float vf_coef, float ent_coef,
                                                  float* __restrict__ d_logits, float* __restrict__ d_value, double* __restrict__ sums) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const float invB = 1.0f / (float)B;
    float pg = 0.0f, vl = 0.0f, en = 0.0f, dls0 = 0.0f, dls1 = 0.0f;
    if (i < B) {
        float l[N_LOGITS], d[N_LOGITS];
        const float4* lp4 = reinterpret_cast<const float4*>(logits + i * N_LOGITS);          // 96 B per sample, 16-byte aligned
        #pragma unroll
        for (int q = 0; q < N_LOGITS / 4; q++) { const float4 v = lp4[q]; l[4 * q] = v.x; l[4 * q + 1] = v.y; l[4 * q + 2] = v.z; l[4 * q + 3] = v.w; }
        const float ls0 = log_std[0], ls1 = log_std[1];
        const float is0 = __expf(-ls0), is1 = __expf(-ls1);
        const float z0 = (a_cont[2 * i] - l[22]) * is0, z1 = (a_cont[2 * i + 1] - l[23]) * is1;
        const float HALF_LOG_2PI = 0.918938533204672742f;
        // first pass for logp (the surrogate's gradient factor needs the ratio): heads without gradients ...
        float logp = -0.5f * z0 * z0 - ls0 - HALF_LOG_2PI - 0.5f * z1 * z1 - ls1 - HALF_LOG_2PI, ent = 0.0f;
        {
            float dummy[N_CAT + N_PRICE + N_OFF];
            head<N_CAT>(l, (int)a_cat[i], 0.0f, 0.0f, dummy, logp, ent);
            head<N_PRICE>(l + N_CAT, (int)a_price[i], 0.0f, 0.0f, dummy + N_CAT, logp, ent);
            head<N_OFF>(l + N_CAT + N_PRICE, (int)a_off[i], 0.0f, 0.0f, dummy + N_CAT + N_PRICE, logp, ent);
        }
        ent += 1.0f + 2.0f * HALF_LOG_2PI + ls0 + ls1;                       // two Gaussian heads: 1/2 + log(2 pi)/2 + log_std each
        const float A = adv[i], ratio = __expf(logp - logp_old[i]);
        const float un = ratio * A, cl = fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip) * A;
        pg = -fminf(un, cl);
        const float g_logp = (un <= cl) ? -un * invB : 0.0f;                // d(-min) / d logp: the unclipped branch carries the gradient
        const float dv = value[i] - ret[i];
        vl = dv * dv;
        en = ent;
        // ... second pass with the gradient factors known
        float lp2 = 0.0f, e2 = 0.0f;
        const float es = ent_coef * invB;                                    // loss has -ent_coef * mean(ent)
        head<N_CAT>(l, (int)a_cat[i], g_logp, es, d, lp2, e2);
        head<N_PRICE>(l + N_CAT, (int)a_price[i], g_logp, es, d + N_CAT, lp2, e2);
        head<N_OFF>(l + N_CAT + N_PRICE, (int)a_off[i], g_logp, es, d + N_CAT + N_PRICE, lp2, e2);
        d[22] = g_logp * z0 * is0;                                           // d logp / d mu = z exp(-log_std)
        d[23] = g_logp * z1 * is1;
        dls0 = g_logp * (z0 * z0 - 1.0f) - es;                               // d logp / d log_std = z^2 - 1 ; d ent / d log_std = 1
        dls1 = g_logp * (z1 * z1 - 1.0f) - es;
        float4* dp4 = reinterpret_cast<float4*>(d_logits + i * N_LOGITS);
        #pragma unroll
        for (int q = 0; q < N_LOGITS / 4; q++) dp4[q] = make_float4(d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]);
        d_value[i] = 2.0f * vf_coef * dv * invB;
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
__global__ __launch_bounds__(256) void k_policy_sample(const float* __restrict__ logits, const float* __restrict__ log_std, long long B,
                                                       unsigned long long seed, const long long* __restrict__ counter,
                                                       long long* __restrict__ a_cat, long long* __restrict__ a_price, long long* __restrict__ a_off,
                                                       float* __restrict__ a_cont, float* __restrict__ logp_out,
                                                       int* __restrict__ env_cat, float* __restrict__ env_mean, float* __restrict__ env_sigma,
                                                       int* __restrict__ env_price, int* __restrict__ env_off) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    float l[N_LOGITS];
    const float4* lp4 = reinterpret_cast<const float4*>(logits + i * N_LOGITS);
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
__global__ void k_bump(long long* counter) { if (threadIdx.x == 0 && blockIdx.x == 0) counter[0] += 1; }
}  // namespace

extern "C" int cda_policy_sample(const float* logits, const float* log_std, int64_t rows, uint64_t seed, int64_t* counter_dev,
                                 int64_t* a_cat, int64_t* a_price, int64_t* a_off, float* a_cont, float* logp,
                                 int32_t* env_category, float* env_size_mean, float* env_size_sigma, int32_t* env_price, int32_t* env_price_offset, void* stream) {
    if (!logits || !log_std || !counter_dev || !a_cat || !a_price || !a_off || !a_cont || !logp || !env_category || !env_size_mean || !env_size_sigma ||
        !env_price || !env_price_offset || rows < 1) return CDA_ERR_INVALID;
    hipLaunchKernelGGL(k_policy_sample, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, logits, log_std, (long long)rows,
                       (unsigned long long)seed, (const long long*)counter_dev, (long long*)a_cat, (long long*)a_price, (long long*)a_off, a_cont, logp,
                       env_category, env_size_mean, env_size_sigma, env_price, env_price_offset);
    hipLaunchKernelGGL(k_bump, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long*)counter_dev);       // the next call draws fresh numbers (graph replays too)
    return hipGetLastError() == hipSuccess ? CDA_OK : CDA_ERR_HIP;
}

extern "C" int cda_ppo_loss(const float* logits, const float* value, const float* log_std, const int64_t* a_cat, const int64_t* a_price,
                            const int64_t* a_off, const float* a_cont, const float* logp_old, const float* adv, const float* ret, int64_t batch,
                            float clip, float vf_coef, float ent_coef, float* d_logits, float* d_value, double* sums5, float* out6, void* stream) {
    if (!logits || !value || !log_std || !a_cat || !a_price || !a_off || !a_cont || !logp_old || !adv || !ret || !d_logits || !d_value || !sums5 || !out6 || batch < 1)
        return CDA_ERR_INVALID;
    if (hipMemsetAsync(sums5, 0, 5 * sizeof(double), (hipStream_t)stream) != hipSuccess) return CDA_ERR_HIP;
    hipLaunchKernelGGL(k_ppo_loss, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, (hipStream_t)stream, logits, value, log_std, (const long long*)a_cat,
                       (const long long*)a_price, (const long long*)a_off, a_cont, logp_old, adv, ret, (long long)batch, clip, vf_coef, ent_coef, d_logits, d_value, sums5);
    hipLaunchKernelGGL(k_ppo_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)sums5, (long long)batch, vf_coef, ent_coef, out6);
    return hipGetLastError() == hipSuccess ? CDA_OK : CDA_ERR_HIP;
}
