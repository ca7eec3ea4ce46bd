// kernel_readout.hpp -- llda_loglik_kernel, llda_readout_phi_kernel, llda_readout_theta_kernel
// Part of the single translation unit llda_gibbs.hip (included in order; see the contents list there).
#pragma once

namespace {

// ---------------------------------------------------------------------------------------------
// log-likelihood read-out (LabeledLDA.py:231-239, 256-265), same group layout
// ---------------------------------------------------------------------------------------------
struct LParams {
    const int64_t *doc_off;
    const int32_t *word;
    const uint16_t *lab_mask;
    const int32_t *n_dk;
    const int32_t *n_kw;
    const int32_t *n_k;
    double *out_doc;
    int64_t D;
    double alpha, beta, vbeta;
};

template <int G>
__device__ __forceinline__ double group_allsum(double x)
{
#pragma unroll
    for (int d = 1; d < G; d <<= 1) x = x + __shfl_xor(x, d, G);
    return x;
}

template <int G, int T>
__global__ void __launch_bounds__(256) llda_loglik_kernel(const LParams P)
{
    constexpr int KP = G * T;
    constexpr int GPB = 256 / G;
    const int tid = threadIdx.x;
    const int lig = tid & (G - 1);
    const int64_t d = (int64_t)blockIdx.x * GPB + tid / G;
    if (d >= P.D) return;
    int ndk[T], nk[T];
    load_lane_row<G, T>(P.n_dk + d * KP, lig, ndk);
    load_lane_row<G, T>(P.n_k, lig, nk);
    const uint32_t mask = P.lab_mask[d * G + lig];
    double th[T], rden[T], rs = 0.0;
#pragma unroll
    for (int s = 0; s < T; ++s) {
        th[s] = (double)ndk[s] + (((mask >> s) & 1u) ? P.alpha : 0.0);     // n_d_k + labs*alpha
        rs = rs + th[s];
        rden[s] = (double)nk[s] + P.vbeta;
    }
    rs = group_allsum<G>(rs);
#pragma unroll
    for (int s = 0; s < T; ++s) th[s] = th[s] / rs;
    double acc = 0.0;
    for (int64_t i = P.doc_off[d]; i < P.doc_off[d + 1]; ++i) {
        int x[T];
        load_lane_row<G, T>(P.n_kw + (int64_t)P.word[i] * KP, lig, x);
        double dot = 0.0;
#pragma unroll
        for (int s = 0; s < T; ++s) dot = dot + th[s] * (((double)x[s] + P.beta) / rden[s]);
        dot = group_allsum<G>(dot);
        acc = acc - log(dot);
    }
    if (lig == 0) P.out_doc[d] = acc;
}


// ---------------------------------------------------------------------------------------------
// Thinning read-outs (LabeledLDA.py:131-153, 231-239; CascadeLDA.py:394-395, 423-434): phi / theta of
// the current counts and their running means, written in the reference's (K, V) / (D, K) layout.
// ---------------------------------------------------------------------------------------------
struct RParams {
    const int32_t *n_kw, *n_k, *n_dk;
    const double *den;
    const uint16_t *lab_mask;
    double *out;
    int32_t *flags;
    int64_t V, D;
    int32_t K, KP, T, mode;
    double alpha, beta, vbeta, keep, share;
    int32_t leaf_start[LLDA_MAX_WIDE_LEAVES], leaf_len[LLDA_MAX_WIDE_LEAVES];
    int32_t last_leaf, tail, tail_row, n_rounds, xor_tree;
    uint32_t rounds_pk[LLDA_MAX_ROUNDS];
};

// topic held by a device position, -1 for padding (inverse of llda_layout.topic_pos)
__device__ __forceinline__ int topic_of_position(const RParams &P, int pos)
{
    int g, slot;
    lane_slot_of_rt(P.KP / P.T, P.T, pos, g, slot);
    const int leaf = g >> 3, rel = (g & 7) + 8 * slot;
    return rel < P.leaf_len[leaf] ? P.leaf_start[leaf] + rel : -1;
}

__device__ __forceinline__ double running_mean(const RParams &P, double old, double cur)
{
    if (P.mode == 0) return cur;
    const double a = P.keep * old, b = P.share * cur;      // two roundings, then the sum (no FMA)
    return a + b;
}

// One workgroup per 64 words: 64 x 64 (word, position) tiles of n_kw go through LDS so that both the
// word-major reads and the topic-major writes are contiguous.
__global__ void __launch_bounds__(256) llda_readout_phi_kernel(const RParams P)
{
    __shared__ int s_tile[64][65];
    __shared__ int s_seen[64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t v0 = (int64_t)blockIdx.x * 64;
    if (tid < 64) s_seen[tid] = 0;
    int bad = 0, seen = 0;
    const int64_t v = v0 + lane;
    for (int c0 = 0; c0 < P.KP; c0 += 64) {
        __syncthreads();
        for (int r = w; r < 64; r += 4)
            if (v0 + r < P.V && c0 + lane < P.KP) s_tile[r][lane] = P.n_kw[(v0 + r) * P.KP + c0 + lane];
        __syncthreads();
        for (int j = w; j < 64 && c0 + j < P.KP; j += 4) {
            const int k = topic_of_position(P, c0 + j);
            if (k < 0 || v >= P.V) continue;
            const double den = P.den ? P.den[c0 + j] : (double)P.n_k[c0 + j] + P.vbeta;
            const double cur = ((double)s_tile[lane][j] + P.beta) / den;
            double *o = P.out + (int64_t)k * P.V + v;
            const double val = running_mean(P, P.mode ? *o : 0.0, cur);
            *o = val;
            if (val < 0.0) bad |= LLDA_READOUT_NEGATIVE;
            if (val != val) bad |= LLDA_READOUT_NAN;
            if (val != 0.0) seen = 1;
        }
    }
    if (seen) atomicOr(&s_seen[lane], 1);
    __syncthreads();
    if (tid < 64 && v0 + tid < P.V && !s_seen[tid]) bad |= LLDA_READOUT_NO_LOAD;
    if (bad && P.flags) atomicOr(P.flags, bad);
}

template <int G, int T, bool HAS_TAIL>
__global__ void __launch_bounds__(256) llda_readout_theta_kernel(const RParams P)
{
    constexpr int KP = G * T;
    constexpr int GPB = 256 / G;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int lig = tid & (G - 1);
    const int64_t d = (int64_t)blockIdx.x * GPB + tid / G;
    if (d >= P.D) return;
    KParams K;                                     // the summation schedule group_sum() reads
    K.last_leaf = P.last_leaf; K.tail = P.tail; K.tail_row = P.tail_row; K.n_rounds = P.n_rounds;
    K.xor_tree = P.xor_tree;
#pragma unroll
    for (int r = 0; r < LLDA_MAX_ROUNDS; ++r) K.rounds_pk[r] = P.rounds_pk[r];
    int ndk[T];
    load_lane_row<G, T>(P.n_dk + d * KP, lig, ndk);
    const uint32_t mask = P.lab_mask[d * G + lig];
    double num[T];
#pragma unroll
    for (int s = 0; s < T; ++s) num[s] = (double)ndk[s] + (((mask >> s) & 1u) ? P.alpha : 0.0);   // n_d_k + labs*alpha
    const double rs = group_sum<G, T, HAS_TAIL>(num, K, lig, lane);                              // np.sum, axis 1
#pragma unroll
    for (int s = 0; s < T; ++s) {
        const int k = topic_of_position(P, pos_of<G, T>(lig, s));
        if (k < 0) continue;
        double *o = P.out + d * P.K + k;
        *o = running_mean(P, P.mode ? *o : 0.0, num[s] / rs);
    }
}

}  // namespace
