// kernel_batch.hpp -- llda_sweep_batch_kernel: one sweep over MANY independent small problems in one launch
// Part of the single translation unit llda_gibbs.hip (included in order; see the contents list there).
#pragma once

namespace {

// ---------------------------------------------------------------------------------------------
// CascadeLDA's ensemble (/root/reference/CascadeLDA.py:135-184) is 122 independent Labeled-LDA problems of 6 ..
// 4171 documents and 2 .. 20 topics each; one launch per problem leaves the chip empty (a 6-document problem is
// one wavefront).  Here every "document instance" (document d as a member of sub-problem p) is a document of ONE
// launch; what differs per problem -- where its n_kw / n_k live, its row length KP, its RNG stream -- is looked up
// per instance.  All problems share the vocabulary (same V, so the same V*beta) and the corpus.
// The per-site arithmetic is the sparse kernel's (one lane per ALLOWED topic, sparse_site<>): the draw is decided
// from unnormalised fp64 prefix sums with a 2^-40 margin, which provably picks the exact pipeline's topic
// (DESIGN.md 4.3).  A site it cannot decide (~1e-11 per site) is resolved on the spot by exact_site_wave(), the
// reference's fp64 pipeline in the dense layout of the instance's own problem (every problem has at most 128 topics:
// one numpy pairwise leaf, 8 lanes) -- so the ensemble's result is the reference's either way.
// Count changes: int32 atomics on the delta image of the fused [n_kw | n_k] buffers of all problems.
// ---------------------------------------------------------------------------------------------
struct BParams {
    const int64_t *inst_off;     // [I+1] site offsets of the instances
    const int32_t *order;        // [n_inst] instances of this launch
    int64_t n_inst;
    const int32_t *word, *freq;  // [S] per instance site
    int32_t *z;                  // [S] device positions (in the instance's problem layout)
    const int32_t *inst_prob;    // [I] problem of the instance
    const int32_t *inst_doc;     // [I] index of the document inside its problem (RNG counter word 1)
    const int64_t *live_off;     // [I+1]
    const int32_t *live_pos;     // allowed positions in draw order: ascending (lane, slot), not memory position
    const int64_t *ndk_off;      // [I] offset of the instance's n_dk row
    int32_t *n_dk;
    const int64_t *kw_off;       // [P] offset of the problem's n_kw (V x KP) in counts / delta
    const int64_t *nk_off;       // [P] offset of the problem's n_k (KP)
    const int32_t *kp;           // [P] row length
    const int32_t *prob_stream;  // [P] RNG stream of the problem (its index in the ensemble's visiting order)
    const int32_t *k;            // [P] number of topics of the problem (<= 128)
    const int32_t *counts;       // sweep-start snapshot of all problems
    int32_t *delta;              // += sweep changes
    int32_t *status;
    double alpha, beta, vbeta, margin_rel;
    uint32_t key0, key1, sweep;
};

// ... or one problem per lane group (at most 128 topics: one pairwise leaf, 8 lanes x KP/8 slots); the layout of the
// group that is served comes from its lanes
__device__ __noinline__ int exact_call(const BParams *, int layK, int layKP, double wx, int pos, int base, int A, double u,
                                       int lane)
{
    const int K = __builtin_amdgcn_readlane(layK, base), KP = __builtin_amdgcn_readlane(layKP, base);
    return exact_site_wave(wx, pos, base, A, u, 8, KP >> 3, 0, K & 7, K >> 3, 0, 1, nullptr, lane);
}

template <int GS>
__global__ void __launch_bounds__(256) llda_sweep_batch_kernel(const BParams P)
{
    constexpr int GPB = 256 / GS;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int lig = tid & (GS - 1);
    const int grp = tid / GS;
    const int gbase = lane & ~(GS - 1);
    const uint64_t gmask = (GS == 64) ? ~0ull : ((1ull << GS) - 1ull);

    // wave-uniform control flow (a lane group without an instance, or with a shorter document, idles through flags):
    // the exact tier of sparse_site() needs all 64 lanes
    const int64_t idx = (int64_t)blockIdx.x * GPB + grp;
    const bool valid = idx < P.n_inst;
    const int64_t inst = valid ? P.order[idx] : P.order[0];
    const int64_t s0 = P.inst_off[inst];
    const int len = valid ? (int)(P.inst_off[inst + 1] - s0) : 0;
    const int prob = P.inst_prob[inst];
    const int KP = P.kp[prob];
    const int32_t *n_kw = P.counts + P.kw_off[prob];
    int32_t *d_kw = P.delta + P.kw_off[prob];
    const int64_t l0 = P.live_off[inst];
    const int A = (int)(P.live_off[inst + 1] - l0);
    const bool live = valid && len > 0 && lig < A;
    const int pos = live ? P.live_pos[l0 + lig] : -1;
    int32_t *ndk_p = P.n_dk + P.ndk_off[inst] + (live ? pos : 0);
    int ndk = live ? *ndk_p : 0;
    const int ndk0 = ndk;
    int nk = live ? P.counts[P.nk_off[prob] + pos] : 0;
    const uint32_t gdoc = (uint32_t)P.inst_doc[inst];
    const uint32_t stream = (uint32_t)P.prob_stream[prob];
    const int layK = P.k[prob];                  // (for the exact tier: the dense layout of THIS group's problem)
    int max_len = len;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) max_len = max(max_len, __shfl_xor(max_len, o, 64));

    for (int n0 = 0; n0 < max_len; n0 += 8) {
        const int nb = max(0, min(8, len - n0));
        const int jj = lig & 7;
        int sv = 0, sf = 0, sz = 0;
        if (nb > 0) {
            const int64_t si = s0 + n0 + (jj < nb ? jj : nb - 1);
            sv = P.word[si]; sf = P.freq[si]; sz = P.z[si];
        }
        int su_lo, su_hi;
        {   // keyed uniform of site n0+jj: Philox block (site >> 1), words (0,1) / (2,3) by parity
            const int n = n0 + jj;
            uint32_t c0 = (uint32_t)(n >> 1), c1 = gdoc, c2 = stream, c3 = P.sweep;
            philox4x32_10(c0, c1, c2, c3, P.key0, P.key1);
            const uint32_t ra = (n & 1) ? c2 : c0, rb = (n & 1) ? c3 : c1;
            const double u = ((double)(ra >> 5) * 67108864.0 + (double)(rb >> 6)) * (1.0 / 9007199254740992.0);
            su_lo = __double2loint(u); su_hi = __double2hiint(u);
        }
        const int w0 = bcast_lane<GS, 0>(sv, lig), w1 = bcast_lane<GS, 1>(sv, lig), w2 = bcast_lane<GS, 2>(sv, lig),
                  w3 = bcast_lane<GS, 3>(sv, lig), w4 = bcast_lane<GS, 4>(sv, lig), w5 = bcast_lane<GS, 5>(sv, lig),
                  w6 = bcast_lane<GS, 6>(sv, lig), w7 = bcast_lane<GS, 7>(sv, lig);
        int xg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (live && nb > 0) {
            const int32_t *col = n_kw + pos;
            xg[0] = col[(int64_t)w0 * KP]; xg[1] = col[(int64_t)w1 * KP]; xg[2] = col[(int64_t)w2 * KP];
            xg[3] = col[(int64_t)w3 * KP]; xg[4] = col[(int64_t)w4 * KP]; xg[5] = col[(int64_t)w5 * KP];
            xg[6] = col[(int64_t)w6 * KP]; xg[7] = col[(int64_t)w7 * KP];
        }
        int my_zn = sz;
        int done = 0;
#define LLDA_BATCH_SITE(J)                                                                                     \
        sparse_site<GS, J, false>(P, layK, KP, nb, sf, sz, su_lo, su_hi, 0, xg, live, pos, A, ndk, nk, my_zn, done, lig, lane, \
                           gbase, gmask);
        LLDA_BATCH_SITE(0) LLDA_BATCH_SITE(1) LLDA_BATCH_SITE(2) LLDA_BATCH_SITE(3)
        LLDA_BATCH_SITE(4) LLDA_BATCH_SITE(5) LLDA_BATCH_SITE(6) LLDA_BATCH_SITE(7)
#undef LLDA_BATCH_SITE
        if (lig < 8 && lig < done) {
            P.z[s0 + n0 + lig] = my_zn;
            if (my_zn != sz) {
                int32_t *row = d_kw + (int64_t)sv * KP;
                atomicAdd(row + sz, -sf);
                atomicAdd(row + my_zn, sf);
            }
        }
    }
    if (live) {
        *ndk_p = ndk;
        if (ndk != ndk0) atomicAdd(P.delta + P.nk_off[prob] + pos, ndk - ndk0);
    }
}

}  // namespace
