// kernel_oct.hpp -- llda_sweep_oct_kernel: the 16-bit-row kernel for the layouts of EIGHT lanes per document (K = 97 .. 128, llda_quad_ok)
// with EIGHT documents per wavefront at FOUR wavefronts per SIMD (round 6 experiment for BASELINE configs[2]).
// MEASURED SLOWER than llda_sweep_quad_kernel<2> (profiles/r06_quad_k128.md: 1.333 vs 1.197 ms on configs[2], bit-identical states):
// this file is committed once for the record and removed by the next commit.
// Part of the single translation unit llda_gibbs.hip (included behind kernel_quad.hpp, whose helpers it uses).
#pragma once

namespace {

// ---------------------------------------------------------------------------------------------
// Why.  llda_sweep_quad_kernel<2> walks a K = 128 document with 4 lanes x 32 slots, sixteen documents per wavefront: 15.8 vector
// instructions per site -- and the vector unit idle in half of its cycles at two wavefronts per SIMD (LDS: 32 slots x 8 bytes per lane;
// 214 VGPRs).  Here a lane walks ONE standard lane (16 slots: 128 bytes of LDS, 128 VGPRs): four wavefronts per SIMD.  The machinery
// is the quad kernel's at half the slots per lane:
//     device position     pos = i << 5 | g << 2 | c          g = standard lane, slot s = 4 i + c   (the layout's own memory order)
//     chains              A = slots 0 .. 7, B = slots 8 .. 15 of the lane: the draw order (lane, slot) is unchanged
//     register / LDS slot rho = 2 a + e: the pair (2a, 2a + 1) holds element a of chain A (slot a) and of chain B (slot 8 + a)
// Rows come from the SAME 16-bit image and per-word flags as the quad kernel's (llda_pack_rows16_all: piece (j, g & 1) of quad lane
// g >> 1 at (2 j + (g & 1)) * 64 + 16 (g >> 1) bytes -- the eight lanes of a document read a contiguous 128-byte block per load, two
// loads per site), the site records, the commit log, documents of at most 65 535 tokens.
// Tier 0: chains of 8 instead of 16 -- every prefix within (12 + 8) v of the lane's share, three scan steps: each term of the quad
// kernel's bound only shrinks, the same data-dependent margin (QM_*) is kept.  Tier 1 and the exact tier as there.
// ---------------------------------------------------------------------------------------------
constexpr int OT = 16;         // slots per lane
constexpr int ONT = 128;       // threads per workgroup: two wavefronts, sixteen documents
constexpr int OG = 8, OKP = OG * OT, OIS = 5;           // lanes per document, positions, shift of the slot chunk i in a position
constexpr uint32_t OKEY_NONE = 0xFC000u | 15u << 9 | (uint32_t)(OKP - 1);   // no slot above lo: the last slot of the last lane
#define OLDS(arr, rho, t) (arr)[(rho) >> 2][t][(rho) & 3]

constexpr int oct_rho_of_slot(int s) { return 2 * (s & 7) + (s >> 3); }
__device__ __forceinline__ int oct_rho(int pos)            // slot number of a device position
{
    // pos = i << 5 | g << 2 | c, s = 4 i + c:  rho = 2 (s & 7) + (s >> 3) = (i & 1) << 3 | c << 1 | i >> 1
    return ((pos >> 2) & 8) | ((pos & 3) << 1) | ((pos >> 6) & 1);
}

// the counts of one document for the cold tiers (the layout IS the standard one: G lanes x 16 slots)
struct OctCounts {
    const int (*s_ndk)[ONT][4];
    const int *s_nk0;
    int t, g;
    bool st;
    __device__ __forceinline__ int word(int s) const { return OLDS(s_ndk, oct_rho_of_slot(s), t); }
    __device__ __forceinline__ int nd(int s) const { return word(s) & 0xffff; }
    __device__ __forceinline__ int nk(int s) const
    {
        const int w = word(s);
        return s_nk0[pos_of<OG, 16>(g, s)] + (w & 0xffff) - (int)((uint32_t)w >> 16);
    }
    __device__ __forceinline__ bool stats() const { return st; }
    __device__ __forceinline__ bool count_unsure() const { return false; }
};

template <bool PAD>
__device__ __noinline__ int oct_cold(const int (*s_ndk)[ONT][4], const int *s_nk0, int tbase, int w, int f, int zo, uint32_t ra, uint32_t rb,
                                     int lane, int64_t d, const KParams *P)
{
    const int g = lane & (OG - 1);
    int x[16];
    gload_lane_row<OG, 16>(P->n_kw, (int64_t)w * OKP, g, x);
    int lo, so;
    lane_slot_of<OG, 16>(zo, lo, so);
#pragma unroll
    for (int s = 0; s < 16; ++s) x[s] -= (g == lo && s == so) ? f : 0;       // the site's own count (LabeledLDA.py:109-111)
    const OctCounts dc{s_ndk, s_nk0, tbase + g, g, lane < OG};
    if constexpr (PAD) return cold_tiers_acc<OG, 16, true, false>(dc, x, P->lab_mask[d * OG + g], uniform53(ra, rb), g, lane, P);
    else return cold_tiers_acc<OG, 16, false, true>(dc, x, 0xFFFFu, uniform53(ra, rb), g, lane, P);
}

typedef float o_v16f __attribute__((ext_vector_type(16)));

// Tier 0 for the eight documents of the wavefront at once (quad_draw with chains of 8).  Returns the ballot of the lanes that are not
// sure; zn = slot rho << 9 | position that every lane's document drew.
template <bool PAD>
__device__ __forceinline__ uint64_t oct_draw(const o_v16f &xv, const q_v2f (&pa)[8], float u, float margin_rel, float margin_data, float beta,
                                             int lg, int &zn)
{
    const q_v2f b2 = {beta, beta};
    q_v2f Q[8];
    LLDA_MARK("scores");
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const q_v2f x2 = {xv[2 * a], xv[2 * a + 1]};
        const q_v2f nb = x2 + b2;
        if (a == 0) Q[0] = nb * pa[0];
        else Q[a] = __builtin_elementwise_fma(nb, pa[a], Q[a - 1]);
    }
    LLDA_MARK("lane_scan");
    __builtin_amdgcn_s_setprio(QP_DEC);
    const float X0 = Q[7].x + Q[7].y;
    float X = X0, tot = X0;
    // two documents share a DPP row: a scan step adds 1.0 * the shifted value, or 0.0 * it in the first lanes of a document
    const float m1 = lg >= 1 ? 1.0f : 0.0f, m2 = lg >= 2 ? 1.0f : 0.0f, m4 = lg >= 4 ? 1.0f : 0.0f;
    X = __builtin_fmaf(dpp_f32<DPP_ROW_SHR + 1>(X), m1, X);
    X = __builtin_fmaf(dpp_f32<DPP_ROW_SHR + 2>(X), m2, X);
    X = __builtin_fmaf(dpp_f32<DPP_ROW_SHR + 4>(X), m4, X);
    tot += dpp_f32<DPP_HALF_MIRROR>(tot);
    tot += dpp_f32<DPP_XOR1>(tot);
    tot += dpp_f32<DPP_XOR2>(tot);
    const float prev = dpp_f32<DPP_ROW_SHR + 1>(X) * m1;
    LLDA_MARK("threshold");
    const float t = u * tot;
    const float tg = t - prev;
    const float md = __builtin_fmaf(QM_L, X0, __builtin_fmaf(QM_T, t, __builtin_fmaf(QM_P, prev, QM_TOT * tot)));
    const float margin = __builtin_fmaf(tot, margin_rel, margin_data * md);
    const float lo0 = tg - margin, hi0 = tg + margin;
    LLDA_MARK("search");
    const bool c0 = Q[7].x <= lo0;                                    // chain A or chain B?
    const float dA = c0 ? Q[7].x : 0.0f;
    const float lo = lo0 - dA, hi = hi0 - dA;
    float q[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) q[a] = c0 ? Q[a].y : Q[a].x;
    // branch-free binary search that keeps the smallest element found above lo
    const bool c5 = q[7] <= lo;
    float ub = c5 ? __int_as_float(0x7f800000) : q[7];
    const bool c1 = q[3] <= lo;
    ub = c1 ? ub : q[3];
    const float m2s = c1 ? q[5] : q[1];
    const float g0 = c1 ? q[4] : q[0], g2 = c1 ? q[6] : q[2];
    const bool c2 = m2s <= lo;
    ub = c2 ? ub : m2s;
    const float m3 = c2 ? g2 : g0;
    const bool c3 = m3 <= lo;
    ub = c3 ? ub : m3;
    uint64_t bad_total;
    asm("v_cmp_class_f32_e64 %0, %1, %2" : "=s"(bad_total) : "v"(tot - margin), "v"(0x2FF));
    uint64_t unsure = __ballot(!(ub > hi)) | bad_total;
    // key = lane << 14 | slot rho << 9 | position; slot s = 8 c0 + 4 c1 + 2 c2 + c3: chunk i = 2 c0 + c1, c = 2 c2 + c3; rho = 2 (s & 7) + c0
    LLDA_MARK("pick");
    const uint32_t p = (c0 ? (2u << OIS | 1u << 9) : 0u) | (c1 ? (1u << OIS | 8u << 9) : 0u) | (c2 ? (2u | 4u << 9) : 0u) | (c3 ? (1u | 2u << 9) : 0u) |
                       ((uint32_t)lg << 2) | ((uint32_t)lg << 14);
    uint32_t key = c5 ? OKEY_NONE : p;
    key = quad_min_key<3>(key);                                       // minimum over the eight lanes of the document
    zn = (int)(key & 0x3FFFu);
    if constexpr (PAD) unsure |= __ballot(key == OKEY_NONE);          // (kernel_quad.hpp: never name a position without a topic)
    return unsure;
}

// Tier 1 in this layout (quad_tier1 with 16 slots per lane): unnormalised fp64 prefix sums, margin 2^-40 of the total
template <bool PAD>
__device__ __forceinline__ uint64_t oct_tier1(const o_v16f &xv, const int (*s_ndk)[ONT][4], const int *s_nk0, int tid, int lg, double u,
                                              double alpha, double beta, double vbeta, double margin_rel, uint32_t vm, int &zn)
{
    double W[OT];
    double run = 0.0;
#pragma unroll
    for (int k = 0; k < OT; ++k) {                       // k = slot = position in the draw order of the lane
        const int rho = oct_rho_of_slot(k);
        const int w = OLDS(s_ndk, rho, tid);
        const int nd = w & 0xffff, nk = s_nk0[((k >> 2) << OIS) | (lg << 2) | (k & 3)] + nd - (int)((uint32_t)w >> 16);
        const double den = (double)nk + vbeta;
        double y = (double)__builtin_amdgcn_rcpf((float)den);
        y = __builtin_fma(__builtin_fma(-den, y, 1.0), y, y);
        y = __builtin_fma(__builtin_fma(-den, y, 1.0), y, y);
        const double ws = ((double)nd + alpha) * (((double)xv[rho] + beta) * y);
        run = run + (((vm >> k) & 1u) ? ws : 0.0);
        W[k] = run;
    }
    const double X = group_scan<OG>(run, lg);
    const double tot = bcast_last<OG>(X, tid & 63);
    double prev = dpp_f64<DPP_ROW_SHR + 1>(X);
    prev = lg ? prev : 0.0;
    const double tg = u * tot - prev;
    const double margin = tot * margin_rel;
    const double lo = tg - margin, hi = tg + margin;
    int cnt_lo = 0, cnt_hi = 0;
#pragma unroll
    for (int k = 0; k < OT; ++k) {
        cnt_lo += (W[k] <= lo) ? 1 : 0;
        cnt_hi += (W[k] <= hi) ? 1 : 0;
    }
    uint64_t unsure = __ballot((cnt_lo != cnt_hi) || !(tot > 0.0) || !(margin < tot));
    const uint32_t k = (uint32_t)cnt_lo;
    const uint32_t p = ((k >> 2) << OIS) | ((uint32_t)lg << 2) | (k & 3u) | ((2u * (k & 7u) + (k >> 3)) << 9) | ((uint32_t)lg << 14);
    uint32_t key = cnt_lo >= OT ? OKEY_NONE : p;
    key = quad_min_key<3>(key);
    zn = (int)(key & 0x3FFFu);
    if constexpr (PAD) unsure |= __ballot(key == OKEY_NONE);
    return unsure;
}

template <bool PAD = false>
__global__ void __launch_bounds__(ONT, 4) llda_sweep_oct_kernel(const KParams P)
{
    constexpr int G = OG, KP = OKP, IS = OIS, DPW = 64 / OG;
    constexpr uint64_t GM = (1ull << OG) - 1;
    __shared__ int s_nk[KP];                   // workgroup accumulator of the n_k changes
    __shared__ int s_nk0[KP];                  // the sweep-start n_k
    __shared__ int s_ndk[OT / 4][ONT][4];      // n_dk | sweep-start n_dk << 16, [rho >> 2][thread][rho & 3]
    __shared__ float s_pa[OT / 4][ONT][4];     // tier-0 factor fl32((n_dk + alpha) / (n_k + V*beta))
    __shared__ float s_u[ONT / G][2 * G];      // the fp32 uniforms of the next 2 G sites of every document
    __shared__ int s_hot[OT][8];               // row rho: -1 (or -65536: the upper half) in the packed register that holds slot rho, else 0

    const int tid = threadIdx.x;
    for (int i = tid; i < KP; i += ONT) {
        s_nk[i] = 0;
        s_nk0[i] = P.n_k[i];
    }
    if (tid < OT * 8) {
        // slot s = 8 e + a (rho = 2 a + e) sits in xp[4 e + (a >> 1)], half a & 1 (the image's 16-byte piece e of the lane)
        const int so = tid >> 3, k = tid & 7, e = so & 1, a = so >> 1;
        s_hot[so][k] = k == 4 * e + (a >> 1) ? -(1 << ((a & 1) << 4)) : 0;
    }
    __syncthreads();

    const int lane = tid & 63, lg = tid & (G - 1), row = lane >> 3, grp = tid >> 3;
    const int gbase = lane & (64 - G);                   // first lane of the document
    const float vbeta32 = (float)P.vbeta, alpha32 = (float)P.alpha, beta32 = (float)P.beta;

    const int64_t site_base = P.doc_off[0];
    int32_t *z_b = P.z + site_base;
    const int32_t *rec_b = P.site_rec + site_base * 4;

    typedef int v4i __attribute__((ext_vector_type(4)));

    auto update = [&](int sg, int pos, int df) {
        const int w = OLDS(s_ndk, sg, tid) + df;                     // (0 <= n_dk + df < 2^16: no carry into the upper half)
        OLDS(s_ndk, sg, tid) = w;
        const int nd = w & 0xffff, nk = s_nk0[pos] + nd - (int)((uint32_t)w >> 16);
        OLDS(s_pa, sg, tid) = tier0_factor(nd, nk, alpha32, vbeta32);
    };

    for (int it = 0; it < P.dpg; ++it) {
        const int64_t idx = ((int64_t)blockIdx.x * P.dpg + it) * (ONT / G) + grp;
        // a lane group without a document walks a copy of the last one with no sites (the cold tiers need every lane of the wavefront)
        const bool valid = idx < P.D;
        const int64_t ic = valid ? idx : P.D - 1;
        const int64_t d = P.doc_order ? (int64_t)P.doc_order[ic] : ic;
        int64_t s0 = P.doc_off[d];
        int len = valid ? (int)(P.doc_off[d + 1] - s0) : 0;
        if (len <= 0) {
            len = 0;
            s0 = site_base;
        }
        int ml = len;
#pragma unroll
        for (int m = G; m < 64; m <<= 1) ml = max(ml, __shfl_xor(ml, m, 64));
        const int maxlen = __builtin_amdgcn_readfirstlane(ml);
        if (maxlen == 0) continue;                                   // (uniform)

        int32_t *ndk_row = P.n_dk + d * KP;
        // the slots of this lane that hold a topic (all ones when K == KP): a position without one keeps the factor 0
        const uint32_t vm = PAD ? (uint32_t)P.lab_mask[d * G + lg] : 0xFFFFu;
        {
            int big = 0, tokens = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const v4i a = ((const v4i *)ndk_row)[i * G + lg];
                const int r[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int s = 4 * i + c, rho = oct_rho_of_slot(s);
                    const int k = s_nk0[(i << IS) + lg * 4 + c];
                    OLDS(s_ndk, rho, tid) = r[c] | (r[c] << 16);
                    OLDS(s_pa, rho, tid) = ((vm >> s) & 1u) ? tier0_factor(r[c], k, alpha32, vbeta32) : 0.0f;
                    big |= r[c];
                    tokens += (int)((uint32_t)r[c] & 0xffffu);
                }
            }
            // the packed word holds the counts of a document of at most 65 535 tokens: status bit 2 otherwise
#pragma unroll
            for (int m = 1; m < G; m <<= 1) tokens += __shfl_xor(tokens, m, G);
            if (valid && (((uint32_t)big >> 16) || tokens > 65535) && P.status) atomicOr(P.status, 4);
        }
        const uint32_t gdoc = (uint32_t)(d + P.doc_base);
        const uint32_t sb = (uint32_t)(s0 - site_base) * 4u;
        const int last = len > 0 ? len - 1 : 0;
        auto off_of = [&](int n) { return opaque_u32(sb + (uint32_t)min(n, last) * 4u); };
        int pv = 0, pf = 0, pc = 0;                                     // the record in flight
        auto load_rec = [&](int &v, int &f, int &c, const uint32_t o) {
            const v4i r = *(const LLDA_GLOBAL v4i *)((const LLDA_GLOBAL char *)rec_b + (o << 2));
            v = r.x; f = r.y; c = r.z;
        };
        auto decode_old = [&](QuadSite &R) {
            R.lo = (R.zo >> 2) & (G - 1);
            R.so = oct_rho(R.zo);
        };
        // the 16-bit row of word v: pieces j = slots 8 j .. 8 j + 7 of this lane, and the row's flag
        int xp[8];
        auto load_row16 = [&](const int v, int &flag) {
            const LLDA_GLOBAL char *q = (const LLDA_GLOBAL char *)P.n_kw16 + (((uint32_t)v << 8) + (uint32_t)(lg & 1) * 64u + (uint32_t)(lg >> 1) * 16u);
            const v4i a = *(const LLDA_GLOBAL v4i *)q, b = *(const LLDA_GLOBAL v4i *)(q + 128);
            xp[0] = a.x; xp[1] = a.y; xp[2] = a.z; xp[3] = a.w;
            xp[4] = b.x; xp[5] = b.y; xp[6] = b.z; xp[7] = b.w;
            flag = *(const LLDA_GLOBAL uint8_t *)((const LLDA_GLOBAL char *)P.row16 + (uint32_t)v);
        };
        // xp -> fp32 in slot order rho (exact: 16-bit counts); the lanes of a document whose row does not fit 16 bits read the int32 row
        // now, without prefetch, and take the own count out here (so, own)
        o_v16f xv;
        auto convert_row = [&](const int v, const int flag, const int so, const float own) {
            LLDA_MARK("convert");
            const uint64_t wide_w = __ballot(flag == 0);
            if (__builtin_expect(wide_w == 0, 1)) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {                           // xp[4 e + m]: slots 8 e + 2 m (low half), 8 e + 2 m + 1
                    const int e = k >> 2, m = k & 3;
                    xv[2 * (2 * m) + e] = (float)((uint32_t)xp[k] & 0xffffu);
                    xv[2 * (2 * m + 1) + e] = (float)((uint32_t)xp[k] >> 16);
                }
            } else {
                LLDA_MARK("rare_wide_row");
                int xi[OT];
#pragma unroll
                for (int t = 0; t < OT; ++t) xi[t] = 0;
                if (flag == 0) {
                    const LLDA_GLOBAL v4i *q = (const LLDA_GLOBAL v4i *)((const LLDA_GLOBAL int32_t *)P.n_kw + ((uint64_t)(uint32_t)v << 7));
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const v4i a = q[i * G + lg];
                        xi[oct_rho_of_slot(4 * i)] = a.x; xi[oct_rho_of_slot(4 * i + 1)] = a.y;
                        xi[oct_rho_of_slot(4 * i + 2)] = a.z; xi[oct_rho_of_slot(4 * i + 3)] = a.w;
                    }
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int e = k >> 2, m = k & 3;
                    const int ra_ = 2 * (2 * m) + e, rb_ = 2 * (2 * m + 1) + e;
                    xv[ra_] = flag == 0 ? (float)xi[ra_] : (float)((uint32_t)xp[k] & 0xffffu);
                    xv[rb_] = flag == 0 ? (float)xi[rb_] : (float)((uint32_t)xp[k] >> 16);
                }
#pragma unroll
                for (int r = 0; r < OT; ++r) xv[r] -= (flag == 0 && so == r) ? own : 0.0f;
            }
        };
        // the site's own count leaves the PACKED row through the one-hot LDS row of its slot (kernel_quad.hpp): two broadcast reads,
        // eight multiply-adds for the eight documents of the wavefront
        auto remove_own_packed = [&](const int so, const int own) {
            LLDA_MARK("own_removal");
            const v4i *hot = (const v4i *)&s_hot[so][0];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const v4i h = hot[j];
                xp[4 * j] += __mul24(h.x, own); xp[4 * j + 1] += __mul24(h.y, own);
                xp[4 * j + 2] += __mul24(h.z, own); xp[4 * j + 3] += __mul24(h.w, own);
            }
        };

        // Software pipeline as in the quad kernel: at the top of iteration n, xv holds the row of site n as fp32 with the site's own count
        // taken out (made during iteration n-1); records run three sites ahead, z two.
        QuadSite R0, R1, R2;
        load_rec(R0.v, R0.f, R0.c, off_of(0)); R0.zo = gload_i32(z_b, off_of(0));
        load_rec(R1.v, R1.f, R1.c, off_of(1)); R1.zo = gload_i32(z_b, off_of(1));
        load_rec(pv, pf, pc, off_of(2));
        R0.zn = R1.zn = 0;
        R2.v = R2.f = R2.zo = R2.c = R2.zn = R2.lo = R2.so = R2.w = 0;
        load_row16(R0.v, R0.w);
        uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        decode_old(R0);
        if (len > 0 && lg == R0.lo) update(R0.so, R0.zo, -R0.f);      // site 0 leaves its topic (LabeledLDA.py:109-111)
        remove_own_packed(R0.so, (len > 0 && lg == R0.lo) ? R0.f : 0);
        convert_row(R0.v, R0.w, R0.so, (len > 0 && lg == R0.lo) ? (float)R0.f : 0.0f);
        load_row16(R1.v, R1.w);                                        // row of site 1

        auto site = [&](const int n, QuadSite &cur, QuadSite &nxt, QuadSite &prv) {
            const bool act = n < len, more = n + 1 < len;
            const int f = cur.f, zo = cur.zo;
            LLDA_MARK("site_top");
            LLDA_MARK("lds_factors");
            __builtin_amdgcn_s_setprio(QP_TOP);
            q_v2f pa[8];
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                pa[a].x = OLDS(s_pa, 2 * a, tid);
                pa[a].y = OLDS(s_pa, 2 * a + 1, tid);
            }
            // the random bits of 16 sites at a time (one Philox block per lane serves two sites)
            LLDA_MARK("rng");
            if ((n & (2 * G - 1)) == 0) {
                LLDA_MARK("rare_philox");
                r0 = (uint32_t)(n >> 1) + (uint32_t)lg; r1 = gdoc; r2 = P.stream_id; r3 = P.sweep;
                philox4x32_10(r0, r1, r2, r3, P.key0, P.key1);
                s_u[grp][2 * lg] = (float)(r0 >> 5) * 0x1p-27f;
                s_u[grp][2 * lg + 1] = (float)(r2 >> 5) * 0x1p-27f;
                LLDA_MARK("rng");
            }
            const float u32 = s_u[grp][n & (2 * G - 1)];
            int zn;
            uint64_t unsure = oct_draw<PAD>(xv, pa, u32, P.margin0_rel, P.margin0_data, beta32, lg, zn) & __ballot(act);
            LLDA_MARK("cold_check");
            if (__builtin_expect(unsure != 0, 0)) {
                LLDA_MARK("rare_cold");
                // tier 1 (fp64, margin 2^-40) right here for all eight documents; what it cannot decide and sites of int32 rows go to the
                // exact tier out of line, one document at a time
                const int holder = (n >> 1) & (G - 1);
                const uint32_t ra_l = (n & 1) ? r2 : r0, rb_l = (n & 1) ? r3 : r1;
                const int bp_h = (gbase | holder) << 2;
                const uint32_t ra = (uint32_t)__builtin_amdgcn_ds_bpermute(bp_h, (int)ra_l), rb = (uint32_t)__builtin_amdgcn_ds_bpermute(bp_h, (int)rb_l);
                const uint64_t t0_w = unsure;
                if (lg == 0 && ((t0_w >> gbase) & GM) && P.status) atomicAdd(P.status + 1, 1);   // statistics
                int z1;
                const uint64_t still = ((P.margin_rel < 1.0 ? oct_tier1<PAD>(xv, s_ndk, s_nk0, tid, lg, uniform53(ra, rb), P.alpha, P.beta, P.vbeta,
                                                                             P.margin_rel, vm, z1) : ~0ull) | __ballot(cur.w == 0)) & __ballot(act);
                const bool mine0 = ((t0_w >> gbase) & GM) != 0;
                zn = mine0 ? z1 : zn;
                uint32_t rows = 0;
#pragma unroll
                for (int r = 0; r < DPW; ++r)
                    rows |= (((t0_w >> (G * r)) & GM) && ((still >> (G * r)) & GM)) ? (1u << r) : 0u;
                rows = (uint32_t)__builtin_amdgcn_readfirstlane((int)rows);
                while (__builtin_expect(rows != 0, 0)) {
                    const int r = __builtin_ctz(rows);
                    rows &= rows - 1;
                    const int src = r * G;
                    const int zo_r = __builtin_amdgcn_readlane(zo, src);
                    int zc = oct_cold<PAD>(s_ndk, s_nk0, (tid & 64) + src, __builtin_amdgcn_readlane(cur.v, src),
                                           __builtin_amdgcn_readlane(f, src), zo_r, (uint32_t)__builtin_amdgcn_readlane((int)ra, src),
                                           (uint32_t)__builtin_amdgcn_readlane((int)rb, src), lane,
                                           (int64_t)__builtin_amdgcn_readlane((int)d, src),
                                           (const KParams *)__builtin_amdgcn_kernarg_segment_ptr());
                    if (__builtin_expect(zc < 0, 0)) {
                        zc = zo_r;
                        if (lane == 0 && P.status) atomicOr(P.status, 1);   // no topic with positive probability
                    }
                    zn = (row == r) ? (zc | (oct_rho(zc) << 9)) : zn;
                }
            }
            LLDA_MARK("decode");
            {
                const int zpos = zn & 511, sn = zn >> 9, ln = (zn >> 2) & (G - 1);
                cur.zn = zpos;
                decode_old(nxt);
                const bool own_new = act && lg == ln, own_old = more && lg == nxt.lo;
                const int sg = own_new ? sn : own_old ? nxt.so : 0;
                const int ps = own_new ? zpos : own_old ? nxt.zo : (lg << 2);
                const int df = own_new ? f : own_old ? -nxt.f : 0;
                LLDA_MARK("count_update");
                const int w0 = OLDS(s_ndk, sg, tid), k0 = s_nk0[ps];
                LLDA_MARK("scalars");
                __builtin_amdgcn_s_setprio(QP_BULK);
                const int w_next = pv;                                 // the record of site n+2
                prv.v = pv; prv.f = pf; prv.c = pc;
                prv.zo = gload_i32(z_b, off_of(n + 2));
                load_rec(pv, pf, pc, off_of(n + 3));
                // site n+1: its row (issued an iteration ago) -> fp32, own count out; then the row of site n+2 is issued
                remove_own_packed(nxt.so, (more && lg == nxt.lo) ? nxt.f : 0);
                convert_row(nxt.v, nxt.w, nxt.so, (more && lg == nxt.lo) ? (float)nxt.f : 0.0f);
                LLDA_MARK("row_prefetch");
                load_row16(w_next, prv.w);
                LLDA_MARK("count_update");
                const int w = w0 + df;
                OLDS(s_ndk, sg, tid) = w;
                const int nd = w & 0xffff, nk = k0 + nd - (int)((uint32_t)w >> 16);
                OLDS(s_pa, sg, tid) = tier0_factor(nd, nk, alpha32, vbeta32);
                if (__builtin_expect(__ballot(own_new && own_old) != 0, 0)) {
                    LLDA_MARK("rare_second_update");
                    if (own_new && own_old) update(nxt.so, nxt.zo, -nxt.f);
                }
                // (the commit comes LAST: kernel_quad.hpp)
                {
                    LLDA_MARK("commit");
                    const uint32_t zoff = opaque_u32(sb + (uint32_t)n * 4u);
                    const LLDA_GLOBAL uint32_t *lp = (const LLDA_GLOBAL uint32_t *)P.commit_log + (uint32_t)(cur.c & 0x7fffffff);
                    const uint32_t word = (uint32_t)zo | ((uint32_t)zpos << 16);
                    if (lg == 0 && act) {
                        gstore_i32(z_b, zoff, zpos);
                        *(LLDA_GLOBAL uint32_t *)lp = word;
                    }
                }
            }
            LLDA_MARK("loop");
        };
        for (int n = 0;; n += 3) {                                  // (uniform trip count: the longest document of the wavefront)
            site(n, R0, R1, R2);
            if (n + 1 >= maxlen) break;
            site(n + 1, R1, R2, R0);
            if (n + 2 >= maxlen) break;
            site(n + 2, R2, R0, R1);
            if (n + 3 >= maxlen) break;
        }

        // document done: fold its n_dk change into the workgroup's n_k accumulator, store the row
        if (valid && len > 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int o[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int w = OLDS(s_ndk, oct_rho_of_slot(4 * i + c), tid);
                    o[c] = w & 0xffff;
                    const int dl = o[c] - (int)((uint32_t)w >> 16);
                    if (dl) atomicAdd(&s_nk[(i << IS) + lg * 4 + c], dl);
                }
                v4i a = {o[0], o[1], o[2], o[3]};
                ((v4i *)ndk_row)[i * G + lg] = a;
            }
        }
    }

    __syncthreads();
    for (int i = tid; i < KP; i += ONT) {
        const int dl = s_nk[i];
        if (dl) atomicAdd(P.n_k_delta + i, dl);
    }
}

}  // namespace
