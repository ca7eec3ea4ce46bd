// kernel_quad.hpp -- llda_sweep_quad_kernel: the K = 512 dense kernel with FOUR documents per wavefront
// Part of the single translation unit llda_gibbs.hip (included in order; see the contents list there).
#pragma once

namespace {

// ---------------------------------------------------------------------------------------------
// Why.  llda_sweep_kernel<32,16,..,R16,W4> (kernel_sweep.hpp) is bound by instruction issue: 84 vector instructions per site of
// which 20 are the arithmetic of the 512 scores (profiles/r05_site_loop_budget.md); the other ~130 per wavefront iteration -- lane scan,
// threshold, search, pick, count update, addresses, commit -- are paid once per iteration whatever the number of documents in the
// wavefront.  Here a document is walked by 16 lanes x 32 slots, four documents per wavefront: the fixed part is shared by four sites.
// LDS holds the same 4 KB per document, so a CU holds the same 32 documents -- in 8 wavefronts (two per SIMD, 256 VGPRs) instead of 16.
//
// Geometry.  The layout of K = 512 is 32 lanes x 16 slots (llda_layout: G = 32, T = 16).  Quad lane lq plays the standard lanes
// 2 lq and 2 lq + 1 one after the other, so the draw order (standard lane, slot) is unchanged:
//     device position     pos   = i << 7 | lq << 3 | e << 2 | c        i = slot >> 2, e = standard lane & 1, c = slot & 3
//     register / LDS slot rho   = 8 i + 2 c + e: the pair (2a, 2a + 1), a = 4 i + c, holds element a of chain A (the slots of
//                         standard lane 2 lq) and of chain B (those of lane 2 lq + 1): one packed fma advances both chains
// Rows come from the 16-bit image of n_kw (llda_pack_rows16_all: EVERY row, with a per-row flag "all counts fit" decided per sweep);
// a site whose row does not fit reads the int32 row without prefetch.  Documents of at most 65 535 tokens (n_dk packed with its
// sweep-start value, as the W4 form).
//
// Tier 0 with two chains of 16 per lane: every prefix within (12 + 16) v of the lane's share, the lane total one more,
// four scan steps: X and the total within 33 v, the target fl(u~ tot~ - X~[g-1]) (one fma) within 69.2 v, its bounds 70.2 v; chain A is
// compared directly (98.2 v), chain B against the bounds minus chain A's total (99.2 v) -- inside the 105 v of the 2-document kernel,
// margin 128 v.  Int32 rows with counts >= 2^24: + 2 v.
// ---------------------------------------------------------------------------------------------
constexpr float LLDA_MARGIN0_QUAD = 0x1.ap-18f;   // tier-0 margin of this kernel relative to the total: 104 * 2^-24 (bound 99.2 v, 101.2 v with
                                                  // int32 counts >= 2^24) -- what the test hooks scale; production uses the sharper form below
// The data-dependent margin (production).  With v = 2^-24 and true values L (lane total), P (the lanes before), t = u * total, the
// compared difference  q~[s] - (tg~ -+ m)  is off by at most
//     28.1 v L  (prefix: 12 v of the terms + 16 roundings)  + 1 v L (chain B against the bounds minus chain A's total)
//   + 34.2 v t + 0.125 v total  (u~: 2^-27 + v u; total~: 29.1 v + 4 scan steps)  + 1 v t (t = u * total rounded on its own)
//   + 33.1 v P  + 3 v |tg|  (tg, its bounds, the chain-B bounds: one rounding each; |tg| <= t + P)  + 2 v for int32 counts >= 2^24
// <= v (31.1 L + 38.2 t + 36.1 P + 0.125 total) -- a third of 104 v * total on average (L ~ total / 16, t ~ P ~ total / 2).  The kernel
// takes m = v * 1.05 * (32 L~ + 39 t~ + 37 P~ + 0.25 total~) from the computed values (each within 34 v of its true value).
// Checked, not only derived: oracle/quad_tier0_model.c replays this arithmetic in fp32 next to 80-bit arithmetic and
// tests/test_quad_tier0_model.py asserts the sub-bounds and the bound position by position (worst compared difference: 0.48 of m);
// tests/test_gpu_parity.py::test_adversarial_near_ties_for_the_quad_tier0 plants thresholds within 2^-28 of a prefix boundary where m is
// smallest, and tools/quad_margin_headroom.py scales m down until planted ties are missed (first at 1/8: profiles/r06_quad_margin_headroom.md).
constexpr float QM_L = 1.05f * 32.0f * 0x1p-24f, QM_T = 1.05f * 39.0f * 0x1p-24f, QM_P = 1.05f * 37.0f * 0x1p-24f, QM_TOT = 1.05f * 0.25f * 0x1p-24f;
#define QLDS(arr, rho, t) (arr)[(rho) >> 2][t][(rho) & 3]      // the per-document LDS arrays, see the kernel
constexpr int QT = 32;        // slots per quad lane
// Issue priority by phase of an iteration (s_setprio; round 6).  The two wavefronts of a SIMD run the same code on documents of the same
// length: left alone they drift into the SAME phase and compete for the vector unit in the bulk phases while both wait in the dependent
// ones.  A wavefront raises its priority for the independent bulk of an iteration -- from the scalar loads / LDS reads of the count
// update through the next site's own-count removal, conversion and row prefetch, the update's arithmetic and the commit (QP_BULK), and
// for the factor reads and the chains of the next site (QP_TOP) -- and drops it for the dependent decision (lane scan, threshold,
// search, key minimum, cold tiers, decode: QP_DEC): the wavefront in its bulk gets the issue slots, the other one's dependent chain
// fills the gaps, and the two stay in antiphase.  configs[3]: 9.39 -> 8.66 ms per 250 000 documents (- 7.7 %), K = 256 - 1.5 %, K = 128
// - 0.4 %; placements measured: profiles/r06_site_loop_budget.md.  -DLLDA_QUAD_PRIO=tdb (three digits) overrides (llda_build_info bit).
#ifdef LLDA_QUAD_PRIO
constexpr int QP_TOP = (LLDA_QUAD_PRIO) / 100 % 10, QP_DEC = (LLDA_QUAD_PRIO) / 10 % 10, QP_BULK = (LLDA_QUAD_PRIO) % 10;
#else
constexpr int QP_TOP = 3, QP_DEC = 0, QP_BULK = 2;
#endif
constexpr int QNT = 128;      // threads per workgroup: two wavefronts, eight documents

// The same walk for the narrower layouts of 16 slots per lane (llda_layout: T = 16): a document is LPD = 2^LB lanes x 32 slots, 64 / LPD
// documents per wavefront --
//     LB = 4   K = 512 (G = 32): 4 documents per wavefront      LB = 3   K = 256 (G = 16): 8      LB = 2   K = 128 (G = 8): 16
//     device position     pos   = i << (3 + LB) | lq << 3 | e << 2 | c
// Everything per lane (chains, search, count update, LDS layout) is the same code; the cross-lane steps stay inside the LPD lanes of a
// document (scan steps masked at the document's first lanes, totals and the minimum by xor butterflies), the random bits come 2 LPD
// sites at a time, and the site's own count leaves the PACKED 16-bit row by compare-and-subtract (sixteen documents would mean sixteen
// indexed writes).  The error bound of tier 0 only shrinks (fewer scan steps), the constants below are kept.
template <int LB>
struct QuadGeo {
    static constexpr int LPD = 1 << LB;            // lanes per document
    static constexpr int G = 2 * LPD;              // lanes of the standard layout (llda_layout.G)
    static constexpr int KP = 32 * LPD;            // positions
    static constexpr int IS = 3 + LB;              // shift of the slot chunk i in a position
    static constexpr int DPW = 64 / LPD;           // documents per wavefront
    static constexpr uint64_t GM = LPD == 64 ? ~0ull : ((1ull << LPD) - 1);     // the lanes of a document in a ballot, shifted down
    static constexpr uint32_t KEY_NONE = 0xFC000u | 31u << 9 | (uint32_t)(KP - 1);   // no slot above lo: the last slot of the last lane
};

// slot number of a device position: rho = 8 i + 2 c + e
template <int LB>
__device__ __forceinline__ int quad_rho(int pos) { return ((pos >> LB) & 0x18) | ((pos & 3) << 1) | ((pos >> 2) & 1); }
constexpr int quad_rho_of(int i, int e, int c) { return 8 * i + 2 * c + e; }

// minimum of a key over the lanes of a document, in every lane: one DPP instruction per step (the compiler's form is three)
template <int LB>
__device__ __forceinline__ uint32_t quad_min_key(uint32_t key)
{
    if constexpr (LB == 4)
        asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                     "v_min_u32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                     "v_min_u32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                     "v_min_u32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(key));
    else if constexpr (LB == 3)
        asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                     "v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                     "v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(key));
    else
        asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                     "v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(key));
    return key;
}

// the counts of one document for the cold tiers, which play it in the STANDARD layout (G lanes x 16 slots)
template <int LB>
struct QuadCounts {
    const int (*s_ndk)[QNT][4];
    const int *s_nk0;
    int t;         // thread of the workgroup that holds this standard lane's slots
    int e;         // standard lane & 1
    int g;         // standard lane
    bool st;
    __device__ __forceinline__ int word(int s) const { return QLDS(s_ndk, 8 * (s >> 2) + 2 * (s & 3) + e, t); }
    __device__ __forceinline__ int nd(int s) const { return word(s) & 0xffff; }
    __device__ __forceinline__ int nk(int s) const
    {
        const int w = word(s);
        return s_nk0[pos_of<QuadGeo<LB>::G, 16>(g, s)] + (w & 0xffff) - (int)((uint32_t)w >> 16);
    }
    __device__ __forceinline__ bool stats() const { return st; }
    __device__ __forceinline__ bool count_unsure() const { return false; }     // (the kernel counted the site before its own tier 1)
};

// One undecided site: every G lanes of the wavefront play the document in the standard layout (all but the first G silently), the row
// comes from n_kw itself.  tbase = thread of the document's quad lane 0; w, f, zo = word, frequency and old position of the site.
template <int LB, bool PAD>
__device__ __noinline__ int quad_cold(const int (*s_ndk)[QNT][4], const int *s_nk0, int tbase, int w, int f, int zo, uint32_t ra,
                                      uint32_t rb, int lane, int64_t d, const KParams *P)
{
    constexpr int G = QuadGeo<LB>::G;
    const int g = lane & (G - 1);
    int x[16];
    gload_lane_row<G, 16>(P->n_kw, (int64_t)w * QuadGeo<LB>::KP, g, x);
    int lo, so;
    lane_slot_of<G, 16>(zo, lo, so);
#pragma unroll
    for (int s = 0; s < 16; ++s) x[s] -= (g == lo && s == so) ? f : 0;       // the site's own count (LabeledLDA.py:109-111)
    const QuadCounts<LB> dc{s_ndk, s_nk0, tbase + (g >> 1), g & 1, g, lane < G};
    // K < KP (positions without a topic, a last leaf with a tail, leaves of unequal length): the masked form with numpy's tail
    if constexpr (PAD) return cold_tiers_acc<G, 16, true, false>(dc, x, P->lab_mask[d * G + g], uniform53(ra, rb), g, lane, P);
    else return cold_tiers_acc<G, 16, false, true>(dc, x, 0xFFFFu, uniform53(ra, rb), g, lane, P);
}

typedef float q_v2f __attribute__((ext_vector_type(2)));
typedef float q_v32f __attribute__((ext_vector_type(32)));

// Tier 0 for all documents of the wavefront at once.  xv = the row minus the site's own count (fp32, exact), pa = the cached factors, both in slot
// order rho: the pair (2a, 2a+1) holds element a of chain A and of chain B, so ONE packed instruction advances both chains.
// Returns the wavefront's ballot of the lanes that are not sure; zn = the position every lane's document drew.
// margin_rel, margin_data: m = total * margin_rel + margin_data * (the data-dependent form): production (0, 1), test hooks (2^-n or 2, 0)
// PAD (K < KP): "no slot above lo in any lane" names position KP - 1, which holds no topic there -- the margin makes that outcome
// impossible for a sure site (the last lane's last TOPIC has the total as its prefix), but a test hook or a later change of the margin
// must not be able to write a topic-less position: the document is reported as not sure, and the exact tier's masked fallback decides.
template <int LB, bool PAD = false>
__device__ __forceinline__ uint64_t quad_draw(const q_v32f &xv, const q_v2f (&pa)[16], float u, float margin_rel, float margin_data, float beta,
                                              int lq, int &zn)
{
    const q_v2f b2 = {beta, beta};
    q_v2f Q[16];
    LLDA_MARK("scores");
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        const q_v2f x2 = {xv[2 * a], xv[2 * a + 1]};
        const q_v2f nb = x2 + b2;
        if (a == 0) Q[0] = nb * pa[0];
        else Q[a] = __builtin_elementwise_fma(nb, pa[a], Q[a - 1]);
    }
    // inclusive scan over the lanes of the document (16 lanes = one DPP row)
    LLDA_MARK("lane_scan");
    __builtin_amdgcn_s_setprio(QP_DEC);
    const float X0 = Q[15].x + Q[15].y;
    float X = X0;
    float tot = X0, prev;
    if constexpr (LB == 4) {
        X += dpp_f32<DPP_ROW_SHR + 1>(X);
        X += dpp_f32<DPP_ROW_SHR + 2>(X);
        X += dpp_f32<DPP_ROW_SHR + 4>(X);
        X += dpp_f32<DPP_ROW_SHR + 8>(X);
        // the document's total in every lane: four rotate-and-add steps over the lane totals, next to the scan (an LDS round trip for
        // the scan's last lane costs the wavefront ~100 cycles of waiting).  Another association order than the scan's: within 33 v too.
        tot += dpp_f32<DPP_ROW_ROR + 8>(tot);
        tot += dpp_f32<DPP_ROW_ROR + 4>(tot);
        tot += dpp_f32<DPP_ROW_ROR + 2>(tot);
        tot += dpp_f32<DPP_ROW_ROR + 1>(tot);
        prev = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(X), DPP_ROW_SHR + 1, 0xF, 0xF, true));
    } else {
        // several documents share a DPP row: a step adds 1.0 * the shifted value, or 0.0 * it in the first lanes of a document (one
        // fma with the DPP operand; x * 1.0 is exact, so the sum rounds once, as the plain add)
        const float m1 = lq >= 1 ? 1.0f : 0.0f, m2 = lq >= 2 ? 1.0f : 0.0f;
        X = __builtin_fmaf(dpp_f32<DPP_ROW_SHR + 1>(X), m1, X);
        X = __builtin_fmaf(dpp_f32<DPP_ROW_SHR + 2>(X), m2, X);
        if constexpr (LB == 3) {
            const float m4 = lq >= 4 ? 1.0f : 0.0f;
            X = __builtin_fmaf(dpp_f32<DPP_ROW_SHR + 4>(X), m4, X);
            tot += dpp_f32<DPP_HALF_MIRROR>(tot);
        }
        tot += dpp_f32<DPP_XOR1>(tot);
        tot += dpp_f32<DPP_XOR2>(tot);
        prev = dpp_f32<DPP_ROW_SHR + 1>(X) * m1;
    }
    LLDA_MARK("threshold");
    const float t = u * tot;
    const float tg = t - prev;
    const float md = __builtin_fmaf(QM_L, X0, __builtin_fmaf(QM_T, t, __builtin_fmaf(QM_P, prev, QM_TOT * tot)));
    const float margin = __builtin_fmaf(tot, margin_rel, margin_data * md);
    const float lo0 = tg - margin, hi0 = tg + margin;
    // chain A or chain B?
    LLDA_MARK("search");
    const bool c0 = Q[15].x <= lo0;
    const float dA = c0 ? Q[15].x : 0.0f;
    const float lo = lo0 - dA, hi = hi0 - dA;
    float q[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) q[a] = c0 ? Q[a].y : Q[a].x;
    // branch-free binary search (draw_tiers.hpp, count_sorted_f32) that keeps the smallest element found above lo
    const bool c5 = q[15] <= lo;
    float ub = c5 ? __int_as_float(0x7f800000) : q[15];
    const bool c1 = q[7] <= lo;
    ub = c1 ? ub : q[7];
    const float m2 = c1 ? q[11] : q[3];
    const float a1 = c1 ? q[9] : q[1], a5 = c1 ? q[13] : q[5];
    const float e0 = c1 ? q[8] : q[0], e2 = c1 ? q[10] : q[2], e4 = c1 ? q[12] : q[4], e6 = c1 ? q[14] : q[6];
    const bool c2 = m2 <= lo;
    ub = c2 ? ub : m2;
    const float m3 = c2 ? a5 : a1;
    const float g0 = c2 ? e4 : e0, g2 = c2 ? e6 : e2;
    const bool c3 = m3 <= lo;
    ub = c3 ? ub : m3;
    const float m4 = c3 ? g2 : g0;
    const bool c4 = m4 <= lo;
    ub = c4 ? ub : m4;
    // guards on the total in one class test (draw_fast_dense_f32): tot - margin is a positive normal number
    uint64_t bad_total;
    asm("v_cmp_class_f32_e64 %0, %1, %2" : "=s"(bad_total) : "v"(tot - margin), "v"(0x2FF));
    uint64_t unsure = __ballot(!(ub > hi)) | bad_total;
    // the position this lane would name, keyed by its lane; the document's first lane with a slot above lo wins (none: 511, the
    // last slot of the last lane).  Row-wide minimum: four DPP steps, one instruction each (the compiler's form is three)
    // key = lane << 14 | slot rho << 9 | position: every search outcome sets its bit of the position AND of the slot number
    LLDA_MARK("pick");
    constexpr uint32_t I1 = 2u << QuadGeo<LB>::IS, I0 = 1u << QuadGeo<LB>::IS;
    const uint32_t p = (c1 ? (I1 | 16u << 9) : 0u) | (c2 ? (I0 | 8u << 9) : 0u) | (c3 ? (2u | 4u << 9) : 0u) | (c4 ? (1u | 2u << 9) : 0u) |
                       (c0 ? (4u | 1u << 9) : 0u) | ((uint32_t)lq << 3) | ((uint32_t)lq << 14);
    uint32_t key = c5 ? QuadGeo<LB>::KEY_NONE : p;          // (no slot above lo: the last slot of the last lane wins only if no lane has one)
    key = quad_min_key<LB>(key);
    zn = (int)(key & 0x3FFFu);                              // slot << 9 | position; the lane is position >> 3 & (LPD - 1)
    if constexpr (PAD) unsure |= __ballot(key == QuadGeo<LB>::KEY_NONE);
    return unsure;
}

// Tier 1 in the quad layout, for the wavefront iterations in which tier 0 is unsure about some document: the decision from
// unnormalised fp64 prefix sums with the margin 2^-40 of the total (draw_tiers.hpp, cold_tiers_acc: the same test on a different
// association order -- the bound there, 254 u < 2^-44, grows by the 16 more additions of a 32-slot chain).  All four documents at once;
// xv = the row minus the site's own count, exact in fp32.  Returns the ballot of the lanes that are STILL not sure; zn as quad_draw.
template <int LB, bool PAD = false>
__device__ __forceinline__ uint64_t quad_tier1(const q_v32f &xv, const int (*s_ndk)[QNT][4], const int *s_nk0, int tid, int lq, double u,
                                               double alpha, double beta, double vbeta, double margin_rel, uint32_t vm, int &zn)
{
    double W[QT];
    double run = 0.0;
#pragma unroll
    for (int k = 0; k < QT; ++k) {                       // k = position in the draw order of the lane: chain A, then chain B
        const int e = k >> 4, a = k & 15, i = a >> 2, c = a & 3;
        const int rho = quad_rho_of(i, e, c);
        const int w = QLDS(s_ndk, rho, tid);
        const int nd = w & 0xffff, nk = s_nk0[(i << QuadGeo<LB>::IS) | (lq << 3) | (e << 2) | c] + nd - (int)((uint32_t)w >> 16);
        const double den = (double)nk + vbeta;
        // 1 / den from the fp32 reciprocal (1 ulp) and two Newton steps in fp64: within 2^-50, as cold_tiers_acc's
        double y = (double)__builtin_amdgcn_rcpf((float)den);
        y = __builtin_fma(__builtin_fma(-den, y, 1.0), y, y);
        y = __builtin_fma(__builtin_fma(-den, y, 1.0), y, y);
        const double ws = ((double)nd + alpha) * (((double)xv[rho] + beta) * y);
        run = run + (((vm >> k) & 1u) ? ws : 0.0);                // (vm: bit 16 e + a = this position holds a topic)
        W[k] = run;
    }
    constexpr int LPD = QuadGeo<LB>::LPD;
    const double X = group_scan<LPD>(run, lq);
    const double tot = bcast_last<LPD>(X, tid & 63);
    double prev = dpp_f64<DPP_ROW_SHR + 1>(X);                         // (0.0 into the first lane of the row)
    if constexpr (LB < 4) prev = lq ? prev : 0.0;
    const double tg = u * tot - prev;
    const double margin = tot * margin_rel;
    const double lo = tg - margin, hi = tg + margin;
    int cnt_lo = 0, cnt_hi = 0;
#pragma unroll
    for (int k = 0; k < QT; ++k) {
        cnt_lo += (W[k] <= lo) ? 1 : 0;
        cnt_hi += (W[k] <= hi) ? 1 : 0;
    }
    uint64_t unsure = __ballot((cnt_lo != cnt_hi) || !(tot > 0.0) || !(margin < tot));
    // position of chain index cnt_lo: e = k >> 4, i = (k >> 2) & 3, c = k & 3
    const uint32_t k = (uint32_t)cnt_lo;
    const uint32_t i_ = (k >> 2) & 3u, e_ = (k >> 4) & 1u, c_ = k & 3u;
    const uint32_t p = (i_ << QuadGeo<LB>::IS) | ((uint32_t)lq << 3) | (e_ << 2) | c_ | ((8u * i_ + 2u * c_ + e_) << 9) | ((uint32_t)lq << 14);
    uint32_t key = cnt_lo >= QT ? QuadGeo<LB>::KEY_NONE : p;
    key = quad_min_key<LB>(key);
    zn = (int)(key & 0x3FFFu);
    if constexpr (PAD) unsure |= __ballot(key == QuadGeo<LB>::KEY_NONE);       // (see quad_draw)
    return unsure;
}

struct QuadSite { int v, f, zo, c, zn, lo, so, w; };  // (lo, so) = quad lane and slot rho of zo; w = flag of the word's row (0: wide)

// -DQUAD_PROFILE (tools/quad_phase_profile.py; never in a production build: llda_build_info reports it): wavefront 0 of workgroup 0
// stamps the shader clock at the phase boundaries of every site and adds the differences up in status[8 + phase]
#ifdef QUAD_PROFILE
#define QP_DECL uint32_t qp_t = 0, qp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define QP_START() do { __builtin_amdgcn_sched_barrier(0); qp_t = (uint32_t)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define QP_MARK(k) do { __builtin_amdgcn_sched_barrier(0); const uint32_t qp_n = (uint32_t)__builtin_amdgcn_s_memtime(); \
                        qp_acc[k] += qp_n - qp_t; qp_t = qp_n; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define QP_DECL
#define QP_START()
#define QP_MARK(k)
#endif

// REC: {word, freq, csc_pos} of a site come as one 16-byte record (llda_sweep_args.site_rec; always for LB < 4)
// PAD: K < KP -- positions without a topic (the K == KP instantiations stay what they were: the validity word is a constant there)
template <int LB, bool REC = (LB < 4), bool PAD = false>
__global__ void __launch_bounds__(QNT, 2) llda_sweep_quad_kernel(const KParams P)
{
    typedef QuadGeo<LB> Geo;
    constexpr int KP = Geo::KP, LPD = Geo::LPD, G = Geo::G, IS = Geo::IS;
    __shared__ int s_nk[KP];                   // workgroup accumulator of the n_k changes
    __shared__ int s_nk0[KP];                  // the sweep-start n_k
    // [rho >> 2][thread][rho & 3]: a lane's four consecutive slots are 16 contiguous bytes, 16 bytes apart from lane to lane -- the 32
    // factors of a lane come with 8 ds_read_b128 (conflict free) instead of 16 two-address reads
    __shared__ int s_ndk[QT / 4][QNT][4];      // n_dk | sweep-start n_dk << 16
    __shared__ float s_pa[QT / 4][QNT][4];     // tier-0 factor fl32((n_dk + alpha) / (n_k + V*beta))
    __shared__ float s_u[QNT / LPD][2 * LPD];  // the fp32 uniforms of the next 2 LPD sites of every document
    __shared__ int s_hot[QT][16];              // row rho: -1 (or -65536: the upper half) in the packed register that holds slot rho, else 0

    const int tid = threadIdx.x;
    for (int i = tid; i < KP; i += QNT) {
        s_nk[i] = 0;
        s_nk0[i] = P.n_k[i];
    }
    for (int i = tid; i < QT * 16; i += QNT) {
        // slot rho = 8 i + 2 c' + e sits in xp[e << 3 | (i >> 1) << 2 | (i & 1) << 1 | c' >> 1], half c' & 1 (convert_row)
        const int so = i >> 4, k = i & 15;
        const int kk = ((so & 1) << 3) | ((so >> 4) << 2) | (((so >> 3) & 1) << 1) | ((so >> 2) & 1);
        s_hot[so][k] = k == kk ? -(1 << (((so >> 1) & 1) << 4)) : 0;
    }
    __syncthreads();

    const int lane = tid & 63, lq = tid & (LPD - 1), row = lane >> LB, grp = tid >> LB;
    const int gbase = lane & (64 - LPD);                 // first lane of the document
    const float vbeta32 = (float)P.vbeta, alpha32 = (float)P.alpha, beta32 = (float)P.beta;

    const int64_t site_base = P.doc_off[0];
    const int32_t *word_b = P.word + site_base, *freq_b = P.freq + site_base, *csc_b = P.csc_pos + site_base;
    int32_t *z_b = P.z + site_base;

    typedef int v4i __attribute__((ext_vector_type(4)));
    QP_DECL;

    auto update = [&](int sg, int pos, int df) {
        const int w = QLDS(s_ndk, sg, tid) + df;                     // (0 <= n_dk + df < 2^16: no carry into the upper half)
        QLDS(s_ndk, sg, tid) = w;
        const int nd = w & 0xffff, nk = s_nk0[pos] + nd - (int)((uint32_t)w >> 16);
        QLDS(s_pa, sg, tid) = tier0_factor(nd, nk, alpha32, vbeta32);
    };

    for (int it = 0; it < P.dpg; ++it) {
        const int64_t idx = ((int64_t)blockIdx.x * P.dpg + it) * (QNT / LPD) + grp;
        // a lane group without a document walks a copy of the last one with no sites: every lane of the wavefront stays in the site
        // loop (the cold tiers need all of them) and nothing it computes is stored
        const bool valid = idx < P.D;
        const int64_t ic = valid ? idx : P.D - 1;
        const int64_t d = P.doc_order ? (int64_t)P.doc_order[ic] : ic;
        int64_t s0 = P.doc_off[d];
        int len = valid ? (int)(P.doc_off[d + 1] - s0) : 0;
        if (len <= 0) {
            len = 0;
            s0 = site_base;
        }
        int maxlen;
        if constexpr (LB == 4) {
            maxlen = max(max(__builtin_amdgcn_readlane(len, 0), __builtin_amdgcn_readlane(len, 16)),
                         max(__builtin_amdgcn_readlane(len, 32), __builtin_amdgcn_readlane(len, 48)));
        } else {
            int ml = len;
#pragma unroll
            for (int m = LPD; m < 64; m <<= 1) ml = max(ml, __shfl_xor(ml, m, 64));
            maxlen = __builtin_amdgcn_readfirstlane(ml);
        }
        if (maxlen == 0) continue;                                   // (uniform)

        int32_t *ndk_row = P.n_dk + d * KP;
        // the positions of this lane that hold a topic (bit 16 e + slot: the masks of the standard lanes 2 lq and 2 lq + 1; all ones
        // when K == KP): a position without one keeps the factor 0 -- it is never drawn, and nothing ever updates it
        const uint32_t vm = PAD ? (uint32_t)P.lab_mask[d * G + 2 * lq] | (uint32_t)P.lab_mask[d * G + 2 * lq + 1] << 16 : ~0u;
        {
            int big = 0, tokens = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const v4i a = ((const v4i *)ndk_row)[i * G + 2 * lq], b = ((const v4i *)ndk_row)[i * G + 2 * lq + 1];
                const int r[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int rho = quad_rho_of(i, j >> 2, j & 3);
                    const int k = s_nk0[(i << IS) + lq * 8 + j];
                    QLDS(s_ndk, rho, tid) = r[j] | (r[j] << 16);
                    QLDS(s_pa, rho, tid) = ((vm >> (16 * (j >> 2) + 4 * i + (j & 3))) & 1u) ? tier0_factor(r[j], k, alpha32, vbeta32) : 0.0f;
                    big |= r[j];
                    tokens += (int)((uint32_t)r[j] & 0xffffu);
                }
            }
            // the packed word holds the counts of a document of at most 65 535 tokens (kernel_sweep.hpp, W4): status bit 2 otherwise
#pragma unroll
            for (int m = 1; m < LPD; m <<= 1) tokens += __shfl_xor(tokens, m, LPD);
            if (valid && (((uint32_t)big >> 16) || tokens > 65535) && P.status) atomicOr(P.status, 4);
        }
        const uint32_t gdoc = (uint32_t)(d + P.doc_base);
        const uint32_t sb = (uint32_t)(s0 - site_base) * 4u;
        const int last = len > 0 ? len - 1 : 0;
        auto off_of = [&](int n) { return opaque_u32(sb + (uint32_t)min(n, last) * 4u); };
        auto load_scalars = [&](QuadSite &R, const uint32_t o) {
            R.v = gload_i32(word_b, o); R.f = gload_i32(freq_b, o); R.c = gload_i32(csc_b, o); R.zo = gload_i32(z_b, o);
        };
        // LB < 4 (eight or sixteen documents per wavefront: every scalar load touches that many cache lines, and the vector-memory address
        // pipeline becomes the bound -- TA busy 0.71 with five scalar loads per site): {word, freq, csc_pos} come as ONE 16-byte record
        // (llda_sweep_args.site_rec), read THREE sites ahead, so that the row of site n+2 is issued from a record that has landed
        const int32_t *rec_b = REC ? P.site_rec + site_base * 4 : nullptr;
        int pv = 0, pf = 0, pc = 0;                                     // the record in flight
        auto load_rec = [&](int &v, int &f, int &c, const uint32_t o) {
            const v4i r = *(const LLDA_GLOBAL v4i *)((const LLDA_GLOBAL char *)rec_b + (o << 2));
            v = r.x; f = r.y; c = r.z;
        };
        auto decode_old = [&](QuadSite &R) {
            R.lo = (R.zo >> 3) & (LPD - 1);
            R.so = quad_rho<LB>(R.zo);
        };
        // the 16-bit row of word v: chunks (e, j) = slots 8j .. 8j+7 of standard lane 2 lq + e, and the row's flag
        // (32-bit byte offsets from the image: llda_sweep checked V * 1024 < 2^32; a row is 2 KP bytes, its second half KP bytes on)
        int xp[16];
        auto load_row16 = [&](const int v, int &flag) {
#ifdef ABL_NOLOAD
#pragma unroll
            for (int k = 0; k < 16; ++k) xp[k] = ((v + k) & 7) * 0x10001;      // ablation: no n_kw traffic
            flag = 1;
            return;
#endif
            // (the image is piece major: the 16 bytes of lane lq's piece (j, e) sit at (2 j + e) * 16 LPD + 16 lq, so that the lanes of
            // a document read CONTIGUOUS bytes with every load -- full 64-byte requests instead of half-used ones)
            const LLDA_GLOBAL char *q = (const LLDA_GLOBAL char *)P.n_kw16 + (((uint32_t)v << (IS + 3)) + (uint32_t)lq * 16u);
            const v4i a = *(const LLDA_GLOBAL v4i *)q, b = *(const LLDA_GLOBAL v4i *)(q + 32 * LPD);
            const v4i c = *(const LLDA_GLOBAL v4i *)(q + 16 * LPD), e = *(const LLDA_GLOBAL v4i *)(q + 48 * LPD);
            xp[0] = a.x; xp[1] = a.y; xp[2] = a.z; xp[3] = a.w;
            xp[4] = b.x; xp[5] = b.y; xp[6] = b.z; xp[7] = b.w;
            xp[8] = c.x; xp[9] = c.y; xp[10] = c.z; xp[11] = c.w;
            xp[12] = e.x; xp[13] = e.y; xp[14] = e.z; xp[15] = e.w;
            flag = *(const LLDA_GLOBAL uint8_t *)((const LLDA_GLOBAL char *)P.row16 + (uint32_t)v);
        };
        // xp -> fp32 in slot order (exact: 16-bit counts); the lanes of a document whose row does not fit 16 bits read the int32 row
        // now, without prefetch (an int32 count beyond 2^24 rounds: tier 0 stays inside its margin, section 4.3; tier 1 is skipped)
        q_v32f xv;
        // (the own count has left xp already, remove_own_packed; the lanes that read an int32 row take it out here: so, own)
        auto convert_row = [&](const int v, const int flag, const int so, const float own) {
            LLDA_MARK("convert");
            const uint64_t wide_w = __ballot(flag == 0);
            if (__builtin_expect(wide_w == 0, 1)) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int e = k >> 3, j = (k >> 2) & 1, m = k & 3;
                    const int i = 2 * j + (m >> 1), c = 2 * (m & 1);
                    xv[quad_rho_of(i, e, c)] = (float)((uint32_t)xp[k] & 0xffffu);
                    xv[quad_rho_of(i, e, c + 1)] = (float)((uint32_t)xp[k] >> 16);
                }
            } else {
                LLDA_MARK("rare_wide_row");
                int xi[QT];
#pragma unroll
                for (int t = 0; t < QT; ++t) xi[t] = 0;
                if (flag == 0) {
                    const LLDA_GLOBAL v4i *q = (const LLDA_GLOBAL v4i *)((const LLDA_GLOBAL int32_t *)P.n_kw + ((uint64_t)(uint32_t)v << (IS + 2)));
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const v4i a = q[i * G + 2 * lq], b = q[i * G + 2 * lq + 1];
                        xi[quad_rho_of(i, 0, 0)] = a.x; xi[quad_rho_of(i, 0, 1)] = a.y; xi[quad_rho_of(i, 0, 2)] = a.z; xi[quad_rho_of(i, 0, 3)] = a.w;
                        xi[quad_rho_of(i, 1, 0)] = b.x; xi[quad_rho_of(i, 1, 1)] = b.y; xi[quad_rho_of(i, 1, 2)] = b.z; xi[quad_rho_of(i, 1, 3)] = b.w;
                    }
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int e = k >> 3, j = (k >> 2) & 1, m = k & 3;
                    const int i = 2 * j + (m >> 1), c = 2 * (m & 1);
                    const int ra_ = quad_rho_of(i, e, c), rb_ = quad_rho_of(i, e, c + 1);
                    xv[ra_] = flag == 0 ? (float)xi[ra_] : (float)((uint32_t)xp[k] & 0xffffu);
                    xv[rb_] = flag == 0 ? (float)xi[rb_] : (float)((uint32_t)xp[k] >> 16);
                }
#pragma unroll
                for (int r = 0; r < QT; ++r) xv[r] -= (flag == 0 && so == r) ? own : 0.0f;
            }
        };
        // The site's own count leaves the PACKED 16-bit row: the slot number so is uniform in a document, and row so of s_hot holds -1 or
        // -65536 in the one register of the sixteen that carries the slot -- four LDS reads (the same address in every lane of a
        // document: broadcasts) and sixteen multiply-adds, for all documents of the wavefront at once.  own = 0 in the lanes that do
        // not hold the slot.  (Round 5 took it out of the fp32 row with a register-indexed subtract per document -- s_set_gpr_idx_on
        // under a hand-set exec mask on pinned registers, correct only behind an s_nop found by experiment; the narrower geometries
        // paid sixteen compare-select-subtract triples.  A row that does not fit 16 bits: see convert_row.)
        auto remove_own_packed = [&](const int so, const int own) {
            LLDA_MARK("own_removal");
            const v4i *hot = (const v4i *)&s_hot[so][0];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const v4i h = hot[j];
                xp[4 * j] += __mul24(h.x, own); xp[4 * j + 1] += __mul24(h.y, own);
                xp[4 * j + 2] += __mul24(h.z, own); xp[4 * j + 3] += __mul24(h.w, own);
            }
        };

        // Software pipeline.  At the top of iteration n, xv holds the row of site n as fp32 with the site's own count taken out -- made
        // during iteration n-1, in the shadow of the LDS reads of its count update, from the row that was issued an iteration earlier
        // still.  Scalars run two sites ahead in three rotating register sets (the loop is unrolled by three), the word ids three (wq).
        QuadSite R0, R1, R2;
        int wq = 0;                                      // (!REC) word of site n+2 at the top of iteration n
        if constexpr (REC) {
            load_rec(R0.v, R0.f, R0.c, off_of(0)); R0.zo = gload_i32(z_b, off_of(0));
            load_rec(R1.v, R1.f, R1.c, off_of(1)); R1.zo = gload_i32(z_b, off_of(1));
            load_rec(pv, pf, pc, off_of(2));
        } else {
            load_scalars(R0, off_of(0));
            load_scalars(R1, off_of(1));
            wq = gload_i32(word_b, off_of(2));
        }
        R0.zn = R1.zn = 0;
        R2.v = R2.f = R2.zo = R2.c = R2.zn = R2.lo = R2.so = R2.w = 0;
        load_row16(R0.v, R0.w);
        uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        decode_old(R0);
        if (len > 0 && lq == R0.lo) update(R0.so, R0.zo, -R0.f);      // site 0 leaves its topic (LabeledLDA.py:109-111)
        remove_own_packed(R0.so, (len > 0 && lq == R0.lo) ? R0.f : 0);
        convert_row(R0.v, R0.w, R0.so, (len > 0 && lq == R0.lo) ? (float)R0.f : 0.0f);
        load_row16(R1.v, R1.w);                                        // row of site 1

        auto site = [&](const int n, QuadSite &cur, QuadSite &nxt, QuadSite &prv) {
            const bool act = n < len, more = n + 1 < len;
            const int f = cur.f, zo = cur.zo;
            QP_START();
            LLDA_MARK("site_top");
            LLDA_MARK("lds_factors");
            __builtin_amdgcn_s_setprio(QP_TOP);
            q_v2f pa[16];
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                pa[a].x = QLDS(s_pa, 2 * a, tid);
                pa[a].y = QLDS(s_pa, 2 * a + 1, tid);
            }
            // the random bits of 32 sites at a time (one Philox block per lane serves two sites); their fp32 images -- the top 27
            // bits, within 2^-24 relative + 2^-27 absolute of u -- go through LDS, the bits themselves are only needed by the cold tiers
            LLDA_MARK("rng");
            if ((n & (2 * LPD - 1)) == 0) {
                LLDA_MARK("rare_philox");
                r0 = (uint32_t)(n >> 1) + (uint32_t)lq; r1 = gdoc; r2 = P.stream_id; r3 = P.sweep;
                philox4x32_10(r0, r1, r2, r3, P.key0, P.key1);
                s_u[grp][2 * lq] = (float)(r0 >> 5) * 0x1p-27f;
                s_u[grp][2 * lq + 1] = (float)(r2 >> 5) * 0x1p-27f;
                LLDA_MARK("rng");
            }
            const float u32 = s_u[grp][n & (2 * LPD - 1)];
            QP_MARK(0);                                                // factors + uniform issued
            int zn;
            uint64_t unsure = quad_draw<LB, PAD>(xv, pa, u32, P.margin0_rel, P.margin0_data, beta32, lq, zn) & __ballot(act);
            QP_MARK(1);                                                // chains, scan, search, pick
            LLDA_MARK("cold_check");
            if (__builtin_expect(unsure != 0, 0)) {
                LLDA_MARK("rare_cold");
                // tier 1 (fp64, margin 2^-40) right here, in this layout, for all four documents; what IT cannot decide (~1e-9 of the
                // sites) goes to the exact tier out of line, one document at a time, the whole wavefront playing it in the standard layout
                const int holder = (n >> 1) & (LPD - 1);
                const uint32_t ra_l = (n & 1) ? r2 : r0, rb_l = (n & 1) ? r3 : r1;
                const int bp_h = (gbase | holder) << 2;
                const uint32_t ra = (uint32_t)__builtin_amdgcn_ds_bpermute(bp_h, (int)ra_l), rb = (uint32_t)__builtin_amdgcn_ds_bpermute(bp_h, (int)rb_l);
                const uint64_t t0_w = unsure;                          // documents tier 0 was unsure about
                if (lq == 0 && ((t0_w >> gbase) & Geo::GM) && P.status) atomicAdd(P.status + 1, 1);   // statistics
                int z1;
                // (a document whose row was read as int32 skips tier 1: a count of 2^24 or more is not exact in xv)
                const uint64_t still = ((P.margin_rel < 1.0 ? quad_tier1<LB, PAD>(xv, s_ndk, s_nk0, tid, lq, uniform53(ra, rb), P.alpha, P.beta, P.vbeta,
                                                                        P.margin_rel, vm, z1) : ~0ull) | __ballot(cur.w == 0)) & __ballot(act);
                const bool mine0 = ((t0_w >> gbase) & Geo::GM) != 0;
                zn = mine0 ? z1 : zn;
                uint32_t rows = 0;
#pragma unroll
                for (int r = 0; r < Geo::DPW; ++r)
                    rows |= (((t0_w >> (LPD * r)) & Geo::GM) && ((still >> (LPD * r)) & Geo::GM)) ? (1u << r) : 0u;
                rows = (uint32_t)__builtin_amdgcn_readfirstlane((int)rows);
                while (__builtin_expect(rows != 0, 0)) {
                    const int r = __builtin_ctz(rows);
                    rows &= rows - 1;
                    const int src = r * LPD;
                    const int zo_r = __builtin_amdgcn_readlane(zo, src);
                    int zc = quad_cold<LB, PAD>(s_ndk, s_nk0, (tid & 64) + src, __builtin_amdgcn_readlane(cur.v, src),
                                       __builtin_amdgcn_readlane(f, src), zo_r, (uint32_t)__builtin_amdgcn_readlane((int)ra, src),
                                       (uint32_t)__builtin_amdgcn_readlane((int)rb, src), lane,
                                       (int64_t)__builtin_amdgcn_readlane((int)d, src),          // (llda_sweep: D < 2^31)
                                       (const KParams *)__builtin_amdgcn_kernarg_segment_ptr());
                    if (__builtin_expect(zc < 0, 0)) {
                        zc = zo_r;
                        if (lane == 0 && P.status) atomicOr(P.status, 1);   // no topic with positive probability
                    }
                    zn = (row == r) ? (zc | (quad_rho<LB>(zc) << 9)) : zn;
                }
            }
            QP_MARK(2);                                                // (cold tiers)
            LLDA_MARK("decode");

            // add the site back (LabeledLDA.py:121-125) and take the NEXT site out of its topic: ONE read-modify-write per lane, branch
            // free -- a lane that owns neither rewrites its slot 0 with what is there (the factor is a function of the counts) --, a
            // second pass (rare) for the documents where one lane owns both.  The commit of this site (z and ONE log word, quad lane 0 of
            // every document that has the site: exec mask by hand, no divergent region) and the scalar loads of site n+2 are issued in
            // the shadow of the LDS reads.
            {
                const int zpos = zn & 511, sn = zn >> 9, ln = (zn >> 3) & (LPD - 1);
                cur.zn = zpos;
                decode_old(nxt);
                const bool own_new = act && lq == ln, own_old = more && lq == nxt.lo;
                const int sg = own_new ? sn : own_old ? nxt.so : 0;
                const int ps = own_new ? zpos : own_old ? nxt.zo : (lq << 3);
                const int df = own_new ? f : own_old ? -nxt.f : 0;
                LLDA_MARK("count_update");
                const int w0 = QLDS(s_ndk, sg, tid), k0 = s_nk0[ps];
                QP_MARK(3);                                            // decode, the update's LDS reads issued
                LLDA_MARK("scalars");
                __builtin_amdgcn_s_setprio(QP_BULK);
                int w_next;                                            // word of site n+2 (loaded an iteration ago)
                if constexpr (REC) {
                    prv.v = w_next = pv; prv.f = pf; prv.c = pc;       // the record of site n+2
                    prv.zo = gload_i32(z_b, off_of(n + 2));
                    load_rec(pv, pf, pc, off_of(n + 3));
                } else {
                    w_next = wq;
                    load_scalars(prv, off_of(n + 2));                  // scalars of site n+2 (clamped)
                    wq = gload_i32(word_b, off_of(n + 3));
                }
                QP_MARK(4);                                            // scalar / record loads of the sites ahead issued
                // site n+1: its row (issued an iteration ago) -> fp32, own count out; then the row of site n+2 is issued
                remove_own_packed(nxt.so, (more && lq == nxt.lo) ? nxt.f : 0);
                convert_row(nxt.v, nxt.w, nxt.so, (more && lq == nxt.lo) ? (float)nxt.f : 0.0f);
                QP_MARK(5);                                            // next row converted (this is where the wait for it sits)
                LLDA_MARK("row_prefetch");
                load_row16(w_next, prv.w);
                LLDA_MARK("count_update");
                const int w = w0 + df;                                  // (0 <= n_dk + df < 2^16: no carry into the upper half)
                QLDS(s_ndk, sg, tid) = w;
                const int nd = w & 0xffff, nk = k0 + nd - (int)((uint32_t)w >> 16);
                QLDS(s_pa, sg, tid) = tier0_factor(nd, nk, alpha32, vbeta32);
                if (__builtin_expect(__ballot(own_new && own_old) != 0, 0)) {
                    LLDA_MARK("rare_second_update");
                    if (own_new && own_old) update(nxt.so, nxt.zo, -nxt.f);
                }
                // (the commit comes LAST, a whole iteration in front of the next vector-memory wait: vmcnt counts in order, so a wait for
                // the row behind two stores that have just been issued waits for the stores as well -- measured with the stores in the
                // middle of the iteration and hidden from the compiler in inline assembly: 42.9 instead of 36.6 ms)
#ifndef ABL_NOCOMMIT
                {
                    LLDA_MARK("commit");
                    const uint32_t zoff = opaque_u32(sb + (uint32_t)n * 4u);
                    const LLDA_GLOBAL uint32_t *lp = (const LLDA_GLOBAL uint32_t *)P.commit_log + (uint32_t)(cur.c & 0x7fffffff);
                    const uint32_t word = (uint32_t)zo | ((uint32_t)zpos << 16);
if (lq == 0 && act) {                                // (plain stores: the compiler's vmcnt bookkeeping sees them)
                        gstore_i32(z_b, zoff, zpos);
                        *(LLDA_GLOBAL uint32_t *)lp = word;
                    }
                }
#endif
            }
            LLDA_MARK("loop");
            QP_MARK(6);                                                // row prefetch, count update written, commit
        };
        for (int n = 0;; n += 3) {                                  // (uniform trip count: the longest document of the wavefront)
            site(n, R0, R1, R2);
            if (n + 1 >= maxlen) break;
            site(n + 1, R1, R2, R0);
            if (n + 2 >= maxlen) break;
            site(n + 2, R2, R0, R1);
            if (n + 3 >= maxlen) break;
        }

#ifdef QUAD_PROFILE
        if (blockIdx.x == 0 && tid == 0 && P.status) {
#pragma unroll
            for (int k = 0; k < 8; ++k) atomicAdd(P.status + 8 + k, (int)qp_acc[k]);
            atomicAdd(P.status + 16, maxlen);
        }
#endif
        // document done: fold its n_dk change into the workgroup's n_k accumulator, store the row
        if (valid && len > 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int w = QLDS(s_ndk, quad_rho_of(i, j >> 2, j & 3), tid);
                    o[j] = w & 0xffff;
                    const int dl = o[j] - (int)((uint32_t)w >> 16);
                    if (dl) atomicAdd(&s_nk[(i << IS) + lq * 8 + j], dl);
                }
                v4i a = {o[0], o[1], o[2], o[3]}, b = {o[4], o[5], o[6], o[7]};
                ((v4i *)ndk_row)[i * G + 2 * lq] = a;
                ((v4i *)ndk_row)[i * G + 2 * lq + 1] = b;
            }
        }
    }

    __syncthreads();
    for (int i = tid; i < KP; i += QNT) {
        const int dl = s_nk[i];
        if (dl) atomicAdd(P.n_k_delta + i, dl);
    }
}

// ---------------------------------------------------------------------------------------------
// llda_pack_rows16_all: the 16-bit image of EVERY row of n_kw (layouts of 16 slots per lane, G = 8, 16, 32 lanes: 2 G threads per row;
// the 16-byte pieces of llda_pack_rows16 -- slots 8 j .. 8 j + 7 of standard lane g -- in the order the quad kernel reads them: piece
// (j, g & 1) of quad lane g >> 1 at 16-byte unit (2 j + (g & 1)) * G / 2 + (g >> 1)) and, per row, whether all of its counts fit 16 bits THIS sweep (row16[v] = 1) -- a row that does
// not is read from n_kw itself by the sweep.
// ---------------------------------------------------------------------------------------------
template <int G>
__global__ void __launch_bounds__(256) llda_pack_rows16_all_kernel(const int32_t *__restrict__ n_kw, uint16_t *__restrict__ out,
                                                                   uint8_t *__restrict__ row16, int64_t V)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t w = t / (2 * G);                                   // (a wavefront holds whole rows)
    if (w >= V) return;
    const int c = (int)(t & (2 * G - 1)), j = c / G, g = c & (G - 1);
    const int4 *src = reinterpret_cast<const int4 *>(n_kw + w * (int64_t)(G * 16));
    const int4 a = src[(2 * j) * G + g], b = src[(2 * j + 1) * G + g];
    const int m = a.x | a.y | a.z | a.w | b.x | b.y | b.z | b.w;
    const uint64_t bal = __ballot((unsigned)m > 0xffffu);            // (a negative count is "wide" too)
    const int first = (int)(threadIdx.x & 63) & ~(2 * G - 1);        // the row's first lane
    const bool wide = G == 32 ? bal != 0 : ((bal >> first) & ((1ull << (2 * G % 64)) - 1)) != 0;
    uint4 o;
    o.x = ((uint32_t)a.x & 0xffffu) | ((uint32_t)a.y << 16);
    o.y = ((uint32_t)a.z & 0xffffu) | ((uint32_t)a.w << 16);
    o.z = ((uint32_t)b.x & 0xffffu) | ((uint32_t)b.y << 16);
    o.w = ((uint32_t)b.z & 0xffffu) | ((uint32_t)b.w << 16);
    reinterpret_cast<uint4 *>(out + w * (int64_t)(G * 16))[(2 * j + (g & 1)) * (G / 2) + (g >> 1)] = o;      // piece major (load_row16)
    if (c == 0) row16[w] = wide ? 0 : 1;
}

}  // namespace
