// draw_tiers.hpp -- tier-0 (fp32) decision, cold tiers (fp64 decision / exact pipeline), commit of a site, per-site random bits
// Part of the single translation unit llda_gibbs.hip (included in order; see the contents list there).
#pragma once

namespace {

// ---------------------------------------------------------------------------------------------
// Two-tier draw (FAST kernels).  The integer state only depends on WHICH topic the draw picks, i.e. on
// the signs of  E[g][s] = q[g][s] - (t - X[g-1])  in the exact fp64 pipeline above.  Tier 1 evaluates
// the same comparison from unnormalised, cheaply rounded scores
//     w~ = a * (num_b * RN(1/den_b)),   Q~ = prefix(w~),   X~ = scan,   T~ = u * X~[G-1] - X~[g-1]
// (no division, no pairwise sum, no normalisation).  Relative to the total every quantity differs from
// its exact counterpart by at most a few hundred units of 2^-53 (DESIGN.md section 4.3 derives
// |E~ - E| <= 2^-44 of the total), so whenever every |Q~ - T~| exceeds 2^-40 of the total the signs -- and
// with them the chosen position -- are those of the exact pipeline.  Otherwise (probability ~1e-9 per
// site) the group falls back to the exact tier.  Returns false when the group must fall back.
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// Tier 0: the same decision in fp32.  Every compared quantity is within 105 * 2^-24 (< 2^-17.2) of the total of
// its real-number value, the exact pipeline within 2^-44; with a margin of 2^-17 = 128 *
// 2^-24 of the total a "sure" fp32 decision therefore has the signs of the exact pipeline.  About 0.6 % of the
// sites (K = 512) are "unsure" and go on to tier 1.
// ---------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f32(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xF, false));
}

// pa = the lane's cached tier-0 factors (s_pa[.][tid]), read by the caller at the top of the site so that their LDS
// latency is covered by the wait for the row
// (XT: the counts as int32, or already converted -- exactly -- to fp32 by the caller)
template <int T, bool DENSE, int S = 0, class XT = int>
__device__ __forceinline__ void prefix_scores_f32(float (&qw)[T], const XT (&x)[T], const float (&pa)[T], uint32_t mask, float beta)
{
    if constexpr (DENSE && T % 2 == 0 && S + 1 < T) {
        // two slots per instruction: (x + beta) * factor as packed fp32 (v_pk_add_f32, v_pk_mul_f32); a wave64
        // VALU instruction takes 4 cycles whether it produces one or two fp32 results per lane
        typedef float v2f __attribute__((ext_vector_type(2)));
        const v2f xf = {(float)x[S], (float)x[S + 1]}, b2 = {beta, beta};
#ifdef ABL_NOFMA
        const v2f p2 = {pa[S], pa[S + 1]};
        const v2f ws = (xf + b2) * p2;
        if constexpr (S == 0) qw[0] = ws.x;
        else qw[S] = qw[S - 1] + ws.x;
        qw[S + 1] = qw[S] + ws.y;
#else
        // the product and the running sum in ONE fused instruction per slot (an explicit fma: the translation unit is built
        // with -ffp-contract=off): 8 v_pk_add + 16 v_fma instead of 8 + 8 v_pk_mul + 16 v_add.  One rounding where the
        // unfused form has two, so the error bound of section 4.3 (one v per rounding) holds a fortiori; the sums stay
        // non-decreasing (RN is monotone and the products are >= 0).
        const v2f nb = xf + b2;
        if constexpr (S == 0) qw[0] = nb.x * pa[0];
        else qw[S] = __builtin_fmaf(nb.x, pa[S], qw[S - 1]);
        qw[S + 1] = __builtin_fmaf(nb.y, pa[S + 1], qw[S]);
#endif
        prefix_scores_f32<T, DENSE, S + 2>(qw, x, pa, mask, beta);
    } else if constexpr (S < T) {
        float ws = ((float)x[S] + beta) * pa[S];                 // num_b * fl32(a / den_b)
        if constexpr (!DENSE) ws = __int_as_float(__float_as_int(ws) & onehot_bit<S>(mask));
        if constexpr (S == 0) qw[0] = ws;
        else qw[S] = qw[S - 1] + ws;
        prefix_scores_f32<T, DENSE, S + 1>(qw, x, pa, mask, beta);
    }
}

// inclusive scan over the G lanes of a group (any association order will do here)
template <int G>
__device__ __forceinline__ float group_scan_f32(float X, int lig)
{
    if constexpr (G == 8) {
        float y;
        y = dpp_f32<DPP_ROW_SHR + 1>(X); X += (lig >= 1) ? y : 0.0f;
        y = dpp_f32<DPP_ROW_SHR + 2>(X); X += (lig >= 2) ? y : 0.0f;
        y = dpp_f32<DPP_ROW_SHR + 4>(X); X += (lig >= 4) ? y : 0.0f;
    } else {
        X += dpp_f32<DPP_ROW_SHR + 1>(X);
        X += dpp_f32<DPP_ROW_SHR + 2>(X);
        X += dpp_f32<DPP_ROW_SHR + 4>(X);
        X += dpp_f32<DPP_ROW_SHR + 8>(X);
        if constexpr (G >= 32) X += dpp_f32<0x142, 0xA>(X);     // row_bcast:15 into rows 1 and 3
        if constexpr (G == 64) X += dpp_f32<0x143, 0xC>(X);     // row_bcast:31 into rows 2 and 3
    }
    return X;
}

template <int G>
__device__ __forceinline__ float bcast_last_f32(float x, int lane)
{
    if constexpr (G == 64) {
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
    } else if constexpr (G == 32) {
        const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 31));
        const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
        return (lane & 32) ? b : a;
    } else if constexpr (G == 16) {
        const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 15));
        const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 31));
        const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 47));
        const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
        const float ab = (lane & 16) ? b : a, cd = (lane & 16) ? d : c;
        return (lane & 32) ? cd : ab;
    } else {
        return __int_as_float(group_pick<G>(__float_as_int(x), G - 1, lane & (G - 1)));
    }
}

// q[0..T-1] is non-decreasing (prefix sums of non-negative terms): cnt = #{s : q[s] <= lo} and whether NO q[s]
// lies in (lo, hi].  T == 16: a branch-free binary search (5 compares) that also tracks the smallest element
// found above lo -- that element is q[cnt], and the gap is clean iff it is above hi too; 30 instructions
// instead of 64 for two linear counts.
template <int T>
__device__ __forceinline__ void count_sorted_f32(const float (&q)[T], float lo, float hi, int &cnt, bool &clean)
{
    if constexpr (T == 16) {
        const bool c5 = q[15] <= lo;                                   // then every element is
        float ub = c5 ? __int_as_float(0x7f800000) : q[15];            // smallest element known to be > lo
        const bool c1 = q[7] <= lo;
        ub = c1 ? ub : q[7];
        const float m2 = c1 ? q[11] : q[3];                            // the later probes, pre-selected by c1
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
        cnt = c5 ? 16 : ((c1 ? 8 : 0) | (c2 ? 4 : 0) | (c3 ? 2 : 0) | (c4 ? 1 : 0));
        clean = ub > hi;
    } else {
        int cnt_lo = 0, cnt_hi = 0;
#pragma unroll
        for (int s = 0; s < T; ++s) {
            cnt_lo += (q[s] <= lo) ? 1 : 0;
            cnt_hi += (q[s] <= hi) ? 1 : 0;
        }
        cnt = cnt_lo;
        clean = cnt_lo == cnt_hi;
    }
}

// lane predicate "my group's half / the whole of the uniform 64-bit mask b is non-zero" for 32- and 64-lane groups:
// scalar instructions and an inverse ballot instead of per-lane 64-bit shifts and compares (the kernel is bound by
// VALU issue; the scalar unit idles)
// upper half of a uniform 64-bit value, hidden from the optimiser (it would turn "hi != 0" into a 64-bit compare,
// which only the vector unit has)
__device__ __forceinline__ uint32_t uniform_hi32(uint64_t b)
{
    uint32_t hi = (uint32_t)(b >> 32);
    asm("" : "+s"(hi));
    return hi;
}
template <int G>
__device__ __forceinline__ uint64_t spread_any(uint64_t b)
{
    static_assert(G == 32 || G == 64, "wide groups only");
    if constexpr (G == 64) {
        return b ? ~0ull : 0ull;
    } else {
        const uint32_t lo = (uint32_t)b, hi = uniform_hi32(b);
        return (lo ? 0xFFFFFFFFull : 0ull) | (hi ? 0xFFFFFFFF00000000ull : 0ull);
    }
}

// gp_all = __ballot(mask != 0) of the whole wavefront (uniform) for G >= 32, the group's own bits (per lane) below
// Tier 0 for a dense mask and one or two documents per wavefront (the 16-slot K = 512 / 1024 kernels, bound by instruction
// issue: every instruction below is one the site pays for).  Returns the wavefront's ballot of the lanes that are NOT sure
// (uniform; 0 on the way all but one wavefront in a hundred take -- which half of it is unsure is the caller's business, behind
// one scalar branch) and the position every lane's group drew in zn.  Every slot is allowed, so "first allowed slot above lo"
// is the binary search's count itself, and the lane writes it down as the DEVICE POSITION it stands for right away:
// pos_of<G,16>(lane, s) = (s >> 2) * 4G | lane << 2 | (s & 3), i.e. the search's four outcomes are four bits of the position
// (a lane with q[15] <= lo has all four set and reads "slot 15").  The hit lane is the first lane with a slot above lo; were
// rounding to leave none, the group's last lane answers with its last slot -- its bit is forced into the ballot -- which is what
// the general path's my_miss / last-allowed-lane fall-back returns for a dense mask.
template <int G>
__device__ __forceinline__ uint64_t draw_fast_dense_f32(const float (&qw)[16], float u, float margin_rel, int lig, int lane, int &zn)
{
    static_assert(G == 32 || G == 64, "one or two documents per wavefront");
    // inclusive scan over the lanes of the group: four row_shr steps inside the rows of 16, then the last lane of row 0 (2) into
    // row 1 (3), and for 64 lanes the last lane of row 1 into rows 2 and 3.  The cross-row steps are ONE instruction each -- a DPP
    // add whose row mask leaves the other rows alone (the compiler's form is v_mov 0 + v_mov_dpp + v_add: it does not fold a
    // partial row mask into a floating-point add) -- written by hand together with the wait states a DPP source and a v_readlane
    // of a freshly written register need (the hazard recogniser does not look inside inline assembly).
    LLDA_MARK("lane_scan");
    float X = qw[15];
    X += dpp_f32<DPP_ROW_SHR + 1>(X);
    X += dpp_f32<DPP_ROW_SHR + 2>(X);
    X += dpp_f32<DPP_ROW_SHR + 4>(X);
    X += dpp_f32<DPP_ROW_SHR + 8>(X);
    if constexpr (G == 32) {
        asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1" : "+v"(X));
    } else {
        asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
                     "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1" : "+v"(X));
    }
    const float tot = bcast_last_f32<G>(X, lane);
    // X of the lane below; 0.0 shifted into lane 0 (bound_ctrl: no register to preset)
    const float prev = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(X), DPP_WAVE_SHR1, 0xF, 0xF, true));
    LLDA_MARK("threshold");
    const float tg = u * tot - (lig ? prev : 0.0f);
    const float margin = tot * margin_rel;
    const float lo = tg - margin, hi = tg + margin;
    LLDA_MARK("search");
    const bool c5 = qw[15] <= lo;
    float ub = c5 ? __int_as_float(0x7f800000) : qw[15];           // smallest element known to be > lo (count_sorted_f32)
    const bool c1 = qw[7] <= lo;
    ub = c1 ? ub : qw[7];
    const float m2 = c1 ? qw[11] : qw[3];
    const float a1 = c1 ? qw[9] : qw[1], a5 = c1 ? qw[13] : qw[5];
    const float e0 = c1 ? qw[8] : qw[0], e2 = c1 ? qw[10] : qw[2], e4 = c1 ? qw[12] : qw[4], e6 = c1 ? qw[14] : qw[6];
    const bool c2 = m2 <= lo;
    ub = c2 ? ub : m2;
    const float m3 = c2 ? a5 : a1;
    const float g0 = c2 ? e4 : e0, g2 = c2 ? e6 : e2;
    const bool c3 = m3 <= lo;
    ub = c3 ? ub : m3;
    const float m4 = c3 ? g2 : g0;
    const bool c4 = m4 <= lo;
    ub = c4 ? ub : m4;
    // the general path's guards on the total (tot > 0, margin < tot, tot finite) in one class test: tot - margin is a positive
    // normal number exactly when all three hold (a difference down in the denormals counts as a failure; under FAST the total
    // is above 1e-24).  0x2FF = every class but "positive normal".
    // (v_cmp_class by hand: the builtin's result reaches the ballot through a select and a second compare.  margin_rel >= 1 --
    // tier 0 switched off, llda_sweep_args.debug_margin -- fails the test too: the caller needs no test of its own.)
    uint64_t bad_total;
    asm("v_cmp_class_f32_e64 %0, %1, %2" : "=s"(bad_total) : "v"(tot - margin), "v"(0x2FF));
    const uint64_t unsure = __ballot(!(ub > hi)) | bad_total;
    LLDA_MARK("pick");
    const int p = (c1 ? 8 * G : 0) | (c2 ? 4 * G : 0) | (lig << 2) | (c3 ? 2 : 0) | (c4 ? 1 : 0);
    const uint64_t above = __ballot(!c5);
    if constexpr (G == 64) {
        zn = __builtin_amdgcn_readlane(p, (int)__builtin_ctzll(above | (1ull << 63)));
    } else {
        const uint32_t f0 = (uint32_t)above | 0x80000000u, f1 = uniform_hi32(above) | 0x80000000u;
        const int z0 = __builtin_amdgcn_readlane(p, (int)__builtin_ctz(f0));
        const int z1 = __builtin_amdgcn_readlane(p, (int)__builtin_ctz(f1) + 32);
        zn = (lane & 32) ? z1 : z0;
    }
    return unsure;
}

template <int G, int T>
__device__ __forceinline__ bool draw_fast_f32(const float (&qw)[T], float u, uint32_t mask, uint64_t gp,
                                              float margin_rel, int lig, int lane, int &zn)
{
    const float X = group_scan_f32<G>(qw[T - 1], lig);
    const float tot = bcast_last_f32<G>(X, lane);
    const float prev = dpp_f32<DPP_WAVE_SHR1>(X);
    const float tg = u * tot - (lig ? prev : 0.0f);
    const float margin = tot * margin_rel;
    const float lo = tg - margin, hi = tg + margin;
    int cnt_lo;
    bool clean;
    count_sorted_f32<T>(qw, lo, hi, cnt_lo, clean);
    if constexpr (G >= 32) {
        // one or two groups per wavefront: everything that is the same for a whole group is computed once, on the
        // scalar unit, from the ballots (a ballot of a comparison IS the comparison's result mask; the ballot of an
        // OR of them costs two vector instructions more)
        const uint64_t ub = __ballot(!clean) | __ballot(!(tot > 0.0f)) | __ballot(!(margin < tot)) | __ballot(!(tot < 3.0e38f));
        if (__builtin_expect(__builtin_amdgcn_inverse_ballot_w64(spread_any<G>(ub)), 0)) return false;
        const uint32_t fm = mask & (0xFFFFu << cnt_lo);
        const uint64_t gf = __ballot(fm != 0);
        const int my_hit = (int)__ffs((int)(fm | 0x10000u)) - 1, my_miss = 31 - (int)__clz((int)(mask | 1u));
        const int my = __builtin_amdgcn_inverse_ballot_w64(spread_any<G>(gf)) ? my_hit : my_miss;
        if constexpr (G == 64) {
            const int sl = gf ? (int)__builtin_ctzll(gf) : 63 - (int)__builtin_clzll(gp | 1ull);
            zn = pos_of<G, T>(sl, __builtin_amdgcn_readlane(my, sl));
        } else {
            const uint32_t f0 = (uint32_t)gf, f1 = uniform_hi32(gf);
            const uint32_t p0 = (uint32_t)gp, p1 = uniform_hi32(gp);
            const int sl0 = f0 ? (int)__builtin_ctz(f0) : 31 - (int)__builtin_clz(p0 | 1u);
            const int sl1 = f1 ? (int)__builtin_ctz(f1) : 31 - (int)__builtin_clz(p1 | 1u);
            const int z0 = pos_of<G, T>(sl0, __builtin_amdgcn_readlane(my, sl0));
            const int z1 = pos_of<G, T>(sl1, __builtin_amdgcn_readlane(my, sl1 + 32));
            zn = (lane & 32) ? z1 : z0;
        }
        return true;
    } else {
        const bool unsure = !clean || !(tot > 0.0f) || !(margin < tot) || !(tot < 3.0e38f);
        const int gbase = lane & ~(G - 1);
        const uint64_t gmask = (1ull << G) - 1ull;
        if (((__ballot(unsure) >> gbase) & gmask) != 0) return false;
        const uint32_t fm = mask & (0xFFFFu << cnt_lo);
        const uint64_t gf = (__ballot(fm != 0) >> gbase) & gmask;
        const bool hit = gf != 0;
        const int sl = hit ? (int)__ffsll((unsigned long long)gf) - 1 : 63 - (int)__clzll((unsigned long long)(gp | 1ull));
        const int my = hit ? (int)__ffs((int)(fm | 0x10000u)) - 1 : 31 - (int)__clz((int)(mask | 1u));
        zn = pos_of<G, T>(sl, group_pick<G>(my, sl, lig));
        return true;
    }
}

// Cold tiers of the FAST kernels, out of line: they run for the ~0.65 % of the sites
// tier 0 is unsure about and work on scratch copies, so the hot loop's register allocation never sees them.
//   tier 1: the decision from unnormalised fp64 prefix sums, margin 2^-40 of the total;
//   exact : the reference's fp64 pipeline bit for bit -- scores, numpy-ordered sum, p = fl(w/S) through
//           ONE IEEE reciprocal y = RN(1/S) and two residual corrections per slot (Markstein: with y
//           correctly rounded and q1 faithful, q2 = RN(q1 + (w - S q1) y) is the correctly rounded
//           quotient; llda_selftest_div checks it against the hardware division), keyed draw.
// Returns the chosen device position or -1.
// ACC: where the document's counts live -- nd(s) = its n_dk at slot s of this lane, nk(s) = the n_k it sees there,
// count_unsure() / stats() = this lane group reports "tier 0 was unsure" / "the exact tier ran" (one group per site)
template <int G, int T, bool HAS_TAIL, bool DENSE, class ACC>
__device__ __noinline__ int cold_tiers_acc(const ACC dc, const int *x, uint32_t mask, double u, int lig, int lane, const KParams *P)
{
    auto nd_of = [&](int s) { return dc.nd(s); };
    auto nk_of = [&](int s) { return dc.nk(s); };
    // Tier 1 is unrolled and lives in registers (the kernel is LDS-limited to 3 waves per SIMD, which leaves 168
    // VGPRs: scratch round trips here stalled the whole wave); the exact tier, ~1e-9 per site, stays rolled over
    // a scratch array.
    const int gbase = lane & ~(G - 1);
    const uint64_t gmask = (G == 64) ? ~0ull : ((1ull << G) - 1ull);
    const double alpha = P->alpha, beta = P->beta, vbeta = P->vbeta;
    const uint32_t lmask = DENSE ? 0xFFFFu : mask;
    double w[T];
    if (lig == 0 && dc.count_unsure() && P->status) atomicAdd(P->status + 1, 1);      // statistics: sites tier 0 was unsure about
    // ---- tier 1: unnormalised fp64 prefix sums, margin 2^-40 of the total ----
    {
        double run = 0.0;
#pragma unroll
        for (int s = 0; s < T; ++s) {
            // 1/den to within 2^-50: hardware estimate + two Newton steps (tier 1 only needs a few 2^-53)
            const double den = (double)nk_of(s) + vbeta;
            double y = __builtin_amdgcn_rcp(den);
            y = __builtin_fma(__builtin_fma(-den, y, 1.0), y, y);
            y = __builtin_fma(__builtin_fma(-den, y, 1.0), y, y);
            const double ws = ((double)nd_of(s) + alpha) * (((double)x[s] + beta) * y);
            run = run + (((lmask >> s) & 1u) ? ws : 0.0);
            w[s] = run;
        }
        const double X = group_scan<G>(run, lig);
        const double tot = bcast_last<G>(X, lane);
        const double prev = dpp_f64<DPP_WAVE_SHR1>(X);
        const double tg = u * tot - (lig ? prev : 0.0);
        const double margin = tot * P->margin_rel;
        int cnt_lo = 0, cnt_hi = 0;
#pragma unroll
        for (int s = 0; s < T; ++s) {
            cnt_lo += (w[s] <= tg - margin) ? 1 : 0;
            cnt_hi += (w[s] <= tg + margin) ? 1 : 0;
        }
        const bool unsure = (cnt_lo != cnt_hi) || !(tot > 0.0) || !(margin < tot);
        if (((__ballot(unsure) >> gbase) & gmask) == 0) {
            const uint32_t fm = mask & (0xFFFFu << cnt_lo);
            const uint64_t gf = (__ballot(fm != 0) >> gbase) & gmask;
            const uint64_t gp = (__ballot(mask != 0) >> gbase) & gmask;
            const bool hit = gf != 0;
            const int sl = hit ? (int)__ffsll((unsigned long long)gf) - 1 : 63 - (int)__clzll((unsigned long long)(gp | 1ull));
            const int my = hit ? (int)__ffs((int)(fm | 0x10000u)) - 1 : 31 - (int)__clz((int)(mask | 1u));
            return pos_of<G, T>(sl, __shfl(my, sl, G));
        }
    }
    // ---- exact tier: the reference's fp64 pipeline, bit for bit ----
    if (lig == 0 && dc.stats() && P->status) { atomicOr(P->status, 2); atomicAdd(P->status + 2, 1); }   // the exact tier ran
    const int leaf = lig >> 3;
    double acc = 0.0, tv = 0.0;
#pragma unroll 1
    for (int s = 0; s < T; ++s) {
        // prob = lab * a * (num_b / den_b)   (LabeledLDA.py:113-116)
        const double ws = ((double)nd_of(s) + alpha) * (((double)x[s] + beta) / ((double)nk_of(s) + vbeta));
        const double v = ((lmask >> s) & 1u) ? ws : 0.0;
        w[s] = v;
        if (HAS_TAIL && s == P->tail_row && leaf == P->last_leaf) tv = v;
        else acc = acc + v;
    }
    const double S = group_sum_tail<G, HAS_TAIL>(acc, tv, *P, lig, lane);      // np.sum(prob), LabeledLDA.py:117
    const double y = 1.0 / S;
    // prob /= np.sum(prob); keyed draw: per-lane prefix, Hillis-Steele scan, first slot with p > 0 and q > t - X[g-1]
    double run = 0.0;
#pragma unroll 1
    for (int s = 0; s < T; ++s) {
        run = (s == 0) ? div_by(w[s], S, y) : run + div_by(w[s], S, y);
        w[s] = run;
    }
    const double X = group_scan<G>(run, lig);
    const double tot = bcast_last<G>(X, lane);
    const double prev = dpp_f64<DPP_WAVE_SHR1>(X);
    const double tg = u * tot - (lig ? prev : 0.0);
    int cnt = 0;
#pragma unroll 1
    for (int s = 0; s < T; ++s) cnt += (w[s] <= tg) ? 1 : 0;
    const uint32_t fm = mask & (0xFFFFu << cnt);
    const uint64_t gf = (__ballot(fm != 0) >> gbase) & gmask;
    const uint64_t gp = (__ballot(mask != 0) >> gbase) & gmask;
    if (gp == 0 || !(S > 0.0)) return -1;
    const bool hit = gf != 0;
    const int sl = hit ? (int)__ffsll((unsigned long long)gf) - 1 : 63 - (int)__clzll((unsigned long long)gp);
    const int my = hit ? (int)__ffs((int)(fm | 0x10000u)) - 1 : 31 - (int)__clz((int)(mask | 1u));
    return pos_of<G, T>(sl, __shfl(my, sl, G));
}

// the counts of the LDS-staged kernels (kernel_sweep.hpp): [slot][thread] arrays.
// W4: s_ndk holds n_dk | sweep-start n_dk << 16 and s_nkc is really the workgroup's copy of the sweep-start n_k, indexed by
// device position (kernel_sweep.hpp, count_update_w4)
template <int G, int T, bool W4>
struct LdsCounts {
    const int (*s_ndk)[256];
    const int (*s_nkc)[256];
    int tid, lig;
    __device__ __forceinline__ int nd(int s) const { return W4 ? s_ndk[s][tid] & 0xffff : s_ndk[s][tid]; }
    __device__ __forceinline__ int nk(int s) const
    {
        if constexpr (W4) {
            const int w = s_ndk[s][tid];
            return ((const int *)s_nkc)[pos_of<G, T>(lig, s)] + (w & 0xffff) - (int)((uint32_t)w >> 16);
        } else {
            return s_nkc[s][tid];
        }
    }
    __device__ __forceinline__ bool stats() const { return true; }
    __device__ __forceinline__ bool count_unsure() const { return true; }
};
template <int G, int T, bool HAS_TAIL, bool DENSE, bool W4 = false>
__device__ __forceinline__ int cold_tiers(const int (*s_ndk)[256], const int *x, const int (*s_nkc)[256], int tid,
                                          uint32_t mask, double u, int lig, int lane, const KParams *P)
{
    const LdsCounts<G, T, W4> acc{s_ndk, s_nkc, tid, lig};
    return cold_tiers_acc<G, T, HAS_TAIL, DENSE>(acc, x, mask, u, lig, lane, P);
}

// store the new assignment of a site and move its count in n_kw_delta (int32 atomics, no return value)
// (c = the site's position in the commit log when there is one; v, f are only needed without a log)
template <class PR>      // PR: const KParams & or a reference into the kernel-argument segment (kernarg_fresh)
__device__ __forceinline__ void commit_site(PR &P, int64_t i, int v, int f, int zo, int zn, int c, int KP)
{
    P.z[i] = zn;
    uint32_t *log = P.commit_log;
    if (log) {
        log[c] = (uint32_t)zo | ((uint32_t)zn << 16);
    } else if (zn != zo) {
        int32_t *row = P.n_kw_delta + (int64_t)v * KP;
        atomicAdd(row + zo, -f);
        atomicAdd(row + zn, f);
    }
}

// the same through base + 32-bit byte offsets (site_off = 4 * site index relative to z_base, c = log position)
template <bool LOGGED>
__device__ __forceinline__ void commit_site_off(const KParams &P, int32_t *z_base, uint32_t site_off, int v, int f, int zo,
                                                int zn, int c, int KP)
{
    gstore_i32(z_base, site_off, zn);
    if (LOGGED) {
        ((LLDA_GLOBAL uint32_t *)P.commit_log)[(uint32_t)c] = (uint32_t)zo | ((uint32_t)zn << 16);   // (log of any length)
    } else if (zn != zo) {
        int32_t *row = P.n_kw_delta + (int64_t)v * KP;
        atomicAdd(row + zo, -f);
        atomicAdd(row + zn, f);
    }
}

// ---------------------------------------------------------------------------------------------
// The sweep kernel
// ---------------------------------------------------------------------------------------------
// Per-site pieces shared by the two sweep kernels ------------------------------------------------
// random bits of site n: one Philox block serves sites 2b and 2b+1; the G lanes of the group compute G
// consecutive blocks at once (every 2G sites, or at the first site of a resumed document) and hand them out
template <int G>
__device__ __forceinline__ void site_random_bits(const KParams &P, int n, bool first, uint32_t gdoc, int lig,
                                                 uint32_t &r0, uint32_t &r1, uint32_t &r2, uint32_t &r3,
                                                 uint32_t &ra, uint32_t &rb)
{
    if (__builtin_expect(first || (n & (2 * G - 1)) == 0, 0)) {
        r0 = (uint32_t)((n >> 1) & ~(G - 1)) + (uint32_t)lig; r1 = gdoc; r2 = P.stream_id; r3 = P.sweep;
        philox4x32_10(r0, r1, r2, r3, P.key0, P.key1);
    }
    const int holder = (n >> 1) & (G - 1);
    ra = (uint32_t)group_pick<G>((int)((n & 1) ? r2 : r0), holder, lig);
    rb = (uint32_t)group_pick<G>((int)((n & 1) ? r3 : r1), holder, lig);
}

// the 53-bit keyed uniform u = ((a >> 5) * 2^26 + (b >> 6)) / 2^53 (every operation exact)
__device__ __forceinline__ double uniform53(uint32_t ra, uint32_t rb)
{
    return ((double)(ra >> 5) * 67108864.0 + (double)(rb >> 6)) * (1.0 / 9007199254740992.0);
}

template <int G>
__device__ __forceinline__ double site_uniform(const KParams &P, int n, bool first, uint32_t gdoc, int lig,
                                               uint32_t &r0, uint32_t &r1, uint32_t &r2, uint32_t &r3)
{
    uint32_t ra, rb;
    site_random_bits<G>(P, n, first, gdoc, lig, r0, r1, r2, r3, ra, rb);
    return uniform53(ra, rb);
}

}  // namespace
