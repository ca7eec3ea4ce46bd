// kernel_sparse.hpp -- llda_sweep_sparse_kernel: one lane per allowed topic
// Part of the single translation unit llda_gibbs.hip (included in order; see the contents list there).
#pragma once

namespace {

// ---------------------------------------------------------------------------------------------
// Sweep kernel for sparse label sets (Labeled LDA proper: a handful of allowed topics out of hundreds,
// e.g. 4.6 of 392 on the abstracts corpus).  One lane per ALLOWED topic, GS = 8..64 lanes per document
// (64/GS documents per wavefront); per site each lane gathers its single n_kw entry, so the traffic is
// 4*A + 32 bytes instead of a 4*KP-byte row.  All per-topic state (n_dk, the n_k the document sees, the
// reciprocal of n_k + V*beta) is a scalar register of the owning lane.
// The draw is the tier-1 decision of DESIGN.md section 4.3 restricted to the live topics: inclusive scan of
// the unnormalised fp64 scores in device-position order, first lane with Q > u*total, sure when every
// |Q - u*total| exceeds 2^-40 of the total.  A document with an unsure site (probability ~1e-11 per site)
// is handed to the dense tiered kernel, which continues from that site (resume list).
// Preconditions as for llda_sweep_kernel (alpha, beta >= 1e-6, V*beta < 2^40).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double rcp_newton(double den)
{
    double y = __builtin_amdgcn_rcp(den);                    // hardware estimate, then two Newton steps:
    y = __builtin_fma(__builtin_fma(-den, y, 1.0), y, y);    // within a few 2^-53 of 1/den
    return __builtin_fma(__builtin_fma(-den, y, 1.0), y, y);
}

// inclusive scan over the GS lanes of a group, any association order (DPP within 16-lane rows + row carries)
template <int GS>
__device__ __forceinline__ double scan_any_f64(double X, int lig)
{
    if constexpr (GS == 8) {
        double y;
        y = dpp_f64<DPP_ROW_SHR + 1>(X); X = X + ((lig >= 1) ? y : 0.0);
        y = dpp_f64<DPP_ROW_SHR + 2>(X); X = X + ((lig >= 2) ? y : 0.0);
        y = dpp_f64<DPP_ROW_SHR + 4>(X); X = X + ((lig >= 4) ? y : 0.0);
        return X;
    } else {
        X = X + dpp_f64<DPP_ROW_SHR + 1>(X);
        X = X + dpp_f64<DPP_ROW_SHR + 2>(X);
        X = X + dpp_f64<DPP_ROW_SHR + 4>(X);
        X = X + dpp_f64<DPP_ROW_SHR + 8>(X);
        if constexpr (GS >= 32) {           // carry the totals of the 16-lane rows upwards, row by row
            const double c1 = __shfl(X, 15, GS);
            X = X + ((lig >= 16 && lig < 32) ? c1 : 0.0);
        }
        if constexpr (GS == 64) {
            const double c2 = __shfl(X, 31, GS);
            X = X + ((lig >= 32 && lig < 48) ? c2 : 0.0);
            const double c3 = __shfl(X, 47, GS);
            X = X + ((lig >= 48) ? c3 : 0.0);
        }
        return X;
    }
}

// value of lane J (J < 8, compile time) of the caller's GS-lane group, in every lane of the group
template <int GS, int J>
__device__ __forceinline__ int bcast_lane(int v, int lig)
{
    if constexpr (GS <= 16) {
        constexpr int QP = (J & 3) * 0x55;                                  // quad_perm [j,j,j,j]
        const int q = __builtin_amdgcn_update_dpp(0, v, QP, 0xF, 0xF, false);
        int r;
        if constexpr ((J >> 2) == 0) {                                      // source quad is the lower one of its 8
            const int up = __builtin_amdgcn_update_dpp(0, q, DPP_ROW_SHR + 4, 0xF, 0xF, false);
            r = (lig & 4) ? up : q;
        } else {
            const int dn = __builtin_amdgcn_update_dpp(0, q, 0x100 + 4, 0xF, 0xF, false);   // row_shl:4
            r = (lig & 4) ? q : dn;
        }
        if constexpr (GS == 16) {                                           // upper 8 lanes take it from the lower 8
            const int up8 = __builtin_amdgcn_update_dpp(0, r, DPP_ROW_SHR + 8, 0xF, 0xF, false);
            r = (lig & 8) ? up8 : r;
        }
        return r;
    } else {
        return __shfl(v, J, GS);
    }
}

// sum over the GS lanes of a group in every lane (any association order)
template <int GS>
__device__ __forceinline__ double allsum_any_f64(double x, int lane)
{
    x = x + dpp_f64<DPP_XOR1>(x);
    x = x + dpp_f64<DPP_XOR2>(x);
    x = x + dpp_f64<DPP_HALF_MIRROR>(x);
    if constexpr (GS >= 16) x = x + dpp_f64<DPP_ROW_ROR + 8>(x);
    if constexpr (GS >= 32) x = x + xor16_f64(x, lane);
    if constexpr (GS == 64) x = x + xor32_f64(x, lane);
    return x;
}

// OR over the GS lanes of a group in every lane
template <int GS>
__device__ __forceinline__ int allor_i32(int x, int lane)
{
    x |= __builtin_amdgcn_update_dpp(0, x, DPP_XOR1, 0xF, 0xF, false);
    x |= __builtin_amdgcn_update_dpp(0, x, DPP_XOR2, 0xF, 0xF, false);
    x |= __builtin_amdgcn_update_dpp(0, x, DPP_HALF_MIRROR, 0xF, 0xF, false);
    if constexpr (GS >= 16) x |= __builtin_amdgcn_update_dpp(0, x, DPP_ROW_ROR + 8, 0xF, 0xF, false);
    if constexpr (GS >= 32) x |= __shfl_xor(x, 16, GS);
    if constexpr (GS == 64) x |= __shfl_xor(x, 32, GS);
    return x;
}

// exact_site_wave with the dense layout of the problem: one problem per launch (its layout is in the kernel
// arguments, read through the kernel-argument segment so that nothing of it stays live in the hot loop) ...
__device__ __noinline__ int exact_call(const KParams *K, int, int, double wx, int pos, int base, int A, double u, int lane)
{
    int P2 = 1;
    while (P2 < K->last_leaf + 1) P2 *= 2;
    const int G = 8 * P2;
    return exact_site_wave(wx, pos, base, A, u, G, K->KP / G, K->last_leaf, K->tail, K->tail_row, K->n_rounds, K->xor_tree,
                           K->rounds_pk, lane);
}

// sum over the GS lanes of a group in every lane, fp32 (any association order)
template <int GS>
__device__ __forceinline__ float allsum_any_f32(float x, int lane)
{
    x += dpp_f32<DPP_XOR1>(x);
    x += dpp_f32<DPP_XOR2>(x);
    x += dpp_f32<DPP_HALF_MIRROR>(x);
    if constexpr (GS >= 16) x += dpp_f32<DPP_ROW_ROR + 8>(x);
    if constexpr (GS >= 32) x += __shfl_xor(x, 16, GS);
    if constexpr (GS == 64) x += __shfl_xor(x, 32, GS);
    return x;
}

// One site of the sparse kernel (J = index inside the current batch of 8 sites), branch free on the decided path:
// every lane of the wavefront executes every instruction and the per-group condition "is site J part of this batch?"
// is data -- divergent branches here cost more in exec-mask bookkeeping (SGPR spills) than the arithmetic they skip.
//
// T0 (round 4): the decision is first taken in fp32 -- tier 0 of DESIGN.md section 4.3 restricted to the live topics.  With
// v = 2^-24 and the total normalised to 1: a = fl(ndk + alpha32), num = fl(x + beta32), den = fl(nk + vbeta32) carry <= 3v each
// (conversion of a count >= 2^24, the rounded prior, the sum), y = v_rcp_f32(den) (1 ulp) is within 5v of 1/den, w = fl(a * fl(num * y))
// within 13v; the inclusive scan over <= 64 lanes adds <= 6v, so every Q and the total are within 19v; t = fl(uf * tot) with uf the
// uniform rounded to fp32 within 20.5v; the compared difference fl(Q - t) within 19v + 20.5v + 1v < 41v of its real value, the exact
// pipeline within 2^-44.  A site is SURE when every live lane of its group has |Q - t| > margin0 * tot with margin0 = 2^-17 = 128v
// (KParams.margin0_rel, the dense kernels' margin): all signs are then those of the exact pipeline and the same topic is selected.
// `tot - margin` being a positive normal number is the guard tot > 0, margin < tot, tot finite in one class test.  When ANY site of the
// wavefront is unsure (about 10^-4 per site at production margins; a scalar branch on a ballot) the whole wavefront takes the fp64
// decision below for this site -- which gives the sure groups the same answer and sends what IT cannot decide to the exact pipeline.
//
// The fp64 decision: the reciprocal y of n_k + V*beta is recomputed ONCE per site, after the removal (it then also covers the previous
// site's add-back).  A site the margin cannot decide (~1e-11 per site; the only branch, wave-uniform) is resolved on
// the spot by exact_site_wave(): the reference's fp64 pipeline in the dense layout of the problem (`lay`), run by
// the whole wavefront for that document -- the topic is the exact pipeline's either way.
// (layK, layKP: topics and row length of the caller's problem in the batched kernel -- per lane group; the sparse
// kernel of one problem passes -1 and the layout is read from the kernel arguments in the rare branch that needs it)
template <int GS, int J, class PT>
__device__ __forceinline__ int sparse_decide_f64(const PT &P, int layK, int layKP, int active, int zo, int su_lo, int su_hi, int x,
                                                 bool live, int pos, int A, int ndk, int nk, int lig, int lane, int gbase, uint64_t gmask)
{
    const double u = __hiloint2double(bcast_lane<GS, J>(su_hi, lig), bcast_lane<GS, J>(su_lo, lig));
    const double y = rcp_newton((double)nk + P.vbeta);
    const double w = live ? ((double)ndk + P.alpha) * (((double)x + P.beta) * y) : 0.0;
    const double Q = scan_any_f64<GS>(w, lig);
    const double tot = allsum_any_f64<GS>(w, lane);
    const double t = u * tot, margin = tot * P.margin_rel;
    // (bitwise on purpose: the short-circuit forms compile to exec-mask branches around single compares)
    const bool unsure = (active != 0) & ((live & !(fabs(Q - t) > margin)) | !(tot > 0.0) | !(margin < tot));
    const uint64_t gf = (__ballot(live && Q > t) >> gbase) & gmask;
    const int sel = gf ? (int)__ffsll((unsigned long long)gf) - 1 : A - 1;          // none: last allowed topic
    int zn = allor_i32<GS>((lig == sel) ? pos : 0, lane);
    uint64_t todo = __ballot(unsure);
    if (__builtin_expect(todo != 0, 0)) {
        // exact tier: one undecided document of the wavefront at a time (all 64 lanes take part)
        if (lane == 0 && P.status) { atomicOr(P.status, 2); }
        const double wx = live ? ((double)ndk + P.alpha) * (((double)x + P.beta) / ((double)nk + P.vbeta)) : 0.0;
        while (todo) {
            const int first = (int)__ffsll((unsigned long long)todo) - 1;
            const int base = first & ~(GS - 1);
            todo &= ~(gmask << base);
            const int A_g = __builtin_amdgcn_readlane(A, base);
            const double u_g = readlane_var_f64(u, base);
            // (the kernel-argument segment, taken here in kernel scope: taking &P would pin all of P in registers)
            const int r = exact_call((const PT *)__builtin_amdgcn_kernarg_segment_ptr(), layK, layKP, wx, pos, base, A_g, u_g, lane);
            if ((lane & ~(GS - 1)) == base) {
                zn = r < 0 ? zo : r;
                if (r < 0 && lig == 0 && P.status) atomicOr(P.status, 1);           // no topic with positive probability
                if (lig == 0 && P.status) atomicAdd(P.status + 2, 1);               // statistics: exact-tier sites
            }
        }
    }
    return zn;
}

template <int GS, int J, bool T0, class PT>
__device__ __forceinline__ void sparse_site(const PT &P, int layK, int layKP, int nb, int sf, int sz, int su_lo,
                                            int su_hi, int su_f, const int (&xg)[8], bool live, int pos, int A, int &ndk, int &nk,
                                            int &my_zn, int &done, int lig, int lane, int gbase, uint64_t gmask)
{
    // (the per-group flags are 0 / -1 integers in VGPRs: as `bool`s they would each occupy an SGPR pair)
    const int f = bcast_lane<GS, J>(sf, lig), zo = bcast_lane<GS, J>(sz, lig);
    const int active = (J < nb) ? -1 : 0;
    const int rm = f & active & ((pos == zo) ? -1 : 0);                             // LabeledLDA.py:109-111
    ndk -= rm; nk -= rm;
    const int x = xg[J] - rm;
    int zn;
    if constexpr (T0) {
        const float uf = __int_as_float(bcast_lane<GS, J>(su_f, lig));
        const float yf = __builtin_amdgcn_rcpf((float)nk + P.vbeta32);
        const float wf = live ? ((float)ndk + P.alpha32) * (((float)x + P.beta32) * yf) : 0.0f;
        const float Qf = group_scan_f32<GS>(wf, lig);
        const float totf = allsum_any_f32<GS>(wf, lane);
        const float tf = uf * totf, mf = totf * P.margin0_rel;
        // sure: every live lane is outside the band, and tot - margin is a positive normal number (tot > 0, margin < tot, finite)
        const bool unsure0 = (active != 0) & ((live & !(fabsf(Qf - tf) > mf)) | !__builtin_amdgcn_classf(totf - mf, 0x100));
        const uint64_t cold = __ballot(unsure0);
        if (__builtin_expect(cold == 0, 1)) {
            const uint64_t gf = (__ballot(live && Qf > tf) >> gbase) & gmask;
            const int sel = gf ? (int)__ffsll((unsigned long long)gf) - 1 : A - 1;
            zn = allor_i32<GS>((lig == sel) ? pos : 0, lane);
        } else {
            if (lane == 0 && P.status) {                                            // statistics: sites the fp32 tier left undecided
                int n = 0;
                for (int g = 0; g < 64; g += GS) n += ((cold >> g) & gmask) ? 1 : 0;
                atomicAdd(P.status + 1, n);
            }
            zn = sparse_decide_f64<GS, J>(P, layK, layKP, active, zo, su_lo, su_hi, x, live, pos, A, ndk, nk, lig, lane, gbase, gmask);
        }
    } else {
        zn = sparse_decide_f64<GS, J>(P, layK, layKP, active, zo, su_lo, su_hi, x, live, pos, A, ndk, nk, lig, lane, gbase, gmask);
    }
    // add the site back (LabeledLDA.py:121-125)
    const int back = active & f & ((pos == zn) ? -1 : 0);
    ndk += back; nk += back;
    // (lig through an opaque move: the eight loop-invariant masks lig == J would otherwise live in 16 SGPRs)
    my_zn = (active & (((int)opaque_u32((uint32_t)lig) == J) ? -1 : 0)) ? zn : my_zn;
    done -= active;                                                                 // active is 0 or -1
}


__device__ void sparse_wide_init();

// PT = KParams (narrow layouts) or WSParams (kernel_wide.hpp: layouts with more than 8 pairwise leaves -- the arithmetic
// of the decided sites does not know the layout at all; what differs is the exact tier an undecided site takes and that the
// n_k changes go straight to global atomics instead of a workgroup accumulator of KP ints)
template <class PT> struct sparse_is_wide { static constexpr bool value = false; };

// IMG = 8 / 16: the gathers read a NARROW IMAGE of n_kw (P.img: one byte / one 16-bit word per count, saturating; written by
// llda_pack_image at the start of the sweep) -- a row spans a quarter / half as many 128-byte lines, so the eight-odd gathers of a
// site fall into fewer distinct lines and four / two times as many rows stay in the L2s: the kernel is bound by the L2's line
// fills (profiles/r03k_summary.md: 4.2 fills per site, 8.8 x the algorithmic bytes).  An entry that reads as the saturation value
// (255 / 65535) ESCAPES to the int32 count in n_kw itself; the escape loads are issued by every lane (a lane whose entry did not
// saturate re-reads n_kw[0], one line the L1 keeps), so the number of loads in flight is the same on every path and the three-deep
// software pipeline below stays exact: at the top of batch b the narrow gathers of b+1 and the escapes of b are in flight.
template <int IMG> struct img_elem { typedef uint8_t T; static constexpr int SAT = 255; };
template <> struct img_elem<16> { typedef uint16_t T; static constexpr int SAT = 65535; };

template <int GS, class PT = KParams, int IMG = 0>
__global__ void __launch_bounds__(256) llda_sweep_sparse_kernel(const PT P)
{
    constexpr int GPB = 256 / GS;
    constexpr bool WIDE = sparse_is_wide<PT>::value;
    __shared__ int s_nk[WIDE ? 1 : LLDA_NARROW_KP];      // workgroup accumulator of the n_k changes (narrow layouts)
    const int tid = threadIdx.x;
    const int KP = P.KP;
    if (!WIDE)
        for (int i = tid; i < KP; i += 256) s_nk[i] = 0;
    if (WIDE && tid == 0) sparse_wide_init();             // (the lock of the workgroup's exact-tier buffer)
    __syncthreads();
    const int lane = tid & 63;
    const int lig = tid & (GS - 1);
    const int grp = tid / GS;
    const int gbase = lane & ~(GS - 1);
    const uint64_t gmask = (GS == 64) ? ~0ull : ((1ull << GS) - 1ull);

    // The document loops are WAVE-UNIFORM (a lane group whose document is shorter, empty or past the end of the shard
    // idles through flags, it does not leave the loop): the exact tier of sparse_site() needs all 64 lanes.
    const int dpg = P.dpg;
    for (int it = 0; it < dpg; ++it) {
        // (fields needed once per document are read through a fresh kernel-argument pointer: an s_load here instead of
        // a register -- or a spill slot -- held across the site loop)
        const LLDA_CONSTANT PT *Q = kernarg_fresh<PT>();
        const int64_t idx = ((int64_t)blockIdx.x * Q->dpg + it) * GPB + grp;
        const bool valid = idx < Q->D;
        if (!__any(valid)) break;
        const int32_t *order = Q->doc_order;
        const int64_t d = valid ? (order ? (int64_t)order[idx] : idx) : 0;
        const int64_t *doc_off = Q->doc_off, *live_off = Q->live_off;
        const int64_t s0 = doc_off[d];
        const int len = valid ? (int)(doc_off[d + 1] - s0) : 0;
        const int64_t l0 = live_off[d];
        const int A = (int)(live_off[d + 1] - l0);
        const bool live = valid && len > 0 && lig < A;
        const int pos = live ? Q->live_pos[l0 + lig] : -1;
        // where the topic's counts sit in a row of the narrow image: the sampler may have reordered the image's columns so that
        // topics that are allowed together share cache lines (llda_pack_image_cols); the counts themselves keep their order
        [[maybe_unused]] int ipos = pos;
        if constexpr (IMG != 0) {
            const int32_t *ic = Q->img_col;
            if (ic && live) ipos = (int)min((uint32_t)ic[pos], (uint32_t)KP - 1u);      // (a caller's table: never outside the image row)
        }
        int32_t *ndk_p = Q->n_dk + d * KP + (live ? pos : 0);
        int ndk = live ? *ndk_p : 0;
        const int ndk0 = ndk;
        int nk = live ? Q->n_k[pos] : 0;
        const uint32_t gdoc = (uint32_t)(d + Q->doc_base);
        int max_len = len;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) max_len = max(max_len, __shfl_xor(max_len, o, 64));

        // Sites are processed in batches of 8: lane j < 8 of the group holds the scalars of site n0+j and draws its
        // uniform, every lane gathers its own topic's n_kw entry for all 8 words, then 8 sites run back to back on
        // registers / DPP only, and lane j commits site n0+j (z store + log word or two atomics).  The memory
        // operations are software-pipelined over the batches -- with the rows resident in L2 the kernel ran 1.9x faster,
        // i.e. half of its time was exposed latency (two dependent round trips per batch: scalars, then gathers): at
        // the top of batch b its gathers and the scalars of batch b+1 are in flight; the body issues the gathers of
        // batch b+1 and the scalars of batch b+2 before it touches the entries of batch b.
        const int jj = lig & 7;
        struct BatchScalars { int v, f, z, c; };
        auto load_scalars = [&](const int n0b, BatchScalars &S) {
            const int nbb = max(0, min(8, len - n0b));
            S.v = S.f = S.z = S.c = 0;
            if (nbb > 0) {
                const LLDA_CONSTANT PT *B = kernarg_fresh<PT>();     // (once per batch of 8 sites)
                const int64_t si = s0 + n0b + (jj < nbb ? jj : nbb - 1);
                const int32_t *cp = B->csc_pos;
                S.v = B->word[si]; S.f = B->freq[si]; S.z = B->z[si]; S.c = cp ? cp[si] : 0;
            }
        };
        auto gather = [&](const int n0b, const int sv, int (&xg)[8]) {
            const int nbb = max(0, min(8, len - n0b));
            // (the broadcasts must run in ALL lanes: a DPP read from a lane that is masked off returns 0)
            const int w0 = bcast_lane<GS, 0>(sv, lig), w1 = bcast_lane<GS, 1>(sv, lig), w2 = bcast_lane<GS, 2>(sv, lig),
                      w3 = bcast_lane<GS, 3>(sv, lig), w4 = bcast_lane<GS, 4>(sv, lig), w5 = bcast_lane<GS, 5>(sv, lig),
                      w6 = bcast_lane<GS, 6>(sv, lig), w7 = bcast_lane<GS, 7>(sv, lig);
#pragma unroll
            for (int j = 0; j < 8; ++j) xg[j] = 0;
            if (live && nbb > 0) {
                const int32_t *col = P.n_kw + pos;
                xg[0] = col[(int64_t)w0 * KP]; xg[1] = col[(int64_t)w1 * KP]; xg[2] = col[(int64_t)w2 * KP];
                xg[3] = col[(int64_t)w3 * KP]; xg[4] = col[(int64_t)w4 * KP]; xg[5] = col[(int64_t)w5 * KP];
                xg[6] = col[(int64_t)w6 * KP]; xg[7] = col[(int64_t)w7 * KP];
            }
        };
        if constexpr (IMG == 0) {
            BatchScalars Sc, Sn;
            load_scalars(0, Sc);
            load_scalars(8, Sn);
            int xg[8];
            gather(0, Sc.v, xg);
            for (int n0 = 0; n0 < max_len; n0 += 8) {
                const int nb = max(0, min(8, len - n0));
                int xg_next[8];
                gather(n0 + 8, Sn.v, xg_next);
                BatchScalars Sf;
                load_scalars(n0 + 16, Sf);
                const int sv = Sc.v, sf = Sc.f, sz = Sc.z, sc = Sc.c;
                int su_lo, su_hi, su_f;
                {   // keyed uniform of site n0+jj: Philox block (site >> 1), words (0,1) / (2,3) by parity
                    const int n = n0 + jj;
                    uint32_t c0 = (uint32_t)(n >> 1), c1 = gdoc, c2 = P.stream_id, c3 = P.sweep;
                    philox4x32_10(c0, c1, c2, c3, P.key0, P.key1);
                    const uint32_t ra = (n & 1) ? c2 : c0, rb = (n & 1) ? c3 : c1;
                    const double u = ((double)(ra >> 5) * 67108864.0 + (double)(rb >> 6)) * (1.0 / 9007199254740992.0);
                    su_lo = __double2loint(u); su_hi = __double2hiint(u); su_f = __float_as_int((float)u);
                }

                int my_zn = sz;
                int done = 0;                     // sites of this batch
    #define LLDA_SPARSE_SITE(J)                                                                                    \
                sparse_site<GS, J, true>(P, -1, -1, nb, sf, sz, su_lo, su_hi, su_f, xg, live, pos, A, ndk, nk, my_zn, done, lig, lane, \
                                   gbase, gmask);
                LLDA_SPARSE_SITE(0) LLDA_SPARSE_SITE(1) LLDA_SPARSE_SITE(2) LLDA_SPARSE_SITE(3)
                LLDA_SPARSE_SITE(4) LLDA_SPARSE_SITE(5) LLDA_SPARSE_SITE(6) LLDA_SPARSE_SITE(7)
    #undef LLDA_SPARSE_SITE
                // commit the sites of the batch: lane j handles site n0+j
                if (lig < 8 && lig < done) commit_site(*kernarg_fresh<PT>(), s0 + n0 + lig, sv, sf, sz, my_zn, sc, KP);
                Sc = Sn; Sn = Sf;
    #pragma unroll
                for (int j = 0; j < 8; ++j) xg[j] = xg_next[j];
            }
        } else {
            typedef typename img_elem<IMG>::T IT;
            constexpr int SAT = img_elem<IMG>::SAT;
            auto narrow = [&](const int n0b, const int sv, int (&xn)[8]) {
                const int nbb = max(0, min(8, len - n0b));
                const int w0 = bcast_lane<GS, 0>(sv, lig), w1 = bcast_lane<GS, 1>(sv, lig), w2 = bcast_lane<GS, 2>(sv, lig),
                          w3 = bcast_lane<GS, 3>(sv, lig), w4 = bcast_lane<GS, 4>(sv, lig), w5 = bcast_lane<GS, 5>(sv, lig),
                          w6 = bcast_lane<GS, 6>(sv, lig), w7 = bcast_lane<GS, 7>(sv, lig);
#pragma unroll
                for (int j = 0; j < 8; ++j) xn[j] = 0;
                if (live && nbb > 0) {
                    const LLDA_GLOBAL IT *col = (const LLDA_GLOBAL IT *)P.img + ipos;
                    xn[0] = col[(int64_t)w0 * KP]; xn[1] = col[(int64_t)w1 * KP]; xn[2] = col[(int64_t)w2 * KP];
                    xn[3] = col[(int64_t)w3 * KP]; xn[4] = col[(int64_t)w4 * KP]; xn[5] = col[(int64_t)w5 * KP];
                    xn[6] = col[(int64_t)w6 * KP]; xn[7] = col[(int64_t)w7 * KP];
                }
            };
            // (lanes without a document or past its allowed topics hold 0 in xn: never saturated, they read n_kw[0])
            auto escape = [&](const int sv, const int (&xn)[8], int (&xe)[8]) {
                const int w0 = bcast_lane<GS, 0>(sv, lig), w1 = bcast_lane<GS, 1>(sv, lig), w2 = bcast_lane<GS, 2>(sv, lig),
                          w3 = bcast_lane<GS, 3>(sv, lig), w4 = bcast_lane<GS, 4>(sv, lig), w5 = bcast_lane<GS, 5>(sv, lig),
                          w6 = bcast_lane<GS, 6>(sv, lig), w7 = bcast_lane<GS, 7>(sv, lig);
                const LLDA_GLOBAL int32_t *cnt = (const LLDA_GLOBAL int32_t *)P.n_kw;
#define LLDA_ESC(J, W) xe[J] = cnt[xn[J] == SAT ? (int64_t)(W) * KP + pos : (int64_t)0];
                LLDA_ESC(0, w0) LLDA_ESC(1, w1) LLDA_ESC(2, w2) LLDA_ESC(3, w3)
                LLDA_ESC(4, w4) LLDA_ESC(5, w5) LLDA_ESC(6, w6) LLDA_ESC(7, w7)
#undef LLDA_ESC
            };
            BatchScalars S0, S1, S2;
            load_scalars(0, S0);
            load_scalars(8, S1);
            load_scalars(16, S2);
            int xn0[8], xe0[8], xn1[8];
            narrow(0, S0.v, xn0);
            narrow(8, S1.v, xn1);
            escape(S0.v, xn0, xe0);
            for (int n0 = 0; n0 < max_len; n0 += 8) {
                const int nb = max(0, min(8, len - n0));
                int xg[8];                                     // the counts of batch n0: image entry, or its escape
#pragma unroll
                for (int j = 0; j < 8; ++j) xg[j] = (xn0[j] == SAT) ? xe0[j] : xn0[j];
                int xn2[8];
                narrow(n0 + 16, S2.v, xn2);
                BatchScalars S3;
                load_scalars(n0 + 24, S3);
                int xe1[8];
                escape(S1.v, xn1, xe1);
                const int sv = S0.v, sf = S0.f, sz = S0.z, sc = S0.c;
                int su_lo, su_hi, su_f;
                {   // keyed uniform of site n0+jj: Philox block (site >> 1), words (0,1) / (2,3) by parity
                    const int n = n0 + jj;
                    uint32_t c0 = (uint32_t)(n >> 1), c1 = gdoc, c2 = P.stream_id, c3 = P.sweep;
                    philox4x32_10(c0, c1, c2, c3, P.key0, P.key1);
                    const uint32_t ra = (n & 1) ? c2 : c0, rb = (n & 1) ? c3 : c1;
                    const double u = ((double)(ra >> 5) * 67108864.0 + (double)(rb >> 6)) * (1.0 / 9007199254740992.0);
                    su_lo = __double2loint(u); su_hi = __double2hiint(u); su_f = __float_as_int((float)u);
                }

                int my_zn = sz;
                int done = 0;                     // sites of this batch
    #define LLDA_SPARSE_SITE(J)                                                                                    \
                sparse_site<GS, J, true>(P, -1, -1, nb, sf, sz, su_lo, su_hi, su_f, xg, live, pos, A, ndk, nk, my_zn, done, lig, lane, \
                                   gbase, gmask);
                LLDA_SPARSE_SITE(0) LLDA_SPARSE_SITE(1) LLDA_SPARSE_SITE(2) LLDA_SPARSE_SITE(3)
                LLDA_SPARSE_SITE(4) LLDA_SPARSE_SITE(5) LLDA_SPARSE_SITE(6) LLDA_SPARSE_SITE(7)
    #undef LLDA_SPARSE_SITE
                // commit the sites of the batch: lane j handles site n0+j
                if (lig < 8 && lig < done) commit_site(*kernarg_fresh<PT>(), s0 + n0 + lig, sv, sf, sz, my_zn, sc, KP);
                S0 = S1; S1 = S2; S2 = S3;
#pragma unroll
                for (int j = 0; j < 8; ++j) { xn0[j] = xn1[j]; xe0[j] = xe1[j]; xn1[j] = xn2[j]; }
            }
        }
        if (live) {
            *ndk_p = ndk;
            if (ndk != ndk0) {
                if (WIDE) atomicAdd(P.n_k_delta + pos, ndk - ndk0);
                else atomicAdd(&s_nk[pos], ndk - ndk0);
            }
        }
    }
    if (!WIDE) {
        __syncthreads();
        for (int i = tid; i < KP; i += 256) {
            const int dl = s_nk[i];
            if (dl) atomicAdd(P.n_k_delta + i, dl);
        }
    }
}

}  // namespace
