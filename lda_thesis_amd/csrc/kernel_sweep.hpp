// kernel_sweep.hpp -- llda_sweep_exact_kernel (general) and llda_sweep_kernel (tiered, the hot kernel)
// Part of the single translation unit llda_gibbs.hip (included in order; see the contents list there).
#pragma once

namespace {

// ---------------------------------------------------------------------------------------------
// Sweep kernel, general form: every site through the reference's fp64 pipeline, inline.  Used when the
// tiered kernel's preconditions do not hold (alpha or beta < 1e-6, V*beta >= 2^40).
// ---------------------------------------------------------------------------------------------
template <int G, int T, bool HAS_TAIL>
__global__ void __launch_bounds__(256) llda_sweep_exact_kernel(const KParams P)
{
    constexpr int KP = G * T;
    constexpr int GPB = 256 / G;              // lane groups (documents in flight) per workgroup
    __shared__ int s_nk[KP];                  // workgroup accumulator of the n_k changes
    const int tid = threadIdx.x;
    for (int i = tid; i < KP; i += 256) s_nk[i] = 0;
    __syncthreads();
    const int lane = tid & 63;
    const int lig = tid & (G - 1);            // lane in group
    const int grp = tid / G;

    for (int it = 0; it < P.dpg; ++it) {
        const int64_t idx = ((int64_t)blockIdx.x * P.dpg + it) * GPB + grp;
        if (idx >= P.D) break;
        const int64_t d = P.doc_order ? (int64_t)P.doc_order[idx] : idx;
        const int64_t s0 = P.doc_off[d];
        const int len = (int)(P.doc_off[d + 1] - s0);
        if (len <= 0) continue;

        int ndk[T], nkb[T];           // nkb = n_k(sweep start) - n_dk(sweep start): n_k seen by the
        int32_t *ndk_row = P.n_dk + d * KP;             // document is nkb + ndk at any time
        load_lane_row<G, T>(ndk_row, lig, ndk);
        load_lane_row<G, T>(P.n_k, lig, nkb);
#pragma unroll
        for (int s = 0; s < T; ++s) nkb[s] -= ndk[s];
        const uint32_t mask = P.lab_mask[d * G + lig];
        const uint32_t gdoc = (uint32_t)(d + P.doc_base);

        // memory pipeline: see llda_sweep_kernel
        int v_c = P.word[s0], f_c = P.freq[s0], zo_c = P.z[s0], c_c = P.csc_pos ? P.csc_pos[s0] : 0;
        const int64_t i1 = s0 + (len > 1 ? 1 : 0);
        int v_1 = P.word[i1], f_1 = P.freq[i1], zo_1 = P.z[i1], c_1 = P.csc_pos ? P.csc_pos[i1] : 0;
        int xn[T];
        load_lane_row<G, T>(P.n_kw + (int64_t)v_c * KP, lig, xn);
        uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        int64_t pend_i = -1;
        int pend_v = 0, pend_f = 0, pend_zo = 0, pend_zn = 0, pend_c = 0;

        for (int n = 0; n < len; ++n) {
            const int v = v_c, f = f_c, zo = zo_c, c = c_c;
            int x[T];
#pragma unroll
            for (int s = 0; s < T; ++s) x[s] = xn[s];
            if (lig == 0 && pend_i >= 0) commit_site(P, pend_i, pend_v, pend_f, pend_zo, pend_zn, pend_c, KP);
            load_lane_row<G, T>(P.n_kw + (int64_t)v_1 * KP, lig, xn);
            v_c = v_1; f_c = f_1; zo_c = zo_1; c_c = c_1;
            {
                const int64_t i2 = s0 + (n + 2 < len ? n + 2 : len - 1);
                v_1 = P.word[i2]; f_1 = P.freq[i2]; zo_1 = P.z[i2];
                if (P.csc_pos) c_1 = P.csc_pos[i2];
            }
            const double u = site_uniform<G>(P, n, n == 0, gdoc, lig, r0, r1, r2, r3);

            // remove the site (LabeledLDA.py:109-111)
            {
                int lo, so;
                lane_slot_of<G, T>(zo, lo, so);
                onehot_add2<T>(ndk, x, (lig == lo) ? (1u << so) : 0u, f);
            }
            // scores (LabeledLDA.py:113-116), np.sum, prob /= sum, keyed draw
            double w[T];
            scores<T>(w, ndk, nkb, x, mask, P.alpha, P.beta, P.vbeta);
            const double S = group_sum<G, T, HAS_TAIL>(w, P, lig, lane);
            const double y = 1.0 / S;
#pragma unroll
            for (int s = 0; s < T; ++s) w[s] = div_by(w[s], S, y);
            int zn = draw_position<G, T, false>(w, u, mask, S > 0.0, lig, lane);
            if (zn < 0) {
                zn = zo;
                if (lig == 0 && P.status) atomicOr(P.status, 1);    // no topic with positive probability
            }
            // add the site back (LabeledLDA.py:121-125)
            {
                int ln, sn;
                lane_slot_of<G, T>(zn, ln, sn);
                onehot_add1<T>(ndk, (lig == ln) ? (1u << sn) : 0u, -f);
            }
            pend_i = s0 + n; pend_v = v; pend_f = f; pend_zo = zo; pend_zn = zn; pend_c = c;
        }
        if (lig == 0 && pend_i >= 0) commit_site(P, pend_i, pend_v, pend_f, pend_zo, pend_zn, pend_c, KP);

        int old[T];
        load_lane_row<G, T>(ndk_row, lig, old);
#pragma unroll
        for (int s = 0; s < T; ++s) {
            const int dl = ndk[s] - old[s];
            if (dl) atomicAdd(&s_nk[pos_of<G, T>(lig, s)], dl);
        }
        store_lane_row<G, T>(ndk_row, lig, ndk);
    }
    __syncthreads();
    for (int i = tid; i < KP; i += 256) {
        const int dl = s_nk[i];
        if (dl) atomicAdd(P.n_k_delta + i, dl);
    }
}

// ---------------------------------------------------------------------------------------------
// Sweep kernel, tiered form (the one that runs in practice).  Preconditions, host-checked in llda_sweep:
// alpha, beta >= 1e-6 (every label-allowed topic has a strictly positive probability, so the "p > 0" tests
// of the draw can be read off the label mask) and V*beta < 2^40.  DENSE (K == KP): every document allows
// every topic and the label mask is not applied.
// The document's n_dk row, the n_k it sees and an fp32 reciprocal of n_k + V*beta live in LDS as
// [slot][thread] arrays: conflict-free, and the owning lane updates ONE dynamically indexed slot per
// change (VGPR arrays would need a 16-deep select chain per update).
// ---------------------------------------------------------------------------------------------
#ifndef LLDA_MARGIN0
#define LLDA_MARGIN0 0x1p-17f   // tier-0 (fp32) decision margin relative to the total score: 128 * 2^-24, the
                                // worst-case error bound of DESIGN.md section 4.3 is 105 * 2^-24
#endif
#ifndef LLDA_WAVES
#define LLDA_WAVES 3          // waves per SIMD the register allocator must leave room for (16 slots per lane: LDS allows 3)
#endif
// Up to 12 slots per lane LDS (<= 40 KB per workgroup) allows one more, and 128 VGPRs suffice without spills in the site
// loop: + 10 % at 16 lanes per document (K = 192), + 17 % at 32 and 64 (K = 384, 768); the 8-lane layouts, bound by the
// vector-memory address pipeline, lose 6 % with a fourth wave and stay at 3 (tools/abl_vocab.py).  With 16 slots per lane
// a fourth wave needs both LDS packing and ~40 fewer VGPRs: a packed-LDS build at 128 VGPRs spilled six values per site,
// and every scratch reload waits for vmcnt(0), i.e. for the row prefetch -- 98 ms instead of 58 at K = 512.  The 16-bit-row kernel
// does run at four (template parameter W4 below: one row tuple, packed LDS, no spill in the site loop).

// tier-0 factor of one topic: fl32(a * y) with a = fl32(n_dk + alpha), y = v_rcp_f32(fl32(n_k + V*beta)).
// n_dk and n_k of a topic always change together, so the product is cached as ONE float per slot.
__device__ __forceinline__ float tier0_factor(int ndk, int nk, float alpha32, float vbeta32)
{
    return ((float)ndk + alpha32) * __builtin_amdgcn_rcpf((float)nk + vbeta32);
}

// a topic count of this document changes by df: n_dk, the n_k the document sees, and the cached tier-0 factor
__device__ __forceinline__ void count_update(int (*s_ndk)[256], int (*s_nkc)[256], float (*s_pa)[256], int slot,
                                             int tid, float alpha32, float vbeta32, int df)
{
    const int nd = s_ndk[slot][tid] + df, nk = s_nkc[slot][tid] + df;
    s_ndk[slot][tid] = nd;
    s_nkc[slot][tid] = nk;
    s_pa[slot][tid] = tier0_factor(nd, nk, alpha32, vbeta32);
}

// W4 form (four waves per SIMD): n_dk and its sweep-start value share one LDS word (n_dk | start << 16; both below 2^16, checked
// when the document is staged), the n_k the document sees is the workgroup's copy of the sweep-start n_k plus the difference
__device__ __forceinline__ void count_update_w4(int (*s_ndk)[256], const int *s_nk0, float (*s_pa)[256], int slot, int pos,
                                                int tid, float alpha32, float vbeta32, int df)
{
    const int w = s_ndk[slot][tid] + df;                              // (0 <= n_dk + df < 2^16: no carry into the upper half)
    s_ndk[slot][tid] = w;
    const int nd = w & 0xffff, nk = s_nk0[pos] + nd - (int)((uint32_t)w >> 16);
    s_pa[slot][tid] = tier0_factor(nd, nk, alpha32, vbeta32);
}

// LOGGED: the commit goes to the word-major commit log (csc_pos / commit_log non-NULL) -- a compile-time fact, so the
// instantiation carries neither the atomics path nor the pointer tests (the kernel is VALU-issue bound and short of
// SGPRs: every uniform test in the site loop costs).
// REC: the scalars of a site come from llda_sweep_args.site_rec -- a compile-time fact too: as a run-time pointer test the two
// load paths met in a phi, the record's fields had to be COPIED into the registers of the other path, the copy needs the
// value, and the compiler put s_waitcnt vmcnt(0) right behind the load -- i.e. behind the row prefetch issued just before.
// R16: rows of n_kw whose counts fit 16 bits are read from their 16-bit image P.n_kw16 (llda_pack_rows16) -- half the
// bytes through the fabric for exactly the rows that miss the L2 (the rare words).  Which sites do so is bit 31 of their
// csc_pos and WHERE their row starts is llda_sweep_args.site_row (read instead of the word id), so the choice costs the
// prefetch nothing.  A 16-bit row arrives as two 16-byte chunks per lane (slots 0..7, 8..15, two per register), the
// conversion to fp32 reads the halves directly (SDWA), and the site's own count leaves the fp32 values through the slot
// index.  What a 16-bit site costs over a 32-bit one: the second conversion arm when the two documents of a wavefront
// differ (47 % of the sites of configs[3]) and the branch around the third and fourth chunk load.  This is the kernel the
// K = 512 / 1024 lines of the bench run on (the int32 form is bound by the fabric, this one by issue).
// W4: the 16-bit-row kernel at FOUR waves per SIMD (it is bound by instruction issue with a SIMD idle in a fifth of its cycles at
// three: DESIGN.md section 4.1) -- 36 / 40 KB of LDS per workgroup (count_update_w4) and at most 128 VGPRs: ONE row tuple (the row
// dies with its conversion; the 0.6 % of the sites that go on to the cold tiers fetch theirs again).  Documents of at most 65 535
// tokens (llda_sweep_args.max_doc_tokens).
template <int G, int T, bool HAS_TAIL, bool DENSE, bool LOGGED, bool REC = false, bool R16 = false, bool W4 = false>
__global__ void __launch_bounds__(256, W4 ? 4 : (T <= 12 && G >= 16) ? LLDA_WAVES + 1 : LLDA_WAVES) llda_sweep_kernel(const KParams P)
{
    static_assert(!R16 || (G >= 32 && T == 16 && DENSE && LOGGED && !REC), "16-bit rows: the dense 16-slot kernel with the commit log");
    static_assert(!W4 || R16, "four waves: the 16-bit-row kernel");
    constexpr int KP = G * T;
    constexpr int GPB = 256 / G;              // lane groups (documents in flight) per workgroup
    __shared__ int s_nk[KP];                  // workgroup accumulator of the n_k changes
    __shared__ int s_ndk[T][256];             // n_dk row of the document (W4: | sweep-start n_dk << 16)
    __shared__ int s_nkc[W4 ? 1 : T][256];    // n_k as the document sees it (W4: derived, see s_nk0)
    __shared__ int s_nk0[W4 ? KP : 1];        // W4: the sweep-start n_k, one copy per workgroup
    __shared__ float s_pa[T][256];            // tier-0 factor fl32((n_dk + alpha) / (n_k + V*beta))

    const int tid = threadIdx.x;
    for (int i = tid; i < KP; i += 256) s_nk[i] = 0;
    if constexpr (W4) {
        for (int i = tid; i < KP; i += 256) s_nk0[i] = P.n_k[i];
    }
    __syncthreads();

    const int lane = tid & 63;
    const int lig = tid & (G - 1);            // lane in group
    const int grp = tid / G;
    const float vbeta32 = (float)P.vbeta, alpha32 = (float)P.alpha, beta32 = (float)P.beta;

    const int64_t site_base = P.doc_off[0];                 // uniform: the bases below stay in SGPRs
    // (16-bit rows: the row starts of llda_sweep_args.site_row stand in for the word ids)
    const int32_t *word_b = (R16 ? P.site_row : P.word) + site_base, *freq_b = P.freq + site_base;
    const int32_t *csc_b = LOGGED ? P.csc_pos + site_base : nullptr;
    constexpr bool PACKED = REC;
    static_assert(!REC || (LOGGED && G <= 16), "site records: commit log, <= 16 lanes per document");
    const int32_t *rec_b = PACKED ? P.site_rec + site_base * 4 : nullptr;
    int32_t *z_b = P.z + site_base;
    for (int it = 0; it < P.dpg; ++it) {
        constexpr int n0 = 0;                 // first site to sample
        const int64_t idx = ((int64_t)blockIdx.x * P.dpg + it) * GPB + grp;
        if (idx >= P.D) break;
        const int64_t d = P.doc_order ? (int64_t)P.doc_order[idx] : idx;
        const int64_t s0 = P.doc_off[d];
        const int len = (int)(P.doc_off[d + 1] - s0);
        if (len <= n0) continue;

        int32_t *ndk_row = P.n_dk + d * KP;
        {
            int r[T], k[T];
            load_lane_row<G, T>(ndk_row, lig, r);
            load_lane_row<G, T>(P.n_k, lig, k);
            [[maybe_unused]] int big = 0;
#pragma unroll
            for (int s = 0; s < T; ++s) {
                if constexpr (W4) {
                    s_ndk[s][tid] = r[s] | (r[s] << 16);
                    big |= r[s];
                } else {
                    s_ndk[s][tid] = r[s];
                    s_nkc[s][tid] = k[s];                          // sweep-start n_k
                }
                s_pa[s][tid] = tier0_factor(r[s], k[s], alpha32, vbeta32);
            }
            if constexpr (W4) {
                // the packed word holds the counts of a document of at most 65 535 tokens: a start value beyond 16 bits, or a
                // document whose counts add up to more (one topic could collect them all before the sweep is over), means the
                // caller's max_doc_tokens was not a bound -- status bit 2, once per document, outside the site loop
                int tokens = 0;
#pragma unroll
                for (int s = 0; s < T; ++s) tokens += (int)((uint32_t)r[s] & 0xffffu);
#pragma unroll
                for (int m = 1; m < G; m <<= 1) tokens += __shfl_xor(tokens, m, G);
                if ((((uint32_t)big >> 16) || tokens > 65535) && P.status) atomicOr(P.status, 4);
            }
        }
        // (DENSE: every slot of every lane is an allowed topic -- a constant, not a register)
        const uint32_t mask = DENSE ? (1u << T) - 1u : P.lab_mask[d * G + lig];
        const uint32_t gdoc = (uint32_t)(d + P.doc_base);
        // lanes with an allowed topic: the wavefront's ballot (uniform) for 32- and 64-lane groups, the group's own
        // bits below that (draw_fast_f32)
        const uint64_t gp_doc = DENSE ? (G >= 32 ? ~0ull : (1ull << (G & 63)) - 1ull) : G >= 32 ? __ballot(mask != 0)
                                        : (__ballot(mask != 0) >> (lane & ~(G - 1))) & ((1ull << (G & 63)) - 1ull);

        // Software pipeline of the memory operations: at the top of iteration n the registers hold the
        // scalars (word, freq, z) of site n, the row of site n is in flight (xn) and so are the scalars of
        // site n+1.  Right after the single s_waitcnt vmcnt(0) of the iteration (first use of xn) the body
        // issues, in this order: the z store + two n_kw_delta atomics of site n-1, the row of site n+1,
        // the scalars of site n+2 -- so nothing the next wait covers is younger than one full site.
        // (site-indexed arrays are addressed as base + 32-bit byte offset from the first document of the call;
        // llda_sweep refuses calls that span 2^30 sites or more)
        const uint32_t sb = (uint32_t)(s0 - site_base) * 4u;
        // The scalars of three consecutive sites live in three register sets whose roles (previous / current /
        // next site) rotate with the site index, and the site loop is unrolled by three: no register-to-register
        // moves for the pipeline (a rolled loop spends 17 v_mov per site on them; the compiler cannot unroll it
        // itself because the body contains convergent cross-lane operations).
        struct SiteRegs { int v, f, zo, c, zn, lo, so; };    // (lo, so) = lane and slot of zo, decoded once per site
        // scalars of one site.  With 8 or 16 lanes per document a wavefront walks 8 / 4 documents, and every scalar
        // load touches that many cache lines: the kernel is then bound by the vector-memory address pipeline (TA
        // busy 74 % at K = 128), not by VALU issue.  PACKED: {word, freq, csc_pos} come as ONE 16-byte record per
        // site (llda_sweep_args.site_rec) -- one stream and one instruction instead of three.
        auto load_scalars = [&](SiteRegs &R, const uint32_t o) {
            if constexpr (PACKED) {
                typedef int v4i __attribute__((ext_vector_type(4)));
                const v4i r = *(const LLDA_GLOBAL v4i *)((const LLDA_GLOBAL char *)rec_b + (o << 2));
                R.v = r.x; R.f = r.y; R.c = r.z;
            } else {
                R.v = gload_i32(word_b, o); R.f = gload_i32(freq_b, o);
                R.c = LOGGED ? gload_i32(csc_b, o) : 0;
            }
            R.zo = gload_i32(z_b, o);
        };
        const uint32_t o0 = opaque_u32(sb + (uint32_t)n0 * 4u), o1 = opaque_u32(sb + (uint32_t)(n0 + 1 < len ? n0 + 1 : n0) * 4u);
        SiteRegs R0, R1, R2;
        R0.c = R1.c = 0;
        load_scalars(R0, o0); R0.zn = 0; R0.lo = R0.so = 0;
        load_scalars(R1, o1); R1.zn = 0; R1.lo = R1.so = 0;
        R2.v = R2.f = R2.zo = R2.c = R2.zn = R2.lo = R2.so = 0;
        // INDEXED (one or two documents per wavefront, 16 slots): the row lives in one of two 16-register tuples whose
        // roles (row of this site / row of the next site, in flight) alternate, and the site's own count is removed
        // from it in place through a register index held in M0 -- 6 vector instructions per document instead of 34
        // for the 16 compare-free selects of onehot_add_to.  The site loop is then unrolled by six (two tuples x
        // three scalar sets).
        constexpr bool INDEXED = G >= 32 && T == 16;
        constexpr int XM = INDEXED && !W4 ? T : 1;      // the second row tuple (W4: none)
        int xn[T], xm[XM];
        // the row of word v into xl; c = the site's csc_pos (bit 31: 16-bit row, slots 8j .. 8j+7 of all lanes contiguous)
        auto load_word_row = [&](int (&xl)[T], const int v, const int c) {
            if constexpr (R16) {
                typedef int v4i __attribute__((ext_vector_type(4)));
                const bool h = c < 0;
                const LLDA_GLOBAL char *q = (const LLDA_GLOBAL char *)P.n_kw + ((int64_t)v << 4) + lig * 16;   // v: site_row
                const v4i a0 = *(const LLDA_GLOBAL v4i *)q, a1 = *(const LLDA_GLOBAL v4i *)(q + G * 16);
                xl[0] = a0.x; xl[1] = a0.y; xl[2] = a0.z; xl[3] = a0.w;
                xl[4] = a1.x; xl[5] = a1.y; xl[6] = a1.z; xl[7] = a1.w;
                // (the upper half of the tuple is DEFINED here, by no instruction: were it merely left alone by 16-bit sites, the
                // values of the previous load would have to survive and the in-place update below would work on a copy)
                v4i a2, a3;
                asm volatile("" : "=v"(a2), "=v"(a3));
                if (!h) {
                    a2 = *(const LLDA_GLOBAL v4i *)(q + 2 * G * 16);
                    a3 = *(const LLDA_GLOBAL v4i *)(q + 3 * G * 16);
                }
                xl[8] = a2.x; xl[9] = a2.y; xl[10] = a2.z; xl[11] = a2.w;
                xl[12] = a3.x; xl[13] = a3.y; xl[14] = a3.z; xl[15] = a3.w;
            } else {
                gload_lane_row<G, T>(P.n_kw, (int64_t)v * KP, lig, xl);
            }
        };
        load_word_row(xn, R0.v, R0.c);
        uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        {   // site 0 leaves its topic (LabeledLDA.py:109-111); later sites do so at the end of the previous site
            lane_slot_of<G, T>(R0.zo, R0.lo, R0.so);
            if (lig == R0.lo) {
                if constexpr (W4) count_update_w4(s_ndk, s_nk0, s_pa, R0.so, R0.zo, tid, alpha32, vbeta32, -R0.f);
                else count_update(s_ndk, s_nkc, s_pa, R0.so, tid, alpha32, vbeta32, -R0.f);
            }
        }

        const uint64_t lig0_w = __ballot(lig == 0);               // (uniform) first lanes of the wavefront's groups
        // one site: `cur` holds its scalars, `nxt` those of site n+1, `prv` those of site n-1 (committed here, then
        // reloaded with the scalars of site n+2)
        auto site = [&](const int n, SiteRegs &cur, SiteRegs &nxt, SiteRegs &prv, int (&xc)[T], int (&xnx)[XM]) {
            const int f = cur.f, zo = cur.zo;
            LLDA_MARK("site_top");
            LLDA_MARK("lds_factors");
            // the cached tier-0 factors of the lane, fetched from LDS first thing: nothing below depends on them until
            // the scores, and the kernel is bound by the latency of one wavefront's instruction stream (39 % of the
            // wave cycles sat in s_waitcnt, mostly on LDS: eight waits per site when the reads trickle in pairs)
            float pa[T];
#pragma unroll
            for (int s = 0; s < T; ++s) pa[s] = s_pa[s][tid];
            // ... and so are the site's random bits (a cross-lane pick through LDS for groups wider than 16 lanes)
            uint32_t ra, rb;
            LLDA_MARK("rng");
            site_random_bits<G>(P, n, n == n0, gdoc, lig, r0, r1, r2, r3, ra, rb);
            __builtin_amdgcn_sched_barrier(0);
            LLDA_MARK("convert");
            // the fetched row minus the site's own count (n_dk / n_k were updated already)
            int x[T];
            [[maybe_unused]] float xf[R16 ? T : 1];                   // R16: the counts as fp32 (exact: 16-bit rows hold < 2^16)
            if constexpr (R16) {
                // the row as it came (the cold tiers take the site's own count out of THEIR copy), converted, and the own count
                // removed from the fp32 values through the slot index itself -- no register / half-word arithmetic for the packed
                // rows.  fl32(x) - f IS fl32(x - f) while x < 2^24 (every 16-bit row, and every int32 row of a real corpus); an
                // int32 count beyond that carries two more roundings (num within 5 v instead of 3 v, the compared difference
                // within 107 v instead of 105 v of section 4.3 -- the margin is 128 v)
#pragma unroll
                for (int s = 0; s < T; ++s) x[s] = xc[s];
                const bool h = cur.c < 0;
                // (two likely blocks, not if / else: the compiler moves BOTH arms of a divergent if / else out of line, three taken
                // branches per site)
                if (__builtin_expect_with_probability(h, 1, 0.6)) {
#pragma unroll
                    for (int s = 0; s < T; ++s) xf[s] = (float)((s & 1) ? (uint32_t)x[s >> 1] >> 16 : (uint32_t)x[s >> 1] & 0xffffu);
                }
                if (__builtin_expect_with_probability(!h, 1, 0.6)) {
#pragma unroll
                    for (int s = 0; s < T; ++s) xf[s] = (float)x[s];
                }
                typedef float v16f __attribute__((ext_vector_type(16)));
                v16f xv;
                LLDA_MARK("own_removal");
#pragma unroll
                for (int s = 0; s < T; ++s) xv[s] = xf[s];
                const float own = (lig == cur.lo) ? (float)f : 0.0f;
#pragma unroll
                for (int g = 0; g < 64 / G; ++g) {
                    const int so_g = __builtin_amdgcn_readlane(cur.so, g * G);
                    xv[so_g & (T - 1)] -= (G == 64 || (lane / G) == g) ? own : 0.0f;
                }
#pragma unroll
                for (int s = 0; s < T; ++s) xf[s] = xv[s];
            } else if constexpr (INDEXED) {
                typedef int v16i __attribute__((ext_vector_type(16)));
                v16i xv;
#pragma unroll
                for (int s = 0; s < T; ++s) xv[s] = xc[s];
                // (lane, slot) of zo were decoded per lane when the previous site took this one out of its topic; the slot is
                // the same in every lane of a group: one v_readlane per document, no scalar decode
                const int own = (lig == cur.lo) ? f : 0;
#pragma unroll
                for (int g = 0; g < 64 / G; ++g) {
                    const int so_g = __builtin_amdgcn_readlane(cur.so, g * G);
                    xv[so_g & (T - 1)] -= (G == 64 || (lane / G) == g) ? own : 0;
                }
#pragma unroll
                for (int s = 0; s < T; ++s) x[s] = xv[s];
            } else {
                // written to a second array so that the next row can be loaded into xn right away
                onehot_add_to<T>(x, xc, (lig == cur.lo) ? (1u << cur.so) : 0u, f);   // m = -1 at the slot: += (-1) * f
            }
            int (&xl)[T] = *(int (*)[T])(INDEXED && !W4 ? (void *)&xnx : (void *)&xc);     // where the next row goes (W4: one tuple)
            LLDA_MARK("commit");
#ifndef ABL_NOCOMMIT
            if (lig == 0 && n > n0)
                commit_site_off<LOGGED>(P, z_b, opaque_u32(sb + (uint32_t)(n - 1) * 4u), prv.v, prv.f, prv.zo, prv.zn,
                                        R16 ? prv.c & 0x7fffffff : prv.c, KP);
#endif
            LLDA_MARK("row_prefetch");
#ifndef ABL_NOLOAD
            load_word_row(xl, nxt.v, nxt.c);                              // row of site n+1 (clamped)
#else
#pragma unroll
            for (int s = 0; s < T; ++s) xl[s] = (nxt.v + s) & 7;          // ablation: no n_kw traffic
#endif
            {
                LLDA_MARK("scalars");
                const uint32_t o2 = opaque_u32(sb + (uint32_t)(n + 2 < len ? n + 2 : len - 1) * 4u);   // scalars of site n+2 (clamped)
                load_scalars(prv, o2);
            }
            // tiered draw
            int zn = -1;
            // the cold tiers for the lanes that enter; the "no topic with positive probability" outcome can only come from there
            // (tier 0 always names a position), so its test lives behind the same rare branch
            auto cold = [&]() {
                int x_c[T];
                if constexpr (W4) load_word_row(x_c, cur.v, cur.c);     // (the row's registers went to the next site's prefetch)
                else {
#pragma unroll
                    for (int s = 0; s < T; ++s) x_c[s] = x[s];
                }
                if constexpr (R16) {
                    if (cur.c < 0) {
#pragma unroll
                        for (int s = T - 1; s >= 0; --s) x_c[s] = (int)((s & 1) ? (uint32_t)x_c[s >> 1] >> 16 : (uint32_t)x_c[s >> 1] & 0xffffu);
                    }
                    if (lig == cur.lo) x_c[cur.so] -= f;              // (x is the row as it came: see above)
                }
                // (the callee reads the parameters from the kernel-argument segment: taking &P would force a scratch
                // copy of all of P and put its pointers into VGPRs)
                zn = cold_tiers<G, T, HAS_TAIL, DENSE, W4>(s_ndk, x_c, W4 ? (const int (*)[256])s_nk0 : s_nkc, tid, mask, uniform53(ra, rb),
                                                           lig, lane, (const KParams *)__builtin_amdgcn_kernarg_segment_ptr());
                if (__builtin_expect(zn < 0, 0)) {
                    zn = zo;
                    if (lig == 0 && P.status) atomicOr(P.status, 1);    // no topic with positive probability
                }
            };
            // fp32 image of the uniform: the top 27 bits (within 2^-24 relative + 2^-27 absolute of u)
            LLDA_MARK("rng");
            const float u32 = (float)(ra >> 5) * 0x1p-27f;
            if constexpr (DENSE && T == 16 && G >= 32) {
                // tier 0: fp32 (a margin >= 1 switches it off by making every site unsure: no test here).  unsure = the
                // wavefront's lanes tier 0 is not sure about (uniform)
                float qf[T];
                LLDA_MARK("scores");
                if constexpr (R16) prefix_scores_f32<T, DENSE>(qf, xf, pa, mask, beta32);
                else prefix_scores_f32<T, DENSE>(qf, x, pa, mask, beta32);
                const uint64_t unsure = draw_fast_dense_f32<G>(qf, u32, P.margin0_rel, lig, lane, zn);
                LLDA_MARK("cold_check");
                if (__builtin_expect(unsure != 0, 0)) {                                  // one scalar branch per site
                    LLDA_MARK("rare_cold");
                    if (__builtin_amdgcn_inverse_ballot_w64(spread_any<G>(unsure))) cold();   // the whole document group enters
                }
            } else {
                bool decided = false;
                if (P.margin0_rel < 1.0f) {           // tier 0: fp32
                    float qf[T];
                    if constexpr (R16) prefix_scores_f32<T, DENSE>(qf, xf, pa, mask, beta32);
                    else prefix_scores_f32<T, DENSE>(qf, x, pa, mask, beta32);
                    decided = draw_fast_f32<G, T>(qf, u32, mask, gp_doc, P.margin0_rel, lig, lane, zn);
                }
                if (__builtin_expect(!decided, 0)) cold();
            }
            cur.zn = zn;
            LLDA_MARK("decode");

            // add the site back (LabeledLDA.py:121-125), and take the NEXT site out of its topic already (its
            // scalars are in registers): the LDS state is final long before the next site's scores read it.
            // Both updates usually belong to different lanes and are done in ONE masked pass; a second pass runs
            // only for groups where the same lane owns both.
            bool more;
            {
                int ln, sn;
                lane_slot_of<G, T>(zn, ln, sn);
                lane_slot_of<G, T>(nxt.zo, nxt.lo, nxt.so);                  // (kept for the next site's removal from x)
                const int lo2 = nxt.lo, so2 = nxt.so;
                LLDA_MARK("count_update");
                if constexpr (INDEXED) {
                    // The three compares by hand, their lane masks as scalars: the compiler keeps such a mask as a per-lane bool
                    // and, where a ballot of a COMBINATION is wanted, rebuilds it through v_cndmask + v_cmp_ne.  more = the
                    // document has another site (also the site loop's exit test: returned).  (16-slot kernels with one or two
                    // documents per wavefront only: the 16-lane K = 256 kernel ran 6 % SLOWER with this shorter sequence --
                    // A/B on one box, same instruction counts elsewhere -- and keeps the compiler's.)
                    uint64_t m_new, m_old_lane, m_more;
                    asm("v_cmp_eq_u32_e64 %0, %1, %2" : "=s"(m_new) : "v"(lig), "v"(ln));
                    asm("v_cmp_eq_u32_e64 %0, %1, %2" : "=s"(m_old_lane) : "v"(lig), "v"(lo2));
                    asm("v_cmp_lt_i32_e64 %0, %1, %2" : "=s"(m_more) : "v"(n + 1), "v"(len));
                    const uint64_t m_old = m_old_lane & m_more, both_w = m_new & m_old;
                    more = __builtin_amdgcn_inverse_ballot_w64(m_more);
                    const bool own_new = __builtin_amdgcn_inverse_ballot_w64(m_new);
                    if (__builtin_expect(__builtin_amdgcn_inverse_ballot_w64(m_new | m_old), 1)) { // (some lane of the wavefront always is)
                        if constexpr (W4) count_update_w4(s_ndk, s_nk0, s_pa, own_new ? sn : so2, own_new ? zn : nxt.zo, tid, alpha32,
                                                          vbeta32, own_new ? f : -nxt.f);
                        else count_update(s_ndk, s_nkc, s_pa, own_new ? sn : so2, tid, alpha32, vbeta32, own_new ? f : -nxt.f);
                    }
                    // (rare blocks behind ONE scalar branch on the ballot: entering and leaving a divergent region costs four
                    // scalar instructions whether or not a lane takes it, and the scalar unit's cycles are not hidden here)
                    if (__builtin_expect(both_w != 0, 0)) {
                        LLDA_MARK("rare_second_update");
                        if (__builtin_amdgcn_inverse_ballot_w64(both_w)) {
                            if constexpr (W4) count_update_w4(s_ndk, s_nk0, s_pa, so2, nxt.zo, tid, alpha32, vbeta32, -nxt.f);
                            else count_update(s_ndk, s_nkc, s_pa, so2, tid, alpha32, vbeta32, -nxt.f);
                        }
                    }
                } else {
                    more = n + 1 < len;
                    const bool own_new = lig == ln, own_old = more && lig == lo2;
                    const uint64_t both_w = __ballot(own_new) & __ballot(more) & __ballot(lig == lo2);   // (taken here: the compares' own masks)
                    if (__builtin_expect(own_new || own_old, 1))                      // (some lane of the wavefront always is)
                        count_update(s_ndk, s_nkc, s_pa, own_new ? sn : so2, tid, alpha32, vbeta32, own_new ? f : -nxt.f);
                    if (__builtin_expect(both_w != 0, 0)) {
                        if (own_new && own_old) count_update(s_ndk, s_nkc, s_pa, so2, tid, alpha32, vbeta32, -nxt.f);
                    }
                }
            }
            LLDA_MARK("loop");
#ifndef ABL_NOCOMMIT
            // the last site of the document is committed right away
            if (__builtin_expect((__ballot(n + 1 == len) & lig0_w) != 0, 0))
              if (lig == 0 && n + 1 == len)
                commit_site_off<LOGGED>(P, z_b, opaque_u32(sb + (uint32_t)n * 4u), cur.v, cur.f, cur.zo, cur.zn,
                                        R16 ? cur.c & 0x7fffffff : cur.c, KP);
#endif
            return more;
        };
        // (a short document LEAVES the loop after its last site: were the remaining sites merely skipped, the
        // compiler would have to keep the unmodified row of the skipped sites alive for the next trip round the loop,
        // and the in-place update would need a copy of the tuple)
        if constexpr (W4) {
            for (int n = n0;; n += 3) {                         // one row tuple: unrolled by the three scalar sets only
                if (!site(n, R0, R1, R2, xn, xm)) break;
                if (!site(n + 1, R1, R2, R0, xn, xm)) break;
                if (!site(n + 2, R2, R0, R1, xn, xm)) break;
            }
        } else if constexpr (INDEXED) {
            for (int n = n0;; n += 6) {                         // len > n0 here
                if (!site(n, R0, R1, R2, xn, xm)) break;                 // (site returns "the document has another site")
                if (!site(n + 1, R1, R2, R0, xm, xn)) break;
                if (!site(n + 2, R2, R0, R1, xn, xm)) break;
                if (!site(n + 3, R0, R1, R2, xm, xn)) break;
                if (!site(n + 4, R1, R2, R0, xn, xm)) break;
                if (!site(n + 5, R2, R0, R1, xm, xn)) break;
            }
        } else {
            for (int n = n0;; n += 3) {
                site(n, R0, R1, R2, xn, xm);
                if (n + 1 >= len) break;
                site(n + 1, R1, R2, R0, xn, xm);
                if (n + 2 >= len) break;
                site(n + 2, R2, R0, R1, xn, xm);
                if (n + 3 >= len) break;
            }
        }

        // document done: fold its n_dk change into the workgroup's n_k accumulator, store the row
        int old[T], cur[T];
        load_lane_row<G, T>(ndk_row, lig, old);
#pragma unroll
        for (int s = 0; s < T; ++s) {
            cur[s] = W4 ? s_ndk[s][tid] & 0xffff : s_ndk[s][tid];
            const int dl = cur[s] - old[s];
            if (dl) atomicAdd(&s_nk[pos_of<G, T>(lig, s)], dl);
        }
        store_lane_row<G, T>(ndk_row, lig, cur);
    }

    __syncthreads();
    for (int i = tid; i < KP; i += 256) {
        const int dl = s_nk[i];
        if (dl) atomicAdd(P.n_k_delta + i, dl);
    }
}

}  // namespace
