// device_common.hpp -- kernel parameters, Philox, row loads, one-hot updates, exact division, cross-lane moves, numpy-ordered group sum, exact keyed draw
// Part of the single translation unit llda_gibbs.hip (included in order; see the contents list there).
#pragma once

namespace {

thread_local int g_last_hip_error = 0;

// -DLLDA_BUDGET_MARKS (tools/site_loop_budget.py; llda_build_info reports it): comment lines in the device assembly at the boundaries
// of the functional classes of a site, so that every instruction of the site loop can be attributed.  The marks are scheduling barriers:
// the marked build is for COUNTING, not for timing.
#ifdef LLDA_BUDGET_MARKS
#define LLDA_MARK(name) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; @" name); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define LLDA_MARK(name)
#endif

#define LLDA_MAX_LIVE 64   // most allowed topics per document the sparse kernel handles
#define LLDA_NARROW_KP 1024  // longest row of a narrow layout (<= 8 leaves: 64 lanes x 16 slots)

struct KParams {
    const int64_t *doc_off;
    const int32_t *doc_order;
    const int32_t *word;
    const int32_t *freq;
    int32_t *z;
    const uint16_t *lab_mask;
    int32_t *n_dk;
    const int32_t *n_kw;
    int32_t *n_kw_delta;
    const int32_t *n_k;
    int32_t *n_k_delta;
    int32_t *status;
    int64_t D;
    int64_t doc_base;
    double alpha, beta, vbeta;
    uint32_t key0, key1, sweep, stream_id;
    int32_t dpg;
    int32_t last_leaf;      // index of the last (tail-carrying) leaf
    int32_t tail, tail_row;
    int32_t n_rounds;
    int32_t xor_tree;       // leaves combine as p^1, p^2, p^4 (balanced recursion, no padded leaves)
    double margin_rel;      // tier-1 decision margin relative to the total score (2^-40; debug: wider / inf)
    float margin0_rel;      // tier-0 (fp32) margin (2^-17; >= 1 disables tier 0)
    float margin0_data;     // quad kernel: weight of its data-dependent margin (production: 1 with margin0_rel = 0; test hooks: 0)
    float alpha32, beta32, vbeta32;   // the priors rounded to fp32 (tier 0 of the sparse-label kernels)
    // sparse-label path: per document the device positions of its allowed topics, in draw order ((lane, slot) ascending)
    const int64_t *live_off;
    const int32_t *live_pos;
    int32_t KP;             // row length (the sparse kernel is not templated on the layout)
    // commit log (both NULL: n_kw_delta atomics): one word per site at its word-major position
    const int32_t *csc_pos;
    uint32_t *commit_log;
    const int32_t *site_rec;   // optional [S][4]: {word, freq, csc_pos, 0} per site (kernels with <= 16 lanes per document)
    const uint16_t *n_kw16;    // optional 16-bit image of n_kw (llda_pack_rows16); sites with bit 31 of csc_pos set read it
    const int32_t *site_row;   // with n_kw16: start of the row of each site's word, 16-byte units from n_kw (read instead of word)
    const void *img;           // sparse-label kernels: optional narrow image of n_kw (llda_pack_image: one byte / 16-bit word per count,
                               // saturating); an entry that reads 255 / 65535 is re-read from n_kw
    const int32_t *img_col;    // with img: image column of every device position (llda_pack_image_cols); NULL = the position itself
    int w4;                    // with n_kw16: documents hold < 2^16 tokens -- the four-wave form of the kernel may run
    const uint8_t *row16;      // quad kernel (kernel_quad.hpp): per word, 1 = every count of its row fits the 16-bit image this sweep
    int quad_pad;              // quad kernel: K < KP (positions without a topic: their factors are 0, the cold tiers take the masked form)
    uint32_t rounds_pk[LLDA_MAX_ROUNDS];   // 4 bits per leaf: partner leaf
};

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11).  Counter (c0..c3), key (k0,k1).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3,
                                              uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// Global-memory accesses of the hot loop: base pointer (uniform, from the kernel arguments) + 32-bit BYTE
// offset.  The explicit global address space gives global_load / global_store with the base in SGPRs and the
// offset in one VGPR -- no 64-bit address arithmetic per access, and no lgkmcnt coupling as with flat_*.
#define LLDA_GLOBAL __attribute__((address_space(1)))
// (the empty asm hides how the offset was computed: otherwise loop strength reduction turns base + offset into
// one 64-bit pointer induction variable per array, i.e. back into 64-bit VALU adds)
__device__ __forceinline__ uint32_t opaque_u32(uint32_t x)
{
    asm("" : "+v"(x));
    return x;
}
// The kernel-argument segment as a FRESH pointer (constant address space: loads through it are s_load from the scalar
// cache).  The empty asm makes the pointer opaque, so a field read through it is re-loaded where the read is written
// instead of being hoisted out of every loop and kept -- or spilled to VGPR lanes -- for the whole kernel.  For fields
// that are needed once per document or once per batch of sites this trades a v_readlane reload for an s_load.
#define LLDA_CONSTANT __attribute__((address_space(4)))
template <class PT>
__device__ __forceinline__ const LLDA_CONSTANT PT *kernarg_fresh()
{
    auto p = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return (const LLDA_CONSTANT PT *)p;
}
__device__ __forceinline__ int gload_i32(const int32_t *base, uint32_t byte_off)
{
    return *(const LLDA_GLOBAL int32_t *)((const LLDA_GLOBAL char *)base + byte_off);
}
__device__ __forceinline__ void gstore_i32(int32_t *base, uint32_t byte_off, int v)
{
    *(LLDA_GLOBAL int32_t *)((LLDA_GLOBAL char *)base + byte_off) = v;
}
// T contiguous int32 at element offset `elem` of a global array
template <int T>
__device__ __forceinline__ void gload_row(const int32_t *base, int64_t elem, int (&x)[T])
{
    const LLDA_GLOBAL int32_t *q = (const LLDA_GLOBAL int32_t *)base + elem;
    if constexpr (T % 4 == 0) {
        typedef int v4i __attribute__((ext_vector_type(4)));
        const LLDA_GLOBAL v4i *p = (const LLDA_GLOBAL v4i *)q;
#pragma unroll
        for (int i = 0; i < T / 4; ++i) {
            const v4i v = p[i];
            x[4 * i + 0] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < T; ++i) x[i] = q[i];
    }
}

// ---------------------------------------------------------------------------------------------
// Row layout in memory.  A row of KP = G*T entries belongs to G lanes x T slots.  The 16-byte
// chunk i (slots 4i .. 4i+3) of ALL lanes is contiguous:  pos(g, s) = ((s >> 2) * G + g) * 4 + (s & 3)  -- so the
// i-th global_load_dwordx4 of the wavefront reads one contiguous run of the row (G * 16 bytes per lane group)
// instead of 16 bytes out of every lane's 64 (which made the vector cache look up every line of the row once per
// instruction: 4x the tag traffic).  Rows with T < 4 (one 4- or 8-byte access per lane) keep pos = g*T + s.
// ---------------------------------------------------------------------------------------------
template <int G, int T>
__device__ __forceinline__ int pos_of(int g, int s)
{
    if constexpr (T % 4 == 0) return ((((s >> 2) * G) + g) << 2) | (s & 3);
    else return g * T + s;
}
template <int G, int T>
__device__ __forceinline__ void lane_slot_of(int pos, int &g, int &s)
{
    if constexpr (T % 4 == 0) {
        const unsigned q = (unsigned)pos >> 2;
        g = (int)(q & (unsigned)(G - 1));
        s = (int)(((q / (unsigned)G) << 2) | ((unsigned)pos & 3u));
    } else {
        g = (int)((unsigned)pos / (unsigned)T);
        s = pos - g * T;
    }
}
// the same with G, T as run-time values (read-outs, exact_site_wave)
__device__ __forceinline__ int pos_of_rt(int G, int T, int g, int s)
{
    return (T & 3) == 0 ? ((((s >> 2) * G) + g) << 2) | (s & 3) : g * T + s;
}
__device__ __forceinline__ void lane_slot_of_rt(int G, int T, int pos, int &g, int &s)
{
    if ((T & 3) == 0) {
        const int q = pos >> 2;
        g = q % G;
        s = ((q / G) << 2) | (pos & 3);
    } else {
        g = pos / T;
        s = pos - g * T;
    }
}

// the T slots of lane `lig` of the row starting at `row`
template <int G, int T>
__device__ __forceinline__ void load_lane_row(const int32_t *__restrict__ row, int lig, int (&x)[T])
{
    if constexpr (T % 4 == 0) {
#pragma unroll
        for (int i = 0; i < T / 4; ++i) {
            const int4 v = reinterpret_cast<const int4 *>(row)[i * G + lig];
            x[4 * i + 0] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w;
        }
    } else if constexpr (T == 2) {
        const int2 v = reinterpret_cast<const int2 *>(row)[lig];
        x[0] = v.x; x[1] = v.y;
    } else {
#pragma unroll
        for (int i = 0; i < T; ++i) x[i] = row[lig * T + i];
    }
}
template <int G, int T>
__device__ __forceinline__ void store_lane_row(int32_t *__restrict__ row, int lig, const int (&x)[T])
{
    if constexpr (T % 4 == 0) {
#pragma unroll
        for (int i = 0; i < T / 4; ++i)
            reinterpret_cast<int4 *>(row)[i * G + lig] = make_int4(x[4 * i], x[4 * i + 1], x[4 * i + 2], x[4 * i + 3]);
    } else if constexpr (T == 2) {
        reinterpret_cast<int2 *>(row)[lig] = make_int2(x[0], x[1]);
    } else {
#pragma unroll
        for (int i = 0; i < T; ++i) row[lig * T + i] = x[i];
    }
}
// the same for the hot loop: uniform base pointer + element offset of the row, global address space
template <int G, int T>
__device__ __forceinline__ void gload_lane_row(const int32_t *base, int64_t row_elem, int lig, int (&x)[T])
{
    const LLDA_GLOBAL int32_t *q = (const LLDA_GLOBAL int32_t *)base + row_elem;
    if constexpr (T % 4 == 0) {
        typedef int v4i __attribute__((ext_vector_type(4)));
        const LLDA_GLOBAL v4i *p = (const LLDA_GLOBAL v4i *)q;
#pragma unroll
        for (int i = 0; i < T / 4; ++i) {
            const v4i v = p[i * G + lig];
            x[4 * i + 0] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < T; ++i) x[i] = q[lig * T + i];
    }
}

// T contiguous int32 starting at p (p is 4*T-byte aligned when T is a multiple of 4).
template <int T>
__device__ __forceinline__ void load_row(const int32_t *__restrict__ p, int (&x)[T])
{
    if constexpr (T % 4 == 0) {
#pragma unroll
        for (int i = 0; i < T / 4; ++i) {
            const int4 v = reinterpret_cast<const int4 *>(p)[i];
            x[4 * i + 0] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w;
        }
    } else if constexpr (T == 2) {
        const int2 v = *reinterpret_cast<const int2 *>(p);
        x[0] = v.x; x[1] = v.y;
    } else {
#pragma unroll
        for (int i = 0; i < T; ++i) x[i] = p[i];
    }
}

template <int T>
__device__ __forceinline__ void store_row(int32_t *__restrict__ p, const int (&x)[T])
{
    if constexpr (T % 4 == 0) {
#pragma unroll
        for (int i = 0; i < T / 4; ++i)
            reinterpret_cast<int4 *>(p)[i] = make_int4(x[4 * i], x[4 * i + 1], x[4 * i + 2], x[4 * i + 3]);
    } else if constexpr (T == 2) {
        *reinterpret_cast<int2 *>(p) = make_int2(x[0], x[1]);
    } else {
#pragma unroll
        for (int i = 0; i < T; ++i) p[i] = x[i];
    }
}

// One-hot slot updates without compares (hipcc turns "(bit) * f" back into v_cmp + v_cndmask and
// spills the masks): m = v_bfe_i32(onehot, S, 1) is 0 or -1, value += m * g via v_mad_i32_i24
// (|g| < 2^23: g is a word frequency inside one document).
template <int S>
__device__ __forceinline__ int onehot_bit(uint32_t oh)
{
    int m;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(oh), "n"(S));
    return m;
}
__device__ __forceinline__ int mad_i24(int a, int b, int c)
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// a[S] += m_S * g and b[S] += m_S * g for every slot S (m_S = 0 / -1)
template <int T, int S = 0>
__device__ __forceinline__ void onehot_add2(int (&a)[T], int (&b)[T], uint32_t oh, int g)
{
    if constexpr (S < T) {
        const int m = onehot_bit<S>(oh);
        a[S] = mad_i24(m, g, a[S]);
        b[S] = mad_i24(m, g, b[S]);
        onehot_add2<T, S + 1>(a, b, oh, g);
    }
}
template <int T, int S = 0>
__device__ __forceinline__ void onehot_add1(int (&a)[T], uint32_t oh, int g)
{
    if constexpr (S < T) {
        a[S] = mad_i24(onehot_bit<S>(oh), g, a[S]);
        onehot_add1<T, S + 1>(a, oh, g);
    }
}

// out-of-place form: b = a + (-1 at the one-hot slot) * g -- saves the register copy when a must survive
template <int T, int S = 0>
__device__ __forceinline__ void onehot_add_to(int (&b)[T], const int (&a)[T], uint32_t oh, int g)
{
    if constexpr (S < T) {
        b[S] = mad_i24(onehot_bit<S>(oh), g, a[S]);
        onehot_add_to<T, S + 1>(b, a, oh, g);
    }
}

template <int T, int S = 0>
__device__ __forceinline__ void scores(double (&w)[T], const int (&ndk)[T], const int (&nkb)[T], const int (&x)[T],
                                       uint32_t mask, double alpha, double beta, double vbeta)
{
    if constexpr (S < T) {
        const double a = (double)ndk[S] + alpha;
        const double num_b = (double)x[S] + beta;
        const double den_b = (double)(nkb[S] + ndk[S]) + vbeta;
        const double ws = a * (num_b / den_b);
        const long long m = (long long)onehot_bit<S>(mask);          // 0 or -1, sign-extended
        w[S] = __longlong_as_double(__double_as_longlong(ws) & m);
        scores<T, S + 1>(w, ndk, nkb, x, mask, alpha, beta, vbeta);
    }
}

// a / b given y = RN(1/b) (IEEE), correctly rounded: q0 = RN(a y); two exact-residual corrections.
__device__ __forceinline__ double div_by(double a, double b, double y)
{
    const double q0 = a * y;
    const double r0 = __builtin_fma(-b, q0, a);
    const double q1 = __builtin_fma(r0, y, q0);
    const double r1 = __builtin_fma(-b, q1, a);
    return __builtin_fma(r1, y, q1);
}

// ---------------------------------------------------------------------------------------------
// Cross-lane moves of doubles without an LDS round trip (DPP / permlane), gfx950.
// ---------------------------------------------------------------------------------------------
// DPP move; lanes whose source lane is outside the row (row_shr) or the wave (wave_shr) receive 0.0.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
constexpr int DPP_XOR1 = 0xB1;          // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;          // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141;  // lane j <- lane 7-j of its 8-lane half (an "xor 4" once quads are uniform)
constexpr int DPP_ROW_SHR = 0x110;      // + n
constexpr int DPP_ROW_ROR = 0x120;      // + n
constexpr int DPP_WAVE_SHR1 = 0x138;

// v_permlane16_swap vdst, src: odd 16-lane rows of vdst <-> even rows of src.  With both operands x:
// r[0] = [R0,R0,R2,R2] (odd rows see the row below), r[1] = [R1,R1,R3,R3] (even rows see the row above).
__device__ __forceinline__ void rows_swapped(double x, double &below, double &above)
{
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(x), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(x), false, false);
    below = __hiloint2double((int)hi[0], (int)lo[0]);
    above = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double xor16_f64(double x, int lane)
{
    double below, above;
    rows_swapped(x, below, above);
    return (lane & 16) ? below : above;
}
// v_permlane32_swap vdst, src: upper half of vdst <-> lower half of src.  r[0] = [lo,lo], r[1] = [hi,hi].
__device__ __forceinline__ double xor32_f64(double x, int lane)
{
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(x), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(x), false, false);
    return (lane & 32) ? __hiloint2double((int)hi[0], (int)lo[0]) : __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double readlane_f64(double x, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l),
                            __builtin_amdgcn_readlane(__double2loint(x), l));
}
// value of the last lane of the caller's group
template <int G>
__device__ __forceinline__ double bcast_last(double x, int lane)
{
    if constexpr (G == 64) {
        return readlane_f64(x, 63);
    } else if constexpr (G == 32) {
        const double a = readlane_f64(x, 31), b = readlane_f64(x, 63);
        return (lane & 32) ? b : a;
    } else if constexpr (G == 16) {
        const double a = readlane_f64(x, 15), b = readlane_f64(x, 31), c = readlane_f64(x, 47), d = readlane_f64(x, 63);
        const double ab = (lane & 16) ? b : a, cd = (lane & 16) ? d : c;
        return (lane & 32) ? cd : ab;
    } else {
        return __shfl(x, G - 1, G);
    }
}
// one Hillis-Steele step of the inclusive scan over the G lanes of a group: X[g] = X[g-D] + X[g], g >= D
template <int G, int D>
__device__ __forceinline__ double scan_step(double X, int lig)
{
    if constexpr (G == 64) {
        const double y = __shfl_up(X, D, G);
        return (lig >= D) ? y + X : X;
    } else if constexpr (G == 32) {
        if constexpr (D < 16) {
            const double y = dpp_f64<DPP_ROW_ROR + D>(X);      // lane i <- lane (i-D) mod 16 of its row
            double below, above;
            rows_swapped(y, below, above);                     // odd rows: the same rotation of the row below
            const double src = ((lig & 15) >= D) ? y : ((lig >= 16) ? below : 0.0);
            return src + X;
        } else {
            double below, above;
            rows_swapped(X, below, above);
            return ((lig >= 16) ? below : 0.0) + X;
        }
    } else {
        const double y = dpp_f64<DPP_ROW_SHR + D>(X);          // 0.0 shifted in at the row start
        if constexpr (G == 16) return y + X;
        else return ((lig >= D) ? y : 0.0) + X;                // 8-lane groups share a row
    }
}
template <int G, int D = 1>
__device__ __forceinline__ double group_scan(double X, int lig)
{
    if constexpr (D < G) return group_scan<G, D * 2>(scan_step<G, D>(X, lig), lig);
    else return X;
}

// value held by lane `src` (same for the whole group, but a run-time value) in every lane of the group.
// 8- and 16-lane groups: OR-butterfly over DPP moves (no LDS round trip); wider groups: ds_bpermute.
template <int G>
__device__ __forceinline__ int group_pick(int v, int src, int lig)
{
    if constexpr (G <= 16) {
        int x = (lig == src) ? v : 0;
        x |= __builtin_amdgcn_update_dpp(0, x, DPP_XOR1, 0xF, 0xF, false);
        x |= __builtin_amdgcn_update_dpp(0, x, DPP_XOR2, 0xF, 0xF, false);
        x |= __builtin_amdgcn_update_dpp(0, x, DPP_HALF_MIRROR, 0xF, 0xF, false);
        if constexpr (G == 16) x |= __builtin_amdgcn_update_dpp(0, x, DPP_ROW_ROR + 8, 0xF, 0xF, false);
        return x;
    } else {
        return __shfl(v, src, G);
    }
}

// Sum of the group's K scores in numpy's pairwise order.  Every lane of the group returns S.
//   chain : per-lane sequential sum over its slots (one of numpy's 8 accumulators)
//   xor butterfly 1,2,4 : ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))   (fp add is commutative)
//   tail  : n % 8 leftovers of the last leaf, added sequentially
//   leaves: combined along numpy's recursion tree by the partner schedule
// cross-lane part of the sum: acc = this lane's chain, tv = this lane's tail element
template <int G, bool HAS_TAIL>
__device__ __forceinline__ double group_sum_tail(double acc, double tv, const KParams &P, int lig, int lane)
{
    const int leaf = lig >> 3;
    acc = acc + dpp_f64<DPP_XOR1>(acc);
    acc = acc + dpp_f64<DPP_XOR2>(acc);
    acc = acc + dpp_f64<DPP_HALF_MIRROR>(acc);
    if (HAS_TAIL) {
        for (int t = 0; t < P.tail; ++t) {
            const double o = __shfl(tv, P.last_leaf * 8 + t, G);
            if (leaf == P.last_leaf) acc = acc + o;
        }
    }
    if constexpr (G > 8) {
        if (P.xor_tree) {
            // balanced recursion (leaf p pairs with p^1, then p^2, p^4): lane xor 8 / 16 / 32
            acc = acc + dpp_f64<DPP_ROW_ROR + 8>(acc);
            if constexpr (G > 16) acc = acc + xor16_f64(acc, lane);
            if constexpr (G > 32) acc = acc + xor32_f64(acc, lane);
        } else {
#pragma unroll
            for (int r = 0; r < LLDA_MAX_ROUNDS; ++r) {
                if (r < P.n_rounds) {
                    const int partner = (P.rounds_pk[r] >> (4 * leaf)) & 15;
                    const double o = __shfl(acc, partner * 8 + (lig & 7), G);
                    if (partner != leaf) acc = acc + o;
                }
            }
            acc = __shfl(acc, 0, G);
        }
    }
    return acc;
}

template <int G, int T, bool HAS_TAIL>
__device__ __forceinline__ double group_sum(const double (&w)[T], const KParams &P, int lig, int lane)
{
    const int leaf = lig >> 3;
    double acc = 0.0, tv = 0.0;
#pragma unroll
    for (int s = 0; s < T; ++s) {
        if (HAS_TAIL && s == P.tail_row && leaf == P.last_leaf) tv = w[s];
        else acc = acc + w[s];
    }
    return group_sum_tail<G, HAS_TAIL>(acc, tv, P, lig, lane);
}

// Keyed categorical draw over the group's K probabilities p (device order, oracle/llda_oracle.py
// draw_keyed): q = per-lane prefix over the slots, X = Hillis-Steele scan of the lane totals,
// t = u * X[G-1]; result = first position with p > 0 and q > t - X[lane-1], else the last position with
// p > 0; -1 if there is none (or !valid).  FAST: "p > 0" is read off the label mask.
template <int G, int T, bool FAST>
__device__ __forceinline__ int draw_position(const double (&w)[T], double u, uint32_t mask, bool valid, int lig, int lane)
{
    const int gbase = lane & ~(G - 1);
    const uint64_t gmask = (G == 64) ? ~0ull : ((1ull << G) - 1ull);
    double q[T];
    q[0] = w[0];
#pragma unroll
    for (int s = 1; s < T; ++s) q[s] = q[s - 1] + w[s];
    const double X = group_scan<G>(q[T - 1], lig);
    const double tot = bcast_last<G>(X, lane);
    const double t = u * tot;
    const double prev = dpp_f64<DPP_WAVE_SHR1>(X);
    const double tg = t - (lig ? prev : 0.0);
    uint32_t fm = 0, pm = 0;
    if (FAST) {
        // q is non-decreasing along the slots, so {s : q[s] > tg} is the suffix starting at
        // cnt = #{s : q[s] <= tg}; positive-probability slots are the label-mask bits.
        int cnt = 0;
#pragma unroll
        for (int s = 0; s < T; ++s) cnt += (q[s] <= tg) ? 1 : 0;
        pm = mask;
        fm = mask & (0xFFFFu << cnt);
    } else {
#pragma unroll
        for (int s = 0; s < T; ++s) {
            const bool pos = w[s] > 0.0;
            pm |= (pos ? 1u : 0u) << s;
            fm |= ((pos && q[s] > tg) ? 1u : 0u) << s;
        }
    }
    const uint64_t gf = (__ballot(fm != 0) >> gbase) & gmask;
    const uint64_t gp = (__ballot(pm != 0) >> gbase) & gmask;
    int zn = -1;
    if (gp != 0 && valid) {
        const bool hit = gf != 0;
        const int sl = hit ? (int)__ffsll((unsigned long long)gf) - 1 : 63 - (int)__clzll((unsigned long long)gp);
        const int my = hit ? (int)__ffs((int)(fm | 0x10000u)) - 1 : 31 - (int)__clz((int)(pm | 1u));
        const int ss = __shfl(my, sl, G);
        zn = pos_of<G, T>(sl, ss);
    }
    return zn;
}

}  // namespace
