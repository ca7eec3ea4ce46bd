// kernel_counts.hpp -- llda_pack_rows16_kernel, llda_commit_log_kernel, llda_apply_delta_kernel, llda_count_init_kernel, division self test
// Part of the single translation unit llda_gibbs.hip (included in order; see the contents list there).
#pragma once

namespace {

// ---------------------------------------------------------------------------------------------
// llda_pack_rows16: the 16-bit image of the n_kw rows flagged in row16 (16 slots per lane only).  A packed row keeps the
// slots 8j .. 8j+7 of ALL lanes contiguous (16 bytes per lane and chunk j = 0, 1), the even slot in the low half of its
// register -- the form llda_sweep_kernel<.., R16> loads with two global_load_dwordx4 per lane.  One thread per 16-byte
// chunk.  A flagged row with a count outside 0 .. 65535 sets bit 2 of status word 0 (the caller flags only rows whose total
// -- which Gibbs sampling conserves -- fits).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) llda_pack_rows16_kernel(const int32_t *__restrict__ n_kw, const uint8_t *__restrict__ row16,
                                                               uint16_t *__restrict__ out, int64_t V, int G, int32_t *status)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int per_row = 2 * G;
    const int64_t w = t / per_row;
    if (w >= V || !row16[w]) return;
    const int c = (int)(t - w * per_row), j = c / G, g = c - j * G;
    const int4 *src = reinterpret_cast<const int4 *>(n_kw + w * (int64_t)(G * 16));
    const int4 a = src[(2 * j) * G + g], b = src[(2 * j + 1) * G + g];
    const int m = a.x | a.y | a.z | a.w | b.x | b.y | b.z | b.w;
    if ((unsigned)m > 0xffffu && status) atomicOr(status, 4);
    uint4 o;
    o.x = ((uint32_t)a.x & 0xffffu) | ((uint32_t)a.y << 16);
    o.y = ((uint32_t)a.z & 0xffffu) | ((uint32_t)a.w << 16);
    o.z = ((uint32_t)b.x & 0xffffu) | ((uint32_t)b.y << 16);
    o.w = ((uint32_t)b.z & 0xffffu) | ((uint32_t)b.w << 16);
    reinterpret_cast<uint4 *>(out + w * (int64_t)(G * 16))[j * G + g] = o;
}

// ---------------------------------------------------------------------------------------------
// llda_pack_image: the saturating narrow image of n_kw the sparse-label kernels gather from (llda_sweep_args.n_kw_img).  One
// thread per four counts: a 16-byte load, a 4- or 8-byte store.
// ---------------------------------------------------------------------------------------------
template <int BITS>
__global__ void __launch_bounds__(256) llda_pack_image_kernel(const int4 *__restrict__ n_kw, void *__restrict__ out, int64_t n4)
{
    constexpr uint32_t SAT = BITS == 8 ? 255u : 65535u;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n4; t += (int64_t)gridDim.x * 256) {
        const int4 a = n_kw[t];
        const uint32_t x = min((uint32_t)a.x, SAT), y = min((uint32_t)a.y, SAT), z = min((uint32_t)a.z, SAT), w = min((uint32_t)a.w, SAT);
        if constexpr (BITS == 8) {
            static_cast<uint32_t *>(out)[t] = x | (y << 8) | (z << 16) | (w << 24);
        } else {
            uint2 o;
            o.x = x | (y << 16);
            o.y = z | (w << 16);
            static_cast<uint2 *>(out)[t] = o;
        }
    }
}

// the same with the image's columns in an order of their own: image column c of every row holds the count at position col_src[c].
// One workgroup per row at a time: the row comes in with 16-byte loads and goes through LDS, where every thread picks the four counts
// of its four image columns (a 4- or 8-byte store) -- the gathers never leave the CU.
template <int BITS>
__global__ void __launch_bounds__(256) llda_pack_image_cols_kernel(const int32_t *__restrict__ n_kw, const int32_t *__restrict__ col_src,
                                                                   void *__restrict__ out, int64_t V, int KP)
{
    constexpr uint32_t SAT = BITS == 8 ? 255u : 65535u;
    extern __shared__ int s_row[];                       // KP counts
    const int q4 = KP / 4;
    for (int64_t v = blockIdx.x; v < V; v += gridDim.x) {
        const int4 *src = reinterpret_cast<const int4 *>(n_kw + v * KP);
        for (int i = threadIdx.x; i < q4; i += 256) reinterpret_cast<int4 *>(s_row)[i] = src[i];
        __syncthreads();
        for (int i = threadIdx.x; i < q4; i += 256) {
            // (col_src is the caller's: an entry outside 0 .. KP-1 reads position KP-1 instead of someone else's LDS -- the image is
            // then not the one the caller meant, but nothing out of bounds is touched; include/llda_gibbs.h states the precondition)
            const int4 cs = reinterpret_cast<const int4 *>(col_src)[i];
            const uint32_t last = (uint32_t)KP - 1u;
            const uint32_t x = min((uint32_t)s_row[min((uint32_t)cs.x, last)], SAT), y = min((uint32_t)s_row[min((uint32_t)cs.y, last)], SAT),
                           z = min((uint32_t)s_row[min((uint32_t)cs.z, last)], SAT), w = min((uint32_t)s_row[min((uint32_t)cs.w, last)], SAT);
            if constexpr (BITS == 8) {
                static_cast<uint32_t *>(out)[v * q4 + i] = x | (y << 8) | (z << 16) | (w << 24);
            } else {
                uint2 o;
                o.x = x | (y << 16);
                o.y = z | (w << 16);
                static_cast<uint2 *>(out)[v * q4 + i] = o;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Fold of the commit log into word-major counts (llda_commit_log, include/llda_gibbs.h): one wavefront per
// item (a run of log entries of one word), a histogram of the word's row per wavefront in LDS.  The histogram
// is flushed either by walking the item's entries again (short items: each touched entry is claimed with an LDS
// exchange, so every entry has exactly one writer) or by scanning the whole row (long items).
// With a row table the rows live at row_off[v] in `target`, and a NEGATIVE offset marks a row of int16 PAIRS:
// position p is the low (p even) or high (p odd) half of word ~row_off[v] + p/2, and counts are added as
// f * 65536^(p & 1) -- plain int32 adds of such words are exact as long as every half stays inside int16
// (llda_apply_rows decodes them), which is what lets the per-sweep all-reduce move half the bytes.
// ---------------------------------------------------------------------------------------------
struct CParams {
    const int64_t *item_begin;
    const int32_t *item_len, *item_word;
    int64_t n_items;
    const uint32_t *log;
    const int32_t *freq;
    const int64_t *row_off;
    int32_t *target, *n_k, *n_k_delta;
    int32_t KP;
};

__device__ __forceinline__ void add_count(int32_t *p, int a, bool shared_row)
{
    if (shared_row) atomicAdd(p, a);
    else *p += a;
}

__global__ void __launch_bounds__(256) llda_commit_log_kernel(const CParams P)
{
    extern __shared__ int s_hist[];               // [4][KP]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int KP = P.KP;
    if (blockIdx.x == 0 && P.n_k)
        for (int p = tid; p < KP; p += 256) {
            P.n_k[p] += P.n_k_delta[p];
            P.n_k_delta[p] = 0;
        }
    int *hist = s_hist + w * KP;
    for (int p = lane; p < KP; p += 64) hist[p] = 0;
    const int64_t item = (int64_t)blockIdx.x * 4 + w;
    if (item >= P.n_items) return;
    const int64_t b = P.item_begin[item];
    const int len = P.item_len[item];
    const int wv = P.item_word[item];
    const bool shared_row = wv < 0;
    const int64_t v = wv & 0x7fffffff;
    const int64_t ro = P.row_off ? P.row_off[v] : v * KP;
    const bool pairs = ro < 0;                    // int16 pairs: entry p -> word p >> 1, weight 65536^(p & 1)
    int32_t *row = P.target + (pairs ? ~ro : ro);
    const int sh = pairs ? 1 : 0, n_words = KP >> sh;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    // the entries of an item are read FOUR per lane and load (16-byte loads behind a head of at most three entries that brings the
    // address to 16 bytes): a quarter of the load instructions and four times the bytes in flight per wavefront -- the fold streams
    // 2.4 GB per sweep of configs[3] and was bound by the latency of its 4-byte loads: 0.81 -> 0.56 ms (profiles/r06_site_loop_budget.md section 5)
    const uint32_t *lg = P.log + b;
    const int32_t *fq = P.freq + b;
    // (the head from the ADDRESSES: a caller's arrays need only be 4-byte aligned; log and freq misaligned differently: one entry per load)
    const unsigned mis_l = (unsigned)(reinterpret_cast<uintptr_t>(lg) & 15), mis_f = (unsigned)(reinterpret_cast<uintptr_t>(fq) & 15);
    const bool vec = mis_l == mis_f && len >= 256;       // (short items -- the words of a sparse-label corpus -- gain nothing from the head / tail split)
    const int head = vec ? min(len, (int)(((16u - mis_l) & 15u) >> 2)) : 0, n4 = vec ? (len - head) >> 2 : 0, tail0 = head + 4 * n4;
    auto count1 = [&](uint32_t e, int f) {
        const int zo = (int)(e & 0xFFFFu), zn = (int)(e >> 16);
        if (zo != zn) {
            atomicAdd(&hist[zo >> sh], -(int)((uint32_t)f << ((zo & sh) * 16)));
            atomicAdd(&hist[zn >> sh], (int)((uint32_t)f << ((zn & sh) * 16)));
        }
    };
    if (lane < head) count1(lg[lane], fq[lane]);
    for (int g = lane; g < n4; g += 64) {
        const uint4 e = *reinterpret_cast<const uint4 *>(lg + head + 4 * g);
        const int4 f = *reinterpret_cast<const int4 *>(fq + head + 4 * g);
        count1(e.x, f.x); count1(e.y, f.y); count1(e.z, f.z); count1(e.w, f.w);
    }
    for (int j = tail0 + lane; j < len; j += 64) count1(lg[j], fq[j]);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    if (len <= n_words) {
        auto flush1 = [&](uint32_t e) {
            const int zo = (int)(e & 0xFFFFu), zn = (int)(e >> 16);
            if (zo != zn) {
                const int a = atomicExch(&hist[zo >> sh], 0), c = atomicExch(&hist[zn >> sh], 0);
                if (a) add_count(row + (zo >> sh), a, shared_row);
                if (c) add_count(row + (zn >> sh), c, shared_row);
            }
        };
        if (lane < head) flush1(lg[lane]);
        for (int g = lane; g < n4; g += 64) {
            const uint4 e = *reinterpret_cast<const uint4 *>(lg + head + 4 * g);
            flush1(e.x); flush1(e.y); flush1(e.z); flush1(e.w);
        }
        for (int j = tail0 + lane; j < len; j += 64) flush1(lg[j]);
    } else {
        for (int p = lane; p < n_words; p += 64) {
            const int a = hist[p];
            if (a) add_count(row + p, a, shared_row);
        }
    }
}

// counts[r][:] += row r of `rows` (decoding int16 pairs), row cleared: one wavefront per row (llda_apply_rows)
__global__ void __launch_bounds__(256) llda_apply_rows_kernel(const int64_t *__restrict__ row_off, int32_t *rows,
                                                             int64_t n_rows, int KP, int32_t *counts)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const int64_t ro = row_off[r];
    int32_t *dst = counts + r * KP;
    if (ro < 0) {
        int32_t *src = rows + ~ro;
        for (int j = lane; j < KP / 2; j += 64) {
            const int32_t s = src[j];
            if (s) {
                const int lo = (int)(int16_t)(s & 0xFFFF);
                const int hi = (int)(((int64_t)s - lo) >> 16);
                src[j] = 0;
                int2 *d2 = reinterpret_cast<int2 *>(dst) + j;
                int2 c = *d2;
                c.x += lo; c.y += hi;
                *d2 = c;
            }
        }
    } else {
        int32_t *src = rows + ro;
        for (int p = lane; p < KP; p += 64) {
            const int32_t s = src[p];
            if (s) { src[p] = 0; dst[p] += s; }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// self test: div_by (reciprocal + two corrections) against the hardware IEEE division
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) llda_selftest_div_kernel(uint64_t seed, int iters, unsigned long long *bad)
{
    uint32_t mism = 0;
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        uint32_t c0 = tid, c1 = (uint32_t)i, c2 = 0x5e1f7e57u, c3 = 0;
        philox4x32_10(c0, c1, c2, c3, (uint32_t)seed, (uint32_t)(seed >> 32));
        // b: a sum-like positive double with a random 52-bit significand and exponent in [-20, 40];
        // a: anything from 0 to 2^60 times smaller than b up to a few times b
        const uint64_t mb = ((uint64_t)(c0 & 0xFFFFFu) << 32) | c1;
        const uint64_t ma = ((uint64_t)(c2 & 0xFFFFFu) << 32) | c3;
        const int eb = (int)((c0 >> 20) % 61) - 20;
        const int ea = eb + 2 - (int)((c2 >> 20) % 64);
        double b = __longlong_as_double((long long)(((uint64_t)(1023 + eb) << 52) | mb));
        double a = __longlong_as_double((long long)(((uint64_t)(1023 + ea) << 52) | ma));
        if ((i & 7) == 7) {            // integer-valued operands, the shape of the count terms
            b = (double)(c0 >> 4) + 1000.0 * 1.0000000000000002;
            a = (double)(c2 >> 12) * 0.1;
        }
        if ((i & 63) == 63) a = 0.0;
        const double y = 1.0 / b;
        if (div_by(a, b, y) != a / b) ++mism;
    }
    if (mism) atomicAdd(bad, (unsigned long long)mism);
}

// ---------------------------------------------------------------------------------------------
// helpers: fold deltas, build counts from assignments
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) llda_apply_delta_kernel(int32_t *__restrict__ counts,
                                                               int32_t *__restrict__ delta, int64_t n4,
                                                               int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int4 *c4 = reinterpret_cast<int4 *>(counts);
    int4 *d4 = reinterpret_cast<int4 *>(delta);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        int4 c = c4[i];
        const int4 d = d4[i];
        if (d.x | d.y | d.z | d.w) {
            c.x += d.x; c.y += d.y; c.z += d.z; c.w += d.w;
            c4[i] = c;
            d4[i] = make_int4(0, 0, 0, 0);
        }
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        counts[i] += delta[i];
        delta[i] = 0;
    }
}

__global__ void __launch_bounds__(256) llda_count_init_kernel(const int64_t *__restrict__ doc_off,
                                                              const int32_t *__restrict__ word,
                                                              const int32_t *__restrict__ freq,
                                                              const int32_t *__restrict__ z, int64_t D, int KP,
                                                              int32_t *n_dk, int32_t *n_kw, int32_t *n_k)
{
    // one wavefront per document at a time, lanes stride over its sites.  The document's n_dk row and the
    // workgroup's share of n_k are histograms in LDS (the row is then written with plain stores, n_k with KP
    // atomics per workgroup); only n_kw takes one global atomic per site.
    extern __shared__ int s_init[];               // [KP] n_k of the workgroup, then [waves][KP] per-wavefront rows
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nthr = blockDim.x, nw = nthr >> 6;  // 4 wavefronts per workgroup; 1 for the long rows of wide layouts (LDS)
    int *s_nk = s_init, *hist = s_init + (1 + w) * KP;
    for (int p = tid; p < KP; p += nthr) s_nk[p] = 0;
    for (int p = lane; p < KP; p += 64) hist[p] = 0;
    __syncthreads();
    for (int64_t d = (int64_t)blockIdx.x * nw + w; d < D; d += (int64_t)gridDim.x * nw) {
        for (int64_t i = doc_off[d] + lane; i < doc_off[d + 1]; i += 64) {
            const int f = freq[i], p = z[i];
            atomicAdd(&hist[p], f);
            atomicAdd(n_kw + (int64_t)word[i] * KP + p, f);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        for (int p = lane; p < KP; p += 64) {
            const int h = hist[p];
            if (h) {
                hist[p] = 0;
                n_dk[d * KP + p] += h;
                atomicAdd(&s_nk[p], h);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    }
    __syncthreads();
    for (int p = tid; p < KP; p += nthr) {
        const int h = s_nk[p];
        if (h) atomicAdd(n_k + p, h);
    }
}

}  // namespace
