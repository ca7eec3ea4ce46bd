// llda_gibbs.hip -- collapsed-Gibbs sweep for Labeled LDA / CascadeLDA on MI355X (gfx950, wave64).
//
// Hot path replaced: LabeledLDA.training_iteration (/root/reference/LabeledLDA.py:101-125) ==
// SubLDA.training_iteration (/root/reference/CascadeLDA.py:397-421).  C ABI: include/llda_gibbs.h.
//
// Execution model (DESIGN.md sections 3-4):
//   * group layout: one lane GROUP of G = 8..64 lanes per document (64/G documents per wavefront), T topic
//     slots per lane; topic k lives at (lane, slot) chosen so that numpy's pairwise summation order (np.sum at
//     LabeledLDA.py:117) is lane-local: one accumulator chain = one lane walking its slots, the 8 accumulators
//     of a 128-topic leaf = 8 neighbouring lanes;
//   * the n_kw row of the current word is one contiguous KP*4-byte read (word-major layout), prefetched one
//     site ahead; count updates are int32 atomics into n_kw_delta and, through an LDS accumulator per workgroup,
//     into n_k_delta.  Snapshot semantics + integer atomics => bit-deterministic;
//   * the draw is TIERED: an fp32 decision with a proven margin, an fp64 decision, and the
//     reference's fp64 pipeline bit for bit (IEEE division, numpy-ordered sum, keyed Philox4x32-10 draw) for
//     the sites the cheaper tiers cannot decide -- the chosen topic is always the exact pipeline's.
// No MFMA (gather/scan, not a contraction).  FMA contraction is OFF: the reference rounds after every ufunc.
//
// Contents (one translation unit; the headers are included in this order)
//   device_common.hpp   kernel parameters, Philox, row loads, one-hot updates, exact division, DPP / permlane
//                       cross-lane moves, numpy-ordered group sum, keyed categorical draw (exact)
//   draw_tiers.hpp      tier-0 (fp32) decision, cold tiers (fp64 decision, exact pipeline), commit of a site
//   exact_generic.hpp   exact_site_wave           the exact pipeline with the layout known at run time (a whole wave per
//                                                 document): the sparse kernels' answer to a site they cannot decide
//   kernel_sweep.hpp    llda_sweep_exact_kernel   general kernel, every site through the exact pipeline
//                       llda_sweep_kernel         tiered kernel, per-document state in LDS (the hot kernel)
//   kernel_quad.hpp     llda_sweep_quad_kernel    K = 512 dense, 16-bit image: FOUR documents per wavefront (the bench's timed kernel)
//   kernel_sparse.hpp   llda_sweep_sparse_kernel  one lane per ALLOWED topic for sparse label sets
//   kernel_batch.hpp    llda_sweep_batch_kernel   one sweep over many independent small problems (CascadeLDA's
//                                                 ensemble) in one launch, sparse-kernel arithmetic
//   kernel_readout.hpp  llda_loglik_kernel, llda_readout_phi / _theta kernels (thinning read-outs)
//   kernel_foldin.hpp   llda_foldin_kernel (test-time sampler)
//   kernel_counts.hpp   llda_commit_log_kernel, llda_apply_delta_kernel, llda_count_init_kernel, self test
//   kernel_wide.hpp     the general path for K with more than 8 pairwise leaves (one wavefront per document)
//   this file           host side: layout (llda_layout_init), dispatch, C entry points
#include <hip/hip_runtime.h>
#include "build_info.hpp"
#ifndef ABL_EXTRA_LDS_BYTES
#define ABL_EXTRA_LDS_BYTES 0      // occupancy ablation: dynamic LDS nobody uses (tools: -DABL_EXTRA_LDS_BYTES=20000 -> 2 workgroups per CU)
#endif
#include <stdint.h>
#include <math.h>
#include <string.h>
#include <map>
#include <memory>
#include <type_traits>

#include "llda_gibbs.h"

#pragma clang fp contract(off)

#include "device_common.hpp"
#include "draw_tiers.hpp"
#include "exact_generic.hpp"
#include "kernel_sweep.hpp"
#include "kernel_quad.hpp"
#include "kernel_sparse.hpp"
#include "kernel_batch.hpp"
#include "kernel_readout.hpp"
#include "kernel_foldin.hpp"
#include "kernel_counts.hpp"
#include "kernel_wide.hpp"

namespace {

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
void add_leaves(llda_layout *L, int n, int start)
{
    if (n <= 128) {
        if (L->n_leaves < LLDA_MAX_WIDE_LEAVES) {
            L->leaf_start[L->n_leaves] = start;
            L->leaf_len[L->n_leaves] = n;
        }
        L->n_leaves++;
        return;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    add_leaves(L, n2, start);
    add_leaves(L, n - n2, start + n2);
}

// numpy's recursion over leaf ranges [first, first+count): returns depth, fills the schedule
int schedule(llda_layout *L, int n, int *next_leaf, int *first_out, int *count_out)
{
    if (n <= 128) {
        *first_out = (*next_leaf)++;
        *count_out = 1;
        return 0;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    int lf, lc, rf, rc;
    const int dl = schedule(L, n2, next_leaf, &lf, &lc);
    const int dr = schedule(L, n - n2, next_leaf, &rf, &rc);
    const int d = (dl > dr ? dl : dr);      // this node combines in round d
    if (d < LLDA_MAX_ROUNDS) {
        for (int a = lf; a < lf + lc; ++a) L->rounds[d][a] = rf;
        for (int b = rf; b < rf + rc; ++b) L->rounds[d][b] = lf;
    }
    if (d + 1 > L->n_rounds) L->n_rounds = d + 1;
    *first_out = lf;
    *count_out = lc + rc;
    return d + 1;
}

// numpy's recursion as in-place adds over the leaf totals, post-order: returns the first leaf of the subtree
int comb_tree(llda_layout *L, int n, int *next_leaf, int *n_comb)
{
    if (n <= 128) return (*next_leaf)++;
    int n2 = n / 2;
    n2 -= n2 % 8;
    const int a = comb_tree(L, n2, next_leaf, n_comb);
    const int b = comb_tree(L, n - n2, next_leaf, n_comb);
    if (*n_comb < LLDA_MAX_WIDE_LEAVES) { L->comb_dst[*n_comb] = a; L->comb_src[*n_comb] = b; }
    ++*n_comb;
    return a;
}

int hip_fail(hipError_t e)
{
    g_last_hip_error = (int)e;
    return LLDA_E_HIP;
}

// The layout of K (128 KB of tables): built once per host thread and K, not once per call.
const llda_layout *layout_of(int32_t K, int *rc)
{
    static thread_local std::map<int32_t, std::unique_ptr<llda_layout>> cache;
    auto it = cache.find(K);
    if (it != cache.end()) { *rc = LLDA_OK; return it->second.get(); }
    std::unique_ptr<llda_layout> L(new llda_layout);
    *rc = llda_layout_init(K, L.get());
    if (*rc) return nullptr;
    if (cache.size() >= 256) cache.clear();
    return (cache[K] = std::move(L)).get();
}

// kernels of the wide path take their LDS as a run-time size; above 64 KB a kernel has to be told once
template <typename Kern>
int allow_lds(Kern kern, size_t bytes)
{
    if (bytes > 160 * 1024) return LLDA_E_BAD_K;
    if (bytes > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return hip_fail(e);
    }
    return LLDA_OK;
}

void fill_wide(const llda_layout &L, WideLayout &W)
{
    W.NT = L.tiers; W.T = L.T; W.G = L.G; W.KP = L.KP;
    W.m = L.n_leaves; W.last_leaf = L.n_leaves - 1; W.tail = L.tail; W.tail_row = L.tail_row;
    for (int i = 0; i < LLDA_MAX_WIDE_LEAVES; ++i) {
        W.comb_dst[i] = (uint8_t)L.comb_dst[i];
        W.comb_src[i] = (uint8_t)L.comb_src[i];
    }
}

// one wavefront per document (or site): enough workgroups to fill the chip a few times over
unsigned wide_blocks(int64_t n)
{
    const int64_t cap = 256 * 16;
    return (unsigned)(n < 1 ? 1 : (n < cap ? n : cap));
}

template <int G, int T>
int launch_sweep(const KParams &P, bool has_tail, bool fast, bool dense, int64_t blocks, hipStream_t st)
{
    const dim3 grid((unsigned)blocks), block(256);
    if constexpr (G <= 16) {
        if (fast && P.commit_log && P.site_rec) {       // 16-byte site records (llda_sweep_args.site_rec)
            if (dense) hipLaunchKernelGGL((llda_sweep_kernel<G, T, false, true, true, true>), grid, block, 0, st, P);
            else if (has_tail) hipLaunchKernelGGL((llda_sweep_kernel<G, T, true, false, true, true>), grid, block, 0, st, P);
            else hipLaunchKernelGGL((llda_sweep_kernel<G, T, false, false, true, true>), grid, block, 0, st, P);
            const hipError_t e = hipGetLastError();
            return e == hipSuccess ? LLDA_OK : hip_fail(e);
        }
    }
    if constexpr (G >= 32 && T == 16) {
        if (P.n_kw16) {                                 // (llda_sweep checked: fast, dense, commit log)
            if (P.w4) hipLaunchKernelGGL((llda_sweep_kernel<G, T, false, true, true, false, true, true>), grid, block, 0, st, P);
            else hipLaunchKernelGGL((llda_sweep_kernel<G, T, false, true, true, false, true>), grid, block, ABL_EXTRA_LDS_BYTES, st, P);
            const hipError_t e = hipGetLastError();
            return e == hipSuccess ? LLDA_OK : hip_fail(e);
        }
    }
    if (!fast) {
        if (has_tail) hipLaunchKernelGGL((llda_sweep_exact_kernel<G, T, true>), grid, block, 0, st, P);
        else hipLaunchKernelGGL((llda_sweep_exact_kernel<G, T, false>), grid, block, 0, st, P);
    } else if (dense) {
        if (P.commit_log) hipLaunchKernelGGL((llda_sweep_kernel<G, T, false, true, true>), grid, block, 0, st, P);
        else hipLaunchKernelGGL((llda_sweep_kernel<G, T, false, true, false>), grid, block, 0, st, P);
    } else if (has_tail) {
        if (P.commit_log) hipLaunchKernelGGL((llda_sweep_kernel<G, T, true, false, true>), grid, block, 0, st, P);
        else hipLaunchKernelGGL((llda_sweep_kernel<G, T, true, false, false>), grid, block, 0, st, P);
    } else {
        if (P.commit_log) hipLaunchKernelGGL((llda_sweep_kernel<G, T, false, false, true>), grid, block, 0, st, P);
        else hipLaunchKernelGGL((llda_sweep_kernel<G, T, false, false, false>), grid, block, 0, st, P);
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LLDA_OK : hip_fail(e);
}

template <int G>
int dispatch_sweep_T(int T, const KParams &P, bool has_tail, bool fast, bool dense, int64_t blocks, hipStream_t st)
{
    if constexpr (G == 8) {
        switch (T) {
        case 1: return launch_sweep<8, 1>(P, has_tail, fast, dense, blocks, st);
        case 2: return launch_sweep<8, 2>(P, has_tail, fast, dense, blocks, st);
        case 4: return launch_sweep<8, 4>(P, has_tail, fast, dense, blocks, st);
        case 8: return launch_sweep<8, 8>(P, has_tail, fast, dense, blocks, st);
        }
    }
    switch (T) {
    case 12: return launch_sweep<G, 12>(P, has_tail, fast, dense, blocks, st);
    case 16: return launch_sweep<G, 16>(P, has_tail, fast, dense, blocks, st);
    }
    return LLDA_E_BAD_K;
}

template <int G, int T>
int launch_loglik(const LParams &P, hipStream_t st)
{
    const int64_t blocks = (P.D + (256 / G) - 1) / (256 / G);
    hipLaunchKernelGGL((llda_loglik_kernel<G, T>), dim3((unsigned)blocks), dim3(256), 0, st, P);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LLDA_OK : hip_fail(e);
}

template <int G>
int dispatch_loglik_T(int T, const LParams &P, hipStream_t st)
{
    if constexpr (G == 8) {
        switch (T) {
        case 1: return launch_loglik<8, 1>(P, st);
        case 2: return launch_loglik<8, 2>(P, st);
        case 4: return launch_loglik<8, 4>(P, st);
        case 8: return launch_loglik<8, 8>(P, st);
        }
    }
    switch (T) {
    case 12: return launch_loglik<G, 12>(P, st);
    case 16: return launch_loglik<G, 16>(P, st);
    }
    return LLDA_E_BAD_K;
}

template <int G, int T>
int launch_foldin(const FParams &P, bool has_tail, hipStream_t st)
{
    if (P.n_sites > 0) {                              // initial assignments: one lane group per site
        const int64_t ib = (P.n_sites + (256 / G) - 1) / (256 / G);
        if (ib > 0x7fffffffLL) return LLDA_E_BAD_ARG;
        if (has_tail) hipLaunchKernelGGL((llda_foldin_init_kernel<G, T, true>), dim3((unsigned)ib), dim3(256), 0, st, P);
        else hipLaunchKernelGGL((llda_foldin_init_kernel<G, T, false>), dim3((unsigned)ib), dim3(256), 0, st, P);
    }
    const int64_t blocks = (P.D + (256 / G) - 1) / (256 / G);
    if (has_tail) hipLaunchKernelGGL((llda_foldin_kernel<G, T, true>), dim3((unsigned)blocks), dim3(256), 0, st, P);
    else hipLaunchKernelGGL((llda_foldin_kernel<G, T, false>), dim3((unsigned)blocks), dim3(256), 0, st, P);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LLDA_OK : hip_fail(e);
}

template <int G>
int dispatch_foldin_T(int T, const FParams &P, bool has_tail, hipStream_t st)
{
    if constexpr (G == 8) {
        switch (T) {
        case 1: return launch_foldin<8, 1>(P, has_tail, st);
        case 2: return launch_foldin<8, 2>(P, has_tail, st);
        case 4: return launch_foldin<8, 4>(P, has_tail, st);
        case 8: return launch_foldin<8, 8>(P, has_tail, st);
        }
    }
    switch (T) {
    case 12: return launch_foldin<G, 12>(P, has_tail, st);
    case 16: return launch_foldin<G, 16>(P, has_tail, st);
    }
    return LLDA_E_BAD_K;
}

template <int G, int T>
int launch_theta(const RParams &P, bool has_tail, hipStream_t st)
{
    const int64_t blocks = (P.D + (256 / G) - 1) / (256 / G);
    if (has_tail) hipLaunchKernelGGL((llda_readout_theta_kernel<G, T, true>), dim3((unsigned)blocks), dim3(256), 0, st, P);
    else hipLaunchKernelGGL((llda_readout_theta_kernel<G, T, false>), dim3((unsigned)blocks), dim3(256), 0, st, P);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LLDA_OK : hip_fail(e);
}

template <int G>
int dispatch_theta_T(int T, const RParams &P, bool has_tail, hipStream_t st)
{
    if constexpr (G == 8) {
        switch (T) {
        case 1: return launch_theta<8, 1>(P, has_tail, st);
        case 2: return launch_theta<8, 2>(P, has_tail, st);
        case 4: return launch_theta<8, 4>(P, has_tail, st);
        case 8: return launch_theta<8, 8>(P, has_tail, st);
        }
    }
    switch (T) {
    case 12: return launch_theta<G, 12>(P, has_tail, st);
    case 16: return launch_theta<G, 16>(P, has_tail, st);
    }
    return LLDA_E_BAD_K;
}

// the summation schedule shared by llda_sweep and llda_foldin
void fill_schedule(const llda_layout &L, int32_t &last_leaf, int32_t &tail, int32_t &tail_row, int32_t &n_rounds,
                   int32_t &xor_tree, uint32_t (&rounds_pk)[LLDA_MAX_ROUNDS])
{
    last_leaf = L.n_leaves - 1; tail = L.tail; tail_row = L.tail_row; n_rounds = L.n_rounds;
    const int P2 = L.G / 8;
    int xt = (L.n_leaves == P2) ? 1 : 0;
    for (int r = 0; (1 << r) < P2 && xt; ++r)
        for (int p = 0; p < P2; ++p)
            if (L.rounds[r][p] != (p ^ (1 << r))) xt = 0;
    xor_tree = xt;
    for (int r = 0; r < LLDA_MAX_ROUNDS; ++r) {
        uint32_t pk = 0;
        for (int p = 0; p < LLDA_MAX_LEAVES; ++p) pk |= (uint32_t)L.rounds[r][p] << (4 * p);
        rounds_pk[r] = pk;
    }
}

}  // namespace

extern "C" {

int llda_abi_version(void) { return LLDA_ABI_VERSION; }

int llda_build_info(void) { return LLDA_BUILD_INFO_BITS; }

int llda_last_hip_error(void) { return g_last_hip_error; }

int llda_struct_size(int which)
{
    switch (which) {
    case 0: return (int)sizeof(llda_layout);
    case 1: return (int)sizeof(llda_sweep_args);
    case 2: return (int)sizeof(llda_batch_args);
    case 3: return (int)sizeof(llda_foldin_args);
    default: return -1;
    }
}

const char *llda_strerror(int code)
{
    switch (code) {
    case LLDA_OK: return "ok";
    case LLDA_E_BAD_K: return "K outside 1..7688, or a wide layout (more than 8 pairwise leaves) handed to an entry point that only takes narrow ones";
    case LLDA_E_BAD_ARG: return "bad argument";
    case LLDA_E_HIP: return "HIP runtime error";
    case LLDA_E_NO_DEVICE: return "no HIP device";
    default: return "unknown error";
    }
}

int llda_layout_init(int32_t K, llda_layout *L)
{
    if (!L) return LLDA_E_BAD_ARG;
    if (K < 1 || K > LLDA_MAX_K) return LLDA_E_BAD_K;
    memset(L, 0, sizeof *L);
    L->K = K;
    add_leaves(L, K, 0);
    if (L->n_leaves > LLDA_MAX_WIDE_LEAVES) return LLDA_E_BAD_K;
    L->wide = L->n_leaves > LLDA_MAX_LEAVES ? 1 : 0;
    int P = 1;
    if (L->wide) P = (L->n_leaves + 7) / 8 * 8;          // 64-lane tiers of one wavefront
    else
        while (P < L->n_leaves) P *= 2;
    L->G = 8 * P;
    L->tiers = L->wide ? P / 8 : 0;
    int t = 0;
    for (int p = 0; p < L->n_leaves; ++p) {
        const int r = (L->leaf_len[p] + 7) / 8;
        if (r > t) t = r;
    }
    if (t > 2) t = (t + 3) / 4 * 4;
    L->T = t;
    L->KP = L->G * t;
    L->tail = L->leaf_len[L->n_leaves - 1] % 8;
    L->tail_row = L->leaf_len[L->n_leaves - 1] / 8;
    for (int i = 0; i < LLDA_MAX_KP; ++i) L->pos_topic[i] = -1;
    // memory position of (lane g, slot s): the 16-byte chunk s >> 2 of all lanes is contiguous (rows with fewer than
    // 4 slots per lane: pos = g*T + s)
    const int w = (t % 4 == 0) ? 4 : t;
    for (int g = 0; g < L->G; ++g)
        for (int s = 0; s < t; ++s) {
            const int pos = ((s / w) * L->G + g) * w + (s % w);
            L->pos_lane[pos] = g;
            L->pos_slot[pos] = s;
        }
    for (int p = 0; p < L->n_leaves; ++p)
        for (int rel = 0; rel < L->leaf_len[p]; ++rel) {
            const int g = 8 * p + (rel & 7), s = rel >> 3;
            const int pos = ((s / w) * L->G + g) * w + (s % w);
            L->topic_pos[L->leaf_start[p] + rel] = pos;
            L->pos_topic[pos] = L->leaf_start[p] + rel;
        }
    {
        int next = 0, n_comb = 0;
        comb_tree(L, K, &next, &n_comb);
    }
    for (int r = 0; r < LLDA_MAX_ROUNDS; ++r)
        for (int p = 0; p < LLDA_MAX_LEAVES; ++p) L->rounds[r][p] = p;
    if (!L->wide) {
        int next = 0, first, count;
        schedule(L, K, &next, &first, &count);
        if (L->n_rounds > LLDA_MAX_ROUNDS) return LLDA_E_BAD_K;
    }
    return LLDA_OK;
}

int64_t llda_sweep_scratch_bytes(int32_t K, int64_t D)
{
    int rc;
    const llda_layout *Lp = layout_of(K, &rc);
    if (rc || !Lp->wide) return 0;
    return (int64_t)wide_blocks(D) * Lp->KP * 8;          // one row of doubles per workgroup of the wide sweep
}

// llda_sweep_args.n_kw_img / img_bits -> KParams.img of a sparse-label launch
static int take_image(const llda_sweep_args *a, KParams &P)
{
    P.img = nullptr;
    if (!a->n_kw_img && !a->img_bits) return LLDA_OK;
    if (!a->n_kw_img || (a->img_bits != 8 && a->img_bits != 16)) return LLDA_E_BAD_ARG;
    if (reinterpret_cast<uintptr_t>(a->n_kw_img) & (a->img_bits == 8 ? 3 : 7)) return LLDA_E_BAD_ARG;     // (as llda_pack_image)
    P.img = a->n_kw_img;
    P.img_col = a->img_col;
    return LLDA_OK;
}

int llda_sweep(const llda_sweep_args *a, void *stream)
{
    if (!a || a->D < 0 || a->V < 1) return LLDA_E_BAD_ARG;
    int rc;
    const llda_layout *Lp = layout_of(a->K, &rc);
    if (rc) return rc;
    const llda_layout &L = *Lp;
    if (a->D == 0) return LLDA_OK;            // an empty shard: nothing to do, array pointers may be NULL
    if (a->n_sites < 0 || a->n_sites >= (1LL << 30)) return LLDA_E_BAD_ARG;   // split the shard (llda_gibbs.h)
    const bool logged = a->csc_pos && a->commit_log;
    if ((a->csc_pos != nullptr) != (a->commit_log != nullptr)) return LLDA_E_BAD_ARG;
    if (!a->doc_off || !a->word || !a->freq || !a->z || !a->lab_mask || !a->n_dk || !a->n_kw ||
        (!a->n_kw_delta && !logged) || !a->n_k || !a->n_k_delta)
        return LLDA_E_BAD_ARG;

    KParams P;
    memset(&P, 0, sizeof P);
    P.doc_off = a->doc_off; P.doc_order = a->doc_order; P.word = a->word; P.freq = a->freq; P.z = a->z;
    P.lab_mask = a->lab_mask; P.n_dk = a->n_dk; P.n_kw = a->n_kw; P.n_kw_delta = a->n_kw_delta;
    P.n_k = a->n_k; P.n_k_delta = a->n_k_delta; P.status = a->status;
    P.csc_pos = a->csc_pos; P.commit_log = a->commit_log;
    P.site_rec = logged ? a->site_rec : nullptr;
    P.D = a->D; P.doc_base = a->doc_base;
    P.alpha = a->alpha; P.beta = a->beta;
    P.vbeta = (double)a->V * a->beta;                       // V * beta evaluated first (LabeledLDA.py:115)
    P.alpha32 = (float)a->alpha; P.beta32 = (float)a->beta; P.vbeta32 = (float)P.vbeta;
    P.key0 = (uint32_t)a->seed; P.key1 = (uint32_t)(a->seed >> 32);
    P.sweep = a->sweep; P.stream_id = a->stream_id;
    fill_schedule(L, P.last_leaf, P.tail, P.tail_row, P.n_rounds, P.xor_tree, P.rounds_pk);
    P.KP = L.KP;
    if (P.site_rec && L.G <= 16 && a->n_sites >= (1LL << 28)) return LLDA_E_BAD_ARG;   // 16-byte records: 32-bit offsets
    const bool has_tail = L.tail != 0;
    // with alpha, beta >= 1e-6 and int32 counts no label-allowed score can underflow to zero ...
    // ... and with V*beta < 2^40 the fp32 / fp64 reciprocals of n_k + V*beta stay in range
    const bool fast = a->alpha >= 1e-6 && a->beta >= 1e-6 && P.vbeta < 1099511627776.0;
    // all-ones label masks and no padded slots: the mask need not be applied at all
    const bool dense = fast && a->dense_mask != 0 && L.K == L.KP;
    // debug_margin: 0 = production margins; n > 0 = 2^-n (wider: more fallbacks); -1 = always exact tier;
    // -2 = no fp32 tier
    P.margin_rel = a->debug_margin == 0 || a->debug_margin == -2 || a->debug_margin == -3 || a->debug_margin == -4 || a->debug_margin == -5 || a->debug_margin == -6 || a->debug_margin == -7 || a->debug_margin == -8 ? 0x1p-40 : (a->debug_margin > 0 ? ldexp(1.0, -a->debug_margin) : 2.0);
    P.margin0_rel = a->debug_margin == 0 || a->debug_margin == -8 ? (float)LLDA_MARGIN0 : (a->debug_margin > 0 && a->debug_margin < 16 ? ldexpf(1.0f, -a->debug_margin) : 2.0f);
    hipStream_t st = (hipStream_t)stream;

    if (L.wide && fast && a->dense_mask == 0 && a->live_off && a->live_pos && a->live_max >= 1 &&
        a->live_max <= LLDA_MAX_LIVE) {
        // sparse label sets on a wide layout: one lane per allowed topic, the wide exact tier for undecided sites
        WSParams W;
        memset(&W, 0, sizeof W);
        static_cast<KParams &>(W) = P;
        W.site_rec = nullptr;
        W.live_off = a->live_off; W.live_pos = a->live_pos;
        if (a->debug_margin < 0) {                          // test hook: every site goes through the exact pipeline
            W.margin_rel = 2.0;
            W.margin0_rel = 2.0f;                           // (also with -8, which keeps the fp32 tier only for the dense 16-bit-row kernel)
        }
        fill_wide(L, W.w);
        const int GS = a->live_max <= 8 ? 8 : a->live_max <= 16 ? 16 : a->live_max <= 32 ? 32 : 64;
        int dpg = a->docs_per_group < 1 ? 1 : a->docs_per_group;
        W.dpg = dpg;
        const int64_t per_block = (int64_t)(256 / GS) * dpg;
        const int64_t blocks = (a->D + per_block - 1) / per_block;
        if (blocks > 0x7fffffffLL) return LLDA_E_BAD_ARG;
        const size_t lds = (size_t)L.KP * 8 + 16;
        const dim3 grid((unsigned)blocks), block(256);
        int rl;
        const int rimg = take_image(a, W);
        if (rimg) return rimg;
#define LLDA_SPARSE_WIDE_I(GS_, IMG_)                                                                       \
        rl = allow_lds(llda_sweep_sparse_kernel<GS_, WSParams, IMG_>, lds);                                 \
        if (rl) return rl;                                                                                  \
        hipLaunchKernelGGL((llda_sweep_sparse_kernel<GS_, WSParams, IMG_>), grid, block, lds, st, W);
#define LLDA_SPARSE_WIDE(GS_)                                                                       \
    case GS_:                                                                                       \
        if (a->img_bits == 8) { LLDA_SPARSE_WIDE_I(GS_, 8) }                                        \
        else if (a->img_bits == 16) { LLDA_SPARSE_WIDE_I(GS_, 16) }                                 \
        else { LLDA_SPARSE_WIDE_I(GS_, 0) }                                                         \
        break;
        switch (GS) { LLDA_SPARSE_WIDE(8) LLDA_SPARSE_WIDE(16) LLDA_SPARSE_WIDE(32) LLDA_SPARSE_WIDE(64) }
#undef LLDA_SPARSE_WIDE
#undef LLDA_SPARSE_WIDE_I
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? LLDA_OK : hip_fail(e);
    }
    if (L.wide) {                                           // more than 8 pairwise leaves: the general path
        if (a->n_kw_img || a->img_bits) return LLDA_E_BAD_ARG;   // the narrow image belongs to the sparse-label kernels
        WParams W;
        memset(&W, 0, sizeof W);
        W.k = P;
        W.k.site_rec = nullptr;
        fill_wide(L, W.w);
        const size_t lds = (size_t)L.KP * 16;               // scores (f64) + n_dk + n_k (int32), per wavefront
        const dim3 grid(wide_blocks(a->D)), block(64);
        int rl = LLDA_OK;
#define LLDA_WIDE_REG(NT_, C_)                                                                               \
    case NT_:                                                                                                \
        rl = allow_lds(llda_sweep_wide_reg_kernel<NT_, C_>, lds_reg);                                        \
        if (rl) return rl;                                                                                   \
        hipLaunchKernelGGL((llda_sweep_wide_reg_kernel<NT_, C_>), grid, block, lds_reg, st, W);              \
        break;
        // counts as start values + int16 changes (no document may then hold 2^15 tokens): 10 instead of 16 bytes of LDS
        // per position
        const bool compact = a->max_doc_tokens > 0 && a->max_doc_tokens < 32768 && a->debug_margin != -4;
        const size_t lds_reg = compact ? (size_t)L.KP * 10 : lds;
        // fp32 tier 0 in front of the fp64 decision (needs the int16 count changes): production
        const bool tier0 = fast && compact && a->debug_margin != -2 && a->debug_margin != -3 && a->debug_margin != -5 &&
                           (a->debug_margin >= -1 || a->debug_margin == -6 || a->debug_margin == -7);
        if (tier0) {
            // margins: production, or the test hook's (n > 0: 2^-n for tier 0 below 16, else tier 0 off; -1: everything exact)
            const float m0 = a->debug_margin == 0 || a->debug_margin == -6 || a->debug_margin == -7 ? LLDA_MARGIN0_WIDE
                             : (a->debug_margin > 0 && a->debug_margin < 16 ? ldexpf(1.0f, -a->debug_margin) : 2.0f);
            // fp32 factors only in LDS (6 bytes per position) when the caller brought the scratch rows of the rare tiers
            // (measured faster only for three and four tiers: see the note at the kernel; -7 forces it for tests and ablations)
            const bool slim = a->scratch && a->scratch_bytes >= llda_sweep_scratch_bytes(a->K, a->D) && a->debug_margin != -6 &&
                              (a->debug_margin == -7 || L.tiers == 3 || L.tiers == 4);
            const size_t lds32 = slim ? (size_t)L.KP * 6 : lds_reg;
            double *scr = slim ? static_cast<double *>(a->scratch) : nullptr;
#define LLDA_WIDE_F32(NT_, TC_)                                                                              \
    if (L.tiers == NT_ && L.T == 4 * TC_) {                                                                  \
        if (slim) {                                                                                          \
            rl = allow_lds(llda_sweep_wide_f32_kernel<NT_, TC_, true>, lds32);                               \
            if (rl) return rl;                                                                               \
            hipLaunchKernelGGL((llda_sweep_wide_f32_kernel<NT_, TC_, true>), grid, block, lds32, st, W, m0, scr);  \
        } else {                                                                                             \
            rl = allow_lds(llda_sweep_wide_f32_kernel<NT_, TC_, false>, lds32);                              \
            if (rl) return rl;                                                                               \
            hipLaunchKernelGGL((llda_sweep_wide_f32_kernel<NT_, TC_, false>), grid, block, lds32, st, W, m0, scr); \
        }                                                                                                    \
    } else
            LLDA_WIDE_F32(2, 3) LLDA_WIDE_F32(2, 4) LLDA_WIDE_F32(3, 4) LLDA_WIDE_F32(4, 3) LLDA_WIDE_F32(4, 4) LLDA_WIDE_F32(5, 4)
            LLDA_WIDE_F32(6, 4) LLDA_WIDE_F32(7, 4) LLDA_WIDE_F32(8, 3) LLDA_WIDE_F32(8, 4)
                return LLDA_E_BAD_K;
#undef LLDA_WIDE_F32
        } else if (fast && a->debug_margin != -3 && compact) {     // the tiered kernel with the row in registers
            switch (L.tiers) {
                LLDA_WIDE_REG(2, true) LLDA_WIDE_REG(3, true) LLDA_WIDE_REG(4, true) LLDA_WIDE_REG(5, true)
                LLDA_WIDE_REG(6, true) LLDA_WIDE_REG(7, true) LLDA_WIDE_REG(8, true)
            default: return LLDA_E_BAD_K;
            }
        } else if (fast && a->debug_margin != -3) {
            switch (L.tiers) {
                LLDA_WIDE_REG(2, false) LLDA_WIDE_REG(3, false) LLDA_WIDE_REG(4, false) LLDA_WIDE_REG(5, false)
                LLDA_WIDE_REG(6, false) LLDA_WIDE_REG(7, false) LLDA_WIDE_REG(8, false)
            default: return LLDA_E_BAD_K;
            }
        } else if (fast) {                                  // (debug_margin -3: the same decision on the LDS-only kernel)
            rl = allow_lds(llda_sweep_wide_kernel<true>, lds);
            if (rl) return rl;
            hipLaunchKernelGGL(llda_sweep_wide_kernel<true>, grid, block, lds, st, W);
        } else {                                            // tiny priors: every site through the exact pipeline
            rl = allow_lds(llda_sweep_wide_kernel<false>, lds);
            if (rl) return rl;
            hipLaunchKernelGGL(llda_sweep_wide_kernel<false>, grid, block, lds, st, W);
        }
#undef LLDA_WIDE_REG
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? LLDA_OK : hip_fail(e);
    }

    // sparse label sets: one lane per allowed topic (a site the margin cannot decide is resolved inside the kernel by
    // the exact pipeline, exact_site_wave)
    const bool sparse = fast && !dense && a->live_off && a->live_pos && a->live_max >= 1 && a->live_max <= LLDA_MAX_LIVE;
    const int G = sparse ? (a->live_max <= 8 ? 8 : a->live_max <= 16 ? 16 : a->live_max <= 32 ? 32 : 64) : L.G;
    const int gpb = 256 / G;
    int dpg = a->docs_per_group;
    // auto: one pass of documents per workgroup.  Workgroups of equal-length documents finish in lock step, so
    // the tail of the launch idles for up to one workgroup's run time: the shorter the workgroup the better
    // (synth2: 3540 M sites/s at 1, 3341 at 4, 3205 at 6 documents per lane group)
    if (dpg < 1) dpg = 1;
    P.dpg = dpg;
    const int64_t per_block = (int64_t)gpb * dpg;
    const int64_t blocks = (a->D + per_block - 1) / per_block;
    if (blocks > 0x7fffffffLL) return LLDA_E_BAD_ARG;

    if (sparse) {
        P.live_off = a->live_off; P.live_pos = a->live_pos;
        if (a->debug_margin < 0) {                         // test hook: every site goes through the exact pipeline
            P.margin_rel = 2.0;
            P.margin0_rel = 2.0f;                          // (also with -8, which keeps the fp32 tier only for the dense 16-bit-row kernel)
        }
        const dim3 grid((unsigned)blocks), block(256);
        const int rimg = take_image(a, P);
        if (rimg) return rimg;
#define LLDA_SPARSE(GS_)                                                                                              \
    case GS_:                                                                                                         \
        if (a->img_bits == 8) hipLaunchKernelGGL((llda_sweep_sparse_kernel<GS_, KParams, 8>), grid, block, 0, st, P);       \
        else if (a->img_bits == 16) hipLaunchKernelGGL((llda_sweep_sparse_kernel<GS_, KParams, 16>), grid, block, 0, st, P); \
        else hipLaunchKernelGGL((llda_sweep_sparse_kernel<GS_, KParams, 0>), grid, block, 0, st, P);                       \
        break;
        switch (G) { LLDA_SPARSE(8) LLDA_SPARSE(16) LLDA_SPARSE(32) LLDA_SPARSE(64) }
#undef LLDA_SPARSE
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? LLDA_OK : hip_fail(e);
    }
    if (a->n_kw_img || a->img_bits) return LLDA_E_BAD_ARG;         // the narrow image belongs to the sparse-label kernels
    if (a->row16) {
        // four / eight / sixteen documents per wavefront (kernel_quad.hpp): K = 512 / 256 / 128 dense with the commit log, every row in
        // the 16-bit image, flags per word from llda_pack_rows16_all, documents below 2^16 tokens
        if (!a->n_kw16 || a->site_row) return LLDA_E_BAD_ARG;
        if (!(fast && a->dense_mask != 0 && logged && llda_quad_ok(a->K))) return LLDA_E_BAD_ARG;
        if (a->D >= (1LL << 31)) return LLDA_E_BAD_ARG;
        P.quad_pad = L.K != L.KP;
        if (P.quad_pad && !P.lab_mask) return LLDA_E_BAD_ARG;
        if (!(a->max_doc_tokens > 0 && a->max_doc_tokens < 65536)) return LLDA_E_BAD_ARG;
        if ((reinterpret_cast<uintptr_t>(a->n_kw16) | reinterpret_cast<uintptr_t>(a->n_kw)) & 15) return LLDA_E_BAD_ARG;
        if (a->V >= (1LL << 22)) return LLDA_E_BAD_ARG;                  // (the image is addressed with 32-bit byte offsets)
        if (L.G <= 16 && !P.site_rec) return LLDA_E_BAD_ARG;             // (8 / 16 documents per wavefront read 16-byte site records)
        if (a->n_sites < 1) return LLDA_OK;                              // (documents without sites: nothing to sample)
        P.n_kw16 = a->n_kw16;
        P.row16 = a->row16;
        if (a->debug_margin == 0) {                                       // (this kernel's own, data-dependent bound: kernel_quad.hpp)
            P.margin0_rel = 0.0f;
            P.margin0_data = 1.0f;
        } else if (a->debug_margin == -9) {                               // test hook: the constant margin 104 * 2^-24 of the total
            P.margin0_rel = LLDA_MARGIN0_QUAD;
            P.margin_rel = 0x1p-40;
        } else if (a->debug_margin <= -10 && a->debug_margin >= -18) {    // test hooks: the data-dependent margin SCALED DOWN -- by 1 / 1.05
            P.margin0_rel = 0.0f;                                         // (the derived bound itself), 1/2, 1/4 ... 1/256: how much of the
            P.margin0_data = a->debug_margin == -10 ? 1.0f / 1.05f : ldexpf(1.0f, a->debug_margin + 10);   // margin the worst site needs
            P.margin_rel = 0x1p-40;
        }
        const int64_t per_q = (int64_t)(2 * QNT / L.G) * dpg;             // (a document is G / 2 lanes)
        const int64_t qblocks = (a->D + per_q - 1) / per_q;
        if (qblocks > 0x7fffffffLL) return LLDA_E_BAD_ARG;
        // (K = 512 with the site records measured SLOWER: 5.19 vs 4.84 ms on 125 000 documents -- four documents per wavefront are not
        // bound by the address pipeline, and the records are 4 more bytes per site)
        const dim3 qgrid((unsigned)qblocks), qblock(QNT);
        if (!P.quad_pad) {
            if (L.G == 32) hipLaunchKernelGGL((llda_sweep_quad_kernel<4>), qgrid, qblock, 0, st, P);
            else if (L.G == 16) hipLaunchKernelGGL((llda_sweep_quad_kernel<3>), qgrid, qblock, 0, st, P);
            else hipLaunchKernelGGL((llda_sweep_quad_kernel<2>), qgrid, qblock, 0, st, P);
        } else {                                                          // K < KP: positions without a topic
            if (L.G == 32) hipLaunchKernelGGL((llda_sweep_quad_kernel<4, false, true>), qgrid, qblock, 0, st, P);
            else if (L.G == 16) hipLaunchKernelGGL((llda_sweep_quad_kernel<3, true, true>), qgrid, qblock, 0, st, P);
            else hipLaunchKernelGGL((llda_sweep_quad_kernel<2, true, true>), qgrid, qblock, 0, st, P);
        }
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? LLDA_OK : hip_fail(e);
    }
    if ((a->n_kw16 != nullptr) != (a->site_row != nullptr)) return LLDA_E_BAD_ARG;
    if (a->n_kw16) {
        // 16-bit rows (bit 31 of csc_pos): the dense 16-slot kernel with the commit log, nothing else knows the flag
        if (!(fast && dense && logged && L.T == 16 && L.G >= 32)) return LLDA_E_BAD_ARG;
        if ((reinterpret_cast<uintptr_t>(a->n_kw16) | reinterpret_cast<uintptr_t>(a->n_kw)) & 15) return LLDA_E_BAD_ARG;
        P.n_kw16 = a->n_kw16;
        P.site_row = a->site_row;
        // four waves per SIMD: n_dk and its sweep-start value share an LDS word (debug_margin -8: the three-wave form regardless)
        P.w4 = a->max_doc_tokens > 0 && a->max_doc_tokens < 65536 && a->debug_margin != -8;
    }
    switch (L.G) {
    case 8: return dispatch_sweep_T<8>(L.T, P, has_tail, fast, dense, blocks, st);
    case 16: return dispatch_sweep_T<16>(L.T, P, has_tail, fast, dense, blocks, st);
    case 32: return dispatch_sweep_T<32>(L.T, P, has_tail, fast, dense, blocks, st);
    case 64: return dispatch_sweep_T<64>(L.T, P, has_tail, fast, dense, blocks, st);
    }
    return LLDA_E_BAD_K;
}

int llda_sweep_batch(const llda_batch_args *a, void *stream)
{
    if (!a || a->n_inst < 0 || a->V < 1) return LLDA_E_BAD_ARG;
    if (a->n_inst == 0) return LLDA_OK;
    if (!a->inst_off || !a->order || !a->word || !a->freq || !a->z || !a->inst_prob || !a->inst_doc || !a->live_off ||
        !a->live_pos || !a->ndk_off || !a->n_dk || !a->kw_off || !a->nk_off || !a->kp || !a->prob_stream || !a->k || !a->counts ||
        !a->delta)
        return LLDA_E_BAD_ARG;
    if (a->lanes != 8 && a->lanes != 16 && a->lanes != 32 && a->lanes != 64) return LLDA_E_BAD_ARG;
    BParams P;
    memset(&P, 0, sizeof P);
    P.inst_off = a->inst_off; P.order = a->order; P.n_inst = a->n_inst; P.word = a->word; P.freq = a->freq; P.z = a->z;
    P.inst_prob = a->inst_prob; P.inst_doc = a->inst_doc; P.live_off = a->live_off; P.live_pos = a->live_pos;
    P.ndk_off = a->ndk_off; P.n_dk = a->n_dk; P.kw_off = a->kw_off; P.nk_off = a->nk_off; P.kp = a->kp;
    P.prob_stream = a->prob_stream; P.k = a->k;
    P.counts = a->counts; P.delta = a->delta; P.status = a->status;
    P.alpha = a->alpha; P.beta = a->beta; P.vbeta = (double)a->V * a->beta;
    // the sparse arithmetic needs strictly positive scores and an in-range reciprocal (as llda_sweep's tiered kernels)
    if (!(a->alpha >= 1e-6 && a->beta >= 1e-6 && P.vbeta < 1099511627776.0)) return LLDA_E_BAD_ARG;
    P.margin_rel = a->debug_margin == 0 ? 0x1p-40 : (a->debug_margin > 0 ? ldexp(1.0, -a->debug_margin) : 2.0);
    P.key0 = (uint32_t)a->seed; P.key1 = (uint32_t)(a->seed >> 32); P.sweep = a->sweep;
    const int gpb = 256 / a->lanes;
    const int64_t blocks = (a->n_inst + gpb - 1) / gpb;
    if (blocks > 0x7fffffffLL) return LLDA_E_BAD_ARG;
    const dim3 grid((unsigned)blocks), block(256);
    hipStream_t st = (hipStream_t)stream;
    switch (a->lanes) {
    case 8: hipLaunchKernelGGL(llda_sweep_batch_kernel<8>, grid, block, 0, st, P); break;
    case 16: hipLaunchKernelGGL(llda_sweep_batch_kernel<16>, grid, block, 0, st, P); break;
    case 32: hipLaunchKernelGGL(llda_sweep_batch_kernel<32>, grid, block, 0, st, P); break;
    default: hipLaunchKernelGGL(llda_sweep_batch_kernel<64>, grid, block, 0, st, P); break;
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LLDA_OK : hip_fail(e);
}

int llda_commit_log(const int64_t *item_begin, const int32_t *item_len, const int32_t *item_word, int64_t n_items,
                    const uint32_t *commit_log, const int32_t *freq_csc, int32_t K, const int64_t *row_off,
                    int32_t *target, int32_t *n_k, int32_t *n_k_delta, void *stream)
{
    if (n_items < 0 || (n_k != nullptr) != (n_k_delta != nullptr)) return LLDA_E_BAD_ARG;
    int rc;
    const llda_layout *Lp = layout_of(K, &rc);
    if (rc) return rc;
    const llda_layout &L = *Lp;
    if (n_items > 0 && (!item_begin || !item_len || !item_word || !commit_log || !freq_csc || !target)) return LLDA_E_BAD_ARG;
    if (n_items == 0 && !n_k) return LLDA_OK;
    CParams P;
    P.item_begin = item_begin; P.item_len = item_len; P.item_word = item_word; P.n_items = n_items;
    P.log = commit_log; P.freq = freq_csc; P.row_off = row_off; P.target = target; P.n_k = n_k; P.n_k_delta = n_k_delta;
    P.KP = L.KP;
    int64_t blocks = (n_items + 3) / 4;
    if (blocks < 1) blocks = 1;
    if (blocks > 0x7fffffffLL) return LLDA_E_BAD_ARG;
    const int rl = allow_lds(llda_commit_log_kernel, 4 * L.KP * sizeof(int));      // (above 64 KB from KP = 4 096 on)
    if (rl) return rl;
    hipLaunchKernelGGL(llda_commit_log_kernel, dim3((unsigned)blocks), dim3(256), 4 * L.KP * sizeof(int), (hipStream_t)stream, P);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LLDA_OK : hip_fail(e);
}

int llda_apply_rows(const int64_t *row_off, int32_t *rows, int64_t n_rows, int32_t K, int32_t *counts, void *stream)
{
    if (n_rows < 0) return LLDA_E_BAD_ARG;
    int rc;
    const llda_layout *Lp = layout_of(K, &rc);
    if (rc) return rc;
    const llda_layout &L = *Lp;
    if (n_rows == 0) return LLDA_OK;
    if (!row_off || !rows || !counts) return LLDA_E_BAD_ARG;
    const int64_t blocks = (n_rows + 3) / 4;
    if (blocks > 0x7fffffffLL) return LLDA_E_BAD_ARG;
    hipLaunchKernelGGL(llda_apply_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, row_off, rows,
                       n_rows, L.KP, counts);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LLDA_OK : hip_fail(e);
}

int llda_rows16_ok(int32_t K)
{
    int rc;
    const llda_layout *Lp = layout_of(K, &rc);
    return !rc && !Lp->wide && Lp->T == 16 && Lp->G >= 32 && Lp->K == Lp->KP ? 1 : 0;
}

int llda_quad_ok(int32_t K)
{
    int rc;
    const llda_layout *Lp = layout_of(K, &rc);
    // (every lane group holds a leaf: the branch-free count update rewrites slot 0 of the lanes that own neither topic of a site
    // with the factor of ITS counts, which must then be a position with a topic -- K = 250, three leaves in a four-leaf layout, is out)
    return !rc && !Lp->wide && Lp->T == 16 && (Lp->G == 8 || Lp->G == 16 || Lp->G == 32) && Lp->n_leaves * 8 == Lp->G ? 1 : 0;
}

int llda_pack_rows16(const int32_t *n_kw, const uint8_t *row16, int64_t V, int32_t K, uint16_t *n_kw16, int32_t *status,
                     void *stream)
{
    if (V < 0) return LLDA_E_BAD_ARG;
    if (!llda_rows16_ok(K)) return LLDA_E_BAD_K;
    if (V == 0) return LLDA_OK;
    if (!n_kw || !row16 || !n_kw16) return LLDA_E_BAD_ARG;
    int rc;
    const llda_layout &L = *layout_of(K, &rc);
    const int64_t blocks = (V * 2 * L.G + 255) / 256;
    if (blocks > 0x7fffffffLL) return LLDA_E_BAD_ARG;
    hipLaunchKernelGGL(llda_pack_rows16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n_kw, row16, n_kw16,
                       V, L.G, status);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LLDA_OK : hip_fail(e);
}

int llda_pack_image_cols(const int32_t *n_kw, int64_t V, int32_t K, int32_t bits, const int32_t *col_src, void *img, void *stream)
{
    if (V < 0 || (bits != 8 && bits != 16)) return LLDA_E_BAD_ARG;
    int rc;
    const llda_layout *Lp = layout_of(K, &rc);
    if (rc) return rc;
    if (V == 0) return LLDA_OK;
    if (!n_kw || !img || !col_src || (Lp->KP & 3)) return LLDA_E_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(n_kw) & 15) || (reinterpret_cast<uintptr_t>(img) & (bits == 8 ? 3 : 7))) return LLDA_E_BAD_ARG;
    if (reinterpret_cast<uintptr_t>(col_src) & 15) return LLDA_E_BAD_ARG;
    int64_t blocks = V < 256 * 32 ? V : 256 * 32;
    const dim3 grid((unsigned)blocks), block(256);
    const size_t lds = (size_t)Lp->KP * sizeof(int);
    if (bits == 8) hipLaunchKernelGGL(llda_pack_image_cols_kernel<8>, grid, block, lds, (hipStream_t)stream, n_kw, col_src, img, V, Lp->KP);
    else hipLaunchKernelGGL(llda_pack_image_cols_kernel<16>, grid, block, lds, (hipStream_t)stream, n_kw, col_src, img, V, Lp->KP);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LLDA_OK : hip_fail(e);
}

int llda_pack_rows16_all(const int32_t *n_kw, int64_t V, int32_t K, uint16_t *n_kw16, uint8_t *row16, void *stream)
{
    if (V < 0) return LLDA_E_BAD_ARG;
    int rc;
    const llda_layout *Lp = layout_of(K, &rc);
    if (rc) return rc;
    if (!llda_quad_ok(K)) return LLDA_E_BAD_K;
    if (V == 0) return LLDA_OK;
    if (!n_kw || !row16 || !n_kw16) return LLDA_E_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(n_kw16) | reinterpret_cast<uintptr_t>(n_kw)) & 15) return LLDA_E_BAD_ARG;
    const int64_t blocks = (V * 2 * Lp->G + 255) / 256;
    if (blocks > 0x7fffffffLL) return LLDA_E_BAD_ARG;
    const dim3 grid((unsigned)blocks), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (Lp->G == 32) hipLaunchKernelGGL(llda_pack_rows16_all_kernel<32>, grid, block, 0, st, n_kw, n_kw16, row16, V);
    else if (Lp->G == 16) hipLaunchKernelGGL(llda_pack_rows16_all_kernel<16>, grid, block, 0, st, n_kw, n_kw16, row16, V);
    else hipLaunchKernelGGL(llda_pack_rows16_all_kernel<8>, grid, block, 0, st, n_kw, n_kw16, row16, V);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LLDA_OK : hip_fail(e);
}

int llda_pack_image(const int32_t *n_kw, int64_t n, int32_t bits, void *img, void *stream)
{
    if (n < 0 || (n & 3) || (bits != 8 && bits != 16)) return LLDA_E_BAD_ARG;
    if (n == 0) return LLDA_OK;
    if (!n_kw || !img) return LLDA_E_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(n_kw) & 15) || (reinterpret_cast<uintptr_t>(img) & (bits == 8 ? 3 : 7))) return LLDA_E_BAD_ARG;
    const int64_t n4 = n / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    const dim3 grid((unsigned)blocks), block(256);
    if (bits == 8)
        hipLaunchKernelGGL(llda_pack_image_kernel<8>, grid, block, 0, (hipStream_t)stream, reinterpret_cast<const int4 *>(n_kw), img, n4);
    else
        hipLaunchKernelGGL(llda_pack_image_kernel<16>, grid, block, 0, (hipStream_t)stream, reinterpret_cast<const int4 *>(n_kw), img, n4);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LLDA_OK : hip_fail(e);
}

int llda_apply_delta(int32_t *counts, int32_t *delta, int64_t n, void *stream)
{
    if (!counts || !delta || n < 0) return LLDA_E_BAD_ARG;
    if (n == 0) return LLDA_OK;
    const bool aligned = ((reinterpret_cast<uintptr_t>(counts) | reinterpret_cast<uintptr_t>(delta)) & 15) == 0;
    const int64_t n4 = aligned ? n / 4 : 0;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(llda_apply_delta_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       counts, delta, n4, n);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LLDA_OK : hip_fail(e);
}

int llda_count_init(const int64_t *doc_off, const int32_t *word, const int32_t *freq, const int32_t *z,
                    int64_t D, int32_t K, int32_t *n_dk, int32_t *n_kw, int32_t *n_k, void *stream)
{
    if (D < 0) return LLDA_E_BAD_ARG;
    int rc;
    const llda_layout *Lp = layout_of(K, &rc);
    if (rc) return rc;
    const llda_layout &L = *Lp;
    if (D == 0) return LLDA_OK;
    if (!doc_off || !word || !freq || !z || !n_dk || !n_kw || !n_k) return LLDA_E_BAD_ARG;
    // rows too long for five LDS histograms per workgroup (wide layouts): one wavefront per workgroup, two histograms
    const int waves = (size_t)5 * L.KP * sizeof(int) <= 48 * 1024 ? 4 : 1;
    const size_t lds = (size_t)(1 + waves) * L.KP * sizeof(int);
    const int rl = allow_lds(llda_count_init_kernel, lds);
    if (rl) return rl;
    int64_t blocks = (D + waves - 1) / waves;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(llda_count_init_kernel, dim3((unsigned)blocks), dim3(64 * waves), lds,
                       (hipStream_t)stream, doc_off, word, freq, z, D, L.KP, n_dk, n_kw, n_k);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LLDA_OK : hip_fail(e);
}

int llda_loglik(const int64_t *doc_off, const int32_t *word, const uint16_t *lab_mask, const int32_t *n_dk,
                const int32_t *n_kw, const int32_t *n_k, int64_t D, int64_t V, int32_t K, double alpha,
                double beta, double *out_doc, void *stream)
{
    if (D < 0 || V < 1) return LLDA_E_BAD_ARG;
    int rc;
    const llda_layout *Lp = layout_of(K, &rc);
    if (rc) return rc;
    const llda_layout &L = *Lp;
    if (D == 0) return LLDA_OK;
    if (!doc_off || !word || !lab_mask || !n_dk || !n_kw || !n_k || !out_doc) return LLDA_E_BAD_ARG;
    LParams P;
    P.doc_off = doc_off; P.word = word; P.lab_mask = lab_mask; P.n_dk = n_dk; P.n_kw = n_kw; P.n_k = n_k;
    P.out_doc = out_doc; P.D = D; P.alpha = alpha; P.beta = beta; P.vbeta = (double)V * beta;
    hipStream_t st = (hipStream_t)stream;
    if (L.wide) {
        WLParams W;
        W.l = P; W.NT = L.tiers; W.T = L.T; W.G = L.G; W.KP = L.KP;
        const size_t lds = (size_t)L.KP * 16;
        const int rl = allow_lds(llda_loglik_wide_kernel, lds);
        if (rl) return rl;
        hipLaunchKernelGGL(llda_loglik_wide_kernel, dim3(wide_blocks(D)), dim3(64), lds, st, W);
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? LLDA_OK : hip_fail(e);
    }
    switch (L.G) {
    case 8: return dispatch_loglik_T<8>(L.T, P, st);
    case 16: return dispatch_loglik_T<16>(L.T, P, st);
    case 32: return dispatch_loglik_T<32>(L.T, P, st);
    case 64: return dispatch_loglik_T<64>(L.T, P, st);
    }
    return LLDA_E_BAD_K;
}

static void readout_layout(const llda_layout &L, RParams &P)
{
    P.K = L.K; P.KP = L.KP; P.T = L.T;
    for (int p = 0; p < LLDA_MAX_WIDE_LEAVES; ++p) { P.leaf_start[p] = L.leaf_start[p]; P.leaf_len[p] = L.leaf_len[p]; }
    fill_schedule(L, P.last_leaf, P.tail, P.tail_row, P.n_rounds, P.xor_tree, P.rounds_pk);
}

int llda_readout_phi(const int32_t *n_kw, const int32_t *n_k, const double *den, int64_t V, int32_t K, double beta,
                     int32_t mode, double keep, double share, double *out, int32_t *flags, void *stream)
{
    if (V < 1 || (mode != 0 && mode != 1) || !n_kw || (!n_k && !den) || !out) return LLDA_E_BAD_ARG;
    int rc;
    const llda_layout *Lp = layout_of(K, &rc);
    if (rc) return rc;
    const llda_layout &L = *Lp;
    RParams P;
    memset(&P, 0, sizeof P);
    readout_layout(L, P);
    P.n_kw = n_kw; P.n_k = n_k; P.den = den; P.out = out; P.flags = flags; P.V = V; P.mode = mode;
    P.beta = beta; P.vbeta = (double)V * beta; P.keep = keep; P.share = share;
    hipLaunchKernelGGL(llda_readout_phi_kernel, dim3((unsigned)((V + 63) / 64)), dim3(256), 0, (hipStream_t)stream, P);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LLDA_OK : hip_fail(e);
}

int llda_readout_theta(const int32_t *n_dk, const uint16_t *lab_mask, int64_t D, int32_t K, double alpha, int32_t mode,
                       double keep, double share, double *out, void *stream)
{
    if (D < 0 || (mode != 0 && mode != 1)) return LLDA_E_BAD_ARG;
    int rc;
    const llda_layout *Lp = layout_of(K, &rc);
    if (rc) return rc;
    const llda_layout &L = *Lp;
    if (D == 0) return LLDA_OK;
    if (!n_dk || !lab_mask || !out) return LLDA_E_BAD_ARG;
    if (L.wide) {
        WRParams W;
        memset(&W, 0, sizeof W);
        W.n_dk = n_dk; W.lab_mask = lab_mask; W.out = out; W.D = D; W.K = L.K; W.mode = mode; W.alpha = alpha;
        W.keep = keep; W.share = share;
        for (int p = 0; p < LLDA_MAX_WIDE_LEAVES; ++p) { W.leaf_start[p] = L.leaf_start[p]; W.leaf_len[p] = L.leaf_len[p]; }
        fill_wide(L, W.w);
        const size_t lds = (size_t)L.KP * 8;
        const int rl = allow_lds(llda_readout_theta_wide_kernel, lds);
        if (rl) return rl;
        hipLaunchKernelGGL(llda_readout_theta_wide_kernel, dim3(wide_blocks(D)), dim3(64), lds, (hipStream_t)stream, W);
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? LLDA_OK : hip_fail(e);
    }
    RParams P;
    memset(&P, 0, sizeof P);
    readout_layout(L, P);
    P.n_dk = n_dk; P.lab_mask = lab_mask; P.out = out; P.D = D; P.mode = mode; P.alpha = alpha;
    P.keep = keep; P.share = share;
    hipStream_t st = (hipStream_t)stream;
    const bool has_tail = L.tail != 0;
    switch (L.G) {
    case 8: return dispatch_theta_T<8>(L.T, P, has_tail, st);
    case 16: return dispatch_theta_T<16>(L.T, P, has_tail, st);
    case 32: return dispatch_theta_T<32>(L.T, P, has_tail, st);
    case 64: return dispatch_theta_T<64>(L.T, P, has_tail, st);
    }
    return LLDA_E_BAD_K;
}

int llda_foldin(const llda_foldin_args *a, void *stream)
{
    if (!a || !a->doc_off || !a->word || !a->init_idx || !a->freq || !a->ph || !a->init_rows || !a->z || !a->n_dk ||
        !a->th || !a->slot_valid || a->D < 0 || a->iters < 0 || a->thinning < 1)
        return LLDA_E_BAD_ARG;
    int rc;
    const llda_layout *Lp = layout_of(a->K, &rc);
    if (rc) return rc;
    const llda_layout &L = *Lp;
    if (a->D == 0) return LLDA_OK;
    FParams P;
    memset(&P, 0, sizeof P);
    P.doc_off = a->doc_off; P.word = a->word; P.init_idx = a->init_idx; P.freq = a->freq; P.z = a->z;
    P.ph = a->ph; P.phn = a->init_rows; P.n_dk = a->n_dk; P.th = a->th; P.slot_valid = a->slot_valid;
    P.status = a->status; P.D = a->D; P.doc_base = a->doc_base; P.doc_ids = a->doc_ids; P.alpha = a->alpha; P.beta = a->beta;
    P.c_init = a->c_init; P.c_loop = a->c_loop;
    P.key0 = (uint32_t)a->seed; P.key1 = (uint32_t)(a->seed >> 32); P.stream_id = a->stream_id;
    P.iters = a->iters; P.thinning = a->thinning; P.beta_fallback = a->beta_fallback; P.avg_mode = a->avg_mode;
    // the decided tier needs non-negative scores and a divisor c whose distance from 1 dwarfs the rounding of a normalised sum
    P.exact_only = (a->exact_only != 0 || !(a->alpha >= 0.0) || !(a->c_loop - 1.0 >= 1e-9)) ? 1 : 0;
    P.n_sites = a->n_sites > 0 ? a->n_sites : 0;
    P.ph_base = a->ph_base; P.doc_stream = a->doc_stream;
    fill_schedule(L, P.last_leaf, P.tail, P.tail_row, P.n_rounds, P.xor_tree, P.rounds_pk);
    hipStream_t st = (hipStream_t)stream;
    const bool has_tail = L.tail != 0;
    if (L.wide) {
        WFParams W;
        memset(&W, 0, sizeof W);
        W.f = P;
        fill_wide(L, W.w);
        // (the wide path always draws the initial assignments with one wavefront per site; n_sites = 0: no site at all)
        int rl = allow_lds(llda_foldin_init_wide_kernel, (size_t)L.KP * 8);
        if (rl) return rl;
        if (P.n_sites > 0)
            hipLaunchKernelGGL(llda_foldin_init_wide_kernel, dim3(wide_blocks(P.n_sites)), dim3(64), (size_t)L.KP * 8, st, W);
        rl = allow_lds(llda_foldin_wide_kernel, (size_t)L.KP * 12);
        if (rl) return rl;
        hipLaunchKernelGGL(llda_foldin_wide_kernel, dim3(wide_blocks(P.D)), dim3(64), (size_t)L.KP * 12, st, W);
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? LLDA_OK : hip_fail(e);
    }
    switch (L.G) {
    case 8: return dispatch_foldin_T<8>(L.T, P, has_tail, st);
    case 16: return dispatch_foldin_T<16>(L.T, P, has_tail, st);
    case 32: return dispatch_foldin_T<32>(L.T, P, has_tail, st);
    case 64: return dispatch_foldin_T<64>(L.T, P, has_tail, st);
    }
    return LLDA_E_BAD_K;
}

int llda_selftest_div(uint64_t seed, int64_t n, unsigned long long *mismatches_dev, void *stream)
{
    if (!mismatches_dev || n < 1) return LLDA_E_BAD_ARG;
    const int iters = 1024;
    int64_t threads = (n + iters - 1) / iters;
    int64_t blocks = (threads + 255) / 256;
    if (blocks > 0x7fffffffLL) return LLDA_E_BAD_ARG;
    hipLaunchKernelGGL(llda_selftest_div_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       seed, iters, mismatches_dev);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LLDA_OK : hip_fail(e);
}

}  // extern "C"
