/* CPU oracle (C restatement) of the collapsed-Gibbs sweep  --  TEST INFRASTRUCTURE ONLY.
 *
 * Checker for the HIP path and the "port" CPU baseline of bench.py.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (lda_thesis_amd) never does.  Parity status: PINNED -- tests/test_oracle_golden.py checks this
 * file bit for bit against tests/golden/ (vectors produced by the unmodified reference, see
 * oracle/gen_golden.py).
 *
 * Restated here (all paths in /root/reference):
 *   sweep body            LabeledLDA.py:101-125 == CascadeLDA.py:397-421
 *   fp64 score DAG        LabeledLDA.py:113-118:  a = n_dk + alpha; num = n_kv[:,v] + beta;
 *                         den = n_zk + V*beta; prob = lab*a*(num/den); prob /= np.sum(prob)
 *   np.sum                numpy 2.2.6 pairwise sum (third party, not vendored; algorithm:
 *                         numpy/_core/src/umath/loops_utils.h.src  @TYPE@_pairwise_sum)
 *   draw                  the module-global multinom_draw (LabeledLDA.py:4,119) is replaced by the
 *                         keyed draw of oracle/llda_oracle.py (draw_keyed / keyed_uniform); this
 *                         file restates that draw.
 * State uses the reference's layout and dtypes: n_d_k (D,K) int64, n_k_v (K,V) int64 row-major
 * (topic-major, so the per-site column read is strided exactly as in the reference), n_zk (K) int64.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no fast-math).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PW_BLOCK 128
#define MAX_LEAVES 64    /* K <= MAX_K always splits into <= 64 leaves */
#define MAX_K 7688
#define MAX_KP 8192      /* 64 leaves x 128 */

/* ---------------- numpy pairwise sum ---------------- */
static double pairwise_sum(const double *a, int64_t n)
{
    if (n < 8) {
        double res = 0.;
        for (int64_t i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= PW_BLOCK) {
        double r[8], res;
        int64_t i;
        for (i = 0; i < 8; i++) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8) {
            r[0] += a[i + 0]; r[1] += a[i + 1]; r[2] += a[i + 2]; r[3] += a[i + 3];
            r[4] += a[i + 4]; r[5] += a[i + 5]; r[6] += a[i + 6]; r[7] += a[i + 7];
        }
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
    }
}

double llda_oracle_pairwise_sum(const double *a, int64_t n) { return pairwise_sum(a, n); }

/* ---------------- Philox4x32-10 ---------------- */
static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

void llda_oracle_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    philox4x32_10(c, key[0], key[1]);
    memcpy(out, c, sizeof c);
}

static double keyed_uniform(uint64_t seed, uint32_t sweep, uint32_t stream, uint32_t doc, uint32_t site)
{
    uint32_t c[4] = {site >> 1, doc, stream, sweep};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    uint32_t a = (site & 1) ? c[2] : c[0];
    uint32_t b = (site & 1) ? c[3] : c[1];
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

double llda_oracle_uniform(uint64_t seed, uint32_t sweep, uint32_t stream, uint32_t doc, uint32_t site)
{
    return keyed_uniform(seed, sweep, stream, doc, site);
}

/* ---------------- group layout (see oracle/llda_oracle.py Layout) ---------------- */
typedef struct {
    int K, m, P, G, T, KP;
    int leaf_start[MAX_LEAVES], leaf_n[MAX_LEAVES];
    int topic_pos[MAX_KP];   /* topic -> position */
    int pos_topic[MAX_KP];   /* position -> topic, -1 padding */
} layout_t;

static void add_leaves(layout_t *L, int n, int start)
{
    if (n <= PW_BLOCK) {
        if (L->m < MAX_LEAVES) { L->leaf_start[L->m] = start; L->leaf_n[L->m] = n; }
        L->m++;
        return;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    add_leaves(L, n2, start);
    add_leaves(L, n - n2, start + n2);
}

static int make_layout(layout_t *L, int K)
{
    if (K < 1 || K > MAX_K) return -1;
    memset(L, 0, sizeof *L);
    L->K = K;
    add_leaves(L, K, 0);
    if (L->m > MAX_LEAVES) return -1;
    if (L->m > 8) L->P = (L->m + 7) / 8 * 8;        /* wide layouts: 64-lane tiers of one wavefront */
    else {
        L->P = 1;
        while (L->P < L->m) L->P *= 2;
    }
    L->G = 8 * L->P;
    int t = 0;
    for (int p = 0; p < L->m; p++) {
        int r = (L->leaf_n[p] + 7) / 8;
        if (r > t) t = r;
    }
    if (t > 2) t = (t + 3) / 4 * 4;
    L->T = t;
    L->KP = L->G * t;
    for (int i = 0; i < L->KP; i++) L->pos_topic[i] = -1;
    for (int p = 0; p < L->m; p++)
        for (int rel = 0; rel < L->leaf_n[p]; rel++) {
            int pos = (8 * p + (rel & 7)) * t + (rel >> 3);
            L->topic_pos[L->leaf_start[p] + rel] = pos;
            L->pos_topic[pos] = L->leaf_start[p] + rel;
        }
    return 0;
}

/* ---------------- the keyed categorical draw (llda_oracle.py draw_keyed) ---------------- */
static int draw_keyed(const layout_t *L, const double *prob, double u)
{
    double p[MAX_KP], q[MAX_KP], x[8 * MAX_LEAVES], y[8 * MAX_LEAVES];
    const int G = L->G, T = L->T;
    for (int i = 0; i < L->KP; i++) p[i] = 0.0;
    for (int k = 0; k < L->K; k++) p[L->topic_pos[k]] = prob[k];
    for (int g = 0; g < G; g++) {
        q[g * T] = p[g * T];
        for (int s = 1; s < T; s++) q[g * T + s] = q[g * T + s - 1] + p[g * T + s];
        x[g] = q[g * T + T - 1];
    }
    for (int d = 1; d < G; d *= 2) {
        for (int g = 0; g < G; g++) y[g] = (g >= d) ? x[g - d] + x[g] : x[g];
        memcpy(x, y, sizeof(double) * G);
    }
    double t = u * x[G - 1];
    int last_pos = -1;
    for (int g = 0; g < G; g++) {
        double tg = t - (g ? x[g - 1] : 0.0);
        for (int s = 0; s < T; s++) {
            int i = g * T + s;
            if (p[i] > 0.0) {
                if (q[i] > tg) return L->pos_topic[i];
                last_pos = i;
            }
        }
    }
    return last_pos < 0 ? -1 : L->pos_topic[last_pos];
}

int llda_oracle_draw(int K, const double *prob, double u)
{
    layout_t L;
    if (make_layout(&L, K)) return -2;
    return draw_keyed(&L, prob, u);
}

/* ---------------- one document ----------------
 * n_zk_work: the n_zk this document sees (mutated in place; caller decides what it starts as)
 * col_adj:   snapshot mode -> the n_k_v column is read from the sweep-start matrix and only the
 *            site's own -f at z_old is applied (word ids are unique inside a document);
 *            sequential mode -> n_k_v is mutated in place like the reference does. */
static int do_doc(const layout_t *L, int snapshot, int64_t d, int K, int64_t V,
                  const int64_t *doc_off, const int32_t *word, const int32_t *freq, int32_t *z,
                  const uint8_t *labs, int64_t *n_d_k, int64_t *n_k_v, int64_t *n_zk_work,
                  double alpha, double beta, uint64_t seed, uint32_t sweep, uint32_t stream,
                  int64_t doc_base, double *prob)
{
    int64_t *row = n_d_k + d * K;
    const uint8_t *lab = labs + d * K;
    const double vbeta = (double)V * beta;
    for (int64_t i = doc_off[d]; i < doc_off[d + 1]; i++) {
        const int64_t v = word[i];
        const int64_t f = freq[i];
        const int zo = z[i];
        if (!snapshot) n_k_v[(int64_t)zo * V + v] -= f;
        row[zo] -= f;
        n_zk_work[zo] -= f;
        for (int k = 0; k < K; k++) {
            int64_t nkv = n_k_v[(int64_t)k * V + v];
            if (snapshot && k == zo) nkv -= f;
            double a = (double)row[k] + alpha;
            double num_b = (double)nkv + beta;
            double den_b = (double)n_zk_work[k] + vbeta;
            prob[k] = ((double)lab[k] * a) * (num_b / den_b);
        }
        double s = pairwise_sum(prob, K);
        for (int k = 0; k < K; k++) prob[k] /= s;
        double u = keyed_uniform(seed, sweep, stream, (uint32_t)(d + doc_base), (uint32_t)(i - doc_off[d]));
        int zn = draw_keyed(L, prob, u);
        if (zn < 0) return -3;
        z[i] = zn;
        if (!snapshot) n_k_v[(int64_t)zn * V + v] += f;
        row[zn] += f;
        n_zk_work[zn] += f;
    }
    return 0;
}

/* mode 0: sequential keyed (O2).  mode 1: per-document snapshot keyed (O3).
 * threads > 1 (snapshot only): documents are split over OpenMP threads. */
int llda_oracle_sweep(int mode, int64_t D, int K, int64_t V,
                      const int64_t *doc_off, const int32_t *word, const int32_t *freq, int32_t *z,
                      const uint8_t *labs, int64_t *n_d_k, int64_t *n_k_v, int64_t *n_zk,
                      double alpha, double beta, uint64_t seed, uint32_t sweep, uint32_t stream,
                      int64_t doc_base, int threads)
{
    layout_t L;
    if (make_layout(&L, K)) return -1;
    if (mode == 0) {
        double prob[MAX_KP];
        for (int64_t d = 0; d < D; d++) {
            int rc = do_doc(&L, 0, d, K, V, doc_off, word, freq, z, labs, n_d_k, n_k_v, n_zk,
                            alpha, beta, seed, sweep, stream, doc_base, prob);
            if (rc) return rc;
        }
        return 0;
    }
    const int64_t S = doc_off[D];
    int32_t *z_old = (int32_t *)malloc(sizeof(int32_t) * (size_t)(S ? S : 1));
    if (!z_old) return -4;
    memcpy(z_old, z, sizeof(int32_t) * (size_t)S);
    int err = 0;
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        double prob[MAX_KP];
        int64_t *work = (int64_t *)malloc(sizeof(int64_t) * (size_t)K);
#pragma omp for schedule(dynamic, 16)
        for (int64_t d = 0; d < D; d++) {
            memcpy(work, n_zk, sizeof(int64_t) * (size_t)K);
            int rc = do_doc(&L, 1, d, K, V, doc_off, word, freq, z, labs, n_d_k, n_k_v, work,
                            alpha, beta, seed, sweep, stream, doc_base, prob);
            if (rc) {
#pragma omp atomic write
                err = rc;
            }
        }
        free(work);
    }
    /* apply the integer deltas (commutative) */
    for (int64_t i = 0; i < S; i++) {
        const int64_t v = word[i], f = freq[i];
        n_k_v[(int64_t)z_old[i] * V + v] -= f;
        n_k_v[(int64_t)z[i] * V + v] += f;
        n_zk[z_old[i]] -= f;
        n_zk[z[i]] += f;
    }
    free(z_old);
    return err;
}

/* Snapshot sweep (O3) of a SELECTION of documents of a larger corpus -- the checker of the full-size parity
 * tests.  Under snapshot semantics a document depends only on the sweep-start n_k_v / n_zk and on itself, so
 * n_sel documents picked at random can be checked without sweeping the rest: local document d (CSR doc_off /
 * word / freq / z, rows labs[d] and n_d_k[d]) has the global id doc_ids[d] for the RNG key.  z and n_d_k are
 * updated; n_k_v and n_zk are the sweep-start counts and are NOT modified (the deltas of the other documents
 * are not known here).  labs == NULL: every topic allowed. */
int llda_oracle_sweep_docs(int64_t n_sel, const int64_t *doc_ids, int K, int64_t V,
                           const int64_t *doc_off, const int32_t *word, const int32_t *freq, int32_t *z,
                           const uint8_t *labs, int64_t *n_d_k, const int64_t *n_k_v, const int64_t *n_zk,
                           double alpha, double beta, uint64_t seed, uint32_t sweep, uint32_t stream, int threads)
{
    layout_t L;
    if (make_layout(&L, K)) return -1;
    uint8_t *ones = NULL;
    if (!labs) {
        ones = (uint8_t *)malloc((size_t)K);
        if (!ones) return -4;
        memset(ones, 1, (size_t)K);
    }
    int err = 0;
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        double prob[MAX_KP];
        int64_t *work = (int64_t *)malloc(sizeof(int64_t) * (size_t)K);
#pragma omp for schedule(dynamic, 4)
        for (int64_t d = 0; d < n_sel; d++) {
            memcpy(work, n_zk, sizeof(int64_t) * (size_t)K);
            /* do_doc indexes labs by d: hand it a base that makes row d the all-ones row when labs is absent */
            const uint8_t *lab_base = labs ? labs : ones - d * K;
            int rc = do_doc(&L, 1, d, K, V, doc_off, word, freq, z, lab_base, n_d_k, (int64_t *)n_k_v, work,
                            alpha, beta, seed, sweep, stream, doc_ids[d] - d, prob);
            if (rc) {
#pragma omp atomic write
                err = rc;
            }
        }
        free(work);
    }
    free(ones);
    return err;
}

/* ---------------- word-major int32 variant (bench.py's strong CPU leg) ----------------
 * The same per-document snapshot sweep (O3, dense label mask: LocalLDA's case and BASELINE configs[2]/[3]) on the layout the GPU
 * kernels use: n_kw [V][K] int32 WORD-major -- the reference's strided column n_k_v[:, v] (LabeledLDA.py:114) is one contiguous
 * row --, n_d_k [D][K] int32, n_k [K] int32.  Same arithmetic, same association order, same keyed draw: the assignments equal
 * llda_oracle_sweep's (tests/test_oracle_units.py).  Not the reference's layout: reported beside the reference-layout leg. */
int llda_oracle_sweep_wm(int64_t D, int K, int64_t V,
                         const int64_t *doc_off, const int32_t *word, const int32_t *freq, int32_t *z,
                         int32_t *n_d_k, int32_t *n_kw, int32_t *n_k,
                         double alpha, double beta, uint64_t seed, uint32_t sweep, uint32_t stream,
                         int64_t doc_base, int threads)
{
    layout_t L;
    if (make_layout(&L, K)) return -1;
    const int64_t S = doc_off[D];
    int32_t *z_old = (int32_t *)malloc(sizeof(int32_t) * (size_t)(S ? S : 1));
    if (!z_old) return -4;
    memcpy(z_old, z, sizeof(int32_t) * (size_t)S);
    const double vbeta = (double)V * beta;
    int err = 0;
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        double prob[MAX_KP];
        int32_t *nk = (int32_t *)malloc(sizeof(int32_t) * (size_t)K);
#pragma omp for schedule(dynamic, 16)
        for (int64_t d = 0; d < D; d++) {
            memcpy(nk, n_k, sizeof(int32_t) * (size_t)K);
            int32_t *row = n_d_k + d * K;
            for (int64_t i = doc_off[d]; i < doc_off[d + 1]; i++) {
                const int32_t *col = n_kw + (int64_t)word[i] * K;
                const int32_t f = freq[i];
                const int zo = z[i];
                row[zo] -= f;
                nk[zo] -= f;
                for (int k = 0; k < K; k++) {
                    const double a = (double)row[k] + alpha;
                    const double num_b = (double)(col[k] - (k == zo ? f : 0)) + beta;
                    const double den_b = (double)nk[k] + vbeta;
                    prob[k] = (1.0 * a) * (num_b / den_b);
                }
                const double s = pairwise_sum(prob, K);
                for (int k = 0; k < K; k++) prob[k] /= s;
                const double u = keyed_uniform(seed, sweep, stream, (uint32_t)(d + doc_base), (uint32_t)(i - doc_off[d]));
                const int zn = draw_keyed(&L, prob, u);
                if (zn < 0) {
#pragma omp atomic write
                    err = -3;
                    break;
                }
                z[i] = zn;
                row[zn] += f;
                nk[zn] += f;
            }
        }
        free(nk);
    }
    for (int64_t i = 0; i < S; i++) {                                 /* the integer deltas (commutative) */
        const int64_t v = word[i];
        const int32_t f = freq[i];
        n_kw[v * K + z_old[i]] -= f;
        n_kw[v * K + z[i]] += f;
        n_k[z_old[i]] -= f;
        n_k[z[i]] += f;
    }
    free(z_old);
    return err;
}
