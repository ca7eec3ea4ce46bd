/* quad_tier0_model.c -- TEST INFRASTRUCTURE (never linked into the product, never timed).
 *
 * A CPU model of TIER 0 of llda_sweep_quad_kernel (lda_thesis_amd/csrc/kernel_quad.hpp: quad_draw), the fp32 decision in front of the
 * fp64 pipeline of /root/reference/LabeledLDA.py:113-119, in the kernel's own association order, next to the same quantities in 80-bit
 * arithmetic.  It turns the error derivation in the header of kernel_quad.hpp into a checked statement: for EVERY position of EVERY
 * lane the compared difference  q~[k] - (tg~ -+ m)  is measured against the real-number  E*[k] = prefix*[k] - u total*  and reported
 * relative to the kernel's margin m and to the derived bound  v (31.1 L* + 38.2 t* + 36.1 P* + 0.125 total*).
 *
 * The model (fp32, one rounding per operation, fmaf where the kernel has v_fma / v_pk_fma):
 *   factor   pa = fl(fl((float)n_dk + alpha32) * rcp(fl((float)n_k + vbeta32)))           (kernel_sweep.hpp: tier0_factor;
 *            v_rcp_f32 is specified to 1 ulp: rcp_mode 0 = the correctly rounded reciprocal, 1 = that moved by -1 / 0 / +1 ulp at random)
 *   chains   a document is LPD = 2^LB lanes x 32 slots; lane l walks chain A (slots 0..15 of standard lane 2 l) and chain B (standard
 *            lane 2 l + 1):  nb = fl(x + beta32);  Q[0] = fl(nb pa);  Q[a] = fma(nb, pa, Q[a-1]);  a position without a topic has pa = 0
 *   scan     X0 = fl(QA[15] + QB[15]);  X = Hillis-Steele inclusive scan over the lanes (steps 1, 2, 4, 8 below LPD);
 *            total by rotate / xor butterflies (LB = 4: 8, 4, 2, 1;  LB = 3: mirror, 1, 2;  LB = 2: 1, 2) -- the same value in every lane
 *   target   t = fl(u32 total), tg = fl(t - X[l-1]);  m = fma(QM_L, X0, fma(QM_T, t, fma(QM_P, X[l-1], QM_TOT total))) (* margin_data,
 *            + total margin_rel);  lo0 = fl(tg - m), hi0 = fl(tg + m);  chain B is compared against fl(lo0 - QA[15]), fl(hi0 - QA[15])
 *   decision per lane: c0 = QA[15] <= lo0 selects chain B; count of elements <= lo; the smallest element above lo must be above hi or the
 *            lane is unsure; the first lane with an element above lo names the position; none: the last slot of the last lane
 *   u32      fl((float)(top 27 bits of u)) * 2^-27
 *
 * Inputs are in DRAW order: index = lane * 32 + k, k < 16 chain A slot k, k >= 16 chain B slot k - 16; x has the site's own count
 * already removed (the kernel removes it from the packed integers, exactly).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define QT 32
#define MAXL 16

static inline uint32_t hash32(uint32_t a)
{
    a ^= a >> 16; a *= 0x7feb352dU; a ^= a >> 15; a *= 0x846ca68bU; a ^= a >> 16;
    return a;
}

static inline float rcp_model(float d, int mode, uint32_t key)
{
    float y = 1.0f / d;                                        /* IEEE division: correctly rounded */
    if (mode == 1) {
        const uint32_t h = hash32(key) % 3u;
        if (h == 1) y = nextafterf(y, INFINITY);
        else if (h == 2) y = nextafterf(y, 0.0f);
    }
    return y;
}

/* out (per vector): pos_t0 = draw-order index tier 0 names, unsure = some lane is not sure, pos_exact = the real-number draw,
 * ratios[8] = { max |delta| / m,  max |delta| / bound*,  max prefix error / (28.1 v L*),  max scan error / (33.1 v (P* + L*)),
 *               total error / (33.1 v total*),  |u32 - u| / (2^-27 + v u),  m / bound* (minimum over lanes),  unsorted chain (0 / 1) }
 * returns 0, or -1 on bad arguments */
int quad_tier0_model(int LB, int64_t n, const int32_t *x, const int32_t *nd, const int32_t *nk, const uint8_t *valid,
                     const double *u53, double alpha, double beta, double vbeta, int rcp_mode, uint32_t rcp_seed,
                     float margin_rel, float margin_data, int32_t *pos_t0, uint8_t *unsure, int32_t *pos_exact, double *ratios)
{
    if (LB < 2 || LB > 4) return -1;
    const int LPD = 1 << LB, KP = QT * LPD;
    const float alpha32 = (float)alpha, beta32 = (float)beta, vbeta32 = (float)vbeta;
    const float QM_L = 1.05f * 32.0f * 0x1p-24f, QM_T = 1.05f * 39.0f * 0x1p-24f, QM_P = 1.05f * 37.0f * 0x1p-24f,
                QM_TOT = 1.05f * 0.25f * 0x1p-24f;
    const long double v = 0x1p-24L;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const int32_t *xi = x + i * KP, *ndi = nd + i * KP, *nki = nk + i * KP;
        float Q[MAXL][QT], X0[MAXL], X[MAXL], tot[MAXL];
        long double cum[MAXL * QT], pre[MAXL * QT], L_[MAXL], P_[MAXL];    /* pre: prefix inside the element's own chain */
        long double run = 0.0L;
        double r_prefix = 0.0, r_unsorted = 0.0;
        for (int l = 0; l < LPD; ++l) {
            long double lane0 = run;
            for (int c = 0; c < 2; ++c) {
                float q = 0.0f;
                long double chain = 0.0L;
                for (int a = 0; a < 16; ++a) {
                    const int k = 16 * c + a, j = l * QT + k;
                    const int ok = valid ? valid[j] : 1;
                    const float af = (float)ndi[j] + alpha32, den = (float)nki[j] + vbeta32;
                    const float pa = ok ? af * rcp_model(den, rcp_mode, rcp_seed ^ (uint32_t)(i * 2654435761u) ^ (uint32_t)j * 40503u) : 0.0f;
                    const float nb = (float)xi[j] + beta32;
                    const float qn = a == 0 ? nb * pa : fmaf(nb, pa, q);
                    if (qn < q) r_unsorted = 1.0;
                    q = qn;
                    Q[l][k] = q;
                    if (ok) {
                        const long double w = ((long double)ndi[j] + alpha) * (((long double)xi[j] + beta) / ((long double)nki[j] + vbeta));
                        run += w;
                        chain += w;
                    }
                    cum[j] = run;
                    pre[j] = chain;                              /* (summed on its own: no cancellation against the lanes before) */
                }
            }
            L_[l] = pre[l * QT + 15] + pre[l * QT + 31];
            P_[l] = lane0;
            X0[l] = Q[l][15] + Q[l][31];
            X[l] = X0[l];
            tot[l] = X0[l];
        }
        const long double total = run;
        /* prefix errors: chain A against its own prefix, chain B against ITS prefix (the kernel shifts the bounds by chain A's total) */
        for (int l = 0; l < LPD; ++l) {
            for (int k = 0; k < QT; ++k) {
                const long double e = fabsl((long double)Q[l][k] - pre[l * QT + k]);
                if (L_[l] > 0 && (double)(e / (28.1L * v * L_[l])) > r_prefix) r_prefix = (double)(e / (28.1L * v * L_[l]));
            }
        }
        /* scan: Hillis-Steele, all lanes at once */
        for (int d = 1; d < LPD; d <<= 1) {
            float Y[MAXL];
            for (int l = 0; l < LPD; ++l) Y[l] = l >= d ? X[l - d] + X[l] : X[l];
            memcpy(X, Y, sizeof(float) * LPD);
        }
        /* total: butterflies */
        {
            float Y[MAXL];
            if (LB == 4) {
                for (int s = 8; s >= 1; s >>= 1) {
                    for (int l = 0; l < 16; ++l) Y[l] = tot[l] + tot[(l + 16 - s) & 15];
                    memcpy(tot, Y, sizeof(float) * 16);
                }
            } else {
                if (LB == 3) {
                    for (int l = 0; l < 8; ++l) Y[l] = tot[l] + tot[7 - l];
                    memcpy(tot, Y, sizeof(float) * 8);
                }
                for (int s = 1; s <= 2; s <<= 1) {
                    for (int l = 0; l < LPD; ++l) Y[l] = tot[l] + tot[l ^ s];
                    memcpy(tot, Y, sizeof(float) * LPD);
                }
            }
        }
        double r_scan = 0.0, r_tot = 0.0;
        for (int l = 0; l < LPD; ++l) {
            const long double s = P_[l] + L_[l];
            if (s > 0) {
                const double r = (double)(fabsl((long double)X[l] - s) / (33.1L * v * s));
                if (r > r_scan) r_scan = r;
            }
            if (tot[l] != tot[0]) r_unsorted = 2.0;                 /* (the butterflies give every lane the same total) */
        }
        if (total > 0) r_tot = (double)(fabsl((long double)tot[0] - total) / (33.1L * v * total));
        const double u = u53[i];
        const float u32 = (float)(uint32_t)floor(u * 134217728.0) * 0x1p-27f;
        const double r_u = fabs((double)u32 - u) / (0x1p-27 + 0x1p-24 * u);
        const long double tstar = (long double)u * total;
        /* the real-number draw: the first position with a topic whose prefix exceeds u total, else the last position with a topic */
        int pe = -1, lastv = -1;
        for (int j = 0; j < KP; ++j) {
            if (valid && !valid[j]) continue;
            lastv = j;
            if (pe < 0 && cum[j] > tstar) pe = j;
        }
        pos_exact[i] = pe >= 0 ? pe : lastv;
        /* tier 0, lane by lane */
        double r_m = 0.0, r_b = 0.0, r_mb = 1e300;
        int winner = -1, any_unsure = 0;
        for (int l = 0; l < LPD; ++l) {
            const float prev = l ? X[l - 1] : 0.0f;
            const float t = u32 * tot[l];
            const float tg = t - prev;
            const float md = fmaf(QM_L, X0[l], fmaf(QM_T, t, fmaf(QM_P, prev, QM_TOT * tot[l])));
            const float margin = fmaf(tot[l], margin_rel, margin_data * md);
            const float lo0 = tg - margin, hi0 = tg + margin;
            const int c0 = Q[l][15] <= lo0;
            const float dA = c0 ? Q[l][15] : 0.0f;
            const float lo = lo0 - dA, hi = hi0 - dA;
            const float *q = &Q[l][c0 ? 16 : 0];
            int cnt = 0;
            float ub = INFINITY;
            for (int a = 0; a < 16; ++a) {
                if (q[a] <= lo) ++cnt;
                else if (q[a] < ub) ub = q[a];
            }
            const float tm = tot[l] - margin;
            const int bad_total = !(isnormal(tm) && tm > 0.0f);
            if (!(ub > hi) || bad_total) any_unsure = 1;
            if (winner < 0 && cnt < 16) winner = l * QT + (c0 ? 16 : 0) + cnt;
            /* every compared difference against the real-number one: chain A against (lo0, hi0), chain B against the bounds minus
               chain A's total (whichever chain THIS lane searched: the bound must hold for both) */
            const float loB = lo0 - Q[l][15], hiB = hi0 - Q[l][15];
            const long double bound = v * (31.1L * L_[l] + 38.2L * tstar + 36.1L * P_[l] + 0.125L * total);
            const long double m = (long double)margin;
            if (bound > 0 && (double)(m / bound) < r_mb) r_mb = (double)(m / bound);
            for (int k = 0; k < QT; ++k) {
                const int j = l * QT + k;
                if (valid && !valid[j]) continue;
                const long double E = cum[j] - tstar;
                const long double qk = (long double)Q[l][k];
                const long double dlo = (qk - (long double)(k < 16 ? lo0 : loB)) - (E + m);
                const long double dhi = (qk - (long double)(k < 16 ? hi0 : hiB)) - (E - m);
                const long double dmax = fabsl(dlo) > fabsl(dhi) ? fabsl(dlo) : fabsl(dhi);
                if (m > 0 && (double)(dmax / m) > r_m) r_m = (double)(dmax / m);
                if (bound > 0 && (double)(dmax / bound) > r_b) r_b = (double)(dmax / bound);
            }
        }
        pos_t0[i] = winner >= 0 ? winner : KP - 1;
        unsure[i] = (uint8_t)any_unsure;
        double *r = ratios + i * 8;
        r[0] = r_m; r[1] = r_b; r[2] = r_prefix; r[3] = r_scan; r[4] = r_tot; r[5] = r_u; r[6] = r_mb; r[7] = r_unsorted;
    }
    return 0;
}
