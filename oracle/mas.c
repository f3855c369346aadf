/* Oracle (TEST INFRASTRUCTURE ONLY): plain-C restatement of the two numba-JIT CPU loops
 * of the reference, used by tests and by bench.py's cpu_baseline leg.
 *
 *   osp_oracle_mas            <- optispeech/model/generator/alignments.py:177-207
 *                                (_monotonic_alignment_search)
 *   osp_oracle_avg_by_dur     <- optispeech/model/generator/alignments.py:242-259
 *                                (_average_by_duration)
 *
 * numba semantics assumed for alignments.py:188 (`log_prob[0, :j+1].sum()` on a float32
 * slice): sequential float32 accumulation, stored into the float64 Q (numba is not
 * installable in the build container, so this point is reasoned, not verified -- see
 * oracle/__init__.py).  Everything else is float64 exactly as written in the reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* lp: (T_mel, ld) row-major float32, valid region [0,T_mel) x [0,T_inp).  path: int64[T_mel]. */
int osp_oracle_mas(const float *lp, int T_mel, int T_inp, int ld, int64_t *path) {
    if (T_mel <= 0 || T_inp <= 0) return -1;
    double *Q = (double *)malloc(sizeof(double) * (size_t)T_inp * (size_t)T_mel);
    if (!Q) return -2;
    for (size_t k = 0; k < (size_t)T_inp * (size_t)T_mel; ++k) Q[k] = -INFINITY;   /* :182 */
#define Qa(i, j) Q[(size_t)(i) * (size_t)T_mel + (size_t)(j)]
#define LP(i, j) lp[(size_t)(j) * (size_t)ld + (size_t)(i)]                           /* :184 transpose */
    float acc = 0.0f;                                                                 /* :186-188 */
    for (int j = 0; j < T_mel; ++j) {
        acc = acc + LP(0, j);
        Qa(0, j) = (double)acc;
    }
    for (int j = 1; j < T_mel; ++j) {                                                 /* :191-193 */
        int lim = (j + 1 < T_inp) ? (j + 1) : T_inp;
        for (int i = 1; i < lim; ++i) {
            double a = Qa(i - 1, j - 1), b = Qa(i, j - 1);
            Qa(i, j) = (a > b ? a : b) + (double)LP(i, j);
        }
    }
    path[T_mel - 1] = T_inp - 1;                                                      /* :196 */
    for (int j = T_mel - 2; j >= 0; --j) {                                            /* :197-206 */
        int64_t i_b = path[j + 1], i_a = i_b - 1, arg;
        if (i_b == 0) arg = 0;
        else if (Qa(i_a, j) >= Qa(i_b, j)) arg = i_a;
        else arg = i_b;
        path[j] = arg;
    }
    free(Q);
    return 0;
#undef Qa
#undef LP
}

/* ds: (B, Tt) float32 durations; xs: (B, Tf) float32; out: (B, Tt) float32 (pre-zeroed by caller or not). */
int osp_oracle_avg_by_dur(const float *ds, const float *xs, const int64_t *text_len,
                          const int64_t *feat_len, int B, int Tt, int Tf, float *out) {
    for (int b = 0; b < B; ++b) {
        int64_t start = 0;
        for (int n = 0; n < Tt; ++n) out[(size_t)b * Tt + n] = 0.0f;                   /* :244 */
        for (int n = 0; n < (int)text_len[b]; ++n) {
            int32_t d = (int32_t)ds[(size_t)b * Tt + n];                               /* :245 */
            int64_t end = start + d;
            int64_t s = start < feat_len[b] ? start : feat_len[b];                     /* x = xs[b,:t_feats] slicing clamps */
            int64_t e = end < feat_len[b] ? end : feat_len[b];
            if (e > s) {                                                               /* :254-255 */
                float acc = 0.0f;
                for (int64_t t = s; t < e; ++t) acc += xs[(size_t)b * Tf + t];
                out[(size_t)b * Tt + n] = acc / (float)(e - s);
            }
            start = end;
        }
    }
    return 0;
}
