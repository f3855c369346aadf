// Alignment learning on the device (reference rows A7, A7b, A8, A9, A10, A10b, A10c; all citations
// optispeech/model/generator/alignments.py unless noted).  Nothing here ever leaves the GPU: the reference
// does >= 2B+4 host round trips per step for these ops.
//
//   osp_lgamma_table / osp_betabinom_prior   :85-123   beta-binomial log prior from a log-factorial table (f64)
//   osp_pairwise_score                       :66-72    score = -||f_t - e_n||_2, -inf on padded tokens
//   osp_logsoftmax_prior_fwd / _bwd          :74-81    log_softmax over tokens + prior; backward emits the
//                                                      pair weights w = g/score consumed by two GEMMs
//   osp_mas                                  :177-239  monotonic alignment search, one wavefront per
//                                                      utterance, f64 DP, bit-exact path + durations + bin loss
//   osp_duration_stats                       :242-280, :167  token-level averages, gaussian centres
//   osp_gaussian_weights                     :163-172  softmax weights of the Gaussian upsampler
//   osp_gather_rows                          utils/segments.py:41-72  segment slicing by start index
//   osp_expand_by_duration                   :283-297  hard length regulator (index gather)
#include "osp_common.h"

// ------------------------------------------------------------------------------------------------ prior
__global__ void lgamma_table_kernel(double* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i > 0 ? lgamma((double)i) : 0.0;
}
extern "C" int osp_lgamma_table(double* out, int64_t n, hipStream_t stream) {
    OSP_CHECK_ARG(out && n > 0, "bad args");
    hipLaunchKernelGGL(lgamma_table_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, stream, out, (int)n);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// logpmf(k; n=N, a=t+1, b=T-t) for t in [0,T), k in [0,N):  lchoose(N,k) + lbeta(k+a, N-k+b) - lbeta(a,b)
// (all arguments are integers for w=1, :110-114), evaluated from lg[i] = lgamma(i).
__global__ void betabinom_prior_kernel(const double* __restrict__ lg, const int64_t* __restrict__ x_len,
                                       const int64_t* __restrict__ y_len, float* __restrict__ out, int Tm, int Nm) {
    const int b = blockIdx.z, t = blockIdx.y;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= Nm) return;
    const int T = (int)y_len[b], N = (int)x_len[b];
    float v = -INFINITY;
    if (t < T && n < N) {
        const int a = t + 1, bb = T - t, k = n;
        const double lchoose = lg[N + 1] - lg[k + 1] - lg[N - k + 1];
        const double lb1 = lg[k + a] + lg[N - k + bb] - lg[N + a + bb];
        const double lb0 = lg[a] + lg[bb] - lg[a + bb];
        v = (float)(lchoose + lb1 - lb0);
    }
    out[((int64_t)b * Tm + t) * Nm + n] = v;
}
extern "C" int osp_betabinom_prior(const double* lg, int64_t lg_len, const int64_t* x_len, const int64_t* y_len,
                                   float* out, int64_t B, int64_t Tm, int64_t Nm, hipStream_t stream) {
    OSP_CHECK_ARG(lg && x_len && y_len && out, "null operand");
    OSP_CHECK_ARG(lg_len >= Tm + Nm + 3, "lgamma table too short");
    hipLaunchKernelGGL(betabinom_prior_kernel, dim3((unsigned)cdiv(Nm, 128), (unsigned)Tm, (unsigned)B), dim3(128), 0,
                       stream, lg, x_len, y_len, out, (int)Tm, (int)Nm);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// ------------------------------------------------------------------------------------------------ pairwise score
// 64 frames x 64 tokens per 256-thread block, 4x4 micro-tile per thread, 16-channel slabs through LDS (k-major).
// Direct differences (not the ||f||^2+||e||^2-2fe expansion): no cancellation, 3 VALU ops per pair-channel.
__global__ __launch_bounds__(256) void pairwise_score_kernel(const float* __restrict__ f, const float* __restrict__ e,
                                                             const int64_t* __restrict__ x_len, float* __restrict__ score,
                                                             int Tm, int Nm, int C) {
    __shared__ float fs[16][64 + 4], es[16][64 + 4];
    const int b = blockIdx.z, t0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const float* fb = f + (int64_t)b * Tm * C;
    const float* eb = e + (int64_t)b * Nm * C;
    float acc[4][4] = {};
    const int lr = tid >> 2, lk = (tid & 3) * 4;   // loader: row 0..63, k offset
    for (int c0 = 0; c0 < C; c0 += 16) {
        float4 vf = make_float4(0, 0, 0, 0), ve = make_float4(0, 0, 0, 0);
        if (t0 + lr < Tm && c0 + lk < C) vf = *reinterpret_cast<const float4*>(fb + (int64_t)(t0 + lr) * C + c0 + lk);
        if (n0 + lr < Nm && c0 + lk < C) ve = *reinterpret_cast<const float4*>(eb + (int64_t)(n0 + lr) * C + c0 + lk);
        __syncthreads();
        fs[lk + 0][lr] = vf.x; fs[lk + 1][lr] = vf.y; fs[lk + 2][lr] = vf.z; fs[lk + 3][lr] = vf.w;
        es[lk + 0][lr] = ve.x; es[lk + 1][lr] = ve.y; es[lk + 2][lr] = ve.z; es[lk + 3][lr] = ve.w;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float a[4], bb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = fs[k][ty * 4 + i]; bb[i] = es[k][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float d = a[i] - bb[j]; acc[i][j] = fmaf(d, d, acc[i][j]); }
        }
    }
    const int N = (int)x_len[b];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = t0 + ty * 4 + i;
        if (t >= Tm) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n < Nm) score[((int64_t)b * Tm + t) * Nm + n] = n < N ? -sqrtf(acc[i][j]) : -INFINITY;
        }
    }
}
extern "C" int osp_pairwise_score(const float* f, const float* e, const int64_t* x_len, float* score, int64_t B,
                                  int64_t Tm, int64_t Nm, int64_t C, hipStream_t stream) {
    OSP_CHECK_ARG(f && e && x_len && score, "null operand");
    OSP_CHECK_ARG(C % 4 == 0, "C must be a multiple of 4");
    hipLaunchKernelGGL(pairwise_score_kernel, dim3((unsigned)cdiv(Nm, 64), (unsigned)cdiv(Tm, 64), (unsigned)B), dim3(256),
                       0, stream, f, e, x_len, score, (int)Tm, (int)Nm, (int)C);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// ------------------------------------------------------------------------------------------------ log-softmax + prior
// one wavefront per (b,t) row; lse saved for the backward.
__global__ __launch_bounds__(256) void logsoftmax_prior_fwd_kernel(const float* __restrict__ score,
                                                                   const float* __restrict__ prior,
                                                                   float* __restrict__ lp, float* __restrict__ lse,
                                                                   int64_t rows, int Nm) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* s = score + row * Nm;
    float mx = -INFINITY;
    for (int n = lane; n < Nm; n += 64) mx = fmaxf(mx, s[n]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int n = lane; n < Nm; n += 64) sum += expf(s[n] - mx);     // exp(-inf) = 0 on padded tokens
    const float l = mx + logf(wave_sum(sum));
    for (int n = lane; n < Nm; n += 64) lp[row * Nm + n] = (s[n] - l) + prior[row * Nm + n];
    if (lane == 0) lse[row] = l;
}
extern "C" int osp_logsoftmax_prior_fwd(const float* score, const float* prior, float* lp, float* lse, int64_t rows,
                                        int64_t Nm, hipStream_t stream) {
    OSP_CHECK_ARG(score && prior && lp && lse, "null operand");
    hipLaunchKernelGGL(logsoftmax_prior_fwd_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, stream, score, prior, lp,
                       lse, rows, (int)Nm);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// backward of (log_softmax + norm): g[n] = dlp[n] - softmax[n] * sum_n' dlp[n'];  w[n] = -g/dist = g/score
// (0 where dist == 0: subgradient of torch.norm at 0, or outside the valid block).  Also emits rowsum(w).
__global__ __launch_bounds__(256) void logsoftmax_prior_bwd_kernel(const float* __restrict__ dlp,
                                                                   const float* __restrict__ score,
                                                                   const float* __restrict__ lse,
                                                                   const int64_t* __restrict__ x_len,
                                                                   const int64_t* __restrict__ y_len,
                                                                   float* __restrict__ w, float* __restrict__ wrow,
                                                                   int B, int Tm, int Nm) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (int64_t)B * Tm) return;
    const int b = (int)(row / Tm), t = (int)(row - (int64_t)b * Tm);
    const int N = (int)x_len[b], T = (int)y_len[b];
    float* wr = w + row * Nm;
    if (t >= T) {
        for (int n = lane; n < Nm; n += 64) wr[n] = 0.f;
        if (lane == 0) wrow[row] = 0.f;
        return;
    }
    const float* d = dlp + row * Nm;
    const float* s = score + row * Nm;
    float tot = 0.f;
    for (int n = lane; n < N; n += 64) tot += d[n];
    tot = wave_sum(tot);
    const float l = lse[row];
    float ws = 0.f;
    for (int n = lane; n < Nm; n += 64) {
        float v = 0.f;
        if (n < N) {
            const float sc = s[n];
            const float g = d[n] - expf(sc - l) * tot;
            v = sc != 0.f ? g / sc : 0.f;
        }
        wr[n] = v;
        ws += v;
    }
    ws = wave_sum(ws);
    if (lane == 0) wrow[row] = ws;
}
extern "C" int osp_logsoftmax_prior_bwd(const float* dlp, const float* score, const float* lse, const int64_t* x_len,
                                        const int64_t* y_len, float* w, float* wrow, int64_t B, int64_t Tm,
                                        int64_t Nm, hipStream_t stream) {
    OSP_CHECK_ARG(dlp && score && lse && x_len && y_len && w && wrow, "null operand");
    hipLaunchKernelGGL(logsoftmax_prior_bwd_kernel, dim3((unsigned)cdiv(B * Tm, 4)), dim3(256), 0, stream, dlp, score, lse,
                       x_len, y_len, w, wrow, (int)B, (int)Tm, (int)Nm);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// ------------------------------------------------------------------------------------------------ MAS
// One wavefront per utterance.  Token i lives in (round r = i / 64, lane = i % 64); Q[., j-1] stays in
// registers as f64 and column j is produced with one shuffle per round -- no barrier in the 800-step loop.
// Bit-exact contract (see oracle/mas.c): row 0 is a sequential f32 prefix sum widened to f64; all other cells
// f64 max/add; the back-track choice `Q[i-1,j] >= Q[i,j]` is recorded as one ballot bit per cell while column
// j+1 is formed (the same two values), and replayed by lane 0.
template <int R>
__global__ __launch_bounds__(64) void mas_kernel(const float* __restrict__ lp, const int64_t* __restrict__ x_len,
                                                 const int64_t* __restrict__ y_len, int* __restrict__ path,
                                                 float* __restrict__ durations, float* __restrict__ bin_sum,
                                                 unsigned long long* __restrict__ bits_ws, int Tm, int Nm,
                                                 int bits_in_lds) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long sbits[];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int T = (int)y_len[b], N = (int)x_len[b];
    const float* L = lp + (int64_t)b * Tm * Nm;
    unsigned long long* bits = bits_in_lds ? sbits : bits_ws + (int64_t)b * Tm * R;
    int* pth = path + (int64_t)b * Tm;
    float* dur = durations + (int64_t)b * Nm;
    for (int n = lane; n < Nm; n += 64) dur[n] = 0.f;
    if (T <= 0 || N <= 0) return;

    double q[R];
    float acc0 = 0.f;
    // The recursion is T dependent steps of a few dozen instructions; what it must never do is wait for memory inside a step.
    // Round 2 requested a row ONE step ahead (every step waited a memory latency: 0.46 ms for T = 800, on the critical chain of
    // the training step).  A register ring refilled inside the steps is no better: the use of an old entry makes the compiler
    // drain vmcnt, the just-issued refill included, and a conditional load is an exec-masked branch.  So: the rows of the NEXT
    // PD frames are requested as a chunk -- unconditional loads from clamped indices, untouched until the chunk is used -- before
    // the PD steps of the current chunk, which consume registers only; one wait per chunk.
    constexpr int PD = R <= 4 ? 8 : (R <= 8 ? 4 : 2);                  // 2 * PD * R row registers: wide texts take shorter chunks
    float cur[PD][R], nxt[PD][R];
    int icl[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { const int i = r * 64 + lane; icl[r] = i < N ? i : 0; q[r] = -INFINITY; }
    auto fetch = [&](float (&dst)[PD][R], int j0) {
#pragma unroll
        for (int k = 0; k < PD; ++k) {
            const int j = j0 + k < T ? j0 + k : T - 1;
#pragma unroll
            for (int r = 0; r < R; ++r) dst[k][r] = L[(int64_t)j * Nm + icl[r]];
        }
    };
    fetch(cur, 0);
    for (int j0 = 0; j0 < T; j0 += PD) {
      fetch(nxt, j0 + PD);
#pragma unroll
      for (int k = 0; k < PD; ++k) {
        const int j = j0 + k;
        if (j >= T) break;                                             // wave-uniform
        double qn[R];
        double carry = 0.0;   // lane 63 of the previous round (column j-1)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = r * 64 + lane;
            const float lpv = i < N ? cur[k][r] : 0.f;
            // column i - 1 of the previous frame: a whole-wave shift by one lane (DPP wave_shr:1, lane 0 takes the carry = lane 63 of
            // the previous 64-column group) and a readlane -- round 2 used __shfl_up / __shfl on doubles = four ds_bpermute round
            // trips through the LDS crossbar per group and step, on the critical path of a T-step recursion
            const int qlo = __double2loint(q[r]), qhi = __double2hiint(q[r]);
            const int plo = __builtin_amdgcn_update_dpp(__double2loint(carry), qlo, 0x138, 0xf, 0xf, false);
            const int phi = __builtin_amdgcn_update_dpp(__double2hiint(carry), qhi, 0x138, 0xf, 0xf, false);
            const double prev = __hiloint2double(phi, plo);
            carry = __hiloint2double(__builtin_amdgcn_readlane(qhi, 63), __builtin_amdgcn_readlane(qlo, 63));
            const bool take_prev = prev >= q[r];
            const unsigned long long mask = __ballot(take_prev);
            // every lane stores the (wave-uniform) mask to the same address: one LDS / memory transaction, and no exec-mask branch
            // in the step; typed stores (a generic pointer is a flat store, which counts against both vmcnt and lgkmcnt)
            if (bits_in_lds) sbits[(int64_t)j * R + r] = mask;
            else bits_ws[((int64_t)b * Tm + j) * R + r] = mask;
            // branch-free (three-way divergent branches cost more scalar latency per step than the arithmetic they skip):
            // column 0 is the sequential f32 prefix sum (:186-188), columns 1 .. min(j, N - 1) the DP update (:191-193), the
            // rest -inf.  Values and operation order per column are exactly those of the branched form.
            if (r == 0) acc0 = acc0 + (i == 0 ? lpv : 0.f);            // only lane 0 of group 0 ever reads it (compile-time r)
            const double v_dp = (take_prev ? prev : q[r]) + (double)lpv;
            const double v_in = (i < N && i <= j) ? v_dp : (double)-INFINITY;
            qn[r] = (i == 0) ? (double)acc0 : v_in;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) q[r] = qn[r];
      }
#pragma unroll
      for (int k = 0; k < PD; ++k)
#pragma unroll
          for (int r = 0; r < R; ++r) cur[k][r] = nxt[k][r];
    }
    __syncthreads();   // single wave: orders the LDS / global bit stores before lane 0 reads them
    if (lane == 0) {
        int i = N - 1, run = 0;                                        // :196
        pth[T - 1] = i;
        run = 1;
        for (int j = T - 2; j >= 0; --j) {                             // :197-206
            int arg;
            if (i == 0) arg = 0;
            else {
                const unsigned long long m = bits[(int64_t)(j + 1) * R + (i >> 6)];
                arg = ((m >> (i & 63)) & 1ull) ? i - 1 : i;
            }
            if (arg != i) { dur[i] = (float)run; run = 0; }
            i = arg;
            pth[j] = i;
            ++run;
        }
        dur[i] = (float)run;                                           // np.bincount(viterbi), :234
    }
    __syncthreads();
    // binarisation loss term of this utterance: -mean_t lp[t, A[t]]  (:236-237)
    float s = 0.f;
    for (int t = lane; t < T; t += 64) s += L[(int64_t)t * Nm + pth[t]];
    s = wave_sum(s);
    if (lane == 0) bin_sum[b] = -s / (float)T;
}

extern "C" int64_t osp_mas_workspace_bytes(int64_t B, int64_t Tm, int64_t Nm) {
    const int64_t R = (Nm + 63) / 64;
    int64_t Rp = 1;
    while (Rp < R) Rp *= 2;
    return Tm * Rp * 8 <= 60000 ? 0 : B * Tm * Rp * 8;
}

extern "C" int osp_mas(const float* lp, const int64_t* x_len, const int64_t* y_len, int* path, float* durations,
                       float* bin_sum, void* workspace, int64_t B, int64_t Tm, int64_t Nm, hipStream_t stream) {
    OSP_CHECK_ARG(lp && x_len && y_len && path && durations && bin_sum, "null operand");
    OSP_CHECK_ARG(Nm <= 2048, "at most 2048 tokens");
    const int64_t R = (Nm + 63) / 64;
    int Rp = 1;
    while (Rp < R) Rp *= 2;
    const int64_t lds = Tm * Rp * 8;
    const int in_lds = lds <= 60000;
    OSP_CHECK_ARG(in_lds || workspace, "workspace required (osp_mas_workspace_bytes)");
    const size_t smem = in_lds ? (size_t)lds : 0;
#define L(RR) hipLaunchKernelGGL((mas_kernel<RR>), dim3((unsigned)B), dim3(64), smem, stream, lp, x_len, y_len, path, durations, bin_sum, (unsigned long long*)workspace, (int)Tm, (int)Nm, in_lds)
    switch (Rp) {
        case 1: L(1); break; case 2: L(2); break; case 4: L(4); break; case 8: L(8); break;
        case 16: L(16); break; default: L(32);
    }
#undef L
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// scatter of the binarisation-loss gradient: dlp[b, t, A[t]] += -gscale / (B * T_b)
__global__ void bin_loss_bwd_kernel(const int* __restrict__ path, const int64_t* __restrict__ y_len,
                                    const float* __restrict__ gscale, float* __restrict__ dlp, int B, int Tm, int Nm) {
    const int b = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int)y_len[b]) return;
    dlp[((int64_t)b * Tm + t) * Nm + path[(int64_t)b * Tm + t]] += -gscale[0] / ((float)B * (float)y_len[b]);
}
extern "C" int osp_bin_loss_bwd(const int* path, const int64_t* y_len, const float* gscale, float* dlp, int64_t B,
                                int64_t Tm, int64_t Nm, hipStream_t stream) {
    OSP_CHECK_ARG(path && y_len && gscale && dlp, "null operand");
    hipLaunchKernelGGL(bin_loss_bwd_kernel, dim3((unsigned)cdiv(Tm, 256), (unsigned)B), dim3(256), 0, stream, path, y_len,
                       gscale, dlp, (int)B, (int)Tm, (int)Nm);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// ------------------------------------------------------------------------------------------------ duration statistics
// One wavefront per utterance: inclusive scan of the (integer-valued) durations, then per token
//   avg_k[n] = mean(xs_k[b, start:end]) clipped to the utterance's frames (0 when empty)     (:242-259)
//   centre[n] = cumsum[n] - d[n]/2                                                            (:167)
__global__ __launch_bounds__(64) void duration_stats_kernel(const float* __restrict__ ds, const float* __restrict__ xs0,
                                                            const float* __restrict__ xs1,
                                                            const int64_t* __restrict__ x_len,
                                                            const int64_t* __restrict__ y_len, float* __restrict__ avg0,
                                                            float* __restrict__ avg1, float* __restrict__ centre, int Tm,
                                                            int Nm) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int N = x_len ? (int)x_len[b] : Nm, T = y_len ? (int)y_len[b] : Tm;
    const float* d = ds + (int64_t)b * Nm;
    float base = 0.f;   // running cumsum over chunks of 64 tokens (exact: small integers in f32)
    for (int n0 = 0; n0 < Nm; n0 += 64) {
        const int n = n0 + lane;
        const float dv = n < Nm ? d[n] : 0.f;
        float inc = dv;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float up = __shfl_up(inc, o, 64);
            if (lane >= o) inc += up;
        }
        const float cum = base + inc;
        base += __shfl(inc, 63, 64);
        if (n < Nm) {
            if (centre) centre[(int64_t)b * Nm + n] = cum - dv / 2.f;
            if (avg0) {
                float a0 = 0.f, a1 = 0.f;
                if (n < N) {
                    const int di = (int)dv;                                  // ds.astype(int32) :245
                    const int end = (int)cum;
                    int s = end - di, e = end;
                    s = s < T ? s : T;
                    e = e < T ? e : T;
                    if (e > s) {
                        float s0 = 0.f, s1 = 0.f;
                        for (int t = s; t < e; ++t) {
                            s0 += xs0[(int64_t)b * Tm + t];
                            if (xs1) s1 += xs1[(int64_t)b * Tm + t];
                        }
                        a0 = s0 / (float)(e - s);
                        a1 = s1 / (float)(e - s);
                    }
                }
                avg0[(int64_t)b * Nm + n] = a0;
                if (avg1) avg1[(int64_t)b * Nm + n] = a1;
            }
        }
    }
}
extern "C" int osp_duration_stats(const float* ds, const float* xs0, const float* xs1, const int64_t* x_len,
                                  const int64_t* y_len, float* avg0, float* avg1, float* centre, int64_t B,
                                  int64_t Tm, int64_t Nm, hipStream_t stream) {
    OSP_CHECK_ARG(ds, "null operand");
    OSP_CHECK_ARG(!avg0 || xs0, "avg0 needs xs0");
    OSP_CHECK_ARG(!avg1 || (xs1 && avg0), "avg1 needs xs1 and avg0");
    hipLaunchKernelGGL(duration_stats_kernel, dim3((unsigned)B), dim3(64), 0, stream, ds, xs0, xs1, x_len, y_len, avg0, avg1,
                       centre, (int)Tm, (int)Nm);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// ------------------------------------------------------------------------------------------------ gaussian upsampling weights
// P[b,t,n] = softmax_n( -delta * (t*[t < T_b] - c[b,n])^2 ), padded tokens -> 0      (:163-172)
__global__ __launch_bounds__(256) void gaussian_weights_kernel(const float* __restrict__ centre,
                                                               const int64_t* __restrict__ x_len,
                                                               const int64_t* __restrict__ y_len, float delta,
                                                               float* __restrict__ P, int B, int Tm, int Nm) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (int64_t)B * Tm) return;
    const int b = (int)(row / Tm), t = (int)(row - (int64_t)b * Tm);
    const int N = (int)x_len[b];
    const float tf = t < (int)y_len[b] ? (float)t : 0.f;                  // t * h_masks  (:164-165)
    const float* c = centre + (int64_t)b * Nm;
    float mx = -INFINITY;
    for (int n = lane; n < N; n += 64) { const float d = tf - c[n]; mx = fmaxf(mx, -1.f * delta * (d * d)); }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int n = lane; n < N; n += 64) { const float d = tf - c[n]; sum += expf(-1.f * delta * (d * d) - mx); }
    const float inv = 1.f / wave_sum(sum);
    for (int n = lane; n < Nm; n += 64) {
        float v = 0.f;
        if (n < N) { const float d = tf - c[n]; v = expf(-1.f * delta * (d * d) - mx) * inv; }
        P[row * Nm + n] = v;
    }
}
extern "C" int osp_gaussian_weights(const float* centre, const int64_t* x_len, const int64_t* y_len, float delta,
                                    float* P, int64_t B, int64_t Tm, int64_t Nm, hipStream_t stream) {
    OSP_CHECK_ARG(centre && x_len && y_len && P, "null operand");
    hipLaunchKernelGGL(gaussian_weights_kernel, dim3((unsigned)cdiv(B * Tm, 4)), dim3(256), 0, stream, centre, x_len, y_len,
                       delta, P, (int)B, (int)Tm, (int)Nm);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// ------------------------------------------------------------------------------------------------ gathers
// out[b, s, :] = src[b, start[b]*mult + s, :]  for s < S   (rows of C floats; C % 4 == 0 -> float4 path)
__global__ void gather_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ start, int64_t mult,
                                   float* __restrict__ out, int T, int S, int C) {
    const int b = blockIdx.z, s = blockIdx.y;
    const int64_t t = start[b] * mult + s;
    const float* p = src + ((int64_t)b * T + t) * C;
    float* o = out + ((int64_t)b * S + s) * C;
    const bool ok = t >= 0 && t < T;
    if ((C & 3) == 0) {
        for (int c = (blockIdx.x * blockDim.x + threadIdx.x) * 4; c < C; c += gridDim.x * blockDim.x * 4)
            *reinterpret_cast<float4*>(o + c) = ok ? *reinterpret_cast<const float4*>(p + c) : make_float4(0, 0, 0, 0);
    } else {
        for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) o[c] = ok ? p[c] : 0.f;
    }
}
extern "C" int osp_gather_rows(const float* src, const int64_t* start, int64_t mult, float* out, int64_t B, int64_t T,
                               int64_t S, int64_t C, hipStream_t stream) {
    OSP_CHECK_ARG(src && start && out, "null operand");
    const int thr = C >= 1024 ? 256 : 64;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(1, (unsigned)S, (unsigned)B), dim3(thr), 0, stream, src, start, mult, out,
                       (int)T, (int)S, (int)C);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// hard length regulator: out[b, t, :] = x[b, n(t), :] with cum[n] <= t < cum[n+1]; zero past the length  (:283-297)
// A workgroup = 32 output frames of one utterance: the cumulative durations are scanned ONCE into LDS (Hillis-Steele over the
// tokens), every wave then takes frames, finds the token by a binary search in LDS (wave-uniform) and copies the row as float4s.
// (Round 1's kernel was one workgroup per frame whose thread 0 walked the durations serially: 80 us at 64 x 768 frames.)
#define EXP_FB 32
__global__ __launch_bounds__(256) void expand_by_duration_kernel(const float* __restrict__ x, const int64_t* __restrict__ dur,
                                                                 float* __restrict__ out, int Nm, int Tout, int C) {
    extern __shared__ int exp_cum[];                               // [2][Nm + 1]: ping-pong buffers of the scan
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = Nm + 1;
    int* a = exp_cum;
    int* bb = exp_cum + L;
    for (int n = tid; n < L; n += 256) a[n] = n == 0 ? 0 : (int)dur[(int64_t)b * Nm + n - 1];      // a[n] = dur[n - 1]: inclusive scan -> cum[n]
    __syncthreads();
    for (int off = 1; off < L; off <<= 1) {
        for (int n = tid; n < L; n += 256) bb[n] = a[n] + (n >= off ? a[n - off] : 0);
        __syncthreads();
        int* t_ = a; a = bb; bb = t_;
    }
    const int total = a[Nm];
    const int t0 = blockIdx.x * EXP_FB;
    for (int f = wave; f < EXP_FB; f += 4) {
        const int t = t0 + f;
        if (t >= Tout) break;
        int tok = -1;
        if (t < total) {                                           // largest n with cum[n] <= t (durations may be 0: the last such n has dur > 0)
            int lo = 0, hi = Nm;                                   // invariant: cum[lo] <= t < cum[hi]
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a[mid] <= t) lo = mid; else hi = mid; }
            tok = lo;
        }
        float* o = out + ((int64_t)b * Tout + t) * C;
        if ((C & 3) == 0) {
            const float4* src = reinterpret_cast<const float4*>(x + ((int64_t)b * Nm + (tok >= 0 ? tok : 0)) * C);
            for (int c = lane; c < C / 4; c += 64) reinterpret_cast<float4*>(o)[c] = tok >= 0 ? src[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            for (int c = lane; c < C; c += 64) o[c] = tok >= 0 ? x[((int64_t)b * Nm + tok) * C + c] : 0.f;
        }
    }
}
extern "C" int osp_expand_by_duration(const float* x, const int64_t* dur, float* out, int64_t B, int64_t Nm,
                                      int64_t Tout, int64_t C, hipStream_t stream) {
    OSP_CHECK_ARG(x && dur && out, "null operand");
    if (Tout == 0) return OSP_OK;
    OSP_CHECK_ARG(Nm > 0 && Nm <= 8000, "token count out of range");
    hipLaunchKernelGGL(expand_by_duration_kernel, dim3((unsigned)cdiv(Tout, EXP_FB), (unsigned)B), dim3(256), 2 * (Nm + 1) * sizeof(int), stream,
                       x, dur, out, (int)Nm, (int)Tout, (int)C);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
