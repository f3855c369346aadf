// Acoustic-model losses (reference rows A13, A14; optispeech/model/generator/loss.py).
//
//   osp_variance_losses      FastSpeech2Loss.forward :83-140 (+ DurationPredictorLoss :28-46), forward and the
//                            gradients w.r.t. the three predictions in one pass
//   osp_forwardsum_ctc       ForwardSumLoss.forward :150-194: blank-padded per-item log_softmax + CTC forward
//                            (alpha) and backward (beta) recursions, one wavefront per utterance, loss and
//                            d loss / d log_p_attn in one launch
#include "osp_common.h"

// ------------------------------------------------------------------------------------------------ A13
// The reference's masked_select broadcasts (B,T,1) against (B,1,T) / (B,1,T,1) masks (loss.py:111-120), so the
// "masked means" are weighted means (see oracle/losses.py):
//   duration: sum_b L_b sum_t (d_hat - log(ds + 1e-8))^2 / (T * sum_b L_b)
//   pitch   : sum_t c_t sum_b smooth_l1(p_hat, ps) / (B * sum_t c_t),  c_t = #{b : L_b > t};  energy alike.
// out[0..2] = losses; gd/gp/ge = d loss / d prediction (unit upstream gradient).
__global__ __launch_bounds__(256) void variance_losses_kernel(const float* __restrict__ d_hat, const float* __restrict__ p_hat,
                                                              const float* __restrict__ e_hat, const float* __restrict__ ds,
                                                              const float* __restrict__ ps, const float* __restrict__ es,
                                                              const int64_t* __restrict__ x_len, float clip_val,
                                                              float* __restrict__ out, float* __restrict__ gd,
                                                              float* __restrict__ gp, float* __restrict__ ge, int B, int T) {
    __shared__ float scratch[16];
    float sumL = 0.f;
    for (int b = 0; b < B; ++b) sumL += (float)min((int)x_len[b], T);
    const float dn = 1.f / ((float)T * sumL), pn = 1.f / ((float)B * sumL);
    float ld = 0.f, lpi = 0.f, le = 0.f;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < B * T; idx += gridDim.x * blockDim.x) {
        const int b = idx / T, t = idx - b * T;
        const float Lb = (float)min((int)x_len[b], T);
        float ct = 0.f;
        for (int bb = 0; bb < B; ++bb) ct += (int)x_len[bb] > t ? 1.f : 0.f;
        const float dd = d_hat[idx] - logf(ds[idx] + clip_val);
        ld += Lb * dd * dd;
        gd[idx] = 2.f * dd * Lb * dn;
        const float dp = p_hat[idx] - ps[idx], ap = fabsf(dp);
        lpi += ct * (ap < 1.f ? 0.5f * dp * dp : ap - 0.5f);
        gp[idx] = ct * pn * (ap < 1.f ? dp : (dp > 0.f ? 1.f : -1.f));
        const float de = e_hat[idx] - es[idx], ae = fabsf(de);
        le += ct * (ae < 1.f ? 0.5f * de * de : ae - 0.5f);
        ge[idx] = ct * pn * (ae < 1.f ? de : (de > 0.f ? 1.f : -1.f));
    }
    ld = block_sum(ld, scratch);
    lpi = block_sum(lpi, scratch);
    le = block_sum(le, scratch);
    if (threadIdx.x == 0) {
        atomicAdd(out + 0, ld * dn);
        atomicAdd(out + 1, lpi * pn);
        atomicAdd(out + 2, le * pn);
    }
}
extern "C" int osp_variance_losses(const float* d_hat, const float* p_hat, const float* e_hat, const float* ds,
                                   const float* ps, const float* es, const int64_t* x_len, float clip_val, float* out,
                                   float* gd, float* gp, float* ge, int64_t B, int64_t T, hipStream_t stream) {
    OSP_CHECK_ARG(d_hat && p_hat && e_hat && ds && ps && es && x_len && out && gd && gp && ge, "null operand");
    hipMemsetAsync(out, 0, 3 * sizeof(float), stream);
    const int64_t blocks = cdiv(B * T, 256);
    hipLaunchKernelGGL(variance_losses_kernel, dim3((unsigned)(blocks < 64 ? blocks : 64)), dim3(256), 0, stream, d_hat, p_hat,
                       e_hat, ds, ps, es, x_len, clip_val, out, gd, gp, ge, (int)B, (int)T);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// ------------------------------------------------------------------------------------------------ A14
__device__ __forceinline__ float lse2(float a, float b) {
    const float m = fmaxf(a, b);
    return m == -INFINITY ? -INFINITY : m + logf(expf(a - m) + expf(b - m));
}
__device__ __forceinline__ float lse3(float a, float b, float c) {
    const float m = fmaxf(a, fmaxf(b, c));
    return m == -INFINITY ? -INFINITY : m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}

// CTC states s in [0, 2N]: even = blank, odd s = 2k+1 <-> token k (targets 1..N are all distinct, :182).
// State s lives in (round r = s / 64, lane = s % 64).  y[t][blank] = log(blank_prob) - lse_t,
// y[t][k] = lp[t,k] - lse_t with lse_t the per-frame log-sum-exp over {blank, tokens < N} (:183-186).
// grad w.r.t. lp follows torch's ctc_loss backward (eq. 16 of Graves et al., softmax folded in):
//   d loss_b / d lp[t,k] = (exp(y[t,k]) - exp(alpha[t,s] + beta[t,s] + nll - y[t,k])) / (N_b * B)
template <int R>
__global__ __launch_bounds__(64) void forwardsum_ctc_kernel(const float* __restrict__ lp, const int64_t* __restrict__ x_len,
                                                            const int64_t* __restrict__ y_len, float log_blank,
                                                            float* __restrict__ alpha_ws, float* __restrict__ lse_ws,
                                                            float* __restrict__ loss_item, float* __restrict__ grad,
                                                            int B, int Tm, int Nm) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int T = (int)y_len[b], N = (int)x_len[b];
    const int S = 2 * N + 1;
    const float* L = lp + (int64_t)b * Tm * Nm;
    float* G = grad ? grad + (int64_t)b * Tm * Nm : nullptr;
    if (G)
        for (int64_t i = lane; i < (int64_t)Tm * Nm; i += 64) G[i] = 0.f;
    if (T <= 0 || N <= 0) { if (lane == 0) loss_item[b] = 0.f; return; }
    const int SW = R * 64;
    float* AW = alpha_ws + (int64_t)b * Tm * SW;
    float* LW = lse_ws + (int64_t)b * Tm;

    // per-lane state bookkeeping
    bool is_lab[R], live[R];
    int tok[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int s = r * 64 + lane;
        live[r] = s < S;
        is_lab[r] = (s & 1) != 0;
        tok[r] = s >> 1;
    }
    auto frame_lse = [&](int t) -> float {
        float mx = log_blank;
        for (int k = lane; k < N; k += 64) mx = fmaxf(mx, L[(int64_t)t * Nm + k]);
        mx = wave_max(mx);
        float sm = 0.f;
        for (int k = lane; k < N; k += 64) sm += expf(L[(int64_t)t * Nm + k] - mx);
        sm = wave_sum(sm) + expf(log_blank - mx);
        return mx + logf(sm);
    };
    float a[R];
    // ---- alpha pass
    for (int t = 0; t < T; ++t) {
        const float lset = frame_lse(t);
        if (lane == 0) LW[t] = lset;
        float an[R];
        float c1 = -INFINITY, c2a = -INFINITY, c2b = -INFINITY;   // carries: prev round lanes 63 / 62,63
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float y = live[r] ? ((is_lab[r] ? L[(int64_t)t * Nm + tok[r]] : log_blank) - lset) : -INFINITY;
            float v;
            if (t == 0) {
                v = (r == 0 && lane < 2 && live[r]) ? y : -INFINITY;
            } else {
                float p1 = __shfl_up(a[r], 1, 64), p2 = __shfl_up(a[r], 2, 64);
                if (lane == 0) { p1 = c1; p2 = c2a; }
                if (lane == 1) p2 = c2b;
                c1 = __shfl(a[r], 63, 64); c2a = __shfl(a[r], 62, 64); c2b = c1;
                v = is_lab[r] ? lse3(a[r], p1, p2) : lse2(a[r], p1);
                v = live[r] ? v + y : -INFINITY;
            }
            an[r] = v;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) { a[r] = an[r]; AW[(int64_t)t * SW + r * 64 + lane] = an[r]; }
    }
    // nll = -logsumexp(alpha[T-1][2N], alpha[T-1][2N-1])
    float tail = -INFINITY;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int s = r * 64 + lane;
        if (s == S - 1 || s == S - 2) tail = lse2(tail, a[r]);
    }
    float mx = wave_max(tail);
    float ll = mx == -INFINITY ? -INFINITY : mx + logf(wave_sum(tail == -INFINITY ? 0.f : expf(tail - mx)));
    const float nll = -ll;
    const bool inf = !(nll < INFINITY);                               // zero_infinity=True (:192)
    if (lane == 0) loss_item[b] = inf ? 0.f : nll / (float)N;          // reduction='mean' divides by target length
    if (!G || inf) return;
    const float gs = 1.f / ((float)N * (float)B);
    __syncthreads();   // single wave: drains the alpha / lse stores before they are re-read below
    // ---- beta pass + gradient
    float be[R];
    for (int t = T - 1; t >= 0; --t) {
        const float lset = LW[t];
        float bn[R];
        float c1 = -INFINITY, c2a = -INFINITY, c2b = -INFINITY;       // carries: NEXT round lanes 0 / 0,1
#pragma unroll
        for (int r = R - 1; r >= 0; --r) {
            const int s = r * 64 + lane;
            const float y = live[r] ? ((is_lab[r] ? L[(int64_t)t * Nm + tok[r]] : log_blank) - lset) : -INFINITY;
            float v;
            if (t == T - 1) {
                v = (s == S - 1 || s == S - 2) ? y : -INFINITY;
            } else {
                float n1 = __shfl_down(be[r], 1, 64), n2 = __shfl_down(be[r], 2, 64);
                if (lane == 63) { n1 = c1; n2 = c2b; }
                if (lane == 62) n2 = c2a;
                c1 = __shfl(be[r], 0, 64); c2a = c1; c2b = __shfl(be[r], 1, 64);
                // skip s -> s+2 only from a label state to the next (different) label state
                v = (is_lab[r] && s + 2 < S) ? lse3(be[r], n1, n2) : lse2(be[r], n1);
                v = live[r] ? v + y : -INFINITY;
            }
            bn[r] = v;
            if (live[r] && is_lab[r]) {
                const float al = AW[(int64_t)t * SW + s];
                const float occ = expf(al + v + nll - y);
                G[(int64_t)t * Nm + tok[r]] = (expf(y) - occ) * gs;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) be[r] = bn[r];
    }
}


// ---- multi-wave variant (S = 2N+1 <= 1024 states): one thread per CTC state, one workgroup per utterance, the
// alpha / beta columns ping-pong through LDS with one barrier per frame; the per-frame log-sum-exp comes from a
// separate fully parallel kernel.  ~50x less serial work per step than the one-wave kernel above.
__global__ __launch_bounds__(256) void ctc_frame_lse_kernel(const float* __restrict__ lp, const int64_t* __restrict__ x_len,
                                                           const int64_t* __restrict__ y_len, float log_blank,
                                                           float* __restrict__ lse, int B, int Tm, int Nm) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (int64_t)B * Tm) return;
    const int b = (int)(row / Tm), t = (int)(row - (int64_t)b * Tm);
    const int N = (int)x_len[b];
    if (t >= (int)y_len[b]) return;
    const float* L = lp + row * Nm;
    float mx = log_blank;
    for (int k = lane; k < N; k += 64) mx = fmaxf(mx, L[k]);
    mx = wave_max(mx);
    float sm = 0.f;
    for (int k = lane; k < N; k += 64) sm += expf(L[k] - mx);
    sm = wave_sum(sm) + expf(log_blank - mx);
    if (lane == 0) lse[row] = mx + logf(sm);
}

// Branch-free log-sum-exp on the hardware exp2 / log2 (v_exp_f32 / v_log_f32 behind __expf / __logf, ~1 ulp each): the recursions
// below are ~1 600 dependent steps whose length IS the instruction stream of one wave -- the library expf / logf (range reduction,
// denormal paths) plus the -inf branches made a step ~200 instructions.  All-(-inf) inputs: the max is replaced by 0 for the
// differences, exp(-inf) = 0, log(0) = -inf -- no branch.  The error per step is ~1e-7 absolute on log(sum <= 3).
__device__ __forceinline__ float lse2_fast(float a, float b) {
    const float m = fmaxf(a, b), ms = m == -INFINITY ? 0.f : m;
    return m + __logf(__expf(a - ms) + __expf(b - ms));
}
__device__ __forceinline__ float lse3_fast(float a, float b, float c) {
    const float m = fmaxf(a, fmaxf(b, c)), ms = m == -INFINITY ? 0.f : m;
    return m + __logf(__expf(a - ms) + __expf(b - ms) + __expf(c - ms));
}

// Barrier for the LDS column exchange of the recursions below.  __syncthreads() carries a workgroup fence that the compiler
// lowers to s_waitcnt vmcnt(0): every step then waited for the global store of its alpha row (and drained the emission prefetch
// ring) -- ~1 us per step, 1.7-1.8 ms for T = 800, which is what the kernel cost until this was found (round 3).  The exchange
// only needs the LDS operations of this wave complete (lgkmcnt(0)) and the barrier itself; global stores stay in flight.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__global__ void forwardsum_ctc_mw_kernel(const float* __restrict__ lp, const int64_t* __restrict__ x_len,
                                         const int64_t* __restrict__ y_len, float log_blank, float* __restrict__ alpha_ws,
                                         const float* __restrict__ lse_ws, float* __restrict__ loss_item,
                                         float* __restrict__ grad, int B, int Tm, int Nm, int SW) {
    extern __shared__ float sh[];                      // [2][SW + 2]  (two leading -inf guard cells per column)
    __shared__ float s_nll;
    const int b = blockIdx.x, s = threadIdx.x;
    const int T = (int)y_len[b], N = (int)x_len[b];
    const int S = 2 * N + 1;
    const float* L = lp + (int64_t)b * Tm * Nm;
    float* G = grad ? grad + (int64_t)b * Tm * Nm : nullptr;
    if (G)
        for (int64_t i = s; i < (int64_t)Tm * Nm; i += blockDim.x) G[i] = 0.f;
    if (T <= 0 || N <= 0) { if (s == 0) loss_item[b] = 0.f; return; }
    float* AW = alpha_ws + (int64_t)b * Tm * SW;
    const float* LW = lse_ws + (int64_t)b * Tm;
    const bool live = s < S, lab = (s & 1) != 0;
    const int tok = s >> 1;
    float* col[2] = {sh + 2, sh + (SW + 2) + 2};
    if (s < 2) { sh[s] = -INFINITY; sh[(SW + 2) + s] = -INFINITY; }
    // Both recursions are T dependent steps (LDS exchange + barrier + three expf / one logf).  Round 2 loaded each step's
    // emission -- and in the second pass the stored alpha -- inside the step: a memory latency per step.  A register RING refilled
    // inside the steps did not help: behind the divergent -inf branches of lse3 the compiler waits with vmcnt(0) in every step,
    // which drains the ring and the alpha store.  So the values are fetched a CHUNK ahead: the loads of chunk c + 1 are issued
    // before the CPD steps of chunk c, which touch no load result; the only vmcnt wait sits at the chunk boundary.
    constexpr int CPD = 8;
    // the loads of a chunk are UNCONDITIONAL (clamped indices) and nothing is computed from them until the step that uses them:
    // a conditional load is an exec-masked branch and arithmetic right behind a load is a wait -- either serialises the chunk's
    // loads, one memory latency each
    const int tokc = (live && lab) ? tok : 0;
    struct Em { float l, w; };
    auto fetch = [&](int t) -> Em { const int tc = t < 0 ? 0 : (t >= T ? T - 1 : t); Em e; e.l = L[(int64_t)tc * Nm + tokc]; e.w = LW[tc]; return e; };
    auto emis_of = [&](const Em& e) -> float { return live ? ((lab ? e.l : log_blank) - e.w) : -INFINITY; };
    // ---- alpha
    float a = -INFINITY;
    Em yc[CPD], yn[CPD];
#pragma unroll
    for (int k = 0; k < CPD; ++k) yc[k] = fetch(k);
    for (int t0 = 0; t0 < T; t0 += CPD) {
#pragma unroll
        for (int k = 0; k < CPD; ++k) yn[k] = fetch(t0 + CPD + k);
#pragma unroll
        for (int k = 0; k < CPD; ++k) {
            const int t = t0 + k;
            if (t >= T) break;                                          // block-uniform
            const float y = emis_of(yc[k]);
            float v;
            if (t == 0) v = (s < 2 && live) ? y : -INFINITY;
            else {
                const float* pc = col[(t - 1) & 1];
                v = lse3_fast(a, pc[s - 1], lab ? pc[s - 2] : -INFINITY);          // (a third term of -inf contributes exp(-inf) = 0)
                v = live ? v + y : -INFINITY;
            }
            a = v;
            col[t & 1][s] = v;
            AW[(int64_t)t * SW + s] = v;
            lds_barrier();
        }
#pragma unroll
        for (int k = 0; k < CPD; ++k) yc[k] = yn[k];
    }
    if (s == 0) {
        const float* pc = col[(T - 1) & 1];
        const float ll = lse2(pc[S - 1], S >= 2 ? pc[S - 2] : -INFINITY);
        s_nll = -ll;
        const bool inf = !(-ll < INFINITY);
        loss_item[b] = inf ? 0.f : -ll / (float)N;
    }
    __syncthreads();
    const float nll = s_nll;
    if (!G || !(nll < INFINITY)) return;
    const float gs = 1.f / ((float)N * (float)B);
    // ---- beta + gradient (columns stored with two trailing guard cells: index s+1, s+2 may run past S)
    float be = -INFINITY;
    float ac[CPD], an[CPD];
    auto alpha_at = [&](int t) -> float { return AW[(int64_t)(t < 0 ? 0 : t) * SW + s]; };        // (every thread owns column s of its rows)
#pragma unroll
    for (int k = 0; k < CPD; ++k) { const int t = T - 1 - k; yc[k] = fetch(t); ac[k] = alpha_at(t); }
    for (int t0 = T - 1; t0 >= 0; t0 -= CPD) {
#pragma unroll
        for (int k = 0; k < CPD; ++k) { const int t = t0 - CPD - k; yn[k] = fetch(t); an[k] = alpha_at(t); }
#pragma unroll
        for (int k = 0; k < CPD; ++k) {
            const int t = t0 - k;
            if (t < 0) break;                                           // block-uniform
            const float y = emis_of(yc[k]), al = ac[k];
            float v;
            if (t == T - 1) v = (s == S - 1 || s == S - 2) ? y : -INFINITY;
            else {
                const float* pc = col[(t + 1) & 1];
                const float n1 = s + 1 < S ? pc[s + 1] : -INFINITY, n2 = s + 2 < S ? pc[s + 2] : -INFINITY;
                v = lse3_fast(be, n1, (lab && s + 2 < S) ? n2 : -INFINITY);
                v = live ? v + y : -INFINITY;
            }
            be = v;
            col[t & 1][s] = v;
            if (live && lab) {
                const float occ = expf(al + v + nll - y);
                G[(int64_t)t * Nm + tok] = (expf(y) - occ) * gs;
            }
            lds_barrier();
        }
#pragma unroll
        for (int k = 0; k < CPD; ++k) { yc[k] = yn[k]; ac[k] = an[k]; }
    }
}

extern "C" int64_t osp_forwardsum_ctc_workspace_floats(int64_t B, int64_t Tm, int64_t Nm) {
    int64_t R = (2 * Nm + 1 + 63) / 64, Rp = 1;
    while (Rp < R) Rp *= 2;
    return B * Tm * Rp * 64 + B * Tm;
}

// loss_item[b] = ctc_b / N_b (0 when infinite); grad (optional) = d (sum_b loss_item[b] / B) / d lp.
extern "C" int osp_forwardsum_ctc(const float* lp, const int64_t* x_len, const int64_t* y_len, float blank_logprob,
                                  float* workspace, float* loss_item, float* grad, int64_t B, int64_t Tm, int64_t Nm,
                                  hipStream_t stream) {
    OSP_CHECK_ARG(lp && x_len && y_len && workspace && loss_item, "null operand");
    OSP_CHECK_ARG(Nm <= 2047, "at most 2047 tokens");
    int64_t R = (2 * Nm + 1 + 63) / 64;
    int Rp = 1;
    while (Rp < R) Rp *= 2;
    float* alpha_ws = workspace;
    float* lse_ws = workspace + B * Tm * Rp * 64;
    if (R * 64 <= 1024) {                               // multi-wave path: one thread per state
        const int SW = (int)(R * 64);
        hipLaunchKernelGGL(ctc_frame_lse_kernel, dim3((unsigned)cdiv(B * Tm, 4)), dim3(256), 0, stream, lp, x_len, y_len,
                           blank_logprob, lse_ws, (int)B, (int)Tm, (int)Nm);
        hipLaunchKernelGGL(forwardsum_ctc_mw_kernel, dim3((unsigned)B), dim3((unsigned)SW), sizeof(float) * 2 * (SW + 2), stream,
                           lp, x_len, y_len, blank_logprob, alpha_ws, lse_ws, loss_item, grad, (int)B, (int)Tm, (int)Nm, SW);
        OSP_LAUNCH_CHECK();
        return OSP_OK;
    }
#define L(RR) hipLaunchKernelGGL((forwardsum_ctc_kernel<RR>), dim3((unsigned)B), dim3(64), 0, stream, lp, x_len, y_len, blank_logprob, alpha_ws, lse_ws, loss_item, grad, (int)B, (int)Tm, (int)Nm)
    switch (Rp) {
        case 1: L(1); break; case 2: L(2); break; case 4: L(4); break; case 8: L(8); break;
        case 16: L(16); break; case 32: L(32); break; default: L(64);
    }
#undef L
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
