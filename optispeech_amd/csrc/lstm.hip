// Single-layer LSTM recurrence (nn.LSTM(dim, dim, num_layers=1, batch_first=True) of the LeanSpeech backbone, reference
// generator/modules/leanspeech.py:49-63) as persistent cooperative kernels.
//
// The input projection X W_ih^T + b_ih + b_hh of ALL steps is one GEMM (conv-GEMM family, done by the caller); what is
// left is the dependent chain  gates_t = Gx_t + h_{t-1} W_hh^T  ->  c_t, h_t  over T steps of a (B x H) x (H x 4H) product
// that is far too small to fill a GPU and far too sequential to launch per step (T = 800 mel frames).  MI355X mapping:
//   * one utterance = a team of NW = H / 32 workgroups, workgroup w owns hidden units [32w, 32w + 32): the 4 x 32 gate rows
//     of W_hh that produce them (128 x H floats) live in REGISTERS for the whole sequence (H / 2 VGPRs per thread at H = 256:
//     no weight is ever re-read), the cell state c of unit j in a register of thread j;
//   * per step a team exchanges only the H floats of h_t through a double-buffered global line + one release / acquire flag
//     per workgroup (agent scope); teams never talk to each other.  Block ids are chosen so that a team sits on ONE XCD
//     (ids congruent mod 8 share an XCD): the exchange then stays inside that XCD's L2;
//   * B x NW <= 256 workgroups per launch (the host wrapper chunks the batch) so that every workgroup of a team is resident
//     -- required for the spin-waits to make progress.
// Backward runs the same team structure in reverse time with the transposed slice (all 4H rows x 32 owned columns of W_hh
// in registers): dh_{t-1} += dG_t W_hh needs every gate gradient of the utterance, so the per-step exchange is the 4H floats
// of dG_t (which is also the kernel's output: dW_ih, dW_hh, db and dX are GEMMs / column sums over dG afterwards).
//
//   i, f, o = sigmoid(.), g = tanh(.);  c_t = f c_{t-1} + i g;  h_t = o tanh(c_t)          (gate row order of torch: i, f, g, o)
#include "osp_common.h"

#define LSTM_U 32          // hidden units per workgroup

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// One acquire / release FENCE per workgroup and step, not one per thread or per poll: an agent-scope fence is an L2 write-back /
// invalidate on this part (21.8 us per step with per-thread fences and acquire polls, measured).  The other threads' stores /
// loads are ordered against the fence thread's by the workgroup barriers around it (scope inclusion).
__device__ __forceinline__ void team_wait(const int* flags, int nw, int need) {
    if ((int)threadIdx.x < nw) {
        while (__hip_atomic_load(const_cast<int*>(flags) + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) __builtin_amdgcn_s_sleep(1);
        if (threadIdx.x == 0) __atomic_thread_fence(__ATOMIC_ACQUIRE);      // wave 0: every polled flag has been seen by now
    }
    __syncthreads();
}
__device__ __forceinline__ void team_post(int* flag, int value) {
    __syncthreads();                                   // every thread's stores of the step are issued ...
    if (threadIdx.x == 0) {
        __threadfence();                               // ... and made visible device-wide before the flag moves
        __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ float ld_shared_line(const float* p) {
    return __hip_atomic_load(const_cast<float*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // bypasses the CU's L1 (the line is rewritten every other step)
}
// workgroup -> (utterance, team member): a team's members have block ids congruent mod 8 (same XCD)
__device__ __forceinline__ void team_ids(int nw, int B, int& b, int& w) {
    const int id = blockIdx.x, x = id & 7, j = id >> 3;
    // utterances are dealt to XCDs round-robin: utterance u -> XCD u % 8, slot u / 8; slot s of an XCD occupies j in [s*nw, (s+1)*nw)
    const int s = j / nw;
    w = j - s * nw;
    b = s * 8 + x;
    (void)B;
}

// gx (B, T, 4H): input projection + both biases; whh (4H, H); h0 / c0 = 0.
// hs (B, T, H) = h_t;  gates (B, T, 4H) post-activation i, f, g, o and cs (B, T, H) = c_t when `save`.
// hx (B, 2, H) exchange lines, flags (B, NW) zeroed by the caller.
template <int H>
__global__ __launch_bounds__(256) void lstm_fwd_kernel(const float* __restrict__ gx, const float* __restrict__ whh, float* __restrict__ hs,
                                                       float* __restrict__ gates, float* __restrict__ cs, float* hx, int* flags, int B, int T) {
    constexpr int NW = H / LSTM_U, HALF = H / 2;
    __shared__ float h_l[H];
    __shared__ float pre_l[4 * LSTM_U];
    int b, w;
    team_ids(NW, B, b, w);
    if (b >= B) return;                                          // whole teams only: no member of a live team exits
    const int tid = threadIdx.x, rr = tid >> 1, half = tid & 1;   // rr: owned gate row (gate = rr / 32, unit = rr % 32), half of the k range
    const int gate = rr >> 5, unit = rr & 31;
    const int row = gate * H + w * LSTM_U + unit;                 // row of W_hh / column of gx
    float wr[HALF];
#pragma unroll
    for (int k = 0; k < HALF; ++k) wr[k] = whh[(int64_t)row * H + half * HALF + k];
    float c = 0.f;                                                // cell state of unit `tid` (threads < 32)
    int* myflag = flags + b * NW + w;
    const int* team = flags + b * NW;
    float* hxb = hx + (int64_t)b * 2 * H;
    for (int t = 0; t < T; ++t) {
        if (t > 0) {
            team_wait(team, NW, t);                               // every member has published h_{t-1}
            for (int k = tid; k < H; k += 256) h_l[k] = ld_shared_line(hxb + ((t - 1) & 1) * H + k);
        } else {
            for (int k = tid; k < H; k += 256) h_l[k] = 0.f;
        }
        const float gxv = gx[((int64_t)b * T + t) * 4 * H + row];
        __syncthreads();
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < HALF; ++k) acc = fmaf(wr[k], h_l[half * HALF + k], acc);
        acc += __shfl_xor(acc, 1);
        if (half == 0) {
            const float pre = gxv + acc;
            pre_l[rr] = gate == 2 ? tanhf(pre) : sigmoidf_(pre);
        }
        __syncthreads();
        if (tid < LSTM_U) {
            const float ig = pre_l[tid], fg = pre_l[LSTM_U + tid], gg = pre_l[2 * LSTM_U + tid], og = pre_l[3 * LSTM_U + tid];
            c = fmaf(fg, c, ig * gg);
            const float h = og * tanhf(c);
            const int64_t bt = (int64_t)b * T + t;
            const int u = w * LSTM_U + tid;
            hs[bt * H + u] = h;
            hxb[(t & 1) * H + u] = h;
            if (cs) {
                cs[bt * H + u] = c;
                gates[bt * 4 * H + u] = ig; gates[bt * 4 * H + H + u] = fg; gates[bt * 4 * H + 2 * H + u] = gg; gates[bt * 4 * H + 3 * H + u] = og;
            }
        }
        team_post(myflag, t + 1);
    }
}

// dhs (B, T, H): gradient w.r.t. every h_t from above;  gates / cs: saved by the forward;  whh (4H, H).
// dg (B, T, 4H): gradient w.r.t. the gate PRE-activations (= w.r.t. gx); it doubles as the team's exchange buffer.
template <int H>
__global__ __launch_bounds__(256) void lstm_bwd_kernel(const float* __restrict__ dhs, const float* __restrict__ gates, const float* __restrict__ cs,
                                                       const float* __restrict__ whh, float* dg, int* flags, int B, int T) {
    constexpr int NW = H / LSTM_U, G4 = 4 * H, PART = G4 / 8;        // thread = (owned column k = tid / 8, eighth of the 4H rows)
    __shared__ float dg_l[G4];
    __shared__ float dh_l[LSTM_U];
    int b, w;
    team_ids(NW, B, b, w);
    if (b >= B) return;
    const int tid = threadIdx.x, kk = tid >> 3, part = tid & 7;
    const int col = w * LSTM_U + kk;                                 // hidden unit whose dh this thread helps to reduce
    float wr[PART];
#pragma unroll
    for (int r = 0; r < PART; ++r) wr[r] = whh[(int64_t)(part * PART + r) * H + col];
    float dc_next = 0.f;                                             // dc_{t+1} * f_{t+1} of unit `tid` (threads < 32)
    int* myflag = flags + b * NW + w;
    const int* team = flags + b * NW;
    for (int s = 0; s < T; ++s) {
        const int t = T - 1 - s;
        const int64_t bt = (int64_t)b * T + t;
        float rec = 0.f;
        if (s > 0) {
            team_wait(team, NW, s);                                  // every member has published its rows of dG_{t+1}
            for (int r = tid; r < G4; r += 256) dg_l[r] = ld_shared_line(dg + (bt + 1) * G4 + r);
            __syncthreads();
#pragma unroll
            for (int r = 0; r < PART; ++r) rec = fmaf(wr[r], dg_l[part * PART + r], rec);
            rec += __shfl_xor(rec, 1); rec += __shfl_xor(rec, 2); rec += __shfl_xor(rec, 4);
        }
        if (part == 0) dh_l[kk] = rec;
        __syncthreads();
        if (tid < LSTM_U) {
            const int u = w * LSTM_U + tid;
            const float dh = dhs[bt * H + u] + dh_l[tid];
            const float ig = gates[bt * G4 + u], fg = gates[bt * G4 + H + u], gg = gates[bt * G4 + 2 * H + u], og = gates[bt * G4 + 3 * H + u];
            const float c = cs[bt * H + u], cprev = t > 0 ? cs[(bt - 1) * H + u] : 0.f;
            const float tc = tanhf(c);
            const float dc = fmaf(dh * og, 1.f - tc * tc, dc_next);
            dg[bt * G4 + u] = dc * gg * ig * (1.f - ig);
            dg[bt * G4 + H + u] = dc * cprev * fg * (1.f - fg);
            dg[bt * G4 + 2 * H + u] = dc * ig * (1.f - gg * gg);
            dg[bt * G4 + 3 * H + u] = dh * tc * og * (1.f - og);
            dc_next = dc * fg;
        }
        team_post(myflag, s + 1);
    }
}

static int lstm_check(int64_t B, int64_t T, int64_t H) {
    OSP_CHECK_ARG(B > 0 && T > 0, "bad shape");
    OSP_CHECK_ARG(H == 64 || H == 128 || H == 256, "hidden size must be 64, 128 or 256");
    OSP_CHECK_ARG(B * (H / LSTM_U) <= 256 && B % 8 == 0, "at most 256 workgroups per launch, whole XCD rounds (B % 8 == 0): chunk / pad the batch");
    return OSP_OK;
}

extern "C" int osp_lstm_fwd(const float* gx, const float* whh, float* hs, float* gates, float* cs, float* hx, int32_t* flags, int64_t B,
                            int64_t T, int64_t H, hipStream_t stream) {
    OSP_CHECK_ARG(gx && whh && hs && hx && flags, "null operand");
    OSP_CHECK_ARG((gates == nullptr) == (cs == nullptr), "gates / cs are saved together");
    if (int rc = lstm_check(B, T, H)) return rc;
    const dim3 grid((unsigned)(B * (H / LSTM_U)));
#define L(H_) hipLaunchKernelGGL((lstm_fwd_kernel<H_>), grid, dim3(256), 0, stream, gx, whh, hs, gates, cs, hx, flags, (int)B, (int)T)
    if (H == 256) L(256); else if (H == 128) L(128); else L(64);
#undef L
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

extern "C" int osp_lstm_bwd(const float* dhs, const float* gates, const float* cs, const float* whh, float* dg, int32_t* flags, int64_t B,
                            int64_t T, int64_t H, hipStream_t stream) {
    OSP_CHECK_ARG(dhs && gates && cs && whh && dg && flags, "null operand");
    if (int rc = lstm_check(B, T, H)) return rc;
    const dim3 grid((unsigned)(B * (H / LSTM_U)));
#define L(H_) hipLaunchKernelGGL((lstm_bwd_kernel<H_>), grid, dim3(256), 0, stream, dhs, gates, cs, whh, dg, flags, (int)B, (int)T)
    if (H == 256) L(256); else if (H == 128) L(128); else L(64);
#undef L
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
