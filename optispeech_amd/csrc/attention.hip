// Masked softmax (+ dropout) of attention scores and its backward: the element-wise half of MultiHeadedAttention
// (generator/modules/_transformer/attention.py:80-98, :120-125).  The two batched GEMMs on either side (Q K^T, P V and
// their gradients) run on the conv-GEMM / wgrad kernels with a batch stride (optispeech_amd/ops.py: AttentionFn).
//   forward : P[z, i, j] = softmax_j(scale * S[z, i, j]) over the valid keys j < klen[z / H], 0 elsewhere   (in place)
//             Pd = dropout(P)  (Philox, regenerated in backward)                                             (optional)
//   backward: dS[z, i, j] = scale * P * (dP - sum_j P dP),  dP = dPd * keep / (1 - p)                        (in place in dPd)
// One wavefront per score row, 16 columns per lane (T2 <= 1024).  HBM-bound: 8 (+4) bytes per score forward, 12 backward.
#include "osp_common.h"

#define ATT_MAXC 16

__global__ __launch_bounds__(256) void attn_softmax_fwd_kernel(float* __restrict__ S, float* __restrict__ Pd,
                                                               const int64_t* __restrict__ klen, int H, int T1, int T2, int64_t rows,
                                                               float scale, float drop_p, uint64_t seed, const int64_t* __restrict__ seed_dev, uint32_t stream_id) {
    if (seed_dev) seed += (uint64_t)*seed_dev;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int z = (int)(row / T1), b = z / H;
    const int kl = (int)min((int64_t)T2, klen[b]);
    float* s = S + row * T2;
    float v[ATT_MAXC];
    float mx = -3.0e38f;
#pragma unroll
    for (int q = 0; q < ATT_MAXC; ++q) {
        const int j = lane + 64 * q;
        v[q] = (j < kl) ? s[j] * scale : -3.0e38f;
        mx = fmaxf(mx, v[q]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < ATT_MAXC; ++q) {
        const int j = lane + 64 * q;
        v[q] = (j < kl) ? __expf(v[q] - mx) : 0.f;
        sum += v[q];
    }
    sum = wave_sum(sum);
    const float inv = sum > 0.f ? 1.f / sum : 0.f;                   // no valid key: all-zero row (masked_fill(mask, 0))
    const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
#pragma unroll
    for (int q = 0; q < ATT_MAXC; ++q) {
        const int j = lane + 64 * q;
        if (j < T2) {
            const float pr = v[q] * inv;
            s[j] = pr;
            if (Pd) Pd[row * T2 + j] = drop_p > 0.f ? pr * dropout_factor(seed, stream_id, (uint64_t)(row * T2 + j), drop_p) : pr;
        }
    }
    (void)keep_scale;
}

__global__ __launch_bounds__(256) void attn_softmax_bwd_kernel(const float* __restrict__ P, float* __restrict__ dPd, int T2, int64_t rows,
                                                               float scale, float drop_p, uint64_t seed, const int64_t* __restrict__ seed_dev, uint32_t stream_id) {
    if (seed_dev) seed += (uint64_t)*seed_dev;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* p = P + row * T2;
    float* g = dPd + row * T2;
    float pv[ATT_MAXC], gv[ATT_MAXC];
    float dot = 0.f;
#pragma unroll
    for (int q = 0; q < ATT_MAXC; ++q) {
        const int j = lane + 64 * q;
        pv[q] = 0.f; gv[q] = 0.f;
        if (j < T2) {
            pv[q] = p[j];
            gv[q] = g[j];
            if (drop_p > 0.f) gv[q] *= dropout_factor(seed, stream_id, (uint64_t)(row * T2 + j), drop_p);
            dot += pv[q] * gv[q];
        }
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int q = 0; q < ATT_MAXC; ++q) {
        const int j = lane + 64 * q;
        if (j < T2) g[j] = scale * pv[q] * (gv[q] - dot);
    }
}

// S: (Z * T1, T2) f32, Z = B * H batch-major; klen: (B) int64 valid key counts.  Pd may be null (no dropout copy).
extern "C" int osp_attn_softmax_fwd(float* S, float* Pd, const int64_t* klen, int64_t B, int64_t H, int64_t T1, int64_t T2, float scale,
                                    float drop_p, int64_t seed, const int64_t* seed_dev, int64_t stream_id, hipStream_t stream) {
    OSP_CHECK_ARG(S && klen && B > 0 && H > 0 && T1 > 0 && T2 > 0 && T2 <= 64 * ATT_MAXC, "bad attention shape (T2 <= 1024)");
    OSP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || Pd), "dropout needs a second output buffer");
    const int64_t rows = B * H * T1;
    hipLaunchKernelGGL(attn_softmax_fwd_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, stream, S, Pd, klen, (int)H, (int)T1, (int)T2,
                       rows, scale, drop_p, (uint64_t)seed, seed_dev, (uint32_t)stream_id);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

extern "C" int osp_attn_softmax_bwd(const float* P, float* dPd, int64_t rows, int64_t T2, float scale, float drop_p, int64_t seed,
                                    const int64_t* seed_dev, int64_t stream_id, hipStream_t stream) {
    OSP_CHECK_ARG(P && dPd && rows > 0 && T2 > 0 && T2 <= 64 * ATT_MAXC, "bad attention shape (T2 <= 1024)");
    hipLaunchKernelGGL(attn_softmax_bwd_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, stream, P, dPd, (int)T2, rows, scale, drop_p,
                       (uint64_t)seed, seed_dev, (uint32_t)stream_id);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// ------------------------------------------------------------------------------------------------ fused (flash-style) forward
// O = softmax(scale * Q K^T over the valid keys) V per (utterance, head) without ever writing the (T x T) scores: the no-grad /
// inference path of MultiHeadedAttention (the decode of the Transformer variant, BASELINE configs[3]: 164 MB of probabilities per
// decoder layer at B = 32, T = 800 in the unfused path).  Training keeps the unfused kernels above (attention dropout and the saved
// probabilities of the backward).
//
// q, k, v: (B, T, H * DK) f32 as the linear layers leave them (head h = channels [h*DK, (h+1)*DK)); o: same layout.
// One workgroup = 4 waves = 128 queries of one (b, h); a wave owns 32 queries.  Keys / values go through LDS in tiles of 32
// (converted to bf16 while staging; rows padded to DK + 8).  Per tile and wave:
//   S^T (32 keys x 32 queries) = K_tile Q^T        v_mfma_f32_32x32x16_bf16, DK / 16 steps: lane = query column, 16 key rows
//   online softmax per QUERY = per lane: the running max / sum need no cross-lane traffic except the lane's partner (l ^ 32)
//   O (32 queries x DK) += P V_tile                P^T's accumulator layout IS the A-operand layout of P when the 16 keys of a
//                                                  k-step are taken in the accumulator's row order (rows 4h..4h+3, 8+4h..8+4h+3):
//                                                  the B operand (V) is read with ds_read_b64_tr_b16 from exactly those rows
// The rescale factor exp(m_old - m_new) and the final 1 / l are per query = per ROW of O's accumulators: they go through a
// 32-float LDS line per wave (lane q writes, every lane reads its 16 rows).
typedef short s16x4_a __attribute__((ext_vector_type(4)));
typedef float f32x16_a __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_a __attribute__((ext_vector_type(8)));

template <int DK>
__global__ __launch_bounds__(256) void attn_fused_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                             const int64_t* __restrict__ klen, float* __restrict__ o, int H, int T, float scale) {
    constexpr int LD = DK + 8;                                   // bf16 elements per LDS row
    constexpr int DS = DK / 16, DB = DK / 32;
    __shared__ __attribute__((aligned(16))) unsigned short k_l[32 * LD];
    __shared__ __attribute__((aligned(16))) unsigned short v_l[32 * LD];
    __shared__ float fac_l[4][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int z = blockIdx.y, b = z / H, hh = z - b * H;
    const int C = H * DK;
    const int kl = (int)min((int64_t)T, klen[b]);
    const int l31 = lane & 31, half = lane >> 5;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int64_t base = (int64_t)b * T * C + (int64_t)hh * DK;
    // Q fragments (B operand of S^T): lane (query l31, half) holds d = 16 s + 8 half .. + 7 for every step s
    bf16x8_a qf[DS];
    {
        const int qi = q0 + l31;
        const float* qp = q + base + (int64_t)(qi < T ? qi : 0) * C;
#pragma unroll
        for (int s = 0; s < DS; ++s) {
            const float4 a = *reinterpret_cast<const float4*>(qp + 16 * s + 8 * half);
            const float4 c = *reinterpret_cast<const float4*>(qp + 16 * s + 8 * half + 4);
            bf16x8_a f;
            f[0] = (__bf16)a.x; f[1] = (__bf16)a.y; f[2] = (__bf16)a.z; f[3] = (__bf16)a.w;
            f[4] = (__bf16)c.x; f[5] = (__bf16)c.y; f[6] = (__bf16)c.z; f[7] = (__bf16)c.w;
            qf[s] = f;
        }
    }
    f32x16_a oacc[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[d][i] = 0.f;
    float m_run = -3.0e38f, l_run = 0.f;                         // of query (q0 + l31); both halves keep identical copies
    const int ntiles = (kl + 31) / 32;
    // K / V rows of the NEXT tile are requested into registers before this tile's MFMAs (round 3, as in attention_train.hip): a
    // tile's global latency flies under compute instead of stalling the workgroup at the staging barrier
    constexpr int NR = 32 * (DK / 4) / 256;
    float4 kreg[NR], vreg[NR];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int it = 0; it < NR; ++it) {
            const int i = tid + it * 256, r = i / (DK / 4), c4 = i - r * (DK / 4), kr = k0 + r;
            kreg[it] = make_float4(0.f, 0.f, 0.f, 0.f); vreg[it] = kreg[it];
            if (kr < kl) {
                kreg[it] = *reinterpret_cast<const float4*>(k + base + (int64_t)kr * C + 4 * c4);
                vreg[it] = *reinterpret_cast<const float4*>(v + base + (int64_t)kr * C + 4 * c4);
            }
        }
    };
    if (ntiles > 0) fetch(0);
    for (int tile = 0; tile < ntiles; ++tile) {
        const int k0 = tile * 32;
        __syncthreads();                                          // the previous tile's reads are done
        // stage K and V rows k0 .. k0 + 31 (zero rows beyond the valid keys), f32 -> bf16
#pragma unroll
        for (int it = 0; it < NR; ++it) {
            const int i = tid + it * 256, r = i / (DK / 4), c4 = i - r * (DK / 4);
            const float4 kv = kreg[it], vv = vreg[it];
            typedef __bf16 b2 __attribute__((ext_vector_type(2)));
            b2 k01, k23, v01, v23;
            k01[0] = (__bf16)kv.x; k01[1] = (__bf16)kv.y; k23[0] = (__bf16)kv.z; k23[1] = (__bf16)kv.w;
            v01[0] = (__bf16)vv.x; v01[1] = (__bf16)vv.y; v23[0] = (__bf16)vv.z; v23[1] = (__bf16)vv.w;
            *reinterpret_cast<uint2*>(k_l + r * LD + 4 * c4) = make_uint2(__builtin_bit_cast(unsigned, k01), __builtin_bit_cast(unsigned, k23));
            *reinterpret_cast<uint2*>(v_l + r * LD + 4 * c4) = make_uint2(__builtin_bit_cast(unsigned, v01), __builtin_bit_cast(unsigned, v23));
        }
        __syncthreads();
        if (tile + 1 < ntiles) fetch(k0 + 32);
        // S^T tile: rows = keys (A operand: lane (key l31, half) holds d = 16 s + 8 half ..), columns = this wave's queries
        f32x16_a st;
#pragma unroll
        for (int i = 0; i < 16; ++i) st[i] = 0.f;
#pragma unroll
        for (int s = 0; s < DS; ++s) {
            const bf16x8_a kf = *reinterpret_cast<const bf16x8_a*>(k_l + l31 * LD + 16 * s + 8 * half);
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st, 0, 0, 0);
        }
        // accumulator element i of lane (query l31, half): key row (i / 4) * 8 + 4 * half + i % 4
        float mx = -3.0e38f;
        float sc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kr = k0 + (i >> 2) * 8 + 4 * half + (i & 3);
            sc[i] = kr < kl ? st[i] * scale : -3.0e38f;
            mx = fmaxf(mx, sc[i]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);               // 0 on the first tile (m_run = -3e38)
        float ps = 0.f;
        bf16x8_a pf[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float p = sc[i] > -1.0e38f ? __expf(sc[i] - m_new) : 0.f;
            ps += p;
            pf[i >> 3][i & 7] = (__bf16)p;
        }
        ps += __shfl_xor(ps, 32);
        l_run = l_run * alpha + ps;
        m_run = m_new;
        // rescale O's rows: factor of query r lives in lane r (either half)
        if (half == 0) fac_l[wave][l31] = alpha;
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float fr[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 f4 = *reinterpret_cast<const float4*>(&fac_l[wave][g * 8 + 4 * half]);
            fr[4 * g] = f4.x; fr[4 * g + 1] = f4.y; fr[4 * g + 2] = f4.z; fr[4 * g + 3] = f4.w;
        }
        __builtin_amdgcn_wave_barrier();
        // O (32 queries x DK) = alpha * O + P V: A = P (lane (query, half): keys in accumulator order), B = V rows read transposed
        const int r16 = lane & 15, g16 = (lane >> 4) & 1;
#pragma unroll
        for (int d = 0; d < DB; ++d) {
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[d][i] *= fr[i];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                // B operand lane (column n = d*32 + (lane & 31), half): keys 16 s + 4 half + 0..3 and 16 s + 8 + 4 half + 0..3
                const int col = d * 32 + 16 * g16 + 4 * (r16 & 3);
                const unsigned short* a0 = v_l + (16 * s + 4 * half + (r16 >> 2)) * LD + col;
                const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned short*)a0;
                s16x4_a lo, hi;
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(addr) : "memory");
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(8 * LD * 2) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi) : : "memory");
                union { struct { s16x4_a l, h; } s2; bf16x8_a vv; } u;
                u.s2.l = lo; u.s2.h = hi;
                oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[s], u.vv, oacc[d], 0, 0, 0);
            }
        }
    }
    // normalise and store: O's accumulator element i of lane (column d, half) is query row (i / 4) * 8 + 4 * half + i % 4
    if (half == 0) fac_l[wave][l31] = l_run > 0.f ? 1.f / l_run : 0.f;       // no valid key: zero row (masked_fill(mask, 0))
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = (i >> 2) * 8 + 4 * half + (i & 3);
        const int qi = q0 + r;
        if (qi < T) {
            const float inv = fac_l[wave][r];
#pragma unroll
            for (int d = 0; d < DB; ++d) o[base + (int64_t)qi * C + d * 32 + l31] = oacc[d][i] * inv;
        }
    }
}

extern "C" int osp_attn_fused_fwd(const float* q, const float* k, const float* v, const int64_t* klen, float* o, int64_t B, int64_t H,
                                  int64_t T, int64_t DK, float scale, hipStream_t stream) {
    OSP_CHECK_ARG(q && k && v && klen && o && B > 0 && H > 0 && T > 0, "bad args");
    OSP_CHECK_ARG(DK == 32 || DK == 64 || DK == 128, "head width must be 32, 64 or 128");
    const dim3 grid((unsigned)cdiv(T, 128), (unsigned)(B * H));
#define L(D_) hipLaunchKernelGGL((attn_fused_fwd_kernel<D_>), grid, dim3(256), 0, stream, q, k, v, klen, o, (int)H, (int)T, scale)
    if (DK == 128) L(128); else if (DK == 64) L(64); else L(32);
#undef L
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
