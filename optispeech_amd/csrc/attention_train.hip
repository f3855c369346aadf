// Fused (flash-style) attention for the TRAINING path of MultiHeadedAttention (generator/modules/_transformer/attention.py:80-98,
// :120-125): forward with attention dropout that keeps only the per-row log-sum-exp, and a backward that recomputes the
// probabilities tile by tile -- the (B*H, T, T) scores / probabilities / their gradients (164 MB per decoder layer at B = 32,
// T = 800, three of them alive in the unfused backward) are never written.  bf16 MFMA operands, f32 accumulation: the
// performance mode's arithmetic (the f32 parity mode keeps the unfused exact-f32 kernels of attention.hip).
//
//   forward : S = Q K^T,  P = softmax_j(scale S) over the valid keys j < klen[b] (0 elsewhere; an all-masked row is 0),
//             Pd = P * keep / (1 - p)  (Philox, element index (z T + i) T + j: the SAME mask as osp_attn_softmax_fwd draws),
//             O = Pd V;   lse[z, i] = max + log(sum)   (natural log of the scaled scores; +3e38 for an all-masked row)
//   backward: D[z, i] = sum_d dO[i, d] O[i, d]  (= sum_j Pd dPd, with or without dropout)
//             P = exp(scale S - lse),  dPd = dO V^T,  dS = scale P (dPd keep/(1-p) - D)
//             dQ = dS K,  dK = dS^T Q,  dV = Pd^T dO
// Two backward kernels, both deterministic (no atomics): `attn_bwd_dq` (a wave owns 32 queries and walks the key tiles -- the
// forward's structure with two score-type products and no running softmax; it also writes D) and `attn_bwd_dkv` (a wave owns
// 32 keys and walks the query tiles; S is recomputed there a second time: 7 instead of 5 products per tile pair, the price of
// not combining dQ across workgroups).
//
// q, k, v, o, dO, dq, dk, dv: (B, T, H * DK) f32 exactly as the linear layers leave / take them (head h = channels
// [h DK, (h + 1) DK)) -- no head-major copies.  Tiles of 32 rows go through LDS as bf16 (rows padded to DK + 8).
//
// MFMA layouts (v_mfma_f32_32x32x16_bf16; l31 = lane & 31, half = lane >> 5):
//   A operand: lane holds A[row l31][k = 8 half .. 8 half + 7];  B operand: lane holds B[k = 8 half ..][col l31]
//   accumulator element i of a lane: row (i >> 2) * 8 + 4 half + (i & 3), column l31
// "score-type" product X^T (32 x 32) = R C^T: R's rows from LDS (A), C's rows as register fragments (B) -> a lane owns ONE
//   column entity (query in fwd / dq, key in dkv) and 16 row entities.
// "value-type" product Y (32 x DK) += X W: the accumulators of X^T ARE the A operand of X when the 16 k-indices of a step
//   are taken in accumulator row order; W's rows are read from LDS with ds_read_b64_tr_b16 in that same order.
#include "osp_common.h"

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

namespace {

// 32 rows (row0 .. row0 + 31 of one (b, h), rows >= nvalid as zeros) x DK f32 -> bf16 LDS tile, by the whole workgroup
template <int DK, int NT>
__device__ __forceinline__ void stage_rows(unsigned short* dst, const float* __restrict__ src, int64_t base, int C, int row0, int nvalid, int tid) {
    constexpr int LD = DK + 8;
    for (int i = tid; i < 32 * (DK / 4); i += NT) {
        const int r = i / (DK / 4), c4 = i - r * (DK / 4);
        const int rr = row0 + r;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rr < nvalid) x = *reinterpret_cast<const float4*>(src + base + (int64_t)rr * C + 4 * c4);
        bf16x2_t a, b;
        a[0] = (__bf16)x.x; a[1] = (__bf16)x.y; b[0] = (__bf16)x.z; b[1] = (__bf16)x.w;
        *reinterpret_cast<uint2*>(dst + r * LD + 4 * c4) = make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
    }
}

// The same in two halves for software pipelining: the NEXT tile's rows are requested into registers before the current tile is
// computed (a tile's global loads then fly under ~1 000 cycles of MFMA work instead of stalling the whole workgroup at the
// staging barrier), converted and written to LDS at the top of the next iteration.
template <int DK, int NT>
__device__ __forceinline__ void load_rows_regs(float4 (&r)[32 * (DK / 4) / NT], const float* __restrict__ src, int64_t base, int C, int row0, int nvalid, int tid) {
#pragma unroll
    for (int it = 0; it < 32 * (DK / 4) / NT; ++it) {
        const int i = tid + it * NT;
        const int rw = i / (DK / 4), c4 = i - rw * (DK / 4);
        const int rr = row0 + rw;
        r[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rr < nvalid) r[it] = *reinterpret_cast<const float4*>(src + base + (int64_t)rr * C + 4 * c4);
    }
}
template <int DK, int NT>
__device__ __forceinline__ void store_rows_lds(unsigned short* dst, const float4 (&r)[32 * (DK / 4) / NT], int tid) {
    constexpr int LD = DK + 8;
#pragma unroll
    for (int it = 0; it < 32 * (DK / 4) / NT; ++it) {
        const int i = tid + it * NT;
        const int rw = i / (DK / 4), c4 = i - rw * (DK / 4);
        bf16x2_t a, b;
        a[0] = (__bf16)r[it].x; a[1] = (__bf16)r[it].y; b[0] = (__bf16)r[it].z; b[1] = (__bf16)r[it].w;
        *reinterpret_cast<uint2*>(dst + rw * LD + 4 * c4) = make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
    }
}

// register fragments (B operand of a score-type product) of row `row` (clamped by the caller): d = 16 s + 8 half .. + 7
template <int DK>
__device__ __forceinline__ void load_frags(bf16x8_t (&f)[DK / 16], const float* __restrict__ p, int half) {
#pragma unroll
    for (int s = 0; s < DK / 16; ++s) {
        const float4 a = *reinterpret_cast<const float4*>(p + 16 * s + 8 * half);
        const float4 c = *reinterpret_cast<const float4*>(p + 16 * s + 8 * half + 4);
        bf16x8_t x;
        x[0] = (__bf16)a.x; x[1] = (__bf16)a.y; x[2] = (__bf16)a.z; x[3] = (__bf16)a.w;
        x[4] = (__bf16)c.x; x[5] = (__bf16)c.y; x[6] = (__bf16)c.z; x[7] = (__bf16)c.w;
        f[s] = x;
    }
}

// X^T (32 LDS rows x 32 fragment columns) = R C^T
template <int DK>
__device__ __forceinline__ f32x16_t score_product(const unsigned short* rows_l, const bf16x8_t (&cf)[DK / 16], int l31, int half) {
    constexpr int LD = DK + 8;
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int s = 0; s < DK / 16; ++s) {
        const bf16x8_t rf = *reinterpret_cast<const bf16x8_t*>(rows_l + l31 * LD + 16 * s + 8 * half);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rf, cf[s], acc, 0, 0, 0);
    }
    return acc;
}

// Y (32 x DK) += X W, X given as two bf16x8 A fragments in accumulator row order, W = 32 LDS rows (read transposed)
template <int DK>
__device__ __forceinline__ void value_product(f32x16_t (&y)[DK / 32], const bf16x8_t (&xf)[2], const unsigned short* w_l, int lane) {
    constexpr int LD = DK + 8;
    const int half = lane >> 5, r16 = lane & 15, g16 = (lane >> 4) & 1;
#pragma unroll
    for (int d = 0; d < DK / 32; ++d) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int col = d * 32 + 16 * g16 + 4 * (r16 & 3);
            const unsigned short* a0 = w_l + (16 * s + 4 * half + (r16 >> 2)) * LD + col;
            const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned short*)a0;
            s16x4_t lo, hi;
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(addr) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(8 * LD * 2) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi) : : "memory");
            union { struct { s16x4_t l, h; } s2; bf16x8_t vv; } u;
            u.s2.l = lo; u.s2.h = hi;
            y[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[s], u.vv, y[d], 0, 0, 0);
        }
    }
}

// keep / (1 - p) factors of 4 consecutive element indices idx0 .. idx0 + 3 (one Philox block when idx0 % 4 == 0)
__device__ __forceinline__ void dropout4(float (&f)[4], uint64_t seed, uint32_t stream, uint64_t idx0, float p) {
    const float ks = 1.0f / (1.0f - p);
    if ((idx0 & 3) == 0) {
        const uint4 r = philox4(seed, idx0 >> 2, stream);
        f[0] = u32_to_unit(r.x) < p ? 0.f : ks; f[1] = u32_to_unit(r.y) < p ? 0.f : ks;
        f[2] = u32_to_unit(r.z) < p ? 0.f : ks; f[3] = u32_to_unit(r.w) < p ? 0.f : ks;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) f[e] = dropout_factor(seed, stream, idx0 + e, p);
    }
}

// Column-entity variant (dK / dV kernel: a lane owns one KEY and 16 query rows): the 4 elements of a lane's row group are 4
// different rows, i.e. 4 different Philox blocks -- but the 4 lanes of a quad own the 4 keys of ONE block.  Each lane of the
// quad draws the block of ONE of the 4 rows; the words are then transposed inside the quad (DPP quad broadcasts), so a lane
// computes one Philox block per 4 elements instead of four (16 -> 4 per tile: the generator was ~2/3 of the kernel's instructions).
// Needs T % 4 == 0 (row starts on a block boundary) and key0 % 4 == 0; the caller falls back to dropout_factor otherwise.
template <int E>
__device__ __forceinline__ unsigned quad_bcast(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, E * 0x55, 0xf, 0xf, false);
}
__device__ __forceinline__ void dropout_rows4_quad(float (&f)[4], uint64_t seed, uint32_t stream, uint64_t zT, int q_first, int T, int key, int lane, float p) {
    const int c = lane & 3;
    const uint64_t idx = (zT + (uint64_t)(q_first + c)) * (uint64_t)T + (uint64_t)(key & ~3);      // block of row q_first + c
    const uint4 r = philox4(seed, idx >> 2, stream);
    const float ks = 1.0f / (1.0f - p);
    // all 16 broadcasts are executed by every lane BEFORE any select (a DPP read from a lane that sits out a branch returns 0)
    const unsigned x0 = quad_bcast<0>(r.x), y0 = quad_bcast<0>(r.y), z0 = quad_bcast<0>(r.z), w0 = quad_bcast<0>(r.w);
    const unsigned x1 = quad_bcast<1>(r.x), y1 = quad_bcast<1>(r.y), z1 = quad_bcast<1>(r.z), w1 = quad_bcast<1>(r.w);
    const unsigned x2 = quad_bcast<2>(r.x), y2 = quad_bcast<2>(r.y), z2 = quad_bcast<2>(r.z), w2 = quad_bcast<2>(r.w);
    const unsigned x3 = quad_bcast<3>(r.x), y3 = quad_bcast<3>(r.y), z3 = quad_bcast<3>(r.z), w3 = quad_bcast<3>(r.w);
    unsigned w[4];
    w[0] = c == 0 ? x0 : c == 1 ? y0 : c == 2 ? z0 : w0;
    w[1] = c == 0 ? x1 : c == 1 ? y1 : c == 2 ? z1 : w1;
    w[2] = c == 0 ? x2 : c == 1 ? y2 : c == 2 ? z2 : w2;
    w[3] = c == 0 ? x3 : c == 1 ? y3 : c == 2 ? z3 : w3;
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = u32_to_unit(w[e]) < p ? 0.f : ks;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------- forward
// One workgroup = 4 waves = 128 queries of one (b, h); a wave owns 32 queries; keys / values in tiles of 32.
template <int DK, bool BIAS>
__global__ __launch_bounds__(256) void attn_train_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                             const int64_t* __restrict__ klen, const float* __restrict__ sbias,
                                                             float* __restrict__ o, float* __restrict__ lse,
                                                             int H, int T, float scale, float drop_p, uint64_t seed,
                                                             const int64_t* __restrict__ seed_dev, uint32_t stream_id) {
    constexpr int LD = DK + 8, DB = DK / 32;
    __shared__ __attribute__((aligned(16))) unsigned short k_l[32 * LD];
    __shared__ __attribute__((aligned(16))) unsigned short v_l[32 * LD];
    __shared__ float fac_l[4][32];
    if (seed_dev) seed += (uint64_t)*seed_dev;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int z = blockIdx.y, b = z / H, hh = z - b * H;
    const int C = H * DK;
    const int kl = (int)min((int64_t)T, klen[b]);
    const int l31 = lane & 31, half = lane >> 5;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int qi = q0 + l31;
    const int64_t base = (int64_t)b * T * C + (int64_t)hh * DK;
    bf16x8_t qf[DK / 16];
    load_frags<DK>(qf, q + base + (int64_t)(qi < T ? qi : 0) * C, half);
    f32x16_t oacc[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[d][i] = 0.f;
    float m_run = -3.0e38f, l_run = 0.f;
    const uint64_t row_idx = ((uint64_t)z * T + (uint64_t)(qi < T ? qi : 0)) * (uint64_t)T;
    const int64_t brow = (int64_t)row_idx;                                        // start of this query's row in a (Z, T, T) tensor
    const int ntiles = (kl + 31) / 32;
    float4 kreg[32 * (DK / 4) / 256], vreg[32 * (DK / 4) / 256];
    if (ntiles > 0) {
        load_rows_regs<DK, 256>(kreg, k, base, C, 0, kl, tid);
        load_rows_regs<DK, 256>(vreg, v, base, C, 0, kl, tid);
    }
    for (int tile = 0; tile < ntiles; ++tile) {
        const int k0 = tile * 32;
        __syncthreads();                                                          // the previous tile's LDS reads are done
        store_rows_lds<DK, 256>(k_l, kreg, tid);
        store_rows_lds<DK, 256>(v_l, vreg, tid);
        __syncthreads();
        if (tile + 1 < ntiles) {                                                  // next tile's rows: in flight under this tile's MFMAs
            load_rows_regs<DK, 256>(kreg, k, base, C, k0 + 32, kl, tid);
            load_rows_regs<DK, 256>(vreg, v, base, C, k0 + 32, kl, tid);
        }
        float sb[16];                                                             // additive score term (relative-position attention), requested before the MFMAs
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kr = k0 + (i >> 2) * 8 + 4 * half + (i & 3);
            sb[i] = (BIAS && kr < kl) ? sbias[brow + kr] : 0.f;
        }
        const f32x16_t st = score_product<DK>(k_l, qf, l31, half);
        float mx = -3.0e38f;
        float sc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kr = k0 + (i >> 2) * 8 + 4 * half + (i & 3);
            sc[i] = kr < kl ? (st[i] + sb[i]) * scale : -3.0e38f;
            mx = fmaxf(mx, sc[i]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        float ps = 0.f;
        bf16x8_t pf[2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float keep[4] = {1.f, 1.f, 1.f, 1.f};
            if (drop_p > 0.f) dropout4(keep, seed, stream_id, row_idx + (uint64_t)(k0 + g * 8 + 4 * half), drop_p);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * g + e;
                const float p = sc[i] > -1.0e38f ? __expf(sc[i] - m_new) : 0.f;
                ps += p;                                        // the normaliser is the sum of the UN-dropped probabilities
                pf[i >> 3][i & 7] = (__bf16)(p * keep[e]);
            }
        }
        ps += __shfl_xor(ps, 32);
        l_run = l_run * alpha + ps;
        m_run = m_new;
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
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[d][i] *= fr[i];
        value_product<DK>(oacc, pf, v_l, lane);
    }
    if (half == 0) fac_l[wave][l31] = l_run > 0.f ? 1.f / l_run : 0.f;
    if (half == 0 && qi < T) lse[(int64_t)z * T + qi] = l_run > 0.f ? m_run + __logf(l_run) : 3.0e38f;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = (i >> 2) * 8 + 4 * half + (i & 3);
        const int qr = q0 + r;
        if (qr < T) {
            const float inv = fac_l[wave][r];
#pragma unroll
            for (int d = 0; d < DB; ++d) o[base + (int64_t)qr * C + d * 32 + l31] = oacc[d][i] * inv;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------- backward: dQ (+ D)
template <int DK, bool BIAS>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                          const float* __restrict__ o, const float* __restrict__ dout,
                                                          const float* __restrict__ lse, const int64_t* __restrict__ klen,
                                                          const float* __restrict__ sbias, float* __restrict__ dsbias,
                                                          float* __restrict__ dq, float* __restrict__ Dbuf, int H, int T, float scale,
                                                          float drop_p, uint64_t seed, const int64_t* __restrict__ seed_dev, uint32_t stream_id) {
    constexpr int LD = DK + 8, DB = DK / 32, DS = DK / 16;
    __shared__ __attribute__((aligned(16))) unsigned short k_l[32 * LD];
    __shared__ __attribute__((aligned(16))) unsigned short v_l[32 * LD];
    if (seed_dev) seed += (uint64_t)*seed_dev;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int z = blockIdx.y, b = z / H, hh = z - b * H;
    const int C = H * DK;
    const int kl = (int)min((int64_t)T, klen[b]);
    const int l31 = lane & 31, half = lane >> 5;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int qi = q0 + l31, qc = qi < T ? qi : 0;
    const int64_t base = (int64_t)b * T * C + (int64_t)hh * DK;
    bf16x8_t qf[DS], dof[DS];
    load_frags<DK>(qf, q + base + (int64_t)qc * C, half);
    load_frags<DK>(dof, dout + base + (int64_t)qc * C, half);
    // D of this lane's query: f32 dot over the lane's half of the channels, completed by the partner lane
    float Dq = 0.f;
    {
        const float* dp = dout + base + (int64_t)qc * C;
        const float* op = o + base + (int64_t)qc * C;
#pragma unroll
        for (int s = 0; s < DS; ++s) {
            const float4 a0 = *reinterpret_cast<const float4*>(dp + 16 * s + 8 * half), a1 = *reinterpret_cast<const float4*>(dp + 16 * s + 8 * half + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(op + 16 * s + 8 * half), b1 = *reinterpret_cast<const float4*>(op + 16 * s + 8 * half + 4);
            Dq += a0.x * b0.x + a0.y * b0.y + a0.z * b0.z + a0.w * b0.w + a1.x * b1.x + a1.y * b1.y + a1.z * b1.z + a1.w * b1.w;
        }
        Dq += __shfl_xor(Dq, 32);
        if (half == 0 && qi < T) Dbuf[(int64_t)z * T + qi] = Dq;
    }
    const float lse_q = qi < T ? lse[(int64_t)z * T + qi] : 3.0e38f;
    f32x16_t acc[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[d][i] = 0.f;
    const uint64_t row_idx = ((uint64_t)z * T + (uint64_t)qc) * (uint64_t)T;
    const int64_t brow = (int64_t)row_idx;
    const int ntiles = (kl + 31) / 32;
    float4 kreg[32 * (DK / 4) / 256], vreg[32 * (DK / 4) / 256];
    if (ntiles > 0) {
        load_rows_regs<DK, 256>(kreg, k, base, C, 0, kl, tid);
        load_rows_regs<DK, 256>(vreg, v, base, C, 0, kl, tid);
    }
    for (int tile = 0; tile < ntiles; ++tile) {
        const int k0 = tile * 32;
        __syncthreads();                                                          // the previous tile's LDS reads are done
        store_rows_lds<DK, 256>(k_l, kreg, tid);
        store_rows_lds<DK, 256>(v_l, vreg, tid);
        __syncthreads();
        if (tile + 1 < ntiles) {                                                  // next tile's rows: in flight under this tile's MFMAs
            load_rows_regs<DK, 256>(kreg, k, base, C, k0 + 32, kl, tid);
            load_rows_regs<DK, 256>(vreg, v, base, C, k0 + 32, kl, tid);
        }
        float sb[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kr = k0 + (i >> 2) * 8 + 4 * half + (i & 3);
            sb[i] = (BIAS && kr < kl) ? sbias[brow + kr] : 0.f;
        }
        const f32x16_t st = score_product<DK>(k_l, qf, l31, half);              // S^T: rows = keys, column = this lane's query
        const f32x16_t dpt = score_product<DK>(v_l, dof, l31, half);            // dPd^T
        bf16x8_t dsf[2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float keep[4] = {1.f, 1.f, 1.f, 1.f};
            if (drop_p > 0.f) dropout4(keep, seed, stream_id, row_idx + (uint64_t)(k0 + g * 8 + 4 * half), drop_p);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * g + e;
                const int kr = k0 + g * 8 + 4 * half + e;
                const float p = kr < kl ? __expf((st[i] + sb[i]) * scale - lse_q) : 0.f;
                const float ds = scale * p * (dpt[i] * keep[e] - Dq);
                dsf[i >> 3][i & 7] = (__bf16)ds;
                if (BIAS && dsbias && qi < T && kr < T) dsbias[brow + kr] = ds;         // the score term's gradient IS dS (f32, before the bf16 rounding)
            }
        }
        value_product<DK>(acc, dsf, k_l, lane);                                 // dQ += dS K
    }
    if (BIAS && dsbias && qi < T)                                                // keys past the last processed tile: zero gradient
        for (int kr = ntiles * 32 + 4 * half; kr < T; kr += 8)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (kr + e < T) dsbias[brow + kr + e] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int qr = q0 + (i >> 2) * 8 + 4 * half + (i & 3);
        if (qr < T) {
#pragma unroll
            for (int d = 0; d < DB; ++d) dq[base + (int64_t)qr * C + d * 32 + l31] = acc[d][i];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------- backward: dK, dV
// One workgroup = 4 waves = 128 keys of one (b, h); a wave owns 32 keys (K / V fragments in registers) and walks the query tiles.
template <int DK, bool BIAS>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                           const float* __restrict__ dout, const float* __restrict__ lse,
                                                           const float* __restrict__ Dbuf, const int64_t* __restrict__ klen,
                                                           const float* __restrict__ sbias,
                                                           float* __restrict__ dk, float* __restrict__ dv, int H, int T, float scale,
                                                           float drop_p, uint64_t seed, const int64_t* __restrict__ seed_dev, uint32_t stream_id) {
    constexpr int LD = DK + 8, DB = DK / 32, DS = DK / 16;
    __shared__ __attribute__((aligned(16))) unsigned short q_l[32 * LD];
    __shared__ __attribute__((aligned(16))) unsigned short do_l[32 * LD];
    __shared__ __attribute__((aligned(16))) float lse_l[32];
    __shared__ __attribute__((aligned(16))) float d_l[32];
    if (seed_dev) seed += (uint64_t)*seed_dev;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int z = blockIdx.y, b = z / H, hh = z - b * H;
    const int C = H * DK;
    const int kl = (int)min((int64_t)T, klen[b]);
    const int l31 = lane & 31, half = lane >> 5;
    const int kb0 = blockIdx.x * 128 + wave * 32;
    const int key = kb0 + l31, kc = key < T ? key : 0;
    const int64_t base = (int64_t)b * T * C + (int64_t)hh * DK;
    f32x16_t dka[DB], dva[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dka[d][i] = 0.f; dva[d][i] = 0.f; }
    if (blockIdx.x * 128 < kl) {                                                 // workgroup-uniform: some key of the block is valid
        bf16x8_t kf[DS], vf[DS];
        load_frags<DK>(kf, k + base + (int64_t)kc * C, half);
        load_frags<DK>(vf, v + base + (int64_t)kc * C, half);
        const bool key_ok = key < kl;
        const int ntiles = (T + 31) / 32;
        float4 qreg[32 * (DK / 4) / 256], doreg[32 * (DK / 4) / 256];
        float lse_r = 3.0e38f, d_r = 0.f;
        auto prefetch = [&](int q0) {
            load_rows_regs<DK, 256>(qreg, q, base, C, q0, T, tid);
            load_rows_regs<DK, 256>(doreg, dout, base, C, q0, T, tid);
            if (tid < 32) {
                const int qi = q0 + tid;
                lse_r = qi < T ? lse[(int64_t)z * T + qi] : 3.0e38f;
                d_r = qi < T ? Dbuf[(int64_t)z * T + qi] : 0.f;
            }
        };
        prefetch(0);
        for (int tile = 0; tile < ntiles; ++tile) {
            const int q0 = tile * 32;
            __syncthreads();                                                      // the previous tile's LDS reads are done
            store_rows_lds<DK, 256>(q_l, qreg, tid);
            store_rows_lds<DK, 256>(do_l, doreg, tid);
            if (tid < 32) { lse_l[tid] = lse_r; d_l[tid] = d_r; }
            __syncthreads();
            if (tile + 1 < ntiles) prefetch(q0 + 32);                             // in flight under this tile's 32 MFMAs per wave
            const f32x16_t s = score_product<DK>(q_l, kf, l31, half);            // S: rows = queries of the tile, column = this lane's key
            const f32x16_t dp = score_product<DK>(do_l, vf, l31, half);          // dPd
            bf16x8_t pdf[2], dsf[2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 l4 = *reinterpret_cast<const float4*>(&lse_l[g * 8 + 4 * half]);
                const float4 d4 = *reinterpret_cast<const float4*>(&d_l[g * 8 + 4 * half]);
                const float lr[4] = {l4.x, l4.y, l4.z, l4.w}, dr[4] = {d4.x, d4.y, d4.z, d4.w};
                float keep4[4] = {1.f, 1.f, 1.f, 1.f};
                if (drop_p > 0.f) {
                    if ((T & 3) == 0) {
                        dropout_rows4_quad(keep4, seed, stream_id, (uint64_t)z * T, q0 + g * 8 + 4 * half, T, key, lane, drop_p);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            keep4[e] = dropout_factor(seed, stream_id, ((uint64_t)z * T + (uint64_t)(q0 + g * 8 + 4 * half + e)) * (uint64_t)T + (uint64_t)key, drop_p);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * g + e;
                    const int qrow = q0 + g * 8 + 4 * half + e;
                    const float sbv = (BIAS && key_ok && qrow < T) ? sbias[((int64_t)z * T + qrow) * T + key] : 0.f;
                    const float p = key_ok ? __expf((s[i] + sbv) * scale - lr[e]) : 0.f;   // rows past T carry lse = +3e38: p = 0
                    const float keep = keep4[e];
                    pdf[i >> 3][i & 7] = (__bf16)(p * keep);
                    dsf[i >> 3][i & 7] = (__bf16)(scale * p * (dp[i] * keep - dr[e]));
                }
            }
            value_product<DK>(dva, pdf, do_l, lane);                             // dV += Pd^T dO
            value_product<DK>(dka, dsf, q_l, lane);                              // dK += dS^T Q
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int kr = kb0 + (i >> 2) * 8 + 4 * half + (i & 3);
        if (kr < T) {
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                dk[base + (int64_t)kr * C + d * 32 + l31] = dka[d][i];
                dv[base + (int64_t)kr * C + d * 32 + l31] = dva[d][i];
            }
        }
    }
}

// q, k, v (B, T, H*DK) f32; klen (B) int64; o (B, T, H*DK); lse (B*H, T).  drop_p in [0, 1); seed / seed_dev / stream_id as in
// osp_attn_softmax_fwd (the same Philox key and element indices: the fused and the unfused path draw identical masks).
// sbias (optional): (B*H, T, T) f32 added to q k^T BEFORE the scaling -- the relative-position term of RelPositionMultiHeadedAttention
// (_transformer/attention.py:290-313).
extern "C" int osp_attn_train_fwd(const float* q, const float* k, const float* v, const int64_t* klen, const float* sbias, float* o,
                                  float* lse, int64_t B, int64_t H, int64_t T, int64_t DK, float scale, float drop_p, int64_t seed,
                                  const int64_t* seed_dev, int64_t stream_id, hipStream_t stream) {
    OSP_CHECK_ARG(q && k && v && klen && o && lse && B > 0 && H > 0 && T > 0, "bad args");
    OSP_CHECK_ARG(DK == 32 || DK == 64 || DK == 128, "head width must be 32, 64 or 128");
    OSP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "dropout probability");
    const dim3 grid((unsigned)cdiv(T, 128), (unsigned)(B * H));
    osp_note_symbol("attn_train_fwd_kernel");
    osp_note_flops(4.0 * (double)B * H * T * T * DK);
#define L(D_, B_) hipLaunchKernelGGL((attn_train_fwd_kernel<D_, B_>), grid, dim3(256), 0, stream, q, k, v, klen, sbias, o, lse, (int)H, (int)T, scale, \
                                     drop_p, (uint64_t)seed, seed_dev, (uint32_t)stream_id)
    if (sbias) { if (DK == 128) L(128, true); else if (DK == 64) L(64, true); else L(32, true); }
    else       { if (DK == 128) L(128, false); else if (DK == 64) L(64, false); else L(32, false); }
#undef L
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// dout (B, T, H*DK); dq, dk, dv (B, T, H*DK) are fully written; Dbuf (B*H, T) scratch (written by the dQ kernel, read by dK/dV).
// sbias / dsbias (optional): the score term of the forward and its gradient ((B*H, T, T) f32, fully written).
extern "C" int osp_attn_train_bwd(const float* q, const float* k, const float* v, const float* o, const float* lse, const float* dout,
                                  const int64_t* klen, const float* sbias, float* dsbias, float* dq, float* dk, float* dv, float* Dbuf,
                                  int64_t B, int64_t H, int64_t T,
                                  int64_t DK, float scale, float drop_p, int64_t seed, const int64_t* seed_dev, int64_t stream_id,
                                  hipStream_t stream) {
    OSP_CHECK_ARG(q && k && v && o && lse && dout && klen && dq && dk && dv && Dbuf && B > 0 && H > 0 && T > 0, "bad args");
    OSP_CHECK_ARG(!dsbias || sbias, "a score-term gradient needs the score term");
    OSP_CHECK_ARG(DK == 32 || DK == 64 || DK == 128, "head width must be 32, 64 or 128");
    OSP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "dropout probability");
    const dim3 grid((unsigned)cdiv(T, 128), (unsigned)(B * H));
    osp_note_symbol("attn_bwd_dq_kernel");
    osp_note_flops(14.0 * (double)B * H * T * T * DK);
#define L(D_, B_)                                                                                                                      \
    do {                                                                                                                               \
        hipLaunchKernelGGL((attn_bwd_dq_kernel<D_, B_>), grid, dim3(256), 0, stream, q, k, v, o, dout, lse, klen, sbias, dsbias, dq, Dbuf, (int)H, (int)T,  \
                           scale, drop_p, (uint64_t)seed, seed_dev, (uint32_t)stream_id);                                              \
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<D_, B_>), grid, dim3(256), 0, stream, q, k, v, dout, lse, Dbuf, klen, sbias, dk, dv, (int)H, (int)T, \
                           scale, drop_p, (uint64_t)seed, seed_dev, (uint32_t)stream_id);                                              \
    } while (0)
    if (sbias) { if (DK == 128) L(128, true); else if (DK == 64) L(64, true); else L(32, true); }
    else       { if (DK == 128) L(128, false); else if (DK == 64) L(64, false); else L(32, false); }
#undef L
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
