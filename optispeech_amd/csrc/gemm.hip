// conv-GEMM family on the CDNA4 matrix cores (exact-f32 MFMA, v_mfma_f32_32x32x2_f32).
//
// One kernel serves every dense contraction of the path (reference rows A1b/A1c/A4/A6/A7/A11/A12,
// SURVEY.md section 8a): pointwise Linear, dense k-tap Conv1d on channels-last frames, their dgrad, and batched
// (B, T, N) x (B, N, C) products.  A k-tap conv over channels-last frames needs no im2col: the operand row
// of frame m for tap j is simply row (m + j - pad) of the activation matrix, valid iff it stays inside the
// utterance [0, T) -- so the K loop walks (tap, channel-chunk) and the loader zero-fills invalid rows.
//
// Numerics: f32 inputs, f32 accumulate; the MFMA is bit-identical to an fmaf chain (guide section 3), so parity
// with the CPU oracle is summation-order only.
//
// Tiling: 256 threads = 4 waves (2x2); block tile BM x BN in {128x128, 64x64}, BK = 16; each wave owns
// (BM/2)x(BN/2) = TMxTN tiles of 32x32 (16 accumulator VGPRs each).  LDS is k-major  As[k][m], Bs[k][n]
// (+4 pad) so a fragment read is 32 consecutive floats per half-wave (conflict-free ds_read_b32) and the
// k-contiguous global float4 is scattered with 4 ds_write_b32 (2-way at worst, free).  Register-prefetch
// double buffering: tile i+1 is in flight from HBM/L2 while tile i feeds the matrix pipe; one barrier per tile.
#include "osp_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BK 16
#define LPAD 4

enum {
    EPI_NONE = 0,            // C = acc (+bias)
    EPI_RELU = 1,            // C = relu(acc + bias)
    EPI_GELU = 2,            // u = acc + bias; aux_out = u (optional); C = gelu(u)
    EPI_SCALE_RES_MASK = 3,  // z = acc + bias; aux_out = z (optional); C = (res + rowscale*gamma*z) * rowmask
    EPI_GELU_BWD = 4,        // C = rowscale * acc * gelu'(aux_in)
    EPI_RELU_BWD = 5,        // C = acc * (aux_in > 0)
    EPI_AXMY = 6,            // C = rowscale * aux_in - acc
    EPI_MASK = 7,            // C = (acc + bias) * rowmask
};

struct GemmP {
    const float* A; int64_t lda; int M, T, Cin, taps, pad;
    const float* a_rowscale;
    const float* B; int64_t sBn, sBtap, sBk; int N;
    float* C; int64_t ldc;
    int epi;
    const float *bias, *gamma, *res; int64_t ldr;
    const float *rowmask, *rowscale;
    float* aux_out; const float* aux_in; int64_t ld_aux;
    int64_t sAb, sBb, sCb, sXb;   // batch strides (A, B, C, aux/res)
    int accumulate, a_vec, b_vec;
};

__device__ __forceinline__ float4 ld4_guard(const float* p, int c, int lim, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vec && c + 3 < lim) {
        v = *reinterpret_cast<const float4*>(p);
    } else {
        if (c < lim) v.x = p[0];
        if (c + 1 < lim) v.y = p[1];
        if (c + 2 < lim) v.z = p[2];
        if (c + 3 < lim) v.w = p[3];
    }
    return v;
}

template <int TM, int TN>
__device__ __forceinline__ void mma_ktile(const float* __restrict__ As, const float* __restrict__ Bs, int lda_s,
                                          int ldb_s, int wm0, int wn0, int lane, f32x16 (&acc)[TM][TN]) {
    const int kh = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
        float a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = As[(kk + kh) * lda_s + wm0 + 32 * i + l31];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = Bs[(kk + kh) * ldb_s + wn0 + 32 * j + l31];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
}

template <int BM, int BN, bool B_KCONTIG>
__global__ __launch_bounds__(256) void conv_gemm_f32_kernel(GemmP p) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int LDA_S = BM + LPAD, LDB_S = BN + LPAD;
    constexpr int A_PER = BM / 64;                      // float4 per thread for the A tile
    constexpr int B_PER = BN / 64;                      // same count in both B modes (BK*BN/4/256)
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (LDA_S + LDB_S)];
    float* As = smem;                                   // [2][BK][LDA_S]
    float* Bs = smem + 2 * BK * LDA_S;                  // [2][BK][LDB_S]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int64_t bz = blockIdx.z;
    const float* A = p.A + bz * p.sAb;
    const float* B = p.B + bz * p.sBb;

    // ---- per-thread A rows
    const int kq = tid & 3, r0 = tid >> 2;
    int a_m[A_PER], a_t[A_PER];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        const int m = m0 + r0 + 64 * i;
        a_m[i] = m;
        a_t[i] = (m < p.M) ? (m % p.T) : -0x40000000;   // invalid rows never pass the range test
    }
    // ---- per-thread B coordinates
    constexpr int BN4 = BN / 4;
    const int bn_col4 = tid % BN4, bn_k0 = tid / BN4;   // N-contig mode
    constexpr int BN_KSTEP = 256 / BN4;                 // rows of k covered per pass

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int chunks = (p.Cin + BK - 1) / BK;
    const int niter = p.taps * chunks;
    float4 ra[A_PER], rb[B_PER];

    auto gload = [&](int it) {
        const int j = it / chunks, c0 = (it - j * chunks) * BK;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int tt = a_t[i] + j - p.pad;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tt >= 0 && tt < p.T) {
                const int64_t row = (int64_t)a_m[i] + j - p.pad;
                const int c = c0 + 4 * kq;
                v = ld4_guard(A + row * p.lda + c, c, p.Cin, p.a_vec);
                if (p.a_rowscale) {
                    const float s = p.a_rowscale[bz * p.M + row];
                    v.x *= s; v.y *= s; v.z *= s; v.w *= s;
                }
            }
            ra[i] = v;
        }
        if (B_KCONTIG) {
#pragma unroll
            for (int i = 0; i < B_PER; ++i) {
                const int n = n0 + r0 + 64 * i, c = c0 + 4 * kq;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n < p.N) v = ld4_guard(B + (int64_t)n * p.sBn + (int64_t)j * p.sBtap + c, c, p.Cin, p.b_vec);
                rb[i] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < B_PER; ++i) {
                const int k = bn_k0 + BN_KSTEP * i, c = c0 + k, n = n0 + 4 * bn_col4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c < p.Cin) v = ld4_guard(B + (int64_t)c * p.sBk + (int64_t)j * p.sBtap + n, n, p.N, p.b_vec);
                rb[i] = v;
            }
        }
    };
    auto sstore = [&](int buf) {
        float* as = As + buf * BK * LDA_S;
        float* bs = Bs + buf * BK * LDB_S;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int r = r0 + 64 * i;
            as[(4 * kq + 0) * LDA_S + r] = ra[i].x;
            as[(4 * kq + 1) * LDA_S + r] = ra[i].y;
            as[(4 * kq + 2) * LDA_S + r] = ra[i].z;
            as[(4 * kq + 3) * LDA_S + r] = ra[i].w;
        }
        if (B_KCONTIG) {
#pragma unroll
            for (int i = 0; i < B_PER; ++i) {
                const int r = r0 + 64 * i;
                bs[(4 * kq + 0) * LDB_S + r] = rb[i].x;
                bs[(4 * kq + 1) * LDB_S + r] = rb[i].y;
                bs[(4 * kq + 2) * LDB_S + r] = rb[i].z;
                bs[(4 * kq + 3) * LDB_S + r] = rb[i].w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < B_PER; ++i) {
                const int k = bn_k0 + BN_KSTEP * i;
                *reinterpret_cast<float4*>(&bs[k * LDB_S + 4 * bn_col4]) = rb[i];
            }
        }
    };

    gload(0);
    sstore(0);
    __syncthreads();
    for (int it = 0; it < niter; ++it) {
        const int buf = it & 1;
        if (it + 1 < niter) gload(it + 1);
        mma_ktile<TM, TN>(As + buf * BK * LDA_S, Bs + buf * BK * LDB_S, LDA_S, LDB_S, wm0, wn0, lane, acc);
        if (it + 1 < niter) sstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* C = p.C + bz * p.sCb;
    const float* res = p.res ? p.res + bz * p.sXb : nullptr;
    const float* aux_in = p.aux_in ? p.aux_in + bz * p.sXb : nullptr;
    float* aux_out = p.aux_out ? p.aux_out + bz * p.sXb : nullptr;
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn0 + 32 * j + l31;
            if (n >= p.N) continue;
            const float bias = p.bias ? p.bias[n] : 0.f;
            const float gam = p.gamma ? p.gamma[n] : 1.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= p.M) continue;
                const int64_t mr = bz * p.M + m;
                float v = acc[i][j][r] + bias, out;
                switch (p.epi) {
                    case EPI_RELU: out = fmaxf(v, 0.f); break;
                    case EPI_GELU:
                        if (aux_out) aux_out[(int64_t)m * p.ld_aux + n] = v;
                        out = gelu_f(v);
                        break;
                    case EPI_SCALE_RES_MASK: {
                        if (aux_out) aux_out[(int64_t)m * p.ld_aux + n] = v;
                        const float rs = p.rowscale ? p.rowscale[mr] : 1.f;
                        const float mk = p.rowmask ? p.rowmask[mr] : 1.f;
                        out = (res[(int64_t)m * p.ldr + n] + rs * gam * v) * mk;
                    } break;
                    case EPI_GELU_BWD: {
                        const float rs = p.rowscale ? p.rowscale[mr] : 1.f;
                        out = rs * v * gelu_grad_f(aux_in[(int64_t)m * p.ld_aux + n]);
                    } break;
                    case EPI_RELU_BWD: out = aux_in[(int64_t)m * p.ld_aux + n] > 0.f ? v : 0.f; break;
                    case EPI_AXMY: {
                        const float rs = p.rowscale ? p.rowscale[mr] : 1.f;
                        out = rs * aux_in[(int64_t)m * p.ld_aux + n] - v;
                    } break;
                    case EPI_MASK: out = v * (p.rowmask ? p.rowmask[mr] : 1.f); break;
                    default: out = v;
                }
                float* dst = C + (int64_t)m * p.ldc + n;
                *dst = p.accumulate ? (*dst + out) : out;
            }
        }
}

int osp_try_gemm_f32_glds(const float* A, int64_t lda, int64_t M, int64_t T, int64_t Cin, int64_t taps, int64_t pad,
                          const float* a_rowscale, const float* B, int64_t sBn, int64_t sBtap, int64_t sBk, int64_t N, float* C,
                          int64_t ldc, int64_t epi, const float* bias, const float* gamma, const float* res, int64_t ldr,
                          const float* rowmask, const float* rowscale, float* aux_out, const float* aux_in, int64_t ld_aux,
                          int64_t batch, int64_t sAb, int64_t sBb, int64_t sCb, int64_t sXb, int64_t accumulate, hipStream_t stream);   // gemm_f32_glds.hip

template <int BM, int BN>
static int launch_gemm(const GemmP& p, int b_kcontig, int batch, hipStream_t stream) {
    dim3 grid((unsigned)cdiv(p.N, BN), (unsigned)cdiv(p.M, BM), (unsigned)batch);
    osp_note_symbol("conv_gemm_f32_kernel");                            // measurement aid (api.cpp)
    osp_note_flops(2.0 * p.M * p.taps * (double)p.Cin * p.N * batch);
    osp_note_bytes(4.0 * batch * ((double)p.M * p.Cin + (double)p.N * p.taps * p.Cin + (double)p.M * p.N));
    if (b_kcontig)
        hipLaunchKernelGGL((conv_gemm_f32_kernel<BM, BN, true>), grid, dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL((conv_gemm_f32_kernel<BM, BN, false>), grid, dim3(256), 0, stream, p);
    return 0;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// C[b, m, n] = epi( sum_{j<taps} sum_{c<Cin} A[b, m + j - pad, c] * Bw(n, j, c) ),  Bw(n,j,c) = B[n*sBn + j*sBtap + c*sBk]
// Rows are frames of utterances of length T (M = n_utt * T); a tap that leaves [0,T) contributes zero.
// Reference ops served: nn.Linear / nn.Conv1d call sites of convnext.py:39-41, core.py:66-71, alignments.py:55-64,
// wavenext/__init__.py:43-44,83 and their autograd dgrad.
extern "C" int osp_conv_gemm_f32(const float* A, int64_t lda, int64_t M, int64_t T, int64_t Cin, int64_t taps,
                                 int64_t pad, const float* a_rowscale, const float* B, int64_t sBn, int64_t sBtap,
                                 int64_t sBk, int64_t N, float* C, int64_t ldc, int64_t epi, const float* bias,
                                 const float* gamma, const float* res, int64_t ldr, const float* rowmask,
                                 const float* rowscale, float* aux_out, const float* aux_in, int64_t ld_aux,
                                 int64_t batch, int64_t sAb, int64_t sBb, int64_t sCb, int64_t sXb,
                                 int64_t accumulate, hipStream_t stream) {
    OSP_CHECK_ARG(A && B && C, "null operand");
    OSP_CHECK_ARG(M > 0 && N > 0 && Cin > 0 && taps > 0 && T > 0 && batch > 0, "bad shape");
    OSP_CHECK_ARG(M % T == 0, "M must be a whole number of utterances of T frames");
    OSP_CHECK_ARG(sBk == 1 || sBn == 1, "B must be contiguous along k or along n");
    OSP_CHECK_ARG(epi >= 0 && epi <= EPI_MASK, "unknown epilogue");
    OSP_CHECK_ARG(epi != EPI_SCALE_RES_MASK || res, "epilogue needs res");
    OSP_CHECK_ARG((epi != EPI_GELU_BWD && epi != EPI_RELU_BWD && epi != EPI_AXMY) || aux_in, "epilogue needs aux_in");
    {   // the direct-to-LDS f32 kernel where its conditions hold (gemm_f32_glds.hip)
        const int r = osp_try_gemm_f32_glds(A, lda, M, T, Cin, taps, pad, a_rowscale, B, sBn, sBtap, sBk, N, C, ldc, epi, bias, gamma, res, ldr,
                                            rowmask, rowscale, aux_out, aux_in, ld_aux, batch, sAb, sBb, sCb, sXb, accumulate, stream);
        if (r < 0) { osp_set_error("osp_conv_gemm_f32: launch failed"); return OSP_ERR_HIP; }
        if (r > 0) return OSP_OK;
    }
    GemmP p;
    p.A = A; p.lda = lda; p.M = (int)M; p.T = (int)T; p.Cin = (int)Cin; p.taps = (int)taps; p.pad = (int)pad;
    p.a_rowscale = a_rowscale;
    p.B = B; p.sBn = sBn; p.sBtap = sBtap; p.sBk = sBk; p.N = (int)N;
    p.C = C; p.ldc = ldc; p.epi = (int)epi; p.bias = bias; p.gamma = gamma; p.res = res; p.ldr = ldr;
    p.rowmask = rowmask; p.rowscale = rowscale; p.aux_out = aux_out; p.aux_in = aux_in; p.ld_aux = ld_aux;
    p.sAb = sAb; p.sBb = sBb; p.sCb = sCb; p.sXb = sXb; p.accumulate = (int)accumulate;
    const int b_kcontig = (sBk == 1) ? 1 : 0;
    p.a_vec = aligned16(A) && (lda % 4 == 0) && (sAb % 4 == 0);
    if (b_kcontig)
        p.b_vec = aligned16(B) && (sBn % 4 == 0) && (sBtap % 4 == 0) && (sBb % 4 == 0);
    else
        p.b_vec = aligned16(B) && (sBk % 4 == 0) && (sBtap % 4 == 0) && (sBb % 4 == 0);
    // tile choice: big tiles once they still fill the 256 CUs
    const int64_t big_blocks = cdiv(M, 128) * cdiv(N, 128) * batch;
    if (big_blocks >= 192) launch_gemm<128, 128>(p, b_kcontig, (int)batch, stream);
    else launch_gemm<64, 64>(p, b_kcontig, (int)batch, stream);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// ----------------------------------------------------------------------------------------------- wgrad
// dW[n, j, c] += oscale[n] * sum_m arow[m] * dY[m, n] * X[m + j - pad, c]   (valid taps only)
// db[n]       += oscale[n] * sum_m arow[m] * dY[m, n]                        (optional)
// Both operands are reduction-major (frames are rows), so tiles go to the k-major LDS image with plain
// ds_write_b128.  The frame dimension is split across blockIdx.z; partial tiles are combined with f32
// atomics straight into the gradient arena (which the step zeroes once).
struct WgradP {
    const float* dY; int64_t ldy; const float* X; int64_t ldx;
    int M, T, N, Cin, taps, pad;
    const float *arow, *oscale;
    float* dW; int64_t ldw; float* db;
    int chunk; int y_vec, x_vec;
    int splits; int64_t sYb, sXb, sWb, sDb;   // batched: blockIdx.z = batch * splits + split
};

template <int BMo, int BNo>
__global__ __launch_bounds__(256) void conv_wgrad_f32_kernel(WgradP p) {
    constexpr int TM = BMo / 64, TN = BNo / 64;
    constexpr int LDA_S = BMo + LPAD, LDB_S = BNo + LPAD;
    constexpr int A4 = BMo / 4, B4 = BNo / 4;
    constexpr int A_PER = BK * A4 / 256, B_PER = BK * B4 / 256;
    constexpr int A_KSTEP = 256 / A4, B_KSTEP = 256 / B4;
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (LDA_S + LDB_S)];
    float* As = smem;
    float* Bs = smem + 2 * BK * LDA_S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * (BMo / 2), wn0 = (wave & 1) * (BNo / 2);
    const int ctiles = (p.Cin + BNo - 1) / BNo;
    const int j = blockIdx.y / ctiles, c0 = (blockIdx.y - j * ctiles) * BNo;
    const int n0 = blockIdx.x * BMo;
    const int bz = blockIdx.z / p.splits, sp = blockIdx.z - bz * p.splits;
    const float* dYb = p.dY + (int64_t)bz * p.sYb;
    const float* Xb = p.X + (int64_t)bz * p.sXb;
    const float* arow = p.arow ? p.arow + (int64_t)bz * p.M : nullptr;
    const int mbeg = sp * p.chunk, mend = min(p.M, mbeg + p.chunk);
    const int a_c4 = tid % A4, a_k0 = tid / A4, b_c4 = tid % B4, b_k0 = tid / B4;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jj = 0; jj < TN; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
    float bsum = 0.f;
    const bool do_bias = (p.db != nullptr) && (blockIdx.y == 0) && (tid < BMo);

    float4 ra[A_PER], rb[B_PER];
    auto gload = [&](int mk) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int m = mk + a_k0 + A_KSTEP * i, n = n0 + 4 * a_c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < mend) {
                v = ld4_guard(dYb + (int64_t)m * p.ldy + n, n, p.N, p.y_vec);
                if (arow) { const float s = arow[m]; v.x *= s; v.y *= s; v.z *= s; v.w *= s; }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int m = mk + b_k0 + B_KSTEP * i, c = c0 + 4 * b_c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < mend) {
                const int tt = (m % p.T) + j - p.pad;
                if (tt >= 0 && tt < p.T) v = ld4_guard(Xb + ((int64_t)m + j - p.pad) * p.ldx + c, c, p.Cin, p.x_vec);
            }
            rb[i] = v;
        }
    };
    auto sstore = [&](int buf) {
        float* as = As + buf * BK * LDA_S;
        float* bs = Bs + buf * BK * LDB_S;
#pragma unroll
        for (int i = 0; i < A_PER; ++i)
            *reinterpret_cast<float4*>(&as[(a_k0 + A_KSTEP * i) * LDA_S + 4 * a_c4]) = ra[i];
#pragma unroll
        for (int i = 0; i < B_PER; ++i)
            *reinterpret_cast<float4*>(&bs[(b_k0 + B_KSTEP * i) * LDB_S + 4 * b_c4]) = rb[i];
    };

    const int niter = (mend - mbeg + BK - 1) / BK;
    if (niter > 0) {
        gload(mbeg);
        sstore(0);
        __syncthreads();
        for (int it = 0; it < niter; ++it) {
            const int buf = it & 1;
            if (it + 1 < niter) gload(mbeg + (it + 1) * BK);
            const float* as = As + buf * BK * LDA_S;
            if (do_bias) {
#pragma unroll
                for (int kk = 0; kk < BK; ++kk) bsum += as[kk * LDA_S + tid];
            }
            mma_ktile<TM, TN>(as, Bs + buf * BK * LDB_S, LDA_S, LDB_S, wm0, wn0, lane, acc);
            if (it + 1 < niter) sstore(buf ^ 1);
            __syncthreads();
        }
    }
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jj = 0; jj < TN; ++jj) {
            const int c = c0 + wn0 + 32 * jj + l31;
            if (c >= p.Cin) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (n >= p.N) continue;
                const float s = p.oscale ? p.oscale[n] : 1.f;
                atomicAdd(p.dW + (int64_t)bz * p.sWb + (int64_t)n * p.ldw + (int64_t)j * p.Cin + c, s * acc[i][jj][r]);
            }
        }
    if (do_bias && n0 + tid < p.N)
        atomicAdd(p.db + (int64_t)bz * p.sDb + n0 + tid, (p.oscale ? p.oscale[n0 + tid] : 1.f) * bsum);
}

extern "C" int osp_conv_wgrad_f32(const float* dY, int64_t ldy, const float* X, int64_t ldx, int64_t M, int64_t T,
                                  int64_t N, int64_t Cin, int64_t taps, int64_t pad, const float* arow,
                                  const float* oscale, float* dW, int64_t ldw, float* db, int64_t batch, int64_t sYb,
                                  int64_t sXb, int64_t sWb, int64_t sDb, hipStream_t stream) {
    OSP_CHECK_ARG(dY && X && dW, "null operand");
    OSP_CHECK_ARG(batch > 0, "batch");
    OSP_CHECK_ARG(M > 0 && N > 0 && Cin > 0 && taps > 0 && T > 0 && M % T == 0, "bad shape");
    WgradP p;
    p.dY = dY; p.ldy = ldy; p.X = X; p.ldx = ldx; p.M = (int)M; p.T = (int)T; p.N = (int)N; p.Cin = (int)Cin;
    p.taps = (int)taps; p.pad = (int)pad; p.arow = arow; p.oscale = oscale; p.dW = dW; p.ldw = ldw; p.db = db;
    p.y_vec = aligned16(dY) && (ldy % 4 == 0) && (sYb % 4 == 0);
    p.x_vec = aligned16(X) && (ldx % 4 == 0) && (sXb % 4 == 0);
    p.sYb = sYb; p.sXb = sXb; p.sWb = sWb; p.sDb = sDb;
    const bool big = (N >= 128 && Cin >= 128);
    const int bmo = big ? 128 : 64;
    const int64_t tiles = cdiv(N, bmo) * taps * cdiv(Cin, bmo) * batch;
    // split the frame dimension until ~2 blocks per CU are in flight (chunk multiple of BK)
    int64_t splits = cdiv(512, tiles);
    int64_t chunk = cdiv(cdiv(M, splits), BK) * BK;
    if (chunk < 4 * BK) chunk = 4 * BK;
    splits = cdiv(M, chunk);
    p.chunk = (int)chunk;
    p.splits = (int)splits;
    dim3 grid((unsigned)cdiv(N, bmo), (unsigned)(taps * cdiv(Cin, bmo)), (unsigned)(splits * batch));
    osp_note_symbol("conv_wgrad_f32_kernel");
    osp_note_flops(2.0 * M * taps * (double)Cin * N * batch);
    if (big) hipLaunchKernelGGL((conv_wgrad_f32_kernel<128, 128>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((conv_wgrad_f32_kernel<64, 64>), grid, dim3(256), 0, stream, p);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
