// Dropped experiment of round 4 (NOT built): the producer / consumer variant of csrc/mlp_fused.hip.  It was pasted into that file
// (it uses its helpers: MlpP, mlp_sfor, mlp_lds_rd16, mlp_gelu, mlp_epilogue, mlp_smem, MLP_MAX_I) and launched with 512 threads and
// MLPPC_LDS bytes of dynamic LDS.  Correct (tests/test_gpu_mlp_fused.py with W2 as the plain bf16 pack), same time per chunk as the
// one-wave-per-SIMD kernel: profiles/r04_mlp_fused_attribution.txt.
// ------------------------------------------------------------------------------------------------ producer / consumer variant
// The kernel above keeps a 32 x C output tile, the S^T tile and the rows of h in ONE wave: one wave per SIMD, and whatever is not
// in an MFMA's shadow (GELU, LDS-DMA issue, barrier turn-arounds) is serial time (profiles/r04_mlp_fused_attribution.txt).  Here a
// workgroup is 8 waves = two per SIMD, split by ROLE:
//   producer wave w (w < 4): rows 32 w .. 32 w + 31, phase 1 + GELU: h fragments (C / 4 registers) + two S^T tiles of the current
//                   64-unit chunk + the previous chunk's two tiles, whose GELU (VALU) is dealt out between this chunk's MFMAs; the
//                   bf16 result goes to LDS in A-operand order (G, two buffers of 128 rows x 64 units)
//   consumer wave w: the same rows, phase 2: out (32 x C f32) += G W2[:, chunk]^T, A fragments from G, B fragments from the ring
// so that one wave's VALU / LDS-DMA issue runs under the other's MFMAs.  The consumer trails the producer by two chunks (S(c) is
// finished in chunk interval c, its GELU written during c + 1, read during c + 2): nchunks + 2 intervals per workgroup.
// Ring: 6 slots of 16 KB.  Per third (half at C = 256) of a chunk interval the producer reads one slot (two W1 k-slabs of the
// chunk: 2 x 64 rows x 64 k) and the consumer one slot (one W2 unit: 128 output rows x the chunk's 64 units); the two slots of
// step s + 2 are requested during step s (4 LDS-DMA instructions per wave and step), waited for with vmcnt(4).
// W2 is the plain (C, I) bf16 pack here: G is stored in natural unit order.
#define MLPPC_RING (6 * 128 * 64 * 2)
#define MLPPC_G (128 * 64 * 2)
#define MLPPC_LDS (MLPPC_RING + 2 * MLPPC_G + MLP_MAX_I * 4)

template <int C>
__global__ __launch_bounds__(512) void convnext_mlp_pc_kernel(const MlpP p) {
    constexpr int KS1 = C / 64, NP = C / 128, NT = C / 32, KH = C / 16;          // W1 k-slabs / W2 units per 64-unit chunk
    static_assert(KS1 == 2 * NP, "two W1 k-slabs per step");
    constexpr int UNITB = 128 * 64 * 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const bool producer = wave < 4;
    const int wr = wave & 3;                                                       // row group of this wave
    const int I = p.I, nchunks = I / 64;
    const int m0 = blockIdx.x * 128 + wr * 32;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)mlp_smem;
    const unsigned g0 = lds0 + MLPPC_RING;
    {
        unsigned* tab = reinterpret_cast<unsigned*>(mlp_smem) + (MLPPC_RING + 2 * MLPPC_G) / 4;
        for (int i = tid; i < I; i += 512) {
            const float b = p.b1[i];
            const __bf16 hi = (__bf16)b, lo = (__bf16)(b - (float)hi);
            tab[i] = (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
        }
        __syncthreads();
    }
    const unsigned tab0 = lds0 + MLPPC_RING + 2 * MLPPC_G + 4 * l31;

    // ---- staging: a slot is 128 rows of 128 B = 16 LDS-DMA instructions; wave w issues instructions 2 w and 2 w + 1
    const int rsub = lane >> 3, pslot = lane & 7;
    unsigned offA[2], offB[2];                                                     // byte offsets into W1 / W2 (without chunk / step)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 8 * (2 * wave + i) + rsub, sw = (pslot ^ ((r >> 1) & 7)) * 8;
        // slot A: rows 0..63 = W1 rows (unit) of the step's first k-slab, rows 64..127 = the same units, second k-slab
        offA[i] = 2u * (unsigned)((r & 63) * C + (r >> 6) * 64 + sw);
        offB[i] = 2u * (unsigned)(r * I + sw);
    }
    // step x (global): producer chunk x / NP, part x % NP; consumer chunk x / NP - 2.  Out-of-range chunks are clamped (requests
    // nobody reads keep the vmcnt arithmetic uniform)
    auto issue_step = [&](int x, auto iidx) {
        constexpr int i = decltype(iidx)::value;                                  // 0, 1: slot A; 2, 3: slot B
        const int cP = x / NP, j = x - cP * NP, base = (x % 3) * 2;
        if constexpr (i < 2) {
            const int cc = cP < nchunks ? cP : nchunks - 1;
            const char* src = reinterpret_cast<const char*>(p.w1 + (int64_t)cc * 64 * C + 128 * j) + offA[i];
            __attribute__((address_space(3))) unsigned short* dst =
                (__attribute__((address_space(3))) unsigned short*)mlp_smem + base * (UNITB / 2) + (2 * wave + i) * (8 * 64);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        } else {
            int cc = cP - 2; cc = cc < 0 ? 0 : (cc < nchunks ? cc : nchunks - 1);
            const char* src = reinterpret_cast<const char*>(p.w2p + (int64_t)(128 * j) * I + cc * 64) + offB[i - 2];
            __attribute__((address_space(3))) unsigned short* dst =
                (__attribute__((address_space(3))) unsigned short*)mlp_smem + (base + 1) * (UNITB / 2) + (2 * wave + i - 2) * (8 * 64);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    mlp_sfor<0, 4>([&](auto iidx) { issue_step(0, iidx); });
    mlp_sfor<0, 4>([&](auto iidx) { issue_step(1, iidx); });

    unsigned fo[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fo[ks] = lds0 + 2 * (l31 * 64 + (((2 * ks + half) ^ ((l31 >> 1) & 7)) << 3));
    const int nsteps = (nchunks + 2) * NP;
    auto open_step = [&]() {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    if (producer) {
        bf16x8 hf[KH];
        {
            const int m = m0 + l31;
            const unsigned short* hp = p.h + (int64_t)(m < p.M ? m : 0) * C + 8 * half;
#pragma unroll
            for (int s = 0; s < KH; ++s) {
                uint4 v = *reinterpret_cast<const uint4*>(hp + 16 * s);
                if (m >= p.M) v = make_uint4(0u, 0u, 0u, 0u);
                hf[s] = __builtin_bit_cast(bf16x8, v);
            }
        }
        const f32x16 zacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int row = wr * 32 + l31;
        const unsigned gw0 = g0 + row * 128 + half * 8;                           // + buffer * MLPPC_G + (slot ^ swz) * 16
        const unsigned gswz = (row >> 1) & 7;
        f32x16 sa[2], sb[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) { sa[t] = zacc; sb[t] = zacc; }
        i32x4 fb[3][2];                                                            // fragment buffers: reads run two k-steps ahead of the MFMAs
        // one chunk interval: MFMAs of chunk c into `cur`, GELU of chunk c - 1 out of `prev` into G[(c - 1) & 1]
        auto chunk = [&](int c, f32x16 (&cur)[2], f32x16 (&prev)[2]) {
            // no conditionals in here: past the last chunk the MFMAs run on the (clamped) last chunk's weights and the GELU of a chunk
            // nobody reads is written to a G buffer nobody reads; chunk -1 is zeros.  (Uniform branches around the MFMAs made the
            // accumulator sets phi nodes: ~200 register copies per step and 40 spilled registers.)
            const unsigned gwb = gw0 + ((c - 1) & 1) * MLPPC_G;
            unsigned gs = gswz;
            asm volatile("" : "+v"(gs));                                         // (opaque: the 8 swizzled slot offsets are recomputed -- 2 VALU
                                                                                  //  each -- instead of living in 8 registers across the loop: they spilled)
            // element group e8 (0..7): tile t = e8 / 4, quarter q = e8 % 4: accumulator elements 4 q .. 4 q + 3 = units 32 t + 8 q + 4 half ..
            auto gelu_group = [&](auto eidx) {
                constexpr int e8 = decltype(eidx)::value, t = e8 >> 2, q = e8 & 3;
                bf16x2 lo2, hi2;
                lo2[0] = (__bf16)mlp_gelu(prev[t][4 * q]); lo2[1] = (__bf16)mlp_gelu(prev[t][4 * q + 1]);
                hi2[0] = (__bf16)mlp_gelu(prev[t][4 * q + 2]); hi2[1] = (__bf16)mlp_gelu(prev[t][4 * q + 3]);
                const uint2 d = make_uint2(__builtin_bit_cast(unsigned, lo2), __builtin_bit_cast(unsigned, hi2));
                const unsigned a = gwb + (((4 * t + q) ^ gs) << 4);
                asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(d) : "memory");
            };
            mlp_sfor<0, NP>([&](auto jidx) {
                constexpr int j = decltype(jidx)::value;
                const int x = c * NP + j;
                open_step();
                const unsigned ub = ((x % 3) * 2) * UNITB;
                if constexpr (j == 0) {
                    // start from the bias: one MFMA per tile against the (hi, lo) table
                    {
                        unsigned bt[2];
                        const unsigned ta = tab0 + (c < nchunks ? c : nchunks - 1) * 256;
                        asm volatile("ds_read_b32 %0, %1 offset:0" : "=v"(bt[0]) : "v"(ta) : "memory");
                        asm volatile("ds_read_b32 %0, %1 offset:128" : "=v"(bt[1]) : "v"(ta) : "memory");
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bt[0]), "+v"(bt[1]) : : "memory");
                        const i32x4 onesf = {half ? 0 : 0x3F803F80, 0, 0, 0};     // (rebuilt per chunk: 4 registers less across the loop)
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const i32x4 af = {(int)bt[t], 0, 0, 0};
                            cur[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, onesf), zacc, 0, 0, 0);
                        }
                    }
                }
                // 8 k-steps: slabs 2 j (rows 0..63 of the slot) and 2 j + 1 (rows 64..127); per k-step 2 reads + 2 MFMAs
                auto rd = [&](auto kidx) {
                    constexpr int k8 = decltype(kidx)::value, sl = k8 >> 2, ks = k8 & 3;
                    const unsigned a = fo[ks] + ub;
                    i32x4 (&f)[2] = fb[k8 % 3];
                    mlp_lds_rd16<sl * 8192>(f[0], a); mlp_lds_rd16<sl * 8192 + 4096>(f[1], a);
                };
                rd(std::integral_constant<int, 0>{});
                rd(std::integral_constant<int, 1>{});
                mlp_sfor<0, 8>([&](auto kidx) {
                    constexpr int k8 = decltype(kidx)::value;
                    if constexpr (k8 % 2 == 0) issue_step(x + 2, std::integral_constant<int, k8 / 2>{});
                    {
                        if constexpr (k8 + 2 < 8) rd(std::integral_constant<int, k8 + 2>{});
                        i32x4 (&f)[2] = fb[k8 % 3];
                        constexpr int ahead = 2 * ((7 - k8) < 2 ? (7 - k8) : 2);       // reads issued after this k-step's
                        if constexpr (ahead == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(f[0]), "+v"(f[1]) : : "memory");
                        else if constexpr (ahead == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(f[0]), "+v"(f[1]) : : "memory");
                        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]) : : "memory");
                        constexpr int kk = 8 * j + k8;                            // k-step of h: slab 2 j + k8 / 4
#pragma unroll
                        for (int t = 0; t < 2; ++t)
                            cur[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[t]), hf[kk], cur[t], 0, 0, 0);
                    }
                    // GELU groups of the previous chunk, dealt over the 8 NP k-steps of the interval
                    constexpr int n0 = (8 * j + k8) * 8 / (8 * NP), n1 = (8 * j + k8 + 1) * 8 / (8 * NP);
                    if constexpr (n1 > n0) gelu_group(std::integral_constant<int, n0>{});
                });
            });
        };
        for (int c = 0; c < nchunks + 2; c += 2) {
            chunk(c, sa, sb);
            if (c + 1 < nchunks + 2) chunk(c + 1, sb, sa);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                             // the consumers' closing barrier
        return;
    }

    // ---- consumer
    f32x16 out[NT];
#pragma unroll
    for (int jn = 0; jn < NT; ++jn)
#pragma unroll
        for (int i = 0; i < 16; ++i) out[jn][i] = 0.f;
    const unsigned ga0 = g0 - lds0 + (wr * 32) * 128;                             // this wave's rows of G (fo carries lds0 + the row)
    // groups of (k-step ks, half hh): 2 B fragments (+ the k-step's A fragment with hh = 0) and 2 MFMAs; the reads of group g + 1 go
    // out before the MFMAs of group g.  (Whole k-steps -- 5 reads, 4 MFMAs, two buffers of 5 fragments -- spilled at C = 384:
    // the 32 x C output tile is 192 of the 256 registers of a wave here.)
    constexpr int CD = C == 384 ? 1 : 2;                                          // groups the reads run ahead of the MFMAs (at C = 384 a
                                                                                  // third fragment buffer spills: out is 192 of 256 registers)
    i32x4 fa[2], fbb[CD + 1][2];
    // the two fill intervals: G of chunk 0 is complete at the end of interval 1; the consumer only keeps the ring schedule going
    for (int x = 0; x < 2 * NP; ++x) {
        open_step();
        mlp_sfor<0, 4>([&](auto kidx) { issue_step(x + 2, kidx); });
    }
    for (int c = 2; c < nchunks + 2; ++c) {                                       // consumes chunk c - 2
        const unsigned gab = ga0 + (c & 1) * MLPPC_G;                             // (c - 2) & 1
        mlp_sfor<0, NP>([&](auto jidx) {
            constexpr int j = decltype(jidx)::value;
            const int x = c * NP + j;
            open_step();
            const unsigned ub = ((x % 3) * 2 + 1) * UNITB;
            auto rd = [&](auto gidx) {
                constexpr int g = decltype(gidx)::value, ks = g >> 1, hh = g & 1;
                if constexpr (hh == 0) mlp_lds_rd16<0>(fa[ks & 1], fo[ks] + gab);
                const unsigned a = fo[ks] + ub;
                i32x4 (&f)[2] = fbb[g % (CD + 1)];
                mlp_lds_rd16<hh * 8192>(f[0], a); mlp_lds_rd16<hh * 8192 + 4096>(f[1], a);
            };
            mlp_sfor<0, CD>([&](auto gidx) { rd(gidx); });
            mlp_sfor<0, 8>([&](auto gidx) {
                constexpr int g = decltype(gidx)::value, ks = g >> 1, hh = g & 1;
                if constexpr (hh == 0) issue_step(x + 2, std::integral_constant<int, ks>{});
                {
                    if constexpr (g + CD < 8) rd(std::integral_constant<int, g + CD>{});
                    i32x4 (&f)[2] = fbb[g % (CD + 1)];
                    // reads issued after this group's: those of groups g + 1 .. g + CD (3 when a group opens a k-step, else 2)
                    constexpr int r1 = g + 1 < 8 ? (((g + 1) & 1) ? 2 : 3) : 0, r2 = (CD > 1 && g + 2 < 8) ? (((g + 2) & 1) ? 2 : 3) : 0;
                    constexpr int ahead = r1 + r2;
                    if constexpr (ahead == 5) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(fa[ks & 1]), "+v"(f[0]), "+v"(f[1]) : : "memory");
                    else if constexpr (ahead == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fa[ks & 1]), "+v"(f[0]), "+v"(f[1]) : : "memory");
                    else if constexpr (ahead == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa[ks & 1]), "+v"(f[0]), "+v"(f[1]) : : "memory");
                    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[ks & 1]), "+v"(f[0]), "+v"(f[1]) : : "memory");
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        out[4 * j + 2 * hh + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[ks & 1]), __builtin_bit_cast(bf16x8, f[i]),
                                                                                          out[4 * j + 2 * hh + i], 0, 0, 0);
                }
            });
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                                 // every wave is done with the ring: the epilogue stages in it

    mlp_epilogue<C, (C == 384 ? 6 : 8)>(p, out, reinterpret_cast<float*>(mlp_smem) + wr * (32 * ((C / 32) % 3 == 0 ? 96 : 64)), m0, lane);
}

