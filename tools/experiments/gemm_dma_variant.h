// Experiment kept for tools/gemm_bench.hip only (NOT part of libvsrcap.so): the stream-K fp32 GEMM with LDS-DMA tile loads.
// Measured bit-identical to and as fast as the register-staged kernel (91.1 vs 91.7 TF/s on the step shapes).
#pragma once
#include "../../vsr-guided-cic_amd/csrc/gemm_f32.h"

namespace vsr {

// ------------------------------------------------------------------------------------------------------------------
// LDS-DMA variant: the same stream-K decomposition and epilogue, but tiles go HBM/L2 -> LDS directly
// (global_load_lds_dwordx4, no VGPR staging, no ds_write) into a ring of THREE stages, so the loads of tile i+2 are in
// flight while tile i is being multiplied (prefetch distance 2 at 2 workgroups per CU: 3 x 24 KB for 128x64 tiles).
//   * LDS rows are unpadded (32 floats = 8 chunks of 16 B, the DMA writes 1 KB per wave instruction linearly);
//     bank conflicts are avoided by an XOR swizzle applied to the SOURCE chunk and to the read address:
//     position p of row r holds chunk p ^ ((r >> 1) & 7)  (conflict-free for the ds_read_b128 lane groups).
//   * K tails and nothing else read a 16-byte block of zeros instead of the matrix (per-lane source address).
//   * one raw s_barrier per k-step, counted s_waitcnt vmcnt(N) (never 0 inside the loop except after an epilogue).
__device__ float g_gemm_zero16[4] = {0.f, 0.f, 0.f, 0.f};

// One LDS-DMA request: every lane sends its 16 bytes at gsrc to LDS byte address lds_dst + 16 * lane.  Written in
// asm on purpose: hipcc puts s_waitcnt vmcnt(0) in front of the next ds_read whenever it has SEEN a DMA in flight
// (guide, "Three .s-level traps"), which would serialise the ring; the waits are counted by hand in the k loop.
__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int TM, int TN, int WM = 2, int WN = 2>
__global__ __launch_bounds__(64 * WM * WN)
__attribute__((amdgpu_waves_per_eu(2, 2)))
void gemm_nt_f32_dma_kernel(const GemmArgs args) {
    constexpr int NT = 64 * WM * WN;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int LR = NT / 8;
    constexpr int LA = BM / LR, LB = BN / LR;
    constexpr int STAGE = (BM + BN) * GEMM_BK;            // floats per stage (unpadded rows)
    constexpr int NL = LA + LB;                           // DMA instructions per thread per tile
    static_assert(BM % 64 == 0 && BM % LR == 0 && BN % LR == 0, "tile shape");
    static_assert(64 * BN <= STAGE, "epilogue staging must fit one stage");
    __shared__ __attribute__((aligned(1024))) float smem[3 * STAGE];

    const int G = args.G;
    const int g = gemm_wg_of_block(args);
    if (g >= G) return;
    const int it0 = gemm_range_begin(g, args.total_iters, G);
    const int it1 = gemm_range_begin(g + 1, args.total_iters, G);
    if (it0 >= it1) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, hh = lane >> 5;
    const int lrow = tid >> 3;
    const int lc4 = 4 * ((tid & 7) ^ ((lrow >> 1) & 7));  // swizzled SOURCE chunk of this lane's LDS position
    const unsigned wave_u = (unsigned)__builtin_amdgcn_readfirstlane(wave);
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(__attribute__((address_space(3))) void*)smem);

    // ---- load cursor (runs two iterations ahead)
    const float* pa[LA];
    const float* pb[LB];
    int l_prob = 0, l_tile = 0, l_tile_left = 0, l_seg = 0, l_seg_left = 0, l_k = 0, l_K = 0;
    auto open_segment = [&](int sg, int first_tile) __attribute__((always_inline)) {
        const GemmProb& P = args.p[l_prob];
        const GemmSeg& S = P.seg[sg];
        const int m0 = (l_tile % P.tiles_m) * BM, n0 = (l_tile / P.tiles_m) * BN;
        l_seg = sg;
        l_K = S.K;
        l_k = first_tile * GEMM_BK;
        l_seg_left = (S.K + GEMM_BK - 1) / GEMM_BK - first_tile;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            int m = m0 + lrow + LR * i;
            m = m < P.M ? m : P.M - 1;
            const long long row = S.a_idx ? (long long)S.a_idx[m] : (long long)m;
            pa[i] = S.A + row * S.lda + lc4;
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            int n = n0 + lrow + LR * i;
            n = n < P.N ? n : P.N - 1;
            pb[i] = S.W + (long long)n * S.ldw + lc4;
        }
    };
    auto open_tile = [&](int prob, int tile, int kt) __attribute__((always_inline)) {
        l_prob = prob;
        l_tile = tile;
        const GemmProb& P = args.p[prob];
        l_tile_left = P.ktiles - kt;
        int sg = 0;
        while (sg < P.nseg - 1 && kt >= (P.seg[sg].K + GEMM_BK - 1) / GEMM_BK) { kt -= (P.seg[sg].K + GEMM_BK - 1) / GEMM_BK; ++sg; }
        open_segment(sg, kt);
    };
    auto issue_next = [&](int stage) __attribute__((always_inline)) {     // DMA the cursor's tile into `stage`, advance
        if (l_tile_left == 0) {
            if (l_tile + 1 < args.p[l_prob].tiles_m * args.p[l_prob].tiles_n) open_tile(l_prob, l_tile + 1, 0);
            else open_tile(l_prob + 1, 0, 0);
        } else if (l_seg_left == 0) {
            open_segment(l_seg + 1, 0);
        }
        const bool kin = l_k + lc4 < l_K;
        // wave-uniform LDS byte address of this wave's first 8-row group in the stage
        const unsigned base = lds0 + (unsigned)__builtin_amdgcn_readfirstlane(stage) * (STAGE * 4) + wave_u * (8 * GEMM_BK * 4);
#pragma unroll
        for (int i = 0; i < LA; ++i) glds16(kin ? pa[i] + l_k : g_gemm_zero16, base + (LR * i) * (GEMM_BK * 4));
#pragma unroll
        for (int i = 0; i < LB; ++i) glds16(kin ? pb[i] + l_k : g_gemm_zero16, base + (BM + LR * i) * (GEMM_BK * 4));
        l_k += GEMM_BK;
        --l_seg_left;
        --l_tile_left;
    };

    // ---- compute-side bookkeeping (identical to the register-staged kernel)
    int c_prob = 0, c_tile = 0, c_left = 0, c_piece = 0;
    bool c_last = false;
    auto decode = [&](int it) __attribute__((always_inline)) {
        int p = 0;
#pragma unroll
        for (int i = 1; i < 4; ++i)
            if (i < args.nprob && it >= args.p[i].it_begin) p = i;
        const GemmProb& P = args.p[p];
        const int local = it - P.it_begin;
        c_prob = p;
        c_tile = local / P.ktiles;
        const int kt = local - c_tile * P.ktiles;
        const int tile_base = it - kt;
        const int g_first = (int)((((long long)tile_base + 1) * G - 1) / args.total_iters);
        c_piece = g - g_first;
        const int rem = P.ktiles - kt;
        c_left = rem < it1 - it ? rem : it1 - it;
        c_last = (c_left == rem);
        return kt;
    };

    constexpr int ST_LD = (64 * (BN + 4) <= STAGE) ? BN + 4 : BN;
    auto flush = [&](const f32x16 (&acc)[TM][TN], float* stage) __attribute__((always_inline)) {
        const GemmProb& P = args.p[c_prob];
        const int m0 = (c_tile % P.tiles_m) * BM, n0 = (c_tile / P.tiles_m) * BN;
        float* C = P.C + (long long)c_piece * P.slab_stride;
        const int extra = c_last ? P.nslab - 1 - c_piece : 0;
        const bool vec_ok = ((P.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(P.C) & 15) == 0) && ((P.slab_stride & 3) == 0);
        constexpr int TPR = BN / 4;
        constexpr int RPP = NT / TPR;
        const int c4 = (tid % TPR) * 4;
        const int n = n0 + c4;
        __builtin_amdgcn_s_barrier();                      // everybody is done reading the stage that becomes the staging buffer
#pragma unroll
        for (int band = 0; band < BM / 64; ++band) {
#pragma unroll
            for (int ti = 0; ti < TM; ++ti) {
                const int trow = wm * TM + ti;
                if ((trow >> 1) == band) {
#pragma unroll
                    for (int tj = 0; tj < TN; ++tj)
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            stage[((trow & 1) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh) * ST_LD + wn * (32 * TN) + tj * 32 + r] = acc[ti][tj][e];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int i = 0; i < 64 / RPP; ++i) {
                const int sr = tid / TPR + RPP * i;
                const int m = m0 + band * 64 + sr;
                if (m < P.M && n < P.N) {
                    const float4 v = *reinterpret_cast<const float4*>(stage + sr * ST_LD + c4);
                    float* dst = C + (long long)m * P.ldc + n;
                    if (vec_ok && n + 3 < P.N) {
                        *reinterpret_cast<float4*>(dst) = v;
                        for (int x = 1; x <= extra; ++x)
                            *reinterpret_cast<float4*>(dst + (long long)x * P.slab_stride) = make_float4(0.f, 0.f, 0.f, 0.f);
                    } else {
                        const float vv[4] = {v.x, v.y, v.z, v.w};
                        for (int q = 0; q < 4; ++q)
                            if (n + q < P.N) {
                                dst[q] = vv[q];
                                for (int x = 1; x <= extra; ++x) dst[(long long)x * P.slab_stride + q] = 0.f;
                            }
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // stores share the VM counter with the DMA: restart the count
    };

    {
        const int kt = decode(it0);
        open_tile(c_prob, c_tile, kt);
    }
    int issued = it0;                                      // next iteration whose tile has not been requested yet
    issue_next(0);
    ++issued;
    if (issued < it1) { issue_next(1); ++issued; }
    int st = 0;                                            // stage of the tile being multiplied
    for (int it = it0; it < it1;) {
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        const int n_it = c_left;
        for (int j_it = 0; j_it < n_it; ++j_it, ++it) {
            // tile `it` has landed once at most the younger tile's NL requests are outstanding
            if (issued > it + 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                  // ... for every wave; and stage (st + 2) % 3 is free again
            int st2 = st + 2;
            st2 = st2 >= 3 ? st2 - 3 : st2;
            if (issued < it1) { issue_next(st2); ++issued; }
            const float* a_base = smem + st * STAGE + (wm * (32 * TM) + r) * GEMM_BK;
            const float* b_base = smem + st * STAGE + BM * GEMM_BK + (wn * (32 * TN) + r) * GEMM_BK;
            const int sw = (r >> 1) & 7;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                float4 av[TM], bv[TN];
                const int ch = 4 * (((2 * kk + hh) ^ sw));
#pragma unroll
                for (int i = 0; i < TM; ++i) av[i] = *reinterpret_cast<const float4*>(a_base + i * 32 * GEMM_BK + ch);
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[j] = *reinterpret_cast<const float4*>(b_base + j * 32 * GEMM_BK + ch);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, bv[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, bv[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, bv[j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, bv[j].w, acc[i][j], 0, 0, 0);
                    }
            }
            st = st + 1 >= 3 ? 0 : st + 1;
        }
        // the stage just multiplied from is (st + 2) % 3: the two others hold / receive the next tiles
        int sfree = st + 2;
        sfree = sfree >= 3 ? sfree - 3 : sfree;
        flush(acc, smem + sfree * STAGE);
        if (it < it1) decode(it);
    }
}

}  // namespace vsr
