// fp32-accurate GEMM on the bf16 matrix cores ("3 x bf16" split), same stream-K decomposition, segments, gathers and
// epilogue as gemm_nt_f32_kernel (gemm_f32.h).  Every fp32 operand is split into three bf16 terms
//     x = hi + mid + lo        (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): 3 x 8 = 24 mantissa bits)
// when its tile is written to LDS, and x*y is accumulated in fp32 from the six partial products hi*hi, hi*mid, mid*hi,
// hi*lo, mid*mid, lo*hi (the three dropped ones are <= 2^-24 |xy| each).  bf16 products are exact in fp32, so the
// result differs from an fp32 fma chain by about one fp32 rounding per product - the same order as a different
// summation order.  v_mfma_f32_32x32x16_bf16 does 16 k per 32 cycles: six of them replace eight fp32 MFMAs of 64
// cycles (192 vs 512 cycles per 16 k).
#pragma once
#include "../../vsr-guided-cic_amd/csrc/gemm_f32.h"

namespace vsr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
    const __bf16 bh = (__bf16)x;
    const float r1 = x - (float)bh;
    const __bf16 bm = (__bf16)r1;
    const float r2 = r1 - (float)bm;
    const __bf16 bl = (__bf16)r2;
    h = __builtin_bit_cast(unsigned short, bh);
    m = __builtin_bit_cast(unsigned short, bm);
    l = __builtin_bit_cast(unsigned short, bl);
}
__device__ __forceinline__ void split3x4(const float4& v, uint2& p0, uint2& p1, uint2& p2) {
    unsigned h[4], m[4], l[4];
    split3(v.x, h[0], m[0], l[0]); split3(v.y, h[1], m[1], l[1]); split3(v.z, h[2], m[2], l[2]); split3(v.w, h[3], m[3], l[3]);
    p0 = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
    p1 = make_uint2(m[0] | (m[1] << 16), m[2] | (m[3] << 16));
    p2 = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
}

template <int TM, int TN, int WM = 2, int WN = 2>
__global__ __launch_bounds__(64 * WM * WN)
__attribute__((amdgpu_waves_per_eu(1, 2)))
void gemm_nt_bf16x3_kernel(const GemmArgs args) {
    constexpr int NT = 64 * WM * WN;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int LR = NT / 8;                           // tile rows covered by one load pass of the workgroup
    constexpr int LA = BM / LR, LB = BN / LR;            // float4 loads per thread per k-tile
    static_assert(BM % 64 == 0 && BM % LR == 0 && BN % LR == 0, "tile shape");
    // one stage = 3 planes (hi, mid, lo) x (BM + BN) rows x 32 bf16 (64 B, unpadded, XOR-swizzled 16-byte chunks)
    constexpr int PLANE = (BM + BN) * 64;                 // bytes
    constexpr int STAGE_B = 3 * PLANE;
    __shared__ __attribute__((aligned(16))) unsigned char smem_b[2 * STAGE_B];
    static_assert(64 * (BN + 4) * 4 <= STAGE_B, "epilogue staging must fit one stage");

    const int G = args.G;
    const int g = gemm_wg_of_block(args);      // grid = 8 * ceil(G / 8)
    if (g >= G) return;
    const int it0 = gemm_range_begin(g, args.total_iters, G);
    const int it1 = gemm_range_begin(g + 1, args.total_iters, G);
    if (it0 >= it1) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, hh = lane >> 5;
    const int lrow = tid >> 3;          // 0..LR-1: 8 threads cover one 128-byte row segment
    const int lc4 = (tid & 7) * 4;      // float offset inside the k-tile

    // ------------------------------------------------------------------ load cursor (runs one iteration ahead)
    float4 ra[LA], rb[LB];
    const float* pa[LA];
    const float* pb[LB];
    int l_prob = 0, l_tile = 0, l_tile_left = 0;          // problem, local tile id, k-tiles left in the tile
    int l_seg = 0, l_seg_left = 0, l_k = 0, l_K = 0;      // segment, k-tiles left in it, next k, its K
    auto open_segment = [&](int s, int first_tile) __attribute__((always_inline)) {
        const GemmProb& P = args.p[l_prob];
        const GemmSeg& S = P.seg[s];
        const int m0 = (l_tile % P.tiles_m) * BM, n0 = (l_tile / P.tiles_m) * BN;
        l_seg = s;
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
    auto open_tile = [&](int prob, int tile, int kt) __attribute__((always_inline)) {     // position the cursor on k-tile kt of a tile
        l_prob = prob;
        l_tile = tile;
        const GemmProb& P = args.p[prob];
        l_tile_left = P.ktiles - kt;
        int s = 0;
        while (s < P.nseg - 1 && kt >= (P.seg[s].K + GEMM_BK - 1) / GEMM_BK) { kt -= (P.seg[s].K + GEMM_BK - 1) / GEMM_BK; ++s; }
        open_segment(s, kt);
    };
    auto load_next = [&]() __attribute__((always_inline)) {
        if (l_tile_left == 0) {                            // wave-uniform, once per tile
            if (l_tile + 1 < args.p[l_prob].tiles_m * args.p[l_prob].tiles_n) open_tile(l_prob, l_tile + 1, 0);
            else open_tile(l_prob + 1, 0, 0);
        } else if (l_seg_left == 0) {
            open_segment(l_seg + 1, 0);
        }
        const bool kin = l_k + lc4 < l_K;                  // K tail: read a valid address, store zeros
#if defined(GEMM_L1HOT)
        const int ko = 0;                                  // diagnostics: every k-tile re-reads the same (L1-resident) lines
#else
        const int ko = kin ? l_k : 0;
#endif
#pragma unroll
        for (int i = 0; i < LA; ++i) ra[i] = *reinterpret_cast<const float4*>(pa[i] + ko);
#pragma unroll
        for (int i = 0; i < LB; ++i) rb[i] = *reinterpret_cast<const float4*>(pb[i] + ko);
        // The tail is zeroed when the tile is written to LDS, NOT here: touching the load destinations now would
        // put an s_waitcnt right behind the loads (hipcc hoisted a vmcnt(4) out of the branch: one exposed L2
        // round trip per k-step).  store_tile() re-derives the tail predicate from the cursor (l_k has advanced by BK).
        l_k += GEMM_BK;
        --l_seg_left;
        --l_tile_left;
    };
    auto store_tile = [&](int buf) __attribute__((always_inline)) {
        if (!(l_k - GEMM_BK + lc4 < l_K)) {                // K tail of the tile in flight (l_k has advanced by BK): zeros
#pragma unroll
            for (int i = 0; i < LA; ++i) ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < LB; ++i) rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        unsigned char* st = smem_b + buf * STAGE_B;
        const int q = tid & 7;                             // this thread's 4 k-values = 8 bytes of chunk q >> 1
#pragma unroll
        for (int i = 0; i < LA + LB; ++i) {
            const float4 v = i < LA ? ra[i] : rb[i - LA];
            const int row = (i < LA ? 0 : BM) + lrow + LR * (i < LA ? i : i - LA);      // row inside the (A ; B) stack
            const int off = row * 64 + ((((q >> 1) ^ ((row >> 2) & 3))) << 4) + (q & 1) * 8;
            uint2 p0, p1, p2;
            split3x4(v, p0, p1, p2);
            *reinterpret_cast<uint2*>(st + off) = p0;
            *reinterpret_cast<uint2*>(st + PLANE + off) = p1;
            *reinterpret_cast<uint2*>(st + 2 * PLANE + off) = p2;
        }
    };

    // ------------------------------------------------------------------ compute-side tile bookkeeping
    int c_prob = 0, c_tile = 0, c_left = 0, c_piece = 0;   // iterations left before this tile's piece is flushed
    bool c_last = false;                                   // this piece completes the tile
    auto decode = [&](int it) __attribute__((always_inline)) {                            // global iteration -> tile, piece, #iterations here
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
        // first workgroup whose range contains the tile's first iteration
        const int g_first = (int)((((long long)tile_base + 1) * G - 1) / args.total_iters);
        c_piece = g - g_first;
        const int rem = P.ktiles - kt;
        c_left = rem < it1 - it ? rem : it1 - it;
        c_last = (c_left == rem);
        return kt;
    };

    // Epilogue of one tile piece.  The accumulator (C/D layout: col = lane & 31, row = (e & 3) + 8 (e >> 2) +
    // 4 (lane >> 5)) is transposed through the LDS buffer the k loop has just finished with, so that the tile
    // leaves as 4 x 16-byte stores per thread (full 256-byte rows) instead of 16 dword stores with per-element
    // address arithmetic; unused slabs of a finished tile get zeros the same way.
    constexpr int ST_LD = BN + 4;
    auto flush = [&](const f32x16 (&acc)[TM][TN], float* stage) __attribute__((always_inline)) {
        const GemmProb& P = args.p[c_prob];
        const int m0 = (c_tile % P.tiles_m) * BM, n0 = (c_tile / P.tiles_m) * BN;
        float* C = P.C + (long long)c_piece * P.slab_stride;
        const int extra = c_last ? P.nslab - 1 - c_piece : 0;     // unused slabs of a finished tile: zeros
        const bool vec_ok = ((P.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(P.C) & 15) == 0) && ((P.slab_stride & 3) == 0);
        constexpr int TPR = BN / 4;                        // threads per staged row
        constexpr int RPP = NT / TPR;                      // rows per store pass
        const int c4 = (tid % TPR) * 4;
        const int n = n0 + c4;
#pragma unroll
        for (int band = 0; band < BM / 64; ++band) {       // 64 tile rows per pass through the staging buffer
#pragma unroll
            for (int ti = 0; ti < TM; ++ti) {
                const int trow = wm * TM + ti;             // 32-row tile index of this wave's tile row ti
                if ((trow >> 1) == band) {
#pragma unroll
                    for (int tj = 0; tj < TN; ++tj)
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            stage[((trow & 1) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh) * ST_LD + wn * (32 * TN) + tj * 32 + r] = acc[ti][tj][e];
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 64 / RPP; ++i) {
                const int sr = tid / TPR + RPP * i;        // staged row 0..63
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
            __syncthreads();                               // the next pass / store_tile() overwrites `stage`
        }
    };

#if defined(GEMM_STAMP)
    const unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
#endif
    {
        const int kt = decode(it0);
        open_tile(c_prob, c_tile, kt);
    }
    load_next();
    store_tile(0);
    __syncthreads();
    int cur = 0;
    // Outer loop: one tile piece; inner loop: its k iterations.  The accumulator is only ever touched by MFMAs
    // inside the inner loop, so it stays in the accumulator registers (a VALU read/zero of it inside the k loop
    // made hipcc shuttle all 16 registers through v_accvgpr_read/write and drain the MFMA pipe every iteration).
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
            const bool more = it + 1 < it1;
            if (more && GEMM_ABLATE < 1) load_next();      // global -> registers, in flight during the MFMAs
            const unsigned char* st = smem_b + cur * STAGE_B;
            const int sw = (r >> 2) & 3;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {               // two 16-deep MFMA steps per 32-deep k-tile
                const int ch = ((2 * ks + hh) ^ sw) << 4;
                bf16x8 a[TM][3], b[TN][3];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        a[i][p] = *reinterpret_cast<const bf16x8*>(st + p * PLANE + (wm * (32 * TM) + i * 32 + r) * 64 + ch);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        b[j][p] = *reinterpret_cast<const bf16x8*>(st + p * PLANE + (BM + wn * (32 * TN) + j * 32 + r) * 64 + ch);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        // x*y ~= sum of the six bf16 partial products with i+j <= 2, smallest first, fp32 accumulate
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
                    }
            }
            if (more && GEMM_ABLATE < 2) {
                store_tile(cur ^ 1);                       // other buffer: nobody reads it in this iteration
                __syncthreads();
                cur ^= 1;
            }
        }
        if (GEMM_ABLATE < 3) flush(acc, reinterpret_cast<float*>(smem_b + (cur ^ 1) * STAGE_B));
        else asm volatile("" ::"v"(acc[0][0][0]), "v"(acc[TM - 1][TN - 1][15]));
        if (it < it1) decode(it);
    }
#if defined(GEMM_STAMP)
    if (args.dbg && tid == 0) {
        args.dbg[4 * g] = __builtin_amdgcn_s_memtime() - st0;
        args.dbg[4 * g + 1] = __builtin_amdgcn_s_memrealtime() - sr0;
        args.dbg[4 * g + 2] = sr0;
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        args.dbg[4 * g + 3] = xcc;
    }
#endif
}


}  // namespace vsr
