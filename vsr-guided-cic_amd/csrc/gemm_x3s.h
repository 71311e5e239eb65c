// "f32x3" for SHORT problems (at most 128 rows): a weight-STREAMING kernel.
//
//   C_p[m][n] = sum over segments g, k:  A_pg[row_g(m)][k] * W_pg[n][k]        fp32 operands, fp32 accumulation, S partial slabs
//
// Same arithmetic as gemm_x3.h (every fp32 element = three bf16 terms, six bf16 MFMAs per product, smallest terms first), another
// shape of the work.  With M <= 128 one m-tile holds every row, a decoder timestep is 240 MB of weights against a few MB of
// activations, and the tiled kernels spend their launches on moving WEIGHT tiles through LDS (two barriers' worth of mover /
// multiplier hand-off per 32 k's for a tile that is used by ONE m-tile).  Here a weight element goes global -> register -> matrix
// core and is never staged:
//   * workgroup = 4 waves = one 128-column block of W over a k-aligned piece of K (gemm_plan_aligned with BN = 128); wave w owns two
//     strips of 16 weight rows, n0 + 32 w .. and + 16: lane (c = lane & 15, q = lane >> 4) loads W[n][k + 8 q .. + 7] of each (two
//     float4: the wave reads whole 128-byte lines of 16 rows), splits the eight values into three bf16x8 fragments - exactly the B
//     operand of v_mfma_f32_16x16x32_bf16 - and multiplies them with the MT 16-row tiles of A: 12 MT MFMAs per 32 k's, issued term by
//     term over the 2 MT independent accumulators (six back-to-back MFMAs on ONE accumulator ran at 43 cycles each instead of 16);
//     W is loaded PF k-tiles ahead through plain loads that hipcc counts exactly (no conditional loads in the steady state, a raw
//     s_barrier instead of __syncthreads(), which would wait for vmcnt(0) and empty the ring at every k-tile).
//   * A (MT 16 rows x 32 k per k-tile, shared by the four waves) is staged through LDS as three bf16 planes, double buffered,
//     one barrier per k-tile: 3 MT ds_read_b128 per wave and k-tile, rows XOR-swizzled as in gemm_x3.h.
//   * accumulators: 2 MT x 4 registers per lane; a k piece ends with MT x 64-byte-per-16-lanes stores into its slab (the k split
//     across workgroups is exact: split slabs, zero-filled up to the problem's slab count by the last piece).
// Per k-tile a workgroup consumes 8 KB of weights in ~6 MT x 16 cycles of MFMA per SIMD: at MT = 7 (batch 100) the chip streams
// weights at about the HBM rate when every CU has a workgroup; below that the launch is bound by the stream alone.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_x3.h"

namespace vsr {

constexpr int X3S_THREADS = 256;
constexpr int x3s_bn(int NS) { return 4 * 16 * NS; }        // columns per workgroup: NS 16-column strips of W per wave, 4 waves
constexpr int X3S_PF = 3;                                  // k-tiles of W (and A) in flight per wave

typedef float f32x4_acc __attribute__((ext_vector_type(4)));

template <int MT, int NS>
__global__ __launch_bounds__(X3S_THREADS)
void gemm_nt_x3s_kernel(const GemmArgs args) {
    constexpr int BK = X3_BK, ROWS = 16 * MT, X3S_BN = x3s_bn(NS);
    constexpr int PLANE = ROWS * X3_ROW;                   // bf16 elements per plane of one k-tile of A
    constexpr int NQ = (ROWS * 8 + X3S_THREADS - 1) / X3S_THREADS;   // float4 quads of A per thread and k-tile (8 per row)
    constexpr int LASTQ = ROWS * 8 - X3S_THREADS * (NQ - 1);         // threads that own a quad in the last pass
    __shared__ __attribute__((aligned(16))) uint16_t smem[2 * 3 * PLANE];

    const int G = args.G;
    const int g = gemm_wg_of_block(args);
    if (g >= G) return;
    const GemmRange rg = gemm_range(args, g);               // k-aligned plan: one piece of one 128-column block
    const int it0 = rg.it0, it1 = rg.it1;
    if (it0 >= it1) return;
    const GemmProb& P = args.p[__builtin_amdgcn_readfirstlane(rg.prob)];
    const int n0 = rg.tile * X3S_BN;                        // (tiles_m = 1: every row of the problem is in this block)
    const int kt0 = it0 - (P.it_begin + rg.tile * P.ktiles), nkt = it1 - it0;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;

    // ---- load cursor: k-tile after k-tile of the piece, W and A of the SAME k-tile together.  Its segment state (bases, this lane's
    // 32-bit row offsets, K) lives in registers and changes only when the cursor enters a segment - a uniform, rare branch; nothing in
    // the steady state reads the argument struct or indexes an array by a run-time value (the first versions did: hipcc put the
    // arrays into scratch, whose loads share vmcnt with the ring and drained it at every k-tile)
    const float* cW = nullptr;
    const float* cA = nullptr;
    int oW[NS], oA[NQ];                                     // n ldw + 8 q  /  row lda + this thread's k offset inside a k-tile
    int cK = 0, ck = 0, cseg = -1, cleft = 0;               // K of the segment, next k, segment index, k-tiles left in it
    auto open_segment = [&](int sg, int first_tile) __attribute__((always_inline)) {
        const GemmSeg& S = P.seg[__builtin_amdgcn_readfirstlane(sg)];
        cseg = sg;
        cK = S.K;
        ck = first_tile * BK;
        cleft = (S.K + BK - 1) / BK - first_tile;
        cW = S.W;
        cA = S.A;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            int n = n0 + 16 * (NS * wave + s) + c;
            n = n < P.N ? n : P.N - 1;
            oW[s] = n * S.ldw + 8 * q;
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int qi = tid + X3S_THREADS * i;
            int m = qi >> 3;
            m = m < P.M ? m : P.M - 1;                      // rows past M read row M - 1: their products are never stored
            const int row = S.a_idx ? S.a_idx[m] : m;
            oA[i] = row * S.lda + (qi & 7) * 4;
        }
    };
    {
        int kt = kt0, sg = 0;
        while (sg < P.nseg - 1 && kt >= (P.seg[sg].K + BK - 1) / BK) { kt -= (P.seg[sg].K + BK - 1) / BK; ++sg; }
        open_segment(sg, kt);
    }
    // the ring: slot u holds W (NS strips x 2 float4) and A (NQ float4) of one k-tile, issued together, consumed in issue order
    f32x4_t wq[X3S_PF][NS][2];
    f32x4_t aq[X3S_PF][NQ];
    int rem[X3S_PF];                                        // valid k's left in the slot's k-tile (tails are zeroed when consumed)
    int issued = 0;                                         // k-tiles of the piece the cursor has passed
    auto load_next = [&](int slot) __attribute__((always_inline)) {
        if (issued < nkt && cleft == 0) open_segment(cseg + 1, 0);       // (uniform; past the piece the last k-tile is re-read and ignored)
        const int k = issued < nkt ? ck : ck - BK;
        rem[slot] = cK - k;
        const bool z0 = !(k + 8 * q < cK), z1 = !(k + 8 * q + 4 < cK);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float* src = cW + oW[s];
            wq[slot][s][0] = *reinterpret_cast<const f32x4_t*>(src + (z0 ? -8 * q : k));
            wq[slot][s][1] = *reinterpret_cast<const f32x4_t*>(src + (z1 ? -8 * q : k + 4));
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int lk = ((tid + X3S_THREADS * i) & 7) * 4;
            aq[slot][i] = *reinterpret_cast<const f32x4_t*>(cA + oA[i] + (k + lk < cK ? k : -lk));
        }
        if (issued < nkt) { ck += BK; --cleft; ++issued; }
    };
    auto store_a = [&](int slot, int b) __attribute__((always_inline)) {
        uint16_t* buf = smem + b * 3 * PLANE;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int qi = tid + X3S_THREADS * i;
            const int R = qi >> 3, lk = (qi & 7) * 4;
            f32x4_t v = aq[slot][i];
            if (!(lk < rem[slot])) v = f32x4_t{0.f, 0.f, 0.f, 0.f};
            uint32_t h0, m0, l0, h1, m1, l1;
            split3(v.x, v.y, h0, m0, l0);
            split3(v.z, v.w, h1, m1, l1);
            const int pos = R * X3_ROW + 8 * ((lk >> 3) ^ ((R >> 2) & 3)) + (lk & 4);
            if (i + 1 < NQ || LASTQ == X3S_THREADS || tid < LASTQ) {
                *reinterpret_cast<uint2*>(buf + pos) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(buf + PLANE + pos) = make_uint2(m0, m1);
                *reinterpret_cast<uint2*>(buf + 2 * PLANE + pos) = make_uint2(l0, l1);
            }
        }
    };

    f32x4_acc acc[NS][MT];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[s][i] = f32x4_acc{0.f, 0.f, 0.f, 0.f};

    // one k-tile: the wave's W fragments x the MT row tiles of A in buffer b
    auto multiply = [&](int slot, int b) __attribute__((always_inline)) {
        bf16x8_t bh[NS], bm[NS], bl[NS];
        const bool z0 = !(8 * q < rem[slot]), z1 = !(8 * q + 4 < rem[slot]);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            f32x4_t w0 = wq[slot][s][0], w1 = wq[slot][s][1];
            if (z0) w0 = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (z1) w1 = f32x4_t{0.f, 0.f, 0.f, 0.f};
            uint32_t h[4], m[4], l[4];
            split3(w0.x, w0.y, h[0], m[0], l[0]); split3(w0.z, w0.w, h[1], m[1], l[1]);
            split3(w1.x, w1.y, h[2], m[2], l[2]); split3(w1.z, w1.w, h[3], m[3], l[3]);
            bh[s] = __builtin_bit_cast(bf16x8_t, make_uint4(h[0], h[1], h[2], h[3]));
            bm[s] = __builtin_bit_cast(bf16x8_t, make_uint4(m[0], m[1], m[2], m[3]));
            bl[s] = __builtin_bit_cast(bf16x8_t, make_uint4(l[0], l[1], l[2], l[3]));
        }
        const uint16_t* base = smem + b * 3 * PLANE;
        // the row tiles in two halves: the A fragments of a half (3 x 16 bytes per tile) stay in registers while the six terms run
        // over its NS x MH independent accumulators - smallest terms first, the leading product last (gemm_x3.h), the same order per
        // accumulator as everywhere else
        constexpr int MH = (MT + 1) / 2;
#define X3S_TERM(AF, BF)                                                                                              \
    _Pragma("unroll") for (int s = 0; s < NS; ++s)                                                                  \
        _Pragma("unroll") for (int i = 0; i < MH; ++i)                                                               \
            if (i0 + i < MT) acc[s][i0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AF[i], BF[s], acc[s][i0 + i], 0, 0, 0);
#pragma unroll
        for (int i0 = 0; i0 < MT; i0 += MH) {
            bf16x8_t ah[MH], am[MH], al[MH];
#pragma unroll
            for (int i = 0; i < MH; ++i) {
                const int R = 16 * ((i0 + i < MT) ? i0 + i : 0) + c;     // lane (c, q) reads row R, k = 8 q .. + 7: chunk q, swizzled
                const uint16_t* p = base + R * X3_ROW + 8 * (q ^ ((R >> 2) & 3));
                ah[i] = *reinterpret_cast<const bf16x8_t*>(p);
                am[i] = *reinterpret_cast<const bf16x8_t*>(p + PLANE);
                al[i] = *reinterpret_cast<const bf16x8_t*>(p + 2 * PLANE);
            }
            X3S_TERM(al, bh) X3S_TERM(ah, bl) X3S_TERM(am, bm) X3S_TERM(am, bh) X3S_TERM(ah, bm) X3S_TERM(ah, bh)
        }
#undef X3S_TERM
    };


    // Workgroup barrier of the k loop: the LDS writes of this wave are drained (lgkmcnt), the global loads of the ring are NOT
    auto lds_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    // ---- prologue: the ring filled with k-tiles 0 .. PF - 1; A of k-tile 0 into buffer 0
#pragma unroll
    for (int u = 0; u < X3S_PF; ++u) load_next(u);
    store_a(0, 0);
    __syncthreads();

    // ---- k loop, PF k-tiles per round (ring slots are compile-time).  k-tile j (slot u): multiply with W(j); put A(j + 1) - slot
    // u + 1, issued right behind W(j) - into the other buffer; refill slot u with k-tile j + PF.  Consumption follows the issue
    // order, no load of the steady state is conditional (past the piece the cursor re-reads its last k-tile), so hipcc's waitcnt
    // pass sees one queue on every path and waits with exact counts; the last partial round only consumes.
    int j0 = 0;
    for (; j0 + X3S_PF <= nkt; j0 += X3S_PF) {
#pragma unroll
        for (int u = 0; u < X3S_PF; ++u) {
            multiply(u, (j0 + u) & 1);
            store_a((u + 1) % X3S_PF, (j0 + u + 1) & 1);    // (after the last k-tile: a buffer nobody reads again)
            load_next(u);
            lds_barrier();
        }
    }
#pragma unroll
    for (int u = 0; u < X3S_PF - 1; ++u) {
        const int j = j0 + u;
        if (j < nkt) {                                      // (uniform)
            multiply(u, j & 1);
            if (j + 1 < nkt) store_a((u + 1) % X3S_PF, (j + 1) & 1);
            lds_barrier();
        }
    }

    // ---- epilogue: accumulator element e of strip s, row tile i is C[16 i + 4 q + e][n0 + 16 (NS wave + s) + c]
    {
        float* C = P.C + (long long)rg.piece * P.slab_stride;
        const int extra = rg.piece == rg.split - 1 ? P.nslab - rg.split : 0;      // slabs this problem's consumers read beyond its split: zeros
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int n = n0 + 16 * (NS * wave + s) + c;
            if (n < P.N) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int m = 16 * i + 4 * q + e;
                        if (m < P.M) {
                            float* dst = C + (long long)m * P.ldc + n;
                            *dst = acc[s][i][e];
                            for (int x = 1; x <= extra; ++x) dst[(long long)x * P.slab_stride] = 0.f;
                        }
                    }
            }
        }
    }
}

}  // namespace vsr
