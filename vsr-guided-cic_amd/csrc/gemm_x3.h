// "f32x3": fp32-accurate "NT" GEMM on the bf16 matrix cores - the parity-mode default for launches the planner gives it.
//
//   C_p[m][n] = sum over segments g, k:  A_pg[row_g(m)][k] * W_pg[n][k]        fp32 operands, fp32 accumulation, S partial slabs
//
// Every fp32 operand element x is three bf16 terms
//     hi = bf16(x),  mid = bf16(x - hi),  lo = bf16(x - hi - mid)         (x = hi + mid + lo exactly: 3 x 8 = 24 mantissa bits)
// and a product a.b is accumulated in fp32 as  lo.hi + hi.lo + mid.mid + mid.hi + hi.mid + hi.hi  (six v_mfma_f32_32x32x16_bf16,
// smallest first); the dropped terms mid.lo, lo.mid, lo.lo are <= 2^-23 of the product - the size of the ONE rounding the fp32
// fma chain commits per product (measured against fp64 at K = 1000: rms 4.0e-7 vs 5.9e-7 for the chain).  Not bit-identical to
// the chain (a different summation order, like any other fp32 GEMM): the parity suite runs in both flavours (tests/conftest.py).
//
// Both operands are fp32 in memory (no copies) and are split by the mover waves on their way into LDS.  Round 3 measured the
// alternative the round-2 review asked for - operands that come as three bf16 PLANES in memory (weights split once per weight
// version, activations by their producers), so that the movers only move: 6 bytes per element instead of 4 through the CU's
// vector-memory path, which is what paces this kernel - and it was never faster (tools/gemm_bench, M = 500: 377 vs 357 us per
// step's GEMMs at best, 688 us with stream-K ranges; M = 100: 114 vs 112 us).  Not kept; DESIGN.md section 4 has the table.
//
// Structure (gemm_bf16.h's): 16 waves - waves 0-7 MULTIPLY (WM x WN waves, TM x TN 32x32 tiles each), waves 8-15 MOVE data
// (asynchronous global loads two k-tiles ahead -> split -> ds_write into the buffer the multipliers are not reading); one
// barrier per k-tile.  Workgroup tile 128 x BN x 32 with BN = 256 (TM = TN = 2), 128 (TM 2, TN 1) or 64 (TM = TN = 1): the
// narrow tiles exist for problems of <= 128 rows (greedy decoding, sampling, the per-step GEMMs of the training pass at batch
// 100, a data-parallel shard): one m-tile holds every row, so the number of tiles is the number of n-tiles, and 256-wide tiles
// would have to be cut into ~10 k pieces each to fill the CUs.  LDS: three bf16 planes of (128 + BN) rows x 32, rows unpadded
// (64 bytes), 16-byte chunk c of row r at position c ^ ((r >> 2) & 3) (conflict-free ds_read_b128 for the 32x32x16 operand
// map), double buffered: 147 / 98 / 74 KB.  Work decomposition: stream-K ranges (gemm_plan) or k-aligned pieces
// (gemm_plan_aligned); slab outputs; LDS-staged 16-byte epilogue stores.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_bf16.h"

namespace vsr {

constexpr int X3_BK = 32;
constexpr int X3_ROW = 32;                               // bf16 elements per LDS row (64 bytes, unpadded, XOR-swizzled chunks)
constexpr int X3_THREADS = 1024;

// x -> (hi, mid, lo) for two values at once: three packed bf16 pairs
__device__ __forceinline__ void split3(float a, float b, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    hi = pack_bf16(a, b);
    const float ah = __uint_as_float(hi << 16), bh = __uint_as_float(hi & 0xffff0000u);
    const float ar = a - ah, br = b - bh;                // exact (Sterbenz-like: hi is a's leading 8 bits)
    mid = pack_bf16(ar, br);
    const float am = __uint_as_float(mid << 16), bm = __uint_as_float(mid & 0xffff0000u);
    lo = pack_bf16(ar - am, br - bm);
}

constexpr size_t x3_lds_bytes(int TM, int TN) { return (size_t)2 * 3 * (128 + 32 * TN * (8 / (4 / TM))) * X3_ROW * sizeof(uint16_t); }

template <int TM, int TN>
__global__ __launch_bounds__(X3_THREADS)
void gemm_nt_x3_kernel(const GemmArgs args) {
    constexpr int WM = 4 / TM, WN = 8 / WM;                // multipliers: WM x WN = 8 waves
    constexpr int BM = 128, BN = 32 * TN * WN, BK = X3_BK;
    static_assert(32 * TM * WM == BM, "tile shape");
    constexpr int PLANE = (BM + BN) * X3_ROW;             // bf16 elements per plane
    constexpr int BUF = 3 * PLANE;                        // hi | mid | lo
    __shared__ __attribute__((aligned(16))) uint16_t smem[2 * BUF];

    const int G = args.G;
    const int g = gemm_wg_of_block(args);
    if (g >= G) return;
    const GemmRange rg = gemm_range(args, g);
    const int it0 = rg.it0, it1 = rg.it1;
    if (it0 >= it1) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const bool mover = wave >= 8;
    const int r = lane & 31, hh = lane >> 5;

    int c_prob = 0, c_tile = 0, c_left = 0, c_piece = 0;
    bool c_last = false;
    auto decode = [&](int it) __attribute__((always_inline)) {
        if (args.aligned) {                                // one piece of one tile: nothing to search
            c_prob = rg.prob; c_tile = rg.tile; c_piece = rg.piece;
            c_left = it1 - it;
            c_last = rg.piece == rg.split - 1;
            return it - (args.p[rg.prob].it_begin + rg.tile * args.p[rg.prob].ktiles);
        }
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

    // Epilogue of one tile piece: bands of 32 tile rows are staged in a k buffer by the multiplier waves that own them and leave
    // as 16-byte row stores issued by all 1024 threads; unused slabs of a finished tile get zeros the same way.
    constexpr int ST_LD = BN + 4;
    static_assert(32 * ST_LD * 4 <= BUF * 2, "staging band must fit one k buffer");
    f32x16 acc[TM][TN];
    auto flush = [&](auto MULT, float* stage) __attribute__((always_inline)) {
        const GemmProb& P = args.p[c_prob];
        const int m0 = (c_tile % P.tiles_m) * BM, n0 = (c_tile / P.tiles_m) * BN;
        float* C = P.C + (long long)c_piece * P.slab_stride;
        const int extra = c_last ? P.nslab - 1 - c_piece : 0;
        const bool vec_ok = ((P.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(P.C) & 15) == 0) && ((P.slab_stride & 3) == 0);
        constexpr int TPR = BN / 4;                        // threads per staged row
        constexpr int RPP = X3_THREADS / TPR;              // rows per store pass (16 / 32 / 64)
        const int c4 = (tid % TPR) * 4;
        const int n = n0 + c4;
        const int wm = wave / WN, wn = wave % WN;          // (multipliers)
        wait_loads<0>();                                   // stores share vmcnt with the asynchronous loads: start from an empty queue
#pragma unroll
        for (int band = 0; band < BM / 32; ++band) {
            if (m0 + band * 32 >= P.M) break;              // rows past the problem (a short m-tile): nothing to stage or store
            if constexpr (decltype(MULT)::value) {
#pragma unroll
                for (int ti = 0; ti < TM; ++ti)
                    if (wm * TM + ti == band) {
#pragma unroll
                        for (int tj = 0; tj < TN; ++tj)
#pragma unroll
                            for (int e = 0; e < 16; ++e)
                                stage[((e & 3) + 8 * (e >> 2) + 4 * hh) * ST_LD + wn * (32 * TN) + tj * 32 + r] = acc[ti][tj][e];
                    }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < (32 + RPP - 1) / RPP; ++i) {
                const int sr = tid / TPR + RPP * i;
                const int m = m0 + band * 32 + sr;
                if (sr < 32 && m < P.M && n < P.N) {
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
            __syncthreads();
        }
        wait_loads<0>();                                   // ... and leave it empty (the hand-written waits count loads only)
    };
    // end of a k-tile for BOTH kinds of waves: barrier, epilogue when the tile piece is complete (staged in the buffer the
    // multipliers have just finished with: the movers wrote the OTHER one during this k-tile)
    int cur = 0, it = it0;
    auto end_of_ktile = [&](auto MULT) __attribute__((always_inline)) {
        ++it;
        const bool piece_done = --c_left == 0;
        __syncthreads();
        if (piece_done) {
            flush(MULT, reinterpret_cast<float*>(smem + cur * BUF));
            if (it < it1) decode(it);
        }
        cur ^= 1;
        return piece_done;
    };

    const int kt0 = decode(it0);

    if (mover) {
        // ================================================================================================ movers
        const int ptid = tid - 512;
        const int lrow = ptid >> 3, lk = (ptid & 7) * 4;   // 8 lanes x float4 cover a row's k-tile (one 128-byte line), 64 rows per pass
        constexpr int LA = BM / 64, LB = BN / 64;          // loads per thread and k-tile
        f32x4_t ra[2][LA], rb[2][LB];                      // tile j (counted from it0) lives in register set j & 1
        bool stl[2] = {false, false};
        const float* pa[LA];
        const float* pb[LB];
        int l_prob = 0, l_tile = 0, l_tile_left = 0;
        int l_seg = 0, l_seg_left = 0, l_k = 0, l_K = 0;
        auto open_segment = [&](int sg, int first_tile) __attribute__((always_inline)) {
            // (readfirstlane: the indices are wave-uniform, but hipcc cannot always prove it and then copies the whole argument
            // struct to scratch to index it per lane)
            const GemmProb& P = args.p[__builtin_amdgcn_readfirstlane(l_prob)];
            const GemmSeg& S = P.seg[__builtin_amdgcn_readfirstlane(sg)];
            const int m0 = (l_tile % P.tiles_m) * BM, n0 = (l_tile / P.tiles_m) * BN;
            l_seg = sg;
            l_K = S.K;
            l_k = first_tile * BK;
            l_seg_left = (S.K + BK - 1) / BK - first_tile;
#pragma unroll
            for (int i = 0; i < LA; ++i) {
                int m = m0 + lrow + 64 * i;
                m = m < P.M ? m : P.M - 1;
                const long long row = S.a_idx ? (long long)S.a_idx[m] : (long long)m;
                pa[i] = S.A + row * S.lda + lk;
            }
#pragma unroll
            for (int i = 0; i < LB; ++i) {
                int n = n0 + lrow + 64 * i;
                n = n < P.N ? n : P.N - 1;
                pb[i] = S.W + (long long)n * S.ldw + lk;
            }
        };
        auto open_tile = [&](int prob, int tile, int kt) __attribute__((always_inline)) {
            l_prob = prob;
            l_tile = tile;
            const GemmProb& P = args.p[prob];
            l_tile_left = P.ktiles - kt;
            int sg = 0;
            while (sg < P.nseg - 1 && kt >= (P.seg[sg].K + BK - 1) / BK) { kt -= (P.seg[sg].K + BK - 1) / BK; ++sg; }
            open_segment(sg, kt);
        };
        int ko = 0;
        bool tail = false;
        auto advance = [&]() __attribute__((always_inline)) {
            if (l_tile_left == 0) {
                if (l_tile + 1 < args.p[l_prob].tiles_m * args.p[l_prob].tiles_n) open_tile(l_prob, l_tile + 1, 0);
                else open_tile(l_prob + 1, 0, 0);
            } else if (l_seg_left == 0) {
                open_segment(l_seg + 1, 0);
            }
            tail = !(l_k + lk < l_K);                      // K is a multiple of 4
            ko = tail ? 0 : l_k;
            l_k += BK;
            --l_seg_left;
            --l_tile_left;
        };
        auto issue = [&](auto S) __attribute__((always_inline)) {
            constexpr int s = decltype(S)::value;
            advance();
#pragma unroll
            for (int i = 0; i < LA; ++i) async_load16(ra[s][i], pa[i] + ko);
#pragma unroll
            for (int i = 0; i < LB; ++i) async_load16(rb[s][i], pb[i] + ko);
            stl[s] = tail;
        };
        auto landed_set = [&](auto S, bool other_in_flight) __attribute__((always_inline)) {
            constexpr int s = decltype(S)::value;
            if (other_in_flight) wait_loads<LA + LB>(); else wait_loads<0>();
#pragma unroll
            for (int i = 0; i < LA; ++i) landed(ra[s][i]);
#pragma unroll
            for (int i = 0; i < LB; ++i) landed(rb[s][i]);
        };
        // row R of the tile (A rows first, then W rows), this thread's 4 k's: 8 bytes per plane at chunk (lk / 8) ^ ((R >> 2) & 3)
        auto put = [&](uint16_t* buf, int R, f32x4_t v, bool zero) __attribute__((always_inline)) {
            if (zero) v = f32x4_t{0.f, 0.f, 0.f, 0.f};
            uint32_t h0, m0, l0, h1, m1, l1;
            split3(v.x, v.y, h0, m0, l0);
            split3(v.z, v.w, h1, m1, l1);
            const int pos = R * X3_ROW + 8 * ((lk >> 3) ^ ((R >> 2) & 3)) + (lk & 4);
            *reinterpret_cast<uint2*>(buf + pos) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(buf + PLANE + pos) = make_uint2(m0, m1);
            *reinterpret_cast<uint2*>(buf + 2 * PLANE + pos) = make_uint2(l0, l1);
        };
        auto store_tile = [&](auto S, int b) __attribute__((always_inline)) {
            constexpr int s = decltype(S)::value;
            uint16_t* buf = smem + b * BUF;
#pragma unroll
            for (int i = 0; i < LA; ++i) put(buf, lrow + 64 * i, ra[s][i], stl[s]);
#pragma unroll
            for (int i = 0; i < LB; ++i) put(buf, BM + lrow + 64 * i, rb[s][i], stl[s]);
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        // k-tile j:  [wait for tile j+1 (issued one k-tile ago; tile j+2 may stay in flight)] [tile j+1 -> three planes of the other
        //            buffer] [issue tile j+3 into the set just emptied] [barrier]
        open_tile(c_prob, c_tile, kt0);
        issue(S0{});                                       // tile 0
        landed_set(S0{}, false);
        store_tile(S0{}, 0);
        if (it0 + 1 < it1) issue(S1{});                    // tile 1
        if (it0 + 2 < it1) issue(S0{});                    // tile 2
        __syncthreads();                                   // buffer 0 is ready
        auto step = [&](auto S) __attribute__((always_inline)) {
            if (it + 1 < it1) {
                landed_set(S, it + 2 < it1);
                store_tile(S, cur ^ 1);
                if (it + 3 < it1) issue(S);
            }
            end_of_ktile(std::false_type{});
        };
        while (it < it1) {
            step(S1{});
            if (it < it1) step(S0{});
        }
    } else {
        // ================================================================================================ multipliers
        const int wm = wave / WN, wn = wave % WN;
        const int swz = (r >> 2) & 3;                      // rows 32 t + r of every subtile share it
        auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        };
        zero_acc();
        __syncthreads();                                   // buffer 0 is ready
        while (it < it1) {
            const uint16_t* base = smem + cur * BUF;
            const uint16_t* a_row = base + (wm * (32 * TM) + r) * X3_ROW;
            const uint16_t* b_row = base + (BM + wn * (32 * TN) + r) * X3_ROW;
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                const int ch = 8 * ((2 * kk + hh) ^ swz);  // lane (r, hh) reads k = 8 hh + 16 kk .. +7: chunk 2 kk + hh, swizzled
                bf16x8_t ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[i] = *reinterpret_cast<const bf16x8_t*>(a_row + i * 32 * X3_ROW + ch);
                    am[i] = *reinterpret_cast<const bf16x8_t*>(a_row + PLANE + i * 32 * X3_ROW + ch);
                    al[i] = *reinterpret_cast<const bf16x8_t*>(a_row + 2 * PLANE + i * 32 * X3_ROW + ch);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bh[j] = *reinterpret_cast<const bf16x8_t*>(b_row + j * 32 * X3_ROW + ch);
                    bm[j] = *reinterpret_cast<const bf16x8_t*>(b_row + PLANE + j * 32 * X3_ROW + ch);
                    bl[j] = *reinterpret_cast<const bf16x8_t*>(b_row + 2 * PLANE + j * 32 * X3_ROW + ch);
                }
                // smallest terms first, the leading product last; term-major over the TM x TN accumulators (consecutive MFMAs of one
                // accumulator are TM x TN issues apart; written accumulator-major hipcc alternates two accumulators: 1-3 % slower on
                // the 128 x 256 tile.  Issuing each product's LDS reads one product ahead by hand - a software pipeline with four live
                // plane operands, sched_barrier between the stages - measured no better: the SIMD's other multiplier wave hides them)
#define X3_TERM(X, Y)                                                                                   \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                     \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                 \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X[i], Y[j], acc[i][j], 0, 0, 0);
                X3_TERM(al, bh)
                X3_TERM(ah, bl)
                X3_TERM(am, bm)
                X3_TERM(am, bh)
                X3_TERM(ah, bm)
                X3_TERM(ah, bh)
#undef X3_TERM
            }
            if (end_of_ktile(std::true_type{})) zero_acc();
        }
    }
}

}  // namespace vsr
