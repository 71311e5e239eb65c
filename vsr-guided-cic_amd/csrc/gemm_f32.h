// Grouped "NT" fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain).
//
//   C_p[m][n] = sum over segments g, k:  A_pg[row_g(m)][k] * W_pg[n][k]        (returned as S partial slabs)
//
// A launch carries up to 4 independent problems (e.g. h1 -> [W1_hg | att_ha] and s_t -> [s_fc | att_sa]);
// each problem sums up to 3 K-segments so that the reference's torch.cat([h2, vbar, x]) (step :147, :176)
// is never materialised: every segment reads its own activation matrix and a column window of the
// reference-layout [out, in] weight, in place.  Activation rows may be gathered through an int32 index
// (embedding rows, beam parents).
//
// Decomposition: STREAM-K.  The decoder GEMMs are skinny (M = 100..500 rows against N = 512..10000), so a
// tile-per-workgroup grid leaves the last round of workgroups mostly idle (measured: 2272 tiles on 1024
// resident slots = 74 % of the achievable rate).  Instead the launch has G resident workgroups and the
// linearised iteration space  sum_p tiles_p * ktiles_p  (one iteration = one 64x64x32 MFMA step) is cut into G
// equal contiguous ranges.  A range that ends inside a tile leaves a partial sum; every tile therefore has
// between 1 and S pieces, piece j goes to slab j and the workgroup that finishes the tile zero-fills the slabs
// it did not use.  Consumers add the S slabs in index order: deterministic, no atomics, no inter-workgroup
// synchronisation, no memset.
//
// Tile: 64 x 64 x 32, 256 threads = 4 waves (2 x 2), one 32x32 accumulator per wave.
// LDS rows are padded to 36 floats: ds_write_b128 / ds_read_b128 are bank-conflict free (guide LDS table).
// Lane (r = lane & 31, h = lane >> 5) reads A[r][8*kk + 4h .. +3] with one ds_read_b128 and feeds the four
// floats to four MFMAs; A and B use the same k permutation, so the sum is unchanged.
// Workgroup ranges are dealt XCD-contiguously (blocks b and b+8 share an XCD), so the m-tiles of one weight
// n-tile run on one XCD back to back and the weight tile is fetched from HBM once.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmSeg {
    const float* A;      // (rows, lda) activations
    const int* a_idx;    // optional row gather: row m reads A + a_idx[m] * lda
    const float* W;      // (N, ldw) weight window start (already offset to the segment's first column)
    int lda;
    int ldw;
    int K;               // multiple of 4
    int exp_idx;         // f16x2 kernels only (gemm_h2.h): slots of GemmArgs::exps with the power-of-two scale exponents of W's image (low 16 bits) and of A's bound (high 16 bits)
    const uint16_t* A16; // bf16 kernel only, optional: a bf16 image of A (same rows, same lda) written by A's producer
};

struct GemmProb {
    GemmSeg seg[3];
    float* C;            // (nslab, M, ldc)
    long long slab_stride;
    int nseg;
    int M;
    int N;
    int ldc;
    int tiles_m;
    int tiles_n;
    int ktiles;          // 32-wide k-tiles over all segments
    int it_begin;        // first global iteration of this problem
    int g_begin;         // aligned plan only: first workgroup of this problem ...
    int split;           // ... and the number of equal k pieces every one of its tiles is cut into
    int nslab;           // slabs THIS problem's tiles write (<= GemmArgs::nslab): its consumers add exactly these
};

struct GemmArgs {
    GemmProb p[4];
    int nprob;
    int total_iters;
    int G;               // workgroups with a non-empty range (1 <= G <= total_iters); grid = 8 * ceil(G / 8)
    int nslab;           // slabs per tile (S)
    int aligned;         // 0: stream-K ranges (gemm_plan); 1: one k-aligned piece of one tile per workgroup (gemm_plan_aligned)
    const int* exps;     // f16x2 kernels only: device table of scale exponents (GemmSeg::w_exp / a_exp index it)
    int xcd_chunk;       // 0: ceil(G / 8) workgroups per XCD.  > 0 (aligned plan, gemm_plan_aligned): the workgroups are dealt in whole GROUPS of
    int xcd_unit;        // xcd_unit (= tiles_m) consecutive numbers, xcd_lo groups per XCD and one more on the first xcd_extra XCDs;
    int xcd_lo, xcd_extra;   // xcd_chunk = the largest per-XCD count = xcd_unit (xcd_lo + (xcd_extra > 0))
};

// Workgroup number of this block.  Blocks are dispatched round-robin over the 8 XCDs (block b runs on XCD b & 7); XCD x takes the
// CONTIGUOUS workgroup numbers [x chunk, (x + 1) chunk): neighbouring tiles share their operand windows in that XCD's L2.
// Returns G (= "no work") for the padding blocks of the grid.
__device__ __forceinline__ int gemm_wg_of_block(const GemmArgs& a) {
    const int i = blockIdx.x >> 3, x = blockIdx.x & 7;
    if (a.xcd_chunk == 0) {
        const int ch = (a.G + 7) >> 3;
        return x * ch + i;                     // (>= G for the padding blocks)
    }
    const int mine = a.xcd_unit * (a.xcd_lo + (x < a.xcd_extra ? 1 : 0));
    const int first = a.xcd_unit * (x * a.xcd_lo + (x < a.xcd_extra ? x : a.xcd_extra));
    return i < mine ? first + i : a.G;
}
inline int gemm_grid(const GemmArgs& a) { return 8 * (a.xcd_chunk ? a.xcd_chunk : (a.G + 7) / 8); }

// tile index -> tile origin: m fastest (the m-tiles of a weight n-tile are neighbours)
__device__ __forceinline__ void gemm_tile_origin(const GemmProb& P, int tile, int BM, int BN, int& m0, int& n0) {
    m0 = (tile % P.tiles_m) * BM; n0 = (tile / P.tiles_m) * BN;
}

constexpr int GEMM_BK = 32;
constexpr int GEMM_LDS = GEMM_BK + 4;

__device__ __host__ inline int gemm_range_begin(int g, int total, int G) { return (int)(((long long)g * total) / G); }

// the k-tile range [it0, it1) of workgroup g (both plans) and, for the aligned plan, its problem / tile / piece
struct GemmRange { int it0, it1, prob, tile, piece, split; };
__device__ __forceinline__ GemmRange gemm_range(const GemmArgs& args, int g) {
    GemmRange r;
    if (!args.aligned) {
        r.it0 = gemm_range_begin(g, args.total_iters, args.G);
        r.it1 = gemm_range_begin(g + 1, args.total_iters, args.G);
        r.prob = r.tile = r.piece = r.split = 0;
        return r;
    }
    int p = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < args.nprob && g >= args.p[i].g_begin) p = i;
    const GemmProb& P = args.p[p];
    const int local = g - P.g_begin, tiles = P.tiles_m * P.tiles_n;
    r.prob = p;
    r.piece = local / tiles;
    r.tile = local - r.piece * tiles;
    r.split = P.split;
    const int base = P.it_begin + r.tile * P.ktiles;
    r.it0 = base + (int)(((long long)r.piece * P.ktiles) / P.split);
    r.it1 = base + (int)(((long long)(r.piece + 1) * P.ktiles) / P.split);
    return r;
}


// TM x TN 32x32 MFMA tiles per wave, WM x WN waves per workgroup; workgroup tile = (32 TM WM) x (32 TN WN)
// Occupancy is LDS-bound (160 KB / (2 (BM + BN) 36 4 B) workgroups per CU); telling the register allocator so
// keeps it from spilling the in-flight tile to scratch in pursuit of waves the LDS could never host.
constexpr int gemm_waves_per_eu(int TM, int TN, int WM, int WN) {
    const int lds = 2 * (32 * TM * WM + 32 * TN * WN) * 36 * 4;
    const int wgs = 163840 / lds;
    const int w = wgs * WM * WN / 4;
    return w < 1 ? 1 : (w > 8 ? 8 : w);
}

template <int TM, int TN, int WM = 2, int WN = 2>
__global__ __launch_bounds__(64 * WM * WN)
__attribute__((amdgpu_waves_per_eu(gemm_waves_per_eu(TM, TN, WM, WN), gemm_waves_per_eu(TM, TN, WM, WN))))
void gemm_nt_f32_kernel(const GemmArgs args) {
    constexpr int NT = 64 * WM * WN;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int LR = NT / 8;                           // tile rows covered by one load pass of the workgroup
    constexpr int LA = BM / LR, LB = BN / LR;            // float4 loads per thread per k-tile
    static_assert(BM % 64 == 0 && BM % LR == 0 && BN % LR == 0, "tile shape");
    __shared__ float smem[2 * (BM + BN) * GEMM_LDS];
    auto sA = [&](int buf) { return smem + buf * (BM + BN) * GEMM_LDS; };
    auto sB = [&](int buf) { return smem + buf * (BM + BN) * GEMM_LDS + BM * GEMM_LDS; };

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
        int m0, n0;
        gemm_tile_origin(P, l_tile, BM, BN, m0, n0);
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
        const int ko = kin ? l_k : 0;
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
#pragma unroll
        for (int i = 0; i < LA; ++i)
            *reinterpret_cast<float4*>(sA(buf) + (lrow + LR * i) * GEMM_LDS + lc4) = ra[i];
#pragma unroll
        for (int i = 0; i < LB; ++i)
            *reinterpret_cast<float4*>(sB(buf) + (lrow + LR * i) * GEMM_LDS + lc4) = rb[i];
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
        int m0, n0;
        gemm_tile_origin(P, c_tile, BM, BN, m0, n0);
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
            if (more) load_next();                         // global -> registers, in flight during the MFMAs
            const float* a_base = sA(cur) + (wm * (32 * TM) + r) * GEMM_LDS + 4 * hh;
            const float* b_base = sB(cur) + (wn * (32 * TN) + r) * GEMM_LDS + 4 * hh;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                float4 av[TM], bv[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) av[i] = *reinterpret_cast<const float4*>(a_base + i * 32 * GEMM_LDS + kk * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[j] = *reinterpret_cast<const float4*>(b_base + j * 32 * GEMM_LDS + kk * 8);
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
            if (more) {
                store_tile(cur ^ 1);                       // other buffer: nobody reads it in this iteration
                __syncthreads();
                cur ^= 1;
            }
        }
        flush(acc, sA(cur ^ 1));
        if (it < it1) decode(it);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// ROWS-16 variant for short problems (M <= 128 per m-tile: greedy decoding, sampling, the per-step GEMMs of the
// training pass at batch 100, a data-parallel shard of 13 images).
//
// The 32x32 MFMA tile pads M = 100 to 128 rows (28 % of the matrix work on zeros) and a 64x64 workgroup tile loads one
// byte per 16 flop - and the k loop of these kernels is paced by the CU's vector-memory issue path, not by the matrix
// pipe.  Here the matrix instruction is v_mfma_f32_16x16x4_f32 (same rate as 32x32x2, exact fp32),
// rows come in units of 16 (100 -> 112: 11 % padding), and ONE workgroup holds every row of the m-tile against
// BN = 64 TN weight rows: a 112 x 128 x 32 step loads 30 KB for 917 kflop (30 flop / byte, the 128x128 figure).
//   workgroup = 8 waves: 4 side by side along n (wave & 3), times 2 halves of every 32-wide k-tile (wave >> 2): the K
//   split stays INSIDE the workgroup (the two halves are added through LDS once per tile piece, in a fixed order), so a
//   short problem (47 .. 125 n-tiles) still fills 256 CUs with at most 8 stream-K pieces per tile, every SIMD hosts two
//   waves of one workgroup (one multiplies while the other waits on memory), and a wave issues 4 loads per k-step
//   wave tile = (16 TM) x (16 TN); accumulators TM x TN x 4 registers
//   lane (r = lane & 15, q = lane >> 4) owns matrix row r of every 16-row tile and the k's {4q..4q+3} + 16 (wave >> 2) of
//   the k-tile: ONE ds_read_b128 feeds four MFMAs (A and B use the same k permutation, the sum is unchanged)
//   LDS rows are unpadded (32 floats = 8 chunks of 16 B); chunk c of row R sits at position c ^ ((R >> 1) & 7): the
//   b128 lane groups of that read pattern hit 64 distinct banks, and the ds_write_b128 of a row covers all 32 banks
//   (a padded row stride cannot be conflict free for this pattern: the groups mix k-chunks q and q + 1).
//   The global loads of the next k-tile are issued BETWEEN the MFMA groups (two loads, 14 MFMAs, ...): issued in one
//   burst in front of them, a wave spends ~2 300 cycles queueing on the CU's address path before its first MFMA
//   (measured: 6 000 cycles per k-step against 3 584 of MFMA with one wave per SIMD).
// Stream-K decomposition, slab outputs, cursor and epilogue staging are those of gemm_nt_f32_kernel.
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int r16_band_tiles(int TM, int BN) {          // 16-row tiles per epilogue pass: the staging rows must fit one k buffer
    int t = TM;
    while (t > 1 && 16 * t * BN > (16 * TM + BN) * GEMM_BK) --t;
    return t;
}

template <int TM, int TN>
__global__ __launch_bounds__(512)
void gemm_nt_f32_r16_kernel(const GemmArgs args) {
    constexpr int NT = 512;
    constexpr int BM = 16 * TM, BN = 64 * TN;
    constexpr int LA = (BM + 63) / 64, LB = BN / 64;     // float4 loads per thread per k-tile (64 rows per pass)
    constexpr int A_TAIL = BM - 64 * (LA - 1);           // rows covered by the last A pass (64 = full)
    constexpr int BUF = (BM + BN) * GEMM_BK;             // floats per k buffer
    __shared__ __attribute__((aligned(16))) float smem[2 * BUF];
    auto sA = [&](int buf) { return smem + buf * BUF; };
    auto sB = [&](int buf) { return smem + buf * BUF + BM * GEMM_BK; };

    const int G = args.G;
    const int g = gemm_wg_of_block(args);
    if (g >= G) return;
    const int it0 = gemm_range_begin(g, args.total_iters, G);
    const int it1 = gemm_range_begin(g + 1, args.total_iters, G);
    if (it0 >= it1) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 3, wk = wave >> 2;              // n position, k half
    const int r = lane & 15, q = lane >> 4;
    const int lrow = tid >> 3;                            // 0..63
    const int chunk = tid & 7;
    const int lc4 = chunk * 4;                            // float offset inside the k-tile (global side)
    const int wpos = 4 * (chunk ^ ((lrow >> 1) & 7));     // swizzled chunk position (LDS side); rows lrow + 64 i share it
    const bool a_last_ok = lrow < A_TAIL;

    // ------------------------------------------------------------------ load cursor (runs one iteration ahead)
    float4 ra[LA], rb[LB];
    const float* pa[LA];
    const float* pb[LB];
    int l_prob = 0, l_tile = 0, l_tile_left = 0;
    int l_seg = 0, l_seg_left = 0, l_k = 0, l_K = 0;
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
            int m = m0 + lrow + 64 * i;
            m = m < P.M ? m : P.M - 1;
            const long long row = S.a_idx ? (long long)S.a_idx[m] : (long long)m;
            pa[i] = S.A + row * S.lda + lc4;
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            int n = n0 + lrow + 64 * i;
            n = n < P.N ? n : P.N - 1;
            pb[i] = S.W + (long long)n * S.ldw + lc4;
        }
    };
    auto open_tile = [&](int prob, int tile, int kt) __attribute__((always_inline)) {
        l_prob = prob;
        l_tile = tile;
        const GemmProb& P = args.p[prob];
        l_tile_left = P.ktiles - kt;
        int s = 0;
        while (s < P.nseg - 1 && kt >= (P.seg[s].K + GEMM_BK - 1) / GEMM_BK) { kt -= (P.seg[s].K + GEMM_BK - 1) / GEMM_BK; ++s; }
        open_segment(s, kt);
    };
    int l_ko = 0;                                          // k offset of the tile being loaded (0 in the K tail: a valid address)
    auto advance = [&]() __attribute__((always_inline)) {  // move the cursor to the next k-tile (wave-uniform bookkeeping)
        if (l_tile_left == 0) {
            if (l_tile + 1 < args.p[l_prob].tiles_m * args.p[l_prob].tiles_n) open_tile(l_prob, l_tile + 1, 0);
            else open_tile(l_prob + 1, 0, 0);
        } else if (l_seg_left == 0) {
            open_segment(l_seg + 1, 0);
        }
        l_ko = (l_k + lc4 < l_K) ? l_k : 0;
        l_k += GEMM_BK;
        --l_seg_left;
        --l_tile_left;
    };
    auto load_a = [&](int i) __attribute__((always_inline)) { ra[i] = *reinterpret_cast<const float4*>(pa[i] + l_ko); };
    auto load_b = [&](int i) __attribute__((always_inline)) { rb[i] = *reinterpret_cast<const float4*>(pb[i] + l_ko); };
    auto store_tile = [&](int buf) __attribute__((always_inline)) {
        if (!(l_k - GEMM_BK + lc4 < l_K)) {                // K tail of the tile in flight (l_k has advanced by BK): zeros
#pragma unroll
            for (int i = 0; i < LA; ++i) ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < LB; ++i) rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < LA; ++i)
            if (i + 1 < LA || a_last_ok)
                *reinterpret_cast<float4*>(sA(buf) + (lrow + 64 * i) * GEMM_BK + wpos) = ra[i];
#pragma unroll
        for (int i = 0; i < LB; ++i)
            *reinterpret_cast<float4*>(sB(buf) + (lrow + 64 * i) * GEMM_BK + wpos) = rb[i];
    };

    // ------------------------------------------------------------------ compute-side tile bookkeeping
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

    // Epilogue: accumulator element e of tile (ti, tj) is C[16 ti + 4 q + e][16 TN wn + 16 tj + r].  TB 16-row tiles per
    // pass go through the idle k buffer: the k-half-1 waves put their sums there, the k-half-0 waves add their own on top
    // (half 0 + half 1, always in that order), then the rows leave as 16-byte stores (512-byte runs per row).
    constexpr int TB = r16_band_tiles(TM, BN);
    auto flush = [&](const f32x4 (&acc)[TM][TN], float* stage) __attribute__((always_inline)) {
        const GemmProb& P = args.p[c_prob];
        const int m0 = (c_tile % P.tiles_m) * BM, n0 = (c_tile / P.tiles_m) * BN;
        float* C = P.C + (long long)c_piece * P.slab_stride;
        const int extra = c_last ? P.nslab - 1 - c_piece : 0;
        const bool vec_ok = ((P.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(P.C) & 15) == 0) && ((P.slab_stride & 3) == 0);
        constexpr int TPR = BN / 4;                        // threads per staged row
        constexpr int RPP = NT / TPR;                      // rows per store pass
        const int c4 = (tid % TPR) * 4;
        const int n = n0 + c4;
#pragma unroll
        for (int b0 = 0; b0 < TM; b0 += TB) {
            if (wk == 1) {
#pragma unroll
                for (int ti = b0; ti < TM && ti < b0 + TB; ++ti)
#pragma unroll
                    for (int tj = 0; tj < TN; ++tj)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            stage[((ti - b0) * 16 + 4 * q + e) * BN + wn * (16 * TN) + tj * 16 + r] = acc[ti][tj][e];
            }
            __syncthreads();
            if (wk == 0) {
#pragma unroll
                for (int ti = b0; ti < TM && ti < b0 + TB; ++ti)
#pragma unroll
                    for (int tj = 0; tj < TN; ++tj)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float* sp = stage + ((ti - b0) * 16 + 4 * q + e) * BN + wn * (16 * TN) + tj * 16 + r;
                            *sp = acc[ti][tj][e] + *sp;
                        }
            }
            __syncthreads();
            constexpr int BAND = 16 * TB;
#pragma unroll
            for (int i = 0; i < (BAND + RPP - 1) / RPP; ++i) {
                const int sr = tid / TPR + RPP * i;
                const int m = m0 + 16 * b0 + sr;
                if (sr < BAND && 16 * b0 + sr < BM && m < P.M && n < P.N) {
                    const float4 v = *reinterpret_cast<const float4*>(stage + sr * BN + c4);
                    float* dst = C + (long long)m * P.ldc + n;
                    if (vec_ok && n + 3 < P.N) {
                        *reinterpret_cast<float4*>(dst) = v;
                        for (int x = 1; x <= extra; ++x)
                            *reinterpret_cast<float4*>(dst + (long long)x * P.slab_stride) = make_float4(0.f, 0.f, 0.f, 0.f);
                    } else {
                        const float vv[4] = {v.x, v.y, v.z, v.w};
                        for (int qq = 0; qq < 4; ++qq)
                            if (n + qq < P.N) {
                                dst[qq] = vv[qq];
                                for (int x = 1; x <= extra; ++x) dst[(long long)x * P.slab_stride + qq] = 0.f;
                            }
                    }
                }
            }
            __syncthreads();
        }
    };

    {
        const int kt = decode(it0);
        open_tile(c_prob, c_tile, kt);
    }
    advance();
#pragma unroll
    for (int i = 0; i < LA; ++i) load_a(i);
#pragma unroll
    for (int i = 0; i < LB; ++i) load_b(i);
    store_tile(0);
    __syncthreads();
    int cur = 0;
    const int ch = 4 * ((q + 4 * wk) ^ (r >> 1));          // this wave's k chunk, swizzled: ((16 t + r) >> 1) & 7 = r >> 1 for every tile t
    for (int it = it0; it < it1;) {
        f32x4 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
        const int n_it = c_left;
        for (int j_it = 0; j_it < n_it; ++j_it, ++it) {
            const bool more = it + 1 < it1;
            if (more) advance();
            const float* a_base = sA(cur) + r * GEMM_BK + ch;
            const float* b_base = sB(cur) + (wn * (16 * TN) + r) * GEMM_BK + ch;
            float4 av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = *reinterpret_cast<const float4*>(a_base + i * 16 * GEMM_BK);
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = *reinterpret_cast<const float4*>(b_base + j * 16 * GEMM_BK);
            // Next tile's global loads, STAGGERED between the two waves of a SIMD (waves w and w + 4 = the two k halves): the
            // k-half-0 wave issues its loads in front of its MFMAs, the k-half-1 wave behind its third MFMA group, so one of
            // them multiplies while the other queues on the CU's address path (lock-stepped, both would queue, then both multiply).
            auto loads = [&]() __attribute__((always_inline)) {
                if (more) {
#pragma unroll
                    for (int i = 0; i < LA; ++i) load_a(i);
#pragma unroll
                    for (int i = 0; i < LB; ++i) load_b(i);
                }
            };
            loads();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].x, bv[j].x, acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].y, bv[j].y, acc[i][j], 0, 0, 0);
            // refill of the OTHER k buffer (free since the barrier that ended the previous iteration) in the middle of the MFMA
            // stream: its ds_writes and the wait for the loads overlap the partner wave's MFMAs; only the barrier is left at the end
            __builtin_amdgcn_sched_barrier(0);
            if (more) store_tile(cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].z, bv[j].z, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].w, bv[j].w, acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
                __syncthreads();
                cur ^= 1;
            }
        }
        flush(acc, sA(cur ^ 1));
        if (it < it1) decode(it);
    }
}

// Host side: fill tiles / iteration offsets, pick the number of workgroups and the slab count.
//   slots      resident workgroups the launch should fill (CUs x 4 for this 36.9 KB-LDS kernel)
//   min_iters  smallest range worth a workgroup (prologue + flush amortisation)
// Returns nslab; the caller then sets every problem's C / slab_stride (slabs are nslab deep).
inline int gemm_plan(GemmArgs& a, int slots, int min_iters = 8, int BM = 64, int BN = 64, int BK = GEMM_BK) {
    int total = 0, kt_max = 1;
    for (int i = 0; i < a.nprob; ++i) {
        GemmProb& p = a.p[i];
        p.tiles_m = (p.M + BM - 1) / BM;
        p.tiles_n = (p.N + BN - 1) / BN;
        p.ktiles = 0;
        for (int s = 0; s < p.nseg; ++s) p.ktiles += (p.seg[s].K + BK - 1) / BK;
        p.it_begin = total;
        total += p.tiles_m * p.tiles_n * p.ktiles;
        if (p.ktiles > kt_max) kt_max = p.ktiles;
    }
    a.total_iters = total;
    int G = total / min_iters;
    if (G > slots) G = slots;
    const int L_min = (kt_max + 6) / 7;            // keep pieces per tile <= 8
    if (G > total / L_min) G = total / L_min;
    if (G < 1) G = 1;
    a.G = G;
    const int L = total / G;                       // shortest range
    a.nslab = G == 1 ? 1 : (kt_max + L - 1) / L + 1;
    if (a.nslab > 8) a.nslab = 8;
    for (int i = 0; i < a.nprob; ++i) a.p[i].nslab = a.nslab;      // (gemm_tight_slabs: fewer for a short-K problem of a merged launch)
    a.aligned = 0;
    a.xcd_chunk = 0;
    return a.nslab;
}

// K-ALIGNED plan (the 128 x 256 kernels of gemm_bf16.h / gemm_f32x3.h): every tile of problem p is cut into split_p equal k
// pieces and every workgroup takes exactly ONE piece of ONE tile; workgroups are numbered piece-major, tiles m-fastest, so the
// 32 workgroups of an XCD (contiguous numbers) are neighbouring tiles AT THE SAME k: they walk the same weight / activation
// k-windows at the same time and share them in L2, where the stream-K ranges (k offsets shifted from tile to tile) send almost
// every tile load to the Infinity Cache.  The slab count is exact (max split) and a tile writes exactly
// split_p slabs.  Chooses the smallest critical path T = max_p ceil(ktiles_p / split_p) with sum_p tiles_p split_p <= slots,
// pieces of at least min_iters k-tiles, split <= 8.  Returns 0 (and leaves the plan untouched) when the tiles alone exceed the
// slots: the caller then uses gemm_plan.
inline int gemm_plan_aligned(GemmArgs& a, int slots, int min_iters, int BM, int BN, int BK) {
    int tiles[4], kt[4], kt_max = 1, sum_tiles = 0;
    for (int i = 0; i < a.nprob; ++i) {
        const GemmProb& p = a.p[i];
        tiles[i] = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
        kt[i] = 0;
        for (int s = 0; s < p.nseg; ++s) kt[i] += (p.seg[s].K + BK - 1) / BK;
        if (kt[i] > kt_max) kt_max = kt[i];
        sum_tiles += tiles[i];
    }
    if (sum_tiles > slots || sum_tiles == 0) return 0;
    int best_T = kt_max;
    for (int T = kt_max; T >= 1; --T) {
        int wgs = 0;
        bool ok = true;
        for (int i = 0; i < a.nprob && ok; ++i) {
            const int s = (kt[i] + T - 1) / T;
            ok = s <= 8 && (s == 1 || kt[i] / s >= min_iters);
            wgs += tiles[i] * s;
        }
        if (!ok || wgs > slots) break;
        best_T = T;
    }
    int total = 0, g = 0, nslab = 1;
    for (int i = 0; i < a.nprob; ++i) {
        GemmProb& p = a.p[i];
        p.tiles_m = (p.M + BM - 1) / BM;
        p.tiles_n = (p.N + BN - 1) / BN;
        p.ktiles = kt[i];
        p.it_begin = total;
        total += tiles[i] * kt[i];
        p.split = (kt[i] + best_T - 1) / best_T;
        p.g_begin = g;
        g += tiles[i] * p.split;
        if (p.split > nslab) nslab = p.split;
    }
    a.total_iters = total;
    a.G = g;
    a.nslab = nslab;
    for (int i = 0; i < a.nprob; ++i) a.p[i].nslab = nslab;       // every problem writes / zero-fills the launch's slab count (gemm_tight_slabs: its own split)
    a.aligned = 1;
    // XCD dealing in whole m-groups (round 6): workgroups are numbered m-fastest, so tiles_m consecutive numbers are the m-tiles of ONE
    // weight n-tile at ONE k window.  With ceil(G / 8) numbers per XCD a group straddles two XCDs whenever that is not a multiple of
    // tiles_m (G = 200, tiles_m = 4: 25 per XCD, every fourth group split) and its weight window is fetched by two L2s.  Deal whole
    // groups instead - when every problem has the same tiles_m and the larger chunk still fits the XCD's share of the slots.
    a.xcd_chunk = 0;
    int u = a.p[0].tiles_m;
    for (int i = 1; i < a.nprob; ++i) if (a.p[i].tiles_m != u) u = 1;
    if (u > 1 && g % u == 0) {
        const int groups = g / u, lo = groups / 8, extra = groups % 8, ch = u * (lo + (extra ? 1 : 0));
        if (ch <= (slots + 7) / 8) { a.xcd_chunk = ch; a.xcd_unit = u; a.xcd_lo = lo; a.xcd_extra = extra; }
    }
    return nslab;
}

// Slabs problem i of a planned launch really needs: a tile of k k-tiles meets at most ceil(k / L) + 1 stream-K ranges of at least L
// iterations (k-aligned plan: exactly its split).  A caller whose consumer reads problem i on its own may lower GemmProb::nslab to
// this (the kernel then writes / zero-fills that many slabs for the problem, the consumer adds that many): the vocabulary GEMM that
// shares a launch with the LSTM1 sums of the next step, att_ga next to LSTM2.
inline int gemm_tight_slabs(const GemmArgs& a, int i) {
    if (a.aligned) return a.p[i].split;
    if (a.G <= 1) return 1;
    const int L = a.total_iters / a.G;
    const int ns = (a.p[i].ktiles + L - 1) / L + 1;
    return ns < a.nslab ? ns : a.nslab;
}

inline double gemm_flops(const GemmArgs& a) {
    double f = 0;
    for (int i = 0; i < a.nprob; ++i)
        for (int s = 0; s < a.p[i].nseg; ++s) f += 2.0 * a.p[i].M * a.p[i].N * a.p[i].seg[s].K;
    return f;
}

// ALGORITHMIC bytes of a launch: every operand element and every output element once (w_bytes = bytes per weight element:
// 4, or 2 when the W operands are bf16 copies); gathered A rows count once per output row.
inline double gemm_bytes(const GemmArgs& a, int w_bytes) {
    double b = 0;
    for (int i = 0; i < a.nprob; ++i) {
        b += 4.0 * a.p[i].M * a.p[i].N;
        for (int s = 0; s < a.p[i].nseg; ++s) b += 4.0 * a.p[i].M * a.p[i].seg[s].K + (double)w_bytes * a.p[i].N * a.p[i].seg[s].K;
    }
    return b;
}

}  // namespace vsr
