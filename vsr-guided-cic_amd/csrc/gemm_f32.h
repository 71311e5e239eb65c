// Grouped "NT" fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain).
//
//   C_p[s][m][n] = sum over the k-tiles of split s, over segments g:  A_pg[row_g(m)][k] * W_pg[n][k]
//
// A launch carries up to 4 independent problems (e.g. h1 -> [W1_hg | att_ha] and s_t -> [s_fc | att_sa]);
// each problem sums up to 3 K-segments so that the reference's torch.cat([h2, vbar, x]) (step :147, :176)
// is never materialised: every segment reads its own activation matrix and a column window of the
// reference-layout [out, in] weight, in place.  Activation rows may be gathered through an int32 index
// (embedding rows, beam parents).  Split-K slabs are summed by the consumer kernels.
//
// Tile: BM x BN x 32, 256 threads = 4 waves (2 x 2), each wave (BM/64) x (BN/64) MFMA tiles of 32x32.
// LDS rows are padded to 36 floats: ds_write_b128 / ds_read_b128 are bank-conflict free (guide LDS table).
// Lane (r = lane & 31, h = lane >> 5) reads A[r][8*kk + 4h .. +3] with one ds_read_b128 and feeds the four
// floats to four MFMAs; A and B use the same k permutation, so the sum is unchanged.
// Work units are dealt so that all m-tiles of one weight n-tile run on ONE XCD back to back: the weight
// tile is fetched from HBM once and re-read from that XCD's L2.
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
    int pad_;
};

struct GemmProb {
    GemmSeg seg[3];
    float* C;            // (nsplit, M, ldc)
    long long split_stride;
    int nseg;
    int M;
    int N;
    int ldc;
    int nsplit;
    int tiles_m;
    int tiles_n;
    int unit_begin;      // first work unit of this problem in the launch
    int ktiles;          // total 32-wide k-tiles over all segments
    int pad_;
};

struct GemmArgs {
    GemmProb p[4];
    int nprob;
    int total_units;
    int chunk;           // ceil(total_units / 8): units per XCD
    int pad_;
};

constexpr int GEMM_BK = 32;
constexpr int GEMM_LDS = GEMM_BK + 4;

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(const GemmArgs args) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int LA = BM / 32, LB = BN / 32;            // float4 loads per thread per k-tile
    __shared__ float smem[2 * (BM + BN) * GEMM_LDS];
    auto sA = [&](int buf) { return smem + buf * (BM + BN) * GEMM_LDS; };
    auto sB = [&](int buf) { return smem + buf * (BM + BN) * GEMM_LDS + BM * GEMM_LDS; };

    // ---- work unit (XCD-contiguous dealing: blocks b and b+8 share an XCD)
    const int bid = blockIdx.x;
    const int unit = (bid & 7) * args.chunk + (bid >> 3);
    if (unit >= args.total_units) return;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < args.nprob && unit >= args.p[i].unit_begin) pi = i;
    const GemmProb& P = args.p[pi];
    int lu = unit - P.unit_begin;
    const int mt = lu % P.tiles_m;
    lu /= P.tiles_m;
    const int nt = lu % P.tiles_n;
    const int split = lu / P.tiles_n;
    const int m0 = mt * BM, n0 = nt * BN;
    const int kt_begin = (int)(((long long)P.ktiles * split) / P.nsplit);
    const int kt_end = (int)(((long long)P.ktiles * (split + 1)) / P.nsplit);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, hh = lane >> 5;

    // ---- per-thread load coordinates (8 threads cover one 128-byte row segment)
    const int lrow = tid >> 3;          // 0..31
    const int lc4 = (tid & 7) * 4;      // float offset inside the k-tile
    float4 ra[LA], rb[LB];

    // running segment cursor
    int seg = 0, seg_kt0 = 0;           // first k-tile index of the current segment
    auto seg_tiles = [&](int s) { return (P.seg[s].K + GEMM_BK - 1) / GEMM_BK; };
    {
        int kt = kt_begin;
        while (seg < P.nseg - 1 && kt >= seg_kt0 + seg_tiles(seg)) { seg_kt0 += seg_tiles(seg); ++seg; }
    }

    auto load_tile = [&](int kt) {
        while (seg < P.nseg - 1 && kt >= seg_kt0 + seg_tiles(seg)) { seg_kt0 += seg_tiles(seg); ++seg; }
        const GemmSeg& S = P.seg[seg];
        const int k = (kt - seg_kt0) * GEMM_BK + lc4;
        const bool kin = k < S.K;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            int m = m0 + lrow + 32 * i;
            m = m < P.M ? m : P.M - 1;
            const long long row = S.a_idx ? (long long)S.a_idx[m] : (long long)m;
            ra[i] = kin ? *reinterpret_cast<const float4*>(S.A + row * S.lda + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            int n = n0 + lrow + 32 * i;
            n = n < P.N ? n : P.N - 1;
            rb[i] = kin ? *reinterpret_cast<const float4*>(S.W + (long long)n * S.ldw + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LA; ++i)
            *reinterpret_cast<float4*>(sA(buf) + (lrow + 32 * i) * GEMM_LDS + lc4) = ra[i];
#pragma unroll
        for (int i = 0; i < LB; ++i)
            *reinterpret_cast<float4*>(sB(buf) + (lrow + 32 * i) * GEMM_LDS + lc4) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (kt_begin < kt_end) {
        load_tile(kt_begin);
        store_tile(0);
        __syncthreads();
        int cur = 0;
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const bool more = kt + 1 < kt_end;
            if (more) load_tile(kt + 1);                 // global -> registers, in flight during the MFMAs
            const float* a_base = sA(cur) + (wm * (BM / 2) + r) * GEMM_LDS + 4 * hh;
            const float* b_base = sB(cur) + (wn * (BN / 2) + r) * GEMM_LDS + 4 * hh;
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
                store_tile(cur ^ 1);                     // other buffer: nobody reads it in this iteration
                __syncthreads();
                cur ^= 1;
            }
        }
    }

    // ---- epilogue: raw partial sums; C/D layout col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
    float* C = P.C + (long long)split * P.split_stride;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / 2) + j * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                if (m < P.M && n < P.N) C[(long long)m * P.ldc + n] = acc[i][j][e];
            }
        }
}

}  // namespace vsr
