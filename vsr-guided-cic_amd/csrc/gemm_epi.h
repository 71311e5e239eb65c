// In-launch combine of the k-pieces of a GEMM tile, with the pointwise consumer of that GEMM as the last arriver's epilogue
// (GemmEpi in gemm_f32.h; used by gemm_h2a.h and by the streaming kernel of gemm_h2.h).
//
// Why: every decoder GEMM is cut along K (stream-K ranges or k-aligned pieces) so that 256 CUs have work at M <= 500, and until round 4
// every consumer kernel (k_lstm1, the slab phase of k_attend, k_lstm2, k_vocab) re-read the 2-4 partial slabs of its input from memory:
// 116 MB of slab reads per beam-5 timestep where 48 MB of sums would do, and two launches that exist only to add them.
//
// Protocol (cdna_hip_programming.md section 6 Guideline 16, counter form; MI355X_MICROARCH.md "Valid forms", first table row):
//   * every piece stores its slab WRITE-THROUGH (16-byte sc1 stores: st16_wt), every storing wave drains its stores (s_waitcnt
//     vmcnt(0)), the workgroup meets at a barrier, ONE lane adds 1 to the tile's ticket (relaxed, agent scope);
//   * the workgroup whose add returned (pieces - 1) is the last arriver: that lane resets the ticket for the next launch, issues ONE
//     agent-scope acquire (this CU's L1 is dropped), the workgroup meets at a barrier, and every wave reads EVERY slab of the tile -
//     its own too - with sc1 loads (L1 bypassed: belt and braces on top of the acquire; a single-piece tile needs no ticket and no
//     acquire, its one slab is read back the same way);
//   * the slabs are added in index order 0 .. pieces - 1: the order slab_sum() / k_attend / k_vocab add them in, so every sum has the
//     bits the slab path gives it, whichever workgroup arrives last;
//   * results do not depend on dispatch order, timing or placement; nobody ever WAITS for another workgroup (no spin, no deadlock).
// The tickets are all zero between launches: vsr_prepare*() zeroes them, each last arriver puts its own back to zero.
#pragma once
#include "gemm_bf16.h"      // async loads, wait_loads, landed
#include "pointwise.h"

namespace vsr {

typedef float epi_f4 __attribute__((ext_vector_type(4)));

// 16-byte write-through store (sc1): leaves the XCD's L2 for memory, visible to every XCD once the storing wave has drained vmcnt
__device__ __forceinline__ void st16_wt(float* p, const float4& v) {
    const epi_f4 x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(x) : "memory");
}
// 16-byte load that bypasses this CU's L1 (sc1); asynchronous: wait_loads<0>() + landed() before the first use
__device__ __forceinline__ void ld16_sc1(epi_f4& d, const float* p) { asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void epi_landed(epi_f4& d) { asm volatile("" : "+v"(d)); }

// Every wave of the workgroup has stored its part of the piece's slab and drained its stores.  Returns true, in every thread, when this
// workgroup's piece is the last of the tile's `pieces` to arrive.  flag: one int of the kernel's (single) LDS array.
__device__ __forceinline__ bool gemm_epi_arrive(int* ticket, int pieces, int* flag) {
    __syncthreads();                                       // ... of every wave of this workgroup
    if (threadIdx.x == 0) {
        int last = 1;
        if (pieces > 1) {
            const int old = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = old == pieces - 1;
            if (last) {
                __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // all zero again for the next launch
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        *flag = last;
    }
    __syncthreads();
    return *flag != 0;
}

// The last arriver's work on tile (m0, n0) of problem P: add the `pieces` slabs in index order, apply P.epi.  NT threads; BM x BN tile;
// N, ldc, the slab stride and every leading dimension are multiples of 4 and the bases 16-byte aligned (the host enables an epilogue
// only then).  Ends with an empty vmcnt queue.
template <int BM, int BN, int NT>
__device__ __forceinline__ void gemm_epi_tile(const GemmProb& P, int m0, int n0, int pieces) {
    constexpr int TPR = BN / 4, RPP = NT / TPR, PASSES = (BM + RPP - 1) / RPP;
    constexpr int CH = PASSES < 4 ? PASSES : 4;           // row passes in flight per thread
    constexpr int JU = 4;                                  // slabs in flight per pass
    const int tid = threadIdx.x;
    const int n = n0 + (tid % TPR) * 4, sr = tid / TPR;
    const GemmEpi& E = P.epi;
    const bool col_ok = n < P.N;
    const int kind = E.kind;
    // per-thread constants of the LSTM2 epilogue: logical column n = 4 u + gate
    const int H = P.wperm_stride, u = n >> 2;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    if (kind == EPI_LSTM2 && col_ok) {
#pragma unroll
        for (int g = 0; g < 4; ++g) bsum[g] = E.a0[g * H + u] + E.a1[g * H + u];
    }
    epi_f4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (kind == EPI_SUM && E.a0 && col_ok) bias4 = *reinterpret_cast<const epi_f4*>(E.a0 + n);
    const bool bias_first = kind == EPI_SUM && E.a0 && E.rpi == 0, bias_last = kind == EPI_SUM && E.a0 && E.rpi != 0;
#pragma unroll 1
    for (int ch = 0; ch < PASSES; ch += CH) {
        if (m0 + ch * RPP >= P.M) break;
        epi_f4 acc[CH];
        bool ok[CH];
        int mrow[CH];
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            mrow[q] = m0 + (ch + q) * RPP + sr;
            ok[q] = col_ok && (ch + q) < PASSES && mrow[q] < P.M && mrow[q] < m0 + BM;
            acc[q] = bias_first ? bias4 : epi_f4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll 1
        for (int j0 = 0; j0 < pieces; j0 += JU) {
            epi_f4 v[JU][CH];
#pragma unroll
            for (int j = 0; j < JU; ++j)
#pragma unroll
                for (int q = 0; q < CH; ++q)
                    if (ok[q] && j0 + j < pieces) ld16_sc1(v[j][q], P.C + (long long)(j0 + j) * P.slab_stride + (long long)mrow[q] * P.ldc + n);
            wait_loads<0>();
#pragma unroll
            for (int j = 0; j < JU; ++j)
#pragma unroll
                for (int q = 0; q < CH; ++q) {
                    epi_landed(v[j][q]);
                    if (ok[q] && j0 + j < pieces) {
                        if (j0 + j == 0 && !bias_first) acc[q] = v[j][q];      // (the slab path starts from slab 0 as well)
                        else acc[q] += v[j][q];
                    }
                }
        }
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            // (the calling kernels run at their register limit: keep the passes' operand loads and transcendental temporaries apart)
            asm volatile("" ::: "memory");
            if (!ok[q]) continue;
            const int m = mrow[q];
            if (kind == EPI_SUM) {
                epi_f4 o = acc[q];
                if (bias_last) o += bias4;
                *reinterpret_cast<epi_f4*>(E.o0 + (long long)m * E.ldo + n) = o;
            } else if (kind == EPI_GT) {
                const long long at = (long long)m * E.ldo + n;
                const epi_f4 gp = *reinterpret_cast<const epi_f4*>(E.a0 + at), cn = *reinterpret_cast<const epi_f4*>(E.a1 + at);
                float4 o;
                o.x = gt_cell(gp.x, acc[q].x, cn.x); o.y = gt_cell(gp.y, acc[q].y, cn.y);
                o.z = gt_cell(gp.z, acc[q].z, cn.z); o.w = gt_cell(gp.w, acc[q].w, cn.w);
                *reinterpret_cast<float4*>(E.o0 + at) = o;
                if (E.o16) img_store4(E.o16, at, o, E.isc);
            } else if (kind == EPI_LSTM2) {
                float qv[4] = {acc[q].x, acc[q].y, acc[q].z, acc[q].w};
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    qv[g] += bsum[g];
                    if (E.a2) qv[g] += E.a2[(long long)(m / E.rpi) * 4 * H + g * H + u];
                }
                const int prow = E.idx ? E.idx[m] : m;
                float c, h2v;
                lstm_cell(qv[0], qv[1], qv[2], qv[3], E.a3[(long long)prow * H + u], h2v, c);
                const long long at = (long long)m * H + u;
                E.o0[at] = h2v;
                E.o1[at] = c;
                if (E.o16) img_store(E.o16, at, h2v, E.isc);
            }
        }
    }
    wait_loads<0>();
}

}  // namespace vsr
