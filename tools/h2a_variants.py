#!/usr/bin/env python3
"""Bit-identical scheduling variants of the all-DMA wide f16x2 kernel (csrc/gemm_h2a.h), built like tools/h2a_ablate.py: a scratch COPY of
csrc/ gets compile-time switches, tools/gemm_bench is built from it once per variant as tools/gemm_bench_var<n>.  The k order of every sum is
untouched, so results are the shipped kernel's (gemm_bench checks them against fp64).
  0  shipped kernel
  1  the movers SPREAD their six DMA issues over the k-tile (s_sleep 2 = 128 cycles between requests) instead of bunching them behind the barrier
  2  ... s_sleep 4 between requests
  3  A requests first, then W (shipped: W first)
  4  movers at lower priority than the multipliers (s_setprio 0 / 3)
  5  the flush of round 4: band by band through one W stage also on k-aligned plans (shipped since round 5: the whole tile staged at once)
  6  ABLATION (wrong sums, timings only; run with GEMM_NOCHECK=1): the multipliers read 12 instead of 16 operand fragments per k-tile (the lo planes of
     the second k-half are not re-read): what would 25 % less LDS fragment traffic - a 64 x 128 wave tile - buy?
  7  ABLATION: 8 instead of 16 fragment reads (no reads at all in the second k-half)
usage: tools/h2a_variants.py ; then tools/gemm_bench_var<n> 500 256 4 5400 1"""
import os, shutil, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = "/tmp/h2avar"
shutil.rmtree(d, ignore_errors=True)
os.makedirs(d + "/vsr-guided-cic_amd")
shutil.copytree(ROOT + "/vsr-guided-cic_amd/csrc", d + "/vsr-guided-cic_amd/csrc")
shutil.copytree(ROOT + "/tools", d + "/tools", ignore=lambda p, names: [n for n in names if not (n.endswith((".hip", ".h")) or n == "experiments")])
p = d + "/vsr-guided-cic_amd/csrc/gemm_h2a.h"
s = open(p).read()
def rep(old, new, cnt=1):
    global s
    assert s.count(old) == cnt, (s.count(old), old[:80])
    s = s.replace(old, new)
rep("namespace vsr {\n\n", "namespace vsr {\n\n#ifndef H2A_VAR\n#define H2A_VAR 0\n#endif\n\n")
rep("    const int tid = threadIdx.x;\n", "    constexpr int var = H2A_VAR;\n    const int tid = threadIdx.x;\n")
rep("""#pragma unroll
            for (int i = 0; i < LB; ++i) h2_glds16(in ? (const void*)(pbW[i] + cw.k) : (const void*)g_h2_zero16, base + (unsigned)(8 * i) * 1024u);
#pragma unroll
            for (int i = 0; i < LA; ++i) h2_glds16(in ? (const void*)(pa[i] + cw.k) : (const void*)g_h2_zero16, base + (unsigned)(WSTAGE * 2) + (unsigned)(8 * i) * 1024u);
""", """            if constexpr (var == 3) {
#pragma unroll
                for (int i = 0; i < LA; ++i) h2_glds16(in ? (const void*)(pa[i] + cw.k) : (const void*)g_h2_zero16, base + (unsigned)(WSTAGE * 2) + (unsigned)(8 * i) * 1024u);
            }
#pragma unroll
            for (int i = 0; i < LB; ++i) {
                h2_glds16(in ? (const void*)(pbW[i] + cw.k) : (const void*)g_h2_zero16, base + (unsigned)(8 * i) * 1024u);
                if constexpr (var == 1) { if (spread) __builtin_amdgcn_s_sleep(2); }
                if constexpr (var == 2) { if (spread) __builtin_amdgcn_s_sleep(4); }
            }
            if constexpr (var != 3) {
#pragma unroll
                for (int i = 0; i < LA; ++i) {
                    h2_glds16(in ? (const void*)(pa[i] + cw.k) : (const void*)g_h2_zero16, base + (unsigned)(WSTAGE * 2) + (unsigned)(8 * i) * 1024u);
                    if constexpr (var == 1) { if (spread && i + 1 < LA) __builtin_amdgcn_s_sleep(2); }
                    if constexpr (var == 2) { if (spread && i + 1 < LA) __builtin_amdgcn_s_sleep(4); }
                }
            }
""")
rep("        int seg_exp = 0;\n", "        int seg_exp = 0;\n        bool spread = false;      // not in the prologue\n")
rep("        __syncthreads();                                   // k-tile 0 is ready\n        // k-tile j: issue k-tile j + NW - 1",
    "        __syncthreads();                                   // k-tile 0 is ready\n        spread = true;\n        if constexpr (var == 4) __builtin_amdgcn_s_setprio(0);\n        // k-tile j: issue k-tile j + NW - 1")
rep("        zero_acc();\n        bool first = true;\n", "        zero_acc();\n        bool first = true;\n        if constexpr (var == 4) __builtin_amdgcn_s_setprio(3);\n")
rep("        if (args.aligned) {\n            // k-aligned plan: this piece is the workgroup's only one", "        if (args.aligned && var != 5) {\n            // k-aligned plan: this piece is the workgroup's only one")
rep("""                f16x8_t ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[i] = *reinterpret_cast<const f16x8_t*>(a_row + i * 32 * 64 + wh);
                    al[i] = *reinterpret_cast<const f16x8_t*>(a_row + i * 32 * 64 + wl);
                }
#pragma unroll
                for (int jj = 0; jj < TN; ++jj) {
                    bh[jj] = *reinterpret_cast<const f16x8_t*>(b_row + jj * 32 * 64 + wh);
                    bl[jj] = *reinterpret_cast<const f16x8_t*>(b_row + jj * 32 * 64 + wl);
                }
""", """#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if (!(var == 7 && kk == 1)) ah[i] = *reinterpret_cast<const f16x8_t*>(a_row + i * 32 * 64 + wh);
                    if (!((var == 6 || var == 7) && kk == 1)) al[i] = *reinterpret_cast<const f16x8_t*>(a_row + i * 32 * 64 + wl);
                }
#pragma unroll
                for (int jj = 0; jj < TN; ++jj) {
                    if (!(var == 7 && kk == 1)) bh[jj] = *reinterpret_cast<const f16x8_t*>(b_row + jj * 32 * 64 + wh);
                    if (!((var == 6 || var == 7) && kk == 1)) bl[jj] = *reinterpret_cast<const f16x8_t*>(b_row + jj * 32 * 64 + wl);
                }
""")
rep("""#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                const int wh = 8 * ((2 * kk + hh) ^ wsw)""", """            f16x8_t ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                const int wh = 8 * ((2 * kk + hh) ^ wsw)""")
open(p, "w").write(s)
b = d + "/tools/gemm_bench.hip"
procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-DH2A_VAR=%d" % n, "-o", ROOT + "/tools/gemm_bench_var%d" % n, b]) for n in (0, 6, 7)]
assert all(p.wait() == 0 for p in procs)
print("built tools/gemm_bench_var0, 6, 7 (edit the tuple for the others)")
