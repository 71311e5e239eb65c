#!/usr/bin/env python3
"""Ablations of the all-DMA wide f16x2 kernel (csrc/gemm_h2a.h) WITHOUT hooks in the shipped header: csrc/ and tools/ are copied to a
scratch directory, the copy of gemm_h2a.h gets compile-time `if (H2A_ABL ...)` switches (run-time ones change the code the compiler makes of the k loop:
ten times slower), and tools/gemm_bench is built from it once per ablation as tools/gemm_bench_abl<n>.  Results are WRONG by construction: timings only.
  1  the movers stop issuing the A DMAs after the prologue        (32 KB per k-tile instead of 48)
  2  ... the W DMAs                                                  (16 KB)
  3  ... both                                                        (no global traffic in the k loop)
  4  the multipliers skip their LDS reads and MFMAs                  (data movement and barriers only)
  5  the multipliers read LDS but issue no MFMA
  6  no epilogue: the tile pieces are not staged and not stored (what do the LDS-staged 16-byte stores of 128 KB per workgroup cost?)
usage: tools/h2a_ablate.py ; then tools/gemm_bench_abl<n> 500 256 4 5400 1"""
import os, shutil, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = "/tmp/h2aabl"
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
rep("namespace vsr {\n\n", "namespace vsr {\n\n#ifndef H2A_ABL\n#define H2A_ABL 0\n#endif\n__device__ unsigned long long g_h2a_life[2048];      // s_memtime ticks from a workgroup's first to its last instruction\n\n")
# workgroup lifetime in shader-clock ticks (thread 0): lifetime ticks / kernel time = the clock the chip holds under this variant's load
rep("    const int tid = threadIdx.x;\n", "    const unsigned long long life0 = __builtin_amdgcn_s_memtime();\n    const int tid = threadIdx.x;\n")
rep("            if (end_of_ktile(std::true_type{})) { zero_acc(); first = true; }\n        }\n    }\n}", "            if (end_of_ktile(std::true_type{})) { zero_acc(); first = true; }\n        }\n        if (tid == 0 && g < 2048) g_h2a_life[g] = __builtin_amdgcn_s_memtime() - life0;\n    }\n}")
rep("    const int tid = threadIdx.x;\n", "    constexpr int abl = H2A_ABL;\n    const int tid = threadIdx.x;\n")
rep("            for (int i = 0; i < LB; ++i) h2_glds16(in ?", "            for (int i = 0; i < LB; ++i) if (!(quiet && (abl == 2 || abl == 3))) h2_glds16(in ?")
rep("            for (int i = 0; i < LA; ++i) h2_glds16(in ?", "            for (int i = 0; i < LA; ++i) if (!(quiet && (abl == 1 || abl == 3))) h2_glds16(in ?")
rep("        int seg_exp = 0;\n", "        int seg_exp = 0;\n        bool quiet = false;\n")
rep("        __syncthreads();                                   // k-tile 0 is ready\n        // k-tile j: issue k-tile j + NW - 1",
    "        __syncthreads();                                   // k-tile 0 is ready\n        quiet = true;\n        // k-tile j: issue k-tile j + NW - 1")
# the counted wait assumes LA + LB requests were issued: with ablations wait for everything
rep("            if (more) { issue(st); wait_loads<(NW - 2) * (LA + LB)>(); } else wait_loads<0>();",
    "            if (more) { issue(st); if (abl >= 1 && abl <= 3) wait_loads<0>(); else wait_loads<(NW - 2) * (LA + LB)>(); } else wait_loads<0>();")
rep("#pragma unroll\n            for (int kk = 0; kk < BK / 16; ++kk) {\n                const int wh",
    "            if constexpr (abl != 4)\n#pragma unroll\n            for (int kk = 0; kk < BK / 16; ++kk) {\n                const int wh")
rep("                H2A_TERM(al, bh)\n                H2A_TERM(ah, bl)\n                H2A_TERM(ah, bh)\n",
    "                if constexpr (abl != 5) {\n                H2A_TERM(al, bh)\n                H2A_TERM(ah, bl)\n                H2A_TERM(ah, bh)\n                } else { asm volatile(\"\" :: \"v\"(ah[0]), \"v\"(al[0]), \"v\"(bh[0]), \"v\"(bl[0]), \"v\"(ah[TM - 1]), \"v\"(al[TM - 1]), \"v\"(bh[TN - 1]), \"v\"(bl[TN - 1])); }\n")
rep("        wait_loads<0>();\n        __syncthreads();                                   // ... for every wave's requests: nothing lands in `stage` from here on\n",
    "        wait_loads<0>();\n        __syncthreads();                                   // ... for every wave's requests: nothing lands in `stage` from here on\n        if constexpr (abl == 6) {\n            if constexpr (decltype(MULT)::value) {\n                _Pragma(\"unroll\") for (int ti = 0; ti < TM; ++ti) _Pragma(\"unroll\") for (int tj = 0; tj < TN; ++tj) _Pragma(\"unroll\") for (int e = 0; e < 16; ++e) asm volatile(\"\" :: \"v\"(acc[ti][tj][e]));\n            }\n            return;\n        }\n")
open(p, "w").write(s)
b = d + "/tools/gemm_bench.hip"
t = open(b).read()
old = "    float ms; CK(hipEventElapsedTime(&ms, e0, e1));\n    return ms / reps;\n}"
assert t.count(old) == 1
t = t.replace(old, """    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (b.tm == 5400) {
        std::vector<unsigned long long> h(2048);
        CK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(vsr::g_h2a_life), h.size() * 8));
        double v = 0, mx = 0; int n = b.a.G < 2048 ? b.a.G : 2048;
        for (int i = 0; i < n; ++i) { v += h[i]; if (h[i] > mx) mx = h[i]; }
        printf("    workgroup lifetime: mean %.0f / max %.0f s_memtime ticks; kernel %.1f us -> max lifetime / kernel time = %.2f ticks per ns\\n", v / n, mx, ms / reps * 1e3, mx / (ms / reps * 1e6));
    }
    return ms / reps;
}""")
open(b, "w").write(t)
procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-DH2A_ABL=%d" % n, "-o", ROOT + "/tools/gemm_bench_abl%d" % n, b]) for n in (0, 1, 2, 3, 4, 6)]
assert all(p.wait() == 0 for p in procs)
print("built tools/gemm_bench_abl0..4, 6")
