#!/usr/bin/env python3
"""In-kernel phase stamps of the f16x2 wide kernel (csrc/gemm_h2.h) WITHOUT hooks in the shipped header: csrc/ and tools/ are copied to a
scratch directory, the copy of gemm_h2.h gets s_memtime stamps around the phases of a k-tile - mover wave 8: issue of the weight DMAs and A loads | wait for the requests of the next k-tile |
A split + LDS stores | barrier (+ flush); multiplier wave 0: LDS reads + MFMAs | barrier (+ flush) - accumulated per
workgroup into a __device__ array, and tools/gemm_bench is built from it as tools/gemm_bench_stamp (its timing runs print the table:
`tools/gemm_bench_stamp 500 256 4 5200 1`).  Stamps cost ~10 % (guide): the numbers are shares of a k-tile, not absolute times.
usage: tools/h2_stamp.py"""
import os, shutil, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = "/tmp/h2stamp"
shutil.rmtree(d, ignore_errors=True)
os.makedirs(d + "/vsr-guided-cic_amd")
shutil.copytree(ROOT + "/vsr-guided-cic_amd/csrc", d + "/vsr-guided-cic_amd/csrc")
os.makedirs(d + "/tools")
for f in os.listdir(ROOT + "/tools"):
    if f.endswith((".hip", ".h")):
        shutil.copy(ROOT + "/tools/" + f, d + "/tools/" + f)
p = d + "/vsr-guided-cic_amd/csrc/gemm_h2.h"
s = open(p).read()
def rep(old, new):
    global s
    assert s.count(old) == 1, (s.count(old), old[:80])
    s = s.replace(old, new)
rep("namespace vsr {\n\ntypedef _Float16 f16x8_t",
    "namespace vsr {\n\n__device__ unsigned long long g_h2_stamp[8 * 2048];\n#define H2T() __builtin_amdgcn_s_memtime()\n\ntypedef _Float16 f16x8_t")
# movers: one period = [issue W DMA + A loads] [wait] [A split + store] [barrier]
rep("            const bool more_w = it + NW - 1 < it1, more_a = it + 2 < it1;\n",
    "            const unsigned long long t0 = H2T();\n            const bool more_w = it + NW - 1 < it1, more_a = it + 2 < it1;\n")
rep("            if (more_a) issue_a(SP);\n            if (it + 1 < it1) {\n",
    "            if (more_a) issue_a(SP);\n            const unsigned long long t1 = H2T();\n            unsigned long long t2 = t1, t3 = t1;\n            if (it + 1 < it1) {\n")
rep("                else wait_loads<0>();\n                store_a(SN, (j + 1) & 1);\n            }\n            end_of_ktile(std::false_type{});\n        };\n",
    "                else wait_loads<0>();\n                t2 = H2T();\n                store_a(SN, (j + 1) & 1);\n                asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n                t3 = H2T();\n            }\n            end_of_ktile(std::false_type{});\n            const unsigned long long t4 = H2T();\n            si += t1 - t0; sw += t2 - t1; ss += t3 - t2; sb += t4 - t3; ++nk;\n        };\n")
rep("        // k-tile j: issue W(j + NW - 1) into the stage freed by the last barrier and A(j + 2) into the set stored last k-tile; wait for\n",
    "        unsigned long long sw = 0, ss = 0, si = 0, sb = 0, nk = 0;\n        // k-tile j: issue W(j + NW - 1) into the stage freed by the last barrier and A(j + 2) into the set stored last k-tile; wait for\n")
rep("            if (it < it1) period(S0{}, S1{});              // j odd\n        }\n",
    "            if (it < it1) period(S0{}, S1{});              // j odd\n        }\n        if (tid == 512 && g < 2048) { g_h2_stamp[8 * g + 0] = sw; g_h2_stamp[8 * g + 1] = ss; g_h2_stamp[8 * g + 2] = si; g_h2_stamp[8 * g + 3] = sb; g_h2_stamp[8 * g + 4] = nk; }\n")
# multipliers
rep("        zero_acc();\n        __syncthreads();\n        while (it < it1) {\n            const uint16_t* a_row = smem + (j & 1) * ABUF",
    "        zero_acc();\n        __syncthreads();\n        unsigned long long sm_ = 0, sbm = 0;\n        while (it < it1) {\n            const unsigned long long m0_ = H2T();\n            const uint16_t* a_row = smem + (j & 1) * ABUF")
rep("            if (end_of_ktile(std::true_type{})) zero_acc();\n        }\n    }\n}\n\n// ------------------------------------------------------------------------------------------------------------------------------\n// Streaming kernel",
    "            const unsigned long long m1_ = H2T();\n            if (end_of_ktile(std::true_type{})) zero_acc();\n            const unsigned long long m2_ = H2T();\n            sm_ += m1_ - m0_; sbm += m2_ - m1_;\n        }\n        if (tid == 0 && g < 2048) { g_h2_stamp[8 * g + 5] = sm_; g_h2_stamp[8 * g + 6] = sbm; }\n    }\n}\n\n// ------------------------------------------------------------------------------------------------------------------------------\n// Streaming kernel")
open(p, "w").write(s)
# the tool prints the table after every timed launch loop
p = d + "/tools/gemm_bench.hip"
t = open(p).read()
old = "    float ms; CK(hipEventElapsedTime(&ms, e0, e1));\n    return ms / reps;\n}"
assert t.count(old) == 1
t = t.replace(old, """    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (b.tm == 5200) {
        std::vector<unsigned long long> h(8 * 2048);
        CK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(vsr::g_h2_stamp), h.size() * 8));
        double v[7] = {0}; int n = b.a.G < 2048 ? b.a.G : 2048;
        for (int i = 0; i < n; ++i) for (int q = 0; q < 7; ++q) v[q] += h[8 * i + q];
        const double nk = v[4] > 0 ? v[4] : 1;
        printf("    per k-tile (s_memtime ticks = shader cycles, mean over %d workgroups): mover wave  issue %.1f | wait %.1f | A split + LDS stores %.1f | barrier (+flush) %.1f = %.1f;  multiplier wave  LDS reads + 24 MFMAs %.1f | barrier (+flush) %.1f = %.1f\\n",
               n, v[2] / nk, v[0] / nk, v[1] / nk, v[3] / nk, (v[0] + v[1] + v[2] + v[3]) / nk, v[5] / nk, v[6] / nk, (v[5] + v[6]) / nk);
    }
    return ms / reps;
}""")
open(p, "w").write(t)
out = ROOT + "/tools/gemm_bench_stamp"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-o", out, d + "/tools/gemm_bench.hip"], check=True)
print(out)
