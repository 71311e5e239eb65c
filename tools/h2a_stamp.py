#!/usr/bin/env python3
"""In-kernel phase stamps of the all-DMA wide f16x2 kernel (csrc/gemm_h2a.h), like tools/h2_stamp.py for gemm_h2.h: a scratch COPY of the header
gets s_memtime stamps - mover wave 8: issue of the k-tile's DMAs | counted wait | barrier; multiplier wave 0: LDS reads + MFMAs | barrier -
accumulated per workgroup (k-tiles that end a piece, i.e. carry a flush, are left out), tools/gemm_bench is built from it as
tools/gemm_bench_stamp_h2a and prints the table after its timing runs: `tools/gemm_bench_stamp_h2a 500 256 4 5400 1` (GEMM_PLAN_ALIGNED=4).
Stamps cost ~10 %: the numbers are SHARES of a k-tile.  usage: tools/h2a_stamp.py"""
import os, shutil, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = "/tmp/h2astamp"
shutil.rmtree(d, ignore_errors=True)
os.makedirs(d + "/vsr-guided-cic_amd")
shutil.copytree(ROOT + "/vsr-guided-cic_amd/csrc", d + "/vsr-guided-cic_amd/csrc")
shutil.copytree(ROOT + "/tools", d + "/tools", ignore=lambda p, names: [n for n in names if not (n.endswith((".hip", ".h")) or n == "experiments")])
p = d + "/vsr-guided-cic_amd/csrc/gemm_h2a.h"
s = open(p).read()
def rep(old, new):
    global s
    assert s.count(old) == 1, (s.count(old), old[:80])
    s = s.replace(old, new)
rep("namespace vsr {\n\n", "namespace vsr {\n\n__device__ unsigned long long g_h2a_stamp[8 * 2048];\n#define H2T() __builtin_amdgcn_s_memtime()\n\n")
# movers
rep("        while (it < it1) {\n            int st = ws + NW - 1;\n            st = st >= NW ? st - NW : st;\n            const bool more = it + NW - 1 < it1;\n            if (more) { issue(st); wait_loads<(NW - 2) * (LA + LB)>(); } else wait_loads<0>();\n            end_of_ktile(std::false_type{});\n        }",
    "        unsigned long long si = 0, sw = 0, sb = 0, nk = 0;\n        while (it < it1) {\n            const unsigned long long t0 = H2T();\n            int st = ws + NW - 1;\n            st = st >= NW ? st - NW : st;\n            const bool more = it + NW - 1 < it1;\n            if (more) issue(st);\n            const unsigned long long t1 = H2T();\n            if (more) wait_loads<(NW - 2) * (LA + LB)>(); else wait_loads<0>();\n            const unsigned long long t2 = H2T();\n            const bool fl = end_of_ktile(std::false_type{});\n            const unsigned long long t3 = H2T();\n            if (!fl) { si += t1 - t0; sw += t2 - t1; sb += t3 - t2; ++nk; }\n        }\n        if (tid == 512 && g < 2048) { g_h2a_stamp[8 * g + 0] = si; g_h2a_stamp[8 * g + 1] = sw; g_h2a_stamp[8 * g + 2] = sb; g_h2a_stamp[8 * g + 3] = nk; }")
# multipliers
rep("        while (it < it1) {\n            const uint16_t* b_row = sW + ws * STG + (wn * (32 * TN) + r) * 64;",
    "        unsigned long long sm_ = 0, sbm = 0, nkm = 0;\n        while (it < it1) {\n            const unsigned long long m0_ = H2T();\n            const uint16_t* b_row = sW + ws * STG + (wn * (32 * TN) + r) * 64;")
rep("            if (end_of_ktile(std::true_type{})) { zero_acc(); first = true; }\n        }\n    }\n}",
    "            const unsigned long long m1_ = H2T();\n            const bool fl = end_of_ktile(std::true_type{});\n            const unsigned long long m2_ = H2T();\n            if (fl) { zero_acc(); first = true; } else { sm_ += m1_ - m0_; sbm += m2_ - m1_; ++nkm; }\n        }\n        if (tid == 0 && g < 2048) { g_h2a_stamp[8 * g + 4] = sm_; g_h2a_stamp[8 * g + 5] = sbm; g_h2a_stamp[8 * g + 6] = nkm; }\n    }\n}")
open(p, "w").write(s)
p = d + "/tools/gemm_bench.hip"
t = open(p).read()
old = "    float ms; CK(hipEventElapsedTime(&ms, e0, e1));\n    return ms / reps;\n}"
assert t.count(old) == 1
t = t.replace(old, """    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (b.tm == 5400) {
        std::vector<unsigned long long> h(8 * 2048);
        CK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(vsr::g_h2a_stamp), h.size() * 8));
        double v[7] = {0}; int n = b.a.G < 2048 ? b.a.G : 2048;
        for (int i = 0; i < n; ++i) for (int q = 0; q < 7; ++q) v[q] += h[8 * i + q];
        const double nk = v[3] > 0 ? v[3] : 1, nm = v[6] > 0 ? v[6] : 1;
        printf("    per k-tile without a flush (s_memtime ticks, mean over %d workgroups): mover wave 8  issue %.0f | counted wait %.0f | barrier %.0f = %.0f;  multiplier wave 0  LDS reads + MFMAs %.0f | barrier %.0f = %.0f\\n",
               n, v[0] / nk, v[1] / nk, v[2] / nk, (v[0] + v[1] + v[2]) / nk, v[4] / nm, v[5] / nm, (v[4] + v[5]) / nm);
    }
    return ms / reps;
}""")
open(p, "w").write(t)
out = ROOT + "/tools/gemm_bench_stamp_h2a"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-o", out, d + "/tools/gemm_bench.hip"], check=True)
print(out)
