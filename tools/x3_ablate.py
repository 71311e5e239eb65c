#!/usr/bin/env python3
"""Timing ablations of the f32x3 kernel (csrc/gemm_x3.h) WITHOUT hooks in the shipped header: the tree's csrc/ and tools/ are copied
to a scratch directory, one textual patch is applied to the copy of gemm_x3.h, and tools/gemm_bench is built from it as
tools/gemm_bench_abl_<name>.  The patched kernels compute garbage - only their launch times mean anything.
  nosplit   movers store the raw fp32 bits into the three planes (no split VALU)
  nomfma    multipliers read their operands from LDS but issue no MFMA
  nolds     multipliers issue their MFMAs on registers they never load (no LDS reads)
  nomove    movers neither load, split nor store: barriers only
  nostore   movers load and split, the planes are not written to LDS
usage: tools/x3_ablate.py [names...]   (default: all)"""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCHES = {
    "nosplit": [("            split3(v.x, v.y, h0, m0, l0);\n            split3(v.z, v.w, h1, m1, l1);\n",
                 "            h0 = __float_as_uint(v.x); m0 = __float_as_uint(v.y); l0 = h0; h1 = __float_as_uint(v.z); m1 = __float_as_uint(v.w); l1 = h1;\n")],
    "nomfma": [("            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X[i], Y[j], acc[i][j], 0, 0, 0);",
                "            asm volatile(\"\" :: \"v\"(X[i]), \"v\"(Y[j]));")],
    "nolds": [("                    ah[i] = *reinterpret_cast<const bf16x8_t*>(a_row + i * 32 * X3_ROW + ch);\n"
               "                    am[i] = *reinterpret_cast<const bf16x8_t*>(a_row + PLANE + i * 32 * X3_ROW + ch);\n"
               "                    al[i] = *reinterpret_cast<const bf16x8_t*>(a_row + 2 * PLANE + i * 32 * X3_ROW + ch);\n",
               "                    asm volatile(\"\" : \"=v\"(ah[i]), \"=v\"(am[i]), \"=v\"(al[i]));\n"),
              ("                    bh[j] = *reinterpret_cast<const bf16x8_t*>(b_row + j * 32 * X3_ROW + ch);\n"
               "                    bm[j] = *reinterpret_cast<const bf16x8_t*>(b_row + PLANE + j * 32 * X3_ROW + ch);\n"
               "                    bl[j] = *reinterpret_cast<const bf16x8_t*>(b_row + 2 * PLANE + j * 32 * X3_ROW + ch);\n",
               "                    asm volatile(\"\" : \"=v\"(bh[j]), \"=v\"(bm[j]), \"=v\"(bl[j]));\n")],
    "nomove": [("            for (int i = 0; i < LA; ++i) async_load16(ra[s][i], pa[i] + ko);\n", "            for (int i = 0; i < 0; ++i) async_load16(ra[s][i], pa[i] + ko);\n"),
               ("            for (int i = 0; i < LB; ++i) async_load16(rb[s][i], pb[i] + ko);\n", "            for (int i = 0; i < 0; ++i) async_load16(rb[s][i], pb[i] + ko);\n"),
               ("            if (other_in_flight) wait_loads<LA + LB>(); else wait_loads<0>();\n", ""),
               ("            for (int i = 0; i < LA; ++i) landed(ra[s][i]);\n", "            for (int i = 0; i < 0; ++i) landed(ra[s][i]);\n"),
               ("            for (int i = 0; i < LB; ++i) landed(rb[s][i]);\n", "            for (int i = 0; i < 0; ++i) landed(rb[s][i]);\n"),
               ("            for (int i = 0; i < LA; ++i) put(buf, lrow + 64 * i, ra[s][i], stl[s]);\n", "            for (int i = 0; i < 0; ++i) put(buf, lrow + 64 * i, ra[s][i], stl[s]);\n"),
               ("            for (int i = 0; i < LB; ++i) put(buf, BM + lrow + 64 * i, rb[s][i], stl[s]);\n", "            for (int i = 0; i < 0; ++i) put(buf, BM + lrow + 64 * i, rb[s][i], stl[s]);\n")],
}
PATCHES["nostore"] = [("            *reinterpret_cast<uint2*>(buf + pos) = make_uint2(h0, h1);\n"
                       "            *reinterpret_cast<uint2*>(buf + PLANE + pos) = make_uint2(m0, m1);\n"
                       "            *reinterpret_cast<uint2*>(buf + 2 * PLANE + pos) = make_uint2(l0, l1);\n",
                       "            asm volatile(\"\" :: \"v\"(h0), \"v\"(h1), \"v\"(m0), \"v\"(m1), \"v\"(l0), \"v\"(l1), \"v\"(buf + pos));\n")]
names = sys.argv[1:] or list(PATCHES)
for n in names:
    d = "/tmp/x3abl_" + n
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d + "/vsr-guided-cic_amd")
    shutil.copytree(ROOT + "/vsr-guided-cic_amd/csrc", d + "/vsr-guided-cic_amd/csrc")
    os.makedirs(d + "/tools")
    for f in os.listdir(ROOT + "/tools"):
        if f.endswith((".hip", ".h")):
            shutil.copy(ROOT + "/tools/" + f, d + "/tools/" + f)
    p = d + "/vsr-guided-cic_amd/csrc/gemm_x3.h"
    s = open(p).read()
    for old, new in PATCHES[n]:
        assert old in s, (n, old[:60])
        s = s.replace(old, new)
    open(p, "w").write(s)
    out = ROOT + "/tools/gemm_bench_abl_" + n
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-o", out, d + "/tools/gemm_bench.hip"], check=True)
    print(out)
