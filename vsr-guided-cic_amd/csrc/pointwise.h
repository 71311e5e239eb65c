// Pointwise arithmetic shared by the decoder's pointwise kernels (kernels.h) and by the in-launch epilogues of the GEMM kernels
// (gemm_epi.h): ONE definition per expression, so that both paths compile the same floating-point operations in the same order.
// Reference equations: /root/reference/models/controllable_captioning.py:151-154 (LSTM1 + gates), :176-177 (LSTM2), :181-182 (g_t).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vsr {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// bf16 image (round-to-nearest-even) of a value / of four consecutive values: the bf16 GEMM mode lets the producers of its A
// operands write this image next to the fp32 value, so that the GEMM loads half the bytes and converts nothing
__device__ __forceinline__ uint16_t bf16_bits(float x) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {x, 0.f};
    return (uint16_t)(__builtin_bit_cast(uint32_t, __builtin_convertvector(v, b2)) & 0xffffu);
}
__device__ __forceinline__ uint2 bf16_bits4(float4 v) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 lo = {v.x, v.y}, hi = {v.z, v.w};
    return make_uint2(__builtin_bit_cast(uint32_t, __builtin_convertvector(lo, b2)), __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, b2)));
}

// Images of A operands for the GEMM kernels that take them (GemmSeg::A16), written by the producers next to the fp32 values.
// isc == 0: bf16, 2 bytes per element (bf16 mode).  isc > 0: the f16x2 flavour's fp16 pairs of x * isc in the fp32 matrix's byte geometry
// (gemm_h2a.h: both operands go global -> LDS by DMA, nothing is converted in the GEMM): elements [8 g, 8 g + 8) -> [hi x 8 | lo x 8],
// `img` then addresses 2-byte units of a 4-byte-per-element buffer.  idx = row * ld + column with ld a multiple of 8.
__device__ __forceinline__ void img_store(uint16_t* __restrict__ img, long long idx, float v, float isc) {
    if (isc == 0.f) { img[idx] = bf16_bits(v); return; }
    const float x = v * isc;
    const _Float16 hi = (_Float16)x;
    const _Float16 lo = (_Float16)(x - (float)hi);
    uint16_t* g = img + ((idx >> 3) << 4) + (idx & 7);
    g[0] = __builtin_bit_cast(uint16_t, hi);
    g[8] = __builtin_bit_cast(uint16_t, lo);
}
__device__ __forceinline__ void img_store4(uint16_t* __restrict__ img, long long idx /* a multiple of 4 */, float4 v, float isc) {
    if (isc == 0.f) { *reinterpret_cast<uint2*>(img + idx) = bf16_bits4(v); return; }
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 x = {v.x * isc, v.y * isc, v.z * isc, v.w * isc};
    const h4 hi = __builtin_convertvector(x, h4);
    const f4 rr = x - __builtin_convertvector(hi, f4);
    const h4 lo = __builtin_convertvector(rr, h4);
    uint16_t* g = img + ((idx >> 3) << 4) + (idx & 7);
    *reinterpret_cast<uint2*>(g) = __builtin_bit_cast(uint2, hi);
    *reinterpret_cast<uint2*>(g + 8) = __builtin_bit_cast(uint2, lo);
}

// LSTM cell from its four pre-activations (PyTorch gate order i, f, g, o)                     step :152 / :177
__device__ __forceinline__ void lstm_cell(float qi, float qf, float qg, float qo, float c_old, float& h, float& c) {
    c = sigmoidf_(qf) * c_old + sigmoidf_(qi) * tanhf(qg);
    h = sigmoidf_(qo) * tanhf(c);
}

// shift-gate vector g_t = sigmoid(gpre + W1_hg h1_new) * tanh(c1_new)                       step :181-182
__device__ __forceinline__ float gt_cell(float gpre, float s, float c1n) { return sigmoidf_(gpre + s) * tanhf(c1n); }

}  // namespace vsr
