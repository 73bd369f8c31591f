// Helpers shared by the kernels built on v_mfma_f32_16x16x32 with the weights as the A operand (D[cout][pixel]) and the
// interleaved fragment pairs of weights.py: pack_fragments16 -- kernels_stream.hip, kernels_exit.hip.
#pragma once
#include "gemm_common.h"

namespace {
using namespace bqk;

typedef float f32x2s __attribute__((ext_vector_type(2)));
typedef _Float16 h16x2s __attribute__((ext_vector_type(2)));
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));

template <typename T> __device__ __forceinline__ f32x4 mma16(const uint4& a, const uint4& b, const f32x4& c);
template <> __device__ __forceinline__ f32x4 mma16<f16_t>(const uint4& a, const uint4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mma16<bf16_t>(const uint4& a, const uint4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// halves of a packed dword as an fp32 pair, by plain conversions (the asm forms of H16<> cost an s_nop each here: hipcc pads
// every asm result it cannot see the latency of)
template <typename T> __device__ __forceinline__ f32x2s unpack2(unsigned u);
template <> __device__ __forceinline__ f32x2s unpack2<f16_t>(unsigned u) {
    return __builtin_convertvector(__builtin_bit_cast(h16x2s, u), f32x2s);
}
template <> __device__ __forceinline__ f32x2s unpack2<bf16_t>(unsigned u) {
    return (f32x2s){__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
}

// maximum of two packed pairs (exact: a maximum rounds nothing).  f16: ONE v_pk_max_f16 -- through
// __builtin_elementwise_max hipcc canonicalises both operands first, three instructions
template <typename T> __device__ __forceinline__ unsigned pmax2(unsigned a, unsigned b);
template <> __device__ __forceinline__ unsigned pmax2<f16_t>(unsigned a, unsigned b) {
    unsigned d;
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
template <> __device__ __forceinline__ unsigned pmax2<bf16_t>(unsigned a, unsigned b) {
    const f32x2s x = unpack2<bf16_t>(a), y = unpack2<bf16_t>(b);
    return H16<bf16_t>::pack2(fmaxf(x.x, y.x), fmaxf(x.y, y.y));
}
// T(float(a) + float(b)) per half: fp32 sum of the two exact values, rounded once to the storage type (the pooling kernels'
// `(T)(mx + (float)re)`).  f16: v_fma_mix_f32 reads both halves in place -- a * 1.0 + b, one rounding: the same sum
template <typename T> __device__ __forceinline__ unsigned padd2(unsigned a, unsigned b);
template <> __device__ __forceinline__ unsigned padd2<f16_t>(unsigned a, unsigned b) {
    float lo, hi;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(lo) : "v"(a), "v"(b));
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(hi) : "v"(a), "v"(b));
    return H16<f16_t>::pack2(lo, hi);
}
template <> __device__ __forceinline__ unsigned padd2<bf16_t>(unsigned a, unsigned b) {
    const f32x2s x = unpack2<bf16_t>(a), y = unpack2<bf16_t>(b);
    return H16<bf16_t>::pack2(x.x + y.x, x.y + y.y);
}
template <typename T> struct NegInf;
template <> struct NegInf<f16_t> { static constexpr unsigned v = 0xfc00fc00u; };
template <> struct NegInf<bf16_t> { static constexpr unsigned v = 0xff80ff80u; };

// folded BN of fragment pair q: lane (pixel px, group g) holds channels 32 q + 8 g .. + 7; sbq = scale + 8 g (bias at + nb)
template <typename T>
__device__ __forceinline__ void bn_pair4(const f32x4& u, const f32x4& v, const float4& s0, const float4& s1, const float4& b0,
                                         const float4& b1, unsigned (&o)[4]) {
    o[0] = H16<T>::pack2(fmaf(u[0], s0.x, b0.x), fmaf(u[1], s0.y, b0.y));
    o[1] = H16<T>::pack2(fmaf(u[2], s0.z, b0.z), fmaf(u[3], s0.w, b0.w));
    o[2] = H16<T>::pack2(fmaf(v[0], s1.x, b1.x), fmaf(v[1], s1.y, b1.y));
    o[3] = H16<T>::pack2(fmaf(v[2], s1.z, b1.z), fmaf(v[3], s1.w, b1.w));
}
// (the four table reads as 16-byte LDS reads: through `float4` hipcc emits ds_read2_b64 -- twice the LDS cycles of a
//  ds_read_b128 -- although every address here is a multiple of 32 bytes; a uint4 read is a ds_read_b128)
__device__ __forceinline__ float4 lds_f4(const float* p) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
template <typename T>
__device__ __forceinline__ void bn_pair(const f32x4& u, const f32x4& v, const float* sbq, int nb, unsigned (&o)[4]) {
    bn_pair4<T>(u, v, lds_f4(sbq), lds_f4(sbq + 4), lds_f4(sbq + nb), lds_f4(sbq + nb + 4), o);
}

}  // namespace
