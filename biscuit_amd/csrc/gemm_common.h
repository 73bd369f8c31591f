// Device helpers shared by the fused MFMA kernels (kernels_gemm.hip, kernels_pipe.hip).
#pragma once
#include "bq_common.h"

namespace bqk {

template <typename T> struct TT;
template <> struct TT<bf16_t> { static constexpr int VEC = 8; };
template <> struct TT<f16_t> { static constexpr int VEC = 8; };
template <> struct TT<float> { static constexpr int VEC = 4; };

// The two 16-bit storage formats behind one interface: halves of a packed dword to fp32 (exact), an fp32 pair to a
// packed dword (round to nearest even, one instruction), ReLU on a packed dword (both formats are sign-magnitude:
// a signed 16-bit max with 0 per half), the mnemonics of the matrix instructions.
template <typename T> struct H16;
template <> struct H16<bf16_t> {
    static constexpr bool F16 = false;
    static __device__ __forceinline__ float lo(unsigned u) { return __uint_as_float(u << 16); }
    static __device__ __forceinline__ float hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        unsigned o;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b));
        return o;
    }
    // d = a * b + c with a = the low / high half of a packed dword, b and c fp32 (one rounding)
    static __device__ __forceinline__ float fma_lo(unsigned u, float b, float c) { return fmaf(lo(u), b, c); }
    static __device__ __forceinline__ float fma_hi(unsigned u, float b, float c) { return fmaf(hi(u), b, c); }
    // c + the low / high half (one rounding)
    static __device__ __forceinline__ float add_lo(float c, unsigned u) { return c + lo(u); }
    static __device__ __forceinline__ float add_hi(float c, unsigned u) { return c + hi(u); }
};
template <> struct H16<f16_t> {
    static constexpr bool F16 = true;
    static __device__ __forceinline__ float lo(unsigned u) {
        float f;
        asm("v_cvt_f32_f16_e32 %0, %1" : "=v"(f) : "v"(u));
        return f;
    }
    static __device__ __forceinline__ float hi(unsigned u) {
        float f;
        asm("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(f) : "v"(u));
        return f;
    }
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        unsigned o;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b));
        return o;
    }
    // v_fma_mix_f32 takes the half straight out of the packed dword: the product and the sum are formed from the
    // exact fp32 values, rounded once -- the same result as lo()/hi() followed by fmaf, without the conversion
    static __device__ __forceinline__ float fma_lo(unsigned u, float b, float c) {
        float d;
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(u), "v"(b), "v"(c));
        return d;
    }
    static __device__ __forceinline__ float fma_hi(unsigned u, float b, float c) {
        float d;
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(u), "v"(b), "v"(c));
        return d;
    }
    static __device__ __forceinline__ float add_lo(float c, unsigned u) {
        float d;
        asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(u), "v"(c));
        return d;
    }
    static __device__ __forceinline__ float add_hi(float c, unsigned u) {
        float d;
        asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(u), "v"(c));
        return d;
    }
};

__device__ __forceinline__ unsigned relu_pk16(unsigned x) {   // ReLU on two packed bf16 / f16: one v_pk_max_i16
    typedef short s16x2r __attribute__((ext_vector_type(2)));
    const s16x2r z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2r, x), z));
}

template <typename T> __device__ __forceinline__ void unpack(const uint4& v, float* f);
template <> __device__ __forceinline__ void unpack<bf16_t>(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
template <> __device__ __forceinline__ void unpack<f16_t>(const uint4& v, float* f) {
    typedef H16<f16_t> F;
    f[0] = F::lo(v.x); f[1] = F::hi(v.x); f[2] = F::lo(v.y); f[3] = F::hi(v.y);
    f[4] = F::lo(v.z); f[5] = F::hi(v.z); f[6] = F::lo(v.w); f[7] = F::hi(v.w);
}
template <> __device__ __forceinline__ void unpack<float>(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y);
    f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
}

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    const unsigned short a = __builtin_bit_cast(unsigned short, (bf16_t)lo);
    const unsigned short b = __builtin_bit_cast(unsigned short, (bf16_t)hi);
    return (unsigned)a | ((unsigned)b << 16);
}

template <typename T> __device__ __forceinline__ uint4 pack(const float* f);
template <> __device__ __forceinline__ uint4 pack<bf16_t>(const float* f) {
    return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]),
                      pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
}
template <> __device__ __forceinline__ uint4 pack<f16_t>(const float* f) {
    typedef H16<f16_t> F;
    return make_uint4(F::pack2(f[0], f[1]), F::pack2(f[2], f[3]), F::pack2(f[4], f[5]), F::pack2(f[6], f[7]));
}
template <> __device__ __forceinline__ uint4 pack<float>(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]),
                      __float_as_uint(f[3]));
}

// 4 consecutive output channels of one pixel
template <typename T> __device__ __forceinline__ void load4(const T* p, float* v);
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float* v) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
}
template <> __device__ __forceinline__ void load4<f16_t>(const f16_t* p, float* v) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    v[0] = H16<f16_t>::lo(u.x); v[1] = H16<f16_t>::hi(u.x); v[2] = H16<f16_t>::lo(u.y); v[3] = H16<f16_t>::hi(u.y);
}
template <> __device__ __forceinline__ void load4<float>(const float* p, float* v) {
    const float4 u = *reinterpret_cast<const float4*>(p);
    v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w;
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float* v);
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float* v) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
}
template <> __device__ __forceinline__ void store4<f16_t>(f16_t* p, const float* v) {
    *reinterpret_cast<uint2*>(p) = make_uint2(H16<f16_t>::pack2(v[0], v[1]), H16<f16_t>::pack2(v[2], v[3]));
}
template <> __device__ __forceinline__ void store4<float>(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}

template <typename T>
__device__ __forceinline__ void mma(f32x16& acc, const uint4& w, const uint4& a);
template <> __device__ __forceinline__ void mma<bf16_t>(f32x16& acc, const uint4& w, const uint4& a) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w),
                                                   __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma<f16_t>(f32x16& acc, const uint4& w, const uint4& a) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w),
                                                  __builtin_bit_cast(f16x8, a), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma<float>(f32x16& acc, const uint4& w, const uint4& a) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.x), __uint_as_float(a.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.y), __uint_as_float(a.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.z), __uint_as_float(a.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.w), __uint_as_float(a.w), acc, 0, 0, 0);
}

// Row iterator over flattened (image, y, x) pixel indices of an H x W map.
struct PixIt {
    int img, y, x;
    __device__ __forceinline__ void init(int p, int H, int W) {
        const int hw = H * W;
        img = p / hw;
        const int rem = p - img * hw;
        y = rem / W;
        x = rem - y * W;
    }
    __device__ __forceinline__ void advance(int step, int H, int W) {
        x += step;
        while (x >= W) { x -= W; ++y; }
        while (y >= H) { y -= H; ++img; }
    }
};


// Epilogue of one wave's accumulators: folded BN (scale/bias), optional residual add,
// optional ReLU, NHWC store.  D layout of v_mfma_*_32x32: lane&31 = pixel (column),
// reg -> output channel row (reg&3) + 8*(reg>>2) + 4*(lane>>5): four consecutive channels
// per register quad, i.e. 8-byte (bf16) / 16-byte (fp32) stores.
template <typename T, int RM, int RN>
__device__ __forceinline__ void epilogue(const GemmParams& p, const f32x16 (&acc)[RM][RN], int nfb,
                                         int m_first, int r32, int h) {
    T* __restrict__ out = reinterpret_cast<T*>(p.out);
    const T* __restrict__ res = reinterpret_cast<const T*>(p.residual);
#pragma unroll
    for (int j = 0; j < RN; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n0 = (nfb + j) * 32 + g * 8 + h * 4;
            if (n0 < p.Nstore) {
                float sc[4] = {1.f, 1.f, 1.f, 1.f}, bi[4] = {0.f, 0.f, 0.f, 0.f};
                if (p.scale) {
                    const float4 t = *reinterpret_cast<const float4*>(p.scale + n0);
                    sc[0] = t.x; sc[1] = t.y; sc[2] = t.z; sc[3] = t.w;
                }
                if (p.bias) {
                    const float4 t = *reinterpret_cast<const float4*>(p.bias + n0);
                    bi[0] = t.x; bi[1] = t.y; bi[2] = t.z; bi[3] = t.w;
                }
#pragma unroll
                for (int i = 0; i < RM; ++i) {
                    const int m = m_first + i * 32 + r32;
                    if (m < p.M) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaf(acc[i][j][g * 4 + e], sc[e], bi[e]);
                        const size_t o = (size_t)m * p.ldo + n0;
                        if (res) {
                            float rv[4];
                            load4<T>(res + o, rv);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += rv[e];
                        }
                        if (p.relu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                        }
                        store4<T>(out + o, v);
                    }
                }
            }
        }
    }
}

// Coalesced epilogue: the accumulator layout gives every lane only 4 consecutive channels
// of one pixel, so storing straight from registers writes 16-byte crumbs into 32 different
// rows per instruction (measured: ~1 TB/s, 40 % of the 728-wide kernel).  Instead finish the
// arithmetic in registers (folded BN, residual, ReLU), park the tile in LDS in its final
// dtype, and let the whole workgroup stream it out row by row in 16-byte pieces: every store
// instruction then writes whole, contiguous 128-byte lines.
// Caller guarantees: all waves are past their last LDS read (barrier) and `smem` has room for
// MT rows of (Nstore*sizeof(T) + 16) bytes.
// Row stride = an ODD number of 16-byte pieces: the 32 rows one staging write touches then start in 16
// different bank groups (2-way conflict, the minimum for 32 x 16 bytes).  728 channels + 16 bytes of padding
// was 92 pieces: 4 bank groups, an 8-way conflict on every write (stamps: 11.6 k cycles for acc -> LDS).
template <typename T>
__device__ __forceinline__ int stage_stride(int nstore) {
    const int bytes = nstore * (int)sizeof(T);
    return ((bytes >> 4) & 1) ? bytes : bytes + 16;
}

template <typename T, int RM, int RN>
__device__ __forceinline__ void epilogue_to_lds(const GemmParams& p, const f32x16 (&acc)[RM][RN], int nfb,
                                                int row_local0, int m0, int r32, int h, unsigned char* smem) {
    const T* __restrict__ res = reinterpret_cast<const T*>(p.residual);
    const int sstride = stage_stride<T>(p.Nstore);
#pragma unroll
    for (int j = 0; j < RN; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n0 = (nfb + j) * 32 + g * 8 + h * 4;
            if (n0 < p.Nstore) {
                float sc[4] = {1.f, 1.f, 1.f, 1.f}, bi[4] = {0.f, 0.f, 0.f, 0.f};
                if (p.scale) {
                    const float4 t = *reinterpret_cast<const float4*>(p.scale + n0);
                    sc[0] = t.x; sc[1] = t.y; sc[2] = t.z; sc[3] = t.w;
                }
                if (p.bias) {
                    const float4 t = *reinterpret_cast<const float4*>(p.bias + n0);
                    bi[0] = t.x; bi[1] = t.y; bi[2] = t.z; bi[3] = t.w;
                }
#pragma unroll
                for (int i = 0; i < RM; ++i) {
                    const int rl = row_local0 + i * 32 + r32;
                    const int m = m0 + rl;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(acc[i][j][g * 4 + e], sc[e], bi[e]);
                    if (res && m < p.M) {
                        float rv[4];
                        load4<T>(res + (size_t)m * p.ldo + n0, rv);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += rv[e];
                    }
                    if (p.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    store4<T>(reinterpret_cast<T*>(smem + (size_t)rl * sstride) + n0, v);
                }
            }
        }
    }
}

template <typename T, int NT, int MT>
__device__ __forceinline__ void lds_rows_to_global(const GemmParams& p, int m0, int tid, const unsigned char* smem) {
    const int sstride = stage_stride<T>(p.Nstore);
    const int ppr = p.Nstore * (int)sizeof(T) / 16;       // 16-byte pieces per row
    const int cpp = ppr < NT ? ppr : NT;
    const int RF = NT / cpp;
    const int tc = tid % cpp, tr = tid / cpp;
    if (tr >= RF) return;
    unsigned char* __restrict__ out = reinterpret_cast<unsigned char*>(p.out);
    const size_t row_bytes = (size_t)p.ldo * sizeof(T);
    for (int pc = tc; pc < ppr; pc += cpp)
        for (int r = tr; r < MT; r += RF) {
            const int m = m0 + r;
            if (m < p.M)
                *reinterpret_cast<uint4*>(out + (size_t)m * row_bytes + pc * 16) =
                    *reinterpret_cast<const uint4*>(smem + (size_t)r * sstride + pc * 16);
        }
}

// XCD-aware block -> tile map (bijective): blocks b, b+8, ... share an XCD/L2; give them
// neighbouring pixel tiles so depthwise halos and weights are L2 hits.
__device__ __forceinline__ int xcd_tile(int bid, int nwg) {
    const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
    return (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
}

}  // namespace bqk
