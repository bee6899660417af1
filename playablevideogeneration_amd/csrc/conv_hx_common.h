// Shared pieces of the split-operand kernels (conv_hx.hip: forward / dgrad tiles + weight packing; conv_hx_wgrad.hip: weight gradients).  Internal linkage: each translation unit
// gets its own copy.
#pragma once
#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Vec;
template <> struct Vec<_Float16> { typedef f16x8 v8; typedef f16x4 v4; };
template <> struct Vec<__bf16> { typedef bf16x8 v8; typedef bf16x4 v4; };

__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

constexpr int KC = HX_KC;      // channels per chunk
#define HX_F16_MAX 65504.f
struct SegRefH { const float* p; long sn; int ld; int C; int bcast; int c0; int idx; const float* bn_scale; const float* bn_shift; int bn_act; int bn_gn; long bn_gs; };
template <typename T> struct is_bf16 { static constexpr bool value = false; };
template <> struct is_bf16<__bf16> { static constexpr bool value = true; };


}  // namespace
