// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.
// Wavefront = 64 lanes; MFMA fragments follow the gfx950 16x16x32 bf16 maps:
//   A: lane l holds A[row l&15][k = 8*(l>>4) + j], j = 0..7
//   B: lane l holds B[k = 8*(l>>4) + j][col l&15]
//   D: lane l holds D[row = 4*(l>>4) + j][col l&15], j = 0..3
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define DLAMD_WAVE 64
#define DLAMD_NUM_XCD 8

enum DlamdAct : int {
  ACT_NONE = 0,
  ACT_RELU = 1,
  ACT_HSWISH = 2,
  ACT_GELU = 3,
  ACT_SIGMOID = 4,
  ACT_TANH = 5,
};

__device__ __forceinline__ float dlamd_act(float v, int act) {
  switch (act) {
    case ACT_RELU: return v > 0.f ? v : 0.f;
    case ACT_HSWISH: {
      float t = fminf(fmaxf(v + 3.f, 0.f), 6.f);
      return v * t * (1.f / 6.f);
    }
    case ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    case ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
    case ACT_TANH: return tanhf(v);
    default: return v;
  }
}

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ u32x4 as_u32x4(bf16x8 v) { return __builtin_bit_cast(u32x4, v); }

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t bits16) {
  return __builtin_bit_cast(float, bits16 << 16);
}

// XCD-aware bijective block remap: hardware deals consecutive workgroup ids
// round-robin over the 8 XCDs (private L2 each).  Give every XCD a contiguous
// range of logical tile ids so neighbouring tiles (which share conv halos /
// weight panels) hit the same L2.  Valid for any grid size.
__device__ __forceinline__ int dlamd_xcd_remap(int bid, int nwg) {
  int q = nwg / DLAMD_NUM_XCD, r = nwg % DLAMD_NUM_XCD;
  int xcd = bid % DLAMD_NUM_XCD;
  int idx = bid / DLAMD_NUM_XCD;
  int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + idx;
}

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

#define DLAMD_CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? 0 : -1)
