// Shared helpers for the gfx950 kernels of libdvae_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dvae_hip.h"

#define DVAE_API extern "C" __attribute__((visibility("default")))

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// Split mode (DVAE_MODE_F32X3): x (4 x fp32) -> three bf16x4 with x[i] == p[0][i] + p[1][i] + p[2][i] exactly (see the X3
// note in gemm.hip): round to nearest-even twice (v_cvt_pk_bf16_f32 packs two values per instruction; x - rne(x) is
// exact in fp32), the last residual has <= 8 significant bits and converts exactly.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// (written on pairs: hipcc turns the 4-wide form into one single-element conversion per value to get rne(x) back as fp32
// — 7.5 VALU operations per element; here a pair costs one v_cvt_pk_bf16_f32, a shift, a mask and a packed subtract per level)
__device__ __forceinline__ unsigned split3_pk(f32x2& r) {   // returns rne(r) as two packed bf16, r -= rne(r)
  const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
  // (inline v_sub_f32: hipcc SLP-packs plain subtracts into v_pk_add_f32, ~13 cycles dearer beside an MFMA)
  float a, b;
  asm("v_sub_f32 %0, %1, %2" : "=v"(a) : "v"(r[0]), "v"(__builtin_bit_cast(float, u << 16)));
  asm("v_sub_f32 %0, %1, %2" : "=v"(b) : "v"(r[1]), "v"(__builtin_bit_cast(float, u & 0xffff0000u)));
  r[0] = a;
  r[1] = b;
  return u;
}
__device__ __forceinline__ void split3(const f32x4& x, bf16x4 (&p)[3]) {
  f32x2 lo = {x[0], x[1]}, hi = {x[2], x[3]};
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    u32x2 w;
    if (q < 2) {
      w[0] = split3_pk(lo);
      w[1] = split3_pk(hi);
    } else {   // the last residual is exact in bf16
      w[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2));
      w[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2));
    }
    p[q] = __builtin_bit_cast(bf16x4, w);
  }
}

// Tile-selection experiments (scripts/one_shape.py, wgrad_sweep.py ...) read environment knobs — in the DEV build only
// (csrc/build.sh dev -> libdvae_dev.so).  The product library has ONE deterministic dispatch: the knobs are constants.
#ifdef DVAE_DEV
#include <cstdlib>
static inline int dvae_dev_knob(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
#else
#define dvae_dev_knob(name, dflt) (dflt)
#endif

extern int g_dvae_compute_mode;   // gemm.hip: process default of the contraction arithmetic (DVAE_MODE_*)
extern int g_dvae_deterministic;  // gemm.hip: dvae_set_deterministic

extern int g_dvae_last_hip_error;

#define DVAE_BN_ROWS_PER_CHUNK 64   // rows per BatchNorm partial-sum chunk (bn.hip; the conv epilogue writes the same layout)

static inline int dvae_check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_dvae_last_hip_error = (int)e;
    return DVAE_ELAUNCH;
  }
  return DVAE_OK;
}

// ---- profiling hooks (prof.cpp) ----
void dvae_prof_begin(int family, hipStream_t s, double flops, unsigned tag, double bytes);
void dvae_prof_end(int family, hipStream_t s);
extern int g_dvae_prof_family;

struct ProfScope {
  int fam;
  hipStream_t s;
  bool on;
  ProfScope(int family, hipStream_t st, double flops, unsigned tag = 0, double bytes = 0.0) : fam(family), s(st) {
    on = (g_dvae_prof_family == family);
    if (on) dvae_prof_begin(family, s, flops, tag, bytes);
  }
  ~ProfScope() {
    if (on) dvae_prof_end(fam, s);
  }
};

// four consecutive elements of an fp32 or bf16 tensor as f32x4 (element index 4*i4)
template <bool B16>
__device__ __forceinline__ f32x4 ld4(const void* p, int64_t i4) {
  if constexpr (B16) return __builtin_convertvector(reinterpret_cast<const bf16x4*>(p)[i4], f32x4);
  else return reinterpret_cast<const f32x4*>(p)[i4];
}
template <bool B16>
__device__ __forceinline__ void st4(void* p, int64_t i4, const f32x4& v) {
  if constexpr (B16) reinterpret_cast<bf16x4*>(p)[i4] = __builtin_convertvector(v, bf16x4);
  else reinterpret_cast<f32x4*>(p)[i4] = v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float act_apply(float u, int act) {
  if (act == DVAE_ACT_RELU) return u > 0.f ? u : 0.f;
  if (act == DVAE_ACT_TANH) return tanhf(u);
  return u;
}
// derivative of the activation expressed through its OUTPUT z
__device__ __forceinline__ float act_grad_from_out(float z, int act) {
  if (act == DVAE_ACT_RELU) return z > 0.f ? 1.f : 0.f;
  if (act == DVAE_ACT_TANH) return 1.f - z * z;
  return 1.f;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
// Gate non-linearities of the LSTM frame kernels: v_exp_f32 + v_rcp_f32 (each ~1 ulp) instead of the libm
// routines (tens of instructions with branches; at H = 64 they were 60 % of a frame).  Absolute error ~1e-7.
__device__ __forceinline__ float gate_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float gate_tanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * x) + 1.f); }
