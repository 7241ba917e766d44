// Issue cost of the vector instructions the path kernels are made of, measured on the device: each
// kernel runs a long unrolled stream of one instruction (eight independent chains per lane) on every
// SIMD at 8 waves per SIMD, and the host reports cycles per wave-instruction per SIMD at the clock the
// run held. Build: hipcc --offload-arch=gfx950 -O2 -o valu_rates tools/valu_rates.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHK(x)                                                                  \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      std::printf("%s: %s\n", #x, hipGetErrorString(e_));                       \
      return 1;                                                                 \
    }                                                                           \
  } while (0)

#define ITERS 4096
#define CHAINS 8

#define KERNEL(NAME, TYPE, INIT, ASM, CONSTRAINT)                                            \
  __global__ __launch_bounds__(256) void NAME(TYPE* out, TYPE seed) {                        \
    TYPE a[CHAINS];                                                                          \
    for (int i = 0; i < CHAINS; ++i) a[i] = INIT;                                            \
    TYPE b = seed;                                                                           \
    for (int it = 0; it < ITERS; ++it) {                                                     \
      _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(ASM : "+" CONSTRAINT(a[i]) : CONSTRAINT(b)); \
    }                                                                                        \
    TYPE s = a[0];                                                                           \
    for (int i = 1; i < CHAINS; ++i) s += a[i];                                              \
    if (s == (TYPE)12345) out[threadIdx.x] = s;                                              \
  }

KERNEL(k_add_u32, unsigned, (unsigned)(threadIdx.x + i), "v_add_u32 %0, %0, %1", "v")
KERNEL(k_mul_lo_u32, unsigned, (unsigned)(threadIdx.x + i), "v_mul_lo_u32 %0, %0, %1", "v")
KERNEL(k_mul_hi_u32, unsigned, (unsigned)(threadIdx.x + i), "v_mul_hi_u32 %0, %0, %1", "v")
KERNEL(k_mul_u32_u24, unsigned, (unsigned)(threadIdx.x + i), "v_mul_u32_u24 %0, %0, %1", "v")
KERNEL(k_mad_u32_u24, unsigned, (unsigned)(threadIdx.x + i), "v_mad_u32_u24 %0, %0, %1, %1", "v")
KERNEL(k_xor_b32, unsigned, (unsigned)(threadIdx.x + i), "v_xor_b32 %0, %0, %1", "v")
KERNEL(k_lshrrev_b64, unsigned long long, (unsigned long long)(threadIdx.x + i), "v_lshrrev_b64 %0, 3, %0", "v")
KERNEL(k_fma_f32, float, (float)(threadIdx.x + i), "v_fma_f32 %0, %0, %1, %1", "v")
KERNEL(k_pk_fma_f32, double, (double)(threadIdx.x + i), "v_pk_fma_f32 %0, %0, %1, %1", "v")
KERNEL(k_add_f64, double, (double)(threadIdx.x + i), "v_add_f64 %0, %0, %1", "v")
KERNEL(k_mul_f64, double, (double)(threadIdx.x + i), "v_mul_f64 %0, %0, %1", "v")
KERNEL(k_fma_f64, double, (double)(threadIdx.x + i), "v_fma_f64 %0, %0, %1, %1", "v")
KERNEL(k_rcp_f64, double, (double)(threadIdx.x + i + 1), "v_rcp_f64 %0, %0", "v")
KERNEL(k_rsq_f64, double, (double)(threadIdx.x + i + 1), "v_rsq_f64 %0, %0", "v")
KERNEL(k_div_fixup_f64, double, (double)(threadIdx.x + i + 1), "v_div_fixup_f64 %0, %0, %1, %1", "v")
KERNEL(k_div_scale_f64, double, (double)(threadIdx.x + i + 1), "v_div_scale_f64 %0, vcc, %0, %1, %0", "v")
KERNEL(k_cndmask_b32, unsigned, (unsigned)(threadIdx.x + i), "v_cndmask_b32 %0, %0, %1, vcc", "v")
KERNEL(k_min_f32, float, (float)(threadIdx.x + i), "v_min_f32 %0, %0, %1", "v")
KERNEL(k_ldexp_f64, double, (double)(threadIdx.x + i + 1), "v_ldexp_f64 %0, %0, 1", "v")

template <typename T>
static int run(const char* name, void (*k)(T*, T), T seed, int cus, double* base) {
  T* out = nullptr;
  CHK(hipMalloc(&out, 256 * sizeof(T)));
  const int blocks = cus * 8;  // 8 blocks of 4 waves per CU = 8 waves per SIMD
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, seed);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, seed);
  CHK(hipEventRecord(e1));
  CHK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  // wave-instructions per SIMD: 8 waves x ITERS x CHAINS, three launches
  const double insts = 3.0 * 8.0 * ITERS * CHAINS;
  const double ns_per = ms * 1e6 / insts;
  if (*base == 0.0) *base = ns_per;
  std::printf("%-18s %7.3f ns per wave-instruction per SIMD  = %5.2f x v_add_u32  (~%.1f cycles at 2.4 GHz)\n", name, ns_per, ns_per / *base,
              ns_per * 2.4);
  CHK(hipFree(out));
  return 0;
}

int main() {
  int cus = 256;
  CHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  double base = 0.0;
#define RUN(NAME, TYPE, SEED) \
  if (run<TYPE>(#NAME, NAME, SEED, cus, &base)) return 1;
  RUN(k_add_u32, unsigned, 3u)
  RUN(k_xor_b32, unsigned, 3u)
  RUN(k_cndmask_b32, unsigned, 3u)
  RUN(k_mul_u32_u24, unsigned, 3u)
  RUN(k_mad_u32_u24, unsigned, 3u)
  RUN(k_mul_lo_u32, unsigned, 3u)
  RUN(k_mul_hi_u32, unsigned, 3u)
  RUN(k_lshrrev_b64, unsigned long long, 3ull)
  RUN(k_fma_f32, float, 1.0001f)
  RUN(k_min_f32, float, 1.0001f)
  RUN(k_pk_fma_f32, double, 1.0001)
  RUN(k_add_f64, double, 1.0001)
  RUN(k_mul_f64, double, 1.0001)
  RUN(k_fma_f64, double, 1.0001)
  RUN(k_ldexp_f64, double, 1.0001)
  RUN(k_div_scale_f64, double, 1.0001)
  RUN(k_div_fixup_f64, double, 1.0001)
  RUN(k_rcp_f64, double, 1.0001)
  RUN(k_rsq_f64, double, 1.0001)
  return 0;
}
