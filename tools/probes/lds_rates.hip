// Probe (not part of the library): LDS read issue rates on gfx950 per wave64 instruction, one workgroup per CU, 4 / 8 / 12 waves:
// ds_read_b128, ds_read_b64, ds_read_b64_tr_b16 (the transposing read the attention kernels take V^T fragments with), conflict-free addresses.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_rates tools/probes/lds_rates.hip && /tmp/lds_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void lds_kernel(long long* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = (float)i;
  __syncthreads();
  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  // b128: lane * 16 (a 1 KiB line per instruction); b64 / tr: lane * 8
  const unsigned a128 = base + (wave & 3) * 8192 + lane * 16, a64 = base + (wave & 3) * 8192 + lane * 8;
  f32x4 r0 = {0}, r1 = {0}, r2 = {0}, r3 = {0};
  f32x2 q0 = {0}, q1 = {0}, q2 = {0}, q3 = {0}, q4 = {0}, q5 = {0}, q6 = {0}, q7 = {0};
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if constexpr (MODE == 0) {
      asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n s_waitcnt lgkmcnt(0)"
                   : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a128) : "memory");
    } else if constexpr (MODE == 1) {
      asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:512\n ds_read_b64 %2, %8 offset:1024\n ds_read_b64 %3, %8 offset:1536\n"
                   "ds_read_b64 %4, %8 offset:2048\n ds_read_b64 %5, %8 offset:2560\n ds_read_b64 %6, %8 offset:3072\n ds_read_b64 %7, %8 offset:3584\n s_waitcnt lgkmcnt(0)"
                   : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(q4), "=v"(q5), "=v"(q6), "=v"(q7) : "v"(a64) : "memory");
    } else {
      asm volatile("ds_read_b64_tr_b16 %0, %8\n ds_read_b64_tr_b16 %1, %8 offset:512\n ds_read_b64_tr_b16 %2, %8 offset:1024\n ds_read_b64_tr_b16 %3, %8 offset:1536\n"
                   "ds_read_b64_tr_b16 %4, %8 offset:2048\n ds_read_b64_tr_b16 %5, %8 offset:2560\n ds_read_b64_tr_b16 %6, %8 offset:3072\n ds_read_b64_tr_b16 %7, %8 offset:3584\n s_waitcnt lgkmcnt(0)"
                   : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(q4), "=v"(q5), "=v"(q6), "=v"(q7) : "v"(a64) : "memory");
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float sink = r0[0] + r1[1] + r2[2] + r3[3] + q0[0] + q1[1] + q2[0] + q3[1] + q4[0] + q5[1] + q6[0] + q7[1];
  if (sink == 123.456f) out[4096] = 1;
  if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
}

template <int MODE>
int run(const char* what, int per_iter, int bytes_per_instr, long long* dev, int waves) {
  const int iters = 4000;
  for (int rep = 0; rep < 2; ++rep) {
    lds_kernel<MODE><<<256, waves * 64, 65536>>>(dev, iters);
    CHECK(hipDeviceSynchronize());
  }
  std::vector<long long> h(16);
  CHECK(hipMemcpy(h.data(), dev, 16 * 8, hipMemcpyDeviceToHost));
  long long mx = 0;
  for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
  const double per = (double)mx / iters / per_iter;
  printf("%-22s waves %2d: %7.2f cycles per wave instruction = %6.2f cycles per CU instruction = %6.1f B / cycle / CU\n", what, waves, per, per / waves,
         bytes_per_instr * waves / per);
  return 0;
}

int main() {
  long long* dev;
  CHECK(hipMalloc(&dev, 8 * 8192));
  CHECK(hipMemset(dev, 0, 8 * 8192));
  for (int waves : {4, 8, 12, 16}) {
    run<0>("ds_read_b128", 4, 1024, dev, waves);
    run<1>("ds_read_b64", 8, 512, dev, waves);
    run<2>("ds_read_b64_tr_b16", 8, 512, dev, waves);
  }
  return 0;
}
