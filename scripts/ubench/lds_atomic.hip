// micro-benchmark: LDS atomic / read / write throughput with scattered addresses (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
constexpr int N = 8192;
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint32_t* idx, unsigned long long* out, int iters) {
  __shared__ unsigned long long acc[N];
  for (int i = threadIdx.x; i < N; i += 256) acc[i] = 0;
  __syncthreads();
  uint32_t d[8];
  for (int k2 = 0; k2 < 8; ++k2) d[k2] = idx[(blockIdx.x * 8 + k2) * 256 + threadIdx.x] & (N - 1);
  unsigned long long s = 0;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) {
      if (MODE == 0) atomicAdd(&acc[d[k2]], 3ull);
      if (MODE == 1) s += atomicExch(&acc[d[k2]], 1ull);
      if (MODE == 2) atomicAdd((unsigned int*)&acc[d[k2]], 3u);
      if (MODE == 3) s += acc[d[k2]];
      if (MODE == 4) acc[d[k2]] = it;
      if (MODE == 5) atomicAdd((float*)&acc[d[k2]], 1.0f);
      if (MODE == 6) s += atomicAdd(&acc[d[k2]], 3ull);
      d[k2] = (d[k2] * 5 + 1) & (N - 1);
    }
  }
  __syncthreads();
  long long t1 = clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = (unsigned long long)(t1 - t0);
  if (s == 12345) out[0] = acc[threadIdx.x];
}
int main() {
  const int blocks = 512, iters = 200;
  std::vector<uint32_t> h(blocks * 8 * 256);
  uint32_t x = 12345;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x >> 8; }
  uint32_t* di; unsigned long long* dout;
  CK(hipMalloc(&di, h.size() * 4)); CK(hipMalloc(&dout, blocks * 8));
  CK(hipMemcpy(di, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  const char* names[] = {"ds_add_u64", "ds_wrxchg_rtn_b64", "ds_add_u32", "ds_read_b64", "ds_write_b64", "ds_add_f32", "ds_add_rtn_u64"};
  for (int mode = 0; mode < 7; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      switch (mode) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, di, dout, iters); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, di, dout, iters); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, di, dout, iters); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, di, dout, iters); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, di, dout, iters); break;
        case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, di, dout, iters); break;
        case 6: hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(256), 0, 0, di, dout, iters); break;
      }
      CK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> o(blocks);
    CK(hipMemcpy(o.data(), dout, blocks * 8, hipMemcpyDeviceToHost));
    double m = 0; for (auto v : o) m += v; m /= blocks;
    // per block: iters*8 wave-instructions per wave, 4 waves; 2 blocks share a CU
    printf("%-20s %.1f cycles per wave-instruction per wave (block-level: %.1f cycles per 64 lanes-op)\n", names[mode], m / (iters * 8), m / (iters * 8 * 4));
  }
  return 0;
}
