// Does the range check of a raw buffer load on gfx950 include the SCALAR offset?  K1's stream variant, K6 and K4c
// prefetch weight fragment groups a few KB past the end of their buffer resource and rely on the answer only for
// tidiness (the allocations are padded so that an unchecked over-run still reads mapped memory).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_buffer_range.hip -o tools/ubench_buffer_range
#include <hip/hip_runtime.h>
#include <stdio.h>
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
__global__ void k(const unsigned *buf, unsigned *out, int soff) {
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(buf), 0, 1024, 0x00020000);
  const u32x4 inside = __builtin_amdgcn_raw_buffer_load_b128(rsrc, threadIdx.x * 16u, 0, 0);
  const u32x4 s_past = __builtin_amdgcn_raw_buffer_load_b128(rsrc, threadIdx.x * 16u, soff, 0);        // scalar offset past the end
  const u32x4 v_past = __builtin_amdgcn_raw_buffer_load_b128(rsrc, threadIdx.x * 16u + soff, 0, 0);   // vector offset past the end
  out[threadIdx.x * 3 + 0] = inside[0];
  out[threadIdx.x * 3 + 1] = s_past[0];
  out[threadIdx.x * 3 + 2] = v_past[0];
}
int main() {
  unsigned *buf, *out, h[64 * 3];
  hipMalloc(&buf, 1 << 20); hipMalloc(&out, sizeof(h));
  hipMemset(buf, 0x5a, 1 << 20);
  for (int soff : {1024, 2048, 65536}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, buf, out, soff);
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("num_records 1024, offset %6d: inside %08x | scalar offset past the end: lane0 %08x lane63 %08x | vector offset past the end: lane0 %08x lane63 %08x\n",
           soff, h[0], h[1], h[63 * 3 + 1], h[2], h[63 * 3 + 2]);
  }
  return 0;
}
