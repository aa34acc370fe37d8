// Probe 2: global_load_lds_dwordx4 and buffer_load ... lds with larger num_records.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const unsigned* p, int nbytes, unsigned* out, int variant) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (int i = threadIdx.x; i < 2048; i += 128) ((unsigned*)smem)[i] = 0xdeadbeefu;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int chunk = (wave * 64 + lane) ^ 5;
  if (variant == 0) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + chunk * 4),
                                     (__attribute__((address_space(3))) void*)(smem + wave * 1024), 16, 0, 0);
  } else {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + wave * 1024), 16, (unsigned)(chunk * 16), 0, 0, 0);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 128) out[i] = ((unsigned*)smem)[i];
}
int main() {
  const int n = 128 * 4;
  std::vector<unsigned> h(4 * n); for (int i = 0; i < 4 * n; ++i) h[i] = i;
  unsigned *d, *o; hipMalloc(&d, 4 * n * 4); hipMalloc(&o, 2048 * 4);
  hipMemcpy(d, h.data(), 4 * n * 4, hipMemcpyHostToDevice);
  for (int variant = 0; variant < 3; ++variant) {
    int nbytes = variant == 2 ? 4 * n * 4 : n * 4;
    k<<<1, 128, 8192>>>(d, nbytes, o, variant == 0 ? 0 : 1);
    std::vector<unsigned> r(2048); hipMemcpy(r.data(), o, 2048 * 4, hipMemcpyDeviceToHost);
    int bad = 0, badlo = 0;
    for (int w = 0; w < 2; ++w) for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
      unsigned got = r[(w * 64 + l) * 4 + e], exp = (unsigned)((((w * 64 + l) ^ 5) * 4) + e);
      if (got != exp) { ++bad; if (l < 32) ++badlo; }
    }
    printf("variant %d (%s, num_records %d): %d bad (%d in lanes<32); sample w0 l40: %08x %08x\n", variant,
           variant == 0 ? "global_load_lds" : "buffer_load lds", nbytes, bad, badlo, r[40 * 4], r[40 * 4 + 1]);
  }
  return 0;
}
