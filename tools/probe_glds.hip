// Probe: buffer_load_dwordx4 ... lds (LDS-DMA) layout and out-of-range behaviour on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const void* p, int nbytes, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (int i = threadIdx.x; i < 2048 / 4 * 4; i += 256) ((unsigned*)smem)[i] = 0xdeadbeefu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, 0x00020000);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // lane l fetches chunk (l ^ 5) of the source (permuted source, linear destination); odd lanes of wave 1 out of range
  unsigned off = (unsigned)(((wave * 64 + lane) ^ 5) * 16);
  if (wave == 1 && (lane & 1)) off = 0xfffffff0u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + wave * 1024), 16, off, 0, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 256) out[i] = ((unsigned*)smem)[i];
}
int main() {
  const int n = 128 * 4;  // 128 chunks of 16 B
  std::vector<unsigned> h(n); for (int i = 0; i < n; ++i) h[i] = i;
  unsigned *d, *o; hipMalloc(&d, n * 4); hipMalloc(&o, 2048 * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  k<<<1, 128, 8192>>>(d, n * 4, o);
  std::vector<unsigned> r(2048); hipMemcpy(r.data(), o, 2048 * 4, hipMemcpyDeviceToHost);
  int bad = 0, zero_ok = 0;
  for (int w = 0; w < 2; ++w) for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
    unsigned got = r[(w * 64 + l) * 4 + e];
    bool oob = (w == 1 && (l & 1));
    unsigned exp = oob ? 0u : (unsigned)((((w * 64 + l) ^ 5) * 4) + e);
    if (got != exp) { if (e == 0) printf("w%d l%d got %08x exp %08x | ", w, l, got, exp); ++bad; }
    else if (oob) ++zero_ok;
  }
  printf("LDS-DMA layout (dest = M0 base + lane*16, per-lane source): %s (%d bad); OOB lanes wrote zeros: %d/128\n", bad ? "MISMATCH" : "OK", bad, zero_ok);
  printf("untouched LDS after the two 1-KiB pieces: %08x (expect deadbeef)\n", r[512]);
  return 0;
}
