// Probe 3: buffer_load_dwordx4 ... lds as the conv kernel would use it: 4 waves x 4 pieces, random OOB lanes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const unsigned* p, int nbytes, unsigned* out, unsigned oobmask_seed) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (int i = threadIdx.x; i < 4096; i += 256) ((unsigned*)smem)[i] = 0xdeadbeefu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, 0x00020000);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int piece = wave * 4 + g;                 // 16 pieces of 1 KiB
    const int chunk = (piece * 64 + lane) ^ 3;      // permuted source
    const bool oob = ((oobmask_seed * 2654435761u) >> ((lane + piece * 7) & 31)) & 1;
    const unsigned off = oob ? 0xfffffff0u : (unsigned)(chunk * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + piece * 1024), 16, off, 0, 0, 0);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 4096; i += 256) out[i] = ((unsigned*)smem)[i];
}
int main() {
  const int nchunks = 1024, n = nchunks * 4;
  std::vector<unsigned> h(n); for (int i = 0; i < n; ++i) h[i] = i + 1;
  unsigned *d, *o; hipMalloc(&d, n * 4); hipMalloc(&o, 4096 * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  for (unsigned seed : {0u, 1u, 7u, 12345u}) {
    k<<<1, 256, 16384>>>(d, n * 4, o, seed);
    std::vector<unsigned> r(4096); hipMemcpy(r.data(), o, 4096 * 4, hipMemcpyDeviceToHost);
    int bad = 0, noob = 0;
    for (int piece = 0; piece < 16; ++piece) for (int l = 0; l < 64; ++l) {
      const bool oob = ((seed * 2654435761u) >> ((l + piece * 7) & 31)) & 1;
      noob += oob;
      for (int e = 0; e < 4; ++e) {
        unsigned got = r[(piece * 64 + l) * 4 + e];
        unsigned exp = oob ? 0u : (unsigned)((((piece * 64 + l) ^ 3) * 4) + e + 1);
        if (got != exp) { if (bad < 6) printf("  seed %u piece %d lane %d e%d got %08x exp %08x\n", seed, piece, l, e, got, exp); ++bad; }
      }
    }
    printf("seed %u: %d OOB lanes, %d bad words\n", seed, noob, bad);
  }
  return 0;
}
