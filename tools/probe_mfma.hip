// Hardware layout probe (dev tool, not product): verifies the MFMA fragment
// layouts and ds_read_b64_tr_b16 semantics this repo's kernels assume.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <math.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
static inline uint16_t f2bf(float f){ uint32_t u; memcpy(&u,&f,4); return (uint16_t)(u>>16);}  // exact for small ints

// A [32][16] row-major bf16, B [16][32] row-major bf16 -> D [32][32]
__global__ void k_mfma32_bf16(const uint16_t* A, const uint16_t* B, float* D) {
  int l = threadIdx.x; bf16x8 a, b;
  for (int j=0;j<8;++j){ int k=(l>>5)*8+j; a[j]=A[(l&31)*16+k]; b[j]=B[k*32+(l&31)]; }
  f32x16 c = {0}; c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a,b,c,0,0,0);
  for (int r=0;r<16;++r){ int row=(r&3)+8*(r>>2)+4*(l>>5), col=l&31; D[row*32+col]=c[r]; }
}
__global__ void k_mfma16_bf16(const uint16_t* A, const uint16_t* B, float* D) { // A[16][32], B[32][16]
  int l = threadIdx.x; bf16x8 a, b;
  for (int j=0;j<8;++j){ int k=(l>>4)*8+j; a[j]=A[(l&15)*32+k]; b[j]=B[k*16+(l&15)]; }
  f32x4 c = {0}; c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a,b,c,0,0,0);
  for (int r=0;r<4;++r){ int row=(l>>4)*4+r, col=l&15; D[row*16+col]=c[r]; }
}
__global__ void k_mfma32_f32(const float* A, const float* B, float* D) { // A[32][2], B[2][32]
  int l = threadIdx.x; float a=A[(l&31)*2+(l>>5)], b=B[(l>>5)*32+(l&31)];
  f32x16 c={0}; c=__builtin_amdgcn_mfma_f32_32x32x2f32(a,b,c,0,0,0);
  for (int r=0;r<16;++r){ int row=(r&3)+8*(r>>2)+4*(l>>5), col=l&31; D[row*32+col]=c[r]; }
}
// tr read: LDS holds u16 index values idx = row*64+col in a [16 k-rows][64 cols] row-major image (128 B rows).
// lane l gives address of row (kbase + (l&15)/4 ... ) per hypothesis H: within a 16-lane group,
// lane i points at row (i>>2), cols cbase + (i&3)*4 ; result lane i gets column cbase+i, rows 0..3.
__global__ void k_tr(uint16_t* out /*[64][4]*/) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[16*64];
  int l=threadIdx.x;
  for (int t=l;t<16*64;t+=64) lds[t]=(uint16_t)t;
  __syncthreads();
  int grp=l>>4, i=l&15;
  int row=grp*4+(i>>2), col=(i&3)*4;       // group g reads k-rows 4g..4g+3, cols 0..15
  uint32_t addr=(uint32_t)(uintptr_t)(&lds[row*64+col]);
  bf16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
  for (int j=0;j<4;++j) out[l*4+j]=(uint16_t)v[j];
}
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
int main(){
  {
    std::vector<uint16_t> A(32*16),B(16*32); std::vector<float> Af(32*16),Bf(16*32),D(32*32),R(32*32,0.f);
    for(int i=0;i<32*16;++i){Af[i]=(float)((i*7+3)%11-5);A[i]=f2bf(Af[i]);} for(int i=0;i<16*32;++i){Bf[i]=(float)((i*5+1)%13-6);B[i]=f2bf(Bf[i]);}
    for(int i=0;i<32;++i)for(int j=0;j<32;++j){float s=0;for(int k=0;k<16;++k)s+=Af[i*16+k]*Bf[k*32+j];R[i*32+j]=s;}
    uint16_t *dA,*dB; float* dD; CK(hipMalloc(&dA,A.size()*2));CK(hipMalloc(&dB,B.size()*2));CK(hipMalloc(&dD,D.size()*4));
    CK(hipMemcpy(dA,A.data(),A.size()*2,hipMemcpyHostToDevice));CK(hipMemcpy(dB,B.data(),B.size()*2,hipMemcpyHostToDevice));
    k_mfma32_bf16<<<1,64>>>(dA,dB,dD); CK(hipMemcpy(D.data(),dD,D.size()*4,hipMemcpyDeviceToHost));
    int bad=0; for(size_t i=0;i<D.size();++i) bad+= D[i]!=R[i]; printf("mfma_f32_32x32x16_bf16 layout: %s (%d bad)\n",bad?"MISMATCH":"OK",bad);
  }
  {
    std::vector<uint16_t> A(16*32),B(32*16); std::vector<float> Af(16*32),Bf(32*16),D(16*16),R(16*16,0.f);
    for(int i=0;i<16*32;++i){Af[i]=(float)((i*7+3)%11-5);A[i]=f2bf(Af[i]);} for(int i=0;i<32*16;++i){Bf[i]=(float)((i*5+1)%13-6);B[i]=f2bf(Bf[i]);}
    for(int i=0;i<16;++i)for(int j=0;j<16;++j){float s=0;for(int k=0;k<32;++k)s+=Af[i*32+k]*Bf[k*16+j];R[i*16+j]=s;}
    uint16_t *dA,*dB; float* dD; CK(hipMalloc(&dA,A.size()*2));CK(hipMalloc(&dB,B.size()*2));CK(hipMalloc(&dD,D.size()*4));
    CK(hipMemcpy(dA,A.data(),A.size()*2,hipMemcpyHostToDevice));CK(hipMemcpy(dB,B.data(),B.size()*2,hipMemcpyHostToDevice));
    k_mfma16_bf16<<<1,64>>>(dA,dB,dD); CK(hipMemcpy(D.data(),dD,D.size()*4,hipMemcpyDeviceToHost));
    int bad=0; for(size_t i=0;i<D.size();++i) bad+= D[i]!=R[i]; printf("mfma_f32_16x16x32_bf16 layout: %s (%d bad)\n",bad?"MISMATCH":"OK",bad);
  }
  {
    std::vector<float> A(32*2),B(2*32),D(32*32),R(32*32,0.f);
    for(int i=0;i<64;++i){A[i]=(float)((i*7+3)%11-5)+0.25f;B[i]=(float)((i*5+1)%13-6)-0.5f;}
    for(int i=0;i<32;++i)for(int j=0;j<32;++j){float s=0;for(int k=0;k<2;++k)s=fmaf(A[i*2+k],B[k*32+j],s);R[i*32+j]=s;}
    float *dA,*dB,*dD; CK(hipMalloc(&dA,256));CK(hipMalloc(&dB,256));CK(hipMalloc(&dD,4096));
    CK(hipMemcpy(dA,A.data(),256,hipMemcpyHostToDevice));CK(hipMemcpy(dB,B.data(),256,hipMemcpyHostToDevice));
    k_mfma32_f32<<<1,64>>>(dA,dB,dD); CK(hipMemcpy(D.data(),dD,4096,hipMemcpyDeviceToHost));
    int bad=0; for(size_t i=0;i<D.size();++i) bad+= D[i]!=R[i]; printf("mfma_f32_32x32x2f32 layout: %s (%d bad)\n",bad?"MISMATCH":"OK",bad);
  }
  {
    uint16_t* dO; CK(hipMalloc(&dO,64*4*2)); std::vector<uint16_t> O(256);
    k_tr<<<1,64>>>(dO); CK(hipMemcpy(O.data(),dO,512,hipMemcpyDeviceToHost));
    int bad=0;
    for(int l=0;l<64;++l){ int grp=l>>4,i=l&15; for(int j=0;j<4;++j){ int exp=(grp*4+j)*64+i; bad+= O[l*4+j]!=exp; } }
    printf("ds_read_b64_tr_b16 hypothesis (lane i of 16-group gets col i, rows 0..3): %s (%d bad)\n",bad?"MISMATCH":"OK",bad);
    if(bad){ for(int l=0;l<64;++l){ printf("lane %2d:",l); for(int j=0;j<4;++j) printf(" (r%d,c%d)",O[l*4+j]/64,O[l*4+j]%64); printf("\n"); } }
  }
  return 0;
}
