// Hardware probe for gfx950: verifies the lane<->element maps this library's
// kernels assume (MFMA fragments, ds_read_b64_tr_b16) and measures a streaming
// copy.  Test infrastructure only; not linked into libgdl_hip.so.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %s:%d\n",hipGetErrorString(e),__FILE__,__LINE__); exit(1);} }while(0)

static unsigned short f2bf(float f){ unsigned u; memcpy(&u,&f,4); unsigned r = u + 0x7fff + ((u>>16)&1); return (unsigned short)(r>>16);} 

__global__ void k_tr(short* out, int rowstride /*shorts*/){
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for(int i=threadIdx.x;i<4096;i+=64) lds[i]=(short)i;
  __syncthreads();
  int l = threadIdx.x, g = l>>4, i = l&15;
  int addr = g*1024 + (i>>2)*rowstride + (i&3)*4;   // lane i supplies row i>>2, cols (i&3)*4..+3 of a 4x16 block
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr));
  for(int j=0;j<4;j++) out[l*4+j]=v[j];
}

// D = A(16x32) * B(32x16): A[i][k] row-major bf16, Bt[j][k] (B transposed) bf16
__global__ void k_mfma_bf16(const unsigned short* A, const unsigned short* Bt, float* D){
  int l = threadIdx.x; int r = l&15, g = l>>4;
  bf16x8 a, b;
  for(int j=0;j<8;j++){ unsigned short ua=A[r*32+g*8+j], ub=Bt[r*32+g*8+j]; a[j]=__builtin_bit_cast(__bf16, ua); b[j]=__builtin_bit_cast(__bf16, ub);} 
  f32x4 acc={0,0,0,0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a,b,acc,0,0,0);
  for(int j=0;j<4;j++) D[(g*4+j)*16 + r] = acc[j];   // row=(lane>>4)*4+reg, col=lane&15
}
// f32 16x16x4: A[i][k] (16x4), Bt[j][k]
__global__ void k_mfma_f32(const float* A, const float* Bt, float* D){
  int l = threadIdx.x; int r = l&15, g = l>>4;
  f32x4 acc={0,0,0,0};
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r*4+g], Bt[r*4+g], acc,0,0,0);
  for(int j=0;j<4;j++) D[(g*4+j)*16 + r] = acc[j];
}
__global__ void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n){
  size_t i = blockIdx.x*(size_t)blockDim.x+threadIdx.x; size_t st=(size_t)gridDim.x*blockDim.x;
  for(;i<n;i+=st) b[i]=a[i];
}

int main(){
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop,0));
  printf("device: %s arch=%s CUs=%d clock=%d kHz mem=%.1f GB L2=%d\n",prop.name,prop.gcnArchName,prop.multiProcessorCount,prop.clockRate,prop.totalGlobalMem/1e9,prop.l2CacheSize);
  // --- tr read
  for(int rs : {16, 64, 72}){
    short* d; CK(hipMalloc(&d,256*2)); k_tr<<<1,64>>>(d, rs); CK(hipDeviceSynchronize());
    short h[256]; CK(hipMemcpy(h,d,512,hipMemcpyDeviceToHost));
    int ok=1; for(int l=0;l<64;l++) for(int j=0;j<4;j++){ int g=l>>4,i=l&15; int expect = g*1024 + j*rs + i; if(h[l*4+j]!=expect) ok=0; }
    printf("tr16_b64 rowstride=%d: %s\n", rs, ok?"MATCH (lane i gets column i of the 4x16 block, elem j = row j)":"MISMATCH");
    if(!ok){ for(int l=0;l<20;l++) printf("  lane %d: %d %d %d %d\n",l,h[l*4],h[l*4+1],h[l*4+2],h[l*4+3]); }
    CK(hipFree(d));
  }
  // --- mfma bf16
  {
    std::vector<unsigned short> A(16*32),B(16*32); std::vector<float> Af(16*32),Bf(16*32),D(256),R(256,0.f);
    for(int i=0;i<512;i++){ float a=(float)((i*37)%23-11)/8.f, b=(float)((i*53)%19-9)/4.f; A[i]=f2bf(a);B[i]=f2bf(b);Af[i]=a;Bf[i]=b; }
    for(int i=0;i<16;i++)for(int j=0;j<16;j++){float s=0;for(int k=0;k<32;k++)s+=Af[i*32+k]*Bf[j*32+k];R[i*16+j]=s;}
    unsigned short *dA,*dB; float* dD; CK(hipMalloc(&dA,1024));CK(hipMalloc(&dB,1024));CK(hipMalloc(&dD,1024));
    CK(hipMemcpy(dA,A.data(),1024,hipMemcpyHostToDevice));CK(hipMemcpy(dB,B.data(),1024,hipMemcpyHostToDevice));
    k_mfma_bf16<<<1,64>>>(dA,dB,dD); CK(hipMemcpy(D.data(),dD,1024,hipMemcpyDeviceToHost));
    float e=0; for(int i=0;i<256;i++) e=fmaxf(e,fabsf(D[i]-R[i])); printf("mfma_f32_16x16x32_bf16 layout check: max err %g %s\n",e,e<1e-3?"OK":"BAD");
  }
  {
    std::vector<float> A(64),B(64),D(256),R(256,0.f);
    for(int i=0;i<64;i++){A[i]=(float)((i*37)%23-11)/8.f;B[i]=(float)((i*53)%19-9)/4.f;}
    for(int i=0;i<16;i++)for(int j=0;j<16;j++){float s=0;for(int k=0;k<4;k++)s+=A[i*4+k]*B[j*4+k];R[i*16+j]=s;}
    float *dA,*dB,*dD; CK(hipMalloc(&dA,256));CK(hipMalloc(&dB,256));CK(hipMalloc(&dD,1024));
    CK(hipMemcpy(dA,A.data(),256,hipMemcpyHostToDevice));CK(hipMemcpy(dB,B.data(),256,hipMemcpyHostToDevice));
    k_mfma_f32<<<1,64>>>(dA,dB,dD); CK(hipMemcpy(D.data(),dD,1024,hipMemcpyDeviceToHost));
    float e=0; for(int i=0;i<256;i++) e=fmaxf(e,fabsf(D[i]-R[i])); printf("mfma_f32_16x16x4f32 layout check: max err %g %s\n",e,e<1e-5?"OK":"BAD");
  }
  // --- streaming copy
  {
    size_t bytes = (size_t)2<<30; float4 *a,*b; CK(hipMalloc(&a,bytes));CK(hipMalloc(&b,bytes)); CK(hipMemset(a,1,bytes));
    hipEvent_t e0,e1; CK(hipEventCreate(&e0));CK(hipEventCreate(&e1));
    for(int w=0;w<2;w++) k_copy<<<2048,256>>>(a,b,bytes/16);
    CK(hipEventRecord(e0)); for(int it=0;it<10;it++) k_copy<<<2048,256>>>(a,b,bytes/16); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms,e0,e1)); printf("float4 copy 2GiB: %.3f ms/iter -> %.2f TB/s (r+w)\n",ms/10, 2.0*bytes/(ms/10*1e-3)/1e12);
  }
  return 0;
}
