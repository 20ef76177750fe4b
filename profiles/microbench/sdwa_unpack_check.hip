#include <hip/hip_runtime.h>
__global__ void k(const uint32_t* in, uint32_t* out) {
  uint32_t x = in[threadIdx.x], lo, hi;
  uint32_t two = 2;
  asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(lo) : "s"(two), "v"(x));
  asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(hi) : "s"(two), "v"(x));
  out[2*threadIdx.x] = lo; out[2*threadIdx.x+1] = hi;
}
int main(){ 
  uint32_t h[64], *d, *o, r[128];
  for(int i=0;i<64;i++) h[i]= (0x7000u+i) | ((0x6fffu-i)<<16);
  hipMalloc(&d,256); hipMalloc(&o,512); hipMemcpy(d,h,256,hipMemcpyHostToDevice);
  k<<<1,64>>>(d,o); hipMemcpy(r,o,512,hipMemcpyDeviceToHost);
  int bad=0; for(int i=0;i<64;i++){ if(r[2*i]!=((h[i]&0xffff)<<2) || r[2*i+1]!=((h[i]>>16)<<2)) bad++; }
  printf("bad=%d %x %x\n",bad,r[0],r[1]); return bad; }
