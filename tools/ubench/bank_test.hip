// bank_test.hip -- does the VGPR bank pattern of the three sources of v_sad_u32 change its rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#define BODY(ASM8) for (int i = 0; i < 4096; ++i) { asm volatile(ASM8 ::: "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25"); }
__global__ void __launch_bounds__(256) k_noconf(unsigned* o) {
  // sources: v4n+1, v4m+3, acc v4k+2 style: three different banks (reg % 4)
  asm volatile("v_mov_b32 v5, 7\n v_mov_b32 v7, 9\n v_mov_b32 v9, 3\n v_mov_b32 v3, 1" ::: "v5","v7","v9","v3");
  BODY("v_sad_u32 v10, v5, v7, v10\n v_sad_u32 v14, v9, v3, v14\n v_sad_u32 v18, v5, v7, v18\n v_sad_u32 v22, v9, v3, v22\n"
       "v_sad_u32 v12, v5, v3, v12\n v_sad_u32 v16, v9, v7, v16\n v_sad_u32 v20, v5, v3, v20\n v_sad_u32 v24, v9, v7, v24")
  unsigned r; asm volatile("v_mov_b32 %0, v10" : "=v"(r)); o[threadIdx.x] = r;
}
__global__ void __launch_bounds__(256) k_conf2(unsigned* o) {
  // two sources on the same bank (v4, v8 -> bank 0), acc elsewhere
  asm volatile("v_mov_b32 v4, 7\n v_mov_b32 v8, 9" ::: "v4","v8");
  BODY("v_sad_u32 v10, v4, v8, v10\n v_sad_u32 v14, v4, v8, v14\n v_sad_u32 v18, v4, v8, v18\n v_sad_u32 v22, v4, v8, v22\n"
       "v_sad_u32 v13, v4, v8, v13\n v_sad_u32 v17, v4, v8, v17\n v_sad_u32 v21, v4, v8, v21\n v_sad_u32 v25, v4, v8, v25")
  unsigned r; asm volatile("v_mov_b32 %0, v10" : "=v"(r)); o[threadIdx.x] = r;
}
__global__ void __launch_bounds__(256) k_conf3(unsigned* o) {
  // all three sources on bank 0
  asm volatile("v_mov_b32 v4, 7\n v_mov_b32 v8, 9" ::: "v4","v8");
  BODY("v_sad_u32 v12, v4, v8, v12\n v_sad_u32 v16, v4, v8, v16\n v_sad_u32 v20, v4, v8, v20\n v_sad_u32 v24, v4, v8, v24\n"
       "v_sad_u32 v12, v4, v8, v12\n v_sad_u32 v16, v4, v8, v16\n v_sad_u32 v20, v4, v8, v20\n v_sad_u32 v24, v4, v8, v24")
  unsigned r; asm volatile("v_mov_b32 %0, v12" : "=v"(r)); o[threadIdx.x] = r;
}
int main() {
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  unsigned* d; (void)hipMalloc(&d, 4096);
  void (*ks[3])(unsigned*) = {k_noconf, k_conf2, k_conf3};
  const char* nm[3] = {"3 banks", "2 srcs same bank", "3 srcs same bank"};
  for (int k = 0; k < 3; ++k) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    ks[k]<<<p.multiProcessorCount * 16, 256>>>(d); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); ks[k]<<<p.multiProcessorCount * 16, 256>>>(d); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double inst = 4096.0 * 8 * 16;       // per SIMD: 16 blocks x 4 waves / 4 SIMDs = 16 waves of 4096 x 8 instructions (rounds 1-3 printed a quarter of this: a stray x 4)
    printf("%-20s %.3f ms  %.2f cycles per wave-instruction per SIMD\n", nm[k], ms, ms * 1e-3 * 2.4e9 / inst);
  }
  return 0;
}
