// Residency census (dev tool): how many 256-thread workgroups with a given dynamic-LDS size does a CU hold?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* cnt, int* mx, int spins) {
  extern __shared__ double lds[];
  if (threadIdx.x == 0) {
    int now = atomicAdd(cnt, 1) + 1;
    atomicMax(mx, now);
    lds[0] = now;
  }
  for (int i = 0; i < spins; ++i) __builtin_amdgcn_s_sleep(100);
  __syncthreads();
  if (threadIdx.x == 0) atomicSub(cnt, 1);
}
int main() {
  int *cnt, *mx;
  hipMalloc(&cnt, 4); hipMalloc(&mx, 4);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("CUs %d, sharedMemPerMultiprocessor %zu, sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu\n", p.multiProcessorCount,
         p.sharedMemPerMultiprocessor, p.sharedMemPerBlock, (size_t)p.maxSharedMemoryPerMultiProcessor);
  for (int kb : {32, 36, 40, 44, 48, 50, 51, 52, 53, 54, 56, 64, 75, 80}) {
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
    hipMemset(cnt, 0, 4); hipMemset(mx, 0, 4);
    k<<<256 * 6, 256, kb * 1024>>>(cnt, mx, 40);
    hipDeviceSynchronize();
    int h; hipMemcpy(&h, mx, 4, hipMemcpyDeviceToHost);
    int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, 256, kb * 1024);
    printf("LDS %2d KB: max concurrent workgroups %4d = %.2f per CU (API says %d)\n", kb, h, h / 256.0, occ);
  }
  return 0;
}
