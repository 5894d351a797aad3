// How many 256-thread workgroups of a kernel with S bytes of dynamic LDS does a CU take?  (hipOccupancyMaxActiveBlocksPerMultiprocessor over a range of S;
// the kernel itself uses few registers, so LDS is the only limit.)   hipcc --offload-arch=gfx950 -o /tmp/lds_occ tools/ab/lds_occupancy.hip && /tmp/lds_occ
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(float* o) {
    extern __shared__ float sm[];
    sm[threadIdx.x] = o[threadIdx.x];
    __syncthreads();
    o[threadIdx.x] = sm[255 - threadIdx.x];
}
int main() {
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerBlock %zu maxSharedMemoryPerMultiProcessor %zu\n", p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor);
    for (int s : {42240, 52000, 54000, 54613, 56000, 67584, 71616, 76000, 78000, 79104, 79872, 80000, 80640, 81920, 82000, 84000, 100000, 163840}) {
        int n = -1;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k, 256, s);
        printf("%7d bytes: %d workgroups per CU (%s)\n", s, n, hipGetErrorString(e));
    }
    return 0;
}
