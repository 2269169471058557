// Every non-negative float (and +inf, and the NaNs as a class): the device's correctly rounded sqrtf(x) against (float)sqrt((double)x), which is what
// upstream's Point3::norm() prescribes (include/corto/point.h:111) and what k_normal.hip's norm3() computed until round 4 - bit for bit.  A double's
// 53-bit square root rounded to 24 bits is the correctly rounded single result (the double rounding is innocuous for sqrt: 53 >= 2*24 + 2), so the
// two agree IF the f32 routine is correctly rounded; this program is that if.  Built and run by tests/test_gpu_parity.py with the library's own flags.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

__global__ void k_check(unsigned long long *bad, uint32_t *first_bad) {
	const uint32_t bits = (uint32_t)blockIdx.x*blockDim.x + threadIdx.x;          // 2^31 threads: every pattern with the sign bit clear
	float x; memcpy(&x, &bits, 4);
	const float a = sqrtf(x), b = (float)sqrt((double)x);
	uint32_t ua, ub; memcpy(&ua, &a, 4); memcpy(&ub, &b, 4);
	const bool same = ua == ub || (a != a && b != b);                                // (NaN in, NaN out: the payload is not part of the contract)
	if(!same) { atomicAdd(bad, 1ull); atomicMin(first_bad, bits); }
}

int main() {
	unsigned long long *bad; uint32_t *first;
	if(hipMalloc(&bad, 8) != hipSuccess || hipMalloc(&first, 4) != hipSuccess) { printf("no device\n"); return 2; }
	hipMemset(bad, 0, 8); hipMemset(first, 0xFF, 4);
	hipLaunchKernelGGL(k_check, dim3(1u << 23), dim3(256), 0, 0, bad, first);
	unsigned long long hb = 0; uint32_t hf = 0;
	if(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("device error\n"); return 2; }
	printf("patterns 2147483648 mismatches %llu first 0x%08x\n", hb, hf);
	return hb ? 1 : 0;
}
