// Where do one-wave workgroups land?  HW_ID of every workgroup of a <<<N, 64, lds>>> launch: SIMD histogram per CU, and the time of a dependent
// scalar/vector chain (a stand-in for a latency-bound lone wave) at 1, 2, 4, 8 workgroups per CU.   hipcc --offload-arch=gfx950 -O2 simd_place.hip -o simd_place
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
template <int mode> __global__ void k(uint32_t *ids, uint32_t *sink, int iters) {
	extern __shared__ uint32_t lds[];
	uint32_t hw;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
	uint32_t xcc;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
	if(threadIdx.x == 0) { ids[2*blockIdx.x] = hw; ids[2*blockIdx.x + 1] = xcc; }
	uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)blockIdx.x), v = threadIdx.x;
	lds[threadIdx.x] = v;
	for(int i = 0; i < iters; i++) {
		if(mode == 0) {                 // dependent scalar + vector chain, a taken branch per iteration
			asm volatile("s_mul_i32 %0, %0, 1664525\n s_add_u32 %0, %0, 1013904223\n v_xor_b32 %1, %0, %1\n v_add_u32 %1, %1, %1\n" : "+s"(s), "+v"(v));
		} else {                        // + an LDS round trip
			asm volatile("s_mul_i32 %0, %0, 1664525\n s_add_u32 %0, %0, 1013904223\n v_xor_b32 %1, %0, %1\n v_and_b32 %1, 0xfc, %1\n ds_read_b32 %1, %1\n s_waitcnt lgkmcnt(0)\n" : "+s"(s), "+v"(v));
		}
	}
	if(v == 0x12345678) sink[0] = v; if(s == 0x12345679u && threadIdx.x == 0) sink[1] = 1;
}
int main() {
	uint32_t *ids, *sink; hipMalloc(&ids, 8*65536); hipMalloc(&sink, 64);
	for(int lds : {1024, 20480}) for(int per_cu : {1, 2, 4, 7}) for(int mode : {0, 1}) {
		const int n = 256*per_cu;
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		if(mode) hipLaunchKernelGGL(k<1>, dim3(n), dim3(64), lds, 0, ids, sink, 1000); else hipLaunchKernelGGL(k<0>, dim3(n), dim3(64), lds, 0, ids, sink, 1000); hipDeviceSynchronize();
		hipEventRecord(e0); if(mode) hipLaunchKernelGGL(k<1>, dim3(n), dim3(64), lds, 0, ids, sink, 20000); else hipLaunchKernelGGL(k<0>, dim3(n), dim3(64), lds, 0, ids, sink, 20000); hipEventRecord(e1); hipDeviceSynchronize();
		float ms; hipEventElapsedTime(&ms, e0, e1);
		std::vector<uint32_t> h(2*n); hipMemcpy(h.data(), ids, 8*n, hipMemcpyDeviceToHost);
		// HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]...
		std::map<uint32_t, int> simd_hist; std::map<uint32_t, std::map<uint32_t,int>> per_cu_simd;
		for(int i = 0; i < n; i++) { const uint32_t hw = h[2*i], simd = (hw >> 4) & 3, cu = ((hw >> 8) & 0xF) | ((hw >> 12) & 0xF) << 4 | (h[2*i+1] & 0xF) << 8; simd_hist[simd]++; per_cu_simd[cu][simd]++; }
		int maxper = 0, cus = (int)per_cu_simd.size(); for(auto &c : per_cu_simd) for(auto &s : c.second) maxper = s.second > maxper ? s.second : maxper;
		printf("lds %5d  %d WG/CU mode %d: %.3f ms (%.1f clk/iter @2.4GHz)  SIMD histogram %d %d %d %d  distinct CUs %d  max waves on one SIMD %d\n", lds, per_cu, mode, ms, ms*1e-3*2.4e9/20000, simd_hist[0], simd_hist[1], simd_hist[2], simd_hist[3], cus, maxper);
	}
	return 0;
}
