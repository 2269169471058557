// What does a LONE wave pay per instruction?  One wave per CU runs 64-instruction blocks of one kind, 2 000 times, between two s_memtime reads:
// independent / dependent VALU, independent / dependent SALU, DPP chains, the VALU -> SGPR -> SALU -> VALU hop, v_readlane with a fresh lane select,
// v_cndmask behind v_cmp.   hipcc --offload-arch=gfx950 -O2 issue_latency.hip -o issue_latency   (why K-TAB's loop costs ~9 clocks an instruction)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
template <int mode> __global__ void k(uint64_t *out, uint32_t *sink, int iters) {
	__shared__ uint32_t lds_[1024]; if(iters < 0) lds_[threadIdx.x] = 1;
	uint32_t v0 = threadIdx.x, v1 = threadIdx.x*3, v2 = 7, v3 = 9, s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)blockIdx.x), s1 = 5, s2 = 11, s3 = 13; uint64_t sm = 0; typedef uint32_t u4 __attribute__((ext_vector_type(4))); u4 q = {1, 2, 3, 4}; v2 = (threadIdx.x*16) & 0x3F0; const uint32_t *gp = (const uint32_t *)sink + (threadIdx.x & 7);
	const uint64_t t0 = __builtin_amdgcn_s_memtime();
	for(int i = 0; i < iters; i++) {
		if(mode == 0) asm volatile(REP16("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));      // each reads the one before
		if(mode == 1) asm volatile(REP16("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));          // four independent chains
		if(mode == 2) asm volatile(REP64("v_add_u32 %0, %0, 1\n") : "+v"(v0));                                                                                                          // one chain
		if(mode == 3) asm volatile(REP64("s_add_u32 %0, %0, 1\n") : "+s"(s0) :: "scc");
		if(mode == 4) asm volatile(REP16("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n") : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc");
		if(mode == 5) asm volatile(REP16("v_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n v_max_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n") : "+v"(v0));   // 64 instructions, 32 of them DPP
		if(mode == 6) asm volatile(REP16("v_readfirstlane_b32 %1, %0\n s_add_u32 %1, %1, 1\n v_add_u32 %0, %1, %0\n v_add_u32 %0, %0, 1\n") : "+v"(v0), "+s"(s0) :: "scc");                       // VALU -> SGPR -> SALU -> VALU
		if(mode == 7) asm volatile(REP16("s_and_b32 %1, %1, 63\n v_readlane_b32 %2, %0, %1\n s_add_u32 %1, %2, 1\n v_add_u32 %0, %0, 1\n") : "+v"(v0), "+s"(s0), "+s"(s1) :: "scc");             // lane select fresh from the SALU
		if(mode == 8) asm volatile(REP16("v_cmp_eq_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n") : "+v"(v0), "+v"(v1), "+v"(v2) :: "vcc");
		if(mode == 9) asm volatile(REP16("v_add_u32 %0, %0, 1\n s_add_u32 %2, %2, 1\n v_add_u32 %1, %1, 1\n s_add_u32 %3, %3, 1\n") : "+v"(v0), "+v"(v1), "+s"(s0), "+s"(s1) :: "scc");          // VALU and SALU alternating, two chains each
		if(mode == 10) asm volatile(REP16("v_add_u32 %0, %0, 1\n s_add_u32 %2, %2, 1\n v_add_u32 %0, %0, 1\n s_add_u32 %2, %2, 1\n") : "+v"(v0), "+v"(v1), "+s"(s0), "+s"(s1) :: "scc");         // ... one chain each
		if(mode == 12) asm volatile(REP16("ds_write_b32 %1, %0\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n") : "+v"(v0) : "v"(v2) : "memory");            // an LDS write every four
		if(mode == 13) asm volatile(REP16("v_cmp_eq_u32 %3, %0, %1\n v_cndmask_b32 %0, %0, %2, %3\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n") : "+v"(v0), "+v"(v1), "+v"(v2), "=s"(sm));   // the mask in an SGPR pair
		if(mode == 14) asm volatile(REP16("s_lshr_b32 %1, %1, 1\n v_mul_u32_u24 %0, %1, %0\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n") : "+v"(v0), "+s"(s0) :: "scc");                    // a VALU reading an SGPR the SALU just wrote
		if(mode == 15) asm volatile(REP16("ds_write_b128 %1, %2\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n") : "+v"(v0) : "v"(v2), "v"(q) : "memory");
		if(mode == 16) { asm volatile("s_mov_b64 exec, 1\n" ::: "exec"); asm volatile(REP16("ds_write_b128 %1, %2\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n") : "+v"(v0) : "v"(v2), "v"(q) : "memory"); asm volatile("s_mov_b64 exec, -1\n" ::: "exec"); }   // one lane
		if(mode == 17) asm volatile(REP16("ds_read_b32 %1, %2\n s_waitcnt lgkmcnt(0)\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, 1\n") : "+v"(v0), "+v"(v1) : "v"(v2) : "memory");       // a round trip every four
		if(mode == 18) asm volatile(REP16("ds_read_b128 %1, %2\n s_waitcnt lgkmcnt(0)\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n") : "+v"(v0), "+v"(q) : "v"(v2) : "memory");
		if(mode == 19) asm volatile(REP16("ds_bpermute_b32 %1, %2, %0\n s_waitcnt lgkmcnt(0)\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, 1\n") : "+v"(v0), "+v"(v1) : "v"(v2) : "memory");
		if(mode == 20) asm volatile(REP16("v_readlane_b32 %1, %0, 5\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n") : "+v"(v0), "+s"(s0));                          // nobody reads the SGPR
		if(mode == 21) asm volatile(REP16("v_writelane_b32 %0, %1, 5\n v_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n v_add_u32 %0, %0, 1\n") : "+v"(v0), "+s"(s0) :: "scc");
		if(mode == 22) asm volatile(REP16("s_cbranch_scc0 1f\n s_nop 0\n1:\n s_cmp_eq_u32 %0, %0\n s_cbranch_scc1 2f\n s_nop 0\n2:\n v_add_u32 %1, %1, 1\n") : "+s"(s0), "+v"(v0) :: "scc");   // one taken branch in four (the s_nops skipped or not)
		if(mode == 23) asm volatile(REP16("s_and_saveexec_b64 %1, vcc\n v_add_u32 %0, %0, 1\n s_or_b64 exec, exec, %1\n v_add_u32 %0, %0, 1\n") : "+v"(v0), "=s"(sm) :: "scc", "exec");
		if(mode == 24) asm volatile(REP16("s_load_dword %1, %2, 0x0\n s_waitcnt lgkmcnt(0)\n s_add_u32 %1, %1, 1\n v_add_u32 %0, %0, 1\n") : "+v"(v0), "+s"(s0) : "s"(out) : "scc", "memory");   // a scalar load round trip (cache hit)
		if(mode == 25) asm volatile(REP16("global_load_dword %1, %2, off\n s_waitcnt vmcnt(0)\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, 1\n") : "+v"(v0), "+v"(v1) : "v"(gp) : "memory");   // a vector load round trip (L1/L2 hit)
		if(mode == 26) asm volatile(REP64("v_mul_lo_u32 %0, %0, %1\n") : "+v"(v0) : "v"(v1));
		if(mode == 27) asm volatile(REP16("v_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n v_add_u32 %0, %0, 1\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
		if(mode == 11) asm volatile(REP16("v_mul_u32_u24 %0, %0, %1\n v_lshrrev_b32 %0, 16, %0\n v_or3_b32 %0, %0, %1, %2\n v_and_or_b32 %0, %0, %2, %1\n") : "+v"(v0), "+v"(v1), "+v"(v2));  // K-TAB's kind of dependent VALU
	}
	const uint64_t t1 = __builtin_amdgcn_s_memtime();
	if(threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
	if(v0 + v1 + v2 + v3 + q.x == 0x12345678u) sink[0] = v0; if(s0 + s1 + s2 + s3 + (uint32_t)sm == 0x12345679u && threadIdx.x == 0) sink[1] = 1;
}
template <int mode> static void run(const char *what, uint64_t *out, uint32_t *sink, int per_cu) {
	const int n = 256*per_cu, iters = 2000;
	hipLaunchKernelGGL(k<mode>, dim3(n), dim3(64), 0, 0, out, sink, 100); hipDeviceSynchronize();
	hipLaunchKernelGGL(k<mode>, dim3(n), dim3(64), 0, 0, out, sink, iters); hipDeviceSynchronize();
	std::vector<uint64_t> h(n); hipMemcpy(h.data(), out, 8*n, hipMemcpyDeviceToHost);
	double sum = 0; for(auto x : h) sum += (double)x;
	printf("%d wave(s) a CU  %-66s %6.2f clocks an instruction\n", per_cu, what, sum/n/iters/64.0);
}
int main() {
	setvbuf(stdout, nullptr, _IONBF, 0);
	uint64_t *out; uint32_t *sink; hipMalloc(&out, 8*65536); hipMalloc(&sink, 64);
	for(int per_cu : {1, 8}) {
		run<0>("VALU, each reads the one before (4 registers)", out, sink, per_cu);
		run<1>("VALU, four independent chains", out, sink, per_cu);
		run<2>("VALU, one chain on one register", out, sink, per_cu);
		run<11>("VALU, one chain of mul24 / lshr / or3 / and_or", out, sink, per_cu);
		run<3>("SALU, one chain", out, sink, per_cu);
		run<4>("SALU, four independent chains", out, sink, per_cu);
		run<9>("VALU / SALU alternating, two chains each", out, sink, per_cu);
		run<10>("VALU / SALU alternating, one chain each", out, sink, per_cu);
		run<5>("DPP chain (v_max_u32_dpp + s_nop 1, both counted)", out, sink, per_cu);
		run<6>("v_readfirstlane -> s_add -> v_add -> v_add", out, sink, per_cu);
		run<7>("s_and -> v_readlane (fresh lane select) -> s_add ; v_add", out, sink, per_cu);
		run<8>("v_cmp -> v_cndmask vcc ; 2 x v_add", out, sink, per_cu);
		run<13>("v_cmp -> v_cndmask on an SGPR pair ; 2 x v_add", out, sink, per_cu);
		run<12>("ds_write ; 3 x v_add", out, sink, per_cu);
		run<14>("s_lshr -> v_mul reading it ; 2 x v_add", out, sink, per_cu);
		run<26>("v_mul_lo_u32, one chain", out, sink, per_cu);
		run<27>("3 x v_mov_b32_dpp of one source ; v_add of the source", out, sink, per_cu);
		run<15>("ds_write_b128 ; 3 x v_add", out, sink, per_cu);
		run<16>("ds_write_b128 of ONE lane ; 3 x v_add", out, sink, per_cu);
		run<17>("ds_read_b32 ; wait ; 2 x v_add  (a round trip every four)", out, sink, per_cu);
		run<18>("ds_read_b128 ; wait ; 2 x v_add", out, sink, per_cu);
		run<19>("ds_bpermute_b32 ; wait ; 2 x v_add", out, sink, per_cu);
		run<20>("v_readlane (constant lane, result unused) ; 3 x v_add", out, sink, per_cu);
		run<21>("v_writelane ; v_add ; s_add ; v_add", out, sink, per_cu);
		run<22>("branches: one not taken, one taken, per four (skipped s_nops not counted)", out, sink, per_cu);
		run<23>("s_and_saveexec ; v_add ; s_or exec ; v_add", out, sink, per_cu);
		run<24>("s_load_dword ; wait ; s_add ; v_add", out, sink, per_cu);
		run<25>("global_load_dword ; wait ; 2 x v_add", out, sink, per_cu);

	}
	return 0;
}
