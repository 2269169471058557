// kernels_common.h — device helpers shared by the HIP kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "device_plan.h"

namespace corto_hip {

// Pointers that come out of job structs in memory are "generic" to the compiler, which then emits FLAT loads and
// stores.  A FLAT store bumps lgkmcnt as well as vmcnt, so the next LDS read stalls until the store has been
// acknowledged by the memory system (~1-2 us) - fatal for the serial per-blob chains.  These casts pin the
// address space so that global_load/global_store (vmcnt only) and ds_* (lgkmcnt only) are emitted.
#define CRT_GLOBAL __attribute__((address_space(1)))
#define CRT_LDS __attribute__((address_space(3)))
template <typename T> __device__ __forceinline__ CRT_GLOBAL T *as_global(T *p) { return (CRT_GLOBAL T *)p; }
template <typename T> __device__ __forceinline__ CRT_LDS T *as_lds(T *p) { return (CRT_LDS T *)p; }
// LDS pointer from its 32-bit byte address (address arithmetic in integers keeps running offsets in one register)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
template <typename T> __device__ __forceinline__ CRT_LDS T *lds_at(uint32_t addr) { return (CRT_LDS T *)addr; }
#pragma clang diagnostic pop

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint32_t wave_id() { return threadIdx.x >> 6; }

// max over the 64 lanes of a wave, result uniform (SGPR).  DPP row operations + row broadcasts (gfx9) instead of
// ds_bpermute shuffles: ~12 VALU instructions with no LDS round trip.
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#define CRT_DPP_MAX(ctrl, rmask) { const uint32_t o_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xf, false); v = o_ > v ? o_ : v; }
	CRT_DPP_MAX(0xB1, 0xf)      // quad_perm [1,0,3,2]
	CRT_DPP_MAX(0x4E, 0xf)      // quad_perm [2,3,0,1]
	CRT_DPP_MAX(0x141, 0xf)     // row_half_mirror
	CRT_DPP_MAX(0x140, 0xf)     // row_mirror: every lane of a 16-lane row holds the row max
	CRT_DPP_MAX(0x142, 0xa)     // row_bcast:15 into rows 1 and 3
	CRT_DPP_MAX(0x143, 0xc)     // row_bcast:31 into rows 2 and 3: lane 63 holds the wave max
#undef CRT_DPP_MAX
	return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// inclusive +scan of a u32 across the 64 lanes of a wave with DPP row shifts / row broadcasts (gfx9): six v_add_u32 with a
// DPP modifier.  The generic __shfl_up version below goes through ds_bpermute (an LDS-pipe round trip per step) and was half
// of the Tunstall decode kernel's time.
// XCD-aware job slots.  Block b runs on XCD b % 8 (observed - MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility" - and a matter
// of speed only: any placement gives the same bytes).  A grid of xcd_grid(n) blocks (kernels.h) gives every XCD a CONTIGUOUS eighth of the n jobs: K-BIT's consecutive
// jobs are the streams of one attribute, which write neighbouring bytes (the four colour components fill the same lines) and re-read one another's logs - through one
// L2 instead of four.  (The same eighth of the BLOBS on one XCD in every kernel of a step - K-TOPO, K-DELTA, K-NRM - was measured too: FETCH_SIZE unchanged, nothing
// survives in an L2 from one kernel to the next; not kept.)
__device__ __forceinline__ uint32_t xcd_slot(uint32_t block, uint32_t n) { return (block & 7u)*((n + 7u) >> 3) + (block >> 3); }   // >= n: no job (the last eighth's tail)

__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v) {
#define CRT_DPP_ADD(ctrl, rmask) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xf, false);
	CRT_DPP_ADD(0x111, 0xf)     // row_shr:1
	CRT_DPP_ADD(0x112, 0xf)     // row_shr:2
	CRT_DPP_ADD(0x114, 0xf)     // row_shr:4
	CRT_DPP_ADD(0x118, 0xf)     // row_shr:8   -> inclusive scan inside every 16-lane row
	CRT_DPP_ADD(0x142, 0xa)     // row_bcast:15 into rows 1 and 3
	CRT_DPP_ADD(0x143, 0xc)     // row_bcast:31 into rows 2 and 3
#undef CRT_DPP_ADD
	return v;
}

// inclusive scan across the 64 lanes of a wave (any additive type)
template <typename T>
__device__ __forceinline__ T wave_inclusive_scan(T v) {
	const uint32_t lane = lane_id();
#pragma unroll
	for(int d = 1; d < 64; d <<= 1) {
		T o = __shfl_up(v, d, 64);
		if(lane >= (uint32_t)d) v += o;
	}
	return v;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it waits until every global
// store of the wave has been acknowledged by the memory system (microseconds under load) - fatal in streaming loops that
// flush to HBM and then reuse an LDS buffer.  Here only lgkmcnt is drained; global loads/stores stay in flight.
__device__ __forceinline__ void lds_barrier() {
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// exclusive scan over a 256-thread block (4 waves); *total = block sum. smem: 4 entries of T.
__device__ __forceinline__ uint32_t wave_inclusive_scan_any(uint32_t v) { return wave_inclusive_scan_u32(v); }
template <typename T> __device__ __forceinline__ T wave_inclusive_scan_any(T v) { return wave_inclusive_scan(v); }

template <typename T, bool LDS_ONLY = false>
__device__ __forceinline__ T block256_exclusive_scan(T v, T *smem, T *total) {
	T inc = wave_inclusive_scan_any(v);
	const uint32_t lane = lane_id(), w = wave_id();
	if(LDS_ONLY) lds_barrier(); else __syncthreads();                 // smem may still be read by a previous call
	if(lane == 63) smem[w] = inc;
	if(LDS_ONLY) lds_barrier(); else __syncthreads();
	T s0 = smem[0], s1 = smem[1], s2 = smem[2], s3 = smem[3];
	T base = w == 0 ? T(0) : w == 1 ? s0 : w == 2 ? T(s0 + s1) : T(s0 + s1 + s2);
	*total = s0 + s1 + s2 + s3;
	return base + inc - v;
}

// MSB-first bit field [o, o+n) of a u32 word stream (src/bitstream.cpp:103-121 as random access).
// Words past nwords read as 0 (malformed streams cannot fault).
template <typename WordPtr>
__device__ __forceinline__ uint32_t bit_field(WordPtr w, uint32_t nwords, uint64_t o, uint32_t n) {
	if(n == 0) return 0;
	const uint64_t i = o >> 5;
	const uint32_t sh = (uint32_t)(o & 31);
	const uint32_t hi = i < nwords ? w[i] : 0u;
	const uint32_t lo = (sh + n > 32 && i + 1 < nwords) ? w[i + 1] : 0u;
	const uint64_t win = ((uint64_t)hi << 32) | lo;
	return (uint32_t)((win << sh) >> (64 - n));
}

// Look-back over the chunks [chunk0, c) of one chain (a long Tunstall stream, the log streams of one bit block): a chunk's output offset is
// the sum of its predecessors' totals.  Every chunk publishes its own total in its state word ("total of this chunk"), later
// "total of everything up to and including this chunk"; a chunk walks back over the words until it meets an inclusive one.
// State words start out 0; the word IS the payload (8-byte agent-scope atomics both sides, no fences: the per-XCD L2s are not
// coherent, sc1 accesses go to memory).
// NOT the textbook version, which spins on a predecessor that has not published yet: that needs the predecessor to be RUNNING, and
// MI355X promises no such thing - each XCD dispatches its share of a grid on its own, and with several launches in flight (eight
// contexts of a decode pool) XCD A can be full of launch X's chunks waiting for a chunk that XCD B has not started because B is full
// of launch Y's chunks waiting for one A has not started.  Measured: steps of 2.5 s (the spin's bound) in the pipelined bench.
// Here nobody waits: a chunk that finds a predecessor's word empty works that predecessor's total out ITSELF (`recompute`, all
// threads of the workgroup; a chunk's total is a function of its input alone) and walks on.  Polling first (`patience` reads, one
// by thread 0 per round) is only worth it where recomputing is dear.
constexpr uint64_t CHAIN_ST_LOCAL = 1ull << 62, CHAIN_ST_INCL = 2ull << 62, CHAIN_ST_MASK = (1ull << 62) - 1ull;
template <class Recompute>
__device__ __forceinline__ uint64_t chain_lookback(uint64_t *state, uint32_t c, uint32_t chunk0, uint64_t total, uint32_t patience, uint64_t *share, Recompute recompute) {
	const bool t0 = threadIdx.x == 0;
	if(c == chunk0) { if(t0) __hip_atomic_store(&state[c], CHAIN_ST_INCL | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return 0; }
	if(t0) __hip_atomic_store(&state[c], CHAIN_ST_LOCAL | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	uint64_t prefix = 0;
	for(uint32_t i = c - 1;; i--) {
		__syncthreads();                                              // (share is free again)
		if(t0) {
			uint64_t v = 0;
			for(uint32_t tries = 0; tries <= patience; tries++) {
				v = __hip_atomic_load(&state[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if(v >> 62) break;
				__builtin_amdgcn_s_sleep(8);
			}
			*share = v;
		}
		__syncthreads();
		const uint64_t v = *share;
		if(v >> 62) {
			prefix += v & CHAIN_ST_MASK;
			if((v >> 62) == 2) break;
		} else {
			const uint64_t t = recompute(i);                            // uniform over the workgroup
			prefix += t;
			if(t0) {                                                     // publish it on the predecessor's behalf (the same value it would store)
				unsigned long long expect = 0;
				__hip_atomic_compare_exchange_strong((unsigned long long *)&state[i], &expect, (unsigned long long)(CHAIN_ST_LOCAL | t), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
		if(i == chunk0) break;
	}
	if(t0) __hip_atomic_store(&state[c], CHAIN_ST_INCL | (prefix + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	return prefix;
}

// x86 cvttss2si: out-of-range / NaN -> INT_MIN (what the reference's (int) casts do on its CPU)
__device__ __forceinline__ int32_t f2i_x86(float x) {
	if(!(x > -2147483904.0f && x < 2147483648.0f)) return (int32_t)0x80000000;
	return (int32_t)x;
}
__device__ __forceinline__ int16_t f2s_x86(float x) { return (int16_t)(uint16_t)(uint32_t)f2i_x86(x); }

} // namespace corto_hip
