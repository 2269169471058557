// k_stream.hip — the stream-level (scan + gather) stages for gfx950:
//   device-wide exclusive scans of per-chunk partials,
//   K-BIT  bit-unpack of log/field streams      include/corto/cstream.h:294-360, src/bitstream.cpp:103-121
//   K-DELTA (point clouds) running sums         include/corto/vertex_attribute.h:177-181, src/normal_attribute.cpp:202-207
//   K-DEQ  dequantisation                       include/corto/vertex_attribute.h:184-193, src/color_attribute.cpp:72-95
//
// Every stage is "scan + gather" over chunks of CHUNK elements; a chunk->job map built by the host
// lets one launch cover every blob of the batch.
#include <type_traits>

#include "kernels_common.h"

namespace corto_hip {

// ------------------------------------------------------------------------------------------------
// exclusive scan of a[0..n) in place, one 1024-thread workgroup (n = number of chunks: small)
__global__ __launch_bounds__(1024) void k_scan_u64(uint64_t *__restrict__ a, uint32_t n) {
	__shared__ uint64_t wsum[16];
	__shared__ uint64_t carry_s;
	if(threadIdx.x == 0) carry_s = 0;
	__syncthreads();
	const uint32_t lane = lane_id(), w = wave_id();
	for(uint32_t base = 0; base < n; base += 1024) {
		const uint32_t i = base + threadIdx.x;
		const uint64_t v = i < n ? a[i] : 0;
		uint64_t inc = wave_inclusive_scan(v);
		if(lane == 63) wsum[w] = inc;
		__syncthreads();
		uint64_t wbase = 0, total = 0;
#pragma unroll
		for(uint32_t k = 0; k < 16; k++) { const uint64_t s = wsum[k]; if(k < w) wbase += s; total += s; }
		const uint64_t carry = carry_s;
		if(i < n) a[i] = carry + wbase + inc - v;
		__syncthreads();
		if(threadIdx.x == 0) carry_s = carry + total;
		__syncthreads();
	}
}

// u32 array scan in three phases (used for CSR offsets / boundary slots): per-chunk sums ...
__global__ __launch_bounds__(256) void k_u32_chunk_sums(const uint32_t *__restrict__ a, uint32_t n, uint64_t *__restrict__ partial) {
	const uint32_t c = blockIdx.x;
	const uint32_t i0 = c*CHUNK + 4*threadIdx.x;
	uint32_t s = 0;
#pragma unroll
	for(int k = 0; k < 4; k++) if(i0 + k < n) s += a[i0 + k];
#pragma unroll
	for(int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
	__shared__ uint32_t red[4];
	if(lane_id() == 0) red[wave_id()] = s;
	__syncthreads();
	if(threadIdx.x == 0) partial[c] = (uint64_t)red[0] + red[1] + red[2] + red[3];
}
// ... and the exclusive apply into out (may alias a; partial already scanned by k_scan_u64)
__global__ __launch_bounds__(256) void k_u32_chunk_apply(const uint32_t *a, uint32_t *out, uint32_t n, const uint64_t *__restrict__ partial) {
	const uint32_t c = blockIdx.x;
	const uint32_t i0 = c*CHUNK + 4*threadIdx.x;
	uint32_t v[4], s = 0;
#pragma unroll
	for(int k = 0; k < 4; k++) { v[k] = i0 + k < n ? a[i0 + k] : 0u; s += v[k]; }
	__shared__ uint32_t smem[4];
	uint32_t total;
	uint32_t o = (uint32_t)partial[c] + block256_exclusive_scan<uint32_t>(s, smem, &total);
#pragma unroll
	for(int k = 0; k < 4; k++) { if(i0 + k < n) out[i0 + k] = o; o += v[k]; }
}

// ------------------------------------------------------------------------------------------------
// K-BIT.  chunk c -> job chunk_job[c]; the job's bit block may be shared by several jobs (one per
// component, component-major: cstream.h:300-317), which is why offsets are taken relative to the first
// chunk of the whole bit block (chain_chunk0) after ONE device-wide scan of all chunk sums.
__device__ __forceinline__ uint32_t unpack_bits_of(const UnpackJob &J, uint32_t log) {
	const uint32_t d = log > 32 ? 32u : log;          // >32 cannot be produced by the encoder (UB in the reference)
	return d*J.fields;
}

// One pass: a chunk adds up the bits of its own 1 024 logs, finds where its fields start by look-back over the earlier chunks of
// its bit block (chain_lookback, wait-free: at most a dozen state words for a 4-component attribute of a C4 blob) and extracts - round 1 ran a
// sums kernel and a device-wide scan in front of this one, two more launches on the attribute chain of every batch.
// the (up to four) logs at i0 .. of a stream as one load: an unaligned dword, or - the stream's last, partial group - its last dword
// shifted down (streams shorter than four logs: bytewise).  r = the number of valid ones.
__device__ __forceinline__ uint32_t unpack_logs4(CRT_GLOBAL const uint8_t *logs, uint32_t count, uint32_t i0, uint32_t &r) {
	r = i0 < count ? min(count - i0, 4u) : 0u;
	if(count >= 4) {
		uint32_t dw = *(CRT_GLOBAL const uint32_t *)(logs + (r == 4u ? i0 : r ? count - 4u : 0u));
		asm volatile("" : "+v"(dw));
		return r == 4u ? dw : r ? dw >> (8u*(4u - r)) : 0u;
	}
	uint32_t raw = 0;
	for(uint32_t k = 0; k < r; k++) raw |= (uint32_t)logs[i0 + k] << (8u*k);
	return raw;
}

__global__ __launch_bounds__(256) void k_unpack_extract(const UnpackJob *__restrict__ jobs, const uint32_t *__restrict__ chunk_job,
                                                        uint32_t nchunks, uint64_t *state) {
	const uint32_t c = blockIdx.x;
	if(c >= nchunks) return;
	const UnpackJob J = jobs[chunk_job[c]];
	const uint32_t i0 = (c - J.chunk0)*CHUNK + 4*threadIdx.x;
	// Every load of a phase is in flight together: a thread's four logs are ONE dword, the bit words of its values are fetched
	// unconditionally on clamped indices and pinned, the stores follow.  (As `i < count ? logs[i] : 0` and a bit_field() per value the
	// compiler made a dozen dependent FLAT round trips per thread - generic pointers, every load in its own branch with its own wait.)
	uint32_t r;
	const uint32_t raw = unpack_logs4(as_global(J.logs), J.count, i0, r);
	uint32_t lg[4], s = 0;
#pragma unroll
	for(int k = 0; k < 4; k++) {
		lg[k] = (uint32_t)k < r ? (raw >> (8*k)) & 255u : 0u;
		if(lg[k] > 32) lg[k] = 32;
		s += lg[k]*J.fields;
	}
	__shared__ uint32_t smem[4];
	__shared__ uint64_t before;                        // look-back scratch
	uint32_t total;
	const uint32_t mine = block256_exclusive_scan<uint32_t>(s, smem, &total);
	__shared__ uint32_t red[4];
	auto chunk_bits = [&](uint32_t ci) -> uint64_t {      // the bits of another chunk of this bit block (maybe another component's log stream)
		const UnpackJob K = jobs[chunk_job[ci]];
		uint32_t rk;
		const uint32_t w4 = unpack_logs4(as_global(K.logs), K.count, (ci - K.chunk0)*CHUNK + 4*threadIdx.x, rk);
		uint32_t t = 0;
#pragma unroll
		for(int k = 0; k < 4; k++) if((uint32_t)k < rk) t += unpack_bits_of(K, (w4 >> (8*k)) & 255u);
#pragma unroll
		for(int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d, 64);
		__syncthreads();
		if(lane_id() == 0) red[wave_id()] = t;
		__syncthreads();
		return (uint64_t)red[0] + red[1] + red[2] + red[3];
	};
	uint64_t o = chain_lookback(state, c, J.chain_chunk0, total, 0u, &before, chunk_bits) + mine;
	CRT_GLOBAL const uint32_t *words = as_global(J.words);
	const uint32_t nwords = J.nwords, last_word = nwords ? nwords - 1u : 0u;
	// two words of the bit block at bit offset `at` (clamped: what bit_field() would not read comes back as the last word and is masked)
	auto window = [&](uint64_t at, uint32_t &hi, uint32_t &lo) {
		const uint64_t wi = at >> 5;
		hi = words[wi < nwords ? (uint32_t)wi : last_word];
		lo = words[wi + 1 < nwords ? (uint32_t)wi + 1u : last_word];
	};
	auto field = [&](uint64_t at, uint32_t n, uint32_t hi, uint32_t lo) -> uint32_t {        // = bit_field(words, nwords, at, n)
		if(n == 0) return 0u;
		const uint64_t wi = at >> 5;
		const uint32_t sh = (uint32_t)(at & 31);
		const uint32_t h = wi < nwords ? hi : 0u, l = (sh + n > 32 && wi + 1 < nwords) ? lo : 0u;
		return (uint32_t)(((((uint64_t)h << 32) | l) << sh) >> (64 - n));
	};
	if(J.mode != 0) {                                  // decodeValues: sign folding (cstream.h:304-316); one field per log
		uint64_t at[4];
		uint32_t hi[4], lo[4];
#pragma unroll
		for(int k = 0; k < 4; k++) { at[k] = o; o += lg[k]; }
		if(nwords) {
#pragma unroll
			for(int k = 0; k < 4; k++) window(at[k], hi[k], lo[k]);
			asm volatile("" : "+v"(hi[0]), "+v"(lo[0]), "+v"(hi[1]), "+v"(lo[1]), "+v"(hi[2]), "+v"(lo[2]), "+v"(hi[3]), "+v"(lo[3]));
		} else {
#pragma unroll
			for(int k = 0; k < 4; k++) hi[k] = lo[k] = 0;
		}
#pragma unroll
		for(int k = 0; k < 4; k++) {
			const uint32_t i = i0 + k, d = lg[k];
			int32_t v = 0;
			if(d) {
				v = (int32_t)field(at[k], d, hi[k], lo[k]);
				const int32_t mid = (int32_t)(1u << (d - 1));
				if(v < mid) v = -v - mid;
			}
			if((uint32_t)k < r && i < J.out_limit) {
				if(J.out_u8) as_global((uint8_t *)J.out)[(size_t)i*J.stride + J.comp] = (uint8_t)v;
				else as_global((int32_t *)J.out)[(size_t)i*J.stride + J.comp] = v;
			}
		}
		return;
	}
	// decodeArray: v = raw - 2^(d-1); d == 0 -> zeros (cstream.h:337-357); `fields` values per log, the same width
#pragma unroll
	for(int k = 0; k < 4; k++) {
		const uint32_t i = i0 + k, d = lg[k];
		if((uint32_t)k >= r) break;
		const bool store = i < J.out_limit;
		CRT_GLOBAL int32_t *out = as_global((int32_t *)J.out) + (size_t)i*J.stride;
		const uint32_t half = (uint32_t)((int32_t)(1u << (d & 31u)) >> 1);   // upstream's `(1<<diff)>>1` in INT (cstream.h:343): 2^(d-1), 0 at d = 0; at d = 32 the compiled reference's shifter takes the count mod 32 (0); at d = 31 its arithmetic shift gives -2^30 (upstream does not round-trip there: the bytes are the contract)
		if(J.fields <= 4 && nwords) {                  // (uniform) every field's words in flight together
			uint32_t hi[4], lo[4];
#pragma unroll
			for(uint32_t f = 0; f < 4; f++) window(o + (uint64_t)(f < J.fields ? f : 0u)*d, hi[f], lo[f]);
			asm volatile("" : "+v"(hi[0]), "+v"(lo[0]), "+v"(hi[1]), "+v"(lo[1]), "+v"(hi[2]), "+v"(lo[2]), "+v"(hi[3]), "+v"(lo[3]));
#pragma unroll
			for(uint32_t f = 0; f < 4; f++) if(f < J.fields) {
				const int32_t v = d ? (int32_t)(field(o + (uint64_t)f*d, d, hi[f], lo[f]) - half) : 0;
				if(store) out[f] = v;
			}
			o += (uint64_t)d*J.fields;
		} else {
			for(uint32_t f = 0; f < J.fields; f++) {
				const int32_t v = d ? (int32_t)(bit_field(words, nwords, o, d) - half) : 0;
				o += d;
				if(store) out[f] = v;
			}
		}
	}
}

// K-BIT for the attributes of LDS-sized blobs (round 3): ONE WAVE PER LOG STREAM, lanes interleaved over the vertices.
// The chunked kernel above gives every 1 024 logs a workgroup of four waves (a C4 batch: 5 632 workgroups, 22 528 waves, each living
// ~6 us, two thirds of it waiting for its loads) and finds a chunk's bit offset by look-back through memory; launched twice it cost a
// pipelined decode 11.8 us of its 89 us per batch (profiles/r03_what_bounds_the_pipeline.txt) - more than its 31 us alone would suggest,
// because what it takes is wave slots and issue slots, 88 M wave-cycles a batch, more than every other kernel of the path together.
// Here a stream is ONE wave's: lane l takes logs l, l + 64, ... (every load and store of a round is 64 consecutive elements), a round's
// bit offsets are one DPP scan, the cursor is carried in a scalar - and the streams in front of it in the bit block (one per
// component, component-major: cstream.h:300-317) are simply added up again from their logs (a few KB), so nobody waits for anybody:
// no state words, no atomics.  A tenth of the waves, a quarter of the wave-cycles (the instruction count is the same).
constexpr uint32_t UW_R = 4;                                                // rounds of 64 logs per block: their loads in flight together
// Bit offsets are 32-bit here (the planner sends a bit block this way only when it has fewer than 2^26 words): round 3's 64-bit cursors made
// every window two 64-bit compares, a 64-bit shift and two 64-bit address computations - a third of the kernel's vector instructions, and
// the pipelined rate is within 2x of the chip's VALU issue rate.
__global__ __launch_bounds__(64) void k_unpack_wave(const UnpackJob *__restrict__ jobs, const uint32_t *__restrict__ job_ids, uint32_t njobs) {
	// (one stream a workgroup of one wave; four streams a workgroup was measured level in round 4)
	// XCD-aware slots (kernels_common.h): the streams of one attribute are consecutive jobs - the four colour components write the bytes of the SAME lines
	// (out[4*i + comp]), and every stream re-adds the logs of the ones in front of it: through one L2, a C4 launch's WRITE_SIZE 23.5 -> 13.1 MB, FETCH_SIZE 5.8 -> 3.5
	const uint32_t slot = xcd_slot(blockIdx.x, njobs);
	if(slot >= njobs) return;
	const uint32_t jid = job_ids[slot];
	const UnpackJob J = jobs[jid];
	const uint32_t lane = threadIdx.x, count = J.count, fields = J.fields;
	CRT_GLOBAL const uint8_t *logs = as_global(J.logs);
	auto width = [](uint32_t l) -> uint32_t { return l > 32u ? 32u : l; };   // >32 cannot be produced by the encoder (UB in the reference)
	// the first block's logs, and everything in front of this stream in its bit block, in flight together
	uint32_t lg[UW_R];
#pragma unroll
	for(uint32_t r = 0; r < UW_R; r++) { const uint32_t i = r*64u + lane; lg[r] = logs[i < count ? i : (count ? count - 1u : 0u)]; }
	uint32_t running = 0;
	for(uint32_t jp = J.chain_chunk0; jp < jid; jp++) {                        // (chain_chunk0: for this kernel, the first JOB of the bit block)
		const UnpackJob &K = jobs[jp];
		CRT_GLOBAL const uint8_t *kl = as_global(K.logs);
		const uint32_t kc = K.count, kf = K.fields;
		uint32_t t = 0;
		for(uint32_t i0 = 4u*lane; i0 < kc; i0 += 4u*64u*4u) {                // four dwords per lane and pass
			uint32_t w4[4], r4[4];
#pragma unroll
			for(uint32_t u = 0; u < 4; u++) w4[u] = unpack_logs4(kl, kc, i0 + 256u*u, r4[u]);
#pragma unroll
			for(uint32_t u = 0; u < 4; u++)
#pragma unroll
				for(uint32_t k = 0; k < 4; k++) if(k < r4[u]) t += width((w4[u] >> (8u*k)) & 255u);
		}
		t = wave_inclusive_scan_u32(t);
		running += (uint32_t)__builtin_amdgcn_readlane((int)t, 63)*kf;
	}
	CRT_GLOBAL const uint32_t *words = as_global(J.words);
	const uint32_t nwords = J.nwords, last_word = nwords ? nwords - 1u : 0u, nbits = nwords << 5;   // (nwords < 2^26: the planner)
	// two words of the bit block at bit offset `at`.  INSIDE: the field lies inside the bit block - every field of a well-formed stream -
	// so only the second word's index can run one past the end (a field that ends on a word boundary) and nothing needs masking;
	// otherwise clamped loads, and field() masks what bit_field() would not have read
	auto window = [&](auto INSIDE, uint32_t at, uint32_t &hi, uint32_t &lo) {
		const uint32_t wi = at >> 5;
		hi = words[min(wi, last_word)];                                         // (a zero-width lane of an exactly full block sits AT nbits: clamp in both modes)
		lo = words[min(wi + 1u, last_word)];
	};
	auto field = [&](auto INSIDE, uint32_t at, uint32_t n, uint32_t hi, uint32_t lo) -> uint32_t {   // = bit_field(words, nwords, at, n); n == 0 -> 0
		const uint32_t sh = at & 31u;
		uint32_t h = hi, l = lo;
		if constexpr(!decltype(INSIDE)::value) {
			const uint32_t wi = at >> 5;
			h = wi < nwords ? hi : 0u; l = (sh + n > 32 && wi + 1 < nwords) ? lo : 0u;
		}
		const uint32_t top = sh ? __builtin_amdgcn_alignbit(h, l, 32u - sh) : h;   // the 32 bits from bit `sh` on: a funnel shift, not a 64-bit one (sh = 0: the shift count 32 would wrap to 0 and give l)
		return (n ? top : 0u) >> ((32u - n) & 31u);                           // top >> (32 - n) with neither the shift by 32 at n = 0 nor the one by -1 at n = 32 (a full word: ADVICE r4)
	};
	const bool values = (J.mode & 1u) != 0;
	const uint32_t stride = J.stride, comp = J.comp, out_limit = J.out_limit;
	const bool out_u8 = J.out_u8 == 1, out_i16 = J.out_u8 == 2;              // (2: every width of the stream is <= 16 bits - its table says so, plan_jobs.cpp - and K-DELTA / K-NRM read int16)
	for(uint32_t base = 0; base < count; base += UW_R*64u) {
		// this block's widths and bit offsets (a scan per round, the cursor a scalar), then the next block's logs go out before the windows
		uint32_t d[UW_R], at[UW_R];
		bool ok = true;
#pragma unroll
		for(uint32_t r = 0; r < UW_R; r++) {
			const uint32_t i = base + r*64u + lane;
			d[r] = i < count ? width(lg[r]) : 0u;
			const uint32_t bits = d[r]*fields, incl = wave_inclusive_scan_u32(bits);
			at[r] = running + (incl - bits);
			ok = ok && running + incl <= nbits;
			running += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
		}
		const bool inside = __all(ok) && nwords != 0;                            // (uniform) the whole block's fields inside the bit block
		if(base + UW_R*64u < count) {
#pragma unroll
			for(uint32_t r = 0; r < UW_R; r++) { const uint32_t i = base + (UW_R + r)*64u + lane; lg[r] = logs[i < count ? i : count - 1u]; }
		}
		auto body = [&](auto INSIDE) {
		if(values) {                                                           // decodeValues: sign folding (cstream.h:304-316); one field per log
			uint32_t hi[UW_R], lo[UW_R];
			if(nwords) {
#pragma unroll
				for(uint32_t r = 0; r < UW_R; r++) window(INSIDE, at[r], hi[r], lo[r]);
#pragma unroll
				for(uint32_t r = 0; r < UW_R; r++) asm volatile("" : "+v"(hi[r]), "+v"(lo[r]));
			} else {
#pragma unroll
				for(uint32_t r = 0; r < UW_R; r++) hi[r] = lo[r] = 0;
			}
#pragma unroll
			for(uint32_t r = 0; r < UW_R; r++) {
				const uint32_t i = base + r*64u + lane, dd = d[r];
				int32_t v = (int32_t)field(INSIDE, at[r], dd, hi[r], lo[r]);
				const int32_t mid = (int32_t)(dd ? 1u << (dd - 1u) : 0u);        // (dd == 0: v = 0, mid = 0: stays 0; dd == 32: 2^31, as the chunked kernel and the oracle)
				v = v < mid ? -v - mid : v;
				if(i < count && i < out_limit) {
					if(out_u8) as_global((uint8_t *)J.out)[i*stride + comp] = (uint8_t)v;
					else if(out_i16) as_global((int16_t *)J.out)[i*stride + comp] = (int16_t)v;
					else as_global((int32_t *)J.out)[i*stride + comp] = v;
				}
			}
		} else {                                                                 // decodeArray: v = raw - 2^(d-1); d == 0 -> zeros (cstream.h:337-357); `fields` values per log
#pragma unroll
			for(uint32_t g = 0; g < UW_R; g += 2) {                               // two rounds at a time: up to eight fields' windows in flight
				uint32_t hi[2][4], lo[2][4];
				const bool fast = fields <= 4 && nwords;                         // (uniform)
				if(fast) {
#pragma unroll
					for(uint32_t k = 0; k < 2; k++)
#pragma unroll
						for(uint32_t f = 0; f < 4; f++) window(INSIDE, at[g + k] + (f < fields ? f : 0u)*d[g + k], hi[k][f], lo[k][f]);
#pragma unroll
					for(uint32_t k = 0; k < 2; k++) asm volatile("" : "+v"(hi[k][0]), "+v"(lo[k][0]), "+v"(hi[k][1]), "+v"(lo[k][1]), "+v"(hi[k][2]), "+v"(lo[k][2]), "+v"(hi[k][3]), "+v"(lo[k][3]));
				}
#pragma unroll
				for(uint32_t k = 0; k < 2; k++) {
					const uint32_t i = base + (g + k)*64u + lane, dd = d[g + k];
					const bool store = i < count && i < out_limit;
					CRT_GLOBAL int32_t *out = as_global((int32_t *)J.out) + i*stride;
					CRT_GLOBAL int16_t *out16 = as_global((int16_t *)J.out) + i*stride;
					const uint32_t half = (uint32_t)((int32_t)(1u << (dd & 31u)) >> 1);   // upstream's `(1<<diff)>>1` in INT (cstream.h:343): 0 at dd = 32 (shift count mod 32), -2^30 at dd = 31 (arithmetic shift of INT_MIN), as k_unpack_extract
					if(fast) {
						int32_t v[4];
#pragma unroll
						for(uint32_t f = 0; f < 4; f++) v[f] = (int32_t)(field(INSIDE, at[g + k] + (f < fields ? f : 0u)*dd, dd, hi[k][f], lo[k][f]) - half);
						if(store && out_i16) {                                       // halfwords: the vertex' record is 2-byte aligned (4 when it has two or four fields)
							typedef int16_t i16x2_t __attribute__((ext_vector_type(2)));
							typedef int16_t i16x4_t __attribute__((ext_vector_type(4)));
							typedef i16x2_t __attribute__((aligned(4))) i16x2u; typedef i16x4_t __attribute__((aligned(4))) i16x4u;
							if(fields == 3) { out16[0] = (int16_t)v[0]; out16[1] = (int16_t)v[1]; out16[2] = (int16_t)v[2]; }
							else if(fields == 2) *(CRT_GLOBAL i16x2u *)out16 = i16x2_t{(int16_t)v[0], (int16_t)v[1]};
							else if(fields == 4) *(CRT_GLOBAL i16x4u *)out16 = i16x4_t{(int16_t)v[0], (int16_t)v[1], (int16_t)v[2], (int16_t)v[3]};
							else out16[0] = (int16_t)v[0];
						} else
						if(store) {                                                  // one vector store a vertex (the caller's ints are 4-byte aligned)
							typedef int32_t i32x2_t __attribute__((ext_vector_type(2)));
							typedef int32_t i32x3_t __attribute__((ext_vector_type(3)));
							typedef int32_t i32x4_t __attribute__((ext_vector_type(4)));
							typedef i32x2_t __attribute__((aligned(4))) i32x2u; typedef i32x3_t __attribute__((aligned(4))) i32x3u; typedef i32x4_t __attribute__((aligned(4))) i32x4u;
							if(fields == 3) *(CRT_GLOBAL i32x3u *)out = i32x3_t{v[0], v[1], v[2]};
							else if(fields == 2) *(CRT_GLOBAL i32x2u *)out = i32x2_t{v[0], v[1]};
							else if(fields == 4) *(CRT_GLOBAL i32x4u *)out = i32x4_t{v[0], v[1], v[2], v[3]};
							else out[0] = v[0];
						}
					} else {
						uint32_t oo = at[g + k];
						for(uint32_t f = 0; f < fields; f++) {
							const int32_t v = dd ? (int32_t)(bit_field(words, nwords, (uint64_t)oo, dd) - half) : 0;
							oo += dd;
							if(store) { if(out_i16) out16[f] = (int16_t)v; else out[f] = v; }
						}
					}
				}
			}
		}
		};
		if(inside) body(std::true_type{}); else body(std::false_type{});
	}
}

// ------------------------------------------------------------------------------------------------
// point-cloud delta: v[i] += v[i-N] over the flat array == per-component inclusive scan (wrap-around)
__device__ __forceinline__ uint32_t cloud_load(const CloudJob &J, uint32_t i, uint32_t comp) {
	return J.is_u8 ? (uint32_t)((const uint8_t *)J.values)[(size_t)i*J.N + comp] : ((const uint32_t *)J.values)[(size_t)i*J.N + comp];
}

__global__ __launch_bounds__(256) void k_cloud_sums(const CloudJob *__restrict__ jobs, const uint32_t *__restrict__ chunk_job,
                                                    uint32_t nchunks, uint64_t *__restrict__ partial) {
	const uint32_t c = blockIdx.x;
	if(c >= nchunks) return;
	const CloudJob J = jobs[chunk_job[c]];
	const uint32_t cpc = (J.nvert + CHUNK - 1)/CHUNK, cj = c - J.chunk0;
	const uint32_t comp = cj/cpc, i0 = (cj - comp*cpc)*CHUNK + 4*threadIdx.x;
	uint32_t s = 0;
#pragma unroll
	for(int k = 0; k < 4; k++) if(i0 + k < J.nvert) s += cloud_load(J, i0 + k, comp);
#pragma unroll
	for(int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
	__shared__ uint32_t red[4];
	if(lane_id() == 0) red[wave_id()] = s;
	__syncthreads();
	if(threadIdx.x == 0) partial[c] = (uint64_t)(uint32_t)(red[0] + red[1] + red[2] + red[3]);
}

__global__ __launch_bounds__(256) void k_cloud_apply(const CloudJob *__restrict__ jobs, const uint32_t *__restrict__ chunk_job,
                                                     uint32_t nchunks, const uint64_t *__restrict__ partial) {
	const uint32_t c = blockIdx.x;
	if(c >= nchunks) return;
	const CloudJob J = jobs[chunk_job[c]];
	const uint32_t cpc = (J.nvert + CHUNK - 1)/CHUNK, cj = c - J.chunk0;
	const uint32_t comp = cj/cpc, i0 = (cj - comp*cpc)*CHUNK + 4*threadIdx.x;
	uint32_t v[4], s = 0;
#pragma unroll
	for(int k = 0; k < 4; k++) { v[k] = i0 + k < J.nvert ? cloud_load(J, i0 + k, comp) : 0u; s += v[k]; }
	__shared__ uint32_t smem[4];
	uint32_t total;
	uint32_t o = (uint32_t)(partial[c] - partial[J.chunk0 + comp*cpc]) + block256_exclusive_scan<uint32_t>(s, smem, &total);
#pragma unroll
	for(int k = 0; k < 4; k++) {
		if(i0 + k >= J.nvert) break;
		o += v[k];
		if(J.is_u8) ((uint8_t *)J.values)[(size_t)(i0 + k)*J.N + comp] = (uint8_t)o;
		else ((uint32_t *)J.values)[(size_t)(i0 + k)*J.N + comp] = o;
	}
}

// ------------------------------------------------------------------------------------------------
// K-DEQ. block b -> job block_job[b]; 256 threads x 4 elements.
__global__ __launch_bounds__(256) void k_dequant(const DequantJob *__restrict__ jobs, const uint32_t *__restrict__ block_job, uint32_t nblocks) {
	const uint32_t b = blockIdx.x;
	if(b >= nblocks) return;
	const DequantJob J = jobs[block_job[b]];
	const uint32_t e0 = (b - J.block0)*CHUNK + 4*threadIdx.x;
	if(!J.is_color && J.format != 6u) {
		// GenericAttr<int>::dequantize with an integer or DOUBLE output format (vertex_attribute.h:195-228): "buffer[i] *= q" through a pointer
		// of the OUTPUT type over the first n = nvert*N elements OF THAT TYPE - the first 2n (n) bytes of the int32 array for the 16-bit (8-bit)
		// formats - and DOUBLE widening in place front to back.  Restated as the reference library behaves when compiled for x86-64 (g++ -O2:
		// scalar loops, cvttss2si conversions); pinned by tests/golden/generic_formats.npz, which the reference itself produced.
		const uint32_t n = J.nvert*J.N;
		const float q = J.q;
		if(J.format == 7u) {                                      // DOUBLE: element i >= 1 is computed from the bytes double[i >> 1] left where coords[i] was:
			CRT_GLOBAL const int32_t *vs = as_global((const int32_t *)J.src);   // D(0) = f(v[0]), D(i) = f(half (i & 1) of D(i >> 1)), f(c) = (double)((float)c*q)
			CRT_GLOBAL double *out = as_global((double *)J.buffer);
			const int32_t v0 = vs[0];
#pragma unroll
			for(int k = 0; k < 4; k++) if(e0 + k < n) {
				const uint32_t i = e0 + k;
				double d = (double)((float)v0*q);
				for(int bit = 31 - (int)__builtin_clz(i | 1u); i && bit >= 0; bit--) {
					const uint64_t w = (uint64_t)__double_as_longlong(d);
					const int32_t c = (int32_t)(uint32_t)(((i >> bit) & 1u) ? w >> 32 : w);
					d = (double)((float)c*q);
				}
				out[i] = d;
			}
			return;
		}
		// integer formats: one thread per dword of the int32 array that holds elements with index < n
		CRT_GLOBAL uint32_t *w32 = as_global((uint32_t *)J.buffer);
		const uint32_t per = J.format <= 1u ? 1u : J.format <= 3u ? 2u : 4u;   // elements of the output type per dword
		const uint32_t ndw = (n + per - 1u)/per;
#pragma unroll
		for(int k = 0; k < 4; k++) if(e0 + k < ndw) {
			const uint32_t d = e0 + k;
			uint32_t w = w32[d];
			if(per == 1u) {                                         // ((uint32_t *)buffer)[i] *= q: u32 -> float -> x q -> cvttss2si (64-bit) -> low 32 bits
				const float f = (float)w*q;
				const int64_t t = (f >= -9223372036854775808.0f && f < 9223372036854775808.0f) ? (int64_t)f : (int64_t)0x8000000000000000ull;
				w = (uint32_t)(uint64_t)t;
			} else if(per == 2u) {                                  // ((uint16_t *)buffer)[i] *= q: u16 -> int -> float -> x q -> cvttss2si -> low 16 bits
				uint32_t r = w;
				if(2u*d < n) r = (r & 0xFFFF0000u) | ((uint32_t)f2i_x86((float)(int32_t)(w & 0xFFFFu)*q) & 0xFFFFu);
				if(2u*d + 1u < n) r = (r & 0x0000FFFFu) | ((uint32_t)f2i_x86((float)(int32_t)(w >> 16)*q) << 16);
				w = r;
			} else {                                                // ((char *)buffer)[i] *= q: char is signed on x86
				uint32_t r = w;
#pragma unroll
				for(uint32_t b = 0; b < 4; b++) if(4u*d + b < n) {
					const int32_t c = (int32_t)(int8_t)(uint8_t)(w >> (8u*b));
					r = (r & ~(255u << (8u*b))) | (((uint32_t)f2i_x86((float)c*q) & 255u) << (8u*b));
				}
				w = r;
			}
			w32[d] = w;
		}
		return;
	}
	if(!J.is_color) {                                          // out = (float)v * q, in place (vertex_attribute.h:190-193)
		const uint32_t n = J.nvert*J.N;
		CRT_GLOBAL int32_t *vi = as_global((int32_t *)J.buffer);
		CRT_GLOBAL float *vf = as_global((float *)J.buffer);
		if(J.stride) {                                          // interleaved vertex buffer: values from packed scratch, floats to vertex i's record
			CRT_GLOBAL const int32_t *vs = as_global((const int32_t *)J.src);
			CRT_GLOBAL uint8_t *base = as_global((uint8_t *)J.buffer);
#pragma unroll
			for(int k = 0; k < 4; k++) if(e0 + k < n) {
				const uint32_t e = e0 + k, i = e/J.N, c = e - i*J.N;
				const float f = (float)vs[e];
				*(CRT_GLOBAL float *)(base + (size_t)i*J.stride + 4u*c) = f*J.q;
			}
		} else if(e0 + 3 < n && (((uintptr_t)J.buffer) & 15) == 0) {      // the usual case: one 16-byte load and store per thread
			typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
			typedef float f32x4 __attribute__((ext_vector_type(4)));
			const i32x4 v = *(CRT_GLOBAL const i32x4 *)(vi + e0);
			f32x4 f;
			f.x = (float)v.x*J.q; f.y = (float)v.y*J.q; f.z = (float)v.z*J.q; f.w = (float)v.w*J.q;
			*(CRT_GLOBAL f32x4 *)(vf + e0) = f;
		} else {
#pragma unroll
			for(int k = 0; k < 4; k++) if(e0 + k < n) { const float f = (float)vi[e0 + k]; vf[e0 + k] = f*J.q; }
		}
	} else {                                                   // YCC -> RGB, x qc, u8 wrap (color_attribute.cpp:76-95, point.h:214)
		CRT_GLOBAL const uint8_t *src = as_global(J.src);
		CRT_GLOBAL uint8_t *dst = as_global((uint8_t *)J.buffer);
#pragma unroll
		for(int k = 0; k < 4; k++) {
			const uint32_t i = e0 + k;
			if(i >= J.nvert) break;
			uint32_t col[4] = {0, 0, 0, 255};
			if(J.N == 4 && (((uintptr_t)J.src) & 3) == 0) { const uint32_t x = *(CRT_GLOBAL const uint32_t *)(src + (size_t)i*4); col[0] = x & 255u; col[1] = (x >> 8) & 255u; col[2] = (x >> 16) & 255u; col[3] = x >> 24; }
			else for(uint32_t c = 0; c < J.N && c < 4; c++) col[c] = src[(size_t)i*J.N + c];
			const uint32_t rgb[4] = {(col[2] + col[0]) & 255u, col[0], (col[1] + col[0]) & 255u, col[3]};
			CRT_GLOBAL uint8_t *out = dst + (size_t)i*(J.stride ? J.stride : J.out_components);
			if(J.out_components == 4 && (((uintptr_t)out) & 3) == 0)
				*(CRT_GLOBAL uint32_t *)out = ((rgb[0]*J.qc[0]) & 255u) | ((rgb[1]*J.qc[1]) & 255u) << 8 | ((rgb[2]*J.qc[2]) & 255u) << 16 | ((rgb[3]*J.qc[3]) & 255u) << 24;
			else for(uint32_t c = 0; c < J.out_components && c < 4; c++) out[c] = (uint8_t)(rgb[c]*J.qc[c]);
		}
	}
}

} // namespace corto_hip
