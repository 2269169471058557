// k_tunstall.hip — Tunstall dictionary build (K-TAB) and table-lookup decode (K-TUN) for gfx950.
//
// Replaces, for a whole batch of streams at once:
//   crt::Tunstall::createDecodingTables2   src/tunstall.cpp:125-256
//   crt::Tunstall::decompress              src/tunstall.cpp:430-452
//   InStream::tunstall_decompress framing  src/cstream.cpp:111-128 (framing itself is walked on the host)
//
// K-TAB: one wave per stream.  The dictionary is grown in LDS; the serial "pick the likeliest word,
//        expand it by every symbol" loop runs with the wave's lanes spread over rows (argmax) and
//        over the child bytes (copy), so an expansion is a few LDS round trips instead of n*len
//        byte copies.
// K-TUN: decode = exclusive scan of word lengths + gather.  Every workgroup stages the stream's
//        table in LDS (<= 9 KiB), reads codewords coalesced, scans lengths wave/block-wide and
//        emits the words.
#include "kernels_common.h"

namespace corto_hip {

// ------------------------------------------------------------------------------------------------
// K-TAB
__global__ __launch_bounds__(64) void k_tun_tables(const TunStream *__restrict__ streams, uint32_t nstreams,
                                                    TunTable *__restrict__ tables) {
	const uint32_t s = blockIdx.x;
	if(s >= nstreams) return;
	const TunStream st = streams[s];
	TunTable &T = tables[st.table];
	const uint32_t n = st.nsym;                 // 2..255 (host guarantees)
	const uint32_t lane = threadIdx.x;

	__shared__ uint32_t eprob[TUN_ENTRY_CAP];   // entry e (creation order) lives in FIFO row e % n
	__shared__ uint16_t eoff[TUN_ENTRY_CAP];
	__shared__ uint16_t elen[TUN_ENTRY_CAP];
	__shared__ uint16_t head[256];              // oldest not-yet-expanded entry of each row
	__shared__ uint32_t P[256];                 // probability << 8  (16.16-ish fixed point)
	__shared__ uint32_t pw[256];                // P0^k (successive (a*b)>>16), low-entropy seed only
	__shared__ uint8_t sym[256];
	__shared__ __attribute__((aligned(16))) uint8_t buf[TUN_TABLE_BYTES];

	for(uint32_t i = lane; i < n; i += 64) { sym[i] = st.probs[2*i]; P[i] = (uint32_t)st.probs[2*i + 1] << 8; }
	for(uint32_t i = lane; i < TUN_ENTRY_CAP; i += 64) { eprob[i] = 0; eoff[i] = 0; elen[i] = 0; }
	__syncthreads();

	// how long a run of the likeliest symbol stays likelier than the runner-up (tunstall.cpp:143-151)
	const uint32_t p0 = P[0], p1 = P[1];
	uint32_t count = 2, run = (p0*p0) >> 16;
	const uint32_t max_count = 255u/(n - 1);
	while(run > p1 && count < max_count) { run = (run*p0) >> 16; count++; }

	uint32_t pos, end, nwords;
	if(count >= 16) {                           // low-entropy seed (tunstall.cpp:153-193)
		// byte store: A | for k>=1: A^(count-1) sym_k ; word (row k, col) = the (col+1)-byte suffix ending at k*count
		const uint32_t total = 1 + (n - 1)*count;
		const uint8_t A = sym[0];
		for(uint32_t b = lane; b < total; b += 64) {
			uint8_t v = A;
			if(b > 0) { uint32_t k = (b - 1)/count + 1, j = (b - 1) - (k - 1)*count; if(j == count - 1) v = sym[k]; }
			buf[b] = v;
		}
		if(lane == 0) {                         // P0^col, col = 1..count
			uint32_t v = p0; pw[1] = v;
			for(uint32_t c = 2; c <= count; c++) { v = (v*p0) >> 16; pw[c] = v; }
		}
		__syncthreads();
		for(uint32_t e = lane; e < count*n; e += 64) {
			const uint32_t col = e/n, row = e - col*n;
			if(row == 0) continue;
			eprob[e] = col == 0 ? P[row] : (pw[col]*P[row]) >> 16;
			eoff[e] = (uint16_t)(row*count - col);
			elen[e] = (uint16_t)(col + 1);
		}
		for(uint32_t k = lane; k < n; k += 64) head[k] = (uint16_t)(k == 0 ? (count - 1)*n : k);
		__syncthreads();
		if(lane == 0) { const uint32_t first = (count - 1)*n; eprob[first] = pw[count]; eoff[first] = 0; elen[first] = (uint16_t)count; }
		nwords = 1 + count*(n - 1);
		end = count*n;
		pos = total;
	} else {                                    // one-symbol words (tunstall.cpp:195-205)
		for(uint32_t i = lane; i < n; i += 64) {
			head[i] = (uint16_t)i; eprob[i] = P[i]; eoff[i] = (uint16_t)i; elen[i] = 1; buf[i] = sym[i];
		}
		nwords = n; end = n; pos = n;
	}
	__syncthreads();

	while(nwords < 256) {                       // tunstall.cpp:207-241
		// likeliest FIFO head; first row wins ties; all-zero -> row 0.  key = prob:16 | (0xFFFF - row)
		uint32_t key = 0;
		for(uint32_t r = lane; r < n; r += 64) {
			const uint32_t h = head[r];
			const uint32_t p = h < TUN_ENTRY_CAP ? eprob[h] : 0u;
			const uint32_t k = p ? ((p << 16) | (0xFFFFu - r)) : 0u;
			key = k > key ? k : key;
		}
#pragma unroll
		for(int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(key, d, 64); key = o > key ? o : key; }
		const uint32_t best = (key >> 16) ? 0xFFFFu - (key & 0xFFFFu) : 0u;
		const uint32_t parent = head[best];
		if(parent >= TUN_ENTRY_CAP) break;      // malformed probabilities (reference: out-of-bounds read)
		const uint32_t pp = eprob[parent], po = eoff[parent], pl = elen[parent];
		const bool full = nwords + n > 255;     // dictionary fills up during this expansion: parent stays
		const uint32_t m = full ? 256 - nwords : n;
		const uint32_t tot = m*(pl + 1);
		if(end + m > TUN_ENTRY_CAP || pos + tot > TUN_TABLE_BYTES) break;
		for(uint32_t r = lane; r < m; r += 64) {
			const uint32_t e = end + r;
			eprob[e] = (pp*P[r]) >> 16;
			eoff[e] = (uint16_t)(pos + r*(pl + 1));
			elen[e] = (uint16_t)(pl + 1);
		}
		for(uint32_t b = lane; b < tot; b += 64) {   // child r = parent bytes + sym[r]
			const uint32_t r = b/(pl + 1), j = b - r*(pl + 1);
			buf[pos + b] = j < pl ? buf[po + j] : sym[r];
		}
		__syncthreads();
		if(!full && lane == 0) head[best] = (uint16_t)(parent + n);
		end += m; pos += tot; nwords += n - 1;
		__syncthreads();
	}

	// survivors in creation order -> codes 0..255 (tunstall.cpp:243-253)
	uint32_t w = 0, used = 0;
	for(uint32_t base = 0; base < end; base += 64) {
		const uint32_t e = base + lane;
		const bool alive = e < end && !(head[e % n] > e);
		const uint64_t mask = __ballot(alive);
		const uint32_t rank = w + __popcll(mask & ((1ull << lane) - 1ull));
		if(alive && rank < 256) {
			T.off[rank] = eoff[e]; T.len[rank] = (uint8_t)elen[e];
			const uint32_t u = (uint32_t)eoff[e] + elen[e];
			used = u > used ? u : used;
		}
		w += __popcll(mask);
	}
#pragma unroll
	for(int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(used, d, 64); used = o > used ? o : used; }
	for(uint32_t c = w + lane; c < 256; c += 64) { T.off[c] = 0; T.len[c] = 0; }   // never on valid input
	if(lane == 0) T.used = used;
	const uint32_t ndw = (used + 3) >> 2;
	const uint32_t *src32 = (const uint32_t *)buf;
	uint32_t *dst32 = (uint32_t *)T.bytes;
	for(uint32_t i = lane; i < ndw; i += 64) dst32[i] = src32[i];
}

// ------------------------------------------------------------------------------------------------
// K-TUN.  One workgroup per (stream, chunk).  A stream is cut into chunks of TUN_CHUNK codewords;
// pass A (k_tun_chunk_sums) adds up the decoded length of each chunk, a device-wide exclusive scan
// turns that into each chunk's output offset, pass B (k_tun_decode) decodes.  Streams of a single
// chunk (the .crt case: a few KiB) skip pass A: their offset is 0.
constexpr uint32_t TUN_TILE = 1024;              // codewords per inner tile (256 threads x 4)

struct TunLds {
	uint16_t off[256];
	uint8_t len[256];
	__attribute__((aligned(16))) uint8_t bytes[TUN_TABLE_BYTES];
	uint32_t scan[4];
};

__device__ __forceinline__ void tun_load_table(TunLds &L, const TunTable &T, uint32_t used) {
	const uint32_t t = threadIdx.x;
	L.off[t] = T.off[t];
	L.len[t] = T.len[t];
	const uint32_t ndw = (used + 3) >> 2;
	const uint32_t *src32 = (const uint32_t *)T.bytes;
	uint32_t *dst32 = (uint32_t *)L.bytes;
	for(uint32_t i = t; i < ndw; i += 256) dst32[i] = src32[i];
}

// pass A: per-chunk decoded byte count
__global__ __launch_bounds__(256) void k_tun_chunk_sums(const TunStream *__restrict__ streams, const uint32_t *__restrict__ chunk_stream,
                                                        uint32_t nchunks, const TunTable *__restrict__ tables,
                                                        uint32_t chunk_codes, uint64_t *__restrict__ chunk_out, uint32_t chunk_base) {
	const uint32_t c = blockIdx.x + chunk_base;
	if(blockIdx.x >= nchunks) return;
	const TunStream st = streams[chunk_stream[c]];
	const TunTable &T = tables[st.table];
	__shared__ uint8_t len[256];
	__shared__ uint32_t red[4];
	len[threadIdx.x] = T.len[threadIdx.x];
	__syncthreads();
	const uint32_t first = (c - st.chunk0)*chunk_codes;
	const uint32_t last = min(first + chunk_codes, st.csize);
	uint32_t sum = 0;
	for(uint32_t j = first + threadIdx.x; j < last; j += 256) sum += len[st.src[j]];
#pragma unroll
	for(int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
	if(lane_id() == 0) red[wave_id()] = sum;
	__syncthreads();
	if(threadIdx.x == 0) chunk_out[c] = (uint64_t)red[0] + red[1] + red[2] + red[3];
}

// pass B: decode one chunk.  chunk_out[c] (after the scan) - chunk_out[st.chunk0] = output offset.
__global__ __launch_bounds__(256) void k_tun_decode(const TunStream *__restrict__ streams, const uint32_t *__restrict__ chunk_stream,
                                                    uint32_t nchunks, const TunTable *__restrict__ tables,
                                                    uint32_t chunk_codes, const uint64_t *__restrict__ chunk_out, uint32_t chunk_base) {
	const uint32_t c = blockIdx.x + chunk_base;
	if(blockIdx.x >= nchunks) return;
	const TunStream st = streams[chunk_stream[c]];
	const TunTable &T = tables[st.table];
	__shared__ TunLds L;
	tun_load_table(L, T, T.used);
	__syncthreads();

	const uint32_t first = (c - st.chunk0)*chunk_codes;
	const uint32_t last = min(first + chunk_codes, st.csize);
	uint64_t base = st.nchunks > 1 ? chunk_out[c] - chunk_out[st.chunk0] : 0;
	const uint64_t size = st.size;
	const uint8_t *__restrict__ src = st.src;
	uint8_t *__restrict__ dst = st.dst;

	for(uint32_t tile = first; tile < last; tile += TUN_TILE) {
		const uint32_t j0 = tile + 4*threadIdx.x;
		uint32_t code[4], l[4], sum = 0;
#pragma unroll
		for(int k = 0; k < 4; k++) {
			const bool ok = j0 + k < last;
			code[k] = ok ? src[j0 + k] : 0u;
			l[k] = ok ? L.len[code[k]] : 0u;
			sum += l[k];
		}
		uint32_t total;
		uint64_t o = base + block256_exclusive_scan<uint32_t>(sum, L.scan, &total);
#pragma unroll
		for(int k = 0; k < 4; k++) {
			if(j0 + k < last) {
				// every word is copied whole; the stream's last codeword emits whatever is left (tunstall.cpp:447-451)
				uint32_t nb = l[k];
				const uint32_t wo = L.off[code[k]];
				if(j0 + k + 1 == st.csize) nb = o < size ? (uint32_t)min((uint64_t)(TUN_TABLE_BYTES - wo), size - o) : 0u;
				else if(o + nb > size) nb = o < size ? (uint32_t)(size - o) : 0u;
				for(uint32_t b = 0; b < nb; b++) dst[o + b] = L.bytes[wo + b];
				o += l[k];
			}
		}
		base += total;
	}
}

// memset path: single-symbol streams (tunstall.cpp:433-436)
__global__ __launch_bounds__(256) void k_fill(const FillJob *__restrict__ jobs, uint32_t njobs) {
	const uint32_t j = blockIdx.x;
	if(j >= njobs) return;
	const FillJob f = jobs[j];
	for(uint32_t i = threadIdx.x; i < f.size; i += 256) f.dst[i] = (uint8_t)f.value;
}

} // namespace corto_hip
