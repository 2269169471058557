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
#include <type_traits>

#include "kernels_common.h"
#include "kernels.h"

namespace corto_hip {

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------
// K-TAB
#include "tun_tables.h"

__global__ __launch_bounds__(64) void k_tun_tables(const TunStream *__restrict__ streams, uint32_t nstreams,
                                                    TunTable *__restrict__ tables) {
	const uint32_t s = blockIdx.x;
	if(s >= nstreams) return;
	const TunStream st = streams[s];
	extern __shared__ __attribute__((aligned(16))) uint8_t big_words[];        // TUN_TABLE_BYTES when the launch has an alphabet of more than 64 symbols, else nothing
	tun_tables_body<true>(st, &tables[st.table], nullptr, nullptr, nullptr, big_words);
#ifdef CORTO_TUN_STAMPS
	if(threadIdx.x == 0 && blockIdx.x < 4096) { g_tun_stamps[blockIdx.x*8 + 4] = g_tun_stamps[blockIdx.x*8 + 3]; g_tun_stamps[blockIdx.x*8 + 5] = st.nsym; g_tun_stamps[blockIdx.x*8 + 6] = st.probs[1]; g_tun_stamps[blockIdx.x*8 + 7] = st.probs[3]; }
#endif
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

__device__ __forceinline__ void tun_load_table(TunLds &L, const TunTable &T, uint32_t used) {      // 16-byte vectors: 48 of offsets / lengths, used/16 of words
	const uint32_t t = threadIdx.x;
	static_assert(offsetof(TunTable, len) == 512 && offsetof(TunTable, bytes) % 16 == 0 && sizeof(TunTable) % 16 == 0 && offsetof(TunLds, len) == 512 && offsetof(TunLds, bytes) % 16 == 0, "vector copy of a TunTable");
	CRT_GLOBAL const u32x4_t *g = (CRT_GLOBAL const u32x4_t *)as_global((const uint8_t *)&T);
	CRT_LDS u32x4_t *l = (CRT_LDS u32x4_t *)as_lds((uint8_t *)&L);
	if(t < 48) l[t] = g[t];
	const uint32_t nv = (min(used, TUN_TABLE_BYTES) + 15u) >> 4;
	CRT_GLOBAL const u32x4_t *gb = (CRT_GLOBAL const u32x4_t *)as_global(T.bytes);
	CRT_LDS u32x4_t *lb = (CRT_LDS u32x4_t *)as_lds(L.bytes);
	for(uint32_t i = t; i < nv; i += 256) lb[i] = gb[i];
}

// pass A: decoded byte count of every quarter chunk (one wave's share of a chunk in pass B): chunk_out[4*c + w]
__global__ __launch_bounds__(256) void k_tun_chunk_sums(const TunStream *__restrict__ streams, const uint32_t *__restrict__ chunk_stream,
                                                        uint32_t nchunks, const TunTable *__restrict__ tables,
                                                        uint64_t *__restrict__ chunk_out, uint32_t chunk_base) {
	const uint32_t c = blockIdx.x + chunk_base;
	if(blockIdx.x >= nchunks) return;
	const TunStream st = streams[chunk_stream[c]];
	const TunTable &T = tables[st.table];
	__shared__ uint8_t len[256];
	len[threadIdx.x] = T.len[threadIdx.x];
	__syncthreads();
	const uint32_t chunk_codes = st.chunk_codes, quarter = chunk_codes/4, w = wave_id(), lane = lane_id();
	const uint32_t cfirst = (c - st.chunk0)*chunk_codes;
	const uint32_t first = min(cfirst + w*quarter, st.csize), last = min(first + quarter, min(cfirst + chunk_codes, st.csize));
	CRT_GLOBAL const uint8_t *src = as_global(st.src);
	uint32_t sum = 0;
	const uint32_t head = min((uint32_t)((0u - (uint32_t)(uintptr_t)(st.src + first)) & 3u), last - first);   // aligned dword body, byte head/tail
	if(lane < head) sum += len[src[first + lane]];
	const uint32_t body0 = first + head, ndw = (last - body0) >> 2;
	CRT_GLOBAL const uint32_t *src32 = (CRT_GLOBAL const uint32_t *)(src + body0);
	for(uint32_t i = lane; i < ndw; i += 64) {
		const uint32_t x = src32[i];
		sum += (uint32_t)len[x & 255u] + len[(x >> 8) & 255u] + len[(x >> 16) & 255u] + len[x >> 24];
	}
	const uint32_t tail0 = body0 + ndw*4;
	if(tail0 + lane < last) sum += len[src[tail0 + lane]];
	sum = wave_inclusive_scan_u32(sum);
	if(lane == 63) chunk_out[(size_t)c*4 + w] = sum;
}

// Emit one thread's run (its up-to-4 words back to back) at byte pointer d.  Table words are read as ALIGNED dwords
// (v_alignbyte shifts out the word's byte phase) into a 64-bit byte FIFO; the run is written as <=3 head bytes (up to the next 4-aligned destination address),
// aligned dwords, and <=3 tail bytes: no unaligned accesses (on gfx950 those stall the LDS pipeline ~70 % of the time, PMC
// SQ_LDS_UNALIGNED_STALL) and no overlap with the neighbouring threads' runs.
template <typename DPtr>
__device__ __forceinline__ void tun_emit_run(DPtr d, uint32_t daddr, CRT_LDS const uint32_t *tab32, const uint32_t (&wo)[4], const uint32_t (&nb)[4]) {
	typedef typename std::conditional<std::is_same<DPtr, CRT_LDS uint8_t *>::value, CRT_LDS uint32_t, CRT_GLOBAL uint32_t>::type dword_t;
	// The FIFO starts at the 4-aligned address at or below d with m = d & 3 placeholder bytes (they belong to the previous
	// thread's run), so that every dword leaves on an aligned address; the first dword is then written byte-wise without
	// its m placeholder bytes, and the last partial dword byte-wise too.
	const uint32_t m = daddr & 3u;
	d -= m;
	uint32_t lo = 0, na = m;                                   // lo holds na (< 4) pending bytes
	bool first = m != 0;
#pragma unroll
	for(int k = 0; k < 4; k++) {
		if(nb[k] == 0) continue;
		CRT_LDS const uint32_t *src = tab32 + (wo[k] >> 2);
		const uint32_t sh = wo[k] & 3u;                        // words keep the reference's shared-suffix layout: any byte offset
		uint32_t prev = *src++;
		for(uint32_t i = 0; i < nb[k]; i += 4) {
			const uint32_t next = *src++;
			uint32_t dw = __builtin_amdgcn_alignbyte(next, prev, sh);
			prev = next;
			const uint32_t vb = min(4u, nb[k] - i);
			if(vb < 4) dw &= (1u << (8*vb)) - 1u;
			const uint32_t full = lo | (dw << (8*na));          // na < 4
			if(na + vb >= 4) {
				if(first) { for(uint32_t b = m; b < 4; b++) d[b] = (uint8_t)(full >> (8*b)); first = false; }
				else *(dword_t *)d = full;
				d += 4;
				lo = na ? dw >> (32 - 8*na) : 0u;
				na = na + vb - 4;
			} else { lo = full; na += vb; }
		}
	}
	for(uint32_t b = first ? m : 0u; b < na; b++) d[b] = (uint8_t)(lo >> (8*b));
}

// shared front half of a tile: codes -> lengths -> block scan -> clipped byte counts
struct TunTile { uint32_t wo[4], nb[4], l[4]; uint32_t total; uint64_t o; };

__device__ __forceinline__ void tun_tile_prepare(TunTile &t, const TunStream &st, CRT_GLOBAL const uint8_t *src, uint32_t tile, uint32_t last,
                                                 uint64_t base, CRT_LDS const uint8_t *len8, CRT_LDS const uint16_t *off16, uint32_t *scan) {
	const uint32_t j0 = tile + 4*threadIdx.x;
	uint32_t code[4], sum = 0;
#pragma unroll
	for(int k = 0; k < 4; k++) {
		const bool ok = j0 + k < last;
		code[k] = ok ? (uint32_t)src[j0 + k] : 0u;
		t.l[k] = ok ? (uint32_t)len8[code[k]] : 0u;
		sum += t.l[k];
	}
	t.o = base + block256_exclusive_scan<uint32_t>(sum, scan, &t.total);
	const uint64_t size = st.size;
	uint64_t oo = t.o;
#pragma unroll
	for(int k = 0; k < 4; k++) {                       // every word whole; the stream's last codeword emits what is left (tunstall.cpp:447-451)
		uint32_t nb = t.l[k];
		t.wo[k] = off16[code[k]];
		if(j0 + k < last) {
			if(j0 + k + 1 == st.csize) nb = oo < size ? (uint32_t)min((uint64_t)(TUN_TABLE_BYTES - t.wo[k]), size - oo) : 0u;
			else if(oo + nb > size) nb = oo < size ? (uint32_t)(size - oo) : 0u;
		} else nb = 0;
		t.nb[k] = nb;
		oo += t.l[k];
	}
}

// the decode of one short stream by one wave from a dictionary in LDS (offsets, lengths, word bytes): four codewords per lane and step
__device__ __forceinline__ void tun_stream_decode(const TunStream &st, const uint16_t *loff, const uint8_t *llen, const uint8_t *words) {
	const uint32_t lane = lane_id(), csize = st.csize;
	const uint64_t size = st.size;
	CRT_GLOBAL const uint8_t *src = as_global(st.src);
	CRT_GLOBAL uint8_t *gdst = as_global(st.dst);
	CRT_LDS const uint8_t *len8 = as_lds(llen);
	CRT_LDS const uint16_t *off16 = as_lds(loff);
	CRT_LDS const uint32_t *tab32 = (CRT_LDS const uint32_t *)as_lds(words);
	uint64_t base = 0;
	for(uint32_t tile = 0; tile < csize; tile += 256) {
		const uint32_t j0 = tile + 4*lane;
		// the lane's (up to) four codewords as ONE load: an unaligned dword at j0, or - the stream's last, partial group - the stream's last
		// dword shifted down; then the four lengths and the four offsets, every read unconditional and pinned.  (Written as four
		// `ok ? src[j0 + k] : 0` the compiler makes four exec-masked byte loads, each with its own wait and a dependent LDS read behind
		// it: eight serial round trips per 256 codewords - most of this kernel's time on a stream of a few hundred.)
		const uint32_t r = j0 < csize ? min(csize - j0, 4u) : 0u;            // valid codewords of this lane
		uint32_t raw;
		if(csize >= 4) {                                                    // (uniform)
			uint32_t dw = *(CRT_GLOBAL const uint32_t *)(src + (r == 4u ? j0 : r ? csize - 4u : 0u));
			asm volatile("" : "+v"(dw));
			raw = r == 4u ? dw : r ? dw >> (8u*(4u - r)) : 0u;
		} else {
			raw = 0;
			for(uint32_t k = 0; k < r; k++) raw |= (uint32_t)src[j0 + k] << (8u*k);
		}
		uint32_t code[4], l[4], sum = 0;
#pragma unroll
		for(int k = 0; k < 4; k++) code[k] = (raw >> (8*k)) & 255u;          // (0 beyond the stream's end: a valid table index)
#pragma unroll
		for(int k = 0; k < 4; k++) l[k] = (uint32_t)len8[code[k]];
		asm volatile("" : "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3]));
#pragma unroll
		for(int k = 0; k < 4; k++) { l[k] = (uint32_t)k < r ? l[k] : 0u; sum += l[k]; }
		const uint32_t inc = wave_inclusive_scan_u32(sum);
		const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
		uint64_t oo = base + inc - sum;
		const uint64_t o_run = oo;
		uint32_t wo[4], nb[4];
#pragma unroll
		for(int k = 0; k < 4; k++) wo[k] = off16[code[k]];
		asm volatile("" : "+v"(wo[0]), "+v"(wo[1]), "+v"(wo[2]), "+v"(wo[3]));
#pragma unroll
		for(int k = 0; k < 4; k++) {                       // every word whole; the stream's last codeword emits what is left (tunstall.cpp:447-451)
			uint32_t n_ = l[k];
			if(j0 + k < csize) {
				if(j0 + k + 1 == csize) n_ = oo < size ? (uint32_t)min((uint64_t)(TUN_TABLE_BYTES - wo[k]), size - oo) : 0u;
				else if(oo + n_ > size) n_ = oo < size ? (uint32_t)(size - oo) : 0u;
			} else n_ = 0;
			nb[k] = n_;
			oo += l[k];
		}
		CRT_GLOBAL uint8_t *d = gdst + o_run;
		tun_emit_run(d, (uint32_t)(uintptr_t)d, tab32, wo, nb);
		base += total;
	}
}

// K-STREAM: dictionary AND decode of one short stream (the .crt case: a few hundred codewords, a few KiB of symbols) by ONE wave.
// A batch of 256 blobs has 2 304 such streams; as two kernels (k_tun_tables, then k_tun_decode with 256 threads per stream) every
// dictionary made a 9 KB round trip through HBM, the decode workgroups sat behind the slowest dictionary of the launch, and a
// step had two more launches on each of its two chains.  Here the dictionary stays in the LDS of the wave that built it (the build is
// ~45 us of dependent steps, the decode of a 2 112-symbol stream a few more), and the symbols are written straight to HBM: four
// codewords per lane and step, lengths scanned across the wave, each lane's run emitted through the aligned byte FIFO.
__global__ __launch_bounds__(64) void k_tun_stream(const TunStream *__restrict__ streams, uint32_t nstreams) {
	const uint32_t s = blockIdx.x;
	if(s >= nstreams) return;
	const TunStream st = streams[s];
	__shared__ uint16_t loff[256];
	__shared__ uint8_t llen[256];
	tun_tables_body(st, nullptr, loff, llen);
	__syncthreads();
	tun_stream_decode(st, loff, llen, g_tun_words);
#ifdef CORTO_TUN_STAMPS
	TUN_STAMP(4);
	if(threadIdx.x == 0 && blockIdx.x < 4096) { g_tun_stamps[blockIdx.x*8 + 5] = st.nsym; g_tun_stamps[blockIdx.x*8 + 6] = st.csize; g_tun_stamps[blockIdx.x*8 + 7] = st.size; }
#endif
}

// K-STREAM, shared dictionaries: the streams of a batch repeat one another's probability tables - an alphabet of two symbols has
// ~127 possible tables - and the dictionary is a function of the table alone, so the planner (batch.cpp) has every DISTINCT table
// built once (k_tun_tables, into its TunTable in HBM) and the streams only load it: 768 bytes of offsets / lengths and the `used` bytes
// of words, from L2 (round 2 did that with one wave per stream, k_tun_stream_shared; round 3 by groups of one dictionary:)

// ... and the streams that share a dictionary share its copy in LDS too (round 3): the planner sorts a launch's streams by dictionary
// and hands groups of up to TUN_GROUP_MAX of them, all of ONE dictionary, to one workgroup of four waves - the dictionary is loaded
// once by 256 threads and every wave decodes its streams from it, one after the other.  What this buys is LDS.time, which is what
// bounds the pipelined decode: one wave per stream held 10 KB for its ~8 us each, 2 304 times a batch - more than any
// other kernel of the path (1.1 of 4.0 GB.us in the steady-state trace); a group of eight holds the same 10 KB for ~20 us.
__global__ __launch_bounds__(256) void k_tun_stream_grouped(const TunStream *__restrict__ streams, const uint32_t *__restrict__ ids, const TunGroup *__restrict__ groups,
                                                            uint32_t ngroups, const TunTable *__restrict__ tables) {
	if(blockIdx.x >= ngroups) return;
	const TunGroup G = groups[blockIdx.x];
	__shared__ uint16_t loff[256];
	__shared__ uint8_t llen[256];
	__shared__ __attribute__((aligned(16))) uint8_t words[TUN_TABLE_BYTES];
	const uint32_t t = threadIdx.x;
	const TunTable &T = tables[streams[ids[G.first]].dict];
	{
		CRT_GLOBAL const u32x4_t *o4 = (CRT_GLOBAL const u32x4_t *)as_global(T.off);
		CRT_GLOBAL const u32x4_t *w4 = (CRT_GLOBAL const u32x4_t *)as_global(T.bytes);
		u32x4_t hv = o4[t < 48 ? t : 47u], wv = w4[t];                          // (the table has 576 vectors of words: t < 256 is inside)
		asm volatile("" : "+v"(hv), "+v"(wv));
		if(t < 32) ((CRT_LDS u32x4_t *)as_lds(loff))[t] = hv;
		else if(t < 48) ((CRT_LDS u32x4_t *)as_lds(llen))[t - 32] = hv;
		((CRT_LDS u32x4_t *)as_lds(words))[t] = wv;
		const uint32_t nv = (min(T.used, TUN_TABLE_BYTES) + 15u) >> 4;
		for(uint32_t i = t + 256; i < nv; i += 256) ((CRT_LDS u32x4_t *)as_lds(words))[i] = w4[i];
	}
	__syncthreads();
	for(uint32_t k = wave_id(); k < G.count; k += 4) tun_stream_decode(streams[ids[G.first + k]], loff, llen, words);
}

// pass B, short streams (the .crt case: one chunk, a few KiB): small LDS footprint so that it can run next to the
// LDS-hungry topology kernel; runs are written straight to HBM.
__global__ __launch_bounds__(256) void k_tun_decode(const TunStream *__restrict__ streams, const uint32_t *__restrict__ chunk_stream,
                                                    uint32_t nchunks, const TunTable *__restrict__ tables,
                                                    const uint64_t *__restrict__ chunk_out, uint32_t chunk_base) {
	const uint32_t c = blockIdx.x + chunk_base;
	if(blockIdx.x >= nchunks) return;
	const TunStream st = streams[chunk_stream[c]];
	const TunTable &T = tables[st.table];
	__shared__ TunLds L;
	tun_load_table(L, T, T.used);
	__syncthreads();
	const uint32_t first = (c - st.chunk0)*st.chunk_codes;
	const uint32_t last = min(first + st.chunk_codes, st.csize);
	uint64_t base = st.nchunks > 1 ? chunk_out[c] - chunk_out[st.chunk0] : 0;
	CRT_GLOBAL const uint8_t *src = as_global(st.src);
	CRT_GLOBAL uint8_t *gdst = as_global(st.dst);
	for(uint32_t tile = first; tile < last; tile += TUN_TILE) {
		TunTile t;
		tun_tile_prepare(t, st, src, tile, last, base, as_lds(L.len), as_lds(L.off), L.scan);
		CRT_GLOBAL uint8_t *d = gdst + t.o;
		tun_emit_run(d, (uint32_t)(uintptr_t)d, (CRT_LDS const uint32_t *)as_lds(L.bytes), t.wo, t.nb);
		base += t.total;
	}
}

// pass B, long streams.  WAVE-AUTONOMOUS: a workgroup shares the stream's table in LDS, but each of its four waves decodes
// its own quarter of the chunk (output offset from pass A's per-quarter sums) with no workgroup barrier in the loop - in
// a first version two block-wide scans per tile (barriers) were half of the kernel time.  Per step of 64*cpl codewords
// (cpl per lane, from the stream's mean word length): lengths -> DPP wave scan -> compose the bytes in the wave's own
// LDS window, laid out at the destination's 16-byte phase -> flush with 16-byte stores, one kilobyte per instruction.
//   compose, branch-free: every word has a zero-padded 16-byte copy (T16); a lane reads W = 1, 2 or 4 dwords of it (W from
//            the dictionary's longest word), moves them to the word's byte phase with W+1 v_alignbyte_b32 and ORs the
//            W+1 dwords into the zeroed window with LDS atomics - neighbouring words touch disjoint bytes of a shared
//            dword, so OR composes them with no ordering at all.
//   long words (> 16 bytes): queued per wave; their further 16-byte pieces are ORed in by up to 64 lanes at once (inline
//            in the compose loop, one long word in any lane would make the whole wave walk that path).
//   flush:   whole 16-byte vectors only; the bytes after the last whole vector stay in the window and become the head of
//            the next step's window, so only a wave's first and last vector are written bytewise.
// LDS ordering inside one wave is program order, so the window needs no barrier, only the s_waitcnt the compiler places.
// A stream's clipped last step, and steps whose bytes exceed the window, take the general byte-FIFO path.
constexpr uint32_t TUN_SUB = 512;                // most codewords per wave per step
constexpr uint32_t TUN_LONGQ = 64;               // per-wave queue of long words
static_assert(TUN_SUB == 64*8 && TUN_CHUNK_CODES % (4*TUN_SUB) == 0, "a wave's quarter chunk is whole steps of 64*cpl codewords (tun_pick_geometry)");

// OR the W dwords x[] into the window so that their first byte lands on window byte p.  P is the LDS byte address of
// window byte p - 1 (the window buffer is 16-byte aligned), N = ~P: the dwords are moved up by p & 3 bytes as
// {x[i], x[i-1]} >> 8*((-p) & 3) - v_alignbyte_b32 takes the shift from the low two bits of N - which for p & 3 == 0
// yields them one dword late; addressing from (p - 1) & ~3 instead of p & ~3 absorbs exactly that.
template <int W> __device__ __forceinline__ void tun_or(uint32_t P, uint32_t N, const uint32_t *x) {
	CRT_LDS uint32_t *o = lds_at<uint32_t>(P & ~3u);
	uint32_t prev = 0;
#pragma unroll
	for(int i = 0; i < W; i++) { atomicOr((uint32_t *)(o + i), __builtin_amdgcn_alignbyte(x[i], prev, N)); prev = x[i]; }
	const uint32_t top = __builtin_amdgcn_alignbyte(0u, prev, N);          // what spills into the next dword: nothing for most short words,
	if(W > 1 || top) atomicOr((uint32_t *)(o + W), top);                   // and the LDS is the busiest unit of the short-word kernel
}

// Bytes 16.. of the queued long words (entry = window position of the word | code << 16): the owning lane streams the
// table's dwords to the window's dwords, {T[i+1], T[i]} >> 8*shift, three instructions per dword.  The dword the stream
// starts in is shared with the word's own bytes 12..15, already there - ORing them again changes nothing, so only the
// last dword needs a mask (the table bytes behind a word belong to other words).
__device__ __forceinline__ void tun_drain_long(uint32_t win0, CRT_LDS const uint32_t *longq, uint32_t n, CRT_LDS const uint16_t *off16,
                                               CRT_LDS const uint8_t *len8, CRT_LDS const uint32_t *tab32) {
	if(lane_id() < n) {
		const uint32_t e = longq[lane_id()], cd = e >> 16;
		const uint32_t d0 = (e & 0xffffu) + 16u;                             // window byte position of the word's byte 16
		const uint32_t sp = (uint32_t)off16[cd] + 16u - (d0 & 3u);           // table byte that lands on the first dword's byte 0
		const uint32_t nbytes = (uint32_t)len8[cd] - 16u + (d0 & 3u);        // bytes from there to the word's end
		CRT_LDS const uint32_t *t = tab32 + (sp >> 2);
		CRT_LDS uint32_t *o = lds_at<uint32_t>(win0 + (d0 & ~3u));
		const uint32_t nd = (nbytes + 3u) >> 2;                               // dwords to OR; the last one masked
		uint32_t lo = t[0], i = 0;
		if(nd > 4) {                                                           // reads of the next four dwords go out before this four's ORs
			uint32_t t1 = t[1], t2 = t[2], t3 = t[3], t4 = t[4];
			for(; i + 4 < nd; i += 4) {
				const uint32_t a0 = __builtin_amdgcn_alignbyte(t1, lo, sp), a1 = __builtin_amdgcn_alignbyte(t2, t1, sp);
				const uint32_t a2 = __builtin_amdgcn_alignbyte(t3, t2, sp), a3 = __builtin_amdgcn_alignbyte(t4, t3, sp);
				lo = t4;
				if(i + 8 < nd) { t1 = t[i + 5]; t2 = t[i + 6]; t3 = t[i + 7]; t4 = t[i + 8]; }
				atomicOr((uint32_t *)(o + i), a0); atomicOr((uint32_t *)(o + i + 1), a1);
				atomicOr((uint32_t *)(o + i + 2), a2); atomicOr((uint32_t *)(o + i + 3), a3);
			}
		}
		for(; i < nd; i++) {
			const uint32_t hi = t[i + 1];
			uint32_t v = __builtin_amdgcn_alignbyte(hi, lo, sp);
			if(i + 1 == nd && (nbytes & 3u)) v &= (1u << (8*(nbytes & 3u))) - 1u;
			atomicOr((uint32_t *)(o + i), v);
			lo = hi;
		}
	}
}


// Between pass A and the decode (the default for long streams): the quarter sums of ONE stream turned into that stream's quarter offsets
// by one workgroup - streams are independent, so there is no device-wide scan (k_scan_u64: one workgroup over every chunk of every
// stream, 71 us on the scaled run against 20 here, launch included) and no dependency between workgroups.  Measured against it on the
// scaled run (adding-up + offsets, us): single pass inside the decode kernel (below) 144; a lean adding-up kernel with a wave-wide wait-free
// look-back 95; the last chunk of a stream scanning it from inside the adding-up kernel ("last one out", a counter per stream) 125 - its
// agent-scope stores and counter cost more than this launch; sums 51 + this kernel 20.
__global__ __launch_bounds__(256) void k_tun_stream_scan(const TunStream *__restrict__ streams, uint32_t nstreams, uint64_t *__restrict__ chunk_out) {
	const uint32_t s = blockIdx.x;
	if(s >= nstreams) return;
	const TunStream st = streams[s];
	CRT_GLOBAL uint64_t *p = as_global(chunk_out) + (size_t)st.chunk0*4;
	const uint32_t n = st.nchunks*4u, tid = threadIdx.x, w = wave_id(), lane = lane_id();
	__shared__ uint32_t wsum[4];
	uint64_t carry = 0;
	constexpr uint32_t PER = 8;                                           // values per thread and pass: eight loads in flight, one block scan per 2 048 values
	for(uint32_t base = 0; base < n; base += 256*PER) {                   // (a quarter chunk decodes to < 2^21 bytes: 2 048 of them fit 32 bits)
		const uint32_t i0 = base + tid*PER;
		uint32_t v[PER], tot = 0;
#pragma unroll
		for(uint32_t k = 0; k < PER; k++) v[k] = i0 + k < n ? (uint32_t)p[i0 + k] : 0u;
#pragma unroll
		for(uint32_t k = 0; k < PER; k++) { const uint32_t x = v[k]; v[k] = tot; tot += x; }   // exclusive inside the thread
		const uint32_t incl = wave_inclusive_scan_u32(tot);
		if(lane == 63) wsum[w] = incl;
		__syncthreads();
		const uint32_t w0 = wsum[0], w1 = wsum[1], w2 = wsum[2], w3 = wsum[3];
		const uint64_t off = carry + (w > 0 ? w0 : 0u) + (w > 1 ? w1 : 0u) + (w > 2 ? w2 : 0u) + (incl - tot);
#pragma unroll
		for(uint32_t k = 0; k < PER; k++) if(i0 + k < n) p[i0 + k] = off + v[k];
		carry += (uint64_t)w0 + w1 + w2 + w3;
		__syncthreads();
	}
}

// (Rounds 2-3 also carried a SINGLE-PASS form - every decode workgroup adding up its own chunk and finding its offset by decoupled
// look-back over its predecessors' state words - as a switch: it measured 0.674 ms against 0.57-0.59 for sums + per-stream scan + decode,
// profiles/EXPERIMENTS.md 3.2 has the table, and it was removed in round 4.  chain_lookback itself lives on in K-BIT's chunked kernel.)
// the decoded bytes leave through non-temporal stores: six bytes are written for every byte read, and as ordinary stores they
// pushed the chunk's codewords out of the XCD's L2 between the look-back's adding-up pass and the decode pass
#ifndef TUN_FLUSH_PLAIN
#define TUN_FLUSH_STORE(v, p) __builtin_nontemporal_store((v), (p))
#else
#define TUN_FLUSH_STORE(v, p) (*(p) = (v))
#endif
template <int W, int CPL> constexpr uint32_t tun_win_bytes() { return W == 1 ? 2048 + 64 : W == 2 ? 4096 + 64 : 6*1024 - 64; }
constexpr uint32_t tun_staged_lds(uint32_t win) { return 4*(win + 64); }                // dynamic LDS: the four waves' windows

template <int W, int CPL>
__device__ __forceinline__ void tun_staged_body(const TunStream &st, const TunTable &T, uint32_t c, uint64_t *chunk_out, uint32_t sums_only,
                                                TunLds &L, uint32_t *t16, uint32_t (*longbuf)[TUN_LONGQ], uint32_t *winbuf) {
	constexpr uint32_t TUN_WIN = tun_win_bytes<W, CPL>();
	const uint32_t tid = threadIdx.x, w = wave_id(), lane = lane_id();
	tun_load_table(L, T, T.used);
	for(uint32_t i = tid; i < 4*(TUN_WIN + 64)/16; i += 256) ((CRT_LDS u32x4_t *)as_lds(winbuf))[i] = u32x4_t{0, 0, 0, 0};
	__syncthreads();
	const uint32_t mylen = L.len[tid];
	constexpr uint32_t width = W;
	{	// zero-padded copy of every word, W dwords per entry (a compact table spreads over more LDS banks)
		const uint32_t wo = L.off[tid], wl = min(mylen, 16u);
		uint32_t d[4] = {0, 0, 0, 0};
		for(uint32_t b = 0; b < wl; b++) d[b >> 2] |= (uint32_t)L.bytes[wo + b] << (8*(b & 3));
		CRT_LDS uint32_t *e = (CRT_LDS uint32_t *)as_lds(t16) + (width == 1 ? 2u : width)*tid;
		if(width == 4) *(CRT_LDS u32x4_t *)e = u32x4_t{d[0], d[1], d[2], d[3]};
		else if(width == 2) *(CRT_LDS u32x2_t *)e = u32x2_t{d[0], d[1]};
		else *(CRT_LDS u32x2_t *)e = u32x2_t{d[0], mylen};                  // one-dword words carry their length: one 8-byte read serves both
	}
	__syncthreads();
	const uint32_t chunk_codes = st.chunk_codes, quarter = chunk_codes/4;
	const uint32_t cfirst = (c - st.chunk0)*chunk_codes;
	const uint32_t first = min(cfirst + w*quarter, st.csize), last = min(first + quarter, min(cfirst + chunk_codes, st.csize));
	const uint64_t size = st.size;
	const uint32_t csize = st.csize;
	CRT_GLOBAL const uint8_t *src = as_global(st.src);
	uint64_t base;
	if(sums_only) {                                          // k_tun_chunk_sums ran before: the stream's quarter SUMS, not yet offsets - add up
		CRT_GLOBAL const uint64_t *qs = as_global(chunk_out) + (size_t)st.chunk0*4;   // the ones in front of this wave's quarter (<= 1 024 of them: the
		const uint32_t nq = (c - st.chunk0)*4u + w;                          // planner sends longer streams through k_tun_stream_scan)
		uint64_t acc = 0;
		for(uint32_t i = lane; i < nq; i += 64) acc += qs[i];
#pragma unroll
		for(int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
		base = acc;
	} else base = chunk_out[(size_t)c*4 + w] - chunk_out[(size_t)st.chunk0*4];   // offsets: k_tun_chunk_sums + a scan ran before
	CRT_GLOBAL uint8_t *gdst = as_global(st.dst);
	CRT_LDS uint8_t *wb = (CRT_LDS uint8_t *)as_lds(winbuf) + w*(TUN_WIN + 64);   // window byte i lives at wb[16 + i]
	CRT_LDS u32x4_t *win = (CRT_LDS u32x4_t *)(wb + 16);
	const uint32_t win0 = (uint32_t)(uintptr_t)win;                      // LDS byte address of window byte 0
	CRT_LDS uint32_t *longq = as_lds(&longbuf[W == 4 ? w : 0][0]);   // (never touched unless W == 4)
	CRT_LDS const uint32_t *t16l = (CRT_LDS const uint32_t *)as_lds(t16);
	CRT_LDS const uint8_t *len8 = as_lds(L.len);
	CRT_LDS const uint16_t *off16 = as_lds(L.off);
	CRT_LDS const uint32_t *tab32 = (CRT_LDS const uint32_t *)as_lds(L.bytes);

	auto fetch4 = [&](uint32_t j) -> uint32_t {                         // four codewords as one (unaligned) dword
		if(j + 4 <= last) return *(CRT_GLOBAL const uint32_t *)(src + j);
		uint32_t v = 0;
		for(uint32_t k = 0; k < 4 && j + k < last; k++) v |= (uint32_t)src[j + k] << (8*k);
		return v;
	};
	// Window state between steps: window byte 0 is the 16-byte aligned destination address below base; of the `phase`
	// bytes in front of base, the first `foreign` are not this wave's to write (another wave's, or already written by the
	// general path), the rest are composed and pending.
	uint32_t foreign = (uint32_t)(uintptr_t)(gdst + base) & 15u;
	bool pending = false;
	auto write_pending = [&]() {                                        // bytes [foreign, phase) of window vector 0 -> HBM, bytewise
		const uint32_t phase = (uint32_t)(uintptr_t)(gdst + base) & 15u;
		CRT_GLOBAL uint8_t *a0 = gdst + base - phase;
		if(lane >= foreign && lane < phase) a0[lane] = wb[16 + lane];
		if(lane == 0) win[0] = u32x4_t{0, 0, 0, 0};
		pending = false; foreign = phase;
	};

	// The loop over a wave's quarter chunk, compiled for each (table width W, codewords per lane CPL) that occurs.  A lane
	// takes the step's codewords in NG groups of GS consecutive ones (group g = codewords g*64*GS + GS*lane ..): the closer
	// neighbouring lanes' words are in the window, the fewer LDS bank conflicts the ORs have; a group is one dword of the fetch.
	{
		constexpr int GS = CPL < 4 ? CPL : 4, NG = CPL/GS;     // (GS = 2 for short words halves the conflicts but its four 2-byte fetches and second scan cost more)
		constexpr uint32_t sub = 64*CPL;
		auto fetchg = [&](uint32_t j) -> uint32_t {                        // a group's GS codewords (GS < 4: with CPL < 4, fetch4's spare bytes are ignored)
			if constexpr(GS == 2 && CPL == 8) {
				if(j + 2 <= last) return *(CRT_GLOBAL const uint16_t *)(src + j);
				return j < last ? (uint32_t)src[j] : 0u;
			} else return fetch4(j);
		};
		// Codewords are fetched two steps ahead and taken out of the fetched registers BEFORE the step's stores are issued:
		// loads and stores share one in-order counter (vmcnt), so a load waited for behind this step's flush would
		// cost the flush's whole HBM write latency, every step.
		uint32_t code[CPL], l[CPL], raw[NG];
		auto extract = [&]() {
#pragma unroll
			for(int k = 0; k < CPL; k++) code[k] = (raw[k/GS] >> (8*(k % GS))) & 255u;
		};
		auto fetch_step = [&](uint32_t t) {
#pragma unroll
			for(int g = 0; g < NG; g++) raw[g] = fetchg(t + g*64*GS + GS*lane);
		};
		fetch_step(first);
		extract();
		fetch_step(first + sub);
		for(uint32_t tile = first; tile < last; tile += sub) {
			const uint32_t j0 = tile + GS*lane;
			uint32_t gsum[NG];
#pragma unroll
			for(int g = 0; g < NG; g++) gsum[g] = 0;
			const bool full = tile + sub <= last;                             // wave-uniform; false only on a stream's last step
			uint32_t x[CPL][W];                                                // the words' padded copies: read now, their latency
#pragma unroll
			for(int k = 0; k < CPL; k++) {                                     // overlaps the length reads and the scan
				if constexpr(W == 1) { const u32x2_t v = *(CRT_LDS const u32x2_t *)(t16l + 2*code[k]); x[k][0] = v.x; l[k] = v.y; }
				else if constexpr(W == 2) { const u32x2_t v = *(CRT_LDS const u32x2_t *)(t16l + 2*code[k]); x[k][0] = v.x; x[k][1] = v.y; }
				else { const u32x4_t v = *(CRT_LDS const u32x4_t *)(t16l + 4*code[k]); x[k][0] = v.x; x[k][1] = v.y; x[k][2] = v.z; x[k][3] = v.w; }
			}
			if constexpr(W != 1) {
#pragma unroll
				for(int k = 0; k < CPL; k++) l[k] = len8[code[k]];
			}
#pragma unroll
			for(int k = 0; k < CPL; k++) {
				if(!full && !(j0 + (k/GS)*64*GS + (k % GS) < last)) l[k] = 0;      // a stream's last step: codewords past its end
				gsum[k/GS] += l[k];
			}
			// exclusive offsets of the lane's groups: groups are scanned two per register (a group's bytes stay below 2^16)
			uint32_t orel[NG], total = 0;
#pragma unroll
			for(int g = 0; g < NG; g += 2) {
				const uint32_t hi = g + 1 < NG ? gsum[g + 1 < NG ? g + 1 : g] : 0u;
				const uint32_t inc = wave_inclusive_scan_u32(gsum[g] | hi << 16);
				const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
				orel[g] = total + (inc & 0xffffu) - gsum[g];
				total += tot & 0xffffu;
				if(g + 1 < NG) { orel[g + 1 < NG ? g + 1 : g] = total + (inc >> 16) - hi; total += tot >> 16; }
			}
			const bool fast = full && total + 32 <= TUN_WIN && tile + sub < csize && base + total <= size;
			auto next_codes = [&]() {                                          // the next step's codewords out, the one after's in flight
				extract();
				fetch_step(tile + 2*sub);
			};
			if(fast) {
				CRT_GLOBAL uint8_t *g0 = gdst + base;
				const uint32_t phase = (uint32_t)(uintptr_t)g0 & 15u;
				{	// compose
					uint32_t P = 0, N = 0, nlong = 0;
#pragma unroll
					for(int k = 0; k < CPL; k++) {
						if(k % GS == 0) { P = win0 + phase + orel[k/GS] - 1u; N = ~P; }   // next group
						tun_or<W>(P, N, x[k]);
						if constexpr(W == 4) {
							const bool lg = l[k] > 16;                                  // queue the rest of a long word
							const uint64_t m = __ballot(lg);
							if(m) {
								if(nlong + (uint32_t)__popcll(m) > TUN_LONGQ) { tun_drain_long(win0, longq, nlong, off16, len8, tab32); nlong = 0; }
								if(lg) longq[nlong + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (P + 1u - win0) | code[k] << 16;
								nlong += (uint32_t)__popcll(m);
							}
						}
						P += l[k]; N -= l[k];
					}
					if constexpr(W == 4) if(nlong) tun_drain_long(win0, longq, nlong, off16, len8, tab32);
				}
				next_codes();
				// flush the whole vectors of [0, phase + total) and re-zero them (same wave: LDS program order, no barrier)
				const uint32_t end = phase + total, nvec = end >> 4;
				CRT_GLOBAL u32x4_t *gv = (CRT_GLOBAL u32x4_t *)(g0 - phase);
				uint32_t i0 = 0;
				if(foreign && nvec) {                                            // vector 0 holds bytes that are not ours
					if(lane >= foreign && lane < 16) ((CRT_GLOBAL uint8_t *)gv)[lane] = wb[16 + lane];
					if(lane == 0) win[0] = u32x4_t{0, 0, 0, 0};
					i0 = 1;
				}
				for(uint32_t i = i0 + lane; i < nvec; i += 64) { TUN_FLUSH_STORE(win[i], &gv[i]); win[i] = u32x4_t{0, 0, 0, 0}; }
				if(nvec) {
					foreign = 0;
					if((end & 15u) && lane == 0) { const u32x4_t t = win[nvec]; win[nvec] = u32x4_t{0, 0, 0, 0}; win[0] = t; }   // carry the tail
				}
				pending = (end & 15u) > foreign;
			} else {
				// general path: byte FIFO straight to HBM, with the clipping rules of the stream's end (tunstall.cpp:447-451)
				uint32_t ccode[CPL];
#pragma unroll
				for(int k = 0; k < CPL; k++) ccode[k] = code[k];
				next_codes();
				if(pending) write_pending();
#pragma unroll
				for(int g = 0; g < NG; g++) {                                    // a group's words are one run of up to four
					uint32_t wo[4], nb[4];
					uint64_t oo = base + orel[g];
					const uint64_t o_run = oo;
#pragma unroll
					for(int r = 0; r < 4; r++) {
						uint32_t n_ = 0;
						wo[r] = 0;
						if(r < GS) {
							const int kk = g*GS + (r < GS ? r : 0);
							const uint32_t j = j0 + g*64*GS + r;
							n_ = l[kk];
							wo[r] = L.off[ccode[kk]];
							if(j < last) {
								if(j + 1 == csize) n_ = oo < size ? (uint32_t)min((uint64_t)(TUN_TABLE_BYTES - wo[r]), size - oo) : 0u;
								else if(oo + n_ > size) n_ = oo < size ? (uint32_t)(size - oo) : 0u;
							} else n_ = 0;
							oo += l[kk];
						}
						nb[r] = n_;
					}
					CRT_GLOBAL uint8_t *d = gdst + o_run;
					tun_emit_run(d, (uint32_t)(uintptr_t)d, tab32, wo, nb);
				}
				foreign = (uint32_t)(uintptr_t)(gdst + base + total) & 15u;     // everything below the next base is written now
			}
			base += total;
		}
	}
	if(pending) write_pending();
}

// Words of at most 4 bytes, of at most 8, and the rest (which picks its step size by the stream) are three bodies of ONE kernel: a
// workgroup runs the body of its stream's class.  Registers and LDS are the long-word class's (the others' are within a few registers of
// it); round 1's three launches - one kernel per class over all chunks, workgroups of another class leaving at once - had two extra grids
// of workgroups and two under-filled tails between the classes (removed in round 4).
__global__ __launch_bounds__(256) void k_tun_decode_staged_any(const TunStream *__restrict__ streams, const uint32_t *__restrict__ chunk_stream,
                                                               uint32_t nchunks, const TunTable *__restrict__ tables,
                                                               uint64_t *chunk_out, uint32_t sums_only) {
	const uint32_t c = blockIdx.x;
	if(blockIdx.x >= nchunks) return;
	const TunStream st = streams[chunk_stream[c]];
	const TunTable &T = tables[st.table];
	const uint32_t W = tun_width(st.cpl, T.maxlen);
	__shared__ TunLds L;
	__shared__ __attribute__((aligned(16))) uint32_t t16[256*4];
	__shared__ uint32_t longbuf[4][TUN_LONGQ];
	extern __shared__ __attribute__((aligned(16))) uint32_t winbuf[];
	if(W == 1) tun_staged_body<1, 8>(st, T, c, chunk_out, sums_only, L, t16, (uint32_t (*)[TUN_LONGQ])longbuf, winbuf);
	else if(W == 2) tun_staged_body<2, 8>(st, T, c, chunk_out, sums_only, L, t16, (uint32_t (*)[TUN_LONGQ])longbuf, winbuf);
	else if(st.cpl == 8) tun_staged_body<4, 8>(st, T, c, chunk_out, sums_only, L, t16, longbuf, winbuf);
	else if(st.cpl == 4) tun_staged_body<4, 4>(st, T, c, chunk_out, sums_only, L, t16, longbuf, winbuf);
	else if(st.cpl == 2) tun_staged_body<4, 2>(st, T, c, chunk_out, sums_only, L, t16, longbuf, winbuf);
	else tun_staged_body<4, 1>(st, T, c, chunk_out, sums_only, L, t16, longbuf, winbuf);
}

// sums_only != 0: chunk_out holds every quarter's decoded SIZE (k_tun_chunk_sums) and a wave adds up the ones in front of its own (streams
// of at most 256 chunks); 0: chunk_out holds the scanned offsets (k_tun_stream_scan ran in between)
int launch_tun_decode_staged(hipStream_t stream, const TunStream *streams, const uint32_t *chunk_stream, uint32_t nchunks, const TunTable *tables,
                             uint64_t *chunk_out, uint32_t sums_only) {
	hipLaunchKernelGGL(k_tun_decode_staged_any, dim3(nchunks), dim3(256), tun_staged_lds(tun_win_bytes<4, 8>()), stream, streams, chunk_stream, nchunks, tables, chunk_out, sums_only);
	return hipGetLastError() == hipSuccess ? 0 : -1;
}

// memset path: single-symbol streams (tunstall.cpp:433-436)
__global__ __launch_bounds__(256) void k_fill(const FillJob *__restrict__ jobs, uint32_t njobs) {
	const uint32_t j = blockIdx.x;
	if(j >= njobs) return;
	const FillJob f = jobs[j];
	for(uint32_t i = threadIdx.x; i < f.size; i += 256) f.dst[i] = (uint8_t)f.value;
}

// one block of memory filled with a byte, 16 bytes a thread and store (crthip_pool poisons a context's output block with it: one
// launch on the context's stream, no host-side work beyond that - hipMemsetAsync of 32 MB cost the calling thread ~0.2 ms)
__global__ __launch_bounds__(256) void k_fill_block(uint8_t *__restrict__ dst, uint64_t bytes, uint32_t value) {
	typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
	const uint32_t w = value*0x01010101u;
	const u32x4_t v = {w, w, w, w};
	const uint64_t head = (16u - (uint32_t)((uintptr_t)dst & 15u)) & 15u, nvec = bytes > head ? (bytes - head) >> 4 : 0;
	u32x4_t *d4 = (u32x4_t *)(dst + head);
	for(uint64_t i = (uint64_t)blockIdx.x*256 + threadIdx.x; i < nvec; i += (uint64_t)gridDim.x*256) __builtin_nontemporal_store(v, d4 + i);
	if(blockIdx.x == 0) {
		for(uint64_t i = threadIdx.x; i < head && i < bytes; i += 256) dst[i] = (uint8_t)value;
		for(uint64_t i = head + nvec*16 + threadIdx.x; i < bytes; i += 256) dst[i] = (uint8_t)value;
	}
}

} // namespace corto_hip

#ifdef CORTO_TUN_STAMPS
extern "C" int crthip_debug_tun_stamps(uint64_t *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(corto_hip::g_tun_stamps), sizeof(uint64_t)*8*4096); }
#endif
