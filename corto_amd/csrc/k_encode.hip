// k_encode.hip — GPU stages of the encoder for gfx950: the Tunstall coder of a batch of byte streams.
//
// Replaces, for many streams at once, the data-parallel part of
//   crt::Tunstall::getProbabilities   src/tunstall.cpp:83-115   (byte histogram)               -> k_enc_hist
//   crt::Tunstall::compress           src/tunstall.cpp:384-428  (greedy parse over the trie)   -> k_enc_tun_parse
// The 256-word dictionary and its 2-symbol-step trie are a few microseconds of serial work per stream and depend on
// std::sort's order of equal probabilities; they stay on the host (encoder.cpp: tun_encoder_tables), between the two kernels.
//
// K-ENC-PARSE: one wave per stream.  The reference parse is a serial chain - a codeword starts where the previous one
// ended - but where a codeword WOULD end if one started at byte p depends on p alone.  So the 64 lanes walk the trie from
// 64 consecutive start positions at once (same loop as the reference's, per lane), and the chain through that window is
// then a handful of scalar readlanes: cur -> next[cur].  Codewords are collected one per lane and stored 64 at a time.
// Source bytes are staged through LDS already remapped to symbol indices; the trie sits in LDS when it fits.
#include "kernels_common.h"
#include "kernels.h"
#include "std_sort_model.h"

namespace corto_hip {

#include "tun_tables.h"

__global__ __launch_bounds__(256) void k_enc_hist(const EncChunk *__restrict__ chunks, uint32_t nchunks, uint32_t *__restrict__ counts) {
	if(blockIdx.x >= nchunks) return;
	const EncChunk c = chunks[blockIdx.x];
	__shared__ uint32_t h[256];
	h[threadIdx.x] = 0;
	__syncthreads();
	CRT_GLOBAL const uint8_t *src = as_global(c.src);
	const uint32_t head = min((uint32_t)((0u - (uint32_t)(uintptr_t)c.src) & 3u), c.size);
	if(threadIdx.x < head) atomicAdd(&h[src[threadIdx.x]], 1u);
	const uint32_t ndw = (c.size - head) >> 2;
	CRT_GLOBAL const uint32_t *src32 = (CRT_GLOBAL const uint32_t *)(src + head);
	for(uint32_t i = threadIdx.x; i < ndw; i += 256) {
		const uint32_t x = src32[i];
		atomicAdd(&h[x & 255u], 1u); atomicAdd(&h[(x >> 8) & 255u], 1u); atomicAdd(&h[(x >> 16) & 255u], 1u); atomicAdd(&h[x >> 24], 1u);
	}
	for(uint32_t i = head + ndw*4 + threadIdx.x; i < c.size; i += 256) atomicAdd(&h[src[i]], 1u);
	__syncthreads();
	const uint32_t v = h[threadIdx.x];
	if(v) atomicAdd(&counts[(size_t)c.stream*256 + threadIdx.x], v);
}

// ------------------------------------------------------------------------------------------------------------------
// K-ENC-PACK: bit widths + bit packing of one value array per workgroup (OutStream::encodeArray / encodeValues,
// include/corto/cstream.h:115-164; BitStream::write, src/bitstream.cpp:86-101).  256 items per tile: every thread sizes its
// item (ARRAY: one width for the N components of an element; VALUES: one width per value, component-major, sign folded),
// a block scan gives the bit offsets, the fields are OR-ed MSB-first into an LDS tile (ds_or), whole words are flushed
// coalesced and the partial last word is carried into the next tile.
constexpr uint32_t ENC_PACK_TILE_WORDS = 256*ENC_PACK_MAX_N;

__device__ __forceinline__ uint32_t enc_needed(int32_t a) {              // cstream.h:105-112
	if(a == 0) return 0;
	if(a == -1) return 1;
	const uint32_t u = a < 0 ? ~(uint32_t)a : (uint32_t)a;
	return 2u + (31u - (uint32_t)__clz((int)u));
}

__device__ __forceinline__ void enc_put_bits(uint32_t *buf, uint32_t b, uint32_t v, uint32_t w) {   // w in 1..32, MSB first
	const uint32_t k = b >> 5, o = b & 31u;
	if(o + w <= 32u) atomicOr(&buf[k], v << (32u - o - w));
	else { const uint32_t r = o + w - 32u; atomicOr(&buf[k], v >> r); atomicOr(&buf[k + 1], v << (32u - r)); }
}

__global__ __launch_bounds__(256) void k_enc_pack(const PackJob *__restrict__ jobs, uint32_t njobs) {
	if(blockIdx.x >= njobs) return;
	const PackJob J = jobs[blockIdx.x];
	__shared__ uint32_t buf[ENC_PACK_TILE_WORDS + 4];
	__shared__ uint32_t scan[4];
	const uint32_t t = threadIdx.x, N = J.N, count = J.count;
	const bool array = J.kind == 1u;                                      // CRTHIP_ENC_ARRAY
	const bool bytes = J.kind == 3u;                                      // CRTHIP_ENC_VALUES_I8
	const uint32_t nitems = array ? count : count*N;
	CRT_GLOBAL const int32_t *v32 = as_global((const int32_t *)J.values);
	CRT_GLOBAL const int8_t *v8 = as_global((const int8_t *)J.values);
	CRT_GLOBAL uint8_t *logs = as_global(J.logs);
	CRT_GLOBAL uint32_t *words = as_global(J.words);
	for(uint32_t i = t; i < ENC_PACK_TILE_WORDS + 4; i += 256) buf[i] = 0;
	__syncthreads();
	uint32_t carry_bits = 0, gw = 0;
	for(uint32_t base = 0; base < nitems; base += 256) {
		const uint32_t it = base + t;
		uint32_t w = 0, nb = 0, val = 0;
		if(it < nitems) {
			if(array) {
				for(uint32_t c = 0; c < N; c++) { const uint32_t d = enc_needed(v32[(size_t)it*N + c]); w = d > w ? d : w; }
				nb = w*N;
			} else {
				const uint32_t c = it/count, i = it - c*count;
				int32_t x = bytes ? (int32_t)v8[(size_t)i*N + c] : v32[(size_t)i*N + c];
				if(x != 0) {
					const uint32_t ax = x < 0 ? 0u - (uint32_t)x : (uint32_t)x;
					w = 32u - (uint32_t)__clz((int)ax);                     // ilog2(abs) + 1
					// upstream: `int middle = (1<<ret)>>1` (cstream.h:133) - in int, so at ret = 31 (|x| >= 2^30) it is (INT_MIN >> 1) = -2^30, not 2^30, and the
					// folded value carries a bit above its field, which BitStream::write ORs onto the bit in front of it (enc_put_bits does the same: the
					// shifts below keep that bit unless the field starts a word, where upstream loses it too) - fixture fields32
					const uint32_t middle = (uint32_t)((int32_t)(1u << (w & 31u)) >> 1);
					val = x < 0 ? ax - middle : (uint32_t)x;
				}
				nb = w;
			}
			logs[it] = (uint8_t)w;
		}
		uint32_t total;
		const uint32_t excl = block256_exclusive_scan<uint32_t>(nb, scan, &total);
		if(nb) {
			uint32_t b = carry_bits + excl;
			if(array) {
				const uint32_t mx = 1u << (w - 1u);
				for(uint32_t c = 0; c < N; c++) { enc_put_bits(buf, b, (uint32_t)v32[(size_t)it*N + c] + mx, w); b += w; }
			} else enc_put_bits(buf, b, val, w);
		}
		__syncthreads();
		const uint32_t tile_bits = carry_bits + total, fw = tile_bits >> 5;
		for(uint32_t k = t; k < fw; k += 256) words[gw + k] = buf[k];
		const uint32_t cw = buf[fw];
		__syncthreads();
		for(uint32_t k = t; k <= fw + 1; k += 256) buf[k] = 0;
		__syncthreads();
		if(t == 0) buf[0] = cw;
		__syncthreads();
		gw += fw; carry_bits = tile_bits & 31u;
	}
	if(t == 0) {
		if(carry_bits) words[gw++] = buf[0];
		*J.nwords = gw;
	}
}

// ------------------------------------------------------------------------------------------------------------------
// Quantisation, elementwise.  The float recipes are upstream's, operation by operation (no FMA: the library is built with
// -ffp-contract=off; IEEE divide): GENERIC (int)(x/q) (vertex_attribute.h:97-99); NORMAL toOcta (normal_attribute.h:75-85):
// s = (|x| + |y|) + |z|, p = (x/s, y/s), folded when z < 0, (int)(p*unit); COLOR byte/qc then (g, b - g, r - g, a) (color_attribute.cpp:30-44,
// point.h:213).  (int) is x86's cvttss2si: INT_MIN when out of range.
__global__ __launch_bounds__(256) void k_enc_quantize(QuantJob J) {
	const uint32_t i = blockIdx.x*256 + threadIdx.x;
	if(i >= J.count) return;
	if(J.kind == 0) {
		const float x = ((const float *)J.in)[i] - 0.0f;
		((int32_t *)J.out)[i] = f2i_x86(x/J.q);
	} else if(J.kind == 1) {
		const float *v = (const float *)J.in + (size_t)i*3;
		const float vx = v[0], vy = v[1], vz = v[2];
		float s = fabsf(vx) + fabsf(vy); s = s + fabsf(vz);
		float px = vx/s, py = vy/s;
		if(vz < 0) {
			const float qx = 1.0f - fabsf(py), qy = 1.0f - fabsf(px);
			px = qx; py = qy;
			if(vx < 0) px = -px;
			if(vy < 0) py = -py;
		}
		int32_t *o = (int32_t *)J.out + (size_t)i*2;
		o[0] = f2i_x86(px*(float)J.unit); o[1] = f2i_x86(py*(float)J.unit);
	} else {
		const uint8_t *c = (const uint8_t *)J.in + (size_t)i*J.N;
		uint8_t y[4] = {0, 0, 0, 0};
		for(uint32_t k = 0; k < J.N && k < 4; k++) y[k] = (uint8_t)(c[k]/J.qc[k]);
		const uint8_t ycc[4] = {y[1], (uint8_t)(y[2] - y[1]), (uint8_t)(y[0] - y[1]), y[3]};
		uint8_t *o = (uint8_t *)J.out + (size_t)i*J.N;
		for(uint32_t k = 0; k < J.N && k < 4; k++) o[k] = ycc[k];
	}
}

// ------------------------------------------------------------------------------------------------------------------
// Encoder tables of one stream, by one wave: what getProbabilities + createDecodingTables2 leave (src/tunstall.cpp:83-115, 125-256).
//   probabilities  count*255/size of every symbol that occurs, in symbol order, then ordered by std::sort with a comparator on the
//                  probability alone - lane 0 runs std_sort_model.h, the restatement of libstdc++'s introsort, because the order
//                  std::sort leaves EQUAL probabilities in decides the dictionary;
//   dictionary     tun_tables.h, the builder the decoder uses (the encoder's dictionary IS the decoder's);
//   level_bound    a word of length L opens at most (L - 1)/2 levels of the 2-symbol-step trie: the host sizes the trie region.
__global__ __launch_bounds__(64) void k_enc_tables(const uint32_t *__restrict__ counts, const uint32_t *__restrict__ sizes, uint32_t nstreams,
                                                    EncTab *__restrict__ tabs) {
	const uint32_t s = blockIdx.x, lane = threadIdx.x;
	if(s >= nstreams) return;
	EncTab &E = tabs[s];
	const uint32_t size = sizes[s];
	__shared__ uint16_t pr[256];                           // symbol | probability << 8
	__shared__ uint8_t pb[512];
	__shared__ uint16_t loff[256];
	__shared__ uint8_t llen[256];
	uint32_t n = 0;
	for(uint32_t r = 0; r < 4; r++) {                      // symbols that occur, in symbol order
		const uint32_t sym = r*64 + lane, c = size ? counts[(size_t)s*256 + sym] : 0u;
		const uint64_t m = __ballot(c > 0);
		if(c > 0) pr[n + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)(sym | ((c*255u/size) & 255u) << 8);
		n += (uint32_t)__popcll(m);
	}
	__syncthreads();
	if(lane == 0) std_sort_model((uint16_t *)pr, (int)n, [](uint16_t a, uint16_t b) -> bool { return (a >> 8) > (b >> 8); });
	__syncthreads();
	for(uint32_t i = lane; i < 256; i += 64) { E.remap[i] = 0; if(i >= n) { E.probs[2*i] = 0; E.probs[2*i + 1] = 0; } }
	__syncthreads();
	for(uint32_t i = lane; i < n; i += 64) {
		const uint32_t v = pr[i];
		pb[2*i] = (uint8_t)v; pb[2*i + 1] = (uint8_t)(v >> 8);
		E.probs[2*i] = (uint8_t)v; E.probs[2*i + 1] = (uint8_t)(v >> 8);
		E.remap[v & 255u] = (uint8_t)i;
	}
	if(lane == 0) { E.nsym = n; E.level_bound = 0; E.used = 0; E.pad = 0; }
	__syncthreads();
	if(n < 2) return;                                      // one symbol: no dictionary, no payload (tunstall.cpp:386-389)
	TunStream st{};
	st.nsym = n;
	const TunBuilt B = tun_tables_body(st, nullptr, loff, llen, pb);
	__syncthreads();
	uint32_t bound = 0;
	for(uint32_t c = lane; c < 256; c += 64) {
		const uint32_t l = llen[c];
		E.lengths[c] = (uint16_t)l; E.index[c] = loff[c];
		bound += l > 2 ? (l - 1)/2 : 0u;
	}
	bound = wave_inclusive_scan_u32(bound);
	for(uint32_t i = lane; i < (B.used + 3)/4; i += 64) ((uint32_t *)E.words)[i] = ((const uint32_t *)g_tun_words)[i];
	if(lane == 63) { E.level_bound = 1 + bound; E.used = B.used; }
}

// createEncodingTables (src/tunstall.cpp:335-382) for the streams ids[]: the 2-symbol-step trie, grown in LDS by one wave.  The
// walk over the 256 words is the reference's, word by word, level by level (a later word overwrites what an earlier one left: the
// order is the result); what the wave adds is width - a new level's nsym^2 entries, and the range of entries a word's tail covers,
// are written by all lanes.  Entries: >= 0 a codeword, < 0 minus a level number; "no word yet" is 255, which is what the host's
// conversion of the reference's 0xffffff gives (it keeps the low byte), and behaves the same in the walk (any value >= 0 is copied).
__global__ __launch_bounds__(64) void k_enc_trie(const EncTab *__restrict__ tabs, const uint32_t *__restrict__ ids, EncStream *__restrict__ streams,
                                                  uint32_t nids, uint32_t trie_cap) {
	const uint32_t j = blockIdx.x, lane = threadIdx.x;
	if(j >= nids) return;
	const EncTab &E = tabs[ids[j]];
	extern __shared__ __attribute__((aligned(16))) uint8_t lds_[];
	CRT_LDS int16_t *t = (CRT_LDS int16_t *)as_lds(lds_);
	__shared__ uint8_t remap[256];
	__shared__ uint16_t index[256], lengths[256];
	__shared__ __attribute__((aligned(16))) uint8_t words[TUN_TABLE_BYTES];
	const uint32_t n = E.nsym, span = n*n;
	for(uint32_t i = lane; i < 256; i += 64) { remap[i] = E.remap[i]; index[i] = E.index[i]; lengths[i] = E.lengths[i]; }
	for(uint32_t i = lane; i < (E.used + 3)/4; i += 64) ((uint32_t *)words)[i] = ((const uint32_t *)E.words)[i];
	for(uint32_t k = lane; k < span && k < trie_cap; k += 64) t[k] = 255;
	__syncthreads();
	uint32_t size = span;
	bool overflow = span > trie_cap;
	for(uint32_t i = 0; i < 256 && !overflow; i++) {
		const uint32_t wl = lengths[i], w0 = index[i];
		uint32_t off = 0, toff = 0;
		for(;;) {
			const uint32_t rem = wl - off;
			uint32_t low = remap[words[w0 + off]], high;
			if(rem >= 2) { low = low*n + remap[words[w0 + off + 1]]; high = low + 1; }
			else { low *= n; high = low + n; }                    // a one-symbol tail covers every second symbol (wordCode, tunstall.h:117-132)
			if(rem <= 2) { for(uint32_t k = low + lane; k < high; k += 64) t[toff + k] = (int16_t)i; break; }
			const int32_t wv = t[toff + low];
			if(wv >= 0) {                                         // a complete word (or nothing) sits here: open a level that starts out as it
				if(size + span > trie_cap) { overflow = true; break; }
				if(lane == 0) t[toff + low] = (int16_t)-(int32_t)(size/span);
				for(uint32_t k = lane; k < span; k += 64) t[size + k] = (int16_t)wv;
				size += span;
			}
			__syncthreads();
			toff = (uint32_t)(-(int32_t)t[toff + low])*span;
			off += 2;
		}
		__syncthreads();
	}
	__syncthreads();
	EncStream &S = streams[j];
	if(!overflow) for(uint32_t k = lane; k < size; k += 64) S.trie[k] = t[k];
	if(lane == 0) S.ntrie = overflow ? 0xFFFFFFFFu : size;
}

// LDS: remap u8[256] | lengths u16[256] | staged symbol indices u8[ENC_STAGE + ENC_STAGE_PAD] | trie i16[ntrie] (when it fits)
template <bool TRIE_IN_LDS>
__device__ __forceinline__ void enc_parse_body(const EncStream &S, CRT_LDS uint8_t *lds) {
	const uint32_t lane = threadIdx.x, size = S.size, n = S.nsym, span = n*n;
	CRT_LDS uint8_t *remap = lds;
	CRT_LDS uint16_t *lengths = (CRT_LDS uint16_t *)(lds + 256);
	CRT_LDS uint8_t *stg = lds + 768;
	CRT_LDS int16_t *ltrie = (CRT_LDS int16_t *)(lds + 768 + ENC_STAGE + ENC_STAGE_PAD);
	CRT_GLOBAL const int16_t *gtrie = as_global(S.trie);
	CRT_GLOBAL const uint8_t *src = as_global(S.src);
	CRT_GLOBAL uint8_t *dst = as_global(S.dst);
	for(uint32_t i = lane; i < 256; i += 64) { remap[i] = S.remap[i]; lengths[i] = S.lengths[i]; }
	if(TRIE_IN_LDS) for(uint32_t i = lane; i < S.ntrie; i += 64) ltrie[i] = gtrie[i];
	__syncthreads();
	auto TR = [&](uint32_t i) -> int32_t { return i < S.ntrie ? (int32_t)(TRIE_IN_LDS ? ltrie[i] : gtrie[i]) : 0; };

	uint32_t base = 0, nout = 0, s0 = 0, s1 = 0;                       // staged: symbol indices of positions [s0, s1)
	int outv = 0;
	while(base < size && nout <= size) {
		if(base + 64 + ENC_STAGE_PAD > s1 && s1 < size) {              // restage from the window's first byte
			__syncthreads();
			s0 = base; s1 = min(size, s0 + ENC_STAGE + ENC_STAGE_PAD);
			for(uint32_t i = lane; i < s1 - s0; i += 64) stg[i] = remap[src[s0 + i]];
			__syncthreads();
		}
		// every lane: the reference's loop (tunstall.cpp:395-420) started at byte p, up to its first codeword
		const uint32_t p = base + lane;
		int32_t code = 0; uint32_t next = p;
		if(p < size) {
			uint32_t in = p, woff = 0, level = 0;                             // level: offset of the trie level being looked at (the reference's -off)
			for(;;) {
				int32_t t;
				if(in >= size) {                                              // ran off the end inside a word: follow the (0,0) entries down
					do { t = TR(level); level = (uint32_t)(-t)*span; } while(t < 0);
					code = t; next = in; break;
				}
				uint32_t low = (uint32_t)stg[in - s0]*n;
				if(size - in >= 2) low += stg[in + 1 - s0];
				t = TR(level + low);
				if(t >= 0) { code = t; next = in + lengths[t & 255] - woff; break; }
				level = (uint32_t)(-t)*span; woff += 2; in += 2;
			}
		}
		// the chain through this window
		uint32_t cur = base;
		while(cur < size && cur - base < 64 && nout <= size) {
			const uint32_t l = cur - base;
			const int c = __builtin_amdgcn_readlane(code, (int)l);
			cur = (uint32_t)__builtin_amdgcn_readlane((int)next, (int)l);
			outv = lane == (nout & 63u) ? c : outv;
			nout++;
			if((nout & 63u) == 0) dst[nout - 64 + lane] = (uint8_t)outv;
		}
		base = cur;
	}
	if(lane < (nout & 63u)) dst[(nout & ~63u) + lane] = (uint8_t)outv;
	if(lane == 0) *S.csize = nout;
}

__global__ __launch_bounds__(64) void k_enc_tun_parse(const EncStream *__restrict__ streams, uint32_t nstreams, uint32_t trie_lds_entries) {
	if(blockIdx.x >= nstreams) return;
	const EncStream S = streams[blockIdx.x];
	extern __shared__ __attribute__((aligned(16))) uint8_t lds_[];
	if(S.ntrie <= trie_lds_entries) enc_parse_body<true>(S, as_lds(lds_));
	else enc_parse_body<false>(S, as_lds(lds_));
}

} // namespace corto_hip
