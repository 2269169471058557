// encoder.cpp — host-side .crt writer: lets tests and bench.py synthesise inputs on the GPU
// box without the reference library.  Byte-identical to upstream's crt::Encoder for the attribute set the decoder path
// covers (positions, normals in all three prediction modes, rgb/rgba colours, uvs, one generic "radius" attribute,
// groups, exif, entropy NONE/TUNSTALL, meshes and point clouds); tests/test_encoder_cpu.py pins that against the
// reference-made golden blobs and, where oracle/_ref exists, against the reference on a random corpus.
//
// What it restates (upstream file:line):
//   container + stage order   src/encoder.cpp:207-296 (encode, encodePointCloud), :311-381 (encodeMesh)
//   quantisation              src/encoder.cpp:49-100, include/corto/vertex_attribute.h:79-128, src/normal_attribute.cpp:61-111,
//                             src/color_attribute.cpp:23-70
//   value coding              include/corto/cstream.h:105-204 (needed, encodeValues, encodeArray), src/bitstream.cpp:86-101,123-129
//   Tunstall                  src/tunstall.cpp:83-115 (getProbabilities), :125-256 (dictionary), :335-428 (encoding trie, compress),
//                             src/cstream.cpp:89-109 (block framing)
//   topology                  src/encoder.cpp:383-504 (buildTopology), :522-722 (encodeFaces)
//   normals                   src/normal_attribute.cpp:113-176 (preDelta, deltaEncode, encode)
// Where the reference's output depends on std::sort's handling of ties (probability order, edge buckets, Morton order)
// the same std::sort call on the same element sequence is made, which reproduces it exactly.
#include <cfloat>
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "../../include/corto_hip.h"
#include "encoder_internal.h"

namespace {

int ilog2u(uint64_t p) { int k = 0; while(p >>= 1) ++k; return k; }                  // src/cstream.cpp:31-35

int32_t f2i(float x) {                                                                 // x86 cvttss2si
	if(!(x > -2147483904.0f && x < 2147483648.0f)) return INT_MIN;
	return (int32_t)x;
}

// ---- streams left to the device (crthip_encode_gpu): recorded instead of written, spliced in at `at` afterwards ----
struct Deferred {
	size_t at = 0;                      // position in the sink's bytes where the stream belongs
	uint32_t kind = 0;                  // CRTHIP_ENC_*, or DEFER_BITS for a bit stream that is already packed (CLERS split bits)
	uint32_t count = 0, N = 1;
	std::vector<uint8_t> bytes;         // symbols / int8 values
	std::vector<int32_t> ints;          // int32 values
	std::vector<uint32_t> words;        // DEFER_BITS
};
constexpr uint32_t DEFER_BITS = 0xFFu;

// ---- byte sink (OutStream, include/corto/cstream.h:42-105) ----
struct Sink {
	std::vector<uint8_t> b;
	std::vector<Deferred> *defer = nullptr;   // non-null: value / symbol / bit streams are recorded, not written
	void u8(uint32_t v) { b.push_back((uint8_t)v); }
	void u16(uint32_t v) { u8(v); u8(v >> 8); }
	void u32(uint32_t v) { u8(v); u8(v >> 8); u8(v >> 16); u8(v >> 24); }
	void f32(float f) { uint32_t u; memcpy(&u, &f, 4); u32(u); }
	void str(const std::string &s) { u16((uint32_t)s.size() + 1); b.insert(b.end(), s.begin(), s.end()); u8(0); }
	void raw(const void *p, size_t n) { const uint8_t *q = (const uint8_t *)p; b.insert(b.end(), q, q + n); }
};

// ---- MSB-first bit writer over u32 words (src/bitstream.cpp:86-101, 123-129) ----
struct BitWriter {
	std::vector<uint32_t> words;
	uint32_t buff = 0;
	int bits = 32;                      // free bits in buff
	void write(uint32_t value, int n) {
		if(n >= bits) {
			buff = (bits == 32 ? 0u : (buff << bits)) | (value >> (n - bits));
			words.push_back(buff);
			const int rest = n - bits;
			value &= rest >= 32 ? 0xFFFFFFFFu : ((1u << rest) - 1u);
			n = rest; bits = 32; buff = 0;
		}
		if(n > 0) { buff = (buff << n) | value; bits -= n; }
	}
	void flush() { if(bits != 32) { words.push_back(buff << bits); buff = 0; bits = 32; } }
	void emit(Sink &s) {                // OutStream::write(BitStream&), cstream.h:79-89
		flush();
		if(s.defer) { Deferred d; d.at = s.b.size(); d.kind = DEFER_BITS; d.words = words; s.defer->push_back(std::move(d)); return; }
		s.u32((uint32_t)words.size());
		while(s.b.size() & 3) s.u8(0);
		for(uint32_t w : words) s.u32(w);
	}
};

// ---- Tunstall encoder ----
struct Sym { uint8_t symbol, probability; };

struct Tunstall {
	std::vector<Sym> probs;
	std::vector<int> index, lengths;    // 256 words after build()
	std::vector<uint8_t> table;
	std::vector<int> offsets;           // 2-symbol-step trie (createEncodingTables)
	std::vector<uint8_t> remap;

	void probabilities(const uint8_t *data, int size) {                               // tunstall.cpp:83-115
		std::vector<int> cnt(256, 0);
		for(int i = 0; i < size; i++) cnt[data[i]]++;
		from_counts(cnt, size);
	}
	void from_counts(const std::vector<int> &cnt, int size) {
		probs.clear();
		for(int i = 0; i < 256; i++) if(cnt[i] > 0) probs.push_back(Sym{(uint8_t)i, (uint8_t)(cnt[i]*255/size)});
		std::sort(probs.begin(), probs.end(), [](const Sym &a, const Sym &b) -> bool { return a.probability > b.probability; });
	}

	void build() {                                                                     // tunstall.cpp:125-256
		const uint32_t n = (uint32_t)probs.size();
		if(n <= 1) return;
		std::vector<uint32_t> q(1024, 0), head(n);
		index.assign(1024, 0); lengths.assign(1024, 0);
		table.assign(8192 + 1024, 0);
		std::vector<uint32_t> P(n);
		for(uint32_t i = 0; i < n; i++) P[i] = (uint32_t)probs[i].probability << 8;
		uint32_t count = 2, run = (P[0]*P[0]) >> 16, max_count = 255/(n - 1), pos = 0, end = 0, nwords = 0;
		while(run > P[1] && count < max_count) { run = (run*P[0]) >> 16; count++; }
		if(count >= 16) {
			table[pos++] = probs[0].symbol;
			for(uint32_t k = 1; k < n; k++) { for(uint32_t i = 0; i + 1 < count; i++) table[pos++] = probs[0].symbol; table[pos++] = probs[k].symbol; }
			head[0] = (count - 1)*n;
			for(uint32_t k = 1; k < n; k++) head[k] = k;
			uint32_t pw = 0;
			for(uint32_t col = 0; col < count; col++) {
				for(uint32_t row = 1; row < n; row++) {
					const uint32_t e = row + col*n;
					q[e] = col == 0 ? P[row] : (pw*P[row]) >> 16;
					index[e] = (int)(row*count - col); lengths[e] = (int)(col + 1);
				}
				pw = col == 0 ? P[0] : (pw*P[0]) >> 16;
			}
			const uint32_t first = (count - 1)*n;
			q[first] = pw; index[first] = 0; lengths[first] = (int)count;
			nwords = 1 + count*(n - 1); end = count*n;
		} else {
			for(uint32_t i = 0; i < n; i++) { head[i] = i; q[end] = P[i]; index[end] = (int)pos; lengths[end] = 1; end++; table[pos++] = probs[i].symbol; }
			nwords = n;
		}
		while(nwords < 256) {
			uint32_t best = 0, maxp = 0;
			for(uint32_t i = 0; i < n; i++) { const uint32_t p = head[i] < q.size() ? q[head[i]] : 0; if(p > maxp) { best = i; maxp = p; } }
			const uint32_t parent = head[best], pp = q[parent], po = (uint32_t)index[parent], pl = (uint32_t)lengths[parent];
			uint32_t r = 0;
			for(; r < n; r++) {
				q[end] = (pp*P[r]) >> 16; index[end] = (int)pos; lengths[end] = (int)pl + 1; end++;
				memmove(&table[pos], &table[po], pl); pos += pl;
				table[pos++] = probs[r].symbol;
				if(nwords + r == 255) break;
			}
			if(r == n) head[best] += n;
			nwords += n - 1;
		}
		size_t w = 0;
		for(size_t e = 0; e < end && w < 256; e++) { if(head[e % n] > e) continue; index[w] = index[e]; lengths[w] = lengths[e]; w++; }
		index.resize(256); lengths.resize(256);
	}

	void code(const uint8_t *w, int length, int &low, int &high) const {              // wordCode, tunstall.h:117-132 (lookup_size 2)
		const int n = (int)probs.size();
		int c = 0;
		for(int i = 0; i < length && i < 2; i++) c = c*n + remap[w[i]];
		low = c; high = c + 1;
		for(int i = length; i < 2; i++) { low *= n; high *= n; }
	}

	void trie() {                                                                      // createEncodingTables, tunstall.cpp:335-382
		const int n = (int)probs.size();
		if(n <= 1) return;
		const int span = n*n;
		remap.assign(256, 0);
		for(int i = 0; i < n; i++) remap[probs[i].symbol] = (uint8_t)i;
		offsets.assign(span, 0xffffff);
		for(size_t i = 0; i < index.size(); i++) {
			int low, high, off = 0, toff = 0;
			for(;;) {
				code(&table[index[i] + off], lengths[i] - off, low, high);
				if(lengths[i] - off <= 2) { for(int k = low; k < high; k++) offsets[toff + k] = (int)i; break; }
				const int w = offsets[toff + low];
				if(w >= 0) { offsets[toff + low] = -(int)offsets.size(); offsets.resize(offsets.size() + span, w); }
				toff = -offsets[toff + low];
				off += 2;
			}
		}
	}

	std::vector<uint8_t> compress(const uint8_t *data, int size) const {               // tunstall.cpp:384-428
		std::vector<uint8_t> out;
		if(probs.size() == 1) return out;
		int in = 0, woff = 0, off = 0;
		while(in < size) {
			const int d = std::min(size - in, 2);
			int low, high;
			code(data + in, d, low, high);
			off = offsets[-off + low];
			if(off >= 0) { out.push_back((uint8_t)off); in += lengths[off] - woff; off = 0; woff = 0; }
			else { woff += 2; in += 2; }
		}
		if(off < 0) { while(off < 0) off = offsets[-off]; out.push_back((uint8_t)off); }
		return out;
	}
};

// entropy-coded byte array (OutStream::compress / tunstall_compress, cstream.cpp:43-64, 89-109)
void put_symbols(Sink &s, uint32_t entropy, const uint8_t *data, uint32_t size) {
	if(s.defer) { Deferred d; d.at = s.b.size(); d.kind = CRTHIP_ENC_SYMBOLS; d.count = size; d.bytes.assign(data, data + size); s.defer->push_back(std::move(d)); return; }
	if(entropy == CRTHIP_ENTROPY_NONE) { s.u32(size); s.raw(data, size); return; }
	Tunstall t;
	t.probabilities(data, (int)size);
	t.build();
	t.trie();
	const std::vector<uint8_t> c = t.compress(data, (int)size);
	s.u8((uint32_t)t.probs.size());
	for(const Sym &p : t.probs) { s.u8(p.symbol); s.u8(p.probability); }
	s.u32(size);
	s.u32((uint32_t)c.size());
	s.raw(c.data(), c.size());
}

int needed(int a) {                                                                    // cstream.h:105-112
	if(a == 0) return 0;
	if(a == -1) return 1;
	if(a < 0) a = -a - 1;
	int n = 2;
	while(a >>= 1) n++;
	return n;
}

// encodeValues (cstream.h:115-141): component-major logs, sign folding
template <class T> void put_values(Sink &s, uint32_t entropy, uint32_t size, const T *values, int N) {
	if(s.defer) {
		Deferred d; d.at = s.b.size(); d.count = size; d.N = (uint32_t)N;
		if(sizeof(T) == 1) { d.kind = CRTHIP_ENC_VALUES_I8; d.bytes.assign((const uint8_t *)values, (const uint8_t *)values + (size_t)size*N); }
		else { d.kind = CRTHIP_ENC_VALUES_I32; d.ints.assign((const int32_t *)values, (const int32_t *)values + (size_t)size*N); }
		s.defer->push_back(std::move(d)); return;
	}
	BitWriter bw;
	std::vector<std::vector<uint8_t>> logs((size_t)N, std::vector<uint8_t>(size));
	for(int c = 0; c < N; c++) for(uint32_t i = 0; i < size; i++) {
		int val = values[(size_t)i*N + c];
		if(val == 0) { logs[c][i] = 0; continue; }
		const int ret = ilog2u((uint64_t)std::abs(val)) + 1;
		logs[c][i] = (uint8_t)ret;
		const int middle = (1 << ret) >> 1;
		if(val < 0) val = -val - middle;
		bw.write((uint32_t)val, ret);
	}
	bw.emit(s);
	for(int c = 0; c < N; c++) put_symbols(s, entropy, logs[c].data(), size);
}

// encodeArray (cstream.h:143-164): one log per element
void put_array(Sink &s, uint32_t entropy, uint32_t size, const int32_t *values, int N) {
	if(s.defer) { Deferred d; d.at = s.b.size(); d.kind = CRTHIP_ENC_ARRAY; d.count = size; d.N = (uint32_t)N; d.ints.assign(values, values + (size_t)size*N); s.defer->push_back(std::move(d)); return; }
	BitWriter bw;
	std::vector<uint8_t> logs(size);
	for(uint32_t i = 0; i < size; i++) {
		const int32_t *p = values + (size_t)i*N;
		int diff = needed(p[0]);
		for(int c = 1; c < N; c++) diff = std::max(diff, needed(p[c]));
		logs[i] = (uint8_t)diff;
		if(diff == 0) continue;
		const int mx = 1 << (diff - 1);
		for(int c = 0; c < N; c++) bw.write((uint32_t)(p[c] + mx), diff);
	}
	bw.emit(s);
	put_symbols(s, entropy, logs.data(), size);
}

struct Quad { uint32_t t, a, b, c; };

// normals: octahedral map (include/corto/normal_attribute.h:75-85)
void to_octa(const float v[3], int unit, int32_t o[2]) {
	float s = std::fabs(v[0]) + std::fabs(v[1]); s = s + std::fabs(v[2]);
	float px = v[0]/s, py = v[1]/s;
	if(v[2] < 0) {
		const float qx = 1.0f - std::fabs(py), qy = 1.0f - std::fabs(px);
		px = qx; py = qy;
		if(v[0] < 0) px = -px;
		if(v[1] < 0) py = -py;
	}
	o[0] = f2i(px*(float)unit); o[1] = f2i(py*(float)unit);
}

struct Attr {
	std::string name;
	int codec = CRTHIP_CODEC_GENERIC, N = 0, format = CRTHIP_FMT_FLOAT, strategy = 0;
	float q = 0;
	std::vector<int32_t> values, diffs;        // generic + normal (normal: 2 per vertex)
	std::vector<uint8_t> cvalues, cdiffs;      // colour
	int qc[4] = {4, 4, 4, 8};
	int prediction = 0;                        // normals
	std::vector<int32_t> boundary;
};

// ---- face adjacency: every directed side of a triangle is a HALF-EDGE, id 3*face + side, side s running opposite corner s
// (side 0 = corners 1->2, side 1 = 2->0, side 2 = 0->1); twin[h] = the half-edge of the neighbouring face running the other way
// along the same two vertices, or NO_TWIN.  Same result as upstream's buildTopology (src/encoder.cpp:450-504) - including on
// non-manifold input, where WHICH two of several candidates get paired depends on the order std::sort leaves equal keys in:
// so the half-edges are bucketed by their smaller vertex in the same order, and every bucket goes through std::sort with an
// equivalent strict-weak order on (smaller vertex, larger vertex).  std::sort's permutation is a function of the comparison
// results alone, so a different element type does not change it.
constexpr uint32_t NO_TWIN = 0xffffffffu;
struct SideKey {
	uint32_t lo, hi;        // the side's two vertices, lo < hi
	uint32_t half;          // 3*face + side
	uint32_t flipped;       // the side runs hi -> lo
	bool operator<(const SideKey &o) const { return lo != o.lo ? lo < o.lo : hi < o.hi; }
};

static void pair_half_edges(const uint32_t *corner, size_t ntri, uint32_t nvert, std::vector<uint32_t> &twin) {
	twin.assign(ntri*3, NO_TWIN);
	// bucket b = sides whose smaller vertex is b, filled face by face, side 0, 1, 2
	std::vector<uint32_t> fill(nvert + 1, 0);
	auto ends = [&](size_t t, int side, uint32_t &from, uint32_t &to) { from = corner[3*t + (side + 1)%3]; to = corner[3*t + (side + 2)%3]; };
	for(size_t t = 0; t < ntri; t++)
		for(int side = 0; side < 3; side++) { uint32_t a, b; ends(t, side, a, b); fill[std::min(a, b) + 1]++; }
	for(uint32_t v = 0; v < nvert; v++) fill[v + 1] += fill[v];
	std::vector<uint32_t> first(fill.begin(), fill.end() - 1);              // bucket starts (fill[] becomes the write cursors)
	std::vector<SideKey> sides(ntri*3);
	for(size_t t = 0; t < ntri; t++)
		for(int side = 0; side < 3; side++) {
			uint32_t a, b; ends(t, side, a, b);
			SideKey k; k.lo = std::min(a, b); k.hi = std::max(a, b); k.half = (uint32_t)(3*t + side); k.flipped = a > b;
			sides[fill[k.lo]++] = k;
		}
	// (upstream skips every bucket after the first that starts at offset 0 - i.e. while no smaller vertex had a side of its own,
	// src/encoder.cpp:481-485 - so the sides of the first non-empty bucket stay unsorted, and pair only where they happen to lie
	// next to each other, when vertex 0 is not the smaller end of any side; byte identity needs the same)
	for(uint32_t v = 0; v < nvert; v++) { if(v > 0 && first[v] == 0) continue; std::sort(sides.begin() + first[v], sides.begin() + fill[v]); }
	// a side pairs with the candidate kept from before it when they run opposite ways along the same vertices and neither has a
	// twin yet; any side that does not pair becomes the candidate
	const SideKey *cand = nullptr;
	for(const SideKey &k : sides) {
		if(cand && cand->lo == k.lo && cand->hi == k.hi && cand->flipped != k.flipped) {
			if(twin[k.half] == NO_TWIN && twin[cand->half] == NO_TWIN) { twin[k.half] = cand->half; twin[cand->half] = k.half; }
		} else cand = &k;
	}
}

enum { VERTEX = 0, LEFT = 1, RIGHT = 2, END = 3, BOUNDARY = 4, DELAY = 5, SPLIT = 6 };

struct Encoder {
	uint32_t nvert, nface, entropy;
	std::vector<uint32_t> faces;                 // original indexing, degenerate faces removed in encode_mesh
	std::vector<uint32_t> group_end;
	std::vector<std::map<std::string, std::string>> group_props;
	std::map<std::string, std::string> exif;
	std::map<std::string, Attr> data;            // std::map: alphabetical like upstream
	std::vector<uint8_t> clers;
	BitWriter split;
	uint32_t max_front = 0, current_vertex = 0, last_index = 0;
	std::vector<int> encoded;
	std::vector<Quad> prediction;
	Sink s;

	// The CLERS writer for faces [start, end) (one group): the region-growing walk of upstream's encodeFaces (src/encoder.cpp:522-722)
	// - same symbols, same split / vertex-id bits, same vertex numbering and parallelogram corners - on this repo's own terms:
	// the advancing front is a circular list threaded THROUGH the half-edges (a half-edge joins the front at most once, so
	// before[] / after[] / where[] are indexed by half-edge id and there is no separate edge store), gates wait in a FIFO of
	// half-edge ids, postponed ones on a stack.
	void encode_faces(int start, int end) {
		const size_t ntri = (size_t)(end - start);
		const uint32_t *corner = faces.data() + (size_t)start*3;
		std::vector<uint32_t> twin;
		pair_half_edges(corner, ntri, nvert, twin);

		enum : uint8_t { OFF_FRONT = 0, ON_FRONT = 1, CLOSED = 2 };
		std::vector<uint8_t> where(ntri*3, OFF_FRONT);                    // a half-edge's life: not reached / on the front / swallowed by a later face
		std::vector<uint32_t> before(ntri*3, 0), after(ntri*3, 0);        // its neighbours along the front
		std::vector<uint8_t> coded(ntri, 0);                              // faces already written
		std::vector<uint32_t> gates, postponed;
		size_t gate_cursor = 0, seed_cursor = 0, remaining = ntri;
		uint32_t joined = 0;                                              // half-edges that ever joined the front (upstream's front.size())
		uint32_t pending = NO_TWIN;                                       // the front edge the previous step made: always the next gate

		std::vector<uint8_t> used(nvert, 0);
		for(uint32_t v : faces) used[v] = 1;
		uint32_t nused = 0; for(uint8_t u : used) nused += u;
		const int idbits = ilog2u(nused) + 1;                              // bits of a known vertex's number in the split stream

		auto onto_front = [&](uint32_t h, uint32_t p, uint32_t n) { where[h] = ON_FRONT; before[h] = p; after[h] = n; joined++; };
		auto introduce = [&](uint32_t v, uint32_t a, uint32_t b, uint32_t c) {      // a vertex seen for the first time gets the next number
			prediction[current_vertex] = Quad{v, a, b, c};
			encoded[v] = (int)current_vertex++;
			last_index = v;
		};
		auto mention = [&](uint32_t v) { split.write((uint32_t)encoded[v], idbits); };

		while(remaining) {
			uint32_t gate;
			if(pending != NO_TWIN) { gate = pending; pending = NO_TWIN; }
			else if(gate_cursor < gates.size()) gate = gates[gate_cursor++];
			else if(!postponed.empty()) { gate = postponed.back(); postponed.pop_back(); }
			else {
				// nothing left to grow from: start a new component with the first face not written yet
				while(seed_cursor < ntri && coded[seed_cursor]) seed_cursor++;
				if(seed_cursor == ntri) break;
				const uint32_t t = (uint32_t)seed_cursor;
				const uint32_t *c3 = corner + 3*(size_t)t;
				uint32_t known = 0;
				for(int k = 0; k < 3; k++) if(encoded[c3[k]] != -1) known |= 1u << k;
				if(known) { clers.push_back(SPLIT); split.write(known, 3); } else clers.push_back(VERTEX);
				for(int k = 0; k < 3; k++) {
					if(encoded[c3[k]] != -1) mention(c3[k]);
					else introduce(c3[k], last_index, last_index, last_index);
				}
				const uint32_t h = 3*t;                                      // its three sides circle the face: side 0 -> 1 -> 2 -> 0
				onto_front(h, h + 2, h + 1); onto_front(h + 1, h, h + 2); onto_front(h + 2, h + 1, h);
				gates.push_back(h); gates.push_back(h + 1); gates.push_back(h + 2);
				coded[t] = 1; remaining--;
				continue;
			}
			if(where[gate] == CLOSED) continue;
			const uint32_t tw = twin[gate];
			if(tw == NO_TWIN || coded[tw/3]) { clers.push_back(BOUNDARY); continue; }
			const uint32_t across = tw/3;                                    // the face on the other side of the gate
			const uint32_t s_far = tw%3, s_a = (s_far + 1)%3, s_b = (s_a + 1)%3;   // its corner opposite the gate, then the gate's two ends
			const uint32_t left = before[gate], right = after[gate];
			const bool zip_left = twin[left] != NO_TWIN && twin[left]/3 == across;
			const bool zip_right = twin[right] != NO_TWIN && twin[right]/3 == across;
			const uint32_t h_a = 3*across + s_a, h_b = 3*across + s_b;       // the two sides of `across` that may join the front
			if(zip_left && zip_right) {                                      // the face closes a triangular hole
				clers.push_back(END);
				const uint32_t ll = before[left], rr = after[right];
				where[left] = CLOSED; where[right] = CLOSED;
				after[ll] = rr; before[rr] = ll;
			} else if(zip_left) {
				clers.push_back(LEFT);
				const uint32_t ll = before[left];
				where[left] = CLOSED;
				onto_front(h_b, ll, right);
				after[ll] = h_b; before[right] = h_b;
				pending = h_b;
			} else if(zip_right) {
				clers.push_back(RIGHT);
				const uint32_t rr = after[right];
				where[right] = CLOSED;
				onto_front(h_a, left, rr);
				before[rr] = h_a; after[left] = h_a;
				pending = h_a;
			} else {
				const uint32_t far = corner[3*(size_t)across + s_far];
				if(encoded[far] != -1 && gate_cursor < gates.size()) {          // a known vertex while gates are still waiting: come back to this one later
					postponed.push_back(gate); clers.push_back(DELAY);
					continue;
				}
				if(encoded[far] != -1) { clers.push_back(SPLIT); mention(far); }
				else {
					clers.push_back(VERTEX);
					// parallelogram corners: the gate's two ends and the corner of the gate's own face opposite it
					introduce(far, corner[3*(size_t)across + s_a], corner[3*(size_t)across + s_b], corner[gate]);
				}
				onto_front(h_a, left, h_b); onto_front(h_b, h_a, right);
				after[left] = h_a; before[right] = h_b;
				gates.push_back(h_b);
				pending = h_a;
			}
			coded[across] = 1; remaining--;
		}
		max_front = std::max(max_front, joined);
	}

	// NormalAttr::preDelta (normal_attribute.cpp:113-143): uses ORIGINAL vertex ids and quantised positions
	void normal_predelta(Attr &a) {
		if(a.prediction == 0) return;
		auto it = data.find("position");
		const std::vector<int32_t> &coords = it->second.values;
		std::vector<float> est((size_t)nvert*3, 0.f);
		a.boundary.assign(nvert, 0);
		for(uint32_t f = 0; f < nface; f++) {
			const uint32_t i0 = faces[(size_t)f*3], i1 = faces[(size_t)f*3 + 1], i2 = faces[(size_t)f*3 + 2];
			const int32_t *p0 = &coords[(size_t)i0*3], *p1 = &coords[(size_t)i1*3], *p2 = &coords[(size_t)i2*3];
			const float ax = (float)p1[0] - (float)p0[0], ay = (float)p1[1] - (float)p0[1], az = (float)p1[2] - (float)p0[2];
			const float bx = (float)p2[0] - (float)p0[0], by = (float)p2[1] - (float)p0[1], bz = (float)p2[2] - (float)p0[2];
			const float n[3] = {ay*bz - az*by, az*bx - ax*bz, ax*by - ay*bx};
			for(int k = 0; k < 3; k++) est[(size_t)i0*3 + k] += n[k];
			for(int k = 0; k < 3; k++) est[(size_t)i1*3 + k] += n[k];
			for(int k = 0; k < 3; k++) est[(size_t)i2*3 + k] += n[k];
			if(a.prediction == 2) {
				a.boundary[i0] ^= (int32_t)i1; a.boundary[i0] ^= (int32_t)i2; a.boundary[i1] ^= (int32_t)i2;
				a.boundary[i1] ^= (int32_t)i0; a.boundary[i2] ^= (int32_t)i0; a.boundary[i2] ^= (int32_t)i1;
			}
		}
		const int unit = f2i(a.q);
		for(uint32_t i = 0; i < nvert; i++) {
			int32_t o[2]; to_octa(&est[(size_t)i*3], unit, o);
			a.values[(size_t)i*2] = (int32_t)((uint32_t)a.values[(size_t)i*2] - (uint32_t)o[0]);
			a.values[(size_t)i*2 + 1] = (int32_t)((uint32_t)a.values[(size_t)i*2 + 1] - (uint32_t)o[1]);
		}
	}

	void delta_encode(Attr &a) {
		const std::vector<Quad> &ctx = prediction;
		if(a.codec == CRTHIP_CODEC_NORMAL) {                                            // normal_attribute.cpp:145-176
			if(a.prediction == 0) {
				a.diffs.assign(ctx.size()*2, 0);
				if(ctx.empty()) return;
				a.diffs[0] = a.values[(size_t)ctx[0].t*2]; a.diffs[1] = a.values[(size_t)ctx[0].t*2 + 1];
				for(size_t i = 1; i < ctx.size(); i++) for(int c = 0; c < 2; c++)
					a.diffs[i*2 + c] = (int32_t)((uint32_t)a.values[(size_t)ctx[i].t*2 + c] - (uint32_t)a.values[(size_t)ctx[i].a*2 + c]);
			} else {
				a.diffs.clear();
				for(const Quad &q : ctx) if(a.prediction != 2 || a.boundary[q.t] != 0) { a.diffs.push_back(a.values[(size_t)q.t*2]); a.diffs.push_back(a.values[(size_t)q.t*2 + 1]); }
			}
			return;
		}
		const int N = a.N;
		if(a.codec == CRTHIP_CODEC_COLOR) {                                             // GenericAttr<uchar>::deltaEncode, strategy 0
			a.cdiffs.assign(ctx.size()*(size_t)N, 0);
			if(ctx.empty()) return;
			for(int c = 0; c < N; c++) a.cdiffs[c] = a.cvalues[(size_t)ctx[0].t*N + c];
			for(size_t i = 1; i < ctx.size(); i++) {
				const Quad &q = ctx[i];
				for(int c = 0; c < N; c++) {
					if(q.a != q.b && (a.strategy & CRTHIP_PARALLEL)) a.cdiffs[i*N + c] = (uint8_t)(a.cvalues[(size_t)q.t*N + c] - (a.cvalues[(size_t)q.a*N + c] + a.cvalues[(size_t)q.b*N + c] - a.cvalues[(size_t)q.c*N + c]));
					else a.cdiffs[i*N + c] = (uint8_t)(a.cvalues[(size_t)q.t*N + c] - a.cvalues[(size_t)q.a*N + c]);
				}
			}
			return;
		}
		a.diffs.assign(ctx.size()*(size_t)N, 0);                                        // vertex_attribute.h:130-144
		if(ctx.empty()) return;
		for(int c = 0; c < N; c++) a.diffs[c] = a.values[(size_t)ctx[0].t*N + c];
		for(size_t i = 1; i < ctx.size(); i++) {
			const Quad &q = ctx[i];
			for(int c = 0; c < N; c++) {
				const uint32_t t = (uint32_t)a.values[(size_t)q.t*N + c], va = (uint32_t)a.values[(size_t)q.a*N + c];
				if(q.a != q.b && (a.strategy & CRTHIP_PARALLEL)) a.diffs[i*N + c] = (int32_t)(t - (va + (uint32_t)a.values[(size_t)q.b*N + c] - (uint32_t)a.values[(size_t)q.c*N + c]));
				else a.diffs[i*N + c] = (int32_t)(t - va);
			}
		}
	}

	void attr_encode(Attr &a) {
		if(a.codec == CRTHIP_CODEC_NORMAL) { s.u8((uint32_t)a.prediction); put_array(s, entropy, (uint32_t)(a.diffs.size()/2), a.diffs.data(), 2); return; }
		if(a.codec == CRTHIP_CODEC_COLOR) {
			for(int c = 0; c < a.N; c++) s.u8((uint32_t)a.qc[c]);
			put_values<int8_t>(s, entropy, nvert, (const int8_t *)a.cdiffs.data(), a.N);
			return;
		}
		if(a.strategy & CRTHIP_CORRELATED) put_array(s, entropy, nvert, a.diffs.data(), a.N);
		else put_values<int32_t>(s, entropy, nvert, a.diffs.data(), a.N);
	}

	void header() {                                                                    // src/encoder.cpp:207-229
		s.u32(0x787A6300); s.u32(1); s.u8(entropy);
		s.u32((uint32_t)exif.size());
		for(auto &kv : exif) { s.str(kv.first); s.str(kv.second); }
		s.u32((uint32_t)data.size());
		for(auto &kv : data) { const Attr &a = kv.second; s.str(kv.first); s.u32((uint32_t)a.codec); s.f32(a.q); s.u8((uint32_t)a.N); s.u8((uint32_t)a.format); s.u8((uint32_t)a.strategy); }
	}

	// IndexAttribute::encodeGroups (include/corto/index_attribute.h:71-81): properties in std::map order = sorted by key
	void groups() {
		s.u32((uint32_t)group_end.size());
		for(size_t g = 0; g < group_end.size(); g++) {
			s.u32(group_end[g]);
			if(g < group_props.size()) {
				s.u8((uint8_t)group_props[g].size());
				for(auto &kv : group_props[g]) { s.str(kv.first); s.str(kv.second); }
			} else s.u8(0);
		}
	}

	void encode_mesh() {                                                               // src/encoder.cpp:311-381
		encoded.assign(nvert, -1);
		if(group_end.empty()) group_end.push_back(nface);
		uint32_t start = 0, count = 0;
		for(uint32_t &g : group_end) {
			for(uint32_t i = start; i < g; i++) {
				const uint32_t *f = &faces[(size_t)i*3];
				if(f[0] == f[1] || f[0] == f[2] || f[1] == f[2]) continue;
				if(count != i) { faces[(size_t)count*3] = f[0]; faces[(size_t)count*3 + 1] = f[1]; faces[(size_t)count*3 + 2] = f[2]; }
				count++;
			}
			start = g; g = count;
		}
		faces.resize((size_t)count*3);
		nface = count;
		prediction.assign(nvert, Quad{0, 0, 0, 0});
		start = 0;
		for(uint32_t g : group_end) { encode_faces((int)start, (int)g); start = g; }
		for(auto &kv : data) if(kv.second.codec == CRTHIP_CODEC_NORMAL) normal_predelta(kv.second);
		nvert = current_vertex;
		prediction.resize(nvert);
		for(auto &kv : data) delta_encode(kv.second);
		s.u32(nvert); s.u32(nface);
		groups();
		s.u32(max_front);
		put_symbols(s, entropy, clers.data(), (uint32_t)clers.size());
		split.emit(s);
		for(auto &kv : data) attr_encode(kv.second);
	}

	struct ZPoint { uint64_t bits; uint32_t pos; bool operator<(const ZPoint &z) const { return bits > z.bits; } };

	void encode_cloud() {                                                              // src/encoder.cpp:238-296
		const std::vector<int32_t> &coords = data.find("position")->second.values;
		int32_t mn[3] = {0, 0, 0};
		for(uint32_t i = 0; i < nvert; i++) for(int k = 0; k < 3; k++) mn[k] = std::min(mn[k], coords[(size_t)i*3 + k]);
		std::vector<ZPoint> z(nvert);
		for(uint32_t i = 0; i < nvert; i++) {
			const uint64_t x = (uint64_t)(int64_t)(coords[(size_t)i*3] - mn[0]), y = (uint64_t)(int64_t)(coords[(size_t)i*3 + 1] - mn[1]), w = (uint64_t)(int64_t)(coords[(size_t)i*3 + 2] - mn[2]);
			uint64_t bits = 0; const uint64_t l = 1;
			for(int k = 0; k < 21; k++) bits |= (x & l << k) << (2*k) | (y & l << k) << (2*k + 1) | (w & l << k) << (2*k + 2);   // include/corto/zpoint.h:34-38
			z[i] = ZPoint{bits, i};
		}
		std::sort(z.rbegin(), z.rend());
		s.u32(nvert); s.u32(0);
		groups();
		prediction.resize(nvert);
		if(nvert) prediction[0] = Quad{z[0].pos, 0xffffffff, 0xffffffff, 0xffffffff};
		for(uint32_t i = 1; i < nvert; i++) prediction[i] = Quad{z[i].pos, z[i - 1].pos, z[i - 1].pos, z[i - 1].pos};
		for(auto &kv : data) if(kv.second.codec == CRTHIP_CODEC_NORMAL) normal_predelta(kv.second);
		for(auto &kv : data) delta_encode(kv.second);
		for(auto &kv : data) attr_encode(kv.second);
	}
};

} // namespace

extern "C" {


// Encode a mesh / point cloud into a .crt blob (host only).  Returns the blob size (also when out == NULL or cap is too
// small: call twice), or <0.  out_nvert/out_nface = counts after unreferenced vertices / degenerate faces are dropped.
static int64_t encode_impl(const crthip_mesh *m, uint8_t *out, size_t cap, uint32_t *out_nvert, uint32_t *out_nface, crthip_ctx *gpu) {
	if(!m || !m->position) return CRTHIP_E_ARGUMENT;
	Encoder E;
	std::vector<Deferred> deferred;
	if(gpu) E.s.defer = &deferred;
	E.nvert = m->nvert; E.nface = m->index ? m->nface : 0; E.entropy = (uint32_t)m->entropy;
	const char *p = m->exif;
	for(uint32_t i = 0; i < m->nexif; i++) { std::string k(p); p += k.size() + 1; std::string v(p); p += v.size() + 1; E.exif[k] = v; }
	for(uint32_t g = 0; g < m->ngroups; g++) E.group_end.push_back(m->group_end[g]);
	if(m->group_nprops && m->group_props) {
		const char *gp = m->group_props;
		E.group_props.resize(m->ngroups);
		for(uint32_t g = 0; g < m->ngroups; g++)
			for(uint32_t i = 0; i < m->group_nprops[g]; i++) {
				std::string k(gp); gp += k.size() + 1;
				std::string v(gp); gp += v.size() + 1;
				E.group_props[g][k] = v;
			}
	}
	if(E.nface) E.faces.assign(m->index, m->index + (size_t)E.nface*3);
	const uint32_t nv = m->nvert;
	std::vector<corto_hip::QuantRequest> quant;                     // (device path) the attributes' quantisation, collected and run in one call
	{	// positions (src/encoder.cpp:49-100, vertex_attribute.h:79-128)
		float q = m->position_q;
		if(m->position_bits > 0) {
			float mn[3] = {m->position[0], m->position[1], m->position[2]}, mx[3] = {mn[0], mn[1], mn[2]};
			for(uint32_t i = 0; i < nv; i++) for(int k = 0; k < 3; k++) { const float v = m->position[(size_t)i*3 + k]; if(v < mn[k]) mn[k] = v; if(v > mx[k]) mx[k] = v; }
			const float intervals = powf(2.0f, (float)m->position_bits);
			float e[3]; for(int k = 0; k < 3; k++) { e[k] = mx[k] - mn[k]; e[k] /= intervals; }
			q = std::max(std::max(e[0], e[1]), e[2]);
		} else if(q == 0.0f && E.nface) {                       // a twentieth of the mean length of each face's first edge (src/encoder.cpp:105-110)
			double average = 0;
			for(uint32_t f = 0; f < E.nface; f++) {
				const float *a = m->position + (size_t)m->index[(size_t)f*3]*3, *b = m->position + (size_t)m->index[(size_t)f*3 + 1]*3;
				const float d[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
				average += (float)sqrt((double)(d[0]*d[0] + d[1]*d[1] + d[2]*d[2]));   // Point3f::norm, include/corto/point.h:111
			}
			q = (float)(average/E.nface)/20.0f;
		} else if(q == 0.0f && nv) {                            // point cloud: from the bounding box volume (src/encoder.cpp:83-91)
			float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
			for(uint32_t i = 0; i < nv; i++) for(int k = 0; k < 3; k++) { const float v = m->position[(size_t)i*3 + k] - 0.0f; if(v < mn[k]) mn[k] = v; if(v > mx[k]) mx[k] = v; }
			for(int k = 0; k < 3; k++) mx[k] -= mn[k];
			q = (float)(0.02*pow(mx[0]*mx[1]*mx[2], 2.0/3.0)/nv);
		}
		Attr &a = E.data["position"];
		a.name = "position"; a.N = 3; a.q = q; a.format = CRTHIP_FMT_FLOAT;
		a.strategy = CRTHIP_CORRELATED | (E.nface > 0 ? CRTHIP_PARALLEL : 0);
		a.values.resize((size_t)nv*3);
		if(gpu) { corto_hip::QuantRequest r; r.kind = 0; r.count = nv*3; r.in = m->position; r.out = a.values.data(); r.q = q; quant.push_back(r); }
		else for(size_t i = 0; i < (size_t)nv*3; i++) a.values[i] = f2i((m->position[i] - 0.0f)/q);
	}
	if(m->normal) {
		Attr &a = E.data["normal"];
		a.name = "normal"; a.codec = CRTHIP_CODEC_NORMAL; a.N = 3; a.q = powf(2.0f, (float)(m->normal_bits - 1));
		a.format = CRTHIP_FMT_FLOAT; a.strategy = CRTHIP_CORRELATED; a.prediction = m->normal_prediction;
		a.values.resize((size_t)nv*2);
		const int unit = f2i(a.q);
		if(gpu) { corto_hip::QuantRequest r; r.kind = 1; r.count = nv; r.in = m->normal; r.out = a.values.data(); r.unit = unit; quant.push_back(r); }
		else for(uint32_t i = 0; i < nv; i++) to_octa(m->normal + (size_t)i*3, unit, &a.values[(size_t)i*2]);
	}
	if(m->color) {
		Attr &a = E.data["color"];
		a.name = "color"; a.codec = CRTHIP_CODEC_COLOR; a.N = m->color_components; a.format = CRTHIP_FMT_UINT8; a.strategy = 0; a.q = 0;
		for(int k = 0; k < 3; k++) a.qc[k] = 1 << (8 - m->color_bits[k]);
		a.qc[3] = m->color_components == 3 ? 1 : 1 << (8 - m->color_bits[3]);           // addColors3: setQ(r, g, b, 8)
		a.cvalues.resize((size_t)nv*a.N);
		if(gpu) { corto_hip::QuantRequest r; r.kind = 2; r.count = nv; r.N = (uint32_t)a.N; r.in = m->color; r.out = a.cvalues.data(); for(int k = 0; k < 4; k++) r.qc[k] = (uint32_t)a.qc[k]; quant.push_back(r); }
		else for(uint32_t i = 0; i < nv; i++) {                                         // color_attribute.cpp:30-44, point.h:213
			uint8_t y[4] = {0, 0, 0, 0};
			for(int k = 0; k < a.N; k++) y[k] = (uint8_t)(m->color[(size_t)i*a.N + k]/a.qc[k]);
			const uint8_t ycc[4] = {y[1], (uint8_t)(y[2] - y[1]), (uint8_t)(y[0] - y[1]), y[3]};
			for(int k = 0; k < a.N; k++) a.cvalues[(size_t)i*a.N + k] = ycc[k];
		}
	}
	auto generic = [&](const char *name, const float *buf, int N, float q) {
		Attr &a = E.data[name];
		a.name = name; a.N = N; a.q = q; a.format = CRTHIP_FMT_FLOAT; a.strategy = 0;
		a.values.resize((size_t)nv*N);
		if(gpu) { corto_hip::QuantRequest r; r.kind = 0; r.count = nv*(uint32_t)N; r.in = buf; r.out = a.values.data(); r.q = q; quant.push_back(r); }
		else for(size_t i = 0; i < (size_t)nv*N; i++) a.values[i] = f2i(buf[i]/q);
	};
	if(m->uv) generic("uv", m->uv, 2, m->uv_q);
	if(m->radius) generic("radius", m->radius, 1, m->radius_q);
	if(gpu) { const int qerr = corto_hip::quantize_device(gpu, quant); if(qerr) return qerr; }   // every attribute's quantisation in one device call (k_enc_quantize)
	E.header();
	if(E.nface > 0) E.encode_mesh(); else E.encode_cloud();
	if(out_nvert) *out_nvert = E.nvert;
	if(out_nface) *out_nface = E.nface;
	if(gpu) {
		// the recorded streams go through the device stages in one call, then everything is spliced together; the zero padding
		// in front of every bit stream (OutStream::write(BitStream&), cstream.h:79-89) depends on the final position, so it is made here
		std::vector<corto_hip::EncValueStream> in;
		for(const Deferred &d : deferred) {
			if(d.kind == DEFER_BITS) continue;
			corto_hip::EncValueStream v;
			v.kind = d.kind; v.count = d.count; v.components = d.N;
			v.values = d.kind == CRTHIP_ENC_SYMBOLS || d.kind == CRTHIP_ENC_VALUES_I8 ? (const void *)d.bytes.data() : (const void *)d.ints.data();
			in.push_back(v);
		}
		std::vector<corto_hip::EncValueResult> res;
		const int err = corto_hip::encode_value_streams(gpu, E.entropy, in, res, nullptr);
		if(err) return err;
		Sink f;
		size_t prev = 0, k = 0;
		auto bits = [&](const std::vector<uint32_t> &w) { f.u32((uint32_t)w.size()); while(f.b.size() & 3) f.u8(0); for(uint32_t x : w) f.u32(x); };
		for(const Deferred &d : deferred) {
			f.raw(E.s.b.data() + prev, d.at - prev); prev = d.at;
			if(d.kind == DEFER_BITS) { bits(d.words); continue; }
			const corto_hip::EncValueResult &r = res[k++];
			if(d.kind != CRTHIP_ENC_SYMBOLS) bits(r.words);
			for(const std::vector<uint8_t> &b : r.blocks) f.raw(b.data(), b.size());
		}
		f.raw(E.s.b.data() + prev, E.s.b.size() - prev);
		E.s.b.swap(f.b);
	}
	if(out && cap >= E.s.b.size()) memcpy(out, E.s.b.data(), E.s.b.size());
	return (int64_t)E.s.b.size();
}

// Arguments are checked before anything is indexed with them (upstream trusts its caller: an index >= nvert writes past its
// vectors, src/encoder.cpp:341-347), and nothing is thrown across the C boundary.
static int64_t encode_checked(const crthip_mesh *m, uint8_t *out, size_t cap, uint32_t *out_nvert, uint32_t *out_nface, crthip_ctx *gpu) {
	if(!m || !m->position) return corto_hip::ctx_fail(CRTHIP_E_ARGUMENT, "crthip_encode: no mesh / no positions");
	if(m->index && m->nface) {
		if((uint64_t)m->nface*3 > 0xFFFFFFFFull) return corto_hip::ctx_fail(CRTHIP_E_LIMIT, "crthip_encode: too many faces");
		for(size_t i = 0; i < (size_t)m->nface*3; i++)
			if(m->index[i] >= m->nvert) return corto_hip::ctx_fail(CRTHIP_E_ARGUMENT, "crthip_encode: face index out of range");
	}
	if(m->color) {
		if(m->color_components != 3 && m->color_components != 4) return corto_hip::ctx_fail(CRTHIP_E_ARGUMENT, "crthip_encode: color_components must be 3 or 4");
		for(int k = 0; k < m->color_components; k++)
			if(m->color_bits[k] < 1 || m->color_bits[k] > 8) return corto_hip::ctx_fail(CRTHIP_E_ARGUMENT, "crthip_encode: color_bits must be 1..8");
	}
	if(m->normal && (m->normal_bits < 1 || m->normal_bits > 16 || m->normal_prediction < 0 || m->normal_prediction > 2))
		return corto_hip::ctx_fail(CRTHIP_E_ARGUMENT, "crthip_encode: normal_bits must be 1..16, normal_prediction 0..2");
	if(m->position_bits > 31) return corto_hip::ctx_fail(CRTHIP_E_ARGUMENT, "crthip_encode: position_bits must be < 32");
	if(m->entropy != CRTHIP_ENTROPY_NONE && m->entropy != CRTHIP_ENTROPY_TUNSTALL) return corto_hip::ctx_fail(CRTHIP_E_ENTROPY, nullptr);
	if(m->ngroups) {
		if(!m->group_end || m->ngroups > (1u << 24)) return corto_hip::ctx_fail(CRTHIP_E_ARGUMENT, "crthip_encode: groups");
		uint32_t prev = 0;
		for(uint32_t g = 0; g < m->ngroups; g++) {
			// (a point cloud's group table is written as given and never used to index anything, src/encoder.cpp:238-296)
			if(m->group_end[g] < prev || (m->index && m->nface && m->group_end[g] > m->nface)) return corto_hip::ctx_fail(CRTHIP_E_ARGUMENT, "crthip_encode: group ends must be ascending and <= nface");
			prev = m->group_end[g];
			if(m->group_nprops && m->group_nprops[g] > 255) return corto_hip::ctx_fail(CRTHIP_E_LIMIT, "crthip_encode: more than 255 properties in a group");
		}
	}
	try {
		return encode_impl(m, out, cap, out_nvert, out_nface, gpu);
	} catch(const std::bad_alloc &) {
		return corto_hip::ctx_fail(CRTHIP_E_NOMEM, nullptr);
	} catch(...) {
		return corto_hip::ctx_fail(CRTHIP_E_ARGUMENT, "crthip_encode: internal error");
	}
}

int64_t crthip_encode(const crthip_mesh *m, uint8_t *out, size_t cap, uint32_t *out_nvert, uint32_t *out_nface) {
	return encode_checked(m, out, cap, out_nvert, out_nface, nullptr);
}

// crthip_encode with the value coding (bit widths, bit packing) and the entropy coder on the device (encode_gpu.cpp)
int64_t crthip_encode_gpu(crthip_ctx *ctx, const crthip_mesh *m, uint8_t *out, size_t cap, uint32_t *out_nvert, uint32_t *out_nface) {
	if(!ctx) return corto_hip::ctx_fail(CRTHIP_E_ARGUMENT, "crthip_encode_gpu: null context (there is no CPU fallback: use crthip_encode for the host encoder)");
	return encode_checked(m, out, cap, out_nvert, out_nface, ctx);
}

} // extern "C"


// the encoder-side Tunstall tables of one stream, from its byte histogram (for the GPU encoder stages, encode_gpu.cpp)
void corto_hip::tun_encoder_tables(const uint32_t counts[256], uint32_t size, TunEncoderTables &out) {
	Tunstall t;
	std::vector<int> cnt(counts, counts + 256);
	t.from_counts(cnt, (int)size);
	t.build();
	t.trie();
	out.nsym = (uint32_t)t.probs.size();
	for(uint32_t i = 0; i < out.nsym; i++) { out.probs[2*i] = t.probs[i].symbol; out.probs[2*i + 1] = t.probs[i].probability; }
	memset(out.remap, 0, sizeof(out.remap));
	memset(out.lengths, 0, sizeof(out.lengths));
	for(size_t i = 0; i < t.remap.size() && i < 256; i++) out.remap[i] = t.remap[i];
	for(size_t i = 0; i < t.lengths.size() && i < 256; i++) out.lengths[i] = (uint16_t)t.lengths[i];
	out.offsets.assign(t.offsets.begin(), t.offsets.end());
}
