// oracle/ref_shim.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// A thin C-ABI driver over the UNMODIFIED reference sources under /root/reference
// (crt::Encoder include/corto/encoder.h:36-96, crt::Decoder include/corto/decoder.h:38-73,
// crt::Tunstall include/corto/tunstall.h:41-150). It is compiled together with the
// reference .cpp files *where they lie* by oracle/Makefile into oracle/_ref/libcorto_ref.so.
// No reference source is copied into this repository; this file only calls the public API.
//
// Used by: tests/ (golden-vector generation + parity checks) and bench.py's cpu_baseline leg
// (cpu_baseline.kind == "reference").
#include <thread>
#include <stdint.h>
#include <string.h>
#include <chrono>
#include <map>
#include <string>
#include <vector>
#include <algorithm>

#include "corto.h"

using namespace crt;

extern "C" {

typedef struct {
	uint32_t nvert, nface;
	const float *position;        // nvert*3, required
	const uint32_t *index;        // nface*3 (NULL for point clouds)
	int32_t position_bits;        // >0: addPositionsBits(); else position_q is used
	float position_q;
	const float *normal;          // nvert*3 or NULL
	int32_t normal_bits;
	int32_t normal_prediction;    // 0 DIFF, 1 ESTIMATED, 2 BORDER
	const uint8_t *color;         // nvert*color_components or NULL
	int32_t color_components;     // 3 or 4
	int32_t color_bits[4];
	const float *uv;              // nvert*2 or NULL
	float uv_q;
	const float *radius;          // nvert*1 or NULL (generic 1-component attribute "radius")
	float radius_q;
	const uint32_t *group_end;    // ngroups face-end markers, or NULL
	uint32_t ngroups;
	int32_t entropy;              // 0 NONE, 1 TUNSTALL
	const char *exif;             // "k\0v\0k\0v\0..." nexif pairs, or NULL
	uint32_t nexif;
	const uint32_t *group_nprops; // pairs per group, or NULL: Encoder::addGroup(end, props) (include/corto/encoder.h:75)
	const char *group_props;      // all pairs of all groups, "k\0v\0..."
} ref_mesh_t;

typedef struct {
	float *position;
	void *normal;
	int32_t normal_format;        // VertexAttribute::FLOAT (6) or INT16 (3)
	uint8_t *color;
	int32_t color_components;
	float *uv;
	float *radius;
	uint32_t *index32;
	uint16_t *index16;
} ref_out_t;

static thread_local std::string g_err;
const char *ref_last_error() { return g_err.c_str(); }

// Encode with the reference crt::Encoder. Returns the .crt size, or -1 on error.
// out may be NULL / cap too small: only the size is returned then. nvert/nface after encoding
// (the encoder drops unreferenced vertices and degenerate faces) are returned via out_nvert/out_nface.
int64_t ref_encode(const ref_mesh_t *m, uint8_t *out, int64_t cap, uint32_t *out_nvert, uint32_t *out_nface) {
	try {
		Encoder enc(m->nvert, m->nface, (Stream::Entropy)m->entropy);
		const char *p = m->exif;
		for(uint32_t i = 0; i < m->nexif; i++) {
			std::string k(p); p += k.size() + 1;
			std::string v(p); p += v.size() + 1;
			enc.exif[k] = v;
		}
		const char *gp = m->group_props;
		for(uint32_t g = 0; g < m->ngroups; g++) {
			if(m->group_nprops && gp) {
				std::map<std::string, std::string> props;
				for(uint32_t i = 0; i < m->group_nprops[g]; i++) {
					std::string k(gp); gp += k.size() + 1;
					std::string v(gp); gp += v.size() + 1;
					props[k] = v;
				}
				enc.addGroup((int)m->group_end[g], props);
			} else enc.addGroup((int)m->group_end[g]);
		}
		if(m->nface == 0 || !m->index) {
			if(m->position_bits > 0) enc.addPositionsBits(m->position, m->position_bits);
			else enc.addPositions(m->position, m->position_q);
		} else {
			if(m->position_bits > 0) enc.addPositionsBits(m->position, (uint32_t *)m->index, m->position_bits);
			else enc.addPositions(m->position, m->index, m->position_q);
		}
		if(m->normal)
			enc.addNormals(m->normal, m->normal_bits, (NormalAttr::Prediction)m->normal_prediction);
		if(m->color) {
			if(m->color_components == 3)
				enc.addColors3(m->color, m->color_bits[0], m->color_bits[1], m->color_bits[2]);
			else
				enc.addColors(m->color, m->color_bits[0], m->color_bits[1], m->color_bits[2], m->color_bits[3]);
		}
		if(m->uv)
			enc.addUvs(m->uv, m->uv_q);
		if(m->radius)
			enc.addAttribute("radius", (const char *)m->radius, VertexAttribute::FLOAT, 1, m->radius_q);
		enc.encode();
		if(out_nvert) *out_nvert = enc.nvert;
		if(out_nface) *out_nface = enc.nface;
		int64_t size = enc.stream.size();
		if(out && cap >= size)
			memcpy(out, enc.stream.data(), size);
		return size;
	} catch(const char *e) {
		g_err = e;
		return -1;
	}
}

// Header facts (crt::Decoder ctor, src/decoder.cpp:41-89). attr_mask: 1 position, 2 normal, 4 color, 8 uv, 16 radius.
int ref_probe(const uint8_t *blob, int len, uint32_t *nvert, uint32_t *nface, uint32_t *attr_mask, int32_t *color_components) {
	try {
		Decoder dec(len, blob);
		*nvert = dec.nvert; *nface = dec.nface;
		uint32_t mask = 0;
		if(dec.hasAttr("position")) mask |= 1;
		if(dec.hasAttr("normal")) mask |= 2;
		if(dec.hasAttr("color")) { mask |= 4; if(color_components) *color_components = dec.data["color"]->N; }
		if(dec.hasAttr("uv")) mask |= 8;
		if(dec.hasAttr("radius")) mask |= 16;
		*attr_mask = mask;
		return 0;
	} catch(const char *e) {
		g_err = e;
		return -1;
	}
}

static void bind(Decoder &dec, const ref_out_t *o) {
	if(o->position) dec.setPositions(o->position);
	if(o->normal) {
		if(o->normal_format == VertexAttribute::INT16) dec.setNormals((int16_t *)o->normal);
		else dec.setNormals((float *)o->normal);
	}
	if(o->color) dec.setColors(o->color, o->color_components);
	if(o->uv) dec.setUvs(o->uv);
	if(o->radius) dec.setAttribute("radius", (char *)o->radius, VertexAttribute::FLOAT);
	if(o->index16) dec.setIndex(o->index16);
	if(o->index32) dec.setIndex(o->index32);
}

// What the reference Decoder leaves in index.groups after decode() (IndexAttribute::decodeGroups, include/corto/index_attribute.h:89-99):
// u32 ngroups | per group: u32 end, u32 nprops, nprops x "k\0v\0".  Returns bytes (also when cap is too small), or -1.
int64_t ref_groups(const uint8_t *blob, int len, uint8_t *out, int64_t cap) {
	try {
		Decoder dec(len, blob);
		std::vector<uint32_t> idx((size_t)dec.nface*3 + 3);
		if(dec.nface) dec.setIndex(idx.data());
		dec.decode();
		std::string flat;
		auto u32 = [&](uint32_t v) { flat.append((const char *)&v, 4); };
		u32((uint32_t)dec.index.groups.size());
		for(auto &g : dec.index.groups) {
			u32(g.end); u32((uint32_t)g.properties.size());
			for(auto &kv : g.properties) { flat += kv.first; flat.push_back('\0'); flat += kv.second; flat.push_back('\0'); }
		}
		if(out && cap >= (int64_t)flat.size()) memcpy(out, flat.data(), flat.size());
		return (int64_t)flat.size();
	} catch(const char *e) {
		g_err = e;
		return -1;
	}
}

// Full decode with the reference crt::Decoder (src/decoder.cpp:126-196).
int ref_decode(const uint8_t *blob, int len, const ref_out_t *o) {
	try {
		Decoder dec(len, blob);
		bind(dec, o);
		dec.decode();
		return 0;
	} catch(const char *e) {
		g_err = e;
		return -1;
	}
}

// ONE generic attribute bound with an arbitrary output format through Decoder::setAttribute(name, buffer, format) (src/decoder.cpp:96-102;
// the format only matters in GenericAttr::dequantize, include/corto/vertex_attribute.h:184-230).  buffer: nvert*N*8 bytes (the decode works
// in place on int32 values whatever the format, DOUBLE widens in place).  index32: nface*3, or NULL for clouds.
int ref_decode_attr_format(const uint8_t *blob, int len, const char *name, int format, uint8_t *buffer, uint32_t *index32) {
	try {
		Decoder dec(len, blob);
		if(!dec.setAttribute(name, (char *)buffer, (VertexAttribute::Format)format)) { g_err = "no such attribute"; return -1; }
		if(dec.nface && index32) dec.setIndex(index32);
		dec.decode();
		return 0;
	} catch(const char *e) {
		g_err = e;
		return -1;
	}
}

// Full decode + the topology intermediates the reference keeps in its public IndexAttribute
// (include/corto/index_attribute.h:48-60): decoded CLERS symbols and per-vertex prediction triples.
int ref_decode_trace(const uint8_t *blob, int len, const ref_out_t *o,
		uint8_t *clers, uint32_t clers_cap, uint32_t *nclers, uint32_t *prediction /* nvert*3 */, uint32_t *max_front) {
	try {
		Decoder dec(len, blob);
		bind(dec, o);
		dec.decode();
		*nclers = (uint32_t)dec.index.clers.size();
		if(clers) memcpy(clers, dec.index.clers.data(), std::min<size_t>(clers_cap, dec.index.clers.size()));
		if(prediction && dec.index.prediction.size())
			memcpy(prediction, dec.index.prediction.data(), dec.index.prediction.size()*sizeof(Face));
		*max_front = dec.index.max_front;
		return 0;
	} catch(const char *e) {
		g_err = e;
		return -1;
	}
}

// Timed region = ctor + set* + decode() with pre-allocated outputs (what src/main.cpp:266-300 times,
// minus its vector::resize). ns[iters] receives per-iteration nanoseconds.
int ref_decode_timed(const uint8_t *blob, int len, const ref_out_t *o, int iters, int64_t *ns) {
	try {
		for(int it = 0; it < iters; it++) {
			auto t0 = std::chrono::steady_clock::now();
			Decoder dec(len, blob);
			bind(dec, o);
			dec.decode();
			auto t1 = std::chrono::steady_clock::now();
			ns[it] = std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
		}
		return 0;
	} catch(const char *e) {
		g_err = e;
		return -1;
	}
}

// The same timed region on `nthreads` host threads at once, thread t decoding blob t % nblobs over and over for
// `seconds` (all blobs take outputs of the shape `o` describes; every thread has its own copy of the buffers).
// Returns the number of completed decodes (context figure for bench.py: all host cores vs one GPU).
int64_t ref_decode_mt(const uint8_t *const *blobs, const int *lens, int nblobs, const ref_out_t *o, uint32_t nvert, uint32_t nface,
                      int nthreads, double seconds) {
	std::vector<int64_t> done(nthreads, 0);
	std::vector<std::thread> th;
	auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double>(seconds);
	for(int t = 0; t < nthreads; t++) th.emplace_back([&, t]() {
		ref_out_t mine = *o;                                // private output buffers of the largest shape (nvert, nface)
		std::vector<std::vector<uint8_t>> bufs;
		auto own = [&](size_t bytes) { bufs.emplace_back(bytes + 64); return (void *)bufs.back().data(); };
		if(o->position) mine.position = (float *)own((size_t)nvert*12);
		if(o->normal) mine.normal = own((size_t)nvert*12);
		if(o->color) mine.color = (uint8_t *)own((size_t)nvert*4);
		if(o->uv) mine.uv = (float *)own((size_t)nvert*8);
		if(o->radius) mine.radius = (float *)own((size_t)nvert*4);
		if(o->index32) mine.index32 = (uint32_t *)own((size_t)nface*12);
		if(o->index16) mine.index16 = (uint16_t *)own((size_t)nface*6);
		const int b = t % nblobs;
		try {
			while(std::chrono::steady_clock::now() < t_end) {
				Decoder dec(lens[b], blobs[b]);
				bind(dec, &mine);
				dec.decode();
				done[t]++;
			}
		} catch(const char *) {}
	});
	for(auto &x : th) x.join();
	int64_t n = 0;
	for(auto d : done) n += d;
	return n;
}

// Stage hooks ------------------------------------------------------------------------------

// Tunstall dictionary (src/tunstall.cpp:125-256): probs = n x (symbol, probability) bytes as stored in the stream.
// index/lengths get 256 entries each; table gets *table_size bytes (cap 8192).
int ref_tunstall_tables(const uint8_t *probs, int n, int32_t *index, int32_t *lengths, uint8_t *table, int32_t *table_size) {
	Tunstall t;
	t.probabilities.resize(n);
	memcpy(t.probabilities.data(), probs, n*2);
	t.createDecodingTables2();
	if(n <= 1) { *table_size = 0; return 0; }
	for(int i = 0; i < 256; i++) { index[i] = t.index[i]; lengths[i] = t.lengths[i]; }
	// only the prefix of the 8192-byte buffer that words point into is meaningful
	int used = 0;
	for(int i = 0; i < 256; i++) used = std::max(used, t.index[i] + t.lengths[i]);
	memcpy(table, t.table.data(), used);
	*table_size = used;
	return 0;
}

// Tunstall decode of one block payload (src/tunstall.cpp:430-452).
int ref_tunstall_decompress(const uint8_t *probs, int n, const uint8_t *data, int csize, uint8_t *out, int size) {
	Tunstall t;
	t.probabilities.resize(n);
	memcpy(t.probabilities.data(), probs, n*2);
	t.createDecodingTables2();
	if(size) t.decompress((unsigned char *)data, csize, out, size);
	return 0;
}

// Tunstall compress one symbol stream exactly as OutStream::tunstall_compress (src/cstream.cpp:89-109):
// out = u8 nsym | nsym*(sym,prob) | u32 size | u32 csize | payload. Returns bytes written or -1.
int64_t ref_tunstall_compress_block(const uint8_t *symbols, int size, uint8_t *out, int64_t cap) {
	OutStream s;
	s.entropy = Stream::TUNSTALL;
	s.compress(size, (uchar *)symbols);
	if((int64_t)s.size() > cap) return -1;
	memcpy(out, s.data(), s.size());
	return s.size();
}

} // extern "C"
