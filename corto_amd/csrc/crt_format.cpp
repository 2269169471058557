// crt_format.cpp — see crt_format.h. Host only.
#include "crt_format.h"

#include <algorithm>
#include <cstring>

#include "../../include/corto_hip.h"

namespace corto_hip {
namespace {

// bounds-checked little-endian cursor (the reference's readers: include/corto/cstream.h:240-291)
struct Cursor {
	const uint8_t *p;
	size_t len, pos = 0;
	bool bad = false;
	bool need(size_t n) {
		if(bad || n > len || pos > len - n) { bad = true; return false; }
		return true;
	}
	uint32_t u8() { return need(1) ? p[pos++] : 0; }
	uint32_t u16() { if(!need(2)) return 0; uint32_t v = p[pos] | (p[pos+1] << 8); pos += 2; return v; }
	uint32_t u32() {
		if(!need(4)) return 0;
		uint32_t v = (uint32_t)p[pos] | ((uint32_t)p[pos+1] << 8) | ((uint32_t)p[pos+2] << 16) | ((uint32_t)p[pos+3] << 24);
		pos += 4; return v;
	}
	float f32() { uint32_t u = u32(); float f; std::memcpy(&f, &u, 4); return f; }
	std::string str() {                  // u16 byte count incl. NUL + bytes (cstream.h:277-280)
		uint32_t n = u16();
		if(!need(n)) return std::string();
		const char *s = (const char *)(p + pos); pos += n;
		return std::string(s, strnlen(s, n));
	}
	bool skip(size_t n) { if(!need(n)) return false; pos += n; return true; }
};

int header(Cursor &c, BlobHeader &h) {
	if(((uintptr_t)c.p) & 3) return CRTHIP_E_ALIGN;            // decoder.cpp:43-44
	if(c.u32() != 0x787A6300u || c.bad) return CRTHIP_E_MAGIC; // decoder.cpp:48-52
	h.version = c.u32();
	h.entropy = c.u8();
	uint32_t nexif = c.u32();
	for(uint32_t i = 0; i < nexif && !c.bad; i++) {
		std::string k = c.str(), v = c.str();
		auto it = std::lower_bound(h.exif.begin(), h.exif.end(), k, [](const auto &a, const std::string &b) { return a.first < b; });
		if(it != h.exif.end() && it->first == k) it->second = v; else h.exif.insert(it, {k, v});
	}
	uint32_t nattr = c.u32();
	for(uint32_t i = 0; i < nattr && !c.bad; i++) {
		AttrHeader a;
		a.name = c.str();
		a.codec = c.u32(); a.q = c.f32(); a.N = c.u8(); a.format = c.u8(); a.strategy = c.u8();
		if(a.codec != CRTHIP_CODEC_NORMAL && a.codec != CRTHIP_CODEC_COLOR) a.codec = CRTHIP_CODEC_GENERIC;  // decoder.cpp:73-80
		auto it = std::lower_bound(h.attrs.begin(), h.attrs.end(), a.name, [](const AttrHeader &x, const std::string &b) { return x.name < b; });
		if(it != h.attrs.end() && it->name == a.name) *it = a; else h.attrs.insert(it, a);   // std::map semantics, decoder.cpp:85
	}
	h.nvert = c.u32();
	h.nface = c.u32();
	h.body_offset = (uint32_t)c.pos;
	if(c.bad) return CRTHIP_E_TRUNCATED;
	if(h.attrs.size() > CRTHIP_MAX_ATTRS) return CRTHIP_E_LIMIT;
	for(auto &a : h.attrs) if(a.name.size() >= CRTHIP_NAME_MAX) return CRTHIP_E_LIMIT;
	return CRTHIP_OK;
}

// "BITS" block: u32 nwords | zero pad to a 4-byte offset from blob start | words   (cstream.h:283-291)
BitsRef bits_block(Cursor &c) {
	BitsRef b;
	b.nwords = c.u32();
	size_t pad = c.pos & 3;
	if(pad) c.skip(4 - pad);
	b.words_off = (uint32_t)c.pos;
	c.skip((size_t)b.nwords * 4);
	return b;
}

// entropy-coded byte array (cstream.cpp:66-87, 111-128)
StreamRef byte_block(Cursor &c, uint32_t entropy, int &err) {
	StreamRef s;
	if(entropy == CRTHIP_ENTROPY_NONE) {
		s.size = s.csize = c.u32();
		s.payload_off = (uint32_t)c.pos;
		c.skip(s.size);
		s.mode = s.size ? STREAM_RAW : STREAM_EMPTY;
		if(!s.size) s.max_sym = 0;
		return s;
	}
	if(entropy != CRTHIP_ENTROPY_TUNSTALL) { err = CRTHIP_E_ENTROPY; return s; }
	s.nsym = c.u8();
	s.probs_off = (uint32_t)c.pos;
	if(c.need((size_t)s.nsym * 2) && s.nsym >= 1) s.fill = c.p[c.pos];
	if(c.need((size_t)s.nsym * 2) && s.nsym >= 2 && s.nsym <= 16) memcpy(s.probs16, c.p + c.pos, (size_t)s.nsym * 2);
	if(c.need((size_t)s.nsym * 2) && s.nsym >= 1) { uint8_t m = 0; for(uint32_t k = 0; k < s.nsym; k++) m = std::max(m, c.p[c.pos + 2*(size_t)k]); s.max_sym = m; }
	c.skip((size_t)s.nsym * 2);
	s.size = c.u32();
	s.csize = c.u32();
	s.payload_off = (uint32_t)c.pos;
	c.skip(s.csize);
	if(s.size == 0) s.mode = STREAM_EMPTY;                 // "if(size)" cstream.cpp:126
	else if(s.nsym == 1) s.mode = STREAM_FILL;             // memset path, tunstall.cpp:433-436
	else if(s.nsym == 0 || s.csize == 0) { s.mode = STREAM_EMPTY; err = err ? err : CRTHIP_E_TRUNCATED; }  // reference would read out of bounds
	else s.mode = STREAM_TUNSTALL;
	return s;
}

} // namespace

int parse_header(const uint8_t *p, size_t len, BlobHeader &h) {
	Cursor c{p, len};
	return header(c, h);
}

void reset_layout(BlobLayout &L) {
	L.h.version = L.h.entropy = L.h.nvert = L.h.nface = L.h.body_offset = 0;
	L.h.exif.clear(); L.h.attrs.clear();
	L.group_end.clear();
	for(auto &g : L.group_props) g.clear();
	L.max_front = 0; L.clers = StreamRef(); L.split = BitsRef();
	for(auto &a : L.attrs) { a.bits = BitsRef(); a.logs.clear(); a.normal_prediction = 0; a.qc[0] = 4; a.qc[1] = 4; a.qc[2] = 4; a.qc[3] = 8; }
	L.end_offset = 0;
}

int walk_blob(const uint8_t *p, size_t len, BlobLayout &L) {
	Cursor c{p, len};
	int err = header(c, L.h);
	if(err) return err;
	const uint32_t entropy = L.h.entropy;

	uint32_t ngroups = c.u32();                              // index_attribute.h:89-99
	if(!c.need((size_t)ngroups * 5)) return CRTHIP_E_TRUNCATED;
	L.group_end.resize(ngroups);
	L.group_props.resize(ngroups);
	for(uint32_t g = 0; g < ngroups && !c.bad; g++) {
		L.group_end[g] = c.u32();
		uint32_t np = c.u8();
		for(uint32_t k = 0; k < np && !c.bad; k++) { std::string key = c.str(), val = c.str(); L.group_props[g].push_back({key, val}); }
	}
	if(L.h.nface > 0) {                                      // index_attribute.h:83-87
		L.max_front = c.u32();
		L.clers = byte_block(c, entropy, err);
		L.split = bits_block(c);
	}
	L.attrs.resize(L.h.attrs.size());
	for(size_t i = 0; i < L.h.attrs.size() && !c.bad && !err; i++) {
		const AttrHeader &a = L.h.attrs[i];
		AttrStreams &s = L.attrs[i];
		if(a.codec == CRTHIP_CODEC_NORMAL) {                 // normal_attribute.cpp:178-185
			s.normal_prediction = c.u8();
			s.bits = bits_block(c);
			s.logs.push_back(byte_block(c, entropy, err));
		} else if(a.codec == CRTHIP_CODEC_COLOR) {           // color_attribute.h:55-59
			for(uint32_t k = 0; k < a.N; k++) { uint32_t q = c.u8(); if(k < 4) s.qc[k] = q; }
			s.bits = bits_block(c);
			for(uint32_t k = 0; k < a.N; k++) s.logs.push_back(byte_block(c, entropy, err));
		} else if(a.strategy & CRTHIP_CORRELATED) {          // decodeArray, cstream.h:324-360
			s.bits = bits_block(c);
			s.logs.push_back(byte_block(c, entropy, err));
		} else {                                             // decodeValues, cstream.h:294-319
			s.bits = bits_block(c);
			for(uint32_t k = 0; k < a.N; k++) s.logs.push_back(byte_block(c, entropy, err));
		}
	}
	if(err) return err;
	if(c.bad) return CRTHIP_E_TRUNCATED;
	L.end_offset = (uint32_t)c.pos;
	return CRTHIP_OK;
}

} // namespace corto_hip
