// include/corto/decoder.h — drop-in crt::Decoder for the MI355X path.
//
// Same class name, public members and method signatures as upstream's include/corto/decoder.h:38-73,
// so code written against libcorto's decoder compiles unchanged against this header and links with
// libcorto_hip.so instead of libcorto.  decode() runs on the GPU through the C ABI (corto_hip.h);
// output buffers are the caller's HOST buffers exactly as upstream (ownership rules: SURVEY.md §8b).
//
// Differences, all deliberate and loud:
//   * there is no CPU path: without a usable HIP device decode() throws;
//   * VertexAttribute here describes an attribute; the built-in codecs (generic, normal, colour) run on the device.  A caller-supplied
//     codec OBJECT (setAttribute(name, buffer, VertexAttribute *), src/decoder.cpp:104-114) is host code: the device decodes the
//     attribute's stream into the buffer as int32 values (what upstream's GenericAttr<int>::decode leaves there,
//     include/corto/vertex_attribute.h:151-156 - a custom object therefore keeps the generic STREAM coding and overrides what comes
//     after it), then decode() calls the object's deltaDecode(nvert, index.prediction), postDelta(...) and dequantize(nvert) on the
//     host, in upstream's order (src/decoder.cpp:186-193).  Other attributes are final by then (upstream: still quantised integers);
//     ESTIMATED / BORDER normals over a position that has a custom object throw, as upstream's do (src/normal_attribute.cpp:210-213);
//   * generic attributes take every VertexAttribute::Format through setAttribute(name, buffer, format): FLOAT is the format upstream's own
//     callers use; the integer formats and DOUBLE leave in the buffer what the compiled reference leaves there (its "*= q" through a pointer
//     of the output type over the int32 array, vertex_attribute.h:195-228 - buffers of nvert*N*4 bytes, nvert*N*8 for DOUBLE, as upstream);
//     colours decode to UINT8 (FLOAT colour output is broken upstream, color_attribute.cpp:96-110), normals to FLOAT or INT16.
// Errors are thrown as `const char *` with upstream's own messages (src/decoder.cpp:44,51,274 ...).
//
// Threading: as upstream (whose only global is the read-only bmask[], src/bitstream.cpp:29-33) - distinct Decoder objects may
// be constructed and decode() on different threads concurrently; one object is not re-entrant and decode() is one-shot.
// decode() borrows one of a process-wide pool of device contexts (at most $CORTO_HIP_CONTEXTS, default 8, created on
// demand; each owns its HIP streams, device scratch and pinned staging, reused from call to call), so up to that many
// decodes overlap on the GPU and further callers wait for a free one.  setDevice() may be called at any time: decodes
// already running finish on the old device.
#ifndef CRT_HIP_DECODER_H
#define CRT_HIP_DECODER_H

#include <cstdint>
#include <map>
#include <string>
#include <vector>

typedef unsigned char uchar;

namespace crt {

// upstream include/corto/index_attribute.h:34-38: a vertex' prediction triple (decode order)
struct Face {
	uint32_t a, b, c;
	Face() {}
	Face(uint32_t v0, uint32_t v1, uint32_t v2): a(v0), b(v1), c(v2) {}
};
class IndexAttribute;

// upstream include/corto/vertex_attribute.h:30-66: the data members, and - for caller-supplied codec objects - the decode-side virtuals that
// are host code (decode(nvert, InStream &) itself is the device's: the generic stream coding)
class VertexAttribute {
public:
	enum Format { UINT32 = 0, INT32, UINT16, INT16, UINT8, INT8, FLOAT, DOUBLE };
	enum Strategy { PARALLEL = 0x1, CORRELATED = 0x2 };
	enum CODEC { GENERIC_CODEC = 1, NORMAL_CODEC = 2, COLOR_CODEC = 3, CUSTOM_CODEC = 100 };

	char *buffer = nullptr;     // output buffer (host), set by the set* calls
	int N = 0;                  // number of components as stored
	float q = 0.0f;             // quantisation step
	int strategy = 0;
	Format format = INT32;      // output format
	uint32_t size = 0;
	int bits = 0;
	int codec_id = GENERIC_CODEC;
	int out_components = 4;     // colours only (upstream ColorAttr::out_components)
	virtual ~VertexAttribute() {}
	virtual int codec() { return codec_id; }
	// the host half of a caller-supplied codec object (vertex_attribute.h:58-65); the defaults do nothing: the built-in codecs' run on the device
	virtual void deltaDecode(uint32_t /*nvert*/, std::vector<Face> & /*context*/) {}
	virtual void postDelta(uint32_t /*nvert*/, uint32_t /*nface*/, std::map<std::string, VertexAttribute *> & /*attrs*/, IndexAttribute & /*index*/) {}
	virtual void dequantize(uint32_t /*nvert*/) {}
};

// upstream include/corto/index_attribute.h:40-60 (what callers touch)
struct Group {
	uint32_t end = 0;           // 1 + last face
	std::map<std::string, std::string> properties;
};

class IndexAttribute {
public:
	uint32_t *faces32 = nullptr;
	uint16_t *faces16 = nullptr;
	std::vector<Face> prediction;   // filled by decode() when an attribute has a caller-supplied codec object (upstream fills it always)
	std::vector<Group> groups;
	uint32_t max_front = 0;
};

class Decoder {
public:
	uint32_t nvert, nface;
	std::map<std::string, std::string> exif;
	std::map<std::string, VertexAttribute *> data;
	IndexAttribute index;

	Decoder(int len, const uchar *input);                // src/decoder.cpp:41-89
	~Decoder();
	Decoder(const Decoder &) = delete;
	Decoder &operator=(const Decoder &) = delete;

	bool hasAttr(const char *name) { return data.count(name) != 0; }

	bool setPositions(float *buffer) { return setAttribute("position", (char *)buffer, VertexAttribute::FLOAT); }
	bool setNormals(float *buffer) { return setAttribute("normal", (char *)buffer, VertexAttribute::FLOAT); }
	bool setNormals(int16_t *buffer) { return setAttribute("normal", (char *)buffer, VertexAttribute::INT16); }
	bool setUvs(float *buffer) { return setAttribute("uv", (char *)buffer, VertexAttribute::FLOAT); }
	bool setColors(uchar *buffer, int components = 4);

	bool setAttribute(const char *name, char *buffer, VertexAttribute::Format format);
	bool setAttribute(const char *name, char *buffer, VertexAttribute *attr);   // a caller-supplied codec object (owned by the Decoder from here on, as upstream): see the top of this file

	void setIndex(uint32_t *buffer) { index.faces32 = buffer; }
	void setIndex(uint16_t *buffer) { index.faces16 = buffer; }

	void decode();                                       // src/decoder.cpp:126-196, on the GPU

	// which HIP device the process-wide context pool of this facade uses (default 0, or $CORTO_HIP_DEVICE)
	static void setDevice(int device);

private:
	const uchar *input_;
	int len_;
};

} // namespace crt
#endif // CRT_HIP_DECODER_H
