// crt_format.h — host-side .crt container walk (no HIP, no device code).
//
// Restates WHAT the reference reads, in the order it reads it:
//   header        crt::Decoder::Decoder            src/decoder.cpp:41-89
//   groups        IndexAttribute::decodeGroups     include/corto/index_attribute.h:89-99
//   index block   IndexAttribute::decode           include/corto/index_attribute.h:83-87
//   attr blocks   GenericAttr::decode / NormalAttr::decode / ColorAttr::decode
//                 include/corto/vertex_attribute.h:153-158, src/normal_attribute.cpp:178-185,
//                 include/corto/color_attribute.h:55-59
//   framing       InStream::read(BitStream&), tunstall_decompress, decompress
//                 include/corto/cstream.h:283-291, src/cstream.cpp:66-87,111-128
// but without decoding anything: every block is self-describing, so one O(#streams) pass yields the
// byte extents the device kernels need, and is where ALL bounds validation happens (the reference
// never validates, cstream.h:226-229).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace corto_hip {

struct AttrHeader {
	std::string name;
	uint32_t codec = 1;      // 1 generic, 2 normal, 3 color (anything else -> generic)
	float q = 0.f;
	uint32_t N = 0, format = 0, strategy = 0;
};

struct BlobHeader {
	uint32_t version = 0, entropy = 0, nvert = 0, nface = 0, body_offset = 0;
	std::vector<std::pair<std::string, std::string>> exif;   // sorted by key (std::map order)
	std::vector<AttrHeader> attrs;                           // sorted by name, duplicates collapsed
};

enum StreamMode : uint32_t { STREAM_EMPTY = 0, STREAM_RAW = 1, STREAM_TUNSTALL = 2, STREAM_FILL = 3 };

// one entropy-coded byte array ("TUNSTALL(x)" / entropy NONE block)
struct StreamRef {
	uint32_t mode = STREAM_EMPTY;
	uint32_t nsym = 0;          // Tunstall: number of (symbol, probability) pairs
	uint32_t probs_off = 0;     // blob offset of the pairs
	uint32_t size = 0;          // decoded bytes
	uint32_t csize = 0;         // payload bytes
	uint32_t payload_off = 0;   // blob offset of payload (codewords, or raw bytes)
	uint32_t fill = 0;          // STREAM_FILL: the single symbol
	uint8_t probs16[32] = {0};  // Tunstall, nsym <= 16: the pairs themselves (streams of a batch with the same table share one dictionary, batch.cpp)
	uint8_t max_sym = 255;      // the largest symbol the stream can decode to (Tunstall: of its table; a fill: the symbol; raw bytes: unknown = 255): a log stream whose
	                            // widths are all <= 16 bits unpacks to values that fit int16 (plan_jobs.cpp)
};

struct BitsRef { uint32_t words_off = 0, nwords = 0; };   // words_off is 4-aligned relative to blob start

struct AttrStreams {
	BitsRef bits;
	std::vector<StreamRef> logs;      // 1 (CORRELATED / normal) or N (otherwise)
	uint32_t normal_prediction = 0;   // normals: 0 DIFF, 1 ESTIMATED, 2 BORDER
	uint32_t qc[4] = {4, 4, 4, 8};    // colours (include/corto/color_attribute.h:31-34 defaults)
};

struct BlobLayout {
	BlobHeader h;
	std::vector<uint32_t> group_end;
	std::vector<std::vector<std::pair<std::string, std::string>>> group_props;
	uint32_t max_front = 0;
	StreamRef clers;
	BitsRef split;
	std::vector<AttrStreams> attrs;   // parallel to h.attrs
	uint32_t end_offset = 0;          // one past the last byte consumed
};

// error codes are the CRTHIP_E_* values of include/corto_hip.h
int parse_header(const uint8_t *p, size_t len, BlobHeader &h);
int walk_blob(const uint8_t *p, size_t len, BlobLayout &L);      // L must be fresh, or reset_layout()
void reset_layout(BlobLayout &L);                                 // back to the default state, keeping every vector's capacity

} // namespace corto_hip
