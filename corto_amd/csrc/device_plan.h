// device_plan.h — POD descriptors shared by the host planner (batch.cpp) and the HIP kernels.
// All pointers are DEVICE addresses.  One array of each job type per batch, uploaded in one H2D copy.
#pragma once
#include <cstdint>
#include <cstdlib>

namespace corto_hip {

constexpr uint32_t TUN_TABLE_BYTES = 8192 + 1024;  // reference buffer is 8192 B (src/tunstall.cpp:137); +3 B padding per word
constexpr uint32_t TUN_ENTRY_CAP = 768;            // creation-order entries; the reference reaches <= 510
constexpr uint32_t CHUNK = 1024;                   // elements per scan chunk (256 threads x 4)

// Decode table of one Tunstall stream as the decode kernel wants it (K-TAB output)
struct TunTable {
	uint16_t off[256];             // word start in bytes[]            (Tunstall::index)
	uint8_t len[256];              // word length                      (Tunstall::lengths)
	uint32_t used;                 // bytes of bytes[] any word reaches
	uint32_t maxlen;               // longest word
	uint32_t pad[2];
	uint8_t bytes[TUN_TABLE_BYTES];
};

struct TunStream {
	const uint8_t *src;            // codewords
	uint8_t *dst;                  // decoded symbols
	const uint8_t *probs;          // nsym x (symbol, probability)
	uint32_t csize, size, nsym;
	uint32_t table;                // TunTable slot
	uint32_t chunk0;               // first entry of this stream in the chunk arrays (long streams)
	uint32_t nchunks;
	uint32_t chunk_codes;          // codewords per chunk (= per K-TUN workgroup), a multiple of 256*cpl
	uint32_t cpl;                  // staged decode: codewords per lane per step (8, 4, 2 or 1)
	uint32_t dict;                 // k_tun_stream_grouped: TunTable slot of the dictionary this stream decodes from
	uint32_t pad_;
};

// Chunk geometry of one stream, from its mean word length size/csize (both are in the stream header): workgroups are
// balanced by OUTPUT bytes (a two-symbol dictionary expands 60x, a flat one 2x), and a wave's step of 64*cpl codewords
// is sized so that its decoded bytes fit the wave's LDS window (k_tunstall.hip).
constexpr uint32_t TUN_CHUNK_CODES = 32768;   // largest chunk
inline void tun_pick_geometry(TunStream &t, uint32_t chunk_cap = TUN_CHUNK_CODES) {   // chunk_cap: DebugConfig::tun_chunk_cap (debug_config.h)
	const uint32_t avg = t.csize ? (uint32_t)(((uint64_t)t.size + t.csize - 1)/t.csize) : 1;
	t.cpl = avg <= 8 ? 8 : avg <= 16 ? 4 : avg <= 32 ? 2 : 1;
	uint32_t codes = chunk_cap;
	while(codes > 256*t.cpl && (uint64_t)codes*avg > 256*1024) codes >>= 1;   // a wave's quarter chunk stays whole steps of 64*cpl
	t.chunk_codes = codes;
	t.nchunks = (t.csize + codes - 1)/codes;
}

// streams [first, first + count) of the launch's by-dictionary order (an index array): all of one dictionary, decoded by one workgroup
struct TunGroup { uint32_t first, count; };
constexpr uint32_t TUN_GROUP_MAX = 8;

struct FillJob { uint8_t *dst; uint32_t size; uint32_t value; };

// CLERS automaton input/output of one mesh blob (src/decoder.cpp:204-358)
struct TopoJob {
	const uint8_t *clers;
	const uint32_t *split_words;
	const uint32_t *group_end;
	void *faces;                   // nface*3 u32 or u16
	uint32_t *pred;                // nvert*3
	uint4 *front_a;                // (v0, v1, v2, deleted)
	uint2 *front_b;                // (prev, next)
	uint32_t *order;               // FIFO of edge ids
	uint32_t *delayed;             // LIFO of edge ids
	int32_t *status;
	int32_t *flags;                // bit 0: the LDS path ran out of edge slots and the blob was redone on the HBM front
	uint32_t nclers, split_nwords, ngroups;
	uint32_t nvert, nface, front_cap, faces_u16;
	uint32_t pad;
	// LDS path (k_mesh.hip): ring of queued-edge records (a power of two; 0: not eligible), pool of surviving-edge records,
	// DELAY stack entries, symbols in the LDS window (a multiple of 8)
	uint32_t lds_ring, lds_pool, lds_delayed_cap, lds_symwin;
};
// TopoJob.pad on the device, bit 1: the automaton keeps a PROGRESS WORD for a consumer that runs BESIDE it (k_delta_tiles on a lone context's second
// stream) - the number of vertices whose prediction triple has been written and has arrived (published every 4 096 vertices and at every slide of the
// symbol window), 0xFFFFFFFF when it is done.  The word sits 16 bytes in front of the triples (zeroed by the host before the launch): no pointer of its
// own - the automaton's ISA block has no scalar register to spare for one (a TopoJob eight bytes longer cost the C4 batch's automata 3 %).
constexpr uint32_t TOPO_PAD_PROGRESS = 2u;
constexpr uint32_t TOPO_PROGRESS_BYTES = 16u;

// one log stream to turn into values (include/corto/cstream.h:294-360)
struct UnpackJob {
	const uint8_t *logs;
	const uint32_t *words;
	void *out;                     // int32* or uint8*
	uint32_t count;                // logs in the stream
	uint32_t nwords;
	uint32_t out_limit;            // never write element index >= out_limit (nvert)
	uint32_t chunk0;               // first chunk of this job in the chunk arrays
	uint32_t chain_chunk0;         // first chunk of the bit block this job shares (component-major chaining); k_unpack_wave: the first JOB of the bit block
	uint16_t fields;               // ARRAY: N fields per log; VALUES: 1
	uint16_t stride;               // output elements per vertex
	uint16_t comp;                 // VALUES: component
	uint8_t mode;                  // 0 ARRAY, 1 VALUES
	uint8_t out_u8;                // 1: bytes (colours); 2: int16 (k_unpack_wave only: a stream whose table holds no width above 16 bits, read by k_delta_lds16 / k_normal_blob)
};

constexpr uint32_t UNPACK_WAVE_MAX_LOGS = 16384;   // bit blocks of at most this many logs (all streams together): one wave per stream (k_unpack_wave); else chunks of 1 024 with look-back

// parallelogram / first-neighbour delta over a mesh (include/corto/vertex_attribute.h:160-176,
// src/normal_attribute.cpp:193-201)
struct DeltaJob {
	void *values;                  // int32* or uint8*, stride N
	const uint32_t *pred;
	uint8_t *fired;                // k_delta_mesh: nvert zeroed flags in HBM (+ the stretch starts behind them); k_delta_tiles: the automaton's progress word (TopoJob.progress) or null
	uint32_t nvert, N;
	uint8_t parallelogram, is_u8, pad[2];   // pad[1]: k_delta_lds16 keeps 32-bit records in LDS (the whole group says the same); pad[0] (device): `values` holds the raw
	                               // deltas as int16 (K-BIT's UnpackJob.out_u8 == 2), packed at the front of the buffer the results go to
	// k_delta_lds16 finishes the attribute on its way out of LDS (no k_dequant launch, no second trip through HBM):
	uint32_t deq;                  // 0: write the integers back; 1: generic, packed: (float)v*q in place (vertex_attribute.h:190-193);
	                               // 2: colour: YCC -> RGB x qc into `out` (color_attribute.cpp:76-95)
	float q;
	uint32_t qc[4];
	void *out;                     // colour destination
	uint32_t out_components, out_stride;
	int32_t *flags;                // k_delta_lds16: set to 1 when the attribute's values relative to vertex 0 left int16 and were redone in HBM
	uint32_t pad2[2];              // pad2[0]: the round loop from vertex 1 (test hook: $CORTO_DELTA_ROUNDS); pad2[1]: words from the blob's status word to `flags` (k_delta_tiles)
};

// point-cloud running sum, one job per (blob, attribute) (vertex_attribute.h:177-181, normal_attribute.cpp:202-207)
struct CloudJob {
	void *values;
	uint32_t nvert, N;
	uint32_t chunk0;               // N * ceil(nvert/CHUNK) chunks: component-major
	uint8_t is_u8, pad[3];
};

struct NormalJob {
	int32_t *diffs;                // 2 ints per (corrected) vertex
	void *out;                     // nvert*3 f32 or i16
	const int32_t *position;       // delta-decoded integer positions (3 per vertex), ESTIMATED/BORDER only
	const void *faces;
	uint32_t nvert, nface, ndiffs;
	uint32_t vbase, fbase;         // offsets into the batch-wide per-vertex / per-face scratch arrays
	int32_t unit;                  // (int)q
	uint8_t prediction, out_i16, faces_u16, fused;   // fused: handled by k_normal_blob (whole pipeline in one workgroup)
	int32_t *status;
	uint32_t out_stride;           // bytes from one vertex's normal to the next (12 or 6 when packed)
	// k_normal_blob is the last reader of the integer positions and turns them into floats itself (no k_dequant launch):
	uint32_t pos_stride;           // bytes from one vertex to the next in pos_out (12 = packed, in place when pos_out == position)
	void *pos_out;                 // null: leave the positions alone
	float pos_q;
	uint32_t diffs_i16;            // k_normal_blob: `diffs` holds int16 pairs (K-BIT's UnpackJob.out_u8 == 2)
	float *fn_scratch;             // k_normal_blob without its face normals in LDS: 3 floats per face in HBM scratch (null: recompute them per incident vertex)
};

struct DequantJob {
	void *buffer;                  // destination (generic, packed: the int32 values are here already and turn into floats in place)
	const uint8_t *src;            // colour: delta-decoded N-component bytes; generic with a stride: the int32 values (packed scratch)
	float q;
	uint32_t nvert, N, out_components;
	uint32_t qc[4];
	uint32_t block0;               // first block of this job in the block->job map
	uint8_t is_color;
	uint8_t format;                // generic attributes: the CRTHIP_FMT_* of `buffer` (FLOAT: (float)v*q; the integer formats and DOUBLE: upstream's
	                               // in-place "*= q" through a pointer of that type, vertex_attribute.h:195-228 - k_dequant restates what the compiled reference does)
	uint8_t pad[2];
	uint32_t stride;               // bytes from one vertex to the next in `buffer`; 0 = packed
	uint32_t pad2;
};

// ---- encoder stages (k_encode.hip, encode_gpu.cpp) ----
// a piece of a source stream for the byte histogram (src/tunstall.cpp:83-115)
struct EncChunk { const uint8_t *src; uint32_t size, stream; };

// one stream for the Tunstall coder (src/tunstall.cpp:384-428): the host-made encoder tables and where the codewords go
struct EncStream {
	const uint8_t *src;            // size symbols
	uint8_t *dst;                  // room for size + 64 codewords
	int16_t *trie;                 // the reference's 2-symbol-step trie, levels of nsym*nsym entries: >= 0 codeword, < 0 minus the next level's number
	                               // (made on the host, or by k_enc_trie in place: then ntrie is what that kernel wrote)
	const uint8_t *remap;          // 256: symbol -> index
	const uint16_t *lengths;       // 256: word length by codeword
	uint32_t *csize;               // out: number of codewords
	uint32_t size, nsym, ntrie, pad;
};
// one attribute to quantise (k_enc_quantize): include/corto/vertex_attribute.h:79-128, src/normal_attribute.cpp:61-111,
// src/color_attribute.cpp:23-70
struct QuantJob {
	const void *in;                // GENERIC: count floats; NORMAL: count x 3 floats; COLOR: count x N bytes
	void *out;                     // GENERIC: count int32; NORMAL: count x 2 int32 (octahedral); COLOR: count x N bytes (YCC)
	uint32_t count, kind, N;       // kind: 0 GENERIC, 1 NORMAL, 2 COLOR
	float q;                       // GENERIC: the step
	int32_t unit;                  // NORMAL: (int)q
	uint32_t qc[4];                // COLOR: per-channel divisors
};
// what k_enc_tables leaves per stream for the Tunstall coder: the block header (probabilities) and the encoder tables
struct EncTab {
	uint32_t nsym;                 // symbols that occur (0: empty stream, 1: no payload)
	uint32_t level_bound;          // upper bound on the levels of the stream's trie: 1 + sum over words of (length - 1)/2
	uint32_t used;                 // dictionary bytes
	uint32_t pad;
	uint8_t probs[512];            // nsym x (symbol, probability) in the reference's order (std::sort of src/tunstall.cpp:111-112)
	uint8_t remap[256];            // symbol -> index in probs
	uint16_t lengths[256];         // word lengths by codeword
	uint16_t index[256];           // word starts in words[]
	uint8_t words[TUN_TABLE_BYTES];
};
// one value array for the bit-width + bit-packing kernel (include/corto/cstream.h:115-164)
struct PackJob {
	const void *values;            // count*N int32 (ARRAY, VALUES_I32) or int8 (VALUES_I8)
	uint8_t *logs;                 // ARRAY: count bytes; VALUES: N arrays of count bytes, component-major
	uint32_t *words;               // MSB-first bit stream (src/bitstream.cpp:86-101)
	uint32_t *nwords;              // out
	uint32_t count, N, kind, pad;  // kind: CRTHIP_ENC_ARRAY / VALUES_I32 / VALUES_I8
};
constexpr uint32_t ENC_PACK_MAX_N = 16;                          // components per element the packing tile is sized for
constexpr uint32_t ENC_STAGE = 4096, ENC_STAGE_PAD = 576;      // bytes staged in LDS per refill; look-ahead a 64-byte window may need (words <= 255 symbols)
constexpr uint32_t ENC_HIST_CHUNK = 1u << 18;

} // namespace corto_hip
