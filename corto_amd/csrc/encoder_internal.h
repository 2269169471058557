// encoder_internal.h — pieces of the host encoder (encoder.cpp) and of the context (batch.cpp) that the GPU encoder
// stages (encode_gpu.cpp, k_encode.hip) use.  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "../../include/corto_hip.h"

namespace corto_hip {

// What Tunstall::compress walks (src/tunstall.cpp:384-428), made from a byte histogram exactly as
// getProbabilities + createDecodingTables2 + createEncodingTables make it (src/tunstall.cpp:83-115, 125-256, 335-382).
struct TunEncoderTables {
	uint32_t nsym = 0;
	uint8_t probs[512];                 // nsym x (symbol, probability), sorted as the reference sorts them
	uint8_t remap[256];                 // symbol -> index in probs
	uint16_t lengths[256];              // word lengths by codeword
	std::vector<int32_t> offsets;       // the 2-symbol-step trie: >= 0 codeword, < 0 minus the offset of the next level
};
void tun_encoder_tables(const uint32_t counts[256], uint32_t size, TunEncoderTables &out);

// value arrays for the device stages (encode_gpu.cpp); values are HOST pointers
struct EncValueStream { uint32_t kind = 0, count = 0, components = 1; const void *values = nullptr; };   // kind: CRTHIP_ENC_*
struct EncValueResult {
	std::vector<uint32_t> words;                    // the bit stream (empty for a symbol stream); the writer pads to 4 bytes before it
	std::vector<std::vector<uint8_t>> blocks;       // entropy-coded log arrays (1 for ARRAY, components for VALUES) or the symbol block
};
int encode_value_streams(crthip_ctx *ctx, uint32_t entropy, const std::vector<EncValueStream> &in, std::vector<EncValueResult> &res,
                         crthip_kernel_times *times);

// quantisation on the device (encode_gpu.cpp: k_enc_quantize): HOST arrays in, HOST arrays out, one upload / download for all of them
struct QuantRequest { uint32_t kind = 0, count = 0, N = 1; const void *in = nullptr; void *out = nullptr; float q = 0; int32_t unit = 0; uint32_t qc[4] = {1, 1, 1, 1}; };
int quantize_device(crthip_ctx *ctx, const std::vector<QuantRequest> &reqs);

// several blobs with HOST output buffers in one batch (batch.cpp): what crthip_decode_host is one of, and what the crt::Decoder facade's
// combiner hands over when several threads call decode() at once.  copy_out: copy every output into the caller's buffer before
// returning; else leave them in the context's pinned landing zone and say where (out_src -> out_dst, out_bytes): valid until the next
// call on this context - the facade lets every waiting thread copy its own.
struct HostDecodeReq {
	const uint8_t *blob; size_t len;
	const crthip_attr_binding *attrs;       // info.nattr entries in attribute order, or null: nothing bound
	void *index; uint32_t index_format;
	void *prediction;                       // host, nvert*3 uint32 or null: upstream's index.prediction (the context of a caller-supplied codec's deltaDecode)
	int32_t status;                         // out: CRTHIP_OK or this blob's CRTHIP_E_*
	uint32_t nout;                          // out (copy_out == false): pieces to copy
	const uint8_t *out_src[CRTHIP_MAX_ATTRS + 2]; void *out_dst[CRTHIP_MAX_ATTRS + 2]; size_t out_bytes[CRTHIP_MAX_ATTRS + 2];
};
int decode_host_many(crthip_ctx *ctx, uint32_t n, HostDecodeReq *reqs, bool copy_out);

// context plumbing (batch.cpp)
int ctx_fail(int code, const char *msg);
int ctx_device(crthip_ctx *ctx);
hipStream_t ctx_stream(crthip_ctx *ctx);
int ctx_quiesce(crthip_ctx *ctx);       // wait for whatever batch is in flight on the context
int ctx_fill_async(crthip_ctx *ctx, void *dst, size_t bytes, int value);   // k_fill_block on the context's main stream
int ctx_copy_to_host_async(crthip_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes);   // D2H behind the decode in flight; sync / done then cover the copy

} // namespace corto_hip
