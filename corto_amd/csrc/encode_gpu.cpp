// encode_gpu.cpp — host side of the GPU encoder stages: crthip_tunstall_encode_blocks.
//
// The entropy coder of crt::Encoder for a batch of byte streams, i.e. OutStream::tunstall_compress
// (src/cstream.cpp:89-109) many times over:  histogram on the device (k_enc_hist)  ->  per stream, on the host, the few
// microseconds of serial work that depend on std::sort's order of equal probabilities: probabilities, 256-word dictionary,
// encoding trie (encoder.cpp: tun_encoder_tables = src/tunstall.cpp:83-115, 125-256, 335-382)  ->  greedy parse on the
// device (k_enc_tun_parse = src/tunstall.cpp:384-428)  ->  block framing on the host.  Every block is byte-identical to
// the reference's (tests/test_gpu_parity.py::test_tunstall_encode_*).  No CPU fallback: without a device the call fails.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/corto_hip.h"
#include "device_plan.h"
#include "encoder_internal.h"
#include "kernels.h"

using namespace corto_hip;

#define ENC_TRY(expr) do { hipError_t e_ = (expr); if(e_ != hipSuccess) { cleanup(); return ctx_fail(CRTHIP_E_DEVICE, (std::string(#expr ": ") + hipGetErrorString(e_)).c_str()); } } while(0)

extern "C" int64_t crthip_tunstall_encode_blocks(crthip_ctx *ctx, uint32_t n, const uint8_t *const *src, const uint32_t *sizes,
                                                 uint8_t *out, size_t cap, uint64_t *block_offset, crthip_kernel_times *times) {
	if(!ctx || (n && (!src || !sizes)) || !block_offset) return ctx_fail(CRTHIP_E_ARGUMENT, "crthip_tunstall_encode_blocks: null argument");
	for(uint32_t i = 0; i < n; i++) {
		if(sizes[i] && !src[i]) return ctx_fail(CRTHIP_E_ARGUMENT, "crthip_tunstall_encode_blocks: null stream");
		if(sizes[i] > (1u << 23)) return ctx_fail(CRTHIP_E_LIMIT, "crthip_tunstall_encode_blocks: stream longer than 2^23 symbols (the reference's count*255 overflows int)");
	}
	void *d_all = nullptr;
	hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
	auto cleanup = [&]() { if(d_all) (void)hipFree(d_all); d_all = nullptr; for(auto &e : ev) if(e) { (void)hipEventDestroy(e); e = nullptr; } };
	if(hipSetDevice(ctx_device(ctx)) != hipSuccess) return ctx_fail(CRTHIP_E_DEVICE, "hipSetDevice");
	{ const int e = ctx_quiesce(ctx); if(e) return e; }
	hipStream_t st = ctx_stream(ctx);
	if(times) memset(times, 0, sizeof(*times));

	// device image: sources | counts | codewords | csize | tables (after the histogram)
	std::vector<uint64_t> src_off(n), dst_off(n);
	uint64_t o = 0;
	for(uint32_t i = 0; i < n; i++) { src_off[i] = o; o += ((uint64_t)sizes[i] + 15) & ~15ull; }
	const uint64_t o_counts = o; o += (uint64_t)n*256*4;
	for(uint32_t i = 0; i < n; i++) { dst_off[i] = o; o += ((uint64_t)sizes[i] + 64 + 15) & ~15ull; }
	const uint64_t o_csize = o; o += ((uint64_t)n*4 + 15) & ~15ull;
	const uint64_t o_chunks = o;
	std::vector<EncChunk> chunks;
	for(uint32_t i = 0; i < n; i++)
		for(uint32_t b = 0; b < sizes[i]; b += ENC_HIST_CHUNK) chunks.push_back(EncChunk{nullptr, std::min(ENC_HIST_CHUNK, sizes[i] - b), i});
	o += (chunks.size()*sizeof(EncChunk) + 15) & ~15ull;
	const uint64_t fixed_bytes = o;

	// sources go up in one copy
	std::vector<uint8_t> h_src(o_counts ? o_counts : 16);
	for(uint32_t i = 0; i < n; i++) if(sizes[i]) memcpy(h_src.data() + src_off[i], src[i], sizes[i]);

	// the tables are sized after the histogram; reserve generously: nothing is known yet, so allocate them separately below
	ENC_TRY(hipMalloc(&d_all, fixed_bytes + 16));
	uint8_t *base = (uint8_t *)d_all;
	for(int k = 0; k < 4; k++) ENC_TRY(hipEventCreate(&ev[k]));
	if(o_counts) ENC_TRY(hipMemcpyAsync(base, h_src.data(), o_counts, hipMemcpyHostToDevice, st));
	if(n) ENC_TRY(hipMemsetAsync(base + o_counts, 0, (size_t)n*256*4, st));
	if(n) ENC_TRY(hipMemsetAsync(base + o_csize, 0, (size_t)(o_chunks - o_csize), st));
	{
		size_t c = 0;
		for(uint32_t i = 0; i < n; i++)
			for(uint32_t b = 0; b < sizes[i]; b += ENC_HIST_CHUNK) chunks[c++].src = base + src_off[i] + b;
	}
	std::vector<uint32_t> counts((size_t)n*256);
	if(!chunks.empty()) {
		ENC_TRY(hipMemcpyAsync(base + o_chunks, chunks.data(), chunks.size()*sizeof(EncChunk), hipMemcpyHostToDevice, st));
		ENC_TRY(hipEventRecord(ev[0], st));
		hipLaunchKernelGGL(k_enc_hist, dim3((uint32_t)chunks.size()), dim3(256), 0, st, (const EncChunk *)(base + o_chunks), (uint32_t)chunks.size(), (uint32_t *)(base + o_counts));
		ENC_TRY(hipEventRecord(ev[1], st));
		ENC_TRY(hipMemcpyAsync(counts.data(), base + o_counts, counts.size()*4, hipMemcpyDeviceToHost, st));
	}
	ENC_TRY(hipStreamSynchronize(st));

	// host: probabilities, dictionary, trie of every stream
	std::vector<TunEncoderTables> tabs(n);
	std::vector<uint32_t> gpu_ids;
	std::vector<uint64_t> tab_off(n, 0);
	uint64_t tbytes = 0;
	uint32_t trie_lds = 0;
	for(uint32_t i = 0; i < n; i++) {
		if(sizes[i] == 0) continue;
		tun_encoder_tables(&counts[(size_t)i*256], sizes[i], tabs[i]);
		if(tabs[i].nsym < 2) continue;                                // one symbol: no payload (tunstall.cpp:386-389)
		gpu_ids.push_back(i);
		tab_off[i] = tbytes;
		tbytes += 256 + 512 + ((tabs[i].offsets.size()*2 + 15) & ~15ull);
		if(tabs[i].offsets.size() <= ENC_TRIE_LDS_MAX) trie_lds = std::max<uint32_t>(trie_lds, (uint32_t)tabs[i].offsets.size());
	}
	void *d_tab = nullptr;
	auto cleanup2 = [&]() { if(d_tab) (void)hipFree(d_tab); d_tab = nullptr; cleanup(); };
#undef ENC_TRY
#define ENC_TRY(expr) do { hipError_t e_ = (expr); if(e_ != hipSuccess) { cleanup2(); return ctx_fail(CRTHIP_E_DEVICE, (std::string(#expr ": ") + hipGetErrorString(e_)).c_str()); } } while(0)
	std::vector<uint32_t> csize(n, 0);
	std::vector<uint8_t> h_codes;
	if(!gpu_ids.empty()) {
		const uint64_t o_streams = tbytes;
		const uint64_t tab_total = tbytes + gpu_ids.size()*sizeof(EncStream) + 16;
		std::vector<uint8_t> h_tab(tab_total);
		ENC_TRY(hipMalloc(&d_tab, tab_total));
		uint8_t *tb = (uint8_t *)d_tab;
		std::vector<EncStream> es;
		for(uint32_t i : gpu_ids) {
			const TunEncoderTables &T = tabs[i];
			uint8_t *h = h_tab.data() + tab_off[i];
			memcpy(h, T.remap, 256);
			memcpy(h + 256, T.lengths, 512);
			int16_t *t16 = (int16_t *)(h + 768);
			const int32_t span = (int32_t)(T.nsym*T.nsym);
			for(size_t k = 0; k < T.offsets.size(); k++) {
				const int32_t v = T.offsets[k];
				t16[k] = v >= 0 ? (int16_t)(v & 255) : (int16_t)-((-v)/span);   // codeword (the reference emits (uchar)off), or minus the level number
			}
			EncStream s{};
			s.src = base + src_off[i]; s.dst = base + dst_off[i];
			s.remap = tb + tab_off[i]; s.lengths = (const uint16_t *)(tb + tab_off[i] + 256); s.trie = (const int16_t *)(tb + tab_off[i] + 768);
			s.csize = (uint32_t *)(base + o_csize) + i;
			s.size = sizes[i]; s.nsym = T.nsym; s.ntrie = (uint32_t)T.offsets.size();
			es.push_back(s);
		}
		memcpy(h_tab.data() + o_streams, es.data(), es.size()*sizeof(EncStream));
		ENC_TRY(hipMemcpyAsync(d_tab, h_tab.data(), tab_total, hipMemcpyHostToDevice, st));
		const uint32_t lds = enc_parse_lds(trie_lds);
		static bool attr_set = false;
		if(lds > 64*1024 && !attr_set) { ENC_TRY(hipFuncSetAttribute((const void *)k_enc_tun_parse, hipFuncAttributeMaxDynamicSharedMemorySize, (int)enc_parse_lds(ENC_TRIE_LDS_MAX))); attr_set = true; }
		ENC_TRY(hipEventRecord(ev[2], st));
		hipLaunchKernelGGL(k_enc_tun_parse, dim3((uint32_t)es.size()), dim3(64), lds, st, (const EncStream *)(tb + o_streams), (uint32_t)es.size(), trie_lds);
		ENC_TRY(hipEventRecord(ev[3], st));
		ENC_TRY(hipMemcpyAsync(csize.data(), base + o_csize, (size_t)n*4, hipMemcpyDeviceToHost, st));
		h_codes.resize(o_csize - dst_off[0]);
		ENC_TRY(hipMemcpyAsync(h_codes.data(), base + dst_off[0], h_codes.size(), hipMemcpyDeviceToHost, st));
		ENC_TRY(hipStreamSynchronize(st));
		if(hipGetLastError() != hipSuccess) { cleanup2(); return ctx_fail(CRTHIP_E_DEVICE, "k_enc_tun_parse launch failed"); }
	}
	if(times) {
		float ms = 0;
		uint32_t k = 0;
		if(!chunks.empty() && hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess) { times->name[k] = "enc_hist"; times->ms[k] = ms; times->launches[k] = 1; k++; }
		if(!gpu_ids.empty() && hipEventElapsedTime(&ms, ev[2], ev[3]) == hipSuccess) { times->name[k] = "enc_tun_parse"; times->ms[k] = ms; times->launches[k] = 1; k++; }
		times->count = k;
	}
	cleanup2();

	// block framing (src/cstream.cpp:96-107): u8 nsym | nsym x (symbol, probability) | i32 size | i32 csize | codewords
	uint64_t w = 0;
	auto put = [&](const void *p, size_t len) { if(out && w + len <= cap) memcpy(out + w, p, len); w += len; };
	for(uint32_t i = 0; i < n; i++) {
		block_offset[i] = w;
		const TunEncoderTables &T = tabs[i];
		const uint8_t ns = (uint8_t)T.nsym;
		const uint32_t cs = T.nsym >= 2 ? csize[i] : 0u;
		if(cs > sizes[i] + 1) return ctx_fail(CRTHIP_E_DEVICE, "k_enc_tun_parse produced an impossible codeword count");
		put(&ns, 1);
		put(T.probs, (size_t)T.nsym*2);
		put(&sizes[i], 4);
		put(&cs, 4);
		if(cs) put(h_codes.data() + (dst_off[i] - dst_off[0]), cs);
	}
	block_offset[n] = w;
	if(out && w > cap) return ctx_fail(CRTHIP_E_ARGUMENT, "crthip_tunstall_encode_blocks: output buffer too small");
	return (int64_t)w;
}
