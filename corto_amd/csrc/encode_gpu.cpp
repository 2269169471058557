// encode_gpu.cpp — host side of the GPU encoder stages.
//
//   crthip_tunstall_encode_blocks   OutStream::tunstall_compress (src/cstream.cpp:89-109) for a batch of byte streams
//   crthip_encode_values            OutStream::encodeArray / encodeValues (include/corto/cstream.h:115-164) for a batch of
//                                   integer arrays: bit-width logs + bit packing on the device, the logs then go through
//                                   the Tunstall coder without leaving HBM
//   corto_hip::encode_value_streams the same for encoder.cpp (crthip_encode_gpu)
//
// Tunstall coder: histogram on the device (k_enc_hist)  ->  per stream, on the host, the few microseconds of serial work
// that depend on std::sort's order of equal probabilities: probabilities, 256-word dictionary, encoding trie (encoder.cpp:
// tun_encoder_tables = src/tunstall.cpp:83-115, 125-256, 335-382)  ->  greedy parse on the device (k_enc_tun_parse =
// src/tunstall.cpp:384-428)  ->  block framing on the host.  Everything is byte-identical to the reference's output
// (tests/test_gpu_parity.py::test_tunstall_encode_*, test_encode_values_*, test_gpu_encoder_*).  No CPU fallback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/corto_hip.h"
#include "device_plan.h"
#include "encoder_internal.h"
#include "kernels.h"

using namespace corto_hip;

namespace {

struct DevMem {                                   // freed on every way out
	void *p = nullptr;
	~DevMem() { if(p) (void)hipFree(p); }
	uint8_t *u8() const { return (uint8_t *)p; }
};
struct Events {
	hipEvent_t e[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
	~Events() { for(auto &x : e) if(x) (void)hipEventDestroy(x); }
};
struct StageTimes { float hist = 0, parse = 0, pack = 0, tables = 0, trie = 0; bool any_hist = false, any_parse = false, any_pack = false, any_tables = false; uint32_t host_table_streams = 0; };

#define ENC_TRY(expr) do { hipError_t e_ = (expr); if(e_ != hipSuccess) return ctx_fail(CRTHIP_E_DEVICE, (std::string(#expr ": ") + hipGetErrorString(e_)).c_str()); } while(0)

// Tunstall blocks of n DEVICE-resident byte streams.  blocks[i] = "u8 nsym | nsym x (symbol, probability) | i32 size | i32 csize | codewords"
int tun_encode_device(hipStream_t st, uint32_t n, const uint8_t *const *d_src, const uint32_t *sizes,
                      std::vector<std::vector<uint8_t>> &blocks, StageTimes &tm) {
	blocks.assign(n, std::vector<uint8_t>());
	for(uint32_t i = 0; i < n; i++)
		if(sizes[i] > (1u << 23)) return ctx_fail(CRTHIP_E_LIMIT, "Tunstall encoder: stream longer than 2^23 symbols (the reference's count*255 overflows int)");
	// device image: counts | codewords | csize | histogram chunks
	std::vector<uint64_t> dst_off(n);
	uint64_t o = (uint64_t)n*256*4;
	for(uint32_t i = 0; i < n; i++) { dst_off[i] = o; o += ((uint64_t)sizes[i] + 64 + 15) & ~15ull; }
	const uint64_t o_csize = o; o += ((uint64_t)n*4 + 15) & ~15ull;
	const uint64_t o_chunks = o;
	std::vector<EncChunk> chunks;
	for(uint32_t i = 0; i < n; i++)
		for(uint32_t b = 0; b < sizes[i]; b += ENC_HIST_CHUNK) chunks.push_back(EncChunk{d_src[i] + b, std::min(ENC_HIST_CHUNK, sizes[i] - b), i});
	o += (chunks.size()*sizeof(EncChunk) + 15) & ~15ull;
	DevMem work; Events ev;
	ENC_TRY(hipMalloc(&work.p, o + 16));
	uint8_t *base = work.u8();
	for(auto &e : ev.e) ENC_TRY(hipEventCreate(&e));
	if(n) ENC_TRY(hipMemsetAsync(base, 0, (size_t)n*256*4, st));
	if(n) ENC_TRY(hipMemsetAsync(base + o_csize, 0, (size_t)(o_chunks - o_csize), st));
	std::vector<uint32_t> counts((size_t)n*256);
	if(!chunks.empty()) {
		ENC_TRY(hipMemcpyAsync(base + o_chunks, chunks.data(), chunks.size()*sizeof(EncChunk), hipMemcpyHostToDevice, st));
		ENC_TRY(hipEventRecord(ev.e[0], st));
		hipLaunchKernelGGL(k_enc_hist, dim3((uint32_t)chunks.size()), dim3(256), 0, st, (const EncChunk *)(base + o_chunks), (uint32_t)chunks.size(), (uint32_t *)base);
		ENC_TRY(hipEventRecord(ev.e[1], st));
	}

	// device: probabilities (in std::sort's order), dictionary and - where it fits the LDS of one workgroup - the encoding trie of every
	// stream (k_enc_tables, k_enc_trie).  The host only sizes the trie regions in between and frames the blocks at the end; a stream
	// whose trie could outgrow ENC_TRIE_LDS_MAX entries (large alphabets) gets its tables from the host routine instead.
	std::vector<TunEncoderTables> tabs(n);
	std::vector<uint32_t> csize(n, 0);
	std::vector<uint8_t> h_codes;
	DevMem dtabs, dtrie, dstreams;
	constexpr size_t HEAD = 16 + 512;                                     // EncTab: four words + the probabilities
	std::vector<uint8_t> heads((size_t)n*HEAD);
	std::vector<uint32_t> dev_ids, host_ids;
	float ms_tables = 0, ms_trie = 0;
	if(n) {
		DevMem dsizes;
		ENC_TRY(hipMalloc(&dsizes.p, (size_t)n*4 + 16));
		ENC_TRY(hipMalloc(&dtabs.p, (size_t)n*sizeof(EncTab) + 16));
		ENC_TRY(hipMemcpyAsync(dsizes.p, sizes, (size_t)n*4, hipMemcpyHostToDevice, st));
		ENC_TRY(hipEventRecord(ev.e[2], st));
		hipLaunchKernelGGL(k_enc_tables, dim3(n), dim3(64), 0, st, (const uint32_t *)base, (const uint32_t *)dsizes.p, n, (EncTab *)dtabs.p);
		ENC_TRY(hipEventRecord(ev.e[3], st));
		ENC_TRY(hipMemcpy2DAsync(heads.data(), HEAD, dtabs.p, sizeof(EncTab), HEAD, n, hipMemcpyDeviceToHost, st));
		ENC_TRY(hipStreamSynchronize(st));
		ENC_TRY(hipGetLastError());
		if(hipEventElapsedTime(&ms_tables, ev.e[2], ev.e[3]) != hipSuccess) ms_tables = 0;
	}
	std::vector<uint64_t> trie_off(n, 0);
	uint64_t trie_bytes = 0;
	for(uint32_t i = 0; i < n; i++) {
		const uint32_t *hd = (const uint32_t *)(heads.data() + (size_t)i*HEAD);
		tabs[i].nsym = sizes[i] ? hd[0] : 0u;
		memcpy(tabs[i].probs, heads.data() + (size_t)i*HEAD + 16, 512);
		if(tabs[i].nsym < 2) continue;                                    // one symbol: no payload (tunstall.cpp:386-389)
		const uint64_t entries = (uint64_t)hd[1]*tabs[i].nsym*tabs[i].nsym;
		if(entries <= ENC_TRIE_LDS_MAX) { dev_ids.push_back(i); trie_off[i] = trie_bytes; trie_bytes += (entries*2 + 15) & ~15ull; }
		else host_ids.push_back(i);
	}
	std::vector<uint64_t> tab_off(n, 0);
	uint64_t tbytes = 0;
	if(!host_ids.empty()) {                                               // the few the device leaves to the host: from the histogram, as before
		ENC_TRY(hipMemcpy(counts.data(), base, counts.size()*4, hipMemcpyDeviceToHost));
		for(uint32_t i : host_ids) {
			tun_encoder_tables(&counts[(size_t)i*256], sizes[i], tabs[i]);
			tab_off[i] = tbytes;
			tbytes += 256 + 512 + ((tabs[i].offsets.size()*2 + 15) & ~15ull);
		}
	}
	const uint32_t ngpu = (uint32_t)(dev_ids.size() + host_ids.size());
	uint32_t trie_lds = 0;
	if(ngpu) {
		const uint64_t o_streams = tbytes, o_ids = o_streams + (((uint64_t)ngpu*sizeof(EncStream) + 15) & ~15ull);
		const uint64_t tab_total = o_ids + (uint64_t)dev_ids.size()*4 + 16;
		std::vector<uint8_t> h_tab(tab_total);
		ENC_TRY(hipMalloc(&dstreams.p, tab_total));
		ENC_TRY(hipMalloc(&dtrie.p, trie_bytes + 16));
		uint8_t *tb = dstreams.u8();
		EncTab *dt = (EncTab *)dtabs.p;
		std::vector<EncStream> es;
		for(uint32_t i : dev_ids) {                                       // device-made tables first: k_enc_trie indexes this array by position
			EncStream s{};
			s.src = d_src[i]; s.dst = base + dst_off[i];
			s.remap = dt[i].remap; s.lengths = dt[i].lengths; s.trie = (int16_t *)(dtrie.u8() + trie_off[i]);
			s.csize = (uint32_t *)(base + o_csize) + i;
			s.size = sizes[i]; s.nsym = tabs[i].nsym; s.ntrie = 0;
			es.push_back(s);
		}
		for(uint32_t i : host_ids) {
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
			s.src = d_src[i]; s.dst = base + dst_off[i];
			s.remap = tb + tab_off[i]; s.lengths = (const uint16_t *)(tb + tab_off[i] + 256); s.trie = (int16_t *)(tb + tab_off[i] + 768);
			s.csize = (uint32_t *)(base + o_csize) + i;
			s.size = sizes[i]; s.nsym = T.nsym; s.ntrie = (uint32_t)T.offsets.size();
			if(T.offsets.size() <= ENC_TRIE_LDS_MAX) trie_lds = std::max<uint32_t>(trie_lds, (uint32_t)T.offsets.size());
			es.push_back(s);
		}
		memcpy(h_tab.data() + o_streams, es.data(), es.size()*sizeof(EncStream));
		if(!dev_ids.empty()) memcpy(h_tab.data() + o_ids, dev_ids.data(), dev_ids.size()*4);
		ENC_TRY(hipMemcpyAsync(dstreams.p, h_tab.data(), tab_total, hipMemcpyHostToDevice, st));
		if(!dev_ids.empty()) {
			const uint32_t nd = (uint32_t)dev_ids.size();
			ENC_TRY(hipEventRecord(ev.e[4], st));
			hipLaunchKernelGGL(k_enc_trie, dim3(nd), dim3(64), ENC_TRIE_LDS_MAX*2, st, (const EncTab *)dtabs.p, (const uint32_t *)(tb + o_ids), (EncStream *)(tb + o_streams), nd, ENC_TRIE_LDS_MAX);
			ENC_TRY(hipEventRecord(ev.e[5], st));
			ENC_TRY(hipMemcpyAsync(es.data(), tb + o_streams, (size_t)nd*sizeof(EncStream), hipMemcpyDeviceToHost, st));
			ENC_TRY(hipStreamSynchronize(st));
			ENC_TRY(hipGetLastError());
			if(hipEventElapsedTime(&ms_trie, ev.e[4], ev.e[5]) != hipSuccess) ms_trie = 0;
			for(uint32_t k = 0; k < nd; k++) {
				if(es[k].ntrie == 0xFFFFFFFFu || es[k].ntrie == 0) return ctx_fail(CRTHIP_E_DEVICE, "k_enc_trie: a trie outgrew the bound the dictionary gives");
				trie_lds = std::max(trie_lds, es[k].ntrie);
			}
		}
		const uint32_t lds = enc_parse_lds(trie_lds);
		hipEvent_t p0 = nullptr, p1 = nullptr;
		ENC_TRY(hipEventCreate(&p0)); ENC_TRY(hipEventCreate(&p1));
		ENC_TRY(hipEventRecord(p0, st));
		hipLaunchKernelGGL(k_enc_tun_parse, dim3(ngpu), dim3(64), lds, st, (const EncStream *)(tb + o_streams), ngpu, trie_lds);
		ENC_TRY(hipEventRecord(p1, st));
		ENC_TRY(hipMemcpyAsync(csize.data(), base + o_csize, (size_t)n*4, hipMemcpyDeviceToHost, st));
		h_codes.resize(o_csize - dst_off[0]);
		ENC_TRY(hipMemcpyAsync(h_codes.data(), base + dst_off[0], h_codes.size(), hipMemcpyDeviceToHost, st));
		ENC_TRY(hipStreamSynchronize(st));
		ENC_TRY(hipGetLastError());
		float pm = 0;
		if(hipEventElapsedTime(&pm, p0, p1) == hipSuccess) { tm.parse += pm; tm.any_parse = true; }
		(void)hipEventDestroy(p0); (void)hipEventDestroy(p1);
	}
	if(n) { tm.tables += ms_tables; tm.trie += ms_trie; tm.any_tables = true; tm.host_table_streams += (uint32_t)host_ids.size(); }
	float ms = 0;
	if(!chunks.empty() && hipEventElapsedTime(&ms, ev.e[0], ev.e[1]) == hipSuccess) { tm.hist += ms; tm.any_hist = true; }

	// block framing (src/cstream.cpp:96-107)
	for(uint32_t i = 0; i < n; i++) {
		const TunEncoderTables &T = tabs[i];
		const uint32_t cs = T.nsym >= 2 ? csize[i] : 0u;
		if(cs > sizes[i] + 1) return ctx_fail(CRTHIP_E_DEVICE, "k_enc_tun_parse produced an impossible codeword count");
		std::vector<uint8_t> &b = blocks[i];
		b.resize(9 + (size_t)T.nsym*2 + cs);
		b[0] = (uint8_t)T.nsym;
		memcpy(&b[1], T.probs, (size_t)T.nsym*2);
		memcpy(&b[1 + T.nsym*2], &sizes[i], 4);
		memcpy(&b[5 + T.nsym*2], &cs, 4);
		if(cs) memcpy(&b[9 + T.nsym*2], h_codes.data() + (dst_off[i] - dst_off[0]), cs);
	}
	return CRTHIP_OK;
}

void report(crthip_kernel_times *times, const StageTimes &tm) {
	if(!times) return;
	uint32_t k = 0;
	if(tm.any_pack) { times->name[k] = "enc_pack"; times->ms[k] = tm.pack; times->launches[k] = 1; k++; }
	if(tm.any_hist) { times->name[k] = "enc_hist"; times->ms[k] = tm.hist; times->launches[k] = 1; k++; }
	if(tm.any_tables) { times->name[k] = "enc_tables"; times->ms[k] = tm.tables; times->launches[k] = 1; k++;
	                    times->name[k] = "enc_trie"; times->ms[k] = tm.trie; times->launches[k] = tm.host_table_streams; k++; }   // (launches of enc_trie: streams whose tables the HOST made instead)
	if(tm.any_parse) { times->name[k] = "enc_tun_parse"; times->ms[k] = tm.parse; times->launches[k] = 1; k++; }
	times->count = k;
}

} // namespace

extern "C" int64_t crthip_tunstall_encode_blocks(crthip_ctx *ctx, uint32_t n, const uint8_t *const *src, const uint32_t *sizes,
                                                 uint8_t *out, size_t cap, uint64_t *block_offset, crthip_kernel_times *times) {
	if(!ctx || (n && (!src || !sizes)) || !block_offset) return ctx_fail(CRTHIP_E_ARGUMENT, "crthip_tunstall_encode_blocks: null argument");
	for(uint32_t i = 0; i < n; i++) if(sizes[i] && !src[i]) return ctx_fail(CRTHIP_E_ARGUMENT, "crthip_tunstall_encode_blocks: null stream");
	if(times) memset(times, 0, sizeof(*times));
	ENC_TRY(hipSetDevice(ctx_device(ctx)));
	{ const int e = ctx_quiesce(ctx); if(e) return e; }
	hipStream_t st = ctx_stream(ctx);
	// sources go up in one copy
	std::vector<uint64_t> src_off(n);
	uint64_t o = 0;
	for(uint32_t i = 0; i < n; i++) { src_off[i] = o; o += ((uint64_t)sizes[i] + 15) & ~15ull; }
	std::vector<uint8_t> h_src(o + 16);
	for(uint32_t i = 0; i < n; i++) if(sizes[i]) memcpy(h_src.data() + src_off[i], src[i], sizes[i]);
	DevMem dsrc;
	ENC_TRY(hipMalloc(&dsrc.p, o + 16));
	if(o) ENC_TRY(hipMemcpy(dsrc.p, h_src.data(), o, hipMemcpyHostToDevice));
	std::vector<const uint8_t *> d_src(n);
	for(uint32_t i = 0; i < n; i++) d_src[i] = dsrc.u8() + src_off[i];
	std::vector<std::vector<uint8_t>> blocks;
	StageTimes tm;
	{ const int e = tun_encode_device(st, n, d_src.data(), sizes, blocks, tm); if(e) return e; }
	report(times, tm);
	uint64_t w = 0;
	for(uint32_t i = 0; i < n; i++) {
		block_offset[i] = w;
		if(out && w + blocks[i].size() <= cap) memcpy(out + w, blocks[i].data(), blocks[i].size());
		w += blocks[i].size();
	}
	block_offset[n] = w;
	if(out && w > cap) return ctx_fail(CRTHIP_E_ARGUMENT, "crthip_tunstall_encode_blocks: output buffer too small");
	return (int64_t)w;
}

// the quantisation step of every attribute of one mesh on the device: one upload, one kernel per attribute, one download
int corto_hip::quantize_device(crthip_ctx *ctx, const std::vector<QuantRequest> &reqs) {
	if(!ctx) return ctx_fail(CRTHIP_E_ARGUMENT, "quantize_device: null context");
	ENC_TRY(hipSetDevice(ctx_device(ctx)));
	{ const int e = ctx_quiesce(ctx); if(e) return e; }
	hipStream_t st = ctx_stream(ctx);
	auto in_bytes = [](const QuantRequest &r) -> uint64_t { return r.kind == 0 ? (uint64_t)r.count*4 : r.kind == 1 ? (uint64_t)r.count*12 : (uint64_t)r.count*r.N; };
	auto out_bytes = [](const QuantRequest &r) -> uint64_t { return r.kind == 0 ? (uint64_t)r.count*4 : r.kind == 1 ? (uint64_t)r.count*8 : (uint64_t)r.count*r.N; };
	std::vector<uint64_t> ioff(reqs.size()), ooff(reqs.size());
	uint64_t o = 0;
	for(size_t k = 0; k < reqs.size(); k++) { ioff[k] = o; o += (in_bytes(reqs[k]) + 15) & ~15ull; }
	const uint64_t in_total = o;
	for(size_t k = 0; k < reqs.size(); k++) { ooff[k] = o; o += (out_bytes(reqs[k]) + 15) & ~15ull; }
	if(o == 0) return CRTHIP_OK;
	DevMem dev;
	ENC_TRY(hipMalloc(&dev.p, o + 16));
	std::vector<uint8_t> h(o);
	for(size_t k = 0; k < reqs.size(); k++) if(in_bytes(reqs[k])) memcpy(h.data() + ioff[k], reqs[k].in, in_bytes(reqs[k]));
	ENC_TRY(hipMemcpyAsync(dev.p, h.data(), in_total, hipMemcpyHostToDevice, st));
	for(size_t k = 0; k < reqs.size(); k++) {
		const QuantRequest &r = reqs[k];
		if(!r.count) continue;
		QuantJob J{};
		J.in = dev.u8() + ioff[k]; J.out = dev.u8() + ooff[k]; J.count = r.count; J.kind = r.kind; J.N = r.N; J.q = r.q; J.unit = r.unit;
		for(int c = 0; c < 4; c++) J.qc[c] = r.qc[c] ? r.qc[c] : 1u;
		hipLaunchKernelGGL(k_enc_quantize, dim3((r.count + 255)/256), dim3(256), 0, st, J);
	}
	ENC_TRY(hipMemcpyAsync(h.data() + in_total, dev.u8() + in_total, o - in_total, hipMemcpyDeviceToHost, st));
	ENC_TRY(hipStreamSynchronize(st));
	ENC_TRY(hipGetLastError());
	for(size_t k = 0; k < reqs.size(); k++) if(out_bytes(reqs[k])) memcpy(reqs[k].out, h.data() + ooff[k], out_bytes(reqs[k]));
	return CRTHIP_OK;
}

// bit-width logs + bit packing of n value arrays on the device, then the Tunstall coder over the logs (or raw logs for entropy NONE)
int corto_hip::encode_value_streams(crthip_ctx *ctx, uint32_t entropy, const std::vector<EncValueStream> &in, std::vector<EncValueResult> &res,
                                    crthip_kernel_times *times) {
	const uint32_t n = (uint32_t)in.size();
	res.assign(n, EncValueResult());
	if(!ctx) return ctx_fail(CRTHIP_E_ARGUMENT, "encode_value_streams: null context");
	if(entropy != CRTHIP_ENTROPY_NONE && entropy != CRTHIP_ENTROPY_TUNSTALL) return ctx_fail(CRTHIP_E_ENTROPY, nullptr);
	if(times) memset(times, 0, sizeof(*times));
	ENC_TRY(hipSetDevice(ctx_device(ctx)));
	{ const int e = ctx_quiesce(ctx); if(e) return e; }
	hipStream_t st = ctx_stream(ctx);

	// device image: values (or symbols) | logs | words | word counts | jobs
	std::vector<uint64_t> v_off(n), l_off(n), w_off(n);
	std::vector<uint32_t> nlogs(n);                      // log arrays of stream i (0 for a symbol stream)
	auto value_bytes = [&](const EncValueStream &s) -> uint64_t {
		return s.kind == CRTHIP_ENC_SYMBOLS ? s.count : (uint64_t)s.count*s.components*(s.kind == CRTHIP_ENC_VALUES_I8 ? 1 : 4);
	};
	uint64_t o = 0;
	for(uint32_t i = 0; i < n; i++) {
		const EncValueStream &s = in[i];
		if(s.count && !s.values) return ctx_fail(CRTHIP_E_ARGUMENT, "encode_value_streams: null values");
		if(s.kind > CRTHIP_ENC_VALUES_I8) return ctx_fail(CRTHIP_E_ARGUMENT, "encode_value_streams: unknown stream kind");
		if(s.kind != CRTHIP_ENC_SYMBOLS && (s.components == 0 || s.components > ENC_PACK_MAX_N)) return ctx_fail(CRTHIP_E_LIMIT, "encode_value_streams: components out of range");
		if((uint64_t)s.count*std::max(1u, s.components) > (1u << 26)) return ctx_fail(CRTHIP_E_LIMIT, "encode_value_streams: array too long");
		v_off[i] = o; o += (value_bytes(s) + 15) & ~15ull;
		nlogs[i] = s.kind == CRTHIP_ENC_SYMBOLS ? 0u : s.kind == CRTHIP_ENC_ARRAY ? 1u : s.components;
	}
	const uint64_t values_bytes = o;
	for(uint32_t i = 0; i < n; i++) { l_off[i] = o; o += ((uint64_t)in[i].count*nlogs[i] + 15) & ~15ull; }
	for(uint32_t i = 0; i < n; i++) { w_off[i] = o; o += nlogs[i] ? ((uint64_t)in[i].count*in[i].components*4 + 8 + 15) & ~15ull : 0; }
	const uint64_t o_nwords = o; o += ((uint64_t)n*4 + 15) & ~15ull;
	const uint64_t o_jobs = o;
	std::vector<PackJob> jobs;
	std::vector<uint32_t> job_stream;
	DevMem dev; Events ev;
	for(auto &e : ev.e) ENC_TRY(hipEventCreate(&e));
	for(uint32_t i = 0; i < n; i++) if(nlogs[i] && in[i].count) { jobs.push_back(PackJob{}); job_stream.push_back(i); }
	o += (jobs.size()*sizeof(PackJob) + 15) & ~15ull;
	ENC_TRY(hipMalloc(&dev.p, o + 16));
	uint8_t *base = dev.u8();
	{
		std::vector<uint8_t> h(values_bytes + 16);
		for(uint32_t i = 0; i < n; i++) if(value_bytes(in[i])) memcpy(h.data() + v_off[i], in[i].values, value_bytes(in[i]));
		if(values_bytes) ENC_TRY(hipMemcpy(base, h.data(), values_bytes, hipMemcpyHostToDevice));
	}
	ENC_TRY(hipMemsetAsync(base + o_nwords, 0, (size_t)(o_jobs - o_nwords), st));
	StageTimes tm;
	std::vector<uint32_t> nwords(n, 0);
	if(!jobs.empty()) {
		for(size_t j = 0; j < jobs.size(); j++) {
			const uint32_t i = job_stream[j];
			PackJob &p = jobs[j];
			p.values = base + v_off[i]; p.logs = base + l_off[i]; p.words = (uint32_t *)(base + w_off[i]); p.nwords = (uint32_t *)(base + o_nwords) + i;
			p.count = in[i].count; p.N = in[i].components; p.kind = in[i].kind;
		}
		ENC_TRY(hipMemcpyAsync(base + o_jobs, jobs.data(), jobs.size()*sizeof(PackJob), hipMemcpyHostToDevice, st));
		ENC_TRY(hipEventRecord(ev.e[4], st));
		hipLaunchKernelGGL(k_enc_pack, dim3((uint32_t)jobs.size()), dim3(256), 0, st, (const PackJob *)(base + o_jobs), (uint32_t)jobs.size());
		ENC_TRY(hipEventRecord(ev.e[5], st));
		ENC_TRY(hipMemcpyAsync(nwords.data(), base + o_nwords, (size_t)n*4, hipMemcpyDeviceToHost, st));
		ENC_TRY(hipStreamSynchronize(st));
		ENC_TRY(hipGetLastError());
		float ms = 0;
		if(hipEventElapsedTime(&ms, ev.e[4], ev.e[5]) == hipSuccess) { tm.pack = ms; tm.any_pack = true; }
	}
	// bit words back to the host
	for(uint32_t i = 0; i < n; i++) {
		if(!nlogs[i]) continue;
		if((uint64_t)nwords[i] > (uint64_t)in[i].count*in[i].components + 1) return ctx_fail(CRTHIP_E_DEVICE, "k_enc_pack produced an impossible word count");
		res[i].words.resize(nwords[i]);
		if(nwords[i]) ENC_TRY(hipMemcpyAsync(res[i].words.data(), base + w_off[i], (size_t)nwords[i]*4, hipMemcpyDeviceToHost, st));
	}
	// entropy coder over the log arrays (device resident) and the symbol streams
	std::vector<const uint8_t *> d_src; std::vector<uint32_t> sizes; std::vector<std::pair<uint32_t, uint32_t>> owner;
	for(uint32_t i = 0; i < n; i++) {
		if(!nlogs[i]) { d_src.push_back(base + v_off[i]); sizes.push_back(in[i].count); owner.push_back({i, 0}); res[i].blocks.resize(1); }
		else {
			res[i].blocks.resize(nlogs[i]);
			for(uint32_t c = 0; c < nlogs[i]; c++) { d_src.push_back(base + l_off[i] + (uint64_t)c*in[i].count); sizes.push_back(in[i].count); owner.push_back({i, c}); }
		}
	}
	if(entropy == CRTHIP_ENTROPY_TUNSTALL) {
		std::vector<std::vector<uint8_t>> blocks;
		{ const int e = tun_encode_device(st, (uint32_t)d_src.size(), d_src.data(), sizes.data(), blocks, tm); if(e) return e; }
		for(size_t k = 0; k < blocks.size(); k++) res[owner[k].first].blocks[owner[k].second] = std::move(blocks[k]);
	} else {                                                       // OutStream::compress with entropy NONE: i32 size | bytes (cstream.cpp:43-64)
		for(size_t k = 0; k < d_src.size(); k++) {
			std::vector<uint8_t> &b = res[owner[k].first].blocks[owner[k].second];
			b.resize(4 + (size_t)sizes[k]);
			memcpy(b.data(), &sizes[k], 4);
			if(sizes[k]) ENC_TRY(hipMemcpyAsync(b.data() + 4, d_src[k], sizes[k], hipMemcpyDeviceToHost, st));
		}
	}
	ENC_TRY(hipStreamSynchronize(st));
	report(times, tm);
	return CRTHIP_OK;
}

extern "C" int64_t crthip_encode_values(crthip_ctx *ctx, uint32_t entropy, uint32_t n, const crthip_enc_stream *streams,
                                        uint8_t *out, size_t cap, uint64_t *stream_offset, crthip_kernel_times *times) {
	if(!ctx || (n && !streams) || !stream_offset) return ctx_fail(CRTHIP_E_ARGUMENT, "crthip_encode_values: null argument");
	std::vector<EncValueStream> in(n);
	for(uint32_t i = 0; i < n; i++) { in[i].kind = streams[i].kind; in[i].count = streams[i].count; in[i].components = streams[i].components; in[i].values = streams[i].values; }
	std::vector<EncValueResult> res;
	{ const int e = encode_value_streams(ctx, entropy, in, res, times); if(e) return e; }
	uint64_t w = 0;
	auto put = [&](const void *p, size_t len) { if(out && w + len <= cap) memcpy(out + w, p, len); w += len; };
	for(uint32_t i = 0; i < n; i++) {
		stream_offset[i] = w;
		if(in[i].kind != CRTHIP_ENC_SYMBOLS) {
			const uint32_t nw = (uint32_t)res[i].words.size();
			put(&nw, 4);
			put(res[i].words.data(), (size_t)nw*4);
		}
		for(auto &b : res[i].blocks) put(b.data(), b.size());
	}
	stream_offset[n] = w;
	if(out && w > cap) return ctx_fail(CRTHIP_E_ARGUMENT, "crthip_encode_values: output buffer too small");
	return (int64_t)w;
}
