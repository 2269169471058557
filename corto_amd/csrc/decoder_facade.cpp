// decoder_facade.cpp — crt::Decoder (include/corto/decoder.h of this repo) on top of the C ABI only.
// Replaces upstream src/decoder.cpp:41-131 (constructor, set*, decode dispatch); the stages themselves run
// on the device behind crthip_decode_host.
#include "../../include/corto/decoder.h"

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/corto_hip.h"

namespace crt {
namespace {

// upstream throws string literals; the C ABI hands back the same literals
[[noreturn]] void raise(int code) {
	switch(code) {
	case CRTHIP_E_ALIGN: throw "Memory must be alignegned on 4 bytes.";
	case CRTHIP_E_MAGIC: throw "Not a crt file.";
	case CRTHIP_E_ENTROPY: throw "Unknown entropy";
	case CRTHIP_E_TOPOLOGY: throw "Decoding topology failed";
	case CRTHIP_E_NORMAL_NEEDS_POSITION: throw "No position attribute found. Use DIFF normal strategy instead.";
	case CRTHIP_E_FORMAT: throw "Format not supported for this attribute on the device path";
	case CRTHIP_E_DEVICE: throw "No usable HIP device (the MI355X path has no CPU fallback)";
	case CRTHIP_E_TRUNCATED: throw "Truncated or inconsistent crt stream.";
	default: throw "corto_hip: decode failed";
	}
}

// Upstream's Decoder objects share nothing (the library's only global is the read-only bmask[], src/bitstream.cpp:29-33), so
// distinct objects may decode on different threads at once.  Here they share the GPU: a small pool of contexts (each with its
// own HIP streams, scratch block, batch object and pinned staging - include/corto_hip.h: crthip_decode_host), one borrowed
// for the duration of a decode().  Up to POOL_MAX decodes overlap on the device; more callers wait for a free context.
// setDevice() retires the pool: idle contexts at once, borrowed ones when they come back.
struct Pool {
	std::mutex m;
	std::condition_variable cv;
	std::vector<crthip_ctx *> idle;
	int live = 0, device = -1, limit = 0;
	uint64_t generation = 0;
} g_pool;

int pool_limit() {
	const char *e = getenv("CORTO_HIP_CONTEXTS");
	int n = e ? atoi(e) : 8;
	return n < 1 ? 1 : n > 64 ? 64 : n;
}

struct Lease {
	crthip_ctx *ctx = nullptr;
	uint64_t generation = 0;
	Lease() {
		std::unique_lock<std::mutex> lock(g_pool.m);
		if(!g_pool.limit) g_pool.limit = pool_limit();
		for(;;) {
			if(!g_pool.idle.empty()) { ctx = g_pool.idle.back(); g_pool.idle.pop_back(); break; }
			if(g_pool.live < g_pool.limit) {
				int dev = g_pool.device;
				if(dev < 0) { const char *e = getenv("CORTO_HIP_DEVICE"); dev = e ? atoi(e) : 0; }
				g_pool.live++;                                   // reserve the slot, create outside the lock
				const uint64_t gen = g_pool.generation;
				lock.unlock();
				crthip_ctx *c = nullptr;
				const int err = crthip_ctx_create(dev, &c);
				lock.lock();
				if(err) { g_pool.live--; g_pool.cv.notify_one(); lock.unlock(); raise(err); }
				if(gen != g_pool.generation) { g_pool.live--; lock.unlock(); crthip_ctx_destroy(c); lock.lock(); continue; }   // setDevice() meanwhile
				ctx = c;
				break;
			}
			g_pool.cv.wait(lock);
		}
		generation = g_pool.generation;
	}
	~Lease() {
		if(!ctx) return;
		std::unique_lock<std::mutex> lock(g_pool.m);
		if(generation == g_pool.generation) { g_pool.idle.push_back(ctx); lock.unlock(); }
		else { g_pool.live--; lock.unlock(); crthip_ctx_destroy(ctx); }
		g_pool.cv.notify_one();
	}
	Lease(const Lease &) = delete;
	Lease &operator=(const Lease &) = delete;
};

} // namespace

void Decoder::setDevice(int device) {
	std::vector<crthip_ctx *> retire;
	{
		std::lock_guard<std::mutex> lock(g_pool.m);
		if(device == g_pool.device) return;
		g_pool.device = device;
		g_pool.generation++;                                  // contexts out on loan are destroyed when they return
		retire.swap(g_pool.idle);
		g_pool.live -= (int)retire.size();
	}
	for(crthip_ctx *c : retire) crthip_ctx_destroy(c);
	g_pool.cv.notify_all();
}

Decoder::Decoder(int len, const uchar *input): nvert(0), nface(0), input_(input), len_(len) {
	crthip_blob_info info;
	int err = crthip_probe(input, (size_t)len, &info);
	if(err) raise(err);
	nvert = info.nvert; nface = info.nface;
	for(uint32_t i = 0; i < info.nattr; i++) {
		VertexAttribute *a = new VertexAttribute();
		a->N = (int)info.attr[i].components; a->q = info.attr[i].q; a->strategy = (int)info.attr[i].strategy;
		a->format = (VertexAttribute::Format)info.attr[i].format; a->codec_id = (int)info.attr[i].codec;
		if(a->codec_id == VertexAttribute::NORMAL_CODEC) a->N = 3;       // NormalAttr(): N = 3
		data[info.attr[i].name] = a;
	}
	int64_t n = crthip_probe_exif(input, (size_t)len, nullptr, 0);
	if(n > 0) {
		std::vector<char> buf((size_t)n);
		crthip_probe_exif(input, (size_t)len, buf.data(), buf.size());
		for(size_t p = 0; p < buf.size();) {
			std::string k(&buf[p]); p += k.size() + 1;
			std::string v(&buf[p]); p += v.size() + 1;
			exif[k] = v;
		}
	}
}

Decoder::~Decoder() {
	for(auto &it : data) delete it.second;
}

bool Decoder::setAttribute(const char *name, char *buffer, VertexAttribute::Format format) {
	auto it = data.find(name);
	if(it == data.end()) return false;
	it->second->format = format;
	it->second->buffer = buffer;
	return true;
}

bool Decoder::setAttribute(const char *name, char *, VertexAttribute *) {
	if(data.find(name) == data.end()) return false;
	throw "Custom attribute codecs cannot run on the device path";
}

bool Decoder::setColors(uchar *buffer, int components) {
	auto it = data.find("color");
	if(it == data.end()) return false;
	it->second->format = VertexAttribute::UINT8;
	it->second->buffer = (char *)buffer;
	it->second->out_components = components;
	return true;
}

void Decoder::decode() {
	// groups are part of what upstream's decode() leaves behind in index.groups (index_attribute.h:89-99)
	int64_t ng = crthip_probe_groups(input_, (size_t)len_, nullptr, 0);
	if(ng < 0) raise((int)ng);
	std::vector<uint32_t> ends((size_t)ng);
	if(ng) crthip_probe_groups(input_, (size_t)len_, ends.data(), ends.size());
	index.groups.resize((size_t)ng);
	for(int64_t g = 0; g < ng; g++) {
		index.groups[(size_t)g].end = ends[(size_t)g];
		int64_t pn = crthip_probe_group_props(input_, (size_t)len_, (uint32_t)g, nullptr, 0);
		if(pn > 0) {
			std::vector<char> buf((size_t)pn);
			crthip_probe_group_props(input_, (size_t)len_, (uint32_t)g, buf.data(), buf.size());
			for(size_t p = 0; p < buf.size();) {
				std::string k(&buf[p]); p += k.size() + 1;
				std::string v(&buf[p]); p += v.size() + 1;
				index.groups[(size_t)g].properties[k] = v;
			}
		}
	}
	std::vector<crthip_attr_binding> binds;
	for(auto &it : data) {                                   // std::map order == the C ABI's attribute order (sorted by name)
		crthip_attr_binding b;
		b.buffer = it.second->buffer;
		b.format = (uint32_t)it.second->format;
		b.out_components = (uint32_t)it.second->out_components;
		b.stride = 0; b.reserved = 0;
		binds.push_back(b);
	}
	void *idx = index.faces16 ? (void *)index.faces16 : (void *)index.faces32;   // faces16 wins (src/decoder.cpp:246-249)
	uint32_t ifmt = index.faces16 ? CRTHIP_FMT_UINT16 : CRTHIP_FMT_UINT32;
	int err;
	{
		Lease lease;                                          // a context of the pool for the duration of this decode
		err = crthip_decode_host(lease.ctx, input_, (size_t)len_, binds.data(), idx, ifmt);
	}
	if(err) raise(err);
}

} // namespace crt
