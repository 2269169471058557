// decoder_facade.cpp — crt::Decoder (include/corto/decoder.h of this repo) on top of the C ABI only.
// Replaces upstream src/decoder.cpp:41-131 (constructor, set*, decode dispatch); the stages themselves run
// on the device behind crthip_decode_host.
#include "../../include/corto/decoder.h"

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/corto_hip.h"
#include "encoder_internal.h"

#include <atomic>
#include <chrono>

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

void read_settings();

struct Lease {
	crthip_ctx *ctx = nullptr;
	uint64_t generation = 0;
	Lease() {
		read_settings();
		std::unique_lock<std::mutex> lock(g_pool.m);
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

// The decode combiner.  A 4K-triangle blob alone is one serial chain on the GPU (its CLERS automaton takes as long as a whole batch's),
// so N threads that each decode their own blob on their own context pay N launch sequences for N chains that one batch would run side by
// side.  Concurrent decode() calls are therefore coalesced: a caller queues its request; whoever finds no leader at work becomes one,
// waits a few microseconds when other callers are about (they are usually a step behind), takes what has queued up - its own request
// included, up to COMBINE_MAX - and decodes it as ONE batch on a pool context (corto_hip::decode_host_many: one upload, one set of
// launches, one download); every caller then copies its own outputs out of the pinned landing zone, the leader waits for them and
// gives the context back.  A lone caller is its own leader and takes the one-blob path at its cost.  $CORTO_HIP_COMBINE_US: the gather
// window (default 30; 0: never wait; negative: no combining, a context per caller); $CORTO_HIP_LEADERS: batches in flight (default 2).
constexpr uint32_t COMBINE_MAX = 64;
struct Request {
	corto_hip::HostDecodeReq h{};
	std::vector<crthip_attr_binding> binds;
	bool taken = false, ready = false;      // picked up by a leader; decoded (outputs waiting in the landing zone, or status != OK)
	std::atomic<int> *pending = nullptr;    // the leader's count of followers still copying
	std::string message;                    // why it failed (the leader's thread-local last error)
};
struct Combiner {
	std::mutex m;
	std::condition_variable cv;
	std::vector<Request *> q;
	int leaders = 0, callers = 0;           // leaders at work; threads inside decode()
	int window_us = -1, max_leaders = 0;
} g_comb;

void copy_out(const Request &r) {
	for(uint32_t k = 0; k < r.h.nout; k++) memcpy(r.h.out_dst[k], r.h.out_src[k], r.h.out_bytes[k]);
}

// the pool's size and the combiner's settings: read once, before any thread can race on them (g_pool.limit is read under g_pool.m by
// Lease and under g_comb.m by the combiner: it must not be written under either alone)
std::once_flag g_settings_once;
void read_settings() {
	std::call_once(g_settings_once, [] {
		g_pool.limit = pool_limit();
		const char *e = getenv("CORTO_HIP_COMBINE_US");
		g_comb.window_us = e ? atoi(e) : 30;
		if(g_comb.window_us < 0) g_comb.window_us = -2;                    // (< 0: every caller decodes its own blob)
		// TWO batches in flight (one being planned and copied out while the other's kernels run), however many threads call: what makes the
		// batches big is that callers queue up behind busy leaders - with a leader per caller (8 contexts) sixteen threads got 97 us a blob,
		// with two 38 (tests/cpp/facade_threads.cpp).  Combining off: as many as the pool has contexts.
		e = getenv("CORTO_HIP_LEADERS");
		const int n = e ? atoi(e) : 2;
		g_comb.max_leaders = g_comb.window_us < 0 ? g_pool.limit : n < 1 ? 1 : n > g_pool.limit ? g_pool.limit : n;
	});
}

int combined_decode(Request &r) {
	read_settings();
	std::unique_lock<std::mutex> lock(g_comb.m);
	g_comb.callers++;
	g_comb.q.push_back(&r);
	for(;;) {
		if(r.ready) break;
		if(!r.taken && g_comb.leaders < g_comb.max_leaders) {
			// lead: let the callers that are a step behind queue up, then take what is there
			g_comb.leaders++;
			if(g_comb.callers > 2 && g_comb.window_us > 0 && g_comb.q.size() < (size_t)g_comb.callers) {      // (two callers: a context each is as good)
				lock.unlock();
				const auto t0 = std::chrono::steady_clock::now();
				while(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < g_comb.window_us) {
#if defined(__x86_64__)
					__builtin_ia32_pause();
#endif
				}
				lock.lock();
			}
			if(r.taken) { g_comb.leaders--; g_comb.cv.notify_all(); continue; }   // another leader took mine while I gathered: follow it
			// MY request first, wherever it stands in the queue (with 65+ callers queued behind two busy leaders the thread that wakes as the
			// next leader may sit beyond the first COMBINE_MAX entries: it must never return without its own blob decoded), then the oldest others
			std::vector<Request *> mine;
			for(auto it = g_comb.q.begin(); it != g_comb.q.end(); ++it) if(*it == &r) { g_comb.q.erase(it); break; }
			r.taken = true; mine.push_back(&r);
			if(g_comb.window_us >= 0)                                    // (combining off: mine alone)
				for(auto it = g_comb.q.begin(); it != g_comb.q.end() && mine.size() < COMBINE_MAX;) { (*it)->taken = true; mine.push_back(*it); it = g_comb.q.erase(it); }
			lock.unlock();
			std::atomic<int> pending((int)mine.size() - 1);           // the followers (everyone in `mine` but me) still copying out of the landing zone
			int batch_err = 0; std::string batch_msg;
			bool published = false;
			std::unique_ptr<Lease> lease;                             // a context of the pool for the duration of this batch: given back only when
			try {                                                     // every follower has copied its outputs out of its landing zone
				std::vector<corto_hip::HostDecodeReq> reqs(mine.size());
				for(size_t k = 0; k < mine.size(); k++) reqs[k] = mine[k]->h;
				lease.reset(new Lease());
				const int rc = corto_hip::decode_host_many(lease->ctx, (uint32_t)reqs.size(), reqs.data(), false);
				const std::string msg = crthip_last_error();
				// a call that failed as a whole (device, memory) must not leave any request looking decoded: every request that still says OK
				// without having been through the kernels takes the call's code
				if(rc == CRTHIP_E_DEVICE || rc == CRTHIP_E_NOMEM) for(auto &q : reqs) if(q.status == CRTHIP_OK) { q.status = rc; q.nout = 0; }
				lock.lock();
				for(size_t k = 0; k < mine.size(); k++) { mine[k]->h = reqs[k]; mine[k]->pending = &pending; if(reqs[k].status) mine[k]->message = msg; mine[k]->ready = true; }
				published = true;
				g_comb.cv.notify_all();
				lock.unlock();
				if(r.h.status == CRTHIP_OK) copy_out(r);              // mine, while the followers copy theirs
				while(pending.load(std::memory_order_acquire) > 0) std::this_thread::yield();   // the landing zone is the context's: keep it until they are done
			} catch(const char *m) { batch_err = CRTHIP_E_DEVICE; batch_msg = m ? m : ""; }
			catch(const std::bad_alloc &) { batch_err = CRTHIP_E_NOMEM; batch_msg = crthip_strerror(CRTHIP_E_NOMEM); }
			catch(...) { batch_err = CRTHIP_E_DEVICE; batch_msg = "corto_hip: decode failed"; }
			if(!lock.owns_lock()) lock.lock();
			if(batch_err && !published) {
				// nobody has been told anything yet: every request of this batch fails with the batch's error, and nobody copies
				for(Request *q : mine) { q->h.status = batch_err; q->h.nout = 0; q->pending = nullptr; q->message = batch_msg.empty() ? crthip_strerror(batch_err) : batch_msg; q->ready = true; }
			} else if(batch_err) {
				// (an exception behind the publication can only come from my own copy_out / the wait: the followers have their results;
				// wait for their copies before the landing zone's context goes back)
				lock.unlock();
				while(pending.load(std::memory_order_acquire) > 0) std::this_thread::yield();
				lock.lock();
				r.h.status = batch_err; r.message = batch_msg;
			}
			g_comb.leaders--;
			g_comb.callers--;
			g_comb.cv.notify_all();
			lock.unlock();
			lease.reset();
			return r.h.status;
		}
		g_comb.cv.wait(lock);
	}
	// a follower: copy my outputs, tell the leader
	g_comb.callers--;
	lock.unlock();
	if(r.h.status == CRTHIP_OK) copy_out(r);
	if(r.pending) r.pending->fetch_sub(1, std::memory_order_release);
	return r.h.status;
}

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

// src/decoder.cpp:104-114: the object takes the stream's q / strategy / N and the buffer, replaces the attribute, and is the Decoder's from here on.
// Its codec() tells decode() that the attribute's deltaDecode / postDelta / dequantize are host calls (anything but the three built-in ids;
// upstream's own convention for such objects is CUSTOM_CODEC).
bool Decoder::setAttribute(const char *name, char *buffer, VertexAttribute *attr) {
	auto it = data.find(name);
	if(it == data.end()) return false;
	VertexAttribute *found = it->second;
	if(found->codec_id != VertexAttribute::GENERIC_CODEC) throw "A custom codec object can replace a generic attribute only on the device path";
	attr->q = found->q;
	attr->strategy = found->strategy;
	attr->N = found->N;
	attr->buffer = buffer;
	attr->codec_id = VertexAttribute::CUSTOM_CODEC;
	delete found;
	it->second = attr;
	return true;
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
	Request r;
	bool custom = false;
	static_assert(sizeof(Face) == 12, "Face is three uint32");
	for(auto &it : data) {                                   // std::map order == the C ABI's attribute order (sorted by name)
		crthip_attr_binding b;
		b.buffer = it.second->buffer;
		b.format = (uint32_t)it.second->format;
		b.out_components = (uint32_t)it.second->out_components;
		b.stride = 0; b.reserved = 0;
		if(it.second->codec_id == VertexAttribute::CUSTOM_CODEC && b.buffer) {   // the stream's int32 values; the object's own steps follow below
			b.format = CRTHIP_FMT_INT32; b.reserved = CRTHIP_BIND_STREAM_VALUES; custom = true;
		}
		r.binds.push_back(b);
	}
	if(custom && nface) { index.prediction.resize(nvert); r.h.prediction = index.prediction.data(); }
	r.h.blob = input_; r.h.len = (size_t)len_;
	r.h.attrs = r.binds.empty() ? nullptr : r.binds.data();
	r.h.index = index.faces16 ? (void *)index.faces16 : (void *)index.faces32;   // faces16 wins (src/decoder.cpp:246-249)
	r.h.index_format = index.faces16 ? CRTHIP_FMT_UINT16 : CRTHIP_FMT_UINT32;
	const int err = combined_decode(r);                       // alone: one blob on a pool context; with other threads decoding: one batch for all
	if(err) raise(err);
	if(custom) {
		// the caller's codec objects, in upstream's order (src/decoder.cpp:186-193 for meshes, :141-146 for clouds: no postDelta there)
		std::vector<Face> none;
		for(auto &it : data) if(it.second->codec_id == VertexAttribute::CUSTOM_CODEC) it.second->deltaDecode(nvert, nface ? index.prediction : none);
		if(nface) for(auto &it : data) if(it.second->codec_id == VertexAttribute::CUSTOM_CODEC) it.second->postDelta(nvert, nface, data, index);
		for(auto &it : data) if(it.second->codec_id == VertexAttribute::CUSTOM_CODEC) it.second->dequantize(nvert);
	}
}

} // namespace crt
