// pool.cpp — the multi-GPU decode pool of include/corto_hip.h: a list of independent batches of .crt blobs
// decoded by every GPU of the node, one shared work queue, no collective.  Built on the public C ABI only (crthip_ctx /
// crthip_batch_*), plus hipMalloc for the lanes' output blocks.
//
// Reference anchor: independent crt::Decoder objects, src/decoder.cpp:126-196 (nothing shared between two decodes).
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/corto_hip.h"
#include "encoder_internal.h"

using corto_hip::ctx_fail;

namespace {

struct Lane {                       // one context = one batch in flight
	uint32_t slot = 0;              // pool device index
	int device = 0;
	crthip_ctx *ctx = nullptr;
	crthip_batch *batch = nullptr;
	void *out = nullptr; size_t out_cap = 0;
	void *host_out = nullptr; size_t host_cap = 0, out_used = 0;   // outputs_to_host: the pinned mirror of `out` (its first out_used bytes are copied behind every decode)
	// bindings of the item the batch object is planned for
	std::vector<crthip_attr_binding> binds;
	std::vector<void *> index_ptr;
	std::vector<uint32_t> index_fmt;
	std::vector<size_t> attr_off, index_off;     // byte offsets inside `out` (attr_off: per binding entry)
	std::vector<uint32_t> first_attr;            // first binding entry of blob i
	std::vector<int32_t> status;
	int64_t item = -1;              // item of the step in flight / last executed
	uint64_t step = 0;              // its global step number
	bool busy = false;
	bool poisoned = false;          // the output block was filled with POISON on the context's stream right before the step in flight / last executed
};
constexpr int POISON = 0xA5;

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

} // namespace

struct crthip_pool {
	uint32_t ndevices = 0, threads_per_device = 0, depth = 0;
	std::vector<int> devices;
	std::vector<Lane> lanes;        // [device][thread][depth]
	std::vector<std::vector<int>> cpus;   // per pool device: the host CPUs of the GPU's NUMA node (empty: unknown, threads are not pinned)
	std::string warning;            // what crthip_pool_create had to say about hardware queues (empty: nothing)
	bool render = false;            // crthip_pool_set_render_layouts: int16 normals, uint16 index where the blob's vertex ids fit
	bool to_host = false;           // crthip_pool_set_outputs_to_host: every step ends with a D2H copy of its outputs into the lane's pinned block
	// state of one run
	std::atomic<uint64_t> next{0}, completed{0};
	std::mutex m;
	int error = CRTHIP_OK; std::string error_msg;
};

static void destroy_lane(Lane &L) {
	(void)hipSetDevice(L.device);
	if(L.batch) crthip_batch_destroy(L.batch);
	if(L.ctx) crthip_ctx_destroy(L.ctx);
	if(L.out) (void)hipFree(L.out);
	if(L.host_out) (void)hipHostFree(L.host_out);
	L.batch = nullptr; L.ctx = nullptr; L.out = nullptr; L.out_cap = 0; L.host_out = nullptr; L.host_cap = 0;
}

extern "C" int crthip_pool_create(uint32_t ndevices, const int *devices, uint32_t threads_per_device, uint32_t depth, crthip_pool **out) {
	if(!out || ndevices == 0 || ndevices > 16 || threads_per_device == 0 || depth == 0 || threads_per_device*depth > 64) return ctx_fail(CRTHIP_E_ARGUMENT, nullptr);
	crthip_pool *p = new crthip_pool();
	p->ndevices = ndevices; p->threads_per_device = threads_per_device; p->depth = depth;
	for(uint32_t d = 0; d < ndevices; d++) p->devices.push_back(devices ? devices[d] : (int)d);
	p->lanes.resize((size_t)ndevices*threads_per_device*depth);
	uint32_t hw_queues = 4;                                      // ROCm's default
	bool hw_queues_set = false;
	{ const char *e = getenv("GPU_MAX_HW_QUEUES"); if(e && atoi(e) > 0) { hw_queues = (uint32_t)atoi(e); hw_queues_set = true; } }
	// contexts per PHYSICAL device: a device id may repeat (several pool devices on one GPU), and it is the GPU's hardware queues
	// that the contexts' streams share
	std::map<int, uint32_t> ctx_per_gpu;
	for(uint32_t d = 0; d < ndevices; d++) ctx_per_gpu[p->devices[d]] += threads_per_device*depth;
	for(size_t i = 0; i < p->lanes.size(); i++) {
		Lane &L = p->lanes[i];
		L.slot = (uint32_t)(i/((size_t)threads_per_device*depth)); L.device = p->devices[L.slot];
		int err = crthip_ctx_create(L.device, &L.ctx);
		// two HIP streams per context only while every stream of the GPU gets a hardware queue of its own (corto_hip.h)
		if(!err && 2*ctx_per_gpu[L.device] > hw_queues) err = crthip_ctx_set_single_stream(L.ctx, 1);
		if(err) { for(auto &x : p->lanes) destroy_lane(x); delete p; return err; }
	}
	for(auto &kv : ctx_per_gpu)
		if(kv.second > hw_queues && p->warning.empty()) {
			char buf[320];
			snprintf(buf, sizeof buf, "corto_hip pool: %u contexts on GPU %d but %u hardware queues (%s): streams that share a queue serialise each other's kernels; "
			         "export GPU_MAX_HW_QUEUES=%u before the process first touches HIP", kv.second, kv.first, hw_queues,
			         hw_queues_set ? "GPU_MAX_HW_QUEUES" : "ROCm's default; GPU_MAX_HW_QUEUES is not set", kv.second > 20 ? 20u : kv.second);   // (20 contexts on 20 queues measured best; 24 on 24 is slower)
			p->warning = buf;
			fprintf(stderr, "%s\n", buf);
		}
	// the host CPUs next to each GPU: PCI bus id -> /sys/bus/pci/devices/<id>/numa_node -> /sys/devices/system/node/node<N>/cpulist
	p->cpus.resize(ndevices);
	for(uint32_t d = 0; d < ndevices; d++) {
		char bus[64] = {0};
		if(hipDeviceGetPCIBusId(bus, (int)sizeof bus, p->devices[d]) != hipSuccess) { (void)hipGetLastError(); continue; }
		for(char *c = bus; *c; c++) if(*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
		char path[192]; int node = -1;
		snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
		if(FILE *f = fopen(path, "r")) { if(fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
		if(node < 0) continue;
		snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
		if(FILE *f = fopen(path, "r")) {
			int a, b; char sep;
			while(fscanf(f, "%d", &a) == 1) {
				b = a;
				if(fscanf(f, "%c", &sep) == 1 && sep == '-') { if(fscanf(f, "%d", &b) != 1) b = a; if(fscanf(f, "%c", &sep) != 1) sep = 0; }
				for(int c = a; c <= b && c < CPU_SETSIZE; c++) p->cpus[d].push_back(c);
				if(sep != ',') break;
			}
			fclose(f);
		}
	}
	*out = p;
	return CRTHIP_OK;
}

extern "C" void crthip_pool_destroy(crthip_pool *p) {
	if(!p) return;
	for(auto &L : p->lanes) destroy_lane(L);
	delete p;
}

extern "C" uint32_t crthip_pool_lanes(const crthip_pool *p) { return p ? (uint32_t)p->lanes.size() : 0; }
extern "C" const char *crthip_pool_warning(const crthip_pool *p) { return p ? p->warning.c_str() : ""; }
extern "C" int64_t crthip_pool_device_cpus(const crthip_pool *p, uint32_t device_slot, int32_t *cpus, size_t cap) {
	if(!p || device_slot >= p->ndevices) return ctx_fail(CRTHIP_E_ARGUMENT, nullptr);
	const std::vector<int> &c = p->cpus[device_slot];
	for(size_t i = 0; i < c.size() && i < cap && cpus; i++) cpus[i] = c[i];
	return (int64_t)c.size();
}
extern "C" int crthip_pool_set_outputs_to_host(crthip_pool *p, int on) {
	if(!p) return ctx_fail(CRTHIP_E_ARGUMENT, nullptr);
	p->to_host = on != 0;
	return CRTHIP_OK;
}
extern "C" int crthip_pool_set_render_layouts(crthip_pool *p, int on) {
	if(!p) return ctx_fail(CRTHIP_E_ARGUMENT, nullptr);
	p->render = on != 0;
	for(auto &L : p->lanes) { L.binds.clear(); L.item = -1; }      // the lanes lay their outputs out again
	return CRTHIP_OK;
}
extern "C" int crthip_pool_set_packed_host_blobs(crthip_pool *p, int on) {
	if(!p) return ctx_fail(CRTHIP_E_ARGUMENT, nullptr);
	for(auto &L : p->lanes) { const int err = crthip_ctx_set_packed_host_blobs(L.ctx, on); if(err) return err; }
	return CRTHIP_OK;
}

// Host-resident items (SURVEY.md 8d's primary region: .crt blobs in pinned host memory -> decoded outputs in HBM): crthip_batch_reset uploads
// a step's blobs itself, with ONE DMA copy at the head of the context's own stream.  Round 4 built and measured four ways of taking that
// copy off the step (C4 batch, 4 x 4 contexts; resident inputs 0.080 ms a step, this path 0.093, PCIe alone 0.075: tools/h2d_probe.py):
// a copy stream per worker thread with an event the context waits for (0.125, and 7 ms stalls), the lane's NEXT batch queued behind the
// running batch's kernels on the lane's own stream (0.15-0.20: a DMA copy queued behind kernels is started late), the same two with a
// copy KERNEL reading the pinned buffer over PCIe (0.13 / 0.156).  All slower; removed.  What does help a little is one more lane per
// thread's worth of contexts (5 x 4: 0.091): the lane whose blobs are on their way is idle for the GPU.
// plan `item` on the lane's batch object, lay its outputs out in the lane's device block and bind them
static int lane_plan(crthip_pool *p, Lane &L, const crthip_pool_item &it, int64_t item_id) {
	const void *arena = it.device_arena ? it.device_arena[L.slot] : nullptr;
	int err = L.batch ? crthip_batch_reset(L.batch, it.nblobs, it.blobs, it.lens, arena)
	                  : crthip_batch_create(L.ctx, it.nblobs, it.blobs, it.lens, arena, &L.batch);
	if(err) return err;
	if(L.item != item_id || L.binds.empty()) {                // the layout of an item's outputs depends on the item alone
		L.binds.clear(); L.attr_off.clear(); L.first_attr.assign(it.nblobs, 0);
		L.index_ptr.assign(it.nblobs, nullptr); L.index_fmt.assign(it.nblobs, CRTHIP_FMT_UINT32); L.index_off.assign(it.nblobs, 0);
		size_t off = 0;
		auto take = [&](size_t n) { off = (off + 255) & ~(size_t)255; const size_t r = off; off += n; return r; };
		crthip_blob_info info;
		for(uint32_t i = 0; i < it.nblobs; i++) {
			if((err = crthip_batch_info(L.batch, i, &info)) != 0) return err;
			L.first_attr[i] = (uint32_t)L.binds.size();
			for(uint32_t k = 0; k < info.nattr; k++) {
				const crthip_attr_info &a = info.attr[k];
				crthip_attr_binding b; b.buffer = nullptr; b.format = CRTHIP_FMT_FLOAT; b.out_components = 0; b.stride = 0; b.reserved = 0;
				size_t n;
				if(a.codec == CRTHIP_CODEC_NORMAL) { if(p->render) b.format = CRTHIP_FMT_INT16; n = (size_t)info.nvert*(p->render ? 6 : 12); }
				else if(a.codec == CRTHIP_CODEC_COLOR) { b.format = CRTHIP_FMT_UINT8; b.out_components = 4; n = (size_t)info.nvert*4; }
				else n = (size_t)info.nvert*a.components*4;
				L.attr_off.push_back(take(n));
				L.binds.push_back(b);
			}
			if(info.nface && p->render && info.nvert < 65536) L.index_fmt[i] = CRTHIP_FMT_UINT16;
			if(info.nface) L.index_off[i] = take((size_t)info.nface*(L.index_fmt[i] == CRTHIP_FMT_UINT16 ? 6 : 12));
		}
		const size_t total = off + 256;
		if(total > L.out_cap) {
			if(L.out) (void)hipFree(L.out);
			L.out = nullptr; L.out_cap = 0;
			if(hipMalloc(&L.out, total + total/8) != hipSuccess) return ctx_fail(CRTHIP_E_NOMEM, nullptr);
			L.out_cap = total + total/8;
		}
		uint8_t *base = (uint8_t *)L.out;
		for(size_t k = 0; k < L.binds.size(); k++) L.binds[k].buffer = base + L.attr_off[k];
		for(uint32_t i = 0; i < it.nblobs; i++) {
			if((err = crthip_batch_info(L.batch, i, &info)) != 0) return err;
			L.index_ptr[i] = info.nface ? base + L.index_off[i] : nullptr;
		}
		L.status.assign(it.nblobs, 0);
		L.out_used = off;
	}
	if(p->to_host && L.host_cap < L.out_cap) {                 // (first use: 32 MB of pinned memory a lane for a C4 item)
		if(L.host_out) (void)hipHostFree(L.host_out);
		L.host_out = nullptr; L.host_cap = 0;
		if(hipHostMalloc(&L.host_out, L.out_cap, hipHostMallocDefault) != hipSuccess) return ctx_fail(CRTHIP_E_NOMEM, nullptr);
		L.host_cap = L.out_cap;
	}
	L.item = item_id;
	return crthip_batch_bind_all(L.batch, L.binds.data(), L.index_ptr.data(), L.index_fmt.data());
}

extern "C" int crthip_pool_run(crthip_pool *p, uint32_t nitems, const crthip_pool_item *items, uint64_t steps, uint64_t warmup,
                               crthip_pool_report *report, double *completion_s) {
	if(!p || !items || nitems == 0 || !report) return ctx_fail(CRTHIP_E_ARGUMENT, nullptr);
	memset(report, 0, sizeof(*report));
	// triangles / vertices of every item (header parse only)
	std::vector<uint64_t> item_tris(nitems, 0), item_verts(nitems, 0);
	for(uint32_t j = 0; j < nitems; j++)
		for(uint32_t i = 0; i < items[j].nblobs; i++) {
			crthip_blob_info info;
			const int err = crthip_probe(items[j].blobs[i], items[j].lens[i], &info);
			if(err) return err;
			item_tris[j] += info.nface; item_verts[j] += info.nvert;
		}
	const uint64_t timed_end = warmup + steps;
	const uint64_t total = timed_end + p->lanes.size();      // the tail keeps every context busy until the last timed completion
	p->next = 0; p->completed = 0; p->error = CRTHIP_OK; p->error_msg.clear();
	for(auto &L : p->lanes) { L.busy = false; L.item = -1; L.binds.clear(); }   // (an item id means this call's items[] only)
	std::vector<double> stamps(timed_end + 1, 0.0);           // stamps[c] = time at which the c-th completion happened (1-based)
	std::atomic<uint64_t> failed{0}, fallbacks{0}, tris{0}, verts{0};
	std::vector<std::atomic<uint64_t>> per_dev(p->ndevices);
	for(auto &x : per_dev) x = 0;
	std::atomic<int32_t> first_error{0};
	std::atomic<uint64_t> host_ns{0}, host_steps{0}, wait_ns{0}, finish_ns{0}, plan_ns{0}, plan_max_ns{0}, launch_max_ns{0};
	auto raise_max = [](std::atomic<uint64_t> &m, uint64_t v) { uint64_t cur = m.load(); while(v > cur && !m.compare_exchange_weak(cur, v)) { } };
	// home shard first: pool device d owns the items j with j % ndevices == d (its shard is resident there), and a device without a home
	// item takes from the others' ("stealing" in a cyclic run: it is the work list that is shared, a faster GPU simply draws more tickets)
	std::vector<std::vector<uint32_t>> home(p->ndevices);
	for(uint32_t j = 0; j < nitems; j++) home[j % p->ndevices].push_back(j);
	std::vector<std::atomic<uint64_t>> home_next(p->ndevices);
	for(auto &x : home_next) x = 0;
	std::atomic<uint64_t> stolen{0};
	// the last round of timed steps and the tail behind them (outputs_to_host: the tail only - poisoning the pinned mirror is 32 MB of memset on the worker thread)
	const uint64_t poison_from = p->to_host ? timed_end : timed_end > p->lanes.size() ? timed_end - p->lanes.size() : 0;
	const double t_launch = now_s();
	stamps[0] = t_launch;

	auto worker = [&](uint32_t slot, uint32_t t) {
		(void)hipSetDevice(p->devices[slot]);
		if(!p->cpus[slot].empty()) {                             // next to the GPU: plan + launch are ~200 us of host work per step and thread
			cpu_set_t set; CPU_ZERO(&set);
			for(int c : p->cpus[slot]) CPU_SET(c, &set);
			(void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);   // (a cpuset that forbids them: stay where we are)
		}
		Lane *mine = &p->lanes[((size_t)slot*p->threads_per_device + t)*p->depth];
		auto finish = [&](Lane &L) -> int {
			const int rc = crthip_batch_sync(L.batch, L.status.data());
			L.busy = false;
			const uint64_t c = ++p->completed;                   // completion order
			if(c <= timed_end) stamps[c] = now_s();
			uint64_t bad = 0;
			for(int32_t s : L.status) if(s) { bad++; int32_t z = 0; first_error.compare_exchange_strong(z, s); }
			failed += bad;
			crthip_batch_stats st;
			if(crthip_batch_get_stats(L.batch, &st) == CRTHIP_OK) fallbacks += st.topology_fallbacks;
			if(c > warmup && c <= timed_end) { per_dev[slot]++; tris += item_tris[(size_t)L.item]; verts += item_verts[(size_t)L.item]; }
			if(rc == CRTHIP_E_DEVICE || rc == CRTHIP_E_NOMEM) return rc;
			return CRTHIP_OK;
		};
		int err = CRTHIP_OK;
		auto tick = [] { return std::chrono::steady_clock::now(); };
		auto ns_since = [](std::chrono::steady_clock::time_point t0) { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); };
		for(uint64_t n = 0; !err; n++) {
			const auto w0 = tick();
			// the next lane to refill: a free one, else whichever of the busy ones finishes first (they mostly finish in the order they
			// were launched, but a thread that waited on the oldest while a younger one was done left that context idle)
			uint32_t pick = p->depth;
			for(uint32_t k = 0; k < p->depth && pick == p->depth; k++) if(!mine[(n + k) % p->depth].busy) pick = (uint32_t)((n + k) % p->depth);
			for(uint32_t spins = 0; pick == p->depth && !err; spins++) {
				for(uint32_t k = 0; k < p->depth; k++) {
					Lane &C = mine[(n + k) % p->depth];
					const int d = crthip_batch_done(C.batch);
					if(d < 0) { err = d; break; }
					if(d) { pick = (uint32_t)((n + k) % p->depth); break; }
				}
				if(pick == p->depth && !err) { if(spins < 64) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(5)); }
			}
			if(err) break;
			wait_ns += ns_since(w0);
			Lane &L = mine[pick];
			const auto f0 = tick();
			if(L.busy) err = finish(L);
			finish_ns += ns_since(f0);
			if(err) break;
			const uint64_t step = p->next.fetch_add(1);
			if(step >= total) break;
			uint32_t j;
			if(!home[slot].empty()) j = home[slot][home_next[slot].fetch_add(1) % home[slot].size()];
			else j = (uint32_t)(stolen.fetch_add(1) % nitems);
			const auto h0 = std::chrono::steady_clock::now();
			err = lane_plan(p, L, items[j], (int64_t)j);
			{ const uint64_t ns_ = ns_since(h0); plan_ns += ns_; raise_max(plan_max_ns, ns_); }
			// the outputs every lane holds after the run were written by a step that STARTED from a poisoned block: the post-run bit-exact
			// check cannot pass on bytes an earlier step left behind (on the context's own stream: ordered before the step's kernels)
			L.poisoned = false;
			if(!err && step >= poison_from && L.out) {
				err = corto_hip::ctx_fill_async(L.ctx, L.out, L.out_cap, POISON);
				if(!err && p->to_host && L.host_out) memset(L.host_out, POISON, L.out_used);
				if(!err) L.poisoned = true;
			}
			const auto d0 = std::chrono::steady_clock::now();
			if(!err) err = crthip_batch_decode(L.batch);
			if(!err && p->to_host) err = corto_hip::ctx_copy_to_host_async(L.ctx, L.host_out, L.out, L.out_used);
			raise_max(launch_max_ns, ns_since(d0));
			host_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - h0).count(); host_steps++;
			if(!err) { L.busy = true; L.step = step; }
		}
		for(uint32_t k = 0; k < p->depth; k++) if(mine[k].busy) { const int e2 = finish(mine[k]); if(!err) err = e2; }
		// a context whose thread drew none of the last 2 x lanes tickets (descheduled while the others emptied the queue: seen with
		// 32-blob items) repeats its last step from a poisoned block, behind the timed region: what lane_read returns is then ALWAYS
		// a poisoned step's output, not only almost always
		for(uint32_t k = 0; k < p->depth && !err; k++) {
			Lane &L = mine[k];
			if(L.item < 0) {                                     // ... and one that drew no ticket at all (a 28-step run on a cold box) decodes its device's first item
				const uint32_t j = home[slot].empty() ? 0u : home[slot][0];
				err = lane_plan(p, L, items[j], (int64_t)j);
				if(err) break;
			}
			if(L.poisoned || !L.out) continue;
			err = corto_hip::ctx_fill_async(L.ctx, L.out, L.out_cap, POISON);
			if(!err && p->to_host && L.host_out) memset(L.host_out, POISON, L.out_used);
			if(!err) err = crthip_batch_decode(L.batch);
			if(!err && p->to_host) err = corto_hip::ctx_copy_to_host_async(L.ctx, L.host_out, L.out, L.out_used);
			if(!err) { L.busy = true; L.poisoned = true; err = finish(L); }
		}
		if(err) {
			std::lock_guard<std::mutex> lock(p->m);
			if(!p->error) { p->error = err; p->error_msg = crthip_last_error(); }
			p->next = total;                                     // stop handing out work
		}
	};
	std::vector<std::thread> threads;
	for(uint32_t d = 0; d < p->ndevices; d++)
		for(uint32_t t = 0; t < p->threads_per_device; t++) threads.emplace_back(worker, d, t);
	for(auto &th : threads) th.join();
	if(p->error) return ctx_fail(p->error, p->error_msg.c_str());
	report->elapsed_s = stamps[timed_end] - stamps[warmup];
	report->steps = steps; report->triangles = tris; report->vertices = verts;
	report->failed_blobs = failed; report->first_error = first_error; report->topology_fallbacks = fallbacks;
	for(uint32_t d = 0; d < p->ndevices; d++) { report->steps_per_device[d] = per_dev[d]; if(per_dev[d]) report->devices_used++; }
	for(auto &L : p->lanes) if(L.item >= 0 && L.poisoned) report->poisoned_lanes++;
	report->host_us_per_step = host_steps ? (float)((double)host_ns/1e3/(double)host_steps) : 0.f;
	if(host_steps) {
		report->host_wait_us = (float)((double)wait_ns/1e3/(double)host_steps); report->host_finish_us = (float)((double)finish_ns/1e3/(double)host_steps);
		report->host_plan_us = (float)((double)plan_ns/1e3/(double)host_steps);
		report->host_plan_max_us = (float)((double)plan_max_ns/1e3); report->host_launch_max_us = (float)((double)launch_max_ns/1e3);
	}
	for(uint32_t d = 0; d < p->ndevices; d++) if(!p->cpus[d].empty()) report->pinned_devices++;
	if(completion_s) for(uint64_t c = 0; c < steps; c++) completion_s[c] = stamps[warmup + 1 + c] - stamps[warmup];
	return CRTHIP_OK;
}

extern "C" int64_t crthip_pool_lane_item(const crthip_pool *p, uint32_t lane, uint32_t *device_slot) {
	if(!p || lane >= p->lanes.size()) return ctx_fail(CRTHIP_E_ARGUMENT, nullptr);
	if(device_slot) *device_slot = p->lanes[lane].slot;
	return p->lanes[lane].item;
}

extern "C" int64_t crthip_pool_lane_read(crthip_pool *p, uint32_t lane, uint32_t blob, const char *what, void *host_out, size_t cap) {
	if(!p || lane >= p->lanes.size() || !what || !host_out) return ctx_fail(CRTHIP_E_ARGUMENT, nullptr);
	Lane &L = p->lanes[lane];
	if(!L.batch || L.item < 0 || blob >= crthip_batch_size(L.batch)) return ctx_fail(CRTHIP_E_ARGUMENT, nullptr);
	crthip_blob_info info;
	int err = crthip_batch_info(L.batch, blob, &info);
	if(err) return err;
	const uint8_t *src = nullptr; size_t n = 0;
	if(!strcmp(what, "#tail")) { n = L.out_cap < 256 ? L.out_cap : 256; src = (const uint8_t *)L.out + (L.out_cap - n); }   // the block's last bytes: behind every output array
	else if(!strcmp(what, "index")) { if(!info.nface) return 0; src = (const uint8_t *)L.index_ptr[blob]; n = (size_t)info.nface*(L.index_fmt[blob] == CRTHIP_FMT_UINT16 ? 6 : 12); }
	else {
		for(uint32_t k = 0; k < info.nattr; k++) if(!strcmp(info.attr[k].name, what)) {
			const crthip_attr_info &a = info.attr[k];
			src = (const uint8_t *)L.binds[L.first_attr[blob] + k].buffer;
			n = a.codec == CRTHIP_CODEC_NORMAL ? (size_t)info.nvert*(L.binds[L.first_attr[blob] + k].format == CRTHIP_FMT_INT16 ? 6 : 12) : a.codec == CRTHIP_CODEC_COLOR ? (size_t)info.nvert*4 : (size_t)info.nvert*a.components*4;
		}
		if(!src) return ctx_fail(CRTHIP_E_ARGUMENT, "no such attribute");
	}
	if(n > cap) n = cap;
	// outputs_to_host: what the step's own D2H copy left in the lane's pinned block (the tail lies behind the copied range: from the device)
	if(p->to_host && L.host_out && strcmp(what, "#tail")) { memcpy(host_out, (const uint8_t *)L.host_out + (src - (const uint8_t *)L.out), n); return (int64_t)n; }
	if(hipSetDevice(L.device) != hipSuccess || hipMemcpy(host_out, src, n, hipMemcpyDeviceToHost) != hipSuccess) return ctx_fail(CRTHIP_E_DEVICE, nullptr);
	return (int64_t)n;
}
