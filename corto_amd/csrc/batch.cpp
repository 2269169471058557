// batch.cpp — host side of the MI355X decode path: context (stream, scratch pool), batch planner
// (walk -> job descriptors -> one H2D upload) and the kernel schedule.  Compiled by hipcc.
//
// The schedule restates the stage order of crt::Decoder::decodeMesh / decodePointCloud
// (src/decoder.cpp:133-196) for a whole batch of blobs at once:
//   decode-all (Tunstall + bit-unpack) -> topology -> delta-all -> postDelta (normals) -> dequantize-all.
#include <chrono>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/corto_hip.h"
#include "crt_format.h"
#include "debug_config.h"
#include "device_plan.h"
#include "encoder_internal.h"
#include "kernels.h"

using namespace corto_hip;

// ------------------------------------------------------------------------------------------------
static thread_local std::string g_error;
static int fail(int code, const std::string &msg) { g_error = msg; return code; }

extern "C" const char *crthip_strerror(int code) {
	switch(code) {
	case CRTHIP_OK: return "ok";
	case CRTHIP_E_ALIGN: return "Memory must be alignegned on 4 bytes.";
	case CRTHIP_E_MAGIC: return "Not a crt file.";
	case CRTHIP_E_TRUNCATED: return "Truncated or inconsistent crt stream.";
	case CRTHIP_E_ENTROPY: return "Unknown entropy";
	case CRTHIP_E_TOPOLOGY: return "Decoding topology failed";
	case CRTHIP_E_NORMAL_NEEDS_POSITION: return "No position attribute found. Use DIFF normal strategy instead.";
	case CRTHIP_E_FORMAT: return "Format not supported for this attribute on the device path";
	case CRTHIP_E_ARGUMENT: return "Invalid argument";
	case CRTHIP_E_DEVICE: return "No usable HIP device (the MI355X path has no CPU fallback)";
	case CRTHIP_E_NOMEM: return "Out of memory";
	case CRTHIP_E_LIMIT: return "Too many attributes or components for this build";
	}
	return "unknown error";
}
static int fail(int code) { return fail(code, crthip_strerror(code)); }
// context plumbing for the encoder stages (encoder_internal.h)
namespace corto_hip { int ctx_fail(int code, const char *msg) { return fail(code, msg ? std::string(msg) : std::string(crthip_strerror(code))); } }
extern "C" const char *crthip_last_error(void) { return g_error.c_str(); }
extern "C" uint32_t crthip_abi_version(void) { return CRTHIP_ABI_VERSION; }

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if(e_ != hipSuccess) return fail(CRTHIP_E_DEVICE, std::string(#expr ": ") + hipGetErrorString(e_)); } while(0)

// ------------------------------------------------------------------------------------------------
// host-only probes
static void fill_info(const BlobHeader &h, crthip_blob_info *info) {
	memset(info, 0, sizeof(*info));
	info->version = h.version; info->entropy = h.entropy; info->nvert = h.nvert; info->nface = h.nface;
	info->nattr = (uint32_t)h.attrs.size(); info->nexif = (uint32_t)h.exif.size(); info->body_offset = h.body_offset;
	for(size_t i = 0; i < h.attrs.size(); i++) {
		crthip_attr_info &a = info->attr[i];
		strncpy(a.name, h.attrs[i].name.c_str(), CRTHIP_NAME_MAX - 1);
		a.codec = h.attrs[i].codec; a.q = h.attrs[i].q; a.components = h.attrs[i].N;
		a.format = h.attrs[i].format; a.strategy = h.attrs[i].strategy;
	}
}

extern "C" int crthip_probe(const uint8_t *blob, size_t len, crthip_blob_info *info) {
	if(!blob || !info) return fail(CRTHIP_E_ARGUMENT);
	BlobHeader h;
	int err = parse_header(blob, len, h);
	if(err) return fail(err);
	fill_info(h, info);
	return CRTHIP_OK;
}

extern "C" int64_t crthip_probe_exif(const uint8_t *blob, size_t len, char *out, size_t cap) {
	BlobHeader h;
	int err = parse_header(blob, len, h);
	if(err) return fail(err);
	std::string flat;
	for(auto &kv : h.exif) { flat += kv.first; flat.push_back('\0'); flat += kv.second; flat.push_back('\0'); }
	if(out && cap >= flat.size()) memcpy(out, flat.data(), flat.size());
	return (int64_t)flat.size();
}

extern "C" int64_t crthip_probe_groups(const uint8_t *blob, size_t len, uint32_t *group_end, size_t cap) {
	BlobLayout L;
	int err = walk_blob(blob, len, L);
	if(err) return fail(err);
	for(size_t i = 0; i < L.group_end.size() && i < cap; i++) group_end[i] = L.group_end[i];
	return (int64_t)L.group_end.size();
}

extern "C" int64_t crthip_probe_group_props(const uint8_t *blob, size_t len, uint32_t g, char *out, size_t cap) {
	BlobLayout L;
	int err = walk_blob(blob, len, L);
	if(err) return fail(err);
	if(g >= L.group_props.size()) return fail(CRTHIP_E_ARGUMENT);
	std::string flat;
	for(auto &kv : L.group_props[g]) { flat += kv.first; flat.push_back('\0'); flat += kv.second; flat.push_back('\0'); }
	if(out && cap >= flat.size()) memcpy(out, flat.data(), flat.size());
	return (int64_t)flat.size();
}

extern "C" uint64_t crthip_arena_layout(uint32_t nblobs, const uint32_t *lens, uint64_t *offsets) {
	uint64_t off = 0;
	for(uint32_t i = 0; i < nblobs; i++) {
		if(offsets) offsets[i] = off;
		off += ((uint64_t)lens[i] + 15) & ~15ull;
	}
	return off;
}

// ------------------------------------------------------------------------------------------------
struct DeviceBuf {
	void *p = nullptr; size_t cap = 0;
	int reserve(size_t n) {
		if(n <= cap) return CRTHIP_OK;
		if(p) { (void)hipFree(p); p = nullptr; cap = 0; }
		size_t want = std::max(n + n/4, (size_t)1 << 20);
		if(hipMalloc(&p, want) != hipSuccess) { p = nullptr; return CRTHIP_E_NOMEM; }
		cap = want;
		return CRTHIP_OK;
	}
	void release() { if(p) (void)hipFree(p); p = nullptr; cap = 0; }
};
struct PinnedBuf {
	void *p = nullptr; size_t cap = 0;
	int reserve(size_t n) {
		if(n <= cap) return CRTHIP_OK;
		if(p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
		size_t want = std::max(n + n/4, (size_t)1 << 16);
		if(hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr; return CRTHIP_E_NOMEM; }
		cap = want;
		return CRTHIP_OK;
	}
	void release() { if(p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

struct KernelTimer {
	std::vector<hipEvent_t> pool;
	struct Rec { const char *name; size_t e0, e1; };
	std::vector<Rec> recs;
	size_t used = 0;
	hipEvent_t get() {
		if(used == pool.size()) { hipEvent_t e; (void)hipEventCreate(&e); pool.push_back(e); }
		return pool[used++];
	}
	void reset() { used = 0; recs.clear(); }
	void release() { for(auto e : pool) (void)hipEventDestroy(e); pool.clear(); reset(); }
};

namespace {
template <typename T> struct HostArr {  // host image of a device array + where it goes
	std::vector<T> v; uint64_t dev_off = 0;
};

struct AttrScratch {
	uint64_t color = ~0ull, diffs = ~0ull, fired = ~0ull, vals = ~0ull, facen = ~0ull; std::vector<uint64_t> sym;   // vals: int32 workspace of a generic attribute bound with a stride; facen: the fused normal kernel's face normals when not in LDS
	void reset() { color = diffs = fired = vals = facen = ~0ull; sym.clear(); }
};
struct BlobScratch {
	uint64_t clers = ~0ull, pred = ~0ull, front_a = ~0ull, front_b = ~0ull, order = ~0ull, delayed = ~0ull, faces = ~0ull;
	uint32_t front_cap = 0, aux_groups = 0;
	std::vector<AttrScratch> attr;
	size_t nattr = 0;                                       // attr[0..nattr) are this decode's (the vector only grows)
	void reset() { clers = pred = front_a = front_b = order = delayed = faces = ~0ull; front_cap = 0; aux_groups = 0; nattr = 0; }
	void set_attrs(size_t n) { if(attr.size() < n) attr.resize(n); for(size_t k = nattr; k < n; k++) attr[k].reset(); if(n > nattr) nattr = n; }
};

struct Plan {
	// job arrays
	HostArr<TunStream> tun, tun_dict; HostArr<uint32_t> tun_chunk_stream;   // tun_dict: one entry per DISTINCT probability table (shared dictionaries)
	HostArr<uint32_t> tun_group_ids; HostArr<TunGroup> tun_groups; uint32_t clers_groups = 0;   // streams by dictionary, in groups of one dictionary each (k_tun_stream_grouped)
	HostArr<FillJob> fill;
	HostArr<TopoJob> topo; HostArr<uint32_t> aux_u32;     // group_end lists
	HostArr<uint32_t> topo_lds_ids, topo_big_ids, topo_glob_ids; uint32_t topo_lds = 0, topo_big_lds = 0;   // LDS automata in two size classes, one launch each
	std::vector<uint32_t> topo_need;                                        // LDS bytes of topo_lds_ids' entries until they are split into the two classes
	HostArr<UnpackJob> unpack; HostArr<uint32_t> unpack_chunk_job, unpack_wave_ids;   // (unpack_wave_ids: the streams of small bit blocks, one wave each: k_unpack_wave)
	HostArr<DeltaJob> delta;
	HostArr<DeltaGroup> delta_groups;                       // blobs whose attributes share one k_delta_lds16 workgroup
	HostArr<CloudJob> cloud; HostArr<uint32_t> cloud_chunk_job;
	HostArr<NormalJob> normal; HostArr<uint32_t> nv_block_job, nv_block_first, nf_block_job, nf_block_first, normal_fused_ids;
	uint32_t normal_fused_lds = 0;
	HostArr<DequantJob> dequant; HostArr<uint32_t> dequant_block_job;
	// scratch regions (offsets)
	uint64_t zero_begin = 0, zero_end = 0;
	uint64_t status_off = 0, tables_off = 0, tun_partial_off = 0, unpack_partial_off = 0, cloud_partial_off = 0;
	uint64_t facen_off = 0, cnt_off = 0, cursor_off = 0, bnd_off = 0, start_off = 0, flag_off = 0, slot_off = 0, adj_off = 0, nscan_partial_off = 0;
	uint64_t jobs_begin = 0, jobs_bytes = 0;
	uint32_t est_nvert = 0, est_nface = 0;                // totals over ESTIMATED/BORDER jobs
	uint32_t delta16_lds = 0;                              // largest LDS request among the k_delta_lds16 groups
	bool tun_multi_chunk = false, any_diff_normal = false, any_est_normal = false;
	uint32_t tun_max_nchunks = 0;
	uint64_t total = 0;
	template <typename A> static void clr(A &a) { a.v.clear(); a.dev_off = 0; }
	void reset() {                                          // keep every vector's capacity
		clr(tun); clr(tun_dict); clr(tun_chunk_stream); clr(tun_group_ids); clr(tun_groups); clers_groups = 0; clr(fill); clr(topo); clr(aux_u32); clr(topo_lds_ids); clr(topo_big_ids); clr(topo_glob_ids); topo_need.clear();
		clr(unpack); clr(unpack_chunk_job); clr(unpack_wave_ids); clr(delta); clr(delta_groups); clr(cloud); clr(cloud_chunk_job); clr(normal); clr(nv_block_job); clr(nv_block_first);
		clr(nf_block_job); clr(nf_block_first); clr(normal_fused_ids); clr(dequant); clr(dequant_block_job);
		topo_lds = topo_big_lds = normal_fused_lds = 0;
		zero_begin = zero_end = status_off = tables_off = tun_partial_off = unpack_partial_off = cloud_partial_off = 0;
		facen_off = cnt_off = cursor_off = bnd_off = start_off = flag_off = slot_off = adj_off = nscan_partial_off = 0;
		jobs_begin = jobs_bytes = 0; est_nvert = est_nface = 0; delta16_lds = 0;
		tun_multi_chunk = any_diff_normal = any_est_normal = false; total = 0; tun_max_nchunks = 0;
	}
};
} // namespace

struct crthip_ctx {
	int device = 0;
	hipStream_t stream = nullptr;
	hipStream_t stream2 = nullptr;  // attribute streams (Tunstall + bit-unpack) run here while the main stream does topology
	hipEvent_t ev_fork = nullptr, ev_join = nullptr;
	hipEvent_t ev_done = nullptr;   // recorded behind a decode's last kernel: what sync / done wait for, so that work a caller queues on the stream BEHIND a decode
	                                // does not delay the harvest of this one
	uint32_t upload_seq = 0, done_covers_seq = 0;   // arena_pin uploads enqueued so far / how many of them sit IN FRONT of ev_done (harvest may only call those complete)
	DebugConfig dbg;                // every environment switch, read once when the context is made (debug_config.h)
	uint32_t normal_fn_max = NORMAL_FN_LDS_MAX;       // largest LDS request for which K-NRM keeps its face normals in LDS (0 for a context that is one of many: crthip_ctx_set_single_stream)
	uint8_t single_stream = 0;                        // crthip_ctx_set_single_stream: no second HIP stream for the attribute streams
	DeviceBuf scratch;        // symbols, tables, fronts, predictions, job arrays ... (one batch in flight at a time)
	PinnedBuf staging;        // host image of the job arrays
	PinnedBuf arena_pin;      // host image of a batch's blobs on their way to the device (batch_fill: one H2D copy, not waited for)
	PinnedBuf status_host;
	bool profiling = false;
	KernelTimer timer;
	crthip_batch *in_flight = nullptr;   // decode enqueued, status not harvested yet
	bool packed_host = false;            // crthip_ctx_set_packed_host_blobs: blobs laid out as an arena in the caller's pinned memory go up from there
	bool arena_upload_pending = false;   // a batch's blobs are (perhaps still) on their way from arena_pin: cleared by whoever synchronises the stream
	crthip_batch *last_decoded = nullptr;// whose intermediates the scratch block holds (crthip_batch_debug_read)
	// crthip_decode_host: everything a one-blob decode with host buffers needs, kept from call to call (no hipMalloc / create in the
	// steady state) and guarded by a mutex so that callers may share a context between threads
	std::mutex host_mutex;
	crthip_batch *host_batch = nullptr;
	DeviceBuf host_out;       // decoded outputs of the one blob, back to back
	PinnedBuf host_pin;       // ... and their landing zone in pinned host memory (one async D2H copy)
	// feedback on the LDS edge slots of the CLERS automaton: raised after a batch with fallbacks, lowered after a long calm run
	uint32_t topo_scale = 1, topo_pool_q8 = 8, topo_pool_cap = 0, topo_calm = 0, topo_patience = 64;   // K-TOPO's learnt slots: ring x topo_scale (a power of two), pool x topo_pool_q8 / 8 (kernels.h: topo_lds_geometry)
	// planner state reused from one decode call to the next (batch.cpp: build_and_launch)
	Plan plan;
	std::vector<BlobScratch> plan_scratch;
	std::vector<const uint8_t *> plan_clers, plan_logs;
	// streams of a batch that carry the same probability table share one dictionary: exact match on the table's bytes (alphabets of up
	// to 16 symbols; bigger ones hardly ever repeat and are quick to build), open addressing on a hash of them
	struct DictKey { uint8_t n, bytes[32]; };
	std::vector<DictKey> dict_keys;
	std::vector<uint32_t> dict_slots, dict_used, dict_ids, dict_count;
	bool delta_wide = false;                          // K-DELTA keeps 32-bit values in LDS: $CORTO_DELTA_WIDE=1, or learnt from a batch whose 16-bit relative values overflowed
	uint32_t delta_calm = 0, delta_patience = 256;
	bool delta_just_narrowed = false;                  // the narrow layout is on trial again after a wide spell (an overflow now doubles the patience)
};

// bytes per component of a generic attribute's caller buffer: upstream decodes in place as int32 whatever the format and DOUBLE widens
// in place (include/corto/vertex_attribute.h:184-228), so every format's buffer is nvert*N*4 bytes but DOUBLE's
static inline size_t generic_work_bytes(uint32_t format) { return format == CRTHIP_FMT_DOUBLE ? 8u : 4u; }

struct Binding { void *buffer = nullptr; uint32_t format = CRTHIP_FMT_FLOAT, out_components = 4, stride = 0; };

struct BlobPlan {
	BlobLayout L;
	uint64_t arena_off = 0;
	uint32_t len = 0;
	std::vector<Binding> bind;
	void *index = nullptr; uint32_t index_u16 = 0;
	int32_t host_status = 0;   // set by the planner (e.g. unsupported format), overrides device status
	// debug handles (scratch offsets valid after decode)
	uint64_t dbg_clers = ~0ull, dbg_pred = ~0ull; uint32_t dbg_nclers = 0;
	bool clers_in_arena = false;
};

struct crthip_batch {
	crthip_ctx *ctx = nullptr;
	std::vector<BlobPlan> blobs;
	const uint8_t *d_arena = nullptr;
	DeviceBuf own_arena;
	uint64_t arena_bytes = 0;
	bool dirty = true;
	crthip_batch_stats stats{};
	std::vector<int32_t> status;
	bool decoded = false;
	bool planned_wide = false;          // the decode in flight was planned with K-DELTA's 32-bit layout
};

// Wait for the batch in flight on this context (if any) and move its per-blob status from the pinned landing zone into the
// batch object.  EVERY place that is about to reuse the context's stream, scratch or status buffer goes through here, so a
// decode(A); decode(B); sync(A) sequence still reports A's failures (status used to be read only by crthip_batch_sync(A) and
// was lost when another call had synchronised first).
static int harvest(crthip_ctx *ctx) {
	crthip_batch *b = ctx->in_flight;
	if(!b) return CRTHIP_OK;
	if(hipEventSynchronize(ctx->ev_done) != hipSuccess) { ctx->in_flight = nullptr; return CRTHIP_E_DEVICE; }
	if(ctx->done_covers_seq == ctx->upload_seq) ctx->arena_upload_pending = false;   // (only an upload that sits in front of the event: one enqueued behind the decode is still on its way)
	const int32_t *hs = (const int32_t *)ctx->status_host.p;
	const size_t n = b->blobs.size();
	for(size_t i = 0; i < n; i++) b->status[i] = b->blobs[i].host_status ? b->blobs[i].host_status : hs[i];
	b->stats.topology_fallbacks = 0;
	for(size_t i = 0; i < n; i++) b->stats.topology_fallbacks += (uint64_t)(hs[n + i] & 1);
	// K-DELTA keeps values in LDS as int16 relative to vertex 0 and redoes an attribute that does not fit in HBM (slow, exact): a context
	// that meets such blobs (positions quantised beyond 15 bits) plans its next batches with 32-bit values; a long run of batches later it
	// tries the narrow layout again, with growing patience if that turns out wrong
	b->stats.delta_redone = 0; b->stats.delta_walked = 0;
	for(size_t i = 0; i < n; i++) b->stats.delta_walked += (uint32_t)(hs[2*n + 2*i + 1] != 0);
	if(!b->planned_wide) {
		for(size_t i = 0; i < n; i++) b->stats.delta_redone += (uint64_t)(hs[2*n + 2*i] != 0);
		if(b->stats.delta_redone) {
			ctx->delta_wide = true;
			if(ctx->delta_just_narrowed && ctx->delta_patience < (1u << 20)) ctx->delta_patience *= 2;   // overflowed right after narrowing again (never on a context's first overflow)
			ctx->delta_calm = 0;
		}
		ctx->delta_just_narrowed = false;                                   // (narrow and fine, or wide from here on)
	} else if(!ctx->dbg.delta_wide && ++ctx->delta_calm >= ctx->delta_patience) { ctx->delta_wide = false; ctx->delta_calm = 0; ctx->delta_just_narrowed = true; }
	// more than one blob in twenty redone on the HBM front (5x slower): more edge slots from the next batch on - as many as the redone blobs say they
	// would have needed (k_topology_lds' redo reports ring and pool slots above bit 0 of the flags word; round 4 went up four-fold whatever was
	// missing, and a batch of Delaunay discs that needed 600 pool slots of its 512 got 2 048 + 2 048: 72 KB of LDS a blob, two automata a CU, the
	// pipeline at a third of its rate); a long run without any: try less again, and be more patient the next time that turns out to be too little
	if(b->stats.topology_fallbacks*20 > n) {
		uint32_t ring_m = 1, pool_q8 = 8;
		const uint32_t cap_before = ctx->topo_pool_cap;
		for(size_t i = 0; i < n; i++) if(hs[n + i] & 1) {
			const auto &h = b->blobs[i].L.h;
			uint32_t ring, pool, symwin;
			topo_lds_geometry(h.nface, b->blobs[i].L.clers.size, 4096, 1, 8, n >= 32 ? 4u : 8u, topo_boundary_estimate(h.nvert, h.nface), ring, pool, symwin);
			const uint32_t need_ring = ((uint32_t)hs[n + i] >> 1) & 0x7FFFu, need_pool = (uint32_t)hs[n + i] >> 16;
			uint32_t m = 1;
			while(ring*m < need_ring && m < 16) m <<= 1;
			ring_m = std::max(ring_m, m);
			pool_q8 = std::max(pool_q8, std::min(128u, (need_pool*8 + pool - 1)/pool + 1u));      // (an eighth on top)
			ctx->topo_pool_cap = std::max(ctx->topo_pool_cap, need_pool + need_pool/8);
		}
		const bool grew = ring_m > ctx->topo_scale || pool_q8 > ctx->topo_pool_q8 || ctx->topo_pool_cap > cap_before;
		ctx->topo_scale = std::max(ctx->topo_scale, ring_m); ctx->topo_pool_q8 = std::max(ctx->topo_pool_q8, pool_q8);
		if(!grew) { ctx->topo_scale = std::min(16u, ctx->topo_scale*2); ctx->topo_pool_q8 = std::min(128u, ctx->topo_pool_q8*2); ctx->topo_pool_cap *= 2; }   // (they fell back with what they asked for: capacity, or a need the redo cannot see)
		if(ctx->topo_calm == 0 && ctx->topo_patience < (1u << 20)) ctx->topo_patience *= 2;    // fell back right after scaling down
		ctx->topo_calm = 0;
	} else if(b->stats.topology_fallbacks == 0 && (ctx->topo_scale > 1 || ctx->topo_pool_q8 > 8) && ++ctx->topo_calm >= ctx->topo_patience) {
		ctx->topo_scale = std::max(1u, ctx->topo_scale/2); ctx->topo_pool_q8 = std::max(8u, ctx->topo_pool_q8*3/4); ctx->topo_pool_cap = ctx->topo_pool_cap*3/4; ctx->topo_calm = 0;
	}
	ctx->in_flight = nullptr;
	return CRTHIP_OK;
}

namespace corto_hip {
int ctx_device(crthip_ctx *ctx) { return ctx->device; }
hipStream_t ctx_stream(crthip_ctx *ctx) { return ctx->stream; }
int ctx_fill_async(crthip_ctx *ctx, void *dst, size_t bytes, int value) {      // on the context's main stream, ordered before its next decode
	if(!bytes) return CRTHIP_OK;
	if(hipSetDevice(ctx->device) != hipSuccess) return fail(CRTHIP_E_DEVICE);
	hipLaunchKernelGGL(k_fill_block, dim3(2048), dim3(256), 0, ctx->stream, (uint8_t *)dst, (uint64_t)bytes, (uint32_t)(value & 255));
	return hipGetLastError() == hipSuccess ? CRTHIP_OK : fail(CRTHIP_E_DEVICE);
}
// SURVEY 8d's secondary region: a decode's outputs copied to (pinned) host memory behind its kernels, on the context's main stream; what
// crthip_batch_sync / crthip_batch_done wait for moves behind the copy
int ctx_copy_to_host_async(crthip_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes) {
	if(!bytes) return CRTHIP_OK;
	if(hipSetDevice(ctx->device) != hipSuccess) return fail(CRTHIP_E_DEVICE);
	if(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return fail(CRTHIP_E_DEVICE);
	if(hipEventRecord(ctx->ev_done, ctx->stream) != hipSuccess) return fail(CRTHIP_E_DEVICE);
	return CRTHIP_OK;
}
int ctx_quiesce(crthip_ctx *ctx) {
	if(harvest(ctx) != CRTHIP_OK) return fail(CRTHIP_E_DEVICE);
	ctx->last_decoded = nullptr;                         // the encoder stages reuse the scratch block
	return CRTHIP_OK;
}
}

extern "C" int crthip_device_count(void) {
	int n = 0;
	if(hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

extern "C" void crthip_ctx_destroy(crthip_ctx *c);
extern "C" void crthip_batch_destroy(crthip_batch *b);
extern "C" int crthip_ctx_create(int device, crthip_ctx **out) {
	if(!out) return fail(CRTHIP_E_ARGUMENT);
	int n = 0;
	if(hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return fail(CRTHIP_E_DEVICE);
	HIP_TRY(hipSetDevice(device));
	crthip_ctx *c = new crthip_ctx();
	c->device = device;
	if(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
	   hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess ||
	   hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
	   hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess ||
	   hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming) != hipSuccess) { delete c; return fail(CRTHIP_E_DEVICE); }
	c->dbg = debug_config_from_env();
	c->delta_wide = c->dbg.delta_wide != 0;
	// kernels that may ask for more than 64 KiB of dynamic LDS: raise their limit on this device, once per context
	// (function attributes are per device; doing it here keeps the launch paths free of shared state between host threads)
	if(hipFuncSetAttribute((const void *)k_topology_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TOPO_LDS_MAX) != hipSuccess ||
	   hipFuncSetAttribute((const void *)k_delta_lds16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DELTA16_LDS_MAX) != hipSuccess ||
	   hipFuncSetAttribute((const void *)k_normal_blob, hipFuncAttributeMaxDynamicSharedMemorySize, (int)NORMAL_LDS_MAX) != hipSuccess ||
	   hipFuncSetAttribute((const void *)k_enc_tun_parse, hipFuncAttributeMaxDynamicSharedMemorySize, (int)enc_parse_lds(ENC_TRIE_LDS_MAX)) != hipSuccess) {
		crthip_ctx_destroy(c); return fail(CRTHIP_E_DEVICE, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
	}
	*out = c;
	return CRTHIP_OK;
}

extern "C" void crthip_ctx_destroy(crthip_ctx *c) {
	if(!c) return;
	(void)hipSetDevice(c->device);
	(void)hipStreamSynchronize(c->stream);
	c->timer.release();
	if(c->host_batch) { crthip_batch *hb = c->host_batch; c->host_batch = nullptr; crthip_batch_destroy(hb); }
	c->scratch.release(); c->staging.release(); c->arena_pin.release(); c->status_host.release(); c->host_out.release(); c->host_pin.release();
	(void)hipStreamSynchronize(c->stream2);
	(void)hipEventDestroy(c->ev_fork); (void)hipEventDestroy(c->ev_join); (void)hipEventDestroy(c->ev_done);
	(void)hipStreamDestroy(c->stream2);
	(void)hipStreamDestroy(c->stream);
	delete c;
}

extern "C" int crthip_ctx_set_profiling(crthip_ctx *c, int enable) {
	if(!c) return fail(CRTHIP_E_ARGUMENT);
	c->profiling = enable != 0;
	return CRTHIP_OK;
}

extern "C" int crthip_ctx_set_packed_host_blobs(crthip_ctx *c, int on) {
	if(!c) return fail(CRTHIP_E_ARGUMENT);
	c->packed_host = on != 0;
	return CRTHIP_OK;
}

extern "C" int crthip_ctx_set_single_stream(crthip_ctx *c, int on) {
	if(!c) return fail(CRTHIP_E_ARGUMENT);
	c->single_stream = on != 0;
	// many batches in flight: kernels wait for LDS to come free, and a request of 29 KB finds room long before one of 78 KB does - the
	// normals kernel with its face normals in an L2-resident scratch array instead of LDS is slower alone (38 vs 34 us per C4 batch) and
	// the pipelined rate higher (round 2, 41 against 91 KB: +10 %)
	c->normal_fn_max = on ? 0u : NORMAL_FN_LDS_MAX;
	return CRTHIP_OK;
}

extern "C" int crthip_ctx_sync(crthip_ctx *c) {
	if(!c) return fail(CRTHIP_E_ARGUMENT);
	HIP_TRY(hipSetDevice(c->device));
	HIP_TRY(hipStreamSynchronize(c->stream));
	return CRTHIP_OK;
}

// ------------------------------------------------------------------------------------------------
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// (re)fill a batch object from a list of blobs: header parse + bounds-checked walk of each, arena layout, upload unless resident
static int batch_fill(crthip_ctx *ctx, crthip_batch *b, uint32_t nblobs, const uint8_t *const *blobs, const uint32_t *lens, const void *device_arena) {
	const double t_create = now_us();
	b->ctx = ctx;
	b->blobs.resize(nblobs);
	b->stats = crthip_batch_stats{};
	b->dirty = true; b->decoded = false;
	uint64_t off = 0;
	for(uint32_t i = 0; i < nblobs; i++) {
		BlobPlan &P = b->blobs[i];
		reset_layout(P.L);
		int err = walk_blob(blobs[i], lens[i], P.L);
		if(err) return fail(err, std::string(crthip_strerror(err)) + " (blob " + std::to_string(i) + ")");
		P.arena_off = off; P.len = lens[i];
		P.bind.assign(P.L.h.attrs.size(), Binding{});
		P.index = nullptr; P.index_u16 = 0; P.host_status = 0;
		P.dbg_clers = P.dbg_pred = ~0ull; P.dbg_nclers = 0; P.clers_in_arena = false;
		off += ((uint64_t)lens[i] + 15) & ~15ull;
		b->stats.total_nvert += P.L.h.nvert; b->stats.total_nface += P.L.h.nface;
		if(P.L.h.nface) { b->stats.clers_symbols += P.L.clers.size; b->stats.split_bytes += (uint64_t)P.L.split.nwords*4; }
	}
	b->arena_bytes = off;
	b->stats.arena_bytes = off;
	b->d_arena = nullptr;
	if(device_arena) b->d_arena = (const uint8_t *)device_arena;
	else if(off) {
		if(b->own_arena.reserve(off) != CRTHIP_OK) return fail(CRTHIP_E_NOMEM);
		// the blobs are gathered in a pinned image of the arena and go up in ONE copy on the context's stream, in front of the kernels that
		// read them - and nobody waits for it (round 3: the hipStreamSynchronize that stood here was 100 us of every from-host step, on the
		// host thread): the image is this context's own buffer, reused only after harvest() has seen the batch it fed complete
		if(harvest(ctx) != CRTHIP_OK) return fail(CRTHIP_E_DEVICE);
		bool in_place = ctx->packed_host && nblobs > 0;                        // the caller's buffer IS the arena's image (corto_hip.h)
		for(uint32_t i = 0; in_place && i < nblobs; i++) in_place = blobs[i] == blobs[0] + b->blobs[i].arena_off;
		if(in_place) {
			const uint64_t bytes = b->blobs[nblobs - 1].arena_off + lens[nblobs - 1];
			// the copy below reads the caller's buffer whenever the DMA engine gets to it: nothing here snapshots it or waits (corto_hip.h: the
			// caller keeps it alive and unchanged until the batch is synced).  A pageable buffer would still work (HIP stages it), a pinned one
			// is what the switch promises: on request, check
			if(ctx->dbg.check_pinned) {
				hipPointerAttribute_t pa;
				if(hipPointerGetAttributes(&pa, blobs[0]) != hipSuccess || pa.type != hipMemoryTypeHost) { (void)hipGetLastError(); return fail(CRTHIP_E_ARGUMENT, "packed host blobs: the buffer is not pinned host memory ($CORTO_HIP_CHECK_PINNED)"); }
			}
			if(hipMemcpyAsync(b->own_arena.p, blobs[0], bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return fail(CRTHIP_E_DEVICE);
		} else {
		if(ctx->arena_upload_pending) {                                      // (a batch that was created and not decoded yet: its upload has to be through before the image is reused)
			if(hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(CRTHIP_E_DEVICE);
			ctx->arena_upload_pending = false; ctx->done_covers_seq = ctx->upload_seq;
		}
		if(ctx->arena_pin.reserve(off) != CRTHIP_OK) return fail(CRTHIP_E_NOMEM);
		uint8_t *h = (uint8_t *)ctx->arena_pin.p;
		for(uint32_t i = 0; i < nblobs; i++) memcpy(h + b->blobs[i].arena_off, blobs[i], lens[i]);
		if(hipMemcpyAsync(b->own_arena.p, h, off, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return fail(CRTHIP_E_DEVICE);
		ctx->arena_upload_pending = true;
		ctx->upload_seq++;
		}
		b->d_arena = (const uint8_t *)b->own_arena.p;
	}
	b->status.assign(nblobs, 0);
	b->stats.host_create_us = (float)(now_us() - t_create);
	return CRTHIP_OK;
}

extern "C" int crthip_batch_create(crthip_ctx *ctx, uint32_t nblobs, const uint8_t *const *blobs, const uint32_t *lens,
                                   const void *device_arena, crthip_batch **out) {
	if(!ctx || !out || (nblobs && (!blobs || !lens))) return fail(CRTHIP_E_ARGUMENT);
	HIP_TRY(hipSetDevice(ctx->device));
	crthip_batch *b = new crthip_batch();
	const int err = batch_fill(ctx, b, nblobs, blobs, lens, device_arena);
	if(err) { b->own_arena.release(); delete b; return err; }
	*out = b;
	return CRTHIP_OK;
}

extern "C" int crthip_batch_reset(crthip_batch *b, uint32_t nblobs, const uint8_t *const *blobs, const uint32_t *lens, const void *device_arena) {
	if(!b || !b->ctx || (nblobs && (!blobs || !lens))) return fail(CRTHIP_E_ARGUMENT);
	crthip_ctx *ctx = b->ctx;
	HIP_TRY(hipSetDevice(ctx->device));
	if(ctx->in_flight == b) { if(harvest(ctx) != CRTHIP_OK) return fail(CRTHIP_E_DEVICE); }
	if(ctx->last_decoded == b) ctx->last_decoded = nullptr;
	return batch_fill(ctx, b, nblobs, blobs, lens, device_arena);       // on failure the batch is left empty-handed: reset or destroy it
}

extern "C" void crthip_batch_destroy(crthip_batch *b) {
	if(!b) return;
	if(b->ctx) {
		(void)hipSetDevice(b->ctx->device);
		if(b->ctx->in_flight == b) { (void)hipStreamSynchronize(b->ctx->stream); b->ctx->in_flight = nullptr; }
		if(b->ctx->last_decoded == b) b->ctx->last_decoded = nullptr;
		if(b->ctx->host_batch == b) b->ctx->host_batch = nullptr;
	}
	b->own_arena.release();
	delete b;
}

extern "C" uint32_t crthip_batch_size(const crthip_batch *b) { return b ? (uint32_t)b->blobs.size() : 0; }

extern "C" int crthip_batch_info(const crthip_batch *b, uint32_t i, crthip_blob_info *info) {
	if(!b || !info || i >= b->blobs.size()) return fail(CRTHIP_E_ARGUMENT);
	fill_info(b->blobs[i].L.h, info);
	return CRTHIP_OK;
}

static int check_binding(const AttrHeader &a, const crthip_attr_binding &bd) {
	if(!bd.buffer) return CRTHIP_OK;
	const uint32_t st = bd.stride;
	if(a.codec == CRTHIP_CODEC_NORMAL) {
		if(bd.format != CRTHIP_FMT_FLOAT && bd.format != CRTHIP_FMT_INT16) return CRTHIP_E_FORMAT;
		const uint32_t el = bd.format == CRTHIP_FMT_INT16 ? 2u : 4u;
		if(((uintptr_t)bd.buffer) % el) return CRTHIP_E_ARGUMENT;            // a float* / int16_t* (decoder.h:52-53) is aligned by its type
		if(st && (st < 3*el || st % el)) return CRTHIP_E_ARGUMENT;
		return CRTHIP_OK;
	}
	if(a.codec == CRTHIP_CODEC_COLOR) {
		if(bd.format != CRTHIP_FMT_UINT8) return CRTHIP_E_FORMAT;        // FLOAT colour output is broken upstream (color_attribute.cpp:96-110)
		uint32_t oc = bd.out_components ? bd.out_components : 4;
		if(a.N < 1 || a.N > 4 || oc > 4 || oc < a.N) return CRTHIP_E_FORMAT;
		if(st && st < oc) return CRTHIP_E_ARGUMENT;
		return CRTHIP_OK;
	}
	if(a.N < 1 || bd.format > CRTHIP_FMT_DOUBLE) return CRTHIP_E_FORMAT;
	// a packed buffer doubles as the int32 workspace and K-DELTA turns it into floats with dword / 16-byte accesses: a float* that is
	// not 4-byte aligned (never one a C++ caller's setPositions(float*) could pass) is refused, not decoded into integers
	if(((uintptr_t)bd.buffer) % (bd.format == CRTHIP_FMT_DOUBLE ? 8 : 4)) return CRTHIP_E_ARGUMENT;
	// the integer formats and DOUBLE (setAttribute(name, buffer, format): vertex_attribute.h:195-228) are upstream's in-place layouts: packed only
	if(bd.format != CRTHIP_FMT_FLOAT) return st ? CRTHIP_E_ARGUMENT : CRTHIP_OK;
	if(st && (st < 4*a.N || st % 4)) return CRTHIP_E_ARGUMENT;
	return CRTHIP_OK;
}

extern "C" int crthip_batch_bind(crthip_batch *b, uint32_t i, const crthip_attr_binding *attrs, void *index, uint32_t index_format) {
	if(!b || i >= b->blobs.size()) return fail(CRTHIP_E_ARGUMENT);
	BlobPlan &P = b->blobs[i];
	if(P.bind.size() && !attrs) return fail(CRTHIP_E_ARGUMENT);
	if(index && index_format != CRTHIP_FMT_UINT32 && index_format != CRTHIP_FMT_UINT16) return fail(CRTHIP_E_FORMAT);
	for(size_t k = 0; k < P.bind.size(); k++) {
		int err = check_binding(P.L.h.attrs[k], attrs[k]);
		if(err) return fail(err, std::string(crthip_strerror(err)) + " (attribute '" + P.L.h.attrs[k].name + "')");
	}
	for(size_t k = 0; k < P.bind.size(); k++) {
		P.bind[k].buffer = attrs[k].buffer; P.bind[k].format = attrs[k].format;
		P.bind[k].out_components = attrs[k].out_components ? attrs[k].out_components : 4;
		// a stride equal to the packed element size is the packed layout (the buffer then doubles as int32 workspace, as upstream's does)
		const AttrHeader &a = P.L.h.attrs[k];
		const uint32_t packed = a.codec == CRTHIP_CODEC_NORMAL ? (attrs[k].format == CRTHIP_FMT_INT16 ? 6u : 12u)
		                      : a.codec == CRTHIP_CODEC_COLOR ? P.bind[k].out_components : 4u*a.N;
		P.bind[k].stride = attrs[k].stride == packed ? 0u : attrs[k].stride;
	}
	P.index = index; P.index_u16 = index && index_format == CRTHIP_FMT_UINT16;
	b->dirty = true;
	return CRTHIP_OK;
}

extern "C" int crthip_batch_bind_all(crthip_batch *b, const crthip_attr_binding *attrs, void *const *index, const uint32_t *index_format) {
	if(!b) return fail(CRTHIP_E_ARGUMENT);
	size_t k = 0;
	for(uint32_t i = 0; i < b->blobs.size(); i++) {
		int err = crthip_batch_bind(b, i, attrs ? attrs + k : nullptr, index ? index[i] : nullptr, index_format ? index_format[i] : CRTHIP_FMT_UINT32);
		if(err) return err;
		k += b->blobs[i].bind.size();
	}
	return CRTHIP_OK;
}

// ------------------------------------------------------------------------------------------------
// planner: everything below turns the walked layouts + bindings into job arrays inside one scratch block
namespace {

struct Carver {                         // bump allocator over the scratch block (offsets only)
	uint64_t off = 0;
	uint64_t take(uint64_t bytes, uint64_t align = 256) { off = (off + align - 1) & ~(align - 1); uint64_t r = off; off += bytes; return r; }
};


static int32_t f2i_x86_host(float x) {
	if(!(x > -2147483904.0f && x < 2147483648.0f)) return (int32_t)0x80000000;
	return (int32_t)x;
}


} // namespace


// launch classes of K-DELTA: 2 - values + prediction graph fit LDS, one wave per attribute (k_delta_lds16, k_delta.hip): as int16 relative to
// vertex 0, or - `wide`: a context that met values beyond int16 - as int32; else the stretch walk over HBM (k_delta_mesh): 0 = large, 1 = small
// (meshes beyond LDS, attributes of more than four components)
static inline bool delta_hosts_a(const DeltaJob &d) { return !d.is_u8 && d.N == 3; }
static inline uint64_t delta_lds_need(const DeltaJob &d, bool wide) {          // alone in a workgroup; ~0: not eligible
	if(d.nvert > DELTA16_NVERT_MAX || d.N < 1 || d.N > 4) return ~0ull;
	return (uint64_t)delta_vbytes(d.nvert, d.N, d.is_u8 != 0, wide) + delta16_graph_lds(d.nvert, delta_hosts_a(d));
}
static inline int delta_class(const DeltaJob &d, bool wide) {
	if(delta_lds_need(d, wide) <= DELTA16_LDS_MAX) return 2;
	return d.nvert > DELTA_SMALL_NVERT ? 0 : 1;
}
static bool normal_fused(uint32_t nvert, uint32_t nface) { return nvert <= 32767 && (uint64_t)3*nface <= 65535 && normal_blob_lds(nvert, nface) <= NORMAL_LDS_MAX; }

struct Launch {
	crthip_ctx *ctx;
	hipStream_t cur = nullptr;
	void begin(const char *name, hipStream_t s = nullptr) {
		cur = s ? s : ctx->stream;
		if(!ctx->profiling) return;
		hipEvent_t e = ctx->timer.get();
		(void)hipEventRecord(e, cur);
		ctx->timer.recs.push_back({name, ctx->timer.used - 1, 0});
	}
	void end() {
		if(!ctx->profiling) return;
		hipEvent_t e = ctx->timer.get();
		(void)hipEventRecord(e, cur);
		ctx->timer.recs.back().e1 = ctx->timer.used - 1;
	}
};

static int build_and_launch_inner(crthip_batch *b);
// A decode call that fails half-way (a HIP error between two launches) may have kernels queued that write per-blob status into the
// context's pinned block and into its scratch; in_flight is not set on that path, so nothing downstream would wait for them before the
// next call clears or moves those blocks.  Drain the context's streams before the error is returned.
static int build_and_launch(crthip_batch *b) {
	const int err = build_and_launch_inner(b);
	if(err) {
		crthip_ctx *ctx = b->ctx;
		(void)hipStreamSynchronize(ctx->stream);
		if(ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
		(void)hipGetLastError();
		b->decoded = false;
	}
	return err;
}

// The planner of one decode call, stage by stage (round 4: this was one function of 660 lines).  carve() lays the batch's scratch out
// (pass 1: sizes and offsets only), jobs() writes the job descriptors of every stage with scratch-relative pseudo pointers (pass 2),
// group() sorts streams by dictionary and attributes into K-DELTA workgroups and places the job arrays, upload() reserves the blocks,
// rebases the pointers and stages the arrays, launch() enqueues the kernels in the order of crt::Decoder::decodeMesh / decodePointCloud
// (src/decoder.cpp:133-196), account() fills crthip_batch_stats.
namespace {
struct Planner {
	crthip_batch *b; crthip_ctx *ctx; Plan &pl; std::vector<BlobScratch> &bs;
	const uint32_t nblobs; const bool wide; const uint8_t *arena;            // wide: K-DELTA with 32-bit values in LDS (this context met values beyond int16)
	Carver cv;
	uint64_t unpack_state_words = 1, n_tun = 0, stat_tin = 0, stat_tout = 0, stat_tt = 0, stat_dicts = 0;
	uint32_t tun_chunks = 0, unpack_chunks = 0, cloud_chunks = 0;
	uint32_t clers_tun = 0, clers_chunks = 0, clers_fill = 0, clers_dict = 0;   // the CLERS streams come first in every stream / chunk / fill / dictionary array
	bool share_clers = false, share_attrs = false;
	int32_t *hs_base = nullptr;                                              // per-blob status words in pinned host memory
	uint8_t *base = nullptr, *stage = nullptr;                               // the scratch block; the host image of the job arrays

	Planner(crthip_batch *b_) : b(b_), ctx(b_->ctx), pl(b_->ctx->plan), bs(b_->ctx->plan_scratch), nblobs((uint32_t)b_->blobs.size()), wide(b_->ctx->delta_wide), arena(b_->d_arena) {}
	static uint8_t *SP(uint64_t off) { return (uint8_t *)(uintptr_t)off; }   // scratch-relative pseudo pointer
	int32_t *HS(uint64_t k) const { return (int32_t *)((uintptr_t)(hs_base + k) | (1ull << 63)); }   // real pointer (bit 63: R() leaves it alone)
	int carve(); int jobs(); void group(); int upload(); int launch(); void account();
};

int Planner::carve() {
	// ---- pass 1: sizes & offsets (device addresses are scratch_base + offset, resolved in pass 2) ----
	// We first carve all scratch, then reserve the block, then fill job structs with real pointers.
	// per-blob scratch offsets live in the context and are reset, not reallocated: a decode call used to spend a third of its host
	// time in malloc/free of these small vectors
	if(bs.size() < nblobs) bs.resize(nblobs);
	for(uint32_t i = 0; i < nblobs; i++) bs[i].reset();

	// zeroed region: status, predictions (vertices the automaton never reaches keep (0,0,0)), and the
	// counters of the ESTIMATED/BORDER normal pipeline
	uint64_t est_v = 0, est_f = 0;
	for(uint32_t i = 0; i < nblobs; i++) {
		const BlobPlan &P = b->blobs[i];
		const BlobLayout &L = P.L;
		if(L.h.nface == 0) continue;
		for(size_t k = 0; k < L.attrs.size(); k++)
			if(L.h.attrs[k].codec == CRTHIP_CODEC_NORMAL && P.bind[k].buffer && L.attrs[k].normal_prediction != 0 && !normal_fused(L.h.nvert, L.h.nface)) { est_v += L.h.nvert; est_f += L.h.nface; }
	}
	// Per-blob status and flags live in the context's PINNED HOST block: the kernels write there directly (only a failing or
	// redone blob does), the host zeroes it before the launch and reads it after the sync - no memset kernel in front of a step
	// and no copy kernel behind it (each stretched to 50-100 us with eight batches in flight).  Prediction triples are not cleared
	// either: the automaton writes every vertex it makes and clears the ones it never reached itself (k_mesh.hip).
	for(uint32_t i = 0; i < nblobs; i++) {
		const BlobLayout &L = b->blobs[i].L;
		if(L.h.nface > 0) bs[i].pred = cv.take((uint64_t)L.h.nvert*12, 16);
	}
	pl.zero_begin = cv.take(0);                                          // zeroed by a memset, present only for big meshes: the counters of the
	pl.est_nvert = (uint32_t)est_v; pl.est_nface = (uint32_t)est_f;      // unfused normal pipeline and the fired flags of k_delta_mesh
	if(est_v) {
		pl.cnt_off = cv.take(est_v*4 + 16); pl.cursor_off = cv.take(est_v*4 + 16); pl.bnd_off = cv.take(est_v*4 + 16);
	}
	for(uint32_t i = 0; i < nblobs; i++) {                           // "fired" flags of delta jobs too large for LDS
		const BlobPlan &P = b->blobs[i];
		const BlobLayout &L = P.L;
		bs[i].set_attrs(L.attrs.size());
		if(L.h.nface == 0) continue;
		for(size_t k = 0; k < L.attrs.size(); k++) {
			if(!P.bind[k].buffer) continue;
			const AttrHeader &a = L.h.attrs[k];
			DeltaJob probe{};
			probe.nvert = L.h.nvert; probe.N = a.codec == CRTHIP_CODEC_NORMAL ? 2u : a.N; probe.is_u8 = a.codec == CRTHIP_CODEC_COLOR;
			if(delta_class(probe, wide) <= 1)                                     // k_delta_mesh: fired flags (zeroed) + the list of stretch starts behind them
				bs[i].attr[k].fired = cv.take((((uint64_t)L.h.nvert + 15) & ~15ull) + 4ull*L.h.nvert + 16, 16);
		}
	}
	pl.zero_end = cv.take(0);
	unpack_state_words = 1;                                               // look-back state words of the bit-unpack chunks (k_unpack_extract): one per
	for(uint32_t i = 0; i < nblobs; i++) {                              // 1 024 logs of every bound stream, + a spare; uploaded as zeros with the jobs
		const BlobPlan &P = b->blobs[i];
		for(size_t k = 0; k < P.L.attrs.size(); k++) if(P.bind[k].buffer) for(const StreamRef &lg : P.L.attrs[k].logs) unpack_state_words += ((uint64_t)lg.size + CHUNK - 1)/CHUNK;
	}

	auto need_stream = [&](const StreamRef &s, uint64_t &sym_off) {
		sym_off = ~0ull;
		if(s.mode == STREAM_TUNSTALL || s.mode == STREAM_FILL) sym_off = cv.take((uint64_t)s.size + 16, 16);
		if(s.mode == STREAM_TUNSTALL) { n_tun++; stat_tin += s.csize; stat_tout += s.size; stat_tt += 9 + 2*(uint64_t)s.nsym; }
	};

	for(uint32_t i = 0; i < nblobs; i++) {
		BlobPlan &P = b->blobs[i];
		const BlobLayout &L = P.L;
		BlobScratch &S = bs[i];
		P.host_status = 0;
		S.set_attrs(L.attrs.size());
		const bool mesh = L.h.nface > 0;
		if(mesh) {
			need_stream(L.clers, S.clers);
			uint32_t maxg = 0, prev = 0;
			for(uint32_t ge : L.group_end) { if(ge > prev) maxg = std::max(maxg, ge - prev); prev = std::max(prev, ge); }
			uint64_t cap = std::min<uint64_t>(L.max_front, (uint64_t)3*maxg);
			S.front_cap = (uint32_t)std::min<uint64_t>(cap, 0xFFFFFFF0u);
			S.front_a = cv.take((uint64_t)(S.front_cap + 4)*16);
			S.front_b = cv.take((uint64_t)(S.front_cap + 4)*8);
			S.order = cv.take((uint64_t)(S.front_cap + 4)*4);
			S.delayed = cv.take((uint64_t)(S.front_cap + 4)*4);
			if(!P.index) S.faces = cv.take((uint64_t)L.h.nface*12);
		}
		for(size_t k = 0; k < L.attrs.size(); k++) {
			const AttrHeader &a = L.h.attrs[k];
			const Binding &bd = P.bind[k];
			if(!bd.buffer) continue;                           // unbound: streams skipped (cstream.h:302,331)
			AttrScratch &A = S.attr[k];
			A.sym.resize(L.attrs[k].logs.size());
			for(size_t j = 0; j < A.sym.size(); j++) need_stream(L.attrs[k].logs[j], A.sym[j]);
			if(a.codec == CRTHIP_CODEC_COLOR) A.color = cv.take((uint64_t)L.h.nvert*a.N + 16, 16);
			if(a.codec != CRTHIP_CODEC_COLOR && a.codec != CRTHIP_CODEC_NORMAL && (bd.stride || bd.format == CRTHIP_FMT_DOUBLE)) A.vals = cv.take((uint64_t)L.h.nvert*a.N*4 + 16, 16);
			if(a.codec == CRTHIP_CODEC_NORMAL) {
				A.diffs = cv.take((uint64_t)L.h.nvert*8 + 16, 16);
				if(mesh && L.attrs[k].normal_prediction != 0 && normal_fused(L.h.nvert, L.h.nface) && normal_blob_lds_fn(L.h.nvert, L.h.nface) > ctx->normal_fn_max)
					A.facen = cv.take((uint64_t)L.h.nface*12 + 16, 16);
			}
		}
	}
	if(est_v) {
		pl.start_off = cv.take(est_v*4 + 16); pl.flag_off = cv.take(est_v*4 + 16); pl.slot_off = cv.take(est_v*4 + 16);
		pl.adj_off = cv.take(est_f*12 + 16); pl.facen_off = cv.take(est_f*12 + 16);
		pl.nscan_partial_off = cv.take(((est_v + CHUNK - 1)/CHUNK + 1)*8);
	}
	pl.tables_off = cv.take(n_tun*sizeof(TunTable));

	return CRTHIP_OK;
}

int Planner::jobs() {
	// ---- pass 2: job structs with offsets stored in pointer fields (rebased after the block is reserved) ----
	// To keep one pass, pointers are built as (uint8_t*)offset and fixed up by adding the scratch base.
	// four words a blob: status | automaton flags (bit 0: redone on the HBM front) | K-DELTA: {an attribute's values left int16, an attribute took the walk}
	if((size_t)nblobs*16 + 16 > ctx->status_host.cap && harvest(ctx) != CRTHIP_OK) return fail(CRTHIP_E_DEVICE);   // the block is about to move: the batch in flight writes to it
	if(ctx->status_host.reserve((size_t)nblobs*16 + 16) != CRTHIP_OK) return fail(CRTHIP_E_NOMEM);
	hs_base = (int32_t *)ctx->status_host.p;

	// the dictionary (TunTable slot) of a stream: a new one, or the one an earlier stream of this launch group with the same table got
	if(ctx->dict_slots.size() != 8192) ctx->dict_slots.assign(8192, 0u);
	for(uint32_t u : ctx->dict_used) ctx->dict_slots[u] = 0;
	ctx->dict_used.clear(); ctx->dict_keys.clear();
	std::vector<uint32_t> &dict_ids = ctx->dict_ids; dict_ids.clear();
	uint32_t dict_group0 = 0;                                              // first dictionary of the current group (CLERS streams / attribute streams)
	auto dict_of = [&](const StreamRef &s, const TunStream &t) -> uint32_t {
		const uint32_t fresh = (uint32_t)pl.tun_dict.v.size();
		auto make = [&]() { TunStream d = t; d.table = fresh; d.dict = fresh; d.nchunks = 1; pl.tun_dict.v.push_back(d); return fresh; };
		if(s.nsym > 16 || fresh - dict_group0 >= 4096 || ctx->dbg.tun_share == 0 || ctx->dbg.tun_share == 2) return make();   // (0 / 2: one dictionary per stream, whatever repeats)
		uint64_t h = 0x9E3779B97F4A7C15ull ^ s.nsym;
		for(uint32_t k = 0; k < 2*s.nsym; k += 8) { uint64_t w; memcpy(&w, s.probs16 + k, 8); h = (h ^ w)*0xFF51AFD7ED558CCDull; h ^= h >> 32; }
		for(uint32_t pos = (uint32_t)h & 8191u;; pos = (pos + 1) & 8191u) {
			const uint32_t e = ctx->dict_slots[pos];
			if(!e) {
				ctx->dict_slots[pos] = (uint32_t)ctx->dict_keys.size() + 1; ctx->dict_used.push_back(pos);
				crthip_ctx::DictKey key; key.n = (uint8_t)s.nsym; memcpy(key.bytes, s.probs16, 32);
				ctx->dict_keys.push_back(key);
				const uint32_t d = make();
				dict_ids.push_back(d);
				return d;
			}
			const crthip_ctx::DictKey &key = ctx->dict_keys[e - 1];
			if(key.n == s.nsym && memcmp(key.bytes, s.probs16, 2*s.nsym) == 0) return dict_ids[e - 1];
		}
	};
	auto add_stream = [&](const StreamRef &s, uint64_t sym_off, uint64_t blob_off) -> const uint8_t * {
		// returns the (pseudo or real) device pointer where the decoded symbols will be; real pointers have bit 63 set
		if(s.mode == STREAM_RAW) return (const uint8_t *)((uintptr_t)(arena + blob_off + s.payload_off) | (1ull << 63));
		if(s.mode == STREAM_EMPTY) return SP(0);
		if(s.mode == STREAM_FILL) { pl.fill.v.push_back(FillJob{SP(sym_off), s.size, s.fill}); return SP(sym_off); }
		TunStream t{};
		t.src = arena + blob_off + s.payload_off; t.dst = SP(sym_off); t.probs = arena + blob_off + s.probs_off;
		t.csize = s.csize; t.size = s.size; t.nsym = s.nsym; t.table = (uint32_t)pl.tun.v.size();
		t.chunk0 = tun_chunks; tun_pick_geometry(t);
		if(t.nchunks > 1) pl.tun_multi_chunk = true;
		pl.tun_max_nchunks = std::max(pl.tun_max_nchunks, t.nchunks);
		for(uint32_t c = 0; c < t.nchunks; c++) pl.tun_chunk_stream.v.push_back((uint32_t)pl.tun.v.size());
		tun_chunks += t.nchunks;
		t.dict = dict_of(s, t);
		pl.tun.v.push_back(t);
		return SP(sym_off);
	};

	// the CLERS streams come first in every stream/chunk/fill array: they are what the topology kernel waits for, the
	// attribute streams are decoded on the second HIP stream while topology runs
	std::vector<const uint8_t *> &clers_ptrs = ctx->plan_clers;
	clers_ptrs.assign(nblobs, nullptr);
	for(uint32_t i = 0; i < nblobs; i++) {
		const BlobLayout &L = b->blobs[i].L;
		if(L.h.nface > 0) clers_ptrs[i] = add_stream(L.clers, bs[i].clers, b->blobs[i].arena_off);
	}
	clers_tun = (uint32_t)pl.tun.v.size(); clers_chunks = tun_chunks; clers_fill = (uint32_t)pl.fill.v.size();
	clers_dict = (uint32_t)pl.tun_dict.v.size();
	// the attribute streams are a launch of their own: their dictionaries are not shared with the CLERS streams' (different HIP streams)
	for(uint32_t u : ctx->dict_used) ctx->dict_slots[u] = 0;
	ctx->dict_used.clear(); ctx->dict_keys.clear(); dict_ids.clear();
	dict_group0 = clers_dict;

	uint32_t est_vbase = 0, est_fbase = 0;
	for(uint32_t i = 0; i < nblobs; i++) {
		BlobPlan &P = b->blobs[i];
		const BlobLayout &L = P.L;
		BlobScratch &S = bs[i];
		const bool mesh = L.h.nface > 0;
		const uint32_t nvert = L.h.nvert, nface = L.h.nface;
		const uint64_t bo = P.arena_off;
		const uint8_t *clers_ptr = nullptr;
		if(mesh) {
			clers_ptr = clers_ptrs[i];
			P.clers_in_arena = L.clers.mode == STREAM_RAW;
			P.dbg_clers = L.clers.mode == STREAM_RAW ? bo + L.clers.payload_off : S.clers;
			P.dbg_nclers = L.clers.size; P.dbg_pred = S.pred;
			TopoJob t{};
			t.clers = clers_ptr;
			t.split_words = (const uint32_t *)(arena + bo + L.split.words_off);
			t.group_end = (const uint32_t *)SP(pl.aux_u32.v.size()*4);   // index into aux, rebased later
			for(uint32_t ge : L.group_end) pl.aux_u32.v.push_back(ge);
			t.faces = P.index ? P.index : (void *)SP(S.faces);
			t.pred = (uint32_t *)SP(S.pred);
			t.front_a = (uint4 *)SP(S.front_a); t.front_b = (uint2 *)SP(S.front_b);
			t.order = (uint32_t *)SP(S.order); t.delayed = (uint32_t *)SP(S.delayed);
			t.status = HS(i);
			t.flags = HS(nblobs + i);
			t.nclers = L.clers.size; t.split_nwords = L.split.nwords; t.ngroups = (uint32_t)L.group_end.size();
			t.nvert = nvert; t.nface = nface; t.front_cap = S.front_cap; t.faces_u16 = P.index ? P.index_u16 : 0;
			t.pad = P.index ? 1u : 0u;                                   // pad = 1: faces is a real pointer
			{
				// every mesh takes the LDS path; a lone big mesh may use most of a CU's LDS, a batch keeps its blobs small
				uint32_t ring, pool, symwin;
				uint32_t scale = ctx->topo_scale, pool_q8 = ctx->topo_pool_q8, need;
				for(;;) {                                                                // as much of what the context has learnt as fits a CU
					topo_lds_geometry(nface, L.clers.size, 4096, scale, pool_q8, nblobs >= 32 ? 4u : 8u, topo_boundary_estimate(nvert, nface), ring, pool, symwin, ctx->topo_pool_cap);
					need = topo_lds_bytes(ring, pool, pool, symwin);                     // every delayed edge is a pool record: same capacity
#ifdef CORTO_TOPO_STAMPS
					if(need <= 32768 && L.clers.size < 8190) need = 65536;              // (the dispatch trace: k_mesh.hip TOPO_ASM_STAMP)
#endif
					if(need <= TOPO_LDS_MAX || (scale == 1 && pool_q8 == 8)) break;
					if(pool_q8 > 8 && (pool > ring || scale == 1)) pool_q8 = std::max(8u, pool_q8/2); else scale >>= 1;
				}
				if(need <= TOPO_LDS_MAX) {
					t.lds_ring = ring; t.lds_pool = pool; t.lds_delayed_cap = pool; t.lds_symwin = symwin;
					pl.topo_lds_ids.v.push_back((uint32_t)pl.topo.v.size()); pl.topo_need.push_back(need);     // (split into two launches below)
				}
				else pl.topo_glob_ids.v.push_back((uint32_t)pl.topo.v.size());
			}
			pl.topo.v.push_back(t);
		}
		// position attribute (needed by ESTIMATED/BORDER normals)
		int pos_k = -1;
		for(size_t k = 0; k < L.attrs.size(); k++) if(L.h.attrs[k].name == "position") pos_k = (int)k;
		// who turns the integer positions into floats: estimated normals read them as integers after K-DELTA, so the fused normal
		// kernel does it as their last reader (pos_by_normal), the separate normal kernels leave it to k_dequant behind them, and
		// without such normals K-DELTA does it on the way out of LDS like for every other attribute
		bool pos_ints_needed = false, pos_by_normal = false;
		{
			uint32_t readers = 0;
			for(size_t k = 0; k < L.attrs.size(); k++)
				if(mesh && L.h.attrs[k].codec == CRTHIP_CODEC_NORMAL && P.bind[k].buffer && (L.attrs[k].normal_prediction == 1 || L.attrs[k].normal_prediction == 2)) readers++;
			pos_ints_needed = readers > 0;
			pos_by_normal = readers == 1 && normal_fused(nvert, nface) && pos_k >= 0 && P.bind[pos_k].format == CRTHIP_FMT_FLOAT;
		}

		for(size_t k = 0; k < L.attrs.size(); k++) {
			const AttrHeader &a = L.h.attrs[k];
			const AttrStreams &as = L.attrs[k];
			const Binding &bd = P.bind[k];
			if(!bd.buffer) continue;
			AttrScratch &A = S.attr[k];
			const uint32_t *words = (const uint32_t *)(arena + bo + as.bits.words_off);
			const uint32_t chain0 = unpack_chunks;
			uint64_t attr_logs = 0;
			for(const StreamRef &lg : as.logs) attr_logs += lg.size;
			const bool by_wave = attr_logs <= UNPACK_WAVE_MAX_LOGS && as.bits.nwords < (1u << 26) && !ctx->dbg.unpack_chunked;   // one wave per stream, no look-back; its bit cursors are 32-bit (k_stream.hip)
			const uint32_t attr_first = (uint32_t)pl.unpack.v.size();
			auto push_unpack = [&](const StreamRef &s, const uint8_t *logs, void *out, bool out_real, uint8_t mode, uint16_t fields, uint16_t stride, uint16_t comp, uint8_t u8) {
				if(s.size == 0) return;
				UnpackJob u{};
				u.logs = logs; u.words = words; u.out = out; u.count = s.size; u.nwords = as.bits.nwords; u.out_limit = nvert;
				u.chunk0 = unpack_chunks; u.chain_chunk0 = chain0; u.fields = fields; u.stride = stride; u.comp = comp; u.mode = mode;
				u.out_u8 = (uint8_t)(u8 | (out_real ? 0x80 : 0));           // bit7: out is a real pointer (cleared at fixup)
				if(by_wave) { u.chain_chunk0 = attr_first; pl.unpack_wave_ids.v.push_back((uint32_t)pl.unpack.v.size()); }
				else {
					const uint32_t nc = (s.size + CHUNK - 1)/CHUNK;
					for(uint32_t c = 0; c < nc; c++) pl.unpack_chunk_job.v.push_back((uint32_t)pl.unpack.v.size());
					unpack_chunks += nc;
				}
				pl.unpack.v.push_back(u);
			};
			std::vector<const uint8_t *> &logs = ctx->plan_logs;
			logs.assign(as.logs.size(), nullptr);
			for(size_t j = 0; j < as.logs.size(); j++) logs[j] = add_stream(as.logs[j], A.sym[j], bo);

			void *values = nullptr; bool values_real = false; uint8_t is_u8 = 0; uint32_t N = a.N; bool para = false; bool do_delta = true;
			if(a.codec == CRTHIP_CODEC_NORMAL) {
				push_unpack(as.logs[0], logs[0], SP(A.diffs), false, 0, 2, 2, 0, 0);
				// a (malformed) stream with fewer diffs than vertices: upstream's vector is zero-filled behind them (normal_attribute.cpp:180-184)
				// (only DIFF reads all nvert entries; the other predictions stop at ndiffs)
				if(as.normal_prediction == 0 && as.logs[0].size < nvert) pl.fill.v.push_back(FillJob{SP(A.diffs + (uint64_t)as.logs[0].size*8), (nvert - as.logs[0].size)*8u, 0u});
				values = SP(A.diffs); N = 2; para = false;
				do_delta = as.normal_prediction == 0;                     // DIFF only (normal_attribute.cpp:190-191)
			} else if(a.codec == CRTHIP_CODEC_COLOR) {
				for(uint32_t c = 0; c < a.N; c++) push_unpack(as.logs[c], logs[c], SP(A.color), false, 1, 1, (uint16_t)a.N, (uint16_t)c, 1);
				values = SP(A.color); is_u8 = 1; para = (a.strategy & CRTHIP_PARALLEL) != 0;
			} else {
				// packed output: the caller's buffer is the int32 workspace (like upstream, vertex_attribute.h:190-193); with a stride, or as
				// DOUBLE (eight bytes a value: upstream widens in place, front to back): scratch
				const bool in_scratch = bd.stride || bd.format == CRTHIP_FMT_DOUBLE;
				void *work = in_scratch ? (void *)SP(A.vals) : bd.buffer;
				const bool work_real = !in_scratch;
				if(a.strategy & CRTHIP_CORRELATED) push_unpack(as.logs[0], logs[0], work, work_real, 0, (uint16_t)a.N, (uint16_t)a.N, 0, 0);
				else for(uint32_t c = 0; c < a.N; c++) push_unpack(as.logs[c], logs[c], work, work_real, 1, 1, (uint16_t)a.N, (uint16_t)c, 0);
				values = work; values_real = work_real; para = (a.strategy & CRTHIP_PARALLEL) != 0;
			}
			bool dequantised = false;                                // by K-DELTA or by the fused normal kernel: no k_dequant job
			if(do_delta && nvert > 1) {
				if(mesh) {
					DeltaJob d{};
					d.values = values; d.pred = (const uint32_t *)SP(S.pred); d.nvert = nvert; d.N = N;
					d.parallelogram = para; d.is_u8 = is_u8; d.pad[0] = values_real; d.pad[1] = wide; d.pad2[0] = ctx->dbg.delta_rounds ? 1u : 0u;                   // pad[1]: 32-bit records in LDS (k_delta_lds16)
					d.fired = A.fired != ~0ull ? SP(A.fired) : nullptr;
					d.flags = HS(2ull*nblobs + 2ull*i);
					if(a.codec != CRTHIP_CODEC_NORMAL && delta_class(d, wide) >= 2) {
						if(a.codec == CRTHIP_CODEC_COLOR) {
							d.deq = 2; d.out = bd.buffer; d.out_components = bd.out_components; d.out_stride = bd.stride;
							for(int c = 0; c < 4; c++) d.qc[c] = as.qc[c];
							dequantised = true;
						} else if(!bd.stride && bd.format == CRTHIP_FMT_FLOAT && !((int)k == pos_k && pos_ints_needed)) { d.deq = 1; d.q = a.q; dequantised = true; }
					}
					pl.delta.v.push_back(d);
				} else {
					CloudJob c{};
					c.values = values; c.nvert = nvert; c.N = N; c.chunk0 = cloud_chunks; c.is_u8 = is_u8; c.pad[0] = values_real;
					const uint32_t nc = N*((nvert + CHUNK - 1)/CHUNK);
					for(uint32_t q = 0; q < nc; q++) pl.cloud_chunk_job.v.push_back((uint32_t)pl.cloud.v.size());
					cloud_chunks += nc;
					pl.cloud.v.push_back(c);
				}
			}
			if(a.codec == CRTHIP_CODEC_NORMAL) {
				const uint32_t pr = as.normal_prediction;
				if(pr == 0 || (mesh && (pr == 1 || pr == 2))) {       // clouds: postDelta never runs (decoder.cpp:142-143)
					NormalJob n{};
					n.diffs = (int32_t *)SP(A.diffs); n.out = bd.buffer; n.nvert = nvert; n.nface = nface;
					n.out_stride = bd.stride ? bd.stride : (bd.format == CRTHIP_FMT_INT16 ? 6u : 12u);
					n.ndiffs = std::min(as.logs[0].size, nvert); n.unit = f2i_x86_host(a.q);
					n.prediction = (uint8_t)pr; n.out_i16 = bd.format == CRTHIP_FMT_INT16;
					n.status = HS(i);
					if(pr != 0) {
						const bool pos_ok = pos_k >= 0 && L.h.attrs[pos_k].codec == CRTHIP_CODEC_GENERIC && L.h.attrs[pos_k].N == 3 && P.bind[pos_k].buffer;
						if(!pos_ok) { P.host_status = CRTHIP_E_NORMAL_NEEDS_POSITION; continue; }
						const bool pos_scratch = P.bind[pos_k].stride != 0 || P.bind[pos_k].format == CRTHIP_FMT_DOUBLE;   // the integer positions: in the caller's packed buffer, or in scratch
						n.position = pos_scratch ? (const int32_t *)SP(S.attr[pos_k].vals) : (const int32_t *)P.bind[pos_k].buffer;
						n.faces = P.index ? P.index : (void *)SP(S.faces);
						n.faces_u16 = (uint8_t)((P.index ? P.index_u16 : 0) | (P.index ? 0x80 : 0) | (pos_scratch ? 0x40 : 0));   // bit7: faces is a real pointer, bit6: position is a scratch offset (both cleared at fixup)
						if(normal_fused(nvert, nface)) {
							n.fused = 1;
							n.fn_scratch = A.facen != ~0ull ? (float *)SP(A.facen) : nullptr;
							if(pos_by_normal) { n.pos_out = P.bind[pos_k].buffer; n.pos_stride = P.bind[pos_k].stride ? P.bind[pos_k].stride : 12u; n.pos_q = L.h.attrs[pos_k].q; }
							pl.normal_fused_ids.v.push_back((uint32_t)pl.normal.v.size());
							pl.normal_fused_lds = std::max(pl.normal_fused_lds, normal_blob_lds_fn(nvert, nface) <= ctx->normal_fn_max ? normal_blob_lds_fn(nvert, nface) : normal_blob_lds(nvert, nface));
						} else {
							n.vbase = est_vbase; n.fbase = est_fbase; est_vbase += nvert; est_fbase += nface;
							pl.any_est_normal = true;
						}
					} else pl.any_diff_normal = true;
					pl.normal.v.push_back(n);
				}
			} else if(!dequantised && !((int)k == pos_k && pos_by_normal)) {
				DequantJob q{};
				q.buffer = bd.buffer; q.q = a.q; q.nvert = nvert; q.N = a.N; q.out_components = bd.out_components;
				for(int c = 0; c < 4; c++) q.qc[c] = as.qc[c];
				q.block0 = (uint32_t)pl.dequant_block_job.v.size();
				q.is_color = a.codec == CRTHIP_CODEC_COLOR;
				q.format = q.is_color ? (uint8_t)CRTHIP_FMT_FLOAT : (uint8_t)bd.format;
				q.stride = bd.stride;
				if(q.is_color) q.src = SP(A.color);
				else if(bd.stride || bd.format == CRTHIP_FMT_DOUBLE) q.src = SP(A.vals);
				const uint64_t elems = q.is_color ? nvert : (uint64_t)nvert*a.N;
				const uint32_t nb = (uint32_t)((elems + CHUNK - 1)/CHUNK);
				for(uint32_t c = 0; c < nb; c++) pl.dequant_block_job.v.push_back((uint32_t)pl.dequant.v.size());
				pl.dequant.v.push_back(q);
			}
		}
	}
	return CRTHIP_OK;
}

void Planner::group() {
	// streams of a launch that share dictionaries: sorted by dictionary (counting sort), cut into groups of one dictionary each
	const uint32_t ntun_all = (uint32_t)pl.tun.v.size(), ndict_all = (uint32_t)pl.tun_dict.v.size();
	// dictionaries by one kernel (K-TAB, 6 KB of LDS per wave), decodes by another (10 KB for a few us), instead of both in one wave per stream
	// (16 KB for ~37 us): a batch of many streams is bound by LDS.time, so the split pays even when NO two streams share a table
	auto shares = [&](uint32_t nstreams, uint32_t ndicts) { (void)ndicts; return ctx->dbg.tun_share == 0 ? false : ctx->dbg.tun_share == 1 ? true : nstreams >= 64; };
	share_clers = !pl.tun_multi_chunk && shares(clers_tun, clers_dict); share_attrs = !pl.tun_multi_chunk && shares(ntun_all - clers_tun, ndict_all - clers_dict);
	{
		std::vector<uint32_t> &cnt = ctx->dict_count;
		auto group_range = [&](uint32_t t0, uint32_t t1, uint32_t d0, uint32_t d1) {
			if(t1 <= t0) return;
			cnt.assign((size_t)(d1 - d0) + 1, 0u);
			for(uint32_t t = t0; t < t1; t++) cnt[pl.tun.v[t].dict - d0 + 1]++;
			for(uint32_t d = 1; d <= d1 - d0; d++) cnt[d] += cnt[d - 1];
			const uint32_t base = (uint32_t)pl.tun_group_ids.v.size();
			pl.tun_group_ids.v.resize((size_t)base + (t1 - t0));
			for(uint32_t d = 0; d < d1 - d0; d++)                                 // (cnt[d] .. cnt[d + 1]: the dictionary's slots; groups before the fill moves the cursors)
				for(uint32_t k = cnt[d]; k < cnt[d + 1]; k += TUN_GROUP_MAX) pl.tun_groups.v.push_back(TunGroup{base + k, std::min(TUN_GROUP_MAX, cnt[d + 1] - k)});
			for(uint32_t t = t0; t < t1; t++) pl.tun_group_ids.v[base + cnt[pl.tun.v[t].dict - d0]++] = t;
		};
		if(share_clers) group_range(0, clers_tun, 0, clers_dict);
		pl.clers_groups = (uint32_t)pl.tun_groups.v.size();
		if(share_attrs) group_range(clers_tun, ntun_all, clers_dict, ndict_all);
	}

	// block maps of the normal jobs (per vertex / per face, 256 per block)
	for(uint32_t j = 0; j < pl.normal.v.size(); j++) {
		const NormalJob &n = pl.normal.v[j];
		pl.nv_block_first.v.push_back((uint32_t)pl.nv_block_job.v.size());
		for(uint32_t c = 0; c < (n.nvert + 255)/256; c++) pl.nv_block_job.v.push_back(j);
		pl.nf_block_first.v.push_back((uint32_t)pl.nf_block_job.v.size());
		if(n.prediction != 0 && !n.fused) for(uint32_t c = 0; c < (n.nface + 255)/256; c++) pl.nf_block_job.v.push_back(j);
	}

	pl.tun_partial_off = cv.take(((uint64_t)tun_chunks*4 + 4)*8);
	pl.cloud_partial_off = cv.take(((uint64_t)cloud_chunks + 1)*8);

	// job arrays region
	pl.jobs_begin = cv.take(0);
	pl.unpack_partial_off = cv.take(unpack_state_words*8, 16);           // (first thing in the uploaded block: zeros)
	auto place = [&](auto &arr) { arr.dev_off = cv.take(arr.v.size()*sizeof(arr.v[0]) + 16, 16); };
	// the LDS automata go up in ONE launch whose LDS request is the largest of theirs - unless some ask for much more than the others (a 66K-triangle
	// mesh among 4K-triangle blobs): those get a launch of their own, so that a big mesh does not cost the small ones their occupancy.  "Much more": beyond
	// 32 KB AND beyond twice the smallest request (round 5: a batch of Delaunay discs asks for 20-40 KB a blob, and cut at 32 KB it became two launches
	// one after the other, each as long as its slowest blob - 1.07 ms instead of 0.57)
	if(!pl.topo_lds_ids.v.empty()) {
		uint32_t lo = 0xFFFFFFFFu;
		for(uint32_t nd : pl.topo_need) lo = std::min(lo, nd);
		const uint32_t cut = std::max(32u*1024u, 2u*lo);
		std::vector<uint32_t> small_ids;
		for(size_t k = 0; k < pl.topo_lds_ids.v.size(); k++) {
			const uint32_t nd = pl.topo_need[k], id = pl.topo_lds_ids.v[k];
			if(nd <= cut) { small_ids.push_back(id); pl.topo_lds = std::max(pl.topo_lds, nd); }
			else { pl.topo_big_ids.v.push_back(id); pl.topo_big_lds = std::max(pl.topo_big_lds, nd); }
		}
		pl.topo_lds_ids.v.swap(small_ids);
	}
	place(pl.tun); place(pl.tun_dict); place(pl.tun_chunk_stream); place(pl.tun_group_ids); place(pl.tun_groups); place(pl.fill); place(pl.topo); place(pl.aux_u32); place(pl.topo_lds_ids); place(pl.topo_big_ids); place(pl.topo_glob_ids); place(pl.unpack); place(pl.unpack_chunk_job); place(pl.unpack_wave_ids);
	// large attributes first: they are launched with four times the threads of the small ones (k_delta_mesh)
	std::stable_partition(pl.delta.v.begin(), pl.delta.v.end(), [wide_ = wide](const DeltaJob &d) { return delta_class(d, wide_) == 0; });
	std::stable_partition(pl.delta.v.begin(), pl.delta.v.end(), [wide_ = wide](const DeltaJob &d) { return delta_class(d, wide_) <= 1; });
	{	// attributes of one blob that fit LDS together share a workgroup and the prediction graph: consecutive jobs of class 2 with the same
		// prediction array, up to DELTA_GROUP_MAX
		size_t j = 0;
		while(j < pl.delta.v.size() && delta_class(pl.delta.v[j], wide) < 2) j++;
		while(j < pl.delta.v.size()) {
			const DeltaJob &d0 = pl.delta.v[j];
			DeltaGroup g{(uint32_t)j, 1};
			uint64_t vals = delta_vbytes(d0.nvert, d0.N, d0.is_u8 != 0, wide);
			bool hosted = delta_hosts_a(d0);
			while(j + g.count < pl.delta.v.size() && g.count < DELTA_GROUP_MAX) {
				const DeltaJob &d = pl.delta.v[j + g.count];
				if(d.pred != d0.pred || d.nvert != d0.nvert) break;
				const uint64_t more = delta_vbytes(d.nvert, d.N, d.is_u8 != 0, wide);
				const bool h2 = hosted || delta_hosts_a(d);
				if(vals + more + delta16_graph_lds(d0.nvert, h2) > DELTA16_LDS_MAX) break;
				vals += more; hosted = h2; g.count++;
			}
			pl.delta16_lds = std::max<uint32_t>(pl.delta16_lds, (uint32_t)(vals + delta16_graph_lds(d0.nvert, hosted)));
			pl.delta_groups.v.push_back(g);
			j += g.count;
		}
	}
	place(pl.delta); place(pl.delta_groups); place(pl.cloud); place(pl.cloud_chunk_job); place(pl.normal); place(pl.nv_block_job); place(pl.nv_block_first);
	place(pl.nf_block_job); place(pl.nf_block_first); place(pl.normal_fused_ids); place(pl.dequant); place(pl.dequant_block_job);
	pl.jobs_bytes = cv.take(0) - pl.jobs_begin;
	pl.total = cv.take(0);

}

int Planner::upload() {
	// ---- reserve device + pinned memory; one batch in flight per context ----
	if(harvest(ctx) != CRTHIP_OK) return fail(CRTHIP_E_DEVICE);        // one batch in flight per context: the previous one's status is kept in its object
	if(ctx->scratch.reserve(pl.total + 256) != CRTHIP_OK) return fail(CRTHIP_E_NOMEM);
	if(ctx->staging.reserve(pl.jobs_bytes + 256) != CRTHIP_OK) return fail(CRTHIP_E_NOMEM);
	base = (uint8_t *)ctx->scratch.p;
	auto R = [&](const void *pseudo) -> uint8_t * {          // rebase a scratch-relative pseudo pointer
		uintptr_t v = (uintptr_t)pseudo;
		if(v >> 63) return (uint8_t *)(v & ~(1ull << 63));     // already real (arena)
		return base + v;
	};
	for(auto &t : pl.tun.v) t.dst = R(t.dst);
	for(auto &t : pl.tun_dict.v) t.dst = nullptr;
	for(auto &f : pl.fill.v) f.dst = R(f.dst);
	for(auto &t : pl.topo.v) {
		t.clers = R(t.clers);
		t.group_end = (const uint32_t *)(base + pl.aux_u32.dev_off + (uintptr_t)t.group_end);
		if(!t.pad) t.faces = R(t.faces);
		t.pad = 0;
		t.pred = (uint32_t *)R(t.pred); t.front_a = (uint4 *)R(t.front_a); t.front_b = (uint2 *)R(t.front_b);
		t.order = (uint32_t *)R(t.order); t.delayed = (uint32_t *)R(t.delayed); t.status = (int32_t *)R(t.status); t.flags = (int32_t *)R(t.flags);
	}
	for(auto &u : pl.unpack.v) {
		u.logs = R(u.logs);
		if(!(u.out_u8 & 0x80)) u.out = R(u.out);
		u.out_u8 &= 0x7F;
	}
	for(auto &d : pl.delta.v) { if(!d.pad[0]) d.values = R(d.values); d.pad[0] = 0; d.pred = (const uint32_t *)R(d.pred); if(d.fired) d.fired = R(d.fired); d.flags = (int32_t *)R(d.flags); }
	for(auto &c : pl.cloud.v) { if(!c.pad[0]) c.values = R(c.values); c.pad[0] = 0; }
	for(auto &n : pl.normal.v) {
		n.diffs = (int32_t *)R(n.diffs); n.status = (int32_t *)R(n.status);
		if(n.prediction != 0 && !(n.faces_u16 & 0x80)) n.faces = R(n.faces);
		if(n.prediction != 0 && (n.faces_u16 & 0x40)) n.position = (const int32_t *)R(n.position);
		if(n.fn_scratch) n.fn_scratch = (float *)R(n.fn_scratch);
		n.faces_u16 &= 0x3F;
	}
	for(auto &q : pl.dequant.v) if(q.is_color || q.stride || q.format == CRTHIP_FMT_DOUBLE) q.src = R(q.src);

	// host image -> device (one copy)
	stage = (uint8_t *)ctx->staging.p;
	memset(stage + (pl.unpack_partial_off - pl.jobs_begin), 0, unpack_state_words*8);
	memset(ctx->status_host.p, 0, (size_t)nblobs*16);                       // (after the harvest above: the previous batch's words have been read)
	auto put = [&](auto &arr) { if(!arr.v.empty()) memcpy(stage + (arr.dev_off - pl.jobs_begin), arr.v.data(), arr.v.size()*sizeof(arr.v[0])); };
	put(pl.tun); put(pl.tun_dict); put(pl.tun_chunk_stream); put(pl.tun_group_ids); put(pl.tun_groups); put(pl.fill); put(pl.topo); put(pl.aux_u32); put(pl.topo_lds_ids); put(pl.topo_big_ids); put(pl.topo_glob_ids); put(pl.unpack); put(pl.unpack_chunk_job); put(pl.unpack_wave_ids);
	put(pl.delta); put(pl.delta_groups); put(pl.cloud); put(pl.cloud_chunk_job); put(pl.normal); put(pl.nv_block_job); put(pl.nv_block_first);
	put(pl.nf_block_job); put(pl.nf_block_first); put(pl.normal_fused_ids); put(pl.dequant); put(pl.dequant_block_job);

	return CRTHIP_OK;
}

int Planner::launch() {
	hipStream_t st = ctx->stream;
	ctx->timer.reset();
	Launch LT{ctx};
	if(pl.jobs_bytes) HIP_TRY(hipMemcpyAsync(base + pl.jobs_begin, stage, pl.jobs_bytes, hipMemcpyHostToDevice, st));
	if(pl.zero_end > pl.zero_begin) HIP_TRY(hipMemsetAsync(base + pl.zero_begin, 0, pl.zero_end - pl.zero_begin, st));

	auto D = [&](auto &arr) { return (decltype(arr.v.data()))(base + arr.dev_off); };
	TunTable *tables = (TunTable *)(base + pl.tables_off);
	uint64_t *tun_partial = (uint64_t *)(base + pl.tun_partial_off);
	uint64_t *unpack_partial = (uint64_t *)(base + pl.unpack_partial_off);
	uint64_t *cloud_partial = (uint64_t *)(base + pl.cloud_partial_off);

	const uint32_t ntun = (uint32_t)pl.tun.v.size();
	const uint32_t nfill = (uint32_t)pl.fill.v.size();
	const uint32_t ndict = (uint32_t)pl.tun_dict.v.size();
	// (a launch's streams share dictionaries when at least half of them repeat another one's table and there are enough of them for it to
	// matter: share_clers / share_attrs, decided where the groups were made)
	stat_dicts = (share_clers ? clers_dict : clers_tun) + (share_attrs ? ndict - clers_dict : ntun - clers_tun);
	auto tunstall = [&](hipStream_t s, uint32_t t0, uint32_t t1, uint32_t c0, uint32_t c1, uint32_t f0, uint32_t f1) {
		if(t1 > t0) {                                        // every stream here is one chunk: one wave per stream
			(void)c0; (void)c1;
			// [t0, t1) is the CLERS streams, the attribute streams, or both (dictionaries are numbered the same way)
			const bool has_clers = t0 == 0 && clers_tun > 0, has_attrs = t1 == ntun && ntun > clers_tun;
			const bool share = (!has_clers || share_clers) && (!has_attrs || share_attrs) && (has_clers || has_attrs);
			if(share) {                                        // distinct tables first, then every stream decodes from its (shared) dictionary
				const uint32_t d0 = has_clers ? 0u : clers_dict, d1 = has_attrs ? ndict : clers_dict;
				uint32_t big = 0;                                  // (an alphabet of more than 64 symbols builds its words in LDS: tun_tables.h)
				for(uint32_t d = d0; d < d1; d++) if(pl.tun_dict.v[d].nsym > 64) big = TUN_TABLE_BYTES;
				LT.begin("tunstall_tables", s); hipLaunchKernelGGL(k_tun_tables, dim3(d1 - d0), dim3(64), big, s, D(pl.tun_dict) + d0, d1 - d0, tables); LT.end();
				const uint32_t g0 = has_clers ? 0u : pl.clers_groups, g1 = has_attrs ? (uint32_t)pl.tun_groups.v.size() : pl.clers_groups;
				LT.begin("tunstall_stream", s); hipLaunchKernelGGL(k_tun_stream_grouped, dim3(g1 - g0), dim3(256), 0, s, D(pl.tun), D(pl.tun_group_ids), D(pl.tun_groups) + g0, g1 - g0, tables); LT.end();
			} else {                                           // dictionary + decode in one kernel
				LT.begin("tunstall_stream", s); hipLaunchKernelGGL(k_tun_stream, dim3(t1 - t0), dim3(64), 0, s, D(pl.tun) + t0, t1 - t0); LT.end();
			}
		}
		if(f1 > f0) { LT.begin("fill", s); hipLaunchKernelGGL(k_fill, dim3(f1 - f0), dim3(256), 0, s, D(pl.fill) + f0, f1 - f0); LT.end(); }
	};
	auto unpack = [&](hipStream_t s) {
		const uint32_t nuw = (uint32_t)pl.unpack_wave_ids.v.size();
		if(nuw) { LT.begin("unpack_extract", s); hipLaunchKernelGGL(k_unpack_wave, dim3(nuw), dim3(64), 0, s, D(pl.unpack), D(pl.unpack_wave_ids), nuw); LT.end(); }
		if(!unpack_chunks) return;
		LT.begin("unpack_extract", s); hipLaunchKernelGGL(k_unpack_extract, dim3(unpack_chunks), dim3(256), 0, s, D(pl.unpack), D(pl.unpack_chunk_job), unpack_chunks, unpack_partial); LT.end();
	};
	auto topology = [&]() -> int {
		if(!pl.topo_lds_ids.v.empty() || !pl.topo_big_ids.v.empty()) {
			LT.begin("topology_lds");
			if(!pl.topo_big_ids.v.empty()) { const uint32_t nj = (uint32_t)pl.topo_big_ids.v.size(); hipLaunchKernelGGL(k_topology_lds, dim3(nj), dim3(64), pl.topo_big_lds, st, D(pl.topo), D(pl.topo_big_ids), nj); }
			if(!pl.topo_lds_ids.v.empty()) { const uint32_t nj = (uint32_t)pl.topo_lds_ids.v.size(); hipLaunchKernelGGL(k_topology_lds, dim3(nj), dim3(64), pl.topo_lds, st, D(pl.topo), D(pl.topo_lds_ids), nj); }
			LT.end();
		}
		if(!pl.topo_glob_ids.v.empty()) {
			const uint32_t nj = (uint32_t)pl.topo_glob_ids.v.size();
			LT.begin("topology"); hipLaunchKernelGGL(k_topology, dim3(nj), dim3(64), 0, st, D(pl.topo), D(pl.topo_glob_ids), nj); LT.end();
		}
		return CRTHIP_OK;
	};
	if(pl.tun_multi_chunk) {
		// long streams (scaled Tunstall runs, very large meshes): chunk offsets need one scan over all chunks; single stream
		uint32_t big = 0;
		for(auto &t : pl.tun.v) if(t.nsym > 64) big = TUN_TABLE_BYTES;
		LT.begin("tunstall_tables"); hipLaunchKernelGGL(k_tun_tables, dim3(ntun), dim3(64), big, st, D(pl.tun), ntun, tables); LT.end();
		LT.begin("tunstall_chunk_sums"); hipLaunchKernelGGL(k_tun_chunk_sums, dim3(tun_chunks), dim3(256), 0, st, D(pl.tun), D(pl.tun_chunk_stream), tun_chunks, tables, tun_partial, 0u); LT.end();
		// (up to 256 chunks a stream: every decode wave adds up the sums in front of it itself; longer streams: one workgroup per stream scans them)
		const bool scanned = pl.tun_max_nchunks > 256;
		if(scanned) { LT.begin("tunstall_stream_scan"); hipLaunchKernelGGL(k_tun_stream_scan, dim3(ntun), dim3(256), 0, st, D(pl.tun), ntun, tun_partial); LT.end(); }
		LT.begin("tunstall_decode"); if(launch_tun_decode_staged(st, D(pl.tun), D(pl.tun_chunk_stream), tun_chunks, tables, tun_partial, scanned ? 0u : 1u)) return fail(CRTHIP_E_DEVICE); LT.end();
		if(nfill) { LT.begin("fill"); hipLaunchKernelGGL(k_fill, dim3(nfill), dim3(256), 0, st, D(pl.fill), nfill); LT.end(); }
		{ int e_ = topology(); if(e_) return e_; }
		unpack(st);
	} else if(!pl.topo.v.empty() && (ntun > clers_tun || nfill > clers_fill || unpack_chunks || !pl.unpack_wave_ids.v.empty()) && !ctx->single_stream) {
		// fork: attribute streams on stream2, CLERS + topology on the main stream
		hipStream_t s2 = ctx->stream2;
		HIP_TRY(hipEventRecord(ctx->ev_fork, st));
		HIP_TRY(hipStreamWaitEvent(s2, ctx->ev_fork, 0));
		tunstall(st, 0, clers_tun, 0, clers_chunks, 0, clers_fill);
		{ int e_ = topology(); if(e_) return e_; }
		tunstall(s2, clers_tun, ntun, clers_chunks, tun_chunks, clers_fill, nfill);
		unpack(s2);
		HIP_TRY(hipEventRecord(ctx->ev_join, s2));
		HIP_TRY(hipStreamWaitEvent(st, ctx->ev_join, 0));
	} else {
		tunstall(st, 0, ntun, 0, tun_chunks, 0, nfill);
		{ int e_ = topology(); if(e_) return e_; }
		unpack(st);
	}
	if(!pl.delta.v.empty()) {
		uint32_t ncls[3] = {0, 0, 0};
		for(auto &d : pl.delta.v) ncls[delta_class(d, wide)]++;
		const uint32_t ngroups = (uint32_t)pl.delta_groups.v.size();
		LT.begin("delta_mesh");
		if(ncls[0]) hipLaunchKernelGGL(k_delta_mesh, dim3(ncls[0]), dim3(DELTA_THREADS), 0, st, D(pl.delta), ncls[0]);
		if(ncls[1]) hipLaunchKernelGGL(k_delta_mesh, dim3(ncls[1]), dim3(DELTA_THREADS/2), 0, st, D(pl.delta) + ncls[0], ncls[1]);
		if(ngroups) hipLaunchKernelGGL(k_delta_lds16, dim3(ngroups), dim3(256), pl.delta16_lds, st, D(pl.delta), D(pl.delta_groups), ngroups);
		LT.end();
	}
	if(cloud_chunks) {
		LT.begin("cloud_sums"); hipLaunchKernelGGL(k_cloud_sums, dim3(cloud_chunks), dim3(256), 0, st, D(pl.cloud), D(pl.cloud_chunk_job), cloud_chunks, cloud_partial); LT.end();
		LT.begin("scan"); hipLaunchKernelGGL(k_scan_u64, dim3(1), dim3(1024), 0, st, cloud_partial, cloud_chunks); LT.end();
		LT.begin("cloud_apply"); hipLaunchKernelGGL(k_cloud_apply, dim3(cloud_chunks), dim3(256), 0, st, D(pl.cloud), D(pl.cloud_chunk_job), cloud_chunks, cloud_partial); LT.end();
	}
	const uint32_t nvb = (uint32_t)pl.nv_block_job.v.size(), nfb = (uint32_t)pl.nf_block_job.v.size();
	if(!pl.normal_fused_ids.v.empty()) {
		const uint32_t nj = (uint32_t)pl.normal_fused_ids.v.size();
		LT.begin("normal_blob"); hipLaunchKernelGGL(k_normal_blob, dim3(nj), dim3(256), pl.normal_fused_lds, st, D(pl.normal), D(pl.normal_fused_ids), nj, pl.normal_fused_lds); LT.end();
	}
	if(pl.any_est_normal) {
		float *facen = (float *)(base + pl.facen_off);
		uint32_t *cnt = (uint32_t *)(base + pl.cnt_off), *cursor = (uint32_t *)(base + pl.cursor_off), *bnd = (uint32_t *)(base + pl.bnd_off);
		uint32_t *start = (uint32_t *)(base + pl.start_off), *flag = (uint32_t *)(base + pl.flag_off), *slot = (uint32_t *)(base + pl.slot_off);
		uint32_t *adj = (uint32_t *)(base + pl.adj_off);
		uint64_t *npart = (uint64_t *)(base + pl.nscan_partial_off);
		const uint32_t nv = pl.est_nvert, nch = (nv + CHUNK - 1)/CHUNK;
		LT.begin("normal_faces"); hipLaunchKernelGGL(k_normal_faces, dim3(nfb), dim3(256), 0, st, D(pl.normal), D(pl.nf_block_job), D(pl.nf_block_first), nfb, facen, cnt, bnd); LT.end();
		LT.begin("normal_scan");
		hipLaunchKernelGGL(k_u32_chunk_sums, dim3(nch), dim3(256), 0, st, cnt, nv, npart);
		hipLaunchKernelGGL(k_scan_u64, dim3(1), dim3(1024), 0, st, npart, nch);
		hipLaunchKernelGGL(k_u32_chunk_apply, dim3(nch), dim3(256), 0, st, cnt, start, nv, npart);
		LT.end();
		LT.begin("normal_fill"); hipLaunchKernelGGL(k_normal_fill, dim3(nfb), dim3(256), 0, st, D(pl.normal), D(pl.nf_block_job), D(pl.nf_block_first), nfb, start, cursor, adj); LT.end();
		LT.begin("normal_flags"); hipLaunchKernelGGL(k_normal_flags, dim3(nvb), dim3(256), 0, st, D(pl.normal), D(pl.nv_block_job), D(pl.nv_block_first), nvb, bnd, flag); LT.end();
		LT.begin("normal_scan");
		hipLaunchKernelGGL(k_u32_chunk_sums, dim3(nch), dim3(256), 0, st, flag, nv, npart);
		hipLaunchKernelGGL(k_scan_u64, dim3(1), dim3(1024), 0, st, npart, nch);
		hipLaunchKernelGGL(k_u32_chunk_apply, dim3(nch), dim3(256), 0, st, flag, slot, nv, npart);
		LT.end();
		LT.begin("normal_vertex"); hipLaunchKernelGGL(k_normal_vertex, dim3(nvb), dim3(256), 0, st, D(pl.normal), D(pl.nv_block_job), D(pl.nv_block_first), nvb, facen, start, cnt, adj, flag, slot); LT.end();
	}
	if(pl.any_diff_normal) { LT.begin("normal_diff"); hipLaunchKernelGGL(k_normal_diff, dim3(nvb), dim3(256), 0, st, D(pl.normal), D(pl.nv_block_job), D(pl.nv_block_first), nvb); LT.end(); }
	const uint32_t ndq = (uint32_t)pl.dequant_block_job.v.size();
	if(ndq) { LT.begin("dequantize"); hipLaunchKernelGGL(k_dequant, dim3(ndq), dim3(256), 0, st, D(pl.dequant), D(pl.dequant_block_job), ndq); LT.end(); }

	HIP_TRY(hipGetLastError());                                            // (status: written by the kernels straight into the pinned block)
	HIP_TRY(hipEventRecord(ctx->ev_done, st));
	ctx->done_covers_seq = ctx->upload_seq;

	return CRTHIP_OK;
}

void Planner::account() {
	const uint32_t ntun = (uint32_t)pl.tun.v.size();
	// stats
	b->stats.tunstall_in = stat_tin; b->stats.tunstall_out = stat_tout; b->stats.tunstall_tables = stat_tt; b->stats.tunstall_streams = ntun; b->stats.tunstall_dictionaries = (uint32_t)stat_dicts;
	b->stats.scratch_bytes = pl.total;
	b->stats.topology_scale = std::max(ctx->topo_scale, (ctx->topo_pool_q8 + 7)/8); b->stats.delta_wide = wide ? 1u : 0u;
	uint64_t ob = 0;
	for(auto &P : b->blobs) {
		const BlobLayout &L = P.L;
		if(P.index) ob += (uint64_t)L.h.nface*3*(P.index_u16 ? 2 : 4);
		for(size_t k = 0; k < P.bind.size(); k++) {
			if(!P.bind[k].buffer) continue;
			const AttrHeader &a = L.h.attrs[k];
			if(a.codec == CRTHIP_CODEC_NORMAL) ob += (uint64_t)L.h.nvert*3*(P.bind[k].format == CRTHIP_FMT_INT16 ? 2 : 4);
			else if(a.codec == CRTHIP_CODEC_COLOR) ob += (uint64_t)L.h.nvert*P.bind[k].out_components;
			else ob += (uint64_t)L.h.nvert*a.N*generic_work_bytes(P.bind[k].format);
		}
	}
	b->stats.output_bytes = ob;
	ctx->in_flight = b; ctx->last_decoded = b;
	b->decoded = true; b->planned_wide = wide;
	b->dirty = false;
}
} // namespace

static int build_and_launch_inner(crthip_batch *b) {
	const double t0 = now_us();                                            // host-side cost of a decode call (crthip_batch_stats::host_*_us)
	b->ctx->plan.reset();
	Planner P(b);
	int err = P.carve();
	if(!err) err = P.jobs();
	if(err) return err;
	P.group();
	const double t1 = now_us();
	if((err = P.upload()) != CRTHIP_OK) return err;
	const double t2 = now_us();
	if((err = P.launch()) != CRTHIP_OK) return err;
	P.account();
	const double t3 = now_us();
	b->stats.host_plan_us = (float)(t1 - t0); b->stats.host_stage_us = (float)(t2 - t1); b->stats.host_launch_us = (float)(t3 - t2);
	return CRTHIP_OK;
}

extern "C" int crthip_batch_decode(crthip_batch *b) {
	if(!b || !b->ctx) return fail(CRTHIP_E_ARGUMENT);
	HIP_TRY(hipSetDevice(b->ctx->device));
	return build_and_launch(b);
}

extern "C" int crthip_batch_sync(crthip_batch *b, int32_t *status) {
	if(!b || !b->ctx) return fail(CRTHIP_E_ARGUMENT);
	crthip_ctx *ctx = b->ctx;
	HIP_TRY(hipSetDevice(ctx->device));
	if(ctx->in_flight == b) { if(harvest(ctx) != CRTHIP_OK) return fail(CRTHIP_E_DEVICE); }   // else: harvested when the context moved on (or never decoded)
	int first = CRTHIP_OK;
	for(size_t i = 0; i < b->blobs.size(); i++) {
		if(status) status[i] = b->status[i];
		if(b->status[i] && !first) { first = b->status[i]; fail(first, std::string(crthip_strerror(first)) + " (blob " + std::to_string(i) + ")"); }
	}
	return first;
}

extern "C" int crthip_batch_done(crthip_batch *b) {
	if(!b || !b->ctx) return fail(CRTHIP_E_ARGUMENT);
	crthip_ctx *ctx = b->ctx;
	if(ctx->in_flight != b) return 1;                                  // harvested already, or never decoded
	if(hipSetDevice(ctx->device) != hipSuccess) return fail(CRTHIP_E_DEVICE);
	const hipError_t e = hipEventQuery(ctx->ev_done);
	if(e == hipSuccess) return 1;
	if(e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
	return fail(CRTHIP_E_DEVICE);
}

extern "C" int crthip_batch_get_stats(const crthip_batch *b, crthip_batch_stats *s) {
	if(!b || !s) return fail(CRTHIP_E_ARGUMENT);
	*s = b->stats;
	return CRTHIP_OK;
}

extern "C" int crthip_batch_kernel_times(crthip_batch *b, crthip_kernel_times *t) {
	if(!b || !t) return fail(CRTHIP_E_ARGUMENT);
	crthip_ctx *ctx = b->ctx;
	memset(t, 0, sizeof(*t));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	for(auto &r : ctx->timer.recs) {
		float ms = 0;
		if(hipEventElapsedTime(&ms, ctx->timer.pool[r.e0], ctx->timer.pool[r.e1]) != hipSuccess) continue;
		uint32_t k = 0;
		for(; k < t->count; k++) if(strcmp(t->name[k], r.name) == 0) break;
		if(k == t->count) { if(t->count == CRTHIP_MAX_KERNELS) continue; t->name[k] = r.name; t->count++; }
		t->ms[k] += ms; t->launches[k]++;
	}
	return CRTHIP_OK;
}

extern "C" int64_t crthip_batch_debug_read(crthip_batch *b, uint32_t i, const char *what, void *host_out, size_t cap) {
	if(!b || i >= b->blobs.size() || !what || !host_out || !b->decoded) return fail(CRTHIP_E_ARGUMENT);
	crthip_ctx *ctx = b->ctx;
	if(ctx->last_decoded != b) return fail(CRTHIP_E_ARGUMENT, "the context's scratch block has been reused by a later call");
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	BlobPlan &P = b->blobs[i];
	const uint8_t *src = nullptr; size_t n = 0;
	if(!strcmp(what, "clers")) {
		if(P.dbg_clers == ~0ull) return 0;
		src = P.clers_in_arena ? b->d_arena + P.dbg_clers : (const uint8_t *)ctx->scratch.p + P.dbg_clers; n = P.dbg_nclers;
	} else if(!strcmp(what, "prediction")) {
		if(P.dbg_pred == ~0ull) return 0;
		src = (const uint8_t *)ctx->scratch.p + P.dbg_pred; n = (size_t)P.L.h.nvert*12;
	} else return fail(CRTHIP_E_ARGUMENT);
	n = std::min(n, cap);
	HIP_TRY(hipMemcpy(host_out, src, n, hipMemcpyDeviceToHost));
	return (int64_t)n;
}

// ------------------------------------------------------------------------------------------------
// blobs with HOST output buffers: the crt::Decoder facade's decode() (one blob, crthip_decode_host) and its combiner (the blobs of
// every thread that called decode() at about the same time, decoder_facade.cpp).  The batch object, the device output block and its
// pinned landing zone live in the context: in the steady state a call is one upload of the blobs, one descriptor upload, the kernels,
// and ONE download of all outputs - no allocation, no per-attribute copies.  Serialised per context (host_mutex).
// A blob the walk rejects fails alone: the others are decoded without it.
namespace corto_hip {
int decode_host_many(crthip_ctx *ctx, uint32_t n, HostDecodeReq *reqs, bool copy_out) {
	if(!ctx || !reqs || n == 0) return fail(CRTHIP_E_ARGUMENT);
	std::lock_guard<std::mutex> lock(ctx->host_mutex);
	// whatever way this call ends, no request may be left saying "OK, nothing to copy": a request's status is CRTHIP_OK only once its
	// blob has been through the kernels (the facade's combiner publishes these words to other threads, decoder_facade.cpp)
	auto fail_all = [&](int code) { for(uint32_t i = 0; i < n; i++) if(reqs[i].status == CRTHIP_OK) { reqs[i].status = code; reqs[i].nout = 0; } return code; };
	for(uint32_t i = 0; i < n; i++) { reqs[i].status = CRTHIP_OK; reqs[i].nout = 0; }
	if(hipSetDevice(ctx->device) != hipSuccess) return fail_all(fail(CRTHIP_E_DEVICE, "hipSetDevice"));
	// the blobs the walk accepts (a malformed one must not take its neighbours down)
	std::vector<const uint8_t *> blobs; std::vector<uint32_t> lens; std::vector<uint32_t> who;
	for(uint32_t i = 0; i < n; i++) {
		HostDecodeReq &r = reqs[i];
		if(!r.blob || r.len > 0xFFFFFFFFull) { r.status = r.blob ? CRTHIP_E_LIMIT : CRTHIP_E_ARGUMENT; continue; }
		if(n > 1) { BlobLayout L; const int e = walk_blob(r.blob, r.len, L); if(e) { r.status = fail(e); continue; } }
		blobs.push_back(r.blob); lens.push_back((uint32_t)r.len); who.push_back(i);
	}
	const uint32_t m = (uint32_t)blobs.size();
	if(m == 0) return reqs[0].status;
	int err;
	if(!ctx->host_batch) err = crthip_batch_create(ctx, m, blobs.data(), lens.data(), nullptr, &ctx->host_batch);
	else err = crthip_batch_reset(ctx->host_batch, m, blobs.data(), lens.data(), nullptr);
	if(err) return fail_all(err);
	crthip_batch *b = ctx->host_batch;
	// outputs of every blob back to back in one device block (16-byte aligned pieces)
	std::vector<crthip_attr_binding> dev; std::vector<size_t> dev_first(m, 0); std::vector<void *> dindex(m, nullptr); std::vector<uint32_t> ifmt(m, CRTHIP_FMT_UINT32);
	struct Piece { uint32_t req, slot; size_t off, bytes; void *host; };
	std::vector<Piece> pieces;
	size_t total = 0;
	for(uint32_t k = 0; k < m; k++) {
		HostDecodeReq &r = reqs[who[k]];
		const BlobLayout &L = b->blobs[k].L;
		const uint32_t nvert = L.h.nvert, nface = L.h.nface;
		const size_t na = L.h.attrs.size();
		dev_first[k] = dev.size();
		for(size_t a = 0; a < na; a++) {
			// attrs == NULL: nothing bound (an index-only decode); host buffers have upstream's packed layouts - a stride is refused, not ignored
			crthip_attr_binding d;
			if(r.attrs) d = r.attrs[a]; else { d.buffer = nullptr; d.format = CRTHIP_FMT_FLOAT; d.out_components = 0; d.stride = 0; }
			if(d.buffer && d.stride) { r.status = fail(CRTHIP_E_ARGUMENT, "crthip_decode_host: host buffers are tightly packed (stride must be 0)"); d.buffer = nullptr; }
			d.stride = 0; d.reserved = 0;
			if(d.buffer && r.status == CRTHIP_OK) {
				const AttrHeader &A = L.h.attrs[a];
				size_t bytes;
				if(A.codec == CRTHIP_CODEC_NORMAL) bytes = (size_t)nvert*3*(d.format == CRTHIP_FMT_INT16 ? 2 : 4);
				else if(A.codec == CRTHIP_CODEC_COLOR) bytes = (size_t)nvert*(d.out_components ? d.out_components : 4);
				else bytes = (size_t)nvert*A.N*generic_work_bytes(d.format);   // (the decode works in int32 / int64 records whatever the output format: DESIGN.md 1)
				pieces.push_back(Piece{who[k], (uint32_t)a, total, bytes, d.buffer});
				d.buffer = (void *)(uintptr_t)(total + 1);                    // (offset + 1: rebased below, once the block is there)
				total += (bytes + 15) & ~(size_t)15;
			} else d.buffer = nullptr;
			dev.push_back(d);
		}
		if(r.index && nface && r.status == CRTHIP_OK) {
			const size_t bytes = (size_t)nface*3*(r.index_format == CRTHIP_FMT_UINT16 ? 2 : 4);
			pieces.push_back(Piece{who[k], CRTHIP_MAX_ATTRS, total, bytes, r.index});
			dindex[k] = (void *)(uintptr_t)(total + 1); ifmt[k] = r.index_format;
			total += (bytes + 15) & ~(size_t)15;
		}
	}
	if(ctx->host_out.reserve(total + 16) != CRTHIP_OK || ctx->host_pin.reserve(total + 16) != CRTHIP_OK) return fail_all(fail(CRTHIP_E_NOMEM));
	uint8_t *dbase = (uint8_t *)ctx->host_out.p, *hbase = (uint8_t *)ctx->host_pin.p;
	for(auto &d : dev) if(d.buffer) d.buffer = dbase + ((uintptr_t)d.buffer - 1);
	for(auto &p : dindex) if(p) p = dbase + ((uintptr_t)p - 1);
	// bound blob by blob: a binding the device path refuses (a format, an alignment) fails ITS blob - which is then decoded with nothing
	// bound - and nobody else's (upstream's Decoder objects share nothing, src/decoder.cpp:126-196)
	err = CRTHIP_OK;
	for(uint32_t k = 0; k < m && !err; k++) {
		HostDecodeReq &r = reqs[who[k]];
		const size_t na = b->blobs[k].L.h.attrs.size();
		int e = r.status != CRTHIP_OK ? r.status : crthip_batch_bind(b, k, na ? dev.data() + dev_first[k] : nullptr, dindex[k], ifmt[k]);
		if(e) {
			if(r.status == CRTHIP_OK) r.status = e;
			for(size_t a = 0; a < na; a++) dev[dev_first[k] + a].buffer = nullptr;
			e = crthip_batch_bind(b, k, na ? dev.data() + dev_first[k] : nullptr, nullptr, CRTHIP_FMT_UINT32);
			if(e) err = e;                                                 // (cannot happen: nothing is bound)
		}
	}
	if(!err) err = crthip_batch_decode(b);
	if(!err && total && hipMemcpyAsync(hbase, dbase, total, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) err = fail(CRTHIP_E_DEVICE);
	std::vector<int32_t> st(m, 0);
	const int serr = crthip_batch_sync(b, st.data());      // (waits for the kernels: the event behind them; keeps the context consistent on error)
	if(hipStreamSynchronize(ctx->stream) != hipSuccess && !err) err = fail(CRTHIP_E_DEVICE);   // ... and this for the copy behind them
	if(!err && (serr == CRTHIP_E_DEVICE || serr == CRTHIP_E_NOMEM)) err = serr;
	for(uint32_t k = 0; k < m; k++) { HostDecodeReq &r = reqs[who[k]]; if(r.status != CRTHIP_OK) continue; r.status = err ? err : st[k]; }
	for(const Piece &p : pieces) {
		HostDecodeReq &r = reqs[p.req];
		if(r.status != CRTHIP_OK) continue;
		if(copy_out) memcpy(p.host, hbase + p.off, p.bytes);
		else if(r.nout < CRTHIP_MAX_ATTRS + 1) { r.out_src[r.nout] = hbase + p.off; r.out_dst[r.nout] = p.host; r.out_bytes[r.nout] = p.bytes; r.nout++; }
	}
	if(err) return err;
	for(uint32_t i = 0; i < n; i++) if(reqs[i].status) return reqs[i].status;            // (the first failing blob's code - its message is the thread's last error; every status is in its request)
	return CRTHIP_OK;
}
}

extern "C" int crthip_decode_host(crthip_ctx *ctx, const uint8_t *blob, size_t len, const crthip_attr_binding *attrs,
                                  void *index, uint32_t index_format) {
	if(!ctx || !blob) return fail(CRTHIP_E_ARGUMENT);
	if(len > 0xFFFFFFFFull) return fail(CRTHIP_E_LIMIT, "blob larger than 4 GiB");
	HostDecodeReq r{};
	r.blob = blob; r.len = len; r.attrs = attrs; r.index = index; r.index_format = index_format;
	return decode_host_many(ctx, 1, &r, true);
}

// ------------------------------------------------------------------------------------------------
// stand-alone Tunstall run over device-resident blocks (roofline measurement of K-TAB/K-TUN)
extern "C" int crthip_tunstall_decode_blocks(crthip_ctx *ctx, uint32_t n, const uint8_t *host_blocks, const void *device_blocks,
                                             const uint64_t *block_offset, void *device_out, const uint64_t *out_offset,
                                             crthip_kernel_times *times) {
	if(!ctx || !host_blocks || !device_blocks || !block_offset || !device_out || !out_offset) return fail(CRTHIP_E_ARGUMENT);
	HIP_TRY(hipSetDevice(ctx->device));
	if(harvest(ctx) != CRTHIP_OK) return fail(CRTHIP_E_DEVICE);
	ctx->last_decoded = nullptr;
	std::vector<TunStream> tun; std::vector<uint32_t> chunk_stream; std::vector<FillJob> fills;
	uint32_t chunks = 0, max_nchunks = 0; bool multi = false;
	for(uint32_t i = 0; i < n; i++) {
		const uint8_t *p = host_blocks + block_offset[i];
		const uint32_t ns = p[0];
		auto rd = [&](const uint8_t *q) { return (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24); };
		const uint32_t size = rd(p + 1 + 2*ns), csize = rd(p + 5 + 2*ns);
		const uint8_t *dblk = (const uint8_t *)device_blocks + block_offset[i];
		uint8_t *dst = (uint8_t *)device_out + out_offset[i];
		if(size == 0) continue;
		if(ns == 1) { fills.push_back(FillJob{dst, size, p[1]}); continue; }
		if(ns == 0 || csize == 0) return fail(CRTHIP_E_TRUNCATED);
		TunStream t{};
		t.src = dblk + 9 + 2*ns; t.dst = dst; t.probs = dblk + 1; t.csize = csize; t.size = size; t.nsym = ns; t.table = (uint32_t)tun.size();
		t.chunk0 = chunks; tun_pick_geometry(t);
		if(t.nchunks > 1) multi = true;
		max_nchunks = std::max(max_nchunks, t.nchunks);
		for(uint32_t c = 0; c < t.nchunks; c++) chunk_stream.push_back((uint32_t)tun.size());
		chunks += t.nchunks;
		tun.push_back(t);
	}
	Carver cv;
	const uint64_t o_tab = cv.take(tun.size()*sizeof(TunTable)), o_part = cv.take(((uint64_t)chunks*4 + 4)*8);
	const uint64_t o_jobs = cv.take(0);
	const uint64_t o_tun = cv.take(tun.size()*sizeof(TunStream) + 16, 16), o_cs = cv.take(chunk_stream.size()*4 + 16, 16), o_fill = cv.take(fills.size()*sizeof(FillJob) + 16, 16);
	const uint64_t total = cv.take(0);
	if(ctx->scratch.reserve(total + 256) != CRTHIP_OK || ctx->staging.reserve(total - o_jobs + 256) != CRTHIP_OK) return fail(CRTHIP_E_NOMEM);
	uint8_t *base = (uint8_t *)ctx->scratch.p, *stage = (uint8_t *)ctx->staging.p;
	if(!tun.empty()) memcpy(stage + (o_tun - o_jobs), tun.data(), tun.size()*sizeof(TunStream));
	if(!chunk_stream.empty()) memcpy(stage + (o_cs - o_jobs), chunk_stream.data(), chunk_stream.size()*4);
	if(!fills.empty()) memcpy(stage + (o_fill - o_jobs), fills.data(), fills.size()*sizeof(FillJob));
	hipStream_t st = ctx->stream;
	HIP_TRY(hipMemcpyAsync(base + o_jobs, stage, total - o_jobs, hipMemcpyHostToDevice, st));
	ctx->timer.reset();
	Launch LT{ctx};
	TunStream *dt = (TunStream *)(base + o_tun); uint32_t *dcs = (uint32_t *)(base + o_cs);
	TunTable *tables = (TunTable *)(base + o_tab); uint64_t *part = (uint64_t *)(base + o_part);
	const uint32_t ntun = (uint32_t)tun.size();
	if(ntun) {
		uint32_t big = 0;
		for(auto &t : tun) if(t.nsym > 64) big = TUN_TABLE_BYTES;
		LT.begin("tunstall_tables"); hipLaunchKernelGGL(k_tun_tables, dim3(ntun), dim3(64), big, st, dt, ntun, tables); LT.end();
		const bool scanned = max_nchunks > 256;
		if(multi) {
			LT.begin("tunstall_chunk_sums"); hipLaunchKernelGGL(k_tun_chunk_sums, dim3(chunks), dim3(256), 0, st, dt, dcs, chunks, tables, part, 0u); LT.end();
			if(scanned) { LT.begin("tunstall_stream_scan"); hipLaunchKernelGGL(k_tun_stream_scan, dim3(ntun), dim3(256), 0, st, dt, ntun, part); LT.end(); }
		}                                                                   // (up to 256 chunks a stream: every decode wave adds up the sums in front of it itself)
		LT.begin("tunstall_decode");
		if(multi) { if(launch_tun_decode_staged(st, dt, dcs, chunks, tables, part, scanned ? 0u : 1u)) return fail(CRTHIP_E_DEVICE); }
		else hipLaunchKernelGGL(k_tun_decode, dim3(chunks), dim3(256), 0, st, dt, dcs, chunks, tables, part, 0u);
		LT.end();
	}
	if(!fills.empty()) { LT.begin("fill"); hipLaunchKernelGGL(k_fill, dim3((uint32_t)fills.size()), dim3(256), 0, st, (FillJob *)(base + o_fill), (uint32_t)fills.size()); LT.end(); }
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(st));
	if(times) {
		memset(times, 0, sizeof(*times));
		for(auto &r : ctx->timer.recs) {
			float ms = 0;
			if(hipEventElapsedTime(&ms, ctx->timer.pool[r.e0], ctx->timer.pool[r.e1]) != hipSuccess) continue;
			uint32_t k = 0;
			for(; k < times->count; k++) if(strcmp(times->name[k], r.name) == 0) break;
			if(k == times->count) { if(times->count == CRTHIP_MAX_KERNELS) continue; times->name[k] = r.name; times->count++; }
			times->ms[k] += ms; times->launches[k]++;
		}
	}
	return CRTHIP_OK;
}
