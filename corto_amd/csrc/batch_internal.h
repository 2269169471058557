// batch_internal.h — what the pieces of the decode path's host side share: the context and batch objects, the plan of one decode call and
// the
// staged Planner (batch.cpp: the C ABI, context, harvest; plan_carve.cpp / plan_jobs.cpp / plan_group.cpp: the planner's stages;
// plan_launch.cpp:
// upload, the kernel schedule, the stats).  Round 5 split batch.cpp (1 640 lines) along the Planner's stages.  Not part of the C ABI.
#pragma once
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

// errors: the thread's last message (crthip_last_error) and the code handed back
int fail(int code, const std::string &msg);
int fail(int code);
#define HIP_TRY(expr) \
	do { hipError_t e_ = (expr); if(e_ != hipSuccess) return fail(CRTHIP_E_DEVICE, std::string(#expr ": ") + hipGetErrorString(e_)); } while(0)
inline double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

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

template <typename T> struct HostArr {  // host image of a device array + where it goes
	std::vector<T> v; uint64_t dev_off = 0;
};

struct AttrScratch {
	// vals: int32 workspace of a generic attribute bound with a stride; facen: the fused normal kernel's face normals when not in LDS
	uint64_t color = ~0ull, diffs = ~0ull, fired = ~0ull, vals = ~0ull, facen = ~0ull; std::vector<uint64_t> sym;
	void reset() { color = diffs = fired = vals = facen = ~0ull; sym.clear(); }
};
struct BlobScratch {
	uint64_t clers = ~0ull, pred = ~0ull, front_a = ~0ull, front_b = ~0ull, order = ~0ull, delayed = ~0ull, faces = ~0ull;
	uint64_t progress = ~0ull;                              // the automaton's progress word (zeroed): blobs with an attribute too big for K-DELTA's LDS records
	uint32_t front_cap = 0, aux_groups = 0;
	std::vector<AttrScratch> attr;
	size_t nattr = 0;                                       // attr[0..nattr) are this decode's (the vector only grows)
	void reset() { clers = pred = front_a = front_b = order = delayed = faces = progress = ~0ull; front_cap = 0; aux_groups = 0; nattr = 0; }
	void set_attrs(size_t n) { if(attr.size() < n) attr.resize(n); for(size_t k = nattr; k < n; k++) attr[k].reset();
		if(n > nattr) nattr = n; }
};

struct Plan {
	// job arrays
	// tun_dict: one entry per DISTINCT probability table (shared dictionaries)
	HostArr<TunStream> tun, tun_dict; HostArr<uint32_t> tun_chunk_stream;
	// streams by dictionary, in groups of one dictionary each (k_tun_stream_grouped)
	HostArr<uint32_t> tun_group_ids; HostArr<TunGroup> tun_groups; uint32_t clers_groups = 0;
	HostArr<FillJob> fill;
	HostArr<TopoJob> topo; HostArr<uint32_t> aux_u32;     // group_end lists
	// LDS automata in two size classes, one launch each
	HostArr<uint32_t> topo_lds_ids, topo_big_ids, topo_glob_ids; uint32_t topo_lds = 0, topo_big_lds = 0;
	// LDS bytes of topo_lds_ids' entries until they are split into the two classes
	std::vector<uint32_t> topo_need;
	// (unpack_wave_ids: the streams of small bit blocks, one wave each: k_unpack_wave)
	HostArr<UnpackJob> unpack; HostArr<uint32_t> unpack_chunk_job, unpack_wave_ids;
	HostArr<DeltaJob> delta;
	HostArr<DeltaGroup> delta_groups;                       // blobs whose attributes share one k_delta_lds16 workgroup
	HostArr<CloudJob> cloud; HostArr<uint32_t> cloud_chunk_job;
	HostArr<NormalJob> normal; HostArr<uint32_t> nv_block_job, nv_block_first, nf_block_job, nf_block_first, normal_fused_ids;
	uint32_t normal_fused_lds = 0;
	HostArr<DequantJob> dequant; HostArr<uint32_t> dequant_block_job;
	// scratch regions (offsets)
	uint64_t zero_begin = 0, zero_end = 0;
	uint64_t status_off = 0, tables_off = 0, tun_partial_off = 0, unpack_partial_off = 0, cloud_partial_off = 0;
	uint64_t facen_off = 0, cnt_off = 0, cursor_off = 0, bnd_off = 0, start_off = 0, flag_off = 0, slot_off = 0, adj_off = 0,
		nscan_partial_off = 0;
	uint64_t jobs_begin = 0, jobs_bytes = 0;
	uint32_t est_nvert = 0, est_nface = 0;                // totals over ESTIMATED/BORDER jobs
	uint32_t delta16_lds = 0;                              // largest LDS request among the k_delta_lds16 groups
	bool tun_multi_chunk = false, any_diff_normal = false, any_est_normal = false;
	uint32_t tun_max_nchunks = 0;
	uint64_t total = 0;
	template <typename A> static void clr(A &a) { a.v.clear(); a.dev_off = 0; }
	void reset() {                                          // keep every vector's capacity
		clr(tun); clr(tun_dict); clr(tun_chunk_stream); clr(tun_group_ids); clr(tun_groups); clers_groups = 0; clr(fill); clr(topo);
			clr(aux_u32); clr(topo_lds_ids); clr(topo_big_ids); clr(topo_glob_ids); topo_need.clear();
		clr(unpack); clr(unpack_chunk_job); clr(unpack_wave_ids); clr(delta); clr(delta_groups); clr(cloud); clr(cloud_chunk_job);
			clr(normal); clr(nv_block_job); clr(nv_block_first);
		clr(nf_block_job); clr(nf_block_first); clr(normal_fused_ids); clr(dequant); clr(dequant_block_job);
		topo_lds = topo_big_lds = normal_fused_lds = 0;
		zero_begin = zero_end = status_off = tables_off = tun_partial_off = unpack_partial_off = cloud_partial_off = 0;
		facen_off = cnt_off = cursor_off = bnd_off = start_off = flag_off = slot_off = adj_off = nscan_partial_off = 0;
		jobs_begin = jobs_bytes = 0; est_nvert = est_nface = 0; delta16_lds = 0;
		tun_multi_chunk = any_diff_normal = any_est_normal = false; total = 0; tun_max_nchunks = 0;
	}
};

struct crthip_ctx {
	int device = 0;
	hipStream_t stream = nullptr;
	hipStream_t stream2 = nullptr;  // attribute streams (Tunstall + bit-unpack) run here while the main stream does topology
	hipEvent_t ev_fork = nullptr, ev_join = nullptr;
	// recorded behind a decode's last kernel: what sync / done wait for, so that work a caller queues on the stream BEHIND a decode
	hipEvent_t ev_done = nullptr;
	                                // does not delay the harvest of this one
	// arena_pin uploads enqueued so far / how many of them sit IN FRONT of ev_done (harvest may only call those complete)
	uint32_t upload_seq = 0, done_covers_seq = 0;
	DebugConfig dbg;                // every environment switch, read once when the context is made (debug_config.h)
	// largest LDS request for which K-NRM keeps its face normals in LDS (0 for a context that is one of many: crthip_ctx_set_single_stream)
	uint32_t normal_fn_max = NORMAL_FN_LDS_MAX;
	uint8_t single_stream = 0;                        // crthip_ctx_set_single_stream: no second HIP stream for the attribute streams
	DeviceBuf scratch;        // symbols, tables, fronts, predictions, job arrays ... (one batch in flight at a time)
	PinnedBuf staging;        // host image of the job arrays
	PinnedBuf arena_pin;      // host image of a batch's blobs on their way to the device (batch_fill: one H2D copy, not waited for)
	PinnedBuf status_host;
	bool profiling = false;
	KernelTimer timer;
	crthip_batch *in_flight = nullptr;   // decode enqueued, status not harvested yet
	// crthip_ctx_set_packed_host_blobs: blobs laid out as an arena in the caller's pinned memory go up from there
	bool packed_host = false;
	// a batch's blobs are (perhaps still) on their way from arena_pin: cleared by whoever synchronises the stream
	bool arena_upload_pending = false;
	crthip_batch *last_decoded = nullptr;// whose intermediates the scratch block holds (crthip_batch_debug_read)
	// crthip_decode_host: everything a one-blob decode with host buffers needs, kept from call to call (no hipMalloc / create in the
	// steady state) and guarded by a mutex so that callers may share a context between threads
	std::mutex host_mutex;
	crthip_batch *host_batch = nullptr;
	DeviceBuf host_out;       // decoded outputs of the one blob, back to back
	PinnedBuf host_pin;       // ... and their landing zone in pinned host memory (one async D2H copy)
	// feedback on the LDS edge slots of the CLERS automaton: raised after a batch with fallbacks, lowered after a long calm run
	// K-TOPO's learnt slots: ring x topo_scale (a power of two), pool x topo_pool_q8 / 8 (kernels.h: topo_lds_geometry)
	uint32_t topo_scale = 1, topo_pool_q8 = 8, topo_pool_cap = 0, topo_calm = 0, topo_patience = 64;
	// planner state reused from one decode call to the next (batch.cpp: build_and_launch)
	Plan plan;
	std::vector<BlobScratch> plan_scratch;
	std::vector<const uint8_t *> plan_clers, plan_logs;
	// streams of a batch that carry the same probability table share one dictionary: exact match on the table's bytes (alphabets of up
	// to 16 symbols; bigger ones hardly ever repeat and are quick to build), open addressing on a hash of them
	struct DictKey { uint8_t n, bytes[32]; };
	std::vector<DictKey> dict_keys;
	std::vector<uint32_t> dict_slots, dict_used, dict_ids, dict_count;
	// K-DELTA keeps 32-bit values in LDS: $CORTO_DELTA_WIDE=1, or learnt from a batch whose 16-bit relative values overflowed
	bool delta_wide = false;
	uint32_t delta_calm = 0, delta_patience = 256;
	// the narrow layout is on trial again after a wide spell (an overflow now doubles the patience)
	bool delta_just_narrowed = false;
};

// bytes per component of a generic attribute's caller buffer: upstream decodes in place as int32 whatever the format and DOUBLE widens
// in place (include/corto/vertex_attribute.h:184-228), so every format's buffer is nvert*N*4 bytes but DOUBLE's
static inline size_t generic_work_bytes(uint32_t format) { return format == CRTHIP_FMT_DOUBLE ? 8u : 4u; }

struct Binding { void *buffer = nullptr; uint32_t format = CRTHIP_FMT_FLOAT, out_components = 4, stride = 0; bool stream_values = false; };   // stream_values: CRTHIP_BIND_STREAM_VALUES

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

int harvest(crthip_ctx *ctx);     // wait for the batch in flight on the context, keep its per-blob status, learn from its flags (batch.cpp)

// ------------------------------------------------------------------------------------------------
// planner: everything below turns the walked layouts + bindings into job arrays inside one scratch block

struct Carver {                         // bump allocator over the scratch block (offsets only)
	uint64_t off = 0;
	uint64_t take(uint64_t bytes, uint64_t align = 256) { off = (off + align - 1) & ~(align - 1); uint64_t r = off; off += bytes;
		return r; }
};


static int32_t f2i_x86_host(float x) {
	if(!(x > -2147483904.0f && x < 2147483648.0f)) return (int32_t)0x80000000;
	return (int32_t)x;
}




// launch classes of K-DELTA: 2 - values + prediction graph fit LDS, one wave per attribute (k_delta_lds16, k_delta.hip): as int16 relative
// to
// vertex 0, or - `wide`: a context that met values beyond int16 - as int32; else the stretch walk over HBM (k_delta_mesh): 0 = large, 1 =
// small
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
static bool normal_fused(uint32_t nvert, uint32_t nface) { return nvert <= 32767 && (uint64_t)3*nface <= 65535 && normal_blob_lds(nvert,
	nface) <= NORMAL_LDS_MAX; }

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

// The planner of one decode call, stage by stage (round 4: this was one function of 660 lines).  carve() lays the batch's scratch out
// (pass 1: sizes and offsets only), jobs() writes the job descriptors of every stage with scratch-relative pseudo pointers (pass 2),
// group() sorts streams by dictionary and attributes into K-DELTA workgroups and places the job arrays, upload() reserves the blocks,
// rebases the pointers and stages the arrays, launch() enqueues the kernels in the order of crt::Decoder::decodeMesh / decodePointCloud
// (src/decoder.cpp:133-196), account() fills crthip_batch_stats.
struct Planner {
	crthip_batch *b; crthip_ctx *ctx; Plan &pl; std::vector<BlobScratch> &bs;
	// wide: K-DELTA with 32-bit values in LDS (this context met values beyond int16)
	const uint32_t nblobs; const bool wide; const uint8_t *arena;
	Carver cv;
	uint64_t unpack_state_words = 1, n_tun = 0, stat_tin = 0, stat_tout = 0, stat_tt = 0, stat_dicts = 0;
	uint32_t tun_chunks = 0, unpack_chunks = 0, cloud_chunks = 0;
	// the CLERS streams come first in every stream / chunk / fill / dictionary array
	uint32_t clers_tun = 0, clers_chunks = 0, clers_fill = 0, clers_dict = 0;
	bool share_clers = false, share_attrs = false;
	int32_t *hs_base = nullptr;                                              // per-blob status words in pinned host memory
	uint8_t *base = nullptr, *stage = nullptr;                               // the scratch block; the host image of the job arrays

	Planner(crthip_batch *b_) : b(b_), ctx(b_->ctx), pl(b_->ctx->plan), bs(b_->ctx->plan_scratch), nblobs((uint32_t)b_->blobs.size()),
		wide(b_->ctx->delta_wide), arena(b_->d_arena) {}
	static uint8_t *SP(uint64_t off) { return (uint8_t *)(uintptr_t)off; }   // scratch-relative pseudo pointer
	// real pointer (bit 63: R() leaves it alone)
	int32_t *HS(uint64_t k) const { return (int32_t *)((uintptr_t)(hs_base + k) | (1ull << 63)); }
	int carve(); int jobs(); void group(); int upload(); int launch(); void account();
};

