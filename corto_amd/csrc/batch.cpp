// batch.cpp — host side of the MI355X decode path: context (stream, scratch pool), batch planner
// (walk -> job descriptors -> one H2D upload) and the kernel schedule.  Compiled by hipcc.
//
// The schedule restates the stage order of crt::Decoder::decodeMesh / decodePointCloud
// (src/decoder.cpp:133-196) for a whole batch of blobs at once:
//   decode-all (Tunstall + bit-unpack) -> topology -> delta-all -> postDelta (normals) -> dequantize-all.
#include "batch_internal.h"

// ------------------------------------------------------------------------------------------------
static thread_local std::string g_error;
int fail(int code, const std::string &msg) { g_error = msg; return code; }

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
int fail(int code) { return fail(code, crthip_strerror(code)); }
// context plumbing for the encoder stages (encoder_internal.h)
namespace corto_hip { int ctx_fail(int code, const char *msg) { return fail(code, msg ? std::string(msg) :
	std::string(crthip_strerror(code))); } }
extern "C" const char *crthip_last_error(void) { return g_error.c_str(); }
extern "C" uint32_t crthip_abi_version(void) { return CRTHIP_ABI_VERSION; }


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

// Wait for the batch in flight on this context (if any) and move its per-blob status from the pinned landing zone into the
// batch object.  EVERY place that is about to reuse the context's stream, scratch or status buffer goes through here, so a
// decode(A); decode(B); sync(A) sequence still reports A's failures (status used to be read only by crthip_batch_sync(A) and
// was lost when another call had synchronised first).
constexpr uint32_t TOPO_SLOTS_MAX = 4096;       // ring and pool slots of the automaton's LDS form (topo_lds_geometry's ring_max)
int harvest(crthip_ctx *ctx) {
	crthip_batch *b = ctx->in_flight;
	if(!b) return CRTHIP_OK;
	if(hipEventSynchronize(ctx->ev_done) != hipSuccess) { ctx->in_flight = nullptr; return CRTHIP_E_DEVICE; }
	// (only an upload that sits in front of the event: one enqueued behind the decode is still on its way)
	if(ctx->done_covers_seq == ctx->upload_seq) ctx->arena_upload_pending = false;
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
			// overflowed right after narrowing again (never on a context's first overflow)
			if(ctx->delta_just_narrowed && ctx->delta_patience < (1u << 20)) ctx->delta_patience *= 2;
			ctx->delta_calm = 0;
		}
		ctx->delta_just_narrowed = false;                                   // (narrow and fine, or wide from here on)
	} else if(!ctx->dbg.delta_wide && ++ctx->delta_calm >= ctx->delta_patience) { ctx->delta_wide = false; ctx->delta_calm = 0;
		ctx->delta_just_narrowed = true; }
	// more than one blob in twenty redone on the HBM front (5x slower): more edge slots from the next batch on - as many as the redone
	// blobs say they
	// would have needed (k_topology_lds' redo reports ring and pool slots above bit 0 of the flags word; round 4 went up four-fold whatever
	// was
	// missing, and a batch of Delaunay discs that needed 600 pool slots of its 512 got 2 048 + 2 048: 72 KB of LDS a blob, two automata a
	// CU, the
	// pipeline at a third of its rate); a long run without any: try less again, and be more patient the next time that turns out to be too
	// little
	if(b->stats.topology_fallbacks*20 > n) {
		uint32_t ring_m = 1, pool_q8 = 8;
		const uint32_t cap_before = ctx->topo_pool_cap;
		uint32_t learnable = 0;
		for(size_t i = 0; i < n; i++) if(hs[n + i] & 1) {
			// a redo that ended in an error (an untrusted stream runs its pool dry before it fails) teaches nothing, and a front beyond the LDS
			// form's 4 096 + 4 096 records is on the HBM path by design: neither may scale every OTHER blob's request up
			if(b->status[i]) continue;
			if((((uint32_t)hs[n + i] >> 1) & 0x7FFFu) > TOPO_SLOTS_MAX || ((uint32_t)hs[n + i] >> 16) > TOPO_SLOTS_MAX) continue;
			learnable++;
			const auto &h = b->blobs[i].L.h;
			uint32_t ring, pool, symwin;
			topo_lds_geometry(h.nface, b->blobs[i].L.clers.size, 4096, 1, 8, n >= 32 ? 4u : 8u, topo_boundary_estimate(h.nvert, h.nface),
				ring, pool, symwin);
			const uint32_t need_ring = ((uint32_t)hs[n + i] >> 1) & 0x7FFFu, need_pool = (uint32_t)hs[n + i] >> 16;
			uint32_t m = 1;
			while(ring*m < need_ring && m < 16) m <<= 1;
			ring_m = std::max(ring_m, m);
			pool_q8 = std::max(pool_q8, std::min(128u, (need_pool*8 + pool - 1)/pool + 1u));      // (an eighth on top)
			ctx->topo_pool_cap = std::max(ctx->topo_pool_cap, need_pool + need_pool/8);
		}
		const bool grew = ring_m > ctx->topo_scale || pool_q8 > ctx->topo_pool_q8 || ctx->topo_pool_cap > cap_before;
		ctx->topo_scale = std::max(ctx->topo_scale, ring_m); ctx->topo_pool_q8 = std::max(ctx->topo_pool_q8, pool_q8);
		// (they fell back with what they asked for: capacity, or a need the redo cannot see)
		if(!grew && learnable) { ctx->topo_scale = std::min(16u, ctx->topo_scale*2); ctx->topo_pool_q8 = std::min(128u, ctx->topo_pool_q8*2);
			ctx->topo_pool_cap *= 2; }
		ctx->topo_pool_cap = std::min(ctx->topo_pool_cap, TOPO_SLOTS_MAX + TOPO_SLOTS_MAX/8);      // (a uint32 that doubled without a bound wrapped to 0 = "no cap")
		if(ctx->topo_calm == 0 && ctx->topo_patience < (1u << 20)) ctx->topo_patience *= 2;    // fell back right after scaling down
		ctx->topo_calm = 0;
	} else if(b->stats.topology_fallbacks == 0 && (ctx->topo_scale > 1 || ctx->topo_pool_q8 > 8) &&
		++ctx->topo_calm >= ctx->topo_patience) {
		ctx->topo_scale = std::max(1u, ctx->topo_scale/2); ctx->topo_pool_q8 = std::max(8u, ctx->topo_pool_q8*3/4); ctx->topo_pool_cap =
			ctx->topo_pool_cap*3/4; ctx->topo_calm = 0;
	}
	ctx->in_flight = nullptr;
	return CRTHIP_OK;
}

namespace corto_hip {
int ctx_device(crthip_ctx *ctx) { return ctx->device; }
hipStream_t ctx_stream(crthip_ctx *ctx) { return ctx->stream; }
// on the context's main stream, ordered before its next decode
int ctx_fill_async(crthip_ctx *ctx, void *dst, size_t bytes, int value) {
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
	   hipFuncSetAttribute((const void *)k_topology_lds_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TOPO_LDS_MAX) != hipSuccess ||
	   hipFuncSetAttribute((const void *)k_delta_lds16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DELTA16_LDS_MAX) != hipSuccess ||
	   hipFuncSetAttribute((const void *)k_normal_blob, hipFuncAttributeMaxDynamicSharedMemorySize, (int)NORMAL_LDS_MAX) != hipSuccess ||
	   hipFuncSetAttribute((const void *)k_enc_tun_parse, hipFuncAttributeMaxDynamicSharedMemorySize,
	   	(int)enc_parse_lds(ENC_TRIE_LDS_MAX)) != hipSuccess) {
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
	c->scratch.release(); c->staging.release(); c->arena_pin.release(); c->status_host.release(); c->host_out.release();
		c->host_pin.release();
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

// (re)fill a batch object from a list of blobs: header parse + bounds-checked walk of each, arena layout, upload unless resident
static int batch_fill(crthip_ctx *ctx, crthip_batch *b, uint32_t nblobs, const uint8_t *const *blobs, const uint32_t *lens,
	const void *device_arena) {
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
			// the copy below reads the caller's buffer whenever the DMA engine gets to it: nothing here snapshots it or waits (corto_hip.h:
			// the
			// caller keeps it alive and unchanged until the batch is synced).  A pageable buffer would still work (HIP stages it), a pinned
			// one
			// is what the switch promises: on request, check
			if(ctx->dbg.check_pinned) {
				hipPointerAttribute_t pa;
				if(hipPointerGetAttributes(&pa, blobs[0]) != hipSuccess || pa.type != hipMemoryTypeHost) { (void)hipGetLastError();
					return fail(CRTHIP_E_ARGUMENT, "packed host blobs: the buffer is not pinned host memory ($CORTO_HIP_CHECK_PINNED)"); }
			}
			if(hipMemcpyAsync(b->own_arena.p, blobs[0], bytes, hipMemcpyHostToDevice,
				ctx->stream) != hipSuccess) return fail(CRTHIP_E_DEVICE);
		} else {
		// (a batch that was created and not decoded yet: its upload has to be through before the image is reused)
		if(ctx->arena_upload_pending) {
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

extern "C" int crthip_batch_reset(crthip_batch *b, uint32_t nblobs, const uint8_t *const *blobs, const uint32_t *lens,
	const void *device_arena) {
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
	if(bd.reserved & ~CRTHIP_BIND_STREAM_VALUES) return CRTHIP_E_ARGUMENT;
	if((bd.reserved & CRTHIP_BIND_STREAM_VALUES) && a.codec != CRTHIP_CODEC_GENERIC) return CRTHIP_E_FORMAT;   // normals / colours have codecs of their own upstream too
	if(a.codec == CRTHIP_CODEC_NORMAL) {
		if(bd.format != CRTHIP_FMT_FLOAT && bd.format != CRTHIP_FMT_INT16) return CRTHIP_E_FORMAT;
		const uint32_t el = bd.format == CRTHIP_FMT_INT16 ? 2u : 4u;
		if(((uintptr_t)bd.buffer) % el) return CRTHIP_E_ARGUMENT;            // a float* / int16_t* (decoder.h:52-53) is aligned by its type
		if(st && (st < 3*el || st % el)) return CRTHIP_E_ARGUMENT;
		return CRTHIP_OK;
	}
	if(a.codec == CRTHIP_CODEC_COLOR) {
		// FLOAT colour output is broken upstream (color_attribute.cpp:96-110)
		if(bd.format != CRTHIP_FMT_UINT8) return CRTHIP_E_FORMAT;
		uint32_t oc = bd.out_components ? bd.out_components : 4;
		if(a.N < 1 || a.N > 4 || oc > 4 || oc < a.N) return CRTHIP_E_FORMAT;
		if(st && st < oc) return CRTHIP_E_ARGUMENT;
		return CRTHIP_OK;
	}
	if(a.N < 1 || bd.format > CRTHIP_FMT_DOUBLE) return CRTHIP_E_FORMAT;
	// the device half of a caller-supplied codec object: int32 stream values, packed (corto_hip.h: CRTHIP_BIND_STREAM_VALUES)
	if(bd.reserved & CRTHIP_BIND_STREAM_VALUES) return bd.format == CRTHIP_FMT_INT32 && !st && ((uintptr_t)bd.buffer) % 4 == 0 ? CRTHIP_OK : CRTHIP_E_ARGUMENT;
	// a packed buffer doubles as the int32 workspace and K-DELTA turns it into floats with dword / 16-byte accesses: a float* that is
	// not 4-byte aligned (never one a C++ caller's setPositions(float*) could pass) is refused, not decoded into integers
	if(((uintptr_t)bd.buffer) % (bd.format == CRTHIP_FMT_DOUBLE ? 8 : 4)) return CRTHIP_E_ARGUMENT;
	// the integer formats and DOUBLE (setAttribute(name, buffer, format): vertex_attribute.h:195-228) are upstream's in-place layouts:
	// packed only
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
		P.bind[k].stream_values = (attrs[k].reserved & CRTHIP_BIND_STREAM_VALUES) != 0;
	}
	P.index = index; P.index_u16 = index && index_format == CRTHIP_FMT_UINT16;
	b->dirty = true;
	return CRTHIP_OK;
}

extern "C" int crthip_batch_bind_all(crthip_batch *b, const crthip_attr_binding *attrs, void *const *index, const uint32_t *index_format) {
	if(!b) return fail(CRTHIP_E_ARGUMENT);
	size_t k = 0;
	for(uint32_t i = 0; i < b->blobs.size(); i++) {
		int err = crthip_batch_bind(b, i, attrs ? attrs + k : nullptr, index ? index[i] : nullptr, index_format ? index_format[i] :
			CRTHIP_FMT_UINT32);
		if(err) return err;
		k += b->blobs[i].bind.size();
	}
	return CRTHIP_OK;
}

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

static int build_and_launch_inner(crthip_batch *b) {
	// host-side cost of a decode call (crthip_batch_stats::host_*_us)
	const double t0 = now_us();
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
	// else: harvested when the context moved on (or never decoded)
	if(ctx->in_flight == b) { if(harvest(ctx) != CRTHIP_OK) return fail(CRTHIP_E_DEVICE); }
	int first = CRTHIP_OK;
	for(size_t i = 0; i < b->blobs.size(); i++) {
		if(status) status[i] = b->status[i];
		if(b->status[i] && !first) { first = b->status[i]; fail(first, std::string(crthip_strerror(first)) + " (blob " +
			std::to_string(i) + ")"); }
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

extern "C" int64_t crthip_batch_read_prediction(crthip_batch *b, uint32_t i, void *host_out, size_t cap) {
	return crthip_batch_debug_read(b, i, "prediction", host_out, cap);
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
	auto fail_all = [&](int code) {
		for(uint32_t i = 0; i < n; i++)
			if(reqs[i].status == CRTHIP_OK) { reqs[i].status = code; reqs[i].nout = 0; }
		return code;
	};
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
	std::vector<crthip_attr_binding> dev; std::vector<size_t> dev_first(m, 0); std::vector<void *> dindex(m, nullptr);
		std::vector<uint32_t> ifmt(m, CRTHIP_FMT_UINT32);
	struct Piece { uint32_t req, slot; size_t off, bytes; void *host; };
	std::vector<Piece> pieces;
	std::vector<std::pair<uint32_t, size_t>> pred_piece;            // (blob of the batch, offset in the output block)
	size_t total = 0;
	for(uint32_t k = 0; k < m; k++) {
		HostDecodeReq &r = reqs[who[k]];
		const BlobLayout &L = b->blobs[k].L;
		const uint32_t nvert = L.h.nvert, nface = L.h.nface;
		const size_t na = L.h.attrs.size();
		dev_first[k] = dev.size();
		for(size_t a = 0; a < na; a++) {
			// attrs == NULL: nothing bound (an index-only decode); host buffers have upstream's packed layouts - a stride is refused, not
			// ignored
			crthip_attr_binding d;
			if(r.attrs) d = r.attrs[a]; else { d.buffer = nullptr; d.format = CRTHIP_FMT_FLOAT; d.out_components = 0; d.stride = 0; d.reserved = 0; }
			if(d.buffer && d.stride) { r.status = fail(CRTHIP_E_ARGUMENT,
				"crthip_decode_host: host buffers are tightly packed (stride must be 0)"); d.buffer = nullptr; }
			d.stride = 0; d.reserved &= CRTHIP_BIND_STREAM_VALUES;
			if(d.buffer && r.status == CRTHIP_OK) {
				const AttrHeader &A = L.h.attrs[a];
				size_t bytes;
				if(A.codec == CRTHIP_CODEC_NORMAL) bytes = (size_t)nvert*3*(d.format == CRTHIP_FMT_INT16 ? 2 : 4);
				else if(A.codec == CRTHIP_CODEC_COLOR) bytes = (size_t)nvert*(d.out_components ? d.out_components : 4);
				// (the decode works in int32 / int64 records whatever the output format: DESIGN.md 1)
				else bytes = (size_t)nvert*A.N*generic_work_bytes(d.format);
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
		if(r.prediction && nface && r.status == CRTHIP_OK) {         // the prediction triples: copied out of the scratch block behind the kernels (below)
			pieces.push_back(Piece{who[k], CRTHIP_MAX_ATTRS + 1, total, (size_t)nvert*12, r.prediction});
			pred_piece.push_back({k, total});
			total += ((size_t)nvert*12 + 15) & ~(size_t)15;
		}
	}
	if(ctx->host_out.reserve(total + 16) != CRTHIP_OK || ctx->host_pin.reserve(total +
		16) != CRTHIP_OK) return fail_all(fail(CRTHIP_E_NOMEM));
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
	for(auto &pp : pred_piece) {
		const BlobPlan &P = b->blobs[pp.first];
		if(!err && P.dbg_pred != ~0ull && hipMemcpyAsync(dbase + pp.second, (const uint8_t *)ctx->scratch.p + P.dbg_pred, (size_t)P.L.h.nvert*12,
			hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) err = fail(CRTHIP_E_DEVICE);
	}
	if(!err && total && hipMemcpyAsync(hbase, dbase, total, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) err = fail(CRTHIP_E_DEVICE);
	std::vector<int32_t> st(m, 0);
	// (waits for the kernels: the event behind them; keeps the context consistent on error)
	const int serr = crthip_batch_sync(b, st.data());
	if(hipStreamSynchronize(ctx->stream) != hipSuccess && !err) err = fail(CRTHIP_E_DEVICE);   // ... and this for the copy behind them
	if(!err && (serr == CRTHIP_E_DEVICE || serr == CRTHIP_E_NOMEM)) err = serr;
	for(uint32_t k = 0; k < m; k++) { HostDecodeReq &r = reqs[who[k]]; if(r.status != CRTHIP_OK) continue; r.status = err ? err : st[k]; }
	for(const Piece &p : pieces) {
		HostDecodeReq &r = reqs[p.req];
		if(r.status != CRTHIP_OK) continue;
		if(copy_out) memcpy(p.host, hbase + p.off, p.bytes);
		else if(r.nout < CRTHIP_MAX_ATTRS + 2) { r.out_src[r.nout] = hbase + p.off; r.out_dst[r.nout] = p.host; r.out_bytes[r.nout] =
			p.bytes; r.nout++; }
	}
	if(err) return err;
	// (the first failing blob's code - its message is the thread's last error; every status is in its request)
	for(uint32_t i = 0; i < n; i++) if(reqs[i].status) return reqs[i].status;
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
		auto rd = [&](const uint8_t *q) { return (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
			};
		const uint32_t size = rd(p + 1 + 2*ns), csize = rd(p + 5 + 2*ns);
		const uint8_t *dblk = (const uint8_t *)device_blocks + block_offset[i];
		uint8_t *dst = (uint8_t *)device_out + out_offset[i];
		if(size == 0) continue;
		if(ns == 1) { fills.push_back(FillJob{dst, size, p[1]}); continue; }
		if(ns == 0 || csize == 0) return fail(CRTHIP_E_TRUNCATED);
		TunStream t{};
		t.src = dblk + 9 + 2*ns; t.dst = dst; t.probs = dblk + 1; t.csize = csize; t.size = size; t.nsym = ns; t.table =
			(uint32_t)tun.size();
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
	const uint64_t o_tun = cv.take(tun.size()*sizeof(TunStream) + 16, 16), o_cs = cv.take(chunk_stream.size()*4 + 16, 16), o_fill =
		cv.take(fills.size()*sizeof(FillJob) + 16, 16);
	const uint64_t total = cv.take(0);
	if(ctx->scratch.reserve(total + 256) != CRTHIP_OK || ctx->staging.reserve(total - o_jobs +
		256) != CRTHIP_OK) return fail(CRTHIP_E_NOMEM);
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
			LT.begin("tunstall_chunk_sums"); hipLaunchKernelGGL(k_tun_chunk_sums, dim3(chunks), dim3(256), 0, st, dt, dcs, chunks, tables,
				part, 0u); LT.end();
			if(scanned) { LT.begin("tunstall_stream_scan"); hipLaunchKernelGGL(k_tun_stream_scan, dim3(ntun), dim3(256), 0, st, dt, ntun,
				part); LT.end(); }
		// (up to 256 chunks a stream: every decode wave adds up the sums in front of it itself)
		}
		LT.begin("tunstall_decode");
		if(multi) { if(launch_tun_decode_staged(st, dt, dcs, chunks, tables, part, scanned ? 0u : 1u)) return fail(CRTHIP_E_DEVICE); }
		else hipLaunchKernelGGL(k_tun_decode, dim3(chunks), dim3(256), 0, st, dt, dcs, chunks, tables, part, 0u);
		LT.end();
	}
	if(!fills.empty()) { LT.begin("fill"); hipLaunchKernelGGL(k_fill, dim3((uint32_t)fills.size()), dim3(256), 0, st, (FillJob *)(base +
		o_fill), (uint32_t)fills.size()); LT.end(); }
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
