// plan_jobs.cpp — Planner::jobs: pass 2 - the job descriptors of every stage, with scratch-relative pseudo pointers.  See batch_internal.h.
#include "batch_internal.h"

int Planner::jobs() {
	// ---- pass 2: job structs with offsets stored in pointer fields (rebased after the block is reserved) ----
	// To keep one pass, pointers are built as (uint8_t*)offset and fixed up by adding the scratch base.
	// four words a blob: status | automaton flags (bit 0: redone on the HBM front) | K-DELTA: {an attribute's values left int16, an
	// attribute took the walk}
	// the block is about to move: the batch in flight writes to it
	if((size_t)nblobs*16 + 16 > ctx->status_host.cap && harvest(ctx) != CRTHIP_OK) return fail(CRTHIP_E_DEVICE);
	if(ctx->status_host.reserve((size_t)nblobs*16 + 16) != CRTHIP_OK) return fail(CRTHIP_E_NOMEM);
	hs_base = (int32_t *)ctx->status_host.p;

	// the dictionary (TunTable slot) of a stream: a new one, or the one an earlier stream of this launch group with the same table got
	if(ctx->dict_slots.size() != 8192) ctx->dict_slots.assign(8192, 0u);
	for(uint32_t u : ctx->dict_used) ctx->dict_slots[u] = 0;
	ctx->dict_used.clear(); ctx->dict_keys.clear();
	std::vector<uint32_t> &dict_ids = ctx->dict_ids; dict_ids.clear();
	// first dictionary of the current group (CLERS streams / attribute streams)
	uint32_t dict_group0 = 0;
	auto dict_of = [&](const StreamRef &s, const TunStream &t) -> uint32_t {
		const uint32_t fresh = (uint32_t)pl.tun_dict.v.size();
		auto make = [&]() { TunStream d = t; d.table = fresh; d.dict = fresh; d.nchunks = 1; pl.tun_dict.v.push_back(d); return fresh; };
		// (0 / 2: one dictionary per stream, whatever repeats)
		if(s.nsym > 16 || fresh - dict_group0 >= 4096 || ctx->dbg.tun_share == 0 || ctx->dbg.tun_share == 2) return make();
		uint64_t h = 0x9E3779B97F4A7C15ull ^ s.nsym;
		for(uint32_t k = 0; k < 2*s.nsym; k += 8) { uint64_t w; memcpy(&w, s.probs16 + k, 8); h = (h ^ w)*0xFF51AFD7ED558CCDull;
			h ^= h >> 32; }
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
			if(S.progress != ~0ull) t.pad |= TOPO_PAD_PROGRESS;
			{
				// every mesh takes the LDS path; a lone big mesh may use most of a CU's LDS, a batch keeps its blobs small
				uint32_t ring, pool, symwin;
				uint32_t scale = ctx->topo_scale, pool_q8 = ctx->topo_pool_q8, need;
				// as much of what the context has learnt as fits a CU
				for(;;) {
					topo_lds_geometry(nface, L.clers.size, 4096, scale, pool_q8, nblobs >= 32 ? 4u : 8u, topo_boundary_estimate(nvert,
						nface), ring, pool, symwin, ctx->topo_pool_cap);
					// every delayed edge is a pool record: same capacity
					need = topo_lds_bytes(ring, pool, pool, symwin);
#ifdef CORTO_TOPO_STAMPS
					if(need <= 32768 && L.clers.size < 8190) need = 65536;              // (the dispatch trace: k_mesh.hip TOPO_ASM_STAMP)
#endif
					if(need <= TOPO_LDS_MAX || (scale == 1 && pool_q8 == 8)) break;
					if(pool_q8 > 8 && (pool > ring || scale == 1)) pool_q8 = std::max(8u, pool_q8/2); else scale >>= 1;
				}
				if(need <= TOPO_LDS_MAX) {
					t.lds_ring = ring; t.lds_pool = pool; t.lds_delayed_cap = pool; t.lds_symwin = symwin;
					// (split into two launches below)
					pl.topo_lds_ids.v.push_back((uint32_t)pl.topo.v.size()); pl.topo_need.push_back(need);
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
				if(mesh && L.h.attrs[k].codec == CRTHIP_CODEC_NORMAL && P.bind[k].buffer && (L.attrs[k].normal_prediction == 1 ||
					L.attrs[k].normal_prediction == 2)) readers++;
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
			// one wave per stream, no look-back; its bit cursors are 32-bit (k_stream.hip)
			const bool by_wave = attr_logs <= UNPACK_WAVE_MAX_LOGS && as.bits.nwords < (1u << 26) && !ctx->dbg.unpack_chunked;
			const uint32_t attr_first = (uint32_t)pl.unpack.v.size();
			auto push_unpack = [&](const StreamRef &s, const uint8_t *logs, void *out, bool out_real, uint8_t mode, uint16_t fields,
				uint16_t stride, uint16_t comp, uint8_t u8) {
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
			// K-BIT hands its values on as int16 where that is PROVEN to hold them: the attribute's log streams are Tunstall-coded and no width in their tables
			// exceeds 16 bits (decodeArray: v in [-2^(d-1), 2^(d-1)); decodeValues folds the sign the other way: |v| < 2^d, 15 bits), the consumer reads
			// halfwords (k_delta_lds16 on 16-bit records, k_normal_blob), and the stream goes a wave a stream.  Half the bytes of that round trip through HBM
			auto widths_fit = [&](const StreamRef &s, uint32_t bits) { return s.mode != STREAM_RAW && s.max_sym <= bits; };
			bool hand_i16 = false;
			if(mesh && by_wave && !wide && !ctx->dbg.values_i32 && nvert > 1) {
				if(a.codec == CRTHIP_CODEC_NORMAL) hand_i16 = as.normal_prediction != 0 && normal_fused(nvert, nface) && widths_fit(as.logs[0], 16);
				else if(a.codec != CRTHIP_CODEC_COLOR && !bd.stream_values) {
					DeltaJob probe{};
					probe.nvert = nvert; probe.N = a.N;
					hand_i16 = delta_class(probe, wide) >= 2;
					if(a.strategy & CRTHIP_CORRELATED) hand_i16 = hand_i16 && widths_fit(as.logs[0], 16);
					else for(uint32_t c = 0; c < a.N && hand_i16; c++) hand_i16 = widths_fit(as.logs[c], 15);
				}
			}
			const uint8_t val_fmt = hand_i16 ? 2 : 0;
			std::vector<const uint8_t *> &logs = ctx->plan_logs;
			logs.assign(as.logs.size(), nullptr);
			for(size_t j = 0; j < as.logs.size(); j++) logs[j] = add_stream(as.logs[j], A.sym[j], bo);

			void *values = nullptr; bool values_real = false; uint8_t is_u8 = 0; uint32_t N = a.N; bool para = false; bool do_delta = true;
			if(a.codec == CRTHIP_CODEC_NORMAL) {
				push_unpack(as.logs[0], logs[0], SP(A.diffs), false, 0, 2, 2, 0, val_fmt);
				// a (malformed) stream with fewer diffs than vertices: upstream's vector is zero-filled behind them
				// (normal_attribute.cpp:180-184)
				// (only DIFF reads all nvert entries; the other predictions stop at ndiffs)
				if(as.normal_prediction == 0 && as.logs[0].size < nvert) pl.fill.v.push_back(FillJob{SP(A.diffs +
					(uint64_t)as.logs[0].size*8), (nvert - as.logs[0].size)*8u, 0u});
				values = SP(A.diffs); N = 2; para = false;
				do_delta = as.normal_prediction == 0;                     // DIFF only (normal_attribute.cpp:190-191)
			} else if(a.codec == CRTHIP_CODEC_COLOR) {
				for(uint32_t c = 0; c < a.N; c++) push_unpack(as.logs[c], logs[c], SP(A.color), false, 1, 1, (uint16_t)a.N, (uint16_t)c, 1);
				values = SP(A.color); is_u8 = 1; para = (a.strategy & CRTHIP_PARALLEL) != 0;
			} else {
				// packed output: the caller's buffer is the int32 workspace (like upstream, vertex_attribute.h:190-193); with a stride, or
				// as
				// DOUBLE (eight bytes a value: upstream widens in place, front to back): scratch
				const bool in_scratch = bd.stride || bd.format == CRTHIP_FMT_DOUBLE;
				void *work = in_scratch ? (void *)SP(A.vals) : bd.buffer;
				const bool work_real = !in_scratch;
				if(a.strategy & CRTHIP_CORRELATED) push_unpack(as.logs[0], logs[0], work, work_real, 0, (uint16_t)a.N, (uint16_t)a.N, 0, val_fmt);
				else for(uint32_t c = 0; c < a.N; c++) push_unpack(as.logs[c], logs[c], work, work_real, 1, 1, (uint16_t)a.N, (uint16_t)c,
					val_fmt);
				values = work; values_real = work_real; para = (a.strategy & CRTHIP_PARALLEL) != 0;
				// the device half of a caller-supplied codec object (CRTHIP_BIND_STREAM_VALUES): the stream's int32 values stay as they are -
				// GenericAttr<int>::decode's result (vertex_attribute.h:151-156); deltaDecode / dequantize are the caller's, on the host
				if(bd.stream_values) continue;
			}
			bool dequantised = false;                                // by K-DELTA or by the fused normal kernel: no k_dequant job
			if(do_delta && nvert > 1) {
				if(mesh) {
					DeltaJob d{};
					d.values = values; d.pred = (const uint32_t *)SP(S.pred); d.nvert = nvert; d.N = N;
					// pad[1]: 32-bit records in LDS (k_delta_lds16)
					d.parallelogram = para; d.is_u8 = is_u8; d.pad[0] = (uint8_t)((values_real ? 1 : 0) | (hand_i16 ? 2 : 0)); d.pad[1] = wide; d.pad2[0] = ctx->dbg.delta_rounds ?
						1u : 0u;                                                // (pad[0] bit 0: `values` is a real pointer - host only; bit 1: int16 raw deltas - what the device sees)
					// (k_delta_mesh's flags, else - k_delta_tiles - the automaton's progress word)
					d.fired = A.fired != ~0ull ? SP(A.fired) : S.progress != ~0ull ? SP(S.progress) : nullptr;
					d.flags = HS(2ull*nblobs + 2ull*i);
					d.pad2[1] = 2u*nblobs + i;                              // (words from the blob's status to its flags: k_delta_tiles reports a progress word that never came)
					if(a.codec != CRTHIP_CODEC_NORMAL && delta_class(d, wide) >= 2) {
						if(a.codec == CRTHIP_CODEC_COLOR) {
							d.deq = 2; d.out = bd.buffer; d.out_components = bd.out_components; d.out_stride = bd.stride;
							for(int c = 0; c < 4; c++) d.qc[c] = as.qc[c];
							dequantised = true;
						} else if(!bd.stride && bd.format == CRTHIP_FMT_FLOAT && !((int)k == pos_k && pos_ints_needed)) { d.deq = 1; d.q =
							a.q; dequantised = true; }
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
						// (a position under a caller-supplied codec holds stream values, not positions: upstream throws there too, normal_attribute.cpp:210-213)
						const bool pos_ok = pos_k >= 0 && L.h.attrs[pos_k].codec == CRTHIP_CODEC_GENERIC && L.h.attrs[pos_k].N == 3 &&
							P.bind[pos_k].buffer && !P.bind[pos_k].stream_values;
						if(!pos_ok) { P.host_status = CRTHIP_E_NORMAL_NEEDS_POSITION; continue; }
						// the integer positions: in the caller's packed buffer, or in scratch
						const bool pos_scratch = P.bind[pos_k].stride != 0 || P.bind[pos_k].format == CRTHIP_FMT_DOUBLE;
						n.position = pos_scratch ? (const int32_t *)SP(S.attr[pos_k].vals) : (const int32_t *)P.bind[pos_k].buffer;
						n.faces = P.index ? P.index : (void *)SP(S.faces);
						// bit7: faces is a real pointer, bit6: position is a scratch offset (both cleared at fixup)
						n.faces_u16 = (uint8_t)((P.index ? P.index_u16 : 0) | (P.index ? 0x80 : 0) | (pos_scratch ? 0x40 : 0));
						if(normal_fused(nvert, nface)) {
							n.fused = 1; n.diffs_i16 = hand_i16;
							n.fn_scratch = A.facen != ~0ull ? (float *)SP(A.facen) : nullptr;
							if(pos_by_normal) { n.pos_out = P.bind[pos_k].buffer; n.pos_stride = P.bind[pos_k].stride ?
								P.bind[pos_k].stride : 12u; n.pos_q = L.h.attrs[pos_k].q; }
							pl.normal_fused_ids.v.push_back((uint32_t)pl.normal.v.size());
							pl.normal_fused_lds = std::max(pl.normal_fused_lds, normal_blob_lds_fn(nvert, nface) <= ctx->normal_fn_max ?
								normal_blob_lds_fn(nvert, nface) : normal_blob_lds(nvert, nface));
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

