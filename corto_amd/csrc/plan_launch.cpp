// plan_launch.cpp — Planner::upload / launch / account: the blocks reserved and the arrays staged, the kernels enqueued in the order of
// crt::Decoder::decodeMesh / decodePointCloud (src/decoder.cpp:133-196), crthip_batch_stats filled.
#include "batch_internal.h"

int Planner::upload() {
	// ---- reserve device + pinned memory; one batch in flight per context ----
	// one batch in flight per context: the previous one's status is kept in its object
	if(harvest(ctx) != CRTHIP_OK) return fail(CRTHIP_E_DEVICE);
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
		if(!(t.pad & 1u)) t.faces = R(t.faces);
		t.pad &= TOPO_PAD_PROGRESS;
		t.pred = (uint32_t *)R(t.pred); t.front_a = (uint4 *)R(t.front_a); t.front_b = (uint2 *)R(t.front_b);
		t.order = (uint32_t *)R(t.order); t.delayed = (uint32_t *)R(t.delayed); t.status = (int32_t *)R(t.status); t.flags =
			(int32_t *)R(t.flags);
	}
	for(auto &u : pl.unpack.v) {
		u.logs = R(u.logs);
		if(!(u.out_u8 & 0x80)) u.out = R(u.out);
		u.out_u8 &= 0x7F;
	}
	for(auto &d : pl.delta.v) { if(!(d.pad[0] & 1)) d.values = R(d.values); d.pad[0] >>= 1; d.pred = (const uint32_t *)R(d.pred);
		if(d.fired) d.fired = R(d.fired); d.flags = (int32_t *)R(d.flags); }
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
	// (after the harvest above: the previous batch's words have been read)
	memset(ctx->status_host.p, 0, (size_t)nblobs*16);
	auto put = [&](auto &arr) { if(!arr.v.empty()) memcpy(stage + (arr.dev_off - pl.jobs_begin), arr.v.data(),
		arr.v.size()*sizeof(arr.v[0])); };
	put(pl.tun); put(pl.tun_dict); put(pl.tun_chunk_stream); put(pl.tun_group_ids); put(pl.tun_groups); put(pl.fill); put(pl.topo);
		put(pl.aux_u32); put(pl.topo_lds_ids); put(pl.topo_big_ids); put(pl.topo_glob_ids); put(pl.unpack); put(pl.unpack_chunk_job);
		put(pl.unpack_wave_ids);
	put(pl.delta); put(pl.delta_groups); put(pl.cloud); put(pl.cloud_chunk_job); put(pl.normal); put(pl.nv_block_job);
		put(pl.nv_block_first);
	put(pl.nf_block_job); put(pl.nf_block_first); put(pl.normal_fused_ids); put(pl.dequant); put(pl.dequant_block_job);

	return CRTHIP_OK;
}

int Planner::launch() {
	hipStream_t st = ctx->stream;
	ctx->timer.reset();
	Launch LT{ctx};
	if(pl.jobs_bytes) HIP_TRY(hipMemcpyAsync(base + pl.jobs_begin, stage, pl.jobs_bytes, hipMemcpyHostToDevice, st));
	if(pl.zero_end > pl.zero_begin) HIP_TRY(hipMemsetAsync(base + pl.zero_begin, 0, pl.zero_end - pl.zero_begin, st));
	for(uint32_t i = 0; i < nblobs; i++)                                         // the progress words of big meshes (a handful a batch at most)
		if(bs[i].progress != ~0ull) HIP_TRY(hipMemsetAsync(base + bs[i].progress, 0, TOPO_PROGRESS_BYTES, st));

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
			// distinct tables first, then every stream decodes from its (shared) dictionary
			if(share) {
				const uint32_t d0 = has_clers ? 0u : clers_dict, d1 = has_attrs ? ndict : clers_dict;
				// (an alphabet of more than 64 symbols builds its words in LDS: tun_tables.h)
				uint32_t big = 0;
				for(uint32_t d = d0; d < d1; d++) if(pl.tun_dict.v[d].nsym > 64) big = TUN_TABLE_BYTES;
				LT.begin("tunstall_tables", s); hipLaunchKernelGGL(k_tun_tables, dim3(d1 - d0), dim3(64), big, s, D(pl.tun_dict) + d0,
					d1 - d0, tables); LT.end();
				const uint32_t g0 = has_clers ? 0u : pl.clers_groups, g1 = has_attrs ? (uint32_t)pl.tun_groups.v.size() : pl.clers_groups;
				LT.begin("tunstall_stream", s); hipLaunchKernelGGL(k_tun_stream_grouped, dim3(g1 - g0), dim3(256), 0, s, D(pl.tun),
					D(pl.tun_group_ids), D(pl.tun_groups) + g0, g1 - g0, tables); LT.end();
			} else {                                           // dictionary + decode in one kernel
				LT.begin("tunstall_stream", s); hipLaunchKernelGGL(k_tun_stream, dim3(t1 - t0), dim3(64), 0, s, D(pl.tun) + t0, t1 - t0);
					LT.end();
			}
		}
		if(f1 > f0) { LT.begin("fill", s); hipLaunchKernelGGL(k_fill, dim3(f1 - f0), dim3(256), 0, s, D(pl.fill) + f0, f1 - f0); LT.end(); }
	};
	auto unpack = [&](hipStream_t s) {
		const uint32_t nuw = (uint32_t)pl.unpack_wave_ids.v.size();
		if(nuw) { LT.begin("unpack_wave", s); hipLaunchKernelGGL(k_unpack_wave, dim3(xcd_grid(nuw)), dim3(64), 0, s, D(pl.unpack),
			D(pl.unpack_wave_ids), nuw); LT.end(); }
		if(!unpack_chunks) return;
		LT.begin("unpack_extract", s); hipLaunchKernelGGL(k_unpack_extract, dim3(unpack_chunks), dim3(256), 0, s, D(pl.unpack),
			D(pl.unpack_chunk_job), unpack_chunks, unpack_partial); LT.end();
	};
	// K-DELTA's jobs by class (sorted that way: 0 / 1 = too big for the LDS records, 2 = the groups of k_delta_lds16)
	uint32_t ncls[3] = {0, 0, 0};
	for(auto &d : pl.delta.v) ncls[delta_class(d, wide)]++;
	bool tiles_launched = false;
	// the big ones in tiles of 1 024 vertices (k_delta_tiles).  On a lone context this goes to the SECOND stream, behind the attribute streams' bit-unpack
	// and BESIDE the automaton, whose progress word the tiles wait for: a single big mesh's delta inversion trails its topology instead of following it
	// (config C2: 1.5 of 4.0 ms).  The automaton is enqueued first on its own stream and waits for nothing of this kernel.
	// BESIDE the automaton only a handful of workgroups: they hold 66 KB of LDS each while they wait, and a batch of hundreds of big meshes' tiles, resident
	// first, could keep the automata (up to 156 KB a workgroup) from ever finding a CU - the tiles would wait for a progress word nobody can write.  Up to 48
	// (sixteen big meshes' three attributes) leave most of the chip free; more run behind the automaton as on a pool context.
	constexpr uint32_t TILES_BESIDE_MAX = 48;
	auto delta_tiles = [&](hipStream_t s) {
		const uint32_t nbig = ncls[0] + ncls[1];
		if(!nbig || ctx->dbg.delta_walk) return;
		if(s != st && nbig > TILES_BESIDE_MAX) return;
		LT.begin("delta_tiles", s); hipLaunchKernelGGL(k_delta_tiles, dim3(nbig), dim3(DELTA_THREADS), 0, s, D(pl.delta), nbig); LT.end();
		tiles_launched = true;
	};
	auto topology = [&]() -> int {
		if(!pl.topo_lds_ids.v.empty() || !pl.topo_big_ids.v.empty()) {
			LT.begin("topology_lds");
			if(!pl.topo_big_ids.v.empty()) { const uint32_t nj = (uint32_t)pl.topo_big_ids.v.size(); hipLaunchKernelGGL(k_topology_lds_big,
				dim3(nj), dim3(64), pl.topo_big_lds, st, D(pl.topo), D(pl.topo_big_ids), nj); }
			if(!pl.topo_lds_ids.v.empty()) { const uint32_t nj = (uint32_t)pl.topo_lds_ids.v.size(); hipLaunchKernelGGL(k_topology_lds,
				dim3(nj), dim3(64), pl.topo_lds, st, D(pl.topo), D(pl.topo_lds_ids), nj); }
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
		LT.begin("tunstall_chunk_sums"); hipLaunchKernelGGL(k_tun_chunk_sums, dim3(tun_chunks), dim3(256), 0, st, D(pl.tun),
			D(pl.tun_chunk_stream), tun_chunks, tables, tun_partial, 0u); LT.end();
		// (up to 256 chunks a stream: every decode wave adds up the sums in front of it itself; longer streams: one workgroup per stream
		// scans them)
		const bool scanned = pl.tun_max_nchunks > 256;
		if(scanned) { LT.begin("tunstall_stream_scan"); hipLaunchKernelGGL(k_tun_stream_scan, dim3(ntun), dim3(256), 0, st, D(pl.tun),
			ntun, tun_partial); LT.end(); }
		LT.begin("tunstall_decode");
		if(launch_tun_decode_staged(st, D(pl.tun), D(pl.tun_chunk_stream), tun_chunks, tables, tun_partial, scanned ? 0u : 1u)) return fail(CRTHIP_E_DEVICE);
		LT.end();
		if(nfill) { LT.begin("fill"); hipLaunchKernelGGL(k_fill, dim3(nfill), dim3(256), 0, st, D(pl.fill), nfill); LT.end(); }
		if(!ctx->single_stream && !ctx->dbg.delta_walk && ncls[0] + ncls[1] && ncls[0] + ncls[1] <= TILES_BESIDE_MAX && !pl.topo.v.empty()) {
			// a big mesh alone: every stream is decoded; the automaton (one serial chain: 2.2 of C2's 4 ms) goes on on the main stream, the attributes'
			// bit-unpack and their delta inversion in tiles on the second one, the tiles trailing the automaton's progress word
			hipStream_t s2 = ctx->stream2;
			HIP_TRY(hipEventRecord(ctx->ev_fork, st));
			HIP_TRY(hipStreamWaitEvent(s2, ctx->ev_fork, 0));
			{ int e_ = topology(); if(e_) return e_; }
			unpack(s2);
			delta_tiles(s2);
			HIP_TRY(hipEventRecord(ctx->ev_join, s2));
			HIP_TRY(hipStreamWaitEvent(st, ctx->ev_join, 0));
		} else {
			{ int e_ = topology(); if(e_) return e_; }
			unpack(st);
		}
	} else if(!pl.topo.v.empty() && (ntun > clers_tun || nfill > clers_fill || unpack_chunks || !pl.unpack_wave_ids.v.empty()) &&
		!ctx->single_stream) {
		// fork: attribute streams on stream2, CLERS + topology on the main stream
		hipStream_t s2 = ctx->stream2;
		HIP_TRY(hipEventRecord(ctx->ev_fork, st));
		HIP_TRY(hipStreamWaitEvent(s2, ctx->ev_fork, 0));
		tunstall(st, 0, clers_tun, 0, clers_chunks, 0, clers_fill);
		{ int e_ = topology(); if(e_) return e_; }
		tunstall(s2, clers_tun, ntun, clers_chunks, tun_chunks, clers_fill, nfill);
		unpack(s2);
		delta_tiles(s2);
		HIP_TRY(hipEventRecord(ctx->ev_join, s2));
		HIP_TRY(hipStreamWaitEvent(st, ctx->ev_join, 0));
	} else {
		tunstall(st, 0, ntun, 0, tun_chunks, 0, nfill);
		{ int e_ = topology(); if(e_) return e_; }
		unpack(st);
	}
	if(!pl.delta.v.empty()) {
		const uint32_t ngroups = (uint32_t)pl.delta_groups.v.size();
		// (the timer's names are the kernels that run: `delta_mesh` = the walk over HBM, `delta_lds16` = the LDS form, one workgroup a blob)
		if(ncls[0] || ncls[1]) {
			// too big for the LDS records: tiles of 1 024 vertices out of an LDS ring (k_delta_tiles, up to four components); rounds 1-5's stretch walk
			// over L2 for attributes of more components, and for everything under $CORTO_DELTA_WALK=1
			const uint32_t nbig = ncls[0] + ncls[1];
			bool many = false;
			for(uint32_t k = 0; k < nbig; k++) many = many || pl.delta.v[k].N > 4;
			if(!ctx->dbg.delta_walk && !tiles_launched) delta_tiles(st);
			if(ctx->dbg.delta_walk || many) {
				LT.begin("delta_mesh");
				const uint32_t only = ctx->dbg.delta_walk ? 0u : 1u;
				if(ncls[0]) hipLaunchKernelGGL(k_delta_mesh, dim3(ncls[0]), dim3(DELTA_THREADS), 0, st, D(pl.delta), ncls[0], only);
				if(ncls[1]) hipLaunchKernelGGL(k_delta_mesh, dim3(ncls[1]), dim3(DELTA_THREADS/2), 0, st, D(pl.delta) + ncls[0], ncls[1], only);
				LT.end();
			}
		}
		if(ngroups) { LT.begin("delta_lds16"); hipLaunchKernelGGL(k_delta_lds16, dim3(ngroups), dim3(256), pl.delta16_lds, st, D(pl.delta), D(pl.delta_groups),
			ngroups); LT.end(); }
	}
	if(cloud_chunks) {
		LT.begin("cloud_sums"); hipLaunchKernelGGL(k_cloud_sums, dim3(cloud_chunks), dim3(256), 0, st, D(pl.cloud), D(pl.cloud_chunk_job),
			cloud_chunks, cloud_partial); LT.end();
		LT.begin("scan"); hipLaunchKernelGGL(k_scan_u64, dim3(1), dim3(1024), 0, st, cloud_partial, cloud_chunks); LT.end();
		LT.begin("cloud_apply"); hipLaunchKernelGGL(k_cloud_apply, dim3(cloud_chunks), dim3(256), 0, st, D(pl.cloud),
			D(pl.cloud_chunk_job), cloud_chunks, cloud_partial); LT.end();
	}
	const uint32_t nvb = (uint32_t)pl.nv_block_job.v.size(), nfb = (uint32_t)pl.nf_block_job.v.size();
	if(!pl.normal_fused_ids.v.empty()) {
		const uint32_t nj = (uint32_t)pl.normal_fused_ids.v.size();
		LT.begin("normal_blob"); hipLaunchKernelGGL(k_normal_blob, dim3(nj), dim3(256), pl.normal_fused_lds, st, D(pl.normal),
			D(pl.normal_fused_ids), nj, pl.normal_fused_lds); LT.end();
	}
	if(pl.any_est_normal) {
		float *facen = (float *)(base + pl.facen_off);
		uint32_t *cnt = (uint32_t *)(base + pl.cnt_off), *cursor = (uint32_t *)(base + pl.cursor_off), *bnd = (uint32_t *)(base +
			pl.bnd_off);
		uint32_t *start = (uint32_t *)(base + pl.start_off), *flag = (uint32_t *)(base + pl.flag_off), *slot = (uint32_t *)(base +
			pl.slot_off);
		uint32_t *adj = (uint32_t *)(base + pl.adj_off);
		uint64_t *npart = (uint64_t *)(base + pl.nscan_partial_off);
		const uint32_t nv = pl.est_nvert, nch = (nv + CHUNK - 1)/CHUNK;
		LT.begin("normal_faces"); hipLaunchKernelGGL(k_normal_faces, dim3(nfb), dim3(256), 0, st, D(pl.normal), D(pl.nf_block_job),
			D(pl.nf_block_first), nfb, facen, cnt, bnd); LT.end();
		LT.begin("normal_scan");
		hipLaunchKernelGGL(k_u32_chunk_sums, dim3(nch), dim3(256), 0, st, cnt, nv, npart);
		hipLaunchKernelGGL(k_scan_u64, dim3(1), dim3(1024), 0, st, npart, nch);
		hipLaunchKernelGGL(k_u32_chunk_apply, dim3(nch), dim3(256), 0, st, cnt, start, nv, npart);
		LT.end();
		LT.begin("normal_fill"); hipLaunchKernelGGL(k_normal_fill, dim3(nfb), dim3(256), 0, st, D(pl.normal), D(pl.nf_block_job),
			D(pl.nf_block_first), nfb, start, cursor, adj); LT.end();
		LT.begin("normal_flags"); hipLaunchKernelGGL(k_normal_flags, dim3(nvb), dim3(256), 0, st, D(pl.normal), D(pl.nv_block_job),
			D(pl.nv_block_first), nvb, bnd, flag); LT.end();
		LT.begin("normal_scan");
		hipLaunchKernelGGL(k_u32_chunk_sums, dim3(nch), dim3(256), 0, st, flag, nv, npart);
		hipLaunchKernelGGL(k_scan_u64, dim3(1), dim3(1024), 0, st, npart, nch);
		hipLaunchKernelGGL(k_u32_chunk_apply, dim3(nch), dim3(256), 0, st, flag, slot, nv, npart);
		LT.end();
		LT.begin("normal_vertex"); hipLaunchKernelGGL(k_normal_vertex, dim3(nvb), dim3(256), 0, st, D(pl.normal), D(pl.nv_block_job),
			D(pl.nv_block_first), nvb, facen, start, cnt, adj, flag, slot); LT.end();
	}
	if(pl.any_diff_normal) { LT.begin("normal_diff"); hipLaunchKernelGGL(k_normal_diff, dim3(nvb), dim3(256), 0, st, D(pl.normal),
		D(pl.nv_block_job), D(pl.nv_block_first), nvb); LT.end(); }
	const uint32_t ndq = (uint32_t)pl.dequant_block_job.v.size();
	if(ndq) { LT.begin("dequantize"); hipLaunchKernelGGL(k_dequant, dim3(ndq), dim3(256), 0, st, D(pl.dequant), D(pl.dequant_block_job),
		ndq); LT.end(); }

	// (status: written by the kernels straight into the pinned block)
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(ctx->ev_done, st));
	ctx->done_covers_seq = ctx->upload_seq;

	return CRTHIP_OK;
}

void Planner::account() {
	const uint32_t ntun = (uint32_t)pl.tun.v.size();
	// stats
	b->stats.tunstall_in = stat_tin; b->stats.tunstall_out = stat_tout; b->stats.tunstall_tables = stat_tt; b->stats.tunstall_streams =
		ntun; b->stats.tunstall_dictionaries = (uint32_t)stat_dicts;
	b->stats.scratch_bytes = pl.total;
	b->stats.descriptor_bytes = (uint32_t)pl.jobs_bytes;
	b->stats.int16_streams = 0;
	for(const UnpackJob &u : pl.unpack.v) b->stats.int16_streams += u.out_u8 == 2;
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
