// plan_carve.cpp — Planner::carve: pass 1 over the batch - the scratch block laid out (sizes and offsets only).  See batch_internal.h.
#include "batch_internal.h"

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
			if(L.h.attrs[k].codec == CRTHIP_CODEC_NORMAL && P.bind[k].buffer && L.attrs[k].normal_prediction != 0 &&
				!normal_fused(L.h.nvert, L.h.nface)) { est_v += L.h.nvert; est_f += L.h.nface; }
	}
	// Per-blob status and flags live in the context's PINNED HOST block: the kernels write there directly (only a failing or
	// redone blob does), the host zeroes it before the launch and reads it after the sync - no memset kernel in front of a step
	// and no copy kernel behind it (each stretched to 50-100 us with eight batches in flight).  Prediction triples are not cleared
	// either: the automaton writes every vertex it makes and clears the ones it never reached itself (k_mesh.hip).
	(void)cv.take(256);                                                  // (offset 0 is a null pseudo pointer: the first blob's progress word must not sit there)
	for(uint32_t i = 0; i < nblobs; i++) {
		const BlobLayout &L = b->blobs[i].L;
		if(L.h.nface > 0) bs[i].pred = cv.take((uint64_t)L.h.nvert*12 + TOPO_PROGRESS_BYTES, 16) + TOPO_PROGRESS_BYTES;   // (the automaton's progress word in front: device_plan.h)
	}
	// zeroed by a memset, present only for big meshes: the counters of the
	pl.zero_begin = cv.take(0);
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
			if(delta_class(probe, wide) <= 1) {
				// k_delta_tiles: one progress word a blob, kept by the automaton; k_delta_mesh (more than four components, or $CORTO_DELTA_WALK): fired
				// flags (zeroed) + the list of stretch starts behind them
				bs[i].progress = bs[i].pred - TOPO_PROGRESS_BYTES;           // (zeroed on its own: Planner::upload)
				if(ctx->dbg.delta_walk || probe.N > 4)
					bs[i].attr[k].fired = cv.take((((uint64_t)L.h.nvert + 15) & ~15ull) + 4ull*L.h.nvert + 16, 16);
			}
		}
	}
	pl.zero_end = cv.take(0);
	// look-back state words of the bit-unpack chunks (k_unpack_extract): one per
	unpack_state_words = 1;
	// 1 024 logs of every bound stream, + a spare; uploaded as zeros with the jobs
	for(uint32_t i = 0; i < nblobs; i++) {
		const BlobPlan &P = b->blobs[i];
		for(size_t k = 0; k < P.L.attrs.size(); k++) if(P.bind[k].buffer) for(const StreamRef &lg :
			P.L.attrs[k].logs) unpack_state_words += ((uint64_t)lg.size + CHUNK - 1)/CHUNK;
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
			if(a.codec != CRTHIP_CODEC_COLOR && a.codec != CRTHIP_CODEC_NORMAL && (bd.stride || bd.format == CRTHIP_FMT_DOUBLE)) A.vals =
				cv.take((uint64_t)L.h.nvert*a.N*4 + 16, 16);
			if(a.codec == CRTHIP_CODEC_NORMAL) {
				A.diffs = cv.take((uint64_t)L.h.nvert*8 + 16, 16);
				if(mesh && L.attrs[k].normal_prediction != 0 && normal_fused(L.h.nvert, L.h.nface) && normal_blob_lds_fn(L.h.nvert,
					L.h.nface) > ctx->normal_fn_max)
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

