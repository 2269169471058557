// plan_group.cpp — Planner::group: streams sorted by dictionary, attributes into K-DELTA workgroups, the automata into launch classes, the
// job arrays placed.
#include "batch_internal.h"

void Planner::group() {
	// streams of a launch that share dictionaries: sorted by dictionary (counting sort), cut into groups of one dictionary each
	const uint32_t ntun_all = (uint32_t)pl.tun.v.size(), ndict_all = (uint32_t)pl.tun_dict.v.size();
	// dictionaries by one kernel (K-TAB, 6 KB of LDS per wave), decodes by another (10 KB for a few us), instead of both in one wave per
	// stream
	// (16 KB for ~37 us): a batch of many streams is bound by LDS.time, so the split pays even when NO two streams share a
	// table
	auto shares = [&](uint32_t nstreams, uint32_t ndicts) { (void)ndicts; return ctx->dbg.tun_share == 0 ? false :
		ctx->dbg.tun_share == 1 ? true : nstreams >= 64; };
	share_clers = !pl.tun_multi_chunk && shares(clers_tun, clers_dict); share_attrs = !pl.tun_multi_chunk && shares(ntun_all - clers_tun,
		ndict_all - clers_dict);
	{
		std::vector<uint32_t> &cnt = ctx->dict_count;
		auto group_range = [&](uint32_t t0, uint32_t t1, uint32_t d0, uint32_t d1) {
			if(t1 <= t0) return;
			cnt.assign((size_t)(d1 - d0) + 1, 0u);
			for(uint32_t t = t0; t < t1; t++) cnt[pl.tun.v[t].dict - d0 + 1]++;
			for(uint32_t d = 1; d <= d1 - d0; d++) cnt[d] += cnt[d - 1];
			const uint32_t base = (uint32_t)pl.tun_group_ids.v.size();
			pl.tun_group_ids.v.resize((size_t)base + (t1 - t0));
			// (cnt[d] .. cnt[d + 1]: the dictionary's slots; groups before the fill moves the cursors)
			for(uint32_t d = 0; d < d1 - d0; d++)
				for(uint32_t k = cnt[d]; k < cnt[d + 1]; k += TUN_GROUP_MAX) pl.tun_groups.v.push_back(TunGroup{base + k,
					std::min(TUN_GROUP_MAX, cnt[d + 1] - k)});
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
	unpack_state_words = (uint64_t)unpack_chunks + 1;                    // (plan_carve's count was every bound stream's; the bit blocks that go a wave a stream keep no state)
	pl.unpack_partial_off = cv.take(unpack_state_words*8, 16);           // (first thing in the uploaded block: zeros)
	auto place = [&](auto &arr) { arr.dev_off = cv.take(arr.v.size()*sizeof(arr.v[0]) + 16, 16); };
	// the LDS automata go up in ONE launch whose LDS request is the largest of theirs - unless some ask for much more than the others (a
	// 66K-triangle
	// mesh among 4K-triangle blobs): those get a launch of their own, so that a big mesh does not cost the small ones their occupancy. 
	// "Much more": beyond
	// 32 KB AND beyond twice the smallest request (round 5: a batch of Delaunay discs asks for 20-40 KB a blob, and cut at 32 KB it became
	// two launches
	// one after the other, each as long as its slowest blob - 1.07 ms instead of 0.57)
	if(!pl.topo_lds_ids.v.empty()) {
		uint32_t lo = 0xFFFFFFFFu;
		for(uint32_t nd : pl.topo_need) lo = std::min(lo, nd);
		const uint32_t cut = std::max(32u*1024u, 2u*lo);
		std::vector<uint32_t> small_ids;
		for(size_t k = 0; k < pl.topo_lds_ids.v.size(); k++) {
			const uint32_t nd = pl.topo_need[k], id = pl.topo_lds_ids.v[k];
			// (a blob whose automaton keeps a progress word - an attribute goes through k_delta_tiles - runs in the big launch: k_topology_lds_big is the kernel that does)
			if(nd <= cut && !(pl.topo.v[id].pad & TOPO_PAD_PROGRESS)) { small_ids.push_back(id); pl.topo_lds = std::max(pl.topo_lds, nd); }
			else { pl.topo_big_ids.v.push_back(id); pl.topo_big_lds = std::max(pl.topo_big_lds, nd); }
		}
		pl.topo_lds_ids.v.swap(small_ids);
	}
	place(pl.tun); place(pl.tun_dict); place(pl.tun_chunk_stream); place(pl.tun_group_ids); place(pl.tun_groups); place(pl.fill);
		place(pl.topo); place(pl.aux_u32); place(pl.topo_lds_ids); place(pl.topo_big_ids); place(pl.topo_glob_ids); place(pl.unpack);
		place(pl.unpack_chunk_job); place(pl.unpack_wave_ids);
	// large attributes first: they are launched with four times the threads of the small ones (k_delta_mesh)
	std::stable_partition(pl.delta.v.begin(), pl.delta.v.end(), [wide_ = wide](const DeltaJob &d) { return delta_class(d, wide_) == 0; });
	std::stable_partition(pl.delta.v.begin(), pl.delta.v.end(), [wide_ = wide](const DeltaJob &d) { return delta_class(d, wide_) <= 1; });
	// attributes of one blob that fit LDS together share a workgroup and the prediction graph: consecutive jobs of class 2 with the same
	{
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
	place(pl.delta); place(pl.delta_groups); place(pl.cloud); place(pl.cloud_chunk_job); place(pl.normal); place(pl.nv_block_job);
		place(pl.nv_block_first);
	place(pl.nf_block_job); place(pl.nf_block_first); place(pl.normal_fused_ids); place(pl.dequant); place(pl.dequant_block_job);
	pl.jobs_bytes = cv.take(0) - pl.jobs_begin;
	pl.total = cv.take(0);

}

