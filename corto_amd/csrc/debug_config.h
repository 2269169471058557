// debug_config.h — every environment switch of the decode path in one struct, filled when a context is made
// (crthip_ctx_create) and never consulted anywhere else: no getenv in the planner, no function-local statics.
//
// Two kinds.  Deployment settings (documented in INTEGRATION.md): how many contexts the crt::Decoder facade keeps, which device it
// uses, whether streams of a batch share dictionaries.  Experiment switches (CORTO_EXP_* / CORTO_TUN_*): alternative kernel paths kept
// for A/B measurements (tools/) - each one is bit-exact and has a parity test that runs the GPU suite's cases through it
// (tests/test_gpu_parity.py::test_experiment_switches_are_bit_exact); none changes results, only which kernels produce them.
// Besides these: decoder_facade.cpp reads $CORTO_HIP_CONTEXTS / $CORTO_HIP_DEVICE / $CORTO_HIP_COMBINE_US / $CORTO_HIP_LEADERS once, pool.cpp reads ROCm's own
// $GPU_MAX_HW_QUEUES to size itself.
#pragma once
#include <cstdint>
#include <cstdlib>

#include "device_plan.h"

namespace corto_hip {

struct DebugConfig {
	// deployment
	int tun_share = -1;             // $CORTO_TUN_SHARE: unset: launches of 64+ streams make every DISTINCT probability table's dictionary once (K-TAB) and decode
	                                // the streams in groups of one dictionary; 2: the same two kernels but one dictionary PER STREAM, whatever repeats (what a
	                                // batch of unrelated meshes looks like); 0: dictionary and decode by one wave per stream (rounds 1-2); 1: two kernels always
	int delta_wide = 0;             // $CORTO_DELTA_WIDE=1: K-DELTA keeps 32-bit values in LDS from the start (a context otherwise learns it from its first overflowing batch)
	bool check_pinned = false;      // $CORTO_HIP_CHECK_PINNED=1: a buffer handed over as a packed pinned arena (crthip_ctx_set_packed_host_blobs) is verified to be pinned host memory
	// experiments (A/B measurements; all bit-exact)
	bool tun_two_pass = false;      // $CORTO_TUN_TWO_PASS=1: long streams - one device-wide scan kernel over the chunk sums (round 1)
	bool tun_single_pass = false;   // $CORTO_TUN_SINGLE_PASS=1: long streams - adding-up and a wait-free look-back inside the decode kernel
	bool tun_three = false;         // $CORTO_TUN_THREE_LAUNCHES=1: one decode kernel per word-width class instead of one for all
	uint32_t tun_chunk_cap = TUN_CHUNK_CODES;   // $CORTO_EXP_TUN_CHUNK: largest chunk of a long stream (a power of two >= 2048)
	bool has_normal_fn_max = false; uint32_t normal_fn_max = 0;   // $CORTO_EXP_NORMAL_FN_MAX: largest LDS request for which K-NRM keeps its face normals in LDS
	bool delta_walk = false;        // $CORTO_EXP_DELTA_WALK=1: K-DELTA's 32-bit kernel without the scan passes (flag-driven walk only)
	bool no_deq_fold = false;       // $CORTO_EXP_NO_DEQ_FOLD=1: every attribute through k_dequant instead of K-DELTA's / K-NRM's copy-out
	uint32_t delta_group = 0;       // $CORTO_EXP_DELTA_GROUP: attributes of a blob per K-DELTA workgroup (1..4; 0 = as many as fit)
	bool unpack_chunked = false;    // $CORTO_EXP_UNPACK_CHUNKED=1: every bit block through the chunked K-BIT with its look-back (rounds 1-2), however small
	bool unpack_twice = false;      // $CORTO_EXP_UNPACK_TWICE=1: K-BIT launched twice (what the kernel costs a pipelined decode: tools/lds_pad_probe.sh)
	bool delta_tree = false;        // $CORTO_EXP_DELTA_TREE=1: attributes without parallelogram prediction (v += v[a]: a tree) by pointer jumping in a workgroup of their own (k_delta_tree) instead of a wave of K-DELTA's window kernel: measured level
	bool delta_global = false;      // $CORTO_EXP_DELTA_GLOBAL=1: K-DELTA of LDS-sized blobs with no LDS at all (k_delta_global)
	uint32_t lds_pad_delta = 0, lds_pad_topo = 0, lds_pad_normal = 0;   // $CORTO_EXP_LDS_PAD_{DELTA,TOPO,NORMAL}: KiB of LDS requested on top of what the kernel uses (what bounds the pipelined rate: tools/lds_pad_probe.sh)
};

inline DebugConfig debug_config_from_env() {
	{
		DebugConfig c;
		auto on = [](const char *name) { const char *e = getenv(name); return e && e[0] == '1'; };
		if(const char *e = getenv("CORTO_TUN_SHARE")) if(e[0] >= '0' && e[0] <= '2') c.tun_share = e[0] - '0';
		c.delta_wide = on("CORTO_DELTA_WIDE");
		c.check_pinned = on("CORTO_HIP_CHECK_PINNED");
		c.tun_two_pass = on("CORTO_TUN_TWO_PASS");
		c.tun_single_pass = on("CORTO_TUN_SINGLE_PASS") && !c.tun_two_pass;
		c.tun_three = on("CORTO_TUN_THREE_LAUNCHES");
		if(const char *e = getenv("CORTO_EXP_TUN_CHUNK")) { const uint32_t v = (uint32_t)atoi(e); if(v >= 2048 && v <= TUN_CHUNK_CODES && !(v & (v - 1))) c.tun_chunk_cap = v; }
		if(const char *e = getenv("CORTO_EXP_NORMAL_FN_MAX")) { c.has_normal_fn_max = true; c.normal_fn_max = (uint32_t)atoi(e); }
		c.delta_walk = on("CORTO_EXP_DELTA_WALK");
		c.no_deq_fold = on("CORTO_EXP_NO_DEQ_FOLD");
		c.delta_global = on("CORTO_EXP_DELTA_GLOBAL");
		c.delta_tree = on("CORTO_EXP_DELTA_TREE");
		c.unpack_twice = on("CORTO_EXP_UNPACK_TWICE");
		c.unpack_chunked = on("CORTO_EXP_UNPACK_CHUNKED");
		if(const char *e = getenv("CORTO_EXP_DELTA_GROUP")) { const uint32_t v = (uint32_t)atoi(e); if(v >= 1 && v <= 4) c.delta_group = v; }
		if(const char *e = getenv("CORTO_EXP_LDS_PAD_DELTA")) c.lds_pad_delta = (uint32_t)atoi(e)*1024u;
		if(const char *e = getenv("CORTO_EXP_LDS_PAD_TOPO")) c.lds_pad_topo = (uint32_t)atoi(e)*1024u;
		if(const char *e = getenv("CORTO_EXP_LDS_PAD_NORMAL")) c.lds_pad_normal = (uint32_t)atoi(e)*1024u;
		return c;
	}
}

} // namespace corto_hip
