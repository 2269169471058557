// debug_config.h — every environment switch of the decode path in one struct, filled when a context is made
// (crthip_ctx_create) and never consulted anywhere else: no getenv in the planner, no function-local statics.
//
// These are deployment settings and test hooks (documented in INTEGRATION.md); none changes results, only which kernels produce them.
// Rounds 2-3 also carried some fifteen experiment switches (alternative kernel paths kept for A/B measurements: a two-pass and a
// single-pass long-stream decode, three decode launches, K-DELTA without LDS / as pointer jumping / walk-only, LDS pad knobs, K-BIT
// twice ...).  Each was measured level or slower (profiles/r03_what_bounds_the_pipeline.txt has the numbers) and they were
// REMOVED in round 4, kernels included: the library has one kernel per stage and size class.
// Besides these: decoder_facade.cpp reads $CORTO_HIP_CONTEXTS / $CORTO_HIP_DEVICE / $CORTO_HIP_COMBINE_US / $CORTO_HIP_LEADERS once, pool.cpp reads
// ROCm's own $GPU_MAX_HW_QUEUES to size itself.
#pragma once
#include <cstdint>
#include <cstdlib>

#include "device_plan.h"

namespace corto_hip {

struct DebugConfig {
	int tun_share = -1;             // $CORTO_TUN_SHARE: unset: launches of 64+ streams make every DISTINCT probability table's dictionary once (K-TAB) and decode
	                                // the streams in groups of one dictionary; 2: the same two kernels but one dictionary PER STREAM, whatever repeats (what a
	                                // batch of unrelated meshes looks like: bench.py's `without_dictionary_sharing`); 0: dictionary and decode by one wave per
	                                // stream (what a launch of fewer than 64 streams takes anyway); 1: two kernels always
	int delta_wide = 0;             // $CORTO_DELTA_WIDE=1: K-DELTA keeps 32-bit values in LDS from the start (a context otherwise learns it from its first overflowing batch)
	bool check_pinned = false;      // $CORTO_HIP_CHECK_PINNED=1: a buffer handed over as a packed pinned arena (crthip_ctx_set_packed_host_blobs) is verified to be pinned host memory
	bool delta_rounds = false;      // $CORTO_DELTA_ROUNDS=1 (test hook): K-DELTA's round loop from vertex 1 for every attribute - int16, 32-bit and byte records, with and
	                                // without parallelogram prediction - instead of after 24 slow window passes (tests/test_gpu_parity.py runs every fixture through it)
	bool delta_walk = false;        // $CORTO_DELTA_WALK=1 (A/B and test hook): attributes too big for K-DELTA's LDS records take rounds 1-5's stretch walk over L2 (k_delta_mesh)
	                                // instead of the tiles of k_delta_tiles
	bool values_i32 = false;        // $CORTO_VALUES_I32=1 (A/B and test hook): K-BIT always hands 32-bit values on (otherwise int16 where an attribute's tables prove every
	                                // width <= 16 bits and the consumer is k_delta_lds16 / k_normal_blob: plan_jobs.cpp)
	bool unpack_chunked = false;    // $CORTO_UNPACK_CHUNKED=1 (test hook): every bit block through the chunked K-BIT with its look-back - the kernel of big meshes -
	                                // however small (tests/test_gpu_parity.py runs ragged sizes through both)
};

inline DebugConfig debug_config_from_env() {
	DebugConfig c;
	auto on = [](const char *name) { const char *e = getenv(name); return e && e[0] == '1'; };
	if(const char *e = getenv("CORTO_TUN_SHARE")) if(e[0] >= '0' && e[0] <= '2') c.tun_share = e[0] - '0';
	c.delta_wide = on("CORTO_DELTA_WIDE");
	c.check_pinned = on("CORTO_HIP_CHECK_PINNED");
	c.unpack_chunked = on("CORTO_UNPACK_CHUNKED");
	c.delta_rounds = on("CORTO_DELTA_ROUNDS");
	c.delta_walk = on("CORTO_DELTA_WALK");
	c.values_i32 = on("CORTO_VALUES_I32");
	return c;
}

} // namespace corto_hip
