// k_mesh.hip — the two per-blob SERIAL stages of a mesh decode on gfx950:
//   K-TOPO  CLERS automaton -> triangle list + per-vertex prediction triples   src/decoder.cpp:204-358
//   K-DELTA parallelogram / first-neighbour delta inversion                     include/corto/vertex_attribute.h:160-176,
//                                                                              src/normal_attribute.cpp:193-201
// Both are chains of dependent reads per blob: the parallel unit is the blob.  One
// wave per blob; many blobs per launch.
#include "kernels_common.h"
#include "kernels.h"

namespace corto_hip {

enum : uint32_t { C_VERTEX = 0, C_LEFT = 1, C_RIGHT = 2, C_END = 3, C_BOUNDARY = 4, C_DELAY = 5, C_SPLIT = 6 };
constexpr int32_t ERR_TOPOLOGY = -5;   // CRTHIP_E_TOPOLOGY

// ------------------------------------------------------------------------------------------------
// K-TOPO.  The reference automaton as it stands, with the front and its queues in HBM scratch sized by the stream's
// max_front (any mesh size): the redo path of blobs whose live front outgrows the LDS slots of topo_lds_body below.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct GlobalFront {
	CRT_GLOBAL u32x4 *fa;            // (v0, v1, v2, deleted)
	CRT_GLOBAL u32x2 *fb;            // (prev, next)
	CRT_GLOBAL uint32_t *order;
	CRT_GLOBAL uint32_t *delayed;
	__device__ void core(uint32_t e, uint32_t &v0, uint32_t &v1, uint32_t &v2, uint32_t &dead) const { const u32x4 t = fa[e]; v0 = t.x; v1 = t.y; v2 = t.z; dead = t.w; }
	__device__ void links(uint32_t e, uint32_t &p, uint32_t &n) const { const u32x2 t = fb[e]; p = t.x; n = t.y; }
	__device__ uint32_t prev(uint32_t e) const { return ((CRT_GLOBAL uint32_t *)fb)[2*(size_t)e]; }
	__device__ uint32_t next(uint32_t e) const { return ((CRT_GLOBAL uint32_t *)fb)[2*(size_t)e + 1]; }
	__device__ uint32_t v0(uint32_t e) const { return ((CRT_GLOBAL uint32_t *)fa)[4*(size_t)e]; }
	__device__ uint32_t v1(uint32_t e) const { return ((CRT_GLOBAL uint32_t *)fa)[4*(size_t)e + 1]; }
	__device__ void set_prev(uint32_t e, uint32_t x) { ((CRT_GLOBAL uint32_t *)fb)[2*(size_t)e] = x; }
	__device__ void set_next(uint32_t e, uint32_t x) { ((CRT_GLOBAL uint32_t *)fb)[2*(size_t)e + 1] = x; }
	__device__ void kill(uint32_t e) { ((CRT_GLOBAL uint32_t *)fa)[4*(size_t)e + 3] = 1; }
	__device__ void put(uint32_t e, uint32_t a, uint32_t b, uint32_t c, uint32_t p, uint32_t n) {
		u32x4 t; t.x = a; t.y = b; t.z = c; t.w = 0; fa[e] = t;
		u32x2 l; l.x = p; l.y = n; fb[e] = l;
	}
	__device__ void order_put(uint32_t i, uint32_t e) { order[i] = e; }
	__device__ uint32_t order_get(uint32_t i) const { return order[i]; }
	__device__ void delayed_put(uint32_t i, uint32_t e) { delayed[i] = e; }
	__device__ uint32_t delayed_get(uint32_t i) const { return delayed[i]; }
	// Next LIVE edge of the queue [iorder, norder), or false if the next 64 entries are all deleted (iorder moves past what was
	// looked at either way).  Nine queued edges in ten are dead by the time they are popped, and here each costs two dependent
	// L2 round trips (the queue entry, then the edge's deleted flag): the wave's 63 parked lanes are switched on for the two
	// loads, so 64 entries are examined per pair of round trips.  The queue holds edge ids this automaton wrote itself.
	__device__ bool pop_live(uint32_t &iorder, uint32_t norder, uint32_t &f) const {
		const uint32_t io = (uint32_t)__builtin_amdgcn_readfirstlane((int)iorder), avail = (uint32_t)__builtin_amdgcn_readfirstlane((int)(norder - iorder));
		const uint64_t po = (uint64_t)(uintptr_t)order, pf = (uint64_t)(uintptr_t)fa;
		const uint64_t ord = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)po) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(po >> 32)) << 32;
		const uint64_t fab = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pf) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(pf >> 32)) << 32;
		uint64_t alive, save;
		uint32_t ent;
		asm volatile(
			"s_mov_b64 %[sv], exec\n"
			"s_mov_b64 exec, -1\n"
			"v_mbcnt_lo_u32_b32 v60, -1, 0\n"
			"v_mbcnt_hi_u32_b32 v60, -1, v60\n"
			"v_cmp_gt_u32 vcc, %[avail], v60\n"
			"s_mov_b64 exec, vcc\n"
			"v_add_u32 v61, %[io], v60\n"
			"v_lshlrev_b32 v61, 2, v61\n"
			"global_load_dword v62, v61, %[ord]\n"
			"s_waitcnt vmcnt(0)\n"
			"v_lshlrev_b32 v61, 4, v62\n"
			"global_load_dword v63, v61, %[fab] offset:12\n"
			"s_waitcnt vmcnt(0)\n"
			"v_cmp_eq_u32 %[alive], 0, v63\n"
			"v_mov_b32 %[ent], v62\n"
			"s_mov_b64 exec, %[sv]\n"
			: [alive] "=s"(alive), [sv] "=&s"(save), [ent] "=&v"(ent)
			: [avail] "s"(avail), [io] "s"(io), [ord] "s"(ord), [fab] "s"(fab)
			: "memory", "vcc", "v60", "v61", "v62", "v63");
		if(!alive) { iorder = io + (avail < 64 ? avail : 64u); return false; }
		const uint32_t j = (uint32_t)__builtin_ctzll(alive);
		f = (uint32_t)__builtin_amdgcn_readlane((int)ent, (int)j);
		iorder = io + j + 1;
		return true;
	}
};

template <class ClersPtr>
struct TopoState {
	const TopoJob &J;
	ClersPtr clers;                  // global, or the LDS copy
	CRT_GLOBAL const uint32_t *split;
	CRT_GLOBAL uint32_t *pred;
	CRT_GLOBAL uint32_t *f32;
	CRT_GLOBAL uint16_t *f16;
	uint32_t cler, vertex_count;
	uint64_t bit;
	int32_t err;
	// what the LDS form of this front would have needed (k_topology_lds' redo reports it, batch.cpp learns from it): the longest the queue got
	// (ring slots), the most pool slots held at once (every BOUNDARY edge for good, every DELAYed one while it sits in the stack), the deepest DELAY stack
	uint32_t peak_queue = 0, chain_ends = 0, peak_delayed = 0, peak_pool = 0;
	__device__ uint32_t bits(uint32_t n) {
		if(bit + n > (uint64_t)J.split_nwords*32) { err = ERR_TOPOLOGY; return 0; }
		const uint32_t v = bit_field(split, J.split_nwords, bit, n);
		bit += n;
		return v;
	}
	__device__ void face(uint32_t at, uint32_t a, uint32_t b, uint32_t c) {
		if(f16) { f16[at] = (uint16_t)a; f16[at + 1] = (uint16_t)b; f16[at + 2] = (uint16_t)c; }
		else { f32[at] = a; f32[at + 1] = b; f32[at + 2] = c; }
	}
	__device__ void predict(uint32_t v, uint32_t a, uint32_t b, uint32_t c) {
		CRT_GLOBAL uint32_t *p = pred + (size_t)v*3;
		p[0] = a; p[1] = b; p[2] = c;
	}
};

template <class Front, class ClersPtr>
__device__ void topo_group(TopoState<ClersPtr> &S, Front &F, uint32_t start, uint32_t end) {
	const TopoJob &J = S.J;
	const uint32_t cap = J.front_cap;
	uint32_t nfront = 0, norder = 0, iorder = 0, ndelayed = 0;
	int64_t new_edge = -1;
	const uint32_t splitbits = 32 - __clz(J.nvert | 1u);      // ilog2(nvert) + 1 (src/cstream.cpp:31-35; nvert >= 1 here)

#define FAIL() do { S.err = ERR_TOPOLOGY; return; } while(0)
	while(start < end) {
		if(new_edge == -1 && iorder >= norder && ndelayed == 0) {      // seed face (decoder.cpp:224-259)
			if(S.cler >= J.nclers) FAIL();
			uint32_t last = S.vertex_count - 1, vi[3], split = 0;
			const uint32_t c = S.clers[S.cler++];
			if(c == C_SPLIT) split = S.bits(3);
			for(int k = 0; k < 3; k++) {
				uint32_t v;
				if(split & (1u << k)) v = S.bits(splitbits);
				else {
					if(S.vertex_count >= J.nvert) FAIL();
					S.predict(S.vertex_count, last, last, last);
					last = v = S.vertex_count++;
				}
				vi[k] = v;
			}
			if(S.err) return;
			S.face(start, vi[0], vi[1], vi[2]); start += 3;
			const uint32_t e = nfront;
			if(e + 3 > cap) FAIL();
			F.order_put(norder++, e); F.order_put(norder++, e + 1); F.order_put(norder++, e + 2);
			if(norder - iorder > S.peak_queue) S.peak_queue = norder - iorder;
			F.put(e, vi[1], vi[2], vi[0], e + 2, e + 1);
			F.put(e + 1, vi[2], vi[0], vi[1], e, e + 2);
			F.put(e + 2, vi[0], vi[1], vi[2], e + 1, e);
			nfront += 3;
			continue;
		}
		uint32_t f;
		if(new_edge != -1) { f = (uint32_t)new_edge; new_edge = -1; }
		else if(iorder < norder) { if(!F.pop_live(iorder, norder, f)) continue; }   // (deleted entries consume no symbol, decoder.cpp:278-279)
		else f = F.delayed_get(--ndelayed);
		if(f >= nfront) FAIL();
		uint32_t v0, v1, v2, dead;
		F.core(f, v0, v1, v2, dead);
		if(dead) continue;                                             // deleted: no symbol consumed (decoder.cpp:278-279)
		if(S.cler >= J.nclers) FAIL();
		const uint32_t c = S.clers[S.cler++];
		if(c == C_BOUNDARY) { S.chain_ends++; continue; }              // (the LDS form keeps no record of it: the sink, below)
		uint32_t ep, en;
		F.links(f, ep, en);
		if(ep >= nfront || en >= nfront) FAIL();
		const uint32_t ne = nfront;
		uint32_t opp;
		new_edge = ne;
		if(c == C_VERTEX || c == C_SPLIT) {                            // decoder.cpp:294-309
			if(c == C_SPLIT) { opp = S.bits(splitbits); if(S.err) return; }
			else {
				if(S.vertex_count >= J.nvert) FAIL();
				S.predict(S.vertex_count, v1, v0, v2);
				opp = S.vertex_count++;
			}
			if(ne + 2 > cap) FAIL();
			F.set_next(ep, ne);
			F.set_prev(en, ne + 1);
			F.put(ne, v0, opp, v1, ep, ne + 1);
			F.order_put(norder++, ne + 1);
			if(norder - iorder > S.peak_queue) S.peak_queue = norder - iorder;
			F.put(ne + 1, opp, v1, v0, ne, en);
			nfront += 2;
		} else if(c == C_LEFT) {                                       // decoder.cpp:311-317
			const uint32_t pp = F.prev(ep);
			if(pp >= nfront || ne + 1 > cap) FAIL();
			opp = F.v0(ep);
			F.kill(ep);
			F.set_next(pp, ne);
			F.set_prev(en, ne);
			F.put(ne, opp, v1, v0, pp, en);
			nfront += 1;
		} else if(c == C_RIGHT) {                                      // decoder.cpp:319-325
			const uint32_t nn = F.next(en);
			if(nn >= nfront || ne + 1 > cap) FAIL();
			opp = F.v1(en);
			F.kill(en);
			F.set_prev(nn, ne);
			F.set_next(ep, ne);
			F.put(ne, v0, opp, v1, ep, nn);
			nfront += 1;
		} else if(c == C_DELAY) {                                      // decoder.cpp:327-331
			if(ndelayed >= cap) FAIL();
			F.delayed_put(ndelayed++, f);
			if(ndelayed > S.peak_delayed) S.peak_delayed = ndelayed;
			if(ndelayed + 1 > S.peak_pool) S.peak_pool = ndelayed + 1;     // (the LDS form's pool: the DELAYed edges while they wait + the sink)
			new_edge = -1;
			continue;
		} else if(c == C_END) {                                        // decoder.cpp:333-339
			const uint32_t pp = F.prev(ep), nn = F.next(en);
			if(pp >= nfront || nn >= nfront) FAIL();
			opp = F.v0(ep);
			F.kill(ep); F.kill(en);
			F.set_next(pp, nn);
			F.set_prev(nn, pp);
			new_edge = -1;
		} else FAIL();
		S.face(start, v1, v0, opp); start += 3;                        // decoder.cpp:348-356
	}
#undef FAIL
}

template <class Front, class ClersPtr>
__device__ uint32_t topo_run(const TopoJob &J, ClersPtr clers, Front &F) {      // returns the slots the LDS form would have needed: ring | pool << 15
	TopoState<ClersPtr> S{J, clers, as_global(J.split_words), as_global(J.pred),
	                      J.faces_u16 ? nullptr : as_global((uint32_t *)J.faces), J.faces_u16 ? as_global((uint16_t *)J.faces) : nullptr, 0, 0, 0, 0};
	uint32_t need = 0;
	CRT_GLOBAL const uint32_t *group_end = as_global(J.group_end);
	uint32_t start = 0;
	for(uint32_t g = 0; g < J.ngroups && !S.err; g++) {                // decoder.cpp:173-178
		const uint32_t ge = group_end[g];
		if(ge > J.nface || ge < start) { S.err = ERR_TOPOLOGY; break; }
		S.peak_queue = S.chain_ends = S.peak_delayed = S.peak_pool = 0;   // (every group starts from an empty front)
		topo_group(S, F, start*3, ge*3);
		const uint32_t nr_ = min(S.peak_queue + 4u, 0x7FFFu), np_ = min(max(S.peak_pool + S.peak_pool/16u + 16u, S.peak_delayed + 1u), 0x7FFFu);   // (a chain-end step takes its pool slots before it gives any back)
		need = max(need & 0x7FFFu, nr_) | max(need >> 15, np_) << 15;         // ring slots | pool slots << 15
		start = ge;
	}
	// vertices the stream never made keep the prediction (0, 0, 0), as in the reference's zero-filled vector (src/decoder.cpp:171): the
	// scratch block is not cleared in front of a batch, so the automaton finishes the array itself (nothing to do for a valid stream)
	for(uint32_t v = S.vertex_count; v < J.nvert; v++) S.predict(v, 0, 0, 0);
	if(S.err) *as_global(J.status) = S.err;
	return need;
}

// general path: any size, state in HBM scratch
__global__ __launch_bounds__(64) void k_topology(const TopoJob *__restrict__ jobs, const uint32_t *__restrict__ job_ids, uint32_t njobs) {
	if(blockIdx.x >= njobs || threadIdx.x != 0) return;
	const TopoJob J = jobs[job_ids[blockIdx.x]];
	GlobalFront F{(CRT_GLOBAL u32x4 *)as_global(J.front_a), (CRT_GLOBAL u32x2 *)as_global(J.front_b), as_global(J.order), as_global(J.delayed)};
	topo_run(J, as_global(J.clers), F);
	if(J.pad & TOPO_PAD_PROGRESS) __hip_atomic_store((CRT_GLOBAL uint32_t *)((CRT_GLOBAL uint8_t *)as_global(J.pred) - TOPO_PROGRESS_BYTES), 0xFFFFFFFFu, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// LDS path: the front of one blob in LDS, hand-tightened.  A lone wave issues one instruction every ~4 cycles, so this
// single-lane loop is issue-bound: every instruction per symbol counts.
//   * 16-byte edge records {v0, v1, v2 | flags, prev | next<<16}: one ds_read_b128 / ds_write_b128 each
//   * LAZY current edge: the edge created by VERTEX/LEFT/RIGHT is the next one processed (decoder.cpp:261-264), and when
//     it is consumed right away by another VERTEX/LEFT/RIGHT/END nobody ever reads its record, and the links its
//     neighbours hold to it are overwritten by that step.  So the current edge lives in registers only; it gets a record
//     slot, its record and the two neighbour links ("materialised") only if it survives (BOUNDARY / DELAY).  Edge ids are
//     internal to the decoder (outputs carry vertex ids only), so this is unobservable.
//   * RECORDS ARE RECYCLED, so the LDS holds the LIVE front, not every edge ever made (the reference's max_front ~ 3*nvert):
//       - queued edges (one per VERTEX / SPLIT, three per seed face) take the slots of a RING in the order they are
//         queued, so the reference's FIFO (`faceorder`) is just the ring's read cursor - no queue storage, no push, and a
//         pop is the record read it needs anyway; a slot is free again the moment it is popped;
//       - surviving chain ends take slots of a POOL with a free list, and give them back when LEFT / RIGHT / END delete
//         them (or, if they sit in the DELAY stack, when they are popped from it).
//     The live front of a mesh is ~3*sqrt(nface) queued edges (251 records for the 4K-triangle blob, 2 000 for a
//     256K-triangle sphere), so a 4K-triangle blob needs 16 KB of LDS instead of 100 KB, ten blobs' automata fit a CU,
//     and meshes of any size run from LDS.  A blob that outgrows its ring or pool (a torus' front is ten times a
//     sphere's) is redone on the HBM front.
//   * RIGHT right after VERTEX closes against the edge VERTEX just created -> its (next, v1) are cached in registers
//   * links stored in the front are produced by this loop, hence always in range; only values that come from the stream
//     are validated.  The symbols sit in LDS as nibbles, eight per dword, in a window the whole wave refills between
//     chains (a blob of up to 8K symbols is loaded once); the padding behind the last symbol is an invalid symbol, so
//     running off the end fails without a per-step bounds test.
// Layout (dynamic LDS): rec[ring + pool] (16 B) | freelist[pool] (u16) | delayed[dcap] (u16) | symbol window (nibbles)
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
#define TOPO_S(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
constexpr uint32_t TOPO_DEAD = 0x80000000u, TOPO_DELAYED = 0x40000000u, TOPO_VMASK = 0x3FFFFFFFu;

// the face store of the ISA block below: three u32, or three u16 (a dword of v1 | v0 << 16, then opp)
#define TOPO_ASM_FACE32(OPP) \
	"  s_lshl_b32 %[t0], %[start], 2\n" \
	"  v_mov_b32 v44, %[v1]\n" \
	"  v_mov_b32 v45, %[v0]\n" \
	"  v_mov_b32 v46, " OPP "\n" \
	"  v_mov_b32 v47, %[t0]\n" \
	"  global_store_dwordx3 v47, v[44:46], %[faceb]\n"
#define TOPO_ASM_FACE16(OPP) \
	"  s_lshl_b32 %[t0], %[start], 1\n" \
	"  s_and_b32 %[t3], %[v1], 0xffff\n" \
	"  s_lshl_b32 %[c], %[v0], 16\n" \
	"  s_or_b32 %[t3], %[t3], %[c]\n" \
	"  v_mov_b32 v44, %[t3]\n" \
	"  v_mov_b32 v46, " OPP "\n" \
	"  v_mov_b32 v47, %[t0]\n" \
	"  global_store_dword v47, v44, %[faceb]\n" \
	"  global_store_short v47, v46, %[faceb] offset:4\n"
// after a consumed symbol: advance the nibble window (every eighth symbol takes the out-of-line refill), test the end of the
// group, and dispatch the NEXT symbol right here - "threaded": one taken branch per symbol instead of four (a taken branch
// restarts the lone wave's instruction fetch, ~20 clocks each)
// -DCORTO_TOPO_STAMPS: every dispatch leaves the shader clock in an LDS trace word of its own (32 KB up, indexed by the symbol it is about to
// take): tools/topo_trace_probe.py turns the differences into what each kind of step costs.  Nothing in the product build.
#ifdef CORTO_TOPO_STAMPS
#define TOPO_ASM_STAMP \
							"  s_memtime s[86:87]\n" \
							"  s_mul_i32 s88, %[cler], 4\n" \
							"  v_mov_b32 v63, s88\n" \
							"  s_waitcnt lgkmcnt(0)\n" \
							"  v_mov_b32 v62, s86\n" \
							"  ds_write_b32 v63, v62 offset:32768\n"
#define TOPO_ASM_STAMP_CLOBBERS , "s86", "s87", "s88"
#else
#define TOPO_ASM_STAMP
#define TOPO_ASM_STAMP_CLOBBERS
#endif
#define TOPO_ASM_TAIL \
							"  s_lshr_b64 s[90:91], s[90:91], 4\n" \
							"  s_add_u32 %[cler], %[cler], 1\n" \
							"  s_and_b32 %[t0], %[cler], 7\n"   /* SCC = result != 0 */ \
							"  s_cbranch_scc0 Lrefill_%=\n" \
							"  s_cmp_lt_u32 %[start], %[end]\n" \
							"  s_cbranch_scc0 Lexit_%=\n" \
							TOPO_ASM_STAMP \
							"  s_and_b32 %[c], s90, 15\n" \
							"  s_cbranch_scc0 Lvertex_%=\n" \
							"  s_cmp_eq_u32 %[c], 1\n" \
							"  s_cbranch_scc1 Lleft_%=\n" \
							"  s_cmp_eq_u32 %[c], 2\n" \
							"  s_cbranch_scc1 Lright_%=\n" \
							"  s_branch Lcold_%=\n"
#define TOPO_FAST_PATH(FACE, RUNFACE, LEADFACE, MIXFACE, FSHIFT) \
						asm volatile( \
							"  s_mov_b32 s90, %[sw]\n"             /* the window registers live in s[90:91] inside the block: one 64-bit shift per symbol keeps EIGHT symbols */ \
							"  s_mov_b32 s91, %[swn]\n"            /* in s90 whatever the alignment (s91: what is left of the next word), so every trigger sees eight */ \
							"Ltop_%=:\n" \
							TOPO_ASM_STAMP \
							"  s_and_b32 %[c], s90, 15\n"   /* (SCC = result != 0) */ \
							"  s_cbranch_scc0 Lvertex_%=\n" \
							"  s_cmp_eq_u32 %[c], 1\n" \
							"  s_cbranch_scc1 Lleft_%=\n" \
							"  s_cmp_eq_u32 %[c], 2\n" \
							"  s_cbranch_scc1 Lright_%=\n" \
							"  s_branch Lcold_%=\n" \
   /* ---------------- VERTEX (decoder.cpp:294-309) */ \
							"Lvertex_%=:\n" \
							"  s_and_b32 %[t0], s90, 0xffff\n"   /* VERTEX LEFT VERTEX LEFT ahead: leave for the run step (TOPO_RUN_STEP) */ \
							"  s_cmp_eq_u32 %[t0], 0x1010\n" \
							"  s_cbranch_scc1 Lvrun_%=\n" \
							"  s_and_b32 %[t1], %[t0], 0xeeee\n"   /* four symbols of VERTEX / LEFT ahead: maybe the mix step (checked out of line) */ \
							"  s_cbranch_scc0 Lvmix_%=\n" \
							"Lvgo_%=:\n" \
							"  s_sub_u32 %[budget], %[budget], 1\n"   /* vertex ids and ring slots left (SCC = borrow: none) */ \
							"  s_cbranch_scc1 Lexit_%=\n" \
							"  s_and_b32 %[t1], %[nq], %[mask]\n"   /* s: slot of the second new edge */ \
							"  s_add_u32 %[nq], %[nq], 1\n" \
							"  s_mul_i32 %[t0], %[vc], 12\n"   /* prediction triple (v1, v0, v2) of the new vertex */ \
							"  v_mov_b32 v40, %[v1]\n" \
							"  v_mov_b32 v41, %[v0]\n" \
							"  v_mov_b32 v42, %[v2]\n" \
							"  v_mov_b32 v43, %[t0]\n" \
							"  global_store_dwordx3 v43, v[40:42], %[predb]\n" \
							FACE("%[vc]")                                           /* face (v1, v0, opp = vc) */ \
							"  s_add_u32 %[start], %[start], 3\n" \
							"  s_lshl_b32 %[t0], %[en], 4\n"   /* front[e.next].prev = s */ \
							"  v_mov_b32 v52, %[t0]\n" \
							"  v_mov_b32 v53, %[t1]\n" \
							"  ds_write_b16 v52, v53 offset:12\n" \
							"  s_lshl_b32 %[t2], %[en], 16\n"   /* rec[s] = {opp, v1, v0, 0xFFFF | en << 16} */ \
							"  s_or_b32 %[t2], %[t2], 0xffff\n" \
							"  v_mov_b32 v48, %[vc]\n" \
							"  v_mov_b32 v49, %[v1]\n" \
							"  v_mov_b32 v50, %[v0]\n" \
							"  v_mov_b32 v51, %[t2]\n" \
							"  s_lshl_b32 %[t0], %[t1], 4\n" \
							"  v_mov_b32 v54, %[t0]\n" \
							"  ds_write_b128 v54, v[48:51]\n" \
							"  s_mov_b32 %[nc], %[t1]\n" \
							"  s_mov_b32 %[ncnext], %[en]\n" \
							"  s_mov_b32 %[ncv1], %[v1]\n" \
							"  s_mov_b32 %[v2], %[v1]\n" \
							"  s_mov_b32 %[v1], %[vc]\n" \
							"  s_mov_b32 %[en], %[t1]\n" \
							"  s_add_u32 %[vc], %[vc], 1\n" \
							TOPO_ASM_TAIL \
   /* ---------------- LEFT (decoder.cpp:311-317), neighbour in the ring */ \
							"Lleft_%=:\n" \
							"  s_cmp_gt_u32 %[ep], %[mask]\n" \
							"  s_cbranch_scc1 Lleftp_%=\n" \
							"  s_and_b32 %[t0], s90, 0xeeee\n" \
							"  s_cbranch_scc0 Llmix_%=\n" \
							"  s_and_b32 %[t1], s90, 0xeeff\n"      /* LEFT RIGHT and two of VERTEX / LEFT, or LEFT LEFT RIGHT and one: the mix step takes ONE RIGHT */ \
							"  s_cmp_eq_u32 %[t1], 0x0021\n"        /* that has no VERTEX in front of it (after a DELAY the next gate goes L R V V ..) */ \
							"  s_cbranch_scc1 Lmixr_%=\n" \
							"  s_and_b32 %[t1], s90, 0xefff\n" \
							"  s_cmp_eq_u32 %[t1], 0x0211\n" \
							"  s_cbranch_scc1 Lmixr_%=\n" \
							"Llgo_%=:\n" \
							"  s_lshl_b32 %[t0], %[ep], 4\n" \
							"  v_mov_b32 v52, %[t0]\n" \
							"  ds_read_b128 v[56:59], v52\n" \
							"  v_mov_b32 v53, 0x8000\n" \
							"  s_waitcnt lgkmcnt(0)\n" \
							"  v_readfirstlane_b32 %[t1], v56\n"   /* opp = prev.v0 */ \
							"  v_readfirstlane_b32 %[t2], v59\n" \
							"  s_and_b32 %[t2], %[t2], 0xffff\n"   /* pp = prev.prev */ \
							"  ds_write_b16 v52, v53 offset:10\n"   /* prev.deleted = true */ \
							FACE("%[t1]") \
							"  s_add_u32 %[start], %[start], 3\n" \
							"  s_mov_b32 %[v2], %[v0]\n" \
							"  s_mov_b32 %[v0], %[t1]\n" \
							"  s_mov_b32 %[ep], %[t2]\n" \
							TOPO_ASM_TAIL \
   /* ---------------- RIGHT (decoder.cpp:319-325): against the edge VERTEX just made (cached), or a ring neighbour */ \
							"Lright_%=:\n" \
							"  s_lshl_b32 %[t0], %[en], 4\n" \
							"  v_mov_b32 v52, %[t0]\n" \
							"  s_cmp_eq_u32 %[en], %[nc]\n" \
							"  s_cbranch_scc1 Lrightc_%=\n" \
							"  s_cmp_gt_u32 %[en], %[mask]\n" \
							"  s_cbranch_scc1 Lrightp_%=\n" \
							"  ds_read_b128 v[56:59], v52\n" \
							"  s_waitcnt lgkmcnt(0)\n" \
							"  v_readfirstlane_b32 %[t1], v57\n"   /* opp = next.v1 */ \
							"  v_readfirstlane_b32 %[t2], v59\n" \
							"  s_lshr_b32 %[t2], %[t2], 16\n"   /* nn = next.next */ \
							"  s_branch Lrightd_%=\n" \
							"Lrightc_%=:\n" \
							"  s_mov_b32 %[t1], %[ncv1]\n" \
							"  s_mov_b32 %[t2], %[ncnext]\n" \
							"Lrightd_%=:\n" \
							"  v_mov_b32 v53, 0x8000\n" \
							"  ds_write_b16 v52, v53 offset:10\n"   /* next.deleted = true */ \
							FACE("%[t1]") \
							"  s_add_u32 %[start], %[start], 3\n" \
							"  s_mov_b32 %[nc], -1\n" \
							"  s_mov_b32 %[v2], %[v1]\n" \
							"  s_mov_b32 %[v1], %[t1]\n" \
							"  s_mov_b32 %[en], %[t2]\n" \
							TOPO_ASM_TAIL \
   /* ---------------- LEFT / RIGHT against a survivor (a pool slot: the neighbour is a chain end - after a DELAY, along a boundary, where a \
      mesh zips up): the same step, and the slot goes back to the free list unless it still sits in the DELAY stack (whose pop returns it). \
      Free list at (mask+1)*32 + 2*fill, fill = pk1 >> 16. */ \
							"Lleftp_%=:\n" \
							"  s_lshl_b32 %[t0], %[ep], 4\n" \
							"  v_mov_b32 v52, %[t0]\n" \
							"  ds_read_b128 v[56:59], v52\n" \
							"  v_mov_b32 v53, 0x8000\n" \
							"  s_waitcnt lgkmcnt(0)\n" \
							"  v_readfirstlane_b32 %[t1], v56\n"   /* opp = prev.v0 */ \
							"  v_readfirstlane_b32 %[t2], v59\n" \
							"  v_readfirstlane_b32 %[t3], v58\n"   /* its flags */ \
							"  ds_write_b16 v52, v53 offset:10\n"   /* prev.deleted = true */ \
							"  s_and_b32 %[t2], %[t2], 0xffff\n"   /* pp = prev.prev */ \
							"  s_bitcmp1_b32 %[t3], 30\n"           /* TOPO_DELAYED */ \
							"  s_cbranch_scc1 Lleftq_%=\n" \
							"  s_lshr_b32 %[t0], %[pk1], 16\n" \
							"  s_lshl_b32 %[t0], %[t0], 1\n" \
							"  s_and_b32 %[t3], %[lay], 0xffff\n" \
							"  s_lshl_b32 %[t3], %[t3], 4\n" \
							"  s_add_u32 %[t0], %[t0], %[t3]\n" \
							"  v_mov_b32 v54, %[t0]\n" \
							"  v_mov_b32 v55, %[ep]\n" \
							"  ds_write_b16 v54, v55\n" \
							"  s_add_u32 %[pk1], %[pk1], 0x10000\n" \
							"Lleftq_%=:\n" \
							FACE("%[t1]") \
							"  s_add_u32 %[start], %[start], 3\n" \
							"  s_mov_b32 %[v2], %[v0]\n" \
							"  s_mov_b32 %[v0], %[t1]\n" \
							"  s_mov_b32 %[ep], %[t2]\n" \
							TOPO_ASM_TAIL \
							"Lrightp_%=:\n" \
							"  ds_read_b128 v[56:59], v52\n"        /* (v52 = en*16 from Lright) */ \
							"  v_mov_b32 v53, 0x8000\n" \
							"  s_waitcnt lgkmcnt(0)\n" \
							"  v_readfirstlane_b32 %[t1], v57\n"   /* opp = next.v1 */ \
							"  v_readfirstlane_b32 %[t2], v59\n" \
							"  v_readfirstlane_b32 %[t3], v58\n" \
							"  ds_write_b16 v52, v53 offset:10\n"   /* next.deleted = true */ \
							"  s_lshr_b32 %[t2], %[t2], 16\n"       /* nn = next.next */ \
							"  s_bitcmp1_b32 %[t3], 30\n" \
							"  s_cbranch_scc1 Lrightq_%=\n" \
							"  s_lshr_b32 %[t0], %[pk1], 16\n" \
							"  s_lshl_b32 %[t0], %[t0], 1\n" \
							"  s_and_b32 %[t3], %[lay], 0xffff\n" \
							"  s_lshl_b32 %[t3], %[t3], 4\n" \
							"  s_add_u32 %[t0], %[t0], %[t3]\n" \
							"  v_mov_b32 v54, %[t0]\n" \
							"  v_mov_b32 v55, %[en]\n" \
							"  ds_write_b16 v54, v55\n" \
							"  s_add_u32 %[pk1], %[pk1], 0x10000\n" \
							"Lrightq_%=:\n" \
							FACE("%[t1]") \
							"  s_add_u32 %[start], %[start], 3\n" \
							"  s_mov_b32 %[nc], -1\n" \
							"  s_mov_b32 %[v2], %[v1]\n" \
							"  s_mov_b32 %[v1], %[t1]\n" \
							"  s_mov_b32 %[en], %[t2]\n" \
							TOPO_ASM_TAIL \
   /* ---------------- every eighth symbol: the next word of the window, then the loop test and the dispatch at the top */ \
							"Lrefill_%=:\n" \
							"  s_lshr_b32 %[t0], %[cler], 3\n" \
							"  s_lshl_b32 %[t0], %[t0], 2\n" \
							"  s_add_u32 %[t0], %[t0], %[clw]\n" \
							"  v_mov_b32 v55, %[t0]\n" \
							"  ds_read_b32 v55, v55\n" \
							"  s_waitcnt lgkmcnt(0)\n" \
							"  v_readfirstlane_b32 s91, v55\n" \
							"  s_cmp_lt_u32 %[start], %[end]\n" \
							"  s_cbranch_scc1 Ltop_%=\n" \
							"  s_branch Lexit_%=\n" \
   /* ---------------- BOUNDARY (decoder.cpp:282-283) and DELAY (:327-331): the chain ends and the current edge survives - it gets a pool slot \
      (free list, else the bump pointer), its record, and its neighbours their links to it - then the next gate is fetched from the ring: \
      the wave's parked lanes look at 64 queue entries at once, the first live one becomes the current edge, and the dispatch goes on \
      without leaving the block.  pk1 = pool bump pointer | free-list fill << 16, pk2 = DELAY stack fill | its capacity << 16. \
      Layout (records at LDS address 0): free list at 16*FL16, DELAY stack at 16*DL16, ring + pool = FL16 slots - the layout word %[lay] = FL16 | DL16 << 16. \
      Anything else (END, an invalid or window-end nibble, no slot left, empty ring and empty DELAY stack, window about to run out) leaves for the C++. */ \
							"Lcold_%=:\n" \
							"  s_cmp_eq_u32 %[c], 4\n"          /* BOUNDARY: the sink takes it (round 6: Lsink below) - no slot, no record: the two links, then the pop */ \
							"  s_cbranch_scc1 Lsink_%=\n" \
							"  s_cmp_eq_u32 %[c], 6\n" \
							"  s_cbranch_scc1 Lsplit_%=\n" \
							"  s_cmp_eq_u32 %[c], 5\n" \
							"  s_cbranch_scc0 Lexit_%=\n" \
							"  s_and_b32 %[t0], %[pk2], 0xffff\n"   /* DELAY: room on the stack? */ \
							"  s_lshr_b32 %[t2], %[pk2], 16\n" \
							"  s_cmp_ge_u32 %[t0], %[t2]\n" \
							"  s_cbranch_scc1 Lexit_%=\n" \
							"  s_lshr_b32 %[t0], %[pk1], 16\n"   /* (DELAY) free-list fill */ \
							"  s_cmp_eq_u32 %[t0], 0\n" \
							"  s_cbranch_scc1 Lbump_%=\n" \
							"  s_sub_u32 %[t0], %[t0], 1\n" \
							"  s_and_b32 %[t2], %[lay], 0xffff\n" \
							"  s_lshl_b32 %[t2], %[t2], 4\n" \
							"  s_lshl_b32 %[t0], %[t0], 1\n" \
							"  s_add_u32 %[t0], %[t0], %[t2]\n" \
							"  v_mov_b32 v52, %[t0]\n" \
							"  ds_read_u16 v52, v52\n" \
							"  s_sub_u32 %[pk1], %[pk1], 0x10000\n" \
							"  s_waitcnt lgkmcnt(0)\n" \
							"  v_readfirstlane_b32 %[t1], v52\n"   /* the slot */ \
							"  s_branch Lhave_%=\n" \
							"Lbump_%=:\n" \
							"  s_and_b32 %[t1], %[pk1], 0xffff\n" \
							"  s_and_b32 %[t2], %[lay], 0xffff\n"         /* ring + pool slots */ \
							"  s_cmp_ge_u32 %[t1], %[t2]\n"   /* pool exhausted: the C++ flags the blob for the HBM redo */ \
							"  s_cbranch_scc1 Lexit_%=\n" \
							"  s_add_u32 %[pk1], %[pk1], 1\n" \
							"Lhave_%=:\n" \
							"  s_mov_b32 %[t2], %[v2]\n" \
							"  s_cmp_eq_u32 %[c], 5\n" \
							"  s_cbranch_scc0 Lput_%=\n" \
							"  s_or_b32 %[t2], %[t2], 0x40000000\n"   /* TOPO_DELAYED; and push the slot */ \
							"  s_and_b32 %[t0], %[pk2], 0xffff\n" \
							"  s_lshl_b32 %[t0], %[t0], 1\n" \
							"  s_lshr_b32 %[t3], %[lay], 16\n" \
							"  s_lshl_b32 %[t3], %[t3], 4\n" \
							"  s_add_u32 %[t0], %[t0], %[t3]\n" \
							"  v_mov_b32 v52, %[t0]\n" \
							"  v_mov_b32 v53, %[t1]\n" \
							"  ds_write_b16 v52, v53\n" \
							"  s_add_u32 %[pk2], %[pk2], 1\n" \
							"Lput_%=:\n" \
							"  s_lshl_b32 %[t3], %[en], 16\n" \
							"  s_or_b32 %[t3], %[t3], %[ep]\n" \
							"  v_mov_b32 v48, %[v0]\n" \
							"  v_mov_b32 v49, %[v1]\n" \
							"  v_mov_b32 v50, %[t2]\n" \
							"  v_mov_b32 v51, %[t3]\n" \
							"  s_lshl_b32 %[t0], %[t1], 4\n" \
							"  v_mov_b32 v54, %[t0]\n" \
							"  ds_write_b128 v54, v[48:51]\n" \
							"Llinks_%=:\n" \
							"  v_mov_b32 v53, %[t1]\n" \
							"  s_lshl_b32 %[t0], %[ep], 4\n" \
							"  v_mov_b32 v52, %[t0]\n" \
							"  ds_write_b16 v52, v53 offset:14\n"   /* front[e.prev].next = slot */ \
							"  s_lshl_b32 %[t0], %[en], 4\n" \
							"  v_mov_b32 v52, %[t0]\n" \
							"  ds_write_b16 v52, v53 offset:12\n"   /* front[e.next].prev = slot */ \
							"  s_lshr_b64 s[90:91], s[90:91], 4\n"   /* the symbol is consumed */ \
							"  s_add_u32 %[cler], %[cler], 1\n" \
							"  s_and_b32 %[t0], %[cler], 7\n" \
							"  s_cbranch_scc1 Lpop_%=\n" \
							"  s_lshr_b32 %[t0], %[cler], 3\n" \
							"  s_lshl_b32 %[t0], %[t0], 2\n" \
							"  s_add_u32 %[t0], %[t0], %[clw]\n" \
							"  v_mov_b32 v55, %[t0]\n" \
							"  ds_read_b32 v55, v55\n" \
							"  s_waitcnt lgkmcnt(0)\n" \
							"  v_readfirstlane_b32 s91, v55\n" \
							"Lpop_%=:\n" \
							"  s_cmp_ge_u32 %[cler], %[slideat]\n" \
							"  s_cbranch_scc1 Lended_%=\n" \
							"  s_sub_u32 %[t2], %[nq], %[qpos]\n"   /* queued entries */ \
							"  s_cmp_eq_u32 %[t2], 0\n" \
							"  s_cbranch_scc1 Ldpop_%=\n" \
							"  s_and_b32 %[t0], s90, 0xee\n"       /* the gate about to be popped is ended at once, and the one after it too (BOUNDARY 4 / DELAY 5 twice): the */ \
							"  s_cmp_eq_u32 %[t0], 0x44\n"           /* chain-end step (TOPO_ASM_ENDS) - it costs what two ends cost one at a time, so a single pair goes the old way */ \
							"  s_cbranch_scc1 Lends_%=\n" \
							"Lpop1_%=:\n" \
							"  s_mov_b64 exec, -1\n" \
							"  v_mbcnt_lo_u32_b32 v60, -1, 0\n" \
							"  v_mbcnt_hi_u32_b32 v60, -1, v60\n" \
							"  v_cmp_gt_u32 vcc, %[t2], v60\n" \
							"  s_mov_b64 exec, vcc\n" \
							"  v_add_u32 v61, %[qpos], v60\n" \
							"  v_and_b32 v61, %[mask], v61\n" \
							"  v_lshlrev_b32 v61, 4, v61\n" \
							"  ds_read_b128 v[56:59], v61\n"        /* every lane its whole record: the live one is then a readlane away, not another round trip */ \
							"  s_waitcnt lgkmcnt(0)\n" \
							"  v_cmp_gt_i32 vcc, v58, -1\n"        /* TOPO_DEAD is the sign bit; lanes past the queue's end report 0 */ \
							"  s_mov_b64 exec, 1\n" \
							"  s_cmp_eq_u64 vcc, 0\n" \
							"  s_cbranch_scc0 Lfound_%=\n" \
							"  s_min_u32 %[t2], %[t2], 64\n"       /* all dead: step over them (no symbol consumed, decoder.cpp:278-279) */ \
							"  s_add_u32 %[qpos], %[qpos], %[t2]\n" \
							"  s_branch Lpop_%=\n" \
							"Lfound_%=:\n" \
							"  s_ff1_i32_b64 %[t0], vcc\n" \
							"  s_add_u32 %[qpos], %[qpos], %[t0]\n" \
							"  s_add_u32 %[qpos], %[qpos], 1\n" \
							"  v_readlane_b32 %[v0], v56, %[t0]\n" \
							"  v_readlane_b32 %[v1], v57, %[t0]\n" \
							"  v_readlane_b32 %[v2], v58, %[t0]\n" \
							"  v_readlane_b32 %[t0], v59, %[t0]\n" \
							"  s_and_b32 %[v2], %[v2], 0x3fffffff\n" \
							"  s_and_b32 %[ep], %[t0], 0xffff\n" \
							"  s_lshr_b32 %[en], %[t0], 16\n" \
							"  s_mov_b32 %[nc], -1\n" \
							"  s_branch Ltop_%=\n" \
							"Lsink_%=:\n" \
							"  s_add_u32 %[t1], %[mask], 1\n"      /* pool slot RING: every BOUNDARY edge (k_mesh.hip: THE SINK) */ \
							"  s_branch Llinks_%=\n" \
   /* the ring is empty: the youngest postponed gate (decoder.cpp:266-270); its pool slot goes back to the free list */ \
							"Ldpop_%=:\n" \
							"  s_and_b32 %[t0], %[pk2], 0xffff\n" \
							"  s_cmp_eq_u32 %[t0], 0\n" \
							"  s_cbranch_scc1 Lended_%=\n" \
							"  s_sub_u32 %[pk2], %[pk2], 1\n" \
							"  s_sub_u32 %[t0], %[t0], 1\n" \
							"  s_lshl_b32 %[t0], %[t0], 1\n" \
							"  s_lshr_b32 %[t2], %[lay], 16\n" \
							"  s_lshl_b32 %[t2], %[t2], 4\n" \
							"  s_add_u32 %[t0], %[t0], %[t2]\n" \
							"  v_mov_b32 v52, %[t0]\n" \
							"  ds_read_u16 v52, v52\n" \
							"  s_lshr_b32 %[t0], %[pk1], 16\n"      /* free list: where the slot id goes */ \
							"  s_lshl_b32 %[t0], %[t0], 1\n" \
							"  s_and_b32 %[t3], %[lay], 0xffff\n" \
							"  s_lshl_b32 %[t3], %[t3], 4\n" \
							"  s_add_u32 %[t0], %[t0], %[t3]\n" \
							"  v_mov_b32 v53, %[t0]\n" \
							"  s_add_u32 %[pk1], %[pk1], 0x10000\n" \
							"  s_waitcnt lgkmcnt(0)\n" \
							"  ds_write_b16 v53, v52\n" \
							"  v_lshlrev_b32 v54, 4, v52\n" \
							"  ds_read_b128 v[56:59], v54\n" \
							"  s_waitcnt lgkmcnt(0)\n" \
							"  v_readfirstlane_b32 %[v2], v58\n" \
							"  s_cmp_lt_i32 %[v2], 0\n"              /* deleted while it waited: no symbol consumed, next */ \
							"  s_cbranch_scc1 Lpop_%=\n" \
							"  v_readfirstlane_b32 %[v0], v56\n" \
							"  v_readfirstlane_b32 %[v1], v57\n" \
							"  v_readfirstlane_b32 %[t0], v59\n" \
							"  s_and_b32 %[v2], %[v2], 0x3fffffff\n" \
							"  s_and_b32 %[ep], %[t0], 0xffff\n" \
							"  s_lshr_b32 %[en], %[t0], 16\n" \
							"  s_mov_b32 %[nc], -1\n" \
							"  s_branch Ltop_%=\n" \
   /* ---------------- SPLIT (decoder.cpp:294-309 with a known vertex): VERTEX's step with the opposite vertex read from the split / \
      vertex-id bits instead of made - no prediction triple, no new id.  The bit cursor, the width of an id and the number of bits the \
      staged words hold sit in cold[3..6], 1056 bytes below the symbol window, the words themselves 1024 below it; a field that does \
      not lie inside the staged words is left to the C++ (which also reports a stream that runs out of bits).  vcc is scratch here: \
      the two words as a 64-bit value, then the vertex id. */ \
							"Lsplit_%=:\n" \
							"  s_sub_u32 %[t0], %[nq], %[qpos]\n"    /* a ring slot - NOT a vertex id: the SPLITs of a mesh that zips up come when every vertex has been made */ \
							"  s_cmp_gt_u32 %[t0], %[mask]\n" \
							"  s_cbranch_scc1 Lexit_%=\n" \
							"  s_cmp_lg_u32 %[budget], 0\n"          /* (the VERTEXes' budget counts ring slots too) */ \
							"  s_cselect_b32 %[t0], 1, 0\n" \
							"  s_sub_u32 %[budget], %[budget], %[t0]\n" \
							"  s_lshr_b32 %[t0], %[lay], 15\n"          /* cold[] sits behind the DELAY stack: 16*(2*DL16 - FL16) (the layout word: FL16 | DL16 << 16, FL16 < 2^15) */ \
							"  s_and_b32 %[t1], %[lay], 0xffff\n" \
							"  s_sub_u32 %[t0], %[t0], %[t1]\n" \
							"  s_lshl_b32 %[t0], %[t0], 4\n" \
							"  v_mov_b32 v52, %[t0]\n" \
							"  ds_read2_b32 v[56:57], v52 offset0:3 offset1:4\n" \
							"  ds_read2_b32 v[58:59], v52 offset0:5 offset1:6\n" \
							"  s_waitcnt lgkmcnt(0)\n" \
							"  v_readfirstlane_b32 %[t0], v56\n" \
							"  v_readfirstlane_b32 %[t1], v57\n" \
							"  v_readfirstlane_b32 %[t2], v58\n" \
							"  v_readfirstlane_b32 %[t3], v59\n" \
							"  s_cmp_lg_u32 %[t1], 0\n" \
							"  s_cbranch_scc1 Lexit_%=\n" \
							"  s_add_u32 %[t1], %[t0], %[t2]\n" \
							"  s_cmp_gt_u32 %[t1], %[t3]\n" \
							"  s_cbranch_scc1 Lexit_%=\n" \
							"  v_mov_b32 v53, %[t1]\n" \
							"  ds_write_b32 v52, v53 offset:12\n"     /* the cursor moves on */ \
							"  s_lshr_b32 %[t1], %[t0], 5\n" \
							"  s_lshl_b32 %[t1], %[t1], 2\n" \
							"  v_add_u32 v54, %[t1], v52\n"               /* (v52: cold[]; the staged words start 32 bytes above it) */ \
							"  v_add_u32 v54, 32, v54\n" \
							"  ds_read2_b32 v[56:57], v54 offset1:1\n" \
							"  s_and_b32 %[t0], %[t0], 31\n" \
							"  s_sub_u32 %[t2], 64, %[t2]\n" \
							"  s_waitcnt lgkmcnt(0)\n" \
							"  v_readfirstlane_b32 vcc_hi, v56\n" \
							"  v_readfirstlane_b32 vcc_lo, v57\n" \
							"  s_nop 0\n" \
							"  s_lshl_b64 vcc, vcc, %[t0]\n" \
							"  s_lshr_b64 vcc, vcc, %[t2]\n" \
							"  s_and_b32 vcc_lo, vcc_lo, 0x3fffffff\n"   /* the opposite vertex */ \
							"  s_and_b32 %[t1], %[nq], %[mask]\n" \
							"  s_add_u32 %[nq], %[nq], 1\n" \
							FACE("vcc_lo") \
							"  s_add_u32 %[start], %[start], 3\n" \
							"  s_lshl_b32 %[t0], %[en], 4\n" \
							"  v_mov_b32 v52, %[t0]\n" \
							"  v_mov_b32 v53, %[t1]\n" \
							"  ds_write_b16 v52, v53 offset:12\n" \
							"  s_lshl_b32 %[t2], %[en], 16\n" \
							"  s_or_b32 %[t2], %[t2], 0xffff\n" \
							"  v_mov_b32 v48, vcc_lo\n" \
							"  v_mov_b32 v49, %[v1]\n" \
							"  v_mov_b32 v50, %[v0]\n" \
							"  v_mov_b32 v51, %[t2]\n" \
							"  s_lshl_b32 %[t0], %[t1], 4\n" \
							"  v_mov_b32 v54, %[t0]\n" \
							"  ds_write_b128 v54, v[48:51]\n" \
							"  s_mov_b32 %[nc], %[t1]\n" \
							"  s_mov_b32 %[ncnext], %[en]\n" \
							"  s_mov_b32 %[ncv1], %[v1]\n" \
							"  s_mov_b32 %[v2], %[v1]\n" \
							"  s_mov_b32 %[v1], vcc_lo\n" \
							"  s_mov_b32 %[en], %[t1]\n" \
							TOPO_ASM_TAIL \
							"Lended_%=:\n" \
							"  s_mov_b32 %[c], 0x200\n"             /* nothing is current: the C++ fetches the next gate (slide, DELAY stack, seed face) */ \
							"  s_branch Lexit_%=\n" \
							   /* ---------------- what the next symbols ask for (s90 shows eight of them whatever the alignment: TOPO_ASM_TAIL).  Four of VERTEX / LEFT: the \
							      mix step (TOPO_ASM_MIX) - unless they are the head of a regular run one symbol on (V VLVLVL V: the run step with that VERTEX as \
							      its lead lane; L VLV..: the LEFT here, then the run step); VERTEX LEFT VERTEX LEFT: the run step if all eight go on like that, else \
							      the mix step takes the short run along; L R x x / L L R x: the mix step with its one RIGHT (Lleft above) */ \
							"Lvlead_%=:\n"                         /* V VLVLVL V: the lone VERTEX in front of a regular run rides along with the run step (TOPO_ASM_RUN's */ \
							"  s_cmp_eq_u32 s90, 0x01010100\n"    /* lead lane); else it goes one at a time */ \
							"  s_cbranch_scc1 Lrunv_%=\n" \
							"  s_branch Lvgo_%=\n" \
							"Lvrun_%=:\n"                          /* VERTEX LEFT VERTEX LEFT: the run step if all eight symbols of the window register go on like that - */ \
							"  s_cmp_eq_u32 s90, 0x10101010\n"    /* a shorter run is the mix step's, which takes what follows it too */ \
							"  s_cbranch_scc1 Lrun_%=\n" \
							"  s_branch Lmix_%=\n" \
							"Lvmix_%=:\n" \
							"  s_cmp_eq_u32 %[t0], 0x0100\n" \
							"  s_cbranch_scc1 Lvlead_%=\n" \
							"  s_branch Lmix_%=\n" \
							"Llmix_%=:\n" \
							"  s_and_b32 %[t0], s90, 0xffff\n" \
							"  s_cmp_eq_u32 %[t0], 0x0101\n" \
							"  s_cbranch_scc1 Llgo_%=\n" \
							TOPO_ASM_MIX(MIXFACE, FSHIFT) \
							TOPO_ASM_RUN(RUNFACE, LEADFACE, FSHIFT) \
							TOPO_ASM_ENDS \
							   /* ---------------- after a step: the group may be done, the window may want sliding (both the C++'s business), else the next symbol */ \
							"Lstepped_%=:\n" \
							"  s_and_b32 %[c], %[cler], 7\n"        /* s91 came back as lane k's whole next word: shifted like s90 */ \
							"  s_lshl_b32 %[c], %[c], 2\n" \
							"  s_lshr_b32 s91, s91, %[c]\n" \
							"  s_mov_b32 %[c], 0\n" \
							"  s_cmp_lt_u32 %[start], %[end]\n" \
							"  s_cbranch_scc0 Lexit_%=\n" \
							"  s_cmp_ge_u32 %[cler], %[slideat]\n" \
							"  s_cbranch_scc1 Lexit_%=\n" \
							"  s_branch Ltop_%=\n" \
							"Lexit_%=:\n" \
							"  s_mov_b32 %[sw], s90\n" \
							"  s_mov_b32 %[swn], s91\n" \
							: [sw] "+s"(sw), [swn] "+s"(swn), [cler] "+s"(cler), [vc] "+s"(vc), [nq] "+s"(nq), [start] "+s"(start), \
							  [v0] "+s"(v0), [v1] "+s"(v1), [v2] "+s"(v2), [ep] "+s"(ep), [en] "+s"(en), \
							  [nc] "+s"(nc), [ncnext] "+s"(nc_next), [ncv1] "+s"(nc_v1), \
							  [t0] "=&s"(t0_), [t1] "=&s"(t1_), [t2] "=&s"(t2_), [c] "=&s"(c_), [t3] "=&s"(t3_), [budget] "+s"(budget_), \
							  [pk1] "+s"(pk1), [pk2] "+s"(pk2), [qpos] "+s"(qpos) \
							: [mask] "s"(MASK), [end] "s"(end), [clw] "s"(clw), [slideat] "s"(slide_at), \
							  [lay] "s"(lay), [predb] "s"(predb), [faceb] "s"(faceb) \
							: "memory", "scc", "vcc", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", \
							  "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99" TOPO_ASM_STAMP_CLOBBERS);

// The symbol window, filled by the whole wave: 32 symbols per lane and pass (two 16-byte loads), each byte checked (anything that
// is not one of the seven CLERS symbols, and everything behind the stream, becomes the invalid nibble 15) and squeezed to a nibble
// with SWAR arithmetic, four words written per lane.  Used for the first fill and for every slide of the window, which lane 0
// alone did a byte at a time before: 0.1 ms per slide, 4 ms for a 256K-triangle mesh's 42 slides - more than its whole automaton now.
//   X: four symbol bytes -> (in the low 16 bits) four nibbles
#define TOPO_PACK4(X, T, U) \
	"  v_and_b32 " T ", 0x7f7f7f7f, " X "\n" \
	"  v_add_u32 " T ", 0x79797979, " T "\n" \
	"  v_or_b32 " T ", " T ", " X "\n" \
	"  v_and_b32 " T ", 0x80808080, " T "\n"       /* 0x80 in every byte >= 7 */ \
	"  v_lshrrev_b32 " T ", 7, " T "\n" \
	"  v_lshlrev_b32 " U ", 4, " T "\n" \
	"  v_sub_u32 " T ", " U ", " T "\n"            /* 15 there */ \
	"  v_or_b32 " X ", " X ", " T "\n" \
	"  v_and_b32 " X ", 0x0f0f0f0f, " X "\n" \
	"  v_lshrrev_b32 " T ", 4, " X "\n" \
	"  v_or_b32 " X ", " X ", " T "\n" \
	"  v_and_b32 " X ", 0x00ff00ff, " X "\n" \
	"  v_lshrrev_b32 " T ", 8, " X "\n" \
	"  v_or_b32 " X ", " X ", " T "\n"
//   W = nibbles of LO | nibbles of HI << 16, then every nibble from (rem - 8*I) on set to 15 (symbols behind the stream)
#define TOPO_PACK_WORD(W, LO, HI, I) \
	TOPO_PACK4(LO, "v62", "v63") \
	TOPO_PACK4(HI, "v62", "v63") \
	"  v_and_b32 " LO ", 0xffff, " LO "\n" \
	"  v_lshl_or_b32 " W ", " HI ", 16, " LO "\n" \
	"  v_add_u32 v62, " I ", v42\n" \
	"  v_max_i32 v62, 0, v62\n" \
	"  v_min_u32 v62, 8, v62\n" \
	"  v_lshlrev_b32 v62, 2, v62\n" \
	"  v_lshlrev_b64 v[56:57], v62, v[58:59]\n" \
	"  v_or_b32 " W ", " W ", v56\n"
#define TOPO_FILL_WINDOW(WINBASE) do { \
	uint64_t fsv_, fm0_, fm1_; uint32_t fwb_; \
	asm volatile( \
		"  s_mov_b64 %[sv], exec\n" \
		"  s_mov_b64 exec, -1\n" \
		"  v_mbcnt_lo_u32_b32 v60, -1, 0\n" \
		"  v_mbcnt_hi_u32_b32 v60, -1, v60\n" \
		"  v_lshlrev_b32 v61, 2, v60\n"                  /* a lane's first word of a pass */ \
		"  v_mov_b32 v58, -1\n" \
		"  v_mov_b32 v59, 0\n" \
		"  s_mov_b32 %[wb], 0\n" \
		"Lfpass_%=:\n" \
		"  v_add_u32 v40, %[wb], v61\n"                  /* word w of the window */ \
		"  v_cmp_gt_u32 vcc, %[symwords], v40\n" \
		"  s_and_saveexec_b64 %[m0], vcc\n" \
		"  s_cbranch_execz Lfnext_%=\n" \
		"  v_lshlrev_b32 v41, 3, v40\n" \
		"  v_add_u32 v41, %[winbase], v41\n"             /* its first symbol */ \
		"  v_sub_u32 v42, %[nclers], v41\n"              /* symbols of the stream from there on (signed) */ \
		"  v_mov_b32 v44, 0\n" \
		"  v_mov_b32 v45, 0\n" \
		"  v_mov_b32 v46, 0\n" \
		"  v_mov_b32 v47, 0\n" \
		"  v_mov_b32 v48, 0\n" \
		"  v_mov_b32 v49, 0\n" \
		"  v_mov_b32 v50, 0\n" \
		"  v_mov_b32 v51, 0\n" \
		"  v_cmp_lt_i32 vcc, 0, v42\n" \
		"  s_and_saveexec_b64 %[m1], vcc\n" \
		"  global_load_dwordx4 v[44:47], v41, %[gcl]\n" \
		"  s_mov_b64 exec, %[m1]\n" \
		"  v_cmp_lt_i32 vcc, 16, v42\n" \
		"  s_and_saveexec_b64 %[m1], vcc\n" \
		"  global_load_dwordx4 v[48:51], v41, %[gcl] offset:16\n" \
		"  s_mov_b64 exec, %[m1]\n" \
		"  s_waitcnt vmcnt(0)\n" \
		TOPO_PACK_WORD("v52", "v44", "v45", "0") \
		TOPO_PACK_WORD("v53", "v46", "v47", "-8") \
		TOPO_PACK_WORD("v54", "v48", "v49", "-16") \
		TOPO_PACK_WORD("v55", "v50", "v51", "-24") \
		"  v_lshlrev_b32 v40, 2, v40\n" \
		"  v_add_u32 v40, %[clbase], v40\n" \
		"  ds_write_b128 v40, v[52:55]\n" \
		"Lfnext_%=:\n" \
		"  s_mov_b64 exec, -1\n" \
		"  s_add_u32 %[wb], %[wb], 256\n" \
		"  s_cmp_lt_u32 %[wb], %[symwords]\n" \
		"  s_cbranch_scc1 Lfpass_%=\n" \
		"  s_mov_b64 exec, 3\n"                          /* two words of "window exhausted" nibbles behind it */ \
		"  v_add_u32 v40, %[symwords], v60\n" \
		"  v_lshlrev_b32 v40, 2, v40\n" \
		"  v_add_u32 v40, %[clbase], v40\n" \
		"  v_mov_b32 v41, 0xeeeeeeee\n" \
		"  ds_write_b32 v40, v41\n" \
		"  s_waitcnt lgkmcnt(0)\n" \
		"  s_mov_b64 exec, %[sv]\n" \
		: [sv] "=&s"(fsv_), [m0] "=&s"(fm0_), [m1] "=&s"(fm1_), [wb] "=&s"(fwb_) \
		: [winbase] "s"(WINBASE), [nclers] "s"(nclers), [symwords] "s"(symwords), [clbase] "s"((uint32_t)(uintptr_t)cl32), [gcl] "s"(gcl) \
		: "memory", "scc", "vcc", "v40", "v41", "v42", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", \
		  "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); } while(0)

// The run step.  A sphere-like mesh's symbol stream is "VERTEX LEFT" repeated (89 % of a 4K-triangle blob's symbols sit in 120
// such runs of 16 pairs on average, 98 % of a 256K-triangle mesh's in runs of 126), and a run is regular: its LEFTs close
// against the edges the previous layer queued, which sit in CONSECUTIVE ring slots linked prev -> slot + 1, its VERTEXes
// take consecutive vertex ids and ring slots.  So pair j of a run is a function of the state before the run, j, and the
// records of ring slots ep+j-2 .. ep+j - and the wave's 63 parked lanes are switched on to do up to 63 pairs in one pass:
//   lane j reads its pair of symbols and the records of slots ep+j, ep+j-1, ep+j-2 (x = v0, w = links);
//   valid_j = symbols are (VERTEX, LEFT) && link(ep+j-1).prev == ep+j (j > 0) && ep+j != e.next && j < kmax; k = first invalid lane;
//   with a_j = j ? x[j-1] : v0,  b_j = j ? vc+j-1 : v1,  c_j = j > 1 ? x[j-2] : j ? v0 : v2   (the current edge before pair j is
//   (a_j, b_j, c_j)), lanes j < k write: prediction[vc+j] = (b_j, a_j, c_j); faces (b_j, a_j, vc+j) and (vc+j, a_j, x[j]);
//   the queued edge's record rec[nq+j] = {vc+j, b_j, a_j, prev = nq+j+1 (lazy for the last), next = nq+j-1 (e.next for the first)};
//   the deleted flag of slot ep+j.  Lane 0 links e.next.prev = nq.  The state after the run comes from lane k-1 (readlane),
//   the symbol window registers from lane k (the words it read are the ones the serial loop continues with).
// (e.next's prev link is stale while the current edge is lazy, hence the ep+j != e.next cut; every other prev link the chain
// follows belongs to a live queued edge.)  Checked against the oracle on the host model tools/topo_run_model.py before it was
// written here.  Wait states for gfx950's VALU-SGPR / readlane / wide-store hazards are inside the string (s_nop).
#define TOPO_RUN_FACE32 \
	"  v_mov_b32 v35, v34\n" \
	"  v_mov_b32 v36, v33\n" \
	"  v_mov_b32 v37, v46\n" \
	"  v_mad_u32_u24 v59, v60, 24, %[c]\n" \
	"  global_store_dwordx4 v59, v[32:35], %[faceb]\n" \
	"  global_store_dwordx2 v59, v[36:37], %[faceb] offset:16\n"
#define TOPO_LEAD_FACE32 \
	"  v_mov_b32 v59, -12\n" \
	"  v_add_u32 v59, %[c], v59\n" \
	"  global_store_dwordx3 v59, v[32:34], %[faceb]\n"
#define TOPO_LEAD_FACE16 \
	"  v_and_b32 v36, 0xffff, v32\n" \
	"  v_lshl_or_b32 v36, v33, 16, v36\n" \
	"  v_mov_b32 v59, -6\n" \
	"  v_add_u32 v59, %[c], v59\n" \
	"  global_store_dword v59, v36, %[faceb]\n" \
	"  global_store_short v59, v34, %[faceb] offset:4\n"
#define TOPO_RUN_FACE16 \
	"  v_and_b32 v36, 0xffff, v32\n" \
	"  v_lshl_or_b32 v36, v33, 16, v36\n" \
	"  v_and_b32 v37, 0xffff, v34\n" \
	"  v_lshl_or_b32 v37, v34, 16, v37\n" \
	"  v_and_b32 v38, 0xffff, v33\n" \
	"  v_lshl_or_b32 v38, v46, 16, v38\n" \
	"  v_mad_u32_u24 v59, v60, 12, %[c]\n" \
	"  global_store_dwordx3 v59, v[36:38], %[faceb]\n"
// Both steps are sections of the ISA block (TOPO_FAST_PATH jumps here from its VERTEX / LEFT entries and dispatches the next symbol
// afterwards).  As asm statements of their own each paid an exit, forty-odd scalar instructions of compiler-made state shuffling either
// side and the way back in; measured, that was worth 7 % on irregular blobs and 5 % on the 128K-vertex mesh, nothing on the regular
// blob (the stamps that had blamed the glue for a quarter of its time were mostly measuring themselves: DESIGN.md 3.1).  Inside they
// live on the block's five scalar temporaries and vcc: conditions are chained by narrowing exec (each compare sees the lanes that passed
// the ones before - and a mask rebuilt with v_cmp is only as wide as exec is at that moment: set exec to -1 first), masks are rebuilt
// from per-lane values where they are needed again, exec is known to be lane 0 on entry.  FSHIFT: log2 of an index's bytes.
#define TOPO_ASM_WINDOW_LEFT /* t2 = symbols the LDS window holds from cler on (63+ when it reaches the end of the stream) */ \
	"  s_add_u32 %[t2], %[slideat], 2048\n" \
	"  s_sub_u32 %[t2], %[t2], %[cler]\n" \
	"  s_max_i32 %[t2], %[t2], 0\n" \
	"  s_cmp_eq_u32 %[slideat], -1\n" \
	"  s_cselect_b32 %[t2], 126, %[t2]\n"
#define TOPO_ASM_RUN(FACE, LEADFACE, FSHIFT) \
	"Lrunv_%=:\n"                                   /* a lone VERTEX, then the run: the state moves on as the VERTEX leaves it (what it was: s92 - s94), and */ \
	"  s_cmp_eq_u32 %[budget], 0\n"                  /* lane 63 - never a pair's: kmax <= 63 - writes what the VERTEX writes, in the pairs' own stores */ \
	"  s_cbranch_scc1 Lvgo_%=\n" \
	"  s_cmp_gt_u32 %[ep], %[mask]\n" \
	"  s_cbranch_scc1 Lvgo_%=\n" \
	"  s_mov_b32 s92, %[v1]\n" \
	"  s_mov_b32 s93, %[v2]\n" \
	"  s_mov_b32 s94, %[en]\n" \
	"  s_mov_b32 %[v2], %[v1]\n" \
	"  s_mov_b32 %[v1], %[vc]\n" \
	"  s_and_b32 %[en], %[nq], %[mask]\n" \
	"  s_add_u32 %[vc], %[vc], 1\n" \
	"  s_add_u32 %[nq], %[nq], 1\n" \
	"  s_add_u32 %[start], %[start], 3\n" \
	"  s_add_u32 %[cler], %[cler], 1\n" \
	"  s_sub_u32 %[budget], %[budget], 1\n" \
	"  s_mov_b32 s96, 0\n"                           /* s[96:97]: the lead lane's exec bit, or nothing */ \
	"  s_brev_b32 s97, 1\n" \
	"  s_lshl_b32 %[t0], s94, 4\n"                   /* the lone VERTEX' own link NOW, not with the step's other writes: what was e.next gets the queued edge for */ \
	"  v_mov_b32 v42, %[en]\n"                       /* prev, and a run that closes the previous layer's slots right up to that edge reads this very link (small */ \
	"  v_mov_b32 v43, %[t0]\n"                       /* closed meshes: tools/stress_topology.py found it).  Not undone if no pair joins: the one-at-a-time */ \
	"  ds_write_b16 v43, v42 offset:12\n"            /* VERTEX that follows writes the same slot there */ \
	"  s_branch Lrunb_%=\n" \
	"Lrun_%=:\n" \
	"  s_cmp_gt_u32 %[ep], %[mask]\n"                 /* e.prev in the pool: one symbol at a time */ \
	"  s_cbranch_scc1 Lvgo_%=\n" \
	"  s_mov_b64 s[96:97], 0\n" \
	"Lrunb_%=:\n" \
	"  s_sub_u32 %[t3], %[end], %[start]\n"           /* kmax = min(63, vertex ids / ring slots left, pairs the group and the window hold) */ \
	"  s_mul_hi_u32 %[t3], %[t3], 0xaaaaaaab\n" \
	"  s_lshr_b32 %[t3], %[t3], 2\n" \
	"  s_min_u32 %[t3], %[t3], 63\n" \
	"  s_min_u32 %[t3], %[t3], %[budget]\n" \
	TOPO_ASM_WINDOW_LEFT \
	"  s_lshr_b32 %[t2], %[t2], 1\n" \
	"  s_min_u32 %[t3], %[t3], %[t2]\n" \
	"  s_mov_b64 exec, -1\n" \
	"  v_mbcnt_lo_u32_b32 v60, -1, 0\n" \
	"  v_mbcnt_hi_u32_b32 v60, -1, v60\n"            /* v60 = j */ \
	"  v_lshlrev_b32 v40, 1, v60\n" \
	"  v_add_u32 v40, %[cler], v40\n"                /* p = cler + 2j: the pair's first symbol */ \
	"  v_lshrrev_b32 v41, 3, v40\n" \
	"  v_lshlrev_b32 v41, 2, v41\n" \
	"  v_add_u32 v41, %[clw], v41\n" \
	"  v_add_u32 v41, -4, v41\n"                     /* (the word of symbol p is cl32[(p >> 3) + wbias - 1]) */ \
	"  ds_read2_b32 v[42:43], v41 offset1:1\n"       /* the word holding symbol p and the next one */ \
	"  v_add_u32 v44, %[ep], v60\n" \
	"  v_add_u32 v48, -1, v44\n" \
	"  v_add_u32 v49, -2, v44\n" \
	"  v_and_b32 v44, %[mask], v44\n"                /* slot ep+j */ \
	"  v_and_b32 v48, %[mask], v48\n" \
	"  v_and_b32 v49, %[mask], v49\n" \
	"  v_lshlrev_b32 v45, 4, v44\n" \
	"  v_lshlrev_b32 v48, 4, v48\n" \
	"  v_lshlrev_b32 v49, 4, v49\n" \
	"  ds_read2_b32 v[46:47], v45 offset1:3\n"       /* x[j], w[j] */ \
	"  ds_read2_b32 v[50:51], v48 offset1:3\n"       /* x[j-1], w[j-1] */ \
	"  ds_read_b32 v52, v49\n"                       /* x[j-2] */ \
	"  v_and_b32 v53, 7, v40\n" \
	"  v_lshlrev_b32 v53, 2, v53\n" \
	"  s_waitcnt lgkmcnt(0)\n" \
	"  v_lshrrev_b64 v[54:55], v53, v[42:43]\n"      /* v54: the symbols from p on, eight nibbles */ \
	"  v_and_b32 v56, 0xff, v54\n" \
	"  v_and_b32 v57, 0xffff, v51\n"                 /* rec[ep+j-1].prev */ \
	"  v_cmp_eq_u32 vcc, 0, v60\n" \
	"  v_cndmask_b32 v57, v57, v44, vcc\n"           /* (lane 0 has no link to check) */ \
	"  v_cmp_eq_u32 vcc, 16, v56\n"                  /* (VERTEX, LEFT) */ \
	"  s_and_b64 exec, exec, vcc\n" \
	"  v_cmp_eq_u32 vcc, v57, v44\n"                 /* link(ep+j-1).prev == slot ep+j */ \
	"  s_and_b64 exec, exec, vcc\n" \
	"  v_cmp_ne_u32 vcc, %[en], v44\n" \
	"  s_and_b64 exec, exec, vcc\n" \
	"  v_cmp_gt_u32 vcc, %[t3], v60\n" \
	"  s_and_b64 exec, exec, vcc\n" \
	"  s_not_b64 vcc, exec\n" \
	"  s_ff1_i32_b64 %[t0], vcc\n"                   /* k: the first lane that cannot join (kmax < 64: there is one) */ \
	"  s_cmp_eq_u32 %[t0], 0\n" \
	"  s_cbranch_scc1 Lrun0_%=\n" \
	"  s_sub_u32 %[t1], %[t0], 1\n" \
	"  s_lshl_b32 %[c], %[start], " FSHIFT "\n"      /* byte offset of the step's first index */ \
	"  s_bfm_b64 exec, %[t0], 0\n"                   /* lanes 0 .. k-1 */ \
	"  s_or_b64 exec, exec, s[96:97]\n"              /* ... and the lead lane */ \
	"  v_mov_b32 v38, %[v0]\n" \
	"  v_mov_b32 v39, %[v1]\n" \
	"  v_add_u32 v34, %[vc], v60\n"                  /* the new vertex vc+j */ \
	"  v_add_u32 v32, -1, v34\n" \
	"  v_cmp_ne_u32 vcc, 0, v60\n" \
	"  v_cndmask_b32 v33, v38, v50, vcc\n"           /* a_j */ \
	"  v_cndmask_b32 v32, v39, v32, vcc\n"           /* b_j */ \
	"  v_mov_b32 v39, %[v2]\n" \
	"  v_cndmask_b32 v58, v39, v38, vcc\n" \
	"  v_cmp_lt_u32 vcc, 1, v60\n" \
	"  v_cndmask_b32 v58, v58, v52, vcc\n"           /* c_j */ \
	"  s_sub_u32 %[t2], %[vc], 1\n"                  /* the lead lane: the edge as the lone VERTEX found it, and the vertex it made */ \
	"  v_writelane_b32 v32, s92, 63\n" \
	"  v_writelane_b32 v33, %[v0], 63\n" \
	"  v_writelane_b32 v58, s93, 63\n" \
	"  v_writelane_b32 v34, %[t2], 63\n" \
	"  v_mov_b32 v56, v32\n" \
	"  v_mov_b32 v57, v33\n" \
	"  v_mul_lo_u32 v59, v34, 12\n" \
	"  global_store_dwordx3 v59, v[56:58], %[predb]\n" \
	"  s_andn2_b64 exec, exec, s[96:97]\n" \
	FACE \
	"  s_mov_b64 exec, s[96:97]\n"                   /* the lone VERTEX' one face, in front of the pairs' */ \
	"  s_cbranch_execz Lrunf_%=\n" \
	LEADFACE \
	"Lrunf_%=:\n" \
	"  s_bfm_b64 exec, %[t0], 0\n" \
	"  s_or_b64 exec, exec, s[96:97]\n" \
	"  v_add_u32 v40, %[nq], v60\n" \
	"  v_add_u32 v53, 1, v40\n" \
	"  v_add_u32 v55, -1, v40\n" \
	"  v_and_b32 v41, %[mask], v40\n"                /* its slot nq+j */ \
	"  v_and_b32 v53, %[mask], v53\n" \
	"  v_and_b32 v55, %[mask], v55\n" \
	"  v_mov_b32 v61, %[en]\n" \
	"  v_cmp_ne_u32 vcc, 0, v60\n" \
	"  v_cndmask_b32 v55, v61, v55, vcc\n"           /* next: e.next for the first, else slot nq+j-1 */ \
	"  v_mov_b32 v61, 0xffff\n" \
	"  v_cmp_eq_u32 vcc, %[t1], v60\n" \
	"  v_cndmask_b32 v53, v53, v61, vcc\n"           /* prev: slot nq+j+1, lazy for the last */ \
	"  v_lshl_or_b32 v51, v55, 16, v53\n" \
	"  v_mov_b32 v48, v34\n" \
	"  v_mov_b32 v49, v32\n" \
	"  v_mov_b32 v50, v33\n" \
	"  v_lshlrev_b32 v41, 4, v41\n" \
	"  s_sub_u32 %[t2], %[nq], 1\n"                  /* the lead lane's record: the slot before the pairs', next = e.next as it was, prev = the first pair's slot */ \
	"  s_and_b32 %[t2], %[t2], %[mask]\n" \
	"  s_lshl_b32 %[t2], %[t2], 4\n" \
	"  v_writelane_b32 v41, %[t2], 63\n" \
	"  s_and_b32 %[t3], %[nq], %[mask]\n" \
	"  s_lshl_b32 %[c], s94, 16\n" \
	"  s_or_b32 %[t3], %[t3], %[c]\n" \
	"  v_writelane_b32 v51, %[t3], 63\n" \
	"  ds_write_b128 v41, v[48:51]\n" \
	"  s_andn2_b64 exec, exec, s[96:97]\n" \
	"  v_mov_b32 v61, 0x8000\n" \
	"  ds_write_b16 v45, v61 offset:10\n"            /* slot ep+j: deleted */ \
	"  v_readlane_b32 s90, v54, %[t0]\n"            /* the window registers: lane k's eight symbols, and what is left of the word behind them (TOPO_ASM_PAIR below) */ \
	"  v_readlane_b32 s91, v43, %[t0]\n" \
	"  v_readlane_b32 %[t2], v47, %[t1]\n"           /* the state after the run: lane k-1's */ \
	"  v_readlane_b32 %[v0], v46, %[t1]\n" \
	"  v_readlane_b32 %[v2], v33, %[t1]\n" \
	"  v_readlane_b32 %[ncv1], v32, %[t1]\n" \
	"  s_mov_b64 exec, 1\n" \
	"  s_and_b32 %[t3], %[nq], %[mask]\n"            /* e.next.prev = the first new slot */ \
	"  s_lshl_b32 %[c], %[en], 4\n" \
	"  v_mov_b32 v40, %[t3]\n" \
	"  v_mov_b32 v41, %[c]\n" \
	"  ds_write_b16 v41, v40 offset:12\n" \
	"  s_and_b32 %[ep], %[t2], 0xffff\n" \
	"  s_add_u32 %[t3], %[nq], %[t0]\n" \
	"  s_sub_u32 %[c], %[t3], 2\n" \
	"  s_and_b32 %[c], %[c], %[mask]\n" \
	"  s_cmp_eq_u32 %[t0], 1\n" \
	"  s_cselect_b32 %[ncnext], %[en], %[c]\n"       /* (next, v1) of the last queued edge, cached for a RIGHT */ \
	"  s_add_u32 %[v1], %[vc], %[t1]\n" \
	"  s_sub_u32 %[c], %[t3], 1\n" \
	"  s_and_b32 %[en], %[c], %[mask]\n" \
	"  s_mov_b32 %[nc], %[en]\n" \
	"  s_add_u32 %[vc], %[vc], %[t0]\n" \
	"  s_mov_b32 %[nq], %[t3]\n" \
	"  s_sub_u32 %[budget], %[budget], %[t0]\n" \
	"  s_mul_i32 %[c], %[t0], 6\n" \
	"  s_add_u32 %[start], %[start], %[c]\n" \
	"  s_lshl_b32 %[c], %[t0], 1\n" \
	"  s_add_u32 %[cler], %[cler], %[c]\n" \
	"  s_branch Lstepped_%=\n" \
	"Lrun0_%=:\n"                                    /* not even one pair: the VERTEX goes the one-at-a-time way (a lead VERTEX: the state as it was) */ \
	"  s_mov_b64 exec, 1\n" \
	"  s_cmp_eq_u32 s97, 0\n" \
	"  s_cbranch_scc1 Lvgo_%=\n" \
	"  s_mov_b32 %[v1], s92\n" \
	"  s_mov_b32 %[v2], s93\n" \
	"  s_mov_b32 %[en], s94\n" \
	"  s_sub_u32 %[vc], %[vc], 1\n" \
	"  s_sub_u32 %[nq], %[nq], 1\n" \
	"  s_sub_u32 %[start], %[start], 3\n" \
	"  s_sub_u32 %[cler], %[cler], 1\n" \
	"  s_add_u32 %[budget], %[budget], 1\n" \
	"  s_branch Lvgo_%=\n"

// The mix step.  Meshes whose quads are not split the same way everywhere (anything that is not a regular grid) do not give (VERTEX LEFT)
// runs: their streams are still nine parts in ten VERTEX and LEFT, but in any order (VVLL, VLLV ...: two thirds of the symbols of a
// 4K-triangle blob with random diagonals sat in the one-symbol-at-a-time path above, 0.65 ms where the regular blob takes 0.20).  Any
// sequence over those two symbols is as regular as the pairs are, seen the right way: VERTEX changes v1 (to the new vertex) and e.next (to
// the edge it queues) and nothing else, LEFT changes v0 (to the next vertex of the chain of ring slots behind e.prev) and e.prev and
// nothing else.  So with nV_j / nL_j = the VERTEXes / LEFTs in front of symbol j (two ballots, two v_mbcnt) and x[i] = the v0 of ring
// slot ep+i, the edge before symbol j is
//   a_j (v0) = nL_j ? x[nL_j - 1] : v0      b_j (v1) = nV_j ? vc + nV_j - 1 : v1
//   c_j (v2) = j == 0 ? v2 : symbol j-1 was a VERTEX ? (nV_j >= 2 ? vc + nV_j - 2 : v1) : (nL_j >= 2 ? x[nL_j - 2] : v0)
// and lane j - ONE symbol a lane, up to 63 a pass - writes the face (b_j, a_j, opp_j), opp_j = vc + nV_j (VERTEX) or x[nL_j] (LEFT); a
// VERTEX lane also its prediction triple (b_j, a_j, c_j) and the record of the edge it queues (slot nq + nV_j: prev = the next VERTEX's
// slot, lazy for the last; next = the previous VERTEX's slot, e.next for the first), a LEFT lane the deleted flag of slot ep + nL_j.
// The step ends at the first lane whose symbol is something else, whose LEFT would need a chain slot behind a broken link (slot ep+i is
// usable while link(ep+i-1).prev == ep+i and ep+i != e.next, as in the run step), whose VERTEX has no vertex id or ring slot left, or
// from which on a regular run of SIXTEEN symbols lies ahead (the run step does two symbols a lane; a shorter run is cheaper taken along:
// round 4).  The state after it is lane k's (a, b, c).
// Round 4: ONE RIGHT, if the step was entered for it (Lmixr: L R x x / L L R x - what follows a DELAY) and no VERTEX stands in front of it.
// A RIGHT touches the other side of the current edge only: it closes against e.next as the step finds it (t = that record, one broadcast
// read), and behind it v1 = t.v1, e.next = t.next - so with r = its lane, b_j takes t.v1 for v1 when j > r, c_{r+1} = the old v1, the
// first VERTEX' record links to t.next, and the chain of slots the LEFTs may close stops in front of t.next as well as in front of
// e.next: the first VERTEX behind the RIGHT rewrites that slot's prev link, which a lane closing it would have read already (a small
// closed front comes round to it - tools/topo_run_model.py `wide` found it, the dozen meshes it used to run did not).
// Two LDS round trips (symbols and links; then a lane's three x, whose addresses depend on nL_j).  Checked against the oracle on the
// host model (tools/topo_run_model.py) before it was written here.
#define TOPO_MIX_FACE32 \
	"  v_mad_u32_u24 v59, v60, 12, %[c]\n" \
	"  global_store_dwordx3 v59, v[32:34], %[faceb]\n"
#define TOPO_MIX_FACE16 \
	"  v_and_b32 v36, 0xffff, v32\n" \
	"  v_lshl_or_b32 v36, v33, 16, v36\n" \
	"  v_mad_u32_u24 v59, v60, 6, %[c]\n" \
	"  global_store_dword v59, v36, %[faceb]\n" \
	"  global_store_short v59, v34, %[faceb] offset:4\n"
#define TOPO_ASM_MIX(FACE, FSHIFT) \
	"Lmix_%=:\n"                                     /* entered on four symbols of VERTEX / LEFT: no RIGHT is looked for (s92 = 64: none; what a step pays */ \
	"  s_mov_b32 s92, 64\n"                          /* for looking is ~40 instructions and a 64-lane LDS read, and a regular blob's mix steps never meet one) */ \
	"  s_mov_b64 s[96:97], 0\n" \
	"  s_mov_b32 s94, -1\n" \
	"  s_branch Lmixb_%=\n" \
	"Lmixr_%=:\n"                                    /* entered on L R .. / L L R ..: the step takes its one RIGHT */ \
	"  s_mov_b32 s92, 0\n" \
	"Lmixb_%=:\n" \
	"  s_cmp_gt_u32 %[ep], %[mask]\n"                 /* e.prev in the pool: one symbol at a time */ \
	"  s_cbranch_scc1 Lmix0_%=\n" \
	"  s_sub_u32 %[t3], %[end], %[start]\n"           /* kmax = min(63, faces the group and symbols the window hold) */ \
	"  s_mul_hi_u32 %[t3], %[t3], 0xaaaaaaab\n" \
	"  s_lshr_b32 %[t3], %[t3], 1\n" \
	"  s_min_u32 %[t3], %[t3], 63\n" \
	TOPO_ASM_WINDOW_LEFT \
	"  s_min_u32 %[t3], %[t3], %[t2]\n" \
	"  s_mov_b64 exec, -1\n" \
	"  v_mbcnt_lo_u32_b32 v60, -1, 0\n" \
	"  v_mbcnt_hi_u32_b32 v60, -1, v60\n"            /* v60 = j */ \
	"  v_add_u32 v40, %[cler], v60\n"                /* p = cler + j: the lane's symbol */ \
	"  v_lshrrev_b32 v41, 3, v40\n" \
	"  v_lshlrev_b32 v41, 2, v41\n" \
	"  v_add_u32 v41, %[clw], v41\n" \
	"  v_add_u32 v41, -4, v41\n" \
	"  ds_read2_b32 v[42:43], v41 offset1:1\n"       /* the word holding symbol p and the next one */ \
	"  v_add_u32 v44, %[ep], v60\n" \
	"  v_add_u32 v48, -1, v44\n" \
	"  v_and_b32 v44, %[mask], v44\n"                /* slot ep+j */ \
	"  v_and_b32 v48, %[mask], v48\n" \
	"  v_lshlrev_b32 v45, 4, v44\n" \
	"  v_lshlrev_b32 v48, 4, v48\n" \
	"  ds_read_b32 v47, v45 offset:12\n"             /* w[j]: the links of slot ep+j */ \
	"  ds_read_b32 v51, v48 offset:12\n"             /* w[j-1] */ \
	"  s_cmp_eq_u32 s92, 64\n" \
	"  s_cbranch_scc1 Lmixa_%=\n" \
	"  s_lshl_b32 %[t0], %[en], 4\n"                 /* e.next's record (every lane the same address: a broadcast), for the step's one RIGHT */ \
	"  v_mov_b32 v52, %[t0]\n" \
	"  ds_read_b128 v[36:39], v52\n" \
	"Lmixa_%=:\n" \
	"  v_and_b32 v53, 7, v40\n" \
	"  v_lshlrev_b32 v53, 2, v53\n" \
	"  s_waitcnt lgkmcnt(0)\n" \
	"  v_lshrrev_b64 v[54:55], v53, v[42:43]\n"      /* v54: the symbols from p on, eight nibbles */ \
	"  v_and_b32 v56, 15, v54\n"                     /* v56: the lane's symbol */ \
	"  v_and_b32 v57, 0xffff, v51\n"                 /* rec[ep+j-1].prev */ \
	"  v_cmp_ne_u32 vcc, 0, v60\n" \
	"  v_cndmask_b32 v35, 0, v54, vcc\n"             /* (lane 0 may be the head of a regular run: the run step had its chance) */ \
	"  s_cmp_eq_u32 s92, 64\n" \
	"  s_cbranch_scc1 Lmixc_%=\n" \
	"  v_readfirstlane_b32 s93, v37\n"               /* s93: e.next's v1 (the RIGHT's opposite vertex, v1 behind it), s94: its next (e.next behind it), s95: its flags */ \
	"  v_readfirstlane_b32 s94, v39\n" \
	"  v_readfirstlane_b32 s95, v38\n" \
	"  v_cmp_eq_u32 vcc, 2, v56\n"                   /* the step's one RIGHT: the first, and only with no VERTEX in front of it (it closes against e.next */ \
	"  s_ff1_i32_b64 s92, vcc\n"                     /* as the step finds it: nothing before it has touched that side).  s92: its lane, 64: none; */ \
	"  v_cmp_eq_u32 vcc, 0, v56\n"                   /* s[96:97]: its exec bit */ \
	"  s_ff1_i32_b64 %[t1], vcc\n" \
	"  s_lshr_b32 s94, s94, 16\n" \
	"  s_cmp_lt_u32 s92, %[t1]\n"                    /* (none: -1 = 0xffffffff, never below) */ \
	"  s_cselect_b32 s92, s92, 64\n" \
	"  s_cselect_b32 %[t1], 1, 0\n" \
	"  s_bfm_b64 s[96:97], %[t1], s92\n" \
	"Lmixc_%=:\n" \
	"  v_cmp_gt_u32 vcc, 2, v56\n"                   /* VERTEX or LEFT (or that RIGHT) ... */ \
	"  s_or_b64 vcc, vcc, s[96:97]\n" \
	"  s_and_b64 exec, exec, vcc\n" \
	"  v_cmp_gt_u32 vcc, %[t3], v60\n"               /* ... within kmax ... */ \
	"  s_and_b64 exec, exec, vcc\n" \
	"  v_cmp_eq_u32 vcc, 0x10101010, v35\n"          /* ... and no regular run of SIXTEEN symbols from here on (lane j + 8 sees its second half; a */ \
	"  s_lshr_b64 s[98:99], vcc, 8\n"                /* shorter one costs the run step what it costs this one to take it along, and a step less) */ \
	"  s_and_b64 s[98:99], s[98:99], vcc\n" \
	"  s_andn2_b64 exec, exec, s[98:99]\n" \
	"  s_not_b64 vcc, exec\n" \
	"  s_ff1_i32_b64 %[t0], vcc\n"                   /* the first lane out as far as its own symbol goes (kmax < 64: there is one) */ \
	"  s_mov_b64 exec, -1\n" \
	"  s_cmp_eq_u32 %[t0], 0\n" \
	"  s_cbranch_scc1 Lmix0_%=\n" \
	"  s_cmp_lt_u32 s92, %[t0]\n"                    /* (a RIGHT behind the first lane out is not this step's) */ \
	"  s_cselect_b32 s92, s92, 64\n" \
	"  s_cselect_b32 %[t1], 1, 0\n" \
	"  s_bfm_b64 s[96:97], %[t1], s92\n" \
	"  v_cmp_gt_u32 vcc, %[t0], v60\n" \
	"  v_cndmask_b32 v56, 15, v56, vcc\n"            /* (from it on: no symbol) */ \
	"  v_cmp_eq_u32 vcc, 0, v60\n" \
	"  v_cndmask_b32 v57, v57, v44, vcc\n"           /* (lane 0 has no link to check) */ \
	"  v_cmp_eq_u32 vcc, v57, v44\n"                 /* link(ep+j-1).prev == slot ep+j */ \
	"  s_and_b64 exec, exec, vcc\n" \
	"  v_cmp_ne_u32 vcc, %[en], v44\n" \
	"  s_and_b64 exec, exec, vcc\n" \
	"  v_cmp_ne_u32 vcc, s94, v44\n"                 /* (nor what is e.next behind a RIGHT: the first VERTEX behind it rewrites that slot's prev link, and a lane */ \
	"  s_and_b64 exec, exec, vcc\n"                  /* closing it has read the link already - a small closed front comes round to it; cutting the chain is always safe) */ \
	"  s_not_b64 vcc, exec\n" \
	"  s_ff1_i32_b64 %[t1], vcc\n"                   /* C: chain slots ep .. ep+C-1 are usable (-1: all 64) */ \
	"  s_mov_b64 exec, -1\n" \
	"  v_cmp_eq_u32 vcc, 0, v56\n" \
	"  s_nop 1\n"                                   /* (gfx90a+: a VALU write of an SGPR / vcc needs two wait states before a VALU reads it as an operand) */ \
	"  v_mbcnt_lo_u32_b32 v38, vcc_lo, 0\n" \
	"  v_mbcnt_hi_u32_b32 v38, vcc_hi, v38\n"        /* nV_j */ \
	"  v_mov_b32 v35, %[t1]\n" \
	"  v_mov_b32 v36, %[budget]\n" \
	"  v_cndmask_b32 v35, v35, v36, vcc\n"           /* what the lane's symbol is bounded by: chain slots (LEFT), vertex ids / ring slots (VERTEX) */ \
	"  v_cmp_eq_u32 vcc, 1, v56\n" \
	"  s_nop 1\n" \
	"  v_mbcnt_lo_u32_b32 v39, vcc_lo, 0\n" \
	"  v_mbcnt_hi_u32_b32 v39, vcc_hi, v39\n"        /* nL_j */ \
	"  v_cndmask_b32 v36, v38, v39, vcc\n"           /* ... and the number it would take */ \
	"  v_cmp_gt_u32 vcc, 2, v56\n" \
	"  s_or_b64 vcc, vcc, s[96:97]\n"                /* (the RIGHT takes neither a chain slot nor a vertex id) */ \
	"  s_and_b64 exec, exec, vcc\n" \
	"  v_cmp_lt_u32 vcc, v36, v35\n" \
	"  s_or_b64 vcc, vcc, s[96:97]\n" \
	"  s_and_b64 exec, exec, vcc\n" \
	"  s_not_b64 vcc, exec\n" \
	"  s_ff1_i32_b64 %[t0], vcc\n"                   /* k: the symbols of this step */ \
	"  s_mov_b64 exec, -1\n" \
	"  s_cmp_eq_u32 %[t0], 0\n" \
	"  s_cbranch_scc1 Lmix0_%=\n" \
	"  s_cmp_lt_u32 s92, %[t0]\n"                    /* the RIGHT is one of them: e.next moves on for everything behind it (s96: what it was, for the RIGHT's own writes) */ \
	"  s_cselect_b32 s92, s92, 64\n" \
	"  s_cbranch_scc0 Lmixr1_%=\n" \
	"  s_mov_b32 s96, %[en]\n" \
	"  s_mov_b32 %[en], s94\n" \
	"Lmixr1_%=:\n" \
	"  v_add_u32 v49, %[ep], v39\n"                  /* x[nL_j], x[nL_j - 1], x[nL_j - 2] */ \
	"  v_add_u32 v50, -1, v49\n" \
	"  v_add_u32 v52, -2, v49\n" \
	"  v_and_b32 v49, %[mask], v49\n" \
	"  v_and_b32 v50, %[mask], v50\n" \
	"  v_and_b32 v52, %[mask], v52\n" \
	"  v_lshlrev_b32 v49, 4, v49\n" \
	"  v_lshlrev_b32 v50, 4, v50\n" \
	"  v_lshlrev_b32 v52, 4, v52\n" \
	"  ds_read_b32 v46, v49\n" \
	"  ds_read_b32 v35, v50\n" \
	"  ds_read_b32 v61, v52\n" \
	"  v_add_u32 v34, %[vc], v38\n"                  /* vc + nV_j: the new vertex of a VERTEX lane */ \
	"  v_add_u32 v32, -1, v34\n" \
	"  v_add_u32 v36, -2, v34\n" \
	"  v_mov_b32 v58, %[v1]\n" \
	"  v_mov_b32 v62, s93\n" \
	"  v_cmp_lt_u32 vcc, s92, v60\n"                 /* behind the RIGHT: v1 is the vertex it closed against */ \
	"  v_cndmask_b32 v58, v58, v62, vcc\n" \
	"  v_cmp_lt_u32 vcc, 0, v38\n" \
	"  v_cndmask_b32 v32, v58, v32, vcc\n"           /* b_j */ \
	"  v_cmp_lt_u32 vcc, 1, v38\n" \
	"  v_cndmask_b32 v36, v58, v36, vcc\n"           /* c_j if symbol j-1 was a VERTEX */ \
	"  v_mov_b32 v58, %[v0]\n" \
	"  s_waitcnt lgkmcnt(0)\n" \
	"  v_cmp_lt_u32 vcc, 0, v39\n" \
	"  v_cndmask_b32 v33, v58, v35, vcc\n"           /* a_j */ \
	"  v_cmp_lt_u32 vcc, 1, v39\n" \
	"  v_cndmask_b32 v37, v58, v61, vcc\n"           /* c_j if symbol j-1 was a LEFT */ \
	"  v_cmp_eq_u32 vcc, 0, v56\n" \
	"  v_cndmask_b32 v34, v46, v34, vcc\n"           /* opp_j */ \
	"  s_lshl_b64 vcc, vcc, 1\n"                     /* lane j: symbol j-1 was a VERTEX */ \
	"  v_cndmask_b32 v37, v37, v36, vcc\n" \
	"  s_cmp_eq_u32 s92, 64\n" \
	"  s_cbranch_scc1 Lmixd_%=\n" \
	"  v_cmp_eq_u32 vcc, s92, v60\n"                 /* the RIGHT's own opposite vertex ... */ \
	"  v_cndmask_b32 v34, v34, v62, vcc\n" \
	"  s_add_u32 %[t1], s92, 1\n"                    /* ... and right behind it v2 is the v1 it found */ \
	"  v_mov_b32 v63, %[v1]\n" \
	"  v_cmp_eq_u32 vcc, %[t1], v60\n" \
	"  v_cndmask_b32 v37, v37, v63, vcc\n" \
	"Lmixd_%=:\n" \
	"  v_mov_b32 v58, %[v2]\n" \
	"  v_cmp_eq_u32 vcc, 0, v60\n" \
	"  v_cndmask_b32 v37, v37, v58, vcc\n"           /* c_j */ \
	"  v_readlane_b32 %[t1], v38, %[t0]\n"           /* VERTEXes and LEFTs of the step: lane k's counts */ \
	"  v_readlane_b32 %[t2], v39, %[t0]\n" \
	"  v_readlane_b32 s90, v54, %[t0]\n"            /* the window registers: lane k's eight symbols, and what is left of the word behind them (TOPO_ASM_PAIR below) */ \
	"  v_readlane_b32 s91, v43, %[t0]\n" \
	"  v_readlane_b32 %[v0], v33, %[t0]\n" \
	"  v_readlane_b32 %[v1], v32, %[t0]\n" \
	"  v_readlane_b32 %[v2], v37, %[t0]\n" \
	"  s_max_u32 %[t3], %[t2], 1\n" \
	"  s_sub_u32 %[t3], %[t3], 1\n" \
	"  v_readlane_b32 %[c], v47, %[t3]\n"            /* e.prev: the prev link of the last slot a LEFT closed */ \
	"  s_and_b32 %[c], %[c], 0xffff\n" \
	"  s_cmp_eq_u32 %[t2], 0\n" \
	"  s_cselect_b32 %[ep], %[ep], %[c]\n" \
	"  s_lshl_b32 %[c], %[start], " FSHIFT "\n"      /* byte offset of the step's first index */ \
	"  s_bfm_b64 exec, %[t0], 0\n"                   /* lanes 0 .. k-1 */ \
	FACE \
	"  v_cmp_eq_u32 vcc, 0, v56\n" \
	"  s_and_b64 exec, exec, vcc\n"                  /* the VERTEX lanes */ \
	"  v_mov_b32 v50, v32\n" \
	"  v_mov_b32 v51, v33\n" \
	"  v_mov_b32 v52, v37\n" \
	"  v_mul_lo_u32 v62, v34, 12\n" \
	"  v_add_u32 v40, %[nq], v38\n"                  /* the edge it queues: slot nq + nV_j */ \
	"  v_add_u32 v53, 1, v40\n" \
	"  v_add_u32 v55, -1, v40\n" \
	"  v_and_b32 v41, %[mask], v40\n" \
	"  v_and_b32 v53, %[mask], v53\n" \
	"  v_and_b32 v55, %[mask], v55\n" \
	"  global_store_dwordx3 v62, v[50:52], %[predb]\n" \
	"  v_mov_b32 v42, %[en]\n" \
	"  v_cmp_eq_u32 vcc, 0, v38\n" \
	"  v_cndmask_b32 v55, v55, v42, vcc\n"           /* next: e.next for the first */ \
	"  s_sub_u32 %[t3], %[t1], 1\n" \
	"  v_mov_b32 v42, 0xffff\n" \
	"  v_cmp_eq_u32 vcc, %[t3], v38\n" \
	"  v_cndmask_b32 v53, v53, v42, vcc\n"           /* prev: lazy for the last */ \
	"  v_lshl_or_b32 v47, v55, 16, v53\n" \
	"  v_mov_b32 v44, v34\n" \
	"  v_mov_b32 v45, v32\n" \
	"  v_mov_b32 v46, v33\n" \
	"  v_lshlrev_b32 v41, 4, v41\n" \
	"  ds_write_b128 v41, v[44:47]\n" \
	"  s_bfm_b64 exec, %[t0], 0\n" \
	"  v_cmp_eq_u32 vcc, 1, v56\n" \
	"  s_and_b64 exec, exec, vcc\n"                  /* the LEFT lanes: slot ep + nL_j is deleted */ \
	"  v_mov_b32 v42, 0x8000\n" \
	"  ds_write_b16 v49, v42 offset:10\n" \
	"  s_mov_b64 exec, 1\n" \
	"  s_cmp_eq_u32 s92, 64\n"                       /* the RIGHT's own writes: what was e.next is deleted, and its slot - a survivor's - goes back to the free list */ \
	"  s_cbranch_scc1 Lmixr2_%=\n"                   /* unless the DELAY stack still owns it (Lrightp's, above) */ \
	"  s_lshl_b32 %[c], s96, 4\n" \
	"  v_mov_b32 v40, %[c]\n" \
	"  v_mov_b32 v41, 0x8000\n" \
	"  ds_write_b16 v40, v41 offset:10\n" \
	"  s_cmp_gt_u32 s96, %[mask]\n" \
	"  s_cbranch_scc0 Lmixr2_%=\n" \
	"  s_bitcmp1_b32 s95, 30\n" \
	"  s_cbranch_scc1 Lmixr2_%=\n" \
	"  s_lshr_b32 %[c], %[pk1], 16\n" \
	"  s_lshl_b32 %[c], %[c], 1\n" \
	"  s_and_b32 %[t3], %[lay], 0xffff\n" \
	"  s_lshl_b32 %[t3], %[t3], 4\n" \
	"  s_add_u32 %[c], %[c], %[t3]\n" \
	"  v_mov_b32 v40, %[c]\n" \
	"  v_mov_b32 v41, s96\n" \
	"  ds_write_b16 v40, v41\n" \
	"  s_add_u32 %[pk1], %[pk1], 0x10000\n" \
	"Lmixr2_%=:\n" \
	"  s_cmp_eq_u32 %[t1], 0\n" \
	"  s_cbranch_scc1 Lmixnov_%=\n" \
	"  s_and_b32 %[t3], %[nq], %[mask]\n"            /* e.next.prev = the first new slot */ \
	"  s_lshl_b32 %[c], %[en], 4\n" \
	"  v_mov_b32 v40, %[t3]\n" \
	"  v_mov_b32 v41, %[c]\n" \
	"  ds_write_b16 v41, v40 offset:12\n" \
	"  s_add_u32 %[nq], %[nq], %[t1]\n" \
	"  s_sub_u32 %[c], %[nq], 1\n" \
	"  s_and_b32 %[en], %[c], %[mask]\n"             /* e.next: the last edge queued */ \
	"  s_add_u32 %[vc], %[vc], %[t1]\n" \
	"  s_sub_u32 %[budget], %[budget], %[t1]\n" \
	"Lmixnov_%=:\n" \
	"  s_mov_b32 %[nc], -1\n"                        /* (every record is in LDS: a RIGHT reads e.next's) */ \
	"  s_mul_i32 %[c], %[t0], 3\n" \
	"  s_add_u32 %[start], %[start], %[c]\n" \
	"  s_add_u32 %[cler], %[cler], %[t0]\n" \
	"  s_branch Lstepped_%=\n" \
	"Lmix0_%=:\n"                                    /* nothing done: the symbol goes the one-at-a-time way */ \
	"  s_mov_b64 exec, 1\n" \
	"  s_and_b32 %[c], s90, 15\n" \
	"  s_cbranch_scc0 Lvgo_%=\n" \
	"  s_branch Llgo_%=\n"

// The chain-end step.  BOUNDARY and DELAY end a chain: the current edge survives (it moves to a pool slot, its neighbours' links follow),
// and the next live gate is popped from the ring - ~70 dependent scalar instructions and two LDS round trips, ~1 000 clocks, and half of a
// regular blob's automaton (222 chain ends) or two thirds of a holey disc's (1 935).  They come in runs (BB, DDDD, DB: 188 of the regular
// blob's 222, 1 637 of the disc's 1 935): the gate just popped is ended by the next symbol at once.  Nothing in such a run depends on the one
// before except through the links of neighbouring gates, so after the current edge has been dealt with the usual way and the NEXT symbol is a
// BOUNDARY / DELAY too, the whole wave looks at the next 64 queue entries (as the pop does anyway): the r-th live one is ended by the r-th
// symbol - lane = queue entry, r = its rank among the live ones (v_mbcnt) - as long as symbols are BOUNDARY / DELAY, a further live gate is in
// sight (the last one popped becomes the current edge), and the pool / DELAY stack have room.  A moving lane takes pool slot number r (free
// list from the top, then the bump pointer), leaves a forwarding mark (0xF0000000 | new slot; vertex ids have 30 bits) in its ring record's
// first word, and every lane follows the marks of its two neighbours: a neighbour that moves in the same step is addressed at its new slot
// (and writes its own link), any other gets its link rewritten as the one-at-a-time code does.  DELAYs set their flag and push their slot at
// (stack fill + DELAYs in front of them).  Checked against the oracle on the host model first (tools/topo_run_model.py).
#define TOPO_ASM_ENDS \
	"Lends_%=:\n"                                     /* t2 = queued entries (not 0) */ \
	"  s_min_u32 %[t2], %[t2], 64\n" \
	"  s_mov_b64 exec, -1\n" \
	"  v_mbcnt_lo_u32_b32 v60, -1, 0\n" \
	"  v_mbcnt_hi_u32_b32 v60, -1, v60\n"            /* v60 = lane */ \
	"  v_add_u32 v40, %[cler], v60\n"                /* lane j: symbol cler + j */ \
	"  v_lshrrev_b32 v41, 3, v40\n" \
	"  v_lshlrev_b32 v41, 2, v41\n" \
	"  v_add_u32 v41, %[clw], v41\n" \
	"  v_add_u32 v41, -4, v41\n" \
	"  ds_read2_b32 v[42:43], v41 offset1:1\n" \
	"  v_add_u32 v61, %[qpos], v60\n"                /* lane i: queue entry qpos + i */ \
	"  v_and_b32 v61, %[mask], v61\n" \
	"  v_lshlrev_b32 v61, 4, v61\n" \
	"  v_mov_b32 v58, -1\n"                          /* (entries past the queue's end: dead) */ \
	"  v_cmp_gt_u32 vcc, %[t2], v60\n" \
	"  s_mov_b64 exec, vcc\n" \
	"  ds_read_b128 v[56:59], v61\n" \
	"  s_mov_b64 exec, -1\n" \
	"  v_and_b32 v53, 7, v40\n" \
	"  v_lshlrev_b32 v53, 2, v53\n" \
	"  s_waitcnt lgkmcnt(0)\n" \
	"  v_lshrrev_b64 v[54:55], v53, v[42:43]\n"      /* v54: the symbols from the lane's on */ \
	"  v_and_b32 v44, 14, v54\n" \
	"  v_cmp_ne_u32 vcc, 4, v44\n"                   /* not BOUNDARY (4) / DELAY (5) */ \
	"  s_ff1_i32_b64 %[t1], vcc\n"                   /* nb: chain-end symbols in a row (-1: 64) */ \
	"  s_min_u32 %[t1], %[t1], 62\n" \
	"  v_cmp_gt_i32 vcc, v58, -1\n"                  /* live entries (TOPO_DEAD is the sign bit) */ \
	"  s_bcnt1_i32_b64 %[t0], vcc\n" \
	"  s_cmp_lt_u32 %[t0], 2\n"                      /* fewer than two live gates in sight: the one-gate way */ \
	"  s_cbranch_scc1 Lends0_%=\n" \
	"  v_mov_b32 v37, 1\n" \
	"  v_cndmask_b32 v39, 0, v37, vcc\n"             /* v39: live */ \
	"  v_mbcnt_lo_u32_b32 v38, vcc_lo, 0\n" \
	"  v_mbcnt_hi_u32_b32 v38, vcc_hi, v38\n"        /* v38: rank among the live entries */ \
	"  s_sub_u32 %[t0], %[t0], 1\n" \
	"  s_min_u32 %[t0], %[t0], %[t1]\n"              /* k = min(symbols, live gates - 1, pool room, DELAY room) */ \
	"  s_lshr_b32 %[t1], %[pk1], 16\n"               /* free-list fill */ \
	"  s_and_b32 %[t3], %[pk1], 0xffff\n"            /* bump pointer */ \
	"  s_and_b32 %[c], %[lay], 0xffff\n"             /* ring + pool slots */ \
	"  s_sub_u32 %[c], %[c], %[t3]\n" \
	"  s_add_u32 %[c], %[c], %[t1]\n" \
	"  s_min_u32 %[t0], %[t0], %[c]\n" \
	"  s_lshr_b32 %[c], %[pk2], 16\n" \
	"  s_and_b32 %[t2], %[pk2], 0xffff\n"            /* DELAY stack fill */ \
	"  s_sub_u32 %[c], %[c], %[t2]\n" \
	"  s_min_u32 %[t0], %[t0], %[c]\n" \
	"  s_cmp_eq_u32 %[t0], 0\n" \
	"  s_cbranch_scc1 Lends0_%=\n" \
	"  v_and_b32 v44, 15, v54\n"                     /* the symbol lanes: symbol | DELAYs in front of it << 8 */ \
	"  s_bfm_b64 exec, %[t0], 0\n" \
	"  v_cmp_eq_u32 vcc, 5, v44\n" \
	"  s_mov_b64 exec, -1\n" \
	"  s_nop 1\n" \
	"  v_mbcnt_lo_u32_b32 v45, vcc_lo, 0\n" \
	"  v_mbcnt_hi_u32_b32 v45, vcc_hi, v45\n" \
	"  v_lshl_or_b32 v45, v45, 8, v44\n" \
	"  v_lshlrev_b32 v46, 2, v38\n" \
	"  ds_bpermute_b32 v46, v46, v45\n"              /* v46: the symbol that ends the lane's gate (lane rank's) */ \
	"  s_waitcnt lgkmcnt(0)\n" \
	"  v_lshrrev_b32 v62, 8, v46\n"                  /* round 6: only a DELAY takes a pool slot (a BOUNDARY edge goes to the sink): its place among the step's DELAYs */ \
	"  v_cmp_gt_u32 vcc, %[t1], v62\n"               /* place < free-list fill: slot freel[fill - 1 - place] */ \
	"  v_sub_u32 v47, %[t1], v62\n" \
	"  v_add_u32 v47, -1, v47\n" \
	"  v_lshlrev_b32 v47, 1, v47\n" \
	"  s_and_b32 %[c], %[lay], 0xffff\n" \
	"  s_lshl_b32 %[c], %[c], 4\n" \
	"  v_add_u32 v47, %[c], v47\n" \
	"  s_mov_b64 exec, vcc\n" \
	"  ds_read_u16 v47, v47\n" \
	"  s_mov_b64 exec, -1\n" \
	"  v_add_u32 v48, %[t3], v62\n"                  /* else bump + place - fill */ \
	"  v_subrev_u32 v48, %[t1], v48\n" \
	"  s_waitcnt lgkmcnt(0)\n" \
	"  v_cndmask_b32 v47, v48, v47, vcc\n"           /* a DELAY's pool slot ... */ \
	"  v_and_b32 v48, 15, v46\n" \
	"  v_cmp_eq_u32 vcc, 5, v48\n" \
	"  s_add_u32 %[c], %[mask], 1\n" \
	"  v_mov_b32 v48, %[c]\n" \
	"  v_cndmask_b32 v47, v48, v47, vcc\n"           /* v47: ... or the sink */ \
	"  v_cmp_ne_u32 vcc, 0, v39\n" \
	"  s_mov_b64 exec, vcc\n"                        /* live lanes ... */ \
	"  v_cmp_gt_u32 vcc, %[t0], v38\n" \
	"  s_mov_b64 exec, vcc\n"                        /* ... that move (rank < k): the forwarding mark */ \
	"  v_or_b32 v50, 0xf0000000, v47\n" \
	"  ds_write_b32 v61, v50\n" \
	"  s_mov_b64 exec, -1\n" \
	"  s_and_b32 %[c], %[lay], 0xffff\n" \
	"  s_sub_u32 %[c], %[c], 1\n"                    /* (links are slot ids below ring + pool; a lazy 0xffff is clamped) */ \
	"  v_and_b32 v48, 0xffff, v59\n" \
	"  v_lshrrev_b32 v49, 16, v59\n" \
	"  v_min_u32 v50, %[c], v48\n" \
	"  v_min_u32 v51, %[c], v49\n" \
	"  v_lshlrev_b32 v50, 4, v50\n" \
	"  v_lshlrev_b32 v51, 4, v51\n" \
	"  ds_read_b32 v50, v50\n" \
	"  ds_read_b32 v51, v51\n" \
	"  s_waitcnt lgkmcnt(0)\n" \
	"  v_lshrrev_b32 v52, 28, v50\n" \
	"  v_cmp_eq_u32 vcc, 15, v52\n" \
	"  v_and_b32 v52, 0xffff, v50\n" \
	"  v_cndmask_b32 v52, v48, v52, vcc\n"           /* v52: prev, forwarded */ \
	"  v_cndmask_b32 v35, 0, v37, vcc\n"             /* v35: no write to prev's record: it moves too ... */ \
	"  v_cmp_lt_u32 vcc, %[c], v48\n" \
	"  v_cndmask_b32 v35, v35, v37, vcc\n"           /* ... or the link is no slot (a lazy 0xffff: kept as it is, like the one-at-a-time code's out-of-range write) */ \
	"  v_cndmask_b32 v52, v52, v48, vcc\n" \
	"  v_lshrrev_b32 v53, 28, v51\n" \
	"  v_cmp_eq_u32 vcc, 15, v53\n" \
	"  v_and_b32 v53, 0xffff, v51\n" \
	"  v_cndmask_b32 v53, v49, v53, vcc\n"           /* v53: next, forwarded */ \
	"  v_cndmask_b32 v36, 0, v37, vcc\n" \
	"  v_cmp_lt_u32 vcc, %[c], v49\n" \
	"  v_cndmask_b32 v36, v36, v37, vcc\n" \
	"  v_cndmask_b32 v53, v53, v49, vcc\n" \
	"  v_cmp_ne_u32 vcc, 0, v39\n" \
	"  s_mov_b64 exec, vcc\n" \
	"  v_cmp_eq_u32 vcc, %[t0], v38\n"               /* the live entry of rank k: the next current edge */ \
	"  s_ff1_i32_b64 %[t1], vcc\n" \
	"  v_cmp_gt_u32 vcc, %[t0], v38\n" \
	"  s_mov_b64 exec, vcc\n"                        /* the movers: their records in the pool */ \
	"  v_mov_b32 v34, 0\n"                           /* (and the mark wiped: a stale link that found it in a later step would be forwarded into a live record) */ \
	"  ds_write_b32 v61, v34\n" \
	"  v_and_b32 v34, 15, v46\n" \
	"  v_and_b32 v58, 0x3fffffff, v58\n" \
	"  v_or_b32 v33, 0x40000000, v58\n"              /* TOPO_DELAYED */ \
	"  v_cmp_eq_u32 vcc, 5, v34\n" \
	"  v_cndmask_b32 v58, v58, v33, vcc\n" \
	"  v_lshl_or_b32 v59, v53, 16, v52\n" \
	"  v_lshlrev_b32 v32, 4, v47\n" \
	"  v_lshrrev_b32 v33, 8, v46\n"                  /* a DELAY's place on the stack */ \
	"  v_add_u32 v33, %[t2], v33\n" \
	"  v_lshlrev_b32 v33, 1, v33\n" \
	"  s_lshr_b32 %[c], %[lay], 16\n" \
	"  s_lshl_b32 %[c], %[c], 4\n" \
	"  v_add_u32 v33, %[c], v33\n" \
	"  s_and_b64 exec, exec, vcc\n"                  /* the DELAYs: their records in the pool (a BOUNDARY edge leaves none), their slots on the stack */ \
	"  ds_write_b128 v32, v[56:59]\n" \
	"  ds_write_b16 v33, v47\n" \
	"  s_mov_b64 exec, -1\n" \
	"  v_cmp_ne_u32 vcc, 0, v39\n"                   /* the neighbours that stay where they are: their links */ \
	"  s_mov_b64 exec, vcc\n" \
	"  v_cmp_gt_u32 vcc, %[t0], v38\n" \
	"  s_mov_b64 exec, vcc\n" \
	"  v_cmp_eq_u32 vcc, 0, v35\n" \
	"  s_and_b64 exec, exec, vcc\n" \
	"  v_lshlrev_b32 v32, 4, v52\n" \
	"  ds_write_b16 v32, v47 offset:14\n"            /* front[prev].next = slot */ \
	"  s_mov_b64 exec, -1\n" \
	"  v_cmp_ne_u32 vcc, 0, v39\n" \
	"  s_mov_b64 exec, vcc\n" \
	"  v_cmp_gt_u32 vcc, %[t0], v38\n" \
	"  s_mov_b64 exec, vcc\n" \
	"  v_cmp_eq_u32 vcc, 0, v36\n" \
	"  s_and_b64 exec, exec, vcc\n" \
	"  v_lshlrev_b32 v32, 4, v53\n" \
	"  ds_write_b16 v32, v47 offset:12\n"            /* front[next].prev = slot */ \
	"  s_mov_b64 exec, 1\n" \
	"  v_readlane_b32 %[v0], v56, %[t1]\n"           /* the next current edge */ \
	"  v_readlane_b32 %[v1], v57, %[t1]\n" \
	"  v_readlane_b32 %[v2], v58, %[t1]\n" \
	"  v_readlane_b32 %[ep], v52, %[t1]\n" \
	"  v_readlane_b32 %[en], v53, %[t1]\n" \
	"  v_readlane_b32 %[c], v45, %[t0]\n"            /* DELAYs of the step (lane k's count) */ \
	"  v_readlane_b32 s90, v54, %[t0]\n"            /* the window registers: lane k's eight symbols, and what is left of the word behind them (TOPO_ASM_PAIR below) */ \
	"  v_readlane_b32 s91, v43, %[t0]\n" \
	"  s_and_b32 %[v2], %[v2], 0x3fffffff\n" \
	"  s_add_u32 %[qpos], %[qpos], %[t1]\n" \
	"  s_add_u32 %[qpos], %[qpos], 1\n" \
	"  s_lshr_b32 %[c], %[c], 8\n" \
	"  s_add_u32 %[pk2], %[pk2], %[c]\n" \
	"  s_lshr_b32 %[t1], %[pk1], 16\n" \
	"  s_min_u32 %[t1], %[t1], %[c]\n"               /* the DELAYs' slots: taken from the free list; the rest from the bump pointer */ \
	"  s_sub_u32 %[c], %[c], %[t1]\n" \
	"  s_add_u32 %[pk1], %[pk1], %[c]\n" \
	"  s_lshl_b32 %[t1], %[t1], 16\n" \
	"  s_sub_u32 %[pk1], %[pk1], %[t1]\n" \
	"  s_add_u32 %[cler], %[cler], %[t0]\n" \
	"  s_mov_b32 %[nc], -1\n" \
	"  s_and_b32 %[c], %[cler], 7\n"                 /* (s91 came back as lane k's whole next word: shifted like s90, as at Lstepped) */ \
	"  s_lshl_b32 %[c], %[c], 2\n" \
	"  s_lshr_b32 s91, s91, %[c]\n" \
	"  s_mov_b32 %[c], 0\n" \
	"  s_cmp_ge_u32 %[cler], %[slideat]\n" \
	"  s_cbranch_scc1 Lexit_%=\n" \
	"  s_branch Ltop_%=\n" \
	"Lends0_%=:\n" \
	"  s_mov_b64 exec, 1\n" \
	"  s_sub_u32 %[t2], %[nq], %[qpos]\n" \
	"  s_branch Lpop1_%=\n"

// -DCORTO_TOPO_STAMPS (CORTO_BUILD_DEFINES=CORTO_TOPO_STAMPS python -m corto_amd.build --force; tools/topo_stamp_probe.py): where the
// automaton's time goes - shader clocks (s_memtime), steps and symbols per phase of workgroups 0..4095, read back with
// crthip_debug_topo_stamps.  Phases: 0 ISA block (run and mix steps included), 3 C++ symbol, 4 gate fetch, 5 prologue; [15] = all of it.
#ifdef CORTO_TOPO_STAMPS
__device__ uint32_t g_topo_stamps[48*4096];
__device__ uint32_t g_topo_trace[16*8192];
#define TOPO_CLK() ((uint32_t)__builtin_amdgcn_s_memtime())
#define TOPO_T0() const uint32_t tt0_ = TOPO_CLK(), tc0_ = cler
#define TOPO_ACC(i) do { st_clk[i] += TOPO_CLK() - tt0_; st_cnt[i]++; st_sym[i] += cler - tc0_; } while(0)
#else
#define TOPO_T0() do { } while(0)
#define TOPO_ACC(i) do { } while(0)
#endif
template <bool U16, bool PROGRESS_>
__device__ __forceinline__ bool topo_lds_body(const TopoJob &J) {
#ifdef TOPO_NO_PARTIAL_PROGRESS
	constexpr bool PROGRESS = false;
#else
	constexpr bool PROGRESS = PROGRESS_;
#endif       // false: out of slots, nothing valid written
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	const uint32_t RING = J.lds_ring, MASK = RING - 1, POOL = J.lds_pool, dcap = J.lds_delayed_cap, SYMW = J.lds_symwin;
	CRT_LDS u32x4 *rec = (CRT_LDS u32x4 *)as_lds(lds);
	CRT_LDS uint16_t *rec16 = (CRT_LDS uint16_t *)rec;
	CRT_LDS uint16_t *freel = (CRT_LDS uint16_t *)(rec + RING + POOL);
	CRT_LDS uint16_t *delayed = freel + ((POOL + 7) & ~7u);
	CRT_LDS uint32_t *cold = (CRT_LDS uint32_t *)(delayed + ((dcap + 7) & ~7u));   // state only the cold paths touch lives here, not in loop-carried registers
	CRT_LDS uint32_t *spl = cold + 8;                                   // the first TOPO_SPLIT_LDS words of the split / vertex-id bits: a SPLIT that loads them from HBM
	CRT_LDS uint32_t *cl32 = spl + TOPO_SPLIT_LDS;                      // the symbol window LAST: the ISA block finds cold[] and spl[] at fixed distances below it
	const uint32_t nspl = J.split_nwords < TOPO_SPLIT_LDS ? J.split_nwords : TOPO_SPLIT_LDS;   // waits ~2 us (the load, and every store in flight before it)
	CRT_GLOBAL const uint8_t *gcl = as_global(J.clers);
	const uint32_t nclers = J.nclers, symwords = SYMW/8;
	// the ISA block addresses records from LDS address 0 and finds the free list and the DELAY stack through the layout word (round 5: the pool
	// is sized on its own - rounds 2-4 had pool = ring, and a mesh whose boundary asked for 700 pool slots paid for a ring of 1 024 it used 120 of):
	// FL16 | DL16 << 16 = where they start, in 16-byte units (FL16 = ring + pool is also the number of slots); cold[] sits one (equally long)
	// DELAY stack behind DL16: anything else takes the HBM path
	if((uint32_t)(uintptr_t)rec != 0u || dcap != POOL || (POOL & 7u) || RING + POOL > 0x7FFFu || (RING & MASK)) return false;
	const uint32_t lay = (RING + POOL) | (RING + POOL + (POOL >> 3)) << 16;
	// automaton state, alive across window refills (uniform: only lane 0 ever changes it)
	CRT_GLOBAL const uint32_t *split = as_global(J.split_words);
	CRT_GLOBAL uint8_t *predb = (CRT_GLOBAL uint8_t *)as_global(J.pred);   // prediction triple of vertex vc at byte 12*vc (vertices are numbered in creation order)
	CRT_GLOBAL uint8_t *faceb = as_global((uint8_t *)J.faces);          // index `start` at byte 4*start (2*start for u16 indices)
	CRT_GLOBAL const uint32_t *group_end = as_global(J.group_end);
	const uint32_t nvert = J.nvert;
	const uint32_t splitbits = 32 - __clz(nvert | 1u);
	uint32_t cler = 0, winbase = 0, vc = 0, err = 0;                     // err: 1 = bad stream, 2 = out of slots
	uint32_t pub = 0;                                                    // vertices published in J.progress
	uint32_t start = 0;
	uint32_t nq = 0, qpos = 0;                                           // ring [qpos, nq)
	const uint64_t bit_end = (uint64_t)J.split_nwords*32;
	enum { K_BIT_LO = 3, K_BIT_HI = 4, K_SPLITBITS = 5, K_BIT_LIMIT = 6 };   // split-bit cursor; bits of a vertex id; bits the staged words hold (for the ISA block's SPLIT)

	// the whole wave fills the symbol window once; from then on lane 0 is alone (and the compiler sees uniform code), and
	// slides the window by itself between chains when a mesh has more symbols than the window (3 instructions per symbol)
	// (two words behind the window: "window exhausted" nibbles - a chain that outruns the window ends in the HBM redo, so that the
	// loop never loads symbols from HBM itself: a load's s_waitcnt would also wait for every face / prediction store in flight)
#ifdef CORTO_TOPO_STAMPS
	uint32_t st_clk[6] = {0, 0, 0, 0, 0, 0}, st_cnt[6] = {0, 0, 0, 0, 0, 0}, st_sym[6] = {0, 0, 0, 0, 0, 0};
	const uint32_t st_begin = TOPO_CLK();
#endif
#ifdef CORTO_TOPO_STAMPS
	for(uint32_t i = threadIdx.x; i < 8192; i += 64) ((CRT_LDS uint32_t *)as_lds(lds))[8192 + i] = 0;
#endif
	TOPO_FILL_WINDOW(0u);
	if(nspl) {                                                            // (<= 256 words: four loads per lane, in flight together)
		uint32_t sw4[4];
#pragma unroll
		for(uint32_t u = 0; u < 4; u++) { const uint32_t w = threadIdx.x + 64*u; sw4[u] = split[w < nspl ? w : nspl - 1u]; }
		asm volatile("" : "+v"(sw4[0]), "+v"(sw4[1]), "+v"(sw4[2]), "+v"(sw4[3]));
#pragma unroll
		for(uint32_t u = 0; u < 4; u++) { const uint32_t w = threadIdx.x + 64*u; if(w < nspl) spl[w] = sw4[u]; }
	}
	__syncthreads();
	if(threadIdx.x != 0) return true;
	cold[K_BIT_LO] = 0; cold[K_BIT_HI] = 0; cold[K_SPLITBITS] = splitbits;
	{ const uint64_t lim = (uint64_t)nspl*32u; cold[K_BIT_LIMIT] = (uint32_t)(lim < bit_end ? lim : bit_end); }
	// pool bump pointer | free-list fill << 16, DELAY stack fill | its capacity << 16: touched at every chain end, kept in registers and
	// packed in pairs - the ISA block carries them, and an asm statement takes 30 operands at most
	uint32_t pk1 = RING + 1u, pk2 = dcap << 16;                          // (pool slot RING is the sink of the BOUNDARY edges: below)
#define TOPO_NFREE() (pk1 >> 16)
#define TOPO_MBUMP() (pk1 & 0xFFFFu)
#define TOPO_NDEL() (pk2 & 0xFFFFu)
#ifdef CORTO_TOPO_STAMPS
	st_clk[5] = TOPO_CLK() - st_begin;
#endif
	__builtin_amdgcn_s_setprio(3);                                      // the serial chain of the whole batch: ahead of any co-resident kernel's waves
	uint32_t sw = TOPO_S(cl32[0]), swn = TOPO_S(cl32[1]);   // TOPO_S: a value lane 0 alone computes is uniform by construction; tell the compiler (SGPR)
	uint32_t wbias = 1, slide_at = SYMW < nclers ? SYMW - 2048u : 0xFFFFFFFFu;   // next symbol word = cl32[(cler >> 3) + wbias]; slide when cler gets here
	uint32_t clw = (uint32_t)(uintptr_t)cl32 + 4u*wbias;                        // ... at LDS byte address clw + 4*(cler >> 3) (the ISA block's one operand for both)
	{
		{
#define TOPO_BITS(dst, n) do { uint64_t bit_ = (uint64_t)cold[K_BIT_LO] | (uint64_t)cold[K_BIT_HI] << 32; if(bit_ + (n) > bit_end) { err = 1; dst = 0; } \
	else { dst = (bit_ + (n) + 31)/32 <= nspl ? bit_field(spl, nspl, bit_, (n)) : bit_field(split, J.split_nwords, bit_, (n)); bit_ += (n); cold[K_BIT_LO] = (uint32_t)bit_; cold[K_BIT_HI] = (uint32_t)(bit_ >> 32); } } while(0)
#define TOPO_FACE(a, b, c) do { if(U16) { CRT_GLOBAL uint16_t *h_ = (CRT_GLOBAL uint16_t *)(faceb + start*2u); h_[0] = (uint16_t)(a); h_[1] = (uint16_t)(b); h_[2] = (uint16_t)(c); } \
	else { u32x3 f_; f_.x = (a); f_.y = (b); f_.z = (c); *(CRT_GLOBAL u32x3 *)(faceb + start*4u) = f_; } start += 3; } while(0)
#define TOPO_PRED(a, b, c) do { u32x3 p_; p_.x = (a); p_.y = (b); p_.z = (c); *(CRT_GLOBAL u32x3 *)(predb + vc*12u) = p_; } while(0)   // always right before vc++
#define TOPO_PUT(e, a, b, c, p, n) do { u32x4 t_; t_.x = (a); t_.y = (b); t_.z = (c); t_.w = (p) | ((n) << 16); rec[e] = t_; } while(0)
	// slide the symbol window up to the current symbol (whole wave, TOPO_FILL_WINDOW) and reload the two window registers
#define TOPO_SLIDE() do { if(PROGRESS) __hip_atomic_store((CRT_GLOBAL uint32_t *)(predb - TOPO_PROGRESS_BYTES), vc, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   /* (the window's refill waits for every store in flight anyway) */ \
	winbase = cler & ~31u; TOPO_FILL_WINDOW(winbase); \
	{ const uint32_t wi_ = (cler - winbase) >> 3; const uint64_t w2_ = ((uint64_t)TOPO_S(cl32[wi_]) | (uint64_t)TOPO_S(cl32[wi_ + 1]) << 32) >> (4*(cler & 7u)); sw = (uint32_t)w2_; swn = (uint32_t)(w2_ >> 32); } \
	wbias = 1u - (winbase >> 3); clw = (uint32_t)(uintptr_t)cl32 + 4u*wbias; slide_at = winbase + SYMW < nclers ? winbase + SYMW - 2048u : 0xFFFFFFFFu; } while(0)
// (sw, swn) = the 64 bits of the current symbol word and the next one, shifted down to the current symbol: sw always holds EIGHT symbols
#define TOPO_SYMBOL(c) do { c = sw & 0xFu; sw = sw >> 4 | swn << 28; swn >>= 4; cler++; if((cler & 7u) == 0) swn = TOPO_S(cl32[(cler >> 3) + wbias]); } while(0)
	// a deleted survivor goes back to the pool, unless it still sits in the DELAY stack (then the pop returns it)
#define TOPO_RELEASE(id, z) do { if((id) > MASK && !(TOPO_S(z) & TOPO_DELAYED)) { freel[TOPO_NFREE()] = (uint16_t)(id); pk1 += 0x10000u; } } while(0)
	// give the surviving current edge a pool slot, its record, and its neighbours their links to it
#define TOPO_MATERIALISE(flags) do { \
	if(TOPO_NFREE()) { pk1 -= 0x10000u; f = TOPO_S(freel[TOPO_NFREE()]); } else if(TOPO_MBUMP() < RING + POOL) { f = TOPO_MBUMP(); pk1++; } else { err = 2; break; } \
	TOPO_PUT(f, v0, v1, v2 | (flags), ep, en); rec16[ep*8 + 7] = (uint16_t)f; rec16[en*8 + 6] = (uint16_t)f; } while(0)

			for(uint32_t g = 0; g < J.ngroups && !err; g++) {              // every group starts from an empty front (decoder.cpp:173-178)
			const uint32_t ge = TOPO_S(group_end[g]);
			if(ge > J.nface || ge*3 < start) { err = 1; break; }
			const uint32_t end = ge*3;
			nq = 0; qpos = 0; pk1 = RING + 1u; pk2 = dcap << 16;
			// THE SINK (round 6).  A BOUNDARY edge's record is never read again: decoder.cpp:311-339 read front[e.prev] / front[e.next] only to close against them, and
			// nothing closes against an edge that has no face behind it (tools/topo_run_model.py asserts it on every family, non-manifold glue included).  Only its slot id
			// lives on, in its neighbours' links - which they only ever overwrite.  So every BOUNDARY edge is ONE pool slot, this one: no slot taken, no record written,
			// two link writes - and the pool holds the DELAYed edges alone (a C4 blob: 65 slots instead of 222, a Delaunay disc: 194 of 469).  Its record exists for
			// malformed streams only: links in range, flagged DELAYED so that a LEFT / RIGHT against it never puts it on the free list.
			{ u32x4 t_; t_.x = 0; t_.y = 0; t_.z = TOPO_DELAYED; t_.w = RING | RING << 16; rec[RING] = t_; }
			while(start < end && !err) {
				if(cler >= slide_at) TOPO_SLIDE();                          // slide the window before it runs low
				// a consumer running beside the automaton (k_delta_tiles): tell it every 4 096 vertices how far the triples have got (a release: ~1 us for
				// the stores in flight to arrive - 2 % of a big mesh's automaton; a mesh whose symbols fit the window never slides)
				if(PROGRESS && vc - pub >= 4096u) { __hip_atomic_store((CRT_GLOBAL uint32_t *)(predb - TOPO_PROGRESS_BYTES), vc, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); pub = vc; }
				TOPO_T0();
				// ---- cold: fetch the next edge to process: ring, DELAY stack, or a new seed face ----
				uint32_t f;
				u32x4 t0;
				uint32_t nd_;
				if(qpos != nq) {
					// Most queued edges are dead by the time they are popped (closed by a neighbouring chain: nine in ten on a
					// sphere-like mesh).  The other 63 lanes of the wave are parked, so for one instruction sequence they are
					// switched back on and each looks at the flags of ONE of the next 64 queue entries: the first live one is
					// found in a single LDS round trip however many dead ones sit in front of it.  A popped slot is free from here on.
					const uint32_t avail = nq - qpos;
					uint64_t alive, save;
					asm volatile(
						"s_mov_b64 %[sv], exec\n"
						"s_mov_b64 exec, -1\n"
						"v_mbcnt_lo_u32_b32 v60, -1, 0\n"
						"v_mbcnt_hi_u32_b32 v60, -1, v60\n"
						"v_add_u32 v61, %[qpos], v60\n"
						"v_and_b32 v61, %[mask], v61\n"
						"v_lshlrev_b32 v61, 4, v61\n"
						"v_add_u32 v61, %[base], v61\n"
						"ds_read_b32 v61, v61 offset:8\n"          /* v2 | flags of entry qpos + lane */
						"s_waitcnt lgkmcnt(0)\n"
						"v_cmp_gt_i32 %[alive], v61, -1\n"        /* TOPO_DEAD is the sign bit */
						"s_mov_b64 exec, %[sv]\n"
						: [alive] "=s"(alive), [sv] "=&s"(save)
						: [qpos] "s"(qpos), [mask] "s"(MASK), [base] "s"((uint32_t)(uintptr_t)rec)
						: "memory", "v60", "v61");
					if(avail < 64) alive &= (1ull << avail) - 1ull;
					if(!alive) { qpos += avail < 64 ? avail : 64u; continue; }         // all dead: no symbol consumed (decoder.cpp:278-279)
					const uint32_t j = (uint32_t)__builtin_ctzll(alive);
					t0 = rec[(qpos + j) & MASK];
					qpos += j + 1;
					f = 0;
				}
				else if((nd_ = TOPO_NDEL()) != 0) { f = delayed[nd_ - 1]; pk2--; t0 = rec[f]; freel[TOPO_NFREE()] = (uint16_t)f; pk1 += 0x10000u; }
				else {                                                     // seed face (decoder.cpp:224-259)
					uint32_t c; TOPO_SYMBOL(c);
					uint32_t last = vc - 1, vi[3], mask = 0;
					if(c == C_SPLIT) TOPO_BITS(mask, 3);
					else if(c != C_VERTEX) { err = c == 14u ? 2u : 1u; break; }
					for(int k = 0; k < 3; k++) {
						uint32_t v;
						if(mask & (1u << k)) TOPO_BITS(v, splitbits);
						else {
							if(vc >= nvert) { err = 1; break; }
							TOPO_PRED(last, last, last);
							last = v = vc++;
						}
						vi[k] = v & TOPO_VMASK;
					}
					if(err) break;
					if(nq - qpos + 3 > RING) { err = 2; break; }
					TOPO_FACE(vi[0], vi[1], vi[2]);
					const uint32_t e0 = nq & MASK, e1 = (nq + 1) & MASK, e2 = (nq + 2) & MASK;
					TOPO_PUT(e0, vi[1], vi[2], vi[0], e2, e1);
					TOPO_PUT(e1, vi[2], vi[0], vi[1], e0, e2);
					TOPO_PUT(e2, vi[0], vi[1], vi[2], e1, e0);
					nq += 3;                                                   // all three wait in the queue
					continue;
				}
				if(t0.z & TOPO_DEAD) continue;                             // deleted: no symbol consumed (decoder.cpp:278-279)
				uint32_t v0 = TOPO_S(t0.x), v1 = TOPO_S(t0.y), v2 = TOPO_S(t0.z) & TOPO_VMASK, ep = TOPO_S(t0.w) & 0xFFFFu, en = TOPO_S(t0.w) >> 16;
				uint32_t nc = 0xFFFFFFFFu, nc_next = 0, nc_v1 = 0;         // cached (next, v1) of edge nc
				TOPO_ACC(4);

				// ---- hot: follow the chain of freshly created edges while the symbols are VERTEX / LEFT / RIGHT ----
				for(;;) {
					if(cler >= slide_at) TOPO_SLIDE();                      // (a chain longer than the window's margin: mid-chain)
					{
						// Every symbol a well-formed mesh contains in hand-scheduled gfx950 ISA: VERTEX, LEFT / RIGHT against a queued (ring) or a
						// surviving (pool) neighbour, SPLIT, BOUNDARY / DELAY with the pop behind them, ~40 instructions per symbol where the
						// compiler's dispatch of the C++ below spends ~68 (scalar copies at every join), and the wave-wide steps as sections of the
						// block: runs of (VERTEX LEFT) pairs with a lone VERTEX in front (TOPO_ASM_RUN, up to 127 symbols), any VERTEX / LEFT sequence
						// with one RIGHT (TOPO_ASM_MIX, 63), runs of chain ends (TOPO_ASM_ENDS).  The block PEEKS at the next symbol and leaves with
						// the state untouched for the rest - END, an invalid or window-end nibble, vertex ids, ring or pool slots running out, the
						// group's last face, the window wanting a slide - which the C++ below then handles.  (Round 4's dispatch trace found SPLIT and
						// the pool neighbours leaving for the C++ at 1 500-2 000 clocks each: DESIGN.md 3.1.)
						uint32_t t0_, t1_, t2_, t3_, c_;
						uint32_t budget_ = TOPO_S(min(nvert - min(vc, nvert), MASK + 1u - (nq - qpos)));   // VERTEX steps the block may take: vertex ids and ring slots left
						{ TOPO_T0();
						if constexpr(U16) { TOPO_FAST_PATH(TOPO_ASM_FACE16, TOPO_RUN_FACE16, TOPO_LEAD_FACE16, TOPO_MIX_FACE16, "1"); } else { TOPO_FAST_PATH(TOPO_ASM_FACE32, TOPO_RUN_FACE32, TOPO_LEAD_FACE32, TOPO_MIX_FACE32, "2"); }
						TOPO_ACC(0); }
						if(start >= end) break;
						if(c_ == 0x200u) break;                                   // the block ended the chain (BOUNDARY / DELAY) and found no gate to go on with
						// (anything else: the next symbol is one the block leaves to the C++ below - or it wants the window slid first, which
						// the top of this loop does after that symbol)
					}
					TOPO_T0();
					uint32_t c; TOPO_SYMBOL(c);
					if(c == C_VERTEX) {                                    // decoder.cpp:294-309
						if(vc >= nvert) { err = 1; break; }
						if(nq - qpos > MASK) { err = 2; break; }
						const uint32_t s = nq & MASK;                      // slot of the second new edge = its place in the queue
						nq++;
						TOPO_PRED(v1, v0, v2);
						const uint32_t opp = vc++;
						TOPO_FACE(v1, v0, opp);
						rec16[en*8 + 6] = (uint16_t)s;                     // front[e.next].prev = new_edge + 1
						TOPO_PUT(s, opp, v1, v0, 0xFFFFu, en);             // second new edge: queued, so it must exist; its prev is the lazy edge
						nc = s; nc_next = en; nc_v1 = v1;
						v2 = v1; v1 = opp; en = s;                         // first new edge (v0, opp, old v1, ep, s): next, lazily
					} else if(c == C_LEFT) {                               // decoder.cpp:311-317
						const u32x4 t = rec[ep];
						const uint32_t pp = TOPO_S(t.w) & 0xFFFFu, opp = TOPO_S(t.x);
						rec16[ep*8 + 5] = 0x8000u;                         // front[e.prev].deleted = true
						TOPO_RELEASE(ep, t.z);
						TOPO_FACE(v1, v0, opp);
						v2 = v0; v0 = opp; ep = pp;                        // new edge (opp, v1, old v0, pp, en): next, lazily
					} else if(c == C_RIGHT) {                              // decoder.cpp:319-325
						uint32_t nn, opp;
						if(en == nc) { nn = nc_next; opp = nc_v1; }        // (a ring slot: nothing to release)
						else { const u32x4 t = rec[en]; nn = TOPO_S(t.w) >> 16; opp = TOPO_S(t.y); TOPO_RELEASE(en, t.z); }
						rec16[en*8 + 5] = 0x8000u;
						TOPO_FACE(v1, v0, opp);
						nc = 0xFFFFFFFFu;
						v2 = v1; v1 = opp; en = nn;                        // new edge (v0, opp, old v1, ep, nn): next, lazily
					} else {                                               // ---- cold symbols end the chain ----
						if(c == C_BOUNDARY) {                              // (the sink: above)
							rec16[ep*8 + 7] = (uint16_t)RING; rec16[en*8 + 6] = (uint16_t)RING;
						} else if(c == C_SPLIT) {
							if(nq - qpos > MASK) { err = 2; break; }
							uint32_t opp; TOPO_BITS(opp, splitbits);
							if(err) break;
							opp = TOPO_S(opp) & TOPO_VMASK;
							const uint32_t s = nq & MASK;
							nq++;
							TOPO_FACE(v1, v0, opp);
							rec16[en*8 + 6] = (uint16_t)s;
							TOPO_PUT(s, opp, v1, v0, 0xFFFFu, en);
							nc = s; nc_next = en; nc_v1 = v1;
							v2 = v1; v1 = opp; en = s;
							if(start < end) continue;                      // SPLIT continues the chain like VERTEX
						} else if(c == C_DELAY) {                          // decoder.cpp:327-331
							const uint32_t nd2_ = TOPO_NDEL();
							if(nd2_ >= dcap) { err = 2; break; }
							TOPO_MATERIALISE(TOPO_DELAYED);
							if(!err) { delayed[nd2_] = (uint16_t)f; pk2++; }
						} else if(c == C_END) {                            // decoder.cpp:333-339
							const u32x4 tp = rec[ep], tn = rec[en];
							const uint32_t pp = tp.w & 0xFFFFu, nn = tn.w >> 16, opp = tp.x;
							rec16[ep*8 + 5] = 0x8000u; rec16[en*8 + 5] = 0x8000u;
							TOPO_RELEASE(ep, tp.z); TOPO_RELEASE(en, tn.z);
							rec16[pp*8 + 7] = (uint16_t)nn;
							rec16[nn*8 + 6] = (uint16_t)pp;
							TOPO_FACE(v1, v0, opp);
						} else err = c == 14u ? 2u : 1u;                   // window exhausted mid-chain (redo on the HBM front) / invalid symbol or past the end
						TOPO_ACC(3);
						break;
					}
					TOPO_ACC(3);
					if(start >= end) break;
				}
			}
			}
#undef TOPO_BITS
#undef TOPO_FACE
#undef TOPO_PRED
#undef TOPO_PUT
#undef TOPO_SYMBOL
#undef TOPO_SLIDE
#undef TOPO_RELEASE
#undef TOPO_MATERIALISE
#undef TOPO_NFREE
#undef TOPO_MBUMP
#undef TOPO_NDEL
		}
	}
#ifdef CORTO_TOPO_STAMPS
	if(blockIdx.x < 4096) {
		uint32_t *o_ = g_topo_stamps + blockIdx.x*48;
		for(int i = 0; i < 6; i++) { o_[i] = st_clk[i]; o_[8 + i] = st_cnt[i]; o_[16 + i] = st_sym[i]; }
		if(blockIdx.x < 16) { const uint32_t tn_ = cler + 1 < 8192 ? cler + 1 : 8192; for(uint32_t i = 0; i < 8192; i++) g_topo_trace[blockIdx.x*8192 + i] = i < tn_ ? ((CRT_LDS uint32_t *)as_lds(lds))[8192 + i] : 0u; g_topo_trace[blockIdx.x*8192 + 8191] = TOPO_CLK(); }
		o_[15] = TOPO_CLK() - st_begin; o_[24] = cler; o_[25] = err; o_[26] = nq; o_[27] = qpos; o_[28] = vc; o_[29] = start; o_[30] = RING; o_[31] = pk1; o_[32] = pk2; o_[33] = dcap;
	}
#endif
	if(err == 2) return false;
	for(uint32_t v = vc; v < nvert; v++) { CRT_GLOBAL uint32_t *p_ = (CRT_GLOBAL uint32_t *)(predb + (size_t)v*12u); p_[0] = 0; p_[1] = 0; p_[2] = 0; }   // never reached: (0, 0, 0) (see topo_run)
	if(err || cler > J.nclers) *as_global(J.status) = ERR_TOPOLOGY;
	return true;
}

// -DCORTO_TOPO_TIMES (tools/topo_times_probe.py): when and where every workgroup of the launch ran - shader clock at entry and exit, HW_ID, XCC_ID -
// read back with crthip_debug_topo_times: how many automata the chip really runs at once.  Nothing of it in the product build.
#ifdef CORTO_TOPO_TIMES
__device__ uint32_t g_topo_times[8*8192];
#endif
template <bool PROGRESS>
__device__ __forceinline__ void topo_lds_kernel(const TopoJob *__restrict__ jobs, const uint32_t *__restrict__ job_ids, uint32_t njobs) {
	if(blockIdx.x >= njobs) return;
#ifdef CORTO_TOPO_TIMES
	const uint64_t tt_begin = __builtin_amdgcn_s_memtime();
#endif
	const TopoJob J = jobs[job_ids[blockIdx.x]];
	const bool done = J.faces_u16 ? topo_lds_body<true, PROGRESS>(J) : topo_lds_body<false, PROGRESS>(J);
#ifdef CORTO_TOPO_TIMES
	if(threadIdx.x == 0 && blockIdx.x < 8192) {
		const uint64_t tt_end = __builtin_amdgcn_s_memtime();
		uint32_t hw, xcc;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
		uint32_t *o = g_topo_times + blockIdx.x*8;
		o[0] = (uint32_t)tt_begin; o[1] = (uint32_t)(tt_begin >> 32); o[2] = (uint32_t)tt_end; o[3] = (uint32_t)(tt_end >> 32); o[4] = hw; o[5] = xcc; o[6] = done ? 1u : 0u; o[7] = njobs;
	}
#endif
	if(!done) {                                                          // thread 0 only: redo the blob with the front in HBM
		GlobalFront F{(CRT_GLOBAL u32x4 *)as_global(J.front_a), (CRT_GLOBAL u32x2 *)as_global(J.front_b), as_global(J.order), as_global(J.delayed)};
		const uint32_t need = topo_run(J, as_global(J.clers), F);
		*as_global(J.flags) = (int32_t)(1u | need << 1);                    // bit 0: redone; above it: the ring slots (15 bits) and the pool slots (15 bits) it would have needed in LDS
	}
	// every triple is written (the redo rewrites what the LDS attempt had published with the same values: the automaton is deterministic)
	if(PROGRESS && threadIdx.x == 0) __hip_atomic_store((CRT_GLOBAL uint32_t *)((CRT_GLOBAL uint8_t *)as_global(J.pred) - TOPO_PROGRESS_BYTES), 0xFFFFFFFFu, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// Two kernels of the same body.  k_topology_lds: a batch's automata, exactly rounds 1-5's code.  k_topology_lds_big: the launch of the blobs that ask for much
// more LDS than the rest - big meshes - which ALSO keeps the progress word in front of its triples up to date (device_plan.h: TOPO_PAD_PROGRESS; every mesh
// has the word, so the kernel needs no test); a kernel of its own because the test, or even a TopoJob eight bytes longer, cost the batch's automata 2-3 %.
__global__ __launch_bounds__(64) void k_topology_lds(const TopoJob *__restrict__ jobs, const uint32_t *__restrict__ job_ids, uint32_t njobs) { topo_lds_kernel<false>(jobs, job_ids, njobs); }
__global__ __launch_bounds__(64) void k_topology_lds_big(const TopoJob *__restrict__ jobs, const uint32_t *__restrict__ job_ids, uint32_t njobs) { topo_lds_kernel<true>(jobs, job_ids, njobs); }

// ------------------------------------------------------------------------------------------------
// K-DELTA (mesh): v[i] += v[a] + v[b] - v[c] (or += v[a]) for i = 1..nvert-1 in index order
// (vertex_attribute.h:165-176).  a, b, c < i, so the recurrence is a DAG; its depth is only ~3.5*sqrt(n)
// (158 levels for the 2 112-vertex C4 unit), and it has a shape: in the breadth-first CLERS order nearly
// every vertex predicts from the vertex made just before it (a = i-1) and two vertices one ring of the front back.
// So the graph is a set of "stretches" - runs of consecutive vertices, each run a serial chain - skewed against each
// other by two steps, and the parallelism is ACROSS stretches.  Both kernels below give each lane / thread whole
// stretches to walk (k_delta_mesh, values in HBM); blobs whose values fit LDS take the window loop of k_delta.hip.

// Attributes too big for LDS (meshes of tens of thousands of vertices, attributes of more than four components): a stretch walk over
// HBM/L2 by one workgroup.  Thread k walks stretches k, k+T, ... in order and carries the value of the vertex it has just
// finished in registers (a = i-1 inside a stretch), so the only waiting is for the two parents one ring back - which the
// neighbouring stretch, two steps ahead, has normally published already.  Flags and values cross waves through L2 with
// release/acquire at workgroup scope.  One loop, test-and-fire in the same iteration: a lane that spun in an inner wait
// loop would keep the lanes it is waiting for (same wave) parked at the reconvergence point.
template <typename T, int NC>
__device__ void delta_stretch_global(CRT_GLOBAL T *v, CRT_GLOBAL uint8_t *fired, CRT_GLOBAL const uint32_t *starts, uint32_t ns,
                                     CRT_GLOBAL const uint32_t *pred, uint32_t nvert, uint32_t Nrt, bool para, uint32_t THREADS) {
	const uint32_t n = NC ? (uint32_t)NC : Nrt;
	uint32_t k = threadIdx.x;
	bool active = k < ns;
	uint32_t i = 0, end = 0;
	if(active) { i = starts[k]; end = k + 1 < ns ? starts[k + 1] : nvert; }
	uint32_t a = 0, b = 0, c = 0, na = 0, nb = 0, nc = 0;
	typedef uint32_t u32x3_t __attribute__((ext_vector_type(3)));
	auto issue = [&](uint32_t j) -> u32x3_t {                              // the triple of vertex j as one 12-byte load (clamped: unconditional)
		return *(CRT_GLOBAL const u32x3_t *)(pred + (size_t)(j < nvert ? j : nvert - 1u)*3);
	};
	auto take = [&](uint32_t j, u32x3_t t, uint32_t &x, uint32_t &y, uint32_t &z) {
		asm volatile("" : "+v"(t));
		x = j < nvert ? t.x : 0u; y = j < nvert ? (para ? t.y : t.x) : 0u; z = j < nvert ? (para ? t.z : t.x) : 0u;
	};
	if(active) { const u32x3_t t0 = issue(i), t1 = issue(i + 1); take(i, t0, a, b, c); take(i + 1, t1, na, nb, nc); }   // the next triple is always one vertex ahead of its use
	T prev[NC ? NC : 1];
	bool at_start = true, have2 = false;
	u32x3_t t2 = {0, 0, 0};                                                // ... and the one behind it is in flight while this vertex waits for its parents and fires
	while(active) {
		if(!have2) { t2 = issue(i + 2); have2 = true; }
		const bool inv = !(a < i && b < i && c < i);                        // malformed triple (and vertex 0): the value stays
		const uint32_t da = inv || !at_start ? 0u : a, db = inv ? 0u : b, dc = inv ? 0u : c;
		const uint32_t ready = __hip_atomic_load(&fired[da], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) &
		                       __hip_atomic_load(&fired[db], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) &
		                       __hip_atomic_load(&fired[dc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		if(ready) {
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
			if(NC) {
				// every value of the step in flight together, then the sums, then the stores (component by component the store of one
				// fenced in the loads of the next - v may alias itself - and a step was five L2 round trips instead of two)
				uint32_t g[NC ? NC : 1][4];
#pragma unroll
				for(uint32_t q = 0; q < (uint32_t)NC; q++) {
					g[q][0] = (uint32_t)v[(size_t)i*n + q]; g[q][1] = (uint32_t)v[(size_t)(inv || !at_start ? 0u : a)*n + q];
					g[q][2] = (uint32_t)v[(size_t)(inv || !para ? 0u : b)*n + q]; g[q][3] = (uint32_t)v[(size_t)(inv || !para ? 0u : c)*n + q];
				}
#pragma unroll
				for(uint32_t q = 0; q < (uint32_t)NC; q++) asm volatile("" : "+v"(g[q][0]), "+v"(g[q][1]), "+v"(g[q][2]), "+v"(g[q][3]));
#pragma unroll
				for(uint32_t q = 0; q < (uint32_t)NC; q++) {
					T x = (T)g[q][0];
					if(!inv) {
						const T pa = at_start ? (T)g[q][1] : prev[q];
						x = (T)(x + pa + (para ? (T)((T)g[q][2] - (T)g[q][3]) : (T)0));
						v[(size_t)i*n + q] = x;
					}
					prev[q] = x;
				}
			} else if(!inv) {
				for(uint32_t q = 0; q < n; q++)
					v[(size_t)i*n + q] = (T)(v[(size_t)i*n + q] + v[(size_t)a*n + q] + (para ? (T)(v[(size_t)b*n + q] - v[(size_t)c*n + q]) : (T)0));
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			__hip_atomic_store(&fired[i], (uint8_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			i++; at_start = false;
			a = na; b = nb; c = nc;
			take(i + 1, t2, na, nb, nc); have2 = false;                        // (issued at the top of this or an earlier pass: back by now)
			if(i == end) {
				k += THREADS; active = k < ns;
				if(active) {
					i = starts[k]; end = k + 1 < ns ? starts[k + 1] : nvert;
					const u32x3_t t0 = issue(i), t1 = issue(i + 1);
					take(i, t0, a, b, c); take(i + 1, t1, na, nb, nc); at_start = true;
				}
			}
		}
		if(!__any(ready)) __builtin_amdgcn_s_sleep(2);                      // nothing to do in this wave: leave the issue slots to the waves that fire
	}
}

__global__ __launch_bounds__(DELTA_THREADS) void k_delta_mesh(const DeltaJob *__restrict__ jobs, uint32_t njobs, uint32_t wide_n_only) {
	if(blockIdx.x >= njobs) return;
	const DeltaJob J = jobs[blockIdx.x];
	if(wide_n_only && J.N <= 4) return;                                      // (k_delta_tiles' job)
	const uint32_t THREADS = blockDim.x, nvert = J.nvert, t = threadIdx.x, lane = lane_id(), w = wave_id(), nwaves = THREADS >> 6;
	CRT_GLOBAL const uint32_t *pred = as_global(J.pred);
	CRT_GLOBAL uint8_t *fired = as_global(J.fired);                       // zero-filled by the host; the stretch starts live behind it
	CRT_GLOBAL uint32_t *starts = (CRT_GLOBAL uint32_t *)(fired + ((nvert + 15u) & ~15u));
	__shared__ uint32_t wcount[DELTA_THREADS/64];
	// stretch starts: vertices that do not predict from the vertex right before them (ordered compaction, THREADS vertices a round)
	uint32_t ns = 0;
	for(uint32_t base = 0; base < nvert; base += THREADS) {
		const uint32_t i = base + t;
		bool start = false;
		if(i < nvert) { const uint32_t a = pred[(size_t)i*3]; start = !(a < i && a + 1 == i); }
		const uint64_t m = __ballot(start);
		if(lane == 0) wcount[w] = __popcll(m);
		__syncthreads();
		uint32_t before = 0, total = 0;
		for(uint32_t q = 0; q < nwaves; q++) { const uint32_t x = wcount[q]; total += x; if(q < w) before += x; }
		if(start) starts[ns + before + __popcll(m & ((1ull << lane) - 1ull))] = i;
		ns += total;
		__syncthreads();
	}
	if(t == 0) __hip_atomic_store(&fired[0], (uint8_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	__threadfence_block();
	__syncthreads();
	const bool para = J.parallelogram != 0;
	if(J.is_u8) {
		CRT_GLOBAL uint8_t *v = as_global((uint8_t *)J.values);
		switch(J.N) {
		case 3: delta_stretch_global<uint8_t, 3>(v, fired, starts, ns, pred, nvert, J.N, para, THREADS); break;
		case 4: delta_stretch_global<uint8_t, 4>(v, fired, starts, ns, pred, nvert, J.N, para, THREADS); break;
		default: delta_stretch_global<uint8_t, 0>(v, fired, starts, ns, pred, nvert, J.N, para, THREADS); break;
		}
	} else {
		CRT_GLOBAL uint32_t *v = as_global((uint32_t *)J.values);
		switch(J.N) {
		case 1: delta_stretch_global<uint32_t, 1>(v, fired, starts, ns, pred, nvert, J.N, para, THREADS); break;
		case 2: delta_stretch_global<uint32_t, 2>(v, fired, starts, ns, pred, nvert, J.N, para, THREADS); break;
		case 3: delta_stretch_global<uint32_t, 3>(v, fired, starts, ns, pred, nvert, J.N, para, THREADS); break;
		default: delta_stretch_global<uint32_t, 0>(v, fired, starts, ns, pred, nvert, J.N, para, THREADS); break;
		}
	}
}

} // namespace corto_hip

#ifdef CORTO_TOPO_TIMES
extern "C" int crthip_debug_topo_times(uint32_t *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(corto_hip::g_topo_times), sizeof(uint32_t)*8*8192); }
#endif
#ifdef CORTO_TOPO_STAMPS
extern "C" int crthip_debug_topo_trace(uint32_t *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(corto_hip::g_topo_trace), sizeof(uint32_t)*16*8192); }
extern "C" int crthip_debug_topo_stamps(uint32_t *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(corto_hip::g_topo_stamps), sizeof(uint32_t)*48*4096); }
#endif
