// k_delta.hip — K-DELTA for blobs whose values and prediction graph fit LDS (round 3): v[i] += v[a] + v[b] - v[c] (or += v[a]) for
// i = 1 .. nvert-1 in index order (include/corto/vertex_attribute.h:160-176, src/normal_attribute.cpp:193-201).
//
// One workgroup per blob, one wave per attribute (up to four) sharing the prediction graph (round 2's k_delta_wave did the same with 32-bit
// values and in-order scans; round 4 retired it: the 32-bit records live here too, LdsW below).  What this kernel does:
//
// * Values live in LDS as 16-BIT integers RELATIVE TO VERTEX 0.  The recurrence is affine with weights +1 +1 -1, so subtracting
//   vertex 0's value from every vertex leaves it unchanged: rel[i] = d[i] + rel[a] + rel[b] - rel[c].  A mesh quantised to n bits
//   spans 2^n steps whatever its offset from the origin, so up to 15 bits the relative values fit an int16 - but the header does
//   not say so (no bounds, no bit count), hence it is CHECKED, not assumed: every raw delta and every result must fit, all sums
//   are formed in 32-bit registers from sign-extended operands (mod 2^32, the reference's int arithmetic), and by induction every
//   stored value then IS v[i] - v[0] mod 2^32.  An attribute that does not fit (18-bit positions, a malformed stream) is redone by
//   its wave on the 32-bit values in HBM with the same loop (slow, exact), the blob's flag word tells the host, and the context plans
//   its next batches on the wide kernel.  A C4 blob's three attributes + graph: 42 KB of LDS instead of 64.
//   Colours are bytes with the reference's mod-256 arithmetic: four to a dword, no base, nothing to check.
// * The graph is 4 bytes a vertex: b | c << 15 | (a == i-1) << 30 | (value stays) << 31, and `a` as a u16 that rides in the spare
//   halfword of a three-component attribute's 8-byte records when the group has one.
// * TWO loops: an OUT-OF-ORDER WINDOW, and - for what it is slow on - the round loop (a scan of affine maps, below).  Lane l looks at vertex s + l, s = the lowest vertex not done.  A vertex
//   can go when b, c (and a, unless it continues its predecessor's sum) are done - done = below s, or set in the 64-bit mask of the
//   window, which is ALL the bookkeeping there is (two SGPRs: nothing above the window is ever done) - and when, if it continues its
//   predecessor, that predecessor is done or goes in this same pass: a flood fill up the lanes, three scalar instructions.  The lanes
//   that go form runs; each run is a prefix sum from its head (one DPP scan per component + a bpermute of the head's exclusive sum).
//   A 4K-triangle grid takes 70 passes (the contiguous blocks of round 2: 110), a holey disc 80 (337), random diagonals 290 - which is why those
//   leave the window after 24 passes for the round loop (45 rounds of about two passes' cost) - tests/test_delta16_model_cpu.py restates both loops
//   on the host and checks them against the oracle.
// * The next window's graph words and raw values are fetched while the current pass gathers and scans.
#include <hip/hip_runtime.h>

#include "device_plan.h"
#include "kernels.h"
#include "kernels_common.h"

namespace corto_hip {
namespace {

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int32_t sx16(uint32_t w) { return (int32_t)(int16_t)(uint16_t)w; }

// ---- values in LDS: K = 1, 2, 3, 4 int16 components (records of 2, 4, 8, 8 bytes), K = 5: four bytes (colours) ----
template <int K> struct LdsVal;
template <> struct LdsVal<1> {
	static constexpr int NC = 1; static constexpr bool CHECK = true; typedef uint32_t Raw;
	CRT_LDS uint16_t *p;
	__device__ __forceinline__ Raw raw(uint32_t i) const { return p[i]; }
	__device__ __forceinline__ static void unpack(Raw w, int32_t (&v)[1]) { v[0] = sx16(w); }
	__device__ __forceinline__ void store(uint32_t i, const int32_t (&v)[1], uint32_t) const { p[i] = (uint16_t)v[0]; }
	__device__ __forceinline__ static int32_t wrap(int32_t v) { return sx16((uint32_t)v); }      // what a stored value reads back as
	__device__ __forceinline__ void sync() const {}
};
template <> struct LdsVal<2> {
	static constexpr int NC = 2; static constexpr bool CHECK = true; typedef uint32_t Raw;
	CRT_LDS uint32_t *p;
	__device__ __forceinline__ Raw raw(uint32_t i) const { return p[i]; }
	__device__ __forceinline__ static void unpack(Raw w, int32_t (&v)[2]) { v[0] = sx16(w); v[1] = (int32_t)w >> 16; }
	__device__ __forceinline__ void store(uint32_t i, const int32_t (&v)[2], uint32_t) const { p[i] = ((uint32_t)v[0] & 0xFFFFu) | ((uint32_t)v[1] << 16); }
	__device__ __forceinline__ static int32_t wrap(int32_t v) { return sx16((uint32_t)v); }
	__device__ __forceinline__ void sync() const {}
};
template <> struct LdsVal<3> {
	static constexpr int NC = 3; static constexpr bool CHECK = true; typedef u32x2 Raw;
	CRT_LDS u32x2 *p;
	__device__ __forceinline__ Raw raw(uint32_t i) const { return p[i]; }
	__device__ __forceinline__ static void unpack(Raw w, int32_t (&v)[3]) { v[0] = sx16(w.x); v[1] = (int32_t)w.x >> 16; v[2] = sx16(w.y); }
	__device__ __forceinline__ void store(uint32_t i, const int32_t (&v)[3], uint32_t a) const {
		p[i] = u32x2{((uint32_t)v[0] & 0xFFFFu) | ((uint32_t)v[1] << 16), ((uint32_t)v[2] & 0xFFFFu) | (a << 16)};
	}
	__device__ __forceinline__ static int32_t wrap(int32_t v) { return sx16((uint32_t)v); }
	__device__ __forceinline__ void sync() const {}
};
template <> struct LdsVal<4> {
	static constexpr int NC = 4; static constexpr bool CHECK = true; typedef u32x2 Raw;
	CRT_LDS u32x2 *p;
	__device__ __forceinline__ Raw raw(uint32_t i) const { return p[i]; }
	__device__ __forceinline__ static void unpack(Raw w, int32_t (&v)[4]) { v[0] = sx16(w.x); v[1] = (int32_t)w.x >> 16; v[2] = sx16(w.y); v[3] = (int32_t)w.y >> 16; }
	__device__ __forceinline__ void store(uint32_t i, const int32_t (&v)[4], uint32_t) const {
		p[i] = u32x2{((uint32_t)v[0] & 0xFFFFu) | ((uint32_t)v[1] << 16), ((uint32_t)v[2] & 0xFFFFu) | ((uint32_t)v[3] << 16)};
	}
	__device__ __forceinline__ static int32_t wrap(int32_t v) { return sx16((uint32_t)v); }
	__device__ __forceinline__ void sync() const {}
};
template <> struct LdsVal<5> {
	static constexpr int NC = 4; static constexpr bool CHECK = false; typedef uint32_t Raw;
	CRT_LDS uint32_t *p;
	__device__ __forceinline__ Raw raw(uint32_t i) const { return p[i]; }
	__device__ __forceinline__ static void unpack(Raw w, int32_t (&v)[4]) { v[0] = (int32_t)(w & 255u); v[1] = (int32_t)((w >> 8) & 255u); v[2] = (int32_t)((w >> 16) & 255u); v[3] = (int32_t)(w >> 24); }
	__device__ __forceinline__ void store(uint32_t i, const int32_t (&v)[4], uint32_t) const {
		p[i] = ((uint32_t)v[0] & 255u) | (((uint32_t)v[1] & 255u) << 8) | (((uint32_t)v[2] & 255u) << 16) | ((uint32_t)v[3] << 24);
	}
	__device__ __forceinline__ static int32_t wrap(int32_t v) { return (int32_t)((uint32_t)v & 255u); }
	__device__ __forceinline__ void sync() const {}
};

// the same four bytes as TWO registers of two 16-bit fields each (b0 | b2 << 16, b1 | b3 << 16), for colours WITHOUT parallelogram prediction
// (v += v[a]: additions only): a pass adds at most 65 bytes into a field, which cannot carry into its neighbour, so two wave scans do the
// work of four, and nothing is unpacked (mod 256 is taken when the record is stored)
template <> struct LdsVal<6> {
	static constexpr int NC = 2; static constexpr bool CHECK = false; typedef uint32_t Raw;
	CRT_LDS uint32_t *p;
	__device__ __forceinline__ Raw raw(uint32_t i) const { return p[i]; }
	__device__ __forceinline__ static void unpack(Raw w, int32_t (&v)[2]) { v[0] = (int32_t)(w & 0x00FF00FFu); v[1] = (int32_t)((w >> 8) & 0x00FF00FFu); }
	__device__ __forceinline__ void store(uint32_t i, const int32_t (&v)[2], uint32_t) const { p[i] = ((uint32_t)v[0] & 0x00FF00FFu) | (((uint32_t)v[1] & 0x00FF00FFu) << 8); }
	__device__ __forceinline__ static int32_t wrap(int32_t v) { return (int32_t)((uint32_t)v & 0x00FF00FFu); }
	__device__ __forceinline__ void sync() const {}
};

// ---- the same records with 32-BIT components (round 4: a context that met values beyond int16 - positions quantised to 16+ bits - plans
// its batches this way; rounds 2-3 kept a second kernel, k_delta_wave, for them).  Records of 4 / 8 / 16 / 16 bytes; absolute values (no
// base, nothing to check); a three-component record's fourth dword carries the graph's `a`.
template <int K> struct LdsW;
template <> struct LdsW<1> {
	static constexpr int NC = 1; static constexpr bool CHECK = false; typedef uint32_t Raw;
	CRT_LDS uint32_t *p;
	__device__ __forceinline__ Raw raw(uint32_t i) const { return p[i]; }
	__device__ __forceinline__ static void unpack(Raw w, int32_t (&v)[1]) { v[0] = (int32_t)w; }
	__device__ __forceinline__ void store(uint32_t i, const int32_t (&v)[1], uint32_t) const { p[i] = (uint32_t)v[0]; }
	__device__ __forceinline__ static int32_t wrap(int32_t v) { return v; }
	__device__ __forceinline__ void sync() const {}
};
template <> struct LdsW<2> {
	static constexpr int NC = 2; static constexpr bool CHECK = false; typedef u32x2 Raw;
	CRT_LDS u32x2 *p;
	__device__ __forceinline__ Raw raw(uint32_t i) const { return p[i]; }
	__device__ __forceinline__ static void unpack(Raw w, int32_t (&v)[2]) { v[0] = (int32_t)w.x; v[1] = (int32_t)w.y; }
	__device__ __forceinline__ void store(uint32_t i, const int32_t (&v)[2], uint32_t) const { p[i] = u32x2{(uint32_t)v[0], (uint32_t)v[1]}; }
	__device__ __forceinline__ static int32_t wrap(int32_t v) { return v; }
	__device__ __forceinline__ void sync() const {}
};
template <> struct LdsW<3> {
	static constexpr int NC = 3; static constexpr bool CHECK = false; typedef u32x4 Raw;
	CRT_LDS u32x4 *p;
	__device__ __forceinline__ Raw raw(uint32_t i) const { return p[i]; }
	__device__ __forceinline__ static void unpack(Raw w, int32_t (&v)[3]) { v[0] = (int32_t)w.x; v[1] = (int32_t)w.y; v[2] = (int32_t)w.z; }
	__device__ __forceinline__ void store(uint32_t i, const int32_t (&v)[3], uint32_t a) const { p[i] = u32x4{(uint32_t)v[0], (uint32_t)v[1], (uint32_t)v[2], a}; }
	__device__ __forceinline__ static int32_t wrap(int32_t v) { return v; }
	__device__ __forceinline__ void sync() const {}
};
template <> struct LdsW<4> {
	static constexpr int NC = 4; static constexpr bool CHECK = false; typedef u32x4 Raw;
	CRT_LDS u32x4 *p;
	__device__ __forceinline__ Raw raw(uint32_t i) const { return p[i]; }
	__device__ __forceinline__ static void unpack(Raw w, int32_t (&v)[4]) { v[0] = (int32_t)w.x; v[1] = (int32_t)w.y; v[2] = (int32_t)w.z; v[3] = (int32_t)w.w; }
	__device__ __forceinline__ void store(uint32_t i, const int32_t (&v)[4], uint32_t) const { p[i] = u32x4{(uint32_t)v[0], (uint32_t)v[1], (uint32_t)v[2], (uint32_t)v[3]}; }
	__device__ __forceinline__ static int32_t wrap(int32_t v) { return v; }
	__device__ __forceinline__ void sync() const {}
};

// ---- values in HBM, as the caller / the bit-unpack left them (the redo path of an attribute whose relative values left int16): any
// N <= 4, int32 or bytes.  One wave owns the attribute (workgroup-scope accesses: the CU's own cache is coherent for it) and a pass's
// stores are waited for before the next pass reads.
template <typename T> struct GlobalVal {
	static constexpr int NC = 4; static constexpr bool CHECK = false;
	struct Raw { uint32_t v[4]; };
	CRT_GLOBAL T *p; uint32_t N;                              // N: components of this run (<= 4)
	uint32_t stride = 0;                                      // elements from one vertex to the next (0: N); p points at the run's first component
	__device__ __forceinline__ uint32_t st() const { return stride ? stride : N; }
	__device__ __forceinline__ Raw raw(uint32_t i) const {
		Raw r;
#pragma unroll
		for(uint32_t q = 0; q < 4; q++) r.v[q] = q < N ? (uint32_t)__hip_atomic_load(p + (size_t)i*st() + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u;
		return r;
	}
	__device__ __forceinline__ static void unpack(const Raw &w, int32_t (&v)[4]) { v[0] = (int32_t)w.v[0]; v[1] = (int32_t)w.v[1]; v[2] = (int32_t)w.v[2]; v[3] = (int32_t)w.v[3]; }
	__device__ __forceinline__ void store(uint32_t i, const int32_t (&v)[4], uint32_t) const {
#pragma unroll
		for(uint32_t q = 0; q < 4; q++) if(q < N) __hip_atomic_store(p + (size_t)i*st() + q, (T)v[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	}
	__device__ __forceinline__ static int32_t wrap(int32_t v) { return (int32_t)(T)v; }
	__device__ __forceinline__ void sync() const { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};

// where a vertex' `a` is: a u16 array of its own (shift 1), or the spare halfword of a three-component attribute's records (shift 3)
struct GaRef {
	uint32_t addr, shift;
	__device__ __forceinline__ uint32_t get(uint32_t i) const { return *lds_at<const uint16_t>(addr + (i << shift)); }
	__device__ __forceinline__ void put(uint32_t i, uint32_t a) const { *lds_at<uint16_t>(addr + (i << shift)) = (uint16_t)a; }
};

constexpr uint32_t GW_CHAINED = 1u << 30, GW_STAYS = 1u << 31, GW_NO_BC = 0x3FFFFFFFu;

// a vertex' graph word and `a`, made once by the builder wave and kept in LDS
__device__ __forceinline__ uint32_t graph_word(uint32_t i, uint32_t a, uint32_t b, uint32_t c) {
	// well-formed streams always predict from earlier vertices.  A triple that does not: the value stays (vertex 0 is one) - for
	// every attribute when it is `a`, for parallelogram attributes only when it is b or c (the others never look at them):
	// GW_STAYS for the first, b = c = 0x7FFF (no vertex: nvert <= 32767) for the second
	const bool va = a < i, vbc = b < i && c < i;
	uint32_t word = !va ? GW_STAYS : vbc ? (b | (c << 15)) : GW_NO_BC;
	if(va && a + 1u == i) word |= GW_CHAINED;
	return word;
}
struct GraphLds {
	CRT_LDS const uint32_t *gw; GaRef ga;
	__device__ __forceinline__ void fetch(uint32_t i, uint32_t &W, uint32_t &A) const { W = gw[i]; A = ga.get(i); }
};

// The window loop.  `base`: what a vertex whose value stays (malformed triple) has to give up to become relative (0 for bytes / HBM).
// Returns the OR over every stored component of (value + 0x8000): anything at or above bit 16 = a value left int16.
// `hand`: null, or where to leave (s, window mask) when the loop gives up in favour of the round loop below - decided at pass 24 from what passes
// 8 .. 23 looked like, and asked again every sixteen passes (a mesh may start as a grid and go on as something else): a window pass costs ~1 100 clocks whatever it finishes, a round of the other loop ~2 200 for ~47 vertices, so fewer
// than 18 vertices a pass says rounds: random diagonals (9-11 a pass), Delaunay meshes, decimated and other irregular closed meshes (3.5), tori
// (13) - not grids (25-29); a holey disc (20) sits at the threshold.  (Rounds 3-4 handed over to a WALK - one lane per stretch, a vertex a pass,
// passes = the DAG's depth: 160-180 for a 4K-triangle blob, but 1 772 for a decimated sphere, which is one stretch; the round loop beats it on
// every family - tests/test_delta16_model_cpu.py - and it is gone.)
struct WindowHand { uint32_t s; uint64_t donew; };                        // s < nvert: the window handed over to the round loop at vertex s, `donew` = what it had finished out of order from there
// PARA: parallelogram prediction (b, c are gathered); else v += v[a] alone - the same loop without the two gathers, their unpacking and
// their ready tests (a third of a pass's vector instructions, for two of a C4 blob's three attributes: the pipelined rate is within 2x of
// the chip's VALU issue rate)
template <bool PARA, class V, class GR>
__device__ __forceinline__ uint32_t delta_window_loop(const V &val, const GR &graph, const uint32_t nvert,
                                                     const int32_t (&base)[V::NC], WindowHand *hand) {
	constexpr int NC = V::NC;
	const uint32_t lane = lane_id();
	const uint64_t lane_le = (2ull << lane) - 1ull;                          // lanes 0 .. mine
	constexpr uint32_t wmask = PARA ? 0xFFFFFFFFu : (GW_CHAINED | GW_STAYS);  // v += v[a] alone: b, c are not looked at
	uint32_t s = 1, bad = 0;
	uint64_t donew = 0;                                                       // bit l: vertex s + l is done (everything below s is; nothing at or above s + 64 can be)
	uint32_t W, A; typename V::Raw D;
	{ const uint32_t ic = s + lane < nvert ? s + lane : nvert - 1u; graph.fetch(ic, W, A); W &= wmask; D = val.raw(ic); }
	uint32_t passes = 0, ngo = 0;
	if(hand) { hand->s = nvert; hand->donew = 0; }
	while(s < nvert) {
		if(hand && passes >= 24u && (passes & 15u) == 8u) {                       // after passes 8-23, 24-39, ...: what did the last sixteen finish?
			if(nvert - s >= 128u && ngo < 288u) { hand->s = s; hand->donew = donew; break; }   // fewer than 18 vertices a pass: the round loop takes over
			ngo = 0;                                                               // (a mesh may turn irregular later: the question is asked again every sixteen passes)
		}
		const uint32_t i = s + lane;
		const bool in = i < nvert;
		const uint32_t b = W & 0x7FFFu, c = (W >> 15) & 0x7FFFu;
		const bool ch = (W & GW_CHAINED) != 0, stays = (W & GW_STAYS) != 0 || (PARA && (W & GW_NO_BC) == GW_NO_BC);
		const uint64_t d1 = (donew << 1) | 1ull;                              // bit r: vertex s - 1 + r is done (r = 0 stands for everything below the window)
		auto ready = [&](uint32_t x) -> uint32_t { const int32_t r = (int32_t)(x - s) + 1; return (uint32_t)(d1 >> (uint32_t)(r > 0 ? r : 0)) & 1u; };   // (x < i: r <= 63)
		const bool done_i = __builtin_amdgcn_inverse_ballot_w64(donew), pred_done = __builtin_amdgcn_inverse_ballot_w64(d1);
		const bool H = stays || !ch || pred_done;                             // starts a sum of its own: its v[a] is fetched, not scanned in
		const uint32_t rdy = stays ? 1u : ((PARA ? ready(b) & ready(c) : 1u) & (ch ? 1u : ready(A)));
		const bool R = in && !done_i && rdy != 0;
		const uint64_t Rm = __ballot(R), Sm = __ballot(R && H);
		// the lanes that go: flood fill from the heads (Sm) up through consecutive ready lanes that continue their predecessor - adding the
		// heads to the ready mask ripples a carry through exactly those
		const uint64_t G = (((Rm + Sm) ^ Rm) & Rm) | Sm;
		const bool go = __builtin_amdgcn_inverse_ballot_w64(G), head = __builtin_amdgcn_inverse_ballot_w64(Sm);
		if(passes >= 8u) ngo += (uint32_t)__builtin_popcountll(G);
		passes++;
		const uint64_t dn = donew | G;
		const uint32_t t = ~dn ? (uint32_t)__builtin_ctzll(~dn) : 64u;       // lane 0 always goes: t >= 1
		const uint32_t s_next = s + t;
		const uint64_t donew_next = t < 64 ? dn >> t : 0ull;
		// this pass's gathers first, then the next window's words: the gathers are what the scans wait for
		const bool use = go && !stays;
		const uint32_t gb = use ? b : 0u, gc = use ? c : 0u, gp = use && head ? (ch ? i - 1u : A) : 0u;
		typename V::Raw Bw, Cw;
		if constexpr(PARA) { Bw = val.raw(gb); Cw = val.raw(gc); }
		const typename V::Raw Pw = val.raw(gp);
		uint32_t W2, A2; typename V::Raw D2;
		{ const uint32_t ic = s_next + lane < nvert ? s_next + lane : nvert - 1u; graph.fetch(ic, W2, A2); W2 &= wmask; D2 = val.raw(ic); }
		int32_t dv[NC], bv[NC], cv[NC], pv[NC], x[NC];
		V::unpack(D, dv); V::unpack(Pw, pv);
		if constexpr(PARA) { V::unpack(Bw, bv); V::unpack(Cw, cv); }
		uint32_t incl[NC], eh[NC];
#pragma unroll
		for(int q = 0; q < NC; q++) {
			int32_t v;
			if constexpr(PARA) v = dv[q] + (use ? bv[q] - cv[q] + (head ? pv[q] : 0) : -base[q]);
			else v = dv[q] + (use ? (head ? pv[q] : 0) : -base[q]);
			x[q] = go ? v : 0;
		}
#pragma unroll
		for(int q = 0; q < NC; q++) incl[q] = wave_inclusive_scan_u32((uint32_t)x[q]);
		// my run's head = the highest head lane at or below me
		const uint64_t hm = Sm & lane_le;
		const uint32_t hidx = hm ? 63u - (uint32_t)__builtin_clzll(hm) : 0u;
#pragma unroll
		for(int q = 0; q < NC; q++) eh[q] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(hidx << 2), (int)(incl[q] - (uint32_t)x[q]));
		int32_t r[NC];
#pragma unroll
		for(int q = 0; q < NC; q++) r[q] = (int32_t)(incl[q] - eh[q]);
		if(go) {
			if(V::CHECK) {
#pragma unroll
				for(int q = 0; q < NC; q++) bad |= (uint32_t)r[q] + 0x8000u;
			}
			val.store(i, r, A);
		}
		val.sync();
		s = s_next; donew = donew_next; W = W2; A = A2; D = D2;
	}
	return bad;
}
template <class V, class GR>
__device__ __forceinline__ uint32_t delta_window_run(const V &val, const GR &graph, const uint32_t nvert, const bool para,
                                                    const int32_t (&base)[V::NC], WindowHand *hand = nullptr) {
	return para ? delta_window_loop<true>(val, graph, nvert, base, hand) : delta_window_loop<false>(val, graph, nvert, base, hand);
}

// ---- the round loop: parents a vertex or two back, by a scan of 2 x 2 affine maps (round 5) ----
// The window finishes a run of vertices per pass as long as b and c are DONE: on a grid they lie a ring back and a pass takes 30 vertices.  On
// anything irregular - flipped diagonals, a Delaunay mesh, a decimated or otherwise irregular closed mesh (fans: c = i - 2 or b = i - 2) - they are the
// vertices just made, a run is 3.5 to 10 vertices, and the window takes hundreds of passes of ~1 100 clocks (a decimated sphere of 2 049 vertices:
// 584, 0.31 ms); rounds 3-4 walked many stretches at once instead (one vertex a lane and pass: the DAG's depth in passes, 160-180 for a 4K-triangle
// blob - and 1 772 for the decimated sphere, which is ONE stretch).  But the dependencies that break the window's runs are almost all of ONE kind: a
// parent ONE or TWO vertices back (measured on every family: a vertex whose nearest in-flight parent lies further back comes once in ~50).  With
//     v[i] = pre[i] + ca[i] v[i-1] + cb[i] v[i-2]        pre = d[i] + the parents that are final,  ca, cb in {-1 .. 3} = how often i-1 / i-2 is a parent (+a +b -c)
// the state (v[i], v[i-1]) is an AFFINE function of (v[i-1], v[i-2]) - T_i = [[ca, cb], [1, 0]], translation (pre, 0) - and affine maps compose
// associatively: an inclusive scan of the maps over 64 lanes (six steps of a 2 x 2 product and a 2 x 2 . vector + vector a component, mod 2^32)
// leaves v[i] in lane i.  A round takes the vertices from the lowest unfinished one up to the first whose in-flight parent lies more than two back
// (that one starts the next round, where its parents are final): 43-52 rounds for a 2K-vertex blob of ANY of those families, ~2 000 clocks each.
// Exact by the same argument as the other loops (sums in 32-bit registers mod 2^32, the stored int16 checked); the linear parts are shared by an
// attribute's components.  Chosen by what the window's passes 8-23 looked like (WindowHand.mode): fewer than 18 vertices a pass.
// (PARA = false: v += v[a] alone - the same loop with b and c left out; byte records: the sums mod 2^32 are stored mod 256.)
template <bool PARA, class V>
__device__ __forceinline__ uint32_t delta_round_loop(const V &val, CRT_LDS const uint32_t *gw, const GaRef ga, const uint32_t nvert,
                                                    const int32_t (&base)[V::NC], const WindowHand hand) {
	constexpr int NC = V::NC;
	const uint32_t lane = lane_id();
	uint32_t bad = 0;
	uint64_t donew = hand.donew;
	uint32_t s = hand.s;
	while(s < nvert) {
		const uint32_t i = s + lane;
		const bool in = i < nvert;
		const uint32_t ic = in ? i : nvert - 1u;
		const uint32_t W = gw[ic], A = ga.get(ic);
		const typename V::Raw D = val.raw(ic);
		const bool done_i = __builtin_amdgcn_inverse_ballot_w64(donew);       // the window finished it out of order: its record IS its value
		const uint32_t b = W & 0x7FFFu, c = (W >> 15) & 0x7FFFu;
		const bool ch = (W & GW_CHAINED) != 0, stays = (W & GW_STAYS) != 0 || (PARA && (W & GW_NO_BC) == GW_NO_BC);
		const bool self = !in || done_i || stays;                             // no parents to add
		const uint32_t pa = ch ? ic - 1u : A;
		// a parent inside the round (>= s) must be one or two back; the first lane with one further back ends the round
		const uint32_t da = ic - pa, db = ic - b, dc = ic - c;                   // (parents are below the vertex: distances >= 1)
		const bool na = !self && pa >= s, nb = PARA && !self && b >= s, nc = PARA && !self && c >= s;
		const uint64_t cut = __ballot(!in || (na && da > 2u) || (nb && db > 2u) || (nc && dc > 2u));
		const uint32_t len = cut ? (uint32_t)__builtin_ctzll(cut) : 64u;          // (lane 0 has no parent inside the round: len >= 1)
		const typename V::Raw Pw = val.raw(self || na ? 0u : pa);
		int32_t dv[NC], pv[NC], bv[NC], cv[NC];
		V::unpack(D, dv); V::unpack(Pw, pv);
		if constexpr(PARA) {
			const typename V::Raw Bw = val.raw(self || nb ? 0u : b), Cw = val.raw(self || nc ? 0u : c);
			V::unpack(Bw, bv); V::unpack(Cw, cv);
		} else {
#pragma unroll
			for(int q = 0; q < NC; q++) bv[q] = cv[q] = 0;
		}
		// the lane's map: (x, y) -> (ca x + cb y + pre, x)
		uint32_t m00 = (na && da == 1u ? 1u : 0u) + (nb && db == 1u ? 1u : 0u) - (nc && dc == 1u ? 1u : 0u);
		uint32_t m01 = (na && da == 2u ? 1u : 0u) + (nb && db == 2u ? 1u : 0u) - (nc && dc == 2u ? 1u : 0u);
		uint32_t m10 = 1u, m11 = 0u;
		uint32_t t0[NC], t1[NC];
#pragma unroll
		for(int q = 0; q < NC; q++) {
			t0[q] = (uint32_t)(done_i ? dv[q] : stays ? dv[q] - base[q] : dv[q] + (na ? 0 : pv[q]) + (!PARA || nb ? 0 : bv[q]) - (!PARA || nc ? 0 : cv[q]));
			t1[q] = 0u;
		}
		// inclusive scan of the maps: inside every 16-lane row by DPP row shifts (1, 2, 4, 8), then the total of row 0 / row 2 into rows 1 / 3 and the total of
		// rows 0-1 into rows 2-3 (row_bcast:15, row_bcast:31) - the steps of wave_inclusive_scan_u32 with the maps' composition for the sum.  A lane takes the
		// prefix that ends in front of its own and puts its own behind it; a lane the step has no source for gets the identity map and stays as it is.
		// (Round 5 began with ds_bpermute: 56 LDS instructions a round at 16 clocks of issue each and six exposed round trips - tools/micro/issue_latency.hip.)
#define CRT_MAP_GET(x, ident, ctrl, rmask) ((uint32_t)__builtin_amdgcn_update_dpp((int)(ident), (int)(x), ctrl, rmask, 0xf, false))
#define CRT_MAP_STEP(ctrl, rmask, last) { \
			uint32_t a00 = 1u, a01 = 0u, a10 = 0u, a11 = 1u, u0[NC], u1[NC]; \
			if(!(last)) { a00 = CRT_MAP_GET(m00, 1u, ctrl, rmask); a01 = CRT_MAP_GET(m01, 0u, ctrl, rmask); a10 = CRT_MAP_GET(m10, 0u, ctrl, rmask); a11 = CRT_MAP_GET(m11, 1u, ctrl, rmask); } \
			_Pragma("unroll") for(int q = 0; q < NC; q++) { u0[q] = CRT_MAP_GET(t0[q], 0u, ctrl, rmask); u1[q] = CRT_MAP_GET(t1[q], 0u, ctrl, rmask); } \
			_Pragma("unroll") for(int q = 0; q < NC; q++) { \
				const uint32_t n0 = m00*u0[q] + m01*u1[q] + t0[q], n1 = m10*u0[q] + m11*u1[q] + t1[q]; \
				t0[q] = n0; t1[q] = n1; \
			} \
			if(!(last)) {                                                          /* (the linear part is not needed behind the last step) */ \
				const uint32_t n00 = m00*a00 + m01*a10, n01 = m00*a01 + m01*a11, n10 = m10*a00 + m11*a10, n11 = m10*a01 + m11*a11; \
				m00 = n00; m01 = n01; m10 = n10; m11 = n11; \
			} }
		CRT_MAP_STEP(0x111, 0xf, false)      // row_shr:1
		CRT_MAP_STEP(0x112, 0xf, false)      // row_shr:2
		CRT_MAP_STEP(0x114, 0xf, false)      // row_shr:4
		CRT_MAP_STEP(0x118, 0xf, false)      // row_shr:8
		CRT_MAP_STEP(0x142, 0xa, false)      // row_bcast:15 into rows 1 and 3
		CRT_MAP_STEP(0x143, 0xc, true)       // row_bcast:31 into rows 2 and 3
#undef CRT_MAP_STEP
#undef CRT_MAP_GET
		if(lane < len && !done_i) {
			int32_t r[NC];
#pragma unroll
			for(int q = 0; q < NC; q++) r[q] = (int32_t)t0[q];
			if(V::CHECK) {
#pragma unroll
				for(int q = 0; q < NC; q++) bad |= (uint32_t)r[q] + 0x8000u;
			}
			val.store(i, r, A);
		}
		val.sync();
		s += len;
		donew = len < 64u ? donew >> len : 0ull;
	}
	return bad;
}

// ---- staging: raw int32 deltas in HBM -> int16 records (checked); results back as the int32 / float the caller wants ----
template <int K, typename S>                                                // S: int32_t, or int16_t when K-BIT wrote halfwords (DeltaJob.pad[0])
__device__ __forceinline__ uint32_t stage_in16(const LdsVal<K> &val, CRT_GLOBAL const S *src, uint32_t nvert, bool ga_here) {
	constexpr int NC = LdsVal<K>::NC;
	uint32_t bad = 0;
	constexpr uint32_t U = NC <= 2 ? 16u : 8u;                              // vertices per lane and round: every load of a round in flight together
	for(uint32_t i0 = lane_id(); i0 < nvert; i0 += 64*U) {                   // (a 2 112-vertex blob: five rounds of 8 x 12 bytes, not nine of four)
		int32_t d[U][NC];
#pragma unroll
		for(uint32_t u = 0; u < U; u++) {
			const uint32_t i = i0 + u*64 < nvert ? i0 + u*64 : nvert - 1u;
#pragma unroll
			for(int q = 0; q < NC; q++) d[u][q] = src[(size_t)i*NC + q];
		}
#pragma unroll
		for(uint32_t u = 0; u < U; u++)
#pragma unroll
			for(int q = 0; q < NC; q++) asm volatile("" : "+v"(d[u][q]));       // all loads of the round in flight before the first is used
#pragma unroll
		for(uint32_t u = 0; u < U; u++) {
			const uint32_t i = i0 + u*64;
			if(i >= nvert) continue;
			if(i == 0) {
#pragma unroll
				for(int q = 0; q < NC; q++) d[u][q] = 0;                        // vertex 0 is the base: relative value 0
			}
#pragma unroll
			for(int q = 0; q < NC; q++) bad |= (uint32_t)d[u][q] + 0x8000u;
			// (K = 3: the spare halfword belongs to the graph builder, which may have written it already: x | y, then z alone)
			if(K == 3 && ga_here) {
				CRT_LDS uint32_t *p32 = (CRT_LDS uint32_t *)(val.p + i);
				p32[0] = ((uint32_t)d[u][0] & 0xFFFFu) | ((uint32_t)d[u][1] << 16);
				*(CRT_LDS uint16_t *)(p32 + 1) = (uint16_t)d[u][NC > 2 ? 2 : 0];
			} else val.store(i, d[u], 0u);
		}
	}
	return bad;
}

// 32-bit records: the raw deltas as they are (absolute values, nothing to check); K = 3 leaves the record's fourth dword to the graph builder
template <int K>
__device__ __forceinline__ void stage_in32(const LdsW<K> &val, CRT_GLOBAL const int32_t *src, uint32_t nvert) {
	constexpr int NC = LdsW<K>::NC;
	constexpr uint32_t U = NC <= 2 ? 16u : 8u;
	for(uint32_t i0 = lane_id(); i0 < nvert; i0 += 64*U) {
		int32_t d[U][NC];
#pragma unroll
		for(uint32_t u = 0; u < U; u++) {
			const uint32_t i = i0 + u*64 < nvert ? i0 + u*64 : nvert - 1u;
#pragma unroll
			for(int q = 0; q < NC; q++) d[u][q] = src[(size_t)i*NC + q];
		}
#pragma unroll
		for(uint32_t u = 0; u < U; u++)
#pragma unroll
			for(int q = 0; q < NC; q++) asm volatile("" : "+v"(d[u][q]));
#pragma unroll
		for(uint32_t u = 0; u < U; u++) {
			const uint32_t i = i0 + u*64;
			if(i >= nvert) continue;
			if(K == 3) {
				CRT_LDS uint32_t *p32 = (CRT_LDS uint32_t *)(val.p + i);
				*(CRT_LDS u32x2 *)p32 = u32x2{(uint32_t)d[u][0], (uint32_t)d[u][1]};
				p32[2] = (uint32_t)d[u][NC > 2 ? 2 : 0];
			} else val.store(i, d[u], 0u);
		}
	}
}
template <int K>
__device__ __forceinline__ void stage_out32(const LdsW<K> &val, CRT_GLOBAL int32_t *dst, uint32_t nvert, bool as_float, float q) {
	constexpr int NC = LdsW<K>::NC;
	CRT_GLOBAL float *fdst = (CRT_GLOBAL float *)dst;
	for(uint32_t i0 = lane_id(); i0 < nvert; i0 += 64*8) {
		typename LdsW<K>::Raw w[8];
#pragma unroll
		for(uint32_t u = 0; u < 8; u++) w[u] = val.raw(i0 + u*64 < nvert ? i0 + u*64 : nvert - 1u);
#pragma unroll
		for(uint32_t u = 0; u < 8; u++) {
			const uint32_t i = i0 + u*64;
			if(i >= nvert) continue;
			int32_t v[NC];
			LdsW<K>::unpack(w[u], v);
			if(as_float) {
#pragma unroll
				for(int c = 0; c < NC; c++) fdst[(size_t)i*NC + c] = (float)v[c]*q;
			} else {
#pragma unroll
				for(int c = 0; c < NC; c++) dst[(size_t)i*NC + c] = v[c];
			}
		}
	}
}

__device__ __forceinline__ void stage_in_bytes(const LdsVal<5> &val, CRT_GLOBAL const uint8_t *src, uint32_t nvert, uint32_t N, uint32_t first = lane_id(), uint32_t step = 64u) {
	if(N == 4 && ((uintptr_t)src & 3) == 0) {
		CRT_GLOBAL const uint32_t *s4 = (CRT_GLOBAL const uint32_t *)src;
		for(uint32_t i0 = first; i0 < nvert; i0 += step*8) {
			uint32_t w[8];
#pragma unroll
			for(uint32_t u = 0; u < 8; u++) w[u] = s4[i0 + u*step < nvert ? i0 + u*step : nvert - 1u];
#pragma unroll
			for(uint32_t u = 0; u < 8; u++) asm volatile("" : "+v"(w[u]));
#pragma unroll
			for(uint32_t u = 0; u < 8; u++) if(i0 + u*step < nvert) val.p[i0 + u*step] = w[u];
		}
		return;
	}
	for(uint32_t i = first; i < nvert; i += step) {
		uint32_t w = 0;
		for(uint32_t q = 0; q < N && q < 4; q++) w |= (uint32_t)src[(size_t)i*N + q] << (8*q);
		val.p[i] = w;
	}
}

// generic attribute out: base + relative value, as int32 (in place of the deltas) or as (float)v*q (vertex_attribute.h:190-193)
template <int K>
__device__ __forceinline__ void stage_out16(const LdsVal<K> &val, CRT_GLOBAL int32_t *dst, uint32_t nvert, const int32_t (&base)[LdsVal<K>::NC], bool as_float, float q) {
	constexpr int NC = LdsVal<K>::NC;
	CRT_GLOBAL float *fdst = (CRT_GLOBAL float *)dst;
	for(uint32_t i0 = lane_id(); i0 < nvert; i0 += 64*8) {
		typename LdsVal<K>::Raw w[8];
#pragma unroll
		for(uint32_t u = 0; u < 8; u++) w[u] = val.raw(i0 + u*64 < nvert ? i0 + u*64 : nvert - 1u);
#pragma unroll
		for(uint32_t u = 0; u < 8; u++) {
			const uint32_t i = i0 + u*64;
			if(i >= nvert) continue;
			int32_t v[NC];
			LdsVal<K>::unpack(w[u], v);
#pragma unroll
			for(int c = 0; c < NC; c++) v[c] += base[c];
			if(as_float) {
#pragma unroll
				for(int c = 0; c < NC; c++) fdst[(size_t)i*NC + c] = (float)v[c]*q;
			} else {
#pragma unroll
				for(int c = 0; c < NC; c++) dst[(size_t)i*NC + c] = v[c];
			}
		}
	}
}

// a colour attribute leaves LDS as RGB(A): (r, g, b, a) = (v2 + v0, v0, v1 + v0, v3) x qc, u8 wrap (color_attribute.cpp:76-95, point.h:214),
// or as the delta-decoded bytes when k_dequant does that later
__device__ __forceinline__ void stage_out_bytes(const LdsVal<5> &val, const DeltaJob &J, uint32_t q0, uint32_t q1, uint32_t q2, uint32_t q3, uint32_t first = lane_id(), uint32_t step = 64u) {
	const uint32_t nvert = J.nvert, N = J.N;
	if(J.deq != 2) {                                                          // bytes back where they came from
		CRT_GLOBAL uint8_t *dst = as_global((uint8_t *)J.values);
		if(N == 4 && ((uintptr_t)dst & 3) == 0) { for(uint32_t i = first; i < nvert; i += step) ((CRT_GLOBAL uint32_t *)dst)[i] = val.p[i]; return; }
		for(uint32_t i = first; i < nvert; i += step) { const uint32_t w = val.p[i]; for(uint32_t c = 0; c < N && c < 4; c++) dst[(size_t)i*N + c] = (uint8_t)(w >> (8*c)); }
		return;
	}
	CRT_GLOBAL uint8_t *dst = as_global((uint8_t *)J.out);
	const uint32_t oc = J.out_components, stride = J.out_stride ? J.out_stride : oc;
	const uint32_t amask = N < 4 ? 0xFF000000u : 0u;                          // fewer than four stored components: alpha 255 (color_attribute.h: out_components > N)
	auto px = [&](uint32_t w) -> uint32_t {                                   // bytes y, u, v, a -> r, g, b, a
		w |= amask;
		const uint32_t y = w & 255u, cu = (w >> 8) & 255u, cv = (w >> 16) & 255u, al = w >> 24;
		return (((cv + y)*q0) & 255u) | ((y*q1) & 255u) << 8 | (((cu + y)*q2) & 255u) << 16 | ((al*q3) & 255u) << 24;
	};
	if(oc == 4 && stride == 4 && ((uintptr_t)dst & 15) == 0) {                // packed RGBA: four vertices per lane, 16-byte stores
		CRT_LDS const u32x4 *v4 = (CRT_LDS const u32x4 *)val.p;               // (the record array starts on a 16-byte multiple)
		CRT_GLOBAL u32x4 *d4 = (CRT_GLOBAL u32x4 *)dst;
		const uint32_t nq = nvert >> 2;
		for(uint32_t k = first; k < nq; k += step) { const u32x4 w = v4[k]; d4[k] = u32x4{px(w.x), px(w.y), px(w.z), px(w.w)}; }
		for(uint32_t i = (nq << 2) + first; i < nvert; i += step) ((CRT_GLOBAL uint32_t *)dst)[i] = px(val.p[i]);
		return;
	}
	for(uint32_t i = first; i < nvert; i += step) {
		const uint32_t w = px(val.p[i]);
		CRT_GLOBAL uint8_t *o = dst + (size_t)i*stride;
		if(oc == 4 && (((uintptr_t)o) & 3) == 0) *(CRT_GLOBAL uint32_t *)o = w;
		else for(uint32_t c = 0; c < oc && c < 4; c++) o[c] = (uint8_t)(w >> (8*c));
	}
}

// one int16 attribute by its wave, in two phases with the workgroup's barrier between them (the graph is another wave's work)
template <int K>
__device__ __forceinline__ uint32_t delta16_in(CRT_LDS uint8_t *rec, const DeltaJob &J, bool ga_here) {
	LdsVal<K> val{(decltype(LdsVal<K>::p))rec};
	if(J.pad[0]) return stage_in16<K, int16_t>(val, as_global((const int16_t *)J.values), J.nvert, ga_here);
	return stage_in16<K, int32_t>(val, as_global((const int32_t *)J.values), J.nvert, ga_here);
}
// returns true if the relative values left int16 (nothing was written back: the caller redoes the attribute in HBM)
template <int K>
__device__ __forceinline__ bool delta16_run(CRT_LDS uint8_t *rec, const DeltaJob &J, CRT_LDS const uint32_t *gw, const GaRef ga, uint32_t bad) {
	constexpr int NC = LdsVal<K>::NC;
	LdsVal<K> val{(decltype(LdsVal<K>::p))rec};
	CRT_GLOBAL const int32_t *src = as_global((const int32_t *)J.values);
	int32_t base[NC];
#pragma unroll
	for(int q = 0; q < NC; q++) base[q] = J.pad[0] ? (int32_t)((CRT_GLOBAL const int16_t *)src)[q] : src[q];   // vertex 0 (every lane: one broadcast load each)
	const uint32_t nvert = (uint32_t)__builtin_amdgcn_readfirstlane((int)J.nvert);
	const bool para = __builtin_amdgcn_readfirstlane((int)J.parallelogram) != 0;
	WindowHand hand{1u, 0ull};
	if(!J.pad2[0]) bad |= delta_window_run(val, GraphLds{gw, ga}, nvert, para, base, &hand);
	if(hand.s < nvert) {
		if(lane_id() == 0) as_global(J.flags)[1] = 1;                          // (statistics: this blob's window handed over to the round loop)
		bad |= para ? delta_round_loop<true>(val, gw, ga, nvert, base, hand) : delta_round_loop<false>(val, gw, ga, nvert, base, hand);
	}
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	if(__ballot((bad >> 16) != 0)) return true;
	stage_out16<K>(val, as_global((int32_t *)J.values), J.nvert, base, J.deq == 1, J.q);
	return false;
}

// the same attribute with 32-bit records (the context met values beyond int16): window, rounds, copy-out; nothing can overflow
template <int K>
__device__ __forceinline__ void delta32_run(CRT_LDS uint8_t *rec, const DeltaJob &J, CRT_LDS const uint32_t *gw, const GaRef ga) {
	LdsW<K> val{(decltype(LdsW<K>::p))rec};
	const int32_t zero[LdsW<K>::NC] = {};
	const uint32_t nvert = (uint32_t)__builtin_amdgcn_readfirstlane((int)J.nvert);
	const bool para = __builtin_amdgcn_readfirstlane((int)J.parallelogram) != 0;
	WindowHand hand{1u, 0ull};
	if(!J.pad2[0]) (void)delta_window_run(val, GraphLds{gw, ga}, nvert, para, zero, &hand);
	if(hand.s < nvert) {
		if(lane_id() == 0) as_global(J.flags)[1] = 1;
		if(para) (void)delta_round_loop<true>(val, gw, ga, nvert, zero, hand); else (void)delta_round_loop<false>(val, gw, ga, nvert, zero, hand);
	}
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	stage_out32<K>(val, as_global((int32_t *)J.values), J.nvert, J.deq == 1, J.q);
}

} // namespace

// One workgroup per blob: up to four attributes, one wave each, share the prediction graph in LDS; the graph is made by the first wave
// that has no attribute (by wave 0 in front of its own staging when all four have one).
// LDS: records of attribute 0 | 1 | ... (each array a 16-byte multiple) | graph words (u32 x nvert) | a (u16 x nvert, unless it rides
// in a three-component attribute's records): delta16_group_lds() in kernels.h is the same sum.
__global__ __launch_bounds__(256) void k_delta_lds16(const DeltaJob *__restrict__ jobs, const DeltaGroup *__restrict__ groups, uint32_t ngroups) {
	if(blockIdx.x >= ngroups) return;
	const DeltaGroup G = groups[blockIdx.x];
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	const uint32_t lane = lane_id(), w = wave_id();
	const uint32_t nvert = (uint32_t)__builtin_amdgcn_readfirstlane((int)jobs[G.first].nvert);   // (uniform: the window loop's control stays scalar)
	const bool wide = __builtin_amdgcn_readfirstlane((int)jobs[G.first].pad[1]) != 0;            // 32-bit records (every job of the group says the same)
	CRT_LDS uint8_t *l8 = (CRT_LDS uint8_t *)as_lds(lds);
	uint32_t off = 0, myoff = 0, ga_addr = 0, ga_shift = 1;
	bool ga_set = false;
	for(uint32_t k = 0; k < G.count; k++) {
		const uint32_t N = jobs[G.first + k].N; const bool u8 = jobs[G.first + k].is_u8 != 0;
		if(k == w) myoff = off;
		if(!ga_set && !u8 && N == 3) { ga_addr = (uint32_t)(uintptr_t)(l8 + off) + (wide ? 12u : 6u); ga_shift = wide ? 4 : 3; ga_set = true; }
		off += delta_vbytes(nvert, N, u8, wide);
	}
	CRT_LDS uint32_t *gw = (CRT_LDS uint32_t *)(l8 + off);
	off += (4u*nvert + 15u) & ~15u;
	if(!ga_set) { ga_addr = (uint32_t)(uintptr_t)(l8 + off); off += (2u*nvert + 15u) & ~15u; }
	const GaRef ga{ga_addr, ga_shift};
	const uint32_t builder = G.count < 4 ? G.count : 0u;
	if(w == builder) {
		// prediction triples -> graph words + a.  Eight rounds of 64 vertices in flight (unconditional loads on clamped indices, pinned).
		CRT_GLOBAL const uint32_t *pred = as_global(jobs[G.first].pred);
		for(uint32_t base = 0; base < nvert; base += 512) {
			uint32_t ta[8], tb[8], tc[8];
#pragma unroll
			for(uint32_t u = 0; u < 8; u++) {
				const uint32_t i = base + u*64 + lane, ic = i < nvert ? i : nvert - 1u;
				const u32x3 t = *(CRT_GLOBAL const u32x3 *)(pred + (size_t)ic*3);
				ta[u] = t.x; tb[u] = t.y; tc[u] = t.z;
			}
#pragma unroll
			for(uint32_t u = 0; u < 8; u++) asm volatile("" : "+v"(ta[u]), "+v"(tb[u]), "+v"(tc[u]));
#pragma unroll
			for(uint32_t u = 0; u < 8; u++) {
				const uint32_t i = base + u*64 + lane;
				if(i >= nvert) continue;
				gw[i] = graph_word(i, ta[u], tb[u], tc[u]);
				ga.put(i, ta[u] < i ? ta[u] : 0u);
			}
		}
	}
	bool redo = false;
	const DeltaJob &J = jobs[G.first + (w < G.count ? w : 0u)];
	const bool mine = w < G.count, bytes = J.is_u8 != 0;
	CRT_LDS uint8_t *rec = l8 + myoff;
	const uint32_t N = J.N;
	uint32_t bad = 0;
	if(mine) {
		const bool ga_here = ga_set && ga_shift == 3 && (uint32_t)(uintptr_t)rec + 6u == ga_addr;
		if(bytes) stage_in_bytes(LdsVal<5>{(CRT_LDS uint32_t *)rec}, as_global((const uint8_t *)J.values), nvert, N);
		else if(wide) {
			CRT_GLOBAL const int32_t *src = as_global((const int32_t *)J.values);
			if(N == 1) stage_in32<1>(LdsW<1>{(CRT_LDS uint32_t *)rec}, src, nvert);
			else if(N == 2) stage_in32<2>(LdsW<2>{(CRT_LDS u32x2 *)rec}, src, nvert);
			else if(N == 3) stage_in32<3>(LdsW<3>{(CRT_LDS u32x4 *)rec}, src, nvert);
			else stage_in32<4>(LdsW<4>{(CRT_LDS u32x4 *)rec}, src, nvert);
		}
		else if(N == 1) bad = delta16_in<1>(rec, J, false);
		else if(N == 2) bad = delta16_in<2>(rec, J, false);
		else if(N == 3) bad = delta16_in<3>(rec, J, ga_here);
		else bad = delta16_in<4>(rec, J, false);
	}
	__syncthreads();                                                           // the graph is there
	if(!mine) return;
	if(bytes) {
		const LdsVal<5> val{(CRT_LDS uint32_t *)rec};
		WindowHand hand{1u, 0ull};
		if(J.parallelogram) {
			const int32_t zero[4] = {0, 0, 0, 0};
			if(!J.pad2[0]) (void)delta_window_loop<true>(val, GraphLds{gw, ga}, nvert, zero, &hand);
			if(hand.s < nvert) {
				if(lane == 0) as_global(J.flags)[1] = 1;
				(void)delta_round_loop<true>(val, gw, ga, nvert, zero, hand);
			}
		} else {                                                               // additions only: two packed registers instead of four components
			const LdsVal<6> val2{(CRT_LDS uint32_t *)rec};
			const int32_t zero[2] = {0, 0};
			if(!J.pad2[0]) (void)delta_window_loop<false>(val2, GraphLds{gw, ga}, nvert, zero, &hand);
			if(hand.s < nvert) {                                                   // (the round loop multiplies: four byte components, not two packed registers)
				if(lane == 0) as_global(J.flags)[1] = 1;
				const int32_t zero4[4] = {0, 0, 0, 0};
				(void)delta_round_loop<false>(val, gw, ga, nvert, zero4, hand);
			}
		}
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		stage_out_bytes(val, J, J.qc[0], J.qc[1], J.qc[2], J.qc[3]);
	}
	else if(wide) {
		if(N == 1) delta32_run<1>(rec, J, gw, ga);
		else if(N == 2) delta32_run<2>(rec, J, gw, ga);
		else if(N == 3) delta32_run<3>(rec, J, gw, ga);
		else delta32_run<4>(rec, J, gw, ga);
	}
	else if(N == 1) redo = delta16_run<1>(rec, J, gw, ga, bad);
	else if(N == 2) redo = delta16_run<2>(rec, J, gw, ga, bad);
	else if(N == 3) redo = delta16_run<3>(rec, J, gw, ga, bad);
	else redo = delta16_run<4>(rec, J, gw, ga, bad);
	if(redo) {
		// the relative values left int16: the raw deltas are still in HBM (nothing was written back) - the same loop on them, 32 bits wide,
		// and a word for the host: its next batches are planned on the wide kernel (batch.cpp: harvest)
		if(lane == 0) *as_global(J.flags) = 1;
		if(J.pad[0]) {
			// (the raw deltas came as halfwords at the front of the buffer: widened in place from the top down - a chunk's 64 values are read before
			// they are written, and what a chunk writes lies above everything still unread)
			CRT_GLOBAL int32_t *v32 = as_global((int32_t *)J.values);
			CRT_GLOBAL const int16_t *v16 = (CRT_GLOBAL const int16_t *)v32;
			const uint32_t n = nvert*N;
			for(uint32_t top = (n + 63u) & ~63u; top > 0; top -= 64) {
				const uint32_t k = top - 64 + lane;
				const int32_t x = k < n ? (int32_t)v16[k] : 0;
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				if(k < n) __hip_atomic_store(v32 + k, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);       // (as GlobalVal's own stores: the loop below reads them back)
			}
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		}
		const int32_t zero[4] = {0, 0, 0, 0};
		const GlobalVal<int32_t> gval{as_global((int32_t *)J.values), N};
		(void)delta_window_run(gval, GraphLds{gw, ga}, nvert, J.parallelogram != 0, zero);
		if(J.deq == 1) {                                                        // floats in place of the integers (vertex_attribute.h:190-193)
			CRT_GLOBAL int32_t *v = as_global((int32_t *)J.values);
			const uint32_t n = nvert*N;
			for(uint32_t k = lane; k < n; k += 64) { const int32_t x = v[k]; ((CRT_GLOBAL float *)v)[k] = (float)x*J.q; }
		}
	}
}


// ------------------------------------------------------------------------------------------------------------------------------------------------
// k_delta_tiles (round 6) - meshes too big for the LDS records above (tens of thousands to millions of vertices; rounds 1-5: k_delta_mesh, a walk
// along the stretches of the prediction graph through L2, three round trips a vertex step: 1.5 of config C2's 4.0 ms).
//
// One workgroup of SIXTEEN waves per (blob, attribute), walking the vertex sequence in TILES of 1 024: thread t has vertex s + t.  What a vertex
// reads is final data except for the parents inside its own tile, so a tile is the window loop above with a 1 024-wide window that does not slide:
//   * the final values of the last DT_RING vertices live in an LDS RING (32-bit components, one array a component: v[x] at ring[q][x mod DT_RING]):
//     a parent is a gather from LDS, not an L2 round trip; one further back than the ring (a seam, an irregular mesh) is read from HBM when the tile
//     begins - it was stored tiles ago;
//   * PASSES inside the tile: a vertex goes when its in-tile parents have gone (a 1 024-bit mask in LDS) - or, if it continues its predecessor
//     (a = i - 1), when that one goes in the same pass: the window loop's flood fill per wave, with the carry handed from wave to wave (each wave
//     publishes whether its lane 63 goes without / with a carry-in, every wave folds the sixteen answers); the vertices that go form runs, each a
//     prefix sum from its head: a wave scan per component, the sum of a run that started in an earlier wave handed on the same way;
//   * a grid finishes a tile in two or three passes (its parents b, c lie a ring of the traversal back: what does not fit the first pass is the
//     part of the tile that predicts from the tile's own beginning); the lowest vertex that has not gone always can, so any graph terminates;
//   * the next tile's triples and raw values are fetched while this one computes; the finished tile goes back to HBM in one coalesced store.
// Exact as everything else here: sums in 32-bit registers mod 2^32 (bytes: stored mod 256).  A malformed triple (a parent that is not an earlier
// vertex; vertex 0) leaves the value as it is, as k_delta_mesh does.  vertex_attribute.h:160-176.
constexpr uint32_t DT_RING = 4096;
template <int NC, typename T>
__device__ __forceinline__ void delta_tiles_body(const DeltaJob &J, CRT_LDS uint32_t *ring, CRT_LDS uint64_t *fm, CRT_LDS uint32_t *gpub, CRT_LDS uint32_t *loc63) {
	constexpr uint32_t NW = DELTA_THREADS/64;
	const uint32_t nvert = J.nvert, t = threadIdx.x, lane = lane_id(), w = wave_id();
	const bool para = J.parallelogram != 0;
	CRT_GLOBAL const uint32_t *pred = as_global(J.pred);
	CRT_GLOBAL T *vals = as_global((T *)J.values);
	CRT_LDS const uint32_t *fm32 = (CRT_LDS const uint32_t *)fm;
	const uint64_t lane_le = (2ull << lane) - 1ull;
	// The automaton may still be running (a lone context decodes the attribute streams beside it and launches this kernel on that second stream):
	// it publishes how many vertices have their prediction triple - at every slide of its symbol window, 0xFFFFFFFF when it is done - and a tile
	// waits for the triples it is about to load.  One thread polls; a wait that outlasts any decode (2 s) is given up rather than hung on.
	CRT_GLOBAL const uint32_t *progress = as_global((const uint32_t *)J.fired);
	uint32_t seen = progress ? 0u : 0xFFFFFFFFu;                               // what the word said when it was last looked at (uniform): polled again only when a tile needs more
	auto wait_for = [&](uint32_t need) {
		if(seen >= need) return;
		if(t == 0) {
			const uint64_t t0 = wall_clock64();
			uint32_t p;
			while((p = __hip_atomic_load(progress, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) < need) {
				__builtin_amdgcn_s_sleep(32);
				if(wall_clock64() - t0 > 200000000ull) {                           // (cannot happen: the automaton is enqueued first and waits for nothing of this kernel;
					as_global(J.flags)[-(int32_t)J.pad2[1]] = -9;                    //  if it ever does, the blob says CRTHIP_E_DEVICE instead of carrying wrong values)
					p = 0xFFFFFFFFu; break;
				}
			}
			gpub[DELTA_THREADS/64] = p;
		}
		lds_barrier();
		seen = gpub[DELTA_THREADS/64];
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
		lds_barrier();                                                         // (the word is read: the next poll may overwrite it)
	};
#ifdef DT_DEBUG
	if(t == 0) printf("tiles job N %u nvert %u progress %p first %u\n", J.N, nvert, (const void *)J.fired, J.fired ? *(const uint32_t *)J.fired : 7u);
#endif
	typedef uint32_t u32x3_t __attribute__((ext_vector_type(3)));
	auto load_pred = [&](uint32_t j) -> u32x3_t { return *(CRT_GLOBAL const u32x3_t *)(pred + (size_t)(j < nvert ? j : nvert - 1u)*3); };
	wait_for(nvert < DELTA_THREADS ? nvert : DELTA_THREADS);
	u32x3_t tri = load_pred(t);
	uint32_t d[NC];
#pragma unroll
	for(int q = 0; q < NC; q++) d[q] = t < nvert ? (uint32_t)vals[(size_t)t*NC + q] : 0u;
	for(uint32_t s = 0; s < nvert; s += DELTA_THREADS) {
		const uint32_t i = s + t;
		const bool in = i < nvert;
		const uint32_t a = tri.x, b = para ? tri.y : tri.x, c = para ? tri.z : tri.x;
		uint32_t dv[NC];
#pragma unroll
		for(int q = 0; q < NC; q++) dv[q] = d[q];
		const bool stays = !(a < i && b < i && c < i);                          // (vertex 0; a malformed triple): the value stays
		const bool a_in = !stays && a >= s, b_in = para && !stays && b >= s, c_in = para && !stays && c >= s;
		const bool ch = a_in && a + 1u == i;                                    // continues its predecessor's sum (t > 0: the predecessor is in the tile)
		const uint32_t ra = a - s, rb = b - s, rc = c - s;
		// parents behind the ring: from HBM, now (stored at least three tiles ago)
		const uint32_t horizon = s + DELTA_THREADS > DT_RING ? s + DELTA_THREADS - DT_RING : 0u;
		const bool a_far = !stays && a < horizon, b_far = para && !stays && b < horizon, c_far = para && !stays && c < horizon;
		// (agent scope: from L2, where the stores of three and more tiles ago have arrived - every wave has since waited for loads it issued
		// behind them, and this CU's L1 may still hold the raw values those addresses had when they were prefetched.)  Issued BEFORE the next
		// tile's prefetch: the memory counter retires in order, and waiting for these must not mean waiting for that
		uint32_t fa[NC], fb[NC], fc[NC];
#pragma unroll
		for(int q = 0; q < NC; q++) {
			fa[q] = a_far ? (uint32_t)__hip_atomic_load(vals + (size_t)a*NC + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
			fb[q] = b_far ? (uint32_t)__hip_atomic_load(vals + (size_t)b*NC + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
			fc[q] = c_far ? (uint32_t)__hip_atomic_load(vals + (size_t)c*NC + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
		}
		asm volatile("" ::: "memory");
		// the next tile's triple and raw values, in flight while this one computes
		if(s + DELTA_THREADS < nvert) {
			wait_for(s + 2*DELTA_THREADS < nvert ? s + 2*DELTA_THREADS : nvert);
			const uint32_t j = i + DELTA_THREADS;
			tri = load_pred(j);
#pragma unroll
			for(int q = 0; q < NC; q++) d[q] = j < nvert ? (uint32_t)vals[(size_t)j*NC + q] : 0u;
		}
		bool fired = !in;
#ifdef DT_DEBUG
		uint32_t npass = 0; const uint64_t dt0 = __builtin_amdgcn_s_memtime();
#endif
		{ const uint64_t m = __ballot(!in); if(lane == 0) fm[w] = m; }
		uint32_t fin[NC];
#pragma unroll
		for(int q = 0; q < NC; q++) fin[q] = dv[q];
		lds_barrier();                                                         // (lgkmcnt only: the next tile's prefetch stays in flight through every barrier of this tile)
		for(;;) {
			// ---- who can go: without, and with, the vertex in front of this wave going too ----
			const uint64_t left = __ballot(!fired);
			uint64_t Sm = 0, G0 = 0, G1 = 0;
			bool H = false;
			if(left) {
				auto gone = [&](uint32_t x) -> bool { return (fm32[x >> 5] >> (x & 31u)) & 1u; };
				const bool af = !a_in || gone(ra), bf = !b_in || gone(rb), cf = !c_in || gone(rc);
				const bool R = !fired && (stays || (bf && cf && (ch || af)));
				H = stays || !ch || af;                                          // starts a sum of its own
				const uint64_t Rm = __ballot(R), Cm = __ballot(R && !H && lane == 0);
				Sm = __ballot(R && H);
				const uint64_t S1 = Sm | Cm;
				G0 = (((Rm + Sm) ^ Rm) & Rm) | Sm; G1 = (((Rm + S1) ^ Rm) & Rm) | S1;
			}
			// bit 0 / 1: lane 63 goes without / with the vertex in front of this wave going; bit 3: vertices of this wave are left
			if(lane == 0) gpub[w] = (uint32_t)(G0 >> 63) | (uint32_t)(G1 >> 63) << 1 | (left ? 8u : 0u);
			lds_barrier();
			// sixteen answers read by sixteen lanes, folded on the scalar unit
			const uint32_t gp = lane < NW ? gpub[lane] : 0u;
			const uint32_t g0m = (uint32_t)__ballot(gp & 1u), g1m = (uint32_t)__ballot(gp & 2u);
			const bool more = __ballot(gp & 8u) != 0;
			if(!more) break;                                                    // every vertex of the tile has gone (uniform over the workgroup)
			uint32_t cinm = 0, cin = 0;                                          // cinm bit k: the vertex in front of wave k goes in this pass
			for(uint32_t k = 0; k < NW; k++) { cinm |= cin << k; cin = ((cin ? g1m : g0m) >> k) & 1u; }
			cin = (cinm >> w) & 1u;
			const uint64_t G = cin ? G1 : G0;
			const bool go = (G >> lane) & 1ull;
			const uint64_t hm = Sm & lane_le;                                    // the heads at or below me (none: my run comes in from the wave before)
			uint32_t r[NC];
#pragma unroll
			for(int q = 0; q < NC; q++) r[q] = 0;
			if(G) {
				const bool use = go && !stays, head = use && H;
				uint32_t x[NC];
#pragma unroll
				for(int q = 0; q < NC; q++) {
					uint32_t v = dv[q];
					if(para) {
						const uint32_t vb = b_far ? fb[q] : ring[q*DT_RING + ((use ? b : 0u) & (DT_RING - 1u))];
						const uint32_t vc = c_far ? fc[q] : ring[q*DT_RING + ((use ? c : 0u) & (DT_RING - 1u))];
						v += use ? vb - vc : 0u;
					}
					const uint32_t va = a_far ? fa[q] : ring[q*DT_RING + ((head ? a : 0u) & (DT_RING - 1u))];
					v += head ? va : 0u;
					x[q] = go ? v : 0u;
				}
				const uint32_t hidx = hm ? 63u - (uint32_t)__builtin_clzll(hm) : 0u;
#pragma unroll
				for(int q = 0; q < NC; q++) {
					const uint32_t incl = wave_inclusive_scan_u32(x[q]);
					const uint32_t eh = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(hidx << 2), (int)(incl - x[q]));
					r[q] = incl - (hm ? eh : 0u);
				}
			}
			// does the run that came in reach this wave's lane 63 (then the next wave's carried sum goes on through this one)?  Only waves whose successor is
			// carried into publish; the others' words are not read
			const bool pass_on = ((cinm >> (w + 1 < NW ? w + 1 : w)) & 1u) && w + 1 < NW;
			if(pass_on && lane == 63) {
				gpub[NW + 1 + w] = go && !hm ? 1u : 0u;
#pragma unroll
				for(int q = 0; q < NC; q++) loc63[w*4 + q] = r[q];
			}
			lds_barrier();
			if(G && cin && (G & 1ull) && !(Sm & 1ull)) {
				// the sum my first run continues = the final value of the vertex in front of this wave = the lane-63 sums of the waves behind me back to the
				// first whose lane 63 started afresh
				const uint32_t o = lane < w && ((cinm >> (lane + 1)) & 1u) ? gpub[NW + 1 + lane] : 0u;
				const uint64_t closed = __ballot(lane < w && !o) & ((1ull << w) - 1ull);
				const uint32_t from = closed ? 63u - (uint32_t)__builtin_clzll(closed) : 0u;
#pragma unroll
				for(int q = 0; q < NC; q++) {
					const uint32_t v = lane < w && lane >= from ? loc63[lane*4 + q] : 0u;
					const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan_u32(v), 63);
					r[q] += go && !hm ? tot : 0u;
				}
			}
			if(go) {
#pragma unroll
				for(int q = 0; q < NC; q++) { ring[q*DT_RING + (i & (DT_RING - 1u))] = r[q]; fin[q] = r[q]; }
				fired = true;
			}
			if(lane == 0 && G) fm[w] |= G;
			lds_barrier();
#ifdef DT_DEBUG
			npass++;
#endif
		}
#ifdef DT_DEBUG
		if(t == 0 && NC == 3 && (s < 4096 || (s & 16383) == 0)) printf("N %u tile %u passes %u clocks %u\n", J.N, s, npass, (uint32_t)(__builtin_amdgcn_s_memtime() - dt0));
#endif
		if(in) {
#pragma unroll
			for(int q = 0; q < NC; q++) vals[(size_t)i*NC + q] = (T)fin[q];
		}
	}
}

__global__ __launch_bounds__(DELTA_THREADS) void k_delta_tiles(const DeltaJob *__restrict__ jobs, uint32_t njobs) {
	__shared__ uint32_t ring[4*DT_RING];                                       // 64 KB: the last DT_RING vertices' final values, a component an array
	__shared__ uint64_t fm[DELTA_THREADS/64];                                   // the tile's vertices that have gone
	__shared__ uint32_t gpub[2*(DELTA_THREADS/64) + 1], loc63[4*DELTA_THREADS/64];  // (gpub[NW]: the progress word as thread 0 last read it; behind it: the waves' open flags)
	if(blockIdx.x >= njobs) return;
	const DeltaJob J = jobs[blockIdx.x];
	if(J.N < 1 || J.N > 4) return;                                              // (more components: k_delta_mesh, launched for those alone)
	if(J.is_u8) {
		switch(J.N) {
		case 1: delta_tiles_body<1, uint8_t>(J, (CRT_LDS uint32_t *)ring, (CRT_LDS uint64_t *)fm, (CRT_LDS uint32_t *)gpub, (CRT_LDS uint32_t *)loc63); break;
		case 2: delta_tiles_body<2, uint8_t>(J, (CRT_LDS uint32_t *)ring, (CRT_LDS uint64_t *)fm, (CRT_LDS uint32_t *)gpub, (CRT_LDS uint32_t *)loc63); break;
		case 3: delta_tiles_body<3, uint8_t>(J, (CRT_LDS uint32_t *)ring, (CRT_LDS uint64_t *)fm, (CRT_LDS uint32_t *)gpub, (CRT_LDS uint32_t *)loc63); break;
		default: delta_tiles_body<4, uint8_t>(J, (CRT_LDS uint32_t *)ring, (CRT_LDS uint64_t *)fm, (CRT_LDS uint32_t *)gpub, (CRT_LDS uint32_t *)loc63); break;
		}
	} else {
		switch(J.N) {
		case 1: delta_tiles_body<1, uint32_t>(J, (CRT_LDS uint32_t *)ring, (CRT_LDS uint64_t *)fm, (CRT_LDS uint32_t *)gpub, (CRT_LDS uint32_t *)loc63); break;
		case 2: delta_tiles_body<2, uint32_t>(J, (CRT_LDS uint32_t *)ring, (CRT_LDS uint64_t *)fm, (CRT_LDS uint32_t *)gpub, (CRT_LDS uint32_t *)loc63); break;
		case 3: delta_tiles_body<3, uint32_t>(J, (CRT_LDS uint32_t *)ring, (CRT_LDS uint64_t *)fm, (CRT_LDS uint32_t *)gpub, (CRT_LDS uint32_t *)loc63); break;
		default: delta_tiles_body<4, uint32_t>(J, (CRT_LDS uint32_t *)ring, (CRT_LDS uint64_t *)fm, (CRT_LDS uint32_t *)gpub, (CRT_LDS uint32_t *)loc63); break;
		}
	}
}

} // namespace corto_hip
