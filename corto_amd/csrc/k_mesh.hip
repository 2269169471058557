// k_mesh.hip — the two per-blob SERIAL stages of a mesh decode on gfx950:
//   K-TOPO  CLERS automaton -> triangle list + per-vertex prediction triples   src/decoder.cpp:204-358
//   K-DELTA parallelogram / first-neighbour delta inversion                     include/corto/vertex_attribute.h:160-176,
//                                                                              src/normal_attribute.cpp:193-201
// Both are chains of dependent reads per blob: the parallel unit is the blob.  One
// wave per blob; many blobs per launch.
#include "kernels_common.h"

namespace corto_hip {

enum : uint32_t { C_VERTEX = 0, C_LEFT = 1, C_RIGHT = 2, C_END = 3, C_BOUNDARY = 4, C_DELAY = 5, C_SPLIT = 6 };
constexpr int32_t ERR_TOPOLOGY = -5;   // CRTHIP_E_TOPOLOGY

// ------------------------------------------------------------------------------------------------
// K-TOPO, general path: front / queues in global scratch sized by the stream's max_front.
struct TopoState {
	const TopoJob &J;
	uint32_t cler, vertex_count;
	uint64_t bit;
	int32_t err;
	__device__ uint32_t bits(uint32_t n) {
		if(bit + n > (uint64_t)J.split_nwords*32) { err = ERR_TOPOLOGY; return 0; }
		const uint32_t v = bit_field(J.split_words, J.split_nwords, bit, n);
		bit += n;
		return v;
	}
	__device__ void face(uint32_t at, uint32_t a, uint32_t b, uint32_t c) {
		if(J.faces_u16) { uint16_t *f = (uint16_t *)J.faces + at; f[0] = (uint16_t)a; f[1] = (uint16_t)b; f[2] = (uint16_t)c; }
		else { uint32_t *f = (uint32_t *)J.faces + at; f[0] = a; f[1] = b; f[2] = c; }
	}
};

__device__ void topo_group(TopoState &S, uint32_t start, uint32_t end) {
	const TopoJob &J = S.J;
	uint4 *__restrict__ fa = J.front_a;
	uint2 *__restrict__ fb = J.front_b;
	uint32_t *__restrict__ order = J.order;
	uint32_t *__restrict__ delayed = J.delayed;
	const uint32_t cap = J.front_cap;
	uint32_t nfront = 0, norder = 0, iorder = 0, ndelayed = 0;
	int64_t new_edge = -1;
	const uint32_t splitbits = 32 - __clz(J.nvert | 1u);      // ilog2(nvert) + 1 (src/cstream.cpp:31-35; nvert >= 1 here)

#define FAIL() do { S.err = ERR_TOPOLOGY; return; } while(0)
	while(start < end) {
		if(new_edge == -1 && iorder >= norder && ndelayed == 0) {      // seed face (decoder.cpp:224-259)
			if(S.cler >= J.nclers) FAIL();
			uint32_t last = S.vertex_count - 1, vi[3], split = 0;
			const uint32_t c = J.clers[S.cler++];
			if(c == C_SPLIT) split = S.bits(3);
			for(int k = 0; k < 3; k++) {
				uint32_t v;
				if(split & (1u << k)) v = S.bits(splitbits);
				else {
					if(S.vertex_count >= J.nvert) FAIL();
					uint32_t *p = J.pred + (size_t)S.vertex_count*3;
					p[0] = last; p[1] = last; p[2] = last;
					last = v = S.vertex_count++;
				}
				vi[k] = v;
			}
			if(S.err) return;
			S.face(start, vi[0], vi[1], vi[2]); start += 3;
			const uint32_t e = nfront;
			if(e + 3 > cap) FAIL();
			order[norder++] = e; order[norder++] = e + 1; order[norder++] = e + 2;
			fa[e] = make_uint4(vi[1], vi[2], vi[0], 0);     fb[e] = make_uint2(e + 2, e + 1);
			fa[e + 1] = make_uint4(vi[2], vi[0], vi[1], 0); fb[e + 1] = make_uint2(e, e + 2);
			fa[e + 2] = make_uint4(vi[0], vi[1], vi[2], 0); fb[e + 2] = make_uint2(e + 1, e);
			nfront += 3;
			continue;
		}
		uint32_t f;
		if(new_edge != -1) { f = (uint32_t)new_edge; new_edge = -1; }
		else if(iorder < norder) f = order[iorder++];
		else f = delayed[--ndelayed];
		if(f >= nfront) FAIL();
		const uint4 ea = fa[f];
		if(ea.w) continue;                                             // deleted: no symbol consumed (decoder.cpp:278-279)
		if(S.cler >= J.nclers) FAIL();
		const uint32_t c = J.clers[S.cler++];
		if(c == C_BOUNDARY) continue;
		const uint2 eb = fb[f];
		const uint32_t v0 = ea.x, v1 = ea.y, ep = eb.x, en = eb.y;
		if(ep >= nfront || en >= nfront) FAIL();
		const uint32_t ne = nfront;
		uint32_t opp;
		new_edge = ne;
		if(c == C_VERTEX || c == C_SPLIT) {                            // decoder.cpp:294-309
			if(c == C_SPLIT) { opp = S.bits(splitbits); if(S.err) return; }
			else {
				if(S.vertex_count >= J.nvert) FAIL();
				uint32_t *p = J.pred + (size_t)S.vertex_count*3;
				p[0] = v1; p[1] = v0; p[2] = ea.z;
				opp = S.vertex_count++;
			}
			if(ne + 2 > cap) FAIL();
			fb[ep].y = ne;
			fb[en].x = ne + 1;
			fa[ne] = make_uint4(v0, opp, v1, 0);     fb[ne] = make_uint2(ep, ne + 1);
			order[norder++] = ne + 1;
			fa[ne + 1] = make_uint4(opp, v1, v0, 0); fb[ne + 1] = make_uint2(ne, en);
			nfront += 2;
		} else if(c == C_LEFT) {                                       // decoder.cpp:311-317
			const uint32_t pp = fb[ep].x;
			if(pp >= nfront || ne + 1 > cap) FAIL();
			opp = fa[ep].x;
			fa[ep].w = 1;
			fb[pp].y = ne;
			fb[en].x = ne;
			fa[ne] = make_uint4(opp, v1, v0, 0); fb[ne] = make_uint2(pp, en);
			nfront += 1;
		} else if(c == C_RIGHT) {                                      // decoder.cpp:319-325
			const uint32_t nn = fb[en].y;
			if(nn >= nfront || ne + 1 > cap) FAIL();
			opp = fa[en].y;
			fa[en].w = 1;
			fb[nn].x = ne;
			fb[ep].y = ne;
			fa[ne] = make_uint4(v0, opp, v1, 0); fb[ne] = make_uint2(ep, nn);
			nfront += 1;
		} else if(c == C_DELAY) {                                      // decoder.cpp:327-331
			if(ndelayed >= cap) FAIL();
			delayed[ndelayed++] = f;
			new_edge = -1;
			continue;
		} else if(c == C_END) {                                        // decoder.cpp:333-339
			const uint32_t pp = fb[ep].x, nn = fb[en].y;
			if(pp >= nfront || nn >= nfront) FAIL();
			opp = fa[ep].x;
			fa[ep].w = 1; fa[en].w = 1;
			fb[pp].y = nn;
			fb[nn].x = pp;
			new_edge = -1;
		} else FAIL();
		S.face(start, v1, v0, opp); start += 3;                        // decoder.cpp:348-356
	}
#undef FAIL
}

__global__ __launch_bounds__(64) void k_topology(const TopoJob *__restrict__ jobs, uint32_t njobs) {
	if(blockIdx.x >= njobs || threadIdx.x != 0) return;
	const TopoJob J = jobs[blockIdx.x];
	TopoState S{J, 0, 0, 0, 0};
	uint32_t start = 0;
	for(uint32_t g = 0; g < J.ngroups && !S.err; g++) {                // decoder.cpp:173-178
		uint32_t ge = J.group_end[g];
		if(ge > J.nface || ge < start) { S.err = ERR_TOPOLOGY; break; }
		topo_group(S, start*3, ge*3);
		start = ge;
	}
	if(S.err) *J.status = S.err;
}

// ------------------------------------------------------------------------------------------------
// K-DELTA (mesh).  One wave per (blob, attribute); lane = component.  v[i] += v[a] + v[b] - v[c]
// (or += v[a]) for i = 1..nvert-1 in index order.  When the attribute fits the LDS budget the whole
// array is staged there (dependent-read latency ~1/4 of an L2 round trip); otherwise in place in HBM.
// Prediction triples are fetched 64 vertices at a time (one per lane) and broadcast with readlane.
template <typename T>
__device__ void delta_chain(T *v, const uint32_t *__restrict__ pred, uint32_t nvert, uint32_t N, bool para) {
	const uint32_t lane = lane_id();
	for(uint32_t c0 = 0; c0 < N; c0 += 64) {
		const uint32_t comp = c0 + lane;
		const bool on = comp < N;
		for(uint32_t i0 = 0; i0 < nvert; i0 += 64) {
			const uint32_t mine = i0 + lane;
			uint32_t pa = 0, pb = 0, pc = 0;
			if(mine < nvert) { pa = pred[(size_t)mine*3]; pb = pred[(size_t)mine*3 + 1]; pc = pred[(size_t)mine*3 + 2]; }
			const uint32_t kend = min(64u, nvert - i0);
			for(uint32_t k = (i0 == 0 ? 1u : 0u); k < kend; k++) {
				const uint32_t a = __shfl(pa, k, 64), b = __shfl(pb, k, 64), c = __shfl(pc, k, 64);
				const uint32_t i = i0 + k;
				if(on && a < nvert && b < nvert && c < nvert) {
					if(para) v[(size_t)i*N + comp] = (T)(v[(size_t)i*N + comp] + v[(size_t)a*N + comp] + v[(size_t)b*N + comp] - v[(size_t)c*N + comp]);
					else v[(size_t)i*N + comp] = (T)(v[(size_t)i*N + comp] + v[(size_t)a*N + comp]);
				}
			}
		}
	}
}

__global__ __launch_bounds__(64) void k_delta_mesh(const DeltaJob *__restrict__ jobs, uint32_t njobs, uint32_t lds_bytes) {
	if(blockIdx.x >= njobs) return;
	const DeltaJob J = jobs[blockIdx.x];
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	const size_t bytes = (size_t)J.nvert*J.N*(J.is_u8 ? 1 : 4);
	const uint32_t lane = lane_id();
	if(bytes <= lds_bytes && (bytes & 3) == 0 && (((uintptr_t)J.values) & 3) == 0) {
		uint32_t *l32 = (uint32_t *)lds;
		uint32_t *g32 = (uint32_t *)J.values;
		const uint32_t ndw = (uint32_t)(bytes >> 2);
		for(uint32_t i = lane; i < ndw; i += 64) l32[i] = g32[i];
		__syncthreads();
		if(J.is_u8) delta_chain<uint8_t>((uint8_t *)lds, J.pred, J.nvert, J.N, J.parallelogram);
		else delta_chain<uint32_t>((uint32_t *)lds, J.pred, J.nvert, J.N, J.parallelogram);
		__syncthreads();
		for(uint32_t i = lane; i < ndw; i += 64) g32[i] = l32[i];
	} else {
		if(J.is_u8) delta_chain<uint8_t>((uint8_t *)J.values, J.pred, J.nvert, J.N, J.parallelogram);
		else delta_chain<uint32_t>((uint32_t *)J.values, J.pred, J.nvert, J.N, J.parallelogram);
	}
}

} // namespace corto_hip
