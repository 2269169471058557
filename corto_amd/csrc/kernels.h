// kernels.h — declarations of the HIP kernels (k_tunstall.hip, k_stream.hip, k_mesh.hip, k_normal.hip)
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>

#include "device_plan.h"

namespace corto_hip {

// k_tunstall.hip
__global__ void k_tun_tables(const TunStream *streams, uint32_t nstreams, TunTable *tables);
__global__ void k_tun_stream(const TunStream *streams, uint32_t nstreams);   // short streams: dictionary + decode by one wave, the dictionary never leaves LDS
__global__ void k_tun_stream_grouped(const TunStream *streams, const uint32_t *ids, const TunGroup *groups, uint32_t ngroups, const TunTable *tables);   // ... streams of ONE dictionary, up to TUN_GROUP_MAX per workgroup, from one copy of it in LDS
__global__ void k_tun_stream_scan(const TunStream *streams, uint32_t nstreams, uint64_t *chunk_out);   // one stream's quarter sums -> its quarter offsets (one workgroup per stream)
__global__ void k_tun_chunk_sums(const TunStream *streams, const uint32_t *chunk_stream, uint32_t nchunks, const TunTable *tables,
                                 uint64_t *chunk_out, uint32_t chunk_base);
__global__ void k_tun_decode(const TunStream *streams, const uint32_t *chunk_stream, uint32_t nchunks, const TunTable *tables,
                             const uint64_t *chunk_out, uint32_t chunk_base);
// Dwords of a word's zero-padded copy the staged decode composes from (k_tunstall.hip): by the dictionary's longest word;
// steps of fewer than 8 codewords per lane (mean word length > 8) are only compiled for 4.
__host__ __device__ inline uint32_t tun_width(uint32_t cpl, uint32_t maxlen) { return cpl < 8 || maxlen > 8 ? 4u : maxlen > 4 ? 2u : 1u; }
int launch_tun_decode_staged(hipStream_t stream, const TunStream *streams, const uint32_t *chunk_stream, uint32_t nchunks, const TunTable *tables,
                             uint64_t *chunk_out, uint32_t sums_only);    // words <= 4 bytes, <= 8 bytes, longer: three bodies of one kernel
// blocks of a launch whose kernel takes its job by xcd_slot() (kernels_common.h: every XCD a contiguous eighth of the jobs)
inline uint32_t xcd_grid(uint32_t n) { return 8u*((n + 7u) >> 3); }
__global__ void k_fill(const FillJob *jobs, uint32_t njobs);
__global__ void k_fill_block(uint8_t *dst, uint64_t bytes, uint32_t value);

// k_stream.hip
__global__ void k_scan_u64(uint64_t *a, uint32_t n);
__global__ void k_u32_chunk_sums(const uint32_t *a, uint32_t n, uint64_t *partial);
__global__ void k_u32_chunk_apply(const uint32_t *a, uint32_t *out, uint32_t n, const uint64_t *partial);
__global__ void k_unpack_extract(const UnpackJob *jobs, const uint32_t *chunk_job, uint32_t nchunks, uint64_t *state);   // state: nchunks + 1 zeroed words (look-back)
__global__ void k_unpack_wave(const UnpackJob *jobs, const uint32_t *job_ids, uint32_t njobs);   // ... the log streams of small bit blocks: one wave per stream, lanes interleaved, no look-back
__global__ void k_cloud_sums(const CloudJob *jobs, const uint32_t *chunk_job, uint32_t nchunks, uint64_t *partial);
__global__ void k_cloud_apply(const CloudJob *jobs, const uint32_t *chunk_job, uint32_t nchunks, const uint64_t *partial);
__global__ void k_dequant(const DequantJob *jobs, const uint32_t *block_job, uint32_t nblocks);

// k_mesh.hip
__global__ void k_topology(const TopoJob *jobs, const uint32_t *job_ids, uint32_t njobs);
__global__ void k_topology_lds(const TopoJob *jobs, const uint32_t *job_ids, uint32_t njobs);
__global__ void k_topology_lds_big(const TopoJob *jobs, const uint32_t *job_ids, uint32_t njobs);   // the same automaton keeping a progress word (k_mesh.hip)
// dynamic LDS bytes k_topology_lds needs for a front of `cap` edges and `nclers` symbols
constexpr uint32_t TOPO_SPLIT_LDS = 256;          // words of the split / vertex-id bit block staged in LDS
// LDS of one blob's CLERS automaton (k_mesh.hip): records of the LIVE front only - a ring for the queued edges, a pool for
// surviving chain ends - plus the DELAY stack and a window of nibble-packed symbols.
inline uint32_t topo_lds_bytes(uint32_t ring, uint32_t pool, uint32_t dcap, uint32_t symwin) {
	return (ring + pool)*16 + ((pool + 7) & ~7u)*2 + ((dcap + 7) & ~7u)*2 + symwin/2 + 8 + 32 + TOPO_SPLIT_LDS*4 + 16;
}
constexpr uint32_t TOPO_SYMWIN_MAX = 8192;
// The queue of a sphere-like mesh peaks near 3*sqrt(nface) (189 for 4 096 faces, 1 497 for 256 000): the RING gets `slots`*sqrt(nface) rounded up to a
// power of two (at least 256, at most `ring_max`) times what the context has learnt (`ring_scale`, a power of two: a torus' queue is ten times a sphere's).
// The POOL is sized on its own (round 5; rounds 2-4 had pool = ring): it keeps every edge that got a BOUNDARY for good and every DELAYed one while it
// waits, and how many boundary edges a mesh has is in the header - B = 2V - F - 2*chi for a manifold mesh (Euler: V - E + F = chi, 2E = 3F + B), 128 for
// the C4 unit, ~V for a ribbon or for confetti of one-face components.  That is a floor (the encoder also writes BOUNDARY where a front meets faces it
// has been to, and the DELAYed edges come on top: Delaunay discs with holes hold 2-2.5x as many), so the context learns a factor for it too (`pool_q8`,
// in eighths) from what redone blobs report.  What does not fit is redone on the HBM front.
// slots = 8 for a few big meshes (LDS is not contended: leave room), 4 for a batch of many blobs: a 4K-triangle blob then takes 12.5 KB,
// and that it is NOT MORE THAN THE 16 KB of a K-STREAM wave matters more than the size itself: the automaton's workgroups are dispatched
// while the attribute streams' 2 000 waves fill every CU ten to a CU, and a 16 KB hole opens whenever one of those ends - a 22 KB
// request waits for two neighbouring ones (0.237 -> 0.200 ms per C4 batch unpipelined, +5 % pipelined).
inline uint32_t topo_boundary_estimate(uint32_t nvert, uint32_t nface) { const uint64_t v2 = 2ull*nvert; return v2 <= nface ? 0u : v2 - nface > (1u << 20) ? (1u << 20) : (uint32_t)(v2 - nface); }
// `pool_cap` (0: none): the most pool slots any redone blob has reported (+ an eighth) - a launch's LDS request is its largest blob's, so the factor is
// not applied beyond what the neediest blob seen so far would have needed
inline void topo_lds_geometry(uint32_t nface, uint32_t nclers, uint32_t ring_max, uint32_t ring_scale, uint32_t pool_q8, uint32_t slots, uint32_t boundary,
                              uint32_t &ring, uint32_t &pool, uint32_t &symwin, uint32_t pool_cap = 0) {
	uint32_t want = 256;
	while((uint64_t)want*want < (uint64_t)slots*slots*nface && want < ring_max) want <<= 1;
	const uint32_t base = want;
	while(ring_scale > 1 && want < ring_max) { want <<= 1; ring_scale >>= 1; }
	ring = want;
	// round 6: the BOUNDARY edges share ONE slot (the sink, k_mesh.hip), so the pool holds the DELAYed edges while they wait + that slot: 65 for a C4 blob
	// (222 with a record per BOUNDARY edge), 131 with flipped diagonals, 194 for a Delaunay disc with holes (469), 511 for a holey disc (1 325) - half the
	// ring's base to begin with, times what the context learns from the blobs that outgrow it.  (`boundary`: rounds 4-5's floor from the header, unused now.)
	(void)boundary;
	uint64_t p = std::max<uint64_t>(64, base/2);
	const uint64_t p1 = p;
	p = (p*pool_q8 + 7)/8;
	if(pool_cap && p > pool_cap) p = std::max<uint64_t>(p1, pool_cap);
	p = (p + 63) & ~63ull;                                                        // (whole 16-byte vectors of free-list and DELAY-stack halfwords)
	pool = (uint32_t)std::min<uint64_t>(p, ring_max);
	const uint32_t all = (nclers + 64 + 31) & ~31u;       // whole 16-byte vectors of nibbles (k_mesh.hip: TOPO_FILL_WINDOW)
	symwin = all < TOPO_SYMWIN_MAX ? all : TOPO_SYMWIN_MAX;
}
constexpr uint32_t TOPO_LDS_MAX = 156*1024;     // of the CU's 160 KiB
constexpr uint32_t DELTA_THREADS = 1024, DELTA_SMALL_NVERT = 8192;      // threads of k_delta_mesh's workgroup for one (blob, attribute) too big for LDS; half of them up to DELTA_SMALL_NVERT vertices
__global__ void k_delta_mesh(const DeltaJob *jobs, uint32_t njobs, uint32_t wide_n_only);       // wide_n_only: the jobs of more than four components alone (the others are k_delta_tiles')
// k_delta.hip: the same jobs (up to four components) in tiles of DELTA_THREADS vertices out of an LDS ring of recent values (round 6)
__global__ void k_delta_tiles(const DeltaJob *jobs, uint32_t njobs);
constexpr uint32_t DELTA_GROUP_MAX = 4;
struct DeltaGroup { uint32_t first, count; };     // DeltaJob entries [first, first + count) of one blob
// k_delta.hip: one workgroup per blob, one wave per attribute (up to four) + one that builds the prediction graph they share: 16-bit values
// relative to vertex 0 (bytes for colours), a 4-byte graph word and one out-of-order window loop.  Records: 2 / 4 / 8 / 8 bytes for 1 / 2 / 3 / 4 int16 components, 4 bytes for up to four colour bytes.
constexpr uint32_t DELTA16_LDS_MAX = 128*1024, DELTA16_NVERT_MAX = 32767;
__host__ __device__ inline uint32_t delta16_rec(uint32_t N, bool is_u8) { return is_u8 ? 4u : N == 1 ? 2u : N == 2 ? 4u : 8u; }
__host__ __device__ inline uint32_t delta16_vbytes(uint32_t nvert, uint32_t N, bool is_u8) { return (nvert*delta16_rec(N, is_u8) + 15u) & ~15u; }
// ... or 32-bit components (a context that met values beyond int16): records of 4 / 8 / 16 / 16 bytes; colours stay four bytes
__host__ __device__ inline uint32_t delta_rec(uint32_t N, bool is_u8, bool wide) { return !wide || is_u8 ? delta16_rec(N, is_u8) : N == 1 ? 4u : N == 2 ? 8u : 16u; }
__host__ __device__ inline uint32_t delta_vbytes(uint32_t nvert, uint32_t N, bool is_u8, bool wide) { return (nvert*delta_rec(N, is_u8, wide) + 15u) & ~15u; }
__host__ __device__ inline bool delta16_eligible(uint32_t nvert, uint32_t N, bool is_u8) { return nvert <= DELTA16_NVERT_MAX && N >= 1 && N <= 4 && (is_u8 || true); }
// graph: 4 bytes a vertex + (unless a three-component int16 attribute of the group lends its spare halfwords) 2 bytes a vertex of `a`
__host__ __device__ inline uint32_t delta16_graph_lds(uint32_t nvert, bool a_embedded) {
	return ((4u*nvert + 15u) & ~15u) + (a_embedded ? 0u : ((2u*nvert + 15u) & ~15u)) + 16u;
}
__global__ void k_delta_lds16(const DeltaJob *jobs, const DeltaGroup *groups, uint32_t ngroups);

// k_normal.hip
__global__ void k_normal_diff(const NormalJob *jobs, const uint32_t *block_job, const uint32_t *block_first, uint32_t nblocks);
__global__ void k_normal_faces(const NormalJob *jobs, const uint32_t *block_job, const uint32_t *block_first, uint32_t nblocks,
                               float *facen, uint32_t *cnt, uint32_t *bnd);
__global__ void k_normal_fill(const NormalJob *jobs, const uint32_t *block_job, const uint32_t *block_first, uint32_t nblocks,
                              const uint32_t *start, uint32_t *cursor, uint32_t *adj);
__global__ void k_normal_flags(const NormalJob *jobs, const uint32_t *block_job, const uint32_t *block_first, uint32_t nblocks,
                               const uint32_t *bnd, uint32_t *flag);
__global__ void k_normal_vertex(const NormalJob *jobs, const uint32_t *block_job, const uint32_t *block_first, uint32_t nblocks,
                                const float *facen, const uint32_t *start, const uint32_t *cnt, const uint32_t *adj,
                                const uint32_t *flag, const uint32_t *slot);

__global__ void k_normal_blob(const NormalJob *jobs, const uint32_t *job_ids, uint32_t njobs, uint32_t lds_bytes);
// (head: cursors u16 | flag bitmap u32 | its prefix counts u16; then the boundary XORs u32 per vertex, later the adjacency u16 per corner)
__host__ __device__ inline uint32_t normal_blob_lds_head(uint32_t nvert) { const uint32_t ndw = 2*((nvert + 63)/64); return (((nvert + 2) & ~1u)*2 + ndw*4 + ndw*2 + 15u) & ~15u; }
__host__ __device__ inline uint32_t normal_blob_lds(uint32_t nvert, uint32_t nface) { const uint32_t adj = (3*nface*2 + 15) & ~15u, bnd = nvert*4; return normal_blob_lds_head(nvert) + (adj > bnd ? adj : bnd) + 64; }
constexpr uint32_t NORMAL_LDS_MAX = 150*1024;
// with the blob's face normals kept in LDS too (3 x f32 per face, behind the layout above): small blobs only
__host__ __device__ inline uint32_t normal_blob_lds_fn(uint32_t nvert, uint32_t nface) { return ((normal_blob_lds(nvert, nface) + 15u) & ~15u) + 12u*nface; }
constexpr uint32_t NORMAL_FN_LDS_MAX = 100*1024;


// k_encode.hip
__global__ void k_enc_hist(const EncChunk *chunks, uint32_t nchunks, uint32_t *counts);
__global__ void k_enc_pack(const PackJob *jobs, uint32_t njobs);
__global__ void k_enc_tun_parse(const EncStream *streams, uint32_t nstreams, uint32_t trie_lds_entries);
// probabilities in std::sort's order + the 256-word dictionary of every stream (one wave each), then the encoding trie of the
// streams listed in ids[] (built in LDS, at most trie_cap entries, written to streams[j].trie; streams[j].ntrie = its size,
// ~0u if it did not fit)
__global__ void k_enc_quantize(QuantJob job);           // float / byte attribute -> quantised integers, elementwise
__global__ void k_enc_tables(const uint32_t *counts, const uint32_t *sizes, uint32_t nstreams, EncTab *tabs);
__global__ void k_enc_trie(const EncTab *tabs, const uint32_t *ids, EncStream *streams, uint32_t nids, uint32_t trie_cap);
inline uint32_t enc_parse_lds(uint32_t trie_entries) { return 768 + ENC_STAGE + ENC_STAGE_PAD + 2*((trie_entries + 7) & ~7u); }
constexpr uint32_t ENC_TRIE_LDS_MAX = 24*1024;         // entries (48 KiB) of trie kept in LDS at most; bigger tries are walked in L2

} // namespace corto_hip
