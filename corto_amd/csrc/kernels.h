// kernels.h — declarations of the HIP kernels (k_tunstall.hip, k_stream.hip, k_mesh.hip, k_normal.hip)
#pragma once
#include <hip/hip_runtime.h>

#include "device_plan.h"

namespace corto_hip {

// k_tunstall.hip
__global__ void k_tun_tables(const TunStream *streams, uint32_t nstreams, TunTable *tables);
__global__ void k_tun_chunk_sums(const TunStream *streams, const uint32_t *chunk_stream, uint32_t nchunks, const TunTable *tables,
                                 uint64_t *chunk_out, uint32_t chunk_base);
__global__ void k_tun_decode(const TunStream *streams, const uint32_t *chunk_stream, uint32_t nchunks, const TunTable *tables,
                             const uint64_t *chunk_out, uint32_t chunk_base);
// Dwords of a word's zero-padded copy the staged decode composes from (k_tunstall.hip): by the dictionary's longest word;
// steps of fewer than 8 codewords per lane (mean word length > 8) are only compiled for 4.
__host__ __device__ inline uint32_t tun_width(uint32_t cpl, uint32_t maxlen) { return cpl < 8 || maxlen > 8 ? 4u : maxlen > 4 ? 2u : 1u; }
int launch_tun_decode_staged(hipStream_t st, const TunStream *streams, const uint32_t *chunk_stream, uint32_t nchunks, const TunTable *tables,
                             const uint64_t *chunk_out);    // three kernels: words <= 4 bytes, <= 8 bytes, longer
__global__ void k_fill(const FillJob *jobs, uint32_t njobs);

// k_stream.hip
__global__ void k_scan_u64(uint64_t *a, uint32_t n);
__global__ void k_u32_chunk_sums(const uint32_t *a, uint32_t n, uint64_t *partial);
__global__ void k_u32_chunk_apply(const uint32_t *a, uint32_t *out, uint32_t n, const uint64_t *partial);
__global__ void k_unpack_sums(const UnpackJob *jobs, const uint32_t *chunk_job, uint32_t nchunks, uint64_t *partial);
__global__ void k_unpack_extract(const UnpackJob *jobs, const uint32_t *chunk_job, uint32_t nchunks, const uint64_t *partial);
__global__ void k_cloud_sums(const CloudJob *jobs, const uint32_t *chunk_job, uint32_t nchunks, uint64_t *partial);
__global__ void k_cloud_apply(const CloudJob *jobs, const uint32_t *chunk_job, uint32_t nchunks, const uint64_t *partial);
__global__ void k_dequant(const DequantJob *jobs, const uint32_t *block_job, uint32_t nblocks);

// k_mesh.hip
__global__ void k_topology(const TopoJob *jobs, const uint32_t *job_ids, uint32_t njobs);
__global__ void k_topology_lds(const TopoJob *jobs, const uint32_t *job_ids, uint32_t njobs);
// dynamic LDS bytes k_topology_lds needs for a front of `cap` edges and `nclers` symbols
inline uint32_t topo_lds_bytes(uint32_t cap, uint32_t dcap, uint32_t nclers) { return (cap + 4)*16 + (((dcap + 4)*2 + 15) & ~15u) + (((nclers + 64 + 7)/8*4 + 15) & ~15u); }
// Edge-record slots the LDS path gets: records exist only for edges that wait in the queue (one per VERTEX / SPLIT, three
// per seed face) or end a chain on the boundary, about one per vertex; a blob that needs more is redone on the HBM front.
inline uint32_t topo_lds_slots(uint32_t front_cap, uint32_t nvert) { const uint32_t want = nvert + nvert/16 + 64; return want < front_cap ? want : front_cap; }
constexpr uint32_t TOPO_LDS_DELAYED = 256;
constexpr uint32_t TOPO_LDS_MAX = 156*1024;     // of the CU's 160 KiB
constexpr uint32_t DELTA_THREADS = 256;       // threads of the dataflow workgroup of one (blob, attribute)
__global__ void k_delta_mesh(const DeltaJob *jobs, uint32_t njobs, uint32_t lds_bytes);

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

__global__ void k_normal_blob(const NormalJob *jobs, const uint32_t *job_ids, uint32_t njobs);
inline uint32_t normal_blob_lds(uint32_t nvert, uint32_t nface) { const uint32_t adj = (3*nface*2 + 15) & ~15u, bnd = nvert*4; return (nvert + 1)*4 + 2*((nvert + 2) & ~1u)*2 + (adj > bnd ? adj : bnd) + 64; }
constexpr uint32_t NORMAL_LDS_MAX = 150*1024;


} // namespace corto_hip
