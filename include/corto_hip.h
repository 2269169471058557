/* corto_hip.h — C ABI of the MI355X-native corto decode path (libcorto_hip.so).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch/HIP types.
 * Reference citations are relative to the upstream tree (cnr-isti-vclab/corto @ 2025-10-03).
 *
 * What it replaces
 *   crt::Decoder::Decoder(len, input)           src/decoder.cpp:41-89        -> crthip_probe / crthip_batch_create
 *   crt::Decoder::setPositions/Normals/Uvs/
 *        setColors/setAttribute/setIndex         include/corto/decoder.h:50-61 -> crthip_batch_bind[_all]
 *   crt::Decoder::decode()                       src/decoder.cpp:126-196      -> crthip_batch_decode (+ _sync)
 *   the legacy one-blob veneers CreateDecoder/DecodeMesh (src/corto_codec.h:41-43) and the WASM glue
 *   (html/js/emscripten/emcorto.cpp:14-89) map onto crthip_decode_host(), see INTEGRATION.md.
 *
 * Error behaviour: every entry point returns CRTHIP_OK (0) or a negative CRTHIP_E_* code and never
 * throws; crthip_last_error() returns the message, which for the conditions the reference throws on
 * is the reference's own string literal ("Not a crt file.", "Memory must be alignegned on 4 bytes.",
 * "Decoding topology failed", "Unknown entropy", ...).
 *
 * There is NO CPU fallback behind this ABI: if no HIP device is usable the calls fail with
 * CRTHIP_E_DEVICE.
 *
 * Device buffers: a context decodes on its own NON-BLOCKING HIP streams, which do not wait for the null stream or for any
 * stream of the caller.  Work the caller queued on the buffers it hands over (a fill of the output block, an upload of the
 * arena on another stream) must have completed before crthip_batch_decode / crthip_tunstall_decode_blocks is called, and the
 * outputs are complete when crthip_batch_sync (or the host-output call) returns - the same contract as handing a buffer to
 * any library that owns its streams.
 */
#ifndef CORTO_HIP_H
#define CORTO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRTHIP_ABI_VERSION 6   /* 2: crthip_attr_binding.stride, crthip_mesh.group_props, crthip_pool_*; 3: crthip_pool_report grew, crthip_pool_warning;
                                  4: integer / DOUBLE output formats of generic attributes, crthip_pool_device_cpus; 5: crthip_pool_set_outputs_to_host;
                                  6: crthip_pool_set_render_layouts, crthip_kernel_times names are the kernels' (unpack_wave, delta_lds16) */

/* VertexAttribute::Format, include/corto/vertex_attribute.h:32 */
enum { CRTHIP_FMT_UINT32 = 0, CRTHIP_FMT_INT32 = 1, CRTHIP_FMT_UINT16 = 2, CRTHIP_FMT_INT16 = 3,
       CRTHIP_FMT_UINT8 = 4, CRTHIP_FMT_INT8 = 5, CRTHIP_FMT_FLOAT = 6, CRTHIP_FMT_DOUBLE = 7 };
/* VertexAttribute::CODEC, include/corto/vertex_attribute.h:34 */
enum { CRTHIP_CODEC_GENERIC = 1, CRTHIP_CODEC_NORMAL = 2, CRTHIP_CODEC_COLOR = 3 };
/* VertexAttribute::Strategy, include/corto/vertex_attribute.h:33 */
enum { CRTHIP_PARALLEL = 1, CRTHIP_CORRELATED = 2 };
/* Stream::Entropy, include/corto/cstream.h:35 */
enum { CRTHIP_ENTROPY_NONE = 0, CRTHIP_ENTROPY_TUNSTALL = 1 };

enum {
	CRTHIP_OK = 0,
	CRTHIP_E_ALIGN = -1,        /* "Memory must be alignegned on 4 bytes."  src/decoder.cpp:44 */
	CRTHIP_E_MAGIC = -2,        /* "Not a crt file."                        src/decoder.cpp:51 */
	CRTHIP_E_TRUNCATED = -3,    /* stream runs past len (the reference never checks) */
	CRTHIP_E_ENTROPY = -4,      /* "Unknown entropy"                        src/cstream.cpp:82 */
	CRTHIP_E_TOPOLOGY = -5,     /* "Decoding topology failed"               src/decoder.cpp:274 */
	CRTHIP_E_NORMAL_NEEDS_POSITION = -6, /* src/normal_attribute.cpp:219-227 */
	CRTHIP_E_FORMAT = -7,       /* output format not supported for that attribute */
	CRTHIP_E_ARGUMENT = -8,
	CRTHIP_E_DEVICE = -9,       /* no usable HIP device / HIP runtime error */
	CRTHIP_E_NOMEM = -10,
	CRTHIP_E_LIMIT = -11        /* more attributes / components than this build supports */
};

#define CRTHIP_MAX_ATTRS 16
#define CRTHIP_NAME_MAX 64

typedef struct {
	char name[CRTHIP_NAME_MAX];  /* NUL-terminated */
	uint32_t codec;              /* CRTHIP_CODEC_* (anything else is treated as GENERIC, src/decoder.cpp:77-79) */
	float q;                     /* quantisation step as stored */
	uint32_t components;         /* N as stored (normals always decode 2 octahedral ints -> 3 outputs) */
	uint32_t format;             /* encoder-side format byte as stored (informational) */
	uint32_t strategy;           /* CRTHIP_PARALLEL | CRTHIP_CORRELATED */
} crthip_attr_info;

typedef struct {
	uint32_t version, entropy;
	uint32_t nvert, nface;
	uint32_t nattr;                              /* attributes in std::map order = sorted by name */
	crthip_attr_info attr[CRTHIP_MAX_ATTRS];
	uint32_t nexif;
	uint32_t body_offset;                        /* first byte after the header */
} crthip_blob_info;

/* Per-attribute output binding (Decoder::setAttribute, src/decoder.cpp:96-123).
 * buffer == NULL leaves the attribute unbound: its streams are skipped, like the reference does.
 * Buffers are DEVICE pointers for the batch API (crthip_batch_*) and HOST pointers for crthip_decode_host.
 *   generic (position, uv, radius, ...): FLOAT - nvert*components*4 bytes (decoded in place as int32 first) - is what upstream's callers use;
 *            the other formats of Decoder::setAttribute(name, buffer, format) (src/decoder.cpp:96-102) are taken too (ABI v4): the integer
 *            formats leave upstream's in-place "*= q" over the int32 array (nvert*components*4 bytes), DOUBLE widens it (nvert*components*8
 *            bytes) - what the compiled reference leaves there, include/corto/vertex_attribute.h:195-228; packed only (stride 0), 4-byte
 *            aligned (DOUBLE: 8)
 *   FLOAT buffers must be 4-byte aligned and INT16 ones 2-byte aligned (what a float* / int16_t* is), else CRTHIP_E_ARGUMENT
 *   normal : FLOAT (nvert*3 f32) or INT16 (nvert*3 i16)
 *   color  : UINT8, out_components = 3 or 4 (>= stored components); nvert*out_components bytes        */
typedef struct {
	void *buffer;
	uint32_t format;
	uint32_t out_components;
	uint32_t stride;             /* bytes from one vertex to the next; 0 = tightly packed (the layouts above).  A stride lets several
	                                attributes land in ONE interleaved vertex buffer (buffer = base + the attribute's offset in the
	                                vertex record): SURVEY.md 8f-3, the render-ready form of Decoder::setNormals(int16_t*) /
	                                setIndex(uint16_t*) (include/corto/decoder.h:53,61).  Must be a multiple of 4 for FLOAT outputs, of 2
	                                for INT16 normals, and at least the element's size; with a stride the attribute is decoded in device
	                                scratch and only its final values touch the buffer (a packed generic buffer doubles as int32 workspace) */
	uint32_t reserved;           /* flags: 0, or CRTHIP_BIND_STREAM_VALUES */
} crthip_attr_binding;
/* crthip_attr_binding.reserved, generic attributes only (ABI 6): stop behind the stream decode - `buffer` (nvert*N int32, packed, format
 * CRTHIP_FMT_INT32) receives the values as the stream holds them, i.e. what upstream's GenericAttr<int>::decode leaves in the buffer
 * (include/corto/vertex_attribute.h:151-156: InStream::decodeArray / decodeValues) BEFORE deltaDecode and dequantize.  This is the device
 * half of a caller-supplied codec object (Decoder::setAttribute(name, buffer, VertexAttribute *), src/decoder.cpp:104-114): its
 * deltaDecode / postDelta / dequantize are host code and run on these values (include/corto/decoder.h of this repo).  A position bound this
 * way cannot feed ESTIMATED / BORDER normals (upstream throws "Position attr has been overloaded" there, src/normal_attribute.cpp:210-213):
 * the blob fails with CRTHIP_E_NORMAL_NEEDS_POSITION. */
#define CRTHIP_BIND_STREAM_VALUES 1u

typedef struct crthip_ctx crthip_ctx;       /* one per device; owns streams + scratch pool */
typedef struct crthip_batch crthip_batch;   /* a planned batch of independent .crt blobs */

uint32_t crthip_abi_version(void);
const char *crthip_last_error(void);         /* thread-local message of the last failing call */
const char *crthip_strerror(int code);

/* Header parse only; host-only, needs no GPU.  blob must be 4-byte aligned (src/decoder.cpp:43-44). */
int crthip_probe(const uint8_t *blob, size_t len, crthip_blob_info *info);
/* exif pairs / group table of one blob, host-only.  Strings are copied into out (NUL-separated
 * key\0value\0...); returns bytes needed, or <0. */
int64_t crthip_probe_exif(const uint8_t *blob, size_t len, char *out, size_t cap);
/* groups: writes min(cap, ngroups) end-face markers, returns ngroups or <0 (include/corto/index_attribute.h:89-99) */
int64_t crthip_probe_groups(const uint8_t *blob, size_t len, uint32_t *group_end, size_t cap);
/* properties of group g as key\0value\0... (Group::properties); returns bytes needed, or <0 */
int64_t crthip_probe_group_props(const uint8_t *blob, size_t len, uint32_t g, char *out, size_t cap);

int crthip_ctx_create(int device, crthip_ctx **out);
void crthip_ctx_destroy(crthip_ctx *ctx);
/* A context decodes a batch on two HIP streams (CLERS streams + topology on one, the attribute streams' entropy decode and bit-unpack
 * on the other, joined before the delta stage): the shortest latency for one batch.  With many contexts on one GPU the streams
 * outnumber the hardware queues ($GPU_MAX_HW_QUEUES, ROCm default 4) and streams that share a queue serialise each other's kernels:
 * from about queues/2 contexts up, one stream per context is faster (12 contexts, 16 queues: +15 %; crthip_pool chooses by itself).
 * The same switch selects the LDS-lean layout of the normals kernel (29 KB instead of 78 KB per blob: slower alone, but with many
 * batches in flight a kernel's wait for LDS is what its latency is made of: +10 %). */
int crthip_ctx_set_single_stream(crthip_ctx *ctx, int on);
/* Blobs that already sit in ONE pinned host buffer (hipHostMalloc / hipHostRegister / torch pin_memory), laid out as
 * crthip_arena_layout says (blob i at blobs[0] + offset i): with this switch on, crthip_batch_create / _reset upload them with one
 * DMA copy straight from there - no gathering into the library's own pinned image first (3.7 MB of memcpy per C4 batch: 150 us of a
 * from-host step's host time).  The caller's promise: the buffer stays valid and unchanged until the batch has been synced.  Blob
 * pointers that are NOT laid out that way take the gathering path as before, whatever the switch says.  (No reference counterpart:
 * crt::Decoder reads its one blob from the caller's memory in place, src/decoder.cpp:41-48.) */
int crthip_ctx_set_packed_host_blobs(crthip_ctx *ctx, int on);
int crthip_device_count(void);

/* Plan a batch: parse every header, walk every body (validating all extents against lens[i]),
 * stage the blobs into one 16-byte-aligned device arena and upload the stream descriptors.
 * blobs[i] are HOST pointers (borrowed only for the duration of this call).
 * device_arena: NULL -> the library uploads the blobs itself;
 *               else  -> DEVICE pointer to the blobs already resident in HBM, laid out back to back with each
 *                        blob starting at the next 16-byte multiple (offsets via crthip_arena_layout). */
int crthip_batch_create(crthip_ctx *ctx, uint32_t nblobs, const uint8_t *const *blobs, const uint32_t *lens,
                        const void *device_arena, crthip_batch **out);
/* Re-plan an existing batch object for a new list of blobs (same meaning as destroy + create, bindings are cleared) reusing
 * its allocations: what a serving loop that decodes batch after batch on one context calls instead of create / destroy. */
int crthip_batch_reset(crthip_batch *b, uint32_t nblobs, const uint8_t *const *blobs, const uint32_t *lens, const void *device_arena);
/* offsets[i] = arena byte offset of blob i under the rule above; returns total arena bytes */
uint64_t crthip_arena_layout(uint32_t nblobs, const uint32_t *lens, uint64_t *offsets);
void crthip_batch_destroy(crthip_batch *b);

uint32_t crthip_batch_size(const crthip_batch *b);
int crthip_batch_info(const crthip_batch *b, uint32_t i, crthip_blob_info *info);

/* Bind outputs of blob i. attrs has info.nattr entries in info.attr order. index: DEVICE pointer to
 * nface*3 entries of index_format (CRTHIP_FMT_UINT32 or CRTHIP_FMT_UINT16), NULL for point clouds. */
int crthip_batch_bind(crthip_batch *b, uint32_t i, const crthip_attr_binding *attrs, void *index, uint32_t index_format);
/* Bind every blob in one call: attrs holds sum(nattr) entries blob after blob; index/index_format have nblobs entries. */
int crthip_batch_bind_all(crthip_batch *b, const crthip_attr_binding *attrs, void *const *index, const uint32_t *index_format);

/* Enqueue the whole decode of the batch on the context's stream (asynchronous). */
int crthip_batch_decode(crthip_batch *b);
/* Wait for completion and collect per-blob status: status[i] = CRTHIP_OK or CRTHIP_E_* (may be NULL).
 * Returns CRTHIP_OK if every blob decoded, else the first failing blob's code. */
int crthip_batch_sync(crthip_batch *b, int32_t *status);
/* Without waiting: 1 if crthip_batch_sync would return at once (the decode has finished, or none is in flight), 0 if it is still
 * running, <0 on error.  For callers that keep several batches in flight and refill whichever finishes first (crthip_pool). */
int crthip_batch_done(crthip_batch *b);

/* One-blob convenience with HOST output buffers: probe + plan + device decode + copy back.
 * This is what the crt::Decoder facade (include/corto/decoder.h of this repo) calls.
 * attrs == NULL: no attribute is bound (an index-only decode, like a Decoder nobody called set*() on).  Host buffers have
 * upstream's tightly packed layouts: a binding with a non-zero stride is refused with CRTHIP_E_ARGUMENT (strides are for
 * DEVICE vertex buffers, crthip_batch_bind). */
int crthip_decode_host(crthip_ctx *ctx, const uint8_t *blob, size_t len, const crthip_attr_binding *attrs,
                       void *index, uint32_t index_format);

/* ---- multi-GPU decode pool ------------------------------------------------------------------------
 * Upstream decodes one blob on one thread; its Decoder objects are independent (src/decoder.cpp:126-196 touches nothing
 * shared), so a list of blobs shards by blob with NO collective.  The pool is that, for the GPUs of one node: `ndevices`
 * devices, `depth` batches in flight per host thread and `threads_per_device` host threads per device, every batch on its
 * own context (crthip_ctx: own HIP streams, scratch, descriptors) with its own output block in that device's HBM.  All
 * threads of all devices pull tickets from ONE queue - an atomic counter - so a faster or less loaded GPU simply takes more steps:
 * no RCCL, no peer traffic.  Which item a ticket decodes is home-shard-first: pool device d prefers the items j with
 * j % ndevices == d (the shard resident in ITS HBM: give device_arena[d] for those and NULL elsewhere), and a device without a home
 * item takes the others' (and uploads them, since they are not resident there).  Worker threads are pinned to the CPUs of their GPU's
 * NUMA node when sysfs names one.
 * devices == NULL: devices 0 .. ndevices-1.  A device id may repeat (several pool "devices" on one GPU: how the N > 1 path
 * is exercised on a one-GPU box). */
typedef struct crthip_pool crthip_pool;
int crthip_pool_create(uint32_t ndevices, const int *devices, uint32_t threads_per_device, uint32_t depth, crthip_pool **out);
void crthip_pool_destroy(crthip_pool *pool);
uint32_t crthip_pool_lanes(const crthip_pool *pool);     /* ndevices * threads_per_device * depth contexts */
/* "" or what crthip_pool_create found wrong with the hardware queues (also printed to stderr once): every context needs a queue of
 * its own, ROCm hands out $GPU_MAX_HW_QUEUES (default 4) per process and reads it when HIP initialises - a pool of 16 contexts on
 * the default runs at a fraction of its rate.  Contexts are counted per physical GPU (a device id may repeat). */
const char *crthip_pool_warning(const crthip_pool *pool);
/* The host CPUs of pool device `device_slot`'s NUMA node (what the pool pins that device's worker threads to): up to cap ids into
 * cpus, returns how many there are (0: sysfs names no node).  A caller that allocates the pinned input buffers of that device's shard
 * on one of these CPUs gets them on the memory next to the GPU's PCIe root (bench.py does). */
int64_t crthip_pool_device_cpus(const crthip_pool *pool, uint32_t device_slot, int32_t *cpus, size_t cap);
/* crthip_ctx_set_packed_host_blobs for every context of the pool (items handed to crthip_pool_run without device arenas). */
int crthip_pool_set_packed_host_blobs(crthip_pool *pool, int on);
/* SURVEY 8d's secondary region: every step of crthip_pool_run ends with ONE device-to-host copy of its decoded outputs into a pinned host block
 * of the lane (queued on the context's stream behind the kernels; a step is complete when the copy is), and crthip_pool_lane_read returns what
 * that copy delivered.  What a host-side consumer of the outputs sees - the reference's own region, decode() into host buffers
 * (src/main.cpp:266-300).  Off by default: outputs stay in HBM. */
int crthip_pool_set_outputs_to_host(crthip_pool *pool, int on);
/* SURVEY 8f3's render layouts for every lane's outputs: normals as int16 (upstream's NormalAttr INT16 output, src/normal_attribute.cpp:203-208,
 * 317-323) and the index as uint16 where a blob has fewer than 65 536 vertices (Decoder::setIndex(uint16_t *), include/corto/decoder.h:61) - 22 % fewer
 * output bytes for a C4 blob, which is what the secondary region's D2H copy moves.  crthip_pool_lane_read returns those bytes. */
int crthip_pool_set_render_layouts(crthip_pool *pool, int on);

/* One work item = one batch of blobs (HOST pointers, borrowed for the duration of crthip_pool_run).
 * device_arena: NULL -> every execution uploads the blobs (pageable or pinned host memory -> HBM) inside the step (SURVEY.md 8d's primary
 *                       region): one DMA copy at the head of the context's own stream, as crthip_batch_create does;
 *               else ndevices DEVICE pointers, entry d = the item's blobs already resident on pool device d in
 *               crthip_arena_layout order (an entry may be NULL: that device uploads). */
typedef struct {
	uint32_t nblobs;
	const uint8_t *const *blobs;
	const uint32_t *lens;
	const void *const *device_arena;
} crthip_pool_item;

typedef struct {
	double elapsed_s;            /* wall time from the completion of the last warm-up step to the completion of the last timed step:
	                                the pipeline is full at both ends (extra steps are queued behind the timed ones and drained untimed) */
	uint64_t steps;              /* timed steps completed (= the `steps` asked for) */
	uint64_t triangles, vertices;/* decoded by the timed steps */
	uint64_t failed_blobs;       /* blobs (over all executed steps) whose status was not CRTHIP_OK */
	int32_t first_error;         /* first failing status seen, or CRTHIP_OK */
	uint32_t devices_used;       /* pool devices that completed at least one timed step */
	uint64_t steps_per_device[16];
	uint64_t topology_fallbacks;
	uint32_t poisoned_lanes;     /* contexts whose LAST executed step started from an output block the pool had just filled with 0xA5 on the
	                                context's stream: the last round of timed steps and the tail behind them are run that way, so what
	                                crthip_pool_lane_read returns afterwards was written by those steps and by nothing earlier */
	uint32_t pinned_devices;     /* pool devices whose worker threads were pinned to the CPUs of the GPU's NUMA node */
	float host_us_per_step;      /* host time per step and thread: plan (walk + bind) + enqueue, averaged over every executed step */
	float host_plan_max_us;      /* the LONGEST single plan call (walk + the upload's enqueue + bind) of the run ... */
	float host_wait_us, host_finish_us, host_plan_us;   /* of a worker thread's time per step: waiting for one of its contexts to finish; harvesting it
	                                (sync, status); the walk + bind part of host_us_per_step */
	float host_launch_max_us;    /* ... and the longest single crthip_batch_decode call: a host thread that blocks inside the runtime shows here */
} crthip_pool_report;

/* Decode warmup + steps batches drawn cyclically from the items (each device from its home items, see above; + a few more steps to
 * keep every context busy until the last timed completion), outputs into the contexts' own device blocks: every attribute bound in its natural format
 * (generic FLOAT, normal FLOAT, colour UINT8 x 4, index UINT32).  completion_s: NULL, or `steps` doubles that receive the
 * completion time of every timed step in seconds since the start of the timed region, in completion order.
 * Blocks until everything has drained.  Returns CRTHIP_OK or the first HIP / planning error (per-blob decode failures are
 * counted in the report, not returned). */
int crthip_pool_run(crthip_pool *pool, uint32_t nitems, const crthip_pool_item *items, uint64_t steps, uint64_t warmup,
                    crthip_pool_report *report, double *completion_s);

/* After a run every lane (context) still holds the outputs of the last step it executed: which item that was, on which pool
 * device, and a copy of one output array of one of its blobs to the host ("position", "normal", "color", "uv", ... or "index").
 * ("#tail": the last 256 bytes of the context's output block, behind every array - 0xA5 after a run, see poisoned_lanes.)
 * Returns bytes written / the item index, or <0.  This is how bench.py and the tests check what every GPU decoded. */
int64_t crthip_pool_lane_item(const crthip_pool *pool, uint32_t lane, uint32_t *device_slot);
int64_t crthip_pool_lane_read(crthip_pool *pool, uint32_t lane, uint32_t blob, const char *what, void *host_out, size_t cap);

/* ---- .crt writer (host only) -------------------------------------------------------------
 * Byte-identical to upstream's crt::Encoder (src/encoder.cpp:207-722) for positions, normals (all three predictions),
 * rgb/rgba colours, uvs, one generic "radius" attribute, groups, exif, entropy NONE/TUNSTALL, meshes and point clouds.
 * Lets tests and bench.py synthesise inputs without the reference library. */
typedef struct {
	uint32_t nvert, nface;
	const float *position;        /* nvert*3, required */
	const uint32_t *index;        /* nface*3, NULL for point clouds */
	int32_t position_bits;        /* >0: step = max extent / 2^bits (Encoder::addPositionsBits); else position_q */
	float position_q;
	const float *normal;          /* nvert*3 or NULL */
	int32_t normal_bits;
	int32_t normal_prediction;    /* 0 DIFF, 1 ESTIMATED, 2 BORDER */
	const uint8_t *color;         /* nvert*color_components or NULL */
	int32_t color_components;     /* 3 or 4 */
	int32_t color_bits[4];
	const float *uv;              /* nvert*2 or NULL */
	float uv_q;
	const float *radius;          /* nvert or NULL */
	float radius_q;
	const uint32_t *group_end;    /* ngroups end-face markers or NULL */
	uint32_t ngroups;
	int32_t entropy;              /* CRTHIP_ENTROPY_* */
	const char *exif;             /* "k\0v\0..." nexif pairs or NULL */
	uint32_t nexif;
	/* Group::properties (Encoder::addGroup(end, props), include/corto/encoder.h:75): group g has group_nprops[g] pairs, all pairs
	 * of all groups flat in group_props as "k\0v\0...".  Written sorted by key like upstream's std::map.  NULL: no properties. */
	const uint32_t *group_nprops;
	const char *group_props;
} crthip_mesh;
/* returns the blob size (also when out == NULL or cap is too small), or <0 */
int64_t crthip_encode(const crthip_mesh *mesh, uint8_t *out, size_t cap, uint32_t *out_nvert, uint32_t *out_nface);

/* ---- measurement / test hooks (not needed by integrators) ---- */
typedef struct {
	uint64_t arena_bytes;       /* compressed input resident in HBM */
	uint64_t output_bytes;      /* bytes of all bound outputs */
	uint64_t tunstall_in;       /* compressed bytes through the Tunstall kernel */
	uint64_t tunstall_out;      /* decoded symbol bytes out of it */
	uint64_t tunstall_tables;   /* probability-table header bytes (1 + 2*nsym + 8 per stream) */
	uint32_t tunstall_streams;
	uint64_t total_nvert, total_nface;
	uint64_t scratch_bytes;
	uint64_t clers_symbols;     /* decoded CLERS symbols over all mesh blobs */
	uint64_t split_bytes;       /* bytes of the split / vertex-id bit blocks */
	uint64_t topology_fallbacks;/* after crthip_batch_sync: mesh blobs whose CLERS automaton outgrew its LDS edge slots and was
	                               redone with the front in HBM (same results, slower) */
	float host_plan_us;         /* host time of the last crthip_batch_decode: job planning (descriptor build) ... */
	float host_stage_us;        /* ... pointer fix-up + staging of the descriptors ... */
	float host_launch_us;       /* ... and issuing the copies / kernel launches (asynchronous; no device wait) */
	float host_create_us;       /* host time of crthip_batch_create: header parse + bounds-checked walk of every blob */
	uint32_t topology_scale;    /* factor on the LDS edge slots this decode was planned with (1, 2, 4 ...): the context raises it
	                               after a batch with fallbacks, so that meshes with long fronts (handles, many boundary loops)
	                               stay in LDS from the next batch on, and lowers it again after a long run without any */
	uint32_t tunstall_dictionaries; /* Tunstall dictionaries the last decode BUILT: streams of a batch that carry the same probability table
	                               share one (the dictionary is a function of the table alone, src/tunstall.cpp:125-256), so this is
	                               <= tunstall_streams; $CORTO_TUN_SHARE=0 builds one per stream */
	uint32_t delta_redone;      /* after crthip_batch_sync: blobs with an attribute whose values, relative to vertex 0, left int16 (K-DELTA's LDS
	                               layout) and were redone on the 32-bit values in HBM (same results, slower); the context plans its next
	                               batches with 32-bit values in LDS when that happens */
	uint32_t delta_walked;      /* after crthip_batch_sync: blobs of which K-DELTA finished an attribute with its ROUND LOOP (a scan of affine maps: parents one or
	                               two vertices back) rather than its window of prefix sums: irregular connectivity (k_delta.hip; the name is rounds 3-4's, when the
	                               fallback was a walk along the stretches of the prediction graph) */
	uint32_t delta_wide;        /* 1: this decode was planned with 32-bit values in K-DELTA's LDS (the context had met such blobs, or $CORTO_DELTA_WIDE=1) */
	uint32_t descriptor_bytes;  /* the job descriptors of the last decode: one host -> HBM copy beside the blobs' own bytes (it shares their PCIe link) */
	uint32_t int16_streams;     /* log streams of the last decode whose values K-BIT handed on as int16 instead of int32: the stream's probability table holds no
	                               width above 16 bits (15 for per-component streams), and the reader is the LDS-resident K-DELTA / K-NRM ($CORTO_VALUES_I32=1: none) */
} crthip_batch_stats;
int crthip_batch_get_stats(const crthip_batch *b, crthip_batch_stats *s);

/* Per-kernel device time of the LAST crthip_batch_decode of this batch, measured with HIP events on the
 * context's stream.  Enable before decode with crthip_ctx_set_profiling(ctx, 1).  names[i] static strings. */
#define CRTHIP_MAX_KERNELS 32
typedef struct {
	uint32_t count;
	const char *name[CRTHIP_MAX_KERNELS];
	float ms[CRTHIP_MAX_KERNELS];
	uint32_t launches[CRTHIP_MAX_KERNELS];
} crthip_kernel_times;
int crthip_ctx_set_profiling(crthip_ctx *ctx, int enable);
int crthip_batch_kernel_times(crthip_batch *b, crthip_kernel_times *t);
/* The prediction triples of blob i of the last decode (upstream's index.prediction, include/corto/index_attribute.h:34-38,76: one Face
 * {a, b, c} per vertex in decode order - what VertexAttribute::deltaDecode takes as its context): nvert*3 uint32 copied to HOST memory;
 * returns the bytes written (0 for a point cloud), < 0 on error.  Valid until the context's next decode. */
int64_t crthip_batch_read_prediction(crthip_batch *b, uint32_t i, void *host_out, size_t cap);

/* Copy an internal intermediate of blob i to a HOST buffer (tests compare these with the oracle):
 * what = "clers" (u8 symbols), "prediction" (nvert*3 u32). Returns bytes written or <0. */
int64_t crthip_batch_debug_read(crthip_batch *b, uint32_t i, const char *what, void *host_out, size_t cap);

/* Stand-alone Tunstall run on DEVICE-resident blocks, used for the HBM-roofline measurement of the
 * Tunstall kernels: n blocks, each "u8 nsym | nsym*(sym,prob) | u32 size | u32 csize | payload"
 * exactly as in the .crt stream (src/cstream.cpp:89-109), block_offset[i] bytes into device_blocks;
 * out_offset[i] bytes into device_out.  host_blocks is the same memory on the host (for framing). */
int crthip_tunstall_decode_blocks(crthip_ctx *ctx, uint32_t n, const uint8_t *host_blocks, const void *device_blocks,
                                  const uint64_t *block_offset, void *device_out, const uint64_t *out_offset,
                                  crthip_kernel_times *times);
int crthip_ctx_sync(crthip_ctx *ctx);

/* GPU encoder stage: the entropy coder of crt::Encoder for a batch of n byte streams, i.e.
 * OutStream::tunstall_compress (src/cstream.cpp:89-109) n times: byte histogram and greedy Tunstall parse
 * (Tunstall::getProbabilities / compress, src/tunstall.cpp:83-115, 384-428) run on the device, one wave per stream;
 * the 256-word dictionary and its trie (src/tunstall.cpp:125-256, 335-382) are built on the host in between.
 * src[i] / sizes[i]: HOST symbol arrays (<= 2^23 symbols each).  Writes the n blocks back to back into out
 * ("u8 nsym | nsym*(sym,prob) | i32 size | i32 csize | codewords", byte-identical to the reference's), their
 * starts into block_offset[0..n] (n+1 entries), and returns the total size, or <0.  out == NULL sizes only.
 * No CPU fallback: CRTHIP_E_DEVICE without a HIP device. */
int64_t crthip_tunstall_encode_blocks(crthip_ctx *ctx, uint32_t n, const uint8_t *const *src, const uint32_t *sizes,
                                      uint8_t *out, size_t cap, uint64_t *block_offset, crthip_kernel_times *times);

/* GPU encoder stage: the value coders of crt::Encoder for a batch of integer arrays - OutStream::encodeArray<int>
 * (one bit width per element, include/corto/cstream.h:143-164), OutStream::encodeValues<int|char> (component-major, one
 * width per value, sign folded, :115-141) and plain symbol streams.  Bit widths and bit packing (src/bitstream.cpp:86-101)
 * run on the device; the width arrays go through the Tunstall coder above without leaving HBM (entropy 1), or are
 * written raw (entropy 0, src/cstream.cpp:43-64).
 * Stream i is written as  "u32 nwords | nwords x u32 | block(s)"  (symbol streams: the block alone).  The reference pads
 * with zeros to a 4-byte stream position between nwords and the words (OutStream::write(BitStream&), cstream.h:79-89);
 * that depends on where the caller puts the stream, so it is the caller's to insert.  Returns the total size or <0. */
#define CRTHIP_ENC_SYMBOLS 0u       /* count bytes */
#define CRTHIP_ENC_ARRAY 1u         /* count x components int32, encodeArray */
#define CRTHIP_ENC_VALUES_I32 2u    /* count x components int32, encodeValues<int> */
#define CRTHIP_ENC_VALUES_I8 3u     /* count x components int8, encodeValues<char> (colours) */
typedef struct crthip_enc_stream {
	uint32_t kind, count, components, reserved;
	const void *values;             /* HOST pointer */
} crthip_enc_stream;
int64_t crthip_encode_values(crthip_ctx *ctx, uint32_t entropy, uint32_t n, const crthip_enc_stream *streams,
                             uint8_t *out, size_t cap, uint64_t *stream_offset, crthip_kernel_times *times);

/* crthip_encode with the value coding and the entropy coder on the device (the topology pass, quantisation and the
 * container stay on the host): same arguments plus the context, byte-identical output. */
int64_t crthip_encode_gpu(crthip_ctx *ctx, const crthip_mesh *mesh, uint8_t *out, size_t cap, uint32_t *out_nvert, uint32_t *out_nface);

#ifdef __cplusplus
}
#endif
#endif /* CORTO_HIP_H */
