/* oracle/corto_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference (cnr-isti-vclab/corto) decode path, stage by stage.
 * It is the checker that travels to the GPU box; the product (corto_amd/) never links or calls it.
 * Parity of this restatement is PINNED against the unmodified reference compiled from
 * /root/reference (oracle/_ref, see oracle/Makefile) by tests/test_oracle_vs_reference.py and by the
 * golden fixtures under tests/golden/ that the reference itself generated
 * (tests/golden/make_golden.py).  The reference ships no tests or golden vectors of its own
 * (SURVEY.md §4), so "outputs of the reference itself run here" is the pin.
 */
#ifndef CORTO_ORACLE_H
#define CORTO_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { CO_FMT_UINT32 = 0, CO_FMT_INT32, CO_FMT_UINT16, CO_FMT_INT16, CO_FMT_UINT8, CO_FMT_INT8, CO_FMT_FLOAT, CO_FMT_DOUBLE };
enum { CO_CODEC_GENERIC = 1, CO_CODEC_NORMAL = 2, CO_CODEC_COLOR = 3 };
enum { CO_PARALLEL = 1, CO_CORRELATED = 2 };
enum { CO_NORMAL_DIFF = 0, CO_NORMAL_ESTIMATED = 1, CO_NORMAL_BORDER = 2 };
#define CO_MAX_ATTRS 16
#define CO_NAME_MAX 64

typedef struct {
	char name[CO_NAME_MAX];
	uint32_t codec;       /* as stored */
	float q;
	uint32_t N, format, strategy;
} co_attr_info;

typedef struct {
	uint32_t version, entropy, nexif, nattr, nvert, nface;
	uint32_t body_offset;             /* byte offset of the first byte after nvert,nface */
	co_attr_info attr[CO_MAX_ATTRS];  /* sorted by name (std::map order), duplicates collapsed */
} co_header;

/* src/decoder.cpp:41-89. returns 0, or <0: -1 misaligned, -2 bad magic, -3 too many attrs / truncated */
int co_parse_header(const uint8_t *blob, size_t len, co_header *h);

/* ---- Tunstall (src/tunstall.cpp:125-256, 430-452) ---- */
typedef struct {
	int n;
	uint8_t sym[256], prob[256];
	uint32_t index[256], length[256];
	uint8_t table[8192 + 512];
	uint32_t table_size;
} co_tunstall;
void co_tunstall_build(co_tunstall *t, const uint8_t *probs /* n x (sym,prob) */, int n);
void co_tunstall_decode(const co_tunstall *t, const uint8_t *in, uint32_t csize, uint8_t *out, uint32_t size);

/* ---- MSB-first bit fields over native u32 words (src/bitstream.cpp:103-121) ---- */
uint32_t co_bits(const uint32_t *words, uint64_t bitoff, uint32_t n);

/* ---- per-attribute binding (Decoder::setAttribute & friends, src/decoder.cpp:96-123) ---- */
typedef struct {
	const char *name;
	void *buffer;
	uint32_t format;          /* CO_FMT_* ; FLOAT for position/uv/generic, FLOAT or INT16 for normal, UINT8 for color */
	uint32_t out_components;  /* color only (3 or 4) */
} co_binding;

typedef struct {
	const co_binding *bind;
	uint32_t nbind;
	uint32_t *index32;
	uint16_t *index16;
} co_outputs;

/* optional capture of intermediates (all caller-allocated or NULL) */
typedef struct {
	uint8_t *clers;       uint32_t nclers;       /* decoded CLERS symbols (cap nclers_cap) */
	uint32_t nclers_cap;
	uint32_t *prediction;                        /* nvert*3 (a,b,c) */
	uint32_t max_front;
	uint32_t front_size;                         /* final front.size() of the last group */
	int32_t *attr_raw[CO_MAX_ATTRS];             /* per attribute (header order): values after bit-unpack, before delta; N*nvert (normals: 2*nvert) */
	int32_t *attr_delta[CO_MAX_ATTRS];           /* after delta, before dequantise */
	uint32_t normal_ndiffs;                      /* number of normal diffs read (BORDER: #boundary) */
	uint64_t tunstall_in, tunstall_out;          /* bytes through the Tunstall stage (compressed, decoded) */
	uint32_t nstreams;
} co_trace;

/* Whole decode, stage order of src/decoder.cpp:133-196. returns 0 or <0 (see .c). */
int co_decode(const uint8_t *blob, size_t len, const co_outputs *o, co_trace *trace);
const char *co_strerror(int err);

#ifdef __cplusplus
}
#endif
#endif
