/* oracle/corto_oracle.c — TEST INFRASTRUCTURE ONLY (see corto_oracle.h).
 *
 * CPU restatement, in plain C, of what the reference decoder computes.  Every function cites the
 * reference file:line whose behaviour it restates (paths relative to /root/reference).  Written
 * from the behavioural description in SURVEY.md §9/§10, not transliterated: streams are walked with
 * random-access bit offsets, Tunstall words live in one flat entry table, the CLERS front is a set
 * of parallel arrays.  Build: oracle/Makefile (-O2 -ffp-contract=off, no -march=native: FMA
 * contraction changes BORDER normals).
 */
#include "corto_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { E_OK = 0, E_ALIGN = -1, E_MAGIC = -2, E_HEADER = -3, E_TRUNC = -4, E_ENTROPY = -5, E_TOPOLOGY = -6,
       E_NORMAL_NEEDS_POSITION = -7, E_FORMAT = -8, E_NOMEM = -9 };

const char *co_strerror(int e) {
	switch(e) {
	case E_OK: return "ok";
	case E_ALIGN: return "Memory must be alignegned on 4 bytes.";   /* src/decoder.cpp:44 (sic) */
	case E_MAGIC: return "Not a crt file.";                          /* src/decoder.cpp:51 */
	case E_HEADER: return "malformed header";
	case E_TRUNC: return "truncated stream";
	case E_ENTROPY: return "Unknown entropy";                        /* src/cstream.cpp:82 */
	case E_TOPOLOGY: return "Decoding topology failed";              /* src/decoder.cpp:274 */
	case E_NORMAL_NEEDS_POSITION: return "No position attribute found. Use DIFF normal strategy instead."; /* src/normal_attribute.cpp:220 */
	case E_FORMAT: return "unsupported output format";
	case E_NOMEM: return "out of memory";
	}
	return "unknown";
}

/* ------------------------------------------------------------------------------------------------
 * little-endian byte cursor (include/corto/cstream.h:240-291).  The reference never bounds-checks;
 * the oracle does (so malformed inputs fail instead of crashing the test process). */
typedef struct { const uint8_t *base; size_t len, pos; int err; } cur_t;

static int need(cur_t *c, size_t n) {
	if(c->err || c->pos + n > c->len) { c->err = E_TRUNC; return 0; }
	return 1;
}
static uint32_t rd8(cur_t *c) { if(!need(c, 1)) return 0; return c->base[c->pos++]; }
static uint32_t rd16(cur_t *c) { if(!need(c, 2)) return 0; uint32_t v = c->base[c->pos] | (c->base[c->pos+1] << 8); c->pos += 2; return v; }
static uint32_t rd32(cur_t *c) {
	if(!need(c, 4)) return 0;
	const uint8_t *p = c->base + c->pos; c->pos += 4;
	return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static float rdf(cur_t *c) { uint32_t u = rd32(c); float f; memcpy(&f, &u, 4); return f; }
/* string = u16 byte count (incl. NUL) + bytes (cstream.h:277-280) */
static const char *rdstr(cur_t *c, uint32_t *n) {
	uint32_t k = rd16(c);
	if(!need(c, k)) { *n = 0; return ""; }
	const char *s = (const char *)(c->base + c->pos); c->pos += k; *n = k; return s;
}

/* src/decoder.cpp:41-89 */
static int parse_header(cur_t *c, co_header *h) {
	memset(h, 0, sizeof(*h));
	if(((uintptr_t)c->base) & 3) return E_ALIGN;
	if(rd32(c) != 0x787A6300u || c->err) return E_MAGIC;
	h->version = rd32(c);
	h->entropy = rd8(c);
	h->nexif = rd32(c);
	for(uint32_t i = 0; i < h->nexif && !c->err; i++) { uint32_t n; rdstr(c, &n); rdstr(c, &n); }
	uint32_t nattr = rd32(c);
	for(uint32_t i = 0; i < nattr && !c->err; i++) {
		uint32_t n; const char *name = rdstr(c, &n);
		co_attr_info a; memset(&a, 0, sizeof(a));
		size_t k = strnlen(name, n);
		if(k >= CO_NAME_MAX) return E_HEADER;
		memcpy(a.name, name, k);
		a.codec = rd32(c); a.q = rdf(c); a.N = rd8(c); a.format = rd8(c); a.strategy = rd8(c);
		/* std::map<string,...>: keep sorted by name, later duplicate replaces earlier (decoder.cpp:85) */
		uint32_t j = 0;
		while(j < h->nattr && strcmp(h->attr[j].name, a.name) < 0) j++;
		if(j < h->nattr && strcmp(h->attr[j].name, a.name) == 0) { h->attr[j] = a; continue; }
		if(h->nattr == CO_MAX_ATTRS) return E_HEADER;
		memmove(&h->attr[j+1], &h->attr[j], (h->nattr - j)*sizeof(co_attr_info));
		h->attr[j] = a; h->nattr++;
	}
	h->nvert = rd32(c);
	h->nface = rd32(c);
	h->body_offset = (uint32_t)c->pos;
	return c->err ? E_TRUNC : E_OK;
}

int co_parse_header(const uint8_t *blob, size_t len, co_header *h) {
	cur_t c = { blob, len, 0, 0 };
	return parse_header(&c, h);
}

/* ------------------------------------------------------------------------------------------------
 * Tunstall dictionary — src/tunstall.cpp:125-256 (createDecodingTables2).
 * Entry e (creation order) = (prob, offset into byte store, length); entry e lives in FIFO row e % n;
 * head[row] = oldest not-yet-expanded entry of that row. */
#define TE_CAP 1024
void co_tunstall_build(co_tunstall *t, const uint8_t *probs, int n) {
	memset(t, 0, sizeof(*t));
	t->n = n;
	for(int i = 0; i < n; i++) { t->sym[i] = probs[2*i]; t->prob[i] = probs[2*i+1]; }
	if(n <= 1) return;                                               /* tunstall.cpp:127 */

	static _Thread_local uint32_t eprob[TE_CAP], eoff[TE_CAP], elen[TE_CAP], head[256], P[256];
	memset(eprob, 0, sizeof(eprob)); memset(eoff, 0, sizeof(eoff)); memset(elen, 0, sizeof(elen));
	uint8_t *buf = t->table;
	const uint32_t un = (uint32_t)n, cap = sizeof(t->table);
	for(uint32_t i = 0; i < un; i++) P[i] = (uint32_t)t->prob[i] << 8;

	/* how long a run of the most probable symbol stays likelier than the 2nd symbol (tunstall.cpp:143-151) */
	uint32_t count = 2, run = (P[0]*P[0]) >> 16, max_count = 255/(un - 1);
	while(run > P[1] && count < max_count) { run = (run*P[0]) >> 16; count++; }

	uint32_t pos = 0, end = 0, nwords = 0;
	if(count >= 16) {                                                /* low entropy seed, tunstall.cpp:153-193 */
		buf[pos++] = t->sym[0];
		for(uint32_t k = 1; k < un; k++) {
			for(uint32_t i = 0; i + 1 < count; i++) buf[pos++] = t->sym[0];
			buf[pos++] = t->sym[k];
		}
		head[0] = (count - 1)*un;
		for(uint32_t k = 1; k < un; k++) head[k] = k;
		uint32_t pw = 0;                                             /* P0^col in 16.16 (valid from col 1) */
		for(uint32_t col = 0; col < count; col++) {
			for(uint32_t row = 1; row < un; row++) {
				uint32_t e = row + col*un;
				eprob[e] = col == 0 ? P[row] : (pw*P[row]) >> 16;
				eoff[e] = row*count - col;                           /* suffix of A^(count-1)·sym_row */
				elen[e] = col + 1;
			}
			pw = col == 0 ? P[0] : (pw*P[0]) >> 16;
		}
		uint32_t first = (count - 1)*un;                             /* the all-A word A^count */
		eprob[first] = pw; eoff[first] = 0; elen[first] = count;
		nwords = 1 + count*(un - 1);
		end = count*un;
	} else {                                                         /* tunstall.cpp:195-205 */
		for(uint32_t i = 0; i < un; i++) {
			head[i] = i; eprob[end] = P[i]; eoff[end] = pos; elen[end] = 1; end++;
			buf[pos++] = t->sym[i];
		}
		nwords = un;
	}

	while(nwords < 256) {                                            /* tunstall.cpp:207-241 */
		uint32_t best = 0, maxp = 0;
		for(uint32_t i = 0; i < un; i++) {                           /* first max wins, strict > from 0 */
			uint32_t p = head[i] < TE_CAP ? eprob[head[i]] : 0;
			if(p > maxp) { best = i; maxp = p; }
		}
		uint32_t parent = head[best];
		if(parent >= TE_CAP) break;                                  /* malformed probabilities; reference is UB here */
		uint32_t pp = eprob[parent], po = eoff[parent], pl = elen[parent];
		uint32_t r = 0;
		for(; r < un; r++) {
			if(end >= TE_CAP || pos + pl + 1 > cap) { nwords = 256; break; }
			eprob[end] = (pp*P[r]) >> 16; eoff[end] = pos; elen[end] = pl + 1; end++;
			memmove(buf + pos, buf + po, pl); pos += pl;
			buf[pos++] = t->sym[r];
			if(nwords + r == 255) break;                             /* dictionary full: parent stays */
		}
		if(r == un) head[best] += un;                                /* parent fully expanded: pop it */
		nwords += un - 1;
	}

	uint32_t w = 0;                                                  /* survivors in creation order, tunstall.cpp:243-253 */
	for(uint32_t e = 0; e < end && w < 256; e++) {
		if(head[e % un] > e) continue;
		t->index[w] = eoff[e]; t->length[w] = elen[e]; w++;
	}
	(void)pos;
	uint32_t used = 0;                                               /* bytes any surviving word can reach */
	for(uint32_t i = 0; i < w; i++) if(t->index[i] + t->length[i] > used) used = t->index[i] + t->length[i];
	t->table_size = used;
}

/* src/tunstall.cpp:430-452 */
void co_tunstall_decode(const co_tunstall *t, const uint8_t *in, uint32_t csize, uint8_t *out, uint32_t size) {
	if(t->n == 1) { memset(out, t->sym[0], size); return; }
	if(size == 0 || csize == 0 || t->n == 0) return;
	uint32_t off = 0;
	for(uint32_t j = 0; j + 1 < csize; j++) {
		uint32_t c = in[j], l = t->length[c];
		if(off + l > size) l = size - off;                           /* never on a valid stream */
		memcpy(out + off, t->table + t->index[c], l); off += l;
	}
	uint32_t c = in[csize - 1], l = size - off;                      /* last word: whatever is left */
	if(t->index[c] + l > sizeof(t->table)) l = sizeof(t->table) - t->index[c];
	memcpy(out + off, t->table + t->index[c], l);
}

/* src/bitstream.cpp:103-121 seen as random access: bits [o, o+n) of the MSB-first word stream */
uint32_t co_bits(const uint32_t *w, uint64_t o, uint32_t n) {
	if(n == 0) return 0;
	uint64_t i = o >> 5; uint32_t sh = (uint32_t)(o & 31);
	uint64_t win = (uint64_t)w[i] << 32;
	if(sh + n > 32) win |= w[i+1];
	return (uint32_t)((win << sh) >> (64 - n));
}

/* ------------------------------------------------------------------------------------------------
 * stream blocks */
typedef struct { const uint32_t *words; uint32_t nwords; } bitblock_t;

/* BITS block: u32 nwords | pad to 4 from blob start | words   (cstream.h:283-291) */
static bitblock_t rd_bits(cur_t *c) {
	bitblock_t b = { NULL, 0 };
	uint32_t n = rd32(c);
	size_t pad = c->pos & 3; if(pad) c->pos += 4 - pad;
	if(!need(c, (size_t)n*4)) return b;
	b.words = (const uint32_t *)(c->base + c->pos); b.nwords = n;
	c->pos += (size_t)n*4;
	return b;
}

/* entropy-coded byte array -> malloc'd symbols (cstream.cpp:66-87, 111-128) */
static uint8_t *rd_symbols(cur_t *c, uint32_t entropy, uint32_t *size, co_trace *tr) {
	*size = 0;
	if(entropy == 0) {
		uint32_t n = rd32(c);
		if(!need(c, n)) return NULL;
		uint8_t *o = (uint8_t *)malloc(n ? n : 1);
		memcpy(o, c->base + c->pos, n); c->pos += n; *size = n;
		return o;
	}
	if(entropy != 1) { c->err = E_ENTROPY; return NULL; }
	uint32_t ns = rd8(c);
	if(!need(c, ns*2)) return NULL;
	const uint8_t *probs = c->base + c->pos; c->pos += ns*2;
	uint32_t n = rd32(c), cs = rd32(c);
	if(!need(c, cs)) return NULL;
	const uint8_t *payload = c->base + c->pos; c->pos += cs;
	uint8_t *o = (uint8_t *)malloc(n ? n : 1);
	if(!o) { c->err = E_NOMEM; return NULL; }
	if(n) {
		co_tunstall *t = (co_tunstall *)malloc(sizeof(co_tunstall));
		co_tunstall_build(t, probs, (int)ns);
		co_tunstall_decode(t, payload, cs, o, n);
		free(t);
	}
	if(tr) { tr->tunstall_in += cs; tr->tunstall_out += n; tr->nstreams++; }
	*size = n;
	return o;
}

/* InStream::decodeValues<T> (cstream.h:294-319): component-major; sign folding. T = int32 or uint8. */
static uint32_t decode_values(cur_t *c, uint32_t entropy, void *values, uint32_t N, int is_u8, co_trace *tr) {
	bitblock_t bb = rd_bits(c);
	uint64_t bit = 0; uint32_t nlog = 0;
	for(uint32_t k = 0; k < N; k++) {
		uint8_t *logs = rd_symbols(c, entropy, &nlog, tr);
		if(!logs) return 0;
		if(values) for(uint32_t i = 0; i < nlog; i++) {
			uint32_t d = logs[i]; int32_t v = 0;
			if(d) {
				v = (int32_t)co_bits(bb.words, bit, d); bit += d;
				int32_t mid = (int32_t)(1u << (d - 1));
				if(v < mid) v = -v - mid;
			}
			if(is_u8) ((uint8_t *)values)[(size_t)i*N + k] = (uint8_t)v;
			else ((int32_t *)values)[(size_t)i*N + k] = v;
		}
		free(logs);
	}
	return nlog;
}

/* InStream::decodeArray<int> (cstream.h:324-360): one log per element, N fields of log bits, v = raw - 2^(log-1) */
static uint32_t decode_array(cur_t *c, uint32_t entropy, int32_t *values, uint32_t N, co_trace *tr) {
	bitblock_t bb = rd_bits(c);
	uint32_t nlog = 0;
	uint8_t *logs = rd_symbols(c, entropy, &nlog, tr);
	if(!logs) return 0;
	uint64_t bit = 0;
	if(values) for(uint32_t i = 0; i < nlog; i++) {
		uint32_t d = logs[i];
		for(uint32_t k = 0; k < N; k++) {
			int32_t v = 0;
			/* cstream.h:343 `max = (1<<diff)>>1` in int: at diff == 32 the compiled reference (x86-64 and AArch64 shifters take the count
			   mod 32) gets (1<<0)>>1 = 0, not 2^31; at diff == 31 it shifts INT_MIN right arithmetically: -2^30 (the encoder added +2^30,
			   cstream.h:157 - upstream does not round-trip such values; its bytes are the contract) - both checked against oracle/_ref in
			   tests/test_oracle_vs_reference.py and pinned by the fixtures fields32 / fields31 */
			if(d) { v = (int32_t)(co_bits(bb.words, bit, d) - (uint32_t)((int32_t)(1u << (d & 31u)) >> 1)); bit += d; }
			values[(size_t)i*N + k] = v;
		}
	}
	free(logs);
	return nlog;
}

/* ------------------------------------------------------------------------------------------------
 * CLERS automaton — src/decoder.cpp:204-358.  One call per group; state shared across groups lives in topo_t. */
enum { VERTEX = 0, LEFT = 1, RIGHT = 2, END = 3, BOUNDARY = 4, DELAY = 5, SPLIT = 6 };

typedef struct {
	const uint8_t *clers; uint32_t nclers, cler;
	bitblock_t bits; uint64_t bit;
	uint32_t nvert, vertex_count, max_front;
	uint32_t *pred;                 /* nvert*3 */
	uint32_t *f32; uint16_t *f16;
	uint32_t front_size;
} topo_t;

static int ilog2u(uint64_t p) { int k = 0; while(p >>= 1) k++; return k; }  /* cstream.cpp:31-35 */

static uint32_t topo_bits(topo_t *t, uint32_t n, int *err) {
	if(t->bit + n > (uint64_t)t->bits.nwords*32) { *err = E_TOPOLOGY; return 0; }
	uint32_t v = co_bits(t->bits.words, t->bit, n); t->bit += n; return v;
}
static void put_face(topo_t *t, uint32_t at, uint32_t a, uint32_t b, uint32_t c) {
	if(t->f16) { t->f16[at] = (uint16_t)a; t->f16[at+1] = (uint16_t)b; t->f16[at+2] = (uint16_t)c; }
	else if(t->f32) { t->f32[at] = a; t->f32[at+1] = b; t->f32[at+2] = c; }
}

static int decode_faces(topo_t *t, uint32_t start, uint32_t end) {
	size_t cap = (size_t)t->max_front + 8;
	uint32_t *v0 = malloc(cap*4), *v1 = malloc(cap*4), *v2 = malloc(cap*4), *prv = malloc(cap*4), *nxt = malloc(cap*4);
	uint8_t *dead = calloc(cap, 1);
	uint32_t *order = malloc(cap*4), *delayed = malloc(cap*4);
	if(!v0 || !v1 || !v2 || !prv || !nxt || !dead || !order || !delayed) return E_NOMEM;
	uint32_t nfront = 0, norder = 0, iorder = 0, ndelayed = 0;
	const int splitbits = ilog2u(t->nvert) + 1;
	int64_t new_edge = -1;
	int err = 0;
#define PUSH_EDGE(A,B,C,P,N) do { if(nfront >= cap) { err = E_TOPOLOGY; goto done; } \
	v0[nfront]=(A); v1[nfront]=(B); v2[nfront]=(C); prv[nfront]=(P); nxt[nfront]=(N); dead[nfront]=0; nfront++; } while(0)
#define CHK(i) do { if((i) >= nfront) { err = E_TOPOLOGY; goto done; } } while(0)

	while(start < end) {
		if(new_edge == -1 && iorder >= norder && ndelayed == 0) {      /* seed face: decoder.cpp:224-259 */
			if(t->cler >= t->nclers) { err = E_TOPOLOGY; goto done; }
			uint32_t last = t->vertex_count - 1, vi[3], split = 0;
			uint32_t c = t->clers[t->cler++];
			if(c == SPLIT) split = topo_bits(t, 3, &err);
			for(int k = 0; k < 3; k++) {
				uint32_t v;
				if(split & (1u << k)) v = topo_bits(t, (uint32_t)splitbits, &err);
				else {
					if(t->vertex_count >= t->nvert) { err = E_TOPOLOGY; goto done; }
					uint32_t *p = t->pred + (size_t)t->vertex_count*3; p[0] = p[1] = p[2] = last;
					last = v = t->vertex_count++;
				}
				vi[k] = v;
			}
			if(err) goto done;
			put_face(t, start, vi[0], vi[1], vi[2]); start += 3;
			uint32_t e = nfront;
			if(norder + 3 > cap) { err = E_TOPOLOGY; goto done; }
			order[norder++] = e;     PUSH_EDGE(vi[1], vi[2], vi[0], e + 2, e + 1);
			order[norder++] = e + 1; PUSH_EDGE(vi[2], vi[0], vi[1], e + 0, e + 2);
			order[norder++] = e + 2; PUSH_EDGE(vi[0], vi[1], vi[2], e + 1, e + 0);
			continue;
		}
		uint32_t f;
		if(new_edge != -1) { f = (uint32_t)new_edge; new_edge = -1; }
		else if(iorder < norder) f = order[iorder++];
		else f = delayed[--ndelayed];
		CHK(f);
		if(dead[f]) continue;                                          /* no symbol consumed, decoder.cpp:278-279 */
		if(t->cler >= t->nclers) { err = E_TOPOLOGY; goto done; }
		uint32_t c = t->clers[t->cler++];
		if(c == BOUNDARY) continue;

		uint32_t a = v0[f], b = v1[f], ep = prv[f], en = nxt[f], opp;
		CHK(ep); CHK(en);
		uint32_t ne = nfront;
		new_edge = ne;
		if(c == VERTEX || c == SPLIT) {                                /* decoder.cpp:294-309 */
			if(c == SPLIT) { opp = topo_bits(t, (uint32_t)splitbits, &err); if(err) goto done; }
			else {
				if(t->vertex_count >= t->nvert) { err = E_TOPOLOGY; goto done; }
				uint32_t *p = t->pred + (size_t)t->vertex_count*3; p[0] = b; p[1] = a; p[2] = v2[f];
				opp = t->vertex_count++;
			}
			nxt[ep] = ne; prv[en] = ne + 1;
			PUSH_EDGE(a, opp, b, ep, ne + 1);
			if(norder >= cap) { err = E_TOPOLOGY; goto done; }
			order[norder++] = nfront;
			PUSH_EDGE(opp, b, a, ne, en);
		} else if(c == LEFT) {                                         /* decoder.cpp:311-317 */
			uint32_t pp = prv[ep]; CHK(pp);
			dead[ep] = 1; nxt[pp] = ne; prv[en] = ne; opp = v0[ep];
			PUSH_EDGE(opp, b, a, pp, en);
		} else if(c == RIGHT) {                                        /* decoder.cpp:319-325 */
			uint32_t nn = nxt[en]; CHK(nn);
			dead[en] = 1; prv[nn] = ne; nxt[ep] = ne; opp = v1[en];
			PUSH_EDGE(a, opp, b, ep, nn);
		} else if(c == DELAY) {                                        /* decoder.cpp:327-331 */
			if(ndelayed >= cap) { err = E_TOPOLOGY; goto done; }
			delayed[ndelayed++] = f; new_edge = -1; continue;
		} else if(c == END) {                                          /* decoder.cpp:333-339 */
			uint32_t pp = prv[ep], nn = nxt[en]; CHK(pp); CHK(nn);
			dead[ep] = 1; dead[en] = 1; nxt[pp] = nn; prv[nn] = pp; opp = v0[ep];
			new_edge = -1;
		} else { err = E_TOPOLOGY; goto done; }
		put_face(t, start, b, a, opp); start += 3;                     /* decoder.cpp:348-356 */
	}
done:
	t->front_size = nfront;
	free(v0); free(v1); free(v2); free(prv); free(nxt); free(dead); free(order); free(delayed);
	return err;
#undef PUSH_EDGE
#undef CHK
}

/* ------------------------------------------------------------------------------------------------
 * float recipes. x86 cvttss2si semantics made explicit. */
static int32_t f2i(float x) {
	if(!(x > -2147483904.0f && x < 2147483648.0f)) return INT_MIN;
	return (int32_t)x;
}
static float norm3(float x, float y, float z) {                      /* include/corto/point.h:111 */
	float s = x*x + y*y; s = s + z*z;
	return (float)sqrt((double)s);
}
/* include/corto/normal_attribute.h:75-85 */
static void to_octa(const float v[3], int32_t unit, int32_t o[2]) {
	float s = fabsf(v[0]) + fabsf(v[1]); s = s + fabsf(v[2]);
	float px = v[0]/s, py = v[1]/s;
	if(v[2] < 0) {
		float qx = 1.0f - fabsf(py), qy = 1.0f - fabsf(px);
		px = qx; py = qy;
		if(v[0] < 0) px = -px;
		if(v[1] < 0) py = -py;
	}
	o[0] = f2i(px*(float)unit); o[1] = f2i(py*(float)unit);
}
/* include/corto/normal_attribute.h:104-112; x,y already reduced to the caller's integer type */
static void to_sphere(int32_t x, int32_t y, int32_t unit, float n[3]) {
	int32_t ax = (int32_t)((uint32_t)(x < 0 ? -(uint32_t)x : (uint32_t)x)), ay = (int32_t)((uint32_t)(y < 0 ? -(uint32_t)y : (uint32_t)y));
	int32_t z = (int32_t)((uint32_t)unit - (uint32_t)ax - (uint32_t)ay);
	n[0] = (float)x; n[1] = (float)y; n[2] = (float)z;
	if(n[2] < 0) {
		n[0] = (float)(int32_t)((x > 0 ? 1u : 0xFFFFFFFFu)*((uint32_t)unit - (uint32_t)ay));
		n[1] = (float)(int32_t)((y > 0 ? 1u : 0xFFFFFFFFu)*((uint32_t)unit - (uint32_t)ax));
	}
	float l = norm3(n[0], n[1], n[2]);
	n[0] /= l; n[1] /= l; n[2] /= l;
}
/* float -> int16 as gcc/x86 does it: cvttss2si to 32 bits, keep the low 16 */
static int16_t f2s(float x) { return (int16_t)(uint16_t)(uint32_t)f2i(x); }

/* ------------------------------------------------------------------------------------------------ */
typedef struct {
	co_attr_info info;
	const co_binding *bind;           /* NULL when unbound */
	int32_t *normal_diffs;            /* normals: 2*nvert ints (NormalAttr::diffs) */
	uint32_t normal_prediction, normal_ndiffs;
	uint32_t qc[4];
} attr_t;

static const co_binding *find_binding(const co_outputs *o, const char *name) {
	if(!o) return NULL;
	for(uint32_t i = 0; i < o->nbind; i++)
		if(o->bind[i].buffer && strcmp(o->bind[i].name, name) == 0) return &o->bind[i];
	return NULL;
}

/* attr->decode(nvert, stream): vertex_attribute.h:153-158, normal_attribute.cpp:178-185, color_attribute.h:55-59 */
static void attr_decode(attr_t *a, cur_t *c, const co_header *h, co_trace *tr, int slot) {
	void *buf = a->bind ? a->bind->buffer : NULL;
	if(a->info.codec == CO_CODEC_NORMAL) {
		a->normal_prediction = rd8(c);
		a->normal_diffs = (int32_t *)calloc((size_t)h->nvert*2 + 2, 4);
		a->normal_ndiffs = decode_array(c, h->entropy, a->normal_diffs, 2, tr);
		if(tr) tr->normal_ndiffs = a->normal_ndiffs;
		if(tr && tr->attr_raw[slot]) memcpy(tr->attr_raw[slot], a->normal_diffs, (size_t)a->normal_ndiffs*8);
	} else if(a->info.codec == CO_CODEC_COLOR) {
		for(uint32_t k = 0; k < a->info.N; k++) { uint32_t q = rd8(c); if(k < 4) a->qc[k] = q; }
		decode_values(c, h->entropy, buf, a->info.N, 1, tr);
		if(tr && tr->attr_raw[slot] && buf)
			for(size_t i = 0; i < (size_t)h->nvert*a->info.N; i++) tr->attr_raw[slot][i] = ((uint8_t *)buf)[i];
	} else {
		if(a->info.strategy & CO_CORRELATED) decode_array(c, h->entropy, (int32_t *)buf, a->info.N, tr);
		else decode_values(c, h->entropy, buf, a->info.N, 0, tr);
		if(tr && tr->attr_raw[slot] && buf) memcpy(tr->attr_raw[slot], buf, (size_t)h->nvert*a->info.N*4);
	}
}

/* attr->deltaDecode: vertex_attribute.h:160-182, normal_attribute.cpp:187-208 */
static void attr_delta(attr_t *a, uint32_t nvert, const uint32_t *pred /* NULL for clouds */) {
	if(!a->bind) return;
	uint32_t N = a->info.N;
	if(a->info.codec == CO_CODEC_NORMAL) {
		if(a->normal_prediction != CO_NORMAL_DIFF) return;
		uint32_t *d = (uint32_t *)a->normal_diffs;
		if(pred) for(uint32_t i = 1; i < nvert; i++) { uint32_t p = pred[(size_t)i*3]; d[2*i] += d[2*p]; d[2*i+1] += d[2*p+1]; }
		else for(size_t i = 2; i < (size_t)nvert*2; i++) d[i] += d[i-2];
		return;
	}
	if(a->info.codec == CO_CODEC_COLOR) {                              /* GenericAttr<uchar>: wraps mod 256 */
		uint8_t *v = (uint8_t *)a->bind->buffer;
		if(pred && (a->info.strategy & CO_PARALLEL))
			for(uint32_t i = 1; i < nvert; i++) { const uint32_t *p = pred + (size_t)i*3;
				for(uint32_t k = 0; k < N; k++) v[(size_t)i*N+k] += (uint8_t)(v[(size_t)p[0]*N+k] + v[(size_t)p[1]*N+k] - v[(size_t)p[2]*N+k]); }
		else if(pred)
			for(uint32_t i = 1; i < nvert; i++) { uint32_t p = pred[(size_t)i*3];
				for(uint32_t k = 0; k < N; k++) v[(size_t)i*N+k] += v[(size_t)p*N+k]; }
		else for(size_t i = N; i < (size_t)nvert*N; i++) v[i] += v[i-N];
		return;
	}
	uint32_t *v = (uint32_t *)a->bind->buffer;                        /* int32 wrap-around arithmetic */
	if(pred && (a->info.strategy & CO_PARALLEL))
		for(uint32_t i = 1; i < nvert; i++) { const uint32_t *p = pred + (size_t)i*3;
			for(uint32_t k = 0; k < N; k++) v[(size_t)i*N+k] += v[(size_t)p[0]*N+k] + v[(size_t)p[1]*N+k] - v[(size_t)p[2]*N+k]; }
	else if(pred)
		for(uint32_t i = 1; i < nvert; i++) { uint32_t p = pred[(size_t)i*3];
			for(uint32_t k = 0; k < N; k++) v[(size_t)i*N+k] += v[(size_t)p*N+k]; }
	else for(size_t i = N; i < (size_t)nvert*N; i++) v[i] += v[i-N];
}

/* NormalAttr::postDelta (normal_attribute.cpp:210-255) with estimateNormals :40-59, markBoundary :24-37,
 * computeNormals :281-325 */
static int normal_post(attr_t *a, const attr_t *position, uint32_t nvert, uint32_t nface, const co_outputs *o) {
	if(!a->bind || a->normal_prediction == CO_NORMAL_DIFF) return 0;
	if(!position) return E_NORMAL_NEEDS_POSITION;
	if(a->bind->format != CO_FMT_FLOAT && a->bind->format != CO_FMT_INT16) return E_FORMAT;
	const int32_t *coords = position->bind ? (const int32_t *)position->bind->buffer : NULL;
	if(!coords) return E_NORMAL_NEEDS_POSITION;   /* the reference dereferences a null buffer here */
	float *est = (float *)calloc((size_t)nvert*3 + 3, 4);
	int32_t *bnd = (int32_t *)calloc((size_t)nvert + 1, 4);
	for(uint32_t f = 0; f < nface; f++) {
		uint32_t i0, i1, i2;
		if(o->index32) { i0 = o->index32[3*(size_t)f]; i1 = o->index32[3*(size_t)f+1]; i2 = o->index32[3*(size_t)f+2]; }
		else { i0 = o->index16[3*(size_t)f]; i1 = o->index16[3*(size_t)f+1]; i2 = o->index16[3*(size_t)f+2]; }
		const int32_t *p0 = coords + 3*(size_t)i0, *p1 = coords + 3*(size_t)i1, *p2 = coords + 3*(size_t)i2;
		float ax = (float)p1[0] - (float)p0[0], ay = (float)p1[1] - (float)p0[1], az = (float)p1[2] - (float)p0[2];
		float bx = (float)p2[0] - (float)p0[0], by = (float)p2[1] - (float)p0[1], bz = (float)p2[2] - (float)p0[2];
		float n[3] = { ay*bz - az*by, az*bx - ax*bz, ax*by - ay*bx };
		for(int k = 0; k < 3; k++) { est[3*(size_t)i0+k] += n[k]; }
		for(int k = 0; k < 3; k++) { est[3*(size_t)i1+k] += n[k]; }
		for(int k = 0; k < 3; k++) { est[3*(size_t)i2+k] += n[k]; }
		if(a->normal_prediction == CO_NORMAL_BORDER) {
			bnd[i0] ^= (int32_t)i1; bnd[i0] ^= (int32_t)i2;
			bnd[i1] ^= (int32_t)i2; bnd[i1] ^= (int32_t)i0;
			bnd[i2] ^= (int32_t)i0; bnd[i2] ^= (int32_t)i1;
		}
	}
	const int32_t unit = f2i(a->info.q);
	uint32_t count = 0;
	for(uint32_t i = 0; i < nvert; i++) {
		float *e = est + 3*(size_t)i;
		int corrected = a->normal_prediction == CO_NORMAL_ESTIMATED || bnd[i] != 0;
		if(a->bind->format == CO_FMT_FLOAT) {
			float *n = (float *)a->bind->buffer + 3*(size_t)i;
			if(corrected) {
				int32_t qn[2]; to_octa(e, unit, qn);
				const int32_t *d = a->normal_diffs + 2*(size_t)count++;
				to_sphere((int32_t)((uint32_t)qn[0] + (uint32_t)d[0]), (int32_t)((uint32_t)qn[1] + (uint32_t)d[1]), unit, n);
			} else {
				float l = norm3(e[0], e[1], e[2]);
				n[0] = e[0]/l; n[1] = e[1]/l; n[2] = e[2]/l;
			}
		} else {
			int16_t *n = (int16_t *)a->bind->buffer + 3*(size_t)i;
			if(corrected) {
				int32_t qn[2]; to_octa(e, unit, qn);
				const int32_t *d = a->normal_diffs + 2*(size_t)count++;
				/* Point2s(qn[0]+d[0], qn[1]+d[1]): truncated to int16 before toSphere (normal_attribute.cpp:293) */
				int16_t sx = (int16_t)(uint16_t)((uint32_t)qn[0] + (uint32_t)d[0]), sy = (int16_t)(uint16_t)((uint32_t)qn[1] + (uint32_t)d[1]);
				float s[3]; to_sphere(sx, sy, unit, s);
				n[0] = f2s(s[0]*32767); n[1] = f2s(s[1]*32767); n[2] = f2s(s[2]*32767);
			} else {
				float l = norm3(e[0], e[1], e[2]);
				if(!(l < 0.00001f)) {                                  /* else: output left unwritten (:296-297) */
					l = 32767.0f/l;
					n[0] = f2s(e[0]*l); n[1] = f2s(e[1]*l); n[2] = f2s(e[2]*l);
				}
			}
		}
	}
	free(est); free(bnd);
	return 0;
}

/* dequantize: vertex_attribute.h:184-230 (FLOAT only), normal_attribute.cpp:257-279, color_attribute.cpp:72-95 */
static int attr_dequantize(attr_t *a, uint32_t nvert) {
	if(!a->bind) return 0;
	if(a->info.codec == CO_CODEC_NORMAL) {
		if(a->normal_prediction != CO_NORMAL_DIFF) return 0;
		const int32_t unit = f2i(a->info.q);
		if(a->bind->format == CO_FMT_FLOAT)
			for(uint32_t i = 0; i < nvert; i++)
				to_sphere(a->normal_diffs[2*(size_t)i], a->normal_diffs[2*(size_t)i+1], unit, (float *)a->bind->buffer + 3*(size_t)i);
		else if(a->bind->format == CO_FMT_INT16)
			for(uint32_t i = 0; i < nvert; i++) {
				float s[3]; int16_t *n = (int16_t *)a->bind->buffer + 3*(size_t)i;
				to_sphere((int16_t)(uint16_t)(uint32_t)a->normal_diffs[2*(size_t)i], (int16_t)(uint16_t)(uint32_t)a->normal_diffs[2*(size_t)i+1], unit, s);
				n[0] = f2s(s[0]*32767); n[1] = f2s(s[1]*32767); n[2] = f2s(s[2]*32767);
			}
		else return E_FORMAT;
		return 0;
	}
	if(a->info.codec == CO_CODEC_COLOR) {
		if(a->bind->format != CO_FMT_UINT8) return E_FORMAT;
		uint32_t N = a->info.N, oc = a->bind->out_components ? a->bind->out_components : 4;
		uint8_t *base = (uint8_t *)a->bind->buffer;
		for(size_t i = nvert; i-- > 0; ) {                             /* in place, back to front */
			uint8_t col[4] = { 0, 0, 0, 255 };
			for(uint32_t k = 0; k < N && k < 4; k++) col[k] = base[i*N + k];
			uint8_t rgb[4] = { (uint8_t)(col[2] + col[0]), col[0], (uint8_t)(col[1] + col[0]), col[3] };  /* point.h:214 */
			for(uint32_t k = 0; k < oc && k < 4; k++) base[i*oc + k] = (uint8_t)(rgb[k]*a->qc[k]);
		}
		return 0;
	}
	/* GenericAttr<int>::dequantize, vertex_attribute.h:184-230.  The values sit in the buffer as n = nvert*N int32.  FLOAT is the
	 * only format upstream's own callers use; the others are restated as the reference library BEHAVES when built as oracle/Makefile
	 * builds it (g++ x86-64 -O2: scalar loops, cvttss2si conversions) - "buffer[i] *= q" through a pointer of the OUTPUT type over the
	 * first n elements OF THAT TYPE, i.e. over the first 2n (n) bytes of the int32 array for the 16-bit (8-bit) formats, and DOUBLE
	 * widening in place front to back, so that element i >= 1 is computed from bytes an earlier store of the same loop wrote.
	 * (Accessing the int array through these pointer types is undefined behaviour upstream; this is what the compiled code does,
	 * pinned by tests/golden/generic_formats.npz, which the reference itself produced.) */
	const size_t n = (size_t)nvert*a->info.N;
	const float q = a->info.q;
	uint8_t *b8 = (uint8_t *)a->bind->buffer;
	switch(a->bind->format) {
	case CO_FMT_FLOAT: {
		int32_t *v = (int32_t *)a->bind->buffer; float *f = (float *)a->bind->buffer;
		for(size_t i = 0; i < n; i++) f[i] = (float)v[i]*q;
		return 0; }
	case CO_FMT_INT32: case CO_FMT_UINT32:                           /* ((uint32_t *)buffer)[i] *= q : u32 -> float -> x q -> cvttss2si (64-bit) -> low 32 bits */
		for(size_t i = 0; i < n; i++) {
			uint32_t u; memcpy(&u, b8 + 4*i, 4);
			const float f = (float)u*q;
			const int64_t w = (f >= -9223372036854775808.0f && f < 9223372036854775808.0f) ? (int64_t)f : INT64_MIN;
			u = (uint32_t)(uint64_t)w; memcpy(b8 + 4*i, &u, 4);
		}
		return 0;
	case CO_FMT_INT16: case CO_FMT_UINT16:                           /* ((uint16_t *)buffer)[i] *= q : u16 -> int -> float -> x q -> cvttss2si (32-bit) -> low 16 bits */
		for(size_t i = 0; i < n; i++) {
			uint16_t u; memcpy(&u, b8 + 2*i, 2);
			u = (uint16_t)(uint32_t)f2i((float)(int32_t)u*q); memcpy(b8 + 2*i, &u, 2);
		}
		return 0;
	case CO_FMT_INT8: case CO_FMT_UINT8:                             /* ((char *)buffer)[i] *= q : char is signed on x86 */
		for(size_t i = 0; i < n; i++) b8[i] = (uint8_t)(uint32_t)f2i((float)(int32_t)(int8_t)b8[i]*q);
		return 0;
	case CO_FMT_DOUBLE:                                              /* ((double *)buffer)[i] = coords[i]*q, i ascending, in place */
		for(size_t i = 0; i < n; i++) {
			int32_t c; memcpy(&c, b8 + 4*i, 4);
			const double d = (double)((float)c*q); memcpy(b8 + 8*i, &d, 8);
		}
		return 0;
	}
	return E_FORMAT;
}

int co_decode(const uint8_t *blob, size_t len, const co_outputs *o, co_trace *tr) {
	cur_t c = { blob, len, 0, 0 };
	co_header h;
	int err = parse_header(&c, &h);
	if(err) return err;
	if(tr) { tr->nclers = 0; tr->tunstall_in = tr->tunstall_out = 0; tr->nstreams = 0; tr->normal_ndiffs = 0; }

	attr_t attrs[CO_MAX_ATTRS]; memset(attrs, 0, sizeof(attrs));
	attr_t *position = NULL;
	for(uint32_t i = 0; i < h.nattr; i++) {
		attrs[i].info = h.attr[i];
		if(attrs[i].info.codec != CO_CODEC_NORMAL && attrs[i].info.codec != CO_CODEC_COLOR) attrs[i].info.codec = CO_CODEC_GENERIC; /* decoder.cpp:77-79 */
		if(attrs[i].info.codec == CO_CODEC_NORMAL) attrs[i].info.N = 3;        /* NormalAttr(): N = 3 */
		attrs[i].bind = find_binding(o, h.attr[i].name);
		attrs[i].qc[0] = attrs[i].qc[1] = attrs[i].qc[2] = 4; attrs[i].qc[3] = 8;   /* color_attribute.h:31-34 */
		if(strcmp(h.attr[i].name, "position") == 0 && attrs[i].info.codec == CO_CODEC_GENERIC) position = &attrs[i];
	}

	/* groups: index_attribute.h:89-99 */
	uint32_t ngroups = rd32(&c);
	if(!need(&c, (size_t)ngroups*5)) return E_TRUNC;
	uint32_t *gend = (uint32_t *)malloc(((size_t)ngroups + 1)*4);
	for(uint32_t g = 0; g < ngroups; g++) {
		gend[g] = rd32(&c);
		uint32_t np = rd8(&c);
		for(uint32_t k = 0; k < np; k++) { uint32_t n; rdstr(&c, &n); rdstr(&c, &n); }
	}

	uint32_t *pred = NULL; uint8_t *clers = NULL;
	topo_t T; memset(&T, 0, sizeof(T));
	if(h.nface > 0) {                                                 /* decodeMesh: decoder.cpp:164-196 */
		T.max_front = rd32(&c);                                        /* index_attribute.h:83-87 */
		clers = rd_symbols(&c, h.entropy, &T.nclers, tr);
		T.bits = rd_bits(&c);
		if(c.err) { err = c.err; goto out; }
		T.clers = clers;
		if(tr) { tr->max_front = T.max_front; tr->nclers = T.nclers;
			if(tr->clers) memcpy(tr->clers, clers, T.nclers < tr->nclers_cap ? T.nclers : tr->nclers_cap); }
	}
	for(uint32_t i = 0; i < h.nattr; i++) attr_decode(&attrs[i], &c, &h, tr, (int)i);
	if(c.err) { err = c.err; goto out; }

	if(h.nface > 0) {
		pred = (uint32_t *)calloc((size_t)h.nvert*3 + 3, 4);
		T.nvert = h.nvert; T.pred = pred;
		T.f32 = o ? o->index32 : NULL; T.f16 = o ? o->index16 : NULL;
		uint32_t start = 0;
		for(uint32_t g = 0; g < ngroups; g++) {
			err = decode_faces(&T, start*3, gend[g]*3);
			if(err) goto out;
			start = gend[g];
		}
		if(tr) { tr->front_size = T.front_size; if(tr->prediction) memcpy(tr->prediction, pred, (size_t)h.nvert*12); }
	}
	for(uint32_t i = 0; i < h.nattr; i++) {
		attr_delta(&attrs[i], h.nvert, h.nface > 0 ? pred : NULL);
		if(tr && tr->attr_delta[i] && attrs[i].bind) {
			if(attrs[i].info.codec == CO_CODEC_NORMAL) memcpy(tr->attr_delta[i], attrs[i].normal_diffs, (size_t)h.nvert*8);
			else if(attrs[i].info.codec == CO_CODEC_COLOR) for(size_t k = 0; k < (size_t)h.nvert*attrs[i].info.N; k++) tr->attr_delta[i][k] = ((uint8_t *)attrs[i].bind->buffer)[k];
			else memcpy(tr->attr_delta[i], attrs[i].bind->buffer, (size_t)h.nvert*attrs[i].info.N*4);
		}
	}
	if(h.nface > 0)                                                   /* postDelta is NOT run for clouds, decoder.cpp:142-143 */
		for(uint32_t i = 0; i < h.nattr; i++)
			if(attrs[i].info.codec == CO_CODEC_NORMAL) { err = normal_post(&attrs[i], position, h.nvert, h.nface, o); if(err) goto out; }
	for(uint32_t i = 0; i < h.nattr; i++) { err = attr_dequantize(&attrs[i], h.nvert); if(err) goto out; }
out:
	for(uint32_t i = 0; i < h.nattr; i++) free(attrs[i].normal_diffs);
	free(gend); free(pred); free(clers);
	return err;
}
