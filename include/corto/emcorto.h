/* emcorto.h — the flat C ABI corto's JavaScript loader binds, served by the MI355X decoder.
 *
 * Replaces upstream html/js/emscripten/emcorto.cpp:14-89 (the EMSCRIPTEN_KEEPALIVE set that corto.em.js / the
 * three.js CORTOLoader call through cwrap): same eighteen symbol names, same argument order and meaning, over the
 * crt::Decoder facade (include/corto/decoder.h).  Built by `python -m corto_amd.build` as libcorto_em_hip.so - a
 * library of its own because the names (`decode`, `nvert`, `groups` ...) are too generic to share a namespace with
 * anything else.  A server-side node process binds it through N-API / ffi exactly as the browser binds the wasm
 * module.
 *
 * Differences from upstream, at the edges only:
 *   - exceptions do not cross the C boundary: newDecoder returns NULL for a blob the reference would throw on,
 *     decode() returns without writing for a decode error - and so that a failed decode is not mistaken for a successful
 *     one, ONE symbol is added to upstream's eighteen: lastError() below.
 *   - every entry point tolerates a NULL decoder (returns 0 / false / does nothing).
 *   - setColors has no default argument in C; pass 4 for RGBA (upstream's C++ default, emcorto.cpp:67).
 */
#ifndef CORTO_HIP_EMCORTO_H
#define CORTO_HIP_EMCORTO_H

#include <stdint.h>

#ifdef __cplusplus
namespace crt { class Decoder; }
typedef crt::Decoder crt_decoder;
extern "C" {
#else
#include <stdbool.h>
typedef struct crt_decoder crt_decoder;
#endif

crt_decoder *newDecoder(int n, const unsigned char *buffer);      /* emcorto.cpp:14-16; buffer borrowed, 4-byte aligned */
int ngroups(crt_decoder *decoder);                                /* :18-20 */
void groups(crt_decoder *decoder, int *groups);                   /* :22-27  groups[i] = end face of group i */
int nvert(crt_decoder *decoder);                                  /* :29-31 */
int nface(crt_decoder *decoder);                                  /* :33-35 */
bool hasAttr(crt_decoder *decoder, const char *attr);             /* :37-39 */
bool hasNormal(crt_decoder *decoder);                             /* :41-43 */
bool hasColor(crt_decoder *decoder);                              /* :45-47 */
bool hasUv(crt_decoder *decoder);                                 /* :49-51 */
void setPositions(crt_decoder *decoder, float *buffer);           /* :55-57 */
void setNormals32(crt_decoder *decoder, float *buffer);           /* :59-61 */
void setNormals16(crt_decoder *decoder, int16_t *buffer);         /* :63-65 */
void setColors(crt_decoder *decoder, unsigned char *buffer, int components);   /* :67-69 */
void setUvs(crt_decoder *decoder, float *buffer);                 /* :71-73 */
void setIndex16(crt_decoder *decoder, uint16_t *buffer);          /* :75-77 */
void setIndex32(crt_decoder *decoder, uint32_t *buffer);          /* :79-81 */
void decode(crt_decoder *decoder);                                /* :83-85  one-shot, like upstream */
void deleteDecoder(crt_decoder *decoder);                         /* :87-89 */
/* Not upstream's: 0 if the decoder's last decode() succeeded (or none ran), else the CRTHIP_E_* code (include/corto_hip.h) of what
 * upstream would have thrown - CRTHIP_E_TOPOLOGY for "Decoding topology failed", CRTHIP_E_DEVICE without a GPU ...;
 * lastError(NULL) = why this thread's last newDecoder() returned NULL (CRTHIP_E_MAGIC for "Not a crt file." ...). */
int lastError(crt_decoder *decoder);

#ifdef __cplusplus
}
#endif
#endif
