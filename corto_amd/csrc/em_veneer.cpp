// em_veneer.cpp — newDecoder ... deleteDecoder over the crt::Decoder facade
// (the ABI of upstream's wasm module, html/js/emscripten/emcorto.cpp:14-89; see include/corto/emcorto.h).
#include "corto/decoder.h"
#include "corto/emcorto.h"

#define EM_API extern "C" __attribute__((visibility("default")))

using crt::Decoder;

EM_API Decoder *newDecoder(int n, const unsigned char *buffer) {
	try { return new Decoder(n, buffer); } catch(const char *) { return nullptr; }
}
EM_API int ngroups(Decoder *d) { return d ? (int)d->index.groups.size() : 0; }
EM_API void groups(Decoder *d, int *out) {
	if(!d || !out) return;
	for(size_t i = 0; i < d->index.groups.size(); i++) out[i] = (int)d->index.groups[i].end;
}
EM_API int nvert(Decoder *d) { return d ? (int)d->nvert : 0; }
EM_API int nface(Decoder *d) { return d ? (int)d->nface : 0; }
EM_API bool hasAttr(Decoder *d, const char *attr) { return d && attr && d->hasAttr(attr); }
EM_API bool hasNormal(Decoder *d) { return d && d->hasAttr("normal"); }
EM_API bool hasColor(Decoder *d) { return d && d->hasAttr("color"); }
EM_API bool hasUv(Decoder *d) { return d && d->hasAttr("uv"); }
EM_API void setPositions(Decoder *d, float *buffer) { if(d) d->setPositions(buffer); }
EM_API void setNormals32(Decoder *d, float *buffer) { if(d) d->setNormals(buffer); }
EM_API void setNormals16(Decoder *d, int16_t *buffer) { if(d) d->setNormals(buffer); }
EM_API void setColors(Decoder *d, unsigned char *buffer, int components) { if(d) d->setColors(buffer, components); }
EM_API void setUvs(Decoder *d, float *buffer) { if(d) d->setUvs(buffer); }
EM_API void setIndex16(Decoder *d, uint16_t *buffer) { if(d) d->setIndex(buffer); }
EM_API void setIndex32(Decoder *d, uint32_t *buffer) { if(d) d->setIndex(buffer); }
EM_API void decode(Decoder *d) {
	if(!d) return;
	try { d->decode(); } catch(const char *) {}
}
EM_API void deleteDecoder(Decoder *d) { delete d; }
