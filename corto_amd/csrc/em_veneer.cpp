// em_veneer.cpp — newDecoder ... deleteDecoder over the crt::Decoder facade
// (the ABI of upstream's wasm module, html/js/emscripten/emcorto.cpp:14-89; see include/corto/emcorto.h).
#include "corto/decoder.h"
#include "corto/emcorto.h"

#include <cstring>
#include <mutex>
#include <unordered_map>

#include "corto_hip.h"

#define EM_API extern "C" __attribute__((visibility("default")))

using crt::Decoder;

namespace {
// what went wrong last, per decoder (and per thread for a constructor that failed: there is no decoder to ask then).  The facade throws the
// reference's string literals (src/decoder.cpp:44,51,274 ...); the code is the CRTHIP_E_* whose message that is.
std::mutex g_m;
std::unordered_map<const Decoder *, int> g_err;
thread_local int g_ctor_err = 0;
int code_of(const char *msg) {
	for(int c = -1; c >= CRTHIP_E_LIMIT; c--) if(msg && !strcmp(msg, crthip_strerror(c))) return c;
	return CRTHIP_E_ARGUMENT;                                   // a message with no code of its own (custom codecs, "decode failed")
}
void note(const Decoder *d, int code) { std::lock_guard<std::mutex> lock(g_m); if(code) g_err[d] = code; else g_err.erase(d); }
}

EM_API Decoder *newDecoder(int n, const unsigned char *buffer) {
	g_ctor_err = 0;
	try { return new Decoder(n, buffer); } catch(const char *msg) { g_ctor_err = code_of(msg); return nullptr; }
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
	try { d->decode(); note(d, 0); } catch(const char *msg) { note(d, code_of(msg)); }
}
EM_API int lastError(Decoder *d) {
	if(!d) return g_ctor_err;
	std::lock_guard<std::mutex> lock(g_m);
	auto it = g_err.find(d);
	return it == g_err.end() ? 0 : it->second;
}
EM_API void deleteDecoder(Decoder *d) { note(d, 0); delete d; }
