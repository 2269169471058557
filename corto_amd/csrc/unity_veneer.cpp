// unity_veneer.cpp — CreateDecoder / DestroyDecoder / DecodeMesh over the crt::Decoder facade
// (the ABI of upstream's cortocodec_unity, src/corto_codec.cpp:6-59; see include/corto/corto_codec.h).
#include "corto/corto_codec.h"

#include <cstdint>
#include <vector>

namespace crt {

#define VENEER_API extern "C" __attribute__((visibility("default")))

VENEER_API Decoder *CreateDecoder(int length, unsigned char *data, Vector2 *decoderInfo) {
	Decoder *d = nullptr;
	try { d = new Decoder(length, data); } catch(const char *) { return nullptr; }
	if(decoderInfo) { decoderInfo[0].x = (float)d->nface; decoderInfo[0].y = (float)d->nvert; }
	return d;
}

VENEER_API void DestroyDecoder(Decoder *decoder) { delete decoder; }

VENEER_API int DecodeMesh(Decoder *decoder, Vector3 *vertices, int *indices, Vector3 *normals, Color *colors, Vector2 *texcoord) {
	if(!decoder) return -2;
	if(decoder->nface == 0) return -1;                                  // "Unity does not support point clouds"
	std::vector<unsigned char> rgba;
	try {
		decoder->setIndex((uint32_t *)indices);
		if(decoder->nvert > 0) decoder->setPositions((float *)vertices);
		if(decoder->hasAttr("normal")) decoder->setNormals((float *)normals);
		if(decoder->hasAttr("color") && colors) { rgba.resize((size_t)decoder->nvert*4); decoder->setColors(rgba.data(), 4); }
		if(decoder->hasAttr("uv")) decoder->setUvs((float *)texcoord);
		decoder->decode();
	} catch(const char *) { return -2; }
	for(size_t i = 0; i < rgba.size()/4; i++)
		colors[i] = Color{rgba[4*i]/255.0f, rgba[4*i + 1]/255.0f, rgba[4*i + 2]/255.0f, rgba[4*i + 3]/255.0f};
	return (int)decoder->nface;
}

} // namespace crt
