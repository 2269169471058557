/* corto_codec.h — the legacy C ABI of corto's Unity plugin, served by the MI355X decoder.
 *
 * Replaces upstream src/corto_codec.h:16-44 / src/corto_codec.cpp:6-59 (library "cortocodec_unity"): same three
 * symbols, same struct layouts, same argument meaning, so unity/CortoMeshLoader.cs:102-107 binds it unchanged
 * (rename libcortocodec_hip.so, or point DllImport at it).  Built by `python -m corto_amd.build` next to
 * libcorto_hip.so, on top of the crt::Decoder facade (include/corto/decoder.h).
 *
 * Differences from upstream, all at the edges:
 *   - exceptions do not cross the C boundary: CreateDecoder returns NULL for a blob the reference would throw on
 *     ("Not a crt file." ...), DecodeMesh returns -2 for a decode error; crthip_last_error() has the message.
 *   - colours: upstream binds the Color array as FLOAT, a format its ColorAttr::dequantize mishandles (it rescales
 *     whatever bytes the caller's array held, src/color_attribute.cpp:96-110).  Here colours are decoded as RGBA8 on
 *     the device - the reference's well-defined path - and written as r,g,b,a = u8/255.
 *   - DecodeMesh is one-shot per decoder, like upstream (the stream position is consumed by decode()).
 */
#ifndef CORTO_HIP_CORTO_CODEC_H
#define CORTO_HIP_CORTO_CODEC_H

#include "decoder.h"

namespace crt {
extern "C" {
struct Color { float r, g, b, a; };
struct Vector2 { float x, y; };
struct Vector3 { float x, y, z; };

/* decoderInfo[0] = (nface, nvert), src/corto_codec.cpp:11-13 */
Decoder *CreateDecoder(int length, unsigned char *data, Vector2 *decoderInfo);
void DestroyDecoder(Decoder *decoder);
/* returns nface; -1 for a point cloud (src/corto_codec.cpp:24-27); -2 on a decode error */
int DecodeMesh(Decoder *decoder, Vector3 *vertices, int *indices, Vector3 *normals, Color *colors, Vector2 *texcoord);
}
}
#endif
