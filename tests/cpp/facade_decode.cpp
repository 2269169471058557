// tests/cpp/facade_decode.cpp — uses crt::Decoder exactly the way upstream's CLI round trip does
// (src/main.cpp:268-298), but against this repo's include/corto/decoder.h + libcorto_hip.so.
// usage: facade_decode in.crt out.bin   -> out.bin = position | normal | color(4) | uv | index as raw little endian
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "corto/decoder.h"

int main(int argc, char **argv) {
	if(argc < 3) return 2;
	FILE *f = fopen(argv[1], "rb");
	if(!f) return 2;
	fseek(f, 0, SEEK_END); long len = ftell(f); fseek(f, 0, SEEK_SET);
	std::vector<uint32_t> storage((len + 3)/4 + 1);             // 4-byte aligned like the stream buffer upstream
	if(fread(storage.data(), 1, len, f) != (size_t)len) return 2;
	fclose(f);
	try {
		crt::Decoder decoder((int)len, (const uchar *)storage.data());
		const uint32_t nvert = decoder.nvert, nface = decoder.nface;
		std::vector<float> coords(nvert*3), norms, uvs;
		std::vector<uchar> colors;
		std::vector<uint32_t> index;
		decoder.setPositions(coords.data());
		if(decoder.data.count("normal")) { norms.resize(nvert*3); decoder.setNormals(norms.data()); }
		if(decoder.data.count("color")) { colors.resize(nvert*4); decoder.setColors(colors.data(), 4); }
		if(decoder.data.count("uv")) { uvs.resize(nvert*2); decoder.setUvs(uvs.data()); }
		if(decoder.nface) { index.resize(nface*3); decoder.setIndex(index.data()); }
		decoder.decode();
		FILE *o = fopen(argv[2], "wb");
		fwrite(coords.data(), 4, coords.size(), o);
		fwrite(norms.data(), 4, norms.size(), o);
		fwrite(colors.data(), 1, colors.size(), o);
		fwrite(uvs.data(), 4, uvs.size(), o);
		fwrite(index.data(), 4, index.size(), o);
		fclose(o);
		printf("nvert %u nface %u groups %zu exif %zu\n", nvert, nface, decoder.index.groups.size(), decoder.exif.size());
		for(auto &g : decoder.index.groups) {                      // Group::end + properties, as upstream callers read them (src/main.cpp:285-294)
			printf("group %u", g.end);
			for(auto &kv : g.properties) printf("\t%s=%s", kv.first.c_str(), kv.second.c_str());
			printf("\n");
		}
	} catch(const char *msg) {
		fprintf(stderr, "error: %s\n", msg);
		return 1;
	}
	return 0;
}
