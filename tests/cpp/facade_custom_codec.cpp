// tests/cpp/facade_custom_codec.cpp — a caller-supplied codec OBJECT behind crt::Decoder::setAttribute(name, buffer, VertexAttribute *)
// (upstream src/decoder.cpp:104-114), compiled against this repo's include/corto/decoder.h + libcorto_hip.so.
//   PlainCodec  restates what upstream's GenericAttr<int> does after its stream decode (include/corto/vertex_attribute.h:160-230:
//               deltaDecode by strategy, dequantize to float): its output must be the built-in codec's, bit for bit;
//   MirrorCodec the same with a twist of its own (x -> 100 - x after dequantisation, and it counts its calls): shows the object's
//               code is what runs, in upstream's order (deltaDecode, postDelta, dequantize).
// usage: facade_custom_codec in.crt <attribute> out.bin [normals]      (normals: bind the normals too)
//   out.bin = the attribute by the built-in codec | by PlainCodec | by MirrorCodec | positions beside PlainCodec (when the attribute is not position)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "corto/decoder.h"

struct PlainCodec: public crt::VertexAttribute {
	int calls[3] = {0, 0, 0};
	int order = 0, seen[3] = {0, 0, 0};
	void deltaDecode(uint32_t nvert, std::vector<crt::Face> &context) override {
		calls[0]++; seen[0] = ++order;
		if(!buffer) return;
		int *values = (int *)buffer;
		if(strategy & PARALLEL) {
			for(uint32_t i = 1; i < context.size(); i++) {
				crt::Face &f = context[i];
				for(int c = 0; c < N; c++) values[i*N + c] += values[f.a*N + c] + values[f.b*N + c] - values[f.c*N + c];
			}
		} else if(context.size()) {
			for(uint32_t i = 1; i < context.size(); i++)
				for(int c = 0; c < N; c++) values[i*N + c] += values[context[i].a*N + c];
		} else {
			for(uint32_t i = N; i < nvert*N; i++) values[i] += values[i - N];
		}
	}
	void postDelta(uint32_t, uint32_t, std::map<std::string, crt::VertexAttribute *> &, crt::IndexAttribute &) override { calls[1]++; seen[1] = ++order; }
	void dequantize(uint32_t nvert) override {
		calls[2]++; seen[2] = ++order;
		if(!buffer) return;
		for(uint32_t i = 0; i < (uint32_t)N*nvert; i++) ((float *)buffer)[i] = ((int *)buffer)[i]*q;
	}
};
struct MirrorCodec: public PlainCodec {
	void dequantize(uint32_t nvert) override {
		PlainCodec::dequantize(nvert);
		for(uint32_t i = 0; i < (uint32_t)N*nvert; i++) ((float *)buffer)[i] = 100.0f - ((float *)buffer)[i];
	}
};

int main(int argc, char **argv) {
	if(argc < 4) return 2;
	FILE *f = fopen(argv[1], "rb");
	if(!f) return 2;
	fseek(f, 0, SEEK_END); long len = ftell(f); fseek(f, 0, SEEK_SET);
	std::vector<uint32_t> storage((len + 3)/4 + 1);
	if(fread(storage.data(), 1, len, f) != (size_t)len) return 2;
	fclose(f);
	const char *name = argv[2];
	try {
		std::vector<float> plain_builtin, by_plain, by_mirror, pos_beside;
		uint32_t nvert = 0, nface = 0; int N = 0;
		{
			crt::Decoder d((int)len, (const uchar *)storage.data());
			if(!d.data.count(name)) { fprintf(stderr, "no such attribute\n"); return 2; }
			nvert = d.nvert; nface = d.nface; N = d.data[name]->N;
			plain_builtin.resize((size_t)nvert*N);
			d.setAttribute(name, (char *)plain_builtin.data(), crt::VertexAttribute::FLOAT);
			d.decode();
		}
		for(int pass = 0; pass < 2; pass++) {
			crt::Decoder d((int)len, (const uchar *)storage.data());
			std::vector<float> &out = pass ? by_mirror : by_plain;
			out.assign((size_t)nvert*N, -1.0f);
			PlainCodec *codec = pass ? new MirrorCodec() : new PlainCodec();       // the Decoder owns it from setAttribute on (decoder.cpp:111-112)
			if(!d.setAttribute(name, (char *)out.data(), codec)) return 3;
			if(d.setAttribute("no such attribute", nullptr, (crt::VertexAttribute *)nullptr)) return 3;
			std::vector<uint32_t> index;
			if(nface) { index.resize((size_t)nface*3); d.setIndex(index.data()); }
			if(!pass && strcmp(name, "position")) { pos_beside.resize((size_t)nvert*3); d.setPositions(pos_beside.data()); }
			std::vector<float> normals;
			if(argc > 4 && d.data.count("normal")) { normals.resize((size_t)nvert*3); d.setNormals(normals.data()); }   // (estimated normals over a custom position: throws)
			d.decode();
			const int want_post = nface ? 1 : 0;                                   // clouds: no postDelta (decoder.cpp:141-146)
			if(codec->calls[0] != 1 || codec->calls[1] != want_post || codec->calls[2] != 1) { fprintf(stderr, "calls %d %d %d\n", codec->calls[0], codec->calls[1], codec->calls[2]); return 4; }
			if(codec->seen[0] != 1 || codec->seen[2] != 2 + want_post) { fprintf(stderr, "order\n"); return 4; }
			if(d.data[name]->codec() != crt::VertexAttribute::CUSTOM_CODEC || d.data[name] != codec) return 4;
			if(nface && d.index.prediction.size() != nvert) return 4;
		}
		FILE *o = fopen(argv[3], "wb");
		fwrite(plain_builtin.data(), 4, plain_builtin.size(), o);
		fwrite(by_plain.data(), 4, by_plain.size(), o);
		fwrite(by_mirror.data(), 4, by_mirror.size(), o);
		fwrite(pos_beside.data(), 4, pos_beside.size(), o);
		fclose(o);
		printf("nvert %u nface %u N %d\n", nvert, nface, N);
	} catch(const char *msg) {
		fprintf(stderr, "error: %s\n", msg);
		return 1;
	}
	return 0;
}
