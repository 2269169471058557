// enc_window_model.cpp — CPU model of k_enc_tun_parse's formulation (tests/test_encode_stage_cpu.py).
// Reads "n | size_i, bytes_i ..." from stdin, builds each stream's encoder tables with the library's own host code
// (corto_hip::tun_encoder_tables), parses the stream the way the kernel does - 64 start positions per window, each
// walking the int16 trie like the reference's loop, then the chain cur -> next[cur] - and writes the framed blocks to
// stdout.  The test compares them with the blocks the reference made (tests/golden/tunstall_enc_kat.npz).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../corto_amd/csrc/encoder_internal.h"

using namespace corto_hip;

static std::vector<uint8_t> parse(const TunEncoderTables &T, const std::vector<uint8_t> &data) {
	const uint32_t size = (uint32_t)data.size(), n = T.nsym, span = n*n;
	std::vector<int16_t> trie(T.offsets.size());
	for(size_t k = 0; k < trie.size(); k++) { const int32_t v = T.offsets[k]; trie[k] = v >= 0 ? (int16_t)(v & 255) : (int16_t)-((-v)/(int32_t)span); }
	auto TR = [&](uint32_t i) -> int32_t { return i < trie.size() ? trie[i] : 0; };
	std::vector<uint8_t> out;
	uint32_t base = 0;
	while(base < size && out.size() <= size) {
		int32_t code[64]; uint32_t next[64];
		for(uint32_t lane = 0; lane < 64; lane++) {
			const uint32_t p = base + lane;
			code[lane] = 0; next[lane] = p;
			if(p >= size) continue;
			uint32_t in = p, woff = 0, level = 0;
			for(;;) {
				int32_t t;
				if(in >= size) { do { t = TR(level); level = (uint32_t)(-t)*span; } while(t < 0); code[lane] = t; next[lane] = in; break; }
				uint32_t low = (uint32_t)T.remap[data[in]]*n;
				if(size - in >= 2) low += T.remap[data[in + 1]];
				t = TR(level + low);
				if(t >= 0) { code[lane] = t; next[lane] = in + T.lengths[t & 255] - woff; break; }
				level = (uint32_t)(-t)*span; woff += 2; in += 2;
			}
		}
		uint32_t cur = base;
		while(cur < size && cur - base < 64 && out.size() <= size) { const uint32_t l = cur - base; out.push_back((uint8_t)code[l]); cur = next[l]; }
		base = cur;
	}
	return out;
}

int main() {
	uint32_t n = 0;
	if(fread(&n, 4, 1, stdin) != 1) return 2;
	for(uint32_t i = 0; i < n; i++) {
		uint32_t size = 0;
		if(fread(&size, 4, 1, stdin) != 1) return 2;
		std::vector<uint8_t> data(size);
		if(size && fread(data.data(), 1, size, stdin) != size) return 2;
		uint32_t counts[256] = {0};
		for(uint8_t b : data) counts[b]++;
		TunEncoderTables T;
		if(size) tun_encoder_tables(counts, size, T);
		std::vector<uint8_t> codes;
		if(T.nsym >= 2) codes = parse(T, data);
		const uint8_t ns = (uint8_t)T.nsym;
		const uint32_t cs = (uint32_t)codes.size();
		fwrite(&ns, 1, 1, stdout); fwrite(T.probs, 1, (size_t)T.nsym*2, stdout); fwrite(&size, 4, 1, stdout); fwrite(&cs, 4, 1, stdout);
		fwrite(codes.data(), 1, codes.size(), stdout);
	}
	return 0;
}
