// tests/cpp/sort_model_test.cpp — corto_amd/csrc/std_sort_model.h against std::sort itself (libstdc++), on the inputs that matter:
// few distinct keys (probabilities 0..255 of up to 256 symbols), so that the order of EQUAL keys is what is being compared.
#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "../../corto_amd/csrc/std_sort_model.h"

struct Sym { uint8_t symbol, probability; };

static uint64_t rng_state = 88172645463325252ull;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 11); }

int main() {
	long cases = 0, bad = 0;
	auto cmp = [](const Sym &a, const Sym &b) -> bool { return a.probability > b.probability; };
	for(int trial = 0; trial < 60000; trial++) {
		const int n = 1 + (int)(rnd() % 256);
		const int kind = trial % 6;
		std::vector<Sym> v((size_t)n);
		for(int i = 0; i < n; i++) {
			uint32_t p;
			switch(kind) {
			case 0: p = rnd() % 256; break;
			case 1: p = rnd() % 4; break;                       // nearly all ties
			case 2: p = (uint32_t)(255 - (i*255)/n); break;     // already sorted
			case 3: p = (uint32_t)((i*255)/n); break;           // reversed
			case 4: p = (i & 1) ? 7u : (uint32_t)(rnd() % 3); break;
			default: p = (uint32_t)(255.0/(1 + (rnd() % (1 + (uint32_t)i)))); break;   // heavy head, long flat tail
			}
			v[(size_t)i] = Sym{(uint8_t)i, (uint8_t)p};
		}
		std::vector<Sym> ref = v, mine = v;
		std::sort(ref.begin(), ref.end(), cmp);
		corto_hip::std_sort_model(mine.data(), n, cmp);
		cases++;
		for(int i = 0; i < n; i++) if(ref[(size_t)i].symbol != mine[(size_t)i].symbol) { bad++; break; }
	}
	// the heapsort fallback: std::sort only takes it after 2*log2(n) bad partitions; checked against the same pieces of
	// libstdc++ it is made of (make_heap + sort_heap = what __partial_sort(first, last, last) does)
	for(int trial = 0; trial < 20000; trial++) {
		const int n = 17 + (int)(rnd() % 240);
		std::vector<Sym> v((size_t)n);
		for(int i = 0; i < n; i++) v[(size_t)i] = Sym{(uint8_t)i, (uint8_t)(rnd() % (trial % 2 ? 5 : 256))};
		std::vector<Sym> ref = v, mine = v;
		std::make_heap(ref.begin(), ref.end(), cmp); std::sort_heap(ref.begin(), ref.end(), cmp);
		// with depth limit 0 the model heap-sorts the whole range, then runs the final insertion sort over sorted data (a no-op)
		corto_hip::std_sort_model(mine.data(), n, cmp, 0);
		cases++;
		for(int i = 0; i < n; i++) if(ref[(size_t)i].symbol != mine[(size_t)i].symbol) { bad++; break; }
	}
	printf("cases %ld mismatches %ld\n", cases, bad);
	return bad ? 1 : 0;
}
