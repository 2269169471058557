// tests/cpp/facade_threads.cpp — distinct crt::Decoder objects on several threads at once, the use upstream allows
// (its Decoder objects share nothing, SURVEY.md §8b "Threading"), against include/corto/decoder.h + libcorto_hip.so.
// usage: facade_threads nthreads rounds out_prefix a.crt [b.crt ...]
//   every thread decodes every file `rounds` times (a fresh Decoder per decode, threads start at different files) and checks
//   that each decode gives the same bytes as its first one; thread 0 then writes <out_prefix><file index>.bin
//   (position | normal | color(4) | uv | index) for the caller to compare with the oracle.
//   With out_prefix "-" nothing is written.  $FACADE_BAD_BIND_THREAD=k: thread k binds its normals with a format the device path refuses
//   (UINT8) and must get the "Format not supported" exception on EVERY decode - while the threads co-batched with it decode as ever.  Prints "per_decode_us <mean>" : wall time per decode() call of one thread, and
//   "wall_us_per_blob": the run's wall time (decodes + the callers' own buffer handling) over all decodes of all threads.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "corto/decoder.h"

static std::vector<uint32_t> slurp(const char *path, long &len) {
	FILE *f = fopen(path, "rb");
	if(!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
	fseek(f, 0, SEEK_END); len = ftell(f); fseek(f, 0, SEEK_SET);
	std::vector<uint32_t> storage((len + 3)/4 + 1);
	if(fread(storage.data(), 1, len, f) != (size_t)len) exit(2);
	fclose(f);
	return storage;
}

static std::vector<uchar> decode_one(const uchar *blob, int len, double *us, bool bad_bind = false) {
	crt::Decoder decoder(len, blob);
	const uint32_t nvert = decoder.nvert, nface = decoder.nface;
	std::vector<float> coords(nvert*3), norms, uvs;
	std::vector<uchar> colors;
	std::vector<uint32_t> index;
	decoder.setPositions(coords.data());
	if(decoder.data.count("normal")) { norms.resize(nvert*3); if(bad_bind) decoder.setAttribute("normal", (char *)norms.data(), crt::VertexAttribute::UINT8); else decoder.setNormals(norms.data()); }
	if(decoder.data.count("color")) { colors.resize(nvert*4); decoder.setColors(colors.data(), 4); }
	if(decoder.data.count("uv")) { uvs.resize(nvert*2); decoder.setUvs(uvs.data()); }
	if(nface) { index.resize(nface*3); decoder.setIndex(index.data()); }
	const auto t0 = std::chrono::steady_clock::now();
	decoder.decode();
	if(us) *us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
	std::vector<uchar> out;
	auto put = [&](const void *p, size_t n) { const uchar *b = (const uchar *)p; out.insert(out.end(), b, b + n); };
	put(coords.data(), coords.size()*4); put(norms.data(), norms.size()*4); put(colors.data(), colors.size());
	put(uvs.data(), uvs.size()*4); put(index.data(), index.size()*4);
	return out;
}

int main(int argc, char **argv) {
	if(argc < 5) return 2;
	const int nthreads = atoi(argv[1]), rounds = atoi(argv[2]);
	const std::string prefix = argv[3];
	const int nfiles = argc - 4;
	std::vector<std::vector<uint32_t>> blobs(nfiles);
	std::vector<long> lens(nfiles);
	for(int i = 0; i < nfiles; i++) blobs[i] = slurp(argv[4 + i], lens[i]);
	std::atomic<int> failures(0);
	std::vector<std::vector<std::vector<uchar>>> first(nthreads, std::vector<std::vector<uchar>>(nfiles));
	std::vector<double> us(nthreads, 0.0);
	std::vector<std::thread> pool;
	try { for(int k = 0; k < 3; k++) (void)decode_one((const uchar *)blobs[0].data(), (int)lens[0], nullptr); }     // HIP start-up and the first context are not what is timed
	catch(const char *) {}                                                                                         // (a file that cannot be decoded: the threads will say so)
	const char *bb = getenv("FACADE_BAD_BIND_THREAD");
	const int bad_thread = bb ? atoi(bb) : -1;
	const auto wall0 = std::chrono::steady_clock::now();
	for(int t = 0; t < nthreads; t++) pool.emplace_back([&, t]() {
		if(t == bad_thread) {
			for(int r = 0; r < rounds; r++)
				for(int k = 0; k < nfiles; k++) {
					const int i = (k + t) % nfiles;
					{ crt::Decoder probe((int)lens[i], (const uchar *)blobs[i].data()); if(!probe.data.count("normal")) continue; }
					bool thrown = false;
					try { (void)decode_one((const uchar *)blobs[i].data(), (int)lens[i], nullptr, true); }
					catch(const char *msg) { thrown = strstr(msg, "Format not supported") != nullptr; }
					if(!thrown) { fprintf(stderr, "thread %d: the refused binding did not throw\n", t); failures++; }
				}
			return;
		}
		try {
			for(int r = 0; r < rounds; r++)
				for(int k = 0; k < nfiles; k++) {
					const int i = (k + t) % nfiles;
					std::vector<uchar> got = decode_one((const uchar *)blobs[i].data(), (int)lens[i], &us[t]);
					if(r == 0) first[t][i].swap(got);
					else if(got != first[t][i]) { fprintf(stderr, "thread %d: decode %d of file %d differs from its first\n", t, r, i); failures++; }
				}
		} catch(const char *msg) { fprintf(stderr, "thread %d: %s\n", t, msg); failures++; }
	});
	for(auto &th : pool) th.join();
	const double wall_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - wall0).count();
	for(int t = 1; t < nthreads; t++)
		for(int i = 0; t != bad_thread && i < nfiles; i++) if(first[t][i] != first[0][i]) { fprintf(stderr, "thread %d disagrees with thread 0 on file %d\n", t, i); failures++; }
	if(prefix != "-")
		for(int i = 0; i < nfiles; i++) {
			FILE *o = fopen((prefix + std::to_string(i) + ".bin").c_str(), "wb");
			fwrite(first[0][i].data(), 1, first[0][i].size(), o); fclose(o);
		}
	double tot = 0; for(double x : us) tot += x;
	printf("per_decode_us %.1f\n", tot/((double)nthreads*rounds*nfiles));
	printf("wall_us_per_blob %.1f\n", wall_us/((double)nthreads*rounds*nfiles));      // all threads together: what the process gets a blob decoded in
	return failures ? 1 : 0;
}
