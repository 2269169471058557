// tun_tables.h — the Tunstall dictionary builder (crt::Tunstall::createDecodingTables2, src/tunstall.cpp:125-256) as a device
// function of ONE wave, shared by the decode kernels (k_tunstall.hip: K-TAB, K-STREAM) and the encoder's table kernel
// (k_encode.hip: k_enc_tables).  Include inside namespace corto_hip, after kernels_common.h / device_plan.h.
#pragma once

// The dictionary of one stream, by one wave.  Tg != null: written to the stream's TunTable in HBM (long streams: many workgroups
// decode from it).  Otherwise it stays in LDS - word bytes in tun_words(), offsets / lengths in loff / llen - for the same wave to
// decode from (k_tun_stream below).  Returns (used bytes, longest word).
struct TunBuilt { uint32_t used, maxlen; };
#ifndef TUN_BATCH_MAX_N
#define TUN_BATCH_MAX_N 8u            // alphabets up to this size grow their dictionary in batches of pops (tun_tables_body; 0: never)
#endif
// -DCORTO_TUN_STAMPS (CORTO_BUILD_DEFINES=CORTO_TUN_STAMPS python -m corto_amd.build --force; tools/tun_stamp_probe.py): where a stream's
// time goes - 100 MHz stamps of workgroups 0..4095 at the phase boundaries, read back with crthip_debug_tun_stamps
#ifdef CORTO_TUN_STAMPS
__device__ uint64_t g_tun_stamps[8*4096];
#define TUN_STAMP(k) do { if(threadIdx.x == 0 && blockIdx.x < 4096) g_tun_stamps[blockIdx.x*8 + (k)] = wall_clock64(); } while(0)
#else
#define TUN_STAMP(k) do { } while(0)
#endif
__shared__ __attribute__((aligned(16))) uint8_t g_tun_words[TUN_TABLE_BYTES];        // (one definition: both kernels below are single-wave workgroups)
// WORDS_TO_HBM (K-TAB only, Tg != null): up to 64 symbols - every stream an encoder really writes - the word bytes are only ever WRITTEN
// while the dictionary is made (seed bytes, then the surviving words spelled out), so they go straight to the TunTable in HBM and the
// kernel's LDS is the 6.4 KB of the growth bookkeeping instead of 15.6 KB (its LDS.time is what a batch's ~250 dictionaries cost the
// pipelined decode).  Bigger alphabets copy parents' bytes: they use `big_words`, 9 KB of dynamic LDS the host adds to the
// launch when a stream of the launch has more than 64 symbols.
template <bool WORDS_TO_HBM = false>
__device__ __forceinline__ TunBuilt tun_tables_body(const TunStream &st, TunTable *Tg, uint16_t *loff, uint8_t *llen, const uint8_t *probs_override = nullptr, uint8_t *big_words = nullptr) {
	const uint8_t *probs = probs_override ? probs_override : st.probs;   // nsym x (symbol, probability), sorted as stored in the stream
	TunTable &T = *Tg;                           // (only touched when Tg != null)
	const bool to_hbm = WORDS_TO_HBM && st.nsym <= 64;
	uint8_t *buf = WORDS_TO_HBM ? big_words : g_tun_words;                 // (LDS: read and written; to_hbm: never touched)
	CRT_GLOBAL uint8_t *gbuf = WORDS_TO_HBM ? as_global(Tg->bytes) : nullptr;
	auto put = [&](uint32_t at, uint8_t v) { if(WORDS_TO_HBM && to_hbm) gbuf[at] = v; else buf[at] = v; };
	const uint32_t n = st.nsym;                 // 2..255 (host guarantees)
	const uint32_t lane = threadIdx.x;

	__shared__ uint32_t epl[TUN_ENTRY_CAP];     // entry e (creation order) lives in FIFO row e % n.  Low half: its probability, 16-bit ((a*b) >> 16 of 16-bit
	                                            // factors); high half: its length (n <= 64: | last symbol << 8) - one LDS read serves both
	__shared__ uint16_t eoff[TUN_ENTRY_CAP];
	__shared__ uint16_t head[256];              // oldest not-yet-expanded entry of each row
	__shared__ uint16_t P[256];                 // probability << 8  (16.16-ish fixed point)
	__shared__ uint16_t pw[256];                // P0^k (successive (a*b)>>16), low-entropy seed only
	__shared__ uint8_t sym[256];

	TUN_STAMP(0);
	for(uint32_t i = lane; i < n; i += 64) { sym[i] = probs[2*i]; P[i] = (uint32_t)probs[2*i + 1] << 8; }
	for(uint32_t i = lane; i < TUN_ENTRY_CAP; i += 64) { epl[i] = 0; eoff[i] = 0; }
	__syncthreads();

	// how long a run of the likeliest symbol stays likelier than the runner-up (tunstall.cpp:143-151)
	const uint32_t p0 = P[0], p1 = P[1];
	uint32_t count = 2, run = (p0*p0) >> 16;
	const uint32_t max_count = 255u/(n - 1);
	while(run > p1 && count < max_count) { run = (run*p0) >> 16; count++; }

	uint32_t pos, end, nwords;
	if(count >= 16) {                           // low-entropy seed (tunstall.cpp:153-193)
		// byte store: A | for k>=1: A^(count-1) sym_k ; word (row k, col) = the (col+1)-byte suffix ending at k*count
		const uint32_t total = 1 + (n - 1)*count;
		const uint8_t A = sym[0];
		for(uint32_t b = lane; b < total; b += 64) {
			uint8_t v = A;
			if(b > 0) { uint32_t k = (b - 1)/count + 1, j = (b - 1) - (k - 1)*count; if(j == count - 1) v = sym[k]; }
			put(b, v);
		}
		if(lane == 0) {                         // P0^col, col = 1..count
			uint32_t v = p0; pw[1] = v;
			for(uint32_t c = 2; c <= count; c++) { v = (v*p0) >> 16; pw[c] = v; }
		}
		__syncthreads();
		for(uint32_t e = lane; e < count*n; e += 64) {
			const uint32_t col = e/n, row = e - col*n;
			if(row == 0) continue;
			const uint32_t pr = col == 0 ? (uint32_t)P[row] : ((uint32_t)pw[col]*(uint32_t)P[row]) >> 16;
			eoff[e] = (uint16_t)(row*count - col);
			epl[e] = (pr & 0xFFFFu) | ((col + 1) | ((uint32_t)sym[row] << 8)) << 16;     // (last symbol: used by the n <= 64 path)
		}
		for(uint32_t k = lane; k < n; k += 64) head[k] = (uint16_t)(k == 0 ? (count - 1)*n : k);
		__syncthreads();
		if(lane == 0) { const uint32_t first = (count - 1)*n; epl[first] = (uint32_t)pw[count] | (count | (uint32_t)A << 8) << 16; eoff[first] = 0; }
		nwords = 1 + count*(n - 1);
		end = count*n;
		pos = total;
	} else {                                    // one-symbol words (tunstall.cpp:195-205)
		for(uint32_t i = lane; i < n; i += 64) {
			head[i] = (uint16_t)i; epl[i] = (uint32_t)P[i] | (1u | ((uint32_t)sym[i] << 8)) << 16; eoff[i] = (uint16_t)i; put(i, sym[i]);
		}
		nwords = n; end = n; pos = n;
	}
	__syncthreads();

	TUN_STAMP(1);
	const bool tree = n <= 64;
	const uint32_t seed_end = end, seed_bytes = pos;
	if(tree) {
		// Fast path (n <= 64 symbols, i.e. every stream the encoder really produces).  Lane r keeps row r's FIFO head (index,
		// probability, length) in registers, so picking the likeliest head is a register-only wave reduction.  A child is
		// recorded as (parent entry, row) - no bytes are copied while the dictionary grows; the 256 surviving words are
		// spelled out once at the end by walking up to the seed.  For the entries made here eoff[] holds the PARENT ENTRY and
		// the top byte of epl[] the word's last symbol.                                    tunstall.cpp:207-241
		// The loop runs up to 254 times (two symbols) and one wave issues an instruction every four clocks at best, so its length in
		// INSTRUCTIONS is what a short stream's decode costs; per lane r < n, packed for few selects:
		//   K   head key: probability << 16 | (0xFFFF - r), probability 0 when the FIFO is empty   (max = likeliest head, first row on ties,
		//       row 0 when all are zero)                 HL  head entry (or, FIFO empty, the entry the row's next child will be) | length << 16
		//   NR  the record (epl[] format) of the entry behind the head, or 0       NX  that entry's index       E  the lane's next child entry
		// A row's FIFO is the children lane r made, in order, so head and next stay in registers: the record behind the new next is read
		// from LDS when a head is popped and merged an iteration later - no LDS round trip on the critical path.  Bounds: end <= 510 + n
		// whatever the probabilities (every expansion adds n entries and n - 1 words), so HL < 640 and NX < 704 < TUN_ENTRY_CAP - 1;
		// lanes >= n write their (ignored) child to the spare entry TUN_ENTRY_CAP - 1.
		// ---- batched growth (round 6; host model and proof: tools/tun_batch_model.py).  The expansion order is a merge of the n FIFOs, each
		// non-increasing when the table is sorted (every stream upstream's encoder writes, tunstall.cpp:108-118: a child is (parent * P[r]) >> 16
		// and parents come in non-increasing order; the low-entropy seed's columns fall below p0^(count-1) >= whatever is popped first).  So with
		// M = the likeliest head and c_max = (M * P[0]) >> 16 = the likeliest entry any expansion from now on can make, EVERY unexpanded entry
		// above c_max is popped before anything that does not exist yet: a whole set of pops is known at once, in the order (probability desc,
		// row asc, FIFO order), and the k-th of them makes entries end + k*n .. + n - 1.  A 2-symbol alphabet's 252 serial expansions (42 us,
		// the length of a launch while 200 other waves idle) become ~20 batches, a 4-symbol one's 83 seven to sixteen.
		// Lane l looks at entry bj = l / n of row br = l % n's FIFO (a window of J = 64 / n entries a row); a row whose whole window is
		// selected may hold more behind it: nothing later in the order than its last candidate is taken.  Ranks by comparing keys over the selected
		// lanes; every popped lane writes its n children.  Anything unusual (all heads zero, an unsorted table) is left to the loop below,
		// which picks up from head[] / epl[] at any point.
		if(n <= TUN_BATCH_MAX_N && nwords + n <= 255) {
			const uint32_t J = 64u/n, br = lane % n, bj = lane/n;
			const bool slot = bj < J;
			const uint32_t nextP = lane + 1 < n ? P[lane + 1] : 0u, thisP = lane < n ? P[lane] : 0u;
			const bool ordered = __ballot(lane + 1 < n && thisP < nextP) == 0;
			uint32_t left = (255 - n - nwords)/(n - 1) + 1;                          // expansions that pop their parent
			uint64_t rowmask = 0;                                                    // the window lanes of my row
			for(uint32_t q = 0; q < n; q++) { const uint64_t m = __ballot(slot && br == q); rowmask = br == q ? m : rowmask; }
			uint32_t H = head[br];
			const uint32_t tie = (63u - br) << 6 | (63u - bj), step = bj*n;
			const uint32_t pmax = P[0];
			while(ordered && left) {
				const uint32_t e = H + step;
				const uint32_t v = slot && e < end ? epl[e] : 0u;
				const uint32_t prob = v & 0xFFFFu;
				const uint32_t M = wave_max_u32(prob);
				if(M == 0) break;                                                    // (every head zero or empty: the loop below has upstream's row-0 rule)
				const uint32_t cmax = (M*pmax) >> 16;
				bool sel = prob > cmax;
				const uint32_t key = prob << 12 | tie;
				// (a row whose whole window is selected may hold more behind it, all of it later in the order than the window's last entry: what is safe is
				// what comes no later than the earliest such last entry - by KEY, so that a flat table's level of equal probabilities still goes a row at a time)
				const uint64_t deep = __ballot(sel && bj == J - 1 && e + n < end);
				if(deep) { const uint32_t cut = wave_max_u32((deep >> lane) & 1 ? key : 0u); sel = sel && key >= cut; }
				const uint64_t smask = __ballot(sel);
				if(!smask) break;
				// ranks: a row's own entries are in order already (lane = bj*n + br), so the row with the most selected entries takes its ranks from its
				// position in the row plus ONE compare against each selected entry of the OTHER rows - and each of those gets its exact rank in the same
				// step (how many selected keys beat it: a ballot and a count).  The loop runs over the other rows' entries only (a two-symbol alphabet's
				// A-row holds two thirds to three quarters of a batch).
				const uint32_t mine = (uint32_t)__popcll(smask & rowmask);
				const uint32_t bigkey = wave_max_u32(slot ? mine << 8 | (255u - br) : 0u);
				const uint32_t bigrow = 255u - (bigkey & 255u);
				const uint64_t bigmask = __ballot(slot && br == bigrow);
				uint32_t rank = (uint32_t)__popcll(smask & rowmask & ((1ull << lane) - 1ull));     // (my place among my row's selected entries)
				if(n == 2) {
					// two rows: how many of the OTHER row's selected entries come before me is a binary search down that row (its keys fall with bj): six
					// cross-lane reads whatever the batch's size - a flat two-symbol table pops 32 a row and batch, and the loop below took a step an entry
					const uint32_t other = (uint32_t)__popcll(smask & ~rowmask);                       // (lanes 0 .. 63 are all window slots when n = 2)
					uint32_t lo = 0, hi = other;
#pragma unroll
					for(int it = 0; it < 6; it++) {
						const uint32_t mid = (lo + hi) >> 1;
						const uint32_t k = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((mid << 1) | (br ^ 1u)) << 2), (int)key);
						const bool before = lo < hi && k > key;
						lo = before ? mid + 1 : lo; hi = before || lo >= hi ? hi : mid;
					}
					rank += lo;
				} else
				for(uint64_t m = smask & ~bigmask; m; m &= m - 1) {
					const uint32_t i = (uint32_t)__builtin_ctzll(m);
					const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key, (int)i);
					rank += br == bigrow && k > key ? 1u : 0u;
					const uint32_t beat = (uint32_t)__popcll(__ballot(sel && key > k));
					rank = lane == i ? beat : rank;
				}
				const bool pop = sel && rank < left;
				const uint64_t pmask = __ballot(pop);
				const uint32_t npop = (uint32_t)__popcll(pmask);
				if(pop) {
					const uint32_t cbase = end + rank*n, len1 = (v & 0xFF0000u) + 0x10000u;
					for(uint32_t q = 0; q < n; q++) {
						const uint32_t t = (uint32_t)__umul24(prob, (uint32_t)P[q]);        // (HIP declares __umul24 as int: shifted as such, a product >= 2^31 smears its sign over the record)
						epl[cbase + q] = (t >> 16) | (uint32_t)sym[q] << 24 | len1;
						eoff[cbase + q] = (uint16_t)e;
					}
				}
				H += (uint32_t)__popcll(pmask & rowmask)*n;
				end += npop*n; nwords += npop*(n - 1); left -= npop;
				asm volatile("" ::: "memory");
			}
			if(lane < n) head[lane] = (uint16_t)H;
			__syncthreads();
		}
		const bool rowlane = lane < n;
		const uint32_t rowc = 0xFFFFu - lane;
		const uint32_t myP = rowlane ? P[lane] : 0u, sym24 = rowlane ? (uint32_t)sym[lane] << 24 : 0u;
		uint32_t K = 0, HL = 0xFFFFu, NR = 0, NX = TUN_ENTRY_CAP - 1, LD = 0;      // (lanes without a row: NX is an address they read every expansion)
		if(rowlane) {
			const uint32_t h = head[lane], v = epl[h];
			K = (v << 16) | rowc; HL = h | (v & 0xFF0000u); NX = h + n;
			if(h + n < end) NR = epl[h + n];
		}
		uint32_t E = rowlane ? end + lane : TUN_ENTRY_CAP - 1;
		const uint32_t inc = rowlane ? n : 0u;
		__builtin_amdgcn_s_waitcnt(0xC07F);                                       // lgkmcnt(0): no LDS read pending into the loop (its waits would land on the loop's top)
		const uint32_t width = n <= 4 ? 4u : n <= 16 ? 16u : 64u;                  // lanes that can hold a row
		// (the reduction's width is a compile-time constant of each of the three loops below: tested inside the loop it is three branches an expansion)
		auto likeliest_w = [&](uint32_t key, auto W) -> uint32_t {
			if(decltype(W)::value == 64) return wave_max_u32(key);
#define CRT_DPP_MAX(ctrl) { const uint32_t o_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, ctrl, 0xf, 0xf, false); key = o_ > key ? o_ : key; }
			CRT_DPP_MAX(0xB1) CRT_DPP_MAX(0x4E)                                   // quad_perm [1,0,3,2], [2,3,0,1]
			if(decltype(W)::value == 16) { CRT_DPP_MAX(0x141) CRT_DPP_MAX(0x140) } // row_half_mirror, row_mirror
#undef CRT_DPP_MAX
			return (uint32_t)__builtin_amdgcn_readfirstlane((int)key);
		};
		auto likeliest = [&](uint32_t key) -> uint32_t {
			return width == 4 ? likeliest_w(key, std::integral_constant<uint32_t, 4>{}) : width == 16 ? likeliest_w(key, std::integral_constant<uint32_t, 16>{})
			                                                                        : likeliest_w(key, std::integral_constant<uint32_t, 64>{});
		};
		bool pend = false;
		uint32_t whole = nwords + n <= 255 ? (255 - n - nwords)/(n - 1) + 1 : 0;   // expansions that pop their parent
		nwords += whole*(n - 1); end += whole*n;
		auto grow = [&](auto W) {
			for(; whole; whole--) {
				const uint32_t key = likeliest_w(K, W);
				const uint32_t best = 0xFFFFu & ~key;
				const uint32_t hq = (uint32_t)__builtin_amdgcn_readlane((int)HL, (int)best);
				const uint32_t parent = hq & 0xFFFFu, len1 = (hq & 0xFF0000u) + 0x10000u;   // the children's length, in place
				const uint32_t t = __umul24(key >> 16, myP);                          // child probability = t >> 16
				const uint32_t recv = (t >> 16) | sym24 | len1;
				epl[E] = recv; eoff[E] = (uint16_t)parent;
				asm volatile("" : "+v"(LD) :: "memory");                              // one wave: LDS executes in program order; the wait for the previous pop's read belongs HERE, behind the reduction
				NR = pend ? LD : NR; pend = false;                                    // (the read issued by the previous pop)
				const bool is_head = (HL & 0xFFFFu) == E;                             // the row's FIFO was empty: the child is its head
				K = is_head ? ((t & 0xFFFF0000u) | rowc) : K;
				HL = is_head ? (E | len1) : HL;
				NR = NX == E ? recv : NR;                                             // ... held only its head: the child is next
				pend = lane == best;                                                   // the parent is fully expanded: its row pops it (selects, not a branch:
				const uint32_t hnew = parent + n;                                      // every lane reads `its` next-but-one record, the popping lane's is the new one)
				K = pend ? (NR << 16) | rowc : K; HL = pend ? hnew | (NR & 0xFF0000u) : HL;
				NX = pend ? hnew + n : NX;
				LD = epl[NX];                                                          // (zero behind the last entry)
				E += inc;
			}
		};
		if(width == 4) grow(std::integral_constant<uint32_t, 4>{});
		else if(width == 16) grow(std::integral_constant<uint32_t, 16>{});
		else grow(std::integral_constant<uint32_t, 64>{});
		NR = pend ? LD : NR;
		if(nwords < 256) {                                                        // the dictionary fills up during the last expansion: its parent stays
			const uint32_t m = 256 - nwords;
			const uint32_t key = likeliest(K);
			const uint32_t best = 0xFFFFu & ~key;
			const uint32_t hq = (uint32_t)__builtin_amdgcn_readlane((int)HL, (int)best);
			const uint32_t t = __umul24(key >> 16, myP);
			if(lane < m) { epl[E] = (t >> 16) | sym24 | ((hq & 0xFF0000u) + 0x10000u); eoff[E] = (uint16_t)(hq & 0xFFFFu); }
			end += m; nwords = 256;
		}
		TUN_STAMP(2);
		const uint32_t h = HL & 0xFFFFu;
		if(lane < n) head[lane] = (uint16_t)min(h, 0xFFFFu);
		__syncthreads();
	} else
	while(nwords < 256) {                       // tunstall.cpp:207-241 (general path, n > 64)
		// likeliest FIFO head; first row wins ties; all-zero -> row 0.  key = prob:16 | (0xFFFF - row)
		uint32_t key = 0;
		for(uint32_t r = lane; r < n; r += 64) {
			const uint32_t h = head[r];
			const uint32_t p = h < TUN_ENTRY_CAP ? epl[h] & 0xFFFFu : 0u;
			const uint32_t k = p ? ((p << 16) | (0xFFFFu - r)) : 0u;
			key = k > key ? k : key;
		}
#pragma unroll
		for(int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(key, d, 64); key = o > key ? o : key; }
		const uint32_t best = (key >> 16) ? 0xFFFFu - (key & 0xFFFFu) : 0u;
		const uint32_t parent = head[best];
		if(parent >= TUN_ENTRY_CAP) break;      // malformed probabilities (reference: out-of-bounds read)
		const uint32_t pp = epl[parent] & 0xFFFFu, po = eoff[parent], pl = (epl[parent] >> 16) & 255u;   // (a seed's top byte is its symbol)
		const bool full = nwords + n > 255;     // dictionary fills up during this expansion: parent stays
		const uint32_t m = full ? 256 - nwords : n;
		const uint32_t tot = m*(pl + 1);
		if(end + m > TUN_ENTRY_CAP || pos + tot > TUN_TABLE_BYTES) break;
		for(uint32_t r = lane; r < m; r += 64) {
			const uint32_t e = end + r;
			epl[e] = (((pp*P[r]) >> 16) & 0xFFFFu) | (pl + 1) << 16;
			eoff[e] = (uint16_t)(pos + r*(pl + 1));
		}
		for(uint32_t b = lane; b < tot; b += 64) {   // child r = parent bytes + sym[r]
			const uint32_t r = b/(pl + 1), j = b - r*(pl + 1);
			buf[pos + b] = j < pl ? buf[po + j] : sym[r];
		}
		__syncthreads();
		if(!full && lane == 0) head[best] = (uint16_t)(parent + n);
		end += m; pos += tot; nwords += n - 1;
		__syncthreads();
	}

	// survivors in creation order -> codes 0..255 (tunstall.cpp:243-253).  Seed words keep their place in the seed bytes
	// (they share suffixes); on the n <= 64 path every surviving word made by an expansion is spelled out behind them:
	// its last symbols come from the rows on the way up to the seed entry, the rest is the seed word A^(k-1) sym[row].
	uint32_t w = 0, used = tree ? seed_bytes : 0u, maxlen = 0, wpos = seed_bytes;
	uint32_t row = lane % n;                                              // e % n, carried along (an integer division per entry otherwise)
	const uint32_t rstep = 64u % n;
	const uint8_t A = sym[0];
	// one word: its table entry, and - made by an expansion - its bytes, spelled from the back up to its seed word A^(k-1) sym[row]
	auto emit = [&](bool take, uint32_t e, uint32_t rank) {
		const uint32_t len_ = take ? (epl[e] >> 16) & 255u : 0u;
		const bool made = tree && take && e >= seed_end;                  // spelled out here
		const uint32_t incl = wave_inclusive_scan_u32(made ? len_ : 0u);
		uint32_t off = made ? wpos + incl - len_ : take ? (uint32_t)eoff[e] : 0u;
		wpos += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
		const bool fits = off + len_ <= TUN_TABLE_BYTES;                    // (always, on streams the reference decodes without overrunning its own buffer)
		if(!fits) { off = 0; }
		const uint32_t len = fits ? len_ : 0u;
		if(take) {
			if(Tg) { T.off[rank] = (uint16_t)off; T.len[rank] = (uint8_t)len; } else { loff[rank] = (uint16_t)off; llen[rank] = (uint8_t)len; }
			used = off + len > used ? off + len : used;
			maxlen = len > maxlen ? len : maxlen;
		}
		if(made && len) {
			uint32_t cur = e, j = off + len;
			while(cur >= seed_end && j > off + 1) { const uint32_t lr = epl[cur]; put(--j, (uint8_t)(lr >> 24)); cur = eoff[cur]; }
			const uint32_t lr = epl[cur];
			put(--j, (uint8_t)(lr >> 24));
			while(j > off) put(--j, A);
		}
	};
	if(tree) {
		// the survivors first, compacted (creation order = code order): the ~510 entries hold 256 survivors, and a pass of the spelling loop is as long
		// as its longest word - four full passes instead of eight half-empty ones.  (pw[] - the low-entropy seed's powers - is free by now.)
		uint16_t *surv = pw;
		for(uint32_t base = 0; base < end; base += 64) {
			const uint32_t e = base + lane;
			const bool alive = e < end && !(head[row] > e);
			row += rstep; row -= row >= n ? n : 0u;
			const uint64_t mask = __ballot(alive);
			const uint32_t rank = w + __popcll(mask & ((1ull << lane) - 1ull));
			if(alive && rank < 256) surv[rank] = (uint16_t)e;
			w += __popcll(mask);
		}
		__syncthreads();
		const uint32_t nw = w < 256 ? w : 256;
		for(uint32_t base = 0; base < nw; base += 64) {
			const uint32_t rank = base + lane;
			const bool take = rank < nw;
			emit(take, take ? (uint32_t)surv[rank] : 0u, rank);
		}
	} else
	for(uint32_t base = 0; base < end; base += 64) {
		const uint32_t e = base + lane;
		const bool alive = e < end && !(head[row] > e);
		row += rstep; row -= row >= n ? n : 0u;
		const uint64_t mask = __ballot(alive);
		const uint32_t rank = w + __popcll(mask & ((1ull << lane) - 1ull));
		emit(alive && rank < 256, e, rank);
		w += __popcll(mask);
	}
	TUN_STAMP(3);
	used = wave_max_u32(used); maxlen = wave_max_u32(maxlen);
	for(uint32_t c = w + lane; c < 256; c += 64) { if(Tg) { T.off[c] = 0; T.len[c] = 0; } else { loff[c] = 0; llen[c] = 0; } }   // never on valid input
	__syncthreads();
	if(Tg) {
		if(lane == 0) { T.used = used; T.maxlen = maxlen; }
		if(WORDS_TO_HBM && to_hbm) return TunBuilt{used, maxlen};           // the words are there already
		typedef uint32_t v4_t __attribute__((ext_vector_type(4)));
		const uint32_t nv = ((used < TUN_TABLE_BYTES ? used : TUN_TABLE_BYTES) + 15u) >> 4;       // 16-byte vectors (both sides are 16-aligned)
		const v4_t *src4 = (const v4_t *)buf;
		v4_t *dst4 = (v4_t *)T.bytes;
		for(uint32_t i = lane; i < nv; i += 64) dst4[i] = src4[i];
	}
	return TunBuilt{used, maxlen};
}

