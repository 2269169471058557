// tun_tables.h — the Tunstall dictionary builder (crt::Tunstall::createDecodingTables2, src/tunstall.cpp:125-256) as a device
// function of ONE wave, shared by the decode kernels (k_tunstall.hip: K-TAB, K-STREAM) and the encoder's table kernel
// (k_encode.hip: k_enc_tables).  Include inside namespace corto_hip, after kernels_common.h / device_plan.h.
#pragma once

// The dictionary of one stream, by one wave.  Tg != null: written to the stream's TunTable in HBM (long streams: many workgroups
// decode from it).  Otherwise it stays in LDS - word bytes in tun_words(), offsets / lengths in loff / llen - for the same wave to
// decode from (k_tun_stream below).  Returns (used bytes, longest word).
struct TunBuilt { uint32_t used, maxlen; };
__shared__ __attribute__((aligned(16))) uint8_t g_tun_words[TUN_TABLE_BYTES];        // (one definition: both kernels below are single-wave workgroups)
__device__ __forceinline__ TunBuilt tun_tables_body(const TunStream &st, TunTable *Tg, uint16_t *loff, uint8_t *llen, const uint8_t *probs_override = nullptr) {
	const uint8_t *probs = probs_override ? probs_override : st.probs;   // nsym x (symbol, probability), sorted as stored in the stream
	TunTable &T = *Tg;                           // (only touched when Tg != null)
	uint8_t *buf = g_tun_words;
	const uint32_t n = st.nsym;                 // 2..255 (host guarantees)
	const uint32_t lane = threadIdx.x;

	__shared__ uint16_t eprob[TUN_ENTRY_CAP];   // entry e (creation order) lives in FIFO row e % n; probabilities are 16-bit ((a*b) >> 16 of 16-bit factors)
	__shared__ uint16_t eoff[TUN_ENTRY_CAP];
	__shared__ uint16_t elen[TUN_ENTRY_CAP];
	__shared__ uint16_t head[256];              // oldest not-yet-expanded entry of each row
	__shared__ uint16_t P[256];                 // probability << 8  (16.16-ish fixed point)
	__shared__ uint16_t pw[256];                // P0^k (successive (a*b)>>16), low-entropy seed only
	__shared__ uint8_t sym[256];

	for(uint32_t i = lane; i < n; i += 64) { sym[i] = probs[2*i]; P[i] = (uint32_t)probs[2*i + 1] << 8; }
	for(uint32_t i = lane; i < TUN_ENTRY_CAP; i += 64) { eprob[i] = 0; eoff[i] = 0; elen[i] = 0; }
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
			buf[b] = v;
		}
		if(lane == 0) {                         // P0^col, col = 1..count
			uint32_t v = p0; pw[1] = v;
			for(uint32_t c = 2; c <= count; c++) { v = (v*p0) >> 16; pw[c] = v; }
		}
		__syncthreads();
		for(uint32_t e = lane; e < count*n; e += 64) {
			const uint32_t col = e/n, row = e - col*n;
			if(row == 0) continue;
			eprob[e] = (uint16_t)(col == 0 ? (uint32_t)P[row] : ((uint32_t)pw[col]*(uint32_t)P[row]) >> 16);
			eoff[e] = (uint16_t)(row*count - col);
			elen[e] = (uint16_t)((col + 1) | (row << 8));     // high byte: row = last symbol (used by the n <= 64 path)
		}
		for(uint32_t k = lane; k < n; k += 64) head[k] = (uint16_t)(k == 0 ? (count - 1)*n : k);
		__syncthreads();
		if(lane == 0) { const uint32_t first = (count - 1)*n; eprob[first] = pw[count]; eoff[first] = 0; elen[first] = (uint16_t)count; }
		nwords = 1 + count*(n - 1);
		end = count*n;
		pos = total;
	} else {                                    // one-symbol words (tunstall.cpp:195-205)
		for(uint32_t i = lane; i < n; i += 64) {
			head[i] = (uint16_t)i; eprob[i] = P[i]; eoff[i] = (uint16_t)i; elen[i] = (uint16_t)(1u | (i << 8)); buf[i] = sym[i];
		}
		nwords = n; end = n; pos = n;
	}
	__syncthreads();

	const bool tree = n <= 64;
	const uint32_t seed_end = end, seed_bytes = pos;
	if(tree) {
		// Fast path (n <= 64 symbols, i.e. every stream the encoder really produces).  Lane r keeps row r's FIFO head (index,
		// probability, length) in registers, so picking the likeliest head is a register-only wave reduction.  A child is
		// recorded as (parent entry, row) - no bytes are copied while the dictionary grows; the 256 surviving words are
		// spelled out once at the end by walking up to the seed.  For the entries made here eoff[] holds the PARENT ENTRY and
		// the high byte of elen[] the row (= index of the last symbol).                                  tunstall.cpp:207-241
		uint32_t h = lane < n ? head[lane] : 0xFFFFu, hp = 0, hl = 0;
		if(lane < n && h < TUN_ENTRY_CAP) { hp = eprob[h]; hl = elen[h] & 255u; }
		const uint32_t myP = lane < n ? P[lane] : 0u;
		while(nwords < 256) {
			// likeliest head, first row wins ties, all-zero -> row 0: one DPP wave reduction + readlane broadcasts
			const uint32_t key = wave_max_u32(hp ? ((hp << 16) | (0xFFFFu - lane)) : 0u);
			const uint32_t best = (key >> 16) ? 0xFFFFu - (key & 0xFFFFu) : 0u;
			const uint32_t parent = (uint32_t)__builtin_amdgcn_readlane((int)h, (int)best);
			if(parent >= TUN_ENTRY_CAP) break;                                    // malformed probabilities
			const uint32_t pp = (uint32_t)__builtin_amdgcn_readlane((int)hp, (int)best), pl = (uint32_t)__builtin_amdgcn_readlane((int)hl, (int)best);
			const bool full = nwords + n > 255;                                   // dictionary fills up during this expansion: parent stays
			const uint32_t m = full ? 256 - nwords : n;
			const uint32_t tot = m*(pl + 1);
			if(end + m > TUN_ENTRY_CAP || pos + tot > TUN_TABLE_BYTES) break;     // where the reference's buffers would overflow
			if(lane < m) {
				const uint32_t e = end + lane, cp = (pp*myP) >> 16;
				eprob[e] = (uint16_t)cp; eoff[e] = (uint16_t)parent; elen[e] = (uint16_t)((pl + 1) | (lane << 8));
				if(h == e) { hp = cp; hl = pl + 1; }                              // the row's FIFO was empty: the child is its new head
			}
			asm volatile("" ::: "memory");                                        // one wave: LDS executes in program order
			if(!full && lane == best) {                                            // parent fully expanded: pop it
				h = parent + n;
				if(h < end + m && h < TUN_ENTRY_CAP) { hp = eprob[h]; hl = elen[h] & 255u; } else { hp = 0; hl = 0; }
			}
			end += m; pos += tot; nwords += n - 1;
		}
		if(lane < n) head[lane] = (uint16_t)min(h, 0xFFFFu);
		__syncthreads();
	} else
	while(nwords < 256) {                       // tunstall.cpp:207-241 (general path, n > 64)
		// likeliest FIFO head; first row wins ties; all-zero -> row 0.  key = prob:16 | (0xFFFF - row)
		uint32_t key = 0;
		for(uint32_t r = lane; r < n; r += 64) {
			const uint32_t h = head[r];
			const uint32_t p = h < TUN_ENTRY_CAP ? eprob[h] : 0u;
			const uint32_t k = p ? ((p << 16) | (0xFFFFu - r)) : 0u;
			key = k > key ? k : key;
		}
#pragma unroll
		for(int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(key, d, 64); key = o > key ? o : key; }
		const uint32_t best = (key >> 16) ? 0xFFFFu - (key & 0xFFFFu) : 0u;
		const uint32_t parent = head[best];
		if(parent >= TUN_ENTRY_CAP) break;      // malformed probabilities (reference: out-of-bounds read)
		const uint32_t pp = eprob[parent], po = eoff[parent], pl = elen[parent];
		const bool full = nwords + n > 255;     // dictionary fills up during this expansion: parent stays
		const uint32_t m = full ? 256 - nwords : n;
		const uint32_t tot = m*(pl + 1);
		if(end + m > TUN_ENTRY_CAP || pos + tot > TUN_TABLE_BYTES) break;
		for(uint32_t r = lane; r < m; r += 64) {
			const uint32_t e = end + r;
			eprob[e] = (pp*P[r]) >> 16;
			eoff[e] = (uint16_t)(pos + r*(pl + 1));
			elen[e] = (uint16_t)(pl + 1);
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
	for(uint32_t base = 0; base < end; base += 64) {
		const uint32_t e = base + lane;
		const bool alive = e < end && !(head[row] > e);
		row += rstep; row -= row >= n ? n : 0u;
		const uint64_t mask = __ballot(alive);
		const uint32_t rank = w + __popcll(mask & ((1ull << lane) - 1ull));
		const bool take = alive && rank < 256;
		const uint32_t len = take ? (uint32_t)elen[e] & 255u : 0u;
		const bool made = tree && take && e >= seed_end;                  // spelled out here
		const uint32_t incl = wave_inclusive_scan_u32(made ? len : 0u);
		const uint32_t off = made ? wpos + incl - len : take ? (uint32_t)eoff[e] : 0u;
		wpos += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
		if(take) {
			if(Tg) { T.off[rank] = (uint16_t)off; T.len[rank] = (uint8_t)len; } else { loff[rank] = (uint16_t)off; llen[rank] = (uint8_t)len; }
			used = off + len > used ? off + len : used;
			maxlen = len > maxlen ? len : maxlen;
		}
		if(made) {
			uint32_t cur = e, j = off + len;
			while(cur >= seed_end) { const uint32_t lr = elen[cur]; buf[--j] = sym[lr >> 8]; cur = eoff[cur]; }
			const uint32_t lr = elen[cur];
			buf[--j] = sym[lr >> 8];
			while(j > off) buf[--j] = A;
		}
		w += __popcll(mask);
	}
	used = wave_max_u32(used); maxlen = wave_max_u32(maxlen);
	for(uint32_t c = w + lane; c < 256; c += 64) { if(Tg) { T.off[c] = 0; T.len[c] = 0; } else { loff[c] = 0; llen[c] = 0; } }   // never on valid input
	__syncthreads();
	if(Tg) {
		if(lane == 0) { T.used = used; T.maxlen = maxlen; }
		const uint32_t ndw = (used + 3) >> 2;
		const uint32_t *src32 = (const uint32_t *)buf;
		uint32_t *dst32 = (uint32_t *)T.bytes;
		for(uint32_t i = lane; i < ndw; i += 64) dst32[i] = src32[i];
	}
	return TunBuilt{used, maxlen};
}

