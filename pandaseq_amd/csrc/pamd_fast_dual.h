// pamd_fast_dual.h -- the lean kernel's geometry for SHORT reads: two pairs per wavefront (included by pamd_fast.hip, inside its
// namespace, behind fast_pair(): it uses that file's helpers).
//
// A pair of 2 x 150 reads keeps 50 of a wave's 64 lanes busy while it is seeded and 16 of them while its overlap is scored, and
// what a pair costs besides its vector work -- the scalar bookkeeping, a dozen exposed LDS round trips, the parking -- is that of a
// 2 x 250 pair (config 4: 461 vector + 227 scalar instructions for 2 708 bytes).  Here a wave takes TWO pairs per pass, one per
// half-wave: lanes 0..31 the first, lanes 32..63 the second, every phase of fast_pair() once for both --
//
//   stage      eight bases per lane (pack4, as the LONG geometry stages), 20 lanes a read; rows, code streams and N maps per half
//   seed       five k-mers per lane on windows of the code streams (BPL = 5 x 32 lanes = 160 bases); ONE set of k-mer tables for
//              both pairs: a table word is epoch | k-mer | half | position, so a word of the other pair never reads as a match
//              and the two pairs' words of one k-mer order as the reference's positions do within each pair
//   votes      the usual pair (one overlap, or one and a stray) is settled per half from ballots' halves
//   scan       one candidate per 16-lane DPP row as in fast_pair(): rows 0, 1 the first pair's two candidates, rows 2, 3 the second's
//   merge      32 bases of each pair per round, every round of the general (mixed) kind -- the rounds' kinds are a half's, not the wave's
//   park       two slots of the pending registers per pass; finish64() is fast_pair()'s
//
// It serves what config 4 is: reads of at most 160 bases, the five counting scorers, any-order sums, no primers or trims, no
// configured maximum overlap, no min_phred module.  A pair it does not settle -- an nt or quality outside the lean domain, a k-mer
// without a slot, votes for more than two overlaps or none, a k-mer twice in the forward read, a sum too close to the LOWQ
// threshold -- is assembled right behind the pass by fast_pair() itself (three bases per lane, the whole wave: redo_pair()), which
// hands on to the general kernel what it always did.  Results do not depend on which path made them
// (tests/test_lean_variants_gpu.py::test_two_pairs_per_wave).  Reference lines: assemble_seq() assembler.c:252-348, align()
// assembler.c:48-250.
//
// MEASURED (round 6, profiles/r06_two_pairs_per_wave_ab.txt): parity-green on the first run, and 22 % SLOWER than a pair per wave
// on config 4 (736 against 942 M pairs/s on one box).  Why: what shrinks per pair is the staging, the scan and the bookkeeping;
// the seeding and the merge cost per BASE (five k-mers a lane instead of three, ten mixed rounds of 32 instead of five of 64),
// every wave-uniform scalar of fast_pair() is a per-half value here (70 scalar registers spilled into vector lanes, selects where
// there were scalar operands), and a pair in eight is assembled twice.  About 1 000 vector instructions per pass + 0.13 x 461 for
// the redone pairs = 560 per pair against 461 -- and the kernel is bound by vector issue.  Kept, OFF by
// default (pamd_set_two_pairs_per_wave), as the measured answer to "two pairs per wave for reads of at most 160".

namespace dual {

// per-wave LDS (bytes): inside fast_pair()'s 7 632.  Rows per half; code streams and N maps per half (a block of 224 bytes: what
// lies before a forward N map and behind a reverse one reads as N); candidate flags (unused by this path: kept clear of the tables);
// the k-mer tables where nothing else ever lies (they are cleared every 127th pass: FK_EPOCH).  The merged nt rows are staged over
// the reads' rows, which are dead by then (every staged base of the merge is read before its first round).
constexpr unsigned D_F = 0, D_R = 640, D_ROW = 320;			// F: 0 / 320, R: 640 / 960
constexpr unsigned D_CS = 1280, D_CS_HALF = 224;			// code-stream block of a half
constexpr unsigned D_FC = 0, D_RC = 48, D_PAD0 = 96, D_FN = 112, D_RN = 160, D_PAD1 = 208;	// inside a block
constexpr unsigned D_NT = 0, D_NT_HALF = 640;				// merged nt staging: 352 bytes per half, over the rows
constexpr unsigned D_T1 = 2048, D_T2 = D_T1 + 4096, D_T3 = D_T2 + 1024, D_T4 = D_T3 + 256, D_END = D_T4 + 128;
static_assert(D_CS + 2 * D_CS_HALF <= D_T1 && D_END <= Geo<false, 10>::WAVE_BYTES && D_T1 % 16 == 0, "inside the short geometry's per-wave region");
constexpr int D_MAXLEN = 160, D_BPL = 5;
constexpr int D_NR = 10;	// rounds of 32 bases a merged row can span (2 x 160 - 2 * minoverlap < 320)
constexpr int D_NF = 5;		// rounds that can hold a forward base
struct LayCS {	// word_counts<>'s view of a half's block
	static constexpr unsigned FC = D_FC, RC = D_RC, FN = D_FN, RN = D_RN;
};

struct Carry2 {	// what a wave carries from one pass to the next: this lane's pair's lengths and its eight bases of either row
	int flen_v, rlen_v;
	uint4 f, r;
};
__device__ __forceinline__ void settle(Carry2 &c) {
	asm volatile("" : "+v"(c.flen_v), "+v"(c.rlen_v), "+v"(c.f.x), "+v"(c.f.y), "+v"(c.f.z), "+v"(c.f.w), "+v"(c.r.x), "+v"(c.r.y), "+v"(c.r.z), "+v"(c.r.w));
}
// lane's pair: q0 for lanes 0..31, q1 for lanes 32..63
__device__ __forceinline__ void load_pass(Carry2 &c, const PamdKernelParams &P, unsigned q0, unsigned q1, int lane) {
	const unsigned q = lane >= 32 ? q1 : q0;
	c.flen_v = P.flen[q];
	c.rlen_v = P.rlen[q];
	c.f = load_row16<false>(P.fwd + (u64) q * P.fwd_row_bytes, P.fwd_row_bytes, lane & 31);
	c.r = load_row16<false>(P.rev + (u64) q * P.rev_row_bytes, P.rev_row_bytes, lane & 31);
}

#define FK_INS(W, A, C) "v_cmp_gt_u32_e32 vcc, %[lim], " C "\n\ts_and_saveexec_b64 %[sv], vcc\n\tds_min_rtn_u32 " W ", " A ", " C "\n\ts_mov_b64 exec, %[sv]\n\t"
__device__ __forceinline__ void insert_all_live5(const unsigned (&addr)[5], unsigned (&carry)[5], unsigned limit) {
	unsigned was[5] = { 0u, 0u, 0u, 0u, 0u };
	u64 sv;
	asm volatile(FK_INS("%[w0]", "%[a0]", "%[c0]") FK_INS("%[w1]", "%[a1]", "%[c1]") FK_INS("%[w2]", "%[a2]", "%[c2]") FK_INS("%[w3]", "%[a3]", "%[c3]") FK_INS("%[w4]", "%[a4]", "%[c4]") "s_waitcnt lgkmcnt(0)"
		     : [w0] "+v"(was[0]), [w1] "+v"(was[1]), [w2] "+v"(was[2]), [w3] "+v"(was[3]), [w4] "+v"(was[4]), [sv] "=&s"(sv)
		     : [a0] "v"(addr[0]), [a1] "v"(addr[1]), [a2] "v"(addr[2]), [a3] "v"(addr[3]), [a4] "v"(addr[4]), [c0] "v"(carry[0]), [c1] "v"(carry[1]), [c2] "v"(carry[2]), [c3] "v"(carry[3]), [c4] "v"(carry[4]), [lim] "s"(limit)
		     : "vcc", "scc", "memory");
#pragma unroll
	for (int j = 0; j < 5; j++)
		carry[j] = was[j] > carry[j] ? was[j] : carry[j];
}
#undef FK_INS

// the two halves of a ballot
__device__ __forceinline__ unsigned lo32(u64 m) {
	return (unsigned) m;
}
__device__ __forceinline__ unsigned hi32(u64 m) {
	return (unsigned) (m >> 32);
}

// what the pass knows about one of its two pairs: wave-uniform scalars
struct HalfState {
	int flen, rlen;
	int status;
	bool redo;		// not settled here: fast_pair() assembles it behind the pass
	bool live;		// still being assembled by this pass
	int ncand, ov0, ov1, bo, examined, mism, degen, seqlen, uf, ur;
	bool slow;
	double best, qsum;
	u64 off;
};

// One pass: the pairs pairA (lanes 0..31) and pairB (lanes 32..63; haveB: there is one).  Parks every pair it settles in slots
// slot0 / slot0 + 1 of the pending registers; redo[k] tells the caller which it did not.
template <int ALGO>
__device__ __forceinline__ void fast_pass2(const PamdKernelParams &P, const double *ltab, unsigned char *wb, WaveArena &arena, Pend64 &pd, int slot0, bool &tables_dirty, unsigned &epoch, u64 pairA, u64 pairB, bool haveB, int lane,
					   Carry2 &cur, const Carry2 &nxt, const HotArgs &H, bool (&redo)[2]) {
	static_assert(counting_algo<ALGO>(), "the counting scorers");
	constexpr int BPL = D_BPL;
	const bool hi = lane >= 32;
	const int hl = lane & 31;
	const int minov = P.minoverlap;
	const uint4 fv = cur.f, rv = cur.r;
	const int flen_v = cur.flen_v, rlen_v = cur.rlen_v;	// this lane's pair's
	HalfState S[2];
	S[0].flen = __builtin_amdgcn_readlane(flen_v, 0);
	S[0].rlen = __builtin_amdgcn_readlane(rlen_v, 0);
	S[1].flen = __builtin_amdgcn_readlane(flen_v, 32);
	S[1].rlen = __builtin_amdgcn_readlane(rlen_v, 32);
#pragma unroll
	for (int k = 0; k < 2; k++) {
		HalfState &s = S[k];
		s.status = PAMD_ST_OK;
		s.redo = false;
		s.live = k == 0 || haveB;
		s.ncand = s.ov0 = s.ov1 = 0;
		s.bo = -1;
		s.examined = s.mism = s.degen = s.seqlen = 0;
		s.slow = false;
		s.best = s.qsum = 0.0;
		s.off = ~0ull;
		s.uf = s.flen;
		s.ur = s.rlen;
		if (s.live) {
			if (s.flen < 2 || s.rlen < 2 || (s.flen < s.rlen ? s.flen : s.rlen) < minov) {	// assembler.c:255-258, :289-292
				s.status = PAMD_ST_BADR;
				s.live = false;
			} else if (minov >= s.flen || minov >= s.rlen) {	// assembler.c:73-82
				s.status = PAMD_ST_NOALGN;
				s.live = false;
			}
		}
	}
	unsigned char *const cs = wb + D_CS + (hi ? D_CS_HALF : 0u);	// this lane's half's code-stream block
	const unsigned lF = lds_addr(wb) + D_F + (hi ? D_ROW : 0u), lR = lds_addr(wb) + D_R + (hi ? D_ROW : 0u);
	bool rotated = false;
	do {
		if (!S[0].live && !S[1].live)
			break;
		wave_sync();	// the previous pass's readers are done with this wave's LDS
		MARK("stage");
		// ---- stage: eight bases per lane (assembler.c reads panda_qual rows; the packing is stage8()'s) ----
		unsigned weird = 0;
		unsigned q4fa, q4fb, q4ra, q4rb;
		{
			int c0 = flen_v - 8 * hl, c1 = c0 - 4;
			c0 = c0 < 0 ? 0 : (c0 > 4 ? 4 : c0);
			c1 = c1 < 0 ? 0 : (c1 > 4 ? 4 : c1);
			unsigned ca, na, cb, nb, sk;
			pack4<true>(fv.x, fv.y, c0, ca, na, weird, sk, q4fa);
			pack4<true>(fv.z, fv.w, c1, cb, nb, weird, sk, q4fb);
			if (hl < 20)
				*reinterpret_cast<uint4 *>(wb + D_F + (hi ? D_ROW : 0u) + 16u * (unsigned) hl) = fv;
			if (hl < 24) {
				reinterpret_cast<uint16_t *>(cs + D_FC)[hl] = (uint16_t) (ca | (cb << 8));
				reinterpret_cast<uint16_t *>(cs + D_FN)[hl] = (uint16_t) (na | (nb << 8));
			}
			c0 = rlen_v - 8 * hl;
			c1 = c0 - 4;
			c0 = c0 < 0 ? 0 : (c0 > 4 ? 4 : c0);
			c1 = c1 < 0 ? 0 : (c1 > 4 ? 4 : c1);
			pack4<false>(rv.x, rv.y, c0, ca, na, weird, sk, q4ra);
			pack4<false>(rv.z, rv.w, c1, cb, nb, weird, sk, q4rb);
			if (hl < 20)
				*reinterpret_cast<uint4 *>(wb + D_R + (hi ? D_ROW : 0u) + 16u * (unsigned) hl) = rv;
			if (hl < 24) {
				reinterpret_cast<uint16_t *>(cs + D_RC)[hl] = (uint16_t) (ca | (cb << 8));
				reinterpret_cast<uint16_t *>(cs + D_RN)[hl] = (uint16_t) (na | (nb << 8));
			}
		}
		{
			const u64 wm = ballot(weird != 0);	// an IUPAC ambiguity code, nt 0, a quality outside 0..46
			if (S[0].live && lo32(wm) != 0) {
				S[0].redo = true;
				S[0].live = false;
			}
			if (S[1].live && hi32(wm) != 0) {
				S[1].redo = true;
				S[1].live = false;
			}
		}
		// the B-cliff length (assembler.c:176-177): the read's length unless it ends in a run of raw quality 2
		{
			auto cliff_len = [&](unsigned q4a, unsigned q4b, int len_v, int (&out)[2], int len0, int len1) {
				const int last = len_v - 1;
				const unsigned qw = (last & 4) ? q4b : q4a;
				const bool mine_is2 = hl == (last >> 3) && ((qw >> (8 * (last & 3))) & 0xFFu) == 2u;
				const u64 ends2 = ballot(mine_is2);
				out[0] = len0;
				out[1] = len1;
				if (__builtin_expect(ends2 != 0, 0)) {
					int c8 = len_v - 8 * hl;
					c8 = c8 < 0 ? 0 : (c8 > 8 ? 8 : c8);
					const unsigned not2 = ~(is2_mask(q4a) | (is2_mask(q4b) << 4)) & ((1u << c8) - 1u);
					const int mine = not2 ? hl * 8 + (31 - __builtin_clz(not2)) + 1 : 0;
					const u64 has = ballot(not2 != 0);
					if (lo32(ends2) != 0)
						out[0] = lo32(has) != 0 ? __builtin_amdgcn_readlane(mine, 31 - __builtin_clz(lo32(has))) : 0;
					if (hi32(ends2) != 0)
						out[1] = hi32(has) != 0 ? __builtin_amdgcn_readlane(mine, 63 - __builtin_clz(hi32(has))) : 0;
				}
			};
			int u[2];
			cliff_len(q4fa, q4fb, flen_v, u, S[0].flen, S[1].flen);
			S[0].uf = u[0];
			S[1].uf = u[1];
			cliff_len(q4ra, q4rb, rlen_v, u, S[0].rlen, S[1].rlen);
			S[0].ur = u[0];
			S[1].ur = u[1];
		}
		if (!S[0].live && !S[1].live)
			break;
		// ---- the k-mer tables: cleared when something else lay over them or the epochs are used up (FK_EPOCH) ----
		{
			const bool clear_now = epoch == 0u || tables_dirty;
			if (clear_now) {
				uint4 *t = reinterpret_cast<uint4 *>(wb + D_T1) + lane;
				unsigned e = EMPTY;
				asm volatile("" : "+v"(e));
				const uint4 e4 = make_uint4(e, e, e, e);
#pragma unroll
				for (unsigned i = 0; i < (D_T3 - D_T1) / 1024u; i++)
					t[64u * i] = e4;
				if (lane < (int) ((D_END - D_T3) / 16u))
					reinterpret_cast<uint4 *>(wb + D_T3)[lane] = e4;
			}
			epoch = clear_now ? 126u : epoch - 1u;
			tables_dirty = false;
		}
		wave_sync();
		MARK("seed-fwd");
		unsigned *const t1 = reinterpret_cast<unsigned *>(wb + D_T1), *const t2 = reinterpret_cast<unsigned *>(wb + D_T2);
		unsigned *const t3 = reinterpret_cast<unsigned *>(wb + D_T3), *const t4 = reinterpret_cast<unsigned *>(wb + D_T4);
		const unsigned l1 = lds_addr(t1), l2 = lds_addr(t2);
		const unsigned limit = (epoch + 1u) << 25;
		const unsigned halfbit = hi ? 0x100u : 0u;	// bit 8 of a table word: which of the two pairs it belongs to
		// the largest overlap looked at: the shorter read's length (assembler.c:79-80; no configured maximum here)
		const int shorter_v = flen_v < rlen_v ? flen_v : rlen_v;
		const unsigned size_v = (unsigned) (shorter_v - minov + 1);
		bool deep = false;
		// ---- seed: forward k-mers (assembler.c:92-101, misc.h:41-45): five positions per lane, p = 5 hl + j ----
		{
			const int fb = 2 * BPL * hl - 16;
			const unsigned *cw = reinterpret_cast<const unsigned *>(cs + D_FC) + (fb >> 5), *nwp = reinterpret_cast<const unsigned *>(cs + D_FN) + (fb >> 5);
			const unsigned c0 = cw[0], c1 = cw[1], n0 = nwp[0], n1 = nwp[1];
			const unsigned w = __builtin_amdgcn_alignbit(c1, c0, (unsigned) fb & 31u), nw = __builtin_amdgcn_alignbit(n1, n0, (unsigned) fb & 31u);
			unsigned carry[BPL], addr[BPL];
#pragma unroll
			for (int j = 0; j < BPL; j++) {
				const unsigned key = __builtin_amdgcn_alignbit(epoch, w << (14 - 2 * j), 16);	// epoch | k-mer
				const unsigned word = (key << 9) | halfbit | (unsigned) (BPL * hl + j);
				carry[j] = (nw & (0x15555u << (2 * j))) == 0 ? word : EMPTY;
				addr[j] = lshl_add<2>(mul24(key, HASH_C) >> 22, l1);
			}
			insert_all_live5(addr, carry, limit);
#pragma unroll
			for (int j = 0; j < BPL; j++)
				addr[j] = lshl_add<2>(__builtin_amdgcn_ubfe(mul24(carry[j] >> 9, HASH_C), 14, 8), l2);
			insert_all_live5(addr, carry, limit);
			u64 lost = 0;
#pragma unroll
			for (int j = 0; j < BPL; j++) {
				if (ballot(carry[j] < limit) != 0) {	// rare from here on
					deep = true;
					carry[j] = insert_level(t3, slot3(carry[j] >> 9), carry[j], limit);
					carry[j] = insert_level(t4, slot4(carry[j] >> 9), carry[j], limit);
					if (ballot(carry[j] < limit) != 0) {
						// a word found no slot: harmless when two smaller words of its k-mer AND PAIR are in (fast_pair())
						const unsigned k = carry[j] >> 9, kk = carry[j] & ~0xFFu;
						const int kept = (int) ((t1[slot1<10>(k)] ^ kk) < 0x100u) + (int) ((t2[slot2(k)] ^ kk) < 0x100u) + (int) ((t3[slot3(k)] ^ kk) < 0x100u) + (int) ((t4[slot4(k)] ^ kk) < 0x100u);
						lost |= ballot(carry[j] < limit && kept < 2);
					}
				}
			}
			// (a displaced word may be the other pair's: whose pair lost it says its half bit, not the lane's)
			if (__builtin_expect(lost != 0, 0)) {
				// which pairs lost a word: conservatively both when any did (rare: tandem repeats)
				if (S[0].live) {
					S[0].redo = true;
					S[0].live = false;
				}
				if (S[1].live) {
					S[1].redo = true;
					S[1].live = false;
				}
				break;
			}
		}
		wave_sync();
		MARK("seed-rev");
		// ---- seed: reverse k-mers vote for overlaps (assembler.c:104-110) ----
		{
			const int rb = 2 * BPL * hl;
			const unsigned *cw = reinterpret_cast<const unsigned *>(cs + D_RC) + (rb >> 5), *nwp = reinterpret_cast<const unsigned *>(cs + D_RN) + (rb >> 5);
			const unsigned c0 = cw[0], c1 = cw[1], n0 = nwp[0], n1 = nwp[1];
			const unsigned w = __builtin_amdgcn_alignbit(c1, c0, (unsigned) rb & 31u), nw = __builtin_amdgcn_alignbit(n1, n0, (unsigned) rb & 31u);
			const unsigned br = __builtin_bitreverse32(w);
			const unsigned top = (unsigned) (flen_v + rlen_v - minov - 1 - BPL * hl);	// assembler.c:107
			unsigned m1[BPL], m2[BPL], key[BPL];
			bool ok[BPL];
#pragma unroll
			for (int j = 0; j < BPL; j++) {
				key[j] = __builtin_amdgcn_alignbit(epoch, br << (2 * j), 16);
				ok[j] = (nw & (0x2AAAAu << (2 * j))) == 0;
				const unsigned prod = mul24(key[j], HASH_C), kk = (key[j] << 9) | halfbit;
				const unsigned e1 = lds_read_u32(lshl_add<2>(prod >> 22, l1)) ^ kk;
				const unsigned e2 = lds_read_u32(lshl_add<2>(__builtin_amdgcn_ubfe(prod, 14, 8), l2)) ^ kk;
				m1[j] = e1 < e2 ? e1 : e2;	// an entry of this k-mer and pair reads as its bare position (< 0x100)
				m2[j] = e1 < e2 ? e2 : e1;
			}
			if (deep) {
				unsigned r3[BPL], r4[BPL];
#pragma unroll
				for (int j = 0; j < BPL; j++) {
					r3[j] = lds_read_u32_early(lshl_add<2>(slot3(key[j]), lds_addr(t3)));
					r4[j] = lds_read_u32_early(lshl_add<2>(slot4(key[j]), lds_addr(t4)));
				}
#pragma unroll
				for (int j = 0; j < BPL; j++) {
					const unsigned kk = (key[j] << 9) | halfbit;
					const unsigned e3 = r3[j] ^ kk, e4 = r4[j] ^ kk;
					const unsigned l34 = e3 < e4 ? e3 : e4, h34 = e3 < e4 ? e4 : e3;
					const unsigned hh = m1[j] < l34 ? l34 : m1[j];
					m1[j] = m1[j] < l34 ? m1[j] : l34;
					const unsigned h = m2[j] < h34 ? m2[j] : h34;
					m2[j] = hh < h ? hh : h;
				}
			}
			u64 second = 0;
			unsigned idx1[BPL];
#pragma unroll
			for (int j = 0; j < BPL; j++) {
				const unsigned idx = top - (unsigned) j - m1[j];	// no match: m1 >= 0x100 ... wraps or lands beyond size (top - j < 0x200 here, see below)
				idx1[j] = ok[j] && m1[j] < 0x100u && idx < size_v ? idx : EMPTY;
				second |= ballot(ok[j]) & ballot(m2[j] < 0x100u);
			}
			// the usual pair's votes: one overlap, or that and a stray one, per half
			unsigned mine = EMPTY;
#pragma unroll
			for (int j = BPL - 1; j >= 0; j--)
				mine = idx1[j] != EMPTY ? idx1[j] : mine;
			const u64 have = ballot(mine != EMPTY);
			unsigned c0s[2] = { EMPTY, EMPTY };
			if (lo32(have) != 0)
				c0s[0] = (unsigned) __builtin_amdgcn_readlane((int) mine, __builtin_ctz(lo32(have)));
			if (hi32(have) != 0)
				c0s[1] = (unsigned) __builtin_amdgcn_readlane((int) mine, 32 + __builtin_ctz(hi32(have)));
			const unsigned c0v = hi ? c0s[1] : c0s[0];
			unsigned other = EMPTY;
#pragma unroll
			for (int j = 0; j < BPL; j++)
				other = (idx1[j] != EMPTY && idx1[j] != c0v) ? idx1[j] : other;
			const u64 oth = ballot(other != EMPTY);
			unsigned c1s[2] = { EMPTY, EMPTY };
			if (lo32(oth) != 0)
				c1s[0] = (unsigned) __builtin_amdgcn_readlane((int) other, __builtin_ctz(lo32(oth)));
			if (hi32(oth) != 0)
				c1s[1] = (unsigned) __builtin_amdgcn_readlane((int) other, 32 + __builtin_ctz(hi32(oth)));
			u64 third = 0;
			if (oth != 0) {
				const unsigned c1v = hi ? c1s[1] : c1s[0];
				bool t = false;
#pragma unroll
				for (int j = 0; j < BPL; j++)
					t |= idx1[j] != EMPTY && idx1[j] != c0v && idx1[j] != c1v;
				third = ballot(t);
			}
#pragma unroll
			for (int k = 0; k < 2; k++) {
				HalfState &s = S[k];
				if (!s.live)
					continue;
				const unsigned hv = k ? hi32(have) : lo32(have), tw = k ? hi32(second) : lo32(second), th = k ? hi32(third) : lo32(third);
				if (hv == 0 || tw != 0 || th != 0) {	// no vote at all (every overlap is a candidate), a k-mer twice in the forward read, a third overlap: fast_pair()'s
					s.redo = true;
					s.live = false;
					continue;
				}
				const unsigned a = c0s[k], b = c1s[k];
				if (b == EMPTY) {
					s.ncand = 1;
					s.ov0 = (int) a + minov;
				} else {
					s.ncand = 2;
					s.ov0 = (int) (a < b ? a : b) + minov;	// ascending, as the scan walks them (assembler.c:120)
					s.ov1 = (int) (a < b ? b : a) + minov;
				}
			}
		}
		if (!S[0].live && !S[1].live)
			break;
		MARK("scan");
		// ---- scan + argmax (assembler.c:118-143): a candidate per 16-lane row, 16 positions per lane ----
		{
			const u64 ovpack = (u64) (unsigned) S[0].ov0 | ((u64) (unsigned) S[0].ov1 << 16) | ((u64) (unsigned) S[1].ov0 << 32) | ((u64) (unsigned) S[1].ov1 << 48);
			const int myov = (int) ((unsigned) (ovpack >> (unsigned) (lane & 0x30)) & 0xFFFFu);
			const int ncand_v = hi ? (S[1].live ? S[1].ncand : 0) : (S[0].live ? S[0].ncand : 0);
			unsigned packed = word_counts<LayCS>(cs, flen_v, rlen_v, myov, 16 * (lane & 15), ((lane >> 4) & 1) < ncand_v);
			packed += (unsigned) __builtin_amdgcn_mov_dpp((int) packed, 0x128, 0xF, 0xF, false);	// row_ror:8
			packed += (unsigned) __builtin_amdgcn_mov_dpp((int) packed, 0x124, 0xF, 0xF, false);	// row_ror:4
			packed += (unsigned) __builtin_amdgcn_mov_dpp((int) packed, 0x122, 0xF, 0xF, false);	// row_ror:2
			packed += (unsigned) __builtin_amdgcn_mov_dpp((int) packed, 0x121, 0xF, 0xF, false);	// row_ror:1
			const int unknowns = (int) (packed & 1023u), matches = (int) (packed >> 10);
			const double sc = count_score<ALGO, true>(H.qual_nn, H.pmatch, H.pmismatch, P.count_table, flen_v, rlen_v, myov, unknowns, matches, myov);
#pragma unroll
			for (int k = 0; k < 2; k++) {
				HalfState &s = S[k];
				if (!s.live)
					continue;
				s.best = H.qual_nn * (double) (s.flen + s.rlen);	// assembler.c:60
				const double p0 = lane_f64(sc, 32 * k);
				if (p0 > s.best) {
					s.best = p0;
					s.bo = s.ov0;
				}
				if (s.ncand == 2) {
					const double p1 = lane_f64(sc, 32 * k + 16);
					if (p1 > s.best) {	// strict: the ascending scan keeps the smaller overlap on ties
						s.best = p1;
						s.bo = s.ov1;
					}
				}
				s.examined = s.ncand;
				s.slow = s.examined == (s.flen < s.rlen ? s.flen : s.rlen) - minov + 1;	// assembler.c:135
				if (s.bo < 0) {
					s.status = PAMD_ST_NOALGN;
					s.best = 0.0;
					s.live = false;
				}
			}
		}
		if (!S[0].live && !S[1].live)
			break;
		// ---- merge (assembler.c:146-247): 32 bases of either pair per round ----
		const int bo_v = hi ? S[1].bo : S[0].bo;	// (a half that is no longer live: whatever; nothing of it is stored or counted)
		const bool live_v = hi ? S[1].live : S[0].live;
		const int A_v = flen_v - bo_v, AF_v = flen_v, seqlen_v = flen_v + rlen_v - bo_v;	// forward-only [0, A), overlap [A, A + bo) = [A, flen), reverse-only up to seqlen
		unsigned ef[D_NF], er[D_NR];
		{
			const unsigned aF = lF + 2u * (unsigned) hl;							// round r: F[32 r + hl] ...
			const unsigned aR = lR + 2u * (unsigned) (rlen_v - 1 + A_v - hl) - 64u * (D_NR - 1);	// ... and R[rlen - 1 - (32 r + hl - A)]
			auto read_round = [&](auto rc) __attribute__((always_inline)) {
				constexpr int r = decltype(rc)::value;
				if constexpr (r < D_NF)
					ef[r] = lds_read_u16_bare<64 * r>(aF);
				er[r] = lds_read_u16_bare<64 * (D_NR - 1 - r)>(aR);
			};
			for_each_int(std::make_integer_sequence<int, D_NR>(), read_round);
		}
		MARK("rotate+alloc");
		cur = nxt;
		settle(cur);
		rotated = true;
		u64 off_v = 0;
		bool store_v = false;
#pragma unroll
		for (int k = 0; k < 2; k++) {
			HalfState &s = S[k];
			if (!s.live)
				continue;
			s.seqlen = s.flen + s.rlen - s.bo;	// len - 1
			const unsigned need16 = ((unsigned) s.seqlen + 15u) >> 4;
			u64 off;
			if (H.stride != 0) {
				off = (k ? pairB : pairA) * H.stride;
			} else {
				if (__builtin_expect(arena.cur + need16 > arena.end, 0)) {
					u64 got = 0;
					if (lane == 0) {
						u64 chunk = P.arena_chunk;
						unsigned long long *cursor = P.seq_cursor;
						asm volatile("global_atomic_add_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(got) : "v"(cursor), "v"(chunk) : "memory");
					}
					const unsigned chunk16 = (unsigned) (P.arena_chunk >> 4);
					arena.cur = (unsigned) uni((int) (unsigned) (got >> 4));
					arena.end = arena.cur + chunk16;
				}
				off = (u64) arena.cur << 4;
				arena.cur += need16;
				if (off + 16ull * need16 > H.capacity)
					off = ~0ull;
			}
			s.off = off;
			if ((k == 1) == hi) {
				off_v = off != ~0ull ? off : 0;
				store_v = off != ~0ull;
			}
		}
		uint8_t *const row_nt = H.nt + off_v;
		double *const row_p = H.p + off_v;
		const int lim_v = (live_v && store_v) ? seqlen_v : 0;
		const int uf_v = hi ? S[1].uf : S[0].uf, ur_v = hi ? S[1].ur : S[0].ur;
		const bool cliff = S[0].uf < S[0].flen || S[0].ur < S[0].rlen || S[1].uf < S[1].flen || S[1].ur < S[1].rlen;
		const unsigned lT = lds_addr(ltab), nN = lds_addr(wb) + D_NT + (hi ? D_NT_HALF : 0u) + (unsigned) hl;
		gbl_f64 *const gP = (gbl_f64 *) (row_p + hl);
		double qacc = 0.0;
		unsigned mismA = 0, mismB = 0, degA = 0, degB = 0;
		lds_wait_all(ef);
		lds_wait_all(er);
		wave_sync();	// (the merged nt bytes go where the reads' rows lay: every lane has its staged bases)
		MARK("merge");
		auto round = [&](auto rc, unsigned &qi_out, unsigned &nt_out) __attribute__((always_inline)) {
			constexpr int R = decltype(rc)::value;
			const unsigned e_r = er[R], e_f = ef[R < D_NF ? R : 0];
			const int j = 32 * R + hl;
			const bool ok = j < lim_v;
			const bool inF = R < D_NF && j < AF_v, inR = j >= A_v;
			const unsigned ab = e_f & e_r;
			const unsigned fq = e_f >> 8, rq = e_r >> 8;
			const bool match = (ab & 0xFFu) != 0;
			const u64 match_lanes = ballot(match);
			unsigned nt = match ? ab : (fq < rq ? e_r : e_f);	// assembler.c:212-220
			unsigned qi;
			if (ALGO == PAMD_ALGO_EA_UTIL) {	// algo_ea_util.c:58-67
				qi = PAMD_LT_SCORE + (fq > rq ? fq : rq);
			} else if (ALGO == PAMD_ALGO_FLASH) {	// algo_flash.c:63-80
				const unsigned d = fq > rq ? fq - rq : rq - fq;
				qi = PAMD_LT_SCORE + (match ? (fq > rq ? fq : rq) : (d < 2 ? 2 : d));
			} else {	// algo_simple_bayes.c:68-75, algo_uparse.c:68-75, algo_stitch.c:59-67
				qi = mat_index(match, fq, rq);
			}
			if (cliff) {
				asm volatile("; B-cliff");
				const bool fmask = j >= uf_v, rmask = rlen_v - 1 + A_v - j >= ur_v;	// assembler.c:192-207
				qi = rmask ? PAMD_LT_SCORE + fq : qi;
				qi = fmask ? PAMD_LT_SCORE + (rmask ? 47u : rq) : qi;
			}
			qi = inR ? qi : PAMD_LT_SCORE + fq;	// forward-only (assembler.c:162-173)
			qi = inF ? qi : PAMD_LT_SCORE + rq;	// reverse-only (assembler.c:231-243)
			nt = inR ? nt : e_f;
			nt = inF ? nt : e_r;
			qi_out = qi;
			nt_out = ok ? nt : 0u;
			const u64 mm = ballot(inF) & ballot(inR) & ~match_lanes;
			mismA += (unsigned) __popc(lo32(mm));
			mismB += (unsigned) __popc(hi32(mm));
		};
		auto trip = [&](auto r0c, auto nc) __attribute__((always_inline)) {
			constexpr int R0 = decltype(r0c)::value, N = decltype(nc)::value;
			unsigned qi[4], ntv[4];
			round(std::integral_constant<int, R0>(), qi[0], ntv[0]);
			round(std::integral_constant<int, R0 + 1>(), qi[1], ntv[1]);
			if constexpr (N > 2) {
				round(std::integral_constant<int, R0 + 2>(), qi[2], ntv[2]);
				round(std::integral_constant<int, R0 + 3>(), qi[3], ntv[3]);
			}
			double pv[4];
#pragma unroll
			for (int k = 0; k < N; k++)
				pv[k] = lds_read_f64(lT + 8u * qi[k]);
#pragma unroll
			for (int k = 0; k < N; k++) {
				lds_write_u8(nN + 32u * (R0 + k), ntv[k]);
				if (hl < lim_v - 32 * (R0 + k)) {
					store_stream(&gP[32 * (R0 + k)], pv[k]);
					qacc += pv[k];
				}
				const u64 dg = ballot((ntv[k] & 0xFFu) == 15u);	// PANDA_NT_IS_DEGN: only N here
				degA += (unsigned) __popc(lo32(dg));
				degB += (unsigned) __popc(hi32(dg));
			}
		};
		{
			const int longest = S[0].live ? (S[1].live ? (S[0].seqlen > S[1].seqlen ? S[0].seqlen : S[1].seqlen) : S[0].seqlen) : S[1].seqlen;
			typedef std::integral_constant<int, 2> two;
			typedef std::integral_constant<int, 4> four;
			trip(std::integral_constant<int, 0>(), four());
			if (longest > 128) {
				trip(std::integral_constant<int, 4>(), four());
				if (longest > 256)
					trip(std::integral_constant<int, 8>(), two());
			}
		}
		wave_sync();
		MARK("row-out+sum");
		if (hl * 16 < lim_v)
			store_stream(&((gbl_u128 *) row_nt)[hl], reinterpret_cast<const u32x4 *>(wb + D_NT + (hi ? D_NT_HALF : 0u))[hl]);
		{
			// each half's sum, any order: rows by DPP, the two rows of a half as scalars
			double t = qacc;
			t += dpp_f64<0xB1>(t);
			t += dpp_f64<0x4E>(t);
			t += dpp_f64<0x141>(t);
			t += dpp_f64<0x140>(t);
			const double sums[2] = { lane_f64(t, 0) + lane_f64(t, 16), lane_f64(t, 32) + lane_f64(t, 48) };
#pragma unroll
			for (int k = 0; k < 2; k++) {
				HalfState &s = S[k];
				if (!s.live)
					continue;
				s.mism = (int) (k ? mismB : mismA);
				s.degen = (int) (k ? degB : degA);
				s.qsum = sums[k];
				// within 1e-9 of the LOWQ threshold the three sums have to be made in the reference's order (assembler.c:171,226,
				// 236,244,334): fast_pair()'s guard, and its pair (the row written here stays behind in the arena as a hole)
				if (__builtin_expect(fabs(s.qsum - H.threshold * (double) (s.seqlen + 1)) <= 1e-9 * (double) (s.seqlen + 1), 0)) {
					s.redo = true;
					s.live = false;
				}
			}
		}
	} while (0);
	MARK("park");
	if (!rotated) {
		cur = nxt;
		settle(cur);
	}
#pragma unroll
	for (int k = 0; k < 2; k++) {
		const HalfState &s = S[k];
		redo[k] = s.redo;
		if (s.redo || (k == 1 && !haveB))
			continue;
		union {
			double d;
			unsigned u[2];
		} qq, bb;
		qq.d = s.qsum;
		bb.d = s.best;
		const bool merged = s.status == PAMD_ST_OK;
		const u64 off = merged ? s.off : ~0ull;
		const unsigned f_len_bo = (unsigned) (merged ? s.seqlen : 0) | ((unsigned) (s.bo < 0 ? 0 : s.bo) << 16), f_ex_mm = (unsigned) s.examined | ((unsigned) (merged ? s.mism : 0) << 16);
		const unsigned f_dg_fo = (unsigned) (merged ? s.degen : 0);
		const unsigned f_ro_st = ((unsigned) s.status << 16) | ((unsigned) s.slow << 24) | (63u << 26);
		const int slot = slot0 + k;
		asm volatile("s_mov_b32 m0, %6\n\t"
			     "v_writelane_b32 %0, %7, m0\n\t"
			     "v_writelane_b32 %1, %8, m0\n\t"
			     "v_writelane_b32 %2, %9, m0\n\t"
			     "v_writelane_b32 %3, %10, m0\n\t"
			     "v_writelane_b32 %4, %11, m0\n\t"
			     "v_writelane_b32 %5, %12, m0"
			     : "+v"(pd.off_lo), "+v"(pd.off_hi), "+v"(pd.q_lo), "+v"(pd.q_hi), "+v"(pd.b_lo), "+v"(pd.b_hi)
			     : "s"(slot), "s"(uni((int) (unsigned) off)), "s"(uni((int) (unsigned) (off >> 32))), "s"(uni((int) qq.u[0])), "s"(uni((int) qq.u[1])), "s"(uni((int) bb.u[0])), "s"(uni((int) bb.u[1])));
		asm volatile("s_mov_b32 m0, %4\n\t"
			     "v_writelane_b32 %0, %5, m0\n\t"
			     "v_writelane_b32 %1, %6, m0\n\t"
			     "v_writelane_b32 %2, %7, m0\n\t"
			     "v_writelane_b32 %3, %8, m0"
			     : "+v"(pd.len_bo), "+v"(pd.ex_mm), "+v"(pd.dg_fo), "+v"(pd.ro_st)
			     : "s"(slot), "s"(uni((int) f_len_bo)), "s"(uni((int) f_ex_mm)), "s"(uni((int) f_dg_fo)), "s"(uni((int) f_ro_st)));
	}
}

}	// namespace dual

// A pair the pass above did not settle, assembled by fast_pair() (three bases per lane, the whole wave).  Out of line: it runs for
// a pair in ten, and one copy serves both halves.  Nothing of the caller's goes in by reference -- a reference to the pending
// registers or to the kernel's parameter block would put THOSE in memory for the whole kernel: the parameters are read from the
// kernel argument segment itself, the pair's parked fields come back by value (lane `slot` holds them) with the arena's state.
struct Redone {
	Pend64 pd;
	unsigned arena_cur, arena_end;
};
template <int ALGO>
__device__ __attribute__((noinline)) Redone redo_pair(const PamdKernelParams *kernarg, const double *ltab, unsigned char *wb, unsigned arena_cur_v, unsigned arena_end_v, int slot_v, u64 pair_v, int lane) {
	// (a function's arguments arrive in vector registers: what is wave-uniform is made scalar again)
	const PamdKernelParams &P = *kernarg;
	auto uni_u64 = [](u64 v) -> u64 {
		return (u64) (unsigned) uni((int) (unsigned) v) | ((u64) (unsigned) uni((int) (unsigned) (v >> 32)) << 32);
	};
	const unsigned arena_cur = (unsigned) uni((int) arena_cur_v), arena_end = (unsigned) uni((int) arena_end_v);
	const int slot = uni(slot_v);
	const u64 pair = uni_u64(pair_v);
	// (read through a pointer these arrive in vector registers too; as scalars the row's offset is scalar arithmetic -- as vectors it
	// was a 64-bit multiply-add whose undefined upper addend the allocator put in a register with a bare LDS read pending, which
	// tools/isa_hazards.py refuses)
	const HotArgs H = { uni_f64(P.qual_nn), uni_f64(P.sb_pmatch), uni_f64(P.sb_pmismatch), uni_f64(P.threshold), uni_u64(P.seq_stride), uni_u64(P.seq_capacity),
			    reinterpret_cast<uint8_t *>(uni_u64((u64) P.seq_nt)), reinterpret_cast<double *>(uni_u64((u64) P.seq_p)) };
	unsigned epoch = 0;
	if (lane < 4)	// what lies before fast_pair()'s forward N map reads as N (the pass's rows lay there)
		reinterpret_cast<unsigned *>(wb + FK_PAD0)[lane] = 0x55555555u;
	Carry<false> one;
	load_pair<false, false>(one, P, (unsigned) pair, lane);
	settle(one);
	const int flen = uni(one.flen_v), rlen = uni(one.rlen_v);
	bool tables_dirty = true, flags_dirty = true;	// (the pass's tables lie elsewhere and hold its own words)
	Redone out;
	out.pd = Pend64{ 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u };
	WaveArena arena;
	arena.cur = arena_cur;
	arena.end = arena_end;
	LenMasks lmf = { -1, 0u, 0u }, lmr = { -1, 0u, 0u };
	fast_pair<ALGO, false, false, false, false, true, false>(P, ltab, nullptr, wb, arena, out.pd, slot, tables_dirty, pair, lane, flen, rlen, one, one, 0u, nullptr, H, flags_dirty, epoch, lmf, lmr);
	out.arena_cur = arena.cur;
	out.arena_end = arena.end;
	return out;
}

template <int ALGO>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4)))
void pamd_fast2_kernel(const PamdKernelParams P) {
	using namespace dual;
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	double *ltab = reinterpret_cast<double *>(lds);
	unsigned *lctr = reinterpret_cast<unsigned *>(lds + P.lds_ctr_off);
	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = uni(tid >> 6);
	const int nwaves = blockDim.x >> 6;
	for (unsigned i = tid; i < P.lds_table_doubles; i += blockDim.x)
		ltab[i] = P.tables[i];
	for (unsigned i = tid; i < PAMD_NUM_COUNTERS; i += blockDim.x)
		lctr[i] = 0;
	unsigned char *wb = lds + P.lds_primer_off + (size_t) wave * Geo<false, 10>::WAVE_BYTES;
	__syncthreads();
	WaveArena arena;
	arena.cur = 0;
	arena.end = 0;
	const unsigned total_waves = gridDim.x * (unsigned) nwaves;
	const unsigned n_pairs = (unsigned) P.n_pairs;
	constexpr unsigned F = 64u;
	const unsigned step_in = P.pair_blocks ? 1u : total_waves;
	unsigned pair = (blockIdx.x * (unsigned) nwaves + (unsigned) wave) * (P.pair_blocks ? F : 1u);
	const unsigned last = n_pairs - 1u;
	Carry2 cur;
	{
		const unsigned q0 = pair < n_pairs ? pair : last, q1 = pair + step_in < n_pairs ? pair + step_in : last;
		load_pass(cur, P, q0, q1, lane);
		settle(cur);
	}
	int npend = 0;
	bool tables_dirty = true;
	unsigned epoch = 0;
	Pend64 pd = { 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u };
	HotArgs H = { P.qual_nn, P.sb_pmatch, P.sb_pmismatch, P.threshold, P.seq_stride, P.seq_capacity, P.seq_nt, P.seq_p };
	unsigned group_first = pair;
	while (pair < n_pairs) {
		const unsigned pairB = pair + step_in;
		const bool haveB = pairB < n_pairs;	// (a block of a finish is 64 pairs, an even number: the second pair of a pass is in its first pair's block)
		const int took = haveB ? 2 : 1;
		const unsigned next = npend + took >= (int) F ? group_first + total_waves * F : pair + 2u * step_in;	// (strided: the same pair either way)
		Carry2 nxt;
		{
			const unsigned q0 = next < n_pairs ? next : last, q1 = next + step_in < n_pairs ? next + step_in : last;
			load_pass(nxt, P, q0, q1, lane);
		}
		// what hangs on the pad words may have been overwritten by a redone pair's layout: made per pass (four stores)
		if (lane < 4) {
			reinterpret_cast<unsigned *>(wb + D_CS + D_PAD0)[lane] = 0x55555555u;
			reinterpret_cast<unsigned *>(wb + D_CS + D_PAD1)[lane] = 0xAAAAAAAAu;
			reinterpret_cast<unsigned *>(wb + D_CS + D_CS_HALF + D_PAD0)[lane] = 0x55555555u;
			reinterpret_cast<unsigned *>(wb + D_CS + D_CS_HALF + D_PAD1)[lane] = 0xAAAAAAAAu;
		}
		bool redo[2];
		fast_pass2<ALGO>(P, ltab, wb, arena, pd, npend, tables_dirty, epoch, pair, pairB, haveB, lane, cur, nxt, H, redo);
		MARK("loop");
#pragma unroll 1
		for (int k = 0; k < took; k++) {
			if (__builtin_expect(redo[k], 0)) {
				wave_sync();
				const int slot = npend + k;
				const Redone r = redo_pair<ALGO>((const PamdKernelParams *) __builtin_amdgcn_kernarg_segment_ptr(), ltab, wb, arena.cur, arena.end, slot, k ? pairB : pair, lane);
				const bool mine = lane == slot;
				pd.off_lo = mine ? r.pd.off_lo : pd.off_lo;
				pd.off_hi = mine ? r.pd.off_hi : pd.off_hi;
				pd.q_lo = mine ? r.pd.q_lo : pd.q_lo;
				pd.q_hi = mine ? r.pd.q_hi : pd.q_hi;
				pd.b_lo = mine ? r.pd.b_lo : pd.b_lo;
				pd.b_hi = mine ? r.pd.b_hi : pd.b_hi;
				pd.len_bo = mine ? r.pd.len_bo : pd.len_bo;
				pd.ex_mm = mine ? r.pd.ex_mm : pd.ex_mm;
				pd.dg_fo = mine ? r.pd.dg_fo : pd.dg_fo;
				pd.ro_st = mine ? r.pd.ro_st : pd.ro_st;
				arena.cur = (unsigned) uni((int) r.arena_cur);
				arena.end = (unsigned) uni((int) r.arena_end);
				tables_dirty = true;	// (fast_pair()'s tables lie elsewhere in the wave's region)
				epoch = 0;
			}
		}
		npend += took;
		if (npend == (int) F) {
			finish64<false, true, false>(P, lctr, pd, npend, group_first, step_in, lane);
			npend = 0;
			group_first = next;
		}
		pair = next;
	}
	if (npend > 0)
		finish64<false, true, false>(P, lctr, pd, npend, group_first, step_in, lane);
	__syncthreads();
	if (P.counters != nullptr) {
		for (unsigned i = tid; i < PAMD_NUM_COUNTERS; i += blockDim.x) {
			const unsigned v = lctr[i];
			if (v != 0)
				atomicAdd(&P.counters[i], (unsigned long long) v);
		}
	}
}
