// ssk_fast.hip -- FAST-9/16 corner score + 3x3 non-max suppression + candidate compaction, fused, one
// pass over the image (the HBM-bound scan of the front-end: every pixel of both images is read once).
// Replaces the detection half of cv::ORB as driven by stereo-vo's processNewImagePair (reference call
// sites EST.cpp:112,267); contract C2-C4 of DESIGN.md "Front-end contract".
//
// Mapping (gfx950, round 6): 256-thread workgroups over 128x32 output tiles, 7 resident per CU (70 VGPRs, 20 KiB of LDS), persistent: each XCD owns a
// contiguous range of tiles (halo re-reads stay in its L2) and its workgroups DRAW tile pairs from the XCD's counter, the next tile's pixels in
// flight during the phases of the one in work.  The kernel is bound by vector-instruction issue (~93 % of the SIMD cycles, ~580 instructions per
// wavefront and tile), not by HBM: every phase below is written for instruction count.  Per tile:
//   phase 0  the (128+16) x (32+8) pixel tile (ring + NMS halo, 4-byte aligned columns) goes from the prefetch registers to LDS (16-byte global
//            loads, 144 B per row coalesced; neighbouring tiles re-read the halo from L2);
//   phase A  item = 4 horizontally adjacent pixels: FIVE dword LDS reads (rows y-3, y, y+3) feed the necessary test on the 4 compass ring pixels
//            of all four (a 9-arc holds two adjacent compass points), on packed 16-bit pairs, the verdict left in the sign of a difference; a
//            lane keeps the survivor bits of its five items in two words (bright / dark) and the wavefront appends them all to an LDS pixel
//            queue in ONE step (a DPP scan of the counts, one ds atomic), each lane walking its set bits;
//   phase B  queued pixels, a quarter of the queue per wavefront: exact score s-1 of the polarity the compass test passed (max over the 16 arcs
//            of 9 of the min signed difference; opposite ring pixels share a register, 31 packed min / max); score > threshold is the corner
//            test -> 8-bit score tile in LDS, the corners packed to the front of the wavefront's quarter;
//   phase C  strict 3x3 NMS, a lane per corner, its 8 neighbours' scores by byte reads; survivors packed score<<24|(4095-y)<<12|(4095-x) into an
//            LDS list (in the image tile's storage); ONE global atomicAdd per workgroup reserves the output range in the image's candidate list.
// The candidate order in HBM is not deterministic (atomic reservation order); every consumer orders by
// the packed key, which is unique per pixel, so results are.
// Algorithmic bytes: W*H per image read once (+ 4 B per surviving corner written); roofline: HBM.
#include "ssk_frontend.h"
#include <cstdlib>

namespace {

#ifndef SSK_FAST_WAVES
#define SSK_FAST_WAVES 7                   // min waves per SIMD = workgroups per CU the register budget must allow
#endif
constexpr int TW = 128, TH = 32;           // output tile; tile origin x is a multiple of 4
constexpr int X_ORG = 28;                  // first tile column (<= SSK_EDGE, multiple of 4)
constexpr int IWV = TW + 16;               // LDS image tile: columns X0-8 .. X0+TW+7   (144 B = 36 dwords)
constexpr int IW = IWV + 4;                // row pitch 148 B = 37 dwords (odd: ring gathers of different rows spread over the banks)
constexpr int IH = TH + 8;                 //                 rows    Y0-4 .. Y0+TH+3
constexpr int SG = (TW + 8) / 4;           // score groups per row: columns X0-4 .. X0+TW+3
constexpr int SH = TH + 2;                 // score rows Y0-1 .. Y0+TH
constexpr int SP = IW;                     // score tile pitch (column index = gx - (X0-4)): the image tile's, so a pixel is ONE offset in both
constexpr int NQ = SG * SH * 4;            // pixels in the score region

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_u16x2(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ uint32_t as_u32x(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 as_s16x2(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ uint32_t as_u32(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }

// Round 5: the score of ONE polarity.  score = max(bright, -dark) with bright = max over the 16 arcs of 9 of the arc's min difference, -dark the same of
// the negated differences.  A pixel whose compass test failed for a polarity at threshold t has no arc of that polarity at t, i.e. that polarity's
// score is <= t -- it cannot be the maximum of a corner (whose other score is > t) and does not matter for a non-corner.  So the polarity the compass
// test passed is enough: half of fast_s_packed's min / max chains.  sgn2 = (+1, +1) or (-1, -1) packed: the ring differences come out signed for
// the polarity by ONE v_pk_mad_i16 per pair (r * sgn - c * sgn) -- what the plain subtraction cost.
// Round 6: A[k] = (d[k], d[k + 8]) -- OPPOSITE ring pixels share a register, so every step below computes an arc and the arc half a turn on with one
// packed instruction, and "the other half of the ring" is a swap of the two 16-bit lanes (op_sel of the packed instruction, no shuffle).
__device__ __forceinline__ void ring_diffs_signed(const uint8_t* p, uint32_t sgn2, uint32_t (&A)[8])
{
	const uint32_t c = p[0], cc = c | (c << 16);
	const uint32_t r[16] = {p[3 * IW], p[3 * IW + 1], p[2 * IW + 2], p[IW + 3], p[3], p[-IW + 3], p[-2 * IW + 2], p[-3 * IW + 1],
	                        p[-3 * IW], p[-3 * IW - 1], p[-2 * IW - 2], p[-IW - 3], p[-3], p[IW - 3], p[2 * IW - 2], p[3 * IW - 1]};
	const s16x2 s2 = as_s16x2(sgn2), ncs = as_s16x2(0u) - as_s16x2(cc) * s2;
#pragma unroll
	for (int k = 0; k < 8; ++k)      // (by name: the compiler rewrites r * s + (-c * s) as (r - c) * s, two packed instructions per pair)
		asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(A[k]) : "v"(r[k] | (r[k + 8] << 16)), "v"(sgn2), "v"(as_u32(ncs)));
}
// max over the 16 arcs of 9 of the arc's min.  With M(j) = min d[2j+1 .. 2j+8], the arcs from 2j and from 2j+1 are min(d[2j], M) and
// min(M, d[2j+9]): their max is min(M(j), max(d[2j], d[2j+9])) (the decomposition of cv's cornerScore<16>).  M(j) = min(d[2j+1], H[j+1], H[j+2], H[j+3],
// d[2j+8]) with H[m] = min(d[2m], d[2m+1]).  Lane pairs hold (j, j+4): 31 packed instructions (the doubling scheme of round 5: 47 + 8 shuffles).
__device__ __forceinline__ s16x2 swap16(s16x2 v) { return __builtin_shufflevector(v, v, 1, 0); }
__device__ __forceinline__ int fast_s_one(const uint32_t (&A)[8])
{
	s16x2 a[8], as[8], h[4], hs[4];
#pragma unroll
	for (int k = 0; k < 8; ++k) { a[k] = as_s16x2(A[k]); as[k] = swap16(a[k]); }         // E(i) = a[i] for i < 8, as[i - 8] above: (d[i], d[i + 8])
#pragma unroll
	for (int j = 0; j < 4; ++j) { h[j] = __builtin_elementwise_min(a[2 * j], a[2 * j + 1]); hs[j] = swap16(h[j]); }   // (H[j], H[j+4]); HH(m >= 4) = hs[m - 4]
	const s16x2 V = __builtin_elementwise_min(h[2], h[3]), W = __builtin_elementwise_min(hs[0], hs[1]);
	const s16x2 T[4] = {__builtin_elementwise_min(h[1], V), __builtin_elementwise_min(V, hs[0]), __builtin_elementwise_min(h[3], W), __builtin_elementwise_min(W, hs[2])};
	s16x2 best;
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		const s16x2 M = __builtin_elementwise_min(__builtin_elementwise_min(a[2 * j + 1], T[j]), as[2 * j]);            // E(2j+1), T, E(2j+8)
		const s16x2 R = __builtin_elementwise_min(M, __builtin_elementwise_max(a[2 * j], as[2 * j + 1]));               // E(2j), E(2j+9)
		best = j ? __builtin_elementwise_max(best, R) : R;
	}
	return max(max((int)best.x, (int)best.y), 0);
}

// queue entry -> score row and column of the pixel (the item map of phase A): bits 5..12 = the lane, bits 0..4 = 8 k + 1 + ra
__device__ __forceinline__ void decode_item(uint32_t code, int& sy, int& col)
{
	const int tq = (code >> 5) & 255, i = code & 31, k = i >> 3, ra = (i & 7) - 1;
	int g;
	if (ra == 4 && tq >= 64) { const int hq = tq - 64; sy = hq >> 1; g = (hq & 1) * (SG - 1); }
	else { sy = 8 * ra + (tq >> 5); g = 1 + (tq & 31); }
	col = 4 * g + k;
}

// n / d for a wave-uniform tile index with d fixed per launch: n * ceil(2^32 / d) >> 32, exact for n * d < 2^32
// (asserted by the launcher).  Runtime integer division has no hardware instruction: three of them per tile were
// ~15% of the kernel's VALU work.
__device__ __forceinline__ int div_magic(int n, uint32_t magic) { return magic ? (int)__umulhi((uint32_t)n, magic) : n; }   // magic 0: d = 1

__global__ __launch_bounds__(256, SSK_FAST_WAVES)
void fast_nms_kernel(const uint8_t* __restrict__ images, int w, int h, int pitch, int img_begin, int n_img,
                     int tiles_x, int tiles_y, uint32_t magic_img, uint32_t magic_x, int threshold,
                     uint32_t* __restrict__ cand, int cand_cap, int32_t* __restrict__ ncand, int32_t* __restrict__ tile_ctr)
{
	__shared__ __attribute__((aligned(16))) uint8_t s_img[IH * IW];
	__shared__ __attribute__((aligned(16))) uint8_t s_sc[(SH * SP + 15) & ~15];
	__shared__ uint16_t s_q1[NQ];
	// the survivors of the NMS (at most one per 2 x 2 pixels of the tile) are listed where the image tile was: phase C reads scores and queue only
	static_assert(sizeof(uint32_t) * (TW * TH / 4) <= sizeof(uint8_t) * IH * IW, "the output list fits the image tile's storage");
	uint32_t* const s_out = reinterpret_cast<uint32_t*>(s_img);
	__shared__ int s_q1n, s_on, s_base, s_next;

	const int tid = threadIdx.x;
#ifdef SSK_FAST_STAMPS
	__shared__ unsigned long long s_t[8];
	unsigned long long t_last = __builtin_amdgcn_s_memtime();
#define FSTAMP(i) do { __syncthreads(); if (tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); s_t[i] = t_ - t_last; t_last = t_; } } while (0)
#else
#define FSTAMP(i) do {} while (0)
#endif
	// persistent workgroups: the tiles are split into 8 contiguous ranges, one per XCD (blocks b and b+8 share an
	// XCD), so the blocks of an XCD sweep neighbouring tiles together and halo re-reads hit that XCD's L2
	const int tiles_img = tiles_x * tiles_y, n_tiles = tiles_img * n_img;
	const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3, stride = gridDim.x >> 3;
	const int per_x = (n_tiles + 7) >> 3;
	const int t_end = min(n_tiles, (xcd + 1) * per_x);
	constexpr int NCH = IH * (IWV / 16);                    // 16-byte chunks of a tile: 9 per row (144 B), 360 in all
	constexpr int NLD = (NCH + 255) / 256;                  // per thread (2)
	uint4 pre[NLD];

	auto issue_loads = [&](int t) {
		const int im = div_magic(t, magic_img), r = t - im * tiles_img, by = div_magic(r, magic_x), bx = r - by * tiles_x;
		const uint8_t* __restrict__ src = images + (size_t)(img_begin + im) * h * pitch;
		const int X0 = X_ORG + bx * TW, Y0 = SSK_EDGE + by * TH;
#pragma unroll
		for (int u = 0; u < NLD; ++u) {
			const int i = min(tid + 256 * u, NCH - 1);           // lanes past the last chunk re-read it (and do not store it): no branch around a load
			uint4 v;
			{
				const int row = i / 9, c = i - row * 9;
				const int gy = min(max(Y0 - 4 + row, 0), h - 1);
				const int gx = X0 - 8 + 16 * c;                   // a multiple of 4: global dwordx4 loads only need dword alignment
				// the last tile column runs past the row pitch (by < 128 bytes: into the next row, or into the 256 bytes of slack behind the
				// last image, ssk_kfbatch_create): those columns are >= w, and no pixel with a full ring reads them.  No branch here: the
				// dword-by-dword tail this used to have made the compiler wait for the first chunk before issuing the second.
				v = *reinterpret_cast<const uint4*>(src + (size_t)gy * pitch + gx);
			}
			pre[u] = v;
		}
	};

	// tile_ctr != NULL (round 6): persistent workgroups that DRAW their tiles -- the first one by position, the others from the XCD's counter, one
	// draw ahead of the prefetch (the atomic's round trip hides behind a whole tile).  A fixed stride visits only tile positions j + k * stride mod
	// (tiles per image): with 100 tiles per image and a stride of 224 a workgroup saw 25 of the 100 positions, always the same part of every
	// image, and the heavy parts made some workgroups finish last (measured: 15 % slower than one tile per workgroup at threshold 20).
	const bool draw = tile_ctr != nullptr;                    // (uniform)
	// a draw is a PAIR of neighbouring tiles (2 p, 2 p + 1 of the XCD's range): one atomic per two tiles -- at one per tile the eight counters, 16 ns a
	// draw each, were the floor of the kernel (0.40 ms per 204 800 tiles)
	int t = xcd * per_x + (draw ? 2 * jb : jb), t_next = draw ? t + 1 : t + stride;
	bool second = false;                                      // (uniform) the tile in work is the second of its pair
	int drawn = 0;                                            // lane 0: the draw in flight
	if (t < t_end) issue_loads(t);
	for (; t < t_end; t = t_next, t_next = !draw ? t_next + stride : (second ? t_next + 1 : s_next), second = !second) {
		// the prefetched tile goes to LDS; the next tile's loads are issued right after and fly during the phases
#pragma unroll
		for (int u = 0; u < NLD; ++u) {
			const int i = min(tid + 256 * u, NCH - 1);           // (the lanes past the last chunk write it once more: no branch around the wait for the loads)
			const int row = i / 9, c = i - row * 9;
			uint32_t* d = reinterpret_cast<uint32_t*>(s_img) + row * (IW / 4) + 4 * c;      // (row pitch 37 dwords: not 16-byte aligned)
			d[0] = pre[u].x; d[1] = pre[u].y; d[2] = pre[u].z; d[3] = pre[u].w;
		}
		static_assert((SH * SP + 15) / 16 <= 512, "score tile cleared by two 16-byte stores per lane");
		reinterpret_cast<uint4*>(s_sc)[tid] = make_uint4(0u, 0u, 0u, 0u);
		if (tid + 256 < (SH * SP + 15) / 16) reinterpret_cast<uint4*>(s_sc)[tid + 256] = make_uint4(0u, 0u, 0u, 0u);
		if (tid == 0) { s_q1n = 0; s_on = 0; }
		__syncthreads();
		if (t_next < t_end) issue_loads(t_next);
		// (built with -amdgpu-atomic-optimizer-strategy=None, build.py: the optimizer's wave-aggregated form reads the result back at once)
		if (draw && !second && tid == 0) drawn = atomicAdd(&tile_ctr[xcd * SSK_FAST_CTR_STRIDE], 1);
		FSTAMP(0);

		const int im = div_magic(t, magic_img), rr = t - im * tiles_img, by = div_magic(rr, magic_x), bx = rr - by * tiles_x;
		const int img = img_begin + im;
		const int X0 = X_ORG + bx * TW, Y0 = SSK_EDGE + by * TH;

		// ---- phase A (round 6): compass test, 4 pixels per lane and round.  Item map without divisions: rounds 0..3 cover score rows 8 ra + (tid >> 5),
		//      core groups 1 + (tid & 31) -- ONE address register, the rounds are immediate offsets; round 4 = rows 32, 33 on lanes 0..63 and the 68
		//      items of the two halo group columns on lanes 64..131.  The tests leave their result as the SIGN of a packed 16-bit difference
		//      (bright: c + t - min(max(N, S), max(E, W)) < 0; dark: max(min(N, S), min(E, W)) - (c - t) < 0; all values within +-510), whose high
		//      byte is 0xff / 0x00-0x01: one v_perm gathers the four pixels' high bytes, one v_and_or files bit (1 + ra) of each into the lane's
		//      survivor words (bit 8 k + 1 + ra: pixel k of round ra).  ~36 vector instructions per round (round 5: ~70). ----
		const bool edge_tile = (X0 + TW + 4 > w - 3) || (h - 2 - Y0 < SH);       // (uniform) some pixel of the score region has no full ring
		const int sy_lim = h - 2 - Y0;                                            // score rows sy < sy_lim have rows gy +- 3 inside the image
		const u16x2 t2 = as_u16x2((uint32_t)threshold * 0x00010001u);
		const uint32_t* const img32 = reinterpret_cast<const uint32_t*>(s_img);
		const int a_core = ((tid >> 5) + 3) * (IW / 4) + (tid & 31) + 2;
		const int hh = tid - 64;
		const int a_r4 = tid < 64 ? a_core + 4 * 8 * (IW / 4) : ((hh >> 1) + 3) * (IW / 4) + (hh & 1) * (SG - 1) + 1;
		uint32_t allb = 0u, alld = 0u;
#pragma unroll
		for (int ra = 0; ra < 5; ++ra) {
			if (ra == 4 && tid >= 64 + 2 * SH) break;
			const uint32_t* r = img32 + (ra < 4 ? a_core + ra * 8 * (IW / 4) : a_r4);      // dword of the group's four pixels
			const uint32_t C = r[0], L = r[-1], R = r[1];
			const uint32_t Dn = r[3 * (IW / 4)], Up = r[-3 * (IW / 4)];
			uint32_t rb[2], rd[2];
#pragma unroll
			for (int hsel = 0; hsel < 2; ++hsel) {
				const uint32_t sel = hsel ? 0x0c030c02u : 0x0c010c00u;   // bytes (0,1) or (2,3) -> the two 16-bit lanes
				const s16x2 c = as_s16x2(__builtin_amdgcn_perm(0u, C, sel));
				const s16x2 a0 = as_s16x2(__builtin_amdgcn_perm(0u, Dn, sel)), a8 = as_s16x2(__builtin_amdgcn_perm(0u, Up, sel));
				// x + 3 of pixels (0,1): bytes C3, R0; of (2,3): R1, R2.  x - 3 of (0,1): L1, L2; of (2,3): L3, C0
				const s16x2 a4 = as_s16x2(hsel ? __builtin_amdgcn_perm(0u, R, 0x0c020c01u) : __builtin_amdgcn_perm(R, C, 0x0c040c03u));
				const s16x2 a12 = as_s16x2(hsel ? __builtin_amdgcn_perm(C, L, 0x0c040c03u) : __builtin_amdgcn_perm(0u, L, 0x0c020c01u));
				const s16x2 hi = c + as_s16x2(as_u32x(t2)), lo = c - as_s16x2(as_u32x(t2));
				const s16x2 mb = __builtin_elementwise_min(__builtin_elementwise_max(a0, a8), __builtin_elementwise_max(a4, a12));
				const s16x2 xd = __builtin_elementwise_max(__builtin_elementwise_min(a0, a8), __builtin_elementwise_min(a4, a12));
				rb[hsel] = as_u32(hi - mb);
				rd[hsel] = as_u32(xd - lo);
			}
			uint32_t FB = __builtin_amdgcn_perm(rb[1], rb[0], 0x07050301u);      // high bytes of pixels 0..3: 0xff = passes
			uint32_t FD = __builtin_amdgcn_perm(rd[1], rd[0], 0x07050301u);
			if (edge_tile) {                                                      // (uniform) right / bottom image border: pixels without a full ring leave
				int sy, g;
				if (ra < 4 || tid < 64) { sy = 8 * ra + (tid >> 5); g = 1 + (tid & 31); }
				else { sy = hh >> 1; g = (hh & 1) * (SG - 1); }
				const int room = min(max(w - 3 - (X0 - 4 + 4 * g), 0), 4);
				const uint32_t rm = (sy < sy_lim && room > 0) ? (0xFFFFFFFFu >> (32 - 8 * room)) : 0u;
				FB &= rm; FD &= rm;
			}
			const uint32_t mk = 0x02020202u << ra;
			allb |= FB & mk;
			alld |= FD & mk;
		}
		{
			// the wavefront appends all its survivors to the pixel queue in one step: inclusive scan of the lanes' counts over the rows of 16
			// (row_shr 1, 2, 4, 8) and across them (row_bcast 15, 31), one ds atomic; a lane then walks its set bits.  Queue entry:
			// tid << 5 | bit index, bit 13 = passes as bright, bit 14 = as dark.
			uint32_t m = allb | alld;
			const int cnt = __popc(m);
			int x = cnt;
			x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);
			x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);
			x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);
			x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);
			x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);
			x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);
			const int total = __builtin_amdgcn_readlane(x, 63);
			if (total != 0) {                                      // (wave uniform)
				int qb = 0;
				if ((tid & 63) == 0) qb = atomicAdd(&s_q1n, total);
				qb = __builtin_amdgcn_readfirstlane(qb) + x - cnt;
				const uint32_t e0 = (uint32_t)tid << 5;
				while (m != 0u) {
					const int i = __builtin_ctz(m);
					m &= m - 1u;
					const uint32_t e = (__builtin_amdgcn_ubfe(alld, (uint32_t)i, 1u) << 14) | (__builtin_amdgcn_ubfe(allb, (uint32_t)i, 1u) << 13) | (e0 | (uint32_t)i);
					s_q1[qb++] = (uint16_t)e;
				}
			}
		}
		__syncthreads();

		FSTAMP(1);
		// ---- phase B: the exact score of every queued pixel; a corner is a pixel whose score exceeds the threshold (an arc of 9 all brighter
		//      than c + t or all darker than c - t  <=>  max over the arcs of the arc's min |difference| > t).  The separate corner test by 16-bit
		//      ring masks that used to stand in front of the score (round 2) cost as much per pixel as the packed score itself. ----
		// (round 6) each wavefront owns a quarter of the queue and packs its corners to the front of its own quarter -- the write index never
		// passes the read index, and nobody else touches the quarter: no second queue (9 KiB of LDS less: with the output list in the image tile's storage, 7 workgroups per CU), no ds atomic per step.
		const int q1n = s_q1n;
		const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
		const int seg = max(64, (q1n + 3) >> 2), seg0 = min(wv * seg, q1n), seg1 = min(seg0 + seg, q1n);   // (whole wavefronts first: 113 entries, the mean, are two wavefronts' work, not four quarter-filled ones)
		int nw = 0;                                                  // corners of this wavefront (uniform)
		for (int base = seg0; base < seg1; base += 64) {
			const int q = base + lane;
			bool corner = false;
			uint16_t code = 0;
			if (q < seg1) {
				code = s_q1[q];
				int sy, col;
				decode_item(code, sy, col);
				// the polarity the compass test passed (both: bright first, dark below -- a wave-uniform branch almost never taken)
				const bool pb = code & 0x2000u, pd = code & 0x4000u;
				const int at = (int)__umul24((unsigned)sy, (unsigned)SP) + col;     // the pixel in the score tile; + 3 rows + 4 columns in the image tile
				const uint8_t* px = s_img + at + 3 * IW + 4;
				uint32_t A[8];
				ring_diffs_signed(px, pb ? 0x00010001u : 0xFFFFFFFFu, A);
				int sc = fast_s_one(A);
				if (__ballot(pb && pd) != 0ull) {                        // (wave uniform)
					if (pb && pd) {
						uint32_t A2[8];
						ring_diffs_signed(px, 0xFFFFFFFFu, A2);
						sc = max(sc, fast_s_one(A2));
					}
				}
				corner = sc > threshold;
				if (corner) s_sc[at] = (uint8_t)(sc - 1);
			}
			const unsigned long long m = __ballot(corner);
			if (corner) s_q1[seg0 + nw + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = code;
			nw += __popcll(m);
		}
		__syncthreads();

		FSTAMP(2);
		FSTAMP(3);
		// ---- phase C: strict 3x3 NMS inside the emit region [31,W-31) x [31,H-31): a lane per CORNER (the wavefront's own corners: a few % of the
		// pixels), its 8 neighbours' scores by byte reads.  (A sweep of all 1024 pixel groups of the tile ran every wavefront through the
		// whole test with an eighth of its lanes holding a corner: ~350 of the kernel's ~1550 instructions per thread.) ----
		for (int base = 0; base < nw; base += 64) {
			const int q = base + lane;
			bool pr = false;
			uint32_t key = 0;
			if (q < nw) {
				int sy, col;                                             // col = score tile column: gx - (X0 - 4)
				decode_item(s_q1[seg0 + q], sy, col);
				const int gy = Y0 - 1 + sy, gx = X0 - 4 + col;
				const uint8_t* sc = s_sc + (int)__umul24((unsigned)sy, (unsigned)SP) + col;
				const int own = sc[0];
				if (own != 0 && sy >= 1 && sy <= TH && col >= 4 && col < TW + 4 && gy < h - SSK_EDGE && gx >= SSK_EDGE && gx < w - SSK_EDGE) {
					const int m0 = max(max((int)sc[-SP - 1], (int)sc[-SP]), max((int)sc[-SP + 1], (int)sc[-1]));
					const int m1 = max(max((int)sc[1], (int)sc[SP - 1]), max((int)sc[SP], (int)sc[SP + 1]));
					if (own > max(m0, m1)) {
						pr = true;
						key = ((uint32_t)own << 24) | ((uint32_t)(4095 - gy) << 12) | (uint32_t)(4095 - gx);
					}
				}
			}
			const unsigned long long m = __ballot(pr);
			if (m != 0ull) {                                             // (wave uniform)
				int b0 = 0;
				if (lane == 0) b0 = atomicAdd(&s_on, __popcll(m));
				b0 = __builtin_amdgcn_readfirstlane(b0);
				if (pr) s_out[b0 + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = key;
			}
		}
		__syncthreads();
		const int on = s_on;
		if (tid == 0 && on > 0) s_base = atomicAdd(&ncand[img], on);
		__syncthreads();
		if (on > 0) {
			const int base = s_base;
			uint32_t* dst = cand + (size_t)img * cand_cap;
			for (int i = tid; i < on; i += 256)
				if (base + i < cand_cap) dst[base + i] = s_out[i];
		}
		if (draw && !second && tid == 0) s_next = xcd * per_x + 2 * (stride + drawn);
		__syncthreads();     // LDS is rewritten by the next tile
#ifdef SSK_FAST_STAMPS
		FSTAMP(4);
		if (tid == 0 && (blockIdx.x % 9973) == 77) printf("fast stamps (cycles) tile %d: load %llu A %llu (q1 %d) B1 %llu (q2 of wave 0: %d) B %llu C+out %llu (%d)\n", t, s_t[0], s_t[1], s_q1n, s_t[2], nw, s_t[3], s_t[4], s_on);
#endif
	}
}

}  // namespace

int ssk_launch_fast(ssk_ctx* ctx, ssk_kfbatch* b, int img_begin, int img_end, int threshold)
{
	const int nimg = img_end - img_begin;
	if (nimg <= 0) return SSK_OK;
	// the candidate counts of the images and, behind the last image's (ssk_kfbatch_create), the eight tile counters of the persistent form -- a
	// cache line apart and more: the counters of all eight XCDs in ONE line made that line travel between the L2s at every draw (92 ns per draw)
	int32_t* const tile_ctr = b->ncand + (size_t)b->n_kf * 2 + SSK_FAST_CTR_STRIDE;
	if (img_end == b->n_kf * 2) SSK_HIP(ctx, hipMemsetAsync(b->ncand + img_begin, 0, sizeof(int32_t) * (size_t)(nimg + SSK_FAST_CTR_INTS), ctx->stream));
	else {
		SSK_HIP(ctx, hipMemsetAsync(b->ncand + img_begin, 0, sizeof(int32_t) * (size_t)nimg, ctx->stream));
		SSK_HIP(ctx, hipMemsetAsync(tile_ctr - SSK_FAST_CTR_STRIDE, 0, sizeof(int32_t) * SSK_FAST_CTR_INTS, ctx->stream));
	}
	const int ex = b->w - SSK_EDGE - X_ORG, eh = b->h - 2 * SSK_EDGE;     // columns X_ORG .. W-32, rows 31 .. H-32
	const int tiles_x = (ex + TW - 1) / TW, tiles_y = (eh + TH - 1) / TH;
	const long long n_tiles = (long long)tiles_x * tiles_y * nimg;
	// exact-division constants for the kernel's tile index decomposition (valid while n * d < 2^32)
	const uint32_t magic_img = tiles_x * tiles_y > 1 ? (uint32_t)((0x100000000ull + (unsigned long long)(tiles_x * tiles_y) - 1) / (unsigned long long)(tiles_x * tiles_y)) : 0u;
	const uint32_t magic_x = tiles_x > 1 ? (uint32_t)((0x100000000ull + (unsigned long long)tiles_x - 1) / (unsigned long long)tiles_x) : 0u;
	if (n_tiles * tiles_x * tiles_y >= 0x100000000ll)
		return ssk_set_error(ctx, SSK_ERR_CAPACITY, "ssk_launch_fast: %lld tiles of %d per image outside the tile index arithmetic", n_tiles, tiles_x * tiles_y);
	// default (round 6): as many persistent workgroups as the CUs hold at once (7 per CU: 20 KB of LDS, 70 VGPRs), drawing their tiles in pairs from
	// the XCD's counter; SSK_FAST_WG_PER_CU=k: k per CU, 0: one tile per workgroup (the form up to round 5); SSK_FAST_DRAW=0: a fixed stride instead
	// of the draw (measured slower: see the kernel)
	static const int wg_per_cu = [] {
		const char* e = getenv("SSK_FAST_WG_PER_CU");
		if (e) return atoi(e);
		int n = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fast_nms_kernel, 256, 0) != hipSuccess || n < 1) { (void)hipGetLastError(); n = 0; }
		return n;
	}();
	static const int draw = [] { const char* e = getenv("SSK_FAST_DRAW"); return e ? atoi(e) : 1; }();
	long long g = wg_per_cu > 0 ? (long long)ctx->n_cu * wg_per_cu : n_tiles;    // 0: one tile per workgroup
	if (g > n_tiles) g = n_tiles;
	g = (g + 7) / 8 * 8;
	ssk_stage_scope prof(ctx, SSK_STAGE_FAST);
	hipLaunchKernelGGL(fast_nms_kernel, dim3((unsigned)g), dim3(256), 0, ctx->stream, b->img, b->w, b->h, b->pitch, img_begin, nimg,
	                   tiles_x, tiles_y, magic_img, magic_x, threshold, b->cand, b->cand_cap, b->ncand, (wg_per_cu > 0 && draw && n_tiles >= 4 * g) ? tile_ctr : nullptr);    // few tiles (the per-keyframe drop-in: 200): one or a few per workgroup by stride
	SSK_LAUNCH_CHECK(ctx, "fast_nms_kernel");
	return SSK_OK;
}
