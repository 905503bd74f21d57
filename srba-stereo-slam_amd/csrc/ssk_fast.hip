// ssk_fast.hip -- FAST-9/16 corner score + 3x3 non-max suppression + candidate compaction, fused, one
// pass over the image (the HBM-bound scan of the front-end: every pixel of both images is read once).
// Replaces the detection half of cv::ORB as driven by stereo-vo's processNewImagePair (reference call
// sites EST.cpp:112,267); contract C2-C4 of DESIGN.md "Front-end contract".
//
// Mapping (gfx950): one 256-thread workgroup per 128x32 output tile (4 resident per CU: <= 128 VGPRs, 33 KiB
// of LDS); each XCD owns a contiguous range of tiles, so halo re-reads stay in its L2.  Per tile:
//   phase 0  the (128+16) x (32+8) pixel tile (ring + NMS halo, 4-byte aligned columns) is staged in LDS with
//            aligned dword loads, 144 B per row coalesced (neighbouring tiles re-read the halo from L2);
//   phase A  item = 4 horizontally adjacent pixels: FIVE dword LDS reads (rows y-3, y, y+3) feed the necessary
//            test on the 4 compass ring pixels of all four (a 9-arc holds two adjacent compass points); a lane keeps
//            the survivor bits of its five items and the wave appends them all to an LDS pixel queue in ONE step
//            (five ballots of the counts, one ds atomic), so the later phases run on dense lanes;
//   phase B  queued pixels: exact score s-1 (max over the 16 arcs of 9 of the min signed difference, by min/max doubling on packed
//            16-bit pairs); score > threshold is the corner test (round 3: the separate 16-bit ring-mask test in front of it cost as
//            much per pixel as the score) -> 8-bit score tile in LDS + second queue of real corners (a few % of the pixels);
//   phase C  strict 3x3 NMS, 4 pixels per lane behind one dword read of their scores (all-zero groups skip),
//            survivors packed score<<24|(4095-y)<<12|(4095-x) into an LDS list; ONE global atomicAdd per
//            workgroup reserves the output range in the image's candidate list.
// The candidate order in HBM is not deterministic (atomic reservation order); every consumer orders by
// the packed key, which is unique per pixel, so results are.
// Algorithmic bytes: W*H per image read once (+ 4 B per surviving corner written); roofline: HBM.
#include "ssk_frontend.h"
#include <cstdlib>

namespace {

#ifndef SSK_FAST_WAVES
#define SSK_FAST_WAVES 4                   // min waves per SIMD = workgroups per CU the register budget must allow
#endif
constexpr int TW = 128, TH = 32;           // output tile; tile origin x is a multiple of 4
constexpr int X_ORG = 28;                  // first tile column (<= SSK_EDGE, multiple of 4)
constexpr int IWV = TW + 16;               // LDS image tile: columns X0-8 .. X0+TW+7   (144 B = 36 dwords)
constexpr int IW = IWV + 4;                // row pitch 148 B = 37 dwords (odd: ring gathers of different rows spread over the banks)
constexpr int IH = TH + 8;                 //                 rows    Y0-4 .. Y0+TH+3
constexpr int SG = (TW + 8) / 4;           // score groups per row: columns X0-4 .. X0+TW+3
constexpr int SH = TH + 2;                 // score rows Y0-1 .. Y0+TH
constexpr int SP = IWV;                    // score tile pitch (column index = gx - (X0-4))
constexpr int NQ = SG * SH * 4;            // pixels in the score region

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_u16x2(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ uint32_t as_u32x(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 as_s16x2(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ uint32_t as_u32(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }

// FAST score of a corner from its 16 ring differences, on PACKED 16-bit lanes (v_pk_min/max_i16): A[k] holds
// (d[2k], d[2k+1]).  lo_n[i] = min of the n ring differences starting at i (doubling), score = max over the 16 arcs
// of 9 of the arc's min (bright) or of minus the arc's max (dark).  Half the instructions and a third of the live
// registers of the scalar form -- the scalar form made the whole kernel spill.
__device__ __forceinline__ int fast_s_packed(const uint32_t (&A)[8])
{
	s16x2 a[8], b[8], lo[8], hi[8];
#pragma unroll
	for (int k = 0; k < 8; ++k) a[k] = as_s16x2(A[k]);
#pragma unroll
	for (int k = 0; k < 8; ++k) b[k] = as_s16x2(__builtin_amdgcn_alignbyte(A[(k + 1) & 7], A[k], 2u));   // (d[2k+1], d[2k+2])
#pragma unroll
	for (int k = 0; k < 8; ++k) { lo[k] = __builtin_elementwise_min(a[k], b[k]); hi[k] = __builtin_elementwise_max(a[k], b[k]); }   // runs of 2
	s16x2 t[8], u[8];
#pragma unroll
	for (int k = 0; k < 8; ++k) { t[k] = __builtin_elementwise_min(lo[k], lo[(k + 1) & 7]); u[k] = __builtin_elementwise_max(hi[k], hi[(k + 1) & 7]); }   // runs of 4
#pragma unroll
	for (int k = 0; k < 8; ++k) { lo[k] = __builtin_elementwise_min(t[k], t[(k + 2) & 7]); hi[k] = __builtin_elementwise_max(u[k], u[(k + 2) & 7]); }     // runs of 8
#pragma unroll
	for (int k = 0; k < 8; ++k) { lo[k] = __builtin_elementwise_min(lo[k], a[(k + 4) & 7]); hi[k] = __builtin_elementwise_max(hi[k], a[(k + 4) & 7]); }   // arcs of 9
	s16x2 sb = lo[0], hm = hi[0];
#pragma unroll
	for (int k = 1; k < 8; ++k) { sb = __builtin_elementwise_max(sb, lo[k]); hm = __builtin_elementwise_min(hm, hi[k]); }
	const int bright = max(max((int)sb.x, (int)sb.y), 0), dark = min(min((int)hm.x, (int)hm.y), 255);
	return max(bright, -dark);
}

// the 16 ring pixels around p (LDS, pitch IW) minus the centre, in ring order, packed in pairs
__device__ __forceinline__ void ring_diffs_packed(const uint8_t* p, uint32_t (&A)[8])
{
	const uint32_t c = p[0], cc = c | (c << 16);
	const uint32_t r[16] = {p[3 * IW], p[3 * IW + 1], p[2 * IW + 2], p[IW + 3], p[3], p[-IW + 3], p[-2 * IW + 2], p[-3 * IW + 1],
	                        p[-3 * IW], p[-3 * IW - 1], p[-2 * IW - 2], p[-IW - 3], p[-3], p[IW - 3], p[2 * IW - 2], p[3 * IW - 1]};
#pragma unroll
	for (int k = 0; k < 8; ++k) A[k] = as_u32(as_s16x2(r[2 * k] | (r[2 * k + 1] << 16)) - as_s16x2(cc));
}

// Round 5: the score of ONE polarity.  score = max(bright, -dark) with bright = max over the 16 arcs of 9 of the arc's min difference, -dark the same of
// the negated differences.  A pixel whose compass test failed for a polarity at threshold t has no arc of that polarity at t, i.e. that polarity's
// score is <= t -- it cannot be the maximum of a corner (whose other score is > t) and does not matter for a non-corner.  So the polarity the compass
// test passed is enough: half of fast_s_packed's min / max chains.  sgn2 = (+1, +1) or (-1, -1) packed: the ring differences come out signed for
// the polarity by ONE v_pk_mad_i16 per pair (r * sgn - c * sgn) -- what the plain subtraction cost.
__device__ __forceinline__ void ring_diffs_signed(const uint8_t* p, uint32_t sgn2, uint32_t (&A)[8])
{
	const uint32_t c = p[0], cc = c | (c << 16);
	const uint32_t r[16] = {p[3 * IW], p[3 * IW + 1], p[2 * IW + 2], p[IW + 3], p[3], p[-IW + 3], p[-2 * IW + 2], p[-3 * IW + 1],
	                        p[-3 * IW], p[-3 * IW - 1], p[-2 * IW - 2], p[-IW - 3], p[-3], p[IW - 3], p[2 * IW - 2], p[3 * IW - 1]};
	const s16x2 s2 = as_s16x2(sgn2), ncs = as_s16x2(0u) - as_s16x2(cc) * s2;
#pragma unroll
	for (int k = 0; k < 8; ++k) A[k] = as_u32(as_s16x2(r[2 * k] | (r[2 * k + 1] << 16)) * s2 + ncs);
}
__device__ __forceinline__ int fast_s_one(const uint32_t (&A)[8])
{
	s16x2 a[8], b[8], lo[8], t[8];
#pragma unroll
	for (int k = 0; k < 8; ++k) a[k] = as_s16x2(A[k]);
#pragma unroll
	for (int k = 0; k < 8; ++k) b[k] = as_s16x2(__builtin_amdgcn_alignbyte(A[(k + 1) & 7], A[k], 2u));   // (d[2k+1], d[2k+2])
#pragma unroll
	for (int k = 0; k < 8; ++k) lo[k] = __builtin_elementwise_min(a[k], b[k]);                      // runs of 2
#pragma unroll
	for (int k = 0; k < 8; ++k) t[k] = __builtin_elementwise_min(lo[k], lo[(k + 1) & 7]);           // runs of 4
#pragma unroll
	for (int k = 0; k < 8; ++k) lo[k] = __builtin_elementwise_min(t[k], t[(k + 2) & 7]);            // runs of 8
#pragma unroll
	for (int k = 0; k < 8; ++k) lo[k] = __builtin_elementwise_min(lo[k], a[(k + 4) & 7]);           // arcs of 9
	s16x2 sb = lo[0];
#pragma unroll
	for (int k = 1; k < 8; ++k) sb = __builtin_elementwise_max(sb, lo[k]);
	return max(max((int)sb.x, (int)sb.y), 0);
}

// append the values of the lanes with pred to an LDS queue: one ds atomic per wave
__device__ __forceinline__ void wave_push(bool pred, uint16_t val, uint16_t* queue, int* counter)
{
	const unsigned long long m = __ballot(pred);
	if (m == 0ull) return;                                   // wave uniform
	const int lane = threadIdx.x & 63;
	int base = 0;
	if (lane == __ffsll((long long)m) - 1) base = atomicAdd(counter, __popcll(m));
	base = __shfl(base, __ffsll((long long)m) - 1, 64);
	if (pred) queue[base + __popcll(m & ((1ull << lane) - 1ull))] = val;
}
// the same for up to 4 flagged values per lane (4 ballots, still ONE atomic per wave)
template <typename T>
__device__ __forceinline__ void wave_push4(const bool (&pr)[4], const T (&val)[4], T* queue, int* counter)
{
	const unsigned long long m0 = __ballot(pr[0]), m1 = __ballot(pr[1]), m2 = __ballot(pr[2]), m3 = __ballot(pr[3]);
	const int n0 = __popcll(m0), n1 = __popcll(m1), n2 = __popcll(m2), n3 = __popcll(m3);
	const int n = n0 + n1 + n2 + n3;
	if (n == 0) return;                                      // wave uniform
	const int lane = threadIdx.x & 63;
	int base = 0;
	if (lane == 0) base = atomicAdd(counter, n);
	base = __shfl(base, 0, 64);
	const unsigned long long lt = (1ull << lane) - 1ull;
	if (pr[0]) queue[base + __popcll(m0 & lt)] = val[0];
	if (pr[1]) queue[base + n0 + __popcll(m1 & lt)] = val[1];
	if (pr[2]) queue[base + n0 + n1 + __popcll(m2 & lt)] = val[2];
	if (pr[3]) queue[base + n0 + n1 + n2 + __popcll(m3 & lt)] = val[3];
}

// n / d for a wave-uniform tile index with d fixed per launch: n * ceil(2^32 / d) >> 32, exact for n * d < 2^32
// (asserted by the launcher).  Runtime integer division has no hardware instruction: three of them per tile were
// ~15% of the kernel's VALU work.
__device__ __forceinline__ int div_magic(int n, uint32_t magic) { return magic ? (int)__umulhi((uint32_t)n, magic) : n; }   // magic 0: d = 1

__global__ __launch_bounds__(256, SSK_FAST_WAVES)
void fast_nms_kernel(const uint8_t* __restrict__ images, int w, int h, int pitch, int img_begin, int n_img,
                     int tiles_x, int tiles_y, uint32_t magic_img, uint32_t magic_x, int threshold,
                     uint32_t* __restrict__ cand, int cand_cap, int32_t* __restrict__ ncand)
{
	__shared__ __attribute__((aligned(16))) uint8_t s_img[IH * IW];
	__shared__ __attribute__((aligned(16))) uint8_t s_sc[SH * SP];
	__shared__ uint16_t s_q1[NQ];
	__shared__ uint8_t s_pol[(SG * SH + 3) & ~3];      // per 4-pixel group: compass survivors as bright (bits 0..3) / dark (bits 4..7) candidates
	__shared__ uint16_t s_q2[NQ];
	__shared__ uint32_t s_out[TW * TH / 4 + 64];
	__shared__ int s_q1n, s_q2n, s_on, s_base;

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
			const int i = tid + 256 * u;
			uint4 v = make_uint4(0u, 0u, 0u, 0u);
			if (i < NCH) {
				const int row = i / 9, c = i - row * 9;
				const int gy = min(max(Y0 - 4 + row, 0), h - 1);
				const int gx = X0 - 8 + 16 * c;                   // a multiple of 4: global dwordx4 loads only need dword alignment
				const uint8_t* p = src + (size_t)gy * pitch + gx;
				if (gx + 16 <= pitch) v = *reinterpret_cast<const uint4*>(p);
				else {                                            // the last tile column runs past the row pitch: dword by dword, zero beyond
					if (gx + 4 <= pitch) v.x = *reinterpret_cast<const uint32_t*>(p);
					if (gx + 8 <= pitch) v.y = *reinterpret_cast<const uint32_t*>(p + 4);
					if (gx + 12 <= pitch) v.z = *reinterpret_cast<const uint32_t*>(p + 8);
				}
			}
			pre[u] = v;
		}
	};

	int t = xcd * per_x + jb;
	if (t < t_end) issue_loads(t);
	for (; t < t_end; t += stride) {
		// the prefetched tile goes to LDS; the next tile's loads are issued right after and fly during the phases
#pragma unroll
		for (int u = 0; u < NLD; ++u) {
			const int i = tid + 256 * u;
			if (i < NCH) {
				const int row = i / 9, c = i - row * 9;
				uint32_t* d = reinterpret_cast<uint32_t*>(s_img) + row * (IW / 4) + 4 * c;      // (row pitch 37 dwords: not 16-byte aligned)
				d[0] = pre[u].x; d[1] = pre[u].y; d[2] = pre[u].z; d[3] = pre[u].w;
			}
		}
		for (int i = tid; i < SH * SP / 4; i += 256) reinterpret_cast<uint32_t*>(s_sc)[i] = 0u;
		if (tid == 0) { s_q1n = 0; s_q2n = 0; s_on = 0; }
		__syncthreads();
		if (t + stride < t_end) issue_loads(t + stride);
		FSTAMP(0);

		const int im = div_magic(t, magic_img), rr = t - im * tiles_img, by = div_magic(rr, magic_x), bx = rr - by * tiles_x;
		const int img = img_begin + im;
		const int X0 = X_ORG + bx * TW, Y0 = SSK_EDGE + by * TH;

		// ---- phase A: compass test, 4 pixels per lane; the lane keeps the survivor bits of its five groups (4 bits each) and the wavefront
		//      appends all of them to the pixel queue in ONE step (five ballots of the lanes' counts, one ds atomic): a push per round plus a
		//      second pass that expanded queued groups to pixels were a fifth of the kernel ----
		constexpr int NRA = (SG * SH + 255) / 256;
		static_assert(4 * NRA <= 32, "survivor bits of a lane");
		unsigned all = 0u;
#pragma unroll
		for (int ra = 0; ra < NRA; ++ra) {
			const int it = 256 * ra + tid;
			unsigned m4 = 0u;                                      // bit k: pixel k of the group passes the compass test
			if (it < SG * SH) {
				const int sy = it / SG, g = it - sy * SG;
				const int gy = Y0 - 1 + sy, gx0 = X0 - 4 + 4 * g;
				if (gy >= 3 && gy < h - 3) {
					const uint32_t* r = reinterpret_cast<const uint32_t*>(s_img) + (sy + 3) * (IW / 4) + (g + 1);   // dword of (gx0, gy)
					const uint32_t C = r[0], L = r[-1], R = r[1];
					const uint32_t Dn = r[3 * (IW / 4)], Up = r[-3 * (IW / 4)];
					const uint32_t E = (C >> 24) | (R << 8);          // bytes: x+3 of pixels 0..3
					const uint32_t Wd = (L >> 8) | (C << 24);         // bytes: x-3 of pixels 0..3
					// the four pixels as two packed 16-bit pairs: a 9-arc holds two ADJACENT compass points, one of {0,8} and
					// one of {4,12}, all brighter than c+t or all darker than c-t.  Saturating packed subtracts turn the
					// compares into "non-zero" and min() into AND: ~48 instructions for 4 pixels instead of ~100.
					const u16x2 t2 = as_u16x2((uint32_t)threshold * 0x00010001u);
					// (round 5: "N or S brighter than c + t" is max(N, S) > c + t, "darker" min(N, S) < c - t: a max / min and ONE saturating subtract per
					// pair of compass points instead of two subtracts and an OR; bright and dark survivors kept apart for the one-polarity score)
					uint32_t rb[2], rd[2];
#pragma unroll
					for (int hsel = 0; hsel < 2; ++hsel) {
						const uint32_t sel = hsel ? 0x0c030c02u : 0x0c010c00u;   // bytes (0,1) or (2,3) -> the two 16-bit lanes
						const u16x2 c = as_u16x2(__builtin_amdgcn_perm(0u, C, sel));
						const u16x2 a0 = as_u16x2(__builtin_amdgcn_perm(0u, Dn, sel)), a8 = as_u16x2(__builtin_amdgcn_perm(0u, Up, sel));
						const u16x2 a4 = as_u16x2(__builtin_amdgcn_perm(0u, E, sel)), a12 = as_u16x2(__builtin_amdgcn_perm(0u, Wd, sel));
						const u16x2 hi = c + t2, lo = __builtin_elementwise_sub_sat(c, t2);
						const u16x2 bv = __builtin_elementwise_sub_sat(__builtin_elementwise_max(a0, a8), hi);
						const u16x2 bh = __builtin_elementwise_sub_sat(__builtin_elementwise_max(a4, a12), hi);
						const u16x2 dv = __builtin_elementwise_sub_sat(lo, __builtin_elementwise_min(a0, a8));
						const u16x2 dh = __builtin_elementwise_sub_sat(lo, __builtin_elementwise_min(a4, a12));
						rb[hsel] = as_u32x(__builtin_elementwise_min(bv, bh));
						rd[hsel] = as_u32x(__builtin_elementwise_min(dv, dh));
					}
					// non-zero 16-bit lanes -> bits: min(x, 1) per lane, then (lo | hi >> 15) of each pair; gx0 >= X_ORG - 4 >= 3 always, only the last
					// tile column can pass w - 3: the pixels k < w - 3 - gx0 as one mask
					// (v_pk_min_u16 by name: the compiler rewrites min(x, 1) as x != 0 and spends four compares, four selects and two
					//  permutes on it)
					const uint32_t one2 = 0x00010001u;
					uint32_t b0, b1, d0, d1;
					asm("v_pk_min_u16 %0, %1, %2" : "=v"(b0) : "v"(rb[0]), "v"(one2));
					asm("v_pk_min_u16 %0, %1, %2" : "=v"(b1) : "v"(rb[1]), "v"(one2));
					asm("v_pk_min_u16 %0, %1, %2" : "=v"(d0) : "v"(rd[0]), "v"(one2));
					asm("v_pk_min_u16 %0, %1, %2" : "=v"(d1) : "v"(rd[1]), "v"(one2));
					const int room = min(max(w - 3 - gx0, 0), 4);
					const unsigned rm = (1u << room) - 1u;
					const unsigned bits_b = (((b0 | (b0 >> 15)) & 3u) | (((b1 | (b1 >> 15)) & 3u) << 2)) & rm;
					const unsigned bits_d = (((d0 | (d0 >> 15)) & 3u) | (((d1 | (d1 >> 15)) & 3u) << 2)) & rm;
					m4 = bits_b | bits_d;
					s_pol[it] = (uint8_t)(bits_b | (bits_d << 4));
				}
			}
			all |= m4 << (4 * ra);
		}
		{
			const unsigned cnt = __popc(all);                        // 0 .. 20
			const unsigned long long b0 = __ballot(cnt & 1u), b1 = __ballot(cnt & 2u), b2 = __ballot(cnt & 4u), b3 = __ballot(cnt & 8u), b4 = __ballot(cnt & 16u);
			const int total = __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2) + 8 * __popcll(b3) + 16 * __popcll(b4);
			if (total != 0) {                                      // (wave uniform)
				const int lane = tid & 63;
				const unsigned long long lt = (1ull << lane) - 1ull;
				const int before = __popcll(b0 & lt) + 2 * __popcll(b1 & lt) + 4 * __popcll(b2 & lt) + 8 * __popcll(b3 & lt) + 16 * __popcll(b4 & lt);
				int qb = 0;
				if (lane == 0) qb = atomicAdd(&s_q1n, total);
				qb = __shfl(qb, 0, 64) + before;
#pragma unroll
				for (int ra = 0; ra < NRA; ++ra) {
					const unsigned m4 = (all >> (4 * ra)) & 15u;
					const unsigned code0 = (unsigned)(256 * ra + tid) << 2;      // code = 4 * group + pixel
#pragma unroll
					for (int k = 0; k < 4; ++k)
						if (m4 & (1u << k)) s_q1[qb + __popc(m4 & ((1u << k) - 1u))] = (uint16_t)(code0 + k);
					qb += __popc(m4);
				}
			}
		}
		__syncthreads();

		FSTAMP(1);
		// ---- phase B: the exact score of every queued pixel; a corner is a pixel whose score exceeds the threshold (an arc of 9 all brighter
		//      than c + t or all darker than c - t  <=>  max over the arcs of the arc's min |difference| > t).  The separate corner test by 16-bit
		//      ring masks that used to stand in front of the score (round 2) cost as much per pixel as the packed score itself. ----
		const int q1n = s_q1n;
		for (int base = 0; base < q1n; base += 256) {
			const int q = base + tid;
			bool corner = false;
			uint16_t code = 0;
			if (q < q1n) {
				code = s_q1[q];
				const int it = code >> 2, k = code & 3;
				const int sy = it / SG, g = it - sy * SG;
				// the polarity the compass test passed (both: bright first, dark below -- a wave-uniform branch almost never taken)
				const unsigned pol = (unsigned)s_pol[it] >> k;
				const bool pb = pol & 1u, pd = pol & 16u;
				const uint8_t* px = s_img + (sy + 3) * IW + 4 * (g + 1) + k;
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
				if (corner) s_sc[sy * SP + 4 * g + k] = (uint8_t)(sc - 1);
			}
			wave_push(corner, code, s_q2, &s_q2n);
		}
		__syncthreads();
		const int q2n = s_q2n;

		FSTAMP(2);
		FSTAMP(3);
		// ---- phase C: strict 3x3 NMS inside the emit region [31,W-31) x [31,H-31): a lane per CORNER (the second queue again: a few % of the
		// pixels), its 8 neighbours' scores by byte reads.  (A sweep of all 1024 pixel groups of the tile ran every wavefront through the
		// whole test with an eighth of its lanes holding a corner: ~350 of the kernel's ~1550 instructions per thread.) ----
		for (int base = 0; base < q2n; base += 256) {
			const int q = base + tid;
			bool pr = false;
			uint32_t key = 0;
			if (q < q2n) {
				const int code = s_q2[q];
				const int it = code >> 2, k = code & 3;
				const int sy = it / SG, g = it - sy * SG;
				const int col = 4 * g + k;                               // score tile column: gx - (X0 - 4)
				const int gy = Y0 - 1 + sy, gx = X0 - 4 + col;
				const uint8_t* sc = s_sc + sy * SP + col;
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
				const int lane = tid & 63;
				int b0 = 0;
				if (lane == 0) b0 = atomicAdd(&s_on, __popcll(m));
				b0 = __shfl(b0, 0, 64);
				if (pr) s_out[b0 + __popcll(m & ((1ull << lane) - 1ull))] = key;
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
		__syncthreads();     // LDS is rewritten by the next tile
#ifdef SSK_FAST_STAMPS
		FSTAMP(4);
		if (tid == 0 && (blockIdx.x % 9973) == 77) printf("fast stamps (cycles) tile %d: load %llu A %llu (q1 %d) B1 %llu (q2 %d) B %llu C+out %llu (%d)\n", t, s_t[0], s_t[1], s_q1n, s_t[2], s_q2n, s_t[3], s_t[4], s_on);
#endif
	}
}

}  // namespace

int ssk_launch_fast(ssk_ctx* ctx, ssk_kfbatch* b, int img_begin, int img_end, int threshold)
{
	const int nimg = img_end - img_begin;
	if (nimg <= 0) return SSK_OK;
	SSK_HIP(ctx, hipMemsetAsync(b->ncand + img_begin, 0, sizeof(int32_t) * (size_t)nimg, ctx->stream));
	const int ex = b->w - SSK_EDGE - X_ORG, eh = b->h - 2 * SSK_EDGE;     // columns X_ORG .. W-32, rows 31 .. H-32
	const int tiles_x = (ex + TW - 1) / TW, tiles_y = (eh + TH - 1) / TH;
	const long long n_tiles = (long long)tiles_x * tiles_y * nimg;
	// exact-division constants for the kernel's tile index decomposition (valid while n * d < 2^32)
	const uint32_t magic_img = tiles_x * tiles_y > 1 ? (uint32_t)((0x100000000ull + (unsigned long long)(tiles_x * tiles_y) - 1) / (unsigned long long)(tiles_x * tiles_y)) : 0u;
	const uint32_t magic_x = tiles_x > 1 ? (uint32_t)((0x100000000ull + (unsigned long long)tiles_x - 1) / (unsigned long long)tiles_x) : 0u;
	if (n_tiles * tiles_x * tiles_y >= 0x100000000ll)
		return ssk_set_error(ctx, SSK_ERR_CAPACITY, "ssk_launch_fast: %lld tiles of %d per image outside the tile index arithmetic", n_tiles, tiles_x * tiles_y);
	// default: one tile per workgroup, tile ranges contiguous per XCD; SSK_FAST_WG_PER_CU=k switches to k persistent
	// workgroups per CU that prefetch their next tile (measured slower on MI355X: the dispatcher overlaps better)
	static const int wg_per_cu = [] { const char* e = getenv("SSK_FAST_WG_PER_CU"); return e ? atoi(e) : 0; }();
	long long g = wg_per_cu > 0 ? (long long)ctx->n_cu * wg_per_cu : n_tiles;    // 0: one tile per workgroup
	if (g > n_tiles) g = n_tiles;
	g = (g + 7) / 8 * 8;
	ssk_stage_scope prof(ctx, SSK_STAGE_FAST);
	hipLaunchKernelGGL(fast_nms_kernel, dim3((unsigned)g), dim3(256), 0, ctx->stream, b->img, b->w, b->h, b->pitch, img_begin, nimg,
	                   tiles_x, tiles_y, magic_img, magic_x, threshold, b->cand, b->cand_cap, b->ncand);
	SSK_LAUNCH_CHECK(ctx, "fast_nms_kernel");
	return SSK_OK;
}
