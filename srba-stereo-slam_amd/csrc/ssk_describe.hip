// ssk_describe.hip -- ORB orientation (intensity centroid) + steered BRIEF-256 on a 7x7 sigma=2 blurred
// patch, one wavefront per keypoint.  Contract C6-C9 of DESIGN.md "Front-end contract"; replaces the
// computeOrientation / GaussianBlur / computeDescriptors steps of cv::ORB (reached from EST.cpp:112,267).
//
// Mapping (gfx950): one 64-lane workgroup per keypoint; the workgroups of an image all run on ONE XCD (workgroup b sits on XCD b % 8), so the image's
// overlapping patches meet in one L2.  The kernel is bound by vector-instruction issue (97 % of the SIMD cycles, ~545 instructions per keypoint:
// vertical pass 210, horizontal 85, tests 75, moments 50, angle 25, sine / cosine 60, loads 40): every part below is written for instruction count.
//   * the 45x45 raw patch (radius 19 of rotated samples + 3 of blur support) is staged in LDS once with
//     ALIGNED dword loads (12 per row; the image is L2/MALL resident after the scan kernel);
//   * moments m10/m01 over the radius-15 disc: two lanes per patch row, v_dot4_u32_u8 against a weight table, DPP wave sums, integers;
//   * angle = degree polynomial in IEEE float; steering (cos,sin) = cv::ORB's (float)cos / (float)sin of that angle (ssk_sincos.h);
//   * the 8.8 fixed-point 7x7 blur is done ONCE per patch and separably inside LDS, on packed data: the horizontal
//     pass reads 6 dwords per 13-output row segment and forms each output with two v_dot4_u32_u8 (taps packed in
//     a constant), the vertical pass reads 10 dwords of the transposed u16 tile and uses four v_dot2_u32_u16 per
//     output (the kernel is VALU-issue bound: 8 keypoint waves per SIMD); no blurred image is ever materialised
//     in HBM;
//   * each lane evaluates 4 tests (8 samples: packed-float rotation, round-half-even by one magic-number add), packs 4 bits, lanes exchange
//     nibbles by shuffle, 32 lanes store one descriptor byte each.
// Data: 2 KB patch read per keypoint, 32 B + 4 B written.  VALU bound.
#include "ssk_frontend.h"
#include "../../include/ssk_sincos.h"

namespace {

constexpr int PR = 22;               // patch radius: 19 (rotated |(13,13)| rounded) + 3 (blur)
constexpr int PD = 2 * PR + 1;       // 45
#ifndef SSK_DESC_PP
#define SSK_DESC_PP 48
#endif
constexpr int PP = SSK_DESC_PP;      // raw patch pitch in bytes (>= 48: 12 dwords of data); pixel c of a row sits at byte shift + c
constexpr int BD = PD - 6;           // 39: blurred patch side
// (pitches and the alias below are macros for same-box A/B: tools/kernel_variant.sh ssk_describe)
#ifndef SSK_DESC_HT
#define SSK_DESC_HT 50                   // 25 dwords: the vertical pass's lanes (a column each) read conflict-free; 48 (24 dwords) put every fourth column into one bank
#endif
#ifndef SSK_DESC_BT
#define SSK_DESC_BT 40
#endif
constexpr int HT = SSK_DESC_HT;      // pitch (u16) of a COLUMN of the transposed horizontal-pass tile (45 rows used, dword aligned, an ODD number of dwords)
constexpr int BT = SSK_DESC_BT;      // pitch (bytes) of a column of the transposed blurred tile (39 rows used)

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_u16x2(uint32_t v) { return __builtin_bit_cast(u16x2, v); }

// Moments of the radius-15 disc (cv::ORB's umax table: |u| <= umax[|v|]): two lanes per row v (u in [-16, -1] and [0, 15]), the 16 pixels as four
// realigned dwords.  Per lane (row, half) and dword k the table holds the packed |u| weights with ZERO bytes outside the disc (m10 = +-sum by
// v_dot4_u32_u8) and the same bytes as ones (the row sum for m01): two 16-byte loads replace the mask arithmetic (round 5: ~50 instructions).
struct moment_table {
	uint32_t w[62][8];
	constexpr moment_table() : w()
	{
		const int umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
		for (int lane = 0; lane < 62; ++lane) {
			const int v = (lane >> 1) - 15, half = lane & 1, um = umax[v < 0 ? -v : v];
			for (int k = 0; k < 4; ++k) {
				uint32_t wt = 0, one = 0;
				for (int i = 0; i < 4; ++i) {
					const int u = half ? 4 * k + i : -16 + 4 * k + i;
					const int au = u < 0 ? -u : u;
					if (au <= um) { wt |= (uint32_t)au << (8 * i); one |= 1u << (8 * i); }
				}
				w[lane][k] = wt; w[lane][4 + k] = one;
			}
		}
	}
};
__constant__ moment_table c_mom = moment_table();

// sum over the wavefront's 64 lanes (all active), returned uniform: row_shr 1, 2, 4, 8 inside the rows of 16, row_bcast 15 / 31 across them
__device__ __forceinline__ int wave_sum(int x)
{
	x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);
	x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);
	x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);
	x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);
	x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);
	x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);
	return __builtin_amdgcn_readlane(x, 63);
}

__device__ __forceinline__ float atan2_deg(float y, float x)
{
	const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
	const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
	const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
	const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
	const float eps = (float)2.2204460492503131e-16;
	const float ax = fabsf(x), ay = fabsf(y);
	float a, c, c2;
	if (ax >= ay) {
		c = __fdiv_rn(ay, __fadd_rn(ax, eps));
		c2 = __fmul_rn(c, c);
		a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
	} else {
		c = __fdiv_rn(ax, __fadd_rn(ay, eps));
		c2 = __fmul_rn(c, c);
		a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
	}
	if (x < 0) a = __fsub_rn(180.f, a);
	if (y < 0) a = __fsub_rn(360.f, a);
	return a;
}

__global__ __launch_bounds__(64)
void describe_kernel(const uint8_t* __restrict__ images, int h, int pitch, int img_begin, int n_img, int n_feats, int max_feats,
                     const short2* __restrict__ kp_xy, const int32_t* __restrict__ nkp,
                     const int8_t* __restrict__ pattern, int steering, float* __restrict__ kp_angle, uint8_t* __restrict__ desc)
{
	__shared__ __attribute__((aligned(16))) uint8_t s_raw[PD * PP + 16];   // + slack: the dword window of the last segment may run 1 dword past a row
	__shared__ __attribute__((aligned(16))) uint16_t s_h[BD * HT + 8];     // horizontal pass, transposed: [column 3..41][row 0..44]
#ifndef SSK_DESC_ALIAS
#define SSK_DESC_ALIAS 1
#endif
#if SSK_DESC_ALIAS
	// (the blurred patch takes the raw patch's place: the raw rows are dead after the horizontal pass, and the kernel's occupancy is its LDS --
	//  7.6 KB per keypoint wave hold 20 of a CU's 32 wave slots)
	static_assert(BD * BT + 48 <= PD * PP + 16, "the blurred tile fits the raw patch's storage");
	uint8_t* s_b = s_raw + 48;                 // blurred patch, transposed: [column 3..41][row 3..41] (48 bytes in: the tests address it from (-1, -1))
#else
	__shared__ uint8_t s_b[BD * BT];           // blurred patch, transposed: [column 3..41][row 3..41]
#endif
	// workgroup b runs on XCD b % 8: an image's keypoints all go to ONE XCD (image i of the launch to XCD i % 8), so that the patches of an image --
	// they overlap 2 x and more -- meet in one L2 (as grid (keypoint, image) the eight L2s each fetched the image: 4.4 GB of L2 misses per
	// 1 M keypoints, r06_pmc_traffic.json, for 0.96 GB of images)
	const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
	const int grp = slot / n_feats, k = slot - grp * n_feats;
	if (8 * grp + xcd >= n_img) return;
	const int img = img_begin + 8 * grp + xcd;
	if (k >= nkp[img]) return;
	const int lane = threadIdx.x;
	const short2 xy = kp_xy[(size_t)img * max_feats + k];
	const int x = xy.x, y = xy.y;
	const uint8_t* __restrict__ src = images + (size_t)img * h * pitch;

	// raw patch: aligned dwords covering columns x-22 .. x+22 of rows y-22 .. y+22
	const int xs = x - PR, xa = xs & ~3, shift = xs - xa;
	if (lane < PD) {                                           // a lane per patch row: 3 x 16 bytes (global dwordx4 only needs dword alignment);
		const uint4* g = reinterpret_cast<const uint4*>(src + (size_t)(y - PR + lane) * pitch + xa);      // (a chunk per lane and turn cost 3 x the address arithmetic)
		const uint4 v0 = g[0], v1 = g[1], v2 = g[2];
		if constexpr (PP % 16 == 0) {
			uint4* d = reinterpret_cast<uint4*>(s_raw) + lane * (PP / 16);
			d[0] = v0; d[1] = v1; d[2] = v2;
		} else if constexpr (PP % 8 == 0) {
			uint2* d = reinterpret_cast<uint2*>(s_raw) + lane * (PP / 8);
			d[0] = uint2{v0.x, v0.y}; d[1] = uint2{v0.z, v0.w}; d[2] = uint2{v1.x, v1.y}; d[3] = uint2{v1.z, v1.w}; d[4] = uint2{v2.x, v2.y}; d[5] = uint2{v2.z, v2.w};
		} else {
			uint32_t* d = reinterpret_cast<uint32_t*>(s_raw) + lane * (PP / 4);
			d[0] = v0.x; d[1] = v0.y; d[2] = v0.z; d[3] = v0.w; d[4] = v1.x; d[5] = v1.y; d[6] = v1.z; d[7] = v1.w; d[8] = v2.x; d[9] = v2.y; d[10] = v2.z; d[11] = v2.w;
		}
	}
	__syncthreads();

	// moments over the radius-15 disc (c_mom above): m10 = sum u I, m01 = sum v I
	int m10 = 0, m01 = 0;
	if (lane < 62) {
		const int v = (lane >> 1) - 15, half = lane & 1;
		const int b = (PR + v) * PP + shift + PR - (half ? 0 : 16);
		const uint32_t* p = reinterpret_cast<const uint32_t*>(s_raw) + (b >> 2);
		const uint32_t sub = (uint32_t)(b & 3);
		const uint4 wt = reinterpret_cast<const uint4*>(c_mom.w[lane])[0], one = reinterpret_cast<const uint4*>(c_mom.w[lane])[1];
		const uint32_t W[4] = {wt.x, wt.y, wt.z, wt.w}, O[4] = {one.x, one.y, one.z, one.w};
		uint32_t D[5];
#pragma unroll
		for (int k = 0; k < 5; ++k) D[k] = p[k];
		uint32_t rs = 0, mu = 0;
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const uint32_t I = __builtin_amdgcn_alignbyte(D[k + 1], D[k], sub);
			mu = __builtin_amdgcn_udot4(I, W[k], mu, false);
			rs = __builtin_amdgcn_udot4(I, O[k], rs, false);
		}
		m10 = half ? (int)mu : -(int)mu;
		m01 = v * (int)rs;
	}
	// horizontal blur pass: item = (row r, segment s of 13 outputs).  The 19 input bytes come in as 6 dwords, are
	// shifted to byte 0 once (v_alignbyte), and every output is two v_dot4_u32_u8 against the packed taps
	// (18,34,49,55 | 49,34,18,0) of its realigned window; results (u16, <= 255*257) go out TRANSPOSED
	// (column-major) so that the vertical pass reads its 19 inputs as contiguous u16 pairs.
	// (segments of 10 outputs, 4 per row: 180 items fill the wavefront's three turns to 64 + 64 + 52 lanes; 135 items of 13 outputs left
	//  the third turn 7 lanes of the longer body)
	for (int it = lane; it < PD * 4; it += 64) {
		const int r = it >> 2, s = it & 3;
		const int b0 = shift + 10 * s;                           // first input byte of the segment inside the raw row
		const uint32_t* p = reinterpret_cast<const uint32_t*>(s_raw + r * PP) + (b0 >> 2);
		const uint32_t sub = (uint32_t)(b0 & 3);
		uint32_t D[5], E[4];
#pragma unroll
		for (int k = 0; k < 5; ++k) D[k] = p[k];
#pragma unroll
		for (int k = 0; k < 4; ++k) E[k] = __builtin_amdgcn_alignbyte(D[k + 1], D[k], sub);
		uint16_t* o = s_h + (10 * s) * HT + r;
		// output j = sum_t tap[t] byte[j + t], taps (18,34,49,55,49,34,18): the window of output 4q + ph starts at byte ph of E[q]; instead
		// of shifting the window to a dword start (12 v_alignbyte per segment) the TAPS are packed at the byte phase: 2 or 3 v_dot4
#pragma unroll
		for (int j = 0; j < 10; ++j) {
			const int q = j >> 2, ph = j & 3;
			uint32_t sum;
			if (ph == 0) sum = __builtin_amdgcn_udot4(E[q], 0x37312212u, __builtin_amdgcn_udot4(E[q + 1], 0x00122231u, 0u, false), false);
			else if (ph == 1) sum = __builtin_amdgcn_udot4(E[q], 0x31221200u, __builtin_amdgcn_udot4(E[q + 1], 0x12223137u, 0u, false), false);
			else if (ph == 2) sum = __builtin_amdgcn_udot4(E[q], 0x22120000u, __builtin_amdgcn_udot4(E[q + 1], 0x22313731u, __builtin_amdgcn_udot4(E[q + 2], 0x00000012u, 0u, false), false), false);
			else sum = __builtin_amdgcn_udot4(E[q], 0x12000000u, __builtin_amdgcn_udot4(E[q + 1], 0x31373122u, __builtin_amdgcn_udot4(E[q + 2], 0x00001222u, 0u, false), false), false);
			if (j < 9 || s < 3) o[j * HT] = (uint16_t)sum;       // (39 = 10 + 10 + 10 + 9 columns)
		}
	}
	m10 = wave_sum(m10); m01 = wave_sum(m01);
	__syncthreads();
	// vertical pass: item = (column j, segment s of 13 outputs): 19 contiguous u16 of the transposed tile as 10 dwords,
	// shifted to an even start once; an output is four v_dot2_u32_u16 on the pairs of its window.  The blurred patch
	// is stored transposed too (s_b[column * BT + row]); the tests index it accordingly.
	for (int it = lane; it < BD * 3; it += 64) {
		const int s = it / BD, j = it - BD * s;
		const int r0 = 13 * s;
		const uint32_t* p = reinterpret_cast<const uint32_t*>(s_h + j * HT) + (r0 >> 1);
		const uint32_t sub = (uint32_t)(r0 & 1) * 2u;
		uint32_t D[10], E[10];
#pragma unroll
		for (int k = 0; k < 10; ++k) D[k] = p[k];
#pragma unroll
		for (int k = 0; k < 9; ++k) E[k] = __builtin_amdgcn_alignbyte(D[k + 1], D[k], sub);
		E[9] = D[9] >> (8 * sub);
		uint8_t* o = s_b + j * BT + r0;
		// E[m] = (in[2m], in[2m+1]); output i = sum_t tap[t] in[i + t]: an even i starts on a pair, an odd one in the middle of one -- the taps
		// are paired accordingly instead of realigning the window (9 v_alignbyte per segment)
#pragma unroll
		for (int i = 0; i < 13; ++i) {
			const int a = i >> 1;
			uint32_t sum = 1u << 15;
			if ((i & 1) == 0) {
				sum = __builtin_amdgcn_udot2(as_u16x2(E[a + 3]), as_u16x2(18u), sum, false);
				sum = __builtin_amdgcn_udot2(as_u16x2(E[a + 2]), as_u16x2(49u | (34u << 16)), sum, false);
				sum = __builtin_amdgcn_udot2(as_u16x2(E[a + 1]), as_u16x2(49u | (55u << 16)), sum, false);
				sum = __builtin_amdgcn_udot2(as_u16x2(E[a]), as_u16x2(18u | (34u << 16)), sum, false);
			} else {
				sum = __builtin_amdgcn_udot2(as_u16x2(E[a + 3]), as_u16x2(34u | (18u << 16)), sum, false);
				sum = __builtin_amdgcn_udot2(as_u16x2(E[a + 2]), as_u16x2(55u | (49u << 16)), sum, false);
				sum = __builtin_amdgcn_udot2(as_u16x2(E[a + 1]), as_u16x2(34u | (49u << 16)), sum, false);
				sum = __builtin_amdgcn_udot2(as_u16x2(E[a]), as_u16x2(18u << 16), sum, false);
			}
			o[i] = (uint8_t)min(sum >> 16, 255u);
		}
	}

	// C7 steering: cv::ORB's (float)cos / (float)sin of the degree angle (shared with the oracle: include/ssk_sincos.h), or (steering 0,
	// rounds 1-2) the exact direction of the centroid
	const float ang = atan2_deg((float)m01, (float)m10);
	float ca = 1.f, sa = 0.f;
	if (steering) ssk_orb_sincos_deg(ang, &ca, &sa);
	else if (m01 != 0 || m10 != 0) {
		const double dx = (double)m10, dy = (double)m01;
		const double r = sqrt(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
		ca = (float)(dx / r);
		sa = (float)(dy / r);
	}
	if (lane == 0) kp_angle[(size_t)img * max_feats + k] = ang;
	__syncthreads();

	// lane L evaluates tests 4L..4L+3  (bits (4L..4L+3)&7 of byte L/2); blurred(px,py) = s_b[(px-3)*BT + py-3].
	// Round 6, by instruction count (the kernel is VALU-issue bound): the pattern comes as floats; a point's rotation is two packed multiplies and one
	// packed add ((x ca - y sa, x sa + y ca): IEEE, no contraction -- the same roundings as the scalar form); round-half-even and the integer
	// conversion are ONE packed add of 2^23 + 20 (an even offset keeps the ties of the sum on the ties of the coordinate; |coordinate| <= 19),
	// after which the low 24 bits of the float ARE 20 + coordinate, and v_mad_i32_i24 (which reads just those bits) forms the tile index.
	typedef float f32x2 __attribute__((ext_vector_type(2)));
	const uint8_t* bc = s_b + (PR - 3 - 20) * BT + (PR - 3 - 20);       // (the tile sits SSK_DESC_BOFF bytes into its storage: this offset stays >= 0)
	unsigned bits = 0;
	const float4* patf = reinterpret_cast<const float4*>(pattern + 1024) + 4 * lane;   // the lane's four tests (x0, y0, x1, y1)
	const f32x2 cs = {ca, sa}, nsc = {-sa, ca}, magic = {8388628.f, 8388628.f};      // (y (-sa) = -(y sa) exactly: the subtraction is an add)
#pragma unroll
	for (int t = 0; t < 4; ++t) {
		const float4 pt = patf[t];
		const f32x2 a0 = (f32x2){pt.x, pt.x} * cs, b0 = (f32x2){pt.y, pt.y} * nsc;
		const f32x2 a1 = (f32x2){pt.z, pt.z} * cs, b1 = (f32x2){pt.w, pt.w} * nsc;
		const f32x2 r0 = (a0 + b0) + magic;                                    // (ix0, iy0) + 2^23 + 20
		const f32x2 r1 = (a1 + b1) + magic;
		const int v0 = bc[__mul24(__float_as_int(r0.x), BT) + (__float_as_int(r0.y) & 0xFFFFFF)];
		const int v1 = bc[__mul24(__float_as_int(r1.x), BT) + (__float_as_int(r1.y) & 0xFFFFFF)];
		bits |= (unsigned)(v0 < v1) << t;
	}
	// byte j = tests 8j..8j+7 = lanes 2j (low nibble) and 2j+1 (high nibble)
	const unsigned other = __shfl_xor(bits, 1, 64);
	if ((lane & 1) == 0)
		desc[((size_t)img * max_feats + k) * 32 + (lane >> 1)] = (uint8_t)(bits | (other << 4));
}

}  // namespace

int ssk_launch_describe(ssk_ctx* ctx, ssk_kfbatch* b, int img_begin, int img_end, int n_feats)
{
	const int nimg = img_end - img_begin;
	if (nimg <= 0 || n_feats <= 0) return SSK_OK;
	ssk_stage_scope prof(ctx, SSK_STAGE_DESCRIBE);
	const long long n_wg = (long long)((nimg + 7) / 8) * 8 * n_feats;
	if (n_wg >= 0x7FFFFFFFll) return ssk_set_error(ctx, SSK_ERR_CAPACITY, "ssk_launch_describe: %d images x %d features outside the grid", nimg, n_feats);
	hipLaunchKernelGGL(describe_kernel, dim3((unsigned)n_wg), dim3(64), 0, ctx->stream, b->img, b->h, b->pitch,
	                   img_begin, nimg, n_feats, b->max_feats, b->kp_xy, b->nkp, ctx->d_pattern, ctx->orb_steering, b->kp_angle, b->desc);
	SSK_LAUNCH_CHECK(ctx, "describe_kernel");
	return SSK_OK;
}
