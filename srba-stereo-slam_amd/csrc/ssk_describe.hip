// ssk_describe.hip -- ORB orientation (intensity centroid) + steered BRIEF-256 on a 7x7 sigma=2 blurred
// patch, one wavefront per keypoint.  Contract C6-C9 of DESIGN.md "Front-end contract"; replaces the
// computeOrientation / GaussianBlur / computeDescriptors steps of cv::ORB (reached from EST.cpp:112,267).
//
// Mapping (gfx950): one 64-lane workgroup per keypoint.
//   * the 45x45 raw patch (radius 19 of rotated samples + 3 of blur support) is staged in LDS once with
//     ALIGNED dword loads (12 per row; the image is L2/MALL resident after the scan kernel);
//   * moments m10/m01 over the radius-15 disc: lanes own patch rows, wave shuffle reduction, integers;
//   * angle = degree polynomial in IEEE float; steering (cos,sin) = (m10,m01)/hypot in double;
//   * the 8.8 fixed-point 7x7 blur is done ONCE per patch and separably inside LDS: horizontal pass with a
//     sliding window over 13-output row segments (u16 results), vertical pass over 13-output column segments
//     (u8 results) -- ~75 LDS accesses per lane instead of 392 for blurring each of the 512 samples alone;
//     no blurred image is ever materialised in HBM;
//   * each lane evaluates 4 tests (8 samples), packs 4 bits, lanes exchange nibbles by shuffle, 32 lanes store
//     one descriptor byte each.
// Data: 2 KB patch read per keypoint, 32 B + 4 B written.  LDS/VALU bound.
#include "ssk_frontend.h"

namespace {

constexpr int PR = 22;               // patch radius: 19 (rotated |(13,13)| rounded) + 3 (blur)
constexpr int PD = 2 * PR + 1;       // 45
constexpr int PP = 48;               // raw patch pitch in bytes (12 dwords); pixel c of a row sits at byte shift + c
constexpr int BD = PD - 6;           // 39: blurred patch side
constexpr int HP = 40;               // pitch (elements) of the horizontal-pass (u16) and blurred (u8) tiles

__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

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
void describe_kernel(const uint8_t* __restrict__ images, int h, int pitch, int img_begin, int max_feats,
                     const short2* __restrict__ kp_xy, const int32_t* __restrict__ nkp,
                     const int8_t* __restrict__ pattern, float* __restrict__ kp_angle, uint8_t* __restrict__ desc)
{
	__shared__ __attribute__((aligned(16))) uint8_t s_raw[PD * PP];
	__shared__ uint16_t s_h[PD * HP];          // horizontal pass: rows 0..44, columns 3..41 of the patch
	__shared__ uint8_t s_b[BD * HP];           // blurred patch: rows/columns 3..41
	const int img = img_begin + blockIdx.y;
	const int k = blockIdx.x;
	if (k >= nkp[img]) return;
	const int lane = threadIdx.x;
	const short2 xy = kp_xy[(size_t)img * max_feats + k];
	const int x = xy.x, y = xy.y;
	const uint8_t* __restrict__ src = images + (size_t)img * h * pitch;

	// raw patch: aligned dwords covering columns x-22 .. x+22 of rows y-22 .. y+22
	const int xs = x - PR, xa = xs & ~3, shift = xs - xa;
	for (int i = lane; i < PD * (PP / 4); i += 64) {
		const int r = i / (PP / 4), c = i - r * (PP / 4);
		reinterpret_cast<uint32_t*>(s_raw)[i] = *reinterpret_cast<const uint32_t*>(src + (size_t)(y - PR + r) * pitch + xa + 4 * c);
	}
	__syncthreads();

	// moments over the radius-15 disc: lane v handles row v-15 (31 rows)
	int m10 = 0, m01 = 0;
	if (lane < 31) {
		const int v = lane - 15;
		const int um = c_umax[v < 0 ? -v : v];
		const uint8_t* p = s_raw + (PR + v) * PP + shift + PR;
		int rs = 0;
		for (int u = -um; u <= um; ++u) { const int I = p[u]; m10 += u * I; rs += I; }
		m01 = v * rs;
	}
	// horizontal blur pass: item = (row r, segment s of 13 outputs), sliding 7-tap window
	for (int it = lane; it < PD * 3; it += 64) {
		const int r = it / 3, s = it - 3 * r;
		const uint8_t* p = s_raw + r * PP + shift + 13 * s;      // patch columns 13s .. 13s+18
		uint16_t* o = s_h + r * HP + 13 * s;
		int w0 = p[0], w1 = p[1], w2 = p[2], w3 = p[3], w4 = p[4], w5 = p[5];
#pragma unroll
		for (int j = 0; j < 13; ++j) {
			const int w6 = p[j + 6];
			o[j] = (uint16_t)(18 * (w0 + w6) + 34 * (w1 + w5) + 49 * (w2 + w4) + 55 * w3);
			w0 = w1; w1 = w2; w2 = w3; w3 = w4; w4 = w5; w5 = w6;
		}
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) { m10 += __shfl_xor(m10, o, 64); m01 += __shfl_xor(m01, o, 64); }
	__syncthreads();
	// vertical pass: item = (column j, segment s of 13 outputs)
	for (int it = lane; it < BD * 3; it += 64) {
		const int s = it / BD, j = it - BD * s;
		const uint16_t* p = s_h + (13 * s) * HP + j;              // rows 13s .. 13s+18 of column j
		uint8_t* o = s_b + (13 * s) * HP + j;
		int w0 = p[0], w1 = p[HP], w2 = p[2 * HP], w3 = p[3 * HP], w4 = p[4 * HP], w5 = p[5 * HP];
#pragma unroll
		for (int i = 0; i < 13; ++i) {
			const int w6 = p[(i + 6) * HP];
			const int sum = 18 * (w0 + w6) + 34 * (w1 + w5) + 49 * (w2 + w4) + 55 * w3;
			o[i * HP] = (uint8_t)min((sum + (1 << 15)) >> 16, 255);
			w0 = w1; w1 = w2; w2 = w3; w3 = w4; w4 = w5; w5 = w6;
		}
	}

	float ca = 1.f, sa = 0.f;
	if (m01 != 0 || m10 != 0) {
		const double dx = (double)m10, dy = (double)m01;
		const double r = sqrt(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
		ca = (float)(dx / r);
		sa = (float)(dy / r);
	}
	if (lane == 0) kp_angle[(size_t)img * max_feats + k] = atan2_deg((float)m01, (float)m10);
	__syncthreads();

	// lane L evaluates tests 4L..4L+3  (bits (4L..4L+3)&7 of byte L/2); blurred(px,py) = s_b[(py-3)*HP + px-3]
	const uint8_t* bc = s_b + (PR - 3) * HP + (PR - 3);
	unsigned bits = 0;
#pragma unroll
	for (int t = 0; t < 4; ++t) {
		const char4 pt = reinterpret_cast<const char4*>(pattern)[4 * lane + t];
		const float x0 = (float)pt.x, y0 = (float)pt.y, x1 = (float)pt.z, y1 = (float)pt.w;
		const int ix0 = (int)rintf(__fsub_rn(__fmul_rn(x0, ca), __fmul_rn(y0, sa)));
		const int iy0 = (int)rintf(__fadd_rn(__fmul_rn(x0, sa), __fmul_rn(y0, ca)));
		const int ix1 = (int)rintf(__fsub_rn(__fmul_rn(x1, ca), __fmul_rn(y1, sa)));
		const int iy1 = (int)rintf(__fadd_rn(__fmul_rn(x1, sa), __fmul_rn(y1, ca)));
		const int v0 = bc[iy0 * HP + ix0];
		const int v1 = bc[iy1 * HP + ix1];
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
	hipLaunchKernelGGL(describe_kernel, dim3(n_feats, nimg), dim3(64), 0, ctx->stream, b->img, b->h, b->pitch,
	                   img_begin, b->max_feats, b->kp_xy, b->nkp, ctx->d_pattern, b->kp_angle, b->desc);
	SSK_LAUNCH_CHECK(ctx, "describe_kernel");
	return SSK_OK;
}
