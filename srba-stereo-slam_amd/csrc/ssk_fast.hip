// ssk_fast.hip -- FAST-9/16 corner score + 3x3 non-max suppression + candidate compaction, fused, one
// pass over the image (the HBM-bound scan of the front-end: every pixel of both images is read once).
// Replaces the detection half of cv::ORB as driven by stereo-vo's processNewImagePair (reference call
// sites EST.cpp:112,267); contract C2-C4 of DESIGN.md "Front-end contract".
//
// Mapping (gfx950): one 256-thread workgroup per 64x16 output tile of one image; grid.z = image.
//   phase 0  the (64+8) x (16+8) pixel tile (3-px ring + 1-px NMS halo) is staged in LDS with aligned
//            dword loads (coalesced 4-B per lane along rows; neighbouring tiles re-read the halo from L2).
//   phase A  cheap necessary test on the 4 compass ring pixels for every pixel of the 66x18 score region;
//            survivors are queued in LDS (ds atomics), so phase B runs on dense lanes.
//   phase B  exact score s-1 (max over the 16 arcs of 9 of the min signed difference, by min/max doubling)
//            for queued pixels -> 8-bit score tile in LDS.
//   phase C  strict 3x3 NMS on the score tile, survivors packed score<<24|(4095-y)<<12|(4095-x) into an LDS
//            list; ONE global atomicAdd per workgroup reserves the output range in the image's list.
// The candidate order in HBM is not deterministic (atomic reservation order); every consumer orders by
// the packed key, which is unique per pixel, so results are.
// Algorithmic bytes: W*H per image read once (+ 4 B per surviving corner written); roofline: HBM.
#include "ssk_frontend.h"

namespace {

constexpr int TW = 64, TH = 16;            // output tile
constexpr int IW = 80;                     // LDS image tile pitch in bytes (20 dwords)
constexpr int IH = TH + 8;
constexpr int SW = TW + 2, SH = TH + 2;    // score region (tile + 1 halo)
constexpr int SP = 68;                     // score tile pitch

__device__ __forceinline__ int fast_s(const int (&d)[16])
{
	int lo2[16], hi2[16], lo4[16], hi4[16], lo8[16], hi8[16];
#pragma unroll
	for (int i = 0; i < 16; ++i) { lo2[i] = min(d[i], d[(i + 1) & 15]); hi2[i] = max(d[i], d[(i + 1) & 15]); }
#pragma unroll
	for (int i = 0; i < 16; ++i) { lo4[i] = min(lo2[i], lo2[(i + 2) & 15]); hi4[i] = max(hi2[i], hi2[(i + 2) & 15]); }
#pragma unroll
	for (int i = 0; i < 16; ++i) { lo8[i] = min(lo4[i], lo4[(i + 4) & 15]); hi8[i] = max(hi4[i], hi4[(i + 4) & 15]); }
	int sb = 0, hmin = 255;
#pragma unroll
	for (int i = 0; i < 16; ++i) {
		sb = max(sb, min(lo8[i], d[(i + 8) & 15]));       // all 9 of arc i brighter by at least this
		hmin = min(hmin, max(hi8[i], d[(i + 8) & 15]));   // all 9 of arc i darker by at least -this
	}
	return max(sb, -hmin);
}

__global__ __launch_bounds__(256)
void fast_nms_kernel(const uint8_t* __restrict__ images, int w, int h, int pitch, int img_begin,
                     int threshold, uint32_t* __restrict__ cand, int cand_cap, int32_t* __restrict__ ncand)
{
	__shared__ __attribute__((aligned(16))) uint8_t s_img[IH * IW];
	__shared__ uint8_t s_score[SH * SP];
	__shared__ uint16_t s_q[SW * SH];
	__shared__ uint32_t s_out[TW * TH / 4 + 64];
	__shared__ int s_qn, s_on, s_base;

	const int img = img_begin + blockIdx.z;
	const uint8_t* __restrict__ src = images + (size_t)img * h * pitch;
	const int x0 = SSK_EDGE + blockIdx.x * TW, y0 = SSK_EDGE + blockIdx.y * TH;
	const int xs = x0 - 4, xa = xs & ~3, shift = xs - xa;
	const int tid = threadIdx.x;

	if (tid == 0) { s_qn = 0; s_on = 0; }
	for (int i = tid; i < IH * (IW / 4); i += 256) {
		const int row = i / (IW / 4), c = i - row * (IW / 4);
		const int gy = min(max(y0 - 4 + row, 0), h - 1);
		const int gx = xa + 4 * c;
		uint32_t v = 0;
		if (gx + 4 <= pitch) v = *reinterpret_cast<const uint32_t*>(src + (size_t)gy * pitch + gx);
		reinterpret_cast<uint32_t*>(s_img)[row * (IW / 4) + c] = v;
	}
	for (int i = tid; i < SH * SP; i += 256) s_score[i] = 0;
	__syncthreads();

	// phase A: necessary condition on the compass points 0,4,8,12 of the ring
	for (int i = tid; i < SW * SH; i += 256) {
		const int sy = i / SW, sx = i - sy * SW;          // score-region coords
		const int gx = x0 - 1 + sx, gy = y0 - 1 + sy;
		if (gx < 3 || gx >= w - 3 || gy < 3 || gy >= h - 3) continue;
		const uint8_t* p = s_img + (sy + 3) * IW + shift + (sx + 3);
		const int c = p[0], hi = c + threshold, lo = c - threshold;
		const int a0 = p[3 * IW], a4 = p[3], a8 = p[-3 * IW], a12 = p[-3];
		const int nb = (a0 > hi) + (a4 > hi) + (a8 > hi) + (a12 > hi);
		const int nd = (a0 < lo) + (a4 < lo) + (a8 < lo) + (a12 < lo);
		if (nb >= 2 || nd >= 2) s_q[atomicAdd(&s_qn, 1)] = (uint16_t)i;
	}
	__syncthreads();

	// phase B: exact score on the queued pixels
	const int qn = s_qn;
	for (int q = tid; q < qn; q += 256) {
		const int i = s_q[q];
		const int sy = i / SW, sx = i - sy * SW;
		const uint8_t* p = s_img + (sy + 3) * IW + shift + (sx + 3);
		const int c = p[0];
		int d[16];
		d[0] = p[3 * IW] - c;       d[1] = p[3 * IW + 1] - c;   d[2] = p[2 * IW + 2] - c;   d[3] = p[IW + 3] - c;
		d[4] = p[3] - c;            d[5] = p[-IW + 3] - c;      d[6] = p[-2 * IW + 2] - c;  d[7] = p[-3 * IW + 1] - c;
		d[8] = p[-3 * IW] - c;      d[9] = p[-3 * IW - 1] - c;  d[10] = p[-2 * IW - 2] - c; d[11] = p[-IW - 3] - c;
		d[12] = p[-3] - c;          d[13] = p[IW - 3] - c;      d[14] = p[2 * IW - 2] - c;  d[15] = p[3 * IW - 1] - c;
		const int s = fast_s(d);
		if (s > threshold) s_score[sy * SP + sx] = (uint8_t)(s - 1);
	}
	__syncthreads();

	// phase C: strict 3x3 NMS inside the emit region [31,W-31) x [31,H-31)
	for (int i = tid; i < TW * TH; i += 256) {
		const int ly = i / TW, lx = i - ly * TW;
		const int gx = x0 + lx, gy = y0 + ly;
		if (gx >= w - SSK_EDGE || gy >= h - SSK_EDGE) continue;
		const uint8_t* s = s_score + (ly + 1) * SP + (lx + 1);
		const int v = s[0];
		if (v == 0) continue;
		if (v > s[-1] && v > s[1] && v > s[-SP - 1] && v > s[-SP] && v > s[-SP + 1] && v > s[SP - 1] && v > s[SP] && v > s[SP + 1])
			s_out[atomicAdd(&s_on, 1)] = ((uint32_t)v << 24) | ((uint32_t)(4095 - gy) << 12) | (uint32_t)(4095 - gx);
	}
	__syncthreads();
	const int on = s_on;
	if (on == 0) return;
	if (tid == 0) s_base = atomicAdd(&ncand[img], on);
	__syncthreads();
	const int base = s_base;
	uint32_t* dst = cand + (size_t)img * cand_cap;
	for (int i = tid; i < on; i += 256)
		if (base + i < cand_cap) dst[base + i] = s_out[i];
}

}  // namespace

int ssk_launch_fast(ssk_ctx* ctx, ssk_kfbatch* b, int img_begin, int img_end, int threshold)
{
	const int nimg = img_end - img_begin;
	if (nimg <= 0) return SSK_OK;
	SSK_HIP(ctx, hipMemsetAsync(b->ncand + img_begin, 0, sizeof(int32_t) * (size_t)nimg, ctx->stream));
	const int ew = b->w - 2 * SSK_EDGE, eh = b->h - 2 * SSK_EDGE;
	dim3 grid((ew + TW - 1) / TW, (eh + TH - 1) / TH, nimg);
	ssk_stage_scope prof(ctx, SSK_STAGE_FAST);
	hipLaunchKernelGGL(fast_nms_kernel, grid, dim3(256), 0, ctx->stream, b->img, b->w, b->h, b->pitch, img_begin,
	                   threshold, b->cand, b->cand_cap, b->ncand);
	SSK_LAUNCH_CHECK(ctx, "fast_nms_kernel");
	return SSK_OK;
}
