// ssk_stereo.hip -- left<->right stereo matching of one keyframe's ORB features: "descriptor row-by-row"
// (stereo-vo's smDescRbR, forced at EST.cpp:1136; parameters KITTI.ini:17-24).  Contract C10 of DESIGN.md.
//
// For every left keypoint: among right keypoints with |yL-yR| <= max_y_diff and
// min_disparity <= xL-xR <= max_disparity, the one with the smallest 256-bit Hamming distance (lowest right
// index on ties), accepted iff distance <= orb_max_distance.  Matches come out in left-keypoint order.
//
// Mapping (gfx950): one 1024-thread workgroup per keyframe.  Left keypoints are processed in tiles of 64
// (one per lane, descriptor in 8 VGPRs); right descriptors + coordinates are staged through LDS in tiles of
// 512 rows (once when they fit one tile); the 16 waves split the right rows (every lane of a wave reads the same LDS row: broadcast).
// The band test runs first and a wave skips the popcounts of a row no lane accepts (ballot).
// Argmin = integer min of the packed key (distance<<16 | right index).  The per-left results are then
// compacted IN ORDER with a workgroup prefix sum, which also writes the gather lists the batched data
// association reads (descriptor row + pixel of the matched left keypoint, Q3 / a5).
#include "ssk_frontend.h"

namespace {

constexpr int TT = 512;
#ifndef SSK_ST_NT
#define SSK_ST_NT 1024
#endif
constexpr int NT = SSK_ST_NT, NW = NT / 64;   // 16 waves: the batch gives one workgroup per CU, so latency is hidden inside it

__global__ __launch_bounds__(NT)
void stereo_match_kernel(int kf_begin, int max_feats, const short2* __restrict__ kp_xy, const int32_t* __restrict__ nkp,
                         const uint8_t* __restrict__ desc, float max_y_diff, float min_disp, float max_disp,
                         int orb_max_distance,
                         int32_t* __restrict__ m_q, int32_t* __restrict__ m_t, int32_t* __restrict__ m_d,
                         int32_t* __restrict__ nm, int32_t* __restrict__ m_row, float2* __restrict__ m_xy, float4* __restrict__ m_obs)
{
	__shared__ uint4 s_desc[TT * 2];
	__shared__ short2 s_rxy[TT];
	__shared__ uint32_t s_part[NW][64];
	__shared__ uint32_t s_best[SSK_MAX_FEATS];
	__shared__ int s_scan[NT];
	constexpr int NB = 1024;                                  // bins of four image rows (y < 4096: the packed candidate key's 12 bits)
	static_assert(NB == NT, "a bin per thread in the counting sorts");
	__shared__ int s_bin[NB];
	__shared__ uint16_t s_rstart[NB + 2], s_ridx[TT], s_lperm[SSK_MAX_FEATS];

	const int kf = kf_begin + blockIdx.x;
	const int il = kf * 2, ir = kf * 2 + 1;
	const int nl = min(nkp[il], max_feats), nr = min(nkp[ir], max_feats);
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const uint4* __restrict__ dl = reinterpret_cast<const uint4*>(desc) + (size_t)il * max_feats * 2;
	const uint4* __restrict__ dr = reinterpret_cast<const uint4*>(desc) + (size_t)ir * max_feats * 2;
	const short2* __restrict__ xl = kp_xy + (size_t)il * max_feats;
	const short2* __restrict__ xr = kp_xy + (size_t)ir * max_feats;

	if (nr <= TT) {
		// Round 6: both sides in ROW order first.  The matcher's band is a few image rows wide, the keypoints come in response order: as it was, every
		// left keypoint met every right one (the band test alone: 250 k pairs a keyframe, and two thirds of the right rows had SOME lane of the 64 in band, so
		// their popcounts ran as well -- the kernel was vector-issue bound on that).  A counting sort by y / 4 (LDS histogram, one scan, a scatter: the order
		// inside a bin does not matter) lets a wavefront of 64 row-neighbouring left keypoints visit only the right rows of its own bins +- the band: ~70 of
		// 500.  Results do not depend on either order: the best key (distance << 16 | ORIGINAL right index) is a minimum, and it is stored by the original
		// left index.  Each wavefront takes whole left tiles on its own: no partial minima across wavefronts, no barrier in the loop.
		const int band = (int)max_y_diff + 2;                   // |yL - yR| <= max_y_diff can only hold inside this many rows
		for (int side = 0; side < 2; ++side) {                    // 0: right (staged in bin order), 1: left (a permutation)
			const int n = side ? nl : nr;
			const short2* __restrict__ xy = side ? xl : xr;
			s_bin[tid] = 0;                                          // (NT == NB)
			__syncthreads();
			short2 me = make_short2(0, 0);
			if (tid < n) { me = xy[tid]; atomicAdd(&s_bin[min(max((int)me.y, 0) >> 2, NB - 1)], 1); }
			__syncthreads();
			// exclusive scan of the 1 024 bins: DPP scan inside the wavefront, the 16 wavefront totals through LDS
			const int v = s_bin[tid];
			int x = v;
			x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);
			x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);
			x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);
			x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);
			x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);
			x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);
			if (lane == 63) s_scan[wave] = x;
			__syncthreads();
			int off = 0;
			for (int w = 0; w < wave; ++w) off += s_scan[w];
			const int excl = off + x - v;
			__syncthreads();
			s_bin[tid] = excl;
			if (side == 0) { s_rstart[tid] = (uint16_t)excl; if (tid == NT - 1) s_rstart[NB] = (uint16_t)(excl + v); }
			__syncthreads();
			if (tid < n) {
				const int pos = atomicAdd(&s_bin[min(max((int)me.y, 0) >> 2, NB - 1)], 1);
				if (side == 0) {
					s_rxy[pos] = me; s_ridx[pos] = (uint16_t)tid;
					s_desc[2 * pos] = dr[2 * tid]; s_desc[2 * pos + 1] = dr[2 * tid + 1];
				} else s_lperm[pos] = (uint16_t)tid;
			}
			__syncthreads();
		}
		for (int lbase = 64 * wave; lbase < nl; lbase += 64 * NW) {
			const bool ok = lbase + lane < nl;
			const int li = s_lperm[ok ? lbase + lane : lbase];       // (the tile's first keypoint again: inside its rows, never accepted)
			const uint4 qa = dl[2 * li], qb = dl[2 * li + 1];
			const short2 p = xl[li];
			const float lx = (float)p.x, ly = (float)p.y;
			// the tile's rows: bin order, so its first and its last valid lane bound them
			const int y_first = __builtin_amdgcn_readfirstlane((int)p.y);
			const int y_last = __builtin_amdgcn_readlane((int)p.y, min(nl - lbase, 64) - 1);
			const int b_lo = min(max((((y_first >> 2) << 2) - band) >> 2, 0), NB - 1), b_hi = min(max((((y_last >> 2) << 2) + 3 + band) >> 2, 0), NB - 1);
			const int r0 = s_rstart[b_lo], r1 = s_rstart[b_hi + 1];
			uint32_t best = 0xFFFFFFFFu;
			for (int r = r0; r < r1; ++r) {
				const short2 rp = s_rxy[r];
				const float disp = __fsub_rn(lx, (float)rp.x);
				const bool cand = ok && !(fabsf(__fsub_rn(ly, (float)rp.y)) > max_y_diff) &&
				                  !(disp < min_disp) && !(disp > max_disp);
				if (__ballot(cand) == 0ull) continue;
				const uint4 ta = s_desc[2 * r], tb = s_desc[2 * r + 1];
				uint32_t d = __popc(qa.x ^ ta.x);
				d += __popc(qa.y ^ ta.y); d += __popc(qa.z ^ ta.z); d += __popc(qa.w ^ ta.w);
				d += __popc(qb.x ^ tb.x); d += __popc(qb.y ^ tb.y); d += __popc(qb.z ^ tb.z); d += __popc(qb.w ^ tb.w);
				if (cand) best = min(best, (d << 16) | (uint32_t)s_ridx[r]);
			}
			if (ok) s_best[li] = best;
		}
		__syncthreads();
	} else
	for (int lbase = 0; lbase < nl; lbase += 64) {
		const int li = lbase + lane;
		const bool ok = li < nl;
		uint4 qa = make_uint4(0, 0, 0, 0), qb = qa;
		float lx = 0.f, ly = 0.f;
		if (ok) {
			qa = dl[2 * li]; qb = dl[2 * li + 1];
			const short2 p = xl[li];
			lx = (float)p.x; ly = (float)p.y;
		}
		uint32_t best = 0xFFFFFFFFu;
		for (int tile = 0; tile < nr; tile += TT) {
			const int rows = min(TT, nr - tile);
			if (nr > TT || lbase == 0) {                     // a right side of <= TT rows is staged once for all left tiles
				__syncthreads();
				for (int c = tid; c < rows * 2; c += NT) s_desc[c] = dr[2 * (size_t)tile + c];
				for (int c = tid; c < rows; c += NT) s_rxy[c] = xr[tile + c];
				__syncthreads();
			}
			for (int r = wave; r < rows; r += NW) {
				const short2 rp = s_rxy[r];
				const float disp = __fsub_rn(lx, (float)rp.x);
				const bool cand = ok && !(fabsf(__fsub_rn(ly, (float)rp.y)) > max_y_diff) &&
				                  !(disp < min_disp) && !(disp > max_disp);
				if (__ballot(cand) == 0ull) continue;
				const uint4 ta = s_desc[2 * r], tb = s_desc[2 * r + 1];
				uint32_t d = __popc(qa.x ^ ta.x);
				d += __popc(qa.y ^ ta.y); d += __popc(qa.z ^ ta.z); d += __popc(qa.w ^ ta.w);
				d += __popc(qb.x ^ tb.x); d += __popc(qb.y ^ tb.y); d += __popc(qb.z ^ tb.z); d += __popc(qb.w ^ tb.w);
				if (cand) best = min(best, (d << 16) | (uint32_t)(tile + r));
			}
		}
		s_part[wave][lane] = best;
		__syncthreads();
		if (wave == 0 && ok) {
			uint32_t m = s_part[0][lane];
#pragma unroll
			for (int w = 1; w < NW; ++w) m = min(m, s_part[w][lane]);
			s_best[li] = m;
		}
		__syncthreads();
	}

	// ordered compaction: thread t owns left keypoints [t*PER, t*PER+PER)
	constexpr int PER = SSK_MAX_FEATS / NT;
	int cnt = 0;
	uint32_t mine[PER];
#pragma unroll
	for (int e = 0; e < PER; ++e) {
		const int li = tid * PER + e;
		uint32_t k = 0xFFFFFFFFu;
		if (li < nl) k = s_best[li];
		const bool acc = k != 0xFFFFFFFFu && (int)(k >> 16) <= orb_max_distance;
		mine[e] = acc ? k : 0xFFFFFFFFu;
		cnt += acc;
	}
	s_scan[tid] = cnt;
	__syncthreads();
	for (int o = 1; o < NT; o <<= 1) {
		const int v = tid >= o ? s_scan[tid - o] : 0;
		__syncthreads();
		s_scan[tid] += v;
		__syncthreads();
	}
	int pos = s_scan[tid] - cnt;
	const size_t ob = (size_t)kf * max_feats;
#pragma unroll
	for (int e = 0; e < PER; ++e) {
		if (mine[e] != 0xFFFFFFFFu) {
			const int li = tid * PER + e;
			m_q[ob + pos] = li;
			m_t[ob + pos] = (int32_t)(mine[e] & 0xFFFFu);
			m_d[ob + pos] = (int32_t)(mine[e] >> 16);
			m_row[ob + pos] = il * max_feats + li;
			const short2 p = xl[li];
			m_xy[ob + pos] = make_float2((float)p.x, (float)p.y);
			const short2 pr = xr[mine[e] & 0xFFFFu];
			m_obs[ob + pos] = make_float4((float)p.x, (float)p.y, (float)pr.x, (float)pr.y);
			++pos;
		}
	}
	if (tid == NT - 1) nm[kf] = s_scan[NT - 1];
}

}  // namespace

int ssk_launch_stereo(ssk_ctx* ctx, ssk_kfbatch* b, int kf_begin, int kf_end, const ssk_frontend_params& p)
{
	const int nkf = kf_end - kf_begin;
	if (nkf <= 0) return SSK_OK;
	ssk_stage_scope prof(ctx, SSK_STAGE_STEREO);
	hipLaunchKernelGGL(stereo_match_kernel, dim3(nkf), dim3(NT), 0, ctx->stream, kf_begin, b->max_feats, b->kp_xy,
	                   b->nkp, b->desc, p.max_y_diff, p.min_disparity, p.max_disparity, p.orb_max_distance,
	                   b->m_q, b->m_t, b->m_d, b->nm, b->m_row, b->m_xy, b->m_obs);
	SSK_LAUNCH_CHECK(ctx, "stereo_match_kernel");
	return SSK_OK;
}
