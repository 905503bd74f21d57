// ssk_da_filters.hip -- the two integer/deterministic data-association filters of
// m_internal_data_association, batched over keyframe pairs:
//   direction histogram   m_detect_outliers_with_direction      EST.cpp:1883-1946
//   ORB distance + serial uniqueness (quirk Q1)   m_detect_outliers_with_orb_distance   EST.cpp:1951-2010
// applied in that order (EST.cpp:1482-1515), then the tracked count (EST.cpp:1664-1668).
//
// One wavefront (64 lanes) per keyframe pair.  Everything lives in LDS: per match a status byte and a
// bin byte, the 36-bin histogram (ds atomics), and the uniqueness scan's other_matched[] table.
//   * direction: lanes stride over the matches; the quotient (oy-ty)/(ox-tx) is formed with IEEE float
//     ops exactly as the C++ does, and binned by comparing against the context's 19 thresholds (the
//     preimages of the bin edges under the host libm's atanf, ssk_context.hip) -- so no device atanf.
//   * mode = FIRST maximal bin (strict '>'), survivors = mode-1, mode, mode+1 with wrap.
//   * the uniqueness scan is order dependent (each accepted match stores ++stage1_counter, which later
//     comparisons read back): the wave runs it serially in query order, inputs preloaded 64 at a time and
//     broadcast by v_readlane, state in LDS; M <= 500.
// Bound: latency of the serial LDS chain; not a bandwidth kernel (tens of KB per pair).
#include "ssk_internal.h"
#include <cfloat>

namespace {

constexpr int MAXM = 4096;   // matches per keyframe this kernel supports (reference: n_feats = 500)

__global__ __launch_bounds__(64)
void da_filters_kernel(const int32_t* __restrict__ train_idx, const int32_t* __restrict__ dist,
                       const int32_t* __restrict__ out_beg,
                       const int32_t* __restrict__ q_beg, const int32_t* __restrict__ q_end,
                       const int32_t* __restrict__ t_beg, const int32_t* __restrict__ t_end,
                       const float* __restrict__ this_xy, const float* __restrict__ other_xy,
                       const float* __restrict__ dir_thresh,
                       int filter_dir, int filter_orb, float rows_offset, float max_orb,
                       int32_t* __restrict__ status_out, int32_t* __restrict__ tracked_out)
{
	__shared__ uint8_t s_status[MAXM];
	__shared__ int8_t s_bin[MAXM];
	__shared__ int16_t s_first[MAXM];
	__shared__ float s_second[MAXM];
	__shared__ int s_hist[36];
	__shared__ int s_mode;
	__shared__ float s_th[19];

	const int pair = blockIdx.x, lane = threadIdx.x;
	const int q0 = q_beg[pair], n_this = min(q_end[pair] - q0, MAXM);
	const int t0 = t_beg[pair], n_other = min(t_end[pair] - t0, MAXM);
	const int o0 = out_beg ? out_beg[pair] : q0;          // row of match 0 in train_idx/dist/status

	if (lane < 36) s_hist[lane] = 0;
	if (lane < 19) s_th[lane] = dir_thresh[lane];
	for (int k = lane; k < n_this; k += 64) s_status[k] = SSK_DA_TRACKED;
	for (int j = lane; j < n_other; j += 64) { s_first[j] = -1; s_second[j] = FLT_MAX; }
	__syncthreads();

	if (n_other <= 0) {       // BFMatcher returned nothing: nothing can be tracked
		for (int k = lane; k < n_this; k += 64) status_out[o0 + k] = SSK_DA_NON_TRACKED;
		if (lane == 0) tracked_out[pair] = 0;
		return;
	}

	if (filter_dir) {
		for (int k = lane; k < n_this; k += 64) {
			const int j = train_idx[o0 + k];
			const float tx = this_xy[2 * (size_t)(q0 + k)], ty = this_xy[2 * (size_t)(q0 + k) + 1];
			const float ox = other_xy[2 * (size_t)(t0 + j)];
			const float oy = __fadd_rn(other_xy[2 * (size_t)(t0 + j) + 1], rows_offset);   // EST.cpp:1913
			const float q = __fdiv_rn(__fsub_rn(oy, ty), __fsub_rn(ox, tx));              // EST.cpp:1915
			int bin;
			if (q != q) bin = -2;                                   // NaN: defined as rejected, not histogrammed
			else {
				bin = 8;
#pragma unroll
				for (int b = 0; b < 19; ++b) bin += (q >= s_th[b]) ? 1 : 0;
				atomicAdd(&s_hist[bin], 1);
			}
			s_bin[k] = (int8_t)bin;
		}
		__syncthreads();
		if (lane == 0) {
			int mode = 0, best = 0;
			for (int b = 0; b < 36; ++b) if (s_hist[b] > best) { best = s_hist[b]; mode = b; }   // :1924-1931
			s_mode = mode;
		}
		__syncthreads();
		const int mode = s_mode;
		const int b0 = mode == 0 ? 35 : mode - 1, b2 = mode == 35 ? 0 : mode + 1;                // :1934-1936
		for (int k = lane; k < n_this; k += 64) {
			const int b = s_bin[k];
			if (b == -2 || !(b == b0 || b == mode || b == b2)) s_status[k] = SSK_DA_REJ_SLOPE;    // :1940-1945
		}
		__syncthreads();
	}

	if (filter_orb) {
		// serial, literal order of EST.cpp:1966-2008 (queryIdx == k for BFMatcher output).  The whole wave walks the
		// matches in lock step: 64 matches at a time are loaded one per lane (status, distance, train index -- none of
		// them is changed by the scan before its turn: the scan only demotes EARLIER matches) and broadcast with
		// v_readlane, so the dependent chain per match is the two LDS reads of other_matched[t]; lane 0 writes.
		float counter = 0.f;   // size_t stage1_counter; exact in float for counts < 2^24
		for (int base = 0; base < n_this; base += 64) {
			const int kl = base + lane;
			int st_l = -1, d_l = 0, t_l = 0;
			if (kl < n_this) { st_l = s_status[kl]; d_l = dist[o0 + kl]; t_l = train_idx[o0 + kl]; }
			const int nk = min(64, n_this - base);
			for (int l = 0; l < nk; ++l) {
				if (__builtin_amdgcn_readlane(st_l, l) != SSK_DA_TRACKED) continue;               // :1969
				const int k = base + l;
				const float d = (float)__builtin_amdgcn_readlane(d_l, l);
				const int t = __builtin_amdgcn_readlane(t_l, l);
				if ((double)d > (double)max_orb) { if (lane == 0) s_status[k] = SSK_DA_REJ_ORB; continue; }       // :1980
				const float second = s_second[t];
				const int prev = s_first[t];
				if (d > second) { if (lane == 0) s_status[k] = SSK_DA_NON_TRACKED; continue; }                    // :1988
				if (prev != -1) { if (lane == 0) s_status[prev] = SSK_DA_NON_TRACKED; counter -= 1.f; }           // :1995-1999
				counter += 1.f;
				if (lane == 0) { s_first[t] = (int16_t)k; s_second[t] = counter; }                                // :2002-2007 (Q1)
			}
		}
	}
	__syncthreads();

	int cnt = 0;
	for (int k = lane; k < n_this; k += 64) {
		const int s = s_status[k];
		status_out[o0 + k] = s;
		cnt += (s == SSK_DA_TRACKED);
	}
	for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
	if (lane == 0) tracked_out[pair] = cnt;
}

}  // namespace

int ssk_launch_da_filters(ssk_ctx* ctx, const int32_t* d_train_idx, const int32_t* d_dist, const int32_t* d_out_beg,
                          const int32_t* d_q_beg, const int32_t* d_q_end, const int32_t* d_t_beg, const int32_t* d_t_end,
                          const float* d_this_xy, const float* d_other_xy,
                          int n_pairs, const ssk_da_params& prm, int32_t* d_status, int32_t* d_tracked)
{
	if (n_pairs <= 0) return SSK_OK;
	ssk_stage_scope prof(ctx, SSK_STAGE_DA_FILTERS);
	hipLaunchKernelGGL(da_filters_kernel, dim3(n_pairs), dim3(64), 0, ctx->stream,
	                   d_train_idx, d_dist, d_out_beg, d_q_beg, d_q_end, d_t_beg, d_t_end, d_this_xy, d_other_xy, ctx->d_dir_thresh,
	                   prm.filter_by_direction, prm.filter_by_orb_distance, (float)(size_t)prm.rows_offset,
	                   prm.max_orb_distance_da, d_status, d_tracked);
	SSK_LAUNCH_CHECK(ctx, "da_filters_kernel");
	return SSK_OK;
}

int ssk_da_max_matches() { return MAXM; }
