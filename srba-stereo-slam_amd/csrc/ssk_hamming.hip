// ssk_hamming.hip -- brute-force 256-bit Hamming K=1 matcher (BFMatcher(NORM_HAMMING,false).match,
// reference call site EST.cpp:1447-1452), batched over independent (query,train) pairs.
//
// Mapping (gfx950, wave64): one 256-thread workgroup = 64 query rows x 4 waves.  Lane l of every wave
// keeps query row (tile*64 + l) in 8 VGPRs; the train rows of the pair are staged through LDS in tiles
// of TT rows (coalesced 16-B loads, optional row gather fused into the load address: a5,
// EST.cpp:1354-1358,1435-1439).  Wave w scans train rows w, w+4, w+8, ... of the tile: every lane of a
// wave reads the SAME LDS row (a broadcast, no bank conflict), xors/popcounts 256 bits
// (8 x v_bcnt_u32_b32 with accumulate) and keeps the minimum of the packed key (distance << 16 | row),
// whose integer minimum is "smallest distance, then lowest train index" -- OpenCV's first-minimum rule.
// The 4 waves' keys meet in LDS; wave 0 writes trainIdx / distance.
// Bound: popcount VALU work, M*M'*~18 lane-ops per pair against 64 KB of descriptor bytes -- far from
// HBM; descriptors of a whole batch stay L2-resident.
#include "ssk_internal.h"

namespace {

constexpr int QT = 64;      // query rows per workgroup (one per lane)
constexpr int NW = 4;       // waves per workgroup, each takes every 4th train row
constexpr int TT = 512;     // train rows per LDS tile (16 KiB)

__global__ __launch_bounds__(QT * NW)
void hamming_match_kernel(const uint4* __restrict__ desc_q, const int32_t* __restrict__ q_beg,
                          const int32_t* __restrict__ q_end, const int32_t* __restrict__ q_gather,
                          const uint4* __restrict__ desc_t, const int32_t* __restrict__ t_beg,
                          const int32_t* __restrict__ t_end, const int32_t* __restrict__ t_gather,
                          const int32_t* __restrict__ out_beg,
                          int32_t* __restrict__ out_idx, int32_t* __restrict__ out_dist)
{
	__shared__ uint4 s_train[TT * 2];
	__shared__ uint32_t s_key[NW][QT];

	const int pair = blockIdx.y;
	const int q0 = q_beg[pair], q1 = q_end[pair];
	const int t0 = t_beg[pair], t1 = t_end[pair];
	const int o0 = out_beg ? out_beg[pair] - q0 : 0;      // output row = o0 + virtual query row
	const int qbase = q0 + blockIdx.x * QT;
	if (qbase >= q1) return;                      // whole workgroup: uniform exit
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int qrow = qbase + lane;
	const bool q_ok = qrow < q1;

	uint4 qa = make_uint4(0, 0, 0, 0), qb = qa;
	if (q_ok) {
		const int src = q_gather ? q_gather[qrow] : qrow;
		qa = desc_q[2 * (size_t)src];
		qb = desc_q[2 * (size_t)src + 1];
	}

	uint32_t best = 0xFFFFFFFFu;
	const int nt = t1 - t0;
	for (int tile = 0; tile < nt; tile += TT) {
		const int rows = min(TT, nt - tile);
		__syncthreads();                          // previous tile fully consumed
		for (int c = threadIdx.x; c < rows * 2; c += QT * NW) {
			const int r = t0 + tile + (c >> 1);
			const int src = t_gather ? t_gather[r] : r;
			s_train[c] = desc_t[2 * (size_t)src + (c & 1)];
		}
		__syncthreads();
		for (int r = wave; r < rows; r += NW) {
			const uint4 ta = s_train[2 * r], tb = s_train[2 * r + 1];
			uint32_t d = __popc(qa.x ^ ta.x);
			d += __popc(qa.y ^ ta.y);
			d += __popc(qa.z ^ ta.z);
			d += __popc(qa.w ^ ta.w);
			d += __popc(qb.x ^ tb.x);
			d += __popc(qb.y ^ tb.y);
			d += __popc(qb.z ^ tb.z);
			d += __popc(qb.w ^ tb.w);
			// tiles are visited in increasing row order, so a 16-bit in-tile row + the tile base kept
			// separately would also do; rows < 2^22 keeps one packed 32-bit key simple and exact.
			const uint32_t key = (d << 22) | (uint32_t)(tile + r);
			best = min(best, key);
		}
	}
	s_key[wave][lane] = best;
	__syncthreads();
	if (wave == 0 && q_ok) {
		const uint32_t k = min(min(s_key[0][lane], s_key[1][lane]), min(s_key[2][lane], s_key[3][lane]));
		if (nt > 0) {
			out_idx[o0 + qrow] = (int32_t)(k & 0x3FFFFFu);
			out_dist[o0 + qrow] = (int32_t)(k >> 22);
		} else {
			out_idx[o0 + qrow] = -1;
			out_dist[o0 + qrow] = 0x7FFFFFFF;
		}
	}
}

}  // namespace

int ssk_launch_hamming(ssk_ctx* ctx, const uint8_t* d_q, const int32_t* d_q_beg, const int32_t* d_q_end,
                       const int32_t* d_q_gather, const uint8_t* d_t, const int32_t* d_t_beg, const int32_t* d_t_end,
                       const int32_t* d_t_gather, const int32_t* d_out_beg,
                       int n_pairs, int max_nq, int32_t* d_train_idx, int32_t* d_dist)
{
	if (n_pairs <= 0 || max_nq <= 0) return SSK_OK;
	dim3 grid((max_nq + QT - 1) / QT, n_pairs);
	ssk_stage_scope prof(ctx, SSK_STAGE_HAMMING);
	hipLaunchKernelGGL(hamming_match_kernel, grid, dim3(QT * NW), 0, ctx->stream,
	                   (const uint4*)d_q, d_q_beg, d_q_end, d_q_gather, (const uint4*)d_t, d_t_beg, d_t_end, d_t_gather,
	                   d_out_beg, d_train_idx, d_dist);
	SSK_LAUNCH_CHECK(ctx, "hamming_match_kernel");
	return SSK_OK;
}
