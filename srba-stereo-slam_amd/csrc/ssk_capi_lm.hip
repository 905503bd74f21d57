// ssk_capi_lm.hip -- C-ABI of seam B3 (relative bundle adjustment): builds the symbolic structure of a
// batch of optimisation windows on the host (graph bookkeeping, as SRBA does symbolically on the CPU),
// uploads state + structure to HBM once, and runs the persistent LM kernel of ssk_lm.hip on it.
#include "ssk_lm.h"
#include "ssk_lm_node_build.h"
#include "ssk_lm_split.h"
#include "ssk_frontend.h"
#include <memory_resource>
#include <unordered_map>
#include <algorithm>
#include <atomic>
#include <thread>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <dlfcn.h>
#include <condition_variable>
#include <functional>
#include <mutex>

struct ssk_lm_batch {
	ssk_ctx* ctx = nullptr;
	int n_windows = 0, trace = 0, max_iters_alloc = 0, max_n6 = 0;
	std::vector<lm_win> hdr;
	void* slab = nullptr;             // ONE device allocation holds every array of the batch
	size_t slab_bytes = 0;
	struct req { size_t bytes, copy; const void* src; void** dst; };
	std::vector<req> reqs;            // filled by upload()/scratch(), resolved by finalize()
	lm_dev D{};
	size_t n_pose = 0, n_lm = 0;      // doubles in pose / lm families
	bool tiled_ok = false;            // every window fits the LDS-tiled kernel
	bool acc_ok = false;              // ... and its accumulators fit LDS too: lm_solve_acc_kernel (on-chip accumulation)
	int schur_mfma = 1;               // acc kernel: Schur complement on the FP64 matrix cores (SSK_LM_SCHUR=mfma, default) or by sparse 6x6 items (=sparse)
	int max_tiles = 0;                // most tiles of a window (cooperative kernel: workgroups per window)
	bool has_general = false;         // the general kernel's structures were built too
	size_t tiled_dyn_bytes = 0;       // its dynamic LDS size
	bool lds_sys = false;             // every window's reduced system fits LDS: the kernel variant without the L2-resident solver
	// cooperative launch in flight: the state it started from (pose_bak / lm_bak) and its parameters, so that a launch whose
	// workgroups were not all resident (stop_reason 5) is repeated with one workgroup per window by ssk_lm_batch_results
	// node form (ssk_lm_node.hip): built when every window of the batch fits it (reduced system + a tile of node records in LDS)
	bool gtab_multi = false;          // the per-workgroup suffix tables of the windows whose table lives in L2 scratch were sized for the cooperative kernel (64 copies)
	bool node_ok = false, node_big = false;
	bool node_small = false;          // every node-kernel window has E <= 10: the kernel variant whose only dense solver is the one-wavefront matrix-core LDL^T (ldl_solve_mfma_wave; ND_BLOCKS6=1: ldl_solve_blocks6_wave)
	nd_dev N{};
	size_t node_dyn_bytes = 0;
	// phase-split node form (ssk_lm_split.hip): windows whose reduced system does not fit LDS (the depth-5 local areas) -- and lean batches:
	// built from the node tables alone, the tiled / general families are never constructed
	bool split_ok = false, lean = false;
	sp_dev S{};
	nd_dev N2{};                      // the phase-split windows' node tables (N: the node kernel's); both families' win[] are indexed by the batch's window number
	int node_n = 0, split_n = 0;      // windows per family (a mixed batch runs both)
	std::vector<char> family;         // lean batch, per window: 0 node kernel, 1 phase-split kernels, 2 pose-only (the only family that keeps the raw observation arrays)
	// pose-only windows (ssk_lm_pose.hip: one or two optimised edges, no optimised landmark -- stage 1 of define_new_keyframe): no node tables
	int pose_n = 0, pose_two = 0;
	const int* pose_order = nullptr;
	size_t pose_dyn_bytes = 0;
	double symbolic_ms = 0.0;         // host time of the symbolic build (ssk_lm_batch_symbolic_ms)
	size_t up_bytes = 0, zero_bytes = 0;   // finalize: bytes uploaded / zeroed (SSK_TIMING)
	int symbolic_threads = 1;
	size_t split_solve_dyn_bytes = 0, split_init_dyn_bytes = 0;
	bool coop_pending = false;
	// a large batch's create returns when its uploads are done; the zero fill of its scratch (queued behind them: it may have to wait for the CUs of
	// whatever else runs on the device) is waited for by the first stream that touches the batch
	hipEvent_t ev_ready = nullptr;
	bool ready_pending = false;
	// (round 6) the lean create does not wait for its uploads either: its big tables sit in the context's pinned arena, which the NEXT lean create of the
	// context waits for before it writes there (lm_wait_uploads) -- by then a symbolic build has passed and the DMA is long done; the few host images that
	// live elsewhere are copied here first and die with the batch (whose destroy waits for the stream)
	bool defer_upload_wait = false;
	std::vector<std::unique_ptr<uint8_t[]>> keep;
	bool used_own = false;            // some operation after the create went through the creating context (its stream): ssk_lm_batch_destroy drains that stream
	size_t guard_off = 0, guard_bytes = 0;   // SSK_LM_GUARD (diagnostic)
	// a batch may be reset / run on ANOTHER context's stream than the one that created it (bench.py's symbolic pass: one thread creates, another
	// runs): the end of the last such use, so that ssk_lm_batch_destroy does not recycle a slab that stream still writes (an event, not the context:
	// the other context may be gone by then)
	hipEvent_t ev_last = nullptr;
	// COMPOSITE batch (round 5: per-window fallback): some windows of the batch do not fit the node builder (a landmark with more than 64 observations or
	// 63 nodes, more than 128 optimised edges, a path of more than 8 steps) -- they alone go through the full builder (the tiled / general kernels of rounds
	// 1-2) as a sub-batch of their own, every other window keeps the node / phase-split / pose-only kernels as the other sub-batch.  The composite owns no
	// device memory; every entry point forwards (sub_of[w]: the window's sub-batch, idx_in[w]: its index there).
	std::vector<ssk_lm_batch*> sub;
	std::vector<int> sub_of, idx_in;
	lm_cam coop_P{};
	double* pose_bak = nullptr; double* lm_bak = nullptr;
};

namespace {

// upload()/scratch() only RECORD what the batch needs; finalize() makes one slab (reusing the context's cached one when it
// is large enough), lays the arrays out at 256-byte offsets, moves every host image with a single staged DMA (small
// batches: the per-keyframe host path) or one copy per array (large batches) and zeroes the scratch part with one memset.
template <typename T, typename A>
int upload(ssk_lm_batch* b, const std::vector<T, A>& v, const T** out)
{
	b->reqs.push_back({sizeof(T) * (v.size() + 4), sizeof(T) * v.size(), v.empty() ? nullptr : (const void*)v.data(), (void**)out});
	return SSK_OK;
}
template <typename T>
int scratch(ssk_lm_batch* b, size_t count, T** out)
{
	b->reqs.push_back({sizeof(T) * (count + 4), 0, nullptr, (void**)out});
	return SSK_OK;
}

// the uploads of the context's last lean create (their sources: the pinned arena) are done -- called before the arena is written or freed
int lm_wait_uploads(ssk_ctx* ctx)
{
	if (ctx->lm_up_pending) {
		ctx->lm_up_pending = false;
		SSK_HIP(ctx, hipEventSynchronize(ctx->ev_lm_up));
	}
	return SSK_OK;
}

int finalize(ssk_lm_batch* b)
{
	ssk_ctx* ctx = b->ctx;
	// uploads first, scratch after: the scratch part is one contiguous range to zero
	size_t up = 0, total = 0;
	for (const ssk_lm_batch::req& r : b->reqs) { if (r.copy > 0) up += ssk_pad(r.bytes); total += ssk_pad(r.bytes); }
	if (total == 0) total = 256;
	b->up_bytes = up; b->zero_bytes = total - up;
	// SSK_LM_GUARD=1 (diagnostic): 1 MB behind the batch's arrays filled with a pattern, checked when the batch is destroyed (a kernel writing past its arrays)
	static const bool guard_on = std::getenv("SSK_LM_GUARD") != nullptr;
	const size_t guard = guard_on ? ((size_t)1 << 20) : 0;
	b->guard_off = total; b->guard_bytes = guard;
	total += guard;
	{
		// the smallest cached slab that fits; none: a new one (a cached one that is too small makes room for it)
		int pick = -1;
		for (int k = 0; k < ssk_ctx::LM_SLABS; ++k)
			if (ctx->lm_slab[k] && ctx->lm_slab_bytes[k] >= total && (pick < 0 || ctx->lm_slab_bytes[k] < ctx->lm_slab_bytes[pick])) pick = k;
		if (pick >= 0) { b->slab = ctx->lm_slab[pick]; b->slab_bytes = ctx->lm_slab_bytes[pick]; ctx->lm_slab[pick] = nullptr; ctx->lm_slab_bytes[pick] = 0; }
		else {
			int small = -1;
			for (int k = 0; k < ssk_ctx::LM_SLABS; ++k)
				if (ctx->lm_slab[k] && (small < 0 || ctx->lm_slab_bytes[k] < ctx->lm_slab_bytes[small])) small = k;
			bool full = true;
			for (int k = 0; k < ssk_ctx::LM_SLABS; ++k) full = full && ctx->lm_slab[k];
			if (full && small >= 0) { SSK_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->lm_slab[small]); ctx->lm_slab[small] = nullptr; ctx->lm_slab_bytes[small] = 0; }
			// head room: the next keyframe's window is a little larger (a quarter; a large batch -- local areas of many streams growing in step through
			// their submap cycle -- a half: every regrowth is a hipMalloc of tens of milliseconds)
			const size_t room = total >= ((size_t)64 << 20) ? total / 2 : total / 4;
			SSK_HIP(ctx, hipMalloc(&b->slab, total + room));
			b->slab_bytes = total + room;
		}
	}
	// A small upload (one keyframe's window) is packed into the pinned staging block and goes with ONE DMA (a hipMemcpyAsync costs ~8 us of host time,
	// a batch has two dozen tables); a large one goes table by table, by DMA straight from the host images that live in the context's pinned LM
	// arena (the lean builder's tables: the runtime recognises pinned memory) and through the runtime's own staging for the rest.
	const bool staged = up <= (size_t)8 << 20;
	const bool defer = b->defer_upload_wait && !staged && total > up;
	uint8_t* st = nullptr;
	if (staged && up) {
		int rc = ssk_arena_reserve(ctx, ctx->pin, up + 256);
		if (rc) return rc;
		ssk_arena_reset(ctx->pin);
		st = ssk_arena_take<uint8_t>(ctx->pin, up);
	}
	size_t off = 0;
	for (int pass = 0; pass < 2; ++pass)
		for (const ssk_lm_batch::req& r : b->reqs) {
			if ((r.copy > 0) != (pass == 0)) continue;
			uint8_t* d = static_cast<uint8_t*>(b->slab) + off;
			*r.dst = d;
			if (pass == 0) {
				if (staged) { std::memcpy(st + off, r.src, r.copy); std::memset(st + off + r.copy, 0, ssk_pad(r.bytes) - r.copy); }
				else {
					const void* src = r.src;
					const char* a0 = static_cast<const char*>(ctx->pin_lm.base);
					if (defer && !(a0 && static_cast<const char*>(src) >= a0 && static_cast<const char*>(src) + r.copy <= a0 + ctx->pin_lm.cap)) {
						std::unique_ptr<uint8_t[]> k(new uint8_t[r.copy]);
						std::memcpy(k.get(), src, r.copy);
						src = k.get();
						b->keep.push_back(std::move(k));
					}
					SSK_HIP(ctx, hipMemcpyAsync(d, src, r.copy, hipMemcpyHostToDevice, ctx->stream));
				}
			}
			off += ssk_pad(r.bytes);
		}
	if (staged && up) SSK_HIP(ctx, hipMemcpyAsync(b->slab, st, up, hipMemcpyHostToDevice, ctx->stream));
	if (!staged && total > up) {
		// the uploads' end marked, the fill queued behind it, only the uploads waited for
		if (!ctx->ev_lm_up) SSK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_lm_up, hipEventDisableTiming));
		SSK_HIP(ctx, hipEventRecord(ctx->ev_lm_up, ctx->stream));
		SSK_HIP(ctx, hipMemsetAsync(static_cast<uint8_t*>(b->slab) + up, 0, total - up, ctx->stream));
		SSK_HIP(ctx, hipEventCreateWithFlags(&b->ev_ready, hipEventDisableTiming));
		SSK_HIP(ctx, hipEventRecord(b->ev_ready, ctx->stream));
		b->ready_pending = true;
		if (defer) {
			ctx->lm_up_pending = true;                           // (the arena's next writer waits: lm_wait_uploads)
			// nobody waits here any more, and the runtime hands queued work to the device when somebody does: without this query the copies started
			// at the caller's next synchronisation -- a step later (measured: the step that solves the batch 8.2 -> 9.6 ms)
			(void)hipStreamQuery(ctx->stream);
		}
		else SSK_HIP(ctx, hipEventSynchronize(ctx->ev_lm_up));    // host images are free again when create returns
	} else {
		if (total > up) SSK_HIP(ctx, hipMemsetAsync(static_cast<uint8_t*>(b->slab) + up, 0, total - up, ctx->stream));
		SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));      // host images / pinned staging are free again when create returns
	}
	if (guard) { SSK_HIP(ctx, hipMemsetAsync(static_cast<uint8_t*>(b->slab) + b->guard_off, 0xA5, guard, ctx->stream)); SSK_HIP(ctx, hipStreamSynchronize(ctx->stream)); }
	b->reqs.clear();
	return SSK_OK;
}

// a10/a11: one thread per L<->R match: observation pixels + initial landmark position
// camera_pose_on_robot (+) projectMatchTo3D(xl, yl, xr)   (UTIL.h:558-574, EST.cpp:158,698), FP64, the
// operation order of the reference expression.
__global__ void build_obs_kernel(const ssk_keypoint* __restrict__ kl, const ssk_keypoint* __restrict__ kr,
                                 const ssk_dmatch* __restrict__ m, int n, lm_cam P,
                                 float* __restrict__ obs_px, double* __restrict__ init_pos)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const ssk_keypoint a = kl[m[i].queryIdx], b = kr[m[i].trainIdx];
	obs_px[4 * i] = a.x; obs_px[4 * i + 1] = a.y; obs_px[4 * i + 2] = b.x; obs_px[4 * i + 3] = b.y;
	const double ul = (double)a.x, vl = (double)a.y, ur = (double)b.x;
	const double b_d = P.baseline / __dadd_rn(__dmul_rn(P.fx_l, P.cx_r - ur), __dmul_rn(P.fx_r, ul - P.cx_l));
	const double X = __dmul_rn(__dmul_rn(b_d, P.fx_r), ul - P.cx_l);
	const double Y = __dmul_rn(__dmul_rn(b_d, P.fx_r), vl - P.cy_l);
	const double Z = __dmul_rn(__dmul_rn(b_d, P.fx_l), P.fx_r);
	for (int r = 0; r < 3; ++r)
		init_pos[3 * i + r] = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(P.S[4 * r], X), __dmul_rn(P.S[4 * r + 1], Y)), __dmul_rn(P.S[4 * r + 2], Z)), P.S[4 * r + 3]);
}

// the same for a resident keyframe batch: a thread per (keyframe, L<->R match slot), the stereo observation (xl, yl, xr, yr) the stereo kernel left
// per match -> the match's initial landmark position; the expression of build_obs_kernel term by term (bit for bit the same values)
__global__ void build_obs_batch_kernel(const float4* __restrict__ m_obs, const int32_t* __restrict__ nm, int kf_begin, int n_kf, int N, lm_cam P,
                                       double* __restrict__ init_pos)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_kf * N) return;
	const int kf = kf_begin + i / N, k = i - (i / N) * N;
	if (k >= nm[kf]) return;
	const float4 o = m_obs[(size_t)kf * N + k];
	const double ul = (double)o.x, vl = (double)o.y, ur = (double)o.z;
	const double b_d = P.baseline / __dadd_rn(__dmul_rn(P.fx_l, P.cx_r - ur), __dmul_rn(P.fx_r, ul - P.cx_l));
	const double X = __dmul_rn(__dmul_rn(b_d, P.fx_r), ul - P.cx_l);
	const double Y = __dmul_rn(__dmul_rn(b_d, P.fx_r), vl - P.cy_l);
	const double Z = __dmul_rn(__dmul_rn(b_d, P.fx_l), P.fx_r);
	for (int r = 0; r < 3; ++r)
		init_pos[3 * (size_t)i + r] = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(P.S[4 * r], X), __dmul_rn(P.S[4 * r + 1], Y)), __dmul_rn(P.S[4 * r + 2], Z)), P.S[4 * r + 3]);
}

// (e) multi-GPU exchange (BASELINE configs[4]).  A shared landmark is based at a keyframe that several submaps reach;
// no window optimises it (lm_unknown_slot -1), the exchange moves it:
//   pack     per shared landmark what THIS window's observations say at its current state: 12 doubles = position (3), g = sum Jf^T w r
//            (3), upper triangle of H = sum Jf^T Jf (6) -- observation model and IRLS weight of the LM contract.  A thread per shared
//            landmark walks the window's observations in order (fixed order: deterministic, no atomics).
//   stitch   after the ONE all-gather: g and H summed in RANK ORDER (bitwise identical on every GPU), Gauss-Newton step of the
//            landmark with every camera held: x <- x + (sum H + lambda I)^-1 sum g.
//   set      the new positions back into the window (state and initial state).
// Alternating this with the windows' own LM solves is block-coordinate descent on the JOINT problem of all submaps: its fixed point is
// the joint optimum (tests/test_shared_gpu.py, tests/test_dist_cpu.py: equal to a single-process solve of the union).
// Round 6: ONE WAVEFRONT per shared landmark (a thread per landmark walked the window's whole observation list alone: 1.5 ms of an exchange step on the
// bench's window): the lanes take the observations in turn, every lane adds up what ITS observations of the landmark say, and the 64 partial sums meet in
// a fixed shuffle tree -- one owner per observation, a fixed order: the same bits on every run and every GPU (the sums differ from a sequential walk's in
// the last place).
__device__ __forceinline__ void pack_one_observation(const lm_cam& P, const double* __restrict__ pose, const int* codes, int np, const double (&f)[3], const float z0, const float z1,
                                                     const float z2, const float z3, double (&acc)[9])
{
	double q[3] = {f[0], f[1], f[2]}, R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
	for (int i = np - 1; i >= 0; --i) {                    // from the base's end of the path to the observer
		const int c = codes[i];
		const double* T = pose + 12 * (c >> 1);
		const int dir = c & 1, si = dir ? 1 : 4, sj = dir ? 4 : 1;
		const double d0 = dir ? q[0] - T[3] : q[0], d1 = dir ? q[1] - T[7] : q[1], d2 = dir ? q[2] - T[11] : q[2];
		double t[3], Rn[9];
		for (int a = 0; a < 3; ++a) {
			const double r0 = T[si * a], r1 = T[si * a + sj], r2 = T[si * a + 2 * sj];
			t[a] = r0 * d0 + r1 * d1 + r2 * d2 + (dir ? 0.0 : T[4 * a + 3]);
			for (int b = 0; b < 3; ++b) Rn[3 * a + b] = r0 * R[b] + r1 * R[3 + b] + r2 * R[6 + b];
		}
		for (int a = 0; a < 3; ++a) q[a] = t[a];
		for (int a = 0; a < 9; ++a) R[a] = Rn[a];
	}
	const double e0 = q[0] - P.S[3], e1 = q[1] - P.S[7], e2 = q[2] - P.S[11];
	double pc[3];
	for (int a = 0; a < 3; ++a) pc[a] = P.S[a] * e0 + P.S[4 + a] * e1 + P.S[8 + a] * e2;
	const double iz = 1.0 / pc[2], iz2 = iz * iz;
	const double r[4] = {(double)z0 - (P.cx_l + P.fx_l * pc[0] * iz), (double)z1 - (P.cy_l + P.fy_l * pc[1] * iz),
	                     (double)z2 - (P.cx_r + P.fx_r * (pc[0] - P.baseline) * iz), (double)z3 - (P.cy_r + P.fy_r * pc[1] * iz)};
	const double s2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3];
	const double wgt = P.robust ? 1.0 / sqrt(sqrt(1.0 + s2 / (P.kernel * P.kernel))) : 1.0;
	const double Dm[12] = {P.fx_l * iz, 0, -P.fx_l * pc[0] * iz2, 0, P.fy_l * iz, -P.fy_l * pc[1] * iz2,
	                       P.fx_r * iz, 0, -P.fx_r * (pc[0] - P.baseline) * iz2, 0, P.fy_r * iz, -P.fy_r * pc[1] * iz2};
	double M[12], Jf[12];                                   // M = D R_S^T, Jf = w M R
	for (int a = 0; a < 4; ++a)
		for (int b = 0; b < 3; ++b) M[3 * a + b] = Dm[3 * a] * P.S[4 * b] + Dm[3 * a + 1] * P.S[4 * b + 1] + Dm[3 * a + 2] * P.S[4 * b + 2];
	for (int a = 0; a < 4; ++a)
		for (int b = 0; b < 3; ++b) Jf[3 * a + b] = wgt * (M[3 * a] * R[b] + M[3 * a + 1] * R[3 + b] + M[3 * a + 2] * R[6 + b]);
	for (int b = 0; b < 3; ++b) acc[b] += wgt * (Jf[b] * r[0] + Jf[3 + b] * r[1] + Jf[6 + b] * r[2] + Jf[9 + b] * r[3]);
	int k = 3;
	for (int a = 0; a < 3; ++a)
		for (int b = a; b < 3; ++b) acc[k++] += Jf[a] * Jf[b] + Jf[3 + a] * Jf[3 + b] + Jf[6 + a] * Jf[6 + b] + Jf[9 + a] * Jf[9 + b];
}
// the 64 lanes' partial sums, pairwise in a fixed tree; lane 0 writes position + g + H of the landmark
__device__ __forceinline__ void pack_reduce_store(double (&acc)[9], const double (&f)[3], double* __restrict__ o)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1)
#pragma unroll
		for (int c = 0; c < 9; ++c) acc[c] += __shfl_xor(acc[c], off, 64);
	if ((threadIdx.x & 63) == 0) {
		o[0] = f[0]; o[1] = f[1]; o[2] = f[2];
		for (int c = 0; c < 9; ++c) o[3 + c] = acc[c];
	}
}

// from the raw observation arrays (pose-only windows, batches of the full builder)
__global__ void pack_shared_kernel(lm_dev D, lm_cam P, int w, int n_shared, const int* __restrict__ lm_idx, double* __restrict__ out)
{
	const int s = blockIdx.x, lane = threadIdx.x & 63;
	if (s >= n_shared) return;
	const lm_win W = D.win[w];
	const int li = lm_idx[s];
	double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};            // g (3), H upper (6)
	double f[3] = {0, 0, 0};
	if (li >= 0 && li < W.n_lms) {
		const double* fp = D.lm + 3 * (size_t)(W.lm0 + li);
		f[0] = fp[0]; f[1] = fp[1]; f[2] = fp[2];
		const double* pose = D.pose + 12 * (size_t)W.edge0;
		for (int ob = lane; ob < W.n_obs; ob += 64) {
			if (D.obs_lm[W.obs0 + ob] != li) continue;
			const int pb = D.obs_path_off[W.opo0 + ob], np = min(D.obs_path_off[W.opo0 + ob + 1] - pb, 8);
			int codes[8];
			for (int i = 0; i < np; ++i) codes[i] = D.path_edge[W.path0 + pb + i];
			const float* z = D.obs_px + 4 * (size_t)(W.obs0 + ob);
			pack_one_observation(P, pose, codes, np, f, z[0], z[1], z[2], z[3], acc);
		}
	}
	pack_reduce_store(acc, f, out + 12 * (size_t)s);
}

// the same for a window of the node / phase-split families, from ITS tables (a lean batch keeps no raw observation arrays for them): the window's
// observation records (landmark, suffix of the whole path), the suffix trie (first step, rest) unrolled into the path's steps, the pixel records
__global__ void pack_shared_node_kernel(lm_dev D, nd_dev N, lm_cam P, int w, int n_shared, const int* __restrict__ lm_idx, double* __restrict__ out)
{
	const int s = blockIdx.x, lane = threadIdx.x & 63;
	if (s >= n_shared) return;
	const lm_win W = D.win[w];
	const nd_win NW = N.win[w];
	const int li = lm_idx[s];
	double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};            // g (3), H upper (6)
	double f[3] = {0, 0, 0};
	if (li >= 0 && li < W.n_lms) {
		const double* fp = D.lm + 3 * (size_t)(W.lm0 + li);
		f[0] = fp[0]; f[1] = fp[1]; f[2] = fp[2];
		const double* pose = D.pose + 12 * (size_t)W.edge0;
		for (int ob = lane; ob < NW.n_obs; ob += 64) {
			const nd_i2 rec = N.obs[NW.obs0 + ob];
			if (rec.x != li) continue;
			int codes[8], np = 0;
			for (int sf = rec.y & 0xFFFF; sf != 0 && np < 8; sf = N.suf[NW.suf0 + sf].y) codes[np++] = N.suf[NW.suf0 + sf].x;      // (first step first)
			const nd_f4 z = N.px[NW.obs0 + ob];
			pack_one_observation(P, pose, codes, np, f, z.x, z.y, z.z, z.w, acc);
		}
	}
	pack_reduce_store(acc, f, out + 12 * (size_t)s);
}

__global__ void stitch_shared_kernel(const double* __restrict__ g, int n_ranks, int n_shared, double lambda, double* __restrict__ pos)
{
	const int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= n_shared) return;
	double A[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
	for (int r = 0; r < n_ranks; ++r) {                    // rank order: the same sums, bit for bit, on every GPU
		const double* p = g + 12 * ((size_t)r * n_shared + s);
		b[0] += p[3]; b[1] += p[4]; b[2] += p[5];
		for (int c = 0; c < 6; ++c) A[c] += p[6 + c];
	}
	if (!(A[0] + A[3] + A[5] > 0.0)) return;               // nobody observes it: it stays where it is
	const double a = A[0] + lambda, bb = A[1], c = A[2], d = A[3] + lambda, e = A[4], f = A[5] + lambda;
	const double c00 = d * f - e * e, c01 = c * e - bb * f, c02 = bb * e - c * d;
	const double det = a * c00 + bb * c01 + c * c02;
	// a (near) singular block -- one far observation, depth barely observable, lambda 0 -- would put inf / NaN into the consensus position,
	// from where it reaches the window's initial state too: such a landmark stays where it is this round
	const double tr = a + d + f;
	if (!(det > 1e-12 * tr * tr * tr)) return;
	const double id = 1.0 / det;
	const double i00 = c00 * id, i01 = c01 * id, i02 = c02 * id, i11 = (a * f - c * c) * id, i12 = (bb * c - a * e) * id, i22 = (a * d - bb * bb) * id;
	pos[3 * s] += i00 * b[0] + i01 * b[1] + i02 * b[2];
	pos[3 * s + 1] += i01 * b[0] + i11 * b[1] + i12 * b[2];
	pos[3 * s + 2] += i02 * b[0] + i12 * b[1] + i22 * b[2];
}

__global__ void set_shared_kernel(lm_dev D, double* __restrict__ lm0, int w, int n_shared, const int* __restrict__ lm_idx, const double* __restrict__ pos)
{
	const int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= n_shared) return;
	const lm_win W = D.win[w];
	const int li = lm_idx[s];
	if (li < 0 || li >= W.n_lms) return;
	for (int c = 0; c < 3; ++c) { D.lm[3 * (size_t)(W.lm0 + li) + c] = pos[3 * s + c]; lm0[3 * (size_t)(W.lm0 + li) + c] = pos[3 * s + c]; }
}

}  // namespace

extern "C" {

static int lm_batch_settle(ssk_ctx* ctx, ssk_lm_batch* b);

static lm_cam cam_of(const ssk_lm_params* p)
{
	lm_cam P{};
	P.fx_l = p->fx_l; P.fy_l = p->fy_l; P.cx_l = p->cx_l; P.cy_l = p->cy_l;
	P.fx_r = p->fx_r; P.fy_r = p->fy_r; P.cx_r = p->cx_r; P.cy_r = p->cy_r; P.baseline = p->baseline;
	for (int i = 0; i < 12; ++i) P.S[i] = p->sensor_pose[i];
	P.max_iters = p->max_iters; P.robust = p->use_robust_kernel; P.kernel = p->kernel_param; P.tau = p->tau;
	P.max_lambda = p->max_lambda; P.stop_rmse = p->max_error_per_obs_to_stop; P.conv_rel = p->converged_rel_decrease;
	return P;
}

// end of an operation that queued work on the batch: remembered when the stream is not the creating context's (ssk_lm_batch_destroy waits for it)
static int lm_batch_mark(ssk_ctx* ctx, ssk_lm_batch* b, int rc)
{
	if (ctx == b->ctx) { b->used_own = true; return rc; }
	if (!b->ev_last) SSK_HIP(ctx, hipEventCreateWithFlags(&b->ev_last, hipEventDisableTiming));
	SSK_HIP(ctx, hipEventRecord(b->ev_last, ctx->stream));
	return rc;
}

int ssk_lm_batch_pack_shared(ssk_ctx* ctx, ssk_lm_batch* b, int w, int n_shared, const int32_t* d_lm_idx, const ssk_lm_params* p, double* d_out)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!b || w < 0 || w >= b->n_windows || n_shared < 0 || !p || (n_shared > 0 && (!d_out || !d_lm_idx)))
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_lm_batch_pack_shared: bad argument");
	if (!b->sub.empty()) return ssk_lm_batch_pack_shared(ctx, b->sub[(size_t)b->sub_of[(size_t)w]], b->idx_in[(size_t)w], n_shared, d_lm_idx, p, d_out);
	if (n_shared == 0) return SSK_OK;
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	{ const int rc0 = lm_batch_settle(ctx, b); if (rc0) return rc0; }
	if (b->lean && b->family[(size_t)w] != 2) {
		hipLaunchKernelGGL(pack_shared_node_kernel, dim3(n_shared), dim3(64), 0, ctx->stream, b->D, b->family[(size_t)w] == 1 ? b->N2 : b->N, cam_of(p), w, n_shared, d_lm_idx, d_out);
		SSK_LAUNCH_CHECK(ctx, "pack_shared_node_kernel");
	} else {
		hipLaunchKernelGGL(pack_shared_kernel, dim3(n_shared), dim3(64), 0, ctx->stream, b->D, cam_of(p), w, n_shared, d_lm_idx, d_out);
		SSK_LAUNCH_CHECK(ctx, "pack_shared_kernel");
	}
	return lm_batch_mark(ctx, b, SSK_OK);
}

int ssk_lm_batch_set_shared(ssk_ctx* ctx, ssk_lm_batch* b, int w, int n_shared, const int32_t* d_lm_idx, const double* d_pos)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!b || w < 0 || w >= b->n_windows || n_shared < 0 || (n_shared > 0 && (!d_pos || !d_lm_idx)))
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_lm_batch_set_shared: bad argument");
	if (!b->sub.empty()) return ssk_lm_batch_set_shared(ctx, b->sub[(size_t)b->sub_of[(size_t)w]], b->idx_in[(size_t)w], n_shared, d_lm_idx, d_pos);
	if (n_shared == 0) return SSK_OK;
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	{ const int rc0 = lm_batch_settle(ctx, b); if (rc0) return rc0; }      // (a timed-out cooperative launch is repaired first: its restore would overwrite what is written here)
	hipLaunchKernelGGL(set_shared_kernel, dim3((n_shared + 255) / 256), dim3(256), 0, ctx->stream, b->D, const_cast<double*>(b->D.lm0), w, n_shared, d_lm_idx, d_pos);
	SSK_LAUNCH_CHECK(ctx, "set_shared_kernel");
	return lm_batch_mark(ctx, b, SSK_OK);
}

int ssk_lm_batch_set_export(ssk_ctx* ctx, ssk_lm_batch* b, int on)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!b) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_lm_batch_set_export: batch is NULL");
	if (!b->sub.empty()) { for (ssk_lm_batch* sb : b->sub) { const int rcs = ssk_lm_batch_set_export(ctx, sb, on); if (rcs) return rcs; } return SSK_OK; }
	b->D.store_hff = on ? 1 : 0;
	return SSK_OK;
}

int ssk_lm_stitch_shared(ssk_ctx* ctx, const double* d_gathered, int n_ranks, int n_shared, double lambda, double* d_pos)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (n_ranks < 1 || n_shared < 0 || !(lambda >= 0.0) || (n_shared > 0 && (!d_gathered || !d_pos)))
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_lm_stitch_shared: bad argument");
	if (n_shared == 0) return SSK_OK;
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	hipLaunchKernelGGL(stitch_shared_kernel, dim3((n_shared + 255) / 256), dim3(256), 0, ctx->stream, d_gathered, n_ranks, n_shared, lambda, d_pos);
	SSK_LAUNCH_CHECK(ctx, "stitch_shared_kernel");
	return SSK_OK;
}

// The ONE collective of the path for callers without torch.distributed -- the C++ facade of a multi-GPU deployment:
// ncclAllGather (RCCL) of every rank's packed blocks on the context's stream, behind the pack kernel and ahead of the stitch kernel.
// RCCL is looked up at run time (dlopen: the copy the process already holds -- PyTorch ships one -- or librccl.so.1 of the ROCm
// installation), so libssk_hip.so carries no link-time dependency on it and a process never ends up with two RCCL instances.
int ssk_lm_exchange_shared(ssk_ctx* ctx, void* nccl_comm, const double* d_shared, double* d_gathered, int n_shared)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!nccl_comm || n_shared < 0 || (n_shared > 0 && (!d_shared || !d_gathered)))
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_lm_exchange_shared: bad argument");
	if (n_shared == 0) return SSK_OK;
	typedef int (*allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
	static allgather_fn fn = [] {
		void* h = nullptr;
		for (const char* name : {"librccl.so", "librccl.so.1"}) if ((h = dlopen(name, RTLD_NOW | RTLD_NOLOAD))) break;      // already in the process?
		if (!h) for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
		return h ? reinterpret_cast<allgather_fn>(dlsym(h, "ncclAllGather")) : nullptr;
	}();
	if (!fn) return ssk_set_error(ctx, SSK_ERR_HIP, "ssk_lm_exchange_shared: RCCL (librccl.so) is not available: %s", dlerror());
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	const int rc = fn(d_shared, d_gathered, (size_t)12 * (size_t)n_shared, 8 /* ncclFloat64 */, nccl_comm, ctx->stream);
	if (rc != 0) return ssk_set_error(ctx, SSK_ERR_HIP, "ssk_lm_exchange_shared: ncclAllGather failed (ncclResult_t %d)", rc);
	return SSK_OK;
}

int ssk_build_observations(ssk_ctx* ctx, const ssk_keypoint* kp_left, const ssk_keypoint* kp_right,
                           const ssk_dmatch* matches, int n_matches, const ssk_lm_params* cam,
                           float* obs_px, double* init_pos)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (n_matches < 0 || !cam || (n_matches > 0 && (!kp_left || !kp_right || !matches || !obs_px || !init_pos)))
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_build_observations: bad argument");
	if (n_matches == 0) return SSK_OK;
	int max_l = 0, max_r = 0;
	for (int i = 0; i < n_matches; ++i) {
		if (matches[i].queryIdx < 0 || matches[i].trainIdx < 0) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_build_observations: negative index in match %d", i);
		max_l = std::max(max_l, matches[i].queryIdx + 1); max_r = std::max(max_r, matches[i].trainIdx + 1);
	}
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	int rc;
	const size_t n = (size_t)n_matches;
	if ((rc = ssk_arena_reserve(ctx, ctx->dev, ssk_pad(sizeof(ssk_keypoint) * max_l) + ssk_pad(sizeof(ssk_keypoint) * max_r) +
	                                               ssk_pad(16 * n) + ssk_pad(16 * n) + ssk_pad(24 * n) + 8 * 256))) return rc;
	if ((rc = ssk_arena_reserve(ctx, ctx->pin, ssk_pad(16 * n) + ssk_pad(24 * n) + 4 * 256))) return rc;
	ssk_arena_reset(ctx->dev); ssk_arena_reset(ctx->pin);
	ssk_keypoint* d_kl = ssk_arena_take<ssk_keypoint>(ctx->dev, max_l);
	ssk_keypoint* d_kr = ssk_arena_take<ssk_keypoint>(ctx->dev, max_r);
	ssk_dmatch* d_m = ssk_arena_take<ssk_dmatch>(ctx->dev, n);
	float* d_px = ssk_arena_take<float>(ctx->dev, 4 * n);
	double* d_pos = ssk_arena_take<double>(ctx->dev, 3 * n);
	float* h_px = ssk_arena_take<float>(ctx->pin, 4 * n);
	double* h_pos = ssk_arena_take<double>(ctx->pin, 3 * n);
	SSK_HIP(ctx, hipMemcpyAsync(d_kl, kp_left, sizeof(ssk_keypoint) * max_l, hipMemcpyHostToDevice, ctx->stream));
	SSK_HIP(ctx, hipMemcpyAsync(d_kr, kp_right, sizeof(ssk_keypoint) * max_r, hipMemcpyHostToDevice, ctx->stream));
	SSK_HIP(ctx, hipMemcpyAsync(d_m, matches, sizeof(ssk_dmatch) * n, hipMemcpyHostToDevice, ctx->stream));
	lm_cam P{};
	P.fx_l = cam->fx_l; P.fy_l = cam->fy_l; P.cx_l = cam->cx_l; P.cy_l = cam->cy_l;
	P.fx_r = cam->fx_r; P.fy_r = cam->fy_r; P.cx_r = cam->cx_r; P.cy_r = cam->cy_r; P.baseline = cam->baseline;
	for (int i = 0; i < 12; ++i) P.S[i] = cam->sensor_pose[i];
	hipLaunchKernelGGL(build_obs_kernel, dim3((n_matches + 255) / 256), dim3(256), 0, ctx->stream, d_kl, d_kr, d_m, n_matches, P, d_px, d_pos);
	SSK_LAUNCH_CHECK(ctx, "build_obs_kernel");
	SSK_HIP(ctx, hipMemcpyAsync(h_px, d_px, 16 * n, hipMemcpyDeviceToHost, ctx->stream));
	SSK_HIP(ctx, hipMemcpyAsync(h_pos, d_pos, 24 * n, hipMemcpyDeviceToHost, ctx->stream));
	SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
	std::memcpy(obs_px, h_px, 16 * n);
	std::memcpy(init_pos, h_pos, 24 * n);
	return SSK_OK;
}

int ssk_build_observations_batch(ssk_ctx* ctx, ssk_kfbatch* b, int kf_begin, int kf_end, const ssk_lm_params* cam, float* d_obs_px, double* d_init_pos)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!b || !cam || kf_begin < 0 || kf_end > b->n_kf || kf_begin > kf_end || (!d_obs_px && !d_init_pos))
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_build_observations_batch: bad argument");
	const int n_kf = kf_end - kf_begin, N = b->max_feats;
	if (n_kf == 0) return SSK_OK;
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	if (d_obs_px)      // (the pixels are the stereo kernel's per-match observation records as they are)
		SSK_HIP(ctx, hipMemcpyAsync(d_obs_px, b->m_obs + (size_t)kf_begin * N, sizeof(float4) * (size_t)n_kf * N, hipMemcpyDeviceToDevice, ctx->stream));
	if (d_init_pos) {
		hipLaunchKernelGGL(build_obs_batch_kernel, dim3((unsigned)(((size_t)n_kf * N + 255) / 256)), dim3(256), 0, ctx->stream, b->m_obs, b->nm, kf_begin, n_kf, N, cam_of(cam), d_init_pos);
		SSK_LAUNCH_CHECK(ctx, "build_obs_batch_kernel");
	}
	return SSK_OK;
}

void ssk_lm_batch_destroy(ssk_lm_batch* b)
{
	if (!b) return;
	for (ssk_lm_batch* sb : b->sub) ssk_lm_batch_destroy(sb);      // (a composite owns its sub-batches and nothing else)
	b->sub.clear();
	// (a batch that was only ever run through OTHER contexts -- bench.py's creator thread, whose context is already queuing the next batch's uploads --
	//  waits for its own events below, not for everything its creating context has queued since)
	if (b->ctx) { (void)hipSetDevice(b->ctx->device); if (b->used_own || !b->lean) (void)hipStreamSynchronize(b->ctx->stream); }
	// the slab goes back to the creating context's cache (a later create may take it, a later destroy may free it): nothing may still be in flight on
	// it -- the create's own zero fill (when no stream waited for it yet) and whatever another context's stream was given to do on the batch
	if (b->ev_ready) { if (b->ready_pending) (void)hipEventSynchronize(b->ev_ready); (void)hipEventDestroy(b->ev_ready); }
	if (b->ev_last) { (void)hipEventSynchronize(b->ev_last); (void)hipEventDestroy(b->ev_last); }
	if (b->slab && b->guard_bytes) {
		std::vector<uint8_t> g(b->guard_bytes);
		(void)hipDeviceSynchronize();
		(void)hipMemcpy(g.data(), static_cast<uint8_t*>(b->slab) + b->guard_off, b->guard_bytes, hipMemcpyDeviceToHost);
		size_t bad = 0, first = 0;
		for (size_t i = 0; i < g.size(); ++i) if (g[i] != 0xA5) { if (!bad) first = i; ++bad; }
		if (bad) std::fprintf(stderr, "[ssk lm guard] batch of %d windows (node %d, split %d, pose %d): %zu guard bytes overwritten, first at +%zu behind %zu bytes of arrays\n",
		                      b->n_windows, b->node_n, b->split_n, b->pose_n, bad, first, b->guard_off);
	}
	if (b->slab) {
		ssk_ctx* c = b->ctx;
		// back to the context's cache: an empty place, else the place of a smaller slab (that one is freed), else freed
		int place = -1;
		if (c) {
			for (int k = 0; k < ssk_ctx::LM_SLABS && place < 0; ++k) if (!c->lm_slab[k]) place = k;
			if (place < 0) {
				int small = 0;
				for (int k = 1; k < ssk_ctx::LM_SLABS; ++k) if (c->lm_slab_bytes[k] < c->lm_slab_bytes[small]) small = k;
				if (c->lm_slab_bytes[small] < b->slab_bytes) { (void)hipFree(c->lm_slab[small]); place = small; }
			}
		}
		if (place >= 0) { c->lm_slab[place] = b->slab; c->lm_slab_bytes[place] = b->slab_bytes; }
		else (void)hipFree(b->slab);
	}
	delete b;
}

// argument checks of one window; E / L = optimised edges / landmarks, s_edge / s_lm = the slot -> index maps, n_path = path entries
static int lm_validate_window(ssk_ctx* ctx, const ssk_lm_window& Wn, int w, int& E, int& L, std::vector<int>& s_edge, std::vector<int>& s_lm, int& n_path)
{
	// (ctx == nullptr: no message, just the code -- the builder's threads check their windows side by side)
#define ssk_set_error(c, code, ...) ((c) ? ssk_set_error((c), (code), __VA_ARGS__) : (code))
	if (Wn.n_edges_total < 0 || Wn.n_lms_total < 0 || Wn.n_obs < 0 ||
	    (Wn.n_edges_total > 0 && (!Wn.edge_unknown_slot || !Wn.edge_inv_pose)) ||
	    (Wn.n_lms_total > 0 && (!Wn.lm_unknown_slot || !Wn.lm_pos)) ||
	    (Wn.n_obs > 0 && (!Wn.obs_lm || !Wn.obs_px || !Wn.obs_path_off)))
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_lm_batch_create: window %d has NULL arrays", w);
	E = 0; L = 0;
	for (int e = 0; e < Wn.n_edges_total; ++e) if (Wn.edge_unknown_slot[e] >= 0) ++E;
	for (int l = 0; l < Wn.n_lms_total; ++l) if (Wn.lm_unknown_slot[l] >= 0) ++L;
	if (6 * E > SSK_LM_MAX_N6) return ssk_set_error(ctx, SSK_ERR_CAPACITY, "ssk_lm_batch_create: window %d optimises %d edges (> %d)", w, E, SSK_LM_MAX_N6 / 6);
	s_edge.assign((size_t)E, -1); s_lm.assign((size_t)L, -1);
	for (int e = 0; e < Wn.n_edges_total; ++e) {
		const int s = Wn.edge_unknown_slot[e];
		if (s >= E || (s >= 0 && s_edge[s] != -1)) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_lm_batch_create: window %d edge slots are not a permutation of 0..E-1", w);
		if (s >= 0) s_edge[s] = e;
	}
	for (int l = 0; l < Wn.n_lms_total; ++l) {
		const int s = Wn.lm_unknown_slot[l];
		if (s >= L || (s >= 0 && s_lm[s] != -1)) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_lm_batch_create: window %d landmark slots are not a permutation of 0..L-1", w);
		if (s >= 0) s_lm[s] = l;
	}
	n_path = Wn.n_obs ? Wn.obs_path_off[Wn.n_obs] : 0;
	if (Wn.n_obs && Wn.obs_path_off[0] != 0) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_lm_batch_create: window %d obs_path_off[0] != 0", w);
	for (int o = 0; o < Wn.n_obs; ++o) {
		const int len = Wn.obs_path_off[o + 1] - Wn.obs_path_off[o];
		if (len < 0 || len > SSK_LM_MAX_PATH) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_lm_batch_create: window %d observation %d path length %d outside [0,%d]", w, o, len, SSK_LM_MAX_PATH);
		if (Wn.obs_lm[o] < 0 || Wn.obs_lm[o] >= Wn.n_lms_total) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_lm_batch_create: window %d observation %d landmark out of range", w, o);
	}
	for (int p = 0; p < n_path; ++p)
		if (!Wn.path_edge || Wn.path_edge[p] < 0 || (Wn.path_edge[p] >> 1) >= Wn.n_edges_total)
			return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_lm_batch_create: window %d path entry %d out of range", w, p);
#undef ssk_set_error
	return SSK_OK;
}

// The symbolic build's worker threads: created once (spawning 64 threads costs ~1.5 ms, each of a create's two parallel phases paid it), asleep on
// a condition variable between jobs.  run(): items 0 .. n-1 dealt out one at a time to `threads` workers and the caller; returns when all are done.
// One job at a time (creates on several host threads take turns).
struct lm_build_pool {
	std::mutex job_mx, mx;
	std::condition_variable cv, cv_done;
	std::vector<std::thread> workers;
	const std::function<void(int)>* fn = nullptr;
	std::atomic<int> next{0};
	int n_items = 0, gen = 0, want = 0, busy = 0;
	bool stop = false;
	void work()
	{
		int seen = 0;
		for (;;) {
			const std::function<void(int)>* f;
			{
				std::unique_lock<std::mutex> lk(mx);
				cv.wait(lk, [&] { return stop || (gen != seen && want > 0); });
				if (stop) return;
				seen = gen; --want; f = fn;
			}
			for (int w = next++; w < n_items; w = next++) (*f)(w);
			{
				std::lock_guard<std::mutex> lk(mx);
				if (--busy == 0) cv_done.notify_one();
			}
		}
	}
	void run(int threads, int n, const std::function<void(int)>& f)
	{
		std::lock_guard<std::mutex> job(job_mx);
		const int helpers = std::max(0, threads - 1);
		while ((int)workers.size() < helpers) workers.emplace_back([this] { work(); });
		{
			std::lock_guard<std::mutex> lk(mx);
			fn = &f; n_items = n; next = 0; want = helpers; busy = helpers; ++gen;
		}
		cv.notify_all();
		for (int w = next++; w < n; w = next++) f(w);
		std::unique_lock<std::mutex> lk(mx);
		cv_done.wait(lk, [&] { return busy == 0; });
		fn = nullptr;
	}
	~lm_build_pool()
	{
		{ std::lock_guard<std::mutex> lk(mx); stop = true; }
		cv.notify_all();
		for (std::thread& th : workers) th.join();
	}
};
static lm_build_pool& lm_pool() { static lm_build_pool p; return p; }

// ---- lean batch: node tables only ----
// A batch whose windows all fit the node builder (ssk_lm_node_build.h: <= 128 optimised edges, <= 64 observations and <= 63 nodes per landmark)
// needs none of the tiled / general structures: every window small enough for lm_solve_node_kernel (reduced system + a tile of records in LDS)
// -> the persistent node kernel; otherwise -> the phase-split kernels (ssk_lm_split.h), whatever the windows' sizes.  Returns 1 when the
// batch does not qualify (the caller falls back to the full builder), 0 with *out set, or an error code.
static int lm_batch_create_lean(ssk_ctx* ctx, const ssk_lm_window* windows, int n_windows, int trace, ssk_lm_batch** out, bool want_split, std::vector<char>* unfit = nullptr)
{
	const int MAX_ITERS_ALLOC = 64;
	static const bool timing = std::getenv("SSK_TIMING") != nullptr;
	const auto t_begin = std::chrono::steady_clock::now();
	const size_t node_lds_limit = (size_t)SSK_LMT_LDS_TOTAL - SSK_LMT_STATIC;
	// Three phases: (1) every window validated and its node tables built, side by side (a thread per window in turn); (2) prefix sums of all sizes;
	// (3) the batch-wide host images and tables filled, side by side again.  Nothing heavy is serial.
	struct lean_win { int E = 0, L = 0, n_path = 0; std::vector<int> s_edge, s_lm; nd_host nh; nd_sizes off; char split = 0, bad = 0; };
	std::vector<lean_win> lw((size_t)n_windows);
	// host threads of the symbolic build: SSK_LM_BUILD_THREADS, else half the logical cores, at most 64 -- and at most TWICE the container's CPU quota
	// (cgroup cpu.max; the GPU boxes of this pool: 16 CPUs of 256): measured back to back on such a box, 1 024 bench windows per create: 64 threads
	// 14.9 ms, 32 threads 9.7 ms, 16 threads 13.8 ms, 8 threads 22.2 ms (bursts above the quota are fine, sustained oversubscription is throttled)
	static const int n_thr_auto = [] {
		const int hc = (int)std::thread::hardware_concurrency();
		int n = std::max(1, std::min(hc / 2, 64));
		double quota = 0.0;
		if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {                  // cgroup v2: "<quota us | max> <period us>"
			char q[32]; double per = 0.0;
			if (std::fscanf(f, "%31s %lf", q, &per) == 2 && std::strcmp(q, "max") != 0 && per > 0) quota = std::atof(q) / per;
			std::fclose(f);
		} else if (FILE* f1 = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {      // cgroup v1
			double qv = 0.0, per = 0.0;
			if (std::fscanf(f1, "%lf", &qv) != 1) qv = 0.0;
			std::fclose(f1);
			if (FILE* f2 = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(f2, "%lf", &per) != 1) per = 0.0; std::fclose(f2); }
			if (qv > 0 && per > 0) quota = qv / per;
		}
		if (quota > 0.0) n = std::max(2, std::min(n, (int)(2.0 * quota + 0.5)));
		return n;
	}();
	const int n_thr_max = ctx->env.lm_build_threads > 0 ? ctx->env.lm_build_threads : n_thr_auto;
	const int n_thr_used = std::max(1, std::min(n_thr_max, n_windows / 8));      // (a thread is worth it from a handful of windows on)
	auto parallel = [&](const std::function<void(int)>& fn) {
		if (n_thr_used > 1) lm_pool().run(n_thr_used, n_windows, fn);
		else
			for (int w = 0; w < n_windows; ++w) fn(w);
	};
	// (a kernel forced by SSK_LM_KERNEL keeps every window; SSK_LM_POSE_ONLY=0: same-box A/B)
	// (SSK_LM_FEW_SPLIT=0: a small batch keeps the node kernel for every window that fits it -- tests of that kernel on single windows, same-box A/B)
	const bool few_windows = n_windows <= 8 && ctx->env.lm_kernel.empty() && ctx->env.lm_few_split;
	const bool pose_ok = !want_split && ctx->env.lm_pose_only;
	// ---- (1) per window: checks; its kernel family -- the pose-only kernel for a window without optimised landmarks, the persistent node kernel when the reduced system and a tile of records fit LDS, the
	// phase-split kernels otherwise (or for every window: SSK_LM_KERNEL=split): no all-or-nothing choice for a mixed batch --; its node tables ----
	parallel([&](int w) {
		const ssk_lm_window& Wn = windows[w];
		lean_win& X = lw[(size_t)w];
		if (lm_validate_window(nullptr, Wn, w, X.E, X.L, X.s_edge, X.s_lm, X.n_path)) { X.bad = 2; return; }
		// pose-only: the kernel reads the raw arrays (its edge poses, twice, in LDS: 48 KB at 256 edges -- inside the 64 KB a launch gets without asking)
		if (pose_ok && X.L == 0 && X.E >= 1 && X.E <= 2 && Wn.n_edges_total <= 256) { X.split = 2; return; }
		// (a handful of windows -- the un-batched drop-in -- cannot fill the device with one persistent workgroup each: a window of more than a few
		//  tiles goes to the phase-split kernels, which spread its tiles over workgroups: 3.5 -> ~1 ms for the 14-edge local areas at a stream's start)
		bool sp = want_split || 8 * nd_lds_doubles(Wn.n_edges_total, X.E, 4, 0) > node_lds_limit || (few_windows && Wn.n_obs > 1024);
		if (!sp) {
			if (!nd_build_window(Wn, SSK_SP_WAVES, X.nh, nd_win{}, node_lds_limit / 8, false)) { X.bad = 1; return; }
			if (8 * X.nh.lds_doubles > node_lds_limit) { sp = true; X.nh = nd_host(); }
		}
		if (sp && !nd_build_window(Wn, SSK_SP_WAVES, X.nh, nd_win{}, node_lds_limit / 8, true)) { X.bad = 1; return; }
		X.split = sp;
	});
	const auto t_ph1 = std::chrono::steady_clock::now();
	for (int w = 0; w < n_windows; ++w) {
		if (lw[(size_t)w].bad == 2) { int E, L, np; std::vector<int> a, c; return lm_validate_window(ctx, windows[w], w, E, L, a, c, np); }      // (again, for the message)
	}
	std::vector<ssk_lm_window> fit_windows;
	{
		int n_bad = 0;
		for (int w = 0; w < n_windows; ++w) n_bad += lw[(size_t)w].bad != 0;
		if (n_bad) {                                      // (which windows: the caller solves THEM with the full builder's kernels, not the whole batch)
			if (unfit) { unfit->assign((size_t)n_windows, 0); for (int w = 0; w < n_windows; ++w) (*unfit)[(size_t)w] = lw[(size_t)w].bad; }
			if (!unfit || want_split || n_bad == n_windows) return 1;
			// the batch of the windows that DO fit, from the tables just built (round 5 returned here and the caller ran this function again on the
			// fitting windows: the symbolic build twice, and the small-batch decisions taken on the sub-batch's size instead of the batch's)
			std::vector<lean_win> lw_fit;
			lw_fit.reserve((size_t)(n_windows - n_bad));
			for (int w = 0; w < n_windows; ++w)
				if (!lw[(size_t)w].bad) { fit_windows.push_back(windows[w]); lw_fit.push_back(std::move(lw[(size_t)w])); }
			lw.swap(lw_fit);
			windows = fit_windows.data(); n_windows = (int)fit_windows.size();
		}
	}
	// ---- (2) bases ----
	std::vector<lm_win> hdr((size_t)n_windows);
	long long h_total = 0, dpt = 0, dft = 0;
	int max_n6 = 0, n_node = 0, n_split = 0, n_pose = 0, n_pose2 = 0, max_pose_edges = 0;
	size_t n_edge_t = 0, n_lm_t = 0, n_obs_t = 0, n_opo_t = 0, n_path_t = 0, n_us_t = 0, n_ls_t = 0;
	nd_sizes zn, zs;
	for (int w = 0; w < n_windows; ++w) {
		const ssk_lm_window& Wn = windows[w];
		lean_win& X = lw[(size_t)w];
		lm_win& H = hdr[w];
		std::memset(&H, 0, sizeof H);
		const int E = X.E, L = X.L;
		H.n_edges = Wn.n_edges_total; H.E = E; H.n_lms = Wn.n_lms_total; H.L = L; H.n_obs = Wn.n_obs; H.n_path = X.n_path;
		H.n_pairs = E * (E + 1) / 2; H.n_tiles = -1;
		H.edge0 = (int)n_edge_t; H.lm0 = (int)n_lm_t; H.obs0 = (int)n_obs_t; H.opo0 = (int)n_opo_t; H.path0 = (int)n_path_t;
		H.us0 = (int)n_us_t; H.ls0 = (int)n_ls_t;
		H.trace0 = w * MAX_ITERS_ALLOC; H.h0 = 0; H.dp_trace0 = dpt; H.df_trace0 = dft;
		// (the dense bordered system in HBM is the phase-split kernels': node and pose-only windows keep theirs on chip -- a mixed batch of 1 024
		// windows would otherwise zero tens of MB per create for nothing)
		if (X.split == 1) { H.h0 = h_total; h_total += (long long)(6 * E + 1) * (6 * E + 2); }
		if (trace) { dpt += (long long)MAX_ITERS_ALLOC * 6 * E; dft += (long long)MAX_ITERS_ALLOC * 3 * L; }
		max_n6 = std::max(max_n6, 6 * E);
		n_edge_t += (size_t)Wn.n_edges_total; n_lm_t += (size_t)Wn.n_lms_total; n_us_t += (size_t)E; n_ls_t += (size_t)L;
		// the raw observation arrays (landmark, pixels, path offsets, path entries: 36 bytes an observation, a third of a batch's tables) are what the
		// pose-only kernel reads; the node and phase-split families read their own tables -- the observation and pixel records hold the same -- and so
		// does ssk_lm_batch_pack_shared for their windows (round 6: those windows' raw arrays are neither placed nor uploaded)
		if (X.split == 2) { n_obs_t += (size_t)Wn.n_obs; n_opo_t += (size_t)Wn.n_obs + 1; n_path_t += (size_t)X.n_path; }
		if (X.split == 2) { ++n_pose; if (E == 2) ++n_pose2; max_pose_edges = std::max(max_pose_edges, Wn.n_edges_total); continue; }
		nd_sizes& z = X.split ? zs : zn;
		X.off = z;
		nd_sizes_add(z, nd_sizes_of(X.nh));
		if (X.split) ++n_split; else ++n_node;
	}
	if (n_edge_t > 0x7FFFFFFF || n_obs_t > 0x7FFFFFFF || n_path_t > 0x7FFFFFFF || zn.item > 0x7FFFFFFF || zs.item > 0x7FFFFFFF)
		return ssk_set_error(ctx, SSK_ERR_CAPACITY, "ssk_lm_batch_create: the batch's tables exceed 2^31 entries");
	// the batch-wide host images and tables in the context's pinned arena (up to 4 GB; larger batches: pageable memory)
	nd_bump bump;
	nd_bump* arena = nullptr;
	{
		auto tb = [](const nd_sizes& z) {
			return sizeof(nd_i2) * (z.suf + z.obs + z.qs + z.gqs) + sizeof(nd_tile) * z.tile + sizeof(nd_group) * z.group + sizeof(nd_f4) * z.obs + sizeof(nd_i4) * (z.lane + z.sqs) +
			       sizeof(int) * (z.item + z.gitem + z.poff + z.goff + z.gslot);
		};
		const size_t need = tb(zn) + tb(zs) + sizeof(int) * (n_edge_t + n_lm_t + n_obs_t + n_opo_t + n_path_t + n_us_t + n_ls_t) + sizeof(double) * (12 * n_edge_t + 3 * n_lm_t) +
		                    sizeof(float) * 4 * n_obs_t + 64 * 64;
		const bool no_pin = ctx->env.lm_no_pinned;
		{ const int rcw = lm_wait_uploads(ctx); if (rcw) return rcw; }      // (the previous create's DMA reads the arena this one is about to fill)
		if (!no_pin && need <= ((size_t)4 << 30)) {       // (round 5: 4 GB -- 1 024 local areas of 5 k observations are 1.5-2.5 GB of tables; beyond 1 GB they went through pageable memory: 150 ms per create)
			// the arena only grows inside ssk_arena_reserve: after one large batch a context would hold up to 1 GB of pinned host memory until
			// ssk_destroy.  Eight creates in a row that need less than a quarter of it give it back (a streak, not one small create: bench.py and
			// the facade alternate batch sizes, and hipHostMalloc of a few hundred MB costs tens of ms).  No create of this context is in flight
			// here (a context serves one thread at a time, include/ssk_capi.h) and every earlier create waited for its uploads.
			if (ctx->pin_lm.base && ctx->pin_lm.cap > ((size_t)64 << 20) && 4 * (need + 4096) < ctx->pin_lm.cap) {
				if (++ctx->pin_lm_small_streak >= 8) {
					(void)hipHostFree(ctx->pin_lm.base);
					ctx->pin_lm.base = nullptr; ctx->pin_lm.cap = 0; ctx->pin_lm.used = 0; ctx->pin_lm_small_streak = 0;
				}
			} else ctx->pin_lm_small_streak = 0;
			// (capacity: the need + a quarter in steps of 64 MB, so "up to 4 GB" pins at most ~5; a refused pinned allocation -- locked-memory or
			// cgroup limits are likelier at this size -- is not an error of the create: the tables then go through pageable memory)
			const int rcp = ssk_arena_reserve(ctx, ctx->pin_lm, need, (size_t)64 << 20);
			if (rcp) { (void)hipGetLastError(); ctx->err.clear(); }
			else {
				bump.base = static_cast<char*>(ctx->pin_lm.base); bump.cap = ctx->pin_lm.cap; bump.used = 0;
				arena = &bump;
			}
		}
	}
	using ai = nd_noinit<int>; using ad = nd_noinit<double>; using af = nd_noinit<float>;
	nd_vec<int> edge_slot(n_edge_t, ai(arena)), lm_slot(n_lm_t, ai(arena)), obs_lm(n_obs_t, ai(arena)), obs_path_off(n_opo_t, ai(arena)), path_edge(n_path_t, ai(arena)),
	            slot_edge(n_us_t, ai(arena)), slot_lm(n_ls_t, ai(arena));
	nd_vec<double> pose(12 * n_edge_t, ad(arena)), lm(3 * n_lm_t, ad(arena));
	nd_vec<float> obs_px(4 * n_obs_t, af(arena));
	nd_host NHn, NHs;                                      // the two families' tables; win[] (and swin[]) of both are indexed by the batch's window number
	if (n_node) nd_resize(NHn, zn, (size_t)n_windows, false, arena);
	if (n_split) nd_resize(NHs, zs, (size_t)n_windows, true, arena);
	std::vector<char> w_split((size_t)n_windows, 0);
	const auto t_ph2 = std::chrono::steady_clock::now();
	// ---- (3) placement ----
	parallel([&](int w) {
		const ssk_lm_window& Wn = windows[w];
		lean_win& X = lw[(size_t)w];
		const lm_win& H = hdr[w];
		auto put = [](auto* dst, const auto* src, size_t n) { if (n) std::memcpy(dst, src, sizeof(*src) * n); };
		put(&edge_slot[0] + H.edge0, Wn.edge_unknown_slot, (size_t)Wn.n_edges_total);
		put(&pose[0] + 12 * (size_t)H.edge0, Wn.edge_inv_pose, 12 * (size_t)Wn.n_edges_total);
		put(&lm_slot[0] + H.lm0, Wn.lm_unknown_slot, (size_t)Wn.n_lms_total);
		put(&lm[0] + 3 * (size_t)H.lm0, Wn.lm_pos, 3 * (size_t)Wn.n_lms_total);
		if (X.split == 2) {
			put(&obs_lm[0] + H.obs0, Wn.obs_lm, (size_t)Wn.n_obs);
			put(&obs_px[0] + 4 * (size_t)H.obs0, Wn.obs_px, 4 * (size_t)Wn.n_obs);
			if (Wn.n_obs) put(&obs_path_off[0] + H.opo0, Wn.obs_path_off, (size_t)Wn.n_obs + 1); else obs_path_off[(size_t)H.opo0] = 0;
			put(&path_edge[0] + H.path0, Wn.path_edge, (size_t)X.n_path);
		}
		put(&slot_edge[0] + H.us0, X.s_edge.data(), X.s_edge.size());
		put(&slot_lm[0] + H.ls0, X.s_lm.data(), X.s_lm.size());
		w_split[(size_t)w] = X.split;
		if (X.split == 2) { std::vector<int>().swap(X.s_edge); std::vector<int>().swap(X.s_lm); return; }
		nd_win& nw = X.nh.win[0];
		nw.edge0 = H.edge0; nw.lm0 = H.lm0; nw.us0 = H.us0; nw.ls0 = H.ls0; nw.trace0 = H.trace0; nw.dp_trace0 = H.dp_trace0; nw.df_trace0 = H.df_trace0;
		nw.h0 = H.h0;
		nd_place(X.split ? NHs : NHn, (size_t)w, X.off, X.nh);
		w_split[(size_t)w] = X.split;
		// the window's own tables are done with: freed here, by the thread that filled them (1 024 windows' worth freed on one thread cost 2.7 ms)
		nd_host keep;
		keep.lds_doubles = X.nh.lds_doubles; keep.lds_doubles_big = X.nh.lds_doubles_big; keep.max_items_per_lane = X.nh.max_items_per_lane; keep.max_tile_blocks = X.nh.max_tile_blocks;
		X.nh = std::move(keep);
		std::vector<int>().swap(X.s_edge); std::vector<int>().swap(X.s_lm);
	});
	for (int w = 0; w < n_windows; ++w) {
		if (lw[(size_t)w].split == 2) continue;
		nd_host& Dh = lw[(size_t)w].split ? NHs : NHn;
		const nd_host& W1 = lw[(size_t)w].nh;
		Dh.lds_doubles = std::max(Dh.lds_doubles, W1.lds_doubles); Dh.lds_doubles_big = std::max(Dh.lds_doubles_big, W1.lds_doubles_big);
		Dh.max_items_per_lane = std::max(Dh.max_items_per_lane, W1.max_items_per_lane); Dh.max_tile_blocks = std::max(Dh.max_tile_blocks, W1.max_tile_blocks);
	}
	const auto t_ph3 = std::chrono::steady_clock::now();
	lw.clear();
	const auto t_sym = std::chrono::steady_clock::now();
	ssk_lm_batch* b = new ssk_lm_batch();
	b->ctx = ctx; b->n_windows = n_windows; b->trace = trace; b->max_iters_alloc = MAX_ITERS_ALLOC; b->max_n6 = max_n6;
	b->hdr = hdr; b->n_pose = pose.size(); b->n_lm = lm.size();
	b->lean = true; b->node_ok = n_node > 0; b->split_ok = n_split > 0; b->node_n = n_node; b->split_n = n_split;
	b->family = w_split;
	lm_dev& D = b->D;
	int rc = SSK_OK;
#define UP(vec, field) if (!rc) rc = upload(b, vec, &D.field)
	UP(hdr, win);
	UP(edge_slot, edge_slot); UP(lm_slot, lm_slot); UP(obs_lm, obs_lm); UP(obs_px, obs_px); UP(obs_path_off, obs_path_off);
	UP(path_edge, path_edge); UP(slot_edge, slot_edge); UP(slot_lm, slot_lm);
	UP(pose, pose0); UP(lm, lm0);
#undef UP
	// launch order / XCD placement: windows by observations x optimised edges, largest first
	const bool by_size = ctx->env.lm_order;
	auto ordered = [&](const nd_host& NH, bool sp) {
		std::vector<int> ord;
		for (int w = 0; w < n_windows; ++w) if (w_split[(size_t)w] == (sp ? 1 : 0)) ord.push_back(w);
		if (by_size)
			std::stable_sort(ord.begin(), ord.end(), [&](int a, int c) { return (long long)NH.win[(size_t)a].n_obs * (NH.win[(size_t)a].E + 4) > (long long)NH.win[(size_t)c].n_obs * (NH.win[(size_t)c].E + 4); });
		return ord;
	};
	std::vector<int> ord_n, win_map;
	std::vector<sp_tile> tile_map;
	if (n_node) {
#define UPN(vec, field) if (!rc) rc = upload(b, NHn.vec, &b->N.field)
		UPN(win, win); UPN(tile, tile); UPN(group, group); UPN(suf, suf); UPN(obs, obs); UPN(px, px); UPN(lane, lane);
		UPN(item, item); UPN(gitem, gitem); UPN(gqs, gqs); UPN(qs, qs);
#undef UPN
		ord_n = ordered(NHn, false);
		if (!rc) rc = upload(b, ord_n, &b->N.order);
		b->node_dyn_bytes = 8 * NHn.lds_doubles;
		b->node_small = true;
		for (const nd_win& nw : NHn.win) b->node_small = b->node_small && nw.E <= 10;      // (placeholders of the other families: E = 0)
	}
	if (n_split) {
#define UPN(vec, field) if (!rc) rc = upload(b, NHs.vec, &b->N2.field)
		UPN(win, win); UPN(tile, tile); UPN(group, group); UPN(suf, suf); UPN(obs, obs); UPN(px, px); UPN(lane, lane);
		UPN(item, item); UPN(gitem, gitem); UPN(gqs, gqs);
#undef UPN
		const std::vector<int> ord = ordered(NHs, true);
		// a window's workgroups all go to ONE XCD (workgroup i runs on XCD i mod 8): its suffix table, landmarks and scratch stay in that L2
		constexpr int NX = 8;
		std::vector<std::vector<nd_i2>> tl(NX);
		std::vector<std::vector<int>> wl(NX);
		std::vector<long long> load(NX, 0);
		// (a handful of windows -- the un-batched drop-in: ONE local area of 40-60 tiles -- cannot fill an XCD's L2 anyway: its tiles go round all eight
		// XCDs, a CU each, instead of two or three to a CU of one XCD; SSK_SP_SPREAD=0 / 1 forces either form)
		const bool spread = ctx->env.sp_spread >= 0 ? ctx->env.sp_spread != 0 : n_split <= 4;
		for (int w : ord) {
			int x = 0;
			for (int k = 1; k < NX; ++k) if (load[k] < load[x]) x = k;
			if (spread) {
				for (int t = 0; t < NHs.win[(size_t)w].n_tiles; ++t) { tl[(size_t)((x + t) % NX)].push_back(nd_i2{w, t}); load[(size_t)((x + t) % NX)] += 1; }
				wl[x].push_back(w);
				continue;
			}
			for (int t = 0; t < NHs.win[(size_t)w].n_tiles; ++t) tl[x].push_back(nd_i2{w, t});
			wl[x].push_back(w);
			load[x] += NHs.win[(size_t)w].n_tiles;
		}
		size_t mt = 0, mw = 0;
		for (int x = 0; x < NX; ++x) { mt = std::max(mt, tl[x].size()); mw = std::max(mw, wl[x].size()); }
		sp_tile none{};
		none.wi = -1;
		tile_map.assign(mt * NX, none);
		win_map.assign(mw * NX, -1);
		for (int x = 0; x < NX; ++x) {
			for (size_t k = 0; k < tl[x].size(); ++k) {
				const nd_i2 wt = tl[x][k];
				const nd_win& nw = NHs.win[(size_t)wt.x];
				const nd_split& sw = NHs.swin[(size_t)wt.x];
				const nd_tile& T = NHs.tile[(size_t)nw.tile0 + wt.y];
				sp_tile d{};
				d.wi = wt.x; d.tile = wt.y; d.grp = nw.grp0 + T.grp0; d.n_grp = T.n_grp; d.sqs = sw.sqs0 + T.qs0; d.n_slots = T.n_rounds;
				d.gqs = nw.gqs0 + T.gqs0; d.n_gqs = T.n_gqs; d.gslot = sw.gslot0 + T.gqs0; d.gtile = sw.gtile0 + wt.y;
				d.item0 = nw.item0; d.gitem0 = nw.gitem0;
				tile_map[k * NX + x] = d;
			}
			for (size_t k = 0; k < wl[x].size(); ++k) win_map[k * NX + x] = wl[x][k];
		}
		b->S.n_tile_wgs = (int)tile_map.size(); b->S.n_win_wgs = (int)win_map.size();
		b->S.max_pairs = 0;
		for (const nd_win& nw : NHs.win) b->S.max_pairs = std::max(b->S.max_pairs, nw.E * (nw.E + 1) / 2);
		if (!rc) rc = upload(b, NHs.swin, &b->S.sw);
		if (!rc) rc = upload(b, NHs.sqs, &b->S.sqs);
		if (!rc) rc = upload(b, NHs.poff, &b->S.poff);
		if (!rc) rc = upload(b, NHs.goff, &b->S.goff);
		if (!rc) rc = upload(b, NHs.gslot, &b->S.gslot);
		if (!rc) rc = upload(b, tile_map, &b->S.tile_map);
		if (!rc) rc = upload(b, win_map, &b->S.win_map);
		size_t sd = 0, id = 0;
		int E_max = 0;
		for (const nd_win& nw : NHs.win) E_max = std::max(E_max, nw.E);
		int mfma = ssk_sp_solve_mfma(E_max, ctx->env.k2_regtile);
		for (int attempt = 0; attempt < 2; ++attempt) {
			sd = 0; id = 0;
			for (const nd_win& nw : NHs.win) {
				sd = std::max(sd, ssk_sp_solve_lds_doubles(nw.n_edges, nw.n_suf, nw.E, mfma));
				id = std::max(id, 12 * (size_t)nw.n_edges + (size_t)nw.n_suf + 2);
			}
			if (!mfma || 8 * sd <= node_lds_limit) break;
			mfma = 0;                                  // (a window with so many fixed edges that the MFMA solver's staging does not fit beside them)
		}
		b->S.solve_mfma = mfma;
		b->split_solve_dyn_bytes = 8 * sd; b->split_init_dyn_bytes = 8 * id;
		if (b->split_solve_dyn_bytes > node_lds_limit || b->split_init_dyn_bytes > node_lds_limit) { delete b; return 1; }
	}
	std::vector<int> ord_p;
	if (n_pose) {
		for (int w = 0; w < n_windows; ++w) if (w_split[(size_t)w] == 2) ord_p.push_back(w);
		if (by_size) std::stable_sort(ord_p.begin(), ord_p.end(), [&](int a, int c) { return hdr[(size_t)a].n_obs > hdr[(size_t)c].n_obs; });
		if (!rc) rc = upload(b, ord_p, &b->pose_order);
		b->pose_n = n_pose; b->pose_two = n_pose2; b->pose_dyn_bytes = sizeof(double) * 24 * (size_t)max_pose_edges;
	}
	D.store_hff = 0;
	if (!rc) rc = upload(b, pose, const_cast<const double**>(&D.pose));
	if (!rc) rc = upload(b, lm, const_cast<const double**>(&D.lm));
	const size_t n_ls = slot_lm.size(), n_us = slot_edge.size();
	if (!rc) rc = scratch(b, pose.size(), &D.pose_new);
	if (!rc) rc = scratch(b, lm.size(), &D.lm_new);
	if (!rc) rc = scratch(b, 9 * n_ls, &D.Hff);
	if (!rc) rc = scratch(b, 3 * n_ls, &D.gf);
	if (!rc) rc = scratch(b, n_split ? (size_t)h_total : 0, &D.Hred);
	if (!rc) rc = scratch(b, 6 * n_us, &D.dp);
	if (!rc) rc = scratch(b, (size_t)n_windows, &D.res);
	if (!rc) rc = scratch(b, trace ? (size_t)n_windows * MAX_ITERS_ALLOC : 1, &D.trace);
	if (!rc) rc = scratch(b, (size_t)dpt, &D.dp_trace);
	if (!rc) rc = scratch(b, (size_t)dft, &D.df_trace);
	if (n_node) {
		if (!rc) rc = scratch(b, (size_t)NHn.tab_doubles, &b->N.tab);
		if (!rc) rc = scratch(b, (size_t)SSK_ND_SCR * NHn.lane.size(), &b->N.scr);
	}
	if (n_split) {
		if (!rc) rc = scratch(b, (size_t)NHs.tab_doubles, &b->N2.tab);
		if (!rc) rc = scratch(b, (size_t)SSK_ND_SCR * NHs.lane.size(), &b->N2.scr);
		if (!rc) rc = scratch(b, 36 * (size_t)NHs.part_slots + 36, &b->S.Hpart);
		if (!rc) rc = scratch(b, 6 * (size_t)NHs.gpart_slots + 6, &b->S.gpart);
		if (!rc) rc = scratch(b, NHs.tile.size(), &b->S.vpart);
		if (!rc) rc = scratch(b, NHs.tile.size(), &b->S.epart);
		if (!rc) rc = scratch(b, NHs.tile.size(), &b->S.fpart);
		if (!rc) rc = scratch(b, (size_t)n_windows, &b->S.ctl);
		if (!rc) rc = scratch(b, 4, &b->S.n_active);
	}
	D.trace_on = trace ? 1 : 0;
	const auto t_fin = std::chrono::steady_clock::now();
	b->defer_upload_wait = arena != nullptr && !ctx->env.lm_sync_upload;
	if (!rc) rc = finalize(b);
	if (rc) { ssk_lm_batch_destroy(b); return rc; }
	if (ctx->env.lm_debug)
		std::fprintf(stderr, "[ssk lm] lean batch: %d windows for the node kernel, %d for the phase-split kernels (%zu tiles, %zu items, %lld partial slots, most blocks of a tile %d)\n",
		             n_node, n_split, NHs.tile.size(), NHs.item.size(), NHs.part_slots, NHs.max_tile_blocks);
	b->symbolic_ms = std::chrono::duration<double, std::milli>(t_sym - t_begin).count();
	b->symbolic_threads = n_thr_used;
	if (timing) {
		(void)hipDeviceSynchronize();
		auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) { return std::chrono::duration<double, std::milli>(c - a).count(); };
		std::fprintf(stderr, "[ssk timing] ssk_lm_batch_create, lean (%d windows: %d node, %d split): symbolic %.3f ms (build %.3f, bases + arena %.3f, place %.3f, free %.3f), maps + requests %.3f ms, "
		             "allocate + upload %.3f ms (%.1f MB up, %.1f MB zeroed)\n", n_windows, n_node, n_split,
		             ms(t_begin, t_sym), ms(t_begin, t_ph1), ms(t_ph1, t_ph2), ms(t_ph2, t_ph3), ms(t_ph3, t_sym), ms(t_sym, t_fin), ms(t_fin, std::chrono::steady_clock::now()), b->up_bytes / 1e6, b->zero_bytes / 1e6);
	}
	*out = b;
	return SSK_OK;
}

// with_general: also build what only the general scratch-in-L2 kernel reads (the q entries with their co-observation lists,
// the chunk lists, its scratch arrays).  The first attempt leaves them out; a batch with a window the tiler cannot take (or
// SSK_LM_KERNEL=l2) is built again with them.
static int lm_batch_create_impl(ssk_ctx* ctx, const ssk_lm_window* windows, int n_windows, int trace, ssk_lm_batch** out, bool with_general, bool want_acc)
{
	// SSK_LM_SCHUR (read when the batch is built: the two forms lay the tile out differently): mfma = Y panels + v_mfma_f64_16x16x4
	// into LDS-resident 16x16 tiles, sparse = 6x6 items Y_i Y_j^T as in the slot kernels
	// Default by size (profiles/r02_lm_mfma_vs_sparse.json: the two forms are within 1-7 % of each other at every size measured): matrix
	// cores where the reduced system is larger than the on-chip accumulation limit (E > 12: the local areas of the un-batched drop-in
	// path, E ~ 40), sparse items below it (the E = 10 windows of the batched bench: 8.59 ms against 8.72 per 1024 windows).
	const char* schur_env = ctx->env.lm_schur == 1 ? "mfma" : ctx->env.lm_schur == 2 ? "sparse" : nullptr;
	int max_E = 0;
	for (int w = 0; windows && w < n_windows; ++w) {
		int E = 0;
		for (int e = 0; e < windows[w].n_edges_total && windows[w].edge_unknown_slot; ++e) if (windows[w].edge_unknown_slot[e] >= 0) ++E;
		max_E = std::max(max_E, E);
	}
	// (a batch of many large windows runs one workgroup per window: there the sparse items win -- 91 ms against 130 ms per 1024 depth-5
	// local areas of 40 edges, profiles/r03_lm_depth5.json; the matrix cores keep the few-large-windows case of the cooperative kernel)
	const int schur_mfma_default = max_E > SSK_LMA_MAX_E && ctx && n_windows * 4 <= ctx->n_cu;
	// the matrix-core form keeps a landmark's 16-row-tile set in a 32-bit mask: 6E + 15 <= 16 * 31, i.e. E <= 80
	const bool mfma_fits = ssk_lma_row_tiles(max_E) <= 31;
	const int schur_mfma = mfma_fits && (schur_env ? (std::strcmp(schur_env, "sparse") != 0) : schur_mfma_default);
	const char* force_k = ctx->env.lm_kernel.empty() ? nullptr : ctx->env.lm_kernel.c_str();
	if (want_acc && windows && n_windows > 0) {            // cheap look ahead: one window with too many optimised edges rules the acc kernel out
		for (int w = 0; w < n_windows && want_acc; ++w) {
			int E = 0;
			for (int e = 0; e < windows[w].n_edges_total && windows[w].edge_unknown_slot; ++e) if (windows[w].edge_unknown_slot[e] >= 0) ++E;
			if (E > SSK_LMA_MAX_E) want_acc = false;
		}
	}
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!windows || n_windows < 1 || !out) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_lm_batch_create: bad argument");
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	const int MAX_ITERS_ALLOC = 64;
	static const bool timing = std::getenv("SSK_TIMING") != nullptr;
	const auto t_begin = std::chrono::steady_clock::now();
	double t_sub[4] = {0, 0, 0, 0};                        // inside the tile loop: footprints, cut, records (rest: groups and tasks)
	double t_ph[5] = {0, 0, 0, 0, 0};                      // per window: validation + incidences, suffix trie, tile loop, concatenation
	auto tnow = [] { return std::chrono::steady_clock::now(); };
	auto tms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) { return std::chrono::duration<double, std::milli>(c - a).count(); };
	// batch-wide host images of every array family
	std::vector<int> edge_slot, lm_slot, obs_lm, obs_path_off, path_edge, path_obs, slot_edge, slot_lm;
	std::vector<int> lmo_off, lmo_idx, lsm_off, lme_edge, lme_ls, lme_elpos, lmep_off, lmep_idx, q_off, q_uend, qo_off, el_off, el_idx;
	std::vector<int2> pair_ee, q_mm, qo;
	std::vector<int> qp_q, pc_off, sc_off;
	std::vector<int3> pchunk, schunk;
	std::vector<double> pose, lm;
	std::vector<float> obs_px;
	std::vector<lm_win> hdr((size_t)n_windows);
	std::vector<lm_tile> tiles;
	std::vector<int4> tobs_a;
	std::vector<float4> tobs_px;
	std::vector<int4> tl_rec, inc_rec;
	std::vector<int> lmep_loc, pit, pslot0, pcl_off, pcl_idx, qit, sslot0, scl_off, scl_idx;
	std::vector<int2> pgrp, sgrp, tl_pan, suf_rec;
	std::vector<int> wt_off, wt, tent, tsuf, st_off;
	std::vector<int2> mt;
	std::vector<int4> ilane;
	long long gtab_total = 0, n_stile_total = 0;
	bool acc_all = want_acc;
	size_t n_pslots_total = 0, n_sslots_total = 0, tiled_dyn = 0;
	bool lds_sys_all = true;
	bool tiled_all = true;
	long long h_total = 0, dpt = 0, dft = 0;
	int max_n6 = 0;
	for (int w = 0; w < n_windows; ++w) {
		const auto tp0 = tnow();
		const ssk_lm_window& Wn = windows[w];
		lm_win& H = hdr[w];
		int E = 0, L = 0, n_path = 0;
		std::vector<int> s_edge, s_lm;
		{ const int rcv = lm_validate_window(ctx, Wn, w, E, L, s_edge, s_lm, n_path); if (rcv) return rcv; }
		// ---- symbolic structure ----
		// per landmark index: its (edge slot -> path entries) incidences
		std::vector<int> p_obs((size_t)n_path);
		// the per-landmark / per-block lists below are tens of thousands of small vectors for a 15 k-observation local area: they come
		// from one monotonic arena per window (malloc / free per list was a third of the symbolic time)
		std::pmr::monotonic_buffer_resource pool(size_t(1) << 20);
		using ivec = std::pmr::vector<int>;
		using ivecs = std::pmr::vector<ivec>;
		std::pmr::vector<std::pmr::vector<std::pair<int, int>>> inc((size_t)Wn.n_lms_total, &pool);   // (edge slot, path entry)
		ivecs lm_obs((size_t)L, &pool);
		for (int o = 0; o < Wn.n_obs; ++o) {
			const int li = Wn.obs_lm[o];
			const int ls = Wn.lm_unknown_slot[li];
			if (ls >= 0) lm_obs[ls].push_back(o);
			for (int p = Wn.obs_path_off[o]; p < Wn.obs_path_off[o + 1]; ++p) {
				p_obs[p] = o;
				const int es = Wn.edge_unknown_slot[Wn.path_edge[p] >> 1];
				if (es >= 0) inc[li].push_back(std::make_pair(es, p));
			}
		}
		// landmark order of the lme groups: optimised landmarks in slot order, then fixed ones
		std::vector<int> order;
		for (int s = 0; s < L; ++s) order.push_back(s_lm[s]);
		for (int l = 0; l < Wn.n_lms_total; ++l) if (Wn.lm_unknown_slot[l] < 0 && !inc[l].empty()) order.push_back(l);
		std::vector<int> w_lme_edge, w_lme_ls, w_lmep_off, w_lmep_idx, w_lsm_off;
		std::vector<int> path_to_lme((size_t)n_path, -1);
		ivecs lm_lmes((size_t)order.size(), &pool);
		int n_lme_unknown = 0;
		for (size_t oi = 0; oi < order.size(); ++oi) {
			const int l = order[oi];
			if ((int)oi < L) w_lsm_off.push_back((int)w_lme_edge.size());
			auto& v = inc[l];
			std::stable_sort(v.begin(), v.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first < b.first; });
			for (size_t k = 0; k < v.size();) {
				const int es = v[k].first;
				const int m = (int)w_lme_edge.size();
				w_lme_edge.push_back(es);
				w_lme_ls.push_back(Wn.lm_unknown_slot[l]);
				w_lmep_off.push_back((int)w_lmep_idx.size());
				for (; k < v.size() && v[k].first == es; ++k) { w_lmep_idx.push_back(v[k].second); path_to_lme[v[k].second] = m; }
				lm_lmes[oi].push_back(m);
			}
			if ((int)oi == L - 1) n_lme_unknown = (int)w_lme_edge.size();
		}
		if (L > 0) w_lsm_off.push_back(n_lme_unknown); else n_lme_unknown = 0;
		w_lmep_off.push_back((int)w_lmep_idx.size());
		const int n_lme = (int)w_lme_edge.size();
		// q entries: per landmark, pairs of its lme entries (edge slots ascending), then sorted pid-major
		auto pid_of = [](int e, int e2) { return e2 * (e2 + 1) / 2 + e; };   // e <= e2
		const int n_pairs = E * (E + 1) / 2;
		int n_q = 0;
		std::vector<int> w_q_off((size_t)n_pairs + 1, 0), w_qo_off;
		std::vector<int2> w_q_mm, w_qo;
		std::vector<int> w_q_uend((size_t)n_pairs + 1, 0);
		if (with_general) {
		struct qrec { int pid, lm_order, mi, mj; };
		std::vector<qrec> qs;
		for (size_t oi = 0; oi < order.size(); ++oi)
			for (size_t i = 0; i < lm_lmes[oi].size(); ++i)
				for (size_t j = i; j < lm_lmes[oi].size(); ++j)
					qs.push_back({pid_of(w_lme_edge[lm_lmes[oi][i]], w_lme_edge[lm_lmes[oi][j]]), (int)oi, lm_lmes[oi][i], lm_lmes[oi][j]});
		std::stable_sort(qs.begin(), qs.end(), [](const qrec& a, const qrec& b) { return a.pid != b.pid ? a.pid < b.pid : a.lm_order < b.lm_order; });
		n_q = (int)qs.size();
		for (const qrec& r : qs) { w_q_off[r.pid + 1]++; if (r.lm_order < L) w_q_uend[r.pid]++; }
		for (int p = 0; p < n_pairs; ++p) w_q_off[p + 1] += w_q_off[p];
		for (int p = 0; p < n_pairs; ++p) w_q_uend[p] += w_q_off[p];     // optimised landmarks sort first inside a block
		for (const qrec& r : qs) {
			w_q_mm.push_back(make_int2(r.mi, r.mj));
			w_qo_off.push_back((int)w_qo.size());
			// observations of this landmark whose path holds both edges: merge the two path-entry lists by observation
			const int a0 = w_lmep_off[r.mi], a1 = w_lmep_off[r.mi + 1], b0 = w_lmep_off[r.mj], b1 = w_lmep_off[r.mj + 1];
			int ia = a0, ib = b0;
			while (ia < a1 && ib < b1) {
				const int oa = p_obs[w_lmep_idx[ia]], ob = p_obs[w_lmep_idx[ib]];
				if (oa == ob) { w_qo.push_back(make_int2(w_lmep_idx[ia], w_lmep_idx[ib])); ++ia; ++ib; }
				else if (oa < ob) ++ia; else ++ib;
			}
		}
		w_qo_off.push_back((int)w_qo.size());
		} else w_qo_off.push_back(0);
		ivecs edge_lmes((size_t)E, &pool);
		for (int m = 0; m < n_lme; ++m) edge_lmes[w_lme_edge[m]].push_back(m);
		// chunks of <= 64 items of one block: H_pp over the q entries with co-observations, Schur over the q
		// entries of optimised landmarks (which sort first inside a block)
		std::vector<int> w_qp_q, w_pc_off((size_t)n_pairs + 1, 0), w_sc_off((size_t)n_pairs + 1, 0);
		std::vector<int3> w_pchunk, w_schunk;
		for (int p = 0; with_general && p < n_pairs; ++p) {
			w_pc_off[p] = (int)w_pchunk.size();
			const int first = (int)w_qp_q.size();
			for (int q = w_q_off[p]; q < w_q_off[p + 1]; ++q) if (w_qo_off[q + 1] > w_qo_off[q]) w_qp_q.push_back(q);
			for (int k = first; k < (int)w_qp_q.size(); k += 64) w_pchunk.push_back(make_int3(p, k, std::min(64, (int)w_qp_q.size() - k)));
			w_sc_off[p] = (int)w_schunk.size();
			for (int q = w_q_off[p]; q < w_q_uend[p]; q += 64) w_schunk.push_back(make_int3(p, q, std::min(64, w_q_uend[p] - q)));
		}
		w_pc_off[n_pairs] = (int)w_pchunk.size();
		w_sc_off[n_pairs] = (int)w_schunk.size();


		// ---- tiles of the LDS-tiled kernel (ssk_lm_tiled.hip, data model in ssk_lm.h) ----
		std::vector<lm_tile> w_tiles;
		std::vector<int4> w_tobs_a;
		std::vector<float4> w_tobs_px;
		std::vector<int4> w_tl_rec, w_inc_rec((size_t)n_lme, make_int4(0, 0, 0, 0));
		std::vector<int> w_lmep_loc(w_lmep_idx.size(), 0);
		std::vector<int> w_pit, w_pslot0, w_qit, w_sslot0, w_pcl_off((size_t)n_pairs + 1, 0), w_pcl_idx, w_scl_off((size_t)n_pairs + 1, 0), w_scl_idx;
		std::vector<int2> w_pgrp, w_sgrp, w_tl_pan;
		std::vector<int> w_wt_off, w_wt, w_tent, w_tsuf;
		std::vector<int2> w_mt;                                // (I | J << 8, partial id in generation order -> final id below)
		std::vector<int4> w_ilane;                             // incidence lanes, tile by tile
		auto inc_lanes = [](int cnt) { const int ns = (cnt + SSK_LMT_INC_CAP - 1) / SSK_LMT_INC_CAP; return ns < 1 ? 1 : (ns > 8 ? 8 : ns); };
		std::vector<std::vector<int>> stile_of;                // per upper 16x16 tile u = J (J + 1) / 2 + I: its partials, tile order
		const auto tp1 = tnow();
		// ---- path-suffix trie: suffix k of a path = (c_k .. c_{n-1}); a suffix is (first step, shorter suffix), 0 = the empty one ----
		std::vector<int2> w_suf(1, make_int2(-1, -1));
		std::vector<int> w_suf_len(1, 0), suf_at((size_t)n_path + 1, 0), obs_suf((size_t)Wn.n_obs, 0);
		{
			std::pmr::unordered_map<uint64_t, int> ids(&pool);      // (first step, shorter suffix) -> id
			ids.reserve(4096);
			for (int o = 0; o < Wn.n_obs; ++o) {
				int cur = 0;
				for (int p2 = Wn.obs_path_off[o + 1] - 1; p2 >= Wn.obs_path_off[o]; --p2) {
					const uint64_t key = ((uint64_t)(uint32_t)Wn.path_edge[p2] << 32) | (uint32_t)cur;
					auto it = ids.find(key);
					if (it == ids.end()) {
						it = ids.emplace(key, (int)w_suf.size()).first;
						w_suf.push_back(make_int2(Wn.path_edge[p2], cur));
						w_suf_len.push_back(w_suf_len[cur] + 1);
					}
					cur = it->second;
					suf_at[p2] = cur;
				}
				obs_suf[o] = cur;
			}
			// ids ascending by length, so that one level only needs the levels before it
			std::vector<int> perm(w_suf.size()), inv(w_suf.size());
			for (size_t i = 0; i < perm.size(); ++i) perm[i] = (int)i;
			std::stable_sort(perm.begin(), perm.end(), [&](int a, int b2) { return w_suf_len[a] < w_suf_len[b2]; });
			for (size_t i = 0; i < perm.size(); ++i) inv[perm[i]] = (int)i;
			std::vector<int2> ns(w_suf.size());
			std::vector<int> nl(w_suf.size());
			for (size_t i = 0; i < perm.size(); ++i) { ns[i] = make_int2(w_suf[perm[i]].x, w_suf[perm[i]].y < 0 ? -1 : inv[w_suf[perm[i]].y]); nl[i] = w_suf_len[perm[i]]; }
			w_suf.swap(ns); w_suf_len.swap(nl);
			for (int& v : suf_at) v = inv[v];
			for (int& v : obs_suf) v = inv[v];
		}
		const int n_suf = (int)w_suf.size();
		const auto tp2 = tnow();
		// few suffixes (batched windows): the whole table stays in LDS; many (a local area of tens of keyframes): per-tile copies
		const bool tab_lds = 96LL * n_suf <= 24 * 1024;
		int n_pslots = 0, n_sslots = 0;
		// LDS of a workgroup: static part, the window's edge poses twice, the solution vector and the pivots, the tile region
		const bool acc = want_acc && E <= SSK_LMA_MAX_E;
		const long long region_bytes = (long long)SSK_LMT_LDS_TOTAL - SSK_LMT_STATIC - 2LL * 12 * 8 * Wn.n_edges_total - 2LL * 8 * ((6 * E + 1) & ~1) -
		                               (acc ? 8LL * ssk_lma_acc_doubles(E) : 0) - (tab_lds ? 104LL * n_suf : 0);
		bool tiled = 6 * E <= SSK_LM_MAX_N6 && Wn.n_edges_total <= 128 && region_bytes >= 32 * 1024;
		if (want_acc && !(acc && tiled)) acc_all = false;
		if (tiled) {
			// Y part of a tile: sparse records (19 doubles per incidence) or, acc + MFMA, panels of [3][16] doubles per (optimised landmark,
			// 16-row tile of the reduced system it touches); acc: + 7 doubles per incidence for the reduced-gradient terms
			auto tile_bytes = [](long long nP, long long nO, long long nY, long long nL) {
				return 8 * (SSK_LMT_ENT * nP + SSK_LMT_OBS * nO + nY + SSK_LMT_LMR * nL);
			};
			const bool panels = schur_mfma != 0;
			{ const int TR = ssk_lma_row_tiles(E); stile_of.assign((size_t)(TR * (TR + 1) / 2), std::vector<int>()); }
			int n_stile = 0;
			ivecs obs_of((size_t)Wn.n_lms_total, &pool);
			for (int o = 0; o < Wn.n_obs; ++o) obs_of[Wn.obs_lm[o]].push_back(o);
			// packed observation record: everything a lane needs in one coalesced 32-byte load
			// (x: first local entry | optimised-step mask << 16 | path length << 24, y: landmark, z/w: 8 path codes of 8 bits)
			std::vector<char> in_tile((size_t)Wn.n_obs, 0);
			// (x: first tile-local entry | number of optimised entries << 16, y: landmark, z: suffix id of the whole path as the tile's
			// table knows it, w: its id in the window's table)
			auto pack_obs = [&](int o, int info) { return make_int4(info, Wn.obs_lm[o], obs_suf[o], obs_suf[o]); };
			std::vector<int> omask((size_t)Wn.n_obs, 0);           // bit i: step i of the observation's path is an optimised edge
			for (int o = 0; o < Wn.n_obs; ++o)
				for (int p = Wn.obs_path_off[o]; p < Wn.obs_path_off[o + 1]; ++p)
					if (Wn.edge_unknown_slot[Wn.path_edge[p] >> 1] >= 0) omask[o] |= 1 << (p - Wn.obs_path_off[o]);
			auto opt_mask = [&](int o) { return omask[o]; };
			// balanced cut: the variable footprint of the window spread evenly over the fewest tiles that fit
			std::vector<long long> fp_P(order.size()), fp_O(order.size()), fp_M(order.size()), fp_Y(order.size());
			// !tab_lds: the suffixes a landmark's observations use (whole paths + optimised entries), and which tile last took each suffix
			ivecs lm_sufs(tab_lds ? 0 : order.size(), &pool);
			std::vector<int> suf_tile(tab_lds ? 0 : (size_t)n_suf, -1), suf_local(tab_lds ? 0 : (size_t)n_suf, -1), suf_stamp(tab_lds ? 0 : (size_t)n_suf, -1);
			auto entry_suf = [&](int pb, int len, int i) {
				const int dir = Wn.path_edge[pb + i] & 1;
				return dir ? suf_at[i + 1 < len ? pb + i + 1 : n_path] : suf_at[pb + i];
			};
			std::vector<int> lm_rowmask(order.size(), 0);
			long long tot = 0;
			for (size_t oi = 0; oi < order.size(); ++oi) {
				long long nP = 0;
				for (int o : obs_of[order[oi]]) nP += __builtin_popcount(opt_mask(o));
				fp_P[oi] = nP; fp_O[oi] = (long long)obs_of[order[oi]].size(); fp_M[oi] = (long long)lm_lmes[oi].size();
				if ((int)oi < L)                                   // rows 6e .. 6e+5 of every incident edge e, as a mask of 16-row tiles
					for (int m : lm_lmes[oi]) lm_rowmask[oi] |= (1 << ((6 * w_lme_edge[m]) >> 4)) | (1 << ((6 * w_lme_edge[m] + 5) >> 4));
				fp_Y[oi] = panels ? (long long)SSK_LMA_PANEL * __builtin_popcount(lm_rowmask[oi]) : (long long)SSK_LMT_Y * fp_M[oi];
				if (acc) fp_Y[oi] += 7 * fp_M[oi];
				if (!tab_lds) {
					ivec& v = lm_sufs[oi];                          // each suffix once (stamped), ascending
					auto take = [&](int sid) { if (suf_stamp[(size_t)sid] != (int)oi) { suf_stamp[(size_t)sid] = (int)oi; v.push_back(sid); } };
					for (int o : obs_of[order[oi]]) {
						take(obs_suf[o]);
						const int pb = Wn.obs_path_off[o], len = Wn.obs_path_off[o + 1] - pb, om = opt_mask(o);
						for (int i = 0; i < len; ++i) if ((om >> i) & 1) take(entry_suf(pb, len, i));
					}
					std::sort(v.begin(), v.end());
				}
				tot += 8 * (SSK_LMT_ENT * fp_P[oi] + SSK_LMT_OBS * fp_O[oi] + fp_Y[oi] + SSK_LMT_LMR);
			}
			const auto ts_pre = tnow(); t_sub[0] += tms(tp2, ts_pre);
			const long long est_tiles = std::max<long long>(1, (tot + region_bytes * 9 / 10 - 1) / (region_bytes * 9 / 10));
			const long long target = (tot + est_tiles - 1) / est_tiles;
			std::vector<int> path_loc((size_t)n_path, -1);
			ivecs pslots_of((size_t)n_pairs, &pool), sslots_of((size_t)n_pairs, &pool);
			struct item { int pid, packed; };
			// lane groups of <= R consecutive items of one block; n_groups(R) for a sorted item list
			auto count_groups = [](const std::vector<item>& items, int R) {
				int n = (int)items.size(), g = 0;
				for (int i = 0; i < n;) { int j = i; while (j < n && items[j].pid == items[i].pid) ++j; g += (j - i + R - 1) / R; i = j; }
				return g;
			};
			auto make_groups = [&](std::vector<item>& items, int R, std::vector<int>& it_out, std::vector<int2>& grp_out, std::vector<int>& slot0_out,
			                       ivecs& slots_of, int& slot_counter, int& n_grp, int& n_chunk) {
				const int n = (int)items.size(), base = (int)it_out.size();
				std::vector<int> gkeys;
				for (int i = 0; i < n;) {
					int j = i;
					while (j < n && items[j].pid == items[i].pid) ++j;
					for (int k = i; k < j; k += R) { grp_out.push_back(make_int2(base + k, std::min(R, j - k) | (items[i].pid << 8))); gkeys.push_back(items[i].pid); }
					i = j;
				}
				for (const item& x : items) it_out.push_back(x.packed);
				n_grp = (int)gkeys.size(); n_chunk = (n_grp + 63) / 64;
				for (int c = 0; c < n_chunk; ++c) {
					slot0_out.push_back(slot_counter);
					int prev = -1;
					for (int lane = 0; lane < 64; ++lane) {
						const int key = gkeys[std::min(64 * c + lane, n_grp - 1)];
						if ((lane & 15) == 0 || key != prev) slots_of[key].push_back(slot_counter++);
						prev = key;
					}
				}
			};
			// the items of a tile by block pair id, stable: a counting sort (a tile has a few hundred items over <= E (E + 1) / 2 ids)
			std::vector<item> pitems, qitems, sort_tmp;
			std::vector<int> sort_cnt((size_t)n_pairs + 1);
			auto sort_by_pid = [&](std::vector<item>& items) {
				if (items.size() < 2) return;
				std::fill(sort_cnt.begin(), sort_cnt.end(), 0);
				for (const item& x : items) ++sort_cnt[(size_t)x.pid + 1];
				for (int i = 0; i < n_pairs; ++i) sort_cnt[(size_t)i + 1] += sort_cnt[(size_t)i];
				sort_tmp.resize(items.size());
				for (const item& x : items) sort_tmp[(size_t)sort_cnt[(size_t)x.pid]++] = x;
				items.swap(sort_tmp);
			};
			std::vector<std::pair<int, int>> steps;
			size_t oi = 0;
			int lme_cursor = 0;
			while (oi < order.size() && tiled) {
				const auto ts_a = tnow();
				lm_tile T{};
				T.lml0 = (int)oi; T.obs0 = (int)w_tobs_a.size(); T.inc0 = lme_cursor; T.ent0 = (int)w_tent.size(); T.il0 = (int)w_ilane.size();
				long long nP = 0, nO = 0, nM = 0, nL = 0, nY = 0, var = 0, nS = 0, nIL = 0;
				const int tile_no = (int)w_tiles.size();
				size_t oj = oi;
				for (; oj < order.size(); ++oj) {
					long long aS = nS;
					if (!tab_lds) for (int sid : lm_sufs[oj]) if (suf_tile[sid] != tile_no) aS += 12;       // the tile's table copy grows with every new suffix
					const long long aP = nP + fp_P[oj], aO = nO + fp_O[oj], aM = nM + fp_M[oj], aL = nL + 1, aY = nY + fp_Y[oj] + (aS - nS);
					long long aIL = nIL;                           // incidence lanes: the lanes of one incidence stay inside one wavefront
					for (int m : lm_lmes[oj]) {
						const int ns = inc_lanes(w_lmep_off[m + 1] - w_lmep_off[m]);
						if ((aIL & 15) + ns > 16) aIL = (aIL + 15) & ~15LL;      // (the lanes of an incidence inside one 16-lane row: DPP combine)
						aIL += ns;
					}
					// one round of lanes per phase: observations, landmarks, incidences each within the workgroup size
					if (tile_bytes(aP, aO, aY, aL) > region_bytes || aIL > SSK_LMT_NT || aL > SSK_LMT_NT || aO > SSK_LMT_NT || aP > 65535 || (panels && aL > 128)) break;
					nP = aP; nO = aO; nM = aM; nL = aL; nY = aY; nS = aS; nIL = aIL;
					if (!tab_lds) for (int sid : lm_sufs[oj]) suf_tile[sid] = tile_no;
					var += 8 * (SSK_LMT_ENT * fp_P[oj] + SSK_LMT_OBS * fp_O[oj] + fp_Y[oj] + SSK_LMT_LMR);
					if (var >= target) { ++oj; break; }
				}
				if (oj == oi) { tiled = false; break; }          // one landmark alone does not fit
				const auto ts_b = tnow(); t_sub[1] += tms(ts_a, ts_b);
				// !tab_lds: the tile's suffix list, tile-local id = position
				T.tsuf0 = (int)w_tsuf.size(); T.n_tsuf = 0;
				if (!tab_lds)
					for (size_t ok = oi; ok < oj; ++ok)
						for (int sid : lm_sufs[ok])
							if (suf_local[sid] < T.tsuf0) { suf_local[sid] = (int)w_tsuf.size(); w_tsuf.push_back(sid); }
				T.n_tsuf = (int)w_tsuf.size() - T.tsuf0;
				auto local_suf = [&](int sid) { return tab_lds ? sid : suf_local[sid] - T.tsuf0; };
				pitems.clear(); qitems.clear();
				int n_ent = 0, n_pan = 0;
				for (size_t ok = oi; ok < oj; ++ok) {
					const int l = order[ok], j = (int)(ok - oi);
					w_tl_pan.push_back(make_int2(lm_rowmask[ok], n_pan));
					n_pan += __builtin_popcount(lm_rowmask[ok]);
					w_tl_rec.push_back(make_int4(Wn.lm_unknown_slot[l], (int)w_tobs_a.size() - T.obs0, (int)(w_tobs_a.size() + obs_of[l].size()) - T.obs0, 0));
					for (int o : obs_of[l]) {
						const int kl = (int)w_tobs_a.size() - T.obs0, pb = Wn.obs_path_off[o], len = Wn.obs_path_off[o + 1] - pb, om = opt_mask(o);
						w_tobs_a.push_back(make_int4(n_ent | (__builtin_popcount(om) << 16), Wn.obs_lm[o], local_suf(obs_suf[o]), obs_suf[o]));
						w_tobs_px.push_back(make_float4(Wn.obs_px[4 * o], Wn.obs_px[4 * o + 1], Wn.obs_px[4 * o + 2], Wn.obs_px[4 * o + 3]));
						in_tile[o] = 1;
						steps.clear();                               // (edge slot, local entry)
						for (int i = 0; i < len; ++i)
							if ((om >> i) & 1) {
								const int le = n_ent + __builtin_popcount(om & ((1 << i) - 1));
								// A = +Jf R^T(suffix i) and q = T(suffix i) f for a forward step, A = -Jf R^T(suffix i+1), q = T(suffix i+1) f for an inverse one
								w_tent.push_back((local_suf(entry_suf(pb, len, i)) << 1) | (Wn.path_edge[pb + i] & 1));
								path_loc[pb + i] = le | (kl << 16);
								steps.push_back(std::make_pair(Wn.edge_unknown_slot[Wn.path_edge[pb + i] >> 1], le));
							}
						n_ent += __builtin_popcount(om);
						for (size_t a = 0; a < steps.size(); ++a)
							for (size_t b2 = a; b2 < steps.size(); ++b2) {
								const bool sw = steps[a].first > steps[b2].first;
								const auto& lo = sw ? steps[b2] : steps[a];
								const auto& hi = sw ? steps[a] : steps[b2];
								pitems.push_back({pid_of(lo.first, hi.first), lo.second | (hi.second << 16)});
							}
					}
					for (int m : lm_lmes[ok]) {
						for (int k = w_lmep_off[m]; k < w_lmep_off[m + 1]; ++k) w_lmep_loc[k] = path_loc[w_lmep_idx[k]];
						const int cnt = w_lmep_off[m + 1] - w_lmep_off[m];
						w_inc_rec[m] = make_int4(j | ((Wn.lm_unknown_slot[l] >= 0 ? 1 : 0) << 15) | (cnt << 16), w_lmep_off[m], 0 /* edge-major slot: below */,
						                         cnt > 0 ? w_lmep_loc[w_lmep_off[m]] : 0);
						// the incidence's lanes: its path entries dealt out SSK_LMT_INC_CAP at a time (a tile phase lasts as long as its longest
						// lane: 84 % of the incidences have one entry, the tail has ten and more), all lanes of one incidence in one 16-lane row of a wavefront
						const int ns = inc_lanes(cnt), per = (cnt + ns - 1) / ns;
						int pos = (int)w_ilane.size() - T.il0;
						if ((pos & 15) + ns > 16)
							for (; pos & 15; ++pos) w_ilane.push_back(make_int4(0, 0, 0, 0));       // (padding: lanes of no incidence)
						for (int sub = 0; sub < ns; ++sub) {
							const int k0 = std::min(cnt, sub * per), k1 = std::min(cnt, k0 + per);
							w_ilane.push_back(make_int4((m - T.inc0) | (sub << 16) | (ns << 20), w_lmep_off[m] + k0, k1 - k0, k1 > k0 ? w_lmep_loc[w_lmep_off[m] + k0] : 0));
						}
					}
					if (Wn.lm_unknown_slot[l] >= 0 && !panels)          // (the matrix-core form has no Schur items: 50 k of them in a 40-edge local area)
						for (size_t a = 0; a < lm_lmes[ok].size(); ++a)
							for (size_t b2 = a; b2 < lm_lmes[ok].size(); ++b2)
								qitems.push_back({pid_of(w_lme_edge[lm_lmes[ok][a]], w_lme_edge[lm_lmes[ok][b2]]),
								                  (lm_lmes[ok][a] - T.inc0) | ((lm_lmes[ok][b2] - T.inc0) << 16)});
					lme_cursor += (int)lm_lmes[ok].size();
				}
				const auto ts_c = tnow(); t_sub[2] += tms(ts_b, ts_c);
				T.n_lm = (int)(oj - oi); T.n_obs = (int)w_tobs_a.size() - T.obs0; T.n_ent = n_ent; T.n_inc = lme_cursor - T.inc0; T.n_pan = panels ? n_pan : 0;
				T.n_il = (int)w_ilane.size() - T.il0;
				T.slot0 = 0; T.n_slot = 0;                        // the tile's optimised landmarks: a run of slots (they lead the tile order, in slot order)
				for (size_t ok = oi; ok < oj; ++ok) { const int sl = Wn.lm_unknown_slot[order[ok]]; if (sl >= 0) { if (!T.n_slot) T.slot0 = sl; ++T.n_slot; } }
				T.pg0 = (int)w_pgrp.size(); T.pc0 = (int)w_pslot0.size();
				// one chunk task per wave: the smallest R whose H_pp + Schur chunks fit one round of the workgroup's waves
				sort_by_pid(pitems);
				sort_by_pid(qitems);
				if (!acc && panels) {
				// slot kernels, matrix-core Schur: H_pp chunks on half the waves, beside them one task per 16x16 tile of the reduced system
				// that the tile's landmarks touch, most landmarks first; a task leaves a 256-double partial in HBM
				int R = 1;
				while (R < 255 && (count_groups(pitems, R) + 63) / 64 > SSK_LMT_NT / 128) ++R;
				make_groups(pitems, R, w_pit, w_pgrp, w_pslot0, pslots_of, n_pslots, T.n_pg, T.n_pc);
				T.sg0 = (int)w_sgrp.size(); T.sc0 = (int)w_sslot0.size(); T.n_sg = 0; T.n_sc = 0;
				const int TR = ssk_lma_row_tiles(E);
				std::vector<std::pair<int, int>> tk;               // (-cover, code)
				{
					// landmarks per (I, J): a landmark touches 3-4 row tiles, so walk its bit pairs instead of all T (T + 1) / 2 tiles per landmark
					std::vector<int> cover((size_t)(TR * (TR + 1) / 2), 0);
					for (size_t ok = oi; ok < oj; ++ok) {
						const int mk = lm_rowmask[ok];
						for (int mj = mk; mj; mj &= mj - 1) {
							const int J = __builtin_ctz(mj);
							for (int mi = mk & ((2 << J) - 1); mi; mi &= mi - 1) cover[(size_t)(J * (J + 1) / 2 + __builtin_ctz(mi))]++;
						}
					}
					for (int J = 0; J < TR; ++J)
						for (int I = 0; I <= J; ++I)
							if (cover[(size_t)(J * (J + 1) / 2 + I)]) tk.push_back(std::make_pair(-cover[(size_t)(J * (J + 1) / 2 + I)], I | (J << 8)));
				}
				std::stable_sort(tk.begin(), tk.end());
				T.mt0 = (int)w_mt.size(); T.n_mt = (int)tk.size();
				for (const auto& t : tk) {
					const int I = t.second & 255, J = t.second >> 8;
					stile_of[(size_t)(J * (J + 1) / 2 + I)].push_back(n_stile);
					w_mt.push_back(make_int2(t.second, n_stile++));
				}
				} else if (!acc) {
				// the kernel gives the landmark phase the last ceil(n_lm / 64) wavefronts and the incidence phase the first ceil(n_il / 64); the
				// wavefronts between them (`idle`) take one H_pp chunk task each meanwhile, and after the barrier the H_pp tasks left and the
				// Schur tasks share all NW wavefronts, one task each (two rounds of smaller tasks measured 5 % slower).  Lane-group sizes (Rp, Rs):
				// the pair whose longest task is shortest under that constraint (a task costs ~250 instructions per H_pp item, ~155 per
				// Schur item, ~650 for the segmented scan)
				const int NWt = SSK_LMT_NT / 64;
				const int idle = std::max(NWt - (T.n_lm + 63) / 64 - (T.n_il + 63) / 64, 0);
				int Rp = 1, Rs = 1;
				{
					long long best = -1;
					for (int rp = 1; rp <= 48; ++rp) {
						const int cp = (count_groups(pitems, rp) + 63) / 64, room = NWt - std::max(cp - idle, 0);
						if (room < 1 && !qitems.empty()) continue;
						if (room < 0) continue;
						int rs = 1;
						while (rs < 255 && !qitems.empty() && (count_groups(qitems, rs) + 63) / 64 > room) ++rs;
						const long long cost = std::max(pitems.empty() ? 0LL : 250LL * rp + 650, qitems.empty() ? 0LL : 155LL * rs + 650);
						if (best < 0 || cost < best) { best = cost; Rp = rp; Rs = rs; }
					}
					if (best < 0) {                                  // (never with the tile limits: fall back to one common size)
						while (Rp < 255 && (count_groups(pitems, Rp) + 63) / 64 + (count_groups(qitems, Rp) + 63) / 64 > NWt) ++Rp;
						Rs = Rp;
					}
				}
				if (w == 0 && ctx->env.lm_debug) std::fprintf(stderr, "[ssk lm]   tile %d: H_pp items %zu (<= %d per lane), Schur items %zu (<= %d per lane), %d idle wavefronts\n", tile_no, pitems.size(), Rp, qitems.size(), Rs, idle);
				make_groups(pitems, Rp, w_pit, w_pgrp, w_pslot0, pslots_of, n_pslots, T.n_pg, T.n_pc);
				T.sg0 = (int)w_sgrp.size(); T.sc0 = (int)w_sslot0.size();
				make_groups(qitems, Rs, w_qit, w_sgrp, w_sslot0, sslots_of, n_sslots, T.n_sg, T.n_sc);
				} else {
				// ---- on-chip accumulation: the last lane of a block's run adds the run's sum to the block's LDS accumulator, so a block
				// must belong to ONE wave of the tile: chunks (one per wave pass) are cut at block boundaries and padded with empty groups,
				// and R is at least large enough for the largest block to fit one chunk ----
				const int NWv = SSK_LMT_NT / 64;
				auto chunks_acc = [](const std::vector<item>& items, int R) {
					int n = (int)items.size(), fill = 0, chunks = n ? 1 : 0;
					for (int i = 0; i < n;) {
						int j = i; while (j < n && items[j].pid == items[i].pid) ++j;
						const int g = (j - i + R - 1) / R;
						if (g > 64) return 1 << 20;
						if (fill + g > 64) { ++chunks; fill = 0; }
						fill += g; i = j;
					}
					return chunks;
				};
				auto make_groups_acc = [&](std::vector<item>& items, int R, std::vector<int>& it_out, std::vector<int2>& grp_out, int& n_grp, int& n_chunk) {
					const int n = (int)items.size(), base = (int)it_out.size(), g0 = (int)grp_out.size();
					int fill = 0;
					for (int i = 0; i < n;) {
						int j = i; while (j < n && items[j].pid == items[i].pid) ++j;
						const int g = (j - i + R - 1) / R;
						if (fill + g > 64) {                         // pad the chunk: empty groups that extend the previous block's run with zeros
							const int2 last = grp_out.back();
							for (; fill < 64; ++fill) grp_out.push_back(make_int2(last.x, last.y & ~255));
							fill = 0;
						}
						for (int k = i; k < j; k += R) grp_out.push_back(make_int2(base + k, std::min(R, j - k) | (items[i].pid << 8)));
						fill += g; i = j;
					}
					for (const item& x : items) it_out.push_back(x.packed);
					n_grp = (int)grp_out.size() - g0; n_chunk = (n_grp + 63) / 64;
				};
				int R = 1;
				if (panels) { while (R < 255 && chunks_acc(pitems, R) > NWv / 2) ++R; }
				else { while (R < 255 && chunks_acc(pitems, R) + chunks_acc(qitems, R) > NWv) ++R; }
				make_groups_acc(pitems, R, w_pit, w_pgrp, T.n_pg, T.n_pc);
				T.sg0 = (int)w_sgrp.size(); T.sc0 = 0; T.n_sg = 0; T.n_sc = 0;
				if (!panels) make_groups_acc(qitems, R, w_qit, w_sgrp, T.n_sg, T.n_sc);
				// tasks of the tile's chunk phase, longest first onto the least loaded wave: H_pp chunk c = (0 << 28) | c, sparse Schur chunk
				// c = (1 << 28) | c, matrix-core Schur tile (I, J) = (2 << 28) | (J << 8) | I
				struct task { long long cost; int code; };
				std::vector<task> tasks;
				for (int c = 0; c < T.n_pc; ++c) tasks.push_back({700 + 380LL * R, c});
				for (int c = 0; c < T.n_sc; ++c) tasks.push_back({700 + 420LL * R, (1 << 28) | c});
				if (panels) {
					const int TR = ssk_lma_row_tiles(E);
					for (int J = 0; J < TR; ++J)
						for (int I = 0; I <= J; ++I) {
							int cover = 0;
							for (size_t ok = oi; ok < oj; ++ok) cover += (lm_rowmask[ok] >> I) & (lm_rowmask[ok] >> J) & 1;
							if (cover) tasks.push_back({250 + 80LL * cover, (2 << 28) | (J << 8) | I});
						}
				}
				std::stable_sort(tasks.begin(), tasks.end(), [](const task& a, const task& b) { return a.cost > b.cost; });
				std::vector<std::vector<int>> per_wave((size_t)NWv);
				std::vector<long long> load((size_t)NWv, 0);
				for (const task& t : tasks) {
					int best = 0;
					for (int wv = 1; wv < NWv; ++wv) if (load[wv] < load[best]) best = wv;
					per_wave[best].push_back(t.code); load[best] += t.cost;
				}
				T.wt0 = (int)w_wt_off.size();
				for (int wv = 0; wv < NWv; ++wv) { w_wt_off.push_back((int)w_wt.size()); w_wt.insert(w_wt.end(), per_wave[wv].begin(), per_wave[wv].end()); }
				w_wt_off.push_back((int)w_wt.size());
				}
				w_tiles.push_back(T);
				tiled_dyn = std::max(tiled_dyn, (size_t)(2 * 12 * 8 * Wn.n_edges_total) + (size_t)(2 * 8 * ((6 * E + 1) & ~1)) + (acc ? 8 * (size_t)ssk_lma_acc_doubles(E) : 0) + (tab_lds ? 104 * (size_t)n_suf : 0) +
				                                (size_t)tile_bytes(T.n_ent, T.n_obs, nY, T.n_lm));
				oi = oj;
			}
			// the observations outside every tile (fixed landmark, no optimised edge on the path) still count in the error
			for (int o = 0; o < Wn.n_obs && tiled; ++o)
				if (!in_tile[o]) {
					w_tobs_a.push_back(pack_obs(o, 0));
					w_tobs_px.push_back(make_float4(Wn.obs_px[4 * o], Wn.obs_px[4 * o + 1], Wn.obs_px[4 * o + 2], Wn.obs_px[4 * o + 3]));
				}
			// final slot ids are block-major, so a block's partial sums are one contiguous run [pcl_off[pid], pcl_off[pid+1]);
			// pcl_idx / scl_idx map a chunk segment (generation order) to its slot
			w_pcl_idx.assign((size_t)n_pslots, 0); w_scl_idx.assign((size_t)n_sslots, 0);
			int np_ = 0, ns_ = 0;
			for (int p2 = 0; p2 < n_pairs; ++p2) {
				w_pcl_off[p2] = np_; for (int sl : pslots_of[p2]) w_pcl_idx[sl] = np_++;
				w_scl_off[p2] = ns_; for (int sl : sslots_of[p2]) w_scl_idx[sl] = ns_++;
			}
			w_pcl_off[n_pairs] = np_; w_scl_off[n_pairs] = ns_;
			// matrix-core partials: final ids tile-major (u), so that the sum of tile u runs over st_off[u] .. st_off[u + 1]
			{
				std::vector<int> fin((size_t)n_stile, 0);
				int nf = 0;
				for (size_t u = 0; u < stile_of.size(); ++u) { st_off.push_back(nf); for (int gid : stile_of[u]) fin[gid] = nf++; }
				st_off.push_back(nf);
				for (int2& m : w_mt) m.y = fin[m.y];
			}
			// the reduced system goes to LDS too when it fits beside poses / solution / pivots (else: L2 scratch, Hred)
			const size_t fixed = (size_t)(2 * 12 * 8 * Wn.n_edges_total) + (size_t)(2 * 8 * ((6 * E + 1) & ~1)) + (tab_lds ? 104 * (size_t)n_suf : 0);
			const size_t mat = (size_t)(6 * E + 1) * ((6 * E) | 1) * 8;
			if (fixed + mat <= (size_t)(SSK_LMT_LDS_TOTAL - SSK_LMT_STATIC)) tiled_dyn = std::max(tiled_dyn, fixed + mat);
			else lds_sys_all = false;
		}
		if (!tiled) tiled_all = false;
		const auto tp3 = tnow();
		if (w == 0 && ctx->env.lm_debug) {
			std::fprintf(stderr, "[ssk lm] window 0: E %d L %d obs %d path %d lme %d tiles %d (region %lld B) pslots %d sslots %d\n", E, L, Wn.n_obs, n_path, n_lme,
			             tiled ? (int)w_tiles.size() : -1, region_bytes, n_pslots, n_sslots);
			for (const lm_tile& T : w_tiles)
				std::fprintf(stderr, "[ssk lm]   tile: lms %d obs %d ent %d inc %d (lanes %d) | pgroups %d (chunks %d) sgroups %d (chunks %d) matrix-core tasks %d\n", T.n_lm, T.n_obs, T.n_ent, T.n_inc, T.n_il, T.n_pg, T.n_pc, T.n_sg, T.n_sc, T.n_mt);
		}
		H.n_suf = n_suf; H.suf0 = (int)suf_rec.size(); H.tent0 = (int)tent.size();
		H.tab_lds = tab_lds ? 1 : 0; H.tsuf0 = (int)tsuf.size(); H.gtab0 = gtab_total;
		H.il0 = (int)ilane.size();
		ilane.insert(ilane.end(), w_ilane.begin(), w_ilane.end());
		H.schur_mfma = schur_mfma; H.mt0 = (int)mt.size(); H.st0 = (int)n_stile_total; H.pad_mt = 0;
		H.sto0 = (int)st_off.size() - (tiled ? (int)stile_of.size() + 1 : 0);
		mt.insert(mt.end(), w_mt.begin(), w_mt.end());
		for (const auto& v : stile_of) n_stile_total += (long long)v.size();
		tsuf.insert(tsuf.end(), w_tsuf.begin(), w_tsuf.end());
		// a table per workgroup that may share the window (cooperative kernel: up to 64)
		if (!tab_lds) gtab_total += 12LL * n_suf * (n_windows * 4 <= ctx->n_cu ? 64 : 1);
		for (int k = 0; k <= SSK_LM_MAX_PATH + 1; ++k) {       // suf_lvl[k] = first id of length >= k
			int first = n_suf;
			for (int i = 0; i < n_suf; ++i) if (w_suf_len[i] >= k) { first = i; break; }
			H.suf_lvl[k] = first;
		}
		suf_rec.insert(suf_rec.end(), w_suf.begin(), w_suf.end());
		tent.insert(tent.end(), w_tent.begin(), w_tent.end());
		H.acc = acc && tiled ? 1 : 0;
		H.wt0 = (int)wt.size(); H.wto0 = (int)wt_off.size(); H.pad_acc = 0;
		wt.insert(wt.end(), w_wt.begin(), w_wt.end()); wt_off.insert(wt_off.end(), w_wt_off.begin(), w_wt_off.end());
		tl_pan.insert(tl_pan.end(), w_tl_pan.begin(), w_tl_pan.end());
		H.n_tiles = tiled ? (int)w_tiles.size() : -1;
		H.tile0 = (int)tiles.size(); H.tl0 = (int)tl_rec.size(); H.tobs0 = (int)tobs_a.size();
		H.pit0 = (int)pit.size(); H.pg0 = (int)pgrp.size(); H.pcx0 = (int)pslot0.size(); H.ps0 = (int)n_pslots_total; H.pcl0 = (int)pcl_idx.size();
		H.qit0 = (int)qit.size(); H.sg0 = (int)sgrp.size(); H.scx0 = (int)sslot0.size(); H.ss0 = (int)n_sslots_total; H.scl0 = (int)scl_idx.size();
		n_pslots_total += (size_t)n_pslots; n_sslots_total += (size_t)n_sslots;
		tiles.insert(tiles.end(), w_tiles.begin(), w_tiles.end());
		tl_rec.insert(tl_rec.end(), w_tl_rec.begin(), w_tl_rec.end());
		tobs_a.insert(tobs_a.end(), w_tobs_a.begin(), w_tobs_a.end());
		tobs_px.insert(tobs_px.end(), w_tobs_px.begin(), w_tobs_px.end());
		lmep_loc.insert(lmep_loc.end(), w_lmep_loc.begin(), w_lmep_loc.end());
		pit.insert(pit.end(), w_pit.begin(), w_pit.end()); pgrp.insert(pgrp.end(), w_pgrp.begin(), w_pgrp.end());
		pslot0.insert(pslot0.end(), w_pslot0.begin(), w_pslot0.end());
		pcl_off.insert(pcl_off.end(), w_pcl_off.begin(), w_pcl_off.end()); pcl_idx.insert(pcl_idx.end(), w_pcl_idx.begin(), w_pcl_idx.end());
		qit.insert(qit.end(), w_qit.begin(), w_qit.end()); sgrp.insert(sgrp.end(), w_sgrp.begin(), w_sgrp.end());
		sslot0.insert(sslot0.end(), w_sslot0.begin(), w_sslot0.end());
		scl_off.insert(scl_off.end(), w_scl_off.begin(), w_scl_off.end()); scl_idx.insert(scl_idx.end(), w_scl_idx.begin(), w_scl_idx.end());

		H.n_edges = Wn.n_edges_total; H.E = E; H.n_lms = Wn.n_lms_total; H.L = L; H.n_obs = Wn.n_obs; H.n_path = n_path;
		H.n_pairs = n_pairs; H.n_lme = n_lme; H.n_lme_unknown = n_lme_unknown; H.n_q = n_q;
		H.edge0 = (int)edge_slot.size(); H.lm0 = (int)lm_slot.size(); H.obs0 = (int)obs_lm.size();
		H.opo0 = (int)obs_path_off.size(); H.path0 = (int)path_edge.size();
		H.us0 = (int)slot_edge.size(); H.uso0 = (int)el_off.size(); H.ls0 = (int)slot_lm.size(); H.lso0 = (int)lmo_off.size();
		H.lmo0 = (int)lmo_idx.size(); H.lme0 = (int)lme_edge.size(); H.lmeo0 = (int)lmep_off.size(); H.lmep0 = (int)lmep_idx.size();
		H.pair0 = (int)pair_ee.size(); H.pairo0 = (int)q_off.size(); H.q0 = (int)q_mm.size(); H.qo0 = (int)qo_off.size();
		H.qe0 = (int)qo.size(); H.el0 = (int)el_idx.size();
		H.qp0 = (int)qp_q.size(); H.pc0 = (int)pchunk.size(); H.sc0 = (int)schunk.size();
		H.n_pchunk = (int)w_pchunk.size(); H.n_schunk = (int)w_schunk.size();
		H.trace0 = w * MAX_ITERS_ALLOC; H.h0 = h_total; H.dp_trace0 = dpt; H.df_trace0 = dft;
		h_total += (long long)(6 * E + 1) * (6 * E + 2);      // Hpair (18 E (E+1)) / the reduced system with its border row
		if (trace) { dpt += (long long)MAX_ITERS_ALLOC * 6 * E; dft += (long long)MAX_ITERS_ALLOC * 3 * L; }
		max_n6 = std::max(max_n6, 6 * E);

		edge_slot.insert(edge_slot.end(), Wn.edge_unknown_slot, Wn.edge_unknown_slot + Wn.n_edges_total);
		pose.insert(pose.end(), Wn.edge_inv_pose, Wn.edge_inv_pose + 12 * (size_t)Wn.n_edges_total);
		lm_slot.insert(lm_slot.end(), Wn.lm_unknown_slot, Wn.lm_unknown_slot + Wn.n_lms_total);
		lm.insert(lm.end(), Wn.lm_pos, Wn.lm_pos + 3 * (size_t)Wn.n_lms_total);
		obs_lm.insert(obs_lm.end(), Wn.obs_lm, Wn.obs_lm + Wn.n_obs);
		obs_px.insert(obs_px.end(), Wn.obs_px, Wn.obs_px + 4 * (size_t)Wn.n_obs);
		if (Wn.n_obs) obs_path_off.insert(obs_path_off.end(), Wn.obs_path_off, Wn.obs_path_off + Wn.n_obs + 1);
		else obs_path_off.push_back(0);
		if (n_path) path_edge.insert(path_edge.end(), Wn.path_edge, Wn.path_edge + n_path);
		path_obs.insert(path_obs.end(), p_obs.begin(), p_obs.end());
		slot_edge.insert(slot_edge.end(), s_edge.begin(), s_edge.end());
		slot_lm.insert(slot_lm.end(), s_lm.begin(), s_lm.end());
		int acc_o = 0;
		for (int l = 0; l < L; ++l) { lmo_off.push_back(acc_o); acc_o += (int)lm_obs[l].size(); lmo_idx.insert(lmo_idx.end(), lm_obs[l].begin(), lm_obs[l].end()); }
		lmo_off.push_back(acc_o);
		if (L > 0) lsm_off.insert(lsm_off.end(), w_lsm_off.begin(), w_lsm_off.end()); else lsm_off.push_back(0);
		lme_edge.insert(lme_edge.end(), w_lme_edge.begin(), w_lme_edge.end());
		lme_ls.insert(lme_ls.end(), w_lme_ls.begin(), w_lme_ls.end());
		lmep_off.insert(lmep_off.end(), w_lmep_off.begin(), w_lmep_off.end());
		lmep_idx.insert(lmep_idx.end(), w_lmep_idx.begin(), w_lmep_idx.end());
		for (int e2 = 0; e2 < E; ++e2) for (int e = 0; e <= e2; ++e) pair_ee.push_back(make_int2(e, e2));
		q_off.insert(q_off.end(), w_q_off.begin(), w_q_off.end());
		q_uend.insert(q_uend.end(), w_q_uend.begin(), w_q_uend.end());
		pc_off.insert(pc_off.end(), w_pc_off.begin(), w_pc_off.end());
		sc_off.insert(sc_off.end(), w_sc_off.begin(), w_sc_off.end());
		qp_q.insert(qp_q.end(), w_qp_q.begin(), w_qp_q.end());
		pchunk.insert(pchunk.end(), w_pchunk.begin(), w_pchunk.end());
		schunk.insert(schunk.end(), w_schunk.begin(), w_schunk.end());
		q_mm.insert(q_mm.end(), w_q_mm.begin(), w_q_mm.end());
		qo_off.insert(qo_off.end(), w_qo_off.begin(), w_qo_off.end());
		qo.insert(qo.end(), w_qo.begin(), w_qo.end());
		int ae = 0;
		std::vector<int> w_elpos((size_t)n_lme, 0);
		for (int e = 0; e < E; ++e) {
			el_off.push_back(ae);
			for (size_t k = 0; k < edge_lmes[e].size(); ++k) w_elpos[edge_lmes[e][k]] = ae + (int)k;
			ae += (int)edge_lmes[e].size();
			el_idx.insert(el_idx.end(), edge_lmes[e].begin(), edge_lmes[e].end());
		}
		el_off.push_back(ae);
		lme_elpos.insert(lme_elpos.end(), w_elpos.begin(), w_elpos.end());
		for (int m = 0; m < n_lme; ++m) w_inc_rec[m].z = w_elpos[m];
		inc_rec.insert(inc_rec.end(), w_inc_rec.begin(), w_inc_rec.end());
		t_ph[0] += tms(tp0, tp1); t_ph[1] += tms(tp1, tp2); t_ph[2] += tms(tp2, tp3); t_ph[3] += tms(tp3, tnow());
	}

	// ---- node form: the default for batches whose windows all fit it (SSK_LM_KERNEL=node|tiled|coop|l2|acc picks a kernel) ----
	nd_host NH;
	std::vector<int> node_order;
	bool node_all = !force_k || std::strcmp(force_k, "node") == 0;
	const size_t node_lds_limit = (size_t)SSK_LMT_LDS_TOTAL - SSK_LMT_STATIC;
	// (a window whose reduced system does not fit LDS: the BIG form -- accumulators and system in HBM / L2, ssk_lm_node.hip -- only on request
	//  (SSK_LM_KERNEL=node): measured 360 ms per 1024 depth-5 windows (E = 40) against 94 ms of the tiled kernel, 84 % of it the
	//  global_atomic_add_f64 stream of the pair phase -- 640 k adds per window and pass, the chip's L2 atomic rate)
	const bool node_big_default = false;
	bool node_big = false;
	for (int w = 0; w < n_windows && node_all; ++w) {
		const lm_win& Hw = hdr[w];
		if (!force_k && !node_big_default && 8 * nd_lds_doubles(Hw.n_edges, Hw.E, 4, 0) > node_lds_limit) { node_all = false; break; }      // (before the symbolic work)
		nd_win nw{};
		nw.edge0 = Hw.edge0; nw.lm0 = Hw.lm0; nw.us0 = Hw.us0; nw.ls0 = Hw.ls0; nw.trace0 = Hw.trace0; nw.dp_trace0 = Hw.dp_trace0; nw.df_trace0 = Hw.df_trace0;
		nw.h0 = Hw.h0;
		if (!nd_build_window(windows[w], 4, NH, nw, node_lds_limit / 8)) { node_all = false; break; }
		if (8 * NH.lds_doubles > node_lds_limit) node_big = true;
		if (node_big && (8 * NH.lds_doubles_big > node_lds_limit || (!force_k && !node_big_default))) node_all = false;
	}
	if (force_k && std::strcmp(force_k, "node") == 0 && !node_all)
		return ssk_set_error(ctx, SSK_ERR_CAPACITY, "ssk_lm_batch_create: SSK_LM_KERNEL=node, but a window of the batch does not fit the node kernel");
	if (want_acc && !acc_all) return lm_batch_create_impl(ctx, windows, n_windows, trace, out, with_general, false);
	if (!tiled_all && !with_general) return lm_batch_create_impl(ctx, windows, n_windows, trace, out, true, false);
	const auto t_sym = std::chrono::steady_clock::now();
	ssk_lm_batch* b = new ssk_lm_batch();
	b->symbolic_ms = std::chrono::duration<double, std::milli>(t_sym - t_begin).count();
	b->has_general = with_general;
	b->ctx = ctx; b->n_windows = n_windows; b->trace = trace; b->max_iters_alloc = MAX_ITERS_ALLOC; b->max_n6 = max_n6;
	b->hdr = hdr; b->n_pose = pose.size(); b->n_lm = lm.size();
	lm_dev& D = b->D;
	int rc = SSK_OK;
#define UP(vec, field) if (!rc) rc = upload(b, vec, &D.field)
	UP(hdr, win);
	UP(edge_slot, edge_slot); UP(lm_slot, lm_slot); UP(obs_lm, obs_lm); UP(obs_px, obs_px); UP(obs_path_off, obs_path_off);
	UP(path_edge, path_edge); UP(path_obs, path_obs); UP(slot_edge, slot_edge); UP(slot_lm, slot_lm);
	UP(lmo_off, lmo_off); UP(lmo_idx, lmo_idx); UP(lsm_off, lsm_off);
	UP(lme_edge, lme_edge); UP(lme_ls, lme_ls); UP(lme_elpos, lme_elpos); UP(lmep_off, lmep_off); UP(lmep_idx, lmep_idx);
	UP(pair_ee, pair_ee); UP(q_off, q_off); UP(q_uend, q_uend); UP(q_mm, q_mm); UP(qo_off, qo_off); UP(qo, qo);
	UP(el_off, el_off); UP(el_idx, el_idx);
	UP(qp_q, qp_q); UP(pchunk, pchunk); UP(pc_off, pc_off); UP(schunk, schunk); UP(sc_off, sc_off);
	UP(pose, pose0); UP(lm, lm0);
	UP(tiles, tile); UP(tl_rec, tl_rec); UP(tobs_a, tobs_a); UP(tobs_px, tobs_px);
	UP(inc_rec, inc_rec); UP(lmep_loc, lmep_loc);
	UP(pit, pit); UP(pgrp, pgrp); UP(pslot0, pslot0); UP(pcl_off, pcl_off); UP(pcl_idx, pcl_idx);
	UP(qit, qit); UP(sgrp, sgrp); UP(sslot0, sslot0); UP(scl_off, scl_off); UP(scl_idx, scl_idx);
	UP(wt_off, wt_off); UP(wt, wt); UP(tl_pan, tl_pan); UP(suf_rec, suf_rec); UP(tent, tent); UP(tsuf, tsuf); UP(mt, mt); UP(st_off, st_off); UP(ilane, ilane);
#undef UP
	b->gtab_multi = n_windows * 4 <= ctx->n_cu;
	b->node_ok = node_all;
	if (node_all) {
#define UPN(vec, field) if (!rc) rc = upload(b, NH.vec, &b->N.field)
		UPN(win, win); UPN(tile, tile); UPN(group, group); UPN(suf, suf); UPN(obs, obs); UPN(px, px); UPN(lane, lane);
		UPN(item, item); UPN(qs, qs); UPN(gitem, gitem); UPN(gqs, gqs);
		{
			// launch order: windows by observations x optimised edges, largest first (SSK_LM_ORDER=0: as given)
			node_order.resize((size_t)n_windows);
			std::vector<int>& ord = node_order;       // (lives until finalize() copies it)
			for (int w = 0; w < n_windows; ++w) ord[(size_t)w] = w;
			if (ctx->env.lm_order)
				std::stable_sort(ord.begin(), ord.end(), [&](int a, int c) { return (long long)NH.win[a].n_obs * (NH.win[a].E + 4) > (long long)NH.win[c].n_obs * (NH.win[c].E + 4); });
			if (!rc) rc = upload(b, ord, &b->N.order);
		}
#undef UPN
		b->node_big = node_big;
		b->node_dyn_bytes = 8 * (node_big ? NH.lds_doubles_big : NH.lds_doubles);
		b->node_small = !node_big;
		for (const nd_win& nw : NH.win) b->node_small = b->node_small && nw.E <= 10;
		if (ctx->env.lm_debug)
		{
			long long rounds = 0;
			for (int t = 0; t < NH.win[0].n_tiles; ++t) rounds += NH.tile[NH.win[0].tile0 + t].n_rounds;
			std::fprintf(stderr, "[ssk lm] node form%s: %d windows, window 0: %d tiles, %d groups, %d suffixes, %lld pair rounds, %zu items (all windows); dynamic LDS %zu B, longest item run of a lane %d\n",
			             node_big ? " (BIG)" : "", n_windows, NH.win[0].n_tiles, NH.win[0].n_groups, NH.win[0].n_suf, rounds, NH.item.size(), b->node_dyn_bytes, NH.max_items_per_lane);
		}
	}
	b->tiled_ok = tiled_all; b->tiled_dyn_bytes = tiled_dyn; b->lds_sys = tiled_all && lds_sys_all;
	b->acc_ok = want_acc && acc_all && tiled_all; b->schur_mfma = schur_mfma;
	D.store_hff = 0;
	// SSK_LM_LDL=mfma (needs the matrix-core Schur form): the kernels whose L2-resident reduced systems are factorised in 16-column panels
	// with the trailing update on the matrix cores (lm_solve_*_ldlm_kernel); default: the six-column VALU solver -- measured faster,
	// profiles/r02_lm_ldl_mfma_vs_valu.json
	{ D.ldl_mfma = ctx->env.lm_ldl_mfma && schur_mfma; if (D.ldl_mfma) b->schur_mfma = 2; }
	b->max_tiles = 0;
	for (const lm_win& hw : b->hdr) b->max_tiles = std::max(b->max_tiles, hw.n_tiles);
	if (!rc) rc = upload(b, pose, const_cast<const double**>(&D.pose));     // working copies of the state (pose0 / lm0 keep the initial one)
	if (!rc) rc = upload(b, lm, const_cast<const double**>(&D.lm));
	const size_t n_obs = obs_lm.size(), n_path = path_edge.size(), n_ls = slot_lm.size(), n_us = slot_edge.size();
	const size_t n_lme = lme_edge.size();
	if (!rc) rc = scratch(b, pose.size(), &D.pose_new);
	if (!rc) rc = scratch(b, lm.size(), &D.lm_new);
	if (!rc) rc = scratch(b, with_general ? 4 * n_obs : 0, &D.obs_r);
	if (!rc) rc = scratch(b, with_general ? 12 * n_obs : 0, &D.obs_Jf);
	if (!rc) rc = scratch(b, with_general ? 3 * n_path : 0, &D.path_q);
	if (!rc) rc = scratch(b, with_general ? 24 * n_path : 0, &D.path_Je);
	if (!rc) rc = scratch(b, 9 * n_ls, &D.Hff);     // (the tiled kernel stores the landmark blocks here for the shared-landmark exchange)
	if (!rc) rc = scratch(b, 9 * n_ls, &D.Vinv);
	if (!rc) rc = scratch(b, 3 * n_ls, &D.gf);
	if (!rc) rc = scratch(b, 3 * n_ls, &D.df);
	if (!rc) rc = scratch(b, with_general ? 18 * n_lme : 0, &D.Wb);
	if (!rc) rc = scratch(b, 18 * n_lme, &D.Tb);
	if (!rc) rc = scratch(b, 6 * n_lme, &D.gpe);
	if (!rc) rc = scratch(b, 6 * n_lme, &D.gre);
	if (!rc) rc = scratch(b, 36 * pchunk.size(), &D.Ppart);
	if (!rc) rc = scratch(b, 36 * schunk.size(), &D.Spart);
	// (+ one record: the assembly's clamped loads of an EMPTY run read the slot at the run's start, which for the last run is one past the end)
	if (!rc) rc = scratch(b, 36 * n_pslots_total + 36, &D.Ppart2);
	if (!rc) rc = scratch(b, 36 * n_sslots_total + 36, &D.Spart2);
	if (!rc) rc = scratch(b, with_general ? (size_t)h_total : 0, &D.Hpair);
	if (!rc) rc = scratch(b, (size_t)h_total, &D.Hred);
	if (!rc) rc = scratch(b, 6 * n_us, &D.dp);
	if (!rc) rc = scratch(b, (size_t)n_windows, &D.res);
	if (!rc) rc = scratch(b, 4 * (size_t)n_windows, &D.coop_i);
	if (!rc) rc = scratch(b, 128 * (size_t)n_windows, &D.coop_f);
	if (!rc) rc = scratch(b, trace ? (size_t)n_windows * MAX_ITERS_ALLOC : 1, &D.trace);
	if (!rc) rc = scratch(b, (size_t)dpt, &D.dp_trace);
	if (!rc) rc = scratch(b, (size_t)dft, &D.df_trace);
	if (!rc) rc = scratch(b, (size_t)gtab_total, &D.suf_tab);
	if (!rc) rc = scratch(b, 256 * (size_t)n_stile_total + 256, &D.Stile);
	if (node_all) {
		if (!rc) rc = scratch(b, (size_t)NH.tab_doubles, &b->N.tab);
		if (!rc) rc = scratch(b, (size_t)SSK_ND_SCR * NH.lane.size(), &b->N.scr);
		if (!rc) rc = scratch(b, node_big ? (size_t)NH.hb_doubles : 0, &b->N.hb);
	}
	if (!rc) rc = scratch(b, pose.size(), &b->pose_bak);
	if (!rc) rc = scratch(b, lm.size(), &b->lm_bak);
	D.trace_on = trace ? 1 : 0;
	if (!rc) rc = finalize(b);
	if (rc) { ssk_lm_batch_destroy(b); return rc; }
	if (timing) {
		(void)hipDeviceSynchronize();
		auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) { return std::chrono::duration<double, std::milli>(c - a).count(); };
		std::fprintf(stderr, "[ssk timing] ssk_lm_batch_create (%d windows): symbolic %.3f ms (incidences %.3f, suffix trie %.3f, tiles %.3f, concatenation %.3f), allocate + upload %.3f ms\n",
		             n_windows, ms(t_begin, t_sym), t_ph[0], t_ph[1], t_ph[2], t_ph[3], ms(t_sym, std::chrono::steady_clock::now()));
		std::fprintf(stderr, "[ssk timing]   tiles: footprints %.3f ms, cut %.3f, records %.3f\n", t_sub[0], t_sub[1], t_sub[2]);
	}
	*out = b;
	return SSK_OK;
}

// SSK_LM_DUMP=<directory> (diagnostic, read per call): every window handed to ssk_lm_batch_create is written to <directory>/win_<counter>.bin -- eight
// 64-bit counts (edges, 12 x edges, landmarks, 3 x landmarks, observations, 4 x observations, observations + 1, path entries), then the arrays in
// that order: the format tools / tests replay through the symbolic builder on the CPU
static void lm_dump_windows(const ssk_lm_window* windows, int n_windows)
{
	const char* dir = std::getenv("SSK_LM_DUMP");
	if (!dir || !windows) return;
	static std::atomic<int> counter{0};
	for (int w = 0; w < n_windows; ++w) {
		const ssk_lm_window& W = windows[w];
		char path[512];
		std::snprintf(path, sizeof path, "%s/win_%05d.bin", dir, counter++);
		FILE* f = std::fopen(path, "wb");
		if (!f) return;
		const long long np = W.n_obs ? W.obs_path_off[W.n_obs] : 0;
		const long long n[8] = {W.n_edges_total, 12LL * W.n_edges_total, W.n_lms_total, 3LL * W.n_lms_total, W.n_obs, 4LL * W.n_obs, W.n_obs + 1LL, np};
		std::fwrite(n, sizeof n[0], 8, f);
		std::fwrite(W.edge_unknown_slot, 4, (size_t)n[0], f); std::fwrite(W.edge_inv_pose, 8, (size_t)n[1], f);
		std::fwrite(W.lm_unknown_slot, 4, (size_t)n[2], f); std::fwrite(W.lm_pos, 8, (size_t)n[3], f);
		std::fwrite(W.obs_lm, 4, (size_t)n[4], f); std::fwrite(W.obs_px, 4, (size_t)n[5], f);
		if (W.n_obs) std::fwrite(W.obs_path_off, 4, (size_t)n[6], f); else { const int z = 0; std::fwrite(&z, 4, 1, f); }
		if (np) std::fwrite(W.path_edge, 4, (size_t)np, f);
		std::fclose(f);
	}
}

int ssk_lm_batch_create(ssk_ctx* ctx, const ssk_lm_window* windows, int n_windows, int trace, ssk_lm_batch** out)
{
	if (ctx && n_windows > 0) lm_dump_windows(windows, n_windows);
	// SSK_LM_KERNEL: l2 = the general scratch-in-L2 kernel, tiled / coop = the slot kernels (one / several workgroups per window),
	// acc = on-chip accumulation (windows with E <= SSK_LMA_MAX_E; on request: its accumulators cost LDS, i.e. more tiles per pass)
	const char* force = ctx->env.lm_kernel.empty() ? nullptr : ctx->env.lm_kernel.c_str();
	const bool l2 = force && std::strcmp(force, "l2") == 0;
	const bool want_acc = force && std::strcmp(force, "acc") == 0;
	// the default: node tables only (node kernel for windows whose reduced system fits LDS, the phase-split kernels otherwise; =split forces
	// the latter); a batch the node builder cannot take, or a forced kernel of rounds 1-3, goes through the full builder
	if (!force || std::strcmp(force, "split") == 0) {
		if (!ctx) return SSK_ERR_BAD_ARG;
		if (!windows || n_windows < 1 || !out) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_lm_batch_create: bad argument");
		SSK_HIP(ctx, hipSetDevice(ctx->device));
		std::vector<char> unfit;
		ssk_lm_batch* lean_b = nullptr;
		const int rc = lm_batch_create_lean(ctx, windows, n_windows, trace, &lean_b, force != nullptr, &unfit);
		if (rc != SSK_OK && rc != 1) return rc;
		int n_unfit = 0;
		for (char u : unfit) n_unfit += u != 0;
		if (rc == SSK_OK && n_unfit == 0) { *out = lean_b; return SSK_OK; }
		if (force) return ssk_set_error(ctx, SSK_ERR_CAPACITY, "ssk_lm_batch_create: SSK_LM_KERNEL=split, but a window of the batch does not fit the node builder");
		// per-window fallback: the windows the node builder cannot take go through the full builder ALONE (a sub-batch); the others keep their kernels
		// (lean_b: the batch of the fitting windows, made in the one pass above)
		if (rc == SSK_OK) {
			std::unique_ptr<ssk_lm_batch, void (*)(ssk_lm_batch*)> c(new ssk_lm_batch(), ssk_lm_batch_destroy);
			c->ctx = ctx; c->n_windows = n_windows; c->trace = trace; c->max_iters_alloc = 64;
			c->sub_of.resize((size_t)n_windows); c->idx_in.resize((size_t)n_windows);
			c->sub.assign(2, nullptr);
			c->sub[0] = lean_b;
			std::vector<ssk_lm_window> part1;
			int n_fit = 0;
			for (int w = 0; w < n_windows; ++w) {
				const int k = unfit[(size_t)w] ? 1 : 0;
				c->sub_of[(size_t)w] = k; c->idx_in[(size_t)w] = k ? (int)part1.size() : n_fit++;
				if (k) part1.push_back(windows[w]);
			}
			const int rc2 = lm_batch_create_impl(ctx, part1.data(), (int)part1.size(), trace, &c->sub[1], false, false);
			if (rc2 != SSK_OK) return rc2;                  // (the composite's destructor frees what was made)
			for (ssk_lm_batch* sb : c->sub) { c->symbolic_ms += sb->symbolic_ms; c->symbolic_threads = std::max(c->symbolic_threads, sb->symbolic_threads); c->max_n6 = std::max(c->max_n6, sb->max_n6); }
			*out = c.release();
			return SSK_OK;
		}
	}
	return lm_batch_create_impl(ctx, windows, n_windows, trace, out, l2, want_acc);
}

// A cooperative launch whose workgroups did not all arrive at a barrier (stop_reason 5) leaves a half-updated state: whoever reads the
// batch next -- results, state, trace, the shared-landmark pack, another run or a reset -- first goes back to the state the launch
// started from and repeats the solve with one workgroup per window (no cross-workgroup barrier, same contract).
static int lm_batch_settle(ssk_ctx* ctx, ssk_lm_batch* b)
{
	if (ctx == b->ctx) b->used_own = true;
	if (b->ready_pending) {                           // (the create's zero fill, possibly on another context's stream)
		SSK_HIP(ctx, hipStreamWaitEvent(ctx->stream, b->ev_ready, 0));
		b->ready_pending = false;
	}
	if (!b->coop_pending) return SSK_OK;
	b->coop_pending = false;
	SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
	std::vector<ssk_lm_result> r((size_t)b->n_windows);
	SSK_HIP(ctx, hipMemcpy(r.data(), b->D.res, sizeof(ssk_lm_result) * r.size(), hipMemcpyDeviceToHost));
	bool timed_out = false;
	for (const ssk_lm_result& x : r) timed_out |= x.stop_reason == 5;
	if (!timed_out || ctx->env.lm_coop_no_fallback) return SSK_OK;
	const bool verbose = ctx->env.lm_debug;
	if (verbose) std::fprintf(stderr, "[ssk lm] cooperative launch timed out at a barrier: repeating with one workgroup per window\n");
	SSK_HIP(ctx, hipMemcpyAsync(b->D.pose, b->pose_bak, sizeof(double) * b->n_pose, hipMemcpyDeviceToDevice, ctx->stream));
	SSK_HIP(ctx, hipMemcpyAsync(b->D.lm, b->lm_bak, sizeof(double) * b->n_lm, hipMemcpyDeviceToDevice, ctx->stream));
	return ssk_lm_launch_tiled(ctx, b->D, b->coop_P, b->n_windows, b->tiled_dyn_bytes, b->schur_mfma, b->lds_sys);
}

int ssk_lm_batch_reset(ssk_ctx* ctx, ssk_lm_batch* b)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!b) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_lm_batch_reset: batch is NULL");
	if (!b->sub.empty()) { for (ssk_lm_batch* sb : b->sub) { const int rcs = ssk_lm_batch_reset(ctx, sb); if (rcs) return rcs; } return SSK_OK; }
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	b->coop_pending = false;                          // (the state is overwritten: a timed-out launch needs no repair)
	{ const int rc0 = lm_batch_settle(ctx, b); if (rc0) return rc0; }      // (the create's zero fill)
	return lm_batch_mark(ctx, b, ssk_lm_launch_reset(ctx, b->D.pose, b->D.pose0, b->n_pose, b->D.lm, b->D.lm0, b->n_lm));
}

static int lm_batch_run_impl(ssk_ctx* ctx, ssk_lm_batch* b, const ssk_lm_params* p);
int ssk_lm_batch_run(ssk_ctx* ctx, ssk_lm_batch* b, const ssk_lm_params* p)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (b && !b->sub.empty()) {
		// the few windows of the full builder run BESIDE the many (a side stream of the context, forked from and joined to its stream by events): one
		// persistent workgroup per odd window would otherwise add its whole latency to the batch's
		SSK_HIP(ctx, hipSetDevice(ctx->device));
		if (!ctx->side_stream) {
			SSK_HIP(ctx, hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
			SSK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_side_a, hipEventDisableTiming));
			SSK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_side_b, hipEventDisableTiming));
		}
		SSK_HIP(ctx, hipEventRecord(ctx->ev_side_a, ctx->stream));
		SSK_HIP(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->ev_side_a, 0));
		hipStream_t main_stream = ctx->stream;
		ctx->stream = ctx->side_stream;                 // (a context serves one thread at a time: nobody else looks at it meanwhile)
		const int rc1 = ssk_lm_batch_run(ctx, b->sub[1], p);
		hipError_t e1 = hipEventRecord(ctx->ev_side_b, ctx->side_stream);
		ctx->stream = main_stream;
		if (rc1) return rc1;
		if (e1 != hipSuccess) return ssk_set_error(ctx, SSK_ERR_HIP, "ssk_lm_batch_run: hipEventRecord failed: %s", hipGetErrorString(e1));
		const int rc0 = ssk_lm_batch_run(ctx, b->sub[0], p);
		SSK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_side_b, 0));
		return rc0;
	}
	const int rc = lm_batch_run_impl(ctx, b, p);
	return b ? lm_batch_mark(ctx, b, rc) : rc;
}
static int lm_batch_run_impl(ssk_ctx* ctx, ssk_lm_batch* b, const ssk_lm_params* p)
{
	if (!b || !p) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_lm_batch_run: NULL argument");
	if (p->max_iters < 0 || (b->trace && p->max_iters > b->max_iters_alloc))
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_lm_batch_run: max_iters %d outside [0,%d] for a traced batch", p->max_iters, b->max_iters_alloc);
	if (p->use_robust_kernel && !(p->kernel_param > 0)) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_lm_batch_run: kernel_param must be > 0");
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	lm_cam P;
	P.fx_l = p->fx_l; P.fy_l = p->fy_l; P.cx_l = p->cx_l; P.cy_l = p->cy_l;
	P.fx_r = p->fx_r; P.fy_r = p->fy_r; P.cx_r = p->cx_r; P.cy_r = p->cy_r; P.baseline = p->baseline;
	for (int i = 0; i < 12; ++i) P.S[i] = p->sensor_pose[i];
	P.max_iters = p->max_iters; P.robust = p->use_robust_kernel; P.kernel = p->kernel_param; P.tau = p->tau;
	P.max_lambda = p->max_lambda; P.stop_rmse = p->max_error_per_obs_to_stop; P.conv_rel = p->converged_rel_decrease;
	// two kernels, one contract: the LDS-tiled kernel when every window of the batch fits it (SSK_LM_KERNEL=l2 forces
	// the general scratch-in-L2 kernel, which takes windows of any size)
	const char* force = ctx->env.lm_kernel.empty() ? nullptr : ctx->env.lm_kernel.c_str();
	// few large windows (the un-batched drop-in: one local area of tens of tiles): G workgroups per window share its tiles,
	// all resident at once (one per CU); SSK_LM_KERNEL=tiled keeps one workgroup per window, =coop forces the split
	{ const int rc0 = lm_batch_settle(ctx, b); if (rc0) return rc0; }      // (a run on top of a timed-out cooperative launch continues from the repaired state)
	if (b->lean) {
		// a mixed batch: the small windows on the persistent node kernel, the others on the phase-split kernels, one after the other on the stream
		if (b->pose_n) { const int rc1 = ssk_lm_launch_pose_only(ctx, b->D, b->pose_order, P, b->pose_n, b->pose_two, b->pose_dyn_bytes); if (rc1) return rc1; }
		if (b->node_n) { const int rc1 = ssk_lm_launch_node(ctx, b->D, b->N, P, b->node_n, b->node_dyn_bytes, false, b->node_small); if (rc1) return rc1; }
		if (b->split_n) return ssk_lm_launch_split(ctx, b->D, b->N2, b->S, P, b->split_n, b->split_solve_dyn_bytes, b->split_init_dyn_bytes);
		return SSK_OK;
	}
	if (b->node_ok) return ssk_lm_launch_node(ctx, b->D, b->N, P, b->n_windows, b->node_dyn_bytes, b->node_big, b->node_small);
	if (b->acc_ok) return ssk_lm_launch_acc(ctx, b->D, P, b->n_windows, b->tiled_dyn_bytes, b->schur_mfma);
	if (b->tiled_ok && !(force && (std::strcmp(force, "l2") == 0 || std::strcmp(force, "tiled") == 0))) {
		// the grid barrier of the cooperative kernel needs all n_windows * G workgroups resident: G from what the occupancy
		// query says one CU holds with this launch's LDS, not from the CU count alone
		const int per_cu = ssk_lm_coop_blocks_per_cu(ctx, b->tiled_dyn_bytes, b->schur_mfma);
		// at most 32 workgroups per window: beyond that the barriers (seven, then four per attempt) cost more than the shorter tile phase
		// gives back (measured on 50- and 120-tile local areas: 32 beats 64 by 3 %, 16 loses 7 %); SSK_LM_COOP_G overrides
		int gmax = 32;
		if (ctx->env.lm_coop_g > 0) gmax = ctx->env.lm_coop_g;
		int G = std::min(std::min(b->max_tiles, gmax), per_cu * ctx->n_cu / std::max(b->n_windows, 1));
		// (windows whose suffix table lives in L2 scratch need one table per workgroup: sized for it only when the batch is small enough)
		const bool want = b->gtab_multi && (force && std::strcmp(force, "coop") == 0 ? G >= 2 : (G >= 4 && b->n_windows * 4 <= ctx->n_cu));
		if (ctx->env.lm_debug) std::fprintf(stderr, "[ssk lm] run: %d windows, max tiles %d, %d workgroups per CU -> %s, G %d, max 6E %d, Schur %s\n", b->n_windows, b->max_tiles, per_cu,
		                                              want ? "cooperative" : "one workgroup per window", G, b->max_n6, b->schur_mfma ? "mfma" : "sparse");
		if (want) {
			// residency is still not guaranteed (another stream or process may hold CUs): keep the state the launch starts
			// from, ssk_lm_batch_results repeats a launch that ended in stop_reason 5 with one workgroup per window
			SSK_HIP(ctx, hipMemcpyAsync(b->pose_bak, b->D.pose, sizeof(double) * b->n_pose, hipMemcpyDeviceToDevice, ctx->stream));
			SSK_HIP(ctx, hipMemcpyAsync(b->lm_bak, b->D.lm, sizeof(double) * b->n_lm, hipMemcpyDeviceToDevice, ctx->stream));
			const bool test_timeout = ctx->env.lm_coop_test_timeout;
			const int rc = ssk_lm_launch_coop(ctx, b->D, P, b->n_windows, G, b->tiled_dyn_bytes, test_timeout, b->schur_mfma);
			if (rc == SSK_OK) { b->coop_pending = true; b->coop_P = P; }
			return rc;
		}
	}
	if (b->tiled_ok && !(force && std::strcmp(force, "l2") == 0 && b->has_general)) return ssk_lm_launch_tiled(ctx, b->D, P, b->n_windows, b->tiled_dyn_bytes, b->schur_mfma, b->lds_sys);
	if (!b->has_general) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_lm_batch_run: the batch was built without the general kernel's structures");
	return ssk_lm_launch(ctx, b->D, P, b->n_windows, b->max_n6);
}

int ssk_lm_batch_kernel_name(ssk_ctx* ctx, ssk_lm_batch* b, char* out, int cap)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!b || !out || cap < 1) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_lm_batch_kernel_name: bad argument");
	if (!b->sub.empty()) {                            // "<kernel of the windows the node builder took>+<kernel of the others>"
		char a0[96], a1[96];
		int rcs = ssk_lm_batch_kernel_name(ctx, b->sub[0], a0, (int)sizeof a0);
		if (!rcs) rcs = ssk_lm_batch_kernel_name(ctx, b->sub[1], a1, (int)sizeof a1);
		if (rcs) return rcs;
		std::snprintf(out, (size_t)cap, "%s+%s", a0, a1);
		return SSK_OK;
	}
	// the same decisions as ssk_lm_batch_run
	const char* force = ctx->env.lm_kernel.empty() ? nullptr : ctx->env.lm_kernel.c_str();
	const char* name = "lm_solve_kernel";
	if (b->split_ok) name = "lm_split_linearise_kernel";      // (the heaviest kernel of a pass; a mixed batch also runs lm_solve_node_kernel on its small windows)
	else if (b->node_ok) name = b->node_big ? "lm_solve_node_big_kernel" : b->node_small ? "lm_solve_node_e10_kernel" : "lm_solve_node_kernel";
	else if (b->lean && b->pose_n) name = "lm_pose_only_kernel";
	else if (b->acc_ok) name = b->schur_mfma ? "lm_solve_acc_mfma_kernel" : "lm_solve_acc_kernel";
	else if (b->tiled_ok && !(force && std::strcmp(force, "l2") == 0 && b->has_general)) {
		bool coop = false;
		if (!(force && (std::strcmp(force, "l2") == 0 || std::strcmp(force, "tiled") == 0))) {
			const int per_cu = ssk_lm_coop_blocks_per_cu(ctx, b->tiled_dyn_bytes, b->schur_mfma);
			int gmax = 32;
			if (ctx->env.lm_coop_g > 0) gmax = ctx->env.lm_coop_g;
			const int G = std::min(std::min(b->max_tiles, gmax), per_cu * ctx->n_cu / std::max(b->n_windows, 1));
			coop = b->gtab_multi && (force && std::strcmp(force, "coop") == 0 ? G >= 2 : (G >= 4 && b->n_windows * 4 <= ctx->n_cu));
		}
		if (coop) name = b->schur_mfma == 2 ? "lm_solve_coop_ldlm_kernel" : b->schur_mfma ? "lm_solve_coop_mfma_kernel" : "lm_solve_coop_kernel";
		else name = b->schur_mfma == 2 ? "lm_solve_tiled_ldlm_kernel" : b->schur_mfma ? "lm_solve_tiled_mfma_kernel" : b->lds_sys ? "lm_solve_tiled_lds_kernel" : "lm_solve_tiled_kernel";
	}
	std::snprintf(out, (size_t)cap, "%s", name);
	return SSK_OK;
}

int ssk_lm_batch_solver_name(ssk_ctx* ctx, ssk_lm_batch* b, char* out, int cap)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!b || !out || cap < 1) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_lm_batch_solver_name: bad argument");
	if (!b->sub.empty()) return ssk_lm_batch_solver_name(ctx, b->sub[0], out, cap);
	const char* name = "ldl (rounds 1-3 kernels)";
	if (b->split_ok) name = b->S.solve_mfma == 2 ? "ldl_solve_mfma_tiles" : b->S.solve_mfma == 1 ? "ldl_solve_mfma_lead" : "ldl_solve_regtile";
	else if (b->node_ok) name = b->node_big ? "ldl_solve_regtile" : b->node_small ? "ldl_solve_mfma_wave" : "ldl_solve_tiles4_lds / ldl_solve_lds";
	else if (b->lean && b->pose_n) name = "in registers (pose-only)";
	std::snprintf(out, (size_t)cap, "%s", name);
	return SSK_OK;
}

int ssk_lm_batch_symbolic_ms(ssk_ctx* ctx, ssk_lm_batch* b, double* ms, int* threads)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!b) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_lm_batch_symbolic_ms: batch is NULL");
	if (ms) *ms = b->symbolic_ms;
	if (threads) *threads = b->symbolic_threads;
	return SSK_OK;
}

int ssk_lm_batch_results(ssk_ctx* ctx, ssk_lm_batch* b, ssk_lm_result* out)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!b || !out) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_lm_batch_results: NULL argument");
	if (!b->sub.empty()) {
		std::vector<ssk_lm_result> r0((size_t)b->sub[0]->n_windows), r1((size_t)b->sub[1]->n_windows);
		int rcs = ssk_lm_batch_results(ctx, b->sub[0], r0.data());
		if (!rcs) rcs = ssk_lm_batch_results(ctx, b->sub[1], r1.data());
		if (rcs) return rcs;
		for (int w = 0; w < b->n_windows; ++w) out[w] = (b->sub_of[(size_t)w] ? r1 : r0)[(size_t)b->idx_in[(size_t)w]];
		return SSK_OK;
	}
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	{ const int rc0 = lm_batch_settle(ctx, b); if (rc0) return rc0; }
	SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
	SSK_HIP(ctx, hipMemcpy(out, b->D.res, sizeof(ssk_lm_result) * (size_t)b->n_windows, hipMemcpyDeviceToHost));
	b->coop_pending = false;
	for (int w = 0; w < b->n_windows; ++w)
		if (out[w].stop_reason == 5)
			return ssk_set_error(ctx, SSK_ERR_HIP, "ssk_lm_batch_results: window %d: the cooperative kernel's workgroups did not all arrive at a barrier "
			                     "(not all resident? another kernel holding CUs) -- set SSK_LM_KERNEL=tiled", w);
	return SSK_OK;
}

int ssk_lm_batch_get_state(ssk_ctx* ctx, ssk_lm_batch* b, int w, double* edge_inv_pose, double* lm_pos)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!b || w < 0 || w >= b->n_windows) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_lm_batch_get_state: bad argument");
	if (!b->sub.empty()) return ssk_lm_batch_get_state(ctx, b->sub[(size_t)b->sub_of[(size_t)w]], b->idx_in[(size_t)w], edge_inv_pose, lm_pos);
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	{ const int rc0 = lm_batch_settle(ctx, b); if (rc0) return rc0; }
	SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
	const lm_win& H = b->hdr[w];
	if (edge_inv_pose && H.n_edges) SSK_HIP(ctx, hipMemcpy(edge_inv_pose, b->D.pose + 12 * (size_t)H.edge0, sizeof(double) * 12 * (size_t)H.n_edges, hipMemcpyDeviceToHost));
	if (lm_pos && H.n_lms) SSK_HIP(ctx, hipMemcpy(lm_pos, b->D.lm + 3 * (size_t)H.lm0, sizeof(double) * 3 * (size_t)H.n_lms, hipMemcpyDeviceToHost));
	return SSK_OK;
}

int ssk_lm_batch_get_state_all(ssk_ctx* ctx, ssk_lm_batch* b, double* edge_inv_pose, double* lm_pos)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!b) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_lm_batch_get_state_all: batch is NULL");
	if (!b->sub.empty()) {
		// a composite: every sub-batch's arrays, then window by window to the caller's order
		std::vector<double> pe[2], pl[2];
		std::vector<size_t> eo[2], lo[2];
		for (int k = 0; k < 2; ++k) {
			ssk_lm_batch* sb = b->sub[(size_t)k];
			pe[k].resize(sb->n_pose); pl[k].resize(sb->n_lm);
			const int rcs = ssk_lm_batch_get_state_all(ctx, sb, pe[k].data(), pl[k].data());
			if (rcs) return rcs;
		}
		size_t oe = 0, ol = 0;
		for (int w = 0; w < b->n_windows; ++w) {
			const int k = b->sub_of[(size_t)w];
			const lm_win& H = b->sub[(size_t)k]->hdr[(size_t)b->idx_in[(size_t)w]];
			if (edge_inv_pose) std::memcpy(edge_inv_pose + oe, pe[k].data() + 12 * (size_t)H.edge0, sizeof(double) * 12 * (size_t)H.n_edges);
			if (lm_pos) std::memcpy(lm_pos + ol, pl[k].data() + 3 * (size_t)H.lm0, sizeof(double) * 3 * (size_t)H.n_lms);
			oe += 12 * (size_t)H.n_edges; ol += 3 * (size_t)H.n_lms;
		}
		return SSK_OK;
	}
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	{ const int rc0 = lm_batch_settle(ctx, b); if (rc0) return rc0; }
	SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
	// (the pose / landmark families of a batch are the windows' arrays back to back in window order)
	if (edge_inv_pose && b->n_pose) SSK_HIP(ctx, hipMemcpy(edge_inv_pose, b->D.pose, sizeof(double) * b->n_pose, hipMemcpyDeviceToHost));
	if (lm_pos && b->n_lm) SSK_HIP(ctx, hipMemcpy(lm_pos, b->D.lm, sizeof(double) * b->n_lm, hipMemcpyDeviceToHost));
	return SSK_OK;
}

int ssk_lm_batch_set_state(ssk_ctx* ctx, ssk_lm_batch* b, int w, const double* edge_inv_pose, const double* lm_pos)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!b || w < 0 || w >= b->n_windows) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_lm_batch_set_state: bad argument");
	if (!b->sub.empty()) return ssk_lm_batch_set_state(ctx, b->sub[(size_t)b->sub_of[(size_t)w]], b->idx_in[(size_t)w], edge_inv_pose, lm_pos);
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	{ const int rc0 = lm_batch_settle(ctx, b); if (rc0) return rc0; }
	SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
	const lm_win& H = b->hdr[w];
	// both copies: the working state and the one ssk_lm_batch_reset restores
	if (edge_inv_pose && H.n_edges) {
		const size_t off = 12 * (size_t)H.edge0, bytes = sizeof(double) * 12 * (size_t)H.n_edges;
		SSK_HIP(ctx, hipMemcpy(b->D.pose + off, edge_inv_pose, bytes, hipMemcpyHostToDevice));
		SSK_HIP(ctx, hipMemcpy(const_cast<double*>(b->D.pose0) + off, edge_inv_pose, bytes, hipMemcpyHostToDevice));
	}
	if (lm_pos && H.n_lms) {
		const size_t off = 3 * (size_t)H.lm0, bytes = sizeof(double) * 3 * (size_t)H.n_lms;
		SSK_HIP(ctx, hipMemcpy(b->D.lm + off, lm_pos, bytes, hipMemcpyHostToDevice));
		SSK_HIP(ctx, hipMemcpy(const_cast<double*>(b->D.lm0) + off, lm_pos, bytes, hipMemcpyHostToDevice));
	}
	return SSK_OK;
}

int ssk_lm_batch_get_trace(ssk_ctx* ctx, ssk_lm_batch* b, int w, ssk_lm_trace_entry* entries, int cap,
                           double* delta_p, double* delta_f, int* n_entries)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!b || w < 0 || w >= b->n_windows || !entries || !n_entries) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_lm_batch_get_trace: bad argument");
	if (!b->sub.empty()) return ssk_lm_batch_get_trace(ctx, b->sub[(size_t)b->sub_of[(size_t)w]], b->idx_in[(size_t)w], entries, cap, delta_p, delta_f, n_entries);
	if (!b->trace) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_lm_batch_get_trace: the batch was created without trace");
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	{ const int rc0 = lm_batch_settle(ctx, b); if (rc0) return rc0; }
	SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
	ssk_lm_result r;
	SSK_HIP(ctx, hipMemcpy(&r, b->D.res + w, sizeof r, hipMemcpyDeviceToHost));
	const int n = std::min(r.iters, cap);
	const lm_win& H = b->hdr[w];
	if (n > 0) {
		SSK_HIP(ctx, hipMemcpy(entries, b->D.trace + H.trace0, sizeof(ssk_lm_trace_entry) * (size_t)n, hipMemcpyDeviceToHost));
		if (delta_p && H.E) SSK_HIP(ctx, hipMemcpy(delta_p, b->D.dp_trace + H.dp_trace0, sizeof(double) * (size_t)n * 6 * H.E, hipMemcpyDeviceToHost));
		if (delta_f && H.L) SSK_HIP(ctx, hipMemcpy(delta_f, b->D.df_trace + H.df_trace0, sizeof(double) * (size_t)n * 3 * H.L, hipMemcpyDeviceToHost));
	}
	*n_entries = n;
	return SSK_OK;
}

}  // extern "C"
