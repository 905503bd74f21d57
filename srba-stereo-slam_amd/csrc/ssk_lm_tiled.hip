// ssk_lm_tiled.hip -- the relative-BA Levenberg-Marquardt solve of ssk_lm.hip (same contract L1-L5, same results
// to rounding) with the whole linearisation kept ON CHIP: no Jacobian ever goes to HBM.
//
// Why: per LM iteration the algorithm needs 56 B per observation (pixels, ids, landmark) and the edge poses
// (SURVEY 8(d)); the scratch-in-L2 kernel moves ~9.5 MB per window-iteration because it materialises the 4x6
// Jacobian blocks and gathers them three times.  Here a window's landmarks are cut (host, ssk_capi_lm.hip) into
// tiles whose linearisation fits the 160 KB of LDS of a gfx950 CU, and a persistent 512-thread workgroup per
// window streams the tiles:
//   T   path-suffix table: the observations of a window share a few dozen spanning-tree paths; every distinct SUFFIX of a
//       path (host: a trie) gets its composed transform T_s = X_k (+) ... (+) X_{n-1} once per pass, level by level -> LDS
//   I1  lane = observation: q = T_path f (one transform instead of a chain), stereo projection, IRLS weight,
//       Jf = w M R_path (4x3); the record keeps only V_o = Jf^T Jf (6) and g_o = Jf^T w r (3), and per optimised path
//       entry the chained point q_i = T_suffix f.  The 4x3 block A_i = +-w M R_prefix of the entry is NOT stored:
//       R_path = R_prefix R_suffix, so A_i = +-Jf R_suffix^T and every product the later phases need follows from V_o,
//       g_o and the table: A_i^T Jf = +-R_s V_o, A_i^T w r = +-R_s g_o, A_i^T A_j = +-R_si V_o R_sj^T.  (A third of the LDS
//       bytes per observation of the first form: fewer, larger tiles -- a tile phase costs its latency whatever its size)
//   I2  lane = (landmark, edge) incidence: W = sum Je^T Jf, g_e part, in REGISTERS across the barrier;
//       lane = landmark: V = sum Jf^T Jf, g_f, Cholesky V + lambda I = L L^T, L^-1, c = L^-1 g_f      -> LDS
//   I3  incidence lanes: Y = W L^-T (so the Schur term W V^-1 W'^T = Y Y'^T is symmetric by construction),
//       reduced gradient term g_e - Y c; Y -> LDS (+ HBM for the back-substitution).
//       all waves: H_pp items (observation, pair of its optimised path entries): 6x6 block from
//       C = A_i^T A_j (3x3) and two cross products (90 FMA instead of 144) in 36 registers
//   I4  Schur items (landmark, pair of its incidences): Y_i Y_j^T
//   Items are sorted by block pair id; a lane owns a group of consecutive items of one block, a wave sums the
//   lanes of a chunk by segments of equal block id with a DPP segmented scan (fixed order: deterministic, no
//   floating-point atomics) into partial slots; the final block sum runs over the slot run of the block.
// After the tiles: reduced system assembled in LDS over the dead tile buffers, LDL^T + back-substitution in LDS
// (ssk_lm_device.h), landmark back-substitution from Y, trial state, robust error, lambda schedule: as ssk_lm.hip.
// A rejected step re-linearises at the same state (bit-identical values) instead of keeping W in HBM.
// HBM traffic per window-iteration: inputs (~60 KB, L2 resident) + Y (2 x 144 B per incidence) + partial slots.
#include "ssk_lm.h"

// LM parity is a 1e-6 relative tolerance (contract L5), not bit-exactness: let the compiler fuse multiply-adds here
// (the build disables contraction globally for the bit-exact front-end kernels)
#pragma clang fp contract(fast)

namespace {

#define SSK_LM_NT SSK_LMT_THREADS
#include "ssk_lm_device.h"

static_assert(NT == SSK_LMT_NT, "workgroup size of the tiled kernel");
constexpr int ENT_P = SSK_LMT_ENT, OBS_P = SSK_LMT_OBS, Y_P = SSK_LMT_Y, LMR_P = SSK_LMT_LMR;

__device__ __forceinline__ void cross3(const double* a, const double* b, double* c)
{
	c[0] = a[1] * b[2] - a[2] * b[1];
	c[1] = a[2] * b[0] - a[0] * b[2];
	c[2] = a[0] * b[1] - a[1] * b[0];
}

// Sum of 36-vectors over the lanes of a wave BY SEGMENT, in lane order.  Lanes are sorted by key; a segment starts
// at every key change and at lanes 0/16/32/48.  Segmented inclusive scan inside each row of 16 lanes with DPP row
// shifts (VALU only: the LDS is the contended resource of the chunk tasks, and a transpose through it cost more):
// step d in {1,2,4,8} adds the value of lane-d when it belongs to the same segment; the fixed shift order makes the
// result deterministic.  The last lane of every segment then stores its 36 sums to part[36 * segslot[segment]].
template <int D>
__device__ __forceinline__ double dpp_shr_f64(double v)
{
	const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x110 + D, 0xF, 0xF, true);    // row_shr:D, lanes shifted in from outside the row read 0
	const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x110 + D, 0xF, 0xF, true);
	return __hiloint2double(hi, lo);
}
template <int D>
__device__ __forceinline__ double dpp_shl_f64(double v)      // the value D lanes up inside the 16-lane row (0 from outside it)
{
	const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x100 + D, 0xF, 0xF, true);    // row_shl:D
	const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x100 + D, 0xF, 0xF, true);
	return __hiloint2double(hi, lo);
}
template <int D>
__device__ __forceinline__ bool dpp_same_key(int key)
{
	return __builtin_amdgcn_update_dpp(-1, key, 0x110 + D, 0xF, 0xF, false) == key;             // outside the row: -1, never a key
}

__device__ __forceinline__ void wave_seg_sum36(double (&acc)[36], int key, int lane, double* __restrict__ part, const int* __restrict__ segslot)
{
	const int prev = __shfl_up(key, 1, 64);
	const bool head = (lane & 15) == 0 || key != prev;
	const unsigned long long hm = __ballot(head);
	const bool tail = (lane & 15) == 15 || ((hm >> ((lane + 1) & 63)) & 1ull);
	const int seg = __popcll(hm & ((2ull << lane) - 1ull)) - 1;
	// "add the value d lanes below when it belongs to the same run" as ONE fma on a 0 / 1 mask (the select form costs two v_cndmask and
	// an add per step: 36 values x 4 steps); a lane shifted in from outside the row reads 0, so t * 0 is exact
	const double s1 = dpp_same_key<1>(key) ? 1.0 : 0.0, s2 = dpp_same_key<2>(key) ? 1.0 : 0.0, s4 = dpp_same_key<4>(key) ? 1.0 : 0.0, s8 = dpp_same_key<8>(key) ? 1.0 : 0.0;
#pragma unroll
	for (int i = 0; i < 36; ++i) {
		double v = acc[i];
		v = fma(dpp_shr_f64<1>(v), s1, v);
		v = fma(dpp_shr_f64<2>(v), s2, v);
		v = fma(dpp_shr_f64<4>(v), s4, v);
		v = fma(dpp_shr_f64<8>(v), s8, v);
		acc[i] = v;
	}
	if (tail) {                                            // one burst of stores, nothing waits on it
		double* out = part + 36 * (size_t)segslot[seg];
#pragma unroll
		for (int i = 0; i < 36; ++i) out[i] = acc[i];
	}
}

// H_pp lane group: the 6x6 blocks Je_1^T Je_2 of `cnt` items (pairs of entry records of one observation), summed into acc.
// block = [C, rows(C) x' q2 ; q1 x cols(C), q1 x cols(top right)] with C = A_1^T A_2 = s1 s2 R_1 V_o R_2^T (R_i: rotation of the
// entry's suffix, V_o = Jf^T Jf of the observation) and u x' q = q x u
__device__ __forceinline__ void pp_group(const double* __restrict__ ent, const double* __restrict__ obsr, const double* __restrict__ tab,
                                         const int* __restrict__ items, int cnt, double (&acc)[36])
{
	int nxt = cnt > 0 ? items[0] : 0;
	for (int k = 0; k < cnt; ++k) {
		const int itv = nxt;
		if (k + 1 < cnt) nxt = items[k + 1];              // next item id in flight during this item's arithmetic
		const double* e1 = ent + ENT_P * (itv & 0xFFFF);
		const double* e2 = ent + ENT_P * (itv >> 16);
		const bool diag = e1 == e2;                        // an entry paired with itself (diagonal blocks): one record, one rotation
		const double q1[3] = {e1[0], e1[1], e1[2]};
		const double pk1 = e1[3];
		const int c1 = __double2loint(pk1);
		const double* orr = obsr + OBS_P * __double2hiint(pk1);
		const double* T1 = tab + 12 * (c1 >> 1);
		double q2[3] = {q1[0], q1[1], q1[2]};
		int c2 = c1;
		if (!diag) { q2[0] = e2[0]; q2[1] = e2[1]; q2[2] = e2[2]; c2 = __double2loint(e2[3]); }
		const double* T2 = tab + 12 * (c2 >> 1);
		const double sg = ((c1 ^ c2) & 1) ? -1.0 : 1.0;
		const double V[6] = {orr[0], orr[1], orr[2], orr[3], orr[4], orr[5]};
		const double Vm[9] = {V[0], V[1], V[2], V[1], V[3], V[4], V[2], V[4], V[5]};
		double B[9], C[9], TR[9];
#pragma unroll
		for (int r = 0; r < 3; ++r) {
			const double r0 = sg * T1[4 * r], r1 = sg * T1[4 * r + 1], r2 = sg * T1[4 * r + 2];
#pragma unroll
			for (int c = 0; c < 3; ++c) B[3 * r + c] = r0 * Vm[c] + r1 * Vm[3 + c] + r2 * Vm[6 + c];
		}
		double R1[9];                                       // (kept for the diagonal case: R2 = R1 without a second read)
#pragma unroll
		for (int r = 0; r < 3; ++r) { R1[3 * r] = T1[4 * r]; R1[3 * r + 1] = T1[4 * r + 1]; R1[3 * r + 2] = T1[4 * r + 2]; }
#pragma unroll
		for (int c = 0; c < 3; ++c) {
			double t0 = R1[3 * c], t1 = R1[3 * c + 1], t2 = R1[3 * c + 2];
			if (!diag) { t0 = T2[4 * c]; t1 = T2[4 * c + 1]; t2 = T2[4 * c + 2]; }
#pragma unroll
			for (int r = 0; r < 3; ++r) C[3 * r + c] = B[3 * r] * t0 + B[3 * r + 1] * t1 + B[3 * r + 2] * t2;
		}
#pragma unroll
		for (int r = 0; r < 3; ++r) {
			cross3(q2, C + 3 * r, TR + 3 * r);
#pragma unroll
			for (int c = 0; c < 3; ++c) { acc[6 * r + c] += C[3 * r + c]; acc[6 * r + 3 + c] += TR[3 * r + c]; }
		}
#pragma unroll
		for (int c = 0; c < 3; ++c) {
			const double cc[3] = {C[c], C[3 + c], C[6 + c]}, tc[3] = {TR[c], TR[3 + c], TR[6 + c]};
			double x[3], y[3];
			cross3(q1, cc, x);
			cross3(q1, tc, y);
#pragma unroll
			for (int r = 0; r < 3; ++r) { acc[6 * (3 + r) + c] += x[r]; acc[6 * (3 + r) + 3 + c] += y[r]; }
		}
	}
}

// Schur lane group: Y_1 Y_2^T of `cnt` items (pairs of incidence records), summed into acc
__device__ __forceinline__ void schur_group(const double* __restrict__ Yr, const int* __restrict__ items, int cnt, double (&acc)[36])
{
	int nxt = cnt > 0 ? items[0] : 0;
	for (int k = 0; k < cnt; ++k) {
		const int itv = nxt;
		if (k + 1 < cnt) nxt = items[k + 1];
		const double* Y1 = Yr + Y_P * (itv & 0xFFFF);
		const double* Y2 = Yr + Y_P * (itv >> 16);
		double y1[18], y2[18];
#pragma unroll
		for (int i = 0; i < 18; ++i) y1[i] = Y1[i];
		if (Y1 != Y2) {
#pragma unroll
			for (int i = 0; i < 18; ++i) y2[i] = Y2[i];
		} else {
#pragma unroll
			for (int i = 0; i < 18; ++i) y2[i] = y1[i];
		}
#pragma unroll
		for (int a = 0; a < 6; ++a)
#pragma unroll
			for (int b = 0; b < 6; ++b)
				acc[6 * a + b] += y1[3 * a] * y2[3 * b] + y1[3 * a + 1] * y2[3 * b + 1] + y1[3 * a + 2] * y2[3 * b + 2];
	}
}

// 1/sqrt(x), x > 0, to ~1 ulp: hardware estimate + two Newton steps (no IEEE sqrt / divide sequence)
__device__ __forceinline__ double rsq_f64(double x)
{
	double y = __builtin_amdgcn_rsq(x);
	const double hx = 0.5 * x;
	y = y * (1.5 - hx * y * y);
	y = y * (1.5 - hx * y * y);
	return y;
}

// One step of the SE(3) chain without a branch on the direction: dir 0: t = R q + tr, dir 1: t = R^T (q - tr).
// R(i,j) of the step sits at T[si * i + sj * j] with (si, sj) = (4, 1) or (1, 4).
__device__ __forceinline__ void se3_step(const double* T, int dir, const double* q, double* t)
{
	const int si = dir ? 1 : 4, sj = dir ? 4 : 1;
	const double tr[3] = {T[3], T[7], T[11]};
	const double d0 = dir ? q[0] - tr[0] : q[0], d1 = dir ? q[1] - tr[1] : q[1], d2 = dir ? q[2] - tr[2] : q[2];
#pragma unroll
	for (int i = 0; i < 3; ++i) t[i] = T[si * i] * d0 + T[si * i + sj] * d1 + T[si * i + 2 * sj] * d2 + (dir ? 0.0 : tr[i]);
}

// stereo projection with the reciprocal depth by Newton iteration
__device__ __forceinline__ void project_fast(const lm_cam& P, const double* pc, double* h)
{
	const double iz = rcp_f64(pc[2]);
	h[0] = P.cx_l + P.fx_l * pc[0] * iz;
	h[1] = P.cy_l + P.fy_l * pc[1] * iz;
	h[2] = P.cx_r + P.fx_r * (pc[0] - P.baseline) * iz;
	h[3] = P.cy_r + P.fy_r * pc[1] * iz;
}

// Path-suffix table of a window at the poses `pose` (window-relative, LDS): tab[12 * id] = row-major [R | t] of the composed
// transform of suffix id; id 0 = identity.  Level by level (a suffix = one step (+) a shorter suffix), one barrier per level.
// recs: the (first step, rest) records -- an LDS copy made once per solve when the table itself lives in LDS (the table is rebuilt
// twice per LM attempt and every level would otherwise start with an L2 round trip), else the array in HBM.
__device__ __forceinline__ void suffix_compose(const double* pose, double* __restrict__ tab, int id, const int2 rec)
{
	const double* X = pose + 12 * (rec.x >> 1);
	const double* Tn = tab + 12 * rec.y;
	const int dir = rec.x & 1, si = dir ? 1 : 4, sj = dir ? 4 : 1;      // (R or R^T)(i,j) = X[si * i + sj * j]
	double tn[3] = {Tn[3], Tn[7], Tn[11]};
	if (dir) { tn[0] -= X[3]; tn[1] -= X[7]; tn[2] -= X[11]; }
	double* O = tab + 12 * id;
#pragma unroll
	for (int i = 0; i < 3; ++i) {
		const double r0 = X[si * i], r1 = X[si * i + sj], r2 = X[si * i + 2 * sj];
#pragma unroll
		for (int j = 0; j < 3; ++j) O[4 * i + j] = r0 * Tn[j] + r1 * Tn[4 + j] + r2 * Tn[8 + j];
		O[4 * i + 3] = r0 * tn[0] + r1 * tn[1] + r2 * tn[2] + (dir ? 0.0 : X[4 * i + 3]);
	}
}
// The table in LDS (few suffixes: the batched windows): a lane per suffix walks its own chain of steps -- first the codes from the suffix down to
// the empty one, then the transforms from the innermost step outwards, in registers: the same operations in the same order as the level
// scheme below (bit-identical), the shared prefixes recomputed, and ONE barrier instead of one per level (8.5 k -> ~2.5 k cycles per build
// on the bench window: a level was a chain of dependent LDS reads on two of the eight wavefronts)
__device__ void build_suffix_table_walk(const lm_win& W, const double* pose, double* __restrict__ tab, const int2* __restrict__ recs)
{
	const int tid = threadIdx.x;
	__syncthreads();                                   // (the caller's writes of the step records, once per solve, and of the poses)
	if (tid < 12) tab[tid] = (tid == 0 || tid == 5 || tid == 10) ? 1.0 : 0.0;
	for (int id = 1 + tid; id < W.n_suf; id += NT) {
		int code[SSK_LM_MAX_PATH], n = 0, cur = id;
#pragma unroll
		for (int k = 0; k < SSK_LM_MAX_PATH; ++k) {
			code[k] = 0;
			if (cur != 0) { const int2 r = recs[cur]; code[k] = r.x; cur = r.y; n = k + 1; }
		}
		double T[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
#pragma unroll
		for (int k = SSK_LM_MAX_PATH - 1; k >= 0; --k) {
			if (k >= n) continue;
			const double* X = pose + 12 * (code[k] >> 1);
			const int dir = code[k] & 1, si = dir ? 1 : 4, sj = dir ? 4 : 1;      // (R or R^T)(i,j) = X[si * i + sj * j]
			double tn[3] = {T[3], T[7], T[11]};
			if (dir) { tn[0] -= X[3]; tn[1] -= X[7]; tn[2] -= X[11]; }
			double O[12];
#pragma unroll
			for (int i = 0; i < 3; ++i) {
				const double r0 = X[si * i], r1 = X[si * i + sj], r2 = X[si * i + 2 * sj];
#pragma unroll
				for (int j = 0; j < 3; ++j) O[4 * i + j] = r0 * T[j] + r1 * T[4 + j] + r2 * T[8 + j];
				O[4 * i + 3] = r0 * tn[0] + r1 * tn[1] + r2 * tn[2] + (dir ? 0.0 : X[4 * i + 3]);
			}
#pragma unroll
			for (int i = 0; i < 12; ++i) T[i] = O[i];
		}
		double* o = tab + 12 * id;
#pragma unroll
		for (int i = 0; i < 12; ++i) o[i] = T[i];
	}
	__syncthreads();
}

__device__ void build_suffix_table(const lm_win& W, const double* pose, double* __restrict__ tab, const int2* __restrict__ recs)
{
	const int tid = threadIdx.x;
	if (tid < 12) tab[tid] = (tid == 0 || tid == 5 || tid == 10) ? 1.0 : 0.0;
	__syncthreads();
	for (int lvl = 1; lvl <= SSK_LM_MAX_PATH; ++lvl) {
		const int i0 = W.suf_lvl[lvl], i1 = W.suf_lvl[lvl + 1];
		if (i0 >= W.n_suf) break;                          // (uniform)
		for (int id = i0 + tid; id < i1; id += NT) suffix_compose(pose, tab, id, recs[id]);
		__syncthreads();
	}
}

// Robustified squared error of a packed observation record (tab: the suffix table at the state to score, lm: the window's landmarks)
__device__ __forceinline__ double obs_error_packed(const lm_cam& P, const double* tab, const double* __restrict__ lm, const int4 rec, const float4 px)
{
	const double* f = lm + 3 * (size_t)rec.y;
	const double* T = tab + 12 * rec.w;                    // (w: the id in the window's table; z may be tile-local)
	double q[3], pc[3], h[4];
#pragma unroll
	for (int i = 0; i < 3; ++i) q[i] = T[4 * i] * f[0] + T[4 * i + 1] * f[1] + T[4 * i + 2] * f[2] + T[4 * i + 3];
	se3_apply_inv(P.S, q, pc);
	project_fast(P, pc, h);
	const double r0 = (double)px.x - h[0], r1 = (double)px.y - h[1], r2 = (double)px.z - h[2], r3 = (double)px.w - h[3];
	const double s2 = r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3;
	if (!P.robust) return s2;
	const double k2 = P.kernel * P.kernel, u = 1.0 + s2 * rcp_f64(k2);          // rho = 2 k^2 (sqrt(1 + s2/k^2) - 1)
	return 2.0 * k2 * (u * rsq_f64(u) - 1.0);
}

// observations k = first, first + step, ... (one workgroup: first = thread id, step = NT)
__device__ double total_error_packed(const lm_dev& D, const lm_win& W, const lm_cam& P, const double* tab, const double* lm, double* s_part,
                                     int first, int step)
{
	double e = 0;
	for (int k = first; k < W.n_obs; k += step) e += obs_error_packed(P, tab, lm + 3 * (size_t)W.lm0, D.tobs_a[W.tobs0 + k], D.tobs_px[W.tobs0 + k]);
	return block_sum(e, s_part);
}

// ---- cooperative variant: G workgroups share one window (its tiles, incidences, landmarks and observations are dealt out
// among them; workgroup 0 solves the reduced system).  They meet at barriers in global memory: every thread fences its own
// stores, thread 0 arrives on a monotonic counter and waits for the G arrivals of this round, every thread fences again
// before it reads what the others wrote.  All G workgroups are resident (the host launches at most one per CU).  A wait that
// outlasts COOP_SPIN_LIMIT polls raises the window's abort flag and every workgroup leaves (stop_reason 5).
constexpr int COOP_SPIN_LIMIT = 1 << 22;
// `expect` = arrivals that complete a barrier round (= G; the forced-timeout test hook of ssk_lm_launch_coop passes G + 1, an
// arrival that never comes) and `spin_limit` = polls before a waiting workgroup gives up
struct coop_t { int* cnt; int* abort; double* f; int g, G, round, bank, expect, spin_limit; };

__device__ __forceinline__ bool coop_barrier(coop_t& c)
{
#ifdef SSK_COOP_ALL_FENCE
	__threadfence();
	__syncthreads();
#else
	// every thread's stores are out of the CU (workgroup-scope release + the barrier); ONE thread then publishes for the workgroup:
	// the agent-scope release writes the XCD's dirty L2 lines back once, not once per wavefront
	// (a workgroup-scope release alone does not wait for the stores: s_waitcnt vmcnt(0) -- acknowledged by L2 -- is spelt out)
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	__builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0) expcnt(7) lgkmcnt(15)
	__syncthreads();
#endif
	if (threadIdx.x == 0) {
		++c.round;
		__hip_atomic_fetch_add(c.cnt, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
		int spins = 0;
		while (__hip_atomic_load(c.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < c.round * c.expect) {
			__builtin_amdgcn_s_sleep(1);
			if (++spins > c.spin_limit || __hip_atomic_load(c.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
				__hip_atomic_store(c.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				break;
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // (one invalidation of the CU's L1 / the XCD's non-local L2 lines for the workgroup)
	}
	__syncthreads();
#ifdef SSK_COOP_ALL_FENCE
	__threadfence();
#else
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#endif
	return __hip_atomic_load(c.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
}

// the G workgroups' values v (already reduced inside each workgroup) combined in workgroup order: sum or max
template <bool MAX>
__device__ __forceinline__ double coop_combine(coop_t& c, double v, bool& alive)
{
	double* slot = c.f + 64 * c.bank;
	c.bank ^= 1;                                        // two banks: a workgroup is at most one exchange ahead of the slowest
	if (threadIdx.x == 0) slot[c.g] = v;
	alive = coop_barrier(c);
	double r = slot[0];
	for (int k = 1; k < c.G; ++k) r = MAX ? fmax(r, slot[k]) : r + slot[k];
	return r;
}

// Sum of component ab over the contiguous partial slots k0..k1-1 in order, eight independent loads in flight;
// padding adds +0.0.
template <int PITCH = 36>
__device__ __forceinline__ double slot_sum(const double* __restrict__ part, int k0, int k1, int ab)
{
	double s = 0.0;
	for (int k = k0; k < k1; k += 8) {
		// unconditional loads on a clamped slot (a load under a lane condition is compiled as a branch that waits for it at
		// once: the eight loads were serialised L2 round trips), the padding selected afterwards
		double v[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) v[u] = part[PITCH * (size_t)min(k + u, k1 - 1) + ab];
#pragma unroll
		for (int u = 0; u < 8; ++u) s += k + u < k1 ? v[u] : 0.0;
	}
	return s;
}

// p - s of one component: the H_pp slots k in [p0, p1) (pitch 36, component ab) and the Schur slots / matrix-core partials k in [s0, s1)
// (pitch SP, component sb), each summed in slot order; the loads of both runs are issued together, four and four per turn
template <int SP>
__device__ __forceinline__ double slot_diff(const double* __restrict__ Pp, int p0, int p1, int ab, const double* __restrict__ Sp, int s0, int s1, int sb)
{
	double p = 0.0, sc = 0.0;
	for (int kp = p0, ks = s0; kp < p1 || ks < s1; kp += 4, ks += 4) {
		double v[4], u[4];
#pragma unroll
		for (int t = 0; t < 4; ++t) v[t] = Pp[36 * (size_t)max(min(kp + t, p1 - 1), p0) + ab];
#pragma unroll
		for (int t = 0; t < 4; ++t) u[t] = Sp[SP * (size_t)max(min(ks + t, s1 - 1), s0) + sb];
#pragma unroll
		for (int t = 0; t < 4; ++t) p += kp + t < p1 ? v[t] : 0.0;
#pragma unroll
		for (int t = 0; t < 4; ++t) sc += ks + t < s1 ? u[t] : 0.0;
	}
	return p - sc;
}

// I1: linearise one observation (packed record: first tile-local entry | optimised entries << 16, landmark, suffix id of its path)
// into the LDS records: V_o = Jf^T Jf, g_o = Jf^T w r; per optimised entry the chained point and (suffix, direction, observation).
// tent: the tile's entry table (suffix id << 1 | direction); kobs: tile-local index of the observation.
__device__ void linearize_obs_lds(const lm_dev& D, const lm_win& W, const lm_cam& P, const double* tab, const int4 rec, const float4 px,
                                  const int* __restrict__ tent, int kobs, double* __restrict__ ent, double* __restrict__ orec)
{
	const int e0 = rec.x & 0xFFFF, n_opt = (rec.x >> 16) & 0xFF;
	double f[3];
	{
		const double* fp = D.lm + 3 * (size_t)(W.lm0 + rec.y);
		f[0] = fp[0]; f[1] = fp[1]; f[2] = fp[2];
	}
	int code = n_opt > 0 ? tent[e0] : 0;
	const double* T = tab + 12 * rec.z;
	double R[9], q[3];
#pragma unroll
	for (int i = 0; i < 3; ++i) {
		R[3 * i] = T[4 * i]; R[3 * i + 1] = T[4 * i + 1]; R[3 * i + 2] = T[4 * i + 2];
		q[i] = R[3 * i] * f[0] + R[3 * i + 1] * f[1] + R[3 * i + 2] * f[2] + T[4 * i + 3];
	}
	double pc[3], h[4], r[4];
	se3_apply_inv(P.S, q, pc);
	const double iz = rcp_f64(pc[2]), iz2 = iz * iz;
	h[0] = P.cx_l + P.fx_l * pc[0] * iz;
	h[1] = P.cy_l + P.fy_l * pc[1] * iz;
	h[2] = P.cx_r + P.fx_r * (pc[0] - P.baseline) * iz;
	h[3] = P.cy_r + P.fy_r * pc[1] * iz;
	const float z[4] = {px.x, px.y, px.z, px.w};
	double s2 = 0;
#pragma unroll
	for (int k = 0; k < 4; ++k) { r[k] = (double)z[k] - h[k]; s2 += r[k] * r[k]; }
	double w = 1.0;                                     // IRLS weight sqrt(rho'(s)/2s) = (1 + s2/k^2)^(-1/4)
	if (P.robust) { const double rs = rsq_f64(1.0 + s2 * rcp_f64(P.kernel * P.kernel)); w = rs * rsq_f64(rs); }
	// M = D R_S^T with the structural zeros of the projection Jacobian D (rows: d(u_l, v_l, u_r, v_r)/d(x, y, z))
	const double dxl = P.fx_l * iz, dzl = -P.fx_l * pc[0] * iz2, dyl = P.fy_l * iz, dwl = -P.fy_l * pc[1] * iz2;
	const double dxr = P.fx_r * iz, dzr = -P.fx_r * (pc[0] - P.baseline) * iz2, dyr = P.fy_r * iz, dwr = -P.fy_r * pc[1] * iz2;
	double M[12];
#pragma unroll
	for (int j = 0; j < 3; ++j) {
		M[j] = dxl * P.S[4 * j] + dzl * P.S[4 * j + 2];
		M[3 + j] = dyl * P.S[4 * j + 1] + dwl * P.S[4 * j + 2];
		M[6 + j] = dxr * P.S[4 * j] + dzr * P.S[4 * j + 2];
		M[9 + j] = dyr * P.S[4 * j + 1] + dwr * P.S[4 * j + 2];
	}
	double jf[12];                                      // Jf = w M R_path
#pragma unroll
	for (int i = 0; i < 4; ++i)
#pragma unroll
		for (int j = 0; j < 3; ++j)
			jf[3 * i + j] = w * (M[3 * i] * R[j] + M[3 * i + 1] * R[3 + j] + M[3 * i + 2] * R[6 + j]);
	const double wr[4] = {w * r[0], w * r[1], w * r[2], w * r[3]};
	orec[0] = jf[0] * jf[0] + jf[3] * jf[3] + jf[6] * jf[6] + jf[9] * jf[9];
	orec[1] = jf[0] * jf[1] + jf[3] * jf[4] + jf[6] * jf[7] + jf[9] * jf[10];
	orec[2] = jf[0] * jf[2] + jf[3] * jf[5] + jf[6] * jf[8] + jf[9] * jf[11];
	orec[3] = jf[1] * jf[1] + jf[4] * jf[4] + jf[7] * jf[7] + jf[10] * jf[10];
	orec[4] = jf[1] * jf[2] + jf[4] * jf[5] + jf[7] * jf[8] + jf[10] * jf[11];
	orec[5] = jf[2] * jf[2] + jf[5] * jf[5] + jf[8] * jf[8] + jf[11] * jf[11];
	orec[6] = jf[0] * wr[0] + jf[3] * wr[1] + jf[6] * wr[2] + jf[9] * wr[3];
	orec[7] = jf[1] * wr[0] + jf[4] * wr[1] + jf[7] * wr[2] + jf[10] * wr[3];
	orec[8] = jf[2] * wr[0] + jf[5] * wr[1] + jf[8] * wr[2] + jf[11] * wr[3];
	for (int k = 0; k < n_opt; ++k) {
		const int cd = code;
		if (k + 1 < n_opt) code = tent[e0 + k + 1];         // next entry's code in flight
		const double* Ts = tab + 12 * (cd >> 1);
		double* er = ent + ENT_P * (e0 + k);
#pragma unroll
		for (int i = 0; i < 3; ++i) er[i] = Ts[4 * i] * f[0] + Ts[4 * i + 1] * f[1] + Ts[4 * i + 2] * f[2] + Ts[4 * i + 3];
		er[3] = __hiloint2double(kobs, cd);
	}
}

// sums of a landmark's observation records k0 .. k1-1: V (6) and g_f (3)
__device__ __forceinline__ void landmark_sums(const double* __restrict__ obsr, int k0, int k1, double (&v)[9])
{
#pragma unroll
	for (int c = 0; c < 9; ++c) v[c] = 0.0;
	for (int k = k0; k < k1; ++k) {
		const double* orr = obsr + OBS_P * k;
#pragma unroll
		for (int c = 0; c < 9; ++c) v[c] += orr[c];
	}
}

// V + lambda I = L L^T of a landmark (v: V upper 6 + g 3) -> rec = (L^-1 lower 6, c = L^-1 g 3)
__device__ __forceinline__ void landmark_factor(const double (&v)[9], double lambda, double (&rec)[9])
{
	const double i00 = rsq_f64(v[0] + lambda);                 // 1 / L_kk straight from the reciprocal square root
	const double l10 = v[1] * i00, l20 = v[2] * i00;
	const double i11 = rsq_f64(v[3] + lambda - l10 * l10);
	const double l21 = (v[4] - l20 * l10) * i11;
	const double i22 = rsq_f64(v[5] + lambda - l20 * l20 - l21 * l21);
	const double i10 = -i11 * l10 * i00, i21 = -i22 * l21 * i11, i20 = -(i21 * l10 + i22 * l20) * i00;
	rec[0] = i00; rec[1] = i10; rec[2] = i11; rec[3] = i20; rec[4] = i21; rec[5] = i22;
	rec[6] = i00 * v[6]; rec[7] = i10 * v[6] + i11 * v[7]; rec[8] = i20 * v[6] + i21 * v[7] + i22 * v[8];
}

// One (landmark, edge) incidence: W = sum over its path entries of Je^T Jf (6x3, only when `unk`: the landmark is optimised) and
// the gradient term ge = sum Je^T w r, from the entry / observation records and the suffix table:
//   A^T Jf = s R_s V_o, A^T w r = s R_s g_o (s = +-1), Je = [A | q x A]
// lrec: the LANE's share of the incidence (ilane record: -, first lmep entry, entries, first entry location)
__device__ __forceinline__ void incidence_sums(const lm_dev& D, const lm_win& W, const double* __restrict__ ent, const double* __restrict__ obsr,
                                               const double* __restrict__ tab, const int4 lrec, bool unk, double (&Wm)[18], double (&ge)[6])
{
	const int cnt = lrec.z;
#pragma unroll
	for (int c = 0; c < 18; ++c) Wm[c] = 0.0;
#pragma unroll
	for (int c = 0; c < 6; ++c) ge[c] = 0.0;
	int nxt = lrec.w;
	for (int k = 0; k < cnt; ++k) {
		const int loc = nxt;
		if (k + 1 < cnt) nxt = D.lmep_loc[W.lmep0 + lrec.y + k + 1];
		const double* er = ent + ENT_P * (loc & 0xFFFF);
		const double* orr = obsr + OBS_P * (loc >> 16);
		const double q[3] = {er[0], er[1], er[2]};
		const int cd = __double2loint(er[3]);
		const double sg = (cd & 1) ? -1.0 : 1.0;
		const double* Ts = tab + 12 * (cd >> 1);
		double Rs[9], V[6], g[3], tv[3], x[3];
#pragma unroll
		for (int i = 0; i < 3; ++i) { Rs[3 * i] = sg * Ts[4 * i]; Rs[3 * i + 1] = sg * Ts[4 * i + 1]; Rs[3 * i + 2] = sg * Ts[4 * i + 2]; }
#pragma unroll
		for (int c = 0; c < 3; ++c) g[c] = orr[6 + c];
#pragma unroll
		for (int c = 0; c < 3; ++c) tv[c] = Rs[3 * c] * g[0] + Rs[3 * c + 1] * g[1] + Rs[3 * c + 2] * g[2];
		cross3(q, tv, x);
#pragma unroll
		for (int c = 0; c < 3; ++c) { ge[c] += tv[c]; ge[3 + c] += x[c]; }
		if (unk) {
#pragma unroll
			for (int c = 0; c < 6; ++c) V[c] = orr[c];
			const double Vm[9] = {V[0], V[1], V[2], V[1], V[3], V[4], V[2], V[4], V[5]};
#pragma unroll
			for (int b = 0; b < 3; ++b) {
				double cb[3];
#pragma unroll
				for (int c = 0; c < 3; ++c) cb[c] = Rs[3 * c] * Vm[b] + Rs[3 * c + 1] * Vm[3 + b] + Rs[3 * c + 2] * Vm[6 + b];
				cross3(q, cb, x);
#pragma unroll
				for (int c = 0; c < 3; ++c) { Wm[3 * c + b] += cb[c]; Wm[3 * (3 + c) + b] += x[c]; }
			}
		}
	}
}

typedef double v4d_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_add_f64(double* p, double v)
{
	(void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // ds_add_f64: no wait for a read
}

// wave_seg_sum36 with the accumulators in LDS: acc36 + 36 * key
__device__ __forceinline__ void wave_seg_add36(double (&acc)[36], int key, int lane, double* __restrict__ acc36)
{
	const int prev = __shfl_up(key, 1, 64);
	const bool head = (lane & 15) == 0 || key != prev;
	const unsigned long long hm = __ballot(head);
	const bool tail = (lane & 15) == 15 || ((hm >> ((lane + 1) & 63)) & 1ull);
	// "add the value d lanes below when it belongs to the same run" as ONE fma on a 0 / 1 mask (the select form costs two v_cndmask and
	// an add per step: 36 values x 4 steps); a lane shifted in from outside the row reads 0, so t * 0 is exact
	const double s1 = dpp_same_key<1>(key) ? 1.0 : 0.0, s2 = dpp_same_key<2>(key) ? 1.0 : 0.0, s4 = dpp_same_key<4>(key) ? 1.0 : 0.0, s8 = dpp_same_key<8>(key) ? 1.0 : 0.0;
#pragma unroll
	for (int i = 0; i < 36; ++i) {
		double v = acc[i];
		v = fma(dpp_shr_f64<1>(v), s1, v);
		v = fma(dpp_shr_f64<2>(v), s2, v);
		v = fma(dpp_shr_f64<4>(v), s4, v);
		v = fma(dpp_shr_f64<8>(v), s8, v);
		acc[i] = v;
	}
	double* out = acc36 + 36 * key;
	for (int r = 0; r < 4; ++r)                            // one DPP row per round: a block's runs in different rows add one after the other
		if (tail && (lane >> 4) == r) {
#pragma unroll
			for (int i = 0; i < 36; ++i) lds_add_f64(out + i, acc[i]);
		}
}

// S tile (I, J) += sum over the tile's landmarks that touch row tiles I and J of panel_I panel_J^T.  lmask / lbase: this lane's
// landmark(s) of the tile (lane, lane + 64): row-tile mask and first panel; pan: the tile's panels.
template <bool RMW>
__device__ __forceinline__ void schur_mfma_task(const double* __restrict__ pan, int mask0, int base0, int mask1, int base1, int I, int J, int lane,
                                                double* __restrict__ S_tile)
{
	v4d_t c = {0.0, 0.0, 0.0, 0.0};
	if (RMW) { c[0] = S_tile[lane]; c[1] = S_tile[64 + lane]; c[2] = S_tile[128 + lane]; c[3] = S_tile[192 + lane]; }
	const int l48 = lane < 48 ? lane : 47;
	const bool real = lane < 48;                           // k = 3: the fourth 16-lane group multiplies zeros
#pragma unroll
	for (int half = 0; half < 2; ++half) {
		const int mk = half ? mask1 : mask0, bs = half ? base1 : base0;
		// this lane's landmark: panels of row tiles I and J (rank of the bit inside the mask)
		const int pa = bs + __popc(mk & ((1 << I) - 1)), pb = bs + __popc(mk & ((1 << J) - 1));
		unsigned long long todo = __ballot(((mk >> I) & (mk >> J) & 1) != 0);
		while (todo) {
			const int l = __builtin_ctzll(todo);
			todo &= todo - 1;
			const int qa = __builtin_amdgcn_readlane(pa, l), qb = __builtin_amdgcn_readlane(pb, l);
			double a = pan[SSK_LMA_PANEL * qa + l48];
			double b = pan[SSK_LMA_PANEL * qb + l48];
			a = real ? a : 0.0; b = real ? b : 0.0;
			c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
		}
	}
	S_tile[lane] = c[0]; S_tile[64 + lane] = c[1]; S_tile[128 + lane] = c[2]; S_tile[192 + lane] = c[3];
}

#ifdef SSK_LM_STAMPS
__device__ unsigned long long g_tst[8];
__device__ unsigned long long g_fine[8];      // chunk tasks of wavefront 0 / 4 of workgroup 0: lane-group items, segmented sum + stores, tasks
#define TSTAMP(i) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); g_tst[i] += t_ - tst_last; tst_last = t_; } } while (0)
#else
#define TSTAMP(i) do {} while (0)
#endif
// The lanes of one incidence (sub-lane s of ns, consecutive lanes of one wavefront) add their shares up into sub-lane 0: a fixed
// binary tree of wave shuffles (s + 2^k into s), executed by the whole wavefront.
__device__ __forceinline__ void incidence_combine(double (&Wm)[18], double (&ge)[6], int sub, int ns)
{
	int ns_max = 1;                                     // largest lane count of an incidence in this wavefront (uniform)
	for (int v = 8; v >= 2; --v) if (__ballot(ns >= v) != 0ull) { ns_max = v; break; }
	// the lanes of an incidence sit inside one 16-lane row (the tiler pads), so "the value d lanes up" is a DPP row shift -- a VALU move per
	// half -- not a ds_bpermute through the LDS pipeline (the combine was 4.2 k cycles of the incidence role's 12 k, more than its sums)
	if (ns_max > 1) {
		const bool take = (sub & 1) == 0 && sub + 1 < ns;
#pragma unroll
		for (int c = 0; c < 18; ++c) { const double v = dpp_shl_f64<1>(Wm[c]); Wm[c] += take ? v : 0.0; }
#pragma unroll
		for (int c = 0; c < 6; ++c) { const double v = dpp_shl_f64<1>(ge[c]); ge[c] += take ? v : 0.0; }
	}
	if (ns_max > 2) {
		const bool take = (sub & 3) == 0 && sub + 2 < ns;
#pragma unroll
		for (int c = 0; c < 18; ++c) { const double v = dpp_shl_f64<2>(Wm[c]); Wm[c] += take ? v : 0.0; }
#pragma unroll
		for (int c = 0; c < 6; ++c) { const double v = dpp_shl_f64<2>(ge[c]); ge[c] += take ? v : 0.0; }
	}
	if (ns_max > 4) {
		const bool take = (sub & 7) == 0 && sub + 4 < ns;
#pragma unroll
		for (int c = 0; c < 18; ++c) { const double v = dpp_shl_f64<4>(Wm[c]); Wm[c] += take ? v : 0.0; }
#pragma unroll
		for (int c = 0; c < 6; ++c) { const double v = dpp_shl_f64<4>(ge[c]); ge[c] += take ? v : 0.0; }
	}
}

// One tile of a window at the pose / landmark state the caller holds: phases I1-I3 of the file header.  `reg`: the
// workgroup's tile region in LDS; `tab`: the path-suffix table at that state;
// partial sums go to the window's slot arrays Pp / Sp, Y / gradient terms / landmark factors to its HBM scratch.
template <bool MFMA>
__device__ __forceinline__ void process_tile(const lm_dev& D, const lm_win& W, const lm_cam& P, const lm_tile& T, const double* tab_l, const double* __restrict__ gtab,
                                             double* reg, double lambda, bool pre, bool same_state,
                                             double* __restrict__ Pp, double* __restrict__ Sp, double& vmax, int* s_i2a, int& i2a_target)
{
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef SSK_LM_STAMPS
		unsigned long long tst_last = __builtin_amdgcn_s_memtime();
#endif
		// a window with many suffixes: this tile's part of the table (built in L2 scratch) comes into LDS first, ids in the records are
		// tile-local.  Either way `tab` is an LDS pointer (a pointer that might be global would turn every table read into a FLAT load)
		double* lt = reg;
		if (!W.tab_lds) {
			const int* ids = D.tsuf + W.tsuf0 + T.tsuf0;
			for (int i = tid; i < 12 * T.n_tsuf; i += NT) { const int k = i / 12; lt[i] = gtab[12 * (size_t)ids[k] + (i - 12 * k)]; }
			reg += 12 * T.n_tsuf;
			__syncthreads();
		}
		const double* tab = W.tab_lds ? tab_l : lt;
		double* ent = reg;
		double* obsr = ent + ENT_P * T.n_ent;
		double* Yr = obsr + OBS_P * T.n_obs;               // sparse: Y_P per incidence; matrix cores: SSK_LMA_PANEL per panel
		double* lmr = Yr + (MFMA ? SSK_LMA_PANEL * T.n_pan : Y_P * T.n_inc);
		// the incidence lane's two records are asked for before I1: the incidence phase is the critical path behind it and starts with them
		const bool inc_wave = !pre && (tid >> 6) < ((T.n_il + 63) >> 6);
		const int4 lrec_i = (inc_wave && tid < T.n_il) ? D.ilane[W.il0 + T.il0 + tid] : make_int4(0, 0, 0, 0);
		const int4 irec_i = inc_wave ? D.inc_rec[W.lme0 + T.inc0 + (lrec_i.x & 0xFFFF)] : make_int4(0, 0, 0, 0);
		// ---- I1: observations (the Y panels start from zero) ----
		if (MFMA && !pre) for (int i = tid; i < SSK_LMA_PANEL * T.n_pan; i += NT) Yr[i] = 0.0;
#ifdef SSK_LM_STAMPS
		const unsigned long long i1t0 = __builtin_amdgcn_s_memtime();
#endif
		for (int k = tid; k < T.n_obs; k += NT)
			linearize_obs_lds(D, W, P, tab, D.tobs_a[W.tobs0 + T.obs0 + k], D.tobs_px[W.tobs0 + T.obs0 + k], D.tent + W.tent0 + T.ent0, k, ent, obsr + OBS_P * k);
#ifdef SSK_LM_STAMPS
		if (threadIdx.x == 0 && blockIdx.x == 0) g_fine[3] += __builtin_amdgcn_s_memtime() - i1t0;      // wavefront 0's own I1 work (without the barrier)
#endif
		__syncthreads();
		TSTAMP(0);
		// After I1 the wavefronts have ROLES, not common phases:
		//   landmark wavefronts (the LAST ceil(n_lm / 64): 2 of 8 on the bench window)  I2a: V, g_f, Cholesky -> landmark records, then they arrive
		//                                                                              on a counter in LDS
		//   incidence wavefronts (the FIRST ceil(n_il / 64): 5)                        I2b: W and the gradient term of their incidence lanes from
		//                                                                              what I1 wrote; then they wait for the landmark records;
		//                                                                              then Y = W L^-T
		//   the wavefronts between them                                                 one H_pp chunk task each (H_pp reads only what I1 wrote)
		// and one workgroup barrier later the H_pp tasks left and the Schur tasks (they read Y) take all wavefronts.  The tiler sizes the lane groups
		// of the two task kinds for exactly this (ssk_capi_lm.hip).  A wavefront with both roles (a tile of many incidence lanes) does I2a first.
		// same_state: the previous pass linearised at this very state (the lambda0 pass, or a rejected step): its H_pp slots are still in place
		// and bit-identical, only the lambda-dependent Schur terms are new.
		const int n_pc = same_state ? 0 : T.n_pc;
		const int w_a = (T.n_lm + 63) >> 6, w_b = pre ? 0 : (T.n_il + 63) >> 6;
		const int wa0 = NW - w_a;                          // first landmark wavefront (n_lm <= NT: w_a <= NW)
		const int n_early = min(n_pc, max(wa0 - w_b, 0));
		if (!pre) i2a_target += w_a;                       // (uniform: every wavefront keeps the running count of I2a arrivals)
		int mask0 = 0, base0 = 0, mask1 = 0, base1 = 0;
		auto run_task = [&](int task) {
			if (MFMA && task >= T.n_pc) {
				// Schur tile (I, J) of this tile's landmarks on the matrix cores -> its 256-double partial in HBM (coalesced)
				const int2 mrec = D.mt[W.mt0 + T.mt0 + task - T.n_pc];
				schur_mfma_task<false>(Yr, mask0, base0, mask1, base1, mrec.x & 255, mrec.x >> 8, lane, D.Stile + 256 * (size_t)(W.st0 + mrec.y));
				return;
			}
			double acc[36];
#pragma unroll
			for (int i = 0; i < 36; ++i) acc[i] = 0.0;
#ifdef SSK_LM_STAMPS
			const unsigned long long ft0 = __builtin_amdgcn_s_memtime();
			unsigned long long ft1;
#endif
			if (task < T.n_pc) {
				const int g = 64 * task + lane;
				const int2 rec = D.pgrp[W.pg0 + T.pg0 + (g < T.n_pg ? g : T.n_pg - 1)];
				pp_group(ent, obsr, tab, D.pit + W.pit0 + rec.x, g < T.n_pg ? (rec.y & 255) : 0, acc);
#ifdef SSK_LM_STAMPS
				ft1 = __builtin_amdgcn_s_memtime();
#endif
				wave_seg_sum36(acc, rec.y >> 8, lane, Pp, D.pcl_idx + W.pcl0 + D.pslot0[W.pcx0 + T.pc0 + task]);
			} else {
				const int c = task - T.n_pc, g = 64 * c + lane;
				const int2 rec = D.sgrp[W.sg0 + T.sg0 + (g < T.n_sg ? g : T.n_sg - 1)];
				schur_group(Yr, D.qit + W.qit0 + rec.x, g < T.n_sg ? (rec.y & 255) : 0, acc);
#ifdef SSK_LM_STAMPS
				ft1 = __builtin_amdgcn_s_memtime();
#endif
				wave_seg_sum36(acc, rec.y >> 8, lane, Sp, D.scl_idx + W.scl0 + D.sslot0[W.scx0 + T.sc0 + c]);
			}
#ifdef SSK_LM_STAMPS
			if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4)) {
				const unsigned long long ft2 = __builtin_amdgcn_s_memtime();
				g_fine[(wave == 0 ? 0 : 4) + 0] += ft1 - ft0; g_fine[(wave == 0 ? 0 : 4) + 1] += ft2 - ft1; g_fine[(wave == 0 ? 0 : 4) + 2] += 1;
			}
#endif
		};
		if (wave >= wa0) {
			// ---- I2a: a lane per landmark: V, g_f, Cholesky -> lmr (+ HBM for the back-substitution) ----
			for (int j = tid - 64 * wa0; j < T.n_lm; j += NT) {
				const int4 lrec = D.tl_rec[W.tl0 + T.lml0 + j];   // (slot, first local observation, end)
				const int slot = lrec.x;
				if (slot < 0) continue;
				double v[9];
				landmark_sums(obsr, lrec.y, lrec.z, v);
				if (pre) { vmax = fmax(vmax, fmax(v[0], fmax(v[3], v[5]))); continue; }
				if (D.store_hff) {                                 // only the shared-landmark exchange reads the landmark blocks
					double* Hm = D.Hff + 9 * (size_t)(W.ls0 + slot);
					Hm[0] = v[0]; Hm[1] = v[1]; Hm[2] = v[2]; Hm[3] = v[1]; Hm[4] = v[3]; Hm[5] = v[4]; Hm[6] = v[2]; Hm[7] = v[4]; Hm[8] = v[5];
					double* gl = D.gf + 3 * (size_t)(W.ls0 + slot);
					gl[0] = v[6]; gl[1] = v[7]; gl[2] = v[8];
				}
				double rec[9];
				landmark_factor(v, lambda, rec);
				double* lr = lmr + LMR_P * j;
				double* lg = D.Vinv + 9 * (size_t)(W.ls0 + slot);
	#pragma unroll
				for (int c = 0; c < 9; ++c) { lr[c] = rec[c]; lg[c] = rec[c]; }
			}
			if (!pre) {                                        // this wavefront's landmark records are in LDS: arrive
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
				if (lane == 0) __hip_atomic_fetch_add(s_i2a, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
		}
		if (wave < w_b) {
			// ---- I2b: a lane per incidence lane (n_il <= NT: one each; whole wavefronts, the lanes of an incidence meet in shuffles) ----
			const int4 lrec = lrec_i;
			const int ns = (lrec.x >> 20) & 15, sub = (lrec.x >> 16) & 15, m = lrec.x & 0xFFFF;
			const int4 irec = irec_i;                          // (local landmark | optimised << 15 | #entries << 16, first lmep, edge-major slot, first entry)
			const int lml = irec.x & 0x7FFF;
			const bool unk = (irec.x >> 15) & 1;
			double Wm[18], ge[6];
#ifdef SSK_LM_STAMPS
			const unsigned long long rt0 = __builtin_amdgcn_s_memtime();
#endif
			incidence_sums(D, W, ent, obsr, tab, lrec, unk, Wm, ge);
#ifdef SSK_LM_STAMPS
			const unsigned long long rt1 = __builtin_amdgcn_s_memtime();
#endif
			incidence_combine(Wm, ge, sub, ns);
#ifdef SSK_LM_STAMPS
			const unsigned long long rt2 = __builtin_amdgcn_s_memtime();
#endif
			// the landmark records: wait for the w_a arrivals of this tile pass
			while (__hip_atomic_load(s_i2a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < i2a_target) __builtin_amdgcn_s_sleep(1);
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#ifdef SSK_LM_STAMPS
			if (threadIdx.x == 0 && blockIdx.x == 0) { const unsigned long long rt3 = __builtin_amdgcn_s_memtime(); g_tst[4] += rt1 - rt0; g_tst[6] += rt2 - rt1; g_tst[7] += rt3 - rt2; }
#endif
			if (ns != 0 && sub == 0) {                         // the incidence's first lane carries on with the sums
				double* gr = D.gre + 6 * (size_t)(W.el0 + irec.z);
				if (unk) {
					const double* lr = lmr + LMR_P * lml;
					const double i00 = lr[0], i10 = lr[1], i11 = lr[2], i20 = lr[3], i21 = lr[4], i22 = lr[5], c0 = lr[6], c1 = lr[7], c2 = lr[8];
					double* yl = Yr + Y_P * m;
					double* yg = D.Tb + 18 * (size_t)(W.lme0 + T.inc0 + m);
					int mk = 0, bs = 0, e_slot = 0;
					if (MFMA) { const int2 tp = D.tl_pan[W.tl0 + T.lml0 + lml]; mk = tp.x; bs = tp.y; e_slot = D.lme_edge[W.lme0 + T.inc0 + m]; }
#pragma unroll
					for (int a = 0; a < 6; ++a) {
						const double y0 = Wm[3 * a] * i00;
						const double y1 = Wm[3 * a] * i10 + Wm[3 * a + 1] * i11;
						const double y2 = Wm[3 * a] * i20 + Wm[3 * a + 1] * i21 + Wm[3 * a + 2] * i22;
						if (MFMA) {                                    // rows 6e .. 6e+5 of the landmark's panels, and HBM for the back-substitution
							const int r = 6 * e_slot + a, rt = r >> 4;
							double* pp = Yr + SSK_LMA_PANEL * (bs + __popc(mk & ((1 << rt) - 1))) + (r & 15);
							pp[0] = y0; pp[16] = y1; pp[32] = y2;
							yg[3 * a] = y0; yg[3 * a + 1] = y1; yg[3 * a + 2] = y2;
						} else {
							yl[3 * a] = y0; yl[3 * a + 1] = y1; yl[3 * a + 2] = y2;
						}
						gr[a] = ge[a] - (y0 * c0 + y1 * c1 + y2 * c2);
					}
				} else {
#pragma unroll
					for (int a = 0; a < 6; ++a) gr[a] = ge[a];
				}
			}
		} else if (wave < wa0) {
			if (wave - w_b < n_early) run_task(wave - w_b);
		}
		__syncthreads();
		TSTAMP(1);
		// ---- I3: Y to HBM for the back-substitution (coalesced), then one chunk task per wave and round: the H_pp chunks the idle wavefronts
		//      did not take, then the Schur chunks (they read the Y records) ----
		if (!pre && !MFMA) {
			double* yg = D.Tb + 18 * (size_t)(W.lme0 + T.inc0);
			for (int i = tid; i < 18 * T.n_inc; i += NT) { const int m = i / 18; yg[i] = Yr[m * Y_P + (i - 18 * m)]; }
		}
		if (MFMA && !pre) {                                  // this lane's landmark(s) of the tile: row-tile mask and first panel
			if (lane < T.n_lm) { const int2 tp = D.tl_pan[W.tl0 + T.lml0 + lane]; mask0 = tp.x; base0 = tp.y; }
			if (lane + 64 < T.n_lm) { const int2 tp = D.tl_pan[W.tl0 + T.lml0 + lane + 64]; mask1 = tp.x; base1 = tp.y; }
		}
		const int n_task = T.n_pc + (pre ? 0 : (MFMA ? T.n_mt : T.n_sc));
		for (int task = n_early + (same_state ? T.n_pc : 0) + wave; task < n_task; task += NW) run_task(task);
		__syncthreads();
		TSTAMP(3);
}

// ================= on-chip accumulation (lm_solve_acc_kernel) =================
// Windows with few optimised edges (E <= SSK_LMA_MAX_E): nothing but Y leaves the chip.  LDS holds, beside the tile records,
//   P_acc  36 doubles per block pair    H_pp blocks, summed over the tiles (kept across a rejected step: same state, same blocks)
//   S_acc  256 doubles per upper 16x16 tile of the reduced system (matrix-core form), or 36 per block pair (sparse form)
//   g_acc  6E doubles                   reduced gradient
// H_pp / sparse Schur: the chunk tasks of the slot kernels, but the last lane of a block's run ADDS the run's sum to the block's
// accumulator.  The host gives every block of a tile to one wave and one chunk, and runs cut by the 16-lane DPP rows are added
// row by row, so an accumulator never sees two adds of one instruction and the order of its adds is fixed: deterministic.
// Matrix-core Schur: S = sum_l Y_l Y_l^T with Y_l (6E x 3) the stacked Y blocks of landmark l.  Y is written as PANELS -- per
// (landmark, 16-row tile it touches) a [3][16] block, zero where the landmark has no incidence -- and a task (I, J) runs one
// v_mfma_f64_16x16x4_f64 per landmark that touches both row tiles: A = panel I, B = panel J (k = 3 columns of Y + one zero
// lane group), the 16x16 accumulator lives in 4 VGPR pairs for the whole task and goes back to LDS once.
// One tile with on-chip accumulation.  Region: [entries | observation records | Y (records or panels) | landmark records |
// gradient terms (7 per incidence: 6 values + edge slot)].
template <bool MFMA>
__device__ __forceinline__ void process_tile_acc(const lm_dev& D, const lm_win& W, const lm_cam& P, const lm_tile& T, const double* tab_l, const double* __restrict__ gtab,
                                                 double* reg, double lambda, bool pre, bool same_state,
                                                 double* __restrict__ P_acc, double* __restrict__ S_acc, double* __restrict__ g_acc, double& vmax)
{
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	double* lt = reg;                                      // (as process_tile)
	if (!W.tab_lds) {
		const int* ids = D.tsuf + W.tsuf0 + T.tsuf0;
		for (int i = tid; i < 12 * T.n_tsuf; i += NT) { const int k = i / 12; lt[i] = gtab[12 * (size_t)ids[k] + (i - 12 * k)]; }
		reg += 12 * T.n_tsuf;
		__syncthreads();
	}
	const double* tab = W.tab_lds ? tab_l : lt;
	double* ent = reg;
	double* obsr = ent + ENT_P * T.n_ent;
	double* Yr = obsr + OBS_P * T.n_obs;                   // sparse: Y_P per incidence; MFMA: SSK_LMA_PANEL per panel
	double* lmr = Yr + (MFMA ? SSK_LMA_PANEL * T.n_pan : Y_P * T.n_inc);
	double* gsc = lmr + LMR_P * T.n_lm;                    // 7 per incidence
	const int n6 = 6 * W.E;
#ifdef SSK_LM_STAMPS
	unsigned long long tst_last = __builtin_amdgcn_s_memtime();
#endif
	// ---- I1: observations; the Y panels start from zero ----
	if (MFMA && !pre) for (int i = tid; i < SSK_LMA_PANEL * T.n_pan; i += NT) Yr[i] = 0.0;
	for (int k = tid; k < T.n_obs; k += NT)
		linearize_obs_lds(D, W, P, tab, D.tobs_a[W.tobs0 + T.obs0 + k], D.tobs_px[W.tobs0 + T.obs0 + k], D.tent + W.tent0 + T.ent0, k, ent, obsr + OBS_P * k);
	__syncthreads();
	TSTAMP(0);
	// ---- I2a: a lane per landmark ----
	for (int j = tid; j < T.n_lm; j += NT) {
		const int4 lrec = D.tl_rec[W.tl0 + T.lml0 + j];
		const int slot = lrec.x;
		if (slot < 0) continue;
		double v[9];
		landmark_sums(obsr, lrec.y, lrec.z, v);
		if (pre) { vmax = fmax(vmax, fmax(v[0], fmax(v[3], v[5]))); continue; }
		if (D.store_hff) {
			double* Hm = D.Hff + 9 * (size_t)(W.ls0 + slot);
			Hm[0] = v[0]; Hm[1] = v[1]; Hm[2] = v[2]; Hm[3] = v[1]; Hm[4] = v[3]; Hm[5] = v[4]; Hm[6] = v[2]; Hm[7] = v[4]; Hm[8] = v[5];
			double* gl = D.gf + 3 * (size_t)(W.ls0 + slot);
			gl[0] = v[6]; gl[1] = v[7]; gl[2] = v[8];
		}
		double rec[9];
		landmark_factor(v, lambda, rec);
		double* lr = lmr + LMR_P * j;
		double* lg = D.Vinv + 9 * (size_t)(W.ls0 + slot);
#pragma unroll
		for (int c = 0; c < 9; ++c) { lr[c] = rec[c]; lg[c] = rec[c]; }
	}
	if (!pre) __syncthreads();
	// ---- I2b: a lane per incidence: W, gradient term; Y = W L^-T -> LDS (records or panels) and HBM (back-substitution) ----
	if (!pre)
	for (int il = tid; il < ((T.n_il + 63) & ~63); il += NT) {
		const int4 lrec = il < T.n_il ? D.ilane[W.il0 + T.il0 + il] : make_int4(0, 0, 0, 0);
		const int ns = (lrec.x >> 20) & 15, sub = (lrec.x >> 16) & 15, m = lrec.x & 0xFFFF;
		const int4 irec = D.inc_rec[W.lme0 + T.inc0 + m];
		const int lml = irec.x & 0x7FFF;
		const bool unk = (irec.x >> 15) & 1;
		const int e_slot = D.lme_edge[W.lme0 + T.inc0 + m];
		double Wm[18], ge[6];
		incidence_sums(D, W, ent, obsr, tab, lrec, unk, Wm, ge);
		incidence_combine(Wm, ge, sub, ns);
		if (ns == 0 || sub != 0) continue;
		double* gr = gsc + 7 * m;
		gr[6] = __hiloint2double(0, e_slot);
		if (unk) {
			const double* lr = lmr + LMR_P * lml;
			const double i00 = lr[0], i10 = lr[1], i11 = lr[2], i20 = lr[3], i21 = lr[4], i22 = lr[5], c0 = lr[6], c1 = lr[7], c2 = lr[8];
			double* yg = D.Tb + 18 * (size_t)(W.lme0 + T.inc0 + m);
			int mk = 0, bs = 0;
			if (MFMA) { const int2 tp = D.tl_pan[W.tl0 + T.lml0 + lml]; mk = tp.x; bs = tp.y; }
#pragma unroll
			for (int a = 0; a < 6; ++a) {
				const double y0 = Wm[3 * a] * i00;
				const double y1 = Wm[3 * a] * i10 + Wm[3 * a + 1] * i11;
				const double y2 = Wm[3 * a] * i20 + Wm[3 * a + 1] * i21 + Wm[3 * a + 2] * i22;
				gr[a] = ge[a] - (y0 * c0 + y1 * c1 + y2 * c2);
				if (MFMA) {
					const int r = 6 * e_slot + a, rt = r >> 4;
					double* pp = Yr + SSK_LMA_PANEL * (bs + __popc(mk & ((1 << rt) - 1))) + (r & 15);
					pp[0] = y0; pp[16] = y1; pp[32] = y2;
					yg[3 * a] = y0; yg[3 * a + 1] = y1; yg[3 * a + 2] = y2;
				} else {
					double* yl = Yr + Y_P * m;
					yl[3 * a] = y0; yl[3 * a + 1] = y1; yl[3 * a + 2] = y2;
				}
			}
		} else {
#pragma unroll
			for (int a = 0; a < 6; ++a) gr[a] = ge[a];
		}
	}
	__syncthreads();
	TSTAMP(1);
	// ---- I3 ----
	if (!pre) {
		if (!MFMA) {                                       // Y to HBM for the back-substitution (coalesced)
			double* yg = D.Tb + 18 * (size_t)(W.lme0 + T.inc0);
			for (int i = tid; i < 18 * T.n_inc; i += NT) { const int m = i / 18; yg[i] = Yr[m * Y_P + (i - 18 * m)]; }
		}
		// reduced gradient: component (e, a) by 8 lanes that walk the tile's incidences in order and take those of edge e
		for (int it = tid; it < n6 * 8; it += NT) {
			const int comp = it >> 3, j = it & 7, e = comp / 6, a = comp - e * 6;
			double sacc = 0.0;
			for (int k = j; k < T.n_inc; k += 8) {
				const double* gr = gsc + 7 * k;
				const double v = gr[a];
				sacc += __double2loint(gr[6]) == e ? v : 0.0;
			}
			sacc += __shfl_down(sacc, 4, 8); sacc += __shfl_down(sacc, 2, 8); sacc += __shfl_down(sacc, 1, 8);
			if (j == 0) g_acc[comp] += sacc;
		}
	}
	TSTAMP(2);
	{
		const int* wo = D.wt_off + W.wto0 + T.wt0;
		const int k0 = wo[wave], k1 = wo[wave + 1];
		int mask0 = 0, base0 = 0, mask1 = 0, base1 = 0;
		if (MFMA && !pre) {
			if (lane < T.n_lm) { const int2 tp = D.tl_pan[W.tl0 + T.lml0 + lane]; mask0 = tp.x; base0 = tp.y; }
			if (lane + 64 < T.n_lm) { const int2 tp = D.tl_pan[W.tl0 + T.lml0 + lane + 64]; mask1 = tp.x; base1 = tp.y; }
		}
		for (int k = k0; k < k1; ++k) {
			const int code = D.wt[W.wt0 + k], kind = code >> 28;
			if (kind == 0) {
				if (same_state) continue;                      // the previous pass linearised at this very state: its H_pp blocks stand
				const int g = 64 * (code & 0xFFFFFF) + lane;
				const int2 rec = D.pgrp[W.pg0 + T.pg0 + (g < T.n_pg ? g : T.n_pg - 1)];
				double acc[36];
#pragma unroll
				for (int i = 0; i < 36; ++i) acc[i] = 0.0;
				pp_group(ent, obsr, tab, D.pit + W.pit0 + rec.x, g < T.n_pg ? (rec.y & 255) : 0, acc);
				wave_seg_add36(acc, rec.y >> 8, lane, P_acc);
			} else if (pre) {
				continue;
			} else if (kind == 1) {
				const int g = 64 * (code & 0xFFFFFF) + lane;
				const int2 rec = D.sgrp[W.sg0 + T.sg0 + (g < T.n_sg ? g : T.n_sg - 1)];
				double acc[36];
#pragma unroll
				for (int i = 0; i < 36; ++i) acc[i] = 0.0;
				schur_group(Yr, D.qit + W.qit0 + rec.x, g < T.n_sg ? (rec.y & 255) : 0, acc);
				wave_seg_add36(acc, rec.y >> 8, lane, S_acc);
			} else if (MFMA) {
				const int I = code & 255, J = (code >> 8) & 255;
				schur_mfma_task<true>(Yr, mask0, base0, mask1, base1, I, J, lane, S_acc + 256 * (J * (J + 1) / 2 + I));
			}
		}
#ifdef SSK_LM_STAMPS
		if (threadIdx.x == 0 && blockIdx.x == 0) g_tst[4] += __builtin_amdgcn_s_memtime() - tst_last;      // wave 0's own tasks
#endif
	}
	__syncthreads();
	TSTAMP(3);
}

// F2: H_red block (e,e') = sum of its H_pp slots + lambda I - sum of its Schur slots (slot order), diagonal blocks
// written from their upper triangle; g_red(e) = sum of its incidences' terms (contiguous, edge-major, 8 lanes per
// component) into the border row; G: blocked LDL^T + back-substitution (ssk_lm_device.h).  A: (n6 + 1) x np.
#ifdef SSK_LM_STAMPS
__device__ unsigned long long g_f2g[4];
#define F2G_TICK(i) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); g_f2g[i] += t_ - f2g_last; f2g_last = t_; } } while (0)
#else
#define F2G_TICK(i) do {} while (0)
#endif
// `solve` = false: only the share (first, step) of the assembly (cooperative kernel: the workgroups of a window assemble the
// L2-resident system together, workgroup 0 then factorises it with `assemble` = false)
template <bool GLOBAL_A, bool SMF = false, bool LDLM = false, typename PtrA>
__device__ __forceinline__ bool assemble_and_solve(const lm_dev& D, const lm_win& W, const double* __restrict__ Pp, const double* __restrict__ Sp,
                                                   double lambda, PtrA A, int np, int n6, double* s_b, double* s_inv,
                                                   bool assemble = true, bool solve = true, int first = threadIdx.x, int step = NT,
                                                   double* lds_free = nullptr, int lds_free_doubles = 0)
{
#ifdef SSK_LM_STAMPS
	unsigned long long f2g_last = __builtin_amdgcn_s_memtime();
#endif
	if (assemble) {
	for (int it = first; it < W.n_pairs * 36; it += step) {
		const int pid = it / 36, ab = it - pid * 36, a = ab / 6, b = ab - a * 6;
		const int2 ee = D.pair_ee[W.pair0 + pid];
		if (ee.x == ee.y && a > b) continue;
		// (empty runs: p1 == p0 reads slot p0 - clamped inside the window's arrays - and adds nothing)
		double sres;
		const int p0 = D.pcl_off[W.pairo0 + pid], p1 = D.pcl_off[W.pairo0 + pid + 1];
		if (SMF) {                                         // matrix-core Schur: the partials of the 16x16 tile that holds (row, col), tile order
			const int row = 6 * ee.x + a, col = 6 * ee.y + b, u = (col >> 4) * ((col >> 4) + 1) / 2 + (row >> 4);
			sres = slot_diff<256>(Pp, p0, p1, ab, D.Stile + 256 * (size_t)W.st0, D.st_off[W.sto0 + u], D.st_off[W.sto0 + u + 1],
			                      (((row & 15) >> 2) << 6) + ((row & 3) << 4) + (col & 15));
		} else sres = slot_diff<36>(Pp, p0, p1, ab, Sp, D.scl_off[W.pairo0 + pid], D.scl_off[W.pairo0 + pid + 1], ab);
		if (ee.x == ee.y && a == b) sres += lambda;
		A[(6 * ee.x + a) * np + 6 * ee.y + b] = sres;
		A[(6 * ee.y + b) * np + 6 * ee.x + a] = sres;
	}
	F2G_TICK(0);
	for (int it = first; it < n6 * 8; it += step) {
		const int comp = it >> 3, j = it & 7, e = comp / 6, a = comp - e * 6;
		const int k0 = D.el_off[W.uso0 + e], k1 = D.el_off[W.uso0 + e + 1];
		const double* src = D.gre + 6 * (size_t)W.el0 + a;
		// s0 takes the elements k0 + j + 16 m, s1 the elements k0 + j + 8 + 16 m (below k1), each in ascending order; four
		// unconditional loads on clamped indices in flight per turn, the padding selected to +0.0 afterwards
		double s0 = 0, s1 = 0;
		for (int k = k0 + j; k < k1; k += 32) {
			double v[4];
#pragma unroll
			for (int u = 0; u < 4; ++u) v[u] = src[6 * (size_t)min(k + 8 * u, k1 - 1)];
			s0 += v[0];
			s1 += k + 8 < k1 ? v[1] : 0.0;
			s0 += k + 16 < k1 ? v[2] : 0.0;
			s1 += k + 24 < k1 ? v[3] : 0.0;
		}
		double v = s0 + s1;
		v += __shfl_down(v, 4, 8); v += __shfl_down(v, 2, 8); v += __shfl_down(v, 1, 8);
		if (j == 0) A[n6 * np + comp] = v;
	}
	}
	__syncthreads();
	F2G_TICK(1);
	if (!solve) return true;
	bool ok_ = true;
	if (n6 > 0) {
		if constexpr (GLOBAL_A) {
			// LDLM (kernels lm_solve_*_ldlm_kernel, SSK_LM_LDL=mfma): 16-column panels with the trailing update on the matrix cores
			if constexpr (LDLM) ok_ = ldl_solve_mfma16(A, np, n6, s_b, s_inv, lds_free, lds_free_doubles);
			else ok_ = SSK_LDL_BIG(A, np, n6, s_b, s_inv, lds_free, lds_free_doubles);
		}
		else ok_ = ldl_solve_lds(A, np, n6, s_b, s_inv);
	}
	F2G_TICK(2);
	return ok_;
}

// ACC: 0 = partial slots in HBM, 1 = on-chip accumulation with sparse Schur items, 2 = on-chip accumulation with the Schur
// complement on the matrix cores
template <bool COOP, int ACC = 0, bool SMF = false, bool LDLM = false, bool LDSYS = false>
__device__ __forceinline__ void lm_tiled_body(const lm_dev& D, const lm_cam& P_arg, int dyn_doubles, int G, int coop_expect = 1, int coop_spin_limit = COOP_SPIN_LIMIT)
{
	static_assert(!(COOP && ACC), "the accumulators of a window live in one workgroup's LDS");
	// the camera / schedule parameters (29 doubles) from LDS, not from 60 scalar registers held for the whole kernel: the kernel's ~45 array
	// bases and the window header already overflow the scalar file (spilled scalars come back through v_readlane, a VALU slot each)
	__shared__ lm_cam s_P;
	if (threadIdx.x == 0) s_P = P_arg;
	__syncthreads();
	const lm_cam& P = s_P;
	extern __shared__ __attribute__((aligned(16))) double s_dyn[];   // [pose_cur | pose_new | delta | pivots | tile records / reduced system]
	__shared__ double s_part[NW];
	__shared__ double s_lambda, s_nu, s_err;
	__shared__ int s_accept, s_stop, s_iters, s_accepted;
	__shared__ int s_same;                             // the state has not moved since the previous pass's linearisation
	__shared__ int s_tabcur;                           // the suffix table in place is the current state's (an accepted step's scoring left exactly that)
	__shared__ int s_i2a;                              // arrivals of the landmark wavefronts (I2a) of every tile pass so far: process_tile
	int i2a_target = 0;

	const int wi = COOP ? (int)blockIdx.x / G : (int)blockIdx.x, g = COOP ? (int)blockIdx.x % G : 0, GG = COOP ? G : 1;
	const lm_win W = D.win[wi];
	const int tid = threadIdx.x;
	const int n6 = 6 * W.E, n6p = (n6 + 1) & ~1;
	coop_t co = {COOP ? D.coop_i + 4 * wi : nullptr, COOP ? D.coop_i + 4 * wi + 1 : nullptr, COOP ? D.coop_f + 128 * (size_t)wi : nullptr, g, GG, 0, 0, coop_expect, coop_spin_limit};
	bool alive = true;
	(void)alive;
	double* pose_cur = s_dyn;
	double* pose_new = s_dyn + 12 * W.n_edges;
	double* s_b = s_dyn + 24 * W.n_edges;               // right-hand side / solution of the reduced system (n6)
	double* s_inv = s_b + n6p;                         // reciprocal pivots of the LDS solve (n6)
	// on-chip accumulators (ACC): H_pp blocks, Schur tiles / blocks, reduced gradient
	const int n_accP = ACC ? 36 * W.n_pairs : 0, RT = (n6 + 15) >> 4, n_accS = ACC ? 256 * (RT * (RT + 1) / 2) : 0;
	double* P_acc = s_inv + n6p;
	double* S_acc = P_acc + n_accP;
	double* g_acc = S_acc + n_accS;
	// path-suffix table: 12 doubles per distinct suffix, rebuilt for the state of every pass -- in LDS, or (many suffixes) in the window's
	// L2-resident scratch, one copy per workgroup
	double* tab_l = ACC ? g_acc + n6p : s_inv + n6p;
	int2* recs_l = reinterpret_cast<int2*>(tab_l + (W.tab_lds ? 12 * W.n_suf : 0));
	double* reg = tab_l + (W.tab_lds ? 13 * W.n_suf : 0);
	double* gtab = D.suf_tab + W.gtab0 + (W.tab_lds ? 0 : 12 * (size_t)W.n_suf * g);
	// the suffix table at a state.  The kernel variant of the batches whose systems all fit LDS (LDSYS: the batched per-keyframe windows, few
	// suffixes of few steps) builds an LDS table by a chain walk per suffix: one barrier instead of one per level.  The other variants keep the
	// level scheme and do not even contain the walk: with a few hundred suffixes of up to eight steps it does not pay, and its mere presence
	// moved their register allocation (+4 % on a 40-edge local area)
	auto build_table = [&](const double* pose) {
		if constexpr (LDSYS) {
			if (W.tab_lds) build_suffix_table_walk(W, pose, tab_l, recs_l); else build_suffix_table(W, pose, gtab, D.suf_rec + W.suf0);
		} else {
			if (W.tab_lds) build_suffix_table(W, pose, tab_l, recs_l); else build_suffix_table(W, pose, gtab, D.suf_rec + W.suf0);
		}
	};
	// the reduced system lives over the dead tile records when (n6 + 1) x (n6 | 1) doubles fit there (E <= ~22),
	// otherwise in the window's L2-resident scratch (same blocked factorisation through plain pointers)
	// (LDSYS: the host found that every window's system fits: the L2-resident solver is compiled out)
	const bool lds_mat = LDSYS || (long long)(n6 + 1) * (n6 | 1) <= (long long)dyn_doubles - (reg - s_dyn);
	const int np = n6 | 1;
#ifdef SSK_LM_STAMPS
	__shared__ unsigned long long s_acc[16];
	unsigned long long t_last = __builtin_amdgcn_s_memtime();
	if (tid < 16) s_acc[tid] = 0;
#define STAMP(i) do { __syncthreads(); if (tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); s_acc[i] += t_ - t_last; t_last = t_; } } while (0)
	__shared__ unsigned long long s_fine[16];
	unsigned long long f_last = 0;
	if (tid < 16) s_fine[tid] = 0;
#define FTICK0() do { if (tid == 0 || tid == 256) f_last = __builtin_amdgcn_s_memtime(); } while (0)
#define FTICK(i) do { if (tid == 0 || tid == 256) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); s_fine[(tid == 0 ? 0 : 8) + i] += t_ - f_last; f_last = t_; } } while (0)
#else
#define STAMP(i) do {} while (0)
#define FTICK0() do {} while (0)
#define FTICK(i) do {} while (0)
#endif

	for (int i = tid; i < 12 * W.n_edges; i += NT) {
		const double v = D.pose[12 * (size_t)W.edge0 + i];
		pose_cur[i] = v; pose_new[i] = v;
	}
	for (int i = tid + NT * g; i < 3 * W.n_lms; i += NT * GG) D.lm_new[3 * (size_t)W.lm0 + i] = D.lm[3 * (size_t)W.lm0 + i];
	__syncthreads();
	if (W.tab_lds) for (int i = tid; i < W.n_suf; i += NT) recs_l[i] = D.suf_rec[W.suf0 + i];      // (the build's first barrier covers it)
	build_table(pose_cur);
	double e0 = W.tab_lds ? total_error_packed(D, W, P, tab_l, D.lm, s_part, tid + NT * g, NT * GG) : total_error_packed(D, W, P, gtab, D.lm, s_part, tid + NT * g, NT * GG);
	if (COOP) e0 = coop_combine<false>(co, e0, alive);
	if (tid == 0) {
		s_lambda = -1.0; s_nu = 2.0; s_err = e0; s_iters = 0; s_accepted = 0; s_stop = 0; s_same = 0; s_tabcur = 1; s_i2a = 0;
		if (g == 0) {
			ssk_lm_result& R = D.res[wi];
			R.err_init = e0; R.n_obs = W.n_obs; R.stop_reason = (W.E == 0 && W.L == 0) ? 3 : 0;
		}
	}
	__syncthreads();
	if (W.E == 0 && W.L == 0) {
		if (tid == 0 && g == 0) { ssk_lm_result& R = D.res[wi]; R.iters = 0; R.accepted = 0; R.err_final = e0; R.lambda_final = -1.0; }
		return;
	}
	double* const Pp = D.Ppart2 + 36 * (size_t)W.ps0;
	double* const Sp = D.Spart2 + 36 * (size_t)W.ss0;

	while (true) {
		__syncthreads();
		if (COOP && !alive) { if (tid == 0) s_stop = 5; break; }
		if (s_iters >= P.max_iters) break;
		if (W.n_obs > 0 && sqrt(s_err / W.n_obs) < P.stop_rmse) { if (tid == 0) s_stop = 1; break; }
		const double lambda = s_lambda;
		const bool pre = lambda < 0;                   // first pass of a solve: only the diagonals, for lambda0
		const bool same_state = !pre && s_same != 0;   // (uniform)
		const int it_idx = s_iters;
		double vmax = 0.0;
		// the scoring of a REJECTED attempt left the trial state's table here; an accepted one's is the new current state's own
		if (!s_tabcur) {                                   // (uniform)
			build_table(pose_cur);
			if (tid == 0) s_tabcur = 1;
		}
		STAMP(0);
		if (ACC) {                                     // the accumulators of this pass start from zero (H_pp: unless the state has not moved)
			for (int i = tid; i < n_accS + n6p; i += NT) S_acc[i] = 0.0;
			if (!same_state) for (int i = tid; i < n_accP; i += NT) P_acc[i] = 0.0;
			__syncthreads();
		}
		for (int t = g; t < W.n_tiles; t += GG) {
			if (ACC) process_tile_acc<ACC == 2>(D, W, P, D.tile[W.tile0 + t], tab_l, gtab, reg, lambda, pre, same_state, P_acc, S_acc, g_acc, vmax);
			else process_tile<SMF>(D, W, P, D.tile[W.tile0 + t], tab_l, gtab, reg, lambda, pre, same_state, Pp, Sp, vmax, &s_i2a, i2a_target);
			STAMP(1);
		}
		if (COOP) {                                    // every workgroup's slots, Y, gradient terms and landmark factors are out
			if (pre) vmax = coop_combine<true>(co, block_max(vmax, s_part), alive);
			else alive = coop_barrier(co);
			if (!alive) continue;
		}
		if (pre) {
			// lambda0 = tau * max diag(H): the diagonal blocks' slot sums, and the landmark blocks' diagonals
			double mx = vmax;
			for (int it = tid; it < n6; it += NT) {
				const int e = it / 6, a = it - 6 * e, pid = e * (e + 1) / 2 + e;
				mx = fmax(mx, ACC ? P_acc[36 * pid + 7 * a] : slot_sum(Pp, D.pcl_off[W.pairo0 + pid], D.pcl_off[W.pairo0 + pid + 1], 7 * a));
			}
			mx = block_max(mx, s_part);
			if (tid == 0) { s_lambda = P.tau * mx; s_same = 1; }
			continue;
		}
		// ---- F2 + G: assemble and solve the reduced system; instantiated once for the LDS copy and once for the L2 copy so
		//      that each uses the instructions of its address space (a pointer that may be either compiles to FLAT accesses) ----
		bool ok = true;
		const bool share_f2 = COOP && !lds_mat;         // an L2-resident system is assembled by all the workgroups of the window
		if (share_f2) {
			assemble_and_solve<true, SMF>(D, W, Pp, Sp, lambda, D.Hred + W.h0, np, n6, s_b, s_inv, true, false, tid + NT * g, NT * GG);
			alive = coop_barrier(co);
			if (!alive) continue;
		}
		STAMP(5);
		if (ACC) {
			// the reduced system from the accumulators, all in LDS: A = H_pp - S + lambda I over the dead tile records, border row = gradient
			for (int it = tid; it < W.n_pairs * 36; it += NT) {
				const int pid = it / 36, ab = it - pid * 36, a = ab / 6, b = ab - a * 6;
				const int2 ee = D.pair_ee[W.pair0 + pid];
				if (ee.x == ee.y && a > b) continue;
				const int row = 6 * ee.x + a, col = 6 * ee.y + b;
				const double sc = ACC == 2 ? S_acc[256 * ((col >> 4) * ((col >> 4) + 1) / 2 + (row >> 4)) + (((row & 15) >> 2) << 6) + ((row & 3) << 4) + (col & 15)]
				                           : S_acc[it];
				double sres = P_acc[it] - sc;
				if (row == col) sres += lambda;
				reg[row * np + col] = sres;
				reg[col * np + row] = sres;
			}
			for (int i = tid; i < n6; i += NT) reg[n6 * np + i] = g_acc[i];
			__syncthreads();
			if (n6 > 0) ok = ldl_solve_lds(reg, np, n6, s_b, s_inv);
		} else if (g == 0) {
			ok = lds_mat ? assemble_and_solve<false, SMF>(D, W, Pp, Sp, lambda, reg, np, n6, s_b, s_inv)
			             : assemble_and_solve<true, SMF, LDLM>(D, W, Pp, Sp, lambda, D.Hred + W.h0, np, n6, s_b, s_inv, !share_f2, true, tid, NT,
			                                  reg, dyn_doubles - (int)(reg - s_dyn));
			if (COOP) {
				if (ok) for (int i = tid; i < n6; i += NT) D.dp[6 * (size_t)W.us0 + i] = s_b[i];
				if (tid == 0) D.coop_i[4 * wi + 2] = ok;
			}
		}
		if (COOP) {                                    // the step of the edges, to every workgroup
			alive = coop_barrier(co);
			if (!alive) continue;
			ok = D.coop_i[4 * wi + 2] != 0;
			if (g != 0 && ok) for (int i = tid; i < n6; i += NT) s_b[i] = D.dp[6 * (size_t)W.us0 + i];
			__syncthreads();
		}
		STAMP(6);
#ifdef SSK_LM_STAMPS
		f_last = __builtin_amdgcn_s_memtime();
#endif
		double dpn = 0, dfn = 0;
		if (ok) {
			for (int i = tid; i < n6; i += NT) { if (!COOP) D.dp[6 * (size_t)W.us0 + i] = s_b[i]; dpn += s_b[i] * s_b[i]; }
			// ---- H: landmarks, delta_f = L^-T (c - sum_e Y_e^T delta_e).  First one lane per incidence (coalesced Y stream,
			//      3-vector Y_m^T delta_e staged in the window's L2-resident scratch), then one lane per landmark ----
			// COOP: a workgroup takes the incidences and the landmarks of ITS tiles (a tile's optimised landmarks are a run of slots, their incidences a
			// run of the landmark-major list), so this phase, the scoring and the state copy below need no barrier between workgroups
			const int n_own = COOP ? (W.n_tiles - g + GG - 1) / GG : 1;
			for (int own = 0; own < n_own; ++own) {
				int m_lo = 0, m_hi = W.n_lme_unknown, l_lo = 0, l_hi = W.L;
				if (COOP) {
					const lm_tile& To = D.tile[W.tile0 + g + own * GG];
					l_lo = To.slot0; l_hi = To.slot0 + To.n_slot;
					m_lo = D.lsm_off[W.lso0 + l_lo]; m_hi = D.lsm_off[W.lso0 + l_hi];
				}
			// one pass, a lane per landmark: its incidences' Y blocks are one contiguous run of the landmark-major list (neighbouring lanes read
			// neighbouring runs); the two-pass form -- a lane per incidence, 3-vectors staged in L2 scratch, a barrier, then a lane per landmark --
			// paid three dependent L2 round trips for the same sums (same operations per landmark, same order)
			(void)m_lo; (void)m_hi;
			for (int l = l_lo + tid; l < l_hi; l += NT) {
				const double* lg = D.Vinv + 9 * (size_t)(W.ls0 + l);
				const int m0 = D.lsm_off[W.lso0 + l], m1 = D.lsm_off[W.lso0 + l + 1];
				double li[9];
#pragma unroll
				for (int c = 0; c < 9; ++c) li[c] = lg[c];
				double rhs[3] = {li[6], li[7], li[8]};
				for (int m = m0; m < m1; m += 2) {
					// two incidences per turn, loads first (the second on a clamped index, masked afterwards)
					const int mb = m + 1 < m1 ? m + 1 : m;
					const double* Ya = D.Tb + 18 * (size_t)(W.lme0 + m);
					const double* Yb = D.Tb + 18 * (size_t)(W.lme0 + mb);
					const int ea = D.lme_edge[W.lme0 + m], eb = D.lme_edge[W.lme0 + mb];
					double ya[18], yb[18];
#pragma unroll
					for (int c = 0; c < 18; ++c) { ya[c] = Ya[c]; yb[c] = Yb[c]; }
					double t0 = 0, t1 = 0, t2 = 0, u0 = 0, u1 = 0, u2 = 0;
#pragma unroll
					for (int a = 0; a < 6; ++a) {
						const double da = s_b[6 * ea + a], db = s_b[6 * eb + a];
						t0 += ya[3 * a] * da; t1 += ya[3 * a + 1] * da; t2 += ya[3 * a + 2] * da;
						u0 += yb[3 * a] * db; u1 += yb[3 * a + 1] * db; u2 += yb[3 * a + 2] * db;
					}
					rhs[0] -= t0; rhs[1] -= t1; rhs[2] -= t2;
					if (m + 1 < m1) { rhs[0] -= u0; rhs[1] -= u1; rhs[2] -= u2; }
				}
				const double d3[3] = {li[0] * rhs[0] + li[1] * rhs[1] + li[3] * rhs[2], li[2] * rhs[1] + li[4] * rhs[2], li[5] * rhs[2]};
				const int li_idx = D.slot_lm[W.ls0 + l];
#pragma unroll
				for (int a = 0; a < 3; ++a) {
					D.df[3 * (size_t)(W.ls0 + l) + a] = d3[a];
					if (D.trace_on) D.df_trace[W.df_trace0 + (size_t)it_idx * 3 * W.L + 3 * l + a] = d3[a];
					dfn += d3[a] * d3[a];
					D.lm_new[3 * (size_t)(W.lm0 + li_idx) + a] = D.lm[3 * (size_t)(W.lm0 + li_idx) + a] + d3[a];
				}
			}
			}
			FTICK(0);
			// ---- H: edges, T_new = exp(eps) (+) T ----
			for (int e = tid; e < W.E; e += NT) {
				const int ei = D.slot_edge[W.us0 + e];
				const double* T = pose_cur + 12 * ei;
				double* O = pose_new + 12 * ei;
				double eps[6], R[9];
#pragma unroll
				for (int a = 0; a < 6; ++a) eps[a] = s_b[6 * e + a];
				so3_exp(eps + 3, R);
#pragma unroll
				for (int i = 0; i < 3; ++i) {
#pragma unroll
					for (int j = 0; j < 3; ++j) O[4 * i + j] = R[3 * i] * T[j] + R[3 * i + 1] * T[4 + j] + R[3 * i + 2] * T[8 + j];
					O[4 * i + 3] = R[3 * i] * T[3] + R[3 * i + 1] * T[7] + R[3 * i + 2] * T[11] + eps[i];
				}
			}
			if (D.trace_on && g == 0)
				for (int i = tid; i < n6; i += NT) D.dp_trace[W.dp_trace0 + (size_t)it_idx * n6 + i] = s_b[i];
		}
		FTICK(1);
		dpn = block_sum(dpn, s_part);
		dfn = block_sum(dfn, s_part);
		FTICK(2);
		if (COOP && D.trace_on) dfn = coop_combine<false>(co, dfn, alive);      // (the trace entry reports the whole step's norm)
		STAMP(7);
		// ---- I ----
		double err_new = s_err;
		if (COOP && ok) {
			// the observations of this workgroup's tiles see only its own landmarks' steps (no barrier between workgroups); the
			// observations outside every tile (fixed landmark, fixed path) are dealt out over all workgroups
			__syncthreads();                           // (pose_new and the landmark steps of this workgroup complete)
			build_table(pose_new);
			const double* tabn = W.tab_lds ? tab_l : gtab;
			int n_tiled = 0;
			if (W.n_tiles > 0) { const lm_tile& Tl = D.tile[W.tile0 + W.n_tiles - 1]; n_tiled = Tl.obs0 + Tl.n_obs; }
			double e = 0;
			for (int t = g; t < W.n_tiles; t += GG) {
				const lm_tile& To = D.tile[W.tile0 + t];
				for (int k = To.obs0 + tid; k < To.obs0 + To.n_obs; k += NT)
					e += obs_error_packed(P, tabn, D.lm_new + 3 * (size_t)W.lm0, D.tobs_a[W.tobs0 + k], D.tobs_px[W.tobs0 + k]);
			}
			for (int k = n_tiled + tid + NT * g; k < W.n_obs; k += NT * GG)
				e += obs_error_packed(P, tabn, D.lm_new + 3 * (size_t)W.lm0, D.tobs_a[W.tobs0 + k], D.tobs_px[W.tobs0 + k]);
			err_new = coop_combine<false>(co, block_sum(e, s_part), alive);
			if (!alive) continue;
		}
		if (!COOP && ok) {
			__syncthreads();                               // (pose_new complete)
			FTICK(3);
			build_table(pose_new);
			FTICK(4);
			err_new = W.tab_lds ? total_error_packed(D, W, P, tab_l, D.lm_new, s_part, tid, NT) : total_error_packed(D, W, P, gtab, D.lm_new, s_part, tid, NT);
		}
		FTICK(5);
		__syncthreads();
		if (tid == 0) {
			const int accepted = ok && err_new < s_err;
			if (D.trace_on && g == 0) {
				ssk_lm_trace_entry& t = D.trace[W.trace0 + it_idx];
				t.lambda = lambda; t.err_before = s_err; t.err_after = err_new; t.delta_p_norm = sqrt(dpn);
				t.delta_f_norm = sqrt(dfn); t.accepted = accepted; t.cholesky_ok = ok;
			}
			const int converged = ok && P.conv_rel > 0 && fabs(s_err - err_new) <= P.conv_rel * s_err;
			s_accept = accepted;
			s_same = !accepted;
			if (ok) s_tabcur = accepted;
			s_iters = it_idx + 1;
			if (accepted) {
				s_err = err_new; s_lambda = lambda * 0.1; s_nu = 2.0; s_accepted += 1;
				if (converged) s_stop = 4;
			} else if (converged) {
				s_stop = 4;
			} else {
				s_lambda = lambda * s_nu; s_nu *= 2.0;
				if (s_lambda > P.max_lambda) s_stop = 2;
			}
		}
		__syncthreads();
		if (s_accept) {
			for (int e = tid; e < W.E; e += NT) {
				const int ei = D.slot_edge[W.us0 + e];
				for (int c = 0; c < 12; ++c) pose_cur[12 * ei + c] = pose_new[12 * ei + c];
			}
			// (COOP: the landmarks of this workgroup's tiles -- the next pass linearises them here, no barrier between workgroups)
			const int n_own = COOP ? (W.n_tiles - g + GG - 1) / GG : 1;
			for (int own = 0; own < n_own; ++own) {
				int l_lo = 0, l_hi = W.L;
				if (COOP) { const lm_tile& To = D.tile[W.tile0 + g + own * GG]; l_lo = To.slot0; l_hi = To.slot0 + To.n_slot; }
				for (int l = l_lo + tid; l < l_hi; l += NT) {
					const int li = D.slot_lm[W.ls0 + l];
					for (int c = 0; c < 3; ++c) D.lm[3 * (size_t)(W.lm0 + li) + c] = D.lm_new[3 * (size_t)(W.lm0 + li) + c];
				}
			}
		}
		__syncthreads();
		STAMP(8);
		if (s_stop == 2 || s_stop == 4) break;
	}
	__syncthreads();
	if (g != 0) return;
	// publish the final edge poses (they lived in LDS during the solve)
	for (int e = tid; e < W.E; e += NT) {
		const int ei = D.slot_edge[W.us0 + e];
		for (int c = 0; c < 12; ++c) D.pose[12 * (size_t)(W.edge0 + ei) + c] = pose_cur[12 * ei + c];
	}
	if (tid == 0) {
		ssk_lm_result& R = D.res[wi];
		R.iters = s_iters; R.accepted = s_accepted; R.stop_reason = s_stop; R.err_final = s_err; R.lambda_final = s_lambda;
#ifdef SSK_LM_STAMPS
		if (blockIdx.x == 0)
			{
			printf("LMT stamps (cycles, window 0, %d iters, %d tiles): top %llu I1 %llu I2 %llu I3 %llu I4 %llu F2 %llu G %llu H %llu I+dec %llu\n", s_iters, W.n_tiles,
			       s_acc[0], s_acc[1], s_acc[2], s_acc[3], s_acc[4], s_acc[5], s_acc[6], s_acc[7], s_acc[8]);
			for (int w_ = 0; w_ < 8; ++w_) printf("register tiles, wavefront %d: A %llu B %llu C %llu barriers %llu\n", w_, g_ldl_w[w_][0], g_ldl_w[w_][1], g_ldl_w[w_][2], g_ldl_w[w_][3]);
			printf("LDL wave0 %llu waves1+ %llu step incl barrier %llu backsub %llu | register tiles (thread 0 = tile (0, 0)): A + barrier %llu, B + barrier %llu, C of the last step %llu\n", g_ldl[0], g_ldl[1], g_ldl[2], g_ldl[3], g_ldl[4], g_ldl[5], g_ldl[6]);
			printf("LMT F2 blocks %llu F2 gradient %llu G %llu (slots: pp %d schur %d)\n", g_f2g[0], g_f2g[1], g_f2g[2], D.pcl_off[W.pairo0 + W.n_pairs] - D.pcl_off[W.pairo0], D.scl_off[W.pairo0 + W.n_pairs] - D.scl_off[W.pairo0]);
			printf("tile phases: I1 %llu I2a (landmarks) %llu I2b (incidences) %llu gradient (acc) %llu chunk tasks (all waves) %llu, of which wave 0 busy (acc) %llu\n", g_tst[0], g_tst[5], g_tst[1], g_tst[2], g_tst[3], g_tst[4]);
			printf("chunk tasks after the barrier: wavefront 0 %llu tasks: items %llu, segmented sum + stores %llu | wavefront 4 %llu tasks: items %llu, sum + stores %llu\n", g_fine[2], g_fine[0], g_fine[1], g_fine[6], g_fine[4], g_fine[5]);
			printf("I1, wavefront 0 without the barrier: %llu\n", g_fine[3]);
			printf("incidence role, wavefront 0: sums %llu, lane combine %llu, wait for the landmark records %llu\n", g_tst[4], g_tst[6], g_tst[7]);
			printf("H / I fine (thread 0): landmark back-substitution %llu, edges %llu, norms %llu, sync %llu, table at the trial state %llu, error %llu\n",
			       s_fine[0], s_fine[1], s_fine[2], s_fine[3], s_fine[4], s_fine[5]);
			}
#endif
	}
}

__global__ __launch_bounds__(NT, SSK_LM_WAVES)
void lm_solve_tiled_kernel(lm_dev D, lm_cam P, int dyn_doubles)
{
	lm_tiled_body<false>(D, P, dyn_doubles, 1);
}
// ... with the Schur complement on the FP64 matrix cores (v_mfma_f64_16x16x4_f64 on Y panels, 256-double partials per tile)
__global__ __launch_bounds__(NT, SSK_LM_WAVES)
void lm_solve_tiled_mfma_kernel(lm_dev D, lm_cam P, int dyn_doubles)
{
	lm_tiled_body<false, 0, true>(D, P, dyn_doubles, 1);
}
__global__ __launch_bounds__(NT, SSK_LM_WAVES)
void lm_solve_coop_mfma_kernel(lm_dev D, lm_cam P, int dyn_doubles, int G, int expect, int spin_limit)
{
	lm_tiled_body<true, 0, true>(D, P, dyn_doubles, G, expect, spin_limit);
}
// every reduced system of the batch fits LDS (E <= ~21: the batched per-keyframe windows): without the L2-resident solver
__global__ __launch_bounds__(NT, SSK_LM_WAVES)
void lm_solve_tiled_lds_kernel(lm_dev D, lm_cam P, int dyn_doubles)
{
	lm_tiled_body<false, 0, false, false, true>(D, P, dyn_doubles, 1);
}
// ... and the factorisation of an L2-resident reduced system in 16-column panels on the matrix cores (SSK_LM_LDL=mfma)
__global__ __launch_bounds__(NT, SSK_LM_WAVES)
void lm_solve_tiled_ldlm_kernel(lm_dev D, lm_cam P, int dyn_doubles)
{
	lm_tiled_body<false, 0, true, true>(D, P, dyn_doubles, 1);
}
__global__ __launch_bounds__(NT, SSK_LM_WAVES)
void lm_solve_coop_ldlm_kernel(lm_dev D, lm_cam P, int dyn_doubles, int G, int expect, int spin_limit)
{
	lm_tiled_body<true, 0, true, true>(D, P, dyn_doubles, G, expect, spin_limit);
}

// on-chip accumulation (E <= SSK_LMA_MAX_E): Schur complement by sparse 6x6 items / on the FP64 matrix cores
__global__ __launch_bounds__(NT, SSK_LM_WAVES)
void lm_solve_acc_kernel(lm_dev D, lm_cam P, int dyn_doubles)
{
	lm_tiled_body<false, 1>(D, P, dyn_doubles, 1);
}
__global__ __launch_bounds__(NT, SSK_LM_WAVES)
void lm_solve_acc_mfma_kernel(lm_dev D, lm_cam P, int dyn_doubles)
{
	lm_tiled_body<false, 2>(D, P, dyn_doubles, 1);
}

// G workgroups per window (grid = n_windows * G, all resident)
__global__ __launch_bounds__(NT, SSK_LM_WAVES)
void lm_solve_coop_kernel(lm_dev D, lm_cam P, int dyn_doubles, int G, int expect, int spin_limit)
{
	lm_tiled_body<true>(D, P, dyn_doubles, G, expect, spin_limit);
}

}  // namespace

// How many workgroups of the cooperative kernel one CU can hold with this much dynamic LDS (0 on error): the hand-rolled grid
// barrier needs every workgroup of a window resident at once, so the caller sizes G from this, not from the CU count alone.
int ssk_lm_coop_blocks_per_cu(ssk_ctx* ctx, size_t dyn_lds_bytes, int schur_mfma)
{
	dyn_lds_bytes = (dyn_lds_bytes + 15) & ~(size_t)15;
	const void* fn = schur_mfma == 2 ? reinterpret_cast<const void*>(lm_solve_coop_ldlm_kernel)
	               : schur_mfma ? reinterpret_cast<const void*>(lm_solve_coop_mfma_kernel) : reinterpret_cast<const void*>(lm_solve_coop_kernel);
	(void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, SSK_LMT_LDS_TOTAL - SSK_LMT_STATIC);
	int n = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, NT, dyn_lds_bytes) != hipSuccess) { (void)hipGetLastError(); return 0; }
	return n;
}

int ssk_lm_launch_coop(ssk_ctx* ctx, const lm_dev& D, const lm_cam& P, int n_windows, int G, size_t dyn_lds_bytes, bool test_timeout, int schur_mfma)
{
	dyn_lds_bytes = (dyn_lds_bytes + 15) & ~(size_t)15;
	static ssk_dev_once attr_once;      // (the attribute is per device)
	if (attr_once.need(ctx->device)) {
		SSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(lm_solve_coop_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
		                                 SSK_LMT_LDS_TOTAL - SSK_LMT_STATIC));
		SSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(lm_solve_coop_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
		                                 SSK_LMT_LDS_TOTAL - SSK_LMT_STATIC));
		SSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(lm_solve_coop_ldlm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
		                                 SSK_LMT_LDS_TOTAL - SSK_LMT_STATIC));
		attr_once.done(ctx->device);
	}
	SSK_HIP(ctx, hipMemsetAsync(D.coop_i, 0, sizeof(int) * 4 * (size_t)n_windows, ctx->stream));
	ssk_stage_scope prof(ctx, SSK_STAGE_LM_SOLVE);
	// test_timeout (SSK_LM_COOP_TEST_TIMEOUT=1): every barrier waits for an arrival that never comes and gives up quickly, the
	// path a workgroup takes when a co-tenant of the GPU keeps one of its partners from being scheduled
	if (schur_mfma == 2) hipLaunchKernelGGL(lm_solve_coop_ldlm_kernel, dim3(n_windows * G), dim3(NT), dyn_lds_bytes, ctx->stream, D, P, (int)(dyn_lds_bytes / 8), G,
	                                        test_timeout ? G + 1 : G, test_timeout ? 1 << 10 : COOP_SPIN_LIMIT);
	else if (schur_mfma) hipLaunchKernelGGL(lm_solve_coop_mfma_kernel, dim3(n_windows * G), dim3(NT), dyn_lds_bytes, ctx->stream, D, P, (int)(dyn_lds_bytes / 8), G,
	                                   test_timeout ? G + 1 : G, test_timeout ? 1 << 10 : COOP_SPIN_LIMIT);
	else hipLaunchKernelGGL(lm_solve_coop_kernel, dim3(n_windows * G), dim3(NT), dyn_lds_bytes, ctx->stream, D, P, (int)(dyn_lds_bytes / 8), G,
	                        test_timeout ? G + 1 : G, test_timeout ? 1 << 10 : COOP_SPIN_LIMIT);
	SSK_LAUNCH_CHECK(ctx, "lm_solve_coop_kernel");
	return SSK_OK;
}

int ssk_lm_launch_acc(ssk_ctx* ctx, const lm_dev& D, const lm_cam& P, int n_windows, size_t dyn_lds_bytes, int schur_mfma)
{
	dyn_lds_bytes = (dyn_lds_bytes + 15) & ~(size_t)15;
	static ssk_dev_once attr_once;      // (the attribute is per device)
	if (attr_once.need(ctx->device)) {
		SSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(lm_solve_acc_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SSK_LMT_LDS_TOTAL - SSK_LMT_STATIC));
		SSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(lm_solve_acc_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SSK_LMT_LDS_TOTAL - SSK_LMT_STATIC));
		attr_once.done(ctx->device);
	}
	ssk_stage_scope prof(ctx, SSK_STAGE_LM_SOLVE);
	if (schur_mfma) hipLaunchKernelGGL(lm_solve_acc_mfma_kernel, dim3(n_windows), dim3(NT), dyn_lds_bytes, ctx->stream, D, P, (int)(dyn_lds_bytes / 8));
	else hipLaunchKernelGGL(lm_solve_acc_kernel, dim3(n_windows), dim3(NT), dyn_lds_bytes, ctx->stream, D, P, (int)(dyn_lds_bytes / 8));
	SSK_LAUNCH_CHECK(ctx, schur_mfma ? "lm_solve_acc_mfma_kernel" : "lm_solve_acc_kernel");
	return SSK_OK;
}

int ssk_lm_launch_tiled(ssk_ctx* ctx, const lm_dev& D, const lm_cam& P, int n_windows, size_t dyn_lds_bytes, int schur_mfma, bool lds_sys)
{
	dyn_lds_bytes = (dyn_lds_bytes + 15) & ~(size_t)15;
	static ssk_dev_once attr_once;      // (the attribute is per device)
	if (attr_once.need(ctx->device)) {
		SSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(lm_solve_tiled_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
		                                 SSK_LMT_LDS_TOTAL - SSK_LMT_STATIC));
		SSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(lm_solve_tiled_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
		                                 SSK_LMT_LDS_TOTAL - SSK_LMT_STATIC));
		SSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(lm_solve_tiled_ldlm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
		                                 SSK_LMT_LDS_TOTAL - SSK_LMT_STATIC));
		SSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(lm_solve_tiled_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
		                                 SSK_LMT_LDS_TOTAL - SSK_LMT_STATIC));
		attr_once.done(ctx->device);
	}
	ssk_stage_scope prof(ctx, SSK_STAGE_LM_SOLVE);
	if (schur_mfma == 2) hipLaunchKernelGGL(lm_solve_tiled_ldlm_kernel, dim3(n_windows), dim3(NT), dyn_lds_bytes, ctx->stream, D, P, (int)(dyn_lds_bytes / 8));
	else if (schur_mfma) hipLaunchKernelGGL(lm_solve_tiled_mfma_kernel, dim3(n_windows), dim3(NT), dyn_lds_bytes, ctx->stream, D, P, (int)(dyn_lds_bytes / 8));
	else if (lds_sys) hipLaunchKernelGGL(lm_solve_tiled_lds_kernel, dim3(n_windows), dim3(NT), dyn_lds_bytes, ctx->stream, D, P, (int)(dyn_lds_bytes / 8));
	else hipLaunchKernelGGL(lm_solve_tiled_kernel, dim3(n_windows), dim3(NT), dyn_lds_bytes, ctx->stream, D, P, (int)(dyn_lds_bytes / 8));
	SSK_LAUNCH_CHECK(ctx, "lm_solve_tiled_kernel");
	return SSK_OK;
}
