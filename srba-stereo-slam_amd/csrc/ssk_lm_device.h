// ssk_lm_device.h -- device helpers shared by the two LM kernels (ssk_lm.hip: scratch-in-L2 kernel for windows of
// any size; ssk_lm_tiled.hip: LDS-tiled kernel).  Include inside the translation unit's anonymous namespace.
#pragma once

#ifndef SSK_LM_WAVES
#define SSK_LM_WAVES 2          // min waves per SIMD the register budget must allow (2 = one 512-thread workgroup per CU)
#endif
#ifndef SSK_RT_UNROLL
#define SSK_RT_UNROLL 1                 // unroll factor of the register-tile update's column loop (ldl_solve_regtile)
#endif
#ifndef SSK_LM_NT
#define SSK_LM_NT 512
#endif
constexpr int NT = SSK_LM_NT;
constexpr int NW = NT / 64;

#define WAVE_SYNC()                                                \
	do {                                                           \
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     \
		__builtin_amdgcn_wave_barrier();                           \
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");     \
	} while (0)

__device__ __forceinline__ void se3_apply(const double* T, const double* p, double* q)
{
#pragma unroll
	for (int i = 0; i < 3; ++i) q[i] = T[4 * i] * p[0] + T[4 * i + 1] * p[1] + T[4 * i + 2] * p[2] + T[4 * i + 3];
}
__device__ __forceinline__ void se3_apply_inv(const double* T, const double* p, double* q)
{
	const double d0 = p[0] - T[3], d1 = p[1] - T[7], d2 = p[2] - T[11];
#pragma unroll
	for (int i = 0; i < 3; ++i) q[i] = T[i] * d0 + T[4 + i] * d1 + T[8 + i] * d2;
}
__device__ __forceinline__ double huber_rho(double s2, double k) { return 2.0 * k * k * (sqrt(1.0 + s2 / (k * k)) - 1.0); }

__device__ __forceinline__ void project(const lm_cam& P, const double* pc, double* h)
{
	const double iz = 1.0 / pc[2];
	h[0] = P.cx_l + P.fx_l * pc[0] * iz;
	h[1] = P.cy_l + P.fy_l * pc[1] * iz;
	h[2] = P.cx_r + P.fx_r * (pc[0] - P.baseline) * iz;
	h[3] = P.cy_r + P.fy_r * pc[1] * iz;
}

// robustified squared error of observation o; `pose` is window relative (LDS cache or global), lm batch wide
__device__ double obs_error(const lm_dev& D, const lm_win& W, const lm_cam& P, const double* pose, const double* __restrict__ lm, int o)
{
	double q[3], t[3];
	const double* f = lm + 3 * (size_t)(W.lm0 + D.obs_lm[W.obs0 + o]);
	q[0] = f[0]; q[1] = f[1]; q[2] = f[2];
	const int pb = D.obs_path_off[W.opo0 + o], pe = D.obs_path_off[W.opo0 + o + 1];
	for (int i = pe - 1; i >= pb; --i) {
		const int c = D.path_edge[W.path0 + i];
		const double* T = pose + 12 * (c >> 1);
		if ((c & 1) == 0) se3_apply(T, q, t); else se3_apply_inv(T, q, t);
		q[0] = t[0]; q[1] = t[1]; q[2] = t[2];
	}
	double pc[3], h[4];
	se3_apply_inv(P.S, q, pc);
	project(P, pc, h);
	const float* z = D.obs_px + 4 * (size_t)(W.obs0 + o);
	double s2 = 0;
#pragma unroll
	for (int k = 0; k < 4; ++k) { const double r = (double)z[k] - h[k]; s2 += r * r; }
	return P.robust ? huber_rho(s2, P.kernel) : s2;
}

// deterministic workgroup reductions: fixed shuffle tree inside a wave, then the NW partials in wave order
__device__ double block_sum(double v, double* s_part)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
	__syncthreads();
	if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
	__syncthreads();
	double r = 0;
#pragma unroll
	for (int w = 0; w < NW; ++w) r += s_part[w];
	return r;
}
__device__ double block_max(double v, double* s_part)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
	__syncthreads();
	if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
	__syncthreads();
	double r = s_part[0];
#pragma unroll
	for (int w = 1; w < NW; ++w) r = fmax(r, s_part[w]);
	return r;
}

__device__ double total_error(const lm_dev& D, const lm_win& W, const lm_cam& P, const double* pose, const double* lm, double* s_part)
{
	double e = 0;
	for (int o = threadIdx.x; o < W.n_obs; o += NT) e += obs_error(D, W, P, pose, lm, o);
	return block_sum(e, s_part);
}

__device__ void so3_exp(const double* w, double* R)
{
	const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
	const double th = sqrt(th2);
	double a, b;
	if (th < 1e-8) { a = 1.0 - th2 / 6.0; b = 0.5 - th2 / 24.0; }
	else { a = sin(th) / th; b = (1.0 - cos(th)) / th2; }
	const double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
	double K2[9];
#pragma unroll
	for (int i = 0; i < 3; ++i)
#pragma unroll
		for (int j = 0; j < 3; ++j) K2[3 * i + j] = K[3 * i] * K[j] + K[3 * i + 1] * K[3 + j] + K[3 * i + 2] * K[6 + j];
#pragma unroll
	for (int i = 0; i < 9; ++i) R[i] = a * K[i] + b * K2[i];
	R[0] += 1; R[4] += 1; R[8] += 1;
}

// (the same, always inlined: a call passes w and R through scratch memory)
__device__ __forceinline__ void so3_exp_inl(const double* w, double* R)
{
	const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
	const double th = sqrt(th2);
	double a, b;
	if (th < 1e-8) { a = 1.0 - th2 / 6.0; b = 0.5 - th2 / 24.0; }
	else { a = sin(th) / th; b = (1.0 - cos(th)) / th2; }
	const double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
	double K2[9];
#pragma unroll
	for (int i = 0; i < 3; ++i)
#pragma unroll
		for (int j = 0; j < 3; ++j) K2[3 * i + j] = K[3 * i] * K[j] + K[3 * i + 1] * K[3 + j] + K[3 * i + 2] * K[6 + j];
#pragma unroll
	for (int i = 0; i < 9; ++i) R[i] = a * K[i] + b * K2[i];
	R[0] += 1; R[4] += 1; R[8] += 1;
}

// G (fast path): the reduced system solved in LDS as a square-root-free LDL^T of the matrix bordered by the
// right-hand side (row n = g_red^T), right-looking, by the whole workgroup:
//   A(i,k) -= A(i,j) A(k,j) / d_j   for j < k <= i <= n; column j itself stays unscaled (U_ij = L'_ij d_j), d_j = A(j,j)
//   when its turn comes.  Blocked by the 6 columns of an edge, with look-ahead: wave 0 applies the current panel to the
//   next panel's columns, factorises that 6x6 diagonal block in registers and finishes the panel (a lane per row) while
//   the other waves apply the current panel to the remaining columns: one barrier per block column.
// The border row comes out as z = L'^-1 b (forward substitution for free), so only L'^T x = D^-1 z is left: one
// wavefront keeps x in registers (lane k owns x_k, x_{k+64}), broadcasts x_i with v_readlane and prefetches row
// i-1 of U while it applies row i, so the dependent chain per step is one readlane + one FMA.
// A is (n+1) x n with an odd row pitch np (column walks are bank-conflict free), n a multiple of 6; it may live in LDS
// or (large systems) in global scratch -- the code only uses plain pointers.  Every thread returns the same flag.
__device__ __forceinline__ double rcp_f64(double d)
{
	double x = __builtin_amdgcn_rcp(d);
	x = fma(fma(-d, x, 1.0), x, x);
	x = fma(fma(-d, x, 1.0), x, x);
	return x;
}

__device__ __forceinline__ double readlane_f64(double v, int l)
{
	const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
	return __hiloint2double(hi, lo);
}

// L'^T x = D^-1 z by ONE wave: lane k owns x_k (and x_{k+64} when TWO: 64 < n <= 128).  The caller has pre-scaled the matrix
// (A(i,k) <- A(i,k) / d_k below the diagonal, 0 on and above it, border row included), so a row costs its loads -- issued a
// block of 6 ahead -- plus the dependent chain v_readlane -> FMA; a single in-order wave pays ~9 cycles per dependent
// instruction, so everything else (scaling, masks, address arithmetic on a runtime pitch) is kept out of this loop.
template <bool TWO>
__device__ __forceinline__ void ldl_backsub_wave(const double* A, int np, int n, double* x_out)
{
	const int tid = threadIdx.x;
	const int k0 = tid, k1 = tid + 64;
	const int q0 = min(k0, n - 1), q1 = min(k1, n - 1);     // lanes outside the system read a valid column; nobody reads them
	double y0 = A[n * np + q0], y1 = TWO ? A[n * np + q1] : 0.0;
	double cur[6][2], nxt[6][2];
	const int nb = n / 6;
#pragma unroll
	for (int j = 0; j < 6; ++j) {
		const int i = 6 * (nb - 1) + j;
		cur[j][0] = A[i * np + q0];
		cur[j][1] = TWO ? A[i * np + q1] : 0.0;
		nxt[j][0] = nxt[j][1] = 0.0;
	}
	for (int b = nb - 1; b >= 0; --b) {
		if (b > 0) {
			// the row addresses are rebuilt from b every time (the pitch made opaque): as strength-reduced loop registers
			// they were spilled to scratch and every block waited for their reloads
			int npv = np;
			asm volatile("" : "+v"(npv));
#pragma unroll
			for (int j = 0; j < 6; ++j) {
				const int i = 6 * (b - 1) + j;
				nxt[j][0] = A[i * npv + q0];
				if (TWO) nxt[j][1] = A[i * npv + q1];
			}
		}
#pragma unroll
		for (int j = 5; j >= 0; --j) {
			const int i = 6 * b + j;
			double xi;
			if (TWO) xi = i < 64 ? readlane_f64(y0, i) : readlane_f64(y1, i - 64);
			else xi = readlane_f64(y0, i);
			y0 -= cur[j][0] * xi;                         // coefficient 0 for k >= i: x_k (k >= i) is already final
			if (TWO) y1 -= cur[j][1] * xi;
		}
#pragma unroll
		for (int j = 0; j < 6; ++j) { cur[j][0] = nxt[j][0]; cur[j][1] = nxt[j][1]; }
	}
	if (k0 < n) x_out[k0] = y0;
	if (TWO && k1 < n) x_out[k1] = y1;
}

#ifdef SSK_LM_STAMPS
__device__ unsigned long long g_ldl[8];
__device__ unsigned long long g_ldl_w[8][4];      // register-tile solver, per wavefront: cycles inside A, B, C, at the two barriers
#define LDL_T0() const unsigned long long ldl_t0_ = __builtin_amdgcn_s_memtime()
#define LDL_ADD(i, who) do { if (threadIdx.x == (who) && blockIdx.x == 0) g_ldl[i] += __builtin_amdgcn_s_memtime() - ldl_t0_; } while (0)
#else
#define LDL_T0() do {} while (0)
#define LDL_ADD(i, who) do {} while (0)
#endif
#ifdef SSK_LM_STAMPS
#define LDLH_T() unsigned long long ldlh_t_ = __builtin_amdgcn_s_memtime()
#define LDLW_T() unsigned long long ldlw_t_ = __builtin_amdgcn_s_memtime()
#define LDLW_ADD(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) g_ldl_w[threadIdx.x >> 6][i] += t_ - ldlw_t_; ldlw_t_ = t_; } while (0)
#define LDLH_ADD(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); g_ldl[i] += t_ - ldlh_t_; ldlh_t_ = t_; } } while (0)
#else
#define LDLH_T() do {} while (0)
#define LDLW_T() do {} while (0)
#define LDLW_ADD(i) do {} while (0)
#define LDLH_ADD(i) do {} while (0)
#endif
// index map of the in-LDS factorisation of a packed lower triangle (row i at i (i + 1) / 2)
struct ldl_ix_packed { __device__ __forceinline__ int operator()(int i, int k) const { return i * (i + 1) / 2 + k; } };

// The factorisation proper, L' D L'^T of rows 0 .. n (row n = right-hand side) in LDS under an index map: see ldl_solve_lds.
// (ix(i, k) of an element above the diagonal must stay inside the allocation: it is read, never used or stored.)
// nr >= n: the last row.  nr = n is the square system with its border row; nr > n a block COLUMN of a larger system (n columns
// factorised, the rows below eliminated on the way: ldl_solve_blocked).
template <typename IX>
__device__ __forceinline__ bool ldl_factor_lookahead(double* A, const IX ix, int n, double* inv, int nr)
{
	const int tid = threadIdx.x;
	__shared__ int s_ldl_bad_p;
	if (tid < 64) {
		// panel 0: the diagonal block by broadcast reads, then one lane per row below it (the border row included)
		double a[21], ivr[6];
#pragma unroll
		for (int i = 0; i < 6; ++i)
#pragma unroll
			for (int k = 0; k <= i; ++k) a[i * (i + 1) / 2 + k] = A[ix(i, k)];
		bool bad = false;
#pragma unroll
		for (int j = 0; j < 6; ++j) {
			const double d = a[j * (j + 1) / 2 + j];
			bad |= !(d > 0.0);
			ivr[j] = rcp_f64(d);
#pragma unroll
			for (int i = j + 1; i < 6; ++i) {
				const double t = a[i * (i + 1) / 2 + j] * ivr[j];
#pragma unroll
				for (int k = j + 1; k <= i; ++k) a[i * (i + 1) / 2 + k] -= t * a[k * (k + 1) / 2 + j];
			}
		}
		for (int i = 6 + tid; i <= nr; i += 64) {
			double x[6];
#pragma unroll
			for (int k = 0; k < 6; ++k) x[k] = A[ix(i, k)];
#pragma unroll
			for (int j = 0; j < 5; ++j) {
				const double t = x[j] * ivr[j];
#pragma unroll
				for (int k = j + 1; k < 6; ++k) x[k] -= t * a[k * (k + 1) / 2 + j];
			}
#pragma unroll
			for (int k = 1; k < 6; ++k) A[ix(i, k)] = x[k];
		}
		if (tid == 0) {
			s_ldl_bad_p = bad;
#pragma unroll
			for (int i = 0; i < 6; ++i) {
				inv[i] = ivr[i];
#pragma unroll
				for (int k = 0; k < i; ++k) A[ix(i, k)] = a[i * (i + 1) / 2 + k];
			}
		}
	}
	for (int c0 = 0; c0 < n; c0 += 6) {
		__syncthreads();                                  // panel c0 complete; the update by panel c0 - 6 applied everywhere
		if (s_ldl_bad_p) return false;                      // uniform
		const int r0 = c0 + 6;
		if (r0 >= n) break;
		double iv[6];
#pragma unroll
		for (int c = 0; c < 6; ++c) iv[c] = inv[c0 + c];
		
		if (tid < 64) {
			// ---- wave 0: the next panel (columns r0 .. r0+5): update by panel c0, factorise, eliminate ----
			double uk[6][6];                              // rows r0 .. r0+5 of panel c0 (broadcast reads)
#pragma unroll
			for (int k = 0; k < 6; ++k)
#pragma unroll
				for (int c = 0; c < 6; ++c) uk[k][c] = A[ix((r0 + k), c0 + c)];
			double a[21], ivr[6];
			bool bad = false;
			for (int i0 = r0; i0 <= nr; i0 += 64) {
				const int i = i0 + tid;
				const bool row = i <= nr;
				const int ic = row ? i : nr;                  // loads are unconditional on a clamped row (a load under a lane
				double x[6];                                  // condition is compiled as a branch that waits for it at once)
				{
					double ui[6], v[6];
#pragma unroll
					for (int c = 0; c < 6; ++c) ui[c] = A[ix(ic, c0 + c)];
#pragma unroll
					for (int k = 0; k < 6; ++k) v[k] = A[ix(ic, r0 + k)];
#pragma unroll
					for (int c = 0; c < 6; ++c) ui[c] *= iv[c];
#pragma unroll
					for (int k = 0; k < 6; ++k) {
						const bool on = row && r0 + k <= i;          // the lower triangle (r0 + k < n always: no column n)
						double sacc = 0.0;
#pragma unroll
						for (int c = 0; c < 6; ++c) sacc += ui[c] * uk[k][c];
						x[k] = on ? v[k] - sacc : 0.0;
					}
				}
				if (i0 == r0) {
					// lanes 0..5 hold the updated rows of the next diagonal block: every lane takes a copy and factorises it
#pragma unroll
					for (int r = 0; r < 6; ++r)
#pragma unroll
						for (int k = 0; k <= r; ++k) a[r * (r + 1) / 2 + k] = readlane_f64(x[k], r);
#pragma unroll
					for (int j = 0; j < 6; ++j) {
						const double d = a[j * (j + 1) / 2 + j];
						bad |= !(d > 0.0);
						ivr[j] = rcp_f64(d);
#pragma unroll
						for (int r = j + 1; r < 6; ++r) {
							const double t = a[r * (r + 1) / 2 + j] * ivr[j];
#pragma unroll
							for (int k = j + 1; k <= r; ++k) a[r * (r + 1) / 2 + k] -= t * a[k * (k + 1) / 2 + j];
						}
					}
				}
				if (row && i >= r0 + 6) {
#pragma unroll
					for (int j = 0; j < 5; ++j) {
						const double t = x[j] * ivr[j];
#pragma unroll
						for (int k = j + 1; k < 6; ++k) x[k] -= t * a[k * (k + 1) / 2 + j];
					}
#pragma unroll
					for (int k = 0; k < 6; ++k) A[ix(i, r0 + k)] = x[k];
				}
			}
			if (tid == 0) {
				s_ldl_bad_p = bad;
#pragma unroll
				for (int r = 0; r < 6; ++r) {
					inv[r0 + r] = ivr[r];
#pragma unroll
					for (int k = 0; k <= r; ++k) A[ix((r0 + r), r0 + k)] = a[r * (r + 1) / 2 + k];
				}
			}
			
		} else {
			// ---- waves 1..: A(i,k) -= sum_c U(i,c) U(k,c) / d_c for the columns beyond the next panel, r0+6 <= k <= min(i, n-1),
			//      i <= n: 4 x 2 elements per thread and pass, loads before stores ----
			constexpr int NR = (NT - 64) / 32;            // row groups of 32 lanes
			const int t2 = tid - 64, ty = t2 >> 5, tx = t2 & 31;
			const int q0 = r0 + 6;
			for (int ib = q0 + ty; ib <= nr; ib += 4 * NR)
				for (int kb = q0 + tx; kb < n; kb += 64) {
					if (kb - tx > ib - ty + 4 * NR - 1) break;       // the whole pass lies above the diagonal (uniform)
					// unconditional loads on clamped indices (see above); only the stores are predicated
					double ui[4][6], uk[2][6], v[8];
#pragma unroll
					for (int u = 0; u < 4; ++u) {
						const int i = min(ib + NR * u, nr);
#pragma unroll
						for (int c = 0; c < 6; ++c) ui[u][c] = A[ix(i, c0 + c)];
					}
#pragma unroll
					for (int w = 0; w < 2; ++w) {
						const int k = min(kb + 32 * w, n - 1);
#pragma unroll
						for (int c = 0; c < 6; ++c) uk[w][c] = A[ix(k, c0 + c)];
					}
#pragma unroll
					for (int u = 0; u < 4; ++u)
#pragma unroll
						for (int w = 0; w < 2; ++w) v[2 * u + w] = A[ix(min(ib + NR * u, nr), min(kb + 32 * w, n - 1))];
#pragma unroll
					for (int u = 0; u < 4; ++u)
#pragma unroll
						for (int c = 0; c < 6; ++c) ui[u][c] *= iv[c];
#pragma unroll
					for (int u = 0; u < 4; ++u)
#pragma unroll
						for (int w = 0; w < 2; ++w) {
							const int i = ib + NR * u, k = kb + 32 * w;
							if (i <= nr && k < n && k <= i) {
								double sacc = 0.0;
#pragma unroll
								for (int c = 0; c < 6; ++c) sacc += ui[u][c] * uk[w][c];
								A[ix(i, k)] = v[2 * u + w] - sacc;
							}
						}
				}
			
		}
		__syncthreads();
		
	}
	return true;
}

__device__ __forceinline__ bool ldl_solve_lds(double* A, int np, int n, double* x_out, double* inv)
{
	int tid = threadIdx.x;
	asm volatile("" : "+v"(tid));                      // (opaque: nothing derived from it is hoisted in front of a caller's LM loop, see ldl_solve_tiles4_lds)
	// blocked by the 6 columns of one edge (n = 6E), with LOOK-AHEAD: while waves 1.. apply panel c0 to the columns
	// beyond the next panel, wave 0 applies it to the next panel's 6 columns only, keeps its rows in registers, factorises
	// the next 6x6 diagonal block (in every lane's registers, handed over by v_readlane) and eliminates inside those
	// columns -- the serial part of a step runs beside the update instead of after it: ONE barrier per block column.
	// Every element still receives the same operations in the same order as the step-by-step form.
	__shared__ int s_ldl_bad;
	if (tid < 64) {
		// panel 0: the diagonal block by broadcast reads, then one lane per row below it (the border row included)
		double a[21], ivr[6];
#pragma unroll
		for (int i = 0; i < 6; ++i)
#pragma unroll
			for (int k = 0; k <= i; ++k) a[i * (i + 1) / 2 + k] = A[i * np + k];
		bool bad = false;
#pragma unroll
		for (int j = 0; j < 6; ++j) {
			const double d = a[j * (j + 1) / 2 + j];
			bad |= !(d > 0.0);
			ivr[j] = rcp_f64(d);
#pragma unroll
			for (int i = j + 1; i < 6; ++i) {
				const double t = a[i * (i + 1) / 2 + j] * ivr[j];
#pragma unroll
				for (int k = j + 1; k <= i; ++k) a[i * (i + 1) / 2 + k] -= t * a[k * (k + 1) / 2 + j];
			}
		}
		for (int i = 6 + tid; i <= n; i += 64) {
			double x[6];
#pragma unroll
			for (int k = 0; k < 6; ++k) x[k] = A[i * np + k];
#pragma unroll
			for (int j = 0; j < 5; ++j) {
				const double t = x[j] * ivr[j];
#pragma unroll
				for (int k = j + 1; k < 6; ++k) x[k] -= t * a[k * (k + 1) / 2 + j];
			}
#pragma unroll
			for (int k = 1; k < 6; ++k) A[i * np + k] = x[k];
		}
		if (tid == 0) {
			s_ldl_bad = bad;
#pragma unroll
			for (int i = 0; i < 6; ++i) {
				inv[i] = ivr[i];
#pragma unroll
				for (int k = 0; k < i; ++k) A[i * np + k] = a[i * (i + 1) / 2 + k];
			}
		}
	}
	for (int c0 = 0; c0 < n; c0 += 6) {
		__syncthreads();                                  // panel c0 complete; the update by panel c0 - 6 applied everywhere
		if (s_ldl_bad) return false;                      // uniform
		const int r0 = c0 + 6;
		if (r0 >= n) break;
		double iv[6];
#pragma unroll
		for (int c = 0; c < 6; ++c) iv[c] = inv[c0 + c];
		LDL_T0();
		if (tid < 64) {
			// ---- wave 0: the next panel (columns r0 .. r0+5): update by panel c0, factorise, eliminate ----
			double uk[6][6];                              // rows r0 .. r0+5 of panel c0 (broadcast reads)
#pragma unroll
			for (int k = 0; k < 6; ++k)
#pragma unroll
				for (int c = 0; c < 6; ++c) uk[k][c] = A[(r0 + k) * np + c0 + c];
			double a[21], ivr[6];
			bool bad = false;
			for (int i0 = r0; i0 <= n; i0 += 64) {
				const int i = i0 + tid;
				const bool row = i <= n;
				const int ic = row ? i : n;                   // loads are unconditional on a clamped row (a load under a lane
				double x[6];                                  // condition is compiled as a branch that waits for it at once)
				{
					double ui[6], v[6];
#pragma unroll
					for (int c = 0; c < 6; ++c) ui[c] = A[ic * np + c0 + c];
#pragma unroll
					for (int k = 0; k < 6; ++k) v[k] = A[ic * np + r0 + k];
#pragma unroll
					for (int c = 0; c < 6; ++c) ui[c] *= iv[c];
#pragma unroll
					for (int k = 0; k < 6; ++k) {
						const bool on = row && r0 + k <= i;          // the lower triangle (r0 + k < n always: no column n)
						double sacc = 0.0;
#pragma unroll
						for (int c = 0; c < 6; ++c) sacc += ui[c] * uk[k][c];
						x[k] = on ? v[k] - sacc : 0.0;
					}
				}
				if (i0 == r0) {
					// lanes 0..5 hold the updated rows of the next diagonal block: every lane takes a copy and factorises it
#pragma unroll
					for (int r = 0; r < 6; ++r)
#pragma unroll
						for (int k = 0; k <= r; ++k) a[r * (r + 1) / 2 + k] = readlane_f64(x[k], r);
#pragma unroll
					for (int j = 0; j < 6; ++j) {
						const double d = a[j * (j + 1) / 2 + j];
						bad |= !(d > 0.0);
						ivr[j] = rcp_f64(d);
#pragma unroll
						for (int r = j + 1; r < 6; ++r) {
							const double t = a[r * (r + 1) / 2 + j] * ivr[j];
#pragma unroll
							for (int k = j + 1; k <= r; ++k) a[r * (r + 1) / 2 + k] -= t * a[k * (k + 1) / 2 + j];
						}
					}
				}
				if (row && i >= r0 + 6) {
#pragma unroll
					for (int j = 0; j < 5; ++j) {
						const double t = x[j] * ivr[j];
#pragma unroll
						for (int k = j + 1; k < 6; ++k) x[k] -= t * a[k * (k + 1) / 2 + j];
					}
#pragma unroll
					for (int k = 0; k < 6; ++k) A[i * np + r0 + k] = x[k];
				}
			}
			if (tid == 0) {
				s_ldl_bad = bad;
#pragma unroll
				for (int r = 0; r < 6; ++r) {
					inv[r0 + r] = ivr[r];
#pragma unroll
					for (int k = 0; k <= r; ++k) A[(r0 + r) * np + r0 + k] = a[r * (r + 1) / 2 + k];
				}
			}
			LDL_ADD(0, 0);
		} else {
			// ---- waves 1..: A(i,k) -= sum_c U(i,c) U(k,c) / d_c for the columns beyond the next panel, r0+6 <= k <= min(i, n-1),
			//      i <= n: 4 x 2 elements per thread and pass, loads before stores ----
			constexpr int NR = (NT - 64) / 32;            // row groups of 32 lanes
			const int t2 = tid - 64, ty = t2 >> 5, tx = t2 & 31;
			const int q0 = r0 + 6;
			for (int ib = q0 + ty; ib <= n; ib += 4 * NR)
				for (int kb = q0 + tx; kb < n; kb += 64) {
					if (kb - tx > ib - ty + 4 * NR - 1) break;       // the whole pass lies above the diagonal (uniform)
					// unconditional loads on clamped indices (see above); only the stores are predicated
					double ui[4][6], uk[2][6], v[8];
#pragma unroll
					for (int u = 0; u < 4; ++u) {
						const int i = min(ib + NR * u, n);
#pragma unroll
						for (int c = 0; c < 6; ++c) ui[u][c] = A[i * np + c0 + c];
					}
#pragma unroll
					for (int w = 0; w < 2; ++w) {
						const int k = min(kb + 32 * w, n - 1);
#pragma unroll
						for (int c = 0; c < 6; ++c) uk[w][c] = A[k * np + c0 + c];
					}
#pragma unroll
					for (int u = 0; u < 4; ++u)
#pragma unroll
						for (int w = 0; w < 2; ++w) v[2 * u + w] = A[min(ib + NR * u, n) * np + min(kb + 32 * w, n - 1)];
#pragma unroll
					for (int u = 0; u < 4; ++u)
#pragma unroll
						for (int c = 0; c < 6; ++c) ui[u][c] *= iv[c];
#pragma unroll
					for (int u = 0; u < 4; ++u)
#pragma unroll
						for (int w = 0; w < 2; ++w) {
							const int i = ib + NR * u, k = kb + 32 * w;
							if (i <= n && k < n && k <= i) {
								double sacc = 0.0;
#pragma unroll
								for (int c = 0; c < 6; ++c) sacc += ui[u][c] * uk[w][c];
								A[i * np + k] = v[2 * u + w] - sacc;
							}
						}
				}
			LDL_ADD(1, 64);
		}
		__syncthreads();
		LDL_ADD(2, 0);
	}
	__syncthreads();
	if (n > 128) {
		// any size (the matrix then lives in L2 scratch): x <- D^-1 z in LDS, then row i of U updates x_k, k < i, by
		// the whole workgroup: one barrier per row
		for (int k = tid; k < n; k += NT) x_out[k] = A[(size_t)n * np + k] * inv[k];
		__syncthreads();
		for (int i = n - 1; i > 0; --i) {
			const double xi = x_out[i];
			for (int k = tid; k < i; k += NT) x_out[k] -= A[(size_t)i * np + k] * inv[k] * xi;
			__syncthreads();
		}
		return true;
	}
	// pre-scale for the back-substitution, all threads: A(i,k) <- A(i,k) / d_k for k < i (rows 1 .. n, the border row too), 0 elsewhere
	for (int it = tid; it < (n + 1) * n; it += NT) {
		const int i = it / n, k = it - i * n;
		A[i * np + k] = k < i ? A[i * np + k] * inv[k] : 0.0;
	}
	__syncthreads();
	LDL_T0();
	if (tid < 64) {
		if (n <= 64) ldl_backsub_wave<false>(A, np, n, x_out);
		else ldl_backsub_wave<true>(A, np, n, x_out);
		LDL_ADD(3, 0);
	}
	__syncthreads();
	return true;
}

// G for a system that does not fit LDS (n > ~126: the matrix lives in the window's L2-resident scratch): the same
// factorisation and the same operation order per element as ldl_solve_lds, organised for global-memory latency:
//   - every wave factorises the 6x6 diagonal block redundantly, so all 512 threads finish the panel in one pass;
//   - the finished panel (rows below the block, 6 columns) is staged in LDS: the rank-6 update then costs one global load
//     and one store per element instead of seven loads;
//   - back-substitution stages rows of U in LDS in chunks, so its n sequential steps run on LDS instead of one L2 round
//     trip each.
// lds: the workgroup's free tile region (lds_doubles doubles).  Every thread returns the same flag.
__device__ __forceinline__ bool ldl_solve_global(double* __restrict__ A, int np, int n, double* x_out, double* inv, double* lds, int lds_doubles)
{
	const int tid = threadIdx.x, ty = tid >> 5, tx = tid & 31;
	__shared__ int s_ldl_bad_g;
	double* Pn = lds;                                  // panel rows r0..n, pitch 7
	for (int c0 = 0; c0 < n; c0 += 6) {
		__syncthreads();
		double a[21], ivr[6];
#pragma unroll
		for (int i = 0; i < 6; ++i)
#pragma unroll
			for (int k = 0; k <= i; ++k) a[i * (i + 1) / 2 + k] = A[(size_t)(c0 + i) * np + c0 + k];
		bool bad = false;
#pragma unroll
		for (int j = 0; j < 6; ++j) {
			const double d = a[j * (j + 1) / 2 + j];
			bad |= !(d > 0.0);
			ivr[j] = rcp_f64(d);
#pragma unroll
			for (int i = j + 1; i < 6; ++i) {
				const double t = a[i * (i + 1) / 2 + j] * ivr[j];
#pragma unroll
				for (int k = j + 1; k <= i; ++k) a[i * (i + 1) / 2 + k] -= t * a[k * (k + 1) / 2 + j];
			}
		}
		const int r0 = c0 + 6;
		for (int i = r0 + tid; i <= n; i += NT) {
			double x[6];
#pragma unroll
			for (int k = 0; k < 6; ++k) x[k] = A[(size_t)i * np + c0 + k];
#pragma unroll
			for (int j = 0; j < 5; ++j) {
				const double t = x[j] * ivr[j];
#pragma unroll
				for (int k = j + 1; k < 6; ++k) x[k] -= t * a[k * (k + 1) / 2 + j];
			}
#pragma unroll
			for (int k = 1; k < 6; ++k) A[(size_t)i * np + c0 + k] = x[k];
#pragma unroll
			for (int k = 0; k < 6; ++k) Pn[(i - r0) * 7 + k] = x[k];
		}
		if (tid == 0) {
			s_ldl_bad_g = bad;
#pragma unroll
			for (int i = 0; i < 6; ++i) {
				inv[c0 + i] = ivr[i];
#pragma unroll
				for (int k = 0; k < i; ++k) A[(size_t)(c0 + i) * np + c0 + k] = a[i * (i + 1) / 2 + k];
			}
		}
		__syncthreads();
		if (s_ldl_bad_g) return false;                    // uniform
		for (int ib = r0 + ty; ib <= n; ib += 4 * (NT / 32))
			for (int kb = r0 + tx; kb < n; kb += 64) {
				if (kb - tx > ib - ty + 4 * (NT / 32) - 1) break;    // the whole pass lies above the diagonal (uniform)
				// unconditional loads on clamped indices (a load under a lane condition is compiled as a branch that waits for
				// it at once); only the stores are predicated
				double ui[4][6], uk[2][6], v[8];
#pragma unroll
				for (int u = 0; u < 4; ++u) {
					const int i = min(ib + (NT / 32) * u, n);
#pragma unroll
					for (int c = 0; c < 6; ++c) ui[u][c] = Pn[(i - r0) * 7 + c];
				}
#pragma unroll
				for (int w = 0; w < 2; ++w) {
					const int k = min(kb + 32 * w, n - 1);
#pragma unroll
					for (int c = 0; c < 6; ++c) uk[w][c] = Pn[(k - r0) * 7 + c];
				}
#pragma unroll
				for (int u = 0; u < 4; ++u)
#pragma unroll
					for (int w = 0; w < 2; ++w) v[2 * u + w] = A[(size_t)min(ib + (NT / 32) * u, n) * np + min(kb + 32 * w, n - 1)];
#pragma unroll
				for (int u = 0; u < 4; ++u)
#pragma unroll
					for (int c = 0; c < 6; ++c) ui[u][c] *= ivr[c];
#pragma unroll
				for (int u = 0; u < 4; ++u)
#pragma unroll
					for (int w = 0; w < 2; ++w) {
						const int i = ib + (NT / 32) * u, k = kb + 32 * w;
						if (i <= n && k < n && k <= i) {
							double sacc = 0.0;
#pragma unroll
							for (int c = 0; c < 6; ++c) sacc += ui[u][c] * uk[w][c];
							A[(size_t)i * np + k] = v[2 * u + w] - sacc;
						}
					}
			}
	}
	__syncthreads();
	// x <- D^-1 z, then row i of U updates x_k, k < i, from the last row up; rows staged in LDS `cr` at a time
	for (int k = tid; k < n; k += NT) x_out[k] = A[(size_t)n * np + k] * inv[k];
	int cr = lds_doubles / (n > 0 ? n : 1);
	cr = cr > 64 ? 64 : cr;
	if (cr < 1) {                                      // no room to stage: one L2 round trip per row
		__syncthreads();
		for (int i = n - 1; i > 0; --i) {
			const double xi = x_out[i];
			for (int k = tid; k < i; k += NT) x_out[k] -= A[(size_t)i * np + k] * inv[k] * xi;
			__syncthreads();
		}
		return true;
	}
	for (int hi = n - 1; hi > 0; hi -= cr) {
		const int lo = hi - cr + 1 > 1 ? hi - cr + 1 : 1;          // rows lo..hi
		__syncthreads();
		for (int t = tid; t < (hi - lo + 1) * n; t += NT) {
			const int r = t / n, k = t - r * n, i = lo + r;
			{ const double av = A[(size_t)i * np + k] * inv[k]; lds[r * n + k] = k < i ? av : 0.0; }      // (unconditional load)
		}
		__syncthreads();
		for (int i = hi; i >= lo; --i) {
			const double xi = x_out[i];
			for (int k = tid; k < i; k += NT) x_out[k] -= lds[(i - lo) * n + k] * xi;
			__syncthreads();
		}
	}
	return true;
}

// G for a system in L2 scratch, hybrid: the leading n1 columns are eliminated in global memory (ldl_solve_global's step), then
// the trailing (n - n1) x (n - n1) block -- the part that fits LDS as a PACKED lower triangle, border row included -- is copied
// into LDS once and factorised there at LDS latency.  n1 = 0 when the whole triangle fits (6E <= ~186: the LDS residency that the
// square layout of ldl_solve_lds gives up to 6E = 126).  A global step costs three dependent L2 round trips (~19 us measured), an
// LDS step ~3 us: a 240-unknown system goes from 40 global steps to 9 global + 31 LDS steps.
// Same operation order per element as the other two solvers up to the back-substitution, which is done block-wise:
// x2 (rows >= n1) inside the packed block, then the rows' leading columns (from global memory, staged) onto x1.
#define SSK_PI(i, k) P[(i) * ((i) + 1) / 2 + (k)]
// Back-substitution, the n1 leading unknowns of a system whose trailing unknowns x_out[n1 .. n) are final and whose leading ones hold
// z_k / d_k: the rows' leading columns come from the matrix in global memory, staged in `lds`.
__device__ __forceinline__ bool ldl_backsub_leading(double* __restrict__ A, int np, int n, int n1, double* x_out, double* inv, double* lds, int lds_doubles)
{
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	// ---- step B: the leading unknowns.  The leading columns of the rows come into LDS (the packed block is dead), scaled by the
	//      pivots and zero from the diagonal on, as many rows at a time as fit (all of them up to ~45 edges), from the last row up.
	//      Rows >= n1 are final: unknown k < n1 takes their terms in one pass (a lane per k); the leading triangle goes by 16-row
	//      blocks as above ----
	const int cr = lds_doubles / n1 < n ? lds_doubles / n1 : n;
	for (int hi = n - 1; hi >= 1;) {
		const int lo = hi - cr + 1 > 1 ? hi - cr + 1 : 1;
		__syncthreads();
		for (int t = tid; t < (hi - lo + 1) * n1; t += NT) {
			const int r = t / n1, k = t - r * n1, i = lo + r;
			const double av = A[(size_t)i * np + k] * inv[k];      // (unconditional load)
			lds[t] = k < i ? av : 0.0;
		}
		__syncthreads();
		const int ilo = lo > n1 ? lo : n1;
		if (hi >= ilo) {
			for (int k = tid; k < n1; k += NT) {
				double s0 = 0.0, s1 = 0.0;
				int i = ilo;
				for (; i + 1 <= hi; i += 2) { s0 += lds[(i - lo) * n1 + k] * x_out[i]; s1 += lds[(i + 1 - lo) * n1 + k] * x_out[i + 1]; }
				if (i <= hi) s0 += lds[(i - lo) * n1 + k] * x_out[i];
				x_out[k] -= s0 + s1;
			}
			__syncthreads();
		}
		for (int bh = hi < n1 - 1 ? hi : n1 - 1; bh >= lo;) {
			const int bl = bh - 15 > lo ? bh - 15 : lo, pw = bh - bl + 1;
			if (wave == 0) {
				const int k = lane < pw ? lane : 0;
				double u[16];
#pragma unroll
				for (int i = 0; i < 16; ++i) u[i] = i < pw ? lds[(bl + i - lo) * n1 + bl + k] : 0.0;      // (zero for k >= i)
				double xv = x_out[bl + k];
#pragma unroll
				for (int i = 15; i >= 1; --i) {
					const double xi = readlane_f64(xv, i);
					if (i < pw) xv -= u[i] * xi;
				}
				if (lane < pw) x_out[bl + lane] = xv;
			}
			__syncthreads();
			for (int k = tid; k < bl; k += NT) {
				double sacc = 0.0;
#pragma unroll
				for (int i = 0; i < 16; ++i) sacc += i < pw ? lds[(bl + (i < pw ? i : 0) - lo) * n1 + k] * x_out[bl + (i < pw ? i : 0)] : 0.0;
				x_out[k] -= sacc;
			}
			__syncthreads();
			bh = bl - 1;
		}
		hi = lo - 1;
	}
	return true;
}

// Back-substitution shared by ldl_solve_hybrid and ldl_solve_blocked: the trailing m unknowns inside the packed LDS block P (border row
// = row m), then the n1 leading ones from the matrix in global memory (rows' leading columns staged in `lds`; P is dead by then).
__device__ __forceinline__ bool ldl_backsub_blocks(double* __restrict__ A, int np, int n, int n1, int m, double* P, double* x_out, double* inv, double* lds, int lds_doubles)
{
	const int tid = threadIdx.x;
#ifdef SSK_LM_STAMPS
	LDLH_T();
#endif
	// ---- back-substitution x = D^-1 z - L'^T x by 16-row blocks from the bottom (two barriers per block, not one per row): inside a
	//      block one wavefront (a lane per column, x of the rows above by readlane), then every earlier unknown takes the block's
	//      16 terms at once.  Step A: x2 inside the packed block ----
	const int lane = tid & 63, wave = tid >> 6;
	for (int k = tid; k < m; k += NT) x_out[n1 + k] = SSK_PI(m, k) * inv[n1 + k];
	for (int k = tid; k < n1; k += NT) x_out[k] = A[(size_t)n * np + k] * inv[k];
	__syncthreads();
	for (int c = ((m - 1) >> 4) << 4; c >= 0; c -= 16) {
		const int pw = m - c < 16 ? m - c : 16;
		if (wave == 0) {
			const int k = lane < pw ? lane : 0;
			const double ivk = inv[n1 + c + k];
			double u[16];
#pragma unroll
			for (int i = 0; i < 16; ++i) u[i] = (i < pw && i > k) ? SSK_PI(c + i, c + k) * ivk : 0.0;
			double xv = x_out[n1 + c + k];
#pragma unroll
			for (int i = 15; i >= 1; --i) {
				const double xi = readlane_f64(xv, i);
				if (i < pw) xv -= u[i] * xi;                    // (u[i] = 0 for k >= i)
			}
			if (lane < pw) x_out[n1 + c + lane] = xv;
		}
		__syncthreads();
		for (int k = tid; k < c; k += NT) {
			double sacc = 0.0;
#pragma unroll
			for (int i = 0; i < 16; ++i) { const int ii = c + (i < pw ? i : 0); sacc += i < pw ? SSK_PI(ii, k) * x_out[n1 + ii] : 0.0; }
			x_out[n1 + k] -= sacc * inv[n1 + k];
		}
		__syncthreads();
	}
	if (n1 == 0) { LDLH_ADD(3); return true; }
	const bool ok_ = ldl_backsub_leading(A, np, n, n1, x_out, inv, lds, lds_doubles);
	LDLH_ADD(3);
	return ok_;
}
__device__ __forceinline__ bool ldl_solve_hybrid(double* __restrict__ A, int np, int n, double* x_out, double* inv, double* lds, int lds_doubles)
{
	const int tid = threadIdx.x, ty = tid >> 5, tx = tid & 31;
	__shared__ int s_ldl_bad_h;
	// LDS: [panel of a leading step: rows x 7 | packed trailing block]; the panel area is only needed when there are leading columns
	int n1 = 0;
	while (n1 < n && (long long)(n - n1 + 1) * (n - n1 + 2) / 2 + (n1 ? 7 * (n + 1) : 0) > lds_doubles) n1 += 6;
	if (n1 >= n) return ldl_solve_global(A, np, n, x_out, inv, lds, lds_doubles);      // (no room at all: never with the tiler's regions)
	const int m = n - n1;
	double* Pn = lds;
	double* P = lds + (n1 ? 7 * (n + 1) : 0);
	LDLH_T();
	// ---- phase 1: the trailing block (rows n1 .. n, columns n1 .. row) into LDS, packed: it takes the updates of the leading
	//      steps there, at LDS latency (in global memory a step's update of the whole matrix cost ~19 us, 2/3 of it this block) ----
	for (int r = ty; r <= m; r += NT / 32) {
		const int cend = r < m ? r : m - 1;
		for (int c = tx; c <= cend; c += 32) SSK_PI(r, c) = A[(size_t)(n1 + r) * np + n1 + c];
	}
	LDLH_ADD(0);
	// ---- phase 2: columns 0 .. n1-1 in global memory, six at a time, panel staged in LDS (as ldl_solve_global) ----
	for (int c0 = 0; c0 < n1; c0 += 6) {
		__syncthreads();
		double a[21], ivr[6];
#pragma unroll
		for (int i = 0; i < 6; ++i)
#pragma unroll
			for (int k = 0; k <= i; ++k) a[i * (i + 1) / 2 + k] = A[(size_t)(c0 + i) * np + c0 + k];
		bool bad = false;
#pragma unroll
		for (int j = 0; j < 6; ++j) {
			const double d = a[j * (j + 1) / 2 + j];
			bad |= !(d > 0.0);
			ivr[j] = rcp_f64(d);
#pragma unroll
			for (int i = j + 1; i < 6; ++i) {
				const double t = a[i * (i + 1) / 2 + j] * ivr[j];
#pragma unroll
				for (int k = j + 1; k <= i; ++k) a[i * (i + 1) / 2 + k] -= t * a[k * (k + 1) / 2 + j];
			}
		}
		const int r0 = c0 + 6;
		for (int i = r0 + tid; i <= n; i += NT) {
			double x[6];
#pragma unroll
			for (int k = 0; k < 6; ++k) x[k] = A[(size_t)i * np + c0 + k];
#pragma unroll
			for (int j = 0; j < 5; ++j) {
				const double t = x[j] * ivr[j];
#pragma unroll
				for (int k = j + 1; k < 6; ++k) x[k] -= t * a[k * (k + 1) / 2 + j];
			}
#pragma unroll
			for (int k = 1; k < 6; ++k) A[(size_t)i * np + c0 + k] = x[k];
#pragma unroll
			for (int k = 0; k < 6; ++k) Pn[(i - r0) * 7 + k] = x[k];
		}
		if (tid == 0) {
			s_ldl_bad_h = bad;
#pragma unroll
			for (int i = 0; i < 6; ++i) {
				inv[c0 + i] = ivr[i];
#pragma unroll
				for (int k = 0; k < i; ++k) A[(size_t)(c0 + i) * np + c0 + k] = a[i * (i + 1) / 2 + k];
			}
		}
		__syncthreads();
		if (s_ldl_bad_h) return false;                    // uniform
		// (a) the leading columns still to come, r0 <= k < n1, in global memory: 4 x 2 elements per thread and pass
		if (r0 < n1)
		for (int ib = r0 + ty; ib <= n; ib += 4 * (NT / 32))
			for (int kb = r0 + tx; kb < n1; kb += 64) {
				if (kb - tx > ib - ty + 4 * (NT / 32) - 1) break;    // the whole pass lies above the diagonal (uniform)
				// unconditional loads on clamped indices (a load under a lane condition is compiled as a branch that waits for
				// it at once); only the stores are predicated
				double ui[4][6], uk[2][6], v[8];
#pragma unroll
				for (int u = 0; u < 4; ++u) {
					const int i = min(ib + (NT / 32) * u, n);
#pragma unroll
					for (int c = 0; c < 6; ++c) ui[u][c] = Pn[(i - r0) * 7 + c];
				}
#pragma unroll
				for (int w = 0; w < 2; ++w) {
					const int k = min(kb + 32 * w, n1 - 1);
#pragma unroll
					for (int c = 0; c < 6; ++c) uk[w][c] = Pn[(k - r0) * 7 + c];
				}
#pragma unroll
				for (int u = 0; u < 4; ++u)
#pragma unroll
					for (int w = 0; w < 2; ++w) v[2 * u + w] = A[(size_t)min(ib + (NT / 32) * u, n) * np + min(kb + 32 * w, n1 - 1)];
#pragma unroll
				for (int u = 0; u < 4; ++u)
#pragma unroll
					for (int c = 0; c < 6; ++c) ui[u][c] *= ivr[c];
#pragma unroll
				for (int u = 0; u < 4; ++u)
#pragma unroll
					for (int w = 0; w < 2; ++w) {
						const int i = ib + (NT / 32) * u, k = kb + 32 * w;
						if (i <= n && k < n1 && k <= i) {
							double sacc = 0.0;
#pragma unroll
							for (int c = 0; c < 6; ++c) sacc += ui[u][c] * uk[w][c];
							A[(size_t)i * np + k] = v[2 * u + w] - sacc;
						}
					}
			}
		// (b) the trailing block in LDS: rows n1 + ii, columns n1 + kk, kk <= min(ii, m - 1); panel row of matrix row i: i - r0
		const int po = n1 - r0;
		for (int ib = ty; ib <= m; ib += 4 * (NT / 32))
			for (int kb = tx; kb < m; kb += 64) {
				if (kb - tx > ib - ty + 4 * (NT / 32) - 1) break;    // the whole pass lies above the diagonal (uniform)
				// an (ii, kk) above the diagonal reads some other entry of the packed triangle (inside the allocation), never stored
				double ui[4][6], uk[2][6], v[8];
#pragma unroll
				for (int u = 0; u < 4; ++u) {
					const int i = min(ib + (NT / 32) * u, m);
#pragma unroll
					for (int c = 0; c < 6; ++c) ui[u][c] = Pn[(i + po) * 7 + c];
				}
#pragma unroll
				for (int w = 0; w < 2; ++w) {
					const int k = min(kb + 32 * w, m - 1);
#pragma unroll
					for (int c = 0; c < 6; ++c) uk[w][c] = Pn[(k + po) * 7 + c];
				}
#pragma unroll
				for (int u = 0; u < 4; ++u)
#pragma unroll
					for (int w = 0; w < 2; ++w) {
						const int i = min(ib + (NT / 32) * u, m), k = min(kb + 32 * w, m - 1);
						v[2 * u + w] = SSK_PI(i, k);
					}
#pragma unroll
				for (int u = 0; u < 4; ++u)
#pragma unroll
					for (int c = 0; c < 6; ++c) ui[u][c] *= ivr[c];
#pragma unroll
				for (int u = 0; u < 4; ++u)
#pragma unroll
					for (int w = 0; w < 2; ++w) {
						const int i = ib + (NT / 32) * u, k = kb + 32 * w;
						if (i <= m && k < m && k <= i) {
							double sacc = 0.0;
#pragma unroll
							for (int c = 0; c < 6; ++c) sacc += ui[u][c] * uk[w][c];
							SSK_PI(i, k) = v[2 * u + w] - sacc;
						}
					}
			}
	}
	// ---- phase 3: blocked LDL^T of the packed block with look-ahead (ldl_solve_lds's factorisation on packed indices) ----
	__syncthreads();
	LDLH_ADD(1);
	if (!ldl_factor_lookahead(P, ldl_ix_packed{}, m, inv + n1, m)) return false;
	__syncthreads();
	LDLH_ADD(2);
	return ldl_backsub_blocks(A, np, n, n1, m, P, x_out, inv, lds, lds_doubles);
}

// index map of a block column in LDS: rows x (w columns), odd pitch
struct ldl_ix_rect { int pitch; __device__ __forceinline__ int operator()(int i, int k) const { return i * pitch + k; } };

// C(i, k) -= sum_q S(i, q) inv[q] S(k, q) over a staged chunk of CW finished columns (S: rows of the target, pitch CW + 1; row k of S
// is the matrix row of target column k), for the target's lower part k <= i, k < ncol, i <= nr.  A wavefront takes a column pair,
// its lanes the rows: the S(k, .) reads are broadcasts, S(i, .) and C(i, k) conflict free (odd pitches / triangular offsets).
template <int CW, typename IX>
__device__ __forceinline__ void ldl_chunk_update(double* C, const IX ix, int ncol, int nr, const double* S, const double* ivq, int nq)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	for (int k0 = 2 * wave; k0 < ncol; k0 += 2 * (NT / 64)) {
		const int k1 = k0 + 1 < ncol ? k0 + 1 : k0;
		double sk0[CW], sk1[CW];
#pragma unroll
		for (int q = 0; q < CW; ++q) {
			const double iv = q < nq ? ivq[q] : 0.0;      // (a last chunk of fewer columns: its staging holds zeros there)
			sk0[q] = S[k0 * (CW + 1) + q] * iv; sk1[q] = S[k1 * (CW + 1) + q] * iv;
		}
		for (int ib = (k0 >> 6) << 6; ib <= nr; ib += 64) {
			const int i0 = min(ib + lane, nr);
			double a0[CW];
#pragma unroll
			for (int q = 0; q < CW; ++q) a0[q] = S[i0 * (CW + 1) + q];
			const double c00 = C[ix(i0, k0)], c01 = C[ix(i0, k1)];
			double s00 = 0, s01 = 0;
#pragma unroll
			for (int q = 0; q < CW; ++q) { s00 += a0[q] * sk0[q]; s01 += a0[q] * sk1[q]; }
			const bool r0 = ib + lane <= nr;
			if (r0 && i0 >= k0) C[ix(i0, k0)] = c00 - s00;
			if (r0 && i0 >= k1 && k1 != k0) C[ix(i0, k1)] = c01 - s01;
		}
	}
}

// The updates of a target in LDS (rows r0 .. n of the matrix, ncol columns under the index map) by the finished columns 0 .. c - 1,
// from CW-column chunks staged through S (rows x (CW + 1)): one parallel load per chunk, the next chunk in flight in registers
// during the update.
template <int CW, typename IX>
__device__ __forceinline__ void ldl_apply_leading(const double* __restrict__ A, int np, int n, const double* inv, const IX ixf, double* C, int ncol, int r0, int c, double* S)
{
	constexpr int SR = 8;                              // elements per thread and leg (rows x CW <= 8 NT: 340 rows at 512 threads, 170 at 256; more: further legs)
	const int tid = threadIdx.x, nr = n - r0, cnt = (nr + 1) * CW, legs = (cnt + SR * NT - 1) / (SR * NT);
	auto chunk_load = [&](int p0, int leg, double (&v)[SR]) {
#pragma unroll
		for (int j = 0; j < SR; ++j) {
			const int e = min(tid + NT * (j + SR * leg), cnt - 1), r = e / CW, q = e - r * CW;
			v[j] = A[(size_t)(r0 + r) * np + p0 + q];
		}
	};
	auto chunk_store = [&](int nq, int leg, const double (&v)[SR]) {
#pragma unroll
		for (int j = 0; j < SR; ++j) {
			const int e = tid + NT * (j + SR * leg), r = e / CW, q = e - r * CW;
			if (e < cnt) S[r * (CW + 1) + q] = q < nq ? v[j] : 0.0;
		}
	};
	double v[SR];
	if (c > 0) chunk_load(0, 0, v);
	for (int p0 = 0; p0 < c; p0 += CW) {
		__syncthreads();                               // the staging is free (and, first turn, the target is in LDS)
		const int nq = c - p0 < CW ? c - p0 : CW;
		chunk_store(nq, 0, v);
		for (int leg = 1; leg < legs; ++leg) { chunk_load(p0, leg, v); chunk_store(nq, leg, v); }
		if (p0 + CW < c) chunk_load(p0 + CW, 0, v);
		__syncthreads();
		ldl_chunk_update<CW>(C, ixf, ncol, nr, S, inv + p0, nq);
	}
	__syncthreads();
}

// The leading n1 columns of a system in L2 scratch, left-looking by block columns: a block column (rows c .. n, as many columns as
// fit LDS beside the staging) is loaded ONCE, takes the updates of the columns before it (ldl_apply_leading), is factorised at LDS
// latency (ldl_factor_lookahead on the rectangle) and is stored once.  0: done, 1: a pivot <= 0, 2: a block column does not fit.
template <int CW>
__device__ __forceinline__ int ldl_leading_panels(double* __restrict__ A, int np, int n, int n1, double* inv, double* lds, int lds_doubles)
{
	const int tid = threadIdx.x;
	// (every block column must fit before the first one is touched: a caller falls back to another solver on 2)
	for (int c = 0; c < n1;) {
		const int R = n - c + 1;
		int w = ((lds_doubles - (c ? (CW + 1) * R : 0)) / R - 1) / 6 * 6;
		if (w > n1 - c) w = n1 - c;
		if (w < 6) return 2;                               // (a system far beyond the tiler's sizes)
		c += w;
	}
	for (int c = 0; c < n1;) {
		const int R = n - c + 1;
		const int stage = c ? (CW + 1) * R : 0;
		int w = ((lds_doubles - stage) / R - 1) / 6 * 6;
		if (w > n1 - c) w = n1 - c;
		const ldl_ix_rect ix{w + 1};
		double* Pn = lds;
		double* S = lds + (size_t)R * (w + 1);
		__syncthreads();
		for (int e0 = 0; e0 < R * w; e0 += 8 * NT) {       // (eight loads in flight per thread)
			double v[8];
#pragma unroll
			for (int j = 0; j < 8; ++j) { const int e = min(e0 + tid + NT * j, R * w - 1), r = e / w, k = e - r * w; v[j] = A[(size_t)(c + r) * np + c + k]; }
#pragma unroll
			for (int j = 0; j < 8; ++j) { const int e = e0 + tid + NT * j, r = e / w, k = e - r * w; if (e < R * w) Pn[r * (w + 1) + k] = v[j]; }
		}
		ldl_apply_leading<CW>(A, np, n, inv, ix, Pn, w, c, c, S);
		__syncthreads();
		if (!ldl_factor_lookahead(Pn, ix, w, inv + c, R - 1)) return 1;
		__syncthreads();
		for (int e = tid; e < R * w; e += NT) { const int r = e / w, k = e - r * w; if (k <= r) A[(size_t)(c + r) * np + c + k] = Pn[r * (w + 1) + k]; }
		c += w;
	}
	__syncthreads();                                   // (the last block column's stores are issued; its LDS copy is dead)
	return 0;
}

// G for a system in L2 scratch, BLOCKED: like ldl_solve_hybrid the trailing block is factorised in LDS as a packed triangle, but the
// leading n1 columns no longer cost one global-memory step per six columns (three dependent L2 round trips and two barriers each):
// ldl_leading_panels; the packed trailing block then takes the updates of ALL leading columns the same way and is factorised in place.
// Per element the same operations as the other solvers; the order of the update terms is by column (ascending) as there.
// (Measured equal to ldl_solve_hybrid on a 240-unknown system -- the in-LDS factorisation is bound by LDS bandwidth, 6.5 accesses per
// element and six-column step -- which is why the default is ldl_solve_regtile.)
__device__ __forceinline__ bool ldl_solve_blocked(double* __restrict__ A, int np, int n, double* x_out, double* inv, double* lds, int lds_doubles)
{
	constexpr int CW = 12;
	const int tid = threadIdx.x;
	int m = n;
	while (m > 0 && (long long)(m + 1) * (m + 2) / 2 + (m < n ? (long long)(CW + 1) * (m + 1) : 0) > lds_doubles) m -= 6;
	const int n1 = n - m;
	if (m <= 0) return ldl_solve_hybrid(A, np, n, x_out, inv, lds, lds_doubles);
	LDLH_T();
	const int lead = ldl_leading_panels<CW>(A, np, n, n1, inv, lds, lds_doubles);
	if (lead == 1) return false;
	if (lead == 2) return ldl_solve_hybrid(A, np, n, x_out, inv, lds, lds_doubles);
	LDLH_ADD(0);
	// ---- the trailing block: packed in LDS, the leading columns' updates, the factorisation in place ----
	double* P = lds;
	double* S = lds + (size_t)(m + 1) * (m + 2) / 2;
	for (int r = tid >> 5; r <= m; r += NT / 32) {
		const int cend = r < m ? r : m - 1;
		for (int k = tid & 31; k <= cend; k += 32) SSK_PI(r, k) = A[(size_t)(n1 + r) * np + n1 + k];
	}
	if (n1 > 0) ldl_apply_leading<CW>(A, np, n, inv, ldl_ix_packed{}, P, m, n1, n1, S);
	__syncthreads();
	LDLH_ADD(1);
	if (!ldl_factor_lookahead(P, ldl_ix_packed{}, m, inv + n1, m)) return false;
	__syncthreads();
	LDLH_ADD(2);
	return ldl_backsub_blocks(A, np, n, n1, m, P, x_out, inv, lds, lds_doubles);
}

// G for a system in L2 scratch, the trailing block in REGISTERS: the lower triangle of the last m <= 246 unknowns is cut into 8 x 8
// tiles, one per thread (31 x 32 / 2 = 496 tiles at 512 threads; column-major thread order, so that whole wavefronts retire as the
// factorisation advances), 64 doubles each.  A step eliminates the 8 columns of one tile column:
//   A  the thread of the diagonal tile factorises it (L' D L'^T in its registers), eliminates the right-hand side's 8 entries (it owns
//      them) and publishes the factor;                                                      -- barrier
//   B  the threads of the tiles below eliminate their 8 rows against it and publish them, tile-major (lane = tile row: no conflicts);
//                                                                                            -- barrier
//   C  every thread right of the column: tile -= U_I D^-1 U_J^T, 512 multiply-adds against 128 LDS reads (the LDS-resident
//      factorisation pays 6.5 LDS accesses per 6 multiply-adds and is bound by them); double buffers: no third barrier.
// The leading n - m columns (systems beyond 41 edges) go by ldl_leading_panels, their updates reach the tiles through staged chunks.
// Back-substitution from the registers: per tile row one 8 x 8 triangle by the diagonal thread, then the row's tiles hand their
// 8-term partial sums to the diagonal threads of their columns (fixed order: bit-reproducible).
// Same operations per element as the other solvers up to the order of an update's terms (term by term here, not a 6-term sum first).
__device__ __forceinline__ bool ldl_solve_regtile(double* __restrict__ A, int np, int n, double* x_out, double* inv, double* lds, int lds_doubles)
{
	constexpr int CW = 12, TP = 33;
	constexpr int TMAX = NT >= 496 ? 31 : (NT >= 253 ? 22 : (NT >= 120 ? 15 : 10));      // T (T + 1) / 2 <= NT
	constexpr int L_S2 = 8 * CW * TP, L_SZ = 16, L_PN = 64 * TP, L_DG = 72, L_ZP = 8, L_XS = 8 * 32, L_PART = 8 * 32;
	constexpr int L_TOTAL = L_S2 + L_SZ + 2 * L_PN + 2 * L_DG + 2 * L_ZP + L_XS + L_PART;
	if (lds_doubles < L_TOTAL) return ldl_solve_blocked(A, np, n, x_out, inv, lds, lds_doubles);
	const int tid = threadIdx.x;
	const int m = n < 8 * TMAX / 6 * 6 ? n : 8 * TMAX / 6 * 6, n1 = n - m, T = (m + 7) >> 3;
	__shared__ int s_ldl_bad_r;
	LDLH_T();
	if (n1 > 0) {
		const int lead = ldl_leading_panels<CW>(A, np, n, n1, inv, lds, lds_doubles);
		if (lead == 1) return false;
		if (lead == 2) return ldl_solve_blocked(A, np, n, x_out, inv, lds, lds_doubles);
	}
	LDLH_ADD(0);
	double* S2 = lds;                                  // staged chunk of leading columns, tile-major: [(row in tile) * CW + q][tile row]
	double* Sz = S2 + L_S2;                            // ... of the right-hand side
	double* Pn2 = Sz + L_SZ;                           // the eliminated tile column: [buffer][(row in tile) * 8 + c][tile row]
	double* dg = Pn2 + 2 * L_PN;                       // [buffer]: the diagonal tile's factor (64) and 1 / pivots (8)
	double* Zp = dg + 2 * L_DG;                        // [buffer]: the right-hand side's eliminated entries of the column
	double* xs = Zp + 2 * L_ZP;                        // the solution of the trailing block
	double* part = xs + L_XS;                          // partial sums of the back-substitution, one 8-vector per tile column
	// thread -> tile (I, J), I >= J, column-major
	int J = 0, I = 0;
	{ int idx = tid; while (J < T && idx >= T - J) { idx -= T - J; ++J; } I = J + idx; }
	const bool active = J < T;
	if (!active) { I = 0; J = 0; }
	const bool diag = active && I == J;
	double t[8][8], z[8];
#pragma unroll
	for (int a = 0; a < 8; ++a)
#pragma unroll
		for (int b = 0; b < 8; ++b) {
			const int r = 8 * I + a, k = 8 * J + b;
			const double v = A[(size_t)(n1 + min(r, m - 1)) * np + n1 + min(k, m - 1)];      // (unconditional loads on clamped indices)
			t[a][b] = (r < m && k < m) ? v : (r == k ? 1.0 : 0.0);                             // the padding of the last tile row: identity
		}
#pragma unroll
	for (int b = 0; b < 8; ++b) { const int k = 8 * J + b; const double v = A[(size_t)n * np + n1 + min(k, m - 1)]; z[b] = k < m ? v : 0.0; }
	if (tid == 0) s_ldl_bad_r = 0;
	if (n1 > 0) {
		// ---- the leading columns' updates: chunks of CW columns, rows n1 .. n staged tile-major ----
		for (int e = tid; e < L_S2; e += NT) S2[e] = 0.0;      // (the padding rows stay zero)
		const int cnt = (m + 1) * CW;
		for (int p0 = 0; p0 < n1; p0 += CW) {
			const int nq = n1 - p0 < CW ? n1 - p0 : CW;
			__syncthreads();
			for (int e0 = 0; e0 < cnt; e0 += 8 * NT) {
				double v[8];
#pragma unroll
				for (int j = 0; j < 8; ++j) { const int e = min(e0 + tid + NT * j, cnt - 1), r = e / CW, q = e - r * CW; v[j] = A[(size_t)(n1 + r) * np + p0 + q]; }
#pragma unroll
				for (int j = 0; j < 8; ++j) {
					const int e = e0 + tid + NT * j, r = e / CW, q = e - r * CW;
					const double x = q < nq ? v[j] : 0.0;
					if (e < cnt) { if (r < m) S2[((r & 7) * CW + q) * TP + (r >> 3)] = x; else Sz[q] = x; }
				}
			}
			__syncthreads();
			if (active) {
#pragma unroll 1
				for (int q = 0; q < CW; ++q) {
					const double ivq = q < nq ? inv[p0 + q] : 0.0;
					double ui[8], uj[8];
#pragma unroll
					for (int a = 0; a < 8; ++a) { ui[a] = S2[(a * CW + q) * TP + I] * ivq; uj[a] = S2[(a * CW + q) * TP + J]; }
#pragma unroll
					for (int a = 0; a < 8; ++a)
#pragma unroll
						for (int b = 0; b < 8; ++b) t[a][b] -= ui[a] * uj[b];
					if (diag) {
						const double uz = Sz[q] * ivq;
#pragma unroll
						for (int b = 0; b < 8; ++b) z[b] -= uz * uj[b];
					}
				}
			}
		}
	}
	__syncthreads();
	LDLH_ADD(1);
	// ---- the factorisation, a tile column per step ----
	LDLW_T();
	for (int J0 = 0; J0 < T; ++J0) {
		const int buf = J0 & 1;
		double* dgb = dg + buf * L_DG;
		double* pnb = Pn2 + buf * L_PN;
		double* zpb = Zp + buf * L_ZP;
		if (diag && J == J0) {
			double iv[8];
			bool bad = false;
#pragma unroll
			for (int j = 0; j < 8; ++j) {
				const double d = t[j][j];
				bad |= !(d > 0.0);
				iv[j] = rcp_f64(d);
#pragma unroll
				for (int i = j + 1; i < 8; ++i) {
					const double f = t[i][j] * iv[j];
#pragma unroll
					for (int k = j + 1; k <= i; ++k) t[i][k] -= f * t[k][j];
				}
			}
#pragma unroll
			for (int j = 0; j < 7; ++j) {
				const double f = z[j] * iv[j];
#pragma unroll
				for (int k = j + 1; k < 8; ++k) z[k] -= f * t[k][j];
			}
#pragma unroll
			for (int i = 0; i < 8; ++i) {
#pragma unroll
				for (int j = 0; j < i; ++j) dgb[8 * i + j] = t[i][j];
				dgb[64 + i] = iv[i];
				zpb[i] = z[i];
				if (8 * J0 + i < m) inv[n1 + 8 * J0 + i] = iv[i];
			}
			if (bad) s_ldl_bad_r = 1;
		}
		LDLW_ADD(0);
		__syncthreads();
		LDLW_ADD(3);
		LDLH_ADD(4);
		if (s_ldl_bad_r) return false;                     // uniform
		if (active && J == J0 && I > J0) {
			double iv[8];
#pragma unroll
			for (int j = 0; j < 8; ++j) iv[j] = dgb[64 + j];
#pragma unroll
			for (int j = 0; j < 7; ++j) {
				double ud[8];
#pragma unroll
				for (int k = j + 1; k < 8; ++k) ud[k] = dgb[8 * k + j];
#pragma unroll
				for (int a = 0; a < 8; ++a) {
					const double f = t[a][j] * iv[j];
#pragma unroll
					for (int k = j + 1; k < 8; ++k) t[a][k] -= f * ud[k];
				}
			}
#pragma unroll
			for (int a = 0; a < 8; ++a)
#pragma unroll
				for (int c = 0; c < 8; ++c) pnb[(8 * a + c) * TP + I] = t[a][c];
		}
		LDLW_ADD(1);
		__syncthreads();
		LDLW_ADD(3);
		LDLH_ADD(5);
		if (active && J > J0) {
			// (not software-pipelined: with the operands of column c + 1 in flight beside the 64 multiply-adds of column c the tile, the
			//  right-hand side and two operand sets exceed the 256 registers of two waves per SIMD -- the compiler spills z inside the loop,
			//  7.7 ms against 4.05 ms for one depth-5 window)
#pragma unroll SSK_RT_UNROLL
			for (int c = 0; c < 8; ++c) {
				const double ivc = dgb[64 + c];
				double ui[8], uj[8];
#pragma unroll
				for (int a = 0; a < 8; ++a) { ui[a] = pnb[(8 * a + c) * TP + I] * ivc; uj[a] = pnb[(8 * a + c) * TP + J]; }
#pragma unroll
				for (int a = 0; a < 8; ++a)
#pragma unroll
					for (int b = 0; b < 8; ++b) t[a][b] -= ui[a] * uj[b];
				if (diag) {
					const double uz = zpb[c] * ivc;
#pragma unroll
					for (int b = 0; b < 8; ++b) z[b] -= uz * uj[b];
				}
			}
		}
		LDLW_ADD(2);
	}
	LDLH_ADD(6);
	// ---- back-substitution x = D^-1 z - L'^T x, a tile row per step from the bottom ----
	double acc[8];
#pragma unroll
	for (int b = 0; b < 8; ++b) acc[b] = 0.0;
	for (int J0 = T - 1; J0 >= 0; --J0) {
		if (diag && J == J0) {
			double x[8];
#pragma unroll
			for (int b = 7; b >= 0; --b) {
				double sacc = z[b] - acc[b];
#pragma unroll
				for (int a = b + 1; a < 8; ++a) sacc -= t[a][b] * x[a];
				x[b] = sacc * (8 * J0 + b < m ? inv[n1 + 8 * J0 + b] : 1.0);
				xs[8 * J0 + b] = x[b];
			}
		}
		__syncthreads();
		if (active && I == J0 && J < J0) {
			double xa[8];
#pragma unroll
			for (int a = 0; a < 8; ++a) xa[a] = xs[8 * J0 + a];
#pragma unroll
			for (int b = 0; b < 8; ++b) {
				double p = 0.0;
#pragma unroll
				for (int a = 0; a < 8; ++a) p += t[a][b] * xa[a];
				part[8 * J + b] = p;
			}
		}
		__syncthreads();
		if (diag && J < J0) {
#pragma unroll
			for (int b = 0; b < 8; ++b) acc[b] += part[8 * J + b];
		}
	}
	__syncthreads();
	for (int k = tid; k < m; k += NT) x_out[n1 + k] = xs[k];
	for (int k = tid; k < n1; k += NT) x_out[k] = A[(size_t)n * np + k] * inv[k];
	__syncthreads();
	bool ok_ = true;
	if (n1 > 0) ok_ = ldl_backsub_leading(A, np, n, n1, x_out, inv, lds, lds_doubles);
	LDLH_ADD(3);
	return ok_;
}
// The register-tile factorisation for a reduced system whose block accumulators are in LDS (the node kernel, E <= 14): 4 x 4 tiles, one
// per thread (T (T + 1) / 2 <= NT tiles, T = ceil(6E / 4) <= 22 at 256 threads), loaded STRAIGHT from the 6 x 6 block accumulators Hb
// (H(6e + a, 6e2 + b), e <= e2, at 36 pid + 6a + b; lambda added on the diagonal; right-hand side g) -- no dense copy of the system.
// Steps A / B / C and the back-substitution as in ldl_solve_regtile; a step costs ~1.8 k cycles against ~7 k of the in-LDS six-column
// step (whose rank-6 update is bound by LDS bandwidth).  lds: 2 x (16 T' + 24) + 8 T' doubles, T' = 23.
__device__ __forceinline__ bool ldl_solve_tiles4_lds(const double* Hb, const double* g, double lambda, int n, double* x_out, double* lds)
{
	constexpr int TP = 23;
	constexpr int L_PN = 16 * TP, L_DG = 20, L_ZP = 4;
	// (the thread number made opaque: inside a caller's LM loop the tile coordinates and LDS addresses derived from a loop-invariant thread number are
	// hoisted in front of that loop and held -- or spilled -- for the whole kernel)
	int tid = threadIdx.x;
	asm volatile("" : "+v"(tid));
	const int T = (n + 3) >> 2;
	__shared__ int s_ldl_bad_t;
	double* Pn2 = lds;                                 // [buffer][(row in tile) * 4 + c][tile row]
	double* dg = Pn2 + 2 * L_PN;                       // [buffer]: the diagonal tile's factor (16) and 1 / pivots (4)
	double* Zp = dg + 2 * L_DG;                        // [buffer]: the right-hand side's eliminated entries of the column
	double* xs = Zp + 2 * L_ZP;                        // the solution (4 T)
	double* part = xs + 4 * TP;                        // partial sums of the back-substitution, one 4-vector per tile column
	int J = 0, I = 0;
	{ int idx = tid; while (J < T && idx >= T - J) { idx -= T - J; ++J; } I = J + idx; }
	const bool active = J < T;
	if (!active) { I = 0; J = 0; }
	const bool diag = active && I == J;
	double t[4][4], z[4], ivd[4];
#pragma unroll
	for (int a = 0; a < 4; ++a)
#pragma unroll
		for (int b = 0; b < 4; ++b) {
			const int r0 = 4 * I + a, c0 = 4 * J + b;
			const int r = min(max(r0, c0), n - 1), c = min(min(r0, c0), n - 1);      // the stored triangle: H(c, r), c <= r
			const int er = r / 6, ec = c / 6;
			const double v = Hb[36 * (er * (er + 1) / 2 + ec) + 6 * (c - 6 * ec) + (r - 6 * er)];
			t[a][b] = (r0 < n && c0 < n) ? (r0 == c0 ? v + lambda : v) : (r0 == c0 ? 1.0 : 0.0);
		}
#pragma unroll
	for (int b = 0; b < 4; ++b) { const int c0 = 4 * J + b; const double v = g[min(c0, n - 1)]; z[b] = c0 < n ? v : 0.0; ivd[b] = 1.0; }
	if (tid == 0) s_ldl_bad_t = 0;
	__syncthreads();
	for (int J0 = 0; J0 < T; ++J0) {
		const int buf = J0 & 1;
		double* dgb = dg + buf * L_DG;
		double* pnb = Pn2 + buf * L_PN;
		double* zpb = Zp + buf * L_ZP;
		if (diag && J == J0) {
			bool bad = false;
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const double d = t[j][j];
				bad |= !(d > 0.0);
				ivd[j] = rcp_f64(d);
#pragma unroll
				for (int i = j + 1; i < 4; ++i) {
					const double f = t[i][j] * ivd[j];
#pragma unroll
					for (int k = j + 1; k <= i; ++k) t[i][k] -= f * t[k][j];
				}
			}
#pragma unroll
			for (int j = 0; j < 3; ++j) {
				const double f = z[j] * ivd[j];
#pragma unroll
				for (int k = j + 1; k < 4; ++k) z[k] -= f * t[k][j];
			}
#pragma unroll
			for (int i = 0; i < 4; ++i) {
#pragma unroll
				for (int j = 0; j < i; ++j) dgb[4 * i + j] = t[i][j];
				dgb[16 + i] = ivd[i];
				zpb[i] = z[i];
			}
			if (bad) s_ldl_bad_t = 1;
		}
		__syncthreads();
		if (s_ldl_bad_t) return false;                     // uniform
		if (active && J == J0 && I > J0) {
			double iv[4];
#pragma unroll
			for (int j = 0; j < 4; ++j) iv[j] = dgb[16 + j];
#pragma unroll
			for (int j = 0; j < 3; ++j) {
				double ud[4];
#pragma unroll
				for (int k = j + 1; k < 4; ++k) ud[k] = dgb[4 * k + j];
#pragma unroll
				for (int a = 0; a < 4; ++a) {
					const double f = t[a][j] * iv[j];
#pragma unroll
					for (int k = j + 1; k < 4; ++k) t[a][k] -= f * ud[k];
				}
			}
#pragma unroll
			for (int a = 0; a < 4; ++a)
#pragma unroll
				for (int c = 0; c < 4; ++c) pnb[(4 * a + c) * TP + I] = t[a][c];
		}
		__syncthreads();
		if (active && J > J0) {
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				const double ivc = dgb[16 + c];
				double ui[4], uj[4];
#pragma unroll
				for (int a = 0; a < 4; ++a) { ui[a] = pnb[(4 * a + c) * TP + I] * ivc; uj[a] = pnb[(4 * a + c) * TP + J]; }
#pragma unroll
				for (int a = 0; a < 4; ++a)
#pragma unroll
					for (int b = 0; b < 4; ++b) t[a][b] -= ui[a] * uj[b];
				if (diag) {
					const double uz = zpb[c] * ivc;
#pragma unroll
					for (int b = 0; b < 4; ++b) z[b] -= uz * uj[b];
				}
			}
		}
	}
	// ---- back-substitution x = D^-1 z - L'^T x, a tile row per step from the bottom ----
	double acc[4];
#pragma unroll
	for (int b = 0; b < 4; ++b) acc[b] = 0.0;
	for (int J0 = T - 1; J0 >= 0; --J0) {
		if (diag && J == J0) {
			double x[4];
#pragma unroll
			for (int b = 3; b >= 0; --b) {
				double sacc = z[b] - acc[b];
#pragma unroll
				for (int a = b + 1; a < 4; ++a) sacc -= t[a][b] * x[a];
				x[b] = sacc * ivd[b];
				xs[4 * J0 + b] = x[b];
			}
		}
		__syncthreads();
		if (active && I == J0 && J < J0) {
			double xa[4];
#pragma unroll
			for (int a = 0; a < 4; ++a) xa[a] = xs[4 * J0 + a];
#pragma unroll
			for (int b = 0; b < 4; ++b) {
				double p = 0.0;
#pragma unroll
				for (int a = 0; a < 4; ++a) p += t[a][b] * xa[a];
				part[4 * J + b] = p;
			}
		}
		__syncthreads();
		if (diag && J < J0) {
#pragma unroll
			for (int b = 0; b < 4; ++b) acc[b] += part[4 * J + b];
		}
	}
	__syncthreads();
	for (int k = tid; k < n; k += NT) x_out[k] = xs[k];
	__syncthreads();
	return true;
}
// The reduced system of a window with E <= 10 optimised edges (the bench windows: "local LM window = 10") by ONE wavefront with NO workgroup
// barrier: the lower block triangle's E (E + 1) / 2 <= 55 blocks of 6 x 6 -- exactly the block accumulators Hb -- one per LANE (36 doubles in
// registers; lane = block pair id).  A step eliminates a block column: (A) the diagonal block's lane factorises it (and eliminates the right-hand
// side's six entries: it owns them) and publishes U, 1 / d, z; (B) the lanes below eliminate their six rows against it and publish them; (C) every
// block right of the column takes block -= U_I D^-1 U_J^T (216 multiply-adds against 72 LDS reads, two of three as broadcasts).  The phases of a
// step are ordered by the wavefront's own LDS counter (WAVE_SYNC), ten steps instead of ldl_solve_tiles4_lds's fifteen with three workgroup
// barriers each and two per back-substitution step: a barrier of four wavefronts beside another workgroup's costs ~1 k cycles, and the 75 of them
// were most of that solver's ~70 k cycles.  The other wavefronts of the workgroup wait at the final barrier (asleep: no issue slots).
// Same factorisation (right-looking L' D L'^T, U = L' D unscaled) and the same sums in the same order for a given E: bit-reproducible.
// lds: LDLB6_DOUBLES doubles.
constexpr int LDLB6_BP = 37;            // pitch of a published block (odd: the panel's lanes write side by side)
constexpr int LDLB6_DOUBLES = 48 + 10 * LDLB6_BP + 60 + 60;
__device__ __forceinline__ bool ldl_solve_blocks6_wave(const double* Hb, const double* g, double lambda, int E, double* x_out, double* lds)
{
	__shared__ int s_ldl_bad_b;
	double* Ud = lds;                                  // the diagonal block's U below the diagonal (6 i + j), 1 / d (36 ..), z (42 ..)
	double* Pn = Ud + 48;                              // the column's eliminated blocks, [I][6 a + c]
	double* xs = Pn + 10 * LDLB6_BP;                   // the solution
	double* part = xs + 60;                            // back-substitution: a block row's contributions, [J][6]
	const int tid = threadIdx.x, n = 6 * E;
	if (tid < 64) {
		int I = 0;
		while ((I + 1) * (I + 2) / 2 <= tid) ++I;
		const int J = tid - I * (I + 1) / 2;
		const bool active = I < E, diag = active && I == J;
		double B[6][6], z[6], iv[6], acc[6];
		{
			const double* hb = Hb + 36 * (active ? tid : 0);       // B(a, b) = A(6 I + a, 6 J + b) = H(6 J + b, 6 I + a): the stored (upper) triangle
#pragma unroll
			for (int b = 0; b < 6; ++b)
#pragma unroll
				for (int a = 0; a < 6; ++a) { const double v = hb[6 * b + a]; B[a][b] = (diag && a == b) ? v + lambda : v; }
#pragma unroll
			for (int b = 0; b < 6; ++b) { z[b] = diag ? g[6 * J + b] : 0.0; iv[b] = 1.0; acc[b] = 0.0; }
		}
		bool bad = false;
		for (int k = 0; k < E; ++k) {
			if (diag && I == k) {
#pragma unroll
				for (int j = 0; j < 6; ++j) {
					const double d = B[j][j];
					bad |= !(d > 0.0);
					iv[j] = rcp_f64(d);
#pragma unroll
					for (int i = j + 1; i < 6; ++i) {
						const double f = B[i][j] * iv[j];
#pragma unroll
						for (int c = j + 1; c <= i; ++c) B[i][c] -= f * B[c][j];
					}
				}
#pragma unroll
				for (int j = 0; j < 5; ++j) {
					const double f = z[j] * iv[j];
#pragma unroll
					for (int c = j + 1; c < 6; ++c) z[c] -= f * B[c][j];
				}
#pragma unroll
				for (int i = 0; i < 6; ++i) {
#pragma unroll
					for (int j = 0; j < i; ++j) Ud[6 * i + j] = B[i][j];
					Ud[36 + i] = iv[i];
					Ud[42 + i] = z[i];
				}
			}
			WAVE_SYNC();
			if (__builtin_amdgcn_ballot_w64(bad) != 0) { bad = true; break; }      // (uniform)
			if (active && J == k && I > k) {
				double ivk[6];
#pragma unroll
				for (int j = 0; j < 6; ++j) ivk[j] = Ud[36 + j];
#pragma unroll
				for (int j = 0; j < 5; ++j) {
					double ud[6];
#pragma unroll
					for (int c = j + 1; c < 6; ++c) ud[c] = Ud[6 * c + j];
#pragma unroll
					for (int a = 0; a < 6; ++a) {
						const double f = B[a][j] * ivk[j];
#pragma unroll
						for (int c = j + 1; c < 6; ++c) B[a][c] -= f * ud[c];
					}
				}
				double* o = Pn + LDLB6_BP * I;
#pragma unroll
				for (int a = 0; a < 6; ++a)
#pragma unroll
					for (int c = 0; c < 6; ++c) o[6 * a + c] = B[a][c];
			}
			WAVE_SYNC();
			if (active && J > k) {
				const double* pi = Pn + LDLB6_BP * I;
				const double* pj = Pn + LDLB6_BP * J;
#pragma unroll
				for (int c = 0; c < 6; ++c) {
					const double ivc = Ud[36 + c];
					double ui[6], uj[6];
#pragma unroll
					for (int a = 0; a < 6; ++a) { ui[a] = pi[6 * a + c] * ivc; uj[a] = pj[6 * a + c]; }
#pragma unroll
					for (int a = 0; a < 6; ++a)
#pragma unroll
						for (int b = 0; b < 6; ++b) B[a][b] -= ui[a] * uj[b];
					// (the right-hand side's entries in every lane, with a zero factor off the diagonal: a branch per column makes the compiler gather the
					// six branches in front of ONE block of multiply-adds that then holds all 72 operands of the block pair in registers)
					const double zc = Ud[42 + c];                   // (read by every lane: a guarded read is a branch again)
					const double uz = diag ? zc * ivc : 0.0;
#pragma unroll
					for (int b = 0; b < 6; ++b) z[b] -= uz * uj[b];
					__builtin_amdgcn_sched_barrier(0);             // (a column's twelve operands at a time)
				}
			}
			WAVE_SYNC();
		}
		if (!bad) {
			// ---- back-substitution x = D^-1 z - L'^T x, a block row per step from the bottom ----
			for (int k = E - 1; k >= 0; --k) {
				if (diag && I == k) {
					double x[6];
#pragma unroll
					for (int b = 5; b >= 0; --b) {
						double sacc = z[b] - acc[b];
#pragma unroll
						for (int a = b + 1; a < 6; ++a) sacc -= B[a][b] * x[a];
						x[b] = sacc * iv[b];
						xs[6 * k + b] = x[b];
					}
				}
				WAVE_SYNC();
				if (active && I == k && J < k) {
					double xa[6];
#pragma unroll
					for (int a = 0; a < 6; ++a) xa[a] = xs[6 * k + a];
#pragma unroll
					for (int b = 0; b < 6; ++b) {
						double p = 0.0;
#pragma unroll
						for (int a = 0; a < 6; ++a) p += B[a][b] * xa[a];
						part[6 * J + b] = p;
					}
				}
				WAVE_SYNC();
				if (diag && J < k) {
#pragma unroll
					for (int b = 0; b < 6; ++b) acc[b] += part[6 * J + b];
				}
				WAVE_SYNC();
			}
		}
		if (tid == 0) s_ldl_bad_b = bad ? 1 : 0;
	}
	__syncthreads();
	if (s_ldl_bad_b) return false;                     // uniform
	for (int i = tid; i < n; i += NT) x_out[i] = xs[i];
	__syncthreads();
	return true;
}
#undef SSK_PI
#ifndef SSK_LDL_BLOCKED
#define SSK_LDL_BLOCKED 2               // the solver of systems in L2 scratch: 2 ldl_solve_regtile (default), 1 ldl_solve_blocked, 0 ldl_solve_hybrid (same-box A/B: tools/lm_variant.sh)
#endif
#if SSK_LDL_BLOCKED == 2
#define SSK_LDL_BIG ldl_solve_regtile
#elif SSK_LDL_BLOCKED == 1
#define SSK_LDL_BIG ldl_solve_blocked
#else
#define SSK_LDL_BIG ldl_solve_hybrid
#endif

// G for a system in L2 scratch on the FP64 MATRIX CORES: the same right-looking LDL^T (U = L' D unscaled, border row = right-hand side)
// blocked by 16 columns.  Per panel: the panel's columns of all rows below come into LDS once (pitch 17); one wave factorises the
// 16 x 16 diagonal block in registers (a lane per row, columns exchanged by shuffles: no workgroup barrier inside the block); a lane
// per row eliminates the panel below it; then the trailing matrix takes its rank-16 update as 16 x 16 tiles, a wave per tile:
// four v_mfma_f64_16x16x4_f64 with A = -(U inv) rows, B = U rows from the LDS panel, the accumulator loaded from and stored to the
// matrix.  15 panels for a 240-unknown local area instead of 40 six-column steps, and the update -- 4.6 Mflop of the 4.7 -- runs on
// the matrix pipe.  Back-substitution by 16-row blocks from the bottom: inside a block by one wave (shuffles), then every earlier
// unknown takes the block's 16 terms at once (coalesced rows).  Same operations per element as the other solvers up to the order of
// the 16 terms of an update, which the matrix core fixes (k ascending).
// lds: the workgroup's free tile region; falls back to ldl_solve_hybrid when a panel (n + 1 rows x 17) does not fit.
typedef double ldl_v4d_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bool ldl_solve_hybrid(double* __restrict__ A, int np, int n, double* x_out, double* inv, double* lds, int lds_doubles);
__device__ __forceinline__ bool ldl_solve_mfma16(double* __restrict__ A, int np, int n, double* x_out, double* inv, double* lds, int lds_doubles)
{
	if ((long long)(n + 1) * 17 > lds_doubles) return ldl_solve_hybrid(A, np, n, x_out, inv, lds, lds_doubles);
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	__shared__ int s_ldl_bad_m;
	double* Pn = lds;                                  // panel: rows c .. n, 16 columns, pitch 17
	if (tid == 0) s_ldl_bad_m = 0;
	for (int c = 0; c < n; c += 16) {
		const int pw = n - c < 16 ? n - c : 16;
		__syncthreads();                                   // the previous panel's update is in the matrix
		if (s_ldl_bad_m) return false;
		// ---- the panel's columns of rows c .. n into LDS (unused columns of a narrow last panel: zero) ----
		for (int i = c + tid; i <= n; i += NT) {
			double* pr = Pn + (i - c) * 17;
#pragma unroll
			for (int j = 0; j < 16; ++j) pr[j] = j < pw ? A[(size_t)i * np + c + (j < pw ? j : 0)] : 0.0;
		}
		__syncthreads();
		// ---- diagonal block: lane r of wave 0 = row c + r; column j of the rows below it by shuffle ----
		if (wave == 0) {
			double x[16];
#pragma unroll
			for (int k = 0; k < 16; ++k) x[k] = Pn[(lane < pw ? lane : 0) * 17 + k];
			bool bad = false;
#pragma unroll
			for (int j = 0; j < 16; ++j) {
				const double d = readlane_f64(x[j], j);
				if (j < pw) bad |= !(d > 0.0);
				const double ivj = j < pw ? rcp_f64(d) : 0.0;
				const double t = x[j] * ivj;
#pragma unroll
				for (int k = j + 1; k < 16; ++k) {
					const double akj = readlane_f64(x[j], k);       // U(c + k, c + j): a scalar broadcast
					if (lane > j && k <= lane) x[k] -= t * akj;
				}
				if (lane == j && j < pw) inv[c + j] = ivj;
			}
			if (lane < pw) {
				double* pr = Pn + lane * 17;
#pragma unroll
				for (int k = 0; k < 16; ++k) { pr[k] = x[k]; if (k <= lane && k < pw) A[(size_t)(c + lane) * np + c + k] = x[k]; }
			}
			if (lane == 0 && bad) s_ldl_bad_m = 1;
		}
		__syncthreads();
		if (s_ldl_bad_m) return false;
		double ivr[16];
#pragma unroll
		for (int j = 0; j < 16; ++j) ivr[j] = j < pw ? inv[c + j] : 0.0;
		// ---- rows below the block: eliminate inside the panel (block entries by broadcast reads) ----
		for (int i = c + pw + tid; i <= n; i += NT) {
			double* pr = Pn + (i - c) * 17;
			double x[16];
#pragma unroll
			for (int k = 0; k < 16; ++k) x[k] = pr[k];
#pragma unroll
			for (int j = 0; j < 15; ++j) {
				const double t = x[j] * ivr[j];
#pragma unroll
				for (int k = j + 1; k < 16; ++k) x[k] -= t * Pn[k * 17 + j];
			}
#pragma unroll
			for (int k = 0; k < 16; ++k) { pr[k] = x[k]; if (k < pw) A[(size_t)i * np + c + k] = x[k]; }
		}
		__syncthreads();
		// ---- trailing update: tile (a, b), b <= a, rows r0 + 16 a .., columns r0 + 16 b .., a wave per tile ----
		const int r0 = c + pw;
		if (r0 < n) {
			const int nr = (n + 1 - r0 + 15) >> 4, nc = (n - r0 + 15) >> 4;
			const int m = lane & 15, kk = lane >> 4;
			int idx = 0;
			for (int a = 0; a < nr; ++a)
				for (int b = 0; b <= a && b < nc; ++b, ++idx) {
					if (idx % NW != wave) continue;
					const int i0 = r0 + 16 * a, k0 = r0 + 16 * b;
					ldl_v4d_t acc;
#pragma unroll
					for (int r = 0; r < 4; ++r) {
						const int row = i0 + kk + 4 * r, col = k0 + m;
						const double v = A[(size_t)(row <= n ? row : n) * np + (col < n ? col : n - 1)];
						acc[r] = (row <= n && col < n) ? v : 0.0;
					}
					const double* pa = Pn + ((i0 + m <= n ? i0 + m : n) - c) * 17;
					const double* pb = Pn + ((k0 + m <= n ? k0 + m : n) - c) * 17;
#pragma unroll
					for (int q = 0; q < 4; ++q) {
						// (the pivot reciprocal of column 4 q + kk straight from LDS: a lane-dependent index into a register array would go through scratch)
						const int jc = c + 4 * q + kk;
						const double av = -(pa[4 * q + kk] * inv[jc < n ? jc : n - 1]);
						acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, pb[4 * q + kk], acc, 0, 0, 0);
					}
#pragma unroll
					for (int r = 0; r < 4; ++r) {
						const int row = i0 + kk + 4 * r, col = k0 + m;
						if (row <= n && col < n && col <= row) A[(size_t)row * np + col] = acc[r];
					}
				}
		}
	}
	__syncthreads();
	if (s_ldl_bad_m) return false;
	// ---- back-substitution: x = D^-1 z, then L'^T x: 16-row blocks from the bottom ----
	for (int k = tid; k < n; k += NT) x_out[k] = A[(size_t)n * np + k] * inv[k];
	__syncthreads();
	for (int c = ((n - 1) >> 4) << 4; c >= 0; c -= 16) {
		const int pw = n - c < 16 ? n - c : 16;
		if (wave == 0) {
			// lane k (< pw): column c + k of the block's rows, scaled; x of the block from the last row up
			const int k = lane < pw ? lane : 0;
			double u[16];
			const double ivk = inv[c + k];
#pragma unroll
			for (int i = 0; i < 16; ++i) u[i] = (i < pw && i > k) ? A[(size_t)(c + i) * np + c + k] * ivk : 0.0;
			double xv = x_out[c + k];
#pragma unroll
			for (int i = 15; i >= 1; --i) {
				const double xi = readlane_f64(xv, i);
				if (i < pw) xv -= u[i] * xi;                    // (u[i] = 0 for k >= i)
			}
			if (lane < pw) x_out[c + lane] = xv;
		}
		__syncthreads();
		for (int k = tid; k < c; k += NT) {
			const double ivk = inv[k];
			double sacc = 0.0;
#pragma unroll
			for (int i = 0; i < 16; ++i) sacc += i < pw ? A[(size_t)(c + (i < pw ? i : 0)) * np + k] * x_out[c + (i < pw ? i : 0)] : 0.0;
			x_out[k] -= sacc * ivk;
		}
		__syncthreads();
	}
	return true;
}

// G (fallback): the same factorisation by the whole workgroup on an L2-resident matrix (6E too large for LDS)
__device__ bool chol_solve_block(double* A, int n, double* b, int* s_flag, double* s_piv)
{
	const int tid = threadIdx.x;
	for (int j = 0; j < n; ++j) {
		if (tid == 0) {
			const double d = A[(size_t)j * n + j];
			*s_flag = d > 0.0;
			*s_piv = sqrt(d);
			if (d > 0.0) A[(size_t)j * n + j] = *s_piv;
		}
		__syncthreads();
		if (!*s_flag) { __syncthreads(); return false; }
		const double inv = 1.0 / *s_piv;
		const int m = n - j - 1;
		for (int i = j + 1 + tid; i < n; i += NT) A[(size_t)i * n + j] = A[(size_t)i * n + j] * inv;
		__syncthreads();
		for (int t = tid; t < m * m; t += NT) {
			const int i = j + 1 + t / m, k = j + 1 + t % m;
			if (k <= i) A[(size_t)i * n + k] -= A[(size_t)i * n + j] * A[(size_t)k * n + j];
		}
		__syncthreads();
	}
	for (int i = 0; i < n; ++i) {
		if (tid == 0) b[i] = b[i] / A[(size_t)i * n + i];
		__syncthreads();
		const double yi = b[i];
		for (int k = i + 1 + tid; k < n; k += NT) b[k] -= A[(size_t)k * n + i] * yi;
		__syncthreads();
	}
	for (int i = n - 1; i >= 0; --i) {
		if (tid == 0) b[i] = b[i] / A[(size_t)i * n + i];
		__syncthreads();
		const double xi = b[i];
		for (int k = tid; k < i; k += NT) b[k] -= A[(size_t)i * n + k] * xi;
		__syncthreads();
	}
	return true;
}
