// ssk_lm_node_dev.h -- device functions of the NODE form shared by the persistent kernel (ssk_lm_node.hip) and the phase-split kernels
// (ssk_lm_split.hip): suffix table, group linearisation (observation / run sums / roots / nodes), pair items, trial-state scoring.
// Included INSIDE the including file's anonymous namespace, after ssk_lm_device.h (NT, NW, WAVE_SYNC, rcp_f64 ...) and the layout constants
// REC, STG, LREC, SCR, R_Q, R_P, R_V.  Data model and algebra: ssk_lm_node.h.

typedef double v2d_t __attribute__((ext_vector_type(2)));

#ifdef SSK_LM_STAMPS
__device__ unsigned long long g_nd_fine[16];
#define FINE0() unsigned long long f_t_ = __builtin_amdgcn_s_memtime()
#define FINE(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); g_nd_fine[i] += t_ - f_t_; f_t_ = t_; } } while (0)
#else
#define FINE0() do {} while (0)
#define FINE(i) do {} while (0)
#endif

__device__ __forceinline__ double nd_rsq(double x)       // 1/sqrt(x), x > 0: hardware estimate + two Newton steps
{
	double y = __builtin_amdgcn_rsq(x);
	const double hx = 0.5 * x;
	y = y * (1.5 - hx * y * y);
	y = y * (1.5 - hx * y * y);
	return y;
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
	const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
	const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
	return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void nd_cross(const double* a, const double* b, double* c)
{
	c[0] = a[1] * b[2] - a[2] * b[1];
	c[1] = a[2] * b[0] - a[0] * b[2];
	c[2] = a[0] * b[1] - a[1] * b[0];
}

// 12 doubles of a suffix transform / scratch record from global memory as six 16-byte loads
__device__ __forceinline__ void nd_load12(const double* __restrict__ p, double (&o)[12])
{
	const v2d_t* v = reinterpret_cast<const v2d_t*>(p);
#pragma unroll
	for (int i = 0; i < 6; ++i) { const v2d_t x = v[i]; o[2 * i] = x.x; o[2 * i + 1] = x.y; }
}
__device__ __forceinline__ void nd_store12(double* __restrict__ p, const double (&o)[12])
{
	v2d_t* v = reinterpret_cast<v2d_t*>(p);
#pragma unroll
	for (int i = 0; i < 6; ++i) { v2d_t x; x.x = o[2 * i]; x.y = o[2 * i + 1]; v[i] = x; }
}
// ND_NT = 1 (same-box A/B only): the STREAMING data of a pass -- the back-substitution scratch (96 B per lane out in the linearisation, back in the scoring),
// the static lane / observation records and item codes -- with the non-temporal cache policy, meant to keep the suffix tables, landmarks and poses in the
// XCD's 4 MB of L2 (64 resident windows stream ~37 MB per pass through it).  MEASURED WORSE, round 5: 3.80 -> 4.46 ms per 1 024 bench windows (the phase-split
// path: 37.08 ms either way): the scratch that comes back in the same pass and the records the scoring re-reads DO profit from the L2 / the memory-side cache
// -- the kernel is latency bound and its latency is largely that of these very streams.  Default 0.
#ifndef ND_NT
#define ND_NT 0
#endif
#if ND_NT
#define ND_LDNT(p) __builtin_nontemporal_load(p)
#define ND_STNT(v, p) __builtin_nontemporal_store(v, p)
#else
#define ND_LDNT(p) (*(p))
#define ND_STNT(v, p) (*(p) = (v))
#endif
__device__ __forceinline__ void nd_load12_nt(const double* __restrict__ p, double (&o)[12])
{
	const v2d_t* v = reinterpret_cast<const v2d_t*>(p);
#pragma unroll
	for (int i = 0; i < 6; ++i) { const v2d_t x = ND_LDNT(v + i); o[2 * i] = x.x; o[2 * i + 1] = x.y; }
}
__device__ __forceinline__ void nd_store12_nt(double* __restrict__ p, const double (&o)[12])
{
	v2d_t* v = reinterpret_cast<v2d_t*>(p);
#pragma unroll
	for (int i = 0; i < 6; ++i) { v2d_t x; x.x = o[2 * i]; x.y = o[2 * i + 1]; ND_STNT(x, v + i); }
}

// Suffix table of the window at the poses `pose` (LDS) -> tab (HBM, 12 doubles per suffix, id 0 = identity): a lane per suffix walks its
// own chain -- the codes from the suffix down to the empty one, then the transforms from the innermost step outwards.
__device__ void nd_build_table(const nd_win& W, const nd_i2* suf /* LDS copy */, const double* pose, double* __restrict__ tab)
{
	for (int id = threadIdx.x; id < W.n_suf; id += NT) {
		int code[SSK_LM_MAX_PATH], n = 0, cur = id;
#pragma unroll
		for (int k = 0; k < SSK_LM_MAX_PATH; ++k) {
			code[k] = 0;
			if (cur != 0) { const nd_i2 r = suf[cur]; code[k] = r.x; cur = r.y; n = k + 1; }
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
		nd_store12(tab + 12 * (size_t)id, T);
	}
}

// camera-frame point and stereo prediction of a landmark f under the path transform T; returns the residual r = z - h
__device__ __forceinline__ void nd_residual(const lm_cam& P, const double (&T)[12], const double* f, const nd_f4 px, double (&pc)[3], double& iz, double (&r)[4])
{
	double q[3];
#pragma unroll
	for (int i = 0; i < 3; ++i) q[i] = T[4 * i] * f[0] + T[4 * i + 1] * f[1] + T[4 * i + 2] * f[2] + T[4 * i + 3];
	const double d0 = q[0] - P.S[3], d1 = q[1] - P.S[7], d2 = q[2] - P.S[11];
#pragma unroll
	for (int i = 0; i < 3; ++i) pc[i] = P.S[i] * d0 + P.S[4 + i] * d1 + P.S[8 + i] * d2;
	iz = rcp_f64(pc[2]);
	r[0] = (double)px.x - (P.cx_l + P.fx_l * pc[0] * iz);
	r[1] = (double)px.y - (P.cy_l + P.fy_l * pc[1] * iz);
	r[2] = (double)px.z - (P.cx_r + P.fx_r * (pc[0] - P.baseline) * iz);
	r[3] = (double)px.w - (P.cy_r + P.fy_r * pc[1] * iz);
}

// ---- group phase: one wavefront linearises one landmark group; leaves the node records in `stage` (the wavefront's part of the record
// region: lane i's record at stage + REC * i), the back-substitution records in HBM ----
// the static records of a lane for one group (observation record + pixels, root / node record): fetched a tile (a group) ahead
struct nd_recs { nd_i2 orec; nd_f4 px; nd_i4 lrec; };
__device__ __forceinline__ nd_recs nd_fetch_recs(const nd_dev& N, const nd_win& W, const nd_group G, int lane)
{
	const int oi = W.obs0 + G.obs0 + min(lane, max(G.n_obs - 1, 0));      // loads are unconditional on clamped indices
	const int li = W.lane0 + G.lane0 + min(lane, max(G.n_lanes - 1, 0));
	nd_recs r;
#if ND_NT
	{
		typedef int nd_v2i __attribute__((ext_vector_type(2)));
		typedef int nd_v4i __attribute__((ext_vector_type(4)));
		typedef float nd_v4f __attribute__((ext_vector_type(4)));
		const nd_v2i a = __builtin_nontemporal_load(reinterpret_cast<const nd_v2i*>(N.obs + oi));
		const nd_v4f b = __builtin_nontemporal_load(reinterpret_cast<const nd_v4f*>(N.px + oi));
		const nd_v4i c = __builtin_nontemporal_load(reinterpret_cast<const nd_v4i*>(N.lane + li));
		r.orec = nd_i2{a.x, a.y}; r.px = nd_f4{b.x, b.y, b.z, b.w}; r.lrec = nd_i4{c.x, c.y, c.z, c.w};
	}
#else
	r.orec = N.obs[oi]; r.px = N.px[oi]; r.lrec = N.lane[li];
#endif
	return r;
}

// Round 5 (persistent node kernel): the lane's operands of a group phase -- the observation's path transform and landmark, the node's transform and
// landmark: gathers from the suffix table and the landmark array, L2 or beyond -- fetched a TILE ahead, at the end of the previous tile's pair phase (its
// 36 accumulators are dead there): they fly during the tile's closing barrier instead of in front of the first flop of the next group phase.
struct nd_pre { double T[12], f[3], Tn[12], fn[3]; };
__device__ __forceinline__ void nd_prefetch_group(const lm_dev& D, const nd_win& W, const nd_group G, const nd_recs& rc, const double* __restrict__ tab, int lane, nd_pre& o)
{
	nd_i4 lrec = rc.lrec;
	if (lane >= G.n_lanes) lrec.w = 0;
	const bool is_node = (lrec.w & 0xFF) == 2;
	nd_load12(tab + 12 * (size_t)(rc.orec.y & 0xFFFF), o.T);
	const double* fp = D.lm + 3 * (size_t)(W.lm0 + rc.orec.x);
	o.f[0] = fp[0]; o.f[1] = fp[1]; o.f[2] = fp[2];
	nd_load12(tab + 12 * (size_t)(is_node ? (lrec.z & 0xFFFF) : 0), o.Tn);
	const double* fq = D.lm + 3 * (size_t)(W.lm0 + lrec.x);
	o.fn[0] = fq[0]; o.fn[1] = fq[1]; o.fn[2] = fq[2];
}

// LATE_TN: the node lanes' transform and landmark are loaded behind the observation phase instead of in front of it (30 VGPRs fewer at the
// function's pressure peak: the phase-split kernel runs three wavefronts per SIMD on 168 registers; the load then flies behind the run sums)
template <bool LATE_TN = false, bool PRE = false>
__device__ __forceinline__ void nd_linearise_group(const lm_dev& D, const nd_dev& N, const nd_win& W, const lm_cam& P, const nd_group G, const nd_recs rc,
                                                   const double* __restrict__ tab, double* __restrict__ stage, double lambda, bool pre, int lane, double& vmax,
                                                   const nd_pre* pf = nullptr)
{
	const int li = W.lane0 + G.lane0 + min(lane, G.n_lanes - 1);
	const nd_i2 orec = rc.orec;
	const nd_f4 px = rc.px;
	nd_i4 lrec = rc.lrec;
	if (lane >= G.n_lanes) lrec.w = 0;                             // idle lane
	const int kind = lrec.w & 0xFF, root_lane = (lrec.w >> 8) & 0xFF;
	const bool is_node = kind == 2, is_root = kind == 1;
	// the node's transform and landmark (needed last: in flight behind everything else)
	double Tn[12], fn[3];
	if (PRE) {
#pragma unroll
		for (int i = 0; i < 12; ++i) Tn[i] = pf->Tn[i];
		fn[0] = pf->fn[0]; fn[1] = pf->fn[1]; fn[2] = pf->fn[2];
	} else if (!LATE_TN) {
		nd_load12(tab + 12 * (size_t)(is_node ? (lrec.z & 0xFFFF) : 0), Tn);
		const double* fp = D.lm + 3 * (size_t)(W.lm0 + lrec.x);
		fn[0] = fp[0]; fn[1] = fp[1]; fn[2] = fp[2];
	}
	FINE0();
	// ---- lane = observation: V_o = Jf^T Jf (6), g_o = Jf^T w r (3) -> staging ----
	{
		double T[12], f[3], pc[3], iz, r[4];
		if (PRE) {
#pragma unroll
			for (int i = 0; i < 12; ++i) T[i] = pf->T[i];
			f[0] = pf->f[0]; f[1] = pf->f[1]; f[2] = pf->f[2];
		} else {
			nd_load12(tab + 12 * (size_t)(orec.y & 0xFFFF), T);
			const double* fp = D.lm + 3 * (size_t)(W.lm0 + orec.x);
			f[0] = fp[0]; f[1] = fp[1]; f[2] = fp[2];
		}
		nd_residual(P, T, f, px, pc, iz, r);
		const double iz2 = iz * iz;
		const double s2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3];
		double w = 1.0;                                            // IRLS weight sqrt(rho'(s)/2s) = (1 + s2/k^2)^(-1/4)
		if (P.robust) { const double rs = nd_rsq(1.0 + s2 * rcp_f64(P.kernel * P.kernel)); w = rs * nd_rsq(rs); }
		// M = D_proj R_S^T with the structural zeros of the projection Jacobian
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
		double jf[12];                                             // Jf = w M R_path
#pragma unroll
		for (int i = 0; i < 4; ++i)
#pragma unroll
			for (int j = 0; j < 3; ++j) jf[3 * i + j] = w * (M[3 * i] * T[j] + M[3 * i + 1] * T[4 + j] + M[3 * i + 2] * T[8 + j]);
		const double wr[4] = {w * r[0], w * r[1], w * r[2], w * r[3]};
		double* o = stage + STG * lane;
		double vo[9];
		vo[0] = jf[0] * jf[0] + jf[3] * jf[3] + jf[6] * jf[6] + jf[9] * jf[9];
		vo[1] = jf[0] * jf[1] + jf[3] * jf[4] + jf[6] * jf[7] + jf[9] * jf[10];
		vo[2] = jf[0] * jf[2] + jf[3] * jf[5] + jf[6] * jf[8] + jf[9] * jf[11];
		vo[3] = jf[1] * jf[1] + jf[4] * jf[4] + jf[7] * jf[7] + jf[10] * jf[10];
		vo[4] = jf[1] * jf[2] + jf[4] * jf[5] + jf[7] * jf[8] + jf[10] * jf[11];
		vo[5] = jf[2] * jf[2] + jf[5] * jf[5] + jf[8] * jf[8] + jf[11] * jf[11];
		vo[6] = jf[0] * wr[0] + jf[3] * wr[1] + jf[6] * wr[2] + jf[9] * wr[3];
		vo[7] = jf[1] * wr[0] + jf[4] * wr[1] + jf[7] * wr[2] + jf[10] * wr[3];
		vo[8] = jf[2] * wr[0] + jf[5] * wr[1] + jf[8] * wr[2] + jf[11] * wr[3];
#if ND_SCAN
		// Inclusive prefix sums INSIDE each landmark's run of observation lanes (a landmark's observations are consecutive lanes; orec.y bits 24..30:
		// the lane's distance h from its landmark's first observation): six DPP steps for the nine sums -- shifts by 1, 2, 4, 8 inside a row of
		// 16 lanes, then the totals of the rows before (row_bcast15 / row_bcast31).  A root / node lane then needs TWO staging rows whatever the
		// length of its run (P[last] - P[first - 1]); walking the runs row by row made every wavefront wait for its longest landmark (3.2 k cycles of
		// a group's 11 k).  Fixed order: deterministic; the difference of two prefix sums of at most 64 terms of one sign pattern costs ~1e-14
		// relative, far inside the LM contract's 1e-6.
		{
			const int h = (orec.y >> 24) & 0x7F, lr = lane & 15;
			const double m1 = h >= 1 ? 1.0 : 0.0, m2 = h >= 2 ? 1.0 : 0.0, m4 = h >= 4 ? 1.0 : 0.0, m8 = h >= 8 ? 1.0 : 0.0;
			// (row_bcast15 for rows 1 and 3 only: row 2 takes the lanes 0 .. 31 in one piece from lane 31, which by then holds rows 0 + 1 of its landmark)
			const double m16 = ((lane & 16) && h >= lr + 1) ? 1.0 : 0.0, m32 = (lane >= 32 && h >= lane - 31) ? 1.0 : 0.0;
#pragma unroll
			for (int c = 0; c < 9; ++c) vo[c] = fma(m1, dpp_f64<0x111>(vo[c]), vo[c]);       // row_shr:1 (a source outside the row: 0)
#pragma unroll
			for (int c = 0; c < 9; ++c) vo[c] = fma(m2, dpp_f64<0x112>(vo[c]), vo[c]);
#pragma unroll
			for (int c = 0; c < 9; ++c) vo[c] = fma(m4, dpp_f64<0x114>(vo[c]), vo[c]);
#pragma unroll
			for (int c = 0; c < 9; ++c) vo[c] = fma(m8, dpp_f64<0x118>(vo[c]), vo[c]);
#pragma unroll
			for (int c = 0; c < 9; ++c) vo[c] = fma(m16, dpp_f64<0x142>(vo[c]), vo[c]);      // row_bcast15: lane 15 of the row before
#pragma unroll
			for (int c = 0; c < 9; ++c) vo[c] = fma(m32, dpp_f64<0x143>(vo[c]), vo[c]);      // row_bcast31: lane 31 (rows 2 and 3)
		}
#endif
#pragma unroll
		for (int c = 0; c < 9; ++c) o[c] = vo[c];
	}
	WAVE_SYNC();
	FINE(0);
	if (LATE_TN && !PRE) {
		nd_load12(tab + 12 * (size_t)(is_node ? (lrec.z & 0xFFFF) : 0), Tn);
		const double* fp = D.lm + 3 * (size_t)(W.lm0 + lrec.x);
		fn[0] = fp[0]; fn[1] = fp[1]; fn[2] = fp[2];
	}
	// ---- lane = root / node: sums over its run of the staging (a landmark's observations are sorted so that a trie subtree is a run) ----
	double vs[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
	{
		const int a = lrec.y & 0xFF, n = kind ? ((lrec.y >> 8) & 0xFF) : 0;
#if ND_SCAN
		// the run's sums from the landmark's prefix sums: P[a + n - 1] - P[a - 1] (bit 16 of y: the run starts at the landmark's first observation)
		if (n > 0) {
			const double* pe = stage + STG * (a + n - 1);
			const double* pb = stage + STG * max(a - 1, 0);
			const double mb = ((lrec.y >> 16) & 1) ? 0.0 : 1.0;
			double e9[9], b9[9];
#pragma unroll
			for (int c = 0; c < 9; ++c) { e9[c] = pe[c]; b9[c] = pb[c]; }
#pragma unroll
			for (int c = 0; c < 9; ++c) vs[c] = fma(-mb, b9[c], e9[c]);
		}
	}
#elif ND_SUM4
		// four rows per turn, loads first (rows past the run: the run's last row, weight 0); the order of the adds is the rows' order
		for (int i = 0; i < n; i += ND_SUMROWS) {
			double v[ND_SUMROWS][9];
#pragma unroll
			for (int u = 0; u < ND_SUMROWS; ++u) {
				const double* o = stage + STG * (a + min(i + u, n - 1));
#pragma unroll
				for (int c = 0; c < 9; ++c) v[u][c] = o[c];
			}
#pragma unroll
			for (int u = 0; u < ND_SUMROWS; ++u) {
				const double wgt = i + u < n ? 1.0 : 0.0;
#pragma unroll
				for (int c = 0; c < 9; ++c) vs[c] = fma(wgt, v[u][c], vs[c]);
			}
		}
	}
#else
		for (int i = 0; i < n; ++i) {
			const double* o = stage + STG * (a + i);
#pragma unroll
			for (int c = 0; c < 9; ++c) vs[c] += o[c];
		}
	}
#endif
	WAVE_SYNC();
	FINE(1);
	// ---- roots: V + lambda I = L L^T -> (L, L^-1, c = L^-1 g, fixed flag) for the landmark's nodes; L^-1, c to HBM for the landmark step ----
	if (is_root) {
		const bool opt = lrec.z >= 0;
		double rec[16];
		if (opt && !pre) {
			const double a00 = vs[0] + lambda, i00 = nd_rsq(a00);
			const double l10 = vs[1] * i00, l20 = vs[2] * i00;
			const double a11 = vs[3] + lambda - l10 * l10, i11 = nd_rsq(a11);
			const double l21 = (vs[4] - l20 * l10) * i11;
			const double a22 = vs[5] + lambda - l20 * l20 - l21 * l21, i22 = nd_rsq(a22);
			const double i10 = -i11 * l10 * i00, i21 = -i22 * l21 * i11, i20 = -(i21 * l10 + i22 * l20) * i00;
			rec[0] = a00 * i00; rec[1] = l10; rec[2] = a11 * i11; rec[3] = l20; rec[4] = l21; rec[5] = a22 * i22;
			rec[6] = i00; rec[7] = i10; rec[8] = i11; rec[9] = i20; rec[10] = i21; rec[11] = i22;
			rec[12] = i00 * vs[6]; rec[13] = i10 * vs[6] + i11 * vs[7]; rec[14] = i20 * vs[6] + i21 * vs[7] + i22 * vs[8];
			rec[15] = 0.0;
		} else {
			rec[0] = 1; rec[1] = 0; rec[2] = 1; rec[3] = 0; rec[4] = 0; rec[5] = 1;
			rec[6] = 1; rec[7] = 0; rec[8] = 1; rec[9] = 0; rec[10] = 0; rec[11] = 1;
			rec[12] = 0; rec[13] = 0; rec[14] = 0; rec[15] = 1.0;
		}
		if (pre && opt) vmax = fmax(vmax, fmax(vs[0], fmax(vs[3], vs[5])));
		double* o = stage + LREC * lane;
#pragma unroll
		for (int c = 0; c < 16; ++c) o[c] = rec[c];
		if (!pre) {
			double sc[12];
#pragma unroll
			for (int c = 0; c < 9; ++c) sc[c] = opt ? rec[6 + c] : 0.0;
			sc[9] = sc[10] = sc[11] = 0.0;
			nd_store12_nt(N.scr + SCR * (size_t)li, sc);
			if (D.store_hff && opt) {                              // diagnostics: the landmark blocks of the last linearisation
				double* Hm = D.Hff + 9 * (size_t)(W.ls0 + lrec.z);
				Hm[0] = vs[0]; Hm[1] = vs[1]; Hm[2] = vs[2]; Hm[3] = vs[1]; Hm[4] = vs[3]; Hm[5] = vs[4]; Hm[6] = vs[2]; Hm[7] = vs[4]; Hm[8] = vs[5];
				double* gl = D.gf + 3 * (size_t)(W.ls0 + lrec.z);
				gl[0] = vs[6]; gl[1] = vs[7]; gl[2] = vs[8];
			}
		}
	}
	WAVE_SYNC();
	FINE(2);
	// ---- nodes: Z = Vs L^-T, Pm = L - Z, v = gs - Z c, rotated into the edge's frame ----
	double out[24];
	if (is_node) {
		const double* lr = stage + LREC * root_lane;
		double Lm[6], Li[6], cc[3];
#pragma unroll
		for (int c = 0; c < 6; ++c) { Lm[c] = lr[c]; Li[c] = lr[6 + c]; }
		cc[0] = lr[12]; cc[1] = lr[13]; cc[2] = lr[14];
		const bool fixed = lr[15] != 0.0;
		const double V[9] = {vs[0], vs[1], vs[2], vs[1], vs[3], vs[4], vs[2], vs[4], vs[5]};
		double Z[9], Pm[9], v[3];
#pragma unroll
		for (int i = 0; i < 3; ++i) {
			// Z(i,j) = sum_{k <= j} V(i,k) Li(j,k)      (Li lower triangular, packed 00 10 11 20 21 22)
			Z[3 * i] = V[3 * i] * Li[0];
			Z[3 * i + 1] = V[3 * i] * Li[1] + V[3 * i + 1] * Li[2];
			Z[3 * i + 2] = V[3 * i] * Li[3] + V[3 * i + 1] * Li[4] + V[3 * i + 2] * Li[5];
			v[i] = vs[6 + i] - (Z[3 * i] * cc[0] + Z[3 * i + 1] * cc[1] + Z[3 * i + 2] * cc[2]);
		}
		const double Lf[9] = {Lm[0], 0, 0, Lm[1], Lm[2], 0, Lm[3], Lm[4], Lm[5]};
#pragma unroll
		for (int c = 0; c < 9; ++c) Pm[c] = fixed ? ((c == 0 || c == 4 || c == 8) ? 1.0 : 0.0) : Lf[c] - Z[c];
		const double sg = ((lrec.z >> 24) & 1) ? -1.0 : 1.0;
#pragma unroll
		for (int i = 0; i < 3; ++i) {
			const double r0 = sg * Tn[4 * i], r1 = sg * Tn[4 * i + 1], r2 = sg * Tn[4 * i + 2];
#pragma unroll
			for (int j = 0; j < 3; ++j) {
				out[3 * i + j] = r0 * Z[j] + r1 * Z[3 + j] + r2 * Z[6 + j];
				out[R_P + 3 * i + j] = r0 * Pm[j] + r1 * Pm[3 + j] + r2 * Pm[6 + j];
			}
			out[R_V + i] = r0 * v[0] + r1 * v[1] + r2 * v[2];
			out[R_Q + i] = Tn[4 * i] * fn[0] + Tn[4 * i + 1] * fn[1] + Tn[4 * i + 2] * fn[2] + Tn[4 * i + 3];
		}
	}
	WAVE_SYNC();
	FINE(3);
	if (is_node) {
		v2d_t* o = reinterpret_cast<v2d_t*>(stage + REC * lane);
#pragma unroll
		for (int c = 0; c < 12; ++c) { v2d_t x; x.x = out[2 * c]; x.y = out[2 * c + 1]; o[c] = x; }
		if (!pre) {
			const double sc[12] = {out[0], out[1], out[2], out[3], out[4], out[5], out[6], out[7], out[8], out[R_Q], out[R_Q + 1], out[R_Q + 2]};
			nd_store12_nt(N.scr + SCR * (size_t)li, sc);
		}
	}
	FINE(4);
}

struct nd_item_ops { double X[10], Y[10], qr[3], qc[3]; };
__device__ __forceinline__ void nd_item_load(const double* __restrict__ rec, int it, nd_item_ops& o)
{
	const double* xr = rec + REC * (it & 511);
	const double* yc = rec + REC * ((it >> 10) & 511);
	const v2d_t* Xv = reinterpret_cast<const v2d_t*>(xr + R_P * ((it >> 9) & 1));
	const v2d_t* Yv = reinterpret_cast<const v2d_t*>(yc + R_P * ((it >> 19) & 1));
#pragma unroll
	for (int c = 0; c < 5; ++c) { const v2d_t a = Xv[c], b = Yv[c]; o.X[2 * c] = a.x; o.X[2 * c + 1] = a.y; o.Y[2 * c] = b.x; o.Y[2 * c + 1] = b.y; }
#pragma unroll
	for (int c = 0; c < 3; ++c) { o.qr[c] = xr[R_Q + c]; o.qc[c] = yc[R_Q + c]; }
}
// acc += sg [I; [q_r]x] (X Y^T) [I | -[q_c]x]
__device__ __forceinline__ void nd_item_apply(const nd_item_ops& o, double sg, double (&acc)[36])
{
	const double sqc[3] = {sg * o.qc[0], sg * o.qc[1], sg * o.qc[2]}, sqr[3] = {sg * o.qr[0], sg * o.qr[1], sg * o.qr[2]};
	double Dm[9], U[9];
#pragma unroll
	for (int a = 0; a < 3; ++a) {
#pragma unroll
		for (int b = 0; b < 3; ++b) Dm[3 * a + b] = o.X[3 * a] * o.Y[3 * b] + o.X[3 * a + 1] * o.Y[3 * b + 1] + o.X[3 * a + 2] * o.Y[3 * b + 2];
		nd_cross(sqc, Dm + 3 * a, U + 3 * a);                      // row a of -sg D [q_c]x = (sg q_c) x (row a of D)
#pragma unroll
		for (int b = 0; b < 3; ++b) { acc[6 * a + b] = fma(sg, Dm[3 * a + b], acc[6 * a + b]); acc[6 * a + 3 + b] += U[3 * a + b]; }
	}
#pragma unroll
	for (int b = 0; b < 3; ++b) {                                  // columns of sg [q_r]x D and of [q_r]x U, added component by component
#pragma unroll
		for (int a = 0; a < 3; ++a) {
			const int a1 = (a + 1) % 3, a2 = (a + 2) % 3;
			acc[6 * (3 + a) + b] = fma(-sqr[a2], Dm[3 * a1 + b], fma(sqr[a1], Dm[3 * a2 + b], acc[6 * (3 + a) + b]));
			acc[6 * (3 + a) + 3 + b] = fma(-o.qr[a2], U[3 * a1 + b], fma(o.qr[a1], U[3 * a2 + b], acc[6 * (3 + a) + 3 + b]));
		}
	}
}

// ---- pair phase: the items of one 6x6 block dealt out among the 4 lanes of a quad; 36 sums in registers ----
// item: record X = slot | P-part << 9 (row node), record Y = slot << 10 | P-part << 19 (column node), bit 20: not nested (D = -Zh Zh^T)
// block(r, c) = [I; [q_r]x] D [I | -[q_c]x],  D = +-X Y^T.
// A lane's first ND_ITEMS_REG item codes come from HBM in ONE burst of independent loads (nd_items_fetch, issued in front of the tile's
// barrier) -- a load per loop turn put an HBM round trip into every item.
#ifndef ND_PIPE
#define ND_PIPE 0                       // 1: software-pipelined item turns (loads of item k + 1 in front of the products of item k)
#endif
#ifndef ND_LM_FLIP
#define ND_LM_FLIP 1                    // persistent node kernel: landmarks double-buffered (an accepted step flips the buffers; 0: a copy pass per accepted step)
#endif
#ifndef ND_PREFETCH
#define ND_PREFETCH 1                   // persistent node kernel: a group phase's transforms / landmarks fetched at the end of the previous tile's pair phase (0: at the group's start)
#endif
#ifndef ND_SCAN
#define ND_SCAN 1                       // run sums from per-landmark prefix sums over the observation lanes (DPP scan); 0: every root / node lane walks its run
#endif
#ifndef ND_SUM4
#define ND_SUM4 1                       // 1: ND_SUMROWS staging rows per turn in the run sums
#endif
#ifndef ND_SUMROWS
#define ND_SUMROWS 4
#endif
#ifndef ND_TILES4
#define ND_TILES4 1                     // the reduced system of windows with E <= 14 as 4 x 4 register tiles (0: dense copy + ldl_solve_lds)
#endif
#ifndef ND_BLOCKS6
#define ND_BLOCKS6 2                    // the reduced system of windows with E <= 10 by ONE wavefront: 2 = in matrix-core accumulator tiles (ldl_solve_mfma_wave,
                                        // ssk_lm_ldl_wave.h), 1 = a 6 x 6 block per lane on the vector ALUs (ldl_solve_blocks6_wave), 0 = the 4 x 4 tiles of all threads
#endif
#ifndef ND_HB_ATOMIC
#define ND_HB_ATOMIC 1                  // block accumulators: 1 = ds_add_f64 per component, 0 = 16-byte loads, adds, 16-byte stores
#endif
constexpr int ND_ITEMS_REG = 8;
struct nd_items { int it[ND_ITEMS_REG]; };
// the lane's items are items[first], items[first + step], ...
__device__ __forceinline__ nd_items nd_items_fetch(const int* __restrict__ items, int n, int first, int step)
{
	nd_items r;
#pragma unroll
	for (int k = 0; k < ND_ITEMS_REG; ++k) r.it[k] = ND_LDNT(items + min(first + step * k, max(n - 1, 0)));      // unconditional, clamped
	return r;
}
// (Round 6, measured and removed again -- commit 79535aa holds it, `ND_GRAD_FUSED`: the reduced gradient riding on the DIAGONAL blocks' items.  Block (e, e) holds one item per
// node of edge e with itself, and the edge's gradient is the sum over exactly those nodes of [vh; q x vh]: the quad that owns the block added them up beside
// its 36 sums and the separate gradient pass -- ten quads of wavefront 0 walking an edge's nodes, 5.3 k of a tile's 17 k pair-phase cycles by thread 0's
// stamps -- went.  Same-box A/B, 1 024 bench windows: 3.80 -> 3.77 ms with 2-3 spilled VGPRs in the e10 kernel: the kernel is not bound by wavefront 0's chain.)
__device__ __forceinline__ void nd_pair_items(const double* __restrict__ rec, const int* __restrict__ items, const nd_items& pre_it, int n, int ql, int step, bool pre, double (&acc)[36])
{
	const double neg = pre ? 0.0 : -1.0;
#if ND_PIPE
	// the wavefront's lanes run in lockstep: every lane takes as many turns as the longest run among them (a lane past its run repeats its
	// last item with weight 0), so the turns need no lane branches and the loads of item k + 1 are issued in front of the products of item k
	const int mine = n > ql ? (n - ql + step - 1) / step : 0;
	int turns = mine;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) turns = max(turns, __shfl_xor(turns, o, 64));
	turns = __builtin_amdgcn_readfirstlane(turns);
	nd_item_ops cur, nxt;
	nd_item_load(rec, pre_it.it[0], cur);
#pragma unroll
	for (int k = 0; k < ND_ITEMS_REG; ++k) {
		if (k < turns) {                                           // (uniform)
			const int it = pre_it.it[k];
			if (k + 1 < ND_ITEMS_REG && k + 1 < turns) nd_item_load(rec, pre_it.it[k + 1], nxt);
			nd_item_apply(cur, k < mine ? (((it >> 20) & 1) ? neg : 1.0) : 0.0, acc);
			cur = nxt;
		}
	}
#else
#pragma unroll
	for (int k = 0; k < ND_ITEMS_REG; ++k) {
		if (ql + step * k < n) {
			const int it = pre_it.it[k];
			nd_item_ops o;
			nd_item_load(rec, it, o);
			nd_item_apply(o, ((it >> 20) & 1) ? neg : 1.0, acc);
		}
	}
#endif
	for (int i = ql + step * ND_ITEMS_REG; i < n; i += step) {     // (long runs: the large windows)
		const int it = items[i];
		nd_item_ops o;
		nd_item_load(rec, it, o);
		nd_item_apply(o, ((it >> 20) & 1) ? neg : 1.0, acc);
	}
}

// the four lanes' sums of a quad, in the fixed order (l0 + l1) + (l2 + l3): every lane ends with the total
__device__ __forceinline__ void nd_quad_sum36(double (&acc)[36])
{
#pragma unroll
	for (int c = 0; c < 36; ++c) acc[c] += dpp_f64<0xB1>(acc[c]);      // quad_perm [1,0,3,2]
#pragma unroll
	for (int c = 0; c < 36; ++c) acc[c] += dpp_f64<0x4E>(acc[c]);      // quad_perm [2,3,0,1]
}

// ---- trial state: landmark steps and robust error of one group by one wavefront ----
// backsub: node lanes t_n = Zh_n^T (dt_e + dw_e x q_n) -> root lane delta_f = L^-T (c - sum t_n), f_new = f + delta_f -> the landmark's
// observation lanes; !backsub: the error at the current landmarks (the solve's first pass)
// (round 5, persistent kernel) the lane's operands of a scoring round with back-substitution -- the observation's trial transform, the lane's scratch
// record, its landmark -- fetched a GROUP ahead: they fly during the previous group's dependent chain (node terms -> root sum -> landmark step -> residual)
struct nd_spre { double T[12], sc[12], fl[3]; };
__device__ __forceinline__ void nd_prefetch_score(const lm_dev& D, const nd_dev& N, const nd_win& W, const nd_group G, const nd_recs& rc, const double* __restrict__ tab, int lane, nd_spre& o)
{
	const int li = W.lane0 + G.lane0 + min(lane, G.n_lanes - 1);
	nd_load12(tab + 12 * (size_t)(rc.orec.y & 0xFFFF), o.T);
	nd_load12_nt(N.scr + SCR * (size_t)li, o.sc);
	const double* fp = D.lm + 3 * (size_t)(W.lm0 + rc.lrec.x);
	o.fl[0] = fp[0]; o.fl[1] = fp[1]; o.fl[2] = fp[2];
}

template <bool PRE = false>
__device__ __forceinline__ void nd_score_group(const lm_dev& D, const nd_dev& N, const nd_win& W, const lm_cam& P, const nd_group G, const nd_recs rc,
                                               const double* __restrict__ tab, double* __restrict__ stage, const double* s_b, bool backsub, int it_idx,
                                               int lane, double& err, double& dfn, const nd_spre* pf = nullptr)
{
	const int li = W.lane0 + G.lane0 + min(lane, G.n_lanes - 1);
	const nd_i2 orec = rc.orec;
	const nd_f4 px = rc.px;
	double T[12];
	if (PRE) {
#pragma unroll
		for (int i = 0; i < 12; ++i) T[i] = pf->T[i];
	} else nd_load12(tab + 12 * (size_t)(orec.y & 0xFFFF), T);
	double f[3];
	if (backsub) {
		nd_i4 lrec = rc.lrec;
		if (lane >= G.n_lanes) lrec.w = 0;
		const int kind = lrec.w & 0xFF;
		double sc[12], fl[3];
		if (PRE) {
#pragma unroll
			for (int i = 0; i < 12; ++i) sc[i] = pf->sc[i];
			fl[0] = pf->fl[0]; fl[1] = pf->fl[1]; fl[2] = pf->fl[2];
		} else {
			nd_load12_nt(N.scr + SCR * (size_t)li, sc);
			const double* fp = D.lm + 3 * (size_t)(W.lm0 + lrec.x);
			fl[0] = fp[0]; fl[1] = fp[1]; fl[2] = fp[2];
		}
		if (kind == 2) {
			const double* d6 = s_b + 6 * ((lrec.z >> 16) & 0xFF);
			const double dw[3] = {d6[3], d6[4], d6[5]};
			double x[3];
			nd_cross(dw, sc + 9, x);
			const double a0 = d6[0] + x[0], a1 = d6[1] + x[1], a2 = d6[2] + x[2];
			double* o = stage + 3 * lane;
#pragma unroll
			for (int j = 0; j < 3; ++j) o[j] = sc[j] * a0 + sc[3 + j] * a1 + sc[6 + j] * a2;
		}
		WAVE_SYNC();
		double fnew[3] = {fl[0], fl[1], fl[2]};
		if (kind == 1 && lrec.z >= 0) {
			const int nn = (lrec.w >> 16) & 0xFF;
			double rhs[3] = {sc[6], sc[7], sc[8]};
			for (int i = 1; i <= nn; ++i) {
				const double* o = stage + 3 * (lane + i);
				rhs[0] -= o[0]; rhs[1] -= o[1]; rhs[2] -= o[2];
			}
			// delta_f = L^-T rhs   (sc[0..5] = L^-1 lower triangle packed 00 10 11 20 21 22)
			const double d3[3] = {sc[0] * rhs[0] + sc[1] * rhs[1] + sc[3] * rhs[2], sc[2] * rhs[1] + sc[4] * rhs[2], sc[5] * rhs[2]};
#pragma unroll
			for (int a = 0; a < 3; ++a) {
				fnew[a] = fl[a] + d3[a];
				dfn += d3[a] * d3[a];
				D.lm_new[3 * (size_t)(W.lm0 + lrec.x) + a] = fnew[a];
				if (D.trace_on) D.df_trace[W.df_trace0 + (size_t)it_idx * 3 * W.L + 3 * lrec.z + a] = d3[a];
			}
		}
		WAVE_SYNC();
		if (kind == 1) { double* o = stage + 3 * lane; o[0] = fnew[0]; o[1] = fnew[1]; o[2] = fnew[2]; }
		WAVE_SYNC();
		const double* fo = stage + 3 * ((orec.y >> 16) & 0xFF);
		f[0] = fo[0]; f[1] = fo[1]; f[2] = fo[2];
		WAVE_SYNC();                                               // (the next group of this wavefront overwrites the staging)
	} else {
		const double* fp = D.lm + 3 * (size_t)(W.lm0 + orec.x);
		f[0] = fp[0]; f[1] = fp[1]; f[2] = fp[2];
	}
	double pc[3], iz, r[4];
	nd_residual(P, T, f, px, pc, iz, r);
	const double s2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3];
	double rho = s2;
	if (P.robust) { const double k2 = P.kernel * P.kernel, u = 1.0 + s2 * rcp_f64(k2); rho = 2.0 * k2 * (u * nd_rsq(u) - 1.0); }
	if (lane < G.n_obs) err += rho;
}
