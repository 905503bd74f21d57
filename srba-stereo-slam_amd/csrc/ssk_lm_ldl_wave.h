// ssk_lm_ldl_wave.h -- the dense bordered LDL^T solve of a reduced system with at most 62 unknowns (E <= 10 optimised edges: the bench windows,
// "local LM window = 10") by ONE wavefront on the FP64 matrix cores, for the persistent node kernel (ssk_lm_node.hip).  Reference:
// solver_LM_schur_dense_cholesky, SRBA.h:34 (the dense Cholesky of H_pp - S + lambda I).
//
// The 64 x 64 padded system -- rows 0 .. n - 1 the unknowns, n .. 62 identity, row 63 the right-hand side (border row) -- is the lower triangle of a
// 4 x 4 grid of 16 x 16 tiles, ten tiles, ALL in this wavefront's registers in the layout of ssk_lm_ldl_mfma.h (tile (J, I), J <= I, TRANSPOSED as a
// v_mfma_f64_16x16x4_f64 accumulator: register r of lane l = A(16 I + l % 16, 16 J + 4 r + l / 16)).  A panel step k (16 columns), the phases in program
// order of the one wavefront (no flags, no workgroup barrier: WAVE_SYNC orders its LDS traffic):
//   A  the diagonal tile goes through LDS into a lane-per-row layout and is factorised with v_readlane column broadcasts; lanes 16 .. 31 carry the columns
//      of L'^-1 through the same instruction stream (the factor-wave code of ssk_lm_ldl_mfma.h, the next pivot's reciprocal started a column early);
//   B  the tiles below: U_Ik^T = L'^-1 A_Ik^T, four MFMAs each (A operand L'^-1 from LDS, B operand the tile itself), rows published to LDS;
//   C  the trailing tiles: tile(J, I) -= (U_Jk D^-1) U_Ik^T, four MFMAs each, operands from the published rows.
// 64 MFMAs and four 16-column chains replace ldl_solve_tiles4_lds's 15 steps x 3 workgroup barriers (+ 2 per back-substitution step): measured
// (tools/lm_ab.py) on the bench windows.  Back-substitution block by block from the bottom: contributions U_I'I^T x_I' by DPP row sums,
// x_I = L'^-T D^-1 (z_I - sum) with x going round by v_readlane.  Same factorisation as the other solvers (right-looking L' D L'^T, U = L' D unscaled,
// positive pivots or failure); every sum has a fixed order: bit-reproducible.
// Only wavefront 0 of the workgroup works; the others wait at the closing barrier.  lds: LDLW_DOUBLES doubles.
#pragma once

typedef double ldlw_v4d __attribute__((ext_vector_type(4)));
constexpr int LDLW_UP = 64;             // pitch of the published rows U^T (16 rows x 64 columns)
constexpr int LDLW_LP = 17;             // pitch of a 16 x 16 block in LDS
constexpr int LDLW_DOUBLES = 16 * LDLW_UP + 4 * 16 * LDLW_LP + 16 * LDLW_LP + 4 * 64;      // (>= 2560: the tiles' image at the start)
static_assert(LDLW_DOUBLES >= 10 * 256, "the tile image overlays the solver's LDS");

#define LDLW_PIN17(w, v) asm volatile("" : "+v"(w), "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), \
                                           "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]))

// sum over the 16 lanes that share l / 16, in every one of them (fixed tree: quads, then the row's four quads)
__device__ __forceinline__ double ldlw_row_sum16(double v)
{
	v += dpp_f64<0xB1>(v);              // quad_perm [1,0,3,2]
	v += dpp_f64<0x4E>(v);              // quad_perm [2,3,0,1]
	v += dpp_f64<0x124>(v);             // row_ror:4
	v += dpp_f64<0x128>(v);             // row_ror:8
	return v;
}

// tile (J, I) of the 4 x 4 grid, column-major over the lower triangle: (0,0) (0,1) (0,2) (0,3) (1,1) (1,2) (1,3) (2,2) (2,3) (3,3)
__device__ __forceinline__ constexpr int ldlw_tile(int J, int I) { return 4 * J - J * (J - 1) / 2 + (I - J); }

// Hb: the 6 x 6 block accumulators (H(6 e + a, 6 e2 + b), e <= e2, at 36 (e2 (e2 + 1) / 2 + e) + 6 a + b), g the reduced gradient, n = 6 E <= 60.
// Every thread of the workgroup calls it (three workgroup barriers); every thread returns the same flag.
__device__ __forceinline__ bool ldl_solve_mfma_wave(const double* Hb, const double* g, double lambda, int n, double* x_out, double* lds)
{
	__shared__ int s_ldlw_bad;
	double* Ut = lds;                                  // [16][LDLW_UP]: the eliminated rows U^T of the running panel
	double* Li = Ut + 16 * LDLW_UP;                    // [4][16][LDLW_LP]: L'^-1 of every diagonal block
	double* dgs = Li + 4 * 16 * LDLW_LP;               // the diagonal tile on its way into the lane-per-row layout; later: the block's contribution sums
	double* ivd = dgs + 16 * LDLW_LP;                  // 1 / d (0 beyond the unknowns)
	double* zs = ivd + 64;                             // the eliminated right-hand side
	double* xs = zs + 64;                              // the solution
	const int tid = threadIdx.x;
	// ---- the padded system as an image of the ten tiles' registers (register r of lane l of tile q at 256 q + 64 r + l), written by ALL threads of the
	// workgroup -- ten elements each, where one wavefront's lanes would compute forty indices into the block accumulators -- over the solver's LDS
	// (the published rows and inverses come later) ----
	for (int p = tid; p < 10 * 256; p += NT) {
		const int q = p >> 8, r = (p >> 6) & 3, l = p & 63;
		const int J = (q >= 4) + (q >= 7) + (q >= 9), I = J + q - (4 * J - J * (J - 1) / 2);
		const int a = 16 * J + 4 * r + (l >> 4), b = 16 * I + (l & 15);      // element (row b, column a), a <= b up to the diagonal tiles' upper halves
		const int lo = min(a, b), hi = max(a, b), e_lo = lo / 6, e_hi = hi / 6;
		const bool inside = hi < n;
		const double hv = Hb[inside ? 36 * (e_hi * (e_hi + 1) / 2 + e_lo) + 6 * (lo - 6 * e_lo) + (hi - 6 * e_hi) : 0];      // the stored triangle: H(lo, hi)
		double v = inside ? (a == b ? hv + lambda : hv) : (a == b ? 1.0 : 0.0);
		if (b == 63) v = a < n ? g[a] : (a == 63 ? 1.0 : 0.0);               // the border row: the right-hand side
		lds[p] = v;
	}
	__syncthreads();
	if (tid < 64) {
		// (the lane number made opaque: the caller runs this inside its LM loop, and addresses derived from a loop-invariant lane number are hoisted in
		// front of that loop, where they sit in registers -- or in scratch -- for the whole kernel)
		int lane = tid;
		asm volatile("" : "+v"(lane));
		const int lc = lane & 15, lr = lane >> 4;
		ldlw_v4d t[10];
#pragma unroll
		for (int q = 0; q < 10; ++q)
#pragma unroll
			for (int r = 0; r < 4; ++r) t[q][r] = lds[256 * q + 64 * r + lane];
		WAVE_SYNC();
		bool bad = false;
		// (the panel loop is NOT unrolled: the factorisation of a diagonal tile -- the bulk of the code, and of the scalar registers its column
		// broadcasts need -- exists once; the parts that touch tiles, which live in registers and need constant indices, are dispatched on k)
#pragma unroll 1
		for (int k = 0; k < 4; ++k) {
			double* Lik = Li + k * 16 * LDLW_LP;
			// ---- A: the diagonal tile, a row per lane (lanes 16 .. 31: the columns of L'^-1) ----
			{
				ldlw_v4d dt = t[ldlw_tile(3, 3)];
				if (k == 0) dt = t[ldlw_tile(0, 0)]; else if (k == 1) dt = t[ldlw_tile(1, 1)]; else if (k == 2) dt = t[ldlw_tile(2, 2)];
#pragma unroll
				for (int r = 0; r < 4; ++r) dgs[(4 * r + lr) * LDLW_LP + lc] = dt[r];
			}
			WAVE_SYNC();
			{
				const bool inv_lane = (lane & 48) == 16;
				double v[16];
#pragma unroll
				for (int j = 0; j < 16; ++j) { const double xv = dgs[j * LDLW_LP + lc]; v[j] = inv_lane ? (j == lc ? 1.0 : 0.0) : xv; }
				double d = readlane_f64(v[0], 0);
				bad |= 16 * k < n && !(d > 0.0);
				double ivj = 16 * k < n ? rcp_f64(d) : 0.0;
#pragma unroll
				for (int j = 0; j < 16; ++j) {
					const double f = v[j] * ivj;                            // lanes 0 .. 15: L'(row, j) below the diagonal
					if (lane == 0) ivd[16 * k + j] = ivj;
					double iv_next = 0.0;
					// the column's updates in five groups, one behind each stage of the next pivot's reciprocal (the pins keep the compiler from issuing
					// the dependent stages back to back and from putting updates off: ssk_lm_ldl_mfma.h)
#define LDLW_GROUP(gq) \
					_Pragma("unroll") for (int c = j + 2 + 3 * (gq); c < 16 && (c < j + 5 + 3 * (gq) || (gq) == 4); ++c) { const double lcj = readlane_f64(f, c); v[c] -= v[j] * lcj; }
					if (j + 1 < 16) {
						const double l1 = readlane_f64(f, j + 1);
						v[j + 1] -= v[j] * l1;
						d = readlane_f64(v[j + 1], j + 1);
						const bool real = 16 * k + j + 1 < n;
						bad |= real && !(d > 0.0);
						double rr = __builtin_amdgcn_rcp(d);
						LDLW_GROUP(0); LDLW_PIN17(rr, v);
						double e = fma(-d, rr, 1.0);
						LDLW_GROUP(1); LDLW_PIN17(e, v);
						rr = fma(e, rr, rr);
						LDLW_GROUP(2); LDLW_PIN17(rr, v);
						e = fma(-d, rr, 1.0);
						LDLW_GROUP(3); LDLW_PIN17(e, v);
						rr = fma(e, rr, rr);
						LDLW_GROUP(4); LDLW_PIN17(rr, v);
						iv_next = real ? rr : 0.0;
					}
#undef LDLW_GROUP
					ivj = iv_next;
				}
				if (inv_lane) {
#pragma unroll
					for (int i = 0; i < 16; ++i) Lik[i * LDLW_LP + lc] = v[i];
				}
				if (k == 3 && lane == 15) {                             // the border row (row 15 of the last diagonal tile): the last panel's part of z
#pragma unroll
					for (int j = 0; j < 16; ++j) zs[48 + j] = j < 15 ? v[j] : 0.0;
				}
			}
			WAVE_SYNC();
			if (__builtin_amdgcn_ballot_w64(bad) != 0) { bad = true; break; }      // (uniform)
			if (k < 3) {
				// ---- B: U_Ik^T = L'^-1 A_Ik^T of the tiles below, published as the rows' columns 16 I .. ----
				double la[4];
#pragma unroll
				for (int q = 0; q < 4; ++q) la[q] = Lik[lc * LDLW_LP + 4 * q + lr];
#define LDLW_TRSM(K, I) do { \
					ldlw_v4d c4 = {0.0, 0.0, 0.0, 0.0}; \
					_Pragma("unroll") for (int q = 0; q < 4; ++q) c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(la[q], t[ldlw_tile(K, I)][q], c4, 0, 0, 0); \
					t[ldlw_tile(K, I)] = c4; \
					_Pragma("unroll") for (int r = 0; r < 4; ++r) Ut[(4 * r + lr) * LDLW_UP + 16 * (I) + lc] = c4[r]; \
					if ((I) == 3 && lc == 15) {                         /* the border row's entries of this panel: the eliminated right-hand side */ \
						_Pragma("unroll") for (int r = 0; r < 4; ++r) zs[16 * (K) + 4 * r + lr] = c4[r]; \
					} \
				} while (0)
				if (k == 0) { LDLW_TRSM(0, 1); LDLW_TRSM(0, 2); LDLW_TRSM(0, 3); }
				else if (k == 1) { LDLW_TRSM(1, 2); LDLW_TRSM(1, 3); }
				else { LDLW_TRSM(2, 3); }
#undef LDLW_TRSM
				WAVE_SYNC();
				// ---- C: the trailing tiles ----
				double nd[4];
#pragma unroll
				for (int q = 0; q < 4; ++q) nd[q] = -ivd[16 * k + 4 * q + lr];
#define LDLW_UPD(J, I) do { \
					double a[4], b[4]; \
					_Pragma("unroll") for (int q = 0; q < 4; ++q) { a[q] = Ut[(4 * q + lr) * LDLW_UP + 16 * (J) + lc] * nd[q]; b[q] = Ut[(4 * q + lr) * LDLW_UP + 16 * (I) + lc]; } \
					_Pragma("unroll") for (int q = 0; q < 4; ++q) t[ldlw_tile(J, I)] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], t[ldlw_tile(J, I)], 0, 0, 0); \
				} while (0)
				if (k == 0) { LDLW_UPD(1, 1); LDLW_UPD(1, 2); LDLW_UPD(1, 3); LDLW_UPD(2, 2); LDLW_UPD(2, 3); LDLW_UPD(3, 3); }
				else if (k == 1) { LDLW_UPD(2, 2); LDLW_UPD(2, 3); LDLW_UPD(3, 3); }
				else { LDLW_UPD(3, 3); }
#undef LDLW_UPD
				WAVE_SYNC();
			}
		}
		if (!bad) {
			// ---- back-substitution, block by block from the bottom: x_I = L'_II^-T D^-1 (z_I - sum over the later blocks of U_I'I^T x_I') ----
			double xb[4] = {0.0, 0.0, 0.0, 0.0};               // lane (mod 16) b: x of the blocks solved so far
#pragma unroll
			for (int I = 3; I >= 0; --I) {
				// the later blocks' contributions: tile (I, I') holds U_I'I^T; component 4 r + lr = the sum over the 16 lanes that share lr
				double cs[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
				for (int I2 = 3; I2 > I; --I2) {
#pragma unroll
					for (int r = 0; r < 4; ++r) cs[r] += ldlw_row_sum16(t[ldlw_tile(I, I2)][r] * xb[I2]);
				}
				if (I < 3) {
					if (lc == 0) {
#pragma unroll
						for (int r = 0; r < 4; ++r) dgs[4 * r + lr] = cs[r];
					}
					WAVE_SYNC();
				}
				const int gi = 16 * I + lc;
				const double sum = I < 3 ? dgs[lc] : 0.0;
				const double v = gi < n ? ivd[gi] * (zs[gi] - sum) : 0.0;
				double li[16];
#pragma unroll
				for (int b = 0; b < 16; ++b) li[b] = Li[I * 16 * LDLW_LP + b * LDLW_LP + lc];
				double xa[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
				for (int cc = 0; cc < 16; ++cc) xa[cc & 3] += li[cc] * readlane_f64(v, cc);
				const double x = (xa[0] + xa[1]) + (xa[2] + xa[3]);
				xb[I] = x;
				if (lane < 16) xs[gi] = x;
				WAVE_SYNC();
			}
		}
		if (tid == 0) s_ldlw_bad = bad ? 1 : 0;
	}
	__syncthreads();
	if (s_ldlw_bad) return false;                      // uniform
	for (int i = tid; i < n; i += NT) x_out[i] = xs[i];
	__syncthreads();
	return true;
}
