// ssk_lm_ldl_mfma.h -- the dense bordered LDL^T solve of the phase-split LM's K2 (ssk_lm_split_solve.hip) with the WHOLE matrix in the
// accumulator registers of the FP64 matrix cores.  Reference: solver_LM_schur_dense_cholesky, SRBA.h:34 (the dense Cholesky of the reduced
// system H_pp - S + lambda I); same factorisation as the other solvers of ssk_lm_device.h (right-looking L' D L'^T, U = L' D unscaled, the
// border row = right-hand side eliminated on the way) up to the order of an update's terms.
//
// Layout.  The (n + 1) x (n + 1) bordered system (n <= 255: 42 optimised edges) padded to 16 T rows is cut into 16 x 16 tiles of its lower
// triangle, tile (J, I), J <= I, held TRANSPOSED as one v_mfma_f64_16x16x4_f64 accumulator: register r of lane l = A(16 I + l % 16, 16 J + 4 r + l / 16).
// T (T + 1) / 2 <= 136 tiles dealt column-major over SEVEN of the workgroup's eight wavefronts (20 tiles = 160 VGPRs per lane at most; a tile
// column retires evenly from all of them); the eighth is the FACTOR wave.  An accumulator tile is, register by register, exactly the B operand of
// the instruction, so LEFT multiplication of a tile needs no data movement.  A panel step k (16 columns):
//   A  the factor wave takes the diagonal tile (k, k) from LDS in a lane-per-row layout, factorises it (column broadcasts by v_readlane) and
//      inverts the unit factor on the way (lanes 16 .. 31: column c of L'^-1 -- the same scalars L'(c, j) and the same instruction drive both),
//      publishes L'^-1 and 1 / d and raises a flag;
//   B  the owners of the tiles (k, I) below: U_Ik^T = L'^-1 A_Ik^T -- four MFMAs, A operand L'^-1 from LDS, B operand the tile's own registers --
//      and publish the rows U^T (16 x 16 T doubles).  The critical chain first: tile (k, k + 1), then -- on a flag, not a barrier -- the owner of
//      the NEXT diagonal tile (k + 1, k + 1) updates it with those 16 rows and hands it to the factor wave: step A of panel k + 1 runs beside the
//      rest of B and all of C of panel k (measured, tools/ldl_bench.hip: the 16-column dependent chains of A cost 5 us of a 103 us solve);
//   C  after a barrier among the seven tile waves (an LDS counter; the rows of consecutive panels alternate between two buffers): every tile (J, I),
//      J > k: tile -= (U_Jk D^-1)(U_Ik^T): four MFMAs, both operands from the published rows (conflict-free: a lane reads row 4 q + l / 16,
//      column 16 X + l % 16), the A operand scaled by -1 / d on the way, two tiles' MFMA chains interleaved.
// No workgroup barrier inside the panel loop: the factor wave never waits for the tile waves' phases.
// Back-substitution by 16-row blocks from the bottom, on flags as well: x_I = L'_II^-T (D^-1 (z_I - sum of the later blocks' contributions)) by the
// factor wave (the inverses of all diagonal blocks stay in LDS; so do the tiles (k, k + 1), whose contribution -- the one the next block waits for --
// the factor wave computes itself, x going round by v_readlane); the tile waves hand the contributions U_IJ^T x_I of their other tiles (row sums by
// DPP) to the blocks above through LDS slots that are summed in a fixed order.  Every sum has one owner and a fixed order: bit-reproducible.
// Measured (tools/ldl_bench.hip, one workgroup, per solve): 240 unknowns 103 us against 171 us of ldl_solve_regtile, 120: 41 / 79 us, 12: 9 / 12 us.
// lds: LDLM_DOUBLES doubles (138 KB).  NT must be 512.
#pragma once
#include <type_traits>

typedef double ldlm_v4d __attribute__((ext_vector_type(4)));
#ifdef SSK_LM_STAMPS
// diagnostics build (tools/split_variant.sh stamps): clock sums of workgroup 0 -- thread 0 (tile wave 0): 0 load, 1 wait for L'^-1, 6 B, 2 barrier among the tile waves, 3 C, 4 loop exit, 5 back-substitution;
// lane 0 of the factor wave: 7 wait for the diagonal tile, 8 elimination + L'^-1; 15: solves
__device__ unsigned long long g_k2_t[16];
// (sums kept in registers and written once at the end: a global read-modify-write per stamp would put its latency into the next phase)
#define K2_T0() unsigned long long k2_t_ = __builtin_amdgcn_s_memtime(), k2_a_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define K2_T(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); k2_a_[i] += t_ - k2_t_; k2_t_ = t_; } while (0)
#define K2_D0() k2_t_ = __builtin_amdgcn_s_memtime()
#define K2_D(i) K2_T(i)
#define K2_FLUSH() do { if ((threadIdx.x == 0 || threadIdx.x == NT - 64) && blockIdx.x == 0) { _Pragma("unroll") for (int i_ = 0; i_ < 10; ++i_) if (k2_a_[i_]) atomicAdd(&g_k2_t[i_], k2_a_[i_]); } } while (0)
#else
#define K2_T0() do {} while (0)
#define K2_T(i) do {} while (0)
#define K2_D0() do {} while (0)
#define K2_D(i) do {} while (0)
#define K2_FLUSH() do {} while (0)
#endif
constexpr int LDLM_TW = 7;              // wavefronts that hold tiles (the eighth factorises the diagonal tiles)
constexpr int LDLM_TS = 20;             // tile slots of a wavefront: ceil(136 / 7)
constexpr int LDLM_UP = 256;            // pitch of the published rows U^T
constexpr int LDLM_LP = 17;             // pitch of a 16 x 16 block in LDS
constexpr int LDLM_NMAX = 255;          // largest system (n + 1 <= 256)
constexpr int LDLM_DOUBLES = 2 * 16 * LDLM_UP + 16 * 16 * LDLM_LP + 15 * 16 * LDLM_LP + 16 * LDLM_LP + 3 * 256;

// a workgroup barrier that orders LDS traffic only: what a chunk step needs between its commit and its operand reads (the loads prefetched for the next
// chunk stay in flight).  Measured no different from __syncthreads(): a workgroup-scope fence does not wait on vmcnt in this execution mode (the CU's
// vector memory operations complete in order); kept because it states the intent
#define LDLM_BARRIER_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define LDLM_LD(x) __hip_atomic_load(&x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define LDLM_ST(x, v) __hip_atomic_store(&x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define LDLM_PIN17(w, v) asm volatile("" : "+v"(w), "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), \
                                           "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]))
#define LDLM_PIN16(v) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), \
                                        "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]))

// slot s (a wave-uniform value) -> f(integral_constant<s>): the tiles live in registers, a slot number must be a constant where a tile is touched;
// a binary tree of scalar branches (five to a slot of twenty)
template <int LO, int HI, class F>
__device__ __forceinline__ void ldlm_dispatch(int s, F&& f)
{
	if constexpr (LO == HI) f(std::integral_constant<int, LO>{});
	else {
		constexpr int MID = (LO + HI) / 2;
		if (s <= MID) ldlm_dispatch<LO, MID>(s, f);
		else ldlm_dispatch<MID + 1, HI>(s, f);
	}
}

// sum over the 16 lanes that share l / 16, in every one of them (fixed tree: quads, then the row's four quads)
__device__ __forceinline__ double ldlm_row_sum16(double v)
{
	v += dpp_f64<0xB1>(v);              // quad_perm [1,0,3,2]
	v += dpp_f64<0x4E>(v);              // quad_perm [2,3,0,1]
	v += dpp_f64<0x124>(v);             // row_ror:4
	v += dpp_f64<0x128>(v);             // row_ror:8
	return v;
}

// lead > 0 (round 5: systems beyond 255 unknowns, ldl_solve_mfma_lead below): A is the TRAILING block of a larger system Afull whose `lead` leading columns
// are factorised already (ldl_leading_mfma: U = L' D unscaled in Afull's lower triangle, 1 / d in inv_lead); the tiles take the leading columns'
// updates right after the load -- chunks of 32 leading columns staged by the whole workgroup in the layout of a panel's published rows, eight MFMAs per
// tile and chunk (the trailing update of the panel loop, applied to every tile).
__device__ __forceinline__ bool ldl_solve_mfma_tiles(const double* __restrict__ A, int np, int n, double* x_out, double* lds,
                                                     int lead = 0, const double* __restrict__ Afull = nullptr, const double* inv_lead = nullptr)
{
	static_assert(NT == 512, "ldl_solve_mfma_tiles: 8 wavefronts");
	const int tid = threadIdx.x, lane = tid & 63, lc = lane & 15, lr = lane >> 4;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const bool tiles = wave < LDLM_TW;
	const int T = (n + 16) >> 4;                       // tile rows, border row included
	const int bcol = n - 16 * (T - 1);                 // the border row inside the last tile row
	// the panel loop runs on flags, not barriers (the factor wave must not wait for the tile waves' phases):
	// s_flag: diagonal tiles handed to the factor wave; s_flag_li: panels whose L'^-1 and 1 / d are published; s_flag_ut: panels whose rows of block
	// k + 1 are published; s_cnt: tile waves past step B (7 per panel: the barrier among the tile waves between B and C)
	__shared__ int s_bad, s_flag, s_flag_li, s_flag_ut, s_cnt;
	__shared__ int s_xcnt, s_ccnt[16];                 // back-substitution: blocks solved; contributions received per block
	double* Ut2 = lds;                                 // [2][16][LDLM_UP]: the eliminated rows U^T of panel k in buffer k & 1; after the factorisation: the contribution slots
	double* Li = Ut2 + 2 * 16 * LDLM_UP;               // [T][16][LDLM_LP]: L'^-1 of every diagonal block
	double* Usub = Li + 16 * 16 * LDLM_LP;             // [T - 1][16][LDLM_LP]: the tiles (k, k + 1) below the diagonal, U^T, kept for the back-substitution's critical chain
	double* dgs = Usub + 15 * 16 * LDLM_LP;            // the diagonal tile on its way to the factor wave
	double* ivd = dgs + 16 * LDLM_LP;                  // 1 / d (0 beyond the unknowns)
	double* zs = ivd + 256;                            // the eliminated right-hand side
	double* xs = zs + 256;                             // the solution
	K2_T0();
	if (tid == 0) { s_bad = 0; s_flag = 0; s_flag_li = 0; s_flag_ut = 0; s_cnt = 0; s_xcnt = 0; }
	if (tid < 16) s_ccnt[tid] = 0;
	__syncthreads();
	// The factor wave and the tile waves run separate code from here on (the same barriers in both): the tiles' 160 registers are not live in
	// the factor wave's code, nor its 16 x 16 block in theirs.
	double* contrib = Ut2;                                 // back-substitution: [J][I][16]
	// chunks of 32 leading columns, rows lead .. lead + n (the border row last) of Afull, staged like a panel's published rows: Sc[column][row] (rows beyond:
	// zero -- the identity padding takes no update), pitch = 17 (mod 32): conflict-free transposing writes and operand reads.  The next chunk's loads are in
	// flight under this one's MFMAs.  (32 x 273 doubles: over Ut2 into Li, which the factorisation has not touched yet.)
	constexpr int CUP = 273;
	const int n_chunks = lead > 0 ? (lead + 31) >> 5 : 0;
	const int cntc = (16 * T) << 5;
	double* Sc = lds;
	double cv[16];
	auto fetch_chunk = [&](int p0) {            // (thread: column tid % 32, rows tid / 32 + 16 j: a uniform row base per load, one offset register per thread)
		int tq = tid;
		asm volatile("" : "+v"(tq));                       // (the offsets are derived at every use: hoisted out of the chunk loop they spill)
		const int col = min(p0 + (tq & 31), lead - 1), rq = tq >> 5;
#pragma unroll
		for (int j = 0; j < 16; ++j) {
			if (NT * j < cntc) {
				const int rj = 16 * j, rb = min(rj, max(n - 15, 0));
				const double* base = Afull + (size_t)(lead + rb) * np;
				cv[j] = base[(min(rq + rj, n) - rb) * np + col];
			}
		}
	};
	auto commit_chunk = [&](int p0) {
		int tq = tid;
		asm volatile("" : "+v"(tq));
		const int kk = tq & 31, rq = tq >> 5;
		double* Sk = Sc + kk * CUP;
#pragma unroll
		for (int j = 0; j < 16; ++j) {
			if (NT * j < cntc) { const int r = rq + 16 * j; Sk[r] = r <= n && p0 + kk < lead ? cv[j] : 0.0; }
		}
	};
	if (n_chunks > 0) fetch_chunk(0);
	if (!tiles) {
		// ================= the factor wave =================
		for (int ch = 0; ch < n_chunks; ++ch) {            // (the leading columns' chunks: this wave stages its share; the tile waves apply them)
			LDLM_BARRIER_LDS();
			commit_chunk(32 * ch);
			if (ch + 1 < n_chunks) fetch_chunk(32 * (ch + 1));
			LDLM_BARRIER_LDS();
		}
		if (n_chunks > 0) __syncthreads();                 // (the last chunk's reads are done: Li is the factorisation's from here on)
		bool bad = false;
		for (int k = 0; k < T && !bad; ++k) {
			double* Lik = Li + k * 16 * LDLM_LP;
			K2_D0();
			while (LDLM_LD(s_flag) <= k) __builtin_amdgcn_s_sleep(1);
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
			K2_D(7);
			// lanes 0 .. 15: row lc of the tile; lanes 16 .. 31: column lc of L'^-1 (the same update v[c] -= v[j] L'(c, j) serves both: one instruction)
			const bool inv_lane = (lane & 48) == 16;
			double v[16];
#pragma unroll
			for (int j = 0; j < 16; ++j) { const double xv = dgs[j * LDLM_LP + lc]; v[j] = inv_lane ? (j == lc ? 1.0 : 0.0) : xv; }
			// column j: the next pivot's reciprocal is started as soon as v[j + 1] has its update, so that its dependent chain (rcp + two Newton
			// steps, ~100 cycles of latency on an in-order wave) runs under the column's remaining updates instead of after them
			double d = readlane_f64(v[0], 0);
			bad |= 0 < n - 16 * k && !(d > 0.0);
			double ivj = 0 < n - 16 * k ? rcp_f64(d) : 0.0;
#pragma unroll
			for (int j = 0; j < 16; ++j) {
				const double f = v[j] * ivj;                                // lanes 0 .. 15: L'(row, j) below the diagonal
				if (lane == 0) ivd[16 * k + j] = ivj;
				double iv_next = 0.0;
				// the column's updates in five groups, one behind each stage of the next pivot's reciprocal (v_rcp_f64, two Newton steps): the pins keep
				// the compiler from issuing the dependent stages back to back (an in-order wave would sit through their latencies) and from putting
				// updates off (their 120 scalars would spill the SGPRs)
#define LDLM_GROUP(g) \
				_Pragma("unroll") for (int c = j + 2 + 3 * (g); c < 16 && (c < j + 5 + 3 * (g) || (g) == 4); ++c) { const double lcj = readlane_f64(f, c); v[c] -= v[j] * lcj; }
				if (j + 1 < 16) {
					const double l1 = readlane_f64(f, j + 1);               // L'(c, j): one scalar for the elimination and the inverse
					v[j + 1] -= v[j] * l1;                                  // (rows above c carry unused values)
					d = readlane_f64(v[j + 1], j + 1);
					const bool real = 16 * k + j + 1 < n;
					bad |= real && !(d > 0.0);
					double r = __builtin_amdgcn_rcp(d);
					LDLM_GROUP(0); LDLM_PIN17(r, v);
					double e = fma(-d, r, 1.0);
					LDLM_GROUP(1); LDLM_PIN17(e, v);
					r = fma(e, r, r);
					LDLM_GROUP(2); LDLM_PIN17(r, v);
					e = fma(-d, r, 1.0);
					LDLM_GROUP(3); LDLM_PIN17(e, v);
					r = fma(e, r, r);
					LDLM_GROUP(4); LDLM_PIN17(r, v);
					iv_next = real ? r : 0.0;
				}
#undef LDLM_GROUP
				ivj = iv_next;
			}
			if (inv_lane) {
#pragma unroll
				for (int i = 0; i < 16; ++i) Lik[i * LDLM_LP + lc] = v[i];
			}
			if (k == T - 1 && lane == bcol) {
#pragma unroll
				for (int j = 0; j < 16; ++j) zs[16 * k + j] = j < bcol ? v[j] : 0.0;
			}
			if (bad && lane == 0) s_bad = k + 1;                // (the failing panel + 1: the tile waves all leave AT that panel, see below)
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			if (lane == 0) LDLM_ST(s_flag_li, k + 1);
			K2_D(8);
		}
		__syncthreads();                                   // the factorisation is complete
		if (s_bad) return false;                           // uniform
		// back-substitution, block by block from the bottom: this wave solves a block as soon as the contributions of all later blocks are in
		// (counted per block), the tile waves take a block's solution as soon as it is published -- flags, no barriers
		// (the contribution of the block just solved -- tile (I, I + 1), the one thing the next block waits for -- is this wave's own work, from the
		// copy of the tile in LDS: x stays in the lanes' registers and goes round by v_readlane; the tile waves' contributions are a block or more old)
		// Dependent FP64 operations cost ~30 cycles each on an in-order wave: every sum below runs as four chains, and the next block's tile and
		// inverse are read while this block's chains run.
		double xprev = 0.0;                                // lane (mod 16) b: x of block I + 1
		double us[16], li[16];
#pragma unroll
		for (int b = 0; b < 16; ++b) { us[b] = 0.0; li[b] = Li[(T - 1) * 16 * LDLM_LP + b * LDLM_LP + lc]; }
		for (int I = T - 1; I >= 0; --I) {
			const int g = 16 * I + lc;
			const double zg = zs[g], ig = ivd[g];
			double c[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
			for (int b = 0; b < 16; ++b) c[b & 3] += us[b] * readlane_f64(xprev, b);          // (us = 0 for the last block)
			while (LDLM_LD(s_ccnt[I]) < T - 2 - I) __builtin_amdgcn_s_sleep(0);
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
			double cv[14];
#pragma unroll
			for (int o = 0; o < 14; ++o) cv[o] = contrib[(I * 16 + min(I + 2 + o, 15)) * 16 + lc];
			double lin[16];
#pragma unroll
			for (int b = 0; b < 16; ++b) lin[b] = li[b];
			if (I > 0) {                                   // the next block's operands, on their way while the chains below run
#pragma unroll
				for (int b = 0; b < 16; ++b) {
					us[b] = Usub[(I - 1) * 16 * LDLM_LP + lc * LDLM_LP + b];
					li[b] = Li[(I - 1) * 16 * LDLM_LP + b * LDLM_LP + lc];
				}
			}
#pragma unroll
			for (int o = 0; o < 14; ++o) c[o & 3] += I + 2 + o < T ? cv[o] : 0.0;
			const double sum = (c[0] + c[1]) + (c[2] + c[3]);
			const double v = g < n ? ig * (zg - sum) : 0.0;
			double xa[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
			for (int cc = 0; cc < 16; ++cc) xa[cc & 3] += lin[cc] * readlane_f64(v, cc);
			const double x = (xa[0] + xa[1]) + (xa[2] + xa[3]);
			if (lane < 16) xs[g] = x;
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			if (lane == 0) LDLM_ST(s_xcnt, T - I);
			xprev = x;
		}
		__syncthreads();
	} else {
		// ================= the tile waves =================
		// ---- load: slot s of tile wave w holds tile 7 s + w of the column-major order (the system's upper triangle is stored too: rows along the
		// tile's columns, coalesced) ----
		const int n_tiles = T * (T + 1) / 2;
		ldlm_v4d t[LDLM_TS];
		int sT[LDLM_TS];                                   // the slots' tiles, J | I << 8 (T | T << 8: none): the back-substitution's lookup
		{
			int J = 0, base = 0;
	#pragma unroll
			for (int s = 0; s < LDLM_TS; ++s) {
				const int idx = LDLM_TW * s + wave;
	#pragma unroll 1
				while (J < T && idx >= base + (T - J)) { base += T - J; ++J; }
				const int I = J + idx - base;
				sT[s] = J < T ? (J | I << 8) : (T | T << 8);
				// every load unconditional, from a clamped address (an empty slot: tile (0, 0); beyond the border row: the border row), the masks
				// afterwards: behind a branch per slot (inner tile / last tile row) every load of a last-row tile was followed by a wait for ALL loads
				// in flight (round 5: the same pattern cost ldl_leading_mfma 6.6 us per panel)
				const int Jc = J < T ? J : 0, Ic = J < T ? I : 0;
	#pragma unroll
				for (int r = 0; r < 4; ++r) {
					const int a = min(16 * Jc + 4 * r + lr, n - 1), b = 16 * Ic + lc;      // element (row b, column a) of the system
					t[s][r] = A[b >= n ? (size_t)n * np + a : (size_t)a * np + b];
				}
			}
	#pragma unroll
			for (int s = 0; s < LDLM_TS; ++s) {
				const int J = sT[s] & 0xFF, I = sT[s] >> 8;
	#pragma unroll
				for (int r = 0; r < 4; ++r) {
					const int a = 16 * J + 4 * r + lr, b = 16 * I + lc;
					t[s][r] = J >= T ? 0.0 : (a < n && b <= n) ? t[s][r] : (a == b ? 1.0 : 0.0);
				}
			}
		}
		for (int ch = 0; ch < n_chunks; ++ch) {
			LDLM_BARRIER_LDS();
			commit_chunk(32 * ch);
			if (ch + 1 < n_chunks) fetch_chunk(32 * (ch + 1));
			LDLM_BARRIER_LDS();
			const int hmax = min(32, lead - 32 * ch);          // (a last chunk of 16 columns: one group)
#pragma unroll 1
			for (int h = 0; h < hmax; h += 16) {
				double nd[4];
#pragma unroll
				for (int q = 0; q < 4; ++q) { const int c = 32 * ch + h + 4 * q + lr; nd[q] = c < lead ? -inv_lead[min(c, lead - 1)] : 0.0; }
				// two slots at a time, branch-free (an empty slot -- T | T << 8 -- works on its zero tile with clamped operands): all sixteen operands are
				// read first, the two tiles' MFMA chains alternate
#pragma unroll
				for (int s = 0; s < LDLM_TS; s += 2) {
					int st0 = sT[s], st1 = sT[s + 1];
					asm volatile("" : "+s"(st0), "+s"(st1));
					const int cj0 = 16 * min(st0 & 0xFF, T - 1), ci0 = 16 * min(st0 >> 8, T - 1), cj1 = 16 * min(st1 & 0xFF, T - 1), ci1 = 16 * min(st1 >> 8, T - 1);
					double a0[4], b0[4], a1[4], b1[4];
#pragma unroll
					for (int q = 0; q < 4; ++q) {
						const double* row = Sc + (h + 4 * q + lr) * CUP + lc;
						a0[q] = row[cj0] * nd[q]; b0[q] = row[ci0];
						a1[q] = row[cj1] * nd[q]; b1[q] = row[ci1];
					}
#pragma unroll
					for (int q = 0; q < 4; ++q) {
						t[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[q], b0[q], t[s], 0, 0, 0);
						t[s + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[q], b1[q], t[s + 1], 0, 0, 0);
					}
				}
			}
		}
		if (n_chunks > 0) __syncthreads();                 // (Li and the published rows' buffers are the factorisation's from here on)
		if (wave == 0) {                                   // the first diagonal tile to the factor wave
#pragma unroll
			for (int r = 0; r < 4; ++r) dgs[(4 * r + lr) * LDLM_LP + lc] = t[0][r];
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			if (lane == 0) LDLM_ST(s_flag, 1);
		}
		K2_T(0);
		for (int k = 0; k < T; ++k) {
			double* Lik = Li + k * 16 * LDLM_LP;
			while (LDLM_LD(s_flag_li) <= k) __builtin_amdgcn_s_sleep(1);
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
			K2_T(1);
			// A failed pivot (the contract's failed Cholesky) ends the factorisation for every wave at the SAME panel: the factor wave runs up to a
			// panel ahead of the slowest tile wave (the owners of tile (k, k + 1) and of the next diagonal tile hand panel k + 1 over before the
			// counter barrier of panel k), so a wave still on its way to panel k may already see the failure of panel k + 1 -- it must not leave
			// before panel k's counter has its seven arrivals, or the waves that passed this check earlier spin on it for ever.  s_bad = failing
			// panel + 1, written before that panel's s_flag_li: every tile wave reads it here at the failing panel itself.
			{ const int sb = LDLM_LD(s_bad); if (sb != 0 && sb - 1 <= k) break; }      // uniform
			double* Ut = Ut2 + (k & 1) * 16 * LDLM_UP;
			const int dk = k * T - k * (k - 1) / 2, dk1 = dk + T - k;  // column-major numbers of the diagonal tiles (k, k) and (k + 1, k + 1)
			// this wave's slots of the panel (-1 / empty ranges: none): the tile (k, k + 1), the next diagonal tile, the column's other tiles
			const int s_sub = (k + 1 < T && (dk + 1) % LDLM_TW == wave) ? (dk + 1) / LDLM_TW : -1;
			const int s_dg = (k + 1 < T && dk1 % LDLM_TW == wave) ? dk1 / LDLM_TW : -1;
			const int b_lo = (dk + 2 - wave + LDLM_TW - 1 + LDLM_TW) / LDLM_TW - 1, b_hi = (dk1 - 1 - wave + LDLM_TW) / LDLM_TW - 1;      // 7 s + wave in [dk + 2, dk1 - 1]
			const int c_lo = (dk1 + 1 - wave + LDLM_TW - 1 + LDLM_TW) / LDLM_TW - 1, c_hi = (n_tiles - 1 - wave + LDLM_TW) / LDLM_TW - 1; // 7 s + wave in [dk1 + 1, n_tiles - 1]
			{
				double la[4];
#pragma unroll
				for (int q = 0; q < 4; ++q) la[q] = Lik[lc * LDLM_LP + 4 * q + lr];
				// U_Ik^T = L'^-1 A_Ik^T of the tile in slot sl, published as the rows' columns 16 I ..
#define LDLM_TRSM(sl, I) do { \
					ldlm_v4d c4 = {0.0, 0.0, 0.0, 0.0}; \
					_Pragma("unroll") for (int q = 0; q < 4; ++q) c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(la[q], t[sl][q], c4, 0, 0, 0); \
					t[sl] = c4; \
					_Pragma("unroll") for (int r = 0; r < 4; ++r) Ut[(4 * r + lr) * LDLM_UP + 16 * (I) + lc] = c4[r]; \
					if ((I) == k + 1) { _Pragma("unroll") for (int r = 0; r < 4; ++r) Usub[k * 16 * LDLM_LP + (4 * r + lr) * LDLM_LP + lc] = c4[r]; } \
					if ((I) == T - 1 && lc == bcol) {                       /* the border row's entries of this panel: the eliminated right-hand side */ \
						_Pragma("unroll") for (int r = 0; r < 4; ++r) zs[16 * k + 4 * r + lr] = c4[r]; \
					} \
				} while (0)
#pragma unroll
				for (int s = 0; s < LDLM_TS; ++s) {
					if (s == s_sub) {
						LDLM_TRSM(s, k + 1);
						__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
						if (lane == 0) LDLM_ST(s_flag_ut, k + 1);
					}
				}
#pragma unroll
				for (int s = 0; s < LDLM_TS; ++s) {
					if (s == s_dg) {
						while (LDLM_LD(s_flag_ut) <= k) __builtin_amdgcn_s_sleep(1);
						__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
						double a[4], b[4];
#pragma unroll
						for (int q = 0; q < 4; ++q) { b[q] = Ut[(4 * q + lr) * LDLM_UP + 16 * (k + 1) + lc]; a[q] = b[q] * -ivd[16 * k + 4 * q + lr]; }
#pragma unroll
						for (int q = 0; q < 4; ++q) t[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], t[s], 0, 0, 0);
#pragma unroll
						for (int r = 0; r < 4; ++r) dgs[(4 * r + lr) * LDLM_LP + lc] = t[s][r];
						__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
						if (lane == 0) LDLM_ST(s_flag, k + 2);
					}
				}
#pragma unroll
				for (int s = 0; s < LDLM_TS; ++s) {
					if (s >= b_lo && s <= b_hi) { const int I = k + LDLM_TW * s + wave - dk; LDLM_TRSM(s, I); }
				}
#undef LDLM_TRSM
			}
			K2_T(6);
			// the tile waves' barrier between B and C (the rows of panel k + 1 go to the other buffer: a wave may run ahead into B of the next panel
			// while a slower one still reads this panel's rows, and nobody reaches B of panel k + 2 before all have passed this point of panel k + 1)
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			if (lane == 0) atomicAdd(&s_cnt, 1);
			while (LDLM_LD(s_cnt) < LDLM_TW * (k + 1)) __builtin_amdgcn_s_sleep(1);
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
			K2_T(2);
			// ---- C: the trailing tiles (the next diagonal tile has its update already), two slots at a time: the MFMAs of one accumulator depend on
			// each other, those of two tiles alternate ----
			{
				double nd[4];
#pragma unroll
				for (int q = 0; q < 4; ++q) nd[q] = -ivd[16 * k + 4 * q + lr];
				int J = k + 1, base = dk1;                     // the column of the running tile and its first tile's number
#pragma unroll
				for (int s = 0; s < LDLM_TS; s += 2) {
					const bool on0 = s >= c_lo && s <= c_hi, on1 = s + 1 >= c_lo && s + 1 <= c_hi;
					if (!on0 && !on1) continue;
					int cj0 = 0, ci0 = 0, cj1 = 0, ci1 = 0;
					if (on0) {
						const int idx = LDLM_TW * s + wave;
#pragma unroll 1
						while (idx >= base + (T - J)) { base += T - J; ++J; }
						cj0 = 16 * J; ci0 = 16 * (J + idx - base);
					}
					if (on1) {
						const int idx = LDLM_TW * (s + 1) + wave;
#pragma unroll 1
						while (idx >= base + (T - J)) { base += T - J; ++J; }
						cj1 = 16 * J; ci1 = 16 * (J + idx - base);
					}
					if (on0 && on1) {
						double a0[4], b0[4], a1[4], b1[4];
#pragma unroll
						for (int q = 0; q < 4; ++q) {
							a0[q] = Ut[(4 * q + lr) * LDLM_UP + cj0 + lc] * nd[q]; b0[q] = Ut[(4 * q + lr) * LDLM_UP + ci0 + lc];
							a1[q] = Ut[(4 * q + lr) * LDLM_UP + cj1 + lc] * nd[q]; b1[q] = Ut[(4 * q + lr) * LDLM_UP + ci1 + lc];
						}
#pragma unroll
						for (int q = 0; q < 4; ++q) {
							t[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[q], b0[q], t[s], 0, 0, 0);
							t[s + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[q], b1[q], t[s + 1], 0, 0, 0);
						}
					} else if (on0) {
						double a0[4], b0[4];
#pragma unroll
						for (int q = 0; q < 4; ++q) { a0[q] = Ut[(4 * q + lr) * LDLM_UP + cj0 + lc] * nd[q]; b0[q] = Ut[(4 * q + lr) * LDLM_UP + ci0 + lc]; }
#pragma unroll
						for (int q = 0; q < 4; ++q) t[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[q], b0[q], t[s], 0, 0, 0);
					} else {
						double a1[4], b1[4];
#pragma unroll
						for (int q = 0; q < 4; ++q) { a1[q] = Ut[(4 * q + lr) * LDLM_UP + cj1 + lc] * nd[q]; b1[q] = Ut[(4 * q + lr) * LDLM_UP + ci1 + lc]; }
#pragma unroll
						for (int q = 0; q < 4; ++q) t[s + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[q], b1[q], t[s + 1], 0, 0, 0);
					}
				}
			}
			K2_T(3);
		}
		__syncthreads();                                   // the factorisation is complete
		K2_T(4);
		if (s_bad) return false;                           // uniform
		for (int I = T - 1; I > 1; --I) {
			while (LDLM_LD(s_xcnt) < T - I) __builtin_amdgcn_s_sleep(0);
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
			const double xb = xs[16 * I + lc];
			// this wave's tiles (J, I), the largest J first: block I - 1 is the next to be solved
#pragma unroll
			for (int s = LDLM_TS - 1; s >= 0; --s) {
				int st = sT[s];
				asm volatile("" : "+s"(st));                                // (re-derived at every use: hoisted, the slots' indices and LDS addresses spill)
				if ((st >> 8) == I && (st & 0xFF) < I - 1) {                // (tile (I - 1, I) is the factor wave's)
					const int J = st & 0xFF, cj = (J * 16 + I) * 16;
#pragma unroll
					for (int r = 0; r < 4; ++r) {
						const double p = ldlm_row_sum16(t[s][r] * xb);
						if (lc == 0) contrib[cj + 4 * r + lr] = p;
					}
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
					if (lane == 0) atomicAdd(&s_ccnt[J], 1);
				}
			}
		}
		__syncthreads();
	}
	for (int i = tid; i < n; i += NT) x_out[i] = xs[i];
	__syncthreads();
	K2_T(5);
	K2_FLUSH();
#ifdef SSK_LM_STAMPS
	if (threadIdx.x == 0 && blockIdx.x == 0) g_k2_t[15] += 1;
#endif
	return true;
}

// The LEADING columns of a system beyond 255 unknowns on the matrix cores (round 5, second step: ldl_leading_panels -- block columns through LDS on the
// vector ALUs -- cost 3 us per column, 2/3 of a 372-unknown solve).  Left-looking by panels of 16 columns: the panel's rows c0 .. n (the border row last)
// as accumulator tiles of the same transposed layout as above, tile i (rows c0 + 16 i ..) in slot i / 7 of tile wave i % 7; every EARLIER panel's rows
// of these columns and of the tiles' rows come from the system in L2 into LDS (the whole workgroup, the next panel's loads in flight under this one's
// MFMAs) and update every tile with four MFMAs; the factor wave factorises the diagonal tile (the chain of ldl_solve_mfma_tiles) and publishes L'^-1;
// the other tiles take U^T = L'^-1 A^T (four MFMAs) and go back to the system's lower triangle with the pivots' reciprocals in inv -- what
// ldl_solve_mfma_tiles' leading chunks and ldl_backsub_leading read.  n1: a multiple of 16.  Returns 0, or 1 for a non-positive pivot.
#ifdef LDL_LEAD_STAMPS
// diagnostics (tools/ldl_bench.hip -DLDL_LEAD_STAMPS): clock sums of thread 0 -- 0 tile load, 1 earlier panels' updates, 2 diagonal tile (factor wave), 3 TRSM + stores; 4 leading total, 5 trailing solve, 6 leading back-substitution
__device__ unsigned long long g_lead_t[16];   // (8 .. 11: inside a chunk step -- wait at the first barrier, commit + next fetch issued, wait at the second barrier, MFMAs)
#define LDLL_T0() unsigned long long ll_t_ = __builtin_amdgcn_s_memtime()
#define LDLL_T(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0 && blockIdx.x == 0) g_lead_t[i] += t_ - ll_t_; ll_t_ = t_; } while (0)
#else
#define LDLL_T0() do {} while (0)
#define LDLL_T(i) do {} while (0)
#endif
constexpr int LDLL_TS = 7;              // tile slots of a wave: ceil(49 / 7) -- 769 rows
constexpr int LDLL_SCAP = LDLM_DOUBLES - 2 * 16 * LDLM_LP;      // the staged rows' share of the solver's LDS
// pitch of staged rows, = 17 (mod 32) doubles: the transposing writes (a lane per column) and the MFMA operand reads (lane l: row 4 q + l / 16,
// column c + l % 16) both spread over the banks
__device__ __forceinline__ int ldll_pitch(int rows) { return rows <= 17 ? 17 : (((rows - 17 + 31) >> 5) << 5) + 17; }
__device__ __forceinline__ int ldl_leading_mfma(double* __restrict__ A, int np, int n, int n1, double* inv, double* lds)
{
	static_assert(NT == 512, "ldl_leading_mfma: 8 wavefronts");
	const int tid = threadIdx.x, lane = tid & 63, lc = lane & 15, lr = lane >> 4;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	__shared__ int s_lbad;
	double* Lik = lds;                                 // L'^-1 of the panel's diagonal tile
	double* dgs = Lik + 16 * LDLM_LP;                  // the diagonal tile on its way to the factor wave
	double* S = dgs + 16 * LDLM_LP;                    // [16 or 32][pitch]: earlier columns of rows c0 ..
	__syncthreads();                                   // (a caller that solves again after a failed pivot: every reader of the last call's flag is past it)
	if (tid == 0) s_lbad = 0;
	// the panel's tiles (the system's upper triangle holds the same values along rows: coalesced; the border is stored as a row only).  Every load is
	// unconditional, from a clamped address, and the masks come afterwards: behind per-slot conditions the compiler waited for every pair of loads before
	// the next (measured: 6.6 us per panel for 28 loads)
	auto load_tiles = [&](ldlm_v4d* tt, int c0) {
		const int Tn = (n + 1 - c0 + 15) >> 4;
		if (wave < LDLM_TW) {
#pragma unroll
			for (int s = 0; s < LDLL_TS; ++s) {
				const int i = min(LDLM_TW * s + wave, Tn - 1);
#pragma unroll
				for (int r = 0; r < 4; ++r) {
					const int a = c0 + 4 * r + lr, b = c0 + 16 * i + lc;         // element (row b, column a)
					tt[s][r] = A[b >= n ? (size_t)n * np + a : (size_t)a * np + b];
				}
			}
#pragma unroll
			for (int s = 0; s < LDLL_TS; ++s) {
				const int i = LDLM_TW * s + wave;
#pragma unroll
				for (int r = 0; r < 4; ++r) tt[s][r] = i < Tn && c0 + 16 * i + lc <= n ? tt[s][r] : 0.0;
			}
		}
	};
	for (int c0 = 0; c0 < n1; c0 += 16) {
		const int R = n + 1 - c0, Tn = (R + 15) >> 4, UP = ldll_pitch(16 * Tn);
		LDLL_T0();
		ldlm_v4d t[LDLL_TS];
		load_tiles(t, c0);                             // (in flight beside the first chunk's loads)
		LDLL_T(0);
		// ---- the earlier panels' updates: W = 32 earlier columns at a time where 32 staged columns fit, 16 otherwise (W a constant in either body).
		//      Thread: column tid % W, rows tid / W + j NT / W -- one offset register per thread and a uniform row base per load (full addresses: 34 x 2
		//      registers beside the tiles, or their arithmetic at every load: measured 3 us of instruction issue per chunk).  Nothing is zeroed: the
		//      padding rows of the last tile row hold copies of the border row (they only reach dead rows of the tiles), and a last chunk with 16 columns
		//      runs one 16-column group ----
		auto run_updates = [&](auto sh_c) {
			constexpr int SH = decltype(sh_c)::value, W = 1 << SH, RSTEP = NT >> SH, NV = SH == 5 ? 36 : 28;      // (33 / 25 loads in groups of four)
			const int rows_ = 16 * Tn, R_ = R;
			double v[NV];
			// a load = one offset add + one load from a uniform base (two instructions; the conditions are per group of four loads: per load, with
			// the row base's 64-bit arithmetic on the scalar unit, a fetch was 660 instructions a wave -- the chunk step is issue bound)
			auto fetch = [&](int q0) {
				int tq = tid;
				asm volatile("" : "+v"(tq));               // (derived at every use: hoisted out of the chunk loop the offsets spill)
				const int rq = tq >> SH, col = min(q0 + (tq & (W - 1)), c0 - 1);
				int rows = rows_, R = R_;
				asm volatile("" : "+s"(rows), "+s"(R));    // (the uniform conditions are compared at every use: hoisted, their masks spill the SGPRs)
				const char* base = reinterpret_cast<const char*>(A + (size_t)c0 * np);
				unsigned off = (unsigned)(rq * np + col) * 8u;
				const unsigned step = (unsigned)(RSTEP * np) * 8u, off_last = (unsigned)((R - 1) * np + col) * 8u;
#pragma unroll
				for (int g = 0; g < NV; g += 4) {
					if (RSTEP * g < rows) {
						if (RSTEP * (g + 4) <= R) {
#pragma unroll
							for (int u = 0; u < 4; ++u) { v[g + u] = *reinterpret_cast<const double*>(base + off); off += step; }
						} else {                           // (the border row's neighbourhood: clamped)
#pragma unroll
							for (int u = 0; u < 4; ++u) { v[g + u] = *reinterpret_cast<const double*>(base + (rq + RSTEP * (g + u) < R ? off : off_last)); off += step; }
						}
					}
				}
			};
			// (the previous panel's columns went to the system by global stores: a chunk that holds them is fetched behind a barrier that waits for
			// them -- the first barrier of a chunk step does: the loads it waits for are the ones the commit behind it needs anyway)
			if (c0 <= W) __syncthreads();
			fetch(0);
			for (int q0 = 0; q0 < c0; q0 += W) {
#ifdef LDL_LEAD_STAMPS
				unsigned long long lc_t_ = __builtin_amdgcn_s_memtime();
#define LDLL_C(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0 && blockIdx.x == 0) g_lead_t[i] += t_ - lc_t_; lc_t_ = t_; } while (0)
#else
#define LDLL_C(i) do {} while (0)
#endif
				__syncthreads();                           // (the previous chunk's reads are done; global stores and loads complete)
				LDLL_C(8);
				{
					int tq = tid;
					asm volatile("" : "+v"(tq));
					const int rq = tq >> SH;
					double* Sk = S + (tq & (W - 1)) * UP + rq;
					int rows = rows_;
					asm volatile("" : "+s"(rows));
#pragma unroll
					for (int g = 0; g < NV; g += 4) {
						if (RSTEP * g < rows) {
							if (RSTEP * (g + 4) <= rows) {
#pragma unroll
								for (int u = 0; u < 4; ++u) Sk[RSTEP * (g + u)] = v[g + u];
							} else {
#pragma unroll
								for (int u = 0; u < 4; ++u) if (rq + RSTEP * (g + u) < rows) Sk[RSTEP * (g + u)] = v[g + u];
							}
						}
					}
				}
				if (q0 + W < c0) fetch(q0 + W);
				LDLL_C(9);
				LDLM_BARRIER_LDS();
				LDLL_C(10);
				if (wave < LDLM_TW) {
					// the slots in use are the same number for every wave (a wave's last one may be beyond the panel: it works on a dead tile): branch-free
					// bodies, one per count -- all operands of a 16-column group are read first, the MFMAs of the slots alternate (behind a scalar branch
					// per slot the compiler issued read, wait, MFMA one by one: 160 cycles per MFMA)
					const int hmax = min(W, c0 - q0);
					int lro = lr, lco = lc;
					asm volatile("" : "+v"(lro), "+v"(lco));   // (the operand addresses are derived in every chunk step: hoisted they spill, and a reload from
					                                          //  scratch waits for every prefetched load in flight -- vmcnt counts both)
					ldlm_dispatch<1, LDLL_TS>((Tn + LDLM_TW - 1) / LDLM_TW, [&](auto ns_c) {
						constexpr int NS = decltype(ns_c)::value;
						// (a dead tile's operand rows lie up to 96 doubles beyond the staged ones: inside the solver's LDS, their values reach nothing --
						// one address per 4-column step, the slots at constant offsets: clamped, 28 addresses per body were hoisted and spilled)
						auto group16 = [&](int h) {
							double a[4], b[NS][4];
#pragma unroll
							for (int k4 = 0; k4 < 4; ++k4) {
								const double* Sr = S + (h + 4 * k4 + lro) * UP + lco;
								a[k4] = Sr[0] * -inv[q0 + h + 4 * k4 + lro];
#pragma unroll
								for (int s = 0; s < NS; ++s) b[s][k4] = Sr[16 * wave + 16 * LDLM_TW * s];
							}
#pragma unroll
							for (int k4 = 0; k4 < 4; ++k4) {
#pragma unroll
								for (int s = 0; s < NS; ++s) t[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[k4], b[s][k4], t[s], 0, 0, 0);
							}
						};
						// (one block for both groups of a full chunk where the operands fit: the second group's reads go under the first one's MFMAs)
						if (NS <= 4 && hmax == 32) { group16(0); group16(16); }
						else {
#pragma unroll 1
							for (int h = 0; h < hmax; h += 16) group16(h);
						}
					});
				}
				LDLL_C(11);
			}
		};
		if (c0 > 0) {
			if (32 * UP <= LDLL_SCAP) run_updates(std::integral_constant<int, 5>{});
			else run_updates(std::integral_constant<int, 4>{});
		}
		LDLL_T(1);
		// ---- the diagonal tile to the factor wave ----
		if (wave == 0) {
#pragma unroll
			for (int r = 0; r < 4; ++r) dgs[(4 * r + lr) * LDLM_LP + lc] = t[0][r];
		}
		__syncthreads();
		if (wave == LDLM_TW) {
			// lanes 0 .. 15: row lc of the tile; lanes 16 .. 31: column lc of L'^-1 (ldl_solve_mfma_tiles' chain; every column is an unknown here)
			const bool inv_lane = (lane & 48) == 16;
			bool bad = false;
			double w[16];
#pragma unroll
			for (int j = 0; j < 16; ++j) { const double xv = dgs[j * LDLM_LP + lc]; w[j] = inv_lane ? (j == lc ? 1.0 : 0.0) : xv; }
			double d = readlane_f64(w[0], 0);
			bad |= !(d > 0.0);
			double ivj = rcp_f64(d);
#pragma unroll
			for (int j = 0; j < 16; ++j) {
				const double f = w[j] * ivj;
				if (lane == 0) inv[c0 + j] = ivj;
				double iv_next = 0.0;
#define LDLL_GROUP(g) \
				_Pragma("unroll") for (int c = j + 2 + 3 * (g); c < 16 && (c < j + 5 + 3 * (g) || (g) == 4); ++c) { const double lcj = readlane_f64(f, c); w[c] -= w[j] * lcj; }
				if (j + 1 < 16) {
					const double l1 = readlane_f64(f, j + 1);
					w[j + 1] -= w[j] * l1;
					d = readlane_f64(w[j + 1], j + 1);
					bad |= !(d > 0.0);
					double r = __builtin_amdgcn_rcp(d);
					LDLL_GROUP(0); LDLM_PIN17(r, w);
					double e = fma(-d, r, 1.0);
					LDLL_GROUP(1); LDLM_PIN17(e, w);
					r = fma(e, r, r);
					LDLL_GROUP(2); LDLM_PIN17(r, w);
					e = fma(-d, r, 1.0);
					LDLL_GROUP(3); LDLM_PIN17(e, w);
					r = fma(e, r, r);
					LDLL_GROUP(4); LDLM_PIN17(r, w);
					iv_next = r;
				}
#undef LDLL_GROUP
				ivj = iv_next;
			}
			if (inv_lane) {
#pragma unroll
				for (int i = 0; i < 16; ++i) Lik[i * LDLM_LP + lc] = w[i];
			}
			if (lane < 16) {                               // the diagonal tile's U (pivots on the diagonal) back to the lower triangle
#pragma unroll
				for (int j = 0; j < 16; ++j) if (j <= lane) A[(size_t)(c0 + lane) * np + c0 + j] = w[j];
			}
			if (bad && lane == 0) s_lbad = 1;
		}
		__syncthreads();
		if (s_lbad) return 1;                              // uniform
		LDLL_T(2);
		if (wave < LDLM_TW) {
			double la[4];
#pragma unroll
			for (int k4 = 0; k4 < 4; ++k4) la[k4] = Lik[lc * LDLM_LP + 4 * k4 + lr];
#pragma unroll
			for (int s = 0; s < LDLL_TS; ++s) {
				const int i = LDLM_TW * s + wave;
				if (i >= 1 && i < Tn) {
					ldlm_v4d c4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
					for (int k4 = 0; k4 < 4; ++k4) c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(la[k4], t[s][k4], c4, 0, 0, 0);
					const int b = c0 + 16 * i + lc;
					if (b <= n) {
#pragma unroll
						for (int r = 0; r < 4; ++r) A[(size_t)b * np + c0 + 4 * r + lr] = c4[r];
					}
				}
			}
		}
		LDLL_T(3);
	}
	__syncthreads();
	return 0;
}

// Back-substitution of the n1 leading unknowns (n1 a multiple of 16) by panels of 16 from the last one up: x_out[n1 .. n) is final, x_out[0 .. n1) holds
// z_k / d_k.  Panel c0: every thread sums U(i, c0 + k) x_i over its share of the rows BELOW the panel (thread = column k x one of 32 row groups, eight
// loads in flight; the matrix is read once, 16 doubles of a row per 16 threads), the 32 partials of a column are added in a fixed order, the panel's own
// triangle by one wavefront (x of the later columns by v_readlane).  (ldl_backsub_leading staged ALL rows' leading columns through LDS and updated every
// earlier unknown after each 16-row block: 49 us at 144 leading unknowns, 131 at 320.)
__device__ __forceinline__ void ldl_backsub_leading16(const double* __restrict__ A, int np, int n, int n1, double* x_out, const double* inv, double* lds)
{
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int kc = tid & 15, rg = tid >> 4;                // column of the panel, row group (32 of them)
	double* part = lds;                                    // [32][17]
	for (int c0 = n1 - 16; c0 >= 0; c0 -= 16) {
		// the panel's own triangle, column k of lane k: U(c0 + j, c0 + k), j > k (wave 0; on their way during the sums)
		double u[16];
		if (wave == 0) {
#pragma unroll
			for (int j = 0; j < 16; ++j) u[j] = A[(size_t)(c0 + j) * np + c0 + (lane & 15)];
		}
		const double* col = A + (size_t)(c0 + 16) * np + c0 + kc;
		const double* xb = x_out + c0 + 16;
		const int rows = n - c0 - 16;                      // rows below the panel
		double acc[4] = {0.0, 0.0, 0.0, 0.0};
		for (int r0 = rg; r0 < rows; r0 += 8 * 32) {
			double a[8];
#pragma unroll
			for (int q = 0; q < 8; ++q) a[q] = col[(size_t)min(r0 + 32 * q, rows - 1) * np];
#pragma unroll
			for (int q = 0; q < 8; ++q) acc[q & 3] += r0 + 32 * q < rows ? a[q] * xb[min(r0 + 32 * q, rows - 1)] : 0.0;
		}
		part[rg * 17 + kc] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
		__syncthreads();
		if (wave == 0) {
			const int k = lane & 15;
			double sp[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
			for (int r = 0; r < 32; ++r) sp[r & 3] += part[r * 17 + k];
			const double ivk = inv[c0 + k];
			double xv = x_out[c0 + k] - ((sp[0] + sp[1]) + (sp[2] + sp[3])) * ivk;
#pragma unroll
			for (int j = 15; j >= 1; --j) {
				const double xj = readlane_f64(xv, j);
				xv -= (j > k ? u[j] * ivk : 0.0) * xj;
			}
			if (lane < 16) x_out[c0 + lane] = xv;
		}
		__syncthreads();
	}
}

// Systems beyond 255 unknowns (round 5: the local areas the facade's own stream grows -- 61 optimised edges = 366 unknowns in the bench's 401-frame pass,
// up to 94 at KITTI-00 length -- ran the register-tile solver): the leading 16 ceil((n - 255) / 16) columns by ldl_leading_mfma above (U and 1 / d to the
// system in L2), the trailing 240 .. 255 unknowns by the tile solver (its tiles take the leading columns' updates as MFMAs right after the load), the leading
// unknowns by ldl_backsub_leading.  Same factorisation, the update terms of an element in column order.  A is overwritten (as by ldl_solve_regtile).
// Measured (tools/ldl_bench.hip, net of the restore copy): 396 unknowns 380 us (ldl_leading_panels for the leading columns: 551; register tiles: 600),
// 564: 998 (1 735), 768: 2 320 (4 690).
constexpr int LDLM_LEAD_NMAX = 768;     // 128 optimised edges: every window the node builder takes (the full-length stream's local areas: median 38 edges,
                                        // 90th percentile 68, largest 94 -- profiles/r05_stream_stage2_windows.json)
__device__ __forceinline__ bool ldl_solve_mfma_lead(double* __restrict__ A, int np, int n, double* x_out, double* inv, double* lds, int lds_doubles)
{
	const int tid = threadIdx.x;
#ifndef SSK_K2_LEAD_MFMA
#define SSK_K2_LEAD_MFMA 1               // 1: the leading columns on the matrix cores (ldl_leading_mfma), 0: ldl_leading_panels (same-box A/B: tools/ldl_bench.hip)
#endif
	LDLL_T0();
#if SSK_K2_LEAD_MFMA
	const int n1 = n <= LDLM_NMAX ? 0 : 16 * ((n - LDLM_NMAX + 15) >> 4), m = n - n1;      // (240 .. 255 trailing unknowns)
	if (n1 > 0 && ldl_leading_mfma(A, np, n, n1, inv, lds) != 0) return false;
#else
	const int m = n <= LDLM_NMAX ? n : 252, n1 = n - m;
	if (n1 > 0) {
		const int lead = ldl_leading_panels<12>(A, np, n, n1, inv, lds, lds_doubles);
		if (lead == 1) return false;
		if (lead == 2) return SSK_LDL_BIG(A, np, n, x_out, inv, lds, lds_doubles);      // (a block column does not fit the staging: not at these sizes)
	}
#endif
	LDLL_T(4);
	if (!ldl_solve_mfma_tiles(A + (size_t)n1 * np + n1, np, m, x_out + n1, lds, n1, A, inv)) return false;
	LDLL_T(5);
	if (n1 > 0) {
		for (int k = tid; k < n1; k += NT) x_out[k] = A[(size_t)n * np + k] * inv[k];
		__syncthreads();
#if SSK_K2_LEAD_MFMA
		ldl_backsub_leading16(A, np, n, n1, x_out, inv, lds);
		LDLL_T(6);
		return true;
#else
		const bool ok_ = ldl_backsub_leading(A, np, n, n1, x_out, inv, lds, lds_doubles);
		LDLL_T(6);
		return ok_;
#endif
	}
	return true;
}
