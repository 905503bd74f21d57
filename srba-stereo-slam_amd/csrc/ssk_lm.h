// ssk_lm.h -- device data model of a batch of relative-BA windows (see ssk_lm.hip for the kernel).
#pragma once
#include "ssk_internal.h"
#include "ssk_lm_node.h"

constexpr int SSK_LM_MAX_PATH = 8;     // spanning-tree path length limit (reference max_tree_depth = 4, KITTI.ini:104)
constexpr int SSK_LM_MAX_N6 = 768;     // reduced system limit: 6E <= 768 (E <= 128 optimised edges per window)
constexpr int SSK_LM_LDS_EDGES = 128;  // edge poses cached in LDS up to this many edges per window

// Per-window header.  Every "...0" member is the window's base offset into a batch-wide array family; values
// stored inside the families (offset tables, indices) are window relative.
//
// Symbolic structure (built once on the host, ssk_capi_lm.hip):
//   lme entry m  = (landmark, optimised edge) incidence: the landmark has an observation whose path crosses
//                  the edge.  Entries are grouped by landmark: optimised landmarks first, in slot order
//                  (lsm_off[slot] .. lsm_off[slot+1]), then fixed landmarks.  lmep lists the path entries.
//   q entry      = (landmark, pair of its lme entries m_i <= m_j by edge slot): one 6x6 block contribution to
//                  the reduced system.  q entries are sorted by block pair id (pid-major), so the reduction of
//                  block pid is a contiguous range q_off[pid] .. q_off[pid+1] of the 36-double slabs.
//                  qo lists the (path entry of e_i, path entry of e_j) of the observations containing both.
struct lm_win {
	int n_edges, E, n_lms, L, n_obs, n_path, n_pairs, n_lme, n_lme_unknown, n_q, n_pchunk, n_schunk;
	int edge0;     // edges:        edge_slot[], pose*[] x12
	int lm0;       // landmarks:    lm_slot[], lm*[] x3
	int obs0;      // observations: obs_lm[], obs_px x4, obs_r x4, obs_Jf x12
	int opo0;      // obs_path_off[] (n_obs+1 entries per window)
	int path0;     // path entries: path_edge[], path_obs[], path_q x3, path_Je x24
	int us0;       // optimised-edge slots: slot_edge[], dp x6
	int uso0;      // per optimised edge offset table (E+1): el_off[]
	int ls0;       // optimised-landmark slots: slot_lm[], Hff x9, Vinv x9, gf x3, df x3
	int lso0;      // per optimised landmark offset tables (L+1): lmo_off[], lsm_off[]
	int lmo0;      // lmo_idx[]  (observations of an optimised landmark)
	int lme0;      // lme entries: lme_edge[], lme_ls[], lme_elpos[], Wb x18, Tb x18, gpe x6; tiled kernel: inc_rec[] =
	               // (tile-local landmark | optimised << 15 | #path entries << 16, first lmep entry, edge-major slot, first lmep_loc)
	int lmeo0;     // lmep_off[] (n_lme+1)
	int lmep0;     // lmep_idx[] path entries of an lme entry
	int pair0;     // block pairs e<=e': pair_ee[]
	int pairo0;    // q_off[], q_uend[] (n_pairs+1): slabs of a block; those of optimised landmarks come first
	int q0;        // q entries: q_mm[]
	int qo0;       // qo_off[] (n_q+1)
	int qe0;       // qo[] entries
	int el0;       // el_idx[] lme entries per optimised edge (edge-major order); gre x6 in that order
	int qp0;       // qp_q[]: q entries that have co-observations, sorted by block
	int pc0;       // H_pp chunks: pchunk[] (block, first qp item, count <= 64), Ppart x36; pc_off[] lives at pairo0
	int sc0;       // Schur chunks: schunk[] (block, first q, count <= 64), Spart x36; sc_off[] lives at pairo0
	int trace0;    // trace entries (window * max_iters)
	// ---- LDS-tiled kernel (ssk_lm_tiled.hip); n_tiles < 0: the window does not fit that kernel ----
	int n_tiles, tile0;   // tile[]
	int tl0;              // tile landmarks (tile order): tl_rec[] = (optimised slot or -1, first tile-local observation, end, 0)
	int tobs0;            // packed observation records tobs_a[], tobs_px[]: tile order, then the observations outside every tile (n_obs in all)
	int pit0, pg0, pcx0, ps0, pcl0;   // H_pp items pit[], lane groups pgrp[], chunk slot bases pslot0[], partial slots Ppart2 x36, pcl_idx[]
	int qit0, sg0, scx0, ss0, scl0;   // Schur items qit[], lane groups sgrp[], sslot0[], Spart2 x36, scl_idx[]
	// ---- on-chip accumulation (lm_solve_acc_kernel): acc != 0 when the window's accumulators fit LDS (E <= SSK_LMA_MAX_E) ----
	// ---- path-suffix table (tiled kernels): the distinct suffixes (c_k, ..., c_{n-1}) of the observations' spanning-tree paths, id 0 =
	// the empty one, ids ascending by length; suf_rec[suf0 + id] = (first step code, id of the rest); suf_lvl[k] = first id of length k
	int n_suf, suf0, tent0;            // tent[tent0 + ..]: per tile entry (suffix id << 1 | direction), tile entry order
	int tab_lds;                       // 1: the whole table lives in LDS for the pass and records hold global suffix ids; 0 (many suffixes): the table
	                                   // is built in the window's L2-resident scratch, every tile copies the suffixes it uses (tsuf[]) into LDS and its
	                                   // records hold tile-local ids
	int tsuf0;                         // tsuf[tsuf0 + ..]: per tile the global ids of the suffixes it uses
	int schur_mfma;                    // slot kernels: 1 = Schur complement on the matrix cores (Y panels; per (tile, 16x16 tile of the reduced
	                                   // system) a 256-double partial in HBM: Stile[], summed in tile order), 0 = sparse 6x6 items
	int il0;                           // ilane[il0 + ..]: incidence lanes of the tiles: (local incidence | sub-lane << 16 | lanes of the incidence << 20, first lmep entry, entries, first entry location)
	int mt0, st0, sto0, pad_mt;        // mt[mt0 + ..]: (I | J << 8, partial id) per task; Stile[256 * (st0 + id)]; st_off[sto0 + u]: partials of tile u
	long long gtab0;                   // suf_tab[gtab0 + ..]: 12 doubles per suffix (x workgroups of the cooperative kernel)
	int suf_lvl[SSK_LM_MAX_PATH + 2];
	int acc;              // 0: partial slots in HBM (kernels above); 1: H_pp blocks, Schur tiles and the reduced gradient accumulate in LDS
	int wt0, wto0;        // per-wave task lists of the tiles: entries wt[] (base wt0), per tile NW + 1 offsets into them wt_off[] (base wto0)
	int pad_acc;
	long long h0;  // Hred (6E)^2 doubles (only used when the matrix does not fit LDS); Hpair 36 x n_pairs
	long long dp_trace0, df_trace0;
};

// One tile = a run of the window's landmarks (tile order: optimised landmarks in slot order, then the fixed ones that
// touch an optimised edge) whose linearisation fits LDS: 15 doubles per optimised path entry (A = w M R 4x3, point q),
// 17 per observation (Jf 4x3, w r), 19 per (landmark, edge) incidence (Y = W L^-T 6x3), 9 per landmark (L^-1, c).
// Reductions over the items of a tile: every lane owns a GROUP of <= R consecutive items of ONE block (items are
// sorted by block pair id), 64 groups form a chunk (one wave pass), a chunk's segments (runs of equal block id,
// also cut every 16 lanes) are summed in lane order into 36-double partial slots: segment s of the chunk goes to slot
// pcl_idx[pslot0[chunk] + s]; slot ids are block-major, block pid owns slots pcl_off[pid] .. pcl_off[pid+1].
struct lm_tile {
	int lml0, n_lm;       // tile landmarks tl_*[W.tl0 + lml0 ..)
	int obs0, n_obs;      // tile observations tobs_*[W.tobs0 + obs0 ..)
	int n_ent;            // optimised path entries of those observations (local index = first entry of the obs + rank)
	int inc0, n_inc;      // incidences = lme entries W.lme0 + inc0 ..   (n_inc + n_lm <= workgroup size)
	int pg0, n_pg, pc0, n_pc;   // H_pp lane groups pgrp[W.pg0 + pg0 ..), chunks pslot0[W.pcx0 + pc0 ..)
	int sg0, n_sg, sc0, n_sc;   // Schur lane groups / chunks
	int n_pan;            // acc: Y panels of the tile (a panel = one 16-row tile of the reduced system x one landmark, [3][16] doubles)
	int wt0, ent0, pad1;  // acc: first entry of the tile's wt_off table (window relative); ent0: first tile entry in tent[] (window relative)
	int tsuf0, n_tsuf;    // !tab_lds: the tile's suffixes tsuf[W.tsuf0 + tsuf0 ..) (tile-local id = position)
	int mt0, n_mt;        // slot kernels, matrix-core Schur: the tile's tasks mt[W.mt0 + mt0 ..)
	int il0, n_il, slot0, n_slot;   // slot0 / n_slot: the tile's optimised landmarks are the slots slot0 .. slot0 + n_slot - 1 (tile order = slot order); incidence LANES of the tile: ilane[W.il0 + il0 ..) -- an incidence with many path entries is shared by up to 8 lanes
};

struct lm_dev {
	const lm_win* win;
	const lm_tile* tile; const int4* tl_rec; const int4* tobs_a; const float4* tobs_px;
	const int4* inc_rec; const int* lmep_loc;
	const int* pit; const int2* pgrp; const int* pslot0; const int* pcl_off; const int* pcl_idx;
	const int* qit; const int2* sgrp; const int* sslot0; const int* scl_off; const int* scl_idx;
	const int2* suf_rec; const int* tent; const int* tsuf; double* suf_tab;
	const int2* mt; const int* st_off; double* Stile;
	const int4* ilane;
	const int* wt_off; const int* wt; const int2* tl_pan;   // acc kernel (tl_pan: per tile landmark (row-tile mask, first panel of the tile's panel array))
	int ldl_mfma;         // reduced systems in L2 scratch: 1 = 16-column LDL^T panels with the trailing update on the FP64 matrix cores (default), 0 = 6-column VALU solver (SSK_LM_LDL=valu)
	int store_hff;        // 1: the landmark blocks H_ff / g_f of the last linearisation go to HBM (shared-landmark exchange); 0: they stay on chip
	double* Ppart2; double* Spart2;
	const int* edge_slot; double* pose; const double* pose0; double* pose_new;
	const int* lm_slot; double* lm; const double* lm0; double* lm_new;
	const int* obs_lm; const float* obs_px; const int* obs_path_off;
	const int* path_edge; const int* path_obs;
	const int* slot_edge; const int* slot_lm;
	const int* lmo_off; const int* lmo_idx; const int* lsm_off;
	const int* lme_edge; const int* lme_ls; const int* lme_elpos; const int* lmep_off; const int* lmep_idx;
	const int2* pair_ee; const int* q_off; const int* q_uend; const int2* q_mm; const int* qo_off; const int2* qo;
	const int* el_off; const int* el_idx;
	const int* qp_q; const int3* pchunk; const int* pc_off; const int3* schunk; const int* sc_off;
	double* obs_r; double* obs_Jf; double* path_q; double* path_Je;
	double* Hff; double* Vinv; double* gf; double* df; double* Wb; double* Tb; double* gpe; double* gre;
	double* Ppart; double* Spart; double* Hpair; double* Hred; double* dp;
	ssk_lm_result* res; ssk_lm_trace_entry* trace; double* dp_trace; double* df_trace;
	int trace_on;
	int* coop_i;      // cooperative kernel, per window: [barrier counter, abort flag, solve ok, -]
	double* coop_f;   // cooperative kernel, per window: 2 banks x 64 partial values
};

struct lm_cam {
	double fx_l, fy_l, cx_l, cy_l, fx_r, fy_r, cx_r, cy_r, baseline;
	double S[12];
	int max_iters, robust;
	double kernel, tau, max_lambda, stop_rmse, conv_rel;
};

int ssk_lm_launch(ssk_ctx* ctx, const lm_dev& D, const lm_cam& P, int n_windows, int max_n6);

// LDS-tiled kernel: layout constants shared by the host tiler (ssk_capi_lm.hip) and the kernel
#ifndef SSK_LMT_THREADS
#define SSK_LMT_THREADS 512
#endif
#ifndef SSK_LMT_WG_PER_CU
#define SSK_LMT_WG_PER_CU 1
#endif
constexpr int SSK_LMT_NT = SSK_LMT_THREADS;        // workgroup size
// record pitches (doubles, odd: conflict-free).  Entry: chained point q (3) + packed (suffix id << 1 | direction, tile-local observation);
// observation: V_o = Jf^T Jf (6) + g_o = Jf^T w r (3) -- the 4x3 blocks A of an optimised path entry are NOT stored: A = +-Jf R_suffix^T
// with R_suffix from the window's path-suffix table (12 doubles per distinct suffix, rebuilt in LDS at every pass)
#ifndef SSK_LMT_INC_CAP_V
#define SSK_LMT_INC_CAP_V 3
#endif
constexpr int SSK_LMT_INC_CAP = SSK_LMT_INC_CAP_V;  // path entries per incidence lane (an incidence with more is split, at most 8 lanes)
constexpr int SSK_LMT_ENT = 5, SSK_LMT_OBS = 9, SSK_LMT_Y = 19, SSK_LMT_LMR = 9;
constexpr int SSK_LMT_LDS_TOTAL = 160 * 1024 / SSK_LMT_WG_PER_CU;   // gfx950: 160 KB of LDS per CU, shared by the resident workgroups
constexpr int SSK_LMT_STATIC = 1024;               // reserve for the kernel's static LDS (a few scalars and the wave partials)
int ssk_lm_launch_tiled(ssk_ctx* ctx, const lm_dev& D, const lm_cam& P, int n_windows, size_t dyn_lds_bytes, int schur_mfma, bool lds_sys);
// G workgroups per window (n_windows * G <= number of CUs, G <= 64): single large windows of the un-batched drop-in path
int ssk_lm_launch_coop(ssk_ctx* ctx, const lm_dev& D, const lm_cam& P, int n_windows, int G, size_t dyn_lds_bytes, bool test_timeout, int schur_mfma);
int ssk_lm_coop_blocks_per_cu(ssk_ctx* ctx, size_t dyn_lds_bytes, int schur_mfma);
// on-chip accumulation kernel: layout shared by the host tiler and the kernel
constexpr int SSK_LMA_MAX_E = 12;                  // optimised edges up to which the accumulators live in LDS
constexpr int SSK_LMA_PANEL = 48;                  // doubles per Y panel: [3 columns][16 rows]
inline int ssk_lma_row_tiles(int E) { return (6 * E + 15) / 16; }
// LDS doubles of the accumulators: H_pp blocks (36 per block pair), Schur tiles (256 per upper 16x16 tile), gradient (6E, even)
inline int ssk_lma_acc_doubles(int E) { const int T = ssk_lma_row_tiles(E); return 36 * (E * (E + 1) / 2) + 256 * (T * (T + 1) / 2) + ((6 * E + 1) & ~1); }
int ssk_lm_launch_acc(ssk_ctx* ctx, const lm_dev& D, const lm_cam& P, int n_windows, size_t dyn_lds_bytes, int schur_mfma);
// node form (ssk_lm_node.hip): one 256-thread workgroup per window, dynamic LDS sized for the largest window of the batch
int ssk_lm_launch_node(ssk_ctx* ctx, const lm_dev& D, const nd_dev& N, const lm_cam& P, int n_windows, size_t dyn_lds_bytes, bool big, bool small_e);
// pose-only windows (ssk_lm_pose.hip): E in {1, 2}, no optimised landmark; d_order: the batch's window numbers of that family
int ssk_lm_launch_pose_only(ssk_ctx* ctx, const lm_dev& D, const int* d_order, const lm_cam& P, int n_windows, int n_two_edges, size_t dyn_lds_bytes);
int ssk_lm_launch_reset(ssk_ctx* ctx, double* pose, const double* pose0, size_t n_pose, double* lm, const double* lm0, size_t n_lm);
