// ssk_lm_node.h -- data model of the NODE form of the relative-BA Levenberg-Marquardt solve (kernel: ssk_lm_node.hip, host
// builder: ssk_lm_node_build.h).  Replaces, for windows whose reduced system fits LDS, the observation-level tiles of
// ssk_lm_tiled.hip (reference: rba.define_new_keyframe, EST.cpp:782-784; solver type SRBA.h:34).
//
// The algebra.  An observation o of landmark l reaches l's base keyframe over a path of k2k edges; the path's SUFFIXES live in a
// trie (id 0 = empty, suffix = (first step, shorter suffix)) with one composed transform T_s = [R_s | t_s] each.  For an optimised
// path entry i of o (suffix s_i starting at that step) the Jacobian block is J_i = +-Jf R^T [I | -[q]x] with (R, q = R f + t) taken
// from the transform suffix of the entry (s_i for a forward step, its rest for an inverse one) -- the same for EVERY observation of
// l whose path contains s_i.  So everything the reduced camera system needs aggregates per NODE n = (landmark, suffix s_i):
//   Vs_n = sum of V_o = Jf^T Jf over the observations whose path contains the suffix, gs_n = sum of Jf^T w r,
//   V_l + lambda I = L L^T,  Z_n = Vs_n L^-T,  c = L^-1 g_l,
//   reduced-system block of an ordered node pair (r, c) of one landmark:  [I; [q_r]x] D [I | -[q_c]x]  with the 3x3
//     D = Zh_r Ph_c^T  (the suffix of c is a suffix of r's: "nested", r far)      Zh = s R Z,  Ph = s R (L - Z),  s = +-1
//     D = Ph_r Zh_c^T  (nested, c far)
//     D = -Zh_r Zh_c^T (not nested: Schur term only)
//   (H_pp and the Schur complement in ONE item: Vs_far = Z_far L^T, so Vs_far - Z_r Z_c^T = Z_far (L - Z_near)^T),
//   reduced gradient of an edge: sum over its nodes of [vh; q x vh], vh = s R (gs - Z c),
//   back-substitution: delta_f = L^-T (c - sum_n Zh_n^T (dt_e + dw_e x q_n)).
//   A landmark the window does not optimise (and every landmark in the pass that only collects diag(H_pp) for lambda_0) uses
//   Z = Vs, L - Z := I, c = 0 and has no "not nested" items.  Verified against the dense textbook form to 1e-15
//   (tests/test_lm_node_form.py).
//
// The mapping.  A workgroup of NDW wavefronts owns a window for the whole LM loop; several workgroups share a CU (80 KB of LDS each
// at the bench windows), so one window's dense 60x60 LDL^T -- a latency chain -- runs beside another window's linearisation.
//   GROUP  = landmarks with <= 64 observations and <= 64 lanes (one ROOT lane per landmark + one lane per node) in all: one
//            wavefront linearises a group with no workgroup barrier: lane = observation (V_o, g_o -> wave-private LDS staging),
//            lane = root / node (sums over a contiguous run of the staging: the landmark's observations are sorted so that every
//            trie subtree is a run), roots factorise, nodes rotate and leave a 24-double record (Zh, Ph, q, vh) in LDS.
//   TILE   = NDW groups (one per wavefront), then one barrier, then the PAIR phase: the tile's items sorted by 6x6 block; a QUAD
//            of lanes owns a block (its items dealt out among the 4 lanes, 36 sums in registers, a two-step DPP reduce-scatter,
//            then each lane adds 9 sums to the block's accumulator in LDS -- one owner per block and tile: no atomics, fixed
//            order, bit-reproducible).
//   After the tiles the accumulators are copied into the dense bordered matrix over the dead records and solved by
//   ldl_solve_lds (ssk_lm_device.h); the trial state is scored group by group, the landmark step computed on the way
//   (node lanes -> root lane -> the landmark's observation lanes, all inside one wavefront).
#pragma once
#include <cstdint>

constexpr int SSK_ND_REC = 26;          // doubles per node record in LDS (24 used; 26: a 16-byte aligned pitch whose b128 gathers spread over all banks)
constexpr int SSK_ND_STAGE = 9;         // doubles per observation in the wave-private staging
constexpr int SSK_ND_LREC = 17;         // doubles per root record in the staging: L (6), L^-1 (6), c (3), fixed flag (17: an odd pitch, no bank conflicts between roots)
constexpr int SSK_ND_SCR = 12;          // doubles per lane in the window's HBM scratch: node (Zh 9, q 3) / root (L^-1 6, c 3)
constexpr int SSK_ND_MAX_WAVES = 8;
#ifndef SSK_ND_LANE_RUNS
#define SSK_ND_LANE_RUNS 0              // pair phase: 0 = a quad of lanes per block, DPP sums, one ds_add_f64 per component and quad (default);
#endif                                  // 1 = a LANE owns a run of <= 8 items of one block and adds its 36 sums itself (perfectly balanced turns, but 256 lanes x 36
                                        // conflicting LDS atomics per tile: measured 5.41 ms against 4.18 ms per 1024 bench windows, same box)

// plain mirrors of the HIP vector types (the host builder is HIP free: tools/lm_node_emu.cpp compiles it with g++)
struct alignas(8) nd_i2 { int x, y; };
struct alignas(16) nd_i4 { int x, y, z, w; };
struct alignas(16) nd_f4 { float x, y, z, w; };

struct nd_win {
	int n_edges, E, n_lms, L, n_obs, n_suf;
	int edge0, lm0, us0, ls0;           // bases into the batch-wide pose / landmark / slot_edge / slot_lm families (shared with the other kernels)
	int suf0;                           // nd_suf[suf0 + id] = (first step code, id of the rest)
	int n_tiles, tile0;                 // nd_tile[tile0 ..)
	int n_groups;
	int grp0;                           // nd_group[grp0 ..)
	int obs0;                           // nd_obs / nd_px [obs0 + group.obs0 + lane]
	int lane0;                          // nd_lane[lane0 + group.lane0 + lane]; scratch nd_scr[SSK_ND_SCR * (lane0 + ...)]
	int item0, qs0, gitem0, gqs0;       // nd_item / nd_qs / nd_gitem / nd_gqs bases
	int trace0;
	int n_rounds, max_parts;            // (diagnostics: most rounds of quad slots of a tile, parts of a block)
	long long tab0;                     // nd_tab[tab0 ..): two suffix tables (current / trial state), 12 doubles per suffix each
	long long dp_trace0, df_trace0;
	long long hb0, h0;                  // BIG form (reduced system larger than LDS): block accumulators nd_dev::hb[hb0 ..) (36 doubles per block pair) and
	                                    // the dense bordered system lm_dev::Hred[h0 ..), both in HBM / L2
};

struct nd_tile {
	int grp0, n_grp;                    // groups of the tile (window relative), n_grp <= waves per workgroup
	int qs0, n_rounds;                  // block quad slots: nd_qs[W.qs0 + qs0 + round * NQ + quad]
	int gqs0, n_gqs;                    // gradient quad slots: nd_gqs[W.gqs0 + gqs0 + k]
};

struct nd_group {
	int obs0, n_obs;                    // observation lanes (window relative)
	int lane0, n_lanes;                 // root / node lanes (window relative)
};

// lane record (int4):
//   x  landmark (window index)
//   y  first observation lane | number of observation lanes << 8 | the run starts at the landmark's first observation << 16   (the run of the staging this lane sums)
//   z  root: optimised slot of the landmark or -1;   node: transform suffix | edge slot << 16 | inverse step << 24
//   w  kind (0 idle, 1 root, 2 node) | lane of the landmark's root << 8 | root: number of node lanes that follow << 16
// observation record (int2): x landmark, y suffix of the whole path | lane of the landmark's root << 16 | distance from the landmark's first observation lane << 24
// item (int): record X: slot | P-part << 9;  record Y: slot << 10 | P-part << 19;  bit 20: not nested (D = -Zh Zh^T)
// quad slot (int2): x block pair id (0xFFFF: none) | items << 16, y first item (window relative) | part of the block << 28 |
//   block cut in two << 30 | this wavefront holds such blocks << 31

// phase-split form (ssk_lm_split.hip): per window, beside its nd_win (whose tile[].qs0 / n_rounds then mean: first (tile, block) slot of the
// tile in sqs[], number of such slots)
struct nd_split {
	int sqs0;                           // sp_dev::sqs[sqs0 + tile.qs0 + k], a LANE slot: x block pair id | several-lane blocks in this wavefront << 29 | diagonal block << 30,
	                                    // y items of the block | log2(lanes of the block) << 16 | this lane's rank << 20, z first item (window relative), w partial slot
	int poff0, goff0, gslot0;           // sp_dev::poff[poff0 + pid] .. [pid + 1]: the block's partial slots; goff[goff0 + e]: the edge's gradient slots; gslot[gslot0 + tile.gqs0 + k]
	int n_part, n_gpart;
	int gtile0;                         // the window's first tile in the batch-wide tile numbering (per-tile error / vmax partials)
	int pad;
	long long part0, gpart0;            // Hpart[36 * (part0 + slot)], gpart[6 * (gpart0 + slot)]
};

struct nd_dev {
	const nd_win* win; const nd_tile* tile; const nd_group* group;
	const nd_i2* suf; const nd_i2* obs; const nd_f4* px; const nd_i4* lane;
	const int* item; const nd_i2* qs; const int* gitem; const nd_i2* gqs;
	double* tab; double* scr; double* hb;
	const int* order;                   // launch order: workgroup b solves window order[b] (largest first: the last workgroups to start are the short ones)
};
