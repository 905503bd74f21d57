// ssk_lm_node_build.h -- host side of the NODE form (ssk_lm_node.h): the symbolic structure of one window -- suffix trie, nodes,
// landmark groups, tiles, block-sorted items -- as SRBA does its graph bookkeeping on the CPU.  HIP free (plain C++17): the
// library (ssk_capi_lm.hip) and the CPU emulation of the kernel's data flow (tools/lm_node_emu.cpp, tests/test_lm_node_form.py)
// compile the same code.
#pragma once
#include <algorithm>
#include <cstdint>
#include <unordered_map>
#include <vector>
#include "../../include/ssk_capi.h"
#include "ssk_lm_node.h"

struct nd_host {
	std::vector<nd_win> win;
	std::vector<nd_tile> tile;
	std::vector<nd_group> group;
	std::vector<nd_i2> suf, obs, qs, gqs;
	std::vector<nd_f4> px;
	std::vector<nd_i4> lane;
	std::vector<int> item, gitem;
	long long tab_doubles = 0;          // suffix tables of all windows
	size_t lds_doubles = 0;             // largest dynamic LDS need of a window (doubles)
	size_t lds_doubles_big = 0;         // ... in the BIG form (accumulators and system in HBM / L2)
	long long hb_doubles = 0;           // BIG form: block accumulators of all windows
	int max_items_per_lane = 0;         // (diagnostic: the longest item run of a quad lane)
	// ---- phase-split form (ssk_lm_split.hip): one nd_split per window beside its nd_win ----
	std::vector<nd_split> swin;
	std::vector<nd_i4> sqs;             // (tile, block) slots: x block pair id, y items, z first item (window relative), w partial slot (window relative)
	std::vector<int> poff, goff, gslot; // poff: per window n_pairs + 1 (partial slots of a block, block-major); goff: E + 1; gslot: per gqs entry its gradient partial slot
	long long part_slots = 0, gpart_slots = 0;   // 36-double / 6-double partial slots of all windows
	int max_tile_blocks = 0;
};

// dynamic LDS of a window, doubles: edge poses twice, solution + pivots, block accumulators, gradient accumulator, the trie records, then the region the
// node records of a tile (waves x 64 x SSK_ND_REC) share with the dense bordered system ((6E + 1) x (6E | 1))
// BIG form (big = true): accumulators and system live in HBM / L2; the region holds the node records, later whatever the hybrid solver
// (ldl_solve_hybrid) can use of the packed trailing triangle -- up to `limit` doubles in all.
inline size_t nd_lds_doubles(int n_edges, int E, int waves, int n_suf, int n_tiles = 0, int n_groups = 0, bool big = false, size_t limit = 0)
{
	const size_t n6 = 6 * (size_t)E, n6p = (n6 + 1) & ~(size_t)1;
	const size_t fixed = 24 * (size_t)n_edges + 2 * n6p + n6p + (((size_t)n_suf + 1) & ~(size_t)1) +
	                     3 * (size_t)n_tiles + 2 * (size_t)n_groups + ((size_t)n_tiles & 1);      // (+ the tile / group descriptors: 24 and 16 bytes each)
	const size_t recs = (size_t)waves * 64 * SSK_ND_REC;
	if (!big) return fixed + 36 * ((size_t)E * (E + 1) / 2) + std::max(recs, (n6 + 1) * (n6 | 1));
	const size_t want = fixed + std::max(recs, (n6 + 1) * (n6 + 2) / 2);
	return std::max(fixed + recs, std::min(want, limit));
}

// Appends window Wn to H (bases edge0 / lm0 / us0 / ls0 / trace offsets are the caller's: the families are shared with the other
// kernels' structures).  Returns false -- and leaves H untouched -- when the window does not fit the node kernel:
// more than 128 optimised edges, a landmark with more than 64 observations or more than 63 nodes, more than 65535 suffixes.
// split = true: the tables of the PHASE-SPLIT form (ssk_lm_split.hip): landmarks ordered by the set of optimised edges their nodes touch (so that
// the landmarks of a tile share 6x6 blocks: 3x fewer (tile, block) pairs on a depth-5 local area), one quad slot per (tile, block) with a
// partial-sum slot in HBM (block-major numbering: the slots of a block are contiguous, tile order), gradient slots per (tile, edge) likewise.
inline bool nd_build_window(const ssk_lm_window& Wn, int waves, nd_host& H, nd_win hdr /* bases filled in by the caller */, size_t lds_limit_doubles = 20352, bool split = false)
{
	const int n_obs = Wn.n_obs, n_lms = Wn.n_lms_total;
	int E = 0, L = 0;
	for (int e = 0; e < Wn.n_edges_total; ++e) if (Wn.edge_unknown_slot[e] >= 0) ++E;
	for (int l = 0; l < n_lms; ++l) if (Wn.lm_unknown_slot[l] >= 0) ++L;
	if (E > 128 || waves < 1 || waves > SSK_ND_MAX_WAVES) return false;
	// ---- suffix trie (ids in creation order: the rest of a suffix always has a smaller id) ----
	std::vector<nd_i2> suf(1, nd_i2{-1, -1});
	std::vector<int> obs_suf((size_t)n_obs, 0);
	{
		std::unordered_map<uint64_t, int> ids;
		ids.reserve(1024);
		for (int o = 0; o < n_obs; ++o) {
			int cur = 0;
			for (int p = Wn.obs_path_off[o + 1] - 1; p >= Wn.obs_path_off[o]; --p) {
				const uint64_t key = ((uint64_t)(uint32_t)Wn.path_edge[p] << 32) | (uint32_t)cur;
				auto it = ids.find(key);
				if (it == ids.end()) { it = ids.emplace(key, (int)suf.size()).first; suf.push_back(nd_i2{Wn.path_edge[p], cur}); }
				cur = it->second;
			}
			obs_suf[o] = cur;
		}
	}
	const int n_suf = (int)suf.size();
	if (n_suf > 65535) return false;
	// ---- landmarks: observations sorted so that every trie subtree is a run; nodes ----
	std::vector<std::vector<int>> lm_obs((size_t)n_lms);
	for (int o = 0; o < n_obs; ++o) lm_obs[(size_t)Wn.obs_lm[o]].push_back(o);
	struct node { int key, a, n; };                       // key suffix, run of the landmark's sorted observations
	struct lmk { int l, obs_first; std::vector<int> obs; std::vector<node> nodes; };
	std::vector<lmk> lms;
	auto optimised = [&](int sid) { return Wn.edge_unknown_slot[suf[(size_t)sid].x >> 1] >= 0; };
	std::vector<std::vector<int>> chains;                 // per observation of the current landmark: suffix ids from the shortest to the whole path
	for (int l = 0; l < n_lms; ++l) {
		const std::vector<int>& ob = lm_obs[(size_t)l];
		if (ob.empty()) continue;                         // never observed: nothing to linearise, nothing moves it
		if (ob.size() > 64) return false;
		chains.assign(ob.size(), std::vector<int>());
		for (size_t k = 0; k < ob.size(); ++k) {
			for (int s = obs_suf[(size_t)ob[k]]; s != 0; s = suf[(size_t)s].y) chains[k].push_back(s);
			std::reverse(chains[k].begin(), chains[k].end());
		}
		std::vector<int> ord(ob.size());
		for (size_t k = 0; k < ord.size(); ++k) ord[k] = (int)k;
		std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return chains[(size_t)a] < chains[(size_t)b]; });
		lmk M;
		M.l = l; M.obs_first = 0;
		for (int k : ord) M.obs.push_back(ob[(size_t)k]);
		// nodes in the order of their first observation (depth first): a node's observations are the run of those whose chain holds its key
		for (size_t k = 0; k < ord.size(); ++k)
			for (int s : chains[(size_t)ord[k]]) {
				if (!optimised(s)) continue;
				bool seen = false;
				for (node& nd : M.nodes) if (nd.key == s) { nd.n = (int)k + 1 - nd.a; seen = true; break; }
				if (!seen) M.nodes.push_back(node{s, (int)k, 1});
			}
		if (M.nodes.size() > 63) return false;
		lms.push_back(std::move(M));
	}
	if (!split) {
		// optimised landmarks first, in slot order, then the fixed ones
		std::stable_sort(lms.begin(), lms.end(), [&](const lmk& a, const lmk& b) {
			const int sa = Wn.lm_unknown_slot[a.l], sb = Wn.lm_unknown_slot[b.l];
			if ((sa >= 0) != (sb >= 0)) return sa >= 0;
			return sa >= 0 ? sa < sb : a.l < b.l;
		});
	} else {
		// by the sorted set of optimised edge slots of the landmark's nodes, lexicographically (ties: landmark index)
		std::vector<std::vector<int>> eset(lms.size());
		for (size_t m = 0; m < lms.size(); ++m) {
			for (const node& nd : lms[m].nodes) eset[m].push_back(Wn.edge_unknown_slot[suf[(size_t)nd.key].x >> 1]);
			std::sort(eset[m].begin(), eset[m].end());
			eset[m].erase(std::unique(eset[m].begin(), eset[m].end()), eset[m].end());
		}
		std::vector<int> ord(lms.size());
		for (size_t m = 0; m < ord.size(); ++m) ord[m] = (int)m;
		std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return eset[(size_t)a] != eset[(size_t)b] ? eset[(size_t)a] < eset[(size_t)b] : lms[(size_t)a].l < lms[(size_t)b].l; });
		std::vector<lmk> sorted;
		sorted.reserve(lms.size());
		for (int m : ord) sorted.push_back(std::move(lms[(size_t)m]));
		lms.swap(sorted);
	}
	// ---- groups: greedy runs of landmarks with <= 64 observations and <= 64 lanes ----
	const size_t win_i = H.win.size();
	(void)win_i;
	hdr.n_edges = Wn.n_edges_total; hdr.E = E; hdr.n_lms = n_lms; hdr.L = L; hdr.n_obs = n_obs; hdr.n_suf = n_suf;
	hdr.suf0 = (int)H.suf.size(); hdr.tile0 = (int)H.tile.size(); hdr.grp0 = (int)H.group.size();
	hdr.obs0 = (int)H.obs.size(); hdr.lane0 = (int)H.lane.size();
	hdr.item0 = (int)H.item.size(); hdr.qs0 = (int)H.qs.size(); hdr.gitem0 = (int)H.gitem.size(); hdr.gqs0 = (int)H.gqs.size();
	hdr.tab0 = H.tab_doubles;
	hdr.hb0 = H.hb_doubles;
	std::vector<nd_group> groups;
	std::vector<nd_i2> w_obs;
	std::vector<nd_f4> w_px;
	std::vector<nd_i4> w_lane;
	struct gnode { int lmk, node, slot_lane; };            // per lane of a group: which landmark / node (-1 = root)
	std::vector<std::vector<gnode>> glanes;
	{
		nd_group G{0, 0, 0, 0};
		std::vector<gnode> gl;
		auto flush = [&] {
			if (G.n_lanes == 0) return;
			groups.push_back(G); glanes.push_back(gl);
			G.obs0 += G.n_obs; G.lane0 += G.n_lanes; G.n_obs = 0; G.n_lanes = 0; gl.clear();
		};
		for (size_t m = 0; m < lms.size(); ++m) {
			const lmk& M = lms[m];
			const int no = (int)M.obs.size(), nl = 1 + (int)M.nodes.size();
			if (G.n_obs + no > 64 || G.n_lanes + nl > 64) flush();
			const int root_lane = G.n_lanes, obs_a = G.n_obs;
			for (int o : M.obs) {
				w_obs.push_back(nd_i2{M.l, obs_suf[(size_t)o] | (root_lane << 16)});
				w_px.push_back(nd_f4{Wn.obs_px[4 * o], Wn.obs_px[4 * o + 1], Wn.obs_px[4 * o + 2], Wn.obs_px[4 * o + 3]});
			}
			w_lane.push_back(nd_i4{M.l, obs_a | (no << 8), Wn.lm_unknown_slot[M.l], 1 | (root_lane << 8) | ((int)M.nodes.size() << 16)});
			gl.push_back(gnode{(int)m, -1, root_lane});
			for (size_t k = 0; k < M.nodes.size(); ++k) {
				const node& nd = M.nodes[k];
				const int code = suf[(size_t)nd.key].x, inv = code & 1;
				const int tsuf = inv ? suf[(size_t)nd.key].y : nd.key;
				const int es = Wn.edge_unknown_slot[code >> 1];
				w_lane.push_back(nd_i4{M.l, (obs_a + nd.a) | (nd.n << 8), tsuf | (es << 16) | (inv << 24), 2 | (root_lane << 8)});
				gl.push_back(gnode{(int)m, (int)k, root_lane + 1 + (int)k});
			}
			G.n_obs += no; G.n_lanes += nl;
		}
		flush();
	}
	// ---- tiles: `waves` groups each; the items of a tile sorted by 6x6 block ----
	const int NQ = waves * 16;
	(void)NQ;
	const int n_pairs = E * (E + 1) / 2;
	auto pid_of = [](int e, int e2) { return e2 * (e2 + 1) / 2 + e; };   // e <= e2
	std::vector<nd_tile> tiles;
	std::vector<int> w_item, w_gitem;
	std::vector<nd_i2> w_qs, w_gqs;
	const size_t n_tiles = (groups.size() + (size_t)waves - 1) / (size_t)waves;
	std::vector<std::vector<std::pair<int, int>>> tile_items(n_tiles);       // (block, item) per tile, generation order
	std::vector<std::vector<std::vector<int>>> tile_edge(n_tiles, std::vector<std::vector<int>>((size_t)E));
	for (size_t t = 0; t < n_tiles; ++t) {
		const size_t g0 = t * (size_t)waves;
		const int n_grp = (int)std::min(groups.size() - g0, (size_t)waves);
		for (int wv = 0; wv < n_grp; ++wv) {
			const std::vector<gnode>& gl = glanes[g0 + (size_t)wv];
			for (size_t a = 0; a < gl.size(); ++a) {
				if (gl[a].node < 0) continue;
				const lmk& M = lms[(size_t)gl[a].lmk];
				const bool opt = Wn.lm_unknown_slot[M.l] >= 0;
				const int ka = M.nodes[(size_t)gl[a].node].key, ea = Wn.edge_unknown_slot[suf[(size_t)ka].x >> 1];
				const int slot_a = 64 * wv + gl[a].slot_lane;
				tile_edge[t][(size_t)ea].push_back(slot_a);
				// partners: the later nodes of the same landmark (they follow in the lane order) and the node itself
				for (size_t b = a; b < gl.size() && gl[b].lmk == gl[a].lmk; ++b) {
					const int kb = M.nodes[(size_t)gl[b].node].key, eb = Wn.edge_unknown_slot[suf[(size_t)kb].x >> 1];
					const int slot_b = 64 * wv + gl[b].slot_lane;
					// nested: one key is a suffix of the other (walk the longer one's rests)
					auto is_suffix_of = [&](int shorter, int longer) { for (int s = longer; s != 0; s = suf[(size_t)s].y) if (s == shorter) return true; return false; };
					int far = 0;                                   // 1: a far (b's key is a suffix of a's), 2: b far, 0: not nested
					if (is_suffix_of(kb, ka)) far = 1; else if (is_suffix_of(ka, kb)) far = 2;
					if (far == 0 && !opt) continue;
					// item (row r, column c): the far node of a nested pair gives Zh, the near one Ph; not nested: Zh, Zh and the sign bit
					auto emit = [&](int slot_r, int slot_c, bool r_far, bool c_far, int pid) {
						const int xp = (far != 0 && !r_far) ? 1 : 0, yp = (far != 0 && !c_far) ? 1 : 0;
						tile_items[t].push_back(std::make_pair(pid, slot_r | (xp << 9) | (slot_c << 10) | (yp << 19) | ((far == 0 ? 1 : 0) << 20)));
					};
					if (a == b) emit(slot_a, slot_a, true, false, pid_of(ea, ea));
					else if (ea < eb) emit(slot_a, slot_b, far == 1, far == 2, pid_of(ea, eb));
					else if (eb < ea) emit(slot_b, slot_a, far == 2, far == 1, pid_of(eb, ea));
					else {                                         // two nodes of one landmark on the same edge: both orders go to the diagonal block
						emit(slot_a, slot_b, far == 1, far == 2, pid_of(ea, ea));
						emit(slot_b, slot_a, far == 2, far == 1, pid_of(ea, ea));
					}
				}
			}
		}
	}
	std::vector<nd_i4> w_sqs;
	std::vector<int> w_poff, w_goff, w_gslot;
	int n_part = 0, n_gpart = 0, max_tile_blocks = 0;
	if (split) {
		std::vector<std::vector<int>> by_pid((size_t)n_pairs);
		std::vector<char> is_diag((size_t)n_pairs, 0);         // block (e, e): the only ones the lambda_0 pass needs (bit 30 of the slot's x)
		for (int e = 0; e < E; ++e) is_diag[(size_t)pid_of(e, e)] = 1;
		std::vector<int> used;
		for (size_t t = 0; t < n_tiles; ++t) {
			nd_tile T{};
			T.grp0 = (int)(t * (size_t)waves); T.n_grp = (int)std::min(groups.size() - t * (size_t)waves, (size_t)waves);
			used.clear();
			for (const auto& it : tile_items[t]) { if (by_pid[(size_t)it.first].empty()) used.push_back(it.first); by_pid[(size_t)it.first].push_back(it.second); }
			// heaviest first: the 16 quads of a wavefront run in lockstep
			std::stable_sort(used.begin(), used.end(), [&](int a, int b) { return by_pid[(size_t)a].size() > by_pid[(size_t)b].size(); });
			// LANE slots: a block gets 1, 2 or 4 lanes (its items dealt out among them, the lanes' sums joined by one or two DPP steps), the fewest
			// that keep a lane's run <= R items; R = the shortest run length whose lanes fit the fewest rounds of the workgroup.  Heaviest blocks first,
			// so the 4-lane blocks come first, then the 2-lane ones: every group is aligned and a wavefront's lanes walk runs of similar length.
			const int NT_ = waves * 64;
			const int rounds_min = std::max(1, ((int)used.size() + NT_ - 1) / NT_);
			auto lanes_of = [](int c, int R) { return c <= R ? 1 : (c <= 2 * R ? 2 : 4); };
			int R = 1;
			for (;; ++R) {
				long long lanes = 0;
				for (int p2 : used) lanes += lanes_of((int)by_pid[(size_t)p2].size(), R);
				if (lanes <= (long long)NT_ * rounds_min) break;
			}
			T.qs0 = (int)w_sqs.size();
			max_tile_blocks = std::max(max_tile_blocks, (int)used.size());
			for (int p2 : used) {
				const std::vector<int>& v = by_pid[(size_t)p2];
				const int k = lanes_of((int)v.size(), R), kc = k == 4 ? 2 : (k == 2 ? 1 : 0);
				if ((int)v.size() > 0xFFFF) return false;
				for (int r = 0; r < k; ++r)
					w_sqs.push_back(nd_i4{p2 | (is_diag[(size_t)p2] ? 1 << 30 : 0), (int)v.size() | (kc << 16) | (r << 20), (int)w_item.size(), -1});
				w_item.insert(w_item.end(), v.begin(), v.end());
			}
			T.n_rounds = (int)w_sqs.size() - T.qs0;               // (split form: n_rounds = number of lane slots of the tile)
			// bit 29: this wavefront (64 consecutive lane slots) holds blocks of several lanes
			for (int w0 = T.qs0; w0 < (int)w_sqs.size(); w0 += 64) {
				bool multi = false;
				for (int i = w0; i < std::min(w0 + 64, (int)w_sqs.size()); ++i) multi |= ((w_sqs[(size_t)i].y >> 16) & 3) != 0;
				if (multi) for (int i = w0; i < std::min(w0 + 64, (int)w_sqs.size()); ++i) w_sqs[(size_t)i].x |= 1 << 29;
			}
			for (int p2 : used) by_pid[(size_t)p2].clear();
			T.gqs0 = (int)w_gqs.size(); T.n_gqs = 0;
			for (int e = 0; e < E; ++e) {
				const std::vector<int>& v = tile_edge[t][(size_t)e];
				if (v.empty()) continue;
				w_gqs.push_back(nd_i2{e | ((int)v.size() << 16), (int)w_gitem.size()});
				w_gitem.insert(w_gitem.end(), v.begin(), v.end());
				++T.n_gqs;
			}
			tiles.push_back(T);
		}
		// partial slots, block-major (a block's slots are contiguous, in tile order); gradient slots, edge-major
		w_poff.assign((size_t)n_pairs + 1, 0);
		for (const nd_i4& q : w_sqs) if (((q.y >> 20) & 3) == 0) w_poff[(size_t)(q.x & 0xFFFF) + 1]++;
		for (int p2 = 0; p2 < n_pairs; ++p2) w_poff[(size_t)p2 + 1] += w_poff[(size_t)p2];
		{
			std::vector<int> next(w_poff.begin(), w_poff.end() - 1);
			for (nd_i4& q : w_sqs) q.w = ((q.y >> 20) & 3) == 0 ? next[(size_t)(q.x & 0xFFFF)]++ : next[(size_t)(q.x & 0xFFFF)] - 1;      // (the lanes of a block follow its first one)
		}
		n_part = w_poff[(size_t)n_pairs];
		w_goff.assign((size_t)E + 1, 0);
		for (const nd_i2& q : w_gqs) w_goff[(size_t)(q.x & 0xFFFF) + 1]++;
		for (int e = 0; e < E; ++e) w_goff[(size_t)e + 1] += w_goff[(size_t)e];
		{
			std::vector<int> next(w_goff.begin(), w_goff.end() - 1);
			w_gslot.resize(w_gqs.size());
			for (size_t k = 0; k < w_gqs.size(); ++k) w_gslot[k] = next[(size_t)(w_gqs[k].x & 0xFFFF)]++;
		}
		n_gpart = (int)w_gqs.size();
	} else {
#if SSK_ND_LANE_RUNS
	// lane slots of a tile: the items of a block dealt out in runs of R consecutive items, a lane per run; R = the shortest run length
	// whose lanes fit the workgroup (one round); the lanes of one block stay inside one wavefront (their sums meet in ONE ds_add_f64 per
	// component: the order of the adds is the LDS unit's lane order, the same in every launch)
	int max_per_lane = 0, max_rounds = 0;
	const int NT_ = waves * 64;
	std::vector<std::vector<int>> by_pid((size_t)n_pairs);
	std::vector<int> used;
	for (size_t t = 0; t < n_tiles; ++t) {
		nd_tile T{};
		T.grp0 = (int)(t * (size_t)waves); T.n_grp = (int)std::min(groups.size() - t * (size_t)waves, (size_t)waves);
		used.clear();
		for (const auto& it : tile_items[t]) { if (by_pid[(size_t)it.first].empty()) used.push_back(it.first); by_pid[(size_t)it.first].push_back(it.second); }
		std::stable_sort(used.begin(), used.end(), [&](int a, int b) { return by_pid[(size_t)a].size() > by_pid[(size_t)b].size(); });
		auto lanes_for = [&](int R) {
			long long pos = 0;
			for (int p2 : used) {
				const long long g = ((long long)by_pid[(size_t)p2].size() + R - 1) / R;
				if (g > 64) return (long long)1 << 40;
				if ((pos & 63) + g > 64) pos = (pos + 63) & ~63LL;
				pos += g;
			}
			return pos;
		};
		int R = 1;
		while (R < 8 && lanes_for(R) > NT_) ++R;
		if (lanes_for(R) >= ((long long)1 << 40)) return false;
		T.qs0 = (int)w_qs.size();
		long long pos = 0;
		for (int p2 : used) {
			const std::vector<int>& v = by_pid[(size_t)p2];
			const long long g = ((long long)v.size() + R - 1) / R;
			if ((pos & 63) + g > 64) for (; pos & 63; ++pos) w_qs.push_back(nd_i2{0xFFFF, 0});
			for (long long k = 0; k < g; ++k) {
				const int first = (int)w_item.size(), n = (int)std::min<long long>(R, (long long)v.size() - k * R);
				w_item.insert(w_item.end(), v.begin() + k * R, v.begin() + k * R + n);
				w_qs.push_back(nd_i2{p2 | (n << 16), first});
				++pos;
			}
			max_per_lane = std::max(max_per_lane, std::min(R, (int)v.size()));
		}
		for (; pos % NT_; ++pos) w_qs.push_back(nd_i2{0xFFFF, 0});
		T.n_rounds = (int)(pos / NT_);
		max_rounds = std::max(max_rounds, T.n_rounds);
		for (int p2 : used) by_pid[(size_t)p2].clear();
		T.gqs0 = (int)w_gqs.size(); T.n_gqs = 0;
		for (int e = 0; e < E; ++e) {
			const std::vector<int>& v = tile_edge[t][(size_t)e];
			if (v.empty()) continue;
			w_gqs.push_back(nd_i2{e | ((int)v.size() << 16), (int)w_gitem.size()});
			w_gitem.insert(w_gitem.end(), v.begin(), v.end());
			++T.n_gqs;
		}
		tiles.push_back(T);
	}
#else
	// quad slots of a tile: a block per quad, heaviest first (the 16 quads of a wavefront run in lockstep: similar loads share one); with quads
	// to spare the heaviest blocks are cut in two parts (items dealt out alternately) that sit in the two quads of an aligned lane octet
	// and are added up by a DPP mirror step before the block's accumulator takes them
	int max_per_lane = 0, max_rounds = 0;
	std::vector<std::vector<int>> by_pid((size_t)n_pairs);
	std::vector<int> used;
	for (size_t t = 0; t < n_tiles; ++t) {
		nd_tile T{};
		T.grp0 = (int)(t * (size_t)waves); T.n_grp = (int)std::min(groups.size() - t * (size_t)waves, (size_t)waves);
		used.clear();
		for (const auto& it : tile_items[t]) { if (by_pid[(size_t)it.first].empty()) used.push_back(it.first); by_pid[(size_t)it.first].push_back(it.second); }
		std::stable_sort(used.begin(), used.end(), [&](int a, int b) { return by_pid[(size_t)a].size() > by_pid[(size_t)b].size(); });
		int n_split = 0;
		if ((int)used.size() < NQ) {
			n_split = std::min((NQ - (int)used.size()), (int)used.size());
			while (n_split > 0 && by_pid[(size_t)used[(size_t)n_split - 1]].size() < 8) --n_split;       // (a short run is not worth a second quad)
		}
		struct qslot { int pid, part, parts; };
		std::vector<qslot> slots;
		for (int k = 0; k < n_split; ++k) { slots.push_back(qslot{used[(size_t)k], 0, 2}); slots.push_back(qslot{used[(size_t)k], 1, 2}); }
		for (size_t k = (size_t)n_split; k < used.size(); ++k) slots.push_back(qslot{used[k], 0, 1});
		T.qs0 = (int)w_qs.size();
		T.n_rounds = ((int)slots.size() + NQ - 1) / NQ;
		max_rounds = std::max(max_rounds, T.n_rounds);
		for (size_t k = 0; k < (size_t)T.n_rounds * NQ; ++k) {
			const int wave_flag = (k / 16) * 16 < (size_t)(2 * n_split) ? (int)0x80000000u : 0;      // this wavefront holds blocks cut in two
			if (k >= slots.size()) { w_qs.push_back(nd_i2{0xFFFF, wave_flag}); continue; }
			const qslot& q = slots[k];
			const std::vector<int>& v = by_pid[(size_t)q.pid];
			const int first = (int)w_item.size();
			for (size_t i = (size_t)q.part; i < v.size(); i += (size_t)q.parts) w_item.push_back(v[i]);
			const int n = (int)w_item.size() - first;
			if (n > 32767 || first >= (1 << 28)) return false;
			w_qs.push_back(nd_i2{q.pid | (n << 16), first | (q.part << 28) | (q.parts == 2 ? 1 << 30 : 0) | wave_flag});
			max_per_lane = std::max(max_per_lane, (n + 3) / 4);
		}
		for (int p : used) by_pid[(size_t)p].clear();
		T.gqs0 = (int)w_gqs.size(); T.n_gqs = 0;
		for (int e = 0; e < E; ++e) {
			const std::vector<int>& v = tile_edge[t][(size_t)e];
			if (v.empty()) continue;
			w_gqs.push_back(nd_i2{e | ((int)v.size() << 16), (int)w_gitem.size()});
			w_gitem.insert(w_gitem.end(), v.begin(), v.end());
			++T.n_gqs;
		}
		tiles.push_back(T);
	}
#endif
	hdr.n_rounds = max_rounds; hdr.max_parts = 2;
	H.max_items_per_lane = std::max(H.max_items_per_lane, max_per_lane);
	}
	hdr.n_tiles = (int)tiles.size(); hdr.n_groups = (int)groups.size();
	// ---- commit ----
	H.suf.insert(H.suf.end(), suf.begin(), suf.end());
	H.tile.insert(H.tile.end(), tiles.begin(), tiles.end());
	H.group.insert(H.group.end(), groups.begin(), groups.end());
	H.obs.insert(H.obs.end(), w_obs.begin(), w_obs.end());
	H.px.insert(H.px.end(), w_px.begin(), w_px.end());
	H.lane.insert(H.lane.end(), w_lane.begin(), w_lane.end());
	H.item.insert(H.item.end(), w_item.begin(), w_item.end());
	H.qs.insert(H.qs.end(), w_qs.begin(), w_qs.end());
	H.gitem.insert(H.gitem.end(), w_gitem.begin(), w_gitem.end());
	H.gqs.insert(H.gqs.end(), w_gqs.begin(), w_gqs.end());
	H.tab_doubles += 2LL * 12 * n_suf;
	H.hb_doubles += 36LL * E * (E + 1) / 2;
	H.lds_doubles = std::max(H.lds_doubles, nd_lds_doubles(Wn.n_edges_total, E, waves, n_suf, (int)tiles.size(), (int)groups.size()));
	H.lds_doubles_big = std::max(H.lds_doubles_big, nd_lds_doubles(Wn.n_edges_total, E, waves, n_suf, (int)tiles.size(), (int)groups.size(), true, lds_limit_doubles));
	if (split) {
		nd_split S{};
		S.sqs0 = (int)H.sqs.size(); S.poff0 = (int)H.poff.size(); S.goff0 = (int)H.goff.size(); S.gslot0 = (int)H.gslot.size();
		S.n_part = n_part; S.n_gpart = n_gpart; S.part0 = H.part_slots; S.gpart0 = H.gpart_slots;
		S.gtile0 = (int)H.tile.size() - (int)tiles.size();
		H.sqs.insert(H.sqs.end(), w_sqs.begin(), w_sqs.end());
		H.poff.insert(H.poff.end(), w_poff.begin(), w_poff.end());
		H.goff.insert(H.goff.end(), w_goff.begin(), w_goff.end());
		H.gslot.insert(H.gslot.end(), w_gslot.begin(), w_gslot.end());
		H.part_slots += n_part; H.gpart_slots += n_gpart;
		H.max_tile_blocks = std::max(H.max_tile_blocks, max_tile_blocks);
		H.swin.push_back(S);
	}
	H.win.push_back(hdr);
	return true;
}
