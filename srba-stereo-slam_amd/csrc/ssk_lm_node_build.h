// ssk_lm_node_build.h -- host side of the NODE form (ssk_lm_node.h): the symbolic structure of one window -- suffix trie, nodes,
// landmark groups, tiles, block-sorted items -- as SRBA does its graph bookkeeping on the CPU.  HIP free (plain C++17): the
// library (ssk_capi_lm.hip) and the CPU emulation of the kernel's data flow (tools/lm_node_emu.cpp, tests/test_lm_node_form.py)
// compile the same code.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <utility>
#include <vector>
#include "../../include/ssk_capi.h"
#include "ssk_lm_node.h"

// Vectors of the node tables: resize() leaves new elements uninitialised (the batch-wide tables are sized by prefix sums and then filled by the
// builder's threads: a value-initialising resize wrote 100 MB of zeros on one thread first), and an allocator bound to a bump arena takes its
// storage from there -- the library hands over PINNED host memory, so that a batch's tables go to the device by DMA straight from where the
// builder's threads wrote them (no pageable staging copy).  An arena allocation is never freed piecewise (the arena is reset as a whole).
struct nd_bump { char* base = nullptr; size_t cap = 0, used = 0; };
template <typename T>
struct nd_noinit {
	using value_type = T;
	using propagate_on_container_move_assignment = std::true_type;      // (v = vector(allocator(arena)) binds v to the arena)
	using propagate_on_container_swap = std::true_type;
	nd_bump* arena = nullptr;
	nd_noinit() = default;
	explicit nd_noinit(nd_bump* a) : arena(a) {}
	template <typename U> nd_noinit(const nd_noinit<U>& o) : arena(o.arena) {}
	template <typename U> struct rebind { using other = nd_noinit<U>; };
	T* allocate(size_t n)
	{
		if (arena) {
			const size_t off = (arena->used + 63) & ~(size_t)63, bytes = n * sizeof(T);
			if (off + bytes <= arena->cap) { arena->used = off + bytes; return reinterpret_cast<T*>(arena->base + off); }
		}
		return static_cast<T*>(::operator new(n * sizeof(T)));
	}
	void deallocate(T* p, size_t)
	{
		if (arena && (char*)p >= arena->base && (char*)p < arena->base + arena->cap) return;
		::operator delete(p);
	}
	template <typename U> void construct(U* p) { ::new ((void*)p) U; }
	template <typename U, typename... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
	template <typename U> bool operator==(const nd_noinit<U>& o) const { return arena == o.arena; }
	template <typename U> bool operator!=(const nd_noinit<U>& o) const { return arena != o.arena; }
};
template <typename T> using nd_vec = std::vector<T, nd_noinit<T>>;

struct nd_host {
	std::vector<nd_win> win;
	nd_vec<nd_tile> tile;
	nd_vec<nd_group> group;
	nd_vec<nd_i2> suf, obs, qs, gqs;
	nd_vec<nd_f4> px;
	nd_vec<nd_i4> lane;
	nd_vec<int> item, gitem;
	long long tab_doubles = 0;          // suffix tables of all windows
	size_t lds_doubles = 0;             // largest dynamic LDS need of a window (doubles)
	size_t lds_doubles_big = 0;         // ... in the BIG form (accumulators and system in HBM / L2)
	long long hb_doubles = 0;           // BIG form: block accumulators of all windows
	int max_items_per_lane = 0;         // (diagnostic: the longest item run of a quad lane)
	// ---- phase-split form (ssk_lm_split.hip): one nd_split per window beside its nd_win ----
	std::vector<nd_split> swin;
	nd_vec<nd_i4> sqs;                  // (tile, block) slots: x block pair id, y items, z first item (window relative), w partial slot (window relative)
	nd_vec<int> poff, goff, gslot;      // poff: per window n_pairs + 1 (partial slots of a block, block-major); goff: E + 1; gslot: per gqs entry its gradient partial slot
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
// the landmarks of a tile share 6x6 blocks: 3x fewer (tile, block) pairs on a depth-5 local area), LANE slots (a block has 1, 2 or 4 lanes) with a
// partial-sum slot in HBM per (tile, block) (block-major numbering: the slots of a block are contiguous, tile order), gradient slots per
// (tile, edge) likewise.
// Round 4: flat arrays and counting sorts instead of a vector per landmark / tile / block (3.4 ms -> see DESIGN.md for a depth-5 local area: this
// runs on the per-keyframe host path of the facade and in front of every batch).
static_assert(SSK_ND_LANE_RUNS == 0, "the host builder emits quad slots (SSK_ND_LANE_RUNS = 0)");

namespace nd_detail {
// open-addressing map (suffix record -> id) of the trie: keys (first step code << 32 | rest id), linear probing, power-of-two capacity
struct suf_map {
	std::vector<uint64_t> key;
	std::vector<int> val;
	size_t mask = 0, used = 0;
	void init(size_t cap) { size_t c = 64; while (c < cap) c <<= 1; key.assign(c, ~(uint64_t)0); val.assign(c, 0); mask = c - 1; used = 0; }
	static size_t hash(uint64_t k) { k ^= k >> 29; k *= 0x9E3779B97F4A7C15ull; k ^= k >> 32; return (size_t)k; }
	void grow()
	{
		std::vector<uint64_t> ok; std::vector<int> ov;
		ok.swap(key); ov.swap(val);
		init(2 * (mask + 1));
		for (size_t i = 0; i < ok.size(); ++i) if (ok[i] != ~(uint64_t)0) { size_t h = hash(ok[i]) & mask; while (key[h] != ~(uint64_t)0) h = (h + 1) & mask; key[h] = ok[i]; val[h] = ov[i]; ++used; }
	}
	// returns the id of k, inserting `fresh` when absent (inserted = true)
	int get(uint64_t k, int fresh, bool& inserted)
	{
		size_t h = hash(k) & mask;
		while (key[h] != ~(uint64_t)0) { if (key[h] == k) { inserted = false; return val[h]; } h = (h + 1) & mask; }
		key[h] = k; val[h] = fresh; inserted = true;
		if (++used * 2 > mask) grow();
		return fresh;
	}
};
// set of optimised edge slots (<= 128) in the order of the sorted vector of its elements, lexicographically
struct eset { uint64_t lo = 0, hi = 0; };
inline bool eset_less(const eset& a, const eset& b)
{
	const uint64_t dlo = a.lo ^ b.lo, dhi = a.hi ^ b.hi;
	if (!dlo && !dhi) return false;
	// the lowest element in which they differ: the set that holds it is smaller, unless the other has nothing at or above it (a prefix: shorter, smaller)
	if (dlo) {
		const int i = __builtin_ctzll(dlo);
		const bool a_has = (a.lo >> i) & 1;
		const eset& other = a_has ? b : a;
		const bool other_ends = (i == 63 ? 0 : (other.lo >> (i + 1))) == 0 && other.hi == 0;
		return a_has ? !other_ends : other_ends;
	}
	const int i = __builtin_ctzll(dhi);
	const bool a_has = (a.hi >> i) & 1;
	const eset& other = a_has ? b : a;
	const bool other_ends = (i == 63 ? 0 : (other.hi >> (i + 1))) == 0;
	return a_has ? !other_ends : other_ends;
}
}  // namespace nd_detail

namespace nd_detail {
struct node { int key, a, n; };                       // key suffix, run of the landmark's sorted observations
struct lmk { int l, obs0, n_obs, node0, n_nodes; };   // observations lm_sorted[obs0 ..), nodes nodes[node0 ..)
struct gnode { int lmk, node, slot_lane; };           // per lane of a group: which landmark / node (-1 = root)
struct skey { uint64_t yh, yl; int pc, l, m; };        // sort key of a landmark's set of optimised edge slots (split form)
// The builder's temporaries, one set per host thread, kept between windows: a window of a batch is a few hundred microseconds of work, two dozen
// vectors allocated and freed per window were a fifth of it (round 6).
// what a range of landmarks / a range of tiles leaves when helper threads build one window's parts side by side (nd_team): merged in range order, so the
// window's tables come out byte for byte as from one thread
struct alignas(128) part_lms { std::vector<lmk> lms; std::vector<node> nodes; bool fail = false; void clear() { lms.clear(); nodes.clear(); fail = false; } };
struct alignas(128) part_tiles {      // (alignas: neighbouring parts' vector headers -- written at every push_back -- must not share a cache line)
	std::vector<nd_i4> sqs; std::vector<nd_i2> qs, gqs; std::vector<int> item, gitem; std::vector<nd_tile> tile;
	int max_per_lane = 0, max_rounds = 0, max_tile_blocks = 0; bool fail = false;
	void clear() { sqs.clear(); qs.clear(); gqs.clear(); item.clear(); gitem.clear(); tile.clear(); max_per_lane = max_rounds = max_tile_blocks = 0; fail = false; }
};
struct workspace {
	suf_map ids;
	std::vector<nd_i2> suf;
	std::vector<int> suf_len, obs_suf, chain, suf_es, rank, child, sib, stack, node_mark, node_idx;
	std::vector<char> suf_opt;
	std::vector<uint64_t> path_key;                       // whole-path cache: hash -> (first path entry of a representative, length, id)
	std::vector<int> path_p0, path_n, path_id;
	std::vector<int> lo_off, lo_idx, next, lm_sorted, ord;
	std::vector<uint64_t> okey;
	std::vector<node> nodes;
	std::vector<lmk> lms, lms2;
	std::vector<eset> es;
	std::vector<gnode> glanes;
	std::vector<int> t_pid, t_code, cnt, start, fillp, used, used2, bucket, tri, sorted_code, e_cnt, e_fill, e_slot, e_of;
	std::vector<char> is_diag;
	std::vector<skey> sk;
	std::vector<part_lms> p_lms;                          // (the parts of a window built by helper threads: kept between windows like everything here)
	std::vector<part_tiles> p_tiles;
};
inline workspace& ws() { static thread_local workspace w; return w; }
}  // namespace nd_detail

// Helper threads for ONE window (round 6: the un-batched drop-in builds a single local area of ~10^4 observations per keyframe: 0.85 ms on one thread).
// run(n_parts, fn): fn(0 .. n_parts - 1), each exactly once, on the caller and up to n - 1 helpers; returns when all are done.  A batch of windows does not
// use it (its windows are the parallel items).  MEASURED AND NOT USED BY THE LIBRARY: with spinning helpers woken ahead of the create (no wake-up latency)
// the un-batched stream's creates took 103-112 ms per 100 keyframes against 88-96 on one thread (2-8 threads, GPU box: 256 cores under a 16-CPU quota, the
// helpers land on other core complexes and the two parallel sections, 0.2-0.45 ms, move a megabyte of tables between their caches); the tables are the same
// byte for byte (tests/test_lm_node_form.py builds the golden windows with 2, 3 and 8 threads).  Kept for a host with a known topology.
struct nd_team {
	int n = 1;
	std::function<void(int, const std::function<void(int)>&)> run;
};

#ifdef ND_BUILD_PROF
#include <chrono>
static double g_nd_prof[8];
#define ND_PROF(i) do { const auto t_ = std::chrono::steady_clock::now(); g_nd_prof[i] += std::chrono::duration<double, std::micro>(t_ - nd_prof_t_).count(); nd_prof_t_ = t_; } while (0)
#define ND_PROF0() auto nd_prof_t_ = std::chrono::steady_clock::now()
#else
#define ND_PROF(i) do {} while (0)
#define ND_PROF0() do {} while (0)
#endif
inline bool nd_build_window(const ssk_lm_window& Wn, int waves, nd_host& H, nd_win hdr /* bases filled in by the caller */, size_t lds_limit_doubles = 20352, bool split = false,
                            const nd_team* team = nullptr)
{
	ND_PROF0();
	using nd_detail::node; using nd_detail::lmk; using nd_detail::gnode;
	nd_detail::workspace& S = nd_detail::ws();
	const int n_obs = Wn.n_obs, n_lms = Wn.n_lms_total;
	int E = 0, L = 0;
	for (int e = 0; e < Wn.n_edges_total; ++e) if (Wn.edge_unknown_slot[e] >= 0) ++E;
	for (int l = 0; l < n_lms; ++l) if (Wn.lm_unknown_slot[l] >= 0) ++L;
	if (E > 128 || waves < 1 || waves > SSK_ND_MAX_WAVES) return false;
	ND_PROF(0);
	// ---- suffix trie (ids in creation order: the rest of a suffix always has a smaller id); chain[id][k] = the suffix of length k + 1 on id's
	// chain of rests (chain[id][len - 1] = id) ----
	constexpr int MAXP = 8;                                // (SSK_LM_MAX_PATH)
	std::vector<nd_i2>& suf = S.suf;
	std::vector<int>& suf_len = S.suf_len;
	std::vector<int>& obs_suf = S.obs_suf;
	suf.assign(1, nd_i2{-1, -1});
	suf_len.assign(1, 0);
	obs_suf.resize((size_t)n_obs);
	{
		nd_detail::suf_map& ids = S.ids;
		ids.init(2048);
		// whole paths seen before (direct mapped, 1024 entries): the observations of a keyframe reach a handful of base keyframes in turn, a path
		// step by step is one trie lookup per edge
		constexpr int PC = 1024;
		S.path_key.assign(PC, ~(uint64_t)0); S.path_p0.resize(PC); S.path_n.resize(PC); S.path_id.resize(PC);
		const int* prev = nullptr;
		int prev_n = -1, prev_id = 0;
		for (int o = 0; o < n_obs; ++o) {
			const int p0 = Wn.obs_path_off[o], pn = Wn.obs_path_off[o + 1] - p0;
			if (pn > MAXP) return false;
			// the observations of one keyframe pair follow each other and share their path
			if (pn == prev_n && (pn == 0 || std::memcmp(prev, Wn.path_edge + p0, sizeof(int) * (size_t)pn) == 0)) { obs_suf[(size_t)o] = prev_id; continue; }
			// (the key is only a filter: a hit is a slot whose stored LENGTH and entries equal this path's -- a first version compared the entries over
			//  this path's length alone and hashed [2, 0] to the empty path's key: the estimator's windows caught it, the seeded test windows did not)
			uint64_t hk = 0x243F6A8885A308D3ull ^ (uint64_t)pn;
			for (int p = 0; p < pn; ++p) { hk = (hk ^ (uint32_t)Wn.path_edge[p0 + p]) * 0x9E3779B97F4A7C15ull; hk ^= hk >> 32; }
			if (hk == ~(uint64_t)0) hk = 0;
			const size_t slot = (size_t)(hk >> 7) & (PC - 1);
			int cur = 0;
			if (S.path_key[slot] == hk && S.path_n[slot] == pn && (pn == 0 || std::memcmp(Wn.path_edge + S.path_p0[slot], Wn.path_edge + p0, sizeof(int) * (size_t)pn) == 0)) cur = S.path_id[slot];
			else {
				for (int p = p0 + pn - 1; p >= p0; --p) {
					const uint64_t key = ((uint64_t)(uint32_t)Wn.path_edge[p] << 32) | (uint32_t)cur;
					bool ins;
					const int id = ids.get(key, (int)suf.size(), ins);
					if (ins) { suf.push_back(nd_i2{Wn.path_edge[p], cur}); suf_len.push_back(suf_len[(size_t)cur] + 1); }
					cur = id;
				}
				S.path_key[slot] = hk; S.path_p0[slot] = p0; S.path_n[slot] = pn; S.path_id[slot] = cur;
			}
			obs_suf[(size_t)o] = cur;
			prev = Wn.path_edge + p0; prev_n = pn; prev_id = cur;
		}
	}
	const int n_suf = (int)suf.size();
	if (n_suf > 65535) return false;
	std::vector<int>& chain = S.chain;
	std::vector<char>& suf_opt = S.suf_opt;                // the suffix starts at an optimised edge
	std::vector<int>& suf_es = S.suf_es;                   // ... its slot
	chain.assign((size_t)n_suf * MAXP, 0);
	suf_opt.assign((size_t)n_suf, 0);
	suf_es.assign((size_t)n_suf, -1);
	for (int id = 1; id < n_suf; ++id) {
		const int len = suf_len[(size_t)id], rest = suf[(size_t)id].y;
		for (int k = 0; k + 1 < len; ++k) chain[(size_t)id * MAXP + k] = chain[(size_t)rest * MAXP + k];
		chain[(size_t)id * MAXP + len - 1] = id;
		suf_es[(size_t)id] = Wn.edge_unknown_slot[suf[(size_t)id].x >> 1];
		suf_opt[(size_t)id] = suf_es[(size_t)id] >= 0;
	}
	// rank[id]: position of the suffix in the depth-first walk of the trie that takes a node's children in ascending id -- the lexicographic order of
	// the chains (the sequence of suffix ids from the shortest to the whole path, a prefix first): what a landmark's observations are sorted by
	std::vector<int>& rank = S.rank;
	{
		std::vector<int>& child = S.child; std::vector<int>& sib = S.sib; std::vector<int>& stack = S.stack;
		child.assign((size_t)n_suf, -1); sib.assign((size_t)n_suf, -1); rank.resize((size_t)n_suf);
		for (int id = n_suf - 1; id >= 1; --id) { const int par = suf[(size_t)id].y; sib[(size_t)id] = child[(size_t)par]; child[(size_t)par] = id; }
		stack.clear(); stack.push_back(0);
		int r = 0;
		while (!stack.empty()) {
			const int id = stack.back(); stack.pop_back();
			rank[(size_t)id] = r++;
			// children pushed in descending id, popped ascending
			int c = child[(size_t)id], n0 = (int)stack.size();
			for (; c >= 0; c = sib[(size_t)c]) stack.push_back(c);
			std::reverse(stack.begin() + n0, stack.end());
		}
	}
	ND_PROF(1);
	// ---- landmarks: observations by landmark (counting sort), then per landmark sorted so that every trie subtree is a run; nodes ----
	std::vector<int>& lo_off = S.lo_off; std::vector<int>& lo_idx = S.lo_idx;
	lo_off.assign((size_t)n_lms + 1, 0); lo_idx.resize((size_t)n_obs);
	for (int o = 0; o < n_obs; ++o) lo_off[(size_t)Wn.obs_lm[o] + 1]++;
	for (int l = 0; l < n_lms; ++l) lo_off[(size_t)l + 1] += lo_off[(size_t)l];
	{
		std::vector<int>& next = S.next;
		next.assign(lo_off.begin(), lo_off.end() - 1);
		for (int o = 0; o < n_obs; ++o) lo_idx[(size_t)next[(size_t)Wn.obs_lm[o]]++] = o;
	}
	std::vector<lmk>& lms = S.lms;
	lms.clear(); lms.reserve((size_t)n_lms);
	std::vector<int>& lm_sorted = S.lm_sorted;             // per landmark its observations, sorted (landmark l: [lo_off[l], lo_off[l + 1]))
	lm_sorted.resize((size_t)n_obs);
	std::vector<node>& nodes = S.nodes;
	nodes.clear(); nodes.reserve((size_t)n_obs * 2);
	// landmarks [l0, l1): observations sorted by chain, nodes in the order of their first observation.  WS: the executing thread's workspace (its
	// per-suffix marks), out_lms / out_nodes: where the range's records go (node0 relative to out_nodes' start)
	auto landmarks_range = [&](int l0, int l1, nd_detail::workspace& WS, std::vector<lmk>& out_lms, std::vector<node>& out_nodes) -> bool {
		// node of (current landmark, suffix): node_mark[suffix] == landmark + 1 says node_idx[suffix] is this landmark's
		std::vector<int>& node_mark = WS.node_mark; std::vector<int>& node_idx = WS.node_idx;
		node_mark.assign((size_t)n_suf, 0); node_idx.resize((size_t)n_suf);
		for (int l = l0; l < l1; ++l) {
			const int o0 = lo_off[(size_t)l], no = lo_off[(size_t)l + 1] - o0;
			if (no == 0) continue;                            // never observed: nothing to linearise, nothing moves it
			if (no > 64) return false;
			// stable insertion sort by chain (= by the rank of the observation's suffix)
			int ord[64], key[64];
			for (int k = 0; k < no; ++k) {
				const int o = lo_idx[(size_t)o0 + k], ko = rank[(size_t)obs_suf[(size_t)o]];
				int j = k;
				while (j > 0 && ko < key[j - 1]) { ord[j] = ord[j - 1]; key[j] = key[j - 1]; --j; }
				ord[j] = o; key[j] = ko;
			}
			lmk M{l, o0, no, (int)out_nodes.size(), 0};
			for (int k = 0; k < no; ++k) lm_sorted[(size_t)o0 + k] = ord[k];
			// nodes in the order of their first observation (depth first): a node's observations are the run of those whose chain holds its key
			const int mark = l + 1;
			for (int k = 0; k < no; ++k) {
				const int sid = obs_suf[(size_t)ord[k]], len = suf_len[(size_t)sid];
				const int* ch = &chain[(size_t)sid * MAXP];
				for (int c = 0; c < len; ++c) {
					const int s2 = ch[c];
					if (!suf_opt[(size_t)s2]) continue;
					if (node_mark[(size_t)s2] == mark) { node& nd = out_nodes[(size_t)node_idx[(size_t)s2]]; nd.n = k + 1 - nd.a; }
					else { node_mark[(size_t)s2] = mark; node_idx[(size_t)s2] = (int)out_nodes.size(); out_nodes.push_back(node{s2, k, 1}); }
				}
			}
			M.n_nodes = (int)out_nodes.size() - M.node0;
			if (M.n_nodes > 63) return false;
			out_lms.push_back(M);
		}
		return true;
	};
	// (a single large window with helper threads at hand: ranges of landmarks side by side, merged in order)
	const bool use_team = team && team->n > 1 && team->run && n_obs >= 2048 && n_lms >= 64;
	if (!use_team) {
		if (!landmarks_range(0, n_lms, S, lms, nodes)) return false;
	} else {
		const int n_parts = std::min(team->n, n_lms / 16);
		std::vector<nd_detail::part_lms>& parts = S.p_lms;
		if ((int)parts.size() < n_parts) parts.resize((size_t)n_parts);
		for (int p = 0; p < n_parts; ++p) parts[(size_t)p].clear();
		team->run(n_parts, [&](int p) {
			const int l0 = (int)((long long)n_lms * p / n_parts), l1 = (int)((long long)n_lms * (p + 1) / n_parts);
			parts[(size_t)p].fail = !landmarks_range(l0, l1, nd_detail::ws(), parts[(size_t)p].lms, parts[(size_t)p].nodes);
		});
		for (int p = 0; p < n_parts; ++p) if (parts[(size_t)p].fail) return false;
		for (int p = 0; p < n_parts; ++p) {
			const nd_detail::part_lms& P = parts[(size_t)p];
			const int d = (int)nodes.size();
			for (lmk M : P.lms) { M.node0 += d; lms.push_back(M); }
			nodes.insert(nodes.end(), P.nodes.begin(), P.nodes.end());
		}
	}
	const int n_sorted = n_obs;                            // (every observation belongs to an observed landmark)
	{
		std::vector<int>& ord = S.ord;
		ord.resize(lms.size());
		if (!split) {
			// optimised landmarks first, in slot order, then the fixed ones (by landmark index): the keys are distinct
			std::vector<uint64_t>& okey = S.okey;
			okey.resize(lms.size());
			for (size_t m = 0; m < lms.size(); ++m) {
				const int sa = Wn.lm_unknown_slot[lms[m].l];
				okey[m] = ((uint64_t)(sa >= 0 ? (uint32_t)sa : 0x80000000u + (uint32_t)lms[m].l) << 32) | (uint32_t)m;
			}
			std::sort(okey.begin(), okey.end());
			for (size_t m = 0; m < ord.size(); ++m) ord[m] = (int)(uint32_t)okey[m];
		} else {
			// by the sorted set of optimised edge slots of the landmark's nodes, lexicographically (a prefix first; ties: landmark index).  As ONE key
			// per landmark: slot e -> bit 127 - e of a 128-bit number X; of two sets that differ first at list position i the one with the smaller
			// element there has the larger X, and a proper prefix has the smaller X -- with every bit below X's lowest set bit filled (Y) the prefix
			// comes out >= its extensions (equal only when the extension adds ALL slots up to 127: then the fewer elements first).  Order: Y
			// descending, element count ascending, landmark index ascending (eset_less above is the definition; tests compare the tables' hashes).
			std::vector<nd_detail::eset>& es = S.es;
			es.assign(lms.size(), nd_detail::eset());
			for (size_t m = 0; m < lms.size(); ++m)
				for (int i = 0; i < lms[m].n_nodes; ++i) {
					const int e = suf_es[(size_t)nodes[(size_t)lms[m].node0 + i].key];
					if (e < 64) es[m].lo |= (uint64_t)1 << e; else es[m].hi |= (uint64_t)1 << (e - 64);
				}
			using nd_detail::skey;
			std::vector<skey>& sk = S.sk;
			sk.resize(lms.size());
			auto rev64 = [](uint64_t v) {
				v = ((v >> 1) & 0x5555555555555555ull) | ((v & 0x5555555555555555ull) << 1);
				v = ((v >> 2) & 0x3333333333333333ull) | ((v & 0x3333333333333333ull) << 2);
				v = ((v >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((v & 0x0F0F0F0F0F0F0F0Full) << 4);
				return __builtin_bswap64(v);
			};
			for (size_t m = 0; m < lms.size(); ++m) {
				// slot e (bit e of lo / bit e - 64 of hi) -> bit 127 - e: the high word is the reversed lo, the low word the reversed hi
				uint64_t xh = rev64(es[m].lo), xl = rev64(es[m].hi);
				if (xl) xl |= (xl & (~xl + 1)) - 1;
				else if (xh) { xh |= (xh & (~xh + 1)) - 1; xl = ~(uint64_t)0; }
				else { xh = ~(uint64_t)0; xl = ~(uint64_t)0; }
				sk[m] = skey{xh, xl, __builtin_popcountll(es[m].lo) + __builtin_popcountll(es[m].hi), lms[m].l, (int)m};
			}
			std::sort(sk.begin(), sk.end(), [](const skey& a, const skey& b) {
				if (a.yh != b.yh) return a.yh > b.yh;
				if (a.yl != b.yl) return a.yl > b.yl;
				if (a.pc != b.pc) return a.pc < b.pc;
				return a.l < b.l;
			});
			for (size_t m = 0; m < ord.size(); ++m) ord[m] = sk[m].m;
		}
		std::vector<lmk>& sorted = S.lms2;
		sorted.resize(lms.size());
		for (size_t m = 0; m < ord.size(); ++m) sorted[m] = lms[(size_t)ord[m]];
		lms.swap(sorted);
	}
	ND_PROF(2);
	// ---- groups: greedy runs of landmarks with <= 64 observations and <= 64 lanes ----
	hdr.n_edges = Wn.n_edges_total; hdr.E = E; hdr.n_lms = n_lms; hdr.L = L; hdr.n_obs = n_obs; hdr.n_suf = n_suf;
	hdr.suf0 = (int)H.suf.size(); hdr.tile0 = (int)H.tile.size(); hdr.grp0 = (int)H.group.size();
	hdr.obs0 = (int)H.obs.size(); hdr.lane0 = (int)H.lane.size();
	hdr.item0 = (int)H.item.size(); hdr.qs0 = (int)H.qs.size(); hdr.gitem0 = (int)H.gitem.size(); hdr.gqs0 = (int)H.gqs.size();
	hdr.tab0 = H.tab_doubles;
	hdr.hb0 = H.hb_doubles;
	// (everything is appended to H directly; a window that turns out not to fit is rolled back)
	const size_t r_suf = H.suf.size(), r_tile = H.tile.size(), r_group = H.group.size(), r_obs = H.obs.size(), r_px = H.px.size(), r_lane = H.lane.size(),
	             r_item = H.item.size(), r_qs = H.qs.size(), r_gitem = H.gitem.size(), r_gqs = H.gqs.size(), r_sqs = H.sqs.size(), r_poff = H.poff.size(),
	             r_goff = H.goff.size(), r_gslot = H.gslot.size();
	auto rollback = [&] {
		H.suf.resize(r_suf); H.tile.resize(r_tile); H.group.resize(r_group); H.obs.resize(r_obs); H.px.resize(r_px); H.lane.resize(r_lane);
		H.item.resize(r_item); H.qs.resize(r_qs); H.gitem.resize(r_gitem); H.gqs.resize(r_gqs); H.sqs.resize(r_sqs); H.poff.resize(r_poff);
		H.goff.resize(r_goff); H.gslot.resize(r_gslot);
		return false;
	};
	std::vector<gnode>& glanes = S.glanes;                 // all groups' lanes, group after group (H.group[...].lane0 / n_lanes index it)
	const size_t n_lanes_all = nodes.size() + lms.size();
	glanes.resize(n_lanes_all);
	// (the tables' final sizes are known here: the observation, pixel and lane records are written in place)
	H.obs.resize(r_obs + (size_t)n_sorted); H.px.resize(r_px + (size_t)n_sorted); H.lane.resize(r_lane + n_lanes_all);
	H.group.reserve(H.group.size() + n_lanes_all / 32 + (size_t)n_sorted / 32 + 4);
	{
		nd_i2* w_obs = H.obs.data() + r_obs; nd_f4* w_px = H.px.data() + r_px; nd_i4* w_lane = H.lane.data() + r_lane;
		gnode* w_gl = glanes.data();
		const nd_f4* in_px = reinterpret_cast<const nd_f4*>(Wn.obs_px);
		nd_group G{0, 0, 0, 0};
		auto flush = [&] {
			if (G.n_lanes == 0) return;
			H.group.push_back(G);
			G.obs0 += G.n_obs; G.lane0 += G.n_lanes; G.n_obs = 0; G.n_lanes = 0;
		};
		for (size_t m = 0; m < lms.size(); ++m) {
			const lmk& M = lms[m];
			const int no = M.n_obs, nl = 1 + M.n_nodes;
			if (G.n_obs + no > 64 || G.n_lanes + nl > 64) flush();
			const int root_lane = G.n_lanes, obs_a = G.n_obs;
			const int* so = &lm_sorted[(size_t)M.obs0];
			for (int k = 0; k < no; ++k) {
				const int o = so[k];
				w_obs[k] = nd_i2{M.l, obs_suf[(size_t)o] | (root_lane << 16) | (k << 24)};      // (k: distance from the landmark's first observation lane)
				std::memcpy(&w_px[k], &in_px[o], sizeof(nd_f4));                               // (the caller's array need not be 16-byte aligned)
			}
			w_obs += no; w_px += no;
			*w_lane++ = nd_i4{M.l, obs_a | (no << 8) | (1 << 16), Wn.lm_unknown_slot[M.l], 1 | (root_lane << 8) | (M.n_nodes << 16)};
			*w_gl++ = gnode{(int)m, -1, root_lane};
			const node* nd0 = &nodes[(size_t)M.node0];
			for (int k = 0; k < M.n_nodes; ++k) {
				const node& nd = nd0[k];
				const int code = suf[(size_t)nd.key].x, inv = code & 1;
				const int tsuf = inv ? suf[(size_t)nd.key].y : nd.key;
				const int es = suf_es[(size_t)nd.key];
				*w_lane++ = nd_i4{M.l, (obs_a + nd.a) | (nd.n << 8) | ((nd.a == 0 ? 1 : 0) << 16), tsuf | (es << 16) | (inv << 24), 2 | (root_lane << 8)};
				*w_gl++ = gnode{(int)m, k, root_lane + 1 + k};
			}
			G.n_obs += no; G.n_lanes += nl;
		}
		flush();
	}
	const size_t n_groups = H.group.size() - r_group;
	ND_PROF(3);
	// ---- tiles: `waves` groups each; the items of a tile sorted by 6x6 block (counting sort over the blocks the tile touches) ----
	const int NQ = waves * 16;
	const int n_pairs = E * (E + 1) / 2;
	auto pid_of = [](int e, int e2) { return e2 * (e2 + 1) / 2 + e; };   // e <= e2
	const size_t n_tiles = (n_groups + (size_t)waves - 1) / (size_t)waves;
	std::vector<char>& is_diag = S.is_diag;                // block (e, e): the only ones the lambda_0 pass needs (bit 30 of a lane slot's x)
	is_diag.assign((size_t)n_pairs + 1, 0);
	for (int e = 0; e < E; ++e) is_diag[(size_t)pid_of(e, e)] = 1;
	std::vector<int>& tri = S.tri;                         // tri[e2] = e2 (e2 + 1) / 2: block (e, e2), e <= e2, is tri[e2] + e
	tri.resize((size_t)E + 1);
	for (int e = 0; e <= E; ++e) tri[(size_t)e] = e * (e + 1) / 2;
	int max_per_lane = 0, max_rounds = 0, max_tile_blocks = 0;
	{
		// an upper estimate of the item count: a landmark with k nodes has at most k (k + 1) / 2 pairs (+ the doubled same-edge pairs: rare)
		size_t est = 0;
		for (const lmk& M : lms) est += (size_t)M.n_nodes * ((size_t)M.n_nodes + 1) / 2;
		H.item.reserve(H.item.size() + est + est / 8 + 64);
		H.gitem.reserve(H.gitem.size() + nodes.size());
		H.tile.reserve(H.tile.size() + n_tiles); H.gqs.reserve(H.gqs.size() + n_tiles * (size_t)std::max(E, 1));
		if (split) H.sqs.reserve(H.sqs.size() + n_tiles * (size_t)std::min(n_pairs, 4 * waves * 64));
		else H.qs.reserve(H.qs.size() + n_tiles * (size_t)NQ * 2);
	}
	// tiles [tile0, tile1) into the tables of O (any vector types; o_r_*: their sizes where the window's -- or the part's -- entries start; the offsets
	// stored in the tables are relative to those).  WS: the executing thread's workspace.  Returns false when a table entry would overflow its field.
	auto tiles_range = [&](size_t tile0, size_t tile1, nd_detail::workspace& WS, auto& o_sqs, auto& o_qs, auto& o_gqs, auto& o_item, auto& o_gitem, auto& o_tile,
	                       const size_t o_r_sqs, const size_t o_r_qs, const size_t o_r_gqs, const size_t o_r_item, const size_t o_r_gitem,
	                       int& max_per_lane, int& max_rounds, int& max_tile_blocks) -> bool {
	std::vector<int>& t_pid = WS.t_pid; std::vector<int>& t_code = WS.t_code;      // the current tile's items, generation order
	std::vector<int>& cnt = WS.cnt; std::vector<int>& start = WS.start; std::vector<int>& fillp = WS.fillp; std::vector<int>& used = WS.used; std::vector<int>& sorted_code = WS.sorted_code;
	std::vector<int>& e_cnt = WS.e_cnt; std::vector<int>& e_fill = WS.e_fill; std::vector<int>& e_slot = WS.e_slot; std::vector<int>& e_of = WS.e_of;      // the tile's node slots by edge
	cnt.assign((size_t)n_pairs + 1, 0); start.resize((size_t)n_pairs + 1); fillp.resize((size_t)n_pairs + 1);
	e_cnt.assign((size_t)E + 1, 0); e_fill.resize((size_t)E + 1);
	for (size_t t = tile0; t < tile1; ++t) {
		const size_t g0 = t * (size_t)waves;
		const int n_grp = (int)std::min(n_groups - g0, (size_t)waves);
		t_pid.clear(); t_code.clear(); e_slot.clear(); e_of.clear(); used.clear();
		for (int wv = 0; wv < n_grp; ++wv) {
			const nd_group& Gr = H.group[r_group + g0 + (size_t)wv];
			const gnode* gl = &glanes[(size_t)Gr.lane0];
			const int nl = Gr.n_lanes;
			// per node lane: its edge slot and its run of the landmark's sorted observations [ra, rb).  Two nodes of a landmark are NESTED (one's suffix is a
			// suffix of the other's) exactly when their runs meet -- both suffixes then lie on one observation's chain --, and of a nested pair the one
			// LATER in the lane order is the far one (nodes are listed by first observation, an observation's chain from the shortest suffix up).
			int l_es[64], l_ra[64], l_rb[64];
			for (int a = 0; a < nl; ++a) {
				if (gl[a].node < 0) continue;
				const node& nd = nodes[(size_t)lms[(size_t)gl[a].lmk].node0 + (size_t)gl[a].node];
				l_es[a] = suf_es[(size_t)nd.key]; l_ra[a] = nd.a; l_rb[a] = nd.a + nd.n;
			}
			// item (row r, column c) of a node pair: the far node of a nested pair gives Zh, the near one Ph (bit 9 / bit 19: the P part of the row's / the
			// column's record); not nested: Zh, Zh and the sign bit (20).  Row = the node of the lower edge slot.
			auto emit = [&](int pid, int code) {
				if (cnt[(size_t)pid]++ == 0) used.push_back(pid);
				t_pid.push_back(pid);
				t_code.push_back(code);
			};
			for (int r0 = 0; r0 < nl;) {                           // a root lane, then its landmark's node lanes
				const lmk& M = lms[(size_t)gl[r0].lmk];
				const int a0 = r0 + 1, a1 = a0 + M.n_nodes;
				const bool opt = Wn.lm_unknown_slot[M.l] >= 0;
				const int sb = 64 * wv;
				for (int a = a0; a < a1; ++a) {
					const int ea = l_es[a], rb_a = l_rb[a];
					const int slot_a = sb + a;                       // (a lane's slot in its group is its lane number)
					e_slot.push_back(slot_a); e_of.push_back(ea);
					// the node with itself: nested, "a far": Zh_a Ph_a^T on the diagonal block
					emit(tri[(size_t)ea] + ea, slot_a | (slot_a << 10) | (1 << 19));
					// partners: the later nodes of the same landmark
					for (int b = a + 1; b < a1; ++b) {
						const int eb = l_es[b], slot_b = sb + b;
						if (l_ra[b] < rb_a) {                        // nested, b far: b's record gives Zh, a's Ph
							if (ea < eb) emit(tri[(size_t)eb] + ea, slot_a | (1 << 9) | (slot_b << 10));
							else if (eb < ea) emit(tri[(size_t)ea] + eb, slot_b | (slot_a << 10) | (1 << 19));
							else {                                   // two nodes of one landmark on the same edge: both orders go to the diagonal block
								emit(tri[(size_t)ea] + ea, slot_a | (1 << 9) | (slot_b << 10));
								emit(tri[(size_t)ea] + ea, slot_b | (slot_a << 10) | (1 << 19));
							}
						} else if (opt) {                            // not nested: the Schur term only
							if (ea < eb) emit(tri[(size_t)eb] + ea, slot_a | (slot_b << 10) | (1 << 20));
							else if (eb < ea) emit(tri[(size_t)ea] + eb, slot_b | (slot_a << 10) | (1 << 20));
							else {
								emit(tri[(size_t)ea] + ea, slot_a | (slot_b << 10) | (1 << 20));
								emit(tri[(size_t)ea] + ea, slot_b | (slot_a << 10) | (1 << 20));
							}
						}
					}
				}
				r0 = a1;
			}
		}
		// heaviest block first (ties: first appearance): the lanes of a wavefront run in lockstep.  A counting sort over the counts (a comparison
		// sort of a depth-5 tile's ~500 blocks, 39 tiles a window, was a third of this section)
		if (used.size() > 1) {
			int cmax = 0;
			for (int p2 : used) cmax = std::max(cmax, cnt[(size_t)p2]);
			std::vector<int>& bk = WS.bucket; std::vector<int>& tmp = WS.used2;
			bk.assign((size_t)cmax + 2, 0);
			for (int p2 : used) bk[(size_t)(cmax - cnt[(size_t)p2]) + 1]++;
			for (int c = 0; c <= cmax; ++c) bk[(size_t)c + 1] += bk[(size_t)c];
			tmp.resize(used.size());
			for (int p2 : used) tmp[(size_t)bk[(size_t)(cmax - cnt[(size_t)p2])]++] = p2;
			used.swap(tmp);
		}
		{
			int pos = 0;
			for (int p2 : used) { start[(size_t)p2] = pos; fillp[(size_t)p2] = pos; pos += cnt[(size_t)p2]; }
			sorted_code.resize(t_code.size());
			for (size_t i = 0; i < t_pid.size(); ++i) sorted_code[(size_t)fillp[(size_t)t_pid[i]]++] = t_code[i];
		}
		nd_tile T{};
		T.grp0 = (int)g0; T.n_grp = n_grp;
		const int item_base = (int)(o_item.size() - o_r_item);
		if (split) {
			// LANE slots: a block gets 1, 2 or 4 lanes (its items dealt out among them, the lanes' sums joined by one or two DPP steps), the fewest
			// that keep a lane's run <= R items; R = the shortest run length whose lanes fit the fewest rounds of the workgroup.  Heaviest blocks first,
			// so the 4-lane blocks come first, then the 2-lane ones: every group is aligned and a wavefront's lanes walk runs of similar length.
			const int NT_ = waves * 64;
			const int rounds_min = std::max(1, ((int)used.size() + NT_ - 1) / NT_);
			auto lanes_of = [](int c, int R) { return c <= R ? 1 : (c <= 2 * R ? 2 : 4); };
			int R = 1;
			for (;; ++R) {
				long long lanes = 0;
				for (int p2 : used) lanes += lanes_of(cnt[(size_t)p2], R);
				if (lanes <= (long long)NT_ * rounds_min) break;
			}
			T.qs0 = (int)(o_sqs.size() - o_r_sqs);
			max_tile_blocks = std::max(max_tile_blocks, (int)used.size());
			for (int p2 : used) {
				const int c = cnt[(size_t)p2];
				const int k = lanes_of(c, R), kc = k == 4 ? 2 : (k == 2 ? 1 : 0);
				if (c > 0xFFFF) return false;
				for (int r = 0; r < k; ++r)
					o_sqs.push_back(nd_i4{p2 | (is_diag[(size_t)p2] ? 1 << 30 : 0), c | (kc << 16) | (r << 20), item_base + start[(size_t)p2], -1});
			}
			o_item.insert(o_item.end(), sorted_code.begin(), sorted_code.end());
			T.n_rounds = (int)(o_sqs.size() - o_r_sqs) - T.qs0;     // (split form: n_rounds = number of lane slots of the tile)
			// bit 29: this wavefront (64 consecutive lane slots) holds blocks of several lanes
			for (size_t w0 = o_r_sqs + (size_t)T.qs0; w0 < o_sqs.size(); w0 += 64) {
				const size_t w1 = std::min(w0 + 64, o_sqs.size());
				bool multi = false;
				for (size_t i = w0; i < w1; ++i) multi |= ((o_sqs[i].y >> 16) & 3) != 0;
				if (multi) for (size_t i = w0; i < w1; ++i) o_sqs[i].x |= 1 << 29;
			}
		} else {
			// quad slots of a tile: a block per quad, heaviest first (the 16 quads of a wavefront run in lockstep: similar loads share one); with quads
			// to spare the heaviest blocks are cut in two parts (items dealt out alternately) that sit in the two quads of an aligned lane octet
			// and are added up by a DPP mirror step before the block's accumulator takes them
			int n_split = 0;
			if ((int)used.size() < NQ) {
				n_split = std::min((NQ - (int)used.size()), (int)used.size());
				while (n_split > 0 && cnt[(size_t)used[(size_t)n_split - 1]] < 8) --n_split;       // (a short run is not worth a second quad)
			}
			const int n_slots = (int)used.size() + n_split;
			T.qs0 = (int)(o_qs.size() - o_r_qs);
			T.n_rounds = (n_slots + NQ - 1) / NQ;
			max_rounds = std::max(max_rounds, T.n_rounds);
			for (int k = 0; k < T.n_rounds * NQ; ++k) {
				const int wave_flag = (k / 16) * 16 < 2 * n_split ? (int)0x80000000u : 0;      // this wavefront holds blocks cut in two
				if (k >= n_slots) { o_qs.push_back(nd_i2{0xFFFF, wave_flag}); continue; }
				const int pid = k < 2 * n_split ? used[(size_t)(k >> 1)] : used[(size_t)(k - n_split)];
				const int part = k < 2 * n_split ? (k & 1) : 0, parts = k < 2 * n_split ? 2 : 1;
				const int first = (int)(o_item.size() - o_r_item);
				const int* v = &sorted_code[(size_t)start[(size_t)pid]];
				const int c = cnt[(size_t)pid];
				if (parts == 1) o_item.insert(o_item.end(), v, v + c);
				else for (int i = part; i < c; i += parts) o_item.push_back(v[i]);
				const int n = (int)(o_item.size() - o_r_item) - first;
				if (n > 32767 || first >= (1 << 28)) return false;
				o_qs.push_back(nd_i2{pid | (n << 16), first | (part << 28) | (parts == 2 ? 1 << 30 : 0) | wave_flag});
				max_per_lane = std::max(max_per_lane, (n + 3) / 4);
			}
		}
		for (int p2 : used) cnt[(size_t)p2] = 0;
		// gradient slots: the tile's node slots by edge (edge order; inside an edge the lane order)
		T.gqs0 = (int)(o_gqs.size() - o_r_gqs); T.n_gqs = 0;
		{
			std::fill(e_cnt.begin(), e_cnt.end(), 0);
			for (int e : e_of) e_cnt[(size_t)e]++;
			int pos = (int)(o_gitem.size() - o_r_gitem);
			const size_t base = o_gitem.size();
			o_gitem.resize(base + e_slot.size());
			for (int e = 0; e < E; ++e) {
				if (e_cnt[(size_t)e] == 0) continue;
				o_gqs.push_back(nd_i2{e | (e_cnt[(size_t)e] << 16), pos});
				e_fill[(size_t)e] = pos - (int)(base - o_r_gitem);      // (position inside this tile's part)
				pos += e_cnt[(size_t)e];
				++T.n_gqs;
			}
			for (size_t i = 0; i < e_slot.size(); ++i) o_gitem[base + (size_t)e_fill[(size_t)e_of[i]]++] = e_slot[i];
		}
		o_tile.push_back(T);
	}
	return true;
	};
	if (!use_team || n_tiles < 8) {
		if (!tiles_range(0, n_tiles, S, H.sqs, H.qs, H.gqs, H.item, H.gitem, H.tile, r_sqs, r_qs, r_gqs, r_item, r_gitem, max_per_lane, max_rounds, max_tile_blocks)) return rollback();
	} else {
		// ranges of tiles side by side into tables of their own, then appended in order: the offsets a part stored are relative to ITS start
		const int n_parts = (int)std::min<size_t>((size_t)team->n, n_tiles / 2);
		std::vector<nd_detail::part_tiles>& parts = S.p_tiles;
		if ((int)parts.size() < n_parts) parts.resize((size_t)n_parts);
		for (int p = 0; p < n_parts; ++p) parts[(size_t)p].clear();
		team->run(n_parts, [&](int p) {
			nd_detail::part_tiles& P = parts[(size_t)p];
			const size_t tile0 = n_tiles * (size_t)p / (size_t)n_parts, tile1 = n_tiles * ((size_t)p + 1) / (size_t)n_parts;
			P.fail = !tiles_range(tile0, tile1, nd_detail::ws(), P.sqs, P.qs, P.gqs, P.item, P.gitem, P.tile, 0, 0, 0, 0, 0, P.max_per_lane, P.max_rounds, P.max_tile_blocks);
		});
		for (int p = 0; p < n_parts; ++p) if (parts[(size_t)p].fail) return rollback();
		for (int p = 0; p < n_parts; ++p) {
			const nd_detail::part_tiles& P = parts[(size_t)p];
			const int d_sqs = (int)(H.sqs.size() - r_sqs), d_qs = (int)(H.qs.size() - r_qs), d_gqs = (int)(H.gqs.size() - r_gqs), d_item = (int)(H.item.size() - r_item),
			          d_gitem = (int)(H.gitem.size() - r_gitem);
			for (nd_tile T : P.tile) { T.qs0 += split ? d_sqs : d_qs; T.gqs0 += d_gqs; H.tile.push_back(T); }
			for (nd_i4 q : P.sqs) { q.z += d_item; H.sqs.push_back(q); }
			for (nd_i2 q : P.qs) {
				if ((q.x & 0xFFFF) != 0xFFFF) {
					const int first = (q.y & 0x0FFFFFFF) + d_item;
					if (first >= (1 << 28)) return rollback();
					q.y = (q.y & ~0x0FFFFFFF) | first;
				}
				H.qs.push_back(q);
			}
			for (nd_i2 q : P.gqs) { q.y += d_gitem; H.gqs.push_back(q); }
			H.item.insert(H.item.end(), P.item.begin(), P.item.end());
			H.gitem.insert(H.gitem.end(), P.gitem.begin(), P.gitem.end());
			max_per_lane = std::max(max_per_lane, P.max_per_lane); max_rounds = std::max(max_rounds, P.max_rounds); max_tile_blocks = std::max(max_tile_blocks, P.max_tile_blocks);
		}
	}
	int n_part = 0, n_gpart = 0;
	if (split) {
		// partial slots, block-major (a block's slots are contiguous, in tile order); gradient slots, edge-major
		const size_t p0 = H.poff.size(), g0 = H.goff.size();
		H.poff.resize(p0 + (size_t)n_pairs + 1);
		int* w_poff = &H.poff[p0];
		std::fill(w_poff, w_poff + n_pairs + 1, 0);
		for (size_t i = r_sqs; i < H.sqs.size(); ++i) if (((H.sqs[i].y >> 20) & 3) == 0) w_poff[(size_t)(H.sqs[i].x & 0xFFFF) + 1]++;
		for (int p2 = 0; p2 < n_pairs; ++p2) w_poff[(size_t)p2 + 1] += w_poff[(size_t)p2];
		{
			std::vector<int>& next = S.next;
			next.assign(w_poff, w_poff + n_pairs);
			for (size_t i = r_sqs; i < H.sqs.size(); ++i) {
				nd_i4& q = H.sqs[i];
				q.w = ((q.y >> 20) & 3) == 0 ? next[(size_t)(q.x & 0xFFFF)]++ : next[(size_t)(q.x & 0xFFFF)] - 1;      // (the lanes of a block follow its first one)
			}
		}
		n_part = w_poff[(size_t)n_pairs];
		H.goff.resize(g0 + (size_t)E + 1);
		int* w_goff = &H.goff[g0];
		std::fill(w_goff, w_goff + E + 1, 0);
		for (size_t i = r_gqs; i < H.gqs.size(); ++i) w_goff[(size_t)(H.gqs[i].x & 0xFFFF) + 1]++;
		for (int e = 0; e < E; ++e) w_goff[(size_t)e + 1] += w_goff[(size_t)e];
		{
			std::vector<int>& next = S.next;
			next.assign(w_goff, w_goff + E);
			H.gslot.resize(r_gslot + (H.gqs.size() - r_gqs));
			for (size_t k = r_gqs; k < H.gqs.size(); ++k) H.gslot[r_gslot + (k - r_gqs)] = next[(size_t)(H.gqs[k].x & 0xFFFF)]++;
		}
		n_gpart = (int)(H.gqs.size() - r_gqs);
	}
	hdr.n_rounds = max_rounds; hdr.max_parts = 2;
	hdr.n_tiles = (int)n_tiles; hdr.n_groups = (int)n_groups;
	ND_PROF(4);
	// ---- commit ----
	H.suf.insert(H.suf.end(), suf.begin(), suf.end());
	H.tab_doubles += 2LL * 12 * n_suf;
	H.hb_doubles += 36LL * E * (E + 1) / 2;
	H.lds_doubles = std::max(H.lds_doubles, nd_lds_doubles(Wn.n_edges_total, E, waves, n_suf, (int)n_tiles, (int)n_groups));
	H.lds_doubles_big = std::max(H.lds_doubles_big, nd_lds_doubles(Wn.n_edges_total, E, waves, n_suf, (int)n_tiles, (int)n_groups, true, lds_limit_doubles));
	H.max_items_per_lane = std::max(H.max_items_per_lane, max_per_lane);
	if (split) {
		nd_split Sp{};
		Sp.sqs0 = (int)r_sqs; Sp.poff0 = (int)r_poff; Sp.goff0 = (int)r_goff; Sp.gslot0 = (int)r_gslot;
		Sp.n_part = n_part; Sp.n_gpart = n_gpart; Sp.part0 = H.part_slots; Sp.gpart0 = H.gpart_slots;
		Sp.gtile0 = (int)r_tile;
		H.part_slots += n_part; H.gpart_slots += n_gpart;
		H.max_tile_blocks = std::max(H.max_tile_blocks, max_tile_blocks);
		H.swin.push_back(Sp);
	}
	H.win.push_back(hdr);
	ND_PROF(5);
	return true;
}

// Appends the tables of ONE window (built into a fresh nd_host by nd_build_window with zero bases) to a batch's tables: every window-level base moves
// by the batch's sizes; indices stored inside the tables are window relative and stay.  Windows can so be built side by side (one thread each).
inline void nd_append(nd_host& D, const nd_host& W1)
{
	nd_win w = W1.win[0];
	w.suf0 += (int)D.suf.size(); w.tile0 += (int)D.tile.size(); w.grp0 += (int)D.group.size(); w.obs0 += (int)D.obs.size(); w.lane0 += (int)D.lane.size();
	w.item0 += (int)D.item.size(); w.qs0 += (int)D.qs.size(); w.gitem0 += (int)D.gitem.size(); w.gqs0 += (int)D.gqs.size();
	w.tab0 += D.tab_doubles; w.hb0 += D.hb_doubles;
	if (!W1.swin.empty()) {
		nd_split s = W1.swin[0];
		s.sqs0 += (int)D.sqs.size(); s.poff0 += (int)D.poff.size(); s.goff0 += (int)D.goff.size(); s.gslot0 += (int)D.gslot.size();
		s.part0 += D.part_slots; s.gpart0 += D.gpart_slots; s.gtile0 += (int)D.tile.size();
		D.swin.push_back(s);
	}
	D.win.push_back(w);
#define ND_APP(v) D.v.insert(D.v.end(), W1.v.begin(), W1.v.end())
	ND_APP(suf); ND_APP(tile); ND_APP(group); ND_APP(obs); ND_APP(px); ND_APP(lane); ND_APP(item); ND_APP(qs); ND_APP(gitem); ND_APP(gqs);
	ND_APP(sqs); ND_APP(poff); ND_APP(goff); ND_APP(gslot);
#undef ND_APP
	D.tab_doubles += W1.tab_doubles; D.hb_doubles += W1.hb_doubles; D.part_slots += W1.part_slots; D.gpart_slots += W1.gpart_slots;
	D.lds_doubles = std::max(D.lds_doubles, W1.lds_doubles); D.lds_doubles_big = std::max(D.lds_doubles_big, W1.lds_doubles_big);
	D.max_items_per_lane = std::max(D.max_items_per_lane, W1.max_items_per_lane); D.max_tile_blocks = std::max(D.max_tile_blocks, W1.max_tile_blocks);
}
// a window of the batch that another kernel family solves: an empty header keeps the family's window arrays indexed by the batch's window number
inline void nd_append_placeholder(nd_host& D, bool split)
{
	D.win.push_back(nd_win{});
	if (split) D.swin.push_back(nd_split{});
}

// Sizes of a one-window nd_host, and its placement into pre-sized batch tables at given offsets (the parallel form of nd_append: every window of
// a batch can be placed by its own thread once the offsets -- prefix sums of the sizes -- are known).
struct nd_sizes { size_t suf = 0, tile = 0, group = 0, obs = 0, lane = 0, item = 0, qs = 0, gitem = 0, gqs = 0, sqs = 0, poff = 0, goff = 0, gslot = 0; long long tab = 0, hb = 0, part = 0, gpart = 0; };
inline nd_sizes nd_sizes_of(const nd_host& W1)
{
	nd_sizes z;
	z.suf = W1.suf.size(); z.tile = W1.tile.size(); z.group = W1.group.size(); z.obs = W1.obs.size(); z.lane = W1.lane.size(); z.item = W1.item.size();
	z.qs = W1.qs.size(); z.gitem = W1.gitem.size(); z.gqs = W1.gqs.size(); z.sqs = W1.sqs.size(); z.poff = W1.poff.size(); z.goff = W1.goff.size();
	z.gslot = W1.gslot.size(); z.tab = W1.tab_doubles; z.hb = W1.hb_doubles; z.part = W1.part_slots; z.gpart = W1.gpart_slots;
	return z;
}
inline void nd_sizes_add(nd_sizes& a, const nd_sizes& b)
{
	a.suf += b.suf; a.tile += b.tile; a.group += b.group; a.obs += b.obs; a.lane += b.lane; a.item += b.item; a.qs += b.qs; a.gitem += b.gitem; a.gqs += b.gqs;
	a.sqs += b.sqs; a.poff += b.poff; a.goff += b.goff; a.gslot += b.gslot; a.tab += b.tab; a.hb += b.hb; a.part += b.part; a.gpart += b.gpart;
}
inline void nd_resize(nd_host& D, const nd_sizes& z, size_t n_windows, bool split, nd_bump* arena = nullptr)
{
	D.win.assign(n_windows, nd_win{});
	if (split) D.swin.assign(n_windows, nd_split{});
	if (arena) {
#define ND_BIND(v) D.v = decltype(D.v)(typename decltype(D.v)::allocator_type(arena))
		ND_BIND(suf); ND_BIND(tile); ND_BIND(group); ND_BIND(obs); ND_BIND(px); ND_BIND(lane); ND_BIND(item); ND_BIND(qs); ND_BIND(gitem); ND_BIND(gqs);
		ND_BIND(sqs); ND_BIND(poff); ND_BIND(goff); ND_BIND(gslot);
#undef ND_BIND
	}
	D.suf.resize(z.suf); D.tile.resize(z.tile); D.group.resize(z.group); D.obs.resize(z.obs); D.px.resize(z.obs); D.lane.resize(z.lane); D.item.resize(z.item);
	D.qs.resize(z.qs); D.gitem.resize(z.gitem); D.gqs.resize(z.gqs); D.sqs.resize(z.sqs); D.poff.resize(z.poff); D.goff.resize(z.goff); D.gslot.resize(z.gslot);
	D.tab_doubles = z.tab; D.hb_doubles = z.hb; D.part_slots = z.part; D.gpart_slots = z.gpart;
}
// window `wi` of the batch: tables of W1 to the offsets `o` of D (thread safe for distinct windows; D's running maxima are the caller's)
inline void nd_place(nd_host& D, size_t wi, const nd_sizes& o, const nd_host& W1)
{
	nd_win w = W1.win[0];
	w.suf0 += (int)o.suf; w.tile0 += (int)o.tile; w.grp0 += (int)o.group; w.obs0 += (int)o.obs; w.lane0 += (int)o.lane;
	w.item0 += (int)o.item; w.qs0 += (int)o.qs; w.gitem0 += (int)o.gitem; w.gqs0 += (int)o.gqs;
	w.tab0 += o.tab; w.hb0 += o.hb;
	if (!W1.swin.empty()) {
		nd_split s = W1.swin[0];
		s.sqs0 += (int)o.sqs; s.poff0 += (int)o.poff; s.goff0 += (int)o.goff; s.gslot0 += (int)o.gslot;
		s.part0 += o.part; s.gpart0 += o.gpart; s.gtile0 += (int)o.tile;
		D.swin[wi] = s;
	}
	D.win[wi] = w;
#define ND_PUT(v, off) if (!W1.v.empty()) std::memcpy(&D.v[off], W1.v.data(), sizeof(W1.v[0]) * W1.v.size())
	ND_PUT(suf, o.suf); ND_PUT(tile, o.tile); ND_PUT(group, o.group); ND_PUT(obs, o.obs); ND_PUT(px, o.obs); ND_PUT(lane, o.lane); ND_PUT(item, o.item);
	ND_PUT(qs, o.qs); ND_PUT(gitem, o.gitem); ND_PUT(gqs, o.gqs); ND_PUT(sqs, o.sqs); ND_PUT(poff, o.poff); ND_PUT(goff, o.goff); ND_PUT(gslot, o.gslot);
#undef ND_PUT
}
