// srba_stereo_slam.cpp -- see srba_stereo_slam.h.  SRBA.h = reference src/srba-stereo-slam.h.
#include "srba_stereo_slam.h"
#include <memory>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <functional>
#include <limits>
#include <queue>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <exception>
#include <cstring>
#include <cmath>

namespace ssk {

mySRBA::mySRBA(Context& ctx, const TSRBAStereoSLAMOptions& o) : m_ctx(ctx), m_opts(o)
{
	// CSRBAStereoSLAMEstimator::initialize, EST.cpp:1149-1160
	parameters.use_robust_kernel = o.srba_use_robust_kernel;
	parameters.use_robust_kernel_stage1 = o.srba_use_robust_kernel_stage1;
	parameters.kernel_param = o.srba_kernel_param;
	parameters.max_tree_depth = o.srba_max_tree_depth;
	parameters.max_optimize_depth = o.srba_max_optimize_depth;
	parameters.submap_size = o.srba_submap_size;
	parameters.std_noise_observations = 0.5;
	parameters.literal_lc_edge_init = o.srba_literal_lc_edge_init;
}

size_t mySRBA::create_kf2kf_edge(TKeyFrameID from, TKeyFrameID to)
{
	k2k_edge_t e;
	e.from = from; e.to = to;
	m_state.k2k_edges.push_back(e);
	const size_t id = m_state.k2k_edges.size() - 1;
	if (m_adj.size() <= std::max(from, to)) m_adj.resize(std::max(from, to) + 1);
	m_adj[from].push_back(std::make_pair(to, id));
	m_adj[to].push_back(std::make_pair(from, id));
	return id;
}

void mySRBA::bfs(TKeyFrameID root, size_t max_depth, std::map<TKeyFrameID, step_t>& out) const
{
	out.clear();
	out[root] = step_t{root, (size_t)-1, 0};
	std::queue<TKeyFrameID> q;
	q.push(root);
	while (!q.empty()) {
		const TKeyFrameID u = q.front();
		q.pop();
		const size_t d = out[u].dist;
		if (d >= max_depth || u >= m_adj.size()) continue;
		for (const auto& ne : m_adj[u])
			if (!out.count(ne.first)) { out[ne.first] = step_t{u, ne.second, d + 1}; q.push(ne.first); }
	}
}

size_t mySRBA::topo_distance(TKeyFrameID a, TKeyFrameID b, size_t max_depth) const
{
	std::map<TKeyFrameID, step_t> t;
	bfs(a, max_depth, t);
	const auto it = t.find(b);
	return it == t.end() ? std::numeric_limits<size_t>::max() : it->second.dist;
}

bool mySRBA::get_kf_relative_pose(TKeyFrameID root, TKeyFrameID target, size_t max_depth, Pose3D& out) const
{
	std::map<TKeyFrameID, step_t> t;
	bfs(root, max_depth, t);
	if (!t.count(target)) return false;
	// walk target -> root; pose of target in root = prod over steps (prev <- cur)
	Pose3D acc;      // identity: pose of `cur` in `cur`
	TKeyFrameID cur = target;
	while (cur != root) {
		const step_t& s = t[cur];
		const k2k_edge_t& e = m_state.k2k_edges[s.edge];
		// T_{prev<-cur}: inv_pose maps `from`-frame points into the `to` frame
		const Pose3D step = (e.from == cur && e.to == s.prev) ? e.inv_pose : e.inv_pose.inverse();
		acc = step + acc;
		cur = s.prev;
	}
	out = acc;
	return true;
}

void mySRBA::get_global_graphslam_problem(TKeyFrameID root, std::vector<Pose3D>& nodes, std::vector<int32_t>& edge_ij, std::vector<Pose3D>& edge_z,
                                          std::vector<bool>& reachable) const
{
	const size_t n = num_keyframes();
	nodes.assign(n, Pose3D());
	reachable.assign(n, false);
	std::map<TKeyFrameID, step_t> t;
	bfs(root, std::numeric_limits<size_t>::max(), t);
	// parents before children: breadth-first distance order
	std::vector<std::pair<size_t, TKeyFrameID>> order;
	for (const auto& kv : t) order.push_back(std::make_pair(kv.second.dist, kv.first));
	std::sort(order.begin(), order.end());
	for (const auto& o : order) {
		const TKeyFrameID k = o.second;
		reachable[k] = true;
		if (k == root) continue;
		const step_t& s = t[k];
		const k2k_edge_t& e = m_state.k2k_edges[s.edge];
		// pose of k seen from its tree parent: inv_pose is `from` seen from `to`
		const Pose3D rel = (e.from == k && e.to == s.prev) ? e.inv_pose : e.inv_pose.inverse();
		nodes[k] = nodes[s.prev] + rel;
	}
	edge_ij.clear(); edge_z.clear();
	for (const k2k_edge_t& e : m_state.k2k_edges) {
		if (!reachable[e.from] || !reachable[e.to]) continue;
		edge_ij.push_back((int32_t)e.to); edge_ij.push_back((int32_t)e.from);
		edge_z.push_back(e.inv_pose);
	}
}

// ---- SRBA.h:144-308, restated line by line -------------------------------------------------------------
void mySRBA::edge_creation_policy(TKeyFrameID new_kf_id, const new_kf_observations_t& obs, std::vector<TNewEdgeInfo>& new_k2k_edge_ids)
{
	if (new_kf_id < 1) throw std::runtime_error("edge_creation_policy: called for the first KF");             // :150
	if (new_kf_id == 1) {                                                                                     // :153-158
		m_kf_localmap_center_ID.push_back(std::set<TKeyFrameID>{0});
		m_localmap_kf_IDs[0] = std::set<TKeyFrameID>();
	}
	const size_t MINIMUM_OBS_TO_LOOP_CLOSURE = parameters.min_obs_to_loop_closure;                              // :166
	const size_t SUBMAP_SIZE = parameters.submap_size;                                                         // :167
	const TKeyFrameID currentLocalmapBaseId = isKFLocalmapCenter(new_kf_id - 1) ? new_kf_id - 1 : getLocalmapCenterID(new_kf_id - 1);   // :170-173
	const size_t NUM_KFS_LOCALMAP = m_localmap_kf_IDs.count(currentLocalmapBaseId) ? m_localmap_kf_IDs[currentLocalmapBaseId].size() : (size_t)-1;   // :176
	if (NUM_KFS_LOCALMAP < SUBMAP_SIZE - 1) {                                                                 // :178
		m_kf_localmap_center_ID.push_back(std::set<TKeyFrameID>{currentLocalmapBaseId});                       // :182-183
		TNewEdgeInfo nei;
		nei.has_approx_init_val = false;
		nei.id = create_kf2kf_edge(currentLocalmapBaseId, new_kf_id);                                          // :186 (from = centre, to = new)
		if (NUM_KFS_LOCALMAP == 0) {
			// :194 as written assigns m_initial_kf_pose (the new KF seen from the previous one) to inv_pose, although
			// inv_pose means `from` (the centre = previous KF) seen from `to` (the new KF), i.e. its inverse -- the
			// other branch (:203) is consistent with the inverse.  Reproduced literally by default (only an initial guess).
			m_state.k2k_edges[nei.id].inv_pose = parameters.literal_first_edge_init ? m_initial_kf_pose : m_initial_kf_pose.inverse();
		} else {
			// :203  inv_pose.inverseComposeFrom(prev_edge.inv_pose, m_initial_kf_pose) = initial^-1 (+) prev_edge.inv_pose
			m_state.k2k_edges[nei.id].inv_pose = m_initial_kf_pose.inverse() + m_state.k2k_edges[nei.id - 1].inv_pose;
		}
		new_k2k_edge_ids.push_back(nei);
		m_localmap_kf_IDs[currentLocalmapBaseId].insert(new_kf_id);                                            // :209
	} else {
		// :220-221 make_ordered_list_base_kfs: #observations of already known landmarks per base KF, descending
		std::map<TKeyFrameID, size_t> obs_per_base;
		for (const auto& o : obs) {
			const auto it = m_state.unknown_lms.find(o.obs.feat_id);
			if (it != m_state.unknown_lms.end()) obs_per_base[it->second.base_kf]++;
		}
		std::multimap<size_t, TKeyFrameID, std::greater<size_t>> obs_for_each_base_sorted;
		for (const auto& kv : obs_per_base) obs_for_each_base_sorted.insert(std::make_pair(kv.second, kv.first));
		std::map<TKeyFrameID, size_t> obs_for_each_area;                                                       // :224-232
		for (const auto& kv : obs_for_each_base_sorted) obs_for_each_area[getLocalmapCenterID(kv.second)] += kv.first;
		std::multimap<size_t, TKeyFrameID, std::greater<size_t>> obs_for_each_area_sorted;                     // :235-239
		for (const auto& kv : obs_for_each_area) obs_for_each_area_sorted.insert(std::make_pair(kv.second, kv.first));
		for (const auto& kv : obs_for_each_area_sorted) {                                                      // :242
			const size_t num_obs_this_base = kv.first;
			const TKeyFrameID central_kf_id = kv.second;
			const size_t found_distance = topo_distance(new_kf_id, central_kf_id, parameters.max_tree_depth); // :253-269 (isolated KF: infinity)
			if (found_distance >= parameters.max_optimize_depth) {                                            // :271
				if (num_obs_this_base >= MINIMUM_OBS_TO_LOOP_CLOSURE) {                                       // :273
					TNewEdgeInfo nei;
					nei.id = create_kf2kf_edge(central_kf_id, new_kf_id);                                      // :278
					nei.has_approx_init_val = false;
					new_k2k_edge_ids.push_back(nei);
					if (new_kf_id < m_kf_localmap_center_ID.size()) m_kf_localmap_center_ID[new_kf_id].insert(central_kf_id);   // :284-290
					else m_kf_localmap_center_ID.push_back(std::set<TKeyFrameID>{central_kf_id});
					m_localmap_kf_IDs[central_kf_id].insert(new_kf_id);                                        // :293
					// :294 as written copies the value of the previously created edge.  For the edge to the new keyframe's own local
					// map that is a neighbouring pose; for a further edge to a far local map (a loop-closing link) it is off by the
					// distance between the two maps, and the robust LM of this build does not recover from that (reproduction:
					// tests/test_config3_gpu.py::test_literal_loop_closing_edge_init_diverges).  Unless literal_lc_edge_init is set, such
					// an edge starts from the composition of the current estimates along the graph (the new keyframe is already
					// linked by its first edge).
					Pose3D init = nei.id == 0 ? Pose3D() : m_state.k2k_edges[nei.id - 1].inv_pose;
					if (!parameters.literal_lc_edge_init && new_k2k_edge_ids.size() > 1) {
						Pose3D rel;
						m_state.k2k_edges[nei.id].inv_pose = init;
						// the edge just created must not be used for its own initial value: look the centre up without it
						m_adj[central_kf_id].pop_back(); m_adj[new_kf_id].pop_back();
						const bool found = get_kf_relative_pose(new_kf_id, central_kf_id, std::numeric_limits<size_t>::max(), rel);
						m_adj[central_kf_id].push_back(std::make_pair(new_kf_id, nei.id)); m_adj[new_kf_id].push_back(std::make_pair(central_kf_id, nei.id));
						if (found) init = rel;
					}
					m_state.k2k_edges[nei.id].inv_pose = init;
				}
			}
		}
		if (new_k2k_edge_ids.empty())                                                                          // :304
			throw std::runtime_error("Error for new KF: no suitable linking KF found with the minimum of common observations: the node becomes isolated of the graph!");
		m_localmap_kf_IDs[new_kf_id] = std::set<TKeyFrameID>();                                                // :305
	}
}

struct mySRBA::worker_t {
	std::mutex mu;
	std::condition_variable cv;
	std::function<void()> job;
	bool busy = false, quit = false;
	std::thread th;
	worker_t() : th([this]() {
		std::unique_lock<std::mutex> lk(mu);
		for (;;) {
			cv.wait(lk, [this]() { return quit || (busy && job); });
			if (quit) return;
			std::function<void()> f; f.swap(job);
			lk.unlock(); f(); lk.lock();
			busy = false;
			cv.notify_all();
		}
	}) {}
	void start(std::function<void()> f) { std::unique_lock<std::mutex> lk(mu); job = std::move(f); busy = true; cv.notify_all(); }
	void wait() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [this]() { return !busy; }); }
	~worker_t() { { std::unique_lock<std::mutex> lk(mu); quit = true; cv.notify_all(); } th.join(); }
};

// one LM solve on the MI355X, in two halves: lm_prepare builds the window (opt_edges + everything the referenced observations' paths
// touch) and its device structures -- it only READS the map, so the local-area window of a keyframe can be prepared on a second
// thread beside the solve of its new edges --, lm_finish solves and writes the result back.
struct mySRBA::lm_job {
	std::vector<int32_t> obs_lm, obs_path_off, path_edge, edge_slot, lm_slot;
	std::vector<float> obs_px;
	std::vector<double> pose, lmpos;
	std::vector<TLandmarkID> lm_ids;
	std::vector<landmark_t*> lm_ptr;
	std::vector<size_t> edge_ids;
	ssk_lm_window w{};
	ssk_lm_batch* b = nullptr;
	bool optimize_landmarks = false;
	int n_opt = 0, rc = 0;
	double build_ms = 0, create_ms = 0;
	std::chrono::steady_clock::time_point t_launch;
	~lm_job() { if (b) ssk_lm_batch_destroy(b); }
};

namespace {
inline std::chrono::steady_clock::time_point lm_now() { return std::chrono::steady_clock::now(); }
inline double lm_ms(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) { return std::chrono::duration<double, std::milli>(c - a).count(); }
}

mySRBA::~mySRBA() { delete m_worker; }

void mySRBA::lm_prepare(lm_job& j, const std::set<size_t>& opt_edges, const std::vector<std::pair<TKeyFrameID, size_t>>& obs_refs, bool optimize_landmarks, bool create) const
{
	const auto t0 = lm_now();
	// per observing keyframe its depth-limited breadth-first tree, flattened: step towards the observer per keyframe id
	// (edge == SIZE_MAX: not in the tree); the window is built once per keyframe over ~10^4 observations, so the lookups
	// are arrays and hashes, not ordered maps
	const size_t n_kf = num_keyframes();
	std::unordered_map<TKeyFrameID, std::vector<step_t>>& trees = m_tree_cache[optimize_landmarks ? 1 : 0];      // (kept across keyframes: see the header)
	std::vector<int32_t>& obs_lm = j.obs_lm; std::vector<int32_t>& obs_path_off = j.obs_path_off; std::vector<int32_t>& path_edge = j.path_edge;
	std::vector<float>& obs_px = j.obs_px;
	obs_path_off.assign(1, 0);
	std::vector<int> edge_idx(m_state.k2k_edges.size(), -1);
	const int slot = optimize_landmarks ? 1 : 0;
	std::vector<uint32_t>& stamp = m_lm_stamp[slot];
	std::vector<int32_t>& local = m_lm_local[slot];
	if (stamp.size() < m_n_lm_seq) { stamp.resize(m_n_lm_seq + m_n_lm_seq / 2 + 1024, 0u); local.resize(stamp.size(), -1); }
	if (++m_lm_gen[slot] == 0u) { std::fill(stamp.begin(), stamp.end(), 0u); m_lm_gen[slot] = 1u; }
	const uint32_t gen = m_lm_gen[slot];
	std::vector<TLandmarkID>& lm_ids = j.lm_ids;
	std::vector<landmark_t*>& lm_ptr = j.lm_ptr;
	std::vector<size_t>& edge_ids = j.edge_ids;
	auto edge_index = [&](size_t e) { if (edge_idx[e] >= 0) return edge_idx[e]; const int i = (int)edge_ids.size(); edge_idx[e] = i; edge_ids.push_back(e); return i; };
	for (size_t e : opt_edges) edge_index(e);
	obs_lm.reserve(obs_refs.size()); obs_px.reserve(4 * obs_refs.size()); path_edge.reserve(4 * obs_refs.size());
	TKeyFrameID tree_kf = (TKeyFrameID)-1;
	const std::vector<step_t>* tree_p = nullptr;
	std::vector<int32_t> steps;
	for (const auto& ref : obs_refs) {
		const TKeyFrameID kf = ref.first;
		const kf_observation_t& o = m_state.kf_observations[kf][ref.second];
		landmark_t* const lmk = o.lm;
		if (!lmk) continue;
		if (kf != tree_kf) {
			auto it = trees.find(kf);
			if (it == trees.end()) {
				std::map<TKeyFrameID, step_t> t;
				bfs(kf, parameters.max_tree_depth, t);
				std::vector<step_t>& flat = trees[kf];
				flat.assign(n_kf, step_t{0, (size_t)-1, 0});
				for (const auto& kv : t) if (kv.first < n_kf) flat[kv.first] = kv.second;
				flat[kf].edge = 0;                           // the root is in its own tree
				it = trees.find(kf);
			}
			tree_kf = kf; tree_p = &it->second;
		}
		const std::vector<step_t>& tree = *tree_p;
		const TKeyFrameID base = lmk->base_kf;
		if (base >= tree.size() || tree[base].edge == (size_t)-1) continue;   // farther than max_tree_depth: not in the spanning tree (a cached
		                                                                       // tree is as long as the map was when it was built: bases are older)
		steps.clear();                                       // from the base back to the observer
		for (TKeyFrameID cur = base; cur != kf;) {
			const step_t& st = tree[cur];
			const k2k_edge_t& e = m_state.k2k_edges[st.edge];
			// step observer-side keyframe st.prev <- cur: inv_pose applies when cur == from and prev == to
			steps.push_back(edge_index(st.edge) * 2 + ((e.from == cur && e.to == st.prev) ? 0 : 1));
			cur = st.prev;
		}
		if (steps.size() > 8) continue;
		path_edge.insert(path_edge.end(), steps.rbegin(), steps.rend());
		obs_path_off.push_back((int32_t)path_edge.size());
		if (stamp[lmk->seq] != gen) { stamp[lmk->seq] = gen; local[lmk->seq] = (int32_t)lm_ids.size(); lm_ids.push_back(o.feat_id); lm_ptr.push_back(lmk); }
		obs_lm.push_back(local[lmk->seq]);
		obs_px.insert(obs_px.end(), o.px, o.px + 4);
	}
	j.edge_slot.assign(edge_ids.size(), -1); j.lm_slot.assign(lm_ids.size(), -1);
	j.pose.resize(12 * edge_ids.size()); j.lmpos.resize(3 * lm_ids.size());
	j.optimize_landmarks = optimize_landmarks;
	int s = 0;
	for (size_t i = 0; i < edge_ids.size(); ++i) {
		if (opt_edges.count(edge_ids[i])) j.edge_slot[i] = s++;
		std::memcpy(&j.pose[12 * i], m_state.k2k_edges[edge_ids[i]].inv_pose.m, 12 * sizeof(double));
	}
	j.n_opt = s;
	for (size_t i = 0; i < lm_ids.size(); ++i) {
		if (optimize_landmarks) j.lm_slot[i] = (int32_t)i;
		std::memcpy(&j.lmpos[3 * i], lm_ptr[i]->pos, 3 * sizeof(double));
	}
	ssk_lm_window w;
	w.n_edges_total = (int32_t)edge_ids.size(); w.edge_unknown_slot = j.edge_slot.data(); w.edge_inv_pose = j.pose.data();
	w.n_lms_total = (int32_t)lm_ids.size(); w.lm_unknown_slot = j.lm_slot.data(); w.lm_pos = j.lmpos.data();
	w.n_obs = (int32_t)obs_lm.size(); w.obs_lm = obs_lm.data(); w.obs_px = obs_px.data();
	w.obs_path_off = obs_path_off.data(); w.path_edge = path_edge.data();
	j.w = w;
	j.build_ms = lm_ms(t0, lm_now());
	if (create) lm_create(j);
}

// the device structures of a built window (HIP calls: kept on the caller's thread -- allocation and uploads from a second thread
// beside a running solve measured slower than one after the other)
void mySRBA::lm_create(lm_job& j) const
{
	const auto t1 = lm_now();
	m_ctx.check(ssk_lm_batch_create(m_ctx.get(), &j.w, 1, 0, &j.b), "ssk_lm_batch_create");
	j.create_ms = lm_ms(t1, lm_now());
}

// launches the solve.  refresh: the job was prepared before the map took its latest values (the solve of the new edges): upload the
// current ones first
void mySRBA::lm_launch(lm_job& j, bool robust, bool refresh)
{
	j.t_launch = lm_now();
	int rc = SSK_OK;
	if (refresh) {
		for (size_t i = 0; i < j.edge_ids.size(); ++i) std::memcpy(&j.pose[12 * i], m_state.k2k_edges[j.edge_ids[i]].inv_pose.m, 12 * sizeof(double));
		rc = ssk_lm_batch_set_state(m_ctx.get(), j.b, 0, j.pose.data(), nullptr);      // (the solve of the new edges holds the landmarks)
	}
	const ssk_lm_params prm = m_opts.lm(robust);
	if (rc == SSK_OK) rc = ssk_lm_batch_run(m_ctx.get(), j.b, &prm);
	j.rc = rc;
}

// waits for the solve and writes its result to the map
void mySRBA::lm_collect(lm_job& j, TOptimizeResults& res)
{
	static const bool timing = std::getenv("SSK_FACADE_TIMING") != nullptr;
	int rc = j.rc;
	ssk_lm_result r{};
	if (rc == SSK_OK) rc = ssk_lm_batch_results(m_ctx.get(), j.b, &r);
	const auto t3 = lm_now();
	if (rc == SSK_OK) rc = ssk_lm_batch_get_state(m_ctx.get(), j.b, 0, j.pose.data(), j.lmpos.data());
	ssk_lm_batch_destroy(j.b); j.b = nullptr;
	const auto t4 = lm_now();
	this->timing.window_build_ms += j.build_ms; this->timing.batch_create_ms += j.create_ms; this->timing.solve_ms += lm_ms(j.t_launch, t3);
	this->timing.readback_ms += lm_ms(t3, t4); this->timing.solves++; this->timing.iters += rc == SSK_OK ? (size_t)r.iters : 0;
	this->timing.max_opt_edges = std::max(this->timing.max_opt_edges, (size_t)j.n_opt); this->timing.max_obs = std::max(this->timing.max_obs, j.obs_lm.size());
	if (timing) std::fprintf(stderr, "[facade timing]   run_lm (%d edges opt, %zu obs): window build %.2f ms, batch create %.2f ms, solve %.2f ms (%d iters), state+destroy %.2f ms\n",
	                         j.n_opt, j.obs_lm.size(), j.build_ms, j.create_ms, lm_ms(j.t_launch, t3), (int)r.iters, lm_ms(t3, lm_now()));
	m_ctx.check(rc, "LM solve");
	lm_apply(j, r, res);
}

void mySRBA::lm_apply(lm_job& j, const ssk_lm_result& r, TOptimizeResults& res)
{
	for (size_t i = 0; i < j.edge_ids.size(); ++i)
		if (j.edge_slot[i] >= 0) std::memcpy(m_state.k2k_edges[j.edge_ids[i]].inv_pose.m, &j.pose[12 * i], 12 * sizeof(double));
	if (j.optimize_landmarks)
		for (size_t i = 0; i < j.lm_ids.size(); ++i) std::memcpy(j.lm_ptr[i]->pos, &j.lmpos[3 * i], 3 * sizeof(double));
	res.total_sqr_error_init = r.err_init; res.total_sqr_error_final = r.err_final;
	res.num_observations = (size_t)r.n_obs; res.num_iters = (size_t)r.iters; res.num_opt_edges = (size_t)j.n_opt;
	res.obs_rmse = r.n_obs ? std::sqrt(r.err_final / r.n_obs) : 0.0;
}

void mySRBA::kf_begin(const new_kf_observations_t& obs, TNewKeyFrameInfo& out, std::vector<TNewEdgeInfo>& new_edges)
{
	const TKeyFrameID new_kf_id = m_state.kf_observations.size();               // alloc_keyframe
	m_state.kf_observations.push_back(std::vector<kf_observation_t>());
	if (m_adj.size() <= new_kf_id) m_adj.resize(new_kf_id + 1);
	out = TNewKeyFrameInfo();
	out.kf_id = new_kf_id;
	new_edges.clear();
	if (new_kf_id >= 1) edge_creation_policy(new_kf_id, obs, new_edges);        // virtual edge_creation_policy
	for (const auto& e : new_edges) out.created_edge_ids.push_back(e.id);
	// (a keyframe with one new edge is a leaf: no path between older keyframes changes; with several it may open shorter ones)
	if (new_edges.size() != 1) { m_tree_cache[0].clear(); m_tree_cache[1].clear(); }
	for (auto& c : m_tree_cache) if (c.size() > 256) c.clear();      // (a local area holds a few dozen observers: the rest is left behind)
	// add_observation: a feature id seen for the first time becomes a landmark based at this keyframe with the given
	// relative position (is_unknown_with_init_val, EST.cpp:143-158)
	for (const auto& o : obs) {
		auto lit = m_state.unknown_lms.find(o.obs.feat_id);
		if (lit == m_state.unknown_lms.end()) {
			landmark_t lm;
			lm.base_kf = new_kf_id;
			lm.pos[0] = o.rel_pos[0]; lm.pos[1] = o.rel_pos[1]; lm.pos[2] = o.rel_pos[2];
			lm.seq = m_n_lm_seq++;
			lit = m_state.unknown_lms.emplace(o.obs.feat_id, lm).first;
		}
		kf_observation_t k;
		k.lm = &lit->second;
		k.feat_id = o.obs.feat_id;
		k.px[0] = o.obs.obs_data.left_px.x; k.px[1] = o.obs.obs_data.left_px.y;
		k.px[2] = o.obs.obs_data.right_px.x; k.px[3] = o.obs.obs_data.right_px.y;
		m_state.kf_observations[new_kf_id].push_back(k);
	}
}

// the window of stage 1: optimize_edges on the new edges only, with the new keyframe's observations, landmarks held
void mySRBA::stage1_prepare(lm_job& j, TKeyFrameID new_kf_id, const std::vector<TNewEdgeInfo>& new_edges, bool create) const
{
	std::set<size_t> opt;
	for (const auto& e : new_edges) opt.insert(e.id);
	std::vector<std::pair<TKeyFrameID, size_t>> refs;
	for (size_t i = 0; i < m_state.kf_observations[new_kf_id].size(); ++i) refs.push_back(std::make_pair(new_kf_id, i));
	lm_prepare(j, opt, refs, false, create);
}

// the window of stage 2: optimize_local_area(root = new KF, depth = max_optimize_depth)
void mySRBA::stage2_prepare(lm_job& j2, TKeyFrameID new_kf_id, bool create) const
{
	// (the device solver's reduced system holds 128 edges, include/ssk_capi.h: a local area beyond that -- a revisit that ties several
	// old local maps to the new one within the depth -- is optimised to the largest depth that fits.  SRBA has no such limit.)
	std::map<TKeyFrameID, step_t> area;
	std::set<size_t> opt;
	for (size_t depth = parameters.max_optimize_depth; ; --depth) {
		area.clear(); opt.clear();
		bfs(new_kf_id, depth, area);
		for (size_t e = 0; e < m_state.k2k_edges.size(); ++e)
			if (area.count(m_state.k2k_edges[e].from) && area.count(m_state.k2k_edges[e].to)) opt.insert(e);
		if (opt.size() <= 128 || depth <= 1) break;
	}
	std::vector<std::pair<TKeyFrameID, size_t>> refs;
	for (const auto& kv : area)
		for (size_t i = 0; i < m_state.kf_observations[kv.first].size(); ++i) refs.push_back(std::make_pair(kv.first, i));
	lm_prepare(j2, opt, refs, true, create);
}

void mySRBA::define_new_keyframe(const new_kf_observations_t& obs, TNewKeyFrameInfo& out, bool run_local_optimization)
{
	std::vector<TNewEdgeInfo> new_edges;
	kf_begin(obs, out, new_edges);
	const TKeyFrameID new_kf_id = out.kf_id;
	if (!run_local_optimization || new_edges.empty()) return;
	// stage 1: optimize_edges on the new edges only, with the new keyframe's observations, landmarks held
	// stage 2: optimize_local_area(root = new KF, depth = max_optimize_depth)
	// The local-area window (~10^4 observations, ~0.6 ms of host work) is collected on a second thread while stage 1 is prepared and
	// solved; it reads the map only, stage 1's result is written to the map after the join and uploaded to the prepared window
	// -- on request (SSK_FACADE_OVERLAP=1).  Measured on one box, 401 frames, ms per keyframe frame: one after the other 8.53 / 8.54, overlapped
	// 8.30 / 8.98 (the hand-over costs what it saves); with the batch creation on the second thread too: 9.13 (allocation and uploads beside a
	// running solve serialise in the runtime).  Default: one after the other.
	// Round 4, the default (SSK_FACADE_OVERLAP=2 or unset): the SAME thread collects and creates the local-area window between the launch of stage 1
	// and the read-back of its result -- the host work hides stage 1's solve (~0.25 ms of device time per keyframe), nothing is handed over; the
	// library keeps two device slabs per context for the two batches that are alive together.  0: one after the other, 1: the second thread.
	static const int overlap_mode = std::getenv("SSK_FACADE_OVERLAP") ? std::atoi(std::getenv("SSK_FACADE_OVERLAP")) : 2;
	const bool overlap = overlap_mode == 1;
	lm_job j2;
	auto prepare_stage2 = [&]() { stage2_prepare(j2, new_kf_id, !overlap); };
	std::exception_ptr worker_error;
	if (overlap && !m_worker) m_worker = new worker_t();
	if (overlap) m_worker->start([&]() { try { prepare_stage2(); } catch (...) { worker_error = std::current_exception(); } });
	struct joiner { worker_t* w; ~joiner() { if (w) w->wait(); } } join_on_exit{overlap ? m_worker : nullptr};
	{
		lm_job j1;
		stage1_prepare(j1, new_kf_id, new_edges, true);
		lm_launch(j1, parameters.use_robust_kernel_stage1, false);
		if (overlap_mode == 2) prepare_stage2();           // (reads the map before stage 1's result is written to it: its poses are refreshed below)
		// (lm_collect writes the new edges' poses to the map: only after the worker has stopped reading it)
		if (overlap) m_worker->wait();
		if (worker_error) std::rethrow_exception(worker_error);
		lm_collect(j1, out.optimize_results_stg1);
	}
	if (overlap_mode == 0) prepare_stage2();
	else if (overlap) lm_create(j2);
	lm_launch(j2, parameters.use_robust_kernel, overlap_mode != 0);
	lm_collect(j2, out.optimize_results);
}

// ---- lock-step batch of engines (round 5) ----
namespace {
// The batch form's host threads: a pool per CALLING thread (two groups of streams driven from two threads do not take turns), workers asleep on a
// condition variable between jobs -- the round-5 form spawned and joined n_threads std::threads for each of a call's four parallel sections
// (~100 thread creations per call).  run(): items 0 .. n-1 dealt out one at a time to the workers and the caller; the first exception is rethrown.
struct host_pool {
	std::mutex mx;
	std::condition_variable cv, cv_done;
	std::vector<std::thread> workers;
	const std::function<void(int)>* fn = nullptr;
	std::atomic<int> next{0};
	int n_items = 0, gen = 0, want = 0, busy = 0;
	bool stop = false;
	std::exception_ptr err;
	void body()
	{
		try { for (int i = next++; i < n_items; i = next++) (*fn)(i); }
		catch (...) { std::lock_guard<std::mutex> lk(mx); if (!err) err = std::current_exception(); next = n_items; }
	}
	void work()
	{
		int seen = 0;
		for (;;) {
			{
				std::unique_lock<std::mutex> lk(mx);
				cv.wait(lk, [&] { return stop || (gen != seen && want > 0); });
				if (stop) return;
				seen = gen; --want;
			}
			body();
			{
				std::lock_guard<std::mutex> lk(mx);
				if (--busy == 0) cv_done.notify_one();
			}
		}
	}
	void run(int threads, int n, const std::function<void(int)>& f)
	{
		const int helpers = std::max(0, std::min(threads, n) - 1);
		while ((int)workers.size() < helpers) workers.emplace_back([this] { work(); });
		{
			std::lock_guard<std::mutex> lk(mx);
			fn = &f; n_items = n; next = 0; want = helpers; busy = helpers; ++gen; err = nullptr;
		}
		if (helpers) cv.notify_all();
		body();
		std::unique_lock<std::mutex> lk(mx);
		cv_done.wait(lk, [&] { return busy == 0; });
		fn = nullptr;
		if (err) { std::exception_ptr e = err; err = nullptr; std::rethrow_exception(e); }
	}
	~host_pool()
	{
		{ std::lock_guard<std::mutex> lk(mx); stop = true; }
		cv.notify_all();
		for (std::thread& th : workers) th.join();
	}
};
// (ONE pool per calling thread: a thread_local inside the template below would be one per lambda type)
host_pool& host_pool_of_this_thread() { static thread_local host_pool pool; return pool; }
template <class F>
void parallel_for(int n, int n_threads, F f)
{
	if (n_threads <= 1 || n <= 1) { for (int i = 0; i < n; ++i) f(i); return; }
	const std::function<void(int)> fn = f;
	host_pool_of_this_thread().run(n_threads, n, fn);
}
int batch_host_threads()
{
	if (const char* e = std::getenv("SSK_RBA_BATCH_THREADS")) return std::max(1, std::atoi(e));
	// as the LM builder: half the logical cores, at most 64 and at most twice the container's CPU quota (cgroup v2 cpu.max)
	static const int n_auto = [] {
		int n = std::max(1, std::min((int)std::thread::hardware_concurrency() / 2, 64));
		if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
			char q[32]; double per = 0.0;
			if (std::fscanf(f, "%31s %lf", q, &per) == 2 && std::strcmp(q, "max") != 0 && per > 0) n = std::max(2, std::min(n, (int)(2.0 * std::atof(q) / per + 0.5)));
			std::fclose(f);
		}
		return n;
	}();
	return n_auto;
}
}  // namespace

void mySRBA::define_new_keyframe_batch(const std::vector<mySRBA*>& E, const std::vector<const new_kf_observations_t*>& obs,
                                       std::vector<TNewKeyFrameInfo>& out, bool run_local_optimization, TBatchTiming* tm)
{
	if (obs.size() != E.size()) throw std::runtime_error("define_new_keyframe_batch: one observation list per engine");
	define_new_keyframe_batch_impl(E, &obs, nullptr, out, run_local_optimization, tm);
}

void mySRBA::define_new_keyframe_batch(const std::vector<mySRBA*>& E, const std::function<void(int, new_kf_observations_t&)>& fill,
                                       std::vector<TNewKeyFrameInfo>& out, bool run_local_optimization, TBatchTiming* tm)
{
	define_new_keyframe_batch_impl(E, nullptr, &fill, out, run_local_optimization, tm);
}

void mySRBA::define_new_keyframe_batch_impl(const std::vector<mySRBA*>& E, const std::vector<const new_kf_observations_t*>* obs,
                                            const std::function<void(int, new_kf_observations_t&)>* fill,
                                            std::vector<TNewKeyFrameInfo>& out, bool run_local_optimization, TBatchTiming* tm)
{
	const int n = (int)E.size();
	out.assign((size_t)n, TNewKeyFrameInfo());
	if (n == 0) return;
	Context& ctx = E[0]->m_ctx;
	for (const mySRBA* e : E) if (e->m_ctx.get() != ctx.get()) throw std::runtime_error("define_new_keyframe_batch: the engines of a batch share one context");
	const int n_thr = batch_host_threads();             // (read per call: groups of streams driven side by side share the box's CPU quota -- SSK_RBA_BATCH_THREADS)
	TBatchTiming T;
	T.host_threads = n_thr;
	auto t0 = lm_now();
	const auto t_call = t0;
	// (1) per engine, side by side: graph bookkeeping (alloc_keyframe, edge_creation_policy, add_observation) and BOTH windows -- they read the map only;
	// stage 1's result is written to the map, and to the prepared local-area windows' edge poses, in (3).
	// An engine whose bookkeeping throws (edge_creation_policy: "the node becomes isolated", SRBA.h:304) has taken the keyframe and is left out of the
	// solves; the others go on.  The call then throws with the engines' indices in the text (the state of every OTHER engine is what
	// define_new_keyframe would have left; the failed ones are as after define_new_keyframe's own throw).
	std::vector<std::vector<TNewEdgeInfo>> new_edges((size_t)n);
	std::vector<std::unique_ptr<lm_job>> j1((size_t)n), j2((size_t)n);
	std::vector<std::string> failed((size_t)n);
	parallel_for(n, n_thr, [&](int i) {
		try {
			if (fill) {
				new_kf_observations_t list;
				(*fill)(i, list);
				E[(size_t)i]->kf_begin(list, out[(size_t)i], new_edges[(size_t)i]);
			} else E[(size_t)i]->kf_begin(*(*obs)[(size_t)i], out[(size_t)i], new_edges[(size_t)i]);
			if (!run_local_optimization || new_edges[(size_t)i].empty()) return;
			std::unique_ptr<lm_job> a(new lm_job()), c(new lm_job());
			E[(size_t)i]->stage1_prepare(*a, out[(size_t)i].kf_id, new_edges[(size_t)i], false);
			E[(size_t)i]->stage2_prepare(*c, out[(size_t)i].kf_id, false);
			j1[(size_t)i] = std::move(a); j2[(size_t)i] = std::move(c);
		} catch (const std::exception& ex) { failed[(size_t)i] = ex.what()[0] ? ex.what() : "error"; }
	});
	std::string fail_text;
	{
		int nf = 0;
		for (int i = 0; i < n; ++i) if (!failed[(size_t)i].empty()) { if (nf++ < 8) fail_text += (fail_text.empty() ? "" : "; ") + std::string("engine ") + std::to_string(i) + ": " + failed[(size_t)i]; }
		if (nf > 8) fail_text += "; ... (" + std::to_string(nf) + " engines in all)";
	}
	std::vector<int> act;
	for (int i = 0; i < n; ++i) if (j1[(size_t)i]) act.push_back(i);
	T.bookkeeping_and_windows_ms = lm_ms(t0, lm_now());
	auto finish = [&] {
		T.total_ms = lm_ms(t_call, lm_now());
		if (tm) *tm = T;
		if (!fail_text.empty()) throw std::runtime_error("define_new_keyframe_batch: " + fail_text);
	};
	if (act.empty()) { finish(); return; }
	const int na = (int)act.size();
	struct batch_guard { ssk_lm_batch* b = nullptr; ~batch_guard() { if (b) ssk_lm_batch_destroy(b); } };
	auto solve = [&](std::vector<std::unique_ptr<lm_job>>& J, bool robust, double& create_ms, double& solve_ms, double& sym_ms, char* kernel, std::vector<ssk_lm_result>& res, double& mean_iters) {
		std::vector<ssk_lm_window> wins((size_t)na);
		size_t n_pose = 0, n_lm = 0;
		for (int k = 0; k < na; ++k) { wins[(size_t)k] = J[(size_t)act[(size_t)k]]->w; n_pose += 12 * (size_t)wins[(size_t)k].n_edges_total; n_lm += 3 * (size_t)wins[(size_t)k].n_lms_total; }
		batch_guard g;
		auto ta = lm_now();
		ctx.check(ssk_lm_batch_create(ctx.get(), wins.data(), na, 0, &g.b), "ssk_lm_batch_create");
		auto tb = lm_now();
		create_ms = lm_ms(ta, tb);
		const ssk_lm_params prm = E[0]->m_opts.lm(robust);
		ctx.check(ssk_lm_batch_run(ctx.get(), g.b, &prm), "ssk_lm_batch_run");
		res.resize((size_t)na);
		ctx.check(ssk_lm_batch_results(ctx.get(), g.b, res.data()), "ssk_lm_batch_results");
		// (uninitialised: the download fills them; zeroing 30 MB first cost milliseconds on one thread)
		std::unique_ptr<double[]> pose(new double[n_pose ? n_pose : 1]), lmp(new double[n_lm ? n_lm : 1]);
		ctx.check(ssk_lm_batch_get_state_all(ctx.get(), g.b, pose.get(), lmp.get()), "ssk_lm_batch_get_state_all");
		solve_ms = lm_ms(tb, lm_now());
		int thr = 0;
		(void)ssk_lm_batch_symbolic_ms(ctx.get(), g.b, &sym_ms, &thr);
		(void)ssk_lm_batch_kernel_name(ctx.get(), g.b, kernel, 96);
		// every window's share to its job, side by side (30 MB through one thread were milliseconds)
		std::vector<size_t> oe((size_t)na + 1, 0), ol((size_t)na + 1, 0);
		double it = 0;
		for (int k = 0; k < na; ++k) {
			const lm_job& j = *J[(size_t)act[(size_t)k]];
			oe[(size_t)k + 1] = oe[(size_t)k] + j.pose.size(); ol[(size_t)k + 1] = ol[(size_t)k] + j.lmpos.size();
			it += res[(size_t)k].iters;
		}
		parallel_for(na, n_thr, [&](int k) {
			lm_job& j = *J[(size_t)act[(size_t)k]];
			std::memcpy(j.pose.data(), pose.get() + oe[(size_t)k], sizeof(double) * j.pose.size());
			std::memcpy(j.lmpos.data(), lmp.get() + ol[(size_t)k], sizeof(double) * j.lmpos.size());
		});
		mean_iters = it / na;
	};
	// (2) stage 1 of every engine: one batch (one or two optimised edges, no optimised landmark: the pose-only kernel)
	std::vector<ssk_lm_result> r1, r2;
	solve(j1, E[0]->parameters.use_robust_kernel_stage1, T.stage1_create_ms, T.stage1_solve_ms, T.stage1_symbolic_ms, T.stage1_kernel, r1, T.stage1_mean_iters);
	// (3) its result to the maps, and into the local-area windows prepared before it
	auto t3 = lm_now();
	parallel_for(na, n_thr, [&](int k) {
		const int i = act[(size_t)k];
		mySRBA& e = *E[(size_t)i];
		e.lm_apply(*j1[(size_t)i], r1[(size_t)k], out[(size_t)i].optimize_results_stg1);
		lm_job& j = *j2[(size_t)i];
		for (size_t q = 0; q < j.edge_ids.size(); ++q) std::memcpy(&j.pose[12 * q], e.m_state.k2k_edges[j.edge_ids[q]].inv_pose.m, 12 * sizeof(double));
	});
	T.stage2_refresh_ms = lm_ms(t3, lm_now());
	// (4) stage 2 of every engine: one batch
	solve(j2, E[0]->parameters.use_robust_kernel, T.stage2_create_ms, T.stage2_solve_ms, T.stage2_symbolic_ms, T.stage2_kernel, r2, T.stage2_mean_iters);
	auto t5 = lm_now();
	parallel_for(na, n_thr, [&](int k) {
		const int i = act[(size_t)k];
		E[(size_t)i]->lm_apply(*j2[(size_t)i], r2[(size_t)k], out[(size_t)i].optimize_results);
	});
	T.writeback_ms = lm_ms(t5, lm_now());
	T.n_stage1 = T.n_stage2 = na;
	for (int k = 0; k < na; ++k) {
		const lm_job& a = *j1[(size_t)act[(size_t)k]]; const lm_job& c = *j2[(size_t)act[(size_t)k]];
		T.stage1_mean_obs += (double)a.obs_lm.size() / na; T.stage2_mean_obs += (double)c.obs_lm.size() / na; T.stage2_mean_edges += (double)c.n_opt / na;
		T.stage2_max_edges = std::max(T.stage2_max_edges, c.n_opt); T.stage2_max_obs = std::max(T.stage2_max_obs, (int)c.obs_lm.size());
		mySRBA& e = *E[(size_t)act[(size_t)k]];
		e.timing.solves += 2; e.timing.iters += (size_t)r1[(size_t)k].iters + (size_t)r2[(size_t)k].iters;
		e.timing.max_opt_edges = std::max(e.timing.max_opt_edges, (size_t)c.n_opt); e.timing.max_obs = std::max(e.timing.max_obs, c.obs_lm.size());
	}
	// the windows' host arrays (a dozen vectors per engine and stage) go side by side too: one thread freeing 2 048 windows was milliseconds
	auto t6 = lm_now();
	parallel_for(n, n_thr, [&](int i) { j1[(size_t)i].reset(); j2[(size_t)i].reset(); });
	T.teardown_ms = lm_ms(t6, lm_now());
	finish();
}

}  // namespace ssk
