// srba_stereo_slam.h -- the facade of the reference's RBA engine (src/srba-stereo-slam.h:30-310: mySRBA :
// RbaEngine<kf2kf_poses::SE3, landmarks::Euclidean3D, observations::StereoCamera, my_srba_options>), keeping
// the names the estimator uses: define_new_keyframe, setInitialKFPose, TNewKeyFrameInfo, new_kf_observation_t,
// get_rba_state().k2k_edges / unknown_lms, parameters.*.
//
// What is host code here (graph bookkeeping, as in SRBA itself): keyframe / edge / landmark containers, the
// reference's fixed-size-submap edge_creation_policy (restated literally from SRBA.h:144-308), breadth-first
// spanning-tree paths limited by max_tree_depth, selection of the optimisation window.
// What runs on the MI355X: every LM solve (stage 1 "optimize_edges" on the new edges, stage 2
// "optimize_local_area"), through ssk_lm_batch_* (include/ssk_capi.h).
#pragma once
#include <deque>
#include <functional>
#include <map>
#include <set>
#include <unordered_map>
#include "srba_stereo_slam_options.h"

namespace ssk {

typedef size_t TKeyFrameID;
typedef uint64_t TLandmarkID;

struct new_kf_observation_t {                  // EST.cpp:137-160,671-699
	struct { TLandmarkID feat_id; struct { struct { float x, y; } left_px, right_px; } obs_data; } obs;
	bool is_fixed = false, is_unknown_with_init_val = true;
	double rel_pos[3] = {0, 0, 0};
	void setRelPos(const double p[3]) { rel_pos[0] = p[0]; rel_pos[1] = p[1]; rel_pos[2] = p[2]; }
};
typedef std::vector<new_kf_observation_t> new_kf_observations_t;

struct TOptimizeResults {
	double total_sqr_error_init = 0, total_sqr_error_final = 0, obs_rmse = 0;
	size_t num_observations = 0, num_iters = 0, num_opt_edges = 0;
};
struct TNewKeyFrameInfo {                      // EST.cpp:178-180,790,847-849
	TKeyFrameID kf_id = 0;
	std::vector<size_t> created_edge_ids;
	TOptimizeResults optimize_results, optimize_results_stg1;
};

struct k2k_edge_t { TKeyFrameID from, to; Pose3D inv_pose; };   // inv_pose: `from` as seen from `to`
struct landmark_t { TKeyFrameID base_kf; double pos[3]; uint32_t seq = 0; };      // seq: creation order (index of the per-window scratch of the LM window builder)
struct kf_observation_t { TLandmarkID feat_id; float px[4]; landmark_t* lm = nullptr; };      // lm: the map node (stable: nodes of an unordered_map do not move)

class mySRBA {
public:
	struct TParameters {
		size_t max_tree_depth = 4, max_optimize_depth = 4, submap_size = 15;       // EST.cpp:1154-1156
		size_t min_obs_to_loop_closure = 6;                                         // SRBA default (never set by the reference, SRBA.h:166)
		bool use_robust_kernel = true, use_robust_kernel_stage1 = true;             // EST.cpp:1150-1151
		double kernel_param = 3.0;                                                  // EST.cpp:1152
		double std_noise_observations = 0.5;                                        // EST.cpp:1157 (uniform: cancels in the solve)
		bool literal_first_edge_init = true;                                        // SRBA.h:194 as written (see .cpp)
		bool literal_lc_edge_init = false;                                          // true = SRBA.h:294 as written (see .cpp: it diverges here)
	} parameters;

	// cumulative host-clock milliseconds of the four parts of every LM solve (run_lm): what an un-batched caller pays per keyframe
	struct TTiming { double window_build_ms = 0, batch_create_ms = 0, solve_ms = 0, readback_ms = 0; size_t solves = 0, iters = 0, max_opt_edges = 0, max_obs = 0; } timing;

	mySRBA(Context& ctx, const TSRBAStereoSLAMOptions& opts);
	~mySRBA();
	mySRBA(const mySRBA&) = delete;
	mySRBA& operator=(const mySRBA&) = delete;

	inline void setInitialKFPose(const Pose3D& pose) { m_initial_kf_pose = pose; }  // SRBA.h:95-97
	void loopClosureDetected(bool _lc = true) { m_lc = _lc; }                       // SRBA.h:85-87 (set by the estimator, read by nobody)
	void setLoopClosureOldID(size_t _id) { m_lc_old_kf_id = _id; }                  // SRBA.h:90-92
	// topological distance between two keyframes over the k2k edges; SIZE_MAX when farther than max_depth: what the
	// estimator reads from spanning_tree.sym.next_edge[from][to].distance (EST.cpp:1803-1812)
	size_t topo_distance(TKeyFrameID a, TKeyFrameID b, size_t max_depth) const;
	inline bool isKFLocalmapCenter(TKeyFrameID id) const { return m_localmap_kf_IDs.count(id) != 0; }        // SRBA.h:100-102
	inline TKeyFrameID getLocalmapCenterID(TKeyFrameID id) const { return *m_kf_localmap_center_ID.at(id).rbegin(); }   // SRBA.h:105-108

	// the local-map bookkeeping of edge_creation_policy (SRBA.h:67-68), for inspection
	const std::set<TKeyFrameID>& localmap_centers_of(TKeyFrameID id) const { return m_kf_localmap_center_ID.at(id); }
	const std::set<TKeyFrameID>* localmap_kfs(TKeyFrameID center) const { const auto it = m_localmap_kf_IDs.find(center); return it == m_localmap_kf_IDs.end() ? nullptr : &it->second; }

	// EST.cpp:166,782
	void define_new_keyframe(const new_kf_observations_t& obs, TNewKeyFrameInfo& out_new_kf_info, bool run_local_optimization);

	// Round 5: B engines (independent streams on ONE context) advanced in lock-step -- every engine takes one new keyframe; the graph bookkeeping
	// and both LM windows of every engine are prepared on host threads side by side; stage 1 (optimize_edges on the new edges, landmarks held) of ALL
	// engines is ONE ssk_lm_batch, stage 2 (optimize_local_area) another.  Every engine ends in the state define_new_keyframe would have left it in
	// (the same windows; the LM kernels' results do not depend on the batch around a window).
	struct TBatchTiming {
		double bookkeeping_and_windows_ms = 0, stage1_create_ms = 0, stage1_solve_ms = 0, stage2_refresh_ms = 0, stage2_create_ms = 0, stage2_solve_ms = 0, writeback_ms = 0;
		double stage1_symbolic_ms = 0, stage2_symbolic_ms = 0;
		int host_threads = 0, n_stage1 = 0, n_stage2 = 0, stage2_max_edges = 0, stage2_max_obs = 0;
		double stage2_mean_edges = 0, stage2_mean_obs = 0, stage1_mean_obs = 0, stage1_mean_iters = 0, stage2_mean_iters = 0;
		char stage1_kernel[96] = {0}, stage2_kernel[96] = {0};
		double teardown_ms = 0, total_ms = 0;               // round 6: the windows' host arrays freed (side by side); the whole call
	};
	static void define_new_keyframe_batch(const std::vector<mySRBA*>& engines, const std::vector<const new_kf_observations_t*>& obs,
	                                      std::vector<TNewKeyFrameInfo>& out, bool run_local_optimization, TBatchTiming* timing = nullptr);
	// the same with the observation list of engine i produced by fill(i, list) ON the host thread that does the engine's bookkeeping (a C caller's
	// records are converted there, side by side, instead of one list after the other in front of the call)
	static void define_new_keyframe_batch(const std::vector<mySRBA*>& engines, const std::function<void(int, new_kf_observations_t&)>& fill,
	                                      std::vector<TNewKeyFrameInfo>& out, bool run_local_optimization, TBatchTiming* timing = nullptr);

	// pose of `target` as seen from `root` by composing edges along the breadth-first tree (depth limited):
	// what create_complete_spanning_tree(root, ...)[target] gives the estimator (EST.cpp:909-915).
	bool get_kf_relative_pose(TKeyFrameID root, TKeyFrameID target, size_t max_depth, Pose3D& out) const;

	// rba.get_global_graphslam_problem(poseGraph) (EST.cpp:946-947): global poses of all keyframes reachable from `root`
	// (composition of the k2k edges along the breadth-first tree, no depth limit; unreachable keyframes keep the identity and
	// are reported in `reachable`) and one constraint per k2k edge: X_to^-1 X_from = inv_pose, as (i = to, j = from, Z).
	void get_global_graphslam_problem(TKeyFrameID root, std::vector<Pose3D>& nodes, std::vector<int32_t>& edge_ij, std::vector<Pose3D>& edge_z,
	                                  std::vector<bool>& reachable) const;

	struct rba_state_t {
		std::deque<k2k_edge_t> k2k_edges;
		std::unordered_map<TLandmarkID, landmark_t> unknown_lms;   // looked up per observation: hashed (nothing iterates it in order)
		std::vector<std::vector<kf_observation_t>> kf_observations;                // per keyframe
	};
	rba_state_t& get_rba_state() { return m_state; }
	const rba_state_t& get_rba_state() const { return m_state; }
	size_t num_keyframes() const { return m_state.kf_observations.size(); }

private:
	struct TNewEdgeInfo { size_t id; bool has_approx_init_val; };
	void edge_creation_policy(TKeyFrameID new_kf_id, const new_kf_observations_t& obs, std::vector<TNewEdgeInfo>& new_k2k_edge_ids);   // SRBA.h:144-308
	size_t create_kf2kf_edge(TKeyFrameID from, TKeyFrameID to);
	// breadth-first tree from `root`: for every keyframe within max_depth the step towards the root
	struct step_t { TKeyFrameID prev; size_t edge; size_t dist; };
	void bfs(TKeyFrameID root, size_t max_depth, std::map<TKeyFrameID, step_t>& out) const;
	struct lm_job;
	// window-local landmark index per landmark sequence number, valid where the stamp equals the job's generation (no hash per observation,
	// no clearing); two slots: the new-edges window and the local-area window may be built at the same time
	mutable std::vector<uint32_t> m_lm_stamp[2];
	mutable std::vector<int32_t> m_lm_local[2];
	mutable uint32_t m_lm_gen[2] = {0, 0};
	uint32_t m_n_lm_seq = 0;
	// depth-limited spanning trees per observing keyframe, flattened (step towards the observer per keyframe id), kept across keyframes:
	// every new edge joins the NEW keyframe to an older one, so the paths from an observer to the (older) base keyframes of its landmarks
	// only change when a keyframe arrives with several edges -- then the cache is dropped.  Per slot as the landmark scratch.
	mutable std::unordered_map<TKeyFrameID, std::vector<step_t>> m_tree_cache[2];
	// one persistent helper thread (a thread's first HIP call costs about a millisecond: not one thread per keyframe)
	struct worker_t;
	worker_t* m_worker = nullptr;
	void lm_prepare(lm_job& j, const std::set<size_t>& opt_edges, const std::vector<std::pair<TKeyFrameID, size_t>>& obs_refs, bool optimize_landmarks, bool create) const;
	void lm_create(lm_job& j) const;
	void lm_launch(lm_job& j, bool robust, bool refresh);
	void lm_collect(lm_job& j, TOptimizeResults& res);
	// the pieces of define_new_keyframe the lock-step batch form shares with it
	static void define_new_keyframe_batch_impl(const std::vector<mySRBA*>& engines, const std::vector<const new_kf_observations_t*>* obs,
	                                           const std::function<void(int, new_kf_observations_t&)>* fill, std::vector<TNewKeyFrameInfo>& out,
	                                           bool run_local_optimization, TBatchTiming* timing);
	void kf_begin(const new_kf_observations_t& obs, TNewKeyFrameInfo& out, std::vector<TNewEdgeInfo>& new_edges);
	void stage1_prepare(lm_job& j, TKeyFrameID new_kf_id, const std::vector<TNewEdgeInfo>& new_edges, bool create) const;
	void stage2_prepare(lm_job& j, TKeyFrameID new_kf_id, bool create) const;
	void lm_apply(lm_job& j, const ssk_lm_result& r, TOptimizeResults& res);      // a solved window's state (j.pose / j.lmpos) to the map

	Context& m_ctx;
	TSRBAStereoSLAMOptions m_opts;
	rba_state_t m_state;
	std::vector<std::vector<std::pair<TKeyFrameID, size_t>>> m_adj;                // keyframe -> (neighbour, edge id)
	Pose3D m_initial_kf_pose;
	bool m_lc = false;
	size_t m_lc_old_kf_id = 0;
	std::deque<std::set<TKeyFrameID>> m_kf_localmap_center_ID;                     // SRBA.h:67
	std::map<TKeyFrameID, std::set<TKeyFrameID>> m_localmap_kf_IDs;                // SRBA.h:68
};

}  // namespace ssk
