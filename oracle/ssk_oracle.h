/*
 * ssk_oracle.h -- CPU ORACLE of the srba-stereo-slam per-keyframe hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may call it; libssk_hip.so never links or loads it.
 *
 * PARITY STATUS: *** parity unpinned ***.  The reference's arithmetic for this path lives in
 * un-vendored third-party libraries that are absent from this image and from /root/reference
 * (stereo-vo [unpinned], OpenCV 2.4.x, MRPT/srba [unpinned], MRPT >= 1.3; reference README.md:12-16),
 * the reference tree holds no tests, fixtures or golden vectors, and its build
 * needs those libraries (src/CMakeLists.txt:3-7), so no oracle/_ref build exists.  What is restated
 * LITERALLY from in-tree reference code is marked [literal file:line]; what restates the published
 * algorithm of an absent dependency is marked [published: ...] and every choice the publication
 * leaves open is fixed here, once, and documented in DESIGN.md section "Oracle contract".
 *
 * Plain C99, single-threaded, no dependencies beyond libc/libm.  Build: oracle/Makefile
 * (-O2 -ffp-contract=off: no FMA contraction, so float results are the IEEE sequence written here).
 */
#ifndef SSK_ORACLE_H
#define SSK_ORACLE_H

#include <stdint.h>
#include "../include/ssk_capi.h"   /* record layouts only (ssk_keypoint, ssk_dmatch, params) */

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- data association (oracle_da.c) ---------------- */

/* [published: OpenCV 2.4 BFMatcher(NORM_HAMMING,crossCheck=false).match; call site EST.cpp:1447-1452]
 * returns number of matches written (nq, or 0 when nq==0 or nt==0). */
int ssko_hamming_match(const uint8_t* query, int nq, const uint8_t* train, int nt, ssk_dmatch* out);

/* [literal EST.cpp:1883-1946] direction histogram filter.  status in/out (SSK_DA_*).
 * this_xy[2*k..] = left keypoint of THIS kf for match k; other_xy[2*j..] = left keypoint of the
 * OTHER kf for its match j (indexed through matches[k].trainIdx).  offset = image rows. */
void ssko_da_filter_direction(const ssk_dmatch* matches, int n_this, const float* this_xy,
                              const float* other_xy, int offset, int32_t* status);

/* [literal EST.cpp:1951-2010] ORB-distance threshold + serial uniqueness scan (quirk Q1:
 * other_matched[].second stores ++stage1_counter, not the distance). */
void ssko_da_filter_orb_distance(const ssk_dmatch* matches, int n_this, int n_other,
                                 double max_orb_distance_da, int32_t* status);

/* [literal EST.cpp:1413-1515,1647-1669] gather (Q3) + match + the two filters. returns tracked count.
 * out_matches/out_status capacity: this_kf->n_matches. */
int ssko_data_association(const ssk_kf_view* this_kf, const ssk_kf_view* other_kf,
                          const ssk_da_params* params, ssk_dmatch* out_matches, int32_t* out_status);

/* [literal EST.cpp:572-617] landmark-ID propagation for a new keyframe.
 * n_matches: matches in the new KF; n_sim: similar KFs, visited in the order given (sortedIndices
 * order already applied by the caller).  tracking_first[s*n_matches+m] = other match index or -1
 * (t_kf_da_info::tracking_info[m].first).  other_ids[s] -> that KF's m_matches_ID array.
 * last_match_id in/out (m_last_match_ID).  out_ids[m] = new_kf.m_matches_ID[m].
 * out_tracked_per_kf[s] = vectorNumberTrackedFeats. returns numberTrackedFeats. */
int ssko_propagate_ids(int n_matches, int n_sim, const int32_t* tracking_first,
                       const uint64_t* const* other_ids, uint64_t* last_match_id,
                       uint64_t* out_ids, int32_t* out_tracked_per_kf);

/* ---------------- change-in-pose filter (oracle_cip.c): contract P1-P6 in that file ---------------- */
/* this_obs / other_obs: (xl, yl, xr, yr) per match; train_idx[k] = other match of this match k; status in/out.
 * residuals (optional): one per tracked pair, in match order.  Returns out->valid. */
int ssko_change_in_pose(const float* this_obs, int n_this, const float* other_obs, int n_other,
                        const int32_t* train_idx, int32_t* status, const double init_pose[12],
                        const ssk_cip_params* P, double* residuals, ssk_cip_result* out);

/* ---------------- fundamental-matrix filter (oracle_fmat.c): contract F1-F6 in that file ---------------- */
void ssko_fm_sample(uint64_t seed, int h, int n, int idx[7]);
int ssko_fm_cubic_roots(double c3, double c2, double c1, double c0, double* roots);
int ssko_fm_seven_point(const double* m1, const double* m2, const int idx[7], const ssk_fm_params* P, double F[3][9]);
int ssko_fm_inlier(const double* F, const double* p1, const double* p2, double thr2);
int ssko_fm_update_iters(double p, int good, int n, int max_iters);
/* this_xy / other_xy: left keypoint pixel per match; status in/out; returns out->valid */
int ssko_fundamental_filter(const float* this_xy, int n_this, const float* other_xy, int n_other, const int32_t* train_idx,
                            int32_t* status, const ssk_fm_params* P, ssk_fm_result* out);

/* ---------------- landmark initialisation (oracle_geom.c) ---------------- */

typedef struct ssko_stereo_cam {   /* mrpt::utils::TStereoCamera fields the path reads */
	double fx_l, fy_l, cx_l, cy_l;
	double fx_r, fy_r, cx_r, cy_r;
	double baseline;               /* rightCameraPose[0] */
} ssko_stereo_cam;

/* [literal UTIL.h:558-574] */
void ssko_project_match_to_3d(double ul, double vl, double ur, const ssko_stereo_cam* cam, double out_xyz[3]);

/* ---------------- stereo front-end (oracle_orb.c): contract C1-C10 in that file ---------------- */
void ssko_fast_score_map(const uint8_t* img, int w, int h, int stride, int threshold, uint8_t* score /* w*h */);
int ssko_fast_candidates(const uint8_t* img, int w, int h, int stride, int threshold, uint32_t* keys, int cap);
float ssko_harris_response(const uint8_t* img, int stride, int x, int y, float harris_k);
float ssko_atan2_deg(float y, float x);
/* C7 steering of the descriptors: 1 (default) = cv::ORB's cos / sin of the angle (include/ssk_sincos.h), 0 = exact centroid direction */
void ssko_set_orb_steering(int mode);
void ssko_orient_describe(const uint8_t* img, int stride, int x, int y, const int8_t* pattern,
                          float* angle_deg, uint8_t* desc32);
int ssko_orb_detect(const uint8_t* img, int w, int h, int stride, const ssk_frontend_params* prm,
                    const int8_t* pattern, ssk_keypoint* kp, uint8_t* desc, int cap);
int ssko_stereo_match(const ssk_keypoint* kl, int nl, const uint8_t* dl, const ssk_keypoint* kr, int nr,
                      const uint8_t* dr, const ssk_frontend_params* prm, ssk_dmatch* out);
int ssko_frontend(const uint8_t* left, const uint8_t* right, int w, int h, int stride,
                  const ssk_frontend_params* prm, const int8_t* pattern, ssk_kf_out* out);

/* ---------------- relative bundle adjustment (oracle_lm.c): contract L1-L5 in that file ---------------- */
void ssko_se3_left_update(const double* eps6, const double* T12, double* out12);
double ssko_lm_total_error(const ssk_lm_window* W, const ssk_lm_params* P, const double* poses, const double* lms);
void ssko_lm_gradient(const ssk_lm_window* W, const ssk_lm_params* P, const double* poses, const double* lms,
                      double* gp, double* gf);
int ssko_lm_solve(const ssk_lm_window* W, const ssk_lm_params* P, double* poses, double* lms,
                  ssk_lm_result* res, ssk_lm_trace_entry* trace, double* dp_trace, double* df_trace);
void ssko_build_observations(const ssk_keypoint* kl, const ssk_keypoint* kr, const ssk_dmatch* m, int n,
                             const ssk_lm_params* P, float* obs_px, double* init_pos);

/* ---------------- bag-of-words place recognition (oracle_bow.c; contract W1-W5 there) ---------------- */
typedef struct ssko_bow_vocab ssko_bow_vocab;
typedef struct ssko_bow_db ssko_bow_db;
/* [published: DBoW2 1.0.2 TemplatedVocabulary::load; file demo/voc.yml.gz] records in FILE ORDER (root excluded):
 * node_id > 0, parent_id >= 0, weight, 32-byte descriptor with bit m + 8k of the DBoW2 bitset in bit m of byte k
 * (BOW.h:95-107); word_node[w] = nodeId of word w.  NULL on malformed input. */
ssko_bow_vocab* ssko_bow_vocab_create(int k, int L, int n_rec, const int32_t* node_id, const int32_t* parent_id,
                                      const double* weight, const uint8_t* desc, int n_words, const int32_t* word_node);
void ssko_bow_vocab_destroy(ssko_bow_vocab* v);
/* [published: TemplatedVocabulary::transform(feature, id, weight)] word id of one descriptor (W2) */
int ssko_bow_word(const ssko_bow_vocab* v, const uint8_t* feat, double* weight);
/* [published: TemplatedVocabulary::transform(features, BowVector), TF_IDF + L1; call BOW.h:73-76,86-89] (W3) */
int ssko_bow_transform(const ssko_bow_vocab* v, const uint8_t* feats, int n, int32_t* out_words, double* out_values);
/* [published: TemplatedDatabase::add / queryL1; calls BOW.h:76,89, EST.cpp:118,426,919] (W4, W5) */
ssko_bow_db* ssko_bow_db_create(int n_words);
void ssko_bow_db_destroy(ssko_bow_db* d);
int ssko_bow_db_size(const ssko_bow_db* d);
int ssko_bow_db_add(ssko_bow_db* d, const int32_t* words, const double* values, int n);
int ssko_bow_db_query(const ssko_bow_db* d, const int32_t* words, const double* values, int n, int max_results, int max_id,
                      int32_t* out_ids, double* out_scores);

/* ---------------- keyframe-insertion / loop-closure policy (oracle_policy.c), all [literal] ---------------- */
double ssko_update_translation_threshold(double x, double th);     /* UTIL.cpp:153-161 */
double ssko_update_rotation_threshold(double x, double th);        /* UTIL.cpp:163-171 */
void ssko_vo_check(double inc_translation_kf, double inc_rotation_kf, double incr_translation, double incr_rotation,
                   int64_t tracked_feats_from_last_kf, double max_translation_limit, double max_rotation_limit,
                   double max_translation, double max_rotation, int vo_id_tracking_th, int32_t* out4);   /* EST.cpp:366-392 */
int ssko_get_similar_kfs(int64_t new_kf_id, int n_q, const int64_t* q_ids, const double* q_scores, const int64_t* topo,
                         int64_t submap_size, int64_t lc_distance, int64_t* similar, int* n_similar, int64_t* lc_id);   /* EST.cpp:1737-1826 */
void ssko_kf_decision(int insert_in, int n_da, const int64_t* da_kf_idx, const int64_t* da_tracked, int potential_lc, int64_t lc_id,
                      int vo_force_new_kf, int64_t updated_matches_th, int64_t up_matches_th_plus, double* th_inout,
                      int64_t* sorted, int64_t* highest_out, int32_t* flags_out3);   /* EST.cpp:437-548 */

/* ---------------- global pose-graph optimisation (oracle_posegraph.c; contract G1-G6 there) ---------------- */
void ssko_so3_log(const double* T12, double* w3);
double ssko_pg_cost(int n_edges, const int32_t* edge_ij, const double* edge_z, const double* X);
/* [published: MRPT optimize_graph_spa_levmarq on rba.get_global_graphslam_problem; call EST.cpp:945-957] */
int ssko_pg_optimize(int n_nodes, double* X, int n_edges, const int32_t* edge_ij, const double* edge_z, int fixed_node,
                     const ssk_pg_params* P, ssk_pg_result* out);

#ifdef __cplusplus
}
#endif
/* ---- oracle_edge_policy.c: mySRBA::edge_creation_policy, SRBA.h:144-308 ---- */
typedef struct ssko_ecp ssko_ecp;
ssko_ecp* ssko_ecp_create(int submap_size, int min_obs_to_loop_closure, int max_tree_depth, int max_optimize_depth);
void ssko_ecp_destroy(ssko_ecp* s);
void ssko_ecp_set_initial_kf_pose(ssko_ecp* s, const double* pose12);
void ssko_ecp_set_edge_pose(ssko_ecp* s, int e, const double* pose12);
int ssko_ecp_num_edges(const ssko_ecp* s);
int ssko_ecp_get_edge(const ssko_ecp* s, int e, int* from, int* to, double* pose12);
int ssko_ecp_kf_centers(const ssko_ecp* s, int kf, int* out8);
int ssko_ecp_localmap_kfs(const ssko_ecp* s, int center, int* out64);
int ssko_ecp_define_kf(ssko_ecp* s, const uint64_t* feat_ids, int n_obs);

#endif
