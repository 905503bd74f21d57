// ssk_frontend.h -- device-resident keyframe batch and the front-end kernel launchers.
#pragma once
#include "ssk_internal.h"

constexpr int SSK_EDGE = 31;         // cv::ORB edgeThreshold: keypoints only in [31, W-31) x [31, H-31)
constexpr int SSK_MAX_FEATS = 2048;  // n_feats limit (reference: 500)
constexpr int SSK_FAST_CTR_STRIDE = 64;                      // ints between the per-XCD tile counters of ssk_launch_fast (256 bytes: own cache lines)
constexpr int SSK_FAST_CTR_INTS = 9 * SSK_FAST_CTR_STRIDE;   // behind ncand[B*2]: a stride of padding, then the eight counters

// HBM layout of a batch of B keyframes (SoA; everything the per-keyframe hot path produces stays here):
//   img     u8   [B*2][H][pitch]      eye-major inside a keyframe (left, right); pitch % 128 == 0
//   cand    u32  [B*2][cand_cap]      FAST+NMS candidates, packed score<<24 | (4095-y)<<12 | (4095-x)
//   ncand   i32  [B*2]
//   kp_xy   i16x2[B*2][N]   kp_resp f32 [B*2][N]   kp_angle f32 [B*2][N]   kp_cs f32x2 [B*2][N]   nkp i32 [B*2]
//   desc    u8   [B*2][N][32]
//   m_q,m_t,m_d i32 [B][N]   nm i32 [B]          L<->R matches in left-keypoint order
struct ssk_kfbatch {
	ssk_ctx* ctx = nullptr;
	int n_kf = 0, w = 0, h = 0, pitch = 0, max_feats = 0, cand_cap = 0;
	uint8_t* img = nullptr;
	uint32_t* cand = nullptr;
	int32_t* ncand = nullptr;
	short2* kp_xy = nullptr;
	float* kp_resp = nullptr;
	float* kp_angle = nullptr;
	float2* kp_cs = nullptr;
	int32_t* nkp = nullptr;
	uint8_t* desc = nullptr;
	int32_t *m_q = nullptr, *m_t = nullptr, *m_d = nullptr, *nm = nullptr;
	// match-index gather lists for batched data association: left-keypoint row (absolute in desc pool)
	// and its pixel coordinates per L<->R match, written by the stereo kernel
	int32_t* m_row = nullptr;    // [B][N]   absolute descriptor row of the match's left keypoint
	float2* m_xy = nullptr;      // [B][N]   left keypoint pixel of the match
	float4* m_obs = nullptr;     // [B][N]   stereo observation (xl, yl, xr, yr) of the match (change-in-pose filter)
	// bag-of-words place recognition (ssk_bow.hip), allocated by the first transform on the batch
	int32_t* bow_word = nullptr; // [B][N]   ascending word ids of the left descriptors
	double* bow_val = nullptr;   // [B][N]   L1-normalised values
	int32_t* bow_n = nullptr;    // [B]
	int32_t* q_ids = nullptr;    // [B][SSK_BOW_MAX_RESULTS] last query: entry ids, scores, counts
	double* q_scores = nullptr;
	int32_t* q_n = nullptr;
};

void ssk_kfbatch_bow_free(ssk_kfbatch* b);
int ssk_launch_fast(ssk_ctx* ctx, ssk_kfbatch* b, int img_begin, int img_end, int threshold);
int ssk_launch_select(ssk_ctx* ctx, ssk_kfbatch* b, int img_begin, int img_end, const ssk_frontend_params& p);
int ssk_launch_describe(ssk_ctx* ctx, ssk_kfbatch* b, int img_begin, int img_end, int n_feats);
int ssk_launch_stereo(ssk_ctx* ctx, ssk_kfbatch* b, int kf_begin, int kf_end, const ssk_frontend_params& p);
// change-in-pose filter over pairs (ssk_cip.hip); obs arrays are indexed by match slot (kf * max_feats + k)
int ssk_launch_cip(ssk_ctx* ctx, const int32_t* d_train_idx, int32_t* d_status, const int32_t* d_out_beg, const int32_t* d_q_beg,
                   const int32_t* d_q_end, const int32_t* d_t_beg, const float4* d_this_obs, const float4* d_other_obs,
                   const double* d_init_poses, int n_pairs, int cap, const ssk_cip_params& prm, double* d_residuals, ssk_cip_result* d_results,
                   int32_t* d_tracked);
// fundamental-matrix RANSAC filter over pairs (ssk_fmat.hip); xy arrays are indexed by match slot
int ssk_launch_fm(ssk_ctx* ctx, const int32_t* d_train_idx, int32_t* d_status, const int32_t* d_out_beg, const int32_t* d_q_beg,
                  const int32_t* d_q_end, const int32_t* d_t_beg, const float2* d_this_xy, const float2* d_other_xy,
                  int n_pairs, int cap, const ssk_fm_params& prm, ssk_fm_result* d_results, int32_t* d_tracked,
                  const unsigned long long* d_seeds = nullptr);   // d_seeds: per-pair RANSAC seeds (NULL: prm.seed + pair)
