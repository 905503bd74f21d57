// ssk_internal.h -- shared internals of libssk_hip.so (context, error plumbing, arenas).
// Product code: nothing here may include or call anything under oracle/.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <string>
#include <vector>
#include "../../include/ssk_capi.h"

// Growable bump arena (device or pinned-host).  Host entry points stage one keyframe's worth of
// data through these; batched "_dev" entry points never touch them.
struct ssk_arena {
	void* base = nullptr;
	size_t cap = 0, used = 0;
	bool pinned_host = false;
};

// The library's environment switches, read ONCE per context (ssk_create) -- not per call -- and again only when the caller asks (ssk_env_refresh: the
// Python binding does so in front of an LM batch create, so that tests and same-box A/Bs can switch kernels on a live context).  Defaults = what
// DESIGN.md 6's numbers were measured with; the switches are listed in DESIGN.md 5 "Switches".
struct ssk_env {
	std::string lm_kernel;              // SSK_LM_KERNEL: "" (default: node / phase-split / pose-only), split, node, tiled, coop, acc, l2
	bool lm_few_split = true;           // SSK_LM_FEW_SPLIT=0: a batch of at most 8 windows keeps the node kernel for every window that fits it
	bool lm_pose_only = true;           // SSK_LM_POSE_ONLY=0: pose-only windows on the node kernel
	bool lm_no_pinned = false;          // SSK_LM_NO_PINNED: the batch's tables in pageable memory
	bool lm_sync_upload = false;        // SSK_LM_SYNC_UPLOAD: a lean create waits for its uploads before it returns (the form up to round 5)
	bool lm_order = true;               // SSK_LM_ORDER=0: windows launched as given (default: largest first)
	bool lm_debug = false;              // SSK_LM_DEBUG
	int sp_spread = -1;                 // SSK_SP_SPREAD: 1 = a phase-split window's tiles over all XCDs, 0 = on one XCD, unset: over all for batches of at most 4 such windows
	int lm_schur = 0;                   // SSK_LM_SCHUR: 0 default (matrix cores for E > 12), 1 mfma, 2 sparse
	bool lm_ldl_mfma = false;           // SSK_LM_LDL=mfma
	int lm_coop_g = 0;                  // SSK_LM_COOP_G (0: default, at most 32)
	bool lm_coop_test_timeout = false, lm_coop_no_fallback = false;      // test hooks of the cooperative kernel's barrier time-out
	bool sp_no_poll = false;            // SSK_SP_NO_POLL
	bool k2_regtile = false;            // SSK_K2_SOLVER=regtile
	int lm_build_threads = 0;           // SSK_LM_BUILD_THREADS (0: from the core count and the cgroup's CPU quota)
};
void ssk_env_read(ssk_env& e);

struct ssk_ctx {
	int device = -1;
	ssk_env env;
	hipStream_t stream = nullptr;
	bool own_stream = false;
	hipEvent_t ev_start = nullptr, ev_stop = nullptr;
	hipStream_t side_stream = nullptr;   // a composite LM batch's few full-builder windows run here, beside the many on `stream` (ssk_capi_lm.hip)
	hipEvent_t ev_side_a = nullptr, ev_side_b = nullptr;
	hipEvent_t ev_lm_up = nullptr;     // end of an LM batch's uploads (ssk_capi_lm.hip: create waits for this, not for the zero fill behind it)
	bool lm_up_pending = false;        // a lean create returned with its uploads (from pin_lm) still queued: waited for before pin_lm is written or freed
	std::string err;
	ssk_arena dev, pin;
	ssk_arena pin_lm;                  // pinned host memory the tables of an LM batch are built in (ssk_capi_lm.hip: one DMA per table, no pageable staging)
	int pin_lm_small_streak = 0;       // consecutive creates that used less than a quarter of pin_lm (eight: the arena is given back)
	std::vector<void*> retired;        // arenas replaced by bigger ones; freed at destroy / next sync
	// direction-filter bin thresholds (see ssk_da_filters.hip): smallest float quotient mapped to
	// bin >= b for b = 9..27, computed once on the host with the host libm's atanf.
	float dir_thresh[19];
	float* d_dir_thresh = nullptr;
	int8_t pattern[1024];              // BRIEF sampling pattern (default: include/ssk_brief_pattern.h)
	int8_t* d_pattern = nullptr;       // [1024] int8, then [1024] float (ssk_upload_pattern)
	int orb_steering = 1;              // C7: 1 = cv::ORB's (float)cos / (float)sin of the fastAtan2 angle (include/ssk_sincos.h), 0 = exact centroid direction
	struct ssk_kfbatch* single = nullptr;   // 1-keyframe batch behind the host drop-in ssk_frontend_run
	// device slabs of destroyed LM batches, reused by the next ones (the host drop-in creates and destroys two batches per keyframe -- the new edge's
	// window and the local area's, the second one built while the first is solved: no hipMalloc in steady state)
	static constexpr int LM_SLABS = 3;     // (round 5: three -- a lock-step batch of streams alternates a small and a growing large batch; with two places the
	                                       // large one's regrowth evicted the small one's slab every time: a hipMalloc of tens of ms in each create)
	void* lm_slab[LM_SLABS] = {nullptr, nullptr, nullptr};
	size_t lm_slab_bytes[LM_SLABS] = {0, 0, 0};
	int n_cu = 256;
	// per-stage HIP-event profiler (ssk_profile_*): event pairs recorded around kernel launches on the
	// context stream while enabled; resolved into per-stage sums at ssk_profile_read.
	bool prof_on = false;
	std::vector<hipEvent_t> prof_pool;      // created lazily, reused
	size_t prof_used = 0;
	struct prof_rec { int stage; hipEvent_t a, b; };
	std::vector<prof_rec> prof_open;
	double prof_ms[SSK_STAGE_COUNT] = {0};
	long long prof_calls[SSK_STAGE_COUNT] = {0};
};

// "once per device" for settings that belong to a device, not to the process (hipFuncSetAttribute: a second context on another GPU needs its own
// call); two threads that both see `need` set the same value twice, which is harmless
struct ssk_dev_once {
	std::atomic<unsigned long long> mask{0};
	bool need(int dev) const { return dev < 0 || dev >= 64 || !((mask.load(std::memory_order_acquire) >> dev) & 1ull); }
	void done(int dev) { if (dev >= 0 && dev < 64) mask.fetch_or(1ull << dev, std::memory_order_release); }
};

hipEvent_t ssk_prof_event(ssk_ctx* ctx);
struct ssk_stage_scope {
	ssk_ctx* ctx; int stage; hipEvent_t a = nullptr;
	ssk_stage_scope(ssk_ctx* c, int s) : ctx(c), stage(s)
	{
		if (ctx->prof_on) { a = ssk_prof_event(ctx); if (a) (void)hipEventRecord(a, ctx->stream); }
	}
	~ssk_stage_scope()
	{
		if (a) {
			hipEvent_t b = ssk_prof_event(ctx);
			if (b) { (void)hipEventRecord(b, ctx->stream); ctx->prof_open.push_back({stage, a, b}); }
		}
	}
};

int ssk_set_error(ssk_ctx* ctx, int code, const char* fmt, ...);

#define SSK_HIP(ctx, call)                                                                         \
	do {                                                                                           \
		hipError_t e__ = (call);                                                                   \
		if (e__ != hipSuccess)                                                                     \
			return ssk_set_error((ctx), SSK_ERR_HIP, "%s failed: %s (%s:%d)", #call,               \
			                     hipGetErrorString(e__), __FILE__, __LINE__);                      \
	} while (0)

#define SSK_LAUNCH_CHECK(ctx, name)                                                                \
	do {                                                                                           \
		hipError_t e__ = hipGetLastError();                                                        \
		if (e__ != hipSuccess)                                                                     \
			return ssk_set_error((ctx), SSK_ERR_HIP, "launch of %s failed: %s", name,              \
			                     hipGetErrorString(e__));                                          \
	} while (0)

// arena helpers: reset at the start of a host entry point, then carve 256-B aligned pieces.
// sized_step != 0: capacity = the need + a quarter, rounded up to the step (not the next power of two); a failed pinned allocation is retried at the exact need
int ssk_arena_reserve(ssk_ctx* ctx, ssk_arena& a, size_t bytes, size_t sized_step = 0);
hipError_t ssk_upload_pattern(ssk_ctx* ctx);   // ctx->pattern -> ctx->d_pattern (int8 table + float copy)
inline void ssk_arena_reset(ssk_arena& a) { a.used = 0; }
template <typename T>
inline T* ssk_arena_take(ssk_arena& a, size_t count)
{
	size_t off = (a.used + 255) & ~size_t(255);
	a.used = off + count * sizeof(T);
	return reinterpret_cast<T*>(static_cast<char*>(a.base) + off);
}
inline size_t ssk_pad(size_t bytes) { return (bytes + 255) & ~size_t(255); }

// One transfer per direction for a host entry point (a small hipMemcpyAsync costs ~8 us of host time, a keyframe pair's data
// association used 26 of them): the inputs are laid out back to back in the pinned arena and go to the same layout in the
// device arena with one copy; the outputs are carved from one device block that comes back with one copy.
// Pieces that go both ways (a status array a filter updates in place) sit between the inputs and the outputs.
// Reserve both arenas for the total first (ssk_pad per piece).
struct ssk_xfer_pack {
	uint8_t *h = nullptr, *d = nullptr;
	size_t off = 0;
	void begin(ssk_ctx* ctx, size_t cap) { h = ssk_arena_take<uint8_t>(ctx->pin, cap); d = ssk_arena_take<uint8_t>(ctx->dev, cap); off = 0; }
	template <typename T>
	T* slot(size_t count, T** host)                    // device address of a piece; *host = its mirror in the pinned block
	{
		*host = reinterpret_cast<T*>(h + off);
		T* r = reinterpret_cast<T*>(d + off);
		off += ssk_pad(count * sizeof(T));
		return r;
	}
	template <typename T>
	T* put(const T* src, size_t count) { T* hv; T* r = slot<T>(count, &hv); if (count) std::memcpy(hv, src, count * sizeof(T)); return r; }
	// bytes [from, to) of the block, host -> device / device -> host
	hipError_t upload(hipStream_t s, size_t from, size_t to) const { return to > from ? hipMemcpyAsync(d + from, h + from, to - from, hipMemcpyHostToDevice, s) : hipSuccess; }
	hipError_t download(hipStream_t s, size_t from, size_t to) const { return to > from ? hipMemcpyAsync(h + from, d + from, to - from, hipMemcpyDeviceToHost, s) : hipSuccess; }
};

// kernels' host-side launchers shared between translation units.  Pair p works on virtual rows
// [beg[p], end[p]) of each side; out_beg[p] (NULL = q_beg) is the output row of its first query.
int ssk_launch_hamming(ssk_ctx* ctx, const uint8_t* d_q, const int32_t* d_q_beg, const int32_t* d_q_end,
                       const int32_t* d_q_gather, const uint8_t* d_t, const int32_t* d_t_beg, const int32_t* d_t_end,
                       const int32_t* d_t_gather, const int32_t* d_out_beg,
                       int n_pairs, int max_nq, int32_t* d_train_idx, int32_t* d_dist);
int ssk_launch_da_filters(ssk_ctx* ctx, const int32_t* d_train_idx, const int32_t* d_dist, const int32_t* d_out_beg,
                          const int32_t* d_q_beg, const int32_t* d_q_end, const int32_t* d_t_beg, const int32_t* d_t_end,
                          const float* d_this_xy, const float* d_other_xy,
                          int n_pairs, const ssk_da_params& prm, int32_t* d_status, int32_t* d_tracked);
