// ssk_context.hip -- context, error text, arenas, stream + HIP-event timer of libssk_hip.so.
#include "ssk_internal.h"
#include <algorithm>
#include <cstdlib>
#include <cmath>
#include "../../include/ssk_brief_pattern.h"

void ssk_kfbatch_destroy_internal(struct ssk_kfbatch* b);

static std::string g_create_error;

hipEvent_t ssk_prof_event(ssk_ctx* ctx)
{
	if (ctx->prof_used == ctx->prof_pool.size()) {
		hipEvent_t e = nullptr;
		if (hipEventCreate(&e) != hipSuccess) return nullptr;
		ctx->prof_pool.push_back(e);
	}
	return ctx->prof_pool[ctx->prof_used++];
}

int ssk_set_error(ssk_ctx* ctx, int code, const char* fmt, ...)
{
	char buf[1024];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	if (ctx) ctx->err = buf; else g_create_error = buf;
	return code;
}

void ssk_env_read(ssk_env& e)
{
	auto str = [](const char* k) { const char* v = std::getenv(k); return std::string(v ? v : ""); };
	auto off = [](const char* k) { const char* v = std::getenv(k); return v && std::atoi(v) == 0; };      // "NAME=0" switches a default off
	auto set = [](const char* k) { return std::getenv(k) != nullptr; };
	e = ssk_env();
	e.lm_kernel = str("SSK_LM_KERNEL");
	e.lm_few_split = !off("SSK_LM_FEW_SPLIT");
	e.lm_pose_only = !off("SSK_LM_POSE_ONLY");
	e.lm_no_pinned = set("SSK_LM_NO_PINNED");
	e.lm_sync_upload = set("SSK_LM_SYNC_UPLOAD");
	e.lm_order = !off("SSK_LM_ORDER");
	e.lm_debug = set("SSK_LM_DEBUG");
	if (const char* v = std::getenv("SSK_SP_SPREAD")) e.sp_spread = std::atoi(v) ? 1 : 0;
	{ const std::string v = str("SSK_LM_SCHUR"); e.lm_schur = v == "mfma" ? 1 : v == "sparse" ? 2 : 0; }
	e.lm_ldl_mfma = str("SSK_LM_LDL") == "mfma";
	if (const char* v = std::getenv("SSK_LM_COOP_G")) e.lm_coop_g = std::max(1, std::min(64, std::atoi(v)));
	e.lm_coop_test_timeout = set("SSK_LM_COOP_TEST_TIMEOUT");
	e.lm_coop_no_fallback = set("SSK_LM_COOP_NO_FALLBACK");
	e.sp_no_poll = set("SSK_SP_NO_POLL");
	e.k2_regtile = str("SSK_K2_SOLVER") == "regtile";
	if (const char* v = std::getenv("SSK_LM_BUILD_THREADS")) e.lm_build_threads = std::max(1, std::atoi(v));
}

int ssk_env_refresh(ssk_ctx* ctx)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	ssk_env_read(ctx->env);
	return SSK_OK;
}

int ssk_arena_reserve(ssk_ctx* ctx, ssk_arena& a, size_t bytes, size_t sized_step)
{
	bytes += 4096;
	if (bytes <= a.cap) return SSK_OK;
	size_t ncap = a.cap ? a.cap : (size_t(1) << 20);
	while (ncap < bytes) ncap *= 2;
	// sized_step (the LM batches' pinned arena, hundreds of MB to GBs): a quarter of headroom rounded up to the step instead of the next power of
	// two -- a need of 2.1 GB pinned 4 GB, one of exactly 4 GB would have pinned 8
	if (sized_step) ncap = (bytes + bytes / 4 + sized_step - 1) / sized_step * sized_step;
	// the old block may still be referenced by queued work: make the stream idle first
	SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (a.base) {
		if (a.pinned_host) SSK_HIP(ctx, hipHostFree(a.base)); else SSK_HIP(ctx, hipFree(a.base));
		a.base = nullptr; a.cap = 0;
	}
	if (a.pinned_host) {
		hipError_t e = hipHostMalloc(&a.base, ncap, hipHostMallocDefault);
		if (e != hipSuccess && sized_step && ncap > bytes) {        // (locked-memory limits: the exact need may still fit)
			(void)hipGetLastError();
			ncap = (bytes + 4095) & ~(size_t)4095;
			e = hipHostMalloc(&a.base, ncap, hipHostMallocDefault);
		}
		if (e != hipSuccess) { a.base = nullptr; a.cap = 0; a.used = 0; SSK_HIP(ctx, e); }
	}
	else SSK_HIP(ctx, hipMalloc(&a.base, ncap));
	a.cap = ncap;
	a.used = 0;
	return SSK_OK;
}

// ---- direction-filter thresholds -------------------------------------------------------------------
// The reference bins a match direction as  floor((180/M_PI*atanf(q)+180.)/10)  (EST.cpp:1915-1916) with
// the C library's atanf.  atanf is monotone, so the bin is a step function of the float quotient q.
// We find, with THIS host's libm, the smallest float q whose bin is >= b for b = 9..27 (bins 8..27 are
// the only reachable ones: atanf in [-pi/2,pi/2] rounds to ang in [89.9999975, 270.0000025]).  The GPU
// then bins by comparing q against the 19 thresholds: identical bins to the libm formula for every
// float q, with no device atanf involved.
static int ref_bin(float q)
{
	const double ang = 180 / M_PI * atanf(q) + 180.;
	return (int)std::floor(ang / 10);
}
static inline int32_t float_to_ordered(float f)
{
	int32_t i; std::memcpy(&i, &f, 4);
	return i >= 0 ? i : (int32_t)(0x80000000u - (uint32_t)i);   // monotone in f, -0 == +0 -> 0
}
static inline float ordered_to_float(int32_t o)
{
	int32_t i = o >= 0 ? o : (int32_t)(0x80000000u - (uint32_t)o);
	float f; std::memcpy(&f, &i, 4);
	return f;
}
static void compute_dir_thresholds(float out[19])
{
	const int32_t lo_all = float_to_ordered(-INFINITY), hi_all = float_to_ordered(INFINITY);
	for (int b = 9; b <= 27; ++b) {
		// smallest ordered key k in [lo_all, hi_all] with ref_bin(key) >= b; ref_bin(+inf) == 27 >= b
		int64_t lo = lo_all, hi = hi_all;
		while (lo < hi) {
			const int64_t mid = lo + (hi - lo) / 2;
			if (ref_bin(ordered_to_float((int32_t)mid)) >= b) hi = mid; else lo = mid + 1;
		}
		out[b - 9] = ordered_to_float((int32_t)lo);
	}
}

// the BRIEF pattern on the device: 1024 int8 coordinates, then the same as floats (the describe kernel rotates them; converting there cost 16
// instructions per keypoint wavefront)
hipError_t ssk_upload_pattern(ssk_ctx* ctx)
{
	float f[1024];
	for (int i = 0; i < 1024; ++i) f[i] = (float)ctx->pattern[i];
	hipError_t e = hipMemcpy(ctx->d_pattern, ctx->pattern, 1024, hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemcpy(ctx->d_pattern + 1024, f, sizeof f, hipMemcpyHostToDevice);
	return e;
}

extern "C" {

int ssk_api_version(void) { return SSK_API_VERSION; }

const char* ssk_last_error(const ssk_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int ssk_create(ssk_ctx** out, int device, void* hip_stream)
{
	if (!out) return ssk_set_error(nullptr, SSK_ERR_BAD_ARG, "ssk_create: out is NULL");
	*out = nullptr;
	int ndev = 0;
	hipError_t e = hipGetDeviceCount(&ndev);
	if (e != hipSuccess || ndev <= 0)
		return ssk_set_error(nullptr, SSK_ERR_NO_DEVICE,
		                     "ssk_create: no HIP device (%s); libssk_hip has no CPU fallback",
		                     e != hipSuccess ? hipGetErrorString(e) : "device count 0");
	if (device < 0 || device >= ndev)
		return ssk_set_error(nullptr, SSK_ERR_BAD_ARG, "ssk_create: device %d out of range [0,%d)", device, ndev);
	hipDeviceProp_t prop;
	if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess)
		return ssk_set_error(nullptr, SSK_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
	if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
		return ssk_set_error(nullptr, SSK_ERR_NO_DEVICE,
		                     "ssk_create: device %d is %s; this library carries gfx950 (MI355X) code only",
		                     device, prop.gcnArchName);
	ssk_ctx* ctx = new ssk_ctx();
	ctx->device = device;
	ctx->n_cu = prop.multiProcessorCount;
	ctx->pin.pinned_host = true;
	ctx->pin_lm.pinned_host = true;
#define CREATE_HIP(call)                                                                           \
	do { hipError_t e2 = (call); if (e2 != hipSuccess) {                                           \
		int rc = ssk_set_error(nullptr, SSK_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e2)); \
		delete ctx; return rc; } } while (0)
	CREATE_HIP(hipSetDevice(device));
	if (hip_stream) { ctx->stream = (hipStream_t)hip_stream; ctx->own_stream = false; }
	else { CREATE_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)); ctx->own_stream = true; }
	CREATE_HIP(hipEventCreate(&ctx->ev_start));
	CREATE_HIP(hipEventCreate(&ctx->ev_stop));
	ssk_env_read(ctx->env);
	compute_dir_thresholds(ctx->dir_thresh);
	CREATE_HIP(hipMalloc((void**)&ctx->d_dir_thresh, sizeof ctx->dir_thresh));
	CREATE_HIP(hipMemcpy(ctx->d_dir_thresh, ctx->dir_thresh, sizeof ctx->dir_thresh, hipMemcpyHostToDevice));
	std::memcpy(ctx->pattern, ssk_default_brief_pattern, sizeof ctx->pattern);
	if (const char* sm = std::getenv("SSK_ORB_STEERING")) ctx->orb_steering = std::atoi(sm) ? 1 : 0;      // (0: the centroid-direction steering of rounds 1-2)
	if (const char* pf = std::getenv("SSK_BRIEF_PATTERN_FILE")) {
		// an external 256 x (x0, y0, x1, y1) int8 table, e.g. OpenCV's bit_pattern_31_ (DBoW2 vocabularies are trained on it)
		std::FILE* f = std::fopen(pf, "rb");
		int8_t buf[1024];
		const size_t got = f ? std::fread(buf, 1, sizeof buf, f) : 0;
		if (f) std::fclose(f);
		bool ok = got == sizeof buf;
		for (size_t i = 0; ok && i < sizeof buf; ++i) ok = buf[i] >= -13 && buf[i] <= 13;
		if (!ok) {
			const int rc = ssk_set_error(nullptr, SSK_ERR_BAD_ARG, "ssk_create: SSK_BRIEF_PATTERN_FILE=%s is not a readable 1024-byte table of coordinates within +-13", pf);
			ssk_destroy(ctx);
			return rc;
		}
		std::memcpy(ctx->pattern, buf, sizeof buf);
	}
	CREATE_HIP(hipMalloc((void**)&ctx->d_pattern, sizeof ctx->pattern * 5));          // the int8 table and its float copy (ssk_upload_pattern)
	CREATE_HIP(ssk_upload_pattern(ctx));
#undef CREATE_HIP
	*out = ctx;
	return SSK_OK;
}

void ssk_destroy(ssk_ctx* ctx)
{
	if (!ctx) return;
	(void)hipSetDevice(ctx->device);
	(void)hipStreamSynchronize(ctx->stream);
	if (ctx->dev.base) (void)hipFree(ctx->dev.base);
	if (ctx->pin.base) (void)hipHostFree(ctx->pin.base);
	if (ctx->lm_up_pending && ctx->ev_lm_up) (void)hipEventSynchronize(ctx->ev_lm_up);      // (a lean create's uploads still read pin_lm)
	if (ctx->pin_lm.base) (void)hipHostFree(ctx->pin_lm.base);
	if (ctx->single) ssk_kfbatch_destroy_internal(ctx->single);
	for (void* sl : ctx->lm_slab) if (sl) (void)hipFree(sl);
	if (ctx->d_dir_thresh) (void)hipFree(ctx->d_dir_thresh);
	if (ctx->d_pattern) (void)hipFree(ctx->d_pattern);
	for (hipEvent_t e : ctx->prof_pool) (void)hipEventDestroy(e);
	if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
	if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
	if (ctx->ev_lm_up) (void)hipEventDestroy(ctx->ev_lm_up);
	if (ctx->side_stream) { (void)hipStreamSynchronize(ctx->side_stream); (void)hipStreamDestroy(ctx->side_stream); }
	if (ctx->ev_side_a) (void)hipEventDestroy(ctx->ev_side_a);
	if (ctx->ev_side_b) (void)hipEventDestroy(ctx->ev_side_b);
	if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
	delete ctx;
}

int ssk_synchronize(ssk_ctx* ctx)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return SSK_OK;
}

void* ssk_stream(ssk_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int ssk_timer_start(ssk_ctx* ctx)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	SSK_HIP(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
	return SSK_OK;
}

int ssk_timer_stop_ms(ssk_ctx* ctx, float* ms)
{
	if (!ctx || !ms) return SSK_ERR_BAD_ARG;
	SSK_HIP(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
	SSK_HIP(ctx, hipEventSynchronize(ctx->ev_stop));
	SSK_HIP(ctx, hipEventElapsedTime(ms, ctx->ev_start, ctx->ev_stop));
	return SSK_OK;
}

int ssk_profile_enable(ssk_ctx* ctx, int on)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
	ctx->prof_on = on != 0;
	ctx->prof_open.clear();
	ctx->prof_used = 0;
	for (int i = 0; i < SSK_STAGE_COUNT; ++i) { ctx->prof_ms[i] = 0; ctx->prof_calls[i] = 0; }
	return SSK_OK;
}

int ssk_profile_read(ssk_ctx* ctx, double* ms, int64_t* calls)
{
	if (!ctx || !ms || !calls) return SSK_ERR_BAD_ARG;
	SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
	for (auto& r : ctx->prof_open) {
		float t = 0.f;
		SSK_HIP(ctx, hipEventElapsedTime(&t, r.a, r.b));
		ctx->prof_ms[r.stage] += t;
		ctx->prof_calls[r.stage] += 1;
	}
	ctx->prof_open.clear();
	ctx->prof_used = 0;
	for (int i = 0; i < SSK_STAGE_COUNT; ++i) { ms[i] = ctx->prof_ms[i]; calls[i] = ctx->prof_calls[i]; }
	return SSK_OK;
}

const char* ssk_stage_name(int stage)
{
	static const char* names[SSK_STAGE_COUNT] = {"fast_nms", "select", "describe", "stereo_match", "hamming_match",
	                                             "da_filters", "lm_linearize", "lm_schur", "lm_solve", "lm_update", "change_in_pose", "fund_matrix", "bow", "pose_graph"};
	return stage >= 0 && stage < SSK_STAGE_COUNT ? names[stage] : "?";
}

// test hook (declared in include/ssk_capi.h): the 19 direction-bin thresholds the context computed.
int ssk_debug_direction_thresholds(ssk_ctx* ctx, float* out19)
{
	if (!ctx || !out19) return SSK_ERR_BAD_ARG;
	std::memcpy(out19, ctx->dir_thresh, sizeof ctx->dir_thresh);
	return SSK_OK;
}

}  // extern "C"
