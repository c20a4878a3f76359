// vgsdf_device.cpp — C-ABI layer of libvgsdf.so (include/vgsdf.h): contexts, HBM-resident
// batches, transfers and launches.  No CPU fallback lives here or anywhere in the
// product: if HIP is unusable every entry point reports VGSDF_E_HIP.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/vgsdf.h"
#include "outline_kernels.h"
#include "sdf_kernels.h"

namespace {

thread_local std::string g_create_error;

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// tiles x chunks a workgroup of the span kernel sweeps at most.  16 is the best value for a batch that fills the chip
// several times over (Noto Sans Regular: 3023 workgroups on 1024 slots; 8 costs it 7 %: chunks are staged more often);
// a small batch is bounded by its longest workgroups instead, and halving them helps (Fira Sans, 1679 glyphs: 58.4 -> 52.2 us).
inline uint32_t default_span_budget(uint32_t n_glyphs) { return n_glyphs < 2048u ? 8u : 16u; }

// true when p is page-locked host memory known to HIP (hipHostMalloc / vgsdf_host_alloc):
// such arrays are DMA'd straight from/to the caller without a staging copy
inline bool is_pinned(const void *p, size_t bytes)
{
	if (!p)
		return false;
	for (const void *q : {p, (const void *)((const uint8_t *)p + (bytes ? bytes - 1 : 0))}) { // first and last byte
		hipPointerAttribute_t a;
		if (hipPointerGetAttributes(&a, q) != hipSuccess) {
			(void)hipGetLastError(); // plain malloc memory: clear the sticky error
			return false;
		}
		if (a.type != hipMemoryTypeHost)
			return false;
	}
	return true;
}

// The address a KERNEL may use for page-locked host memory: the mapping the runtime reports for it (equal to the host
// address for hipHostMalloc memory on this platform, but not guaranteed for memory the caller registered itself with
// hipHostRegister).  NULL when p is not page-locked host memory known to HIP, or has no device mapping.
inline void *pinned_device_ptr(void *p, size_t bytes)
{
	if (!is_pinned(p, bytes))
		return nullptr;
	hipPointerAttribute_t a;
	if (hipPointerGetAttributes(&a, p) != hipSuccess) {
		(void)hipGetLastError();
		return nullptr;
	}
	return a.devicePointer;
}

} // namespace

// HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues per device (default 4), and streams that share a
// queue run one after the other.  A renderer keeps two groups in flight on two contexts of two streams each: four streams — as
// soon as the host holds any other stream (PyTorch, its own contexts) two of ours share a queue and the overlap of one group's
// front-end with the other's raster is gone (measured in bench.py's process: Noto Sans all files 8.9 instead of 11.2 M glyphs/s).
// The variable is read when the HIP runtime starts, i.e. at the process's first HIP call: when this library is loaded before
// that and the variable is unset, 8 queues are asked for.  A host that starts HIP first sets it itself.
// (VGSDF_KEEP_HW_QUEUES=1 in the environment: the library leaves the variable alone.)
__attribute__((constructor)) static void vgsdf_default_hw_queues()
{
	const char *keep = std::getenv("VGSDF_KEEP_HW_QUEUES");
	if (!(keep && keep[0] == '1'))
		(void)setenv("GPU_MAX_HW_QUEUES", "8", /*overwrite=*/0);
}

struct vgsdf_ctx {
	int device = 0;
	hipStream_t stream = nullptr;
	hipStream_t copy_stream = nullptr;                 // the front-end's read-back, beside the kernels that follow the plan
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	hipEvent_t ev_plan = nullptr, ev_rects = nullptr;   // plan done (kernel stream) / rects on the host (copy stream)
	int variant = 0;
	std::string err;
	// grow-only scratch of vgsdf_render_batch: no hipMalloc / hipHostMalloc in steady state
	void *d_scratch = nullptr, *h_scratch = nullptr;
	size_t d_scratch_bytes = 0, h_scratch_bytes = 0;
	struct FrontEnd *fe = nullptr; // device outline front-end state (lazy)
	// run counters {blocks, glyphs, pixels} of the work this context did (vgsdf_add_counters), summed over the
	// contexts of a run by vgsdf_reduce_counters; d_counters: 24 bytes on the device for the collective
	uint64_t counters[3] = {0, 0, 0};
	uint64_t *d_counters = nullptr;
	void *comm = nullptr; // ncclComm_t of the communicator this context last reduced in (owned by the cache below)
	std::string reduce_path; // how the last vgsdf_reduce_counters with this context first took its sum (vgsdf_reduce_path)
};

struct vgsdf_dbatch {
	vgsdf_stats stats{};
	// one device arena: [descs | tiles | sx | sy | ex | ey | out]
	void *d_arena = nullptr;
	size_t arena_bytes = 0, input_bytes = 0;
	void *h_stage = nullptr; // pinned staging of the input part
	vgsdf::GlyphDesc *d_glyphs = nullptr;
	uint2 *d_tiles = nullptr;
	void *d_boxes = nullptr; // chunk boxes (span kernel); NULL: none
	double *d_sx = nullptr, *d_sy = nullptr, *d_ex = nullptr, *d_ey = nullptr;
	uint32_t seg_stride = 1; // 1: four SoA arrays (C ABI batches); 4: 32-byte records (device front-end)
	uint8_t *d_out = nullptr;
	size_t out_bytes = 0;
	// work list = [main kernel | brute force]
	uint32_t n_main = 0; // entries [0, n_main): main kernel; the rest: brute-force tiles
	int tile_order = 1;
	bool span_list = false; // main-class entries are (glyph, first pixel | tile count): sdf_tiles_span only
	bool borrowed = false; // arena + staging belong to the context (vgsdf_render_batch)
};

// grow-only device / pinned-host buffers of the outline front-end
struct DevBuf {
	void *p = nullptr;
	size_t cap = 0;
	bool host = false;
	hipError_t ensure(size_t bytes)
	{
		if (bytes <= cap)
			return hipSuccess;
		release();
		const size_t want = bytes + bytes / 4 + 256;
		hipError_t e = host ? hipHostMalloc(&p, want, hipHostMallocDefault) : hipMalloc(&p, want);
		if (e != hipSuccess) {
			p = nullptr;
			return e;
		}
		cap = want;
		return hipSuccess;
	}
	void release()
	{
		if (p)
			(void)(host ? hipHostFree(p) : hipFree(p));
		p = nullptr;
		cap = 0;
	}
};

// a submission between vgsdf_outlines_submit and vgsdf_outlines_wait
struct FePending {
	bool active = false;
	uint32_t n = 0, n_cmds = 0;
	size_t hdr_off = 0, rh_bytes = 0, at_off = 0; // rects | PlanHeader | (in-place PBF assembly) bitmap positions u64[n]
	bool span = false, spec = false;
	bool spec_direct = false; // the raster stores through the device mapping of the caller's page-locked buffer
	const uint32_t *d_pbf_pre = nullptr; // device copies of the in-place PBF inputs (NULL: bitmaps packed back to back)
	const uint8_t *d_pbf_fix = nullptr;
	uint8_t *spec_out = nullptr, *d_spec = nullptr; // destination of the raster enqueued behind the front-end
	size_t spec_cap = 0;
	uint32_t launch_spans = 0, span_max = 4, span_budget = 16;
	double t0 = 0, t1 = 0;
};

struct FrontEnd {
	// device: inputs, per-command / per-ring intermediates, results of measure + plan, the resident batch
	DevBuf cmds, kinds, coords, meta, cmd_open, counts, pt_local, cmd_box, cmd_mask, rings, cmd_ring, rects_hdr, descs, tiles, flag;
	DevBuf seg, out, boxes, pbf_in; // seg: records {sx, sy, ex, ey}
	DevBuf h_rects, h_stage; // pinned
	size_t seg_cap = 0, tile_cap = 0; // elements the segment arrays / the work list hold
	uint32_t last_spans = 0;          // work-list length of the previous batch (grid guess of the one-submission form)
	// error words of the submissions: two 16-byte slots used alternately; the plan kernel of a submission zeroes the other
	// slot for its successor (no memset launch per submission).  flags_clean: both slots are known to be in that state
	uint32_t flag_slot = 0;
	bool flags_clean = false;
	uint32_t *flag_word() const { return (uint32_t *)((uint8_t *)flag.p + 16 * (size_t)flag_slot); }
	uint32_t *next_flag_word() const { return (uint32_t *)((uint8_t *)flag.p + 16 * (size_t)(flag_slot ^ 1u)); }
	FePending pend;
	uint32_t n_glyphs = 0, n_cmds = 0, n_segs = 0;
	uint64_t out_bytes = 0;
	vgsdf_dbatch batch; // borrowed view over the buffers above
	bool prepared = false;
	bool peeked = false; // vgsdf_outlines_peek has waited for the read-back of the pending submission
	FrontEnd()
	{
		h_rects.host = true;
		h_stage.host = true;
		batch.borrowed = true;
	}
	void release_all()
	{
		for (DevBuf *b : {&cmds, &kinds, &coords, &meta, &cmd_open, &counts, &pt_local, &cmd_box, &cmd_mask, &rings, &cmd_ring, &rects_hdr, &descs, &tiles, &flag, &seg, &out, &boxes, &pbf_in,
		                  &h_rects, &h_stage})
			b->release();
	}
};


#define HIP_TRY(ctx, expr)                                                                     \
	do {                                                                                       \
		hipError_t e__ = (expr);                                                               \
		if (e__ != hipSuccess) {                                                               \
			(ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                   \
			return e__ == hipErrorOutOfMemory ? VGSDF_E_OOM : VGSDF_E_HIP;                     \
		}                                                                                      \
	} while (0)

extern "C" {

int vgsdf_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		return 0;
	return n;
}

int vgsdf_create(int device_ordinal, vgsdf_ctx **out)
{
	if (!out) {
		g_create_error = "vgsdf_create: out is NULL";
		return VGSDF_E_ARG;
	}
	*out = nullptr;
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0) {
		g_create_error = std::string("vgsdf_create: no HIP device (") +
		                 (e != hipSuccess ? hipGetErrorString(e) : "device count 0") +
		                 "); this library has no CPU fallback";
		return VGSDF_E_HIP;
	}
	if (device_ordinal < 0 || device_ordinal >= n) {
		g_create_error = "vgsdf_create: device ordinal out of range";
		return VGSDF_E_ARG;
	}
	vgsdf_ctx *ctx = new (std::nothrow) vgsdf_ctx();
	if (!ctx) {
		g_create_error = "vgsdf_create: out of host memory";
		return VGSDF_E_OOM;
	}
	ctx->device = device_ordinal;
	if ((e = hipSetDevice(device_ordinal)) != hipSuccess ||
	    (e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess ||
	    (e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking)) != hipSuccess ||
	    (e = hipEventCreateWithFlags(&ctx->ev_plan, hipEventDisableTiming)) != hipSuccess ||
	    (e = hipEventCreateWithFlags(&ctx->ev_rects, hipEventDisableTiming)) != hipSuccess ||
	    (e = hipEventCreate(&ctx->ev0)) != hipSuccess || (e = hipEventCreate(&ctx->ev1)) != hipSuccess) {
		g_create_error = std::string("vgsdf_create: ") + hipGetErrorString(e);
		vgsdf_destroy(ctx);
		return VGSDF_E_HIP;
	}
	*out = ctx;
	return VGSDF_OK;
}

void vgsdf_destroy(vgsdf_ctx *ctx)
{
	if (!ctx)
		return;
	(void)hipSetDevice(ctx->device);
	if (ctx->stream) {
		(void)hipStreamSynchronize(ctx->stream);
		(void)hipStreamDestroy(ctx->stream);
	}
	if (ctx->copy_stream) {
		(void)hipStreamSynchronize(ctx->copy_stream);
		(void)hipStreamDestroy(ctx->copy_stream);
	}
	if (ctx->ev_plan)
		(void)hipEventDestroy(ctx->ev_plan);
	if (ctx->ev_rects)
		(void)hipEventDestroy(ctx->ev_rects);
	if (ctx->ev0)
		(void)hipEventDestroy(ctx->ev0);
	if (ctx->ev1)
		(void)hipEventDestroy(ctx->ev1);
	if (ctx->fe) {
		ctx->fe->release_all();
		delete ctx->fe;
	}
	if (ctx->d_counters)
		(void)hipFree(ctx->d_counters);
	if (ctx->d_scratch)
		(void)hipFree(ctx->d_scratch);
	if (ctx->h_scratch)
		(void)hipHostFree(ctx->h_scratch);
	delete ctx;
}

const char *vgsdf_last_error(const vgsdf_ctx *ctx)
{
	return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int vgsdf_set_variant(vgsdf_ctx *ctx, int variant)
{
	if (!ctx)
		return VGSDF_E_ARG;
	// the product build knows 0 (default) and 1 (brute force); development builds (-DVGSDF_DEV_VARIANTS)
	// add the earlier generations and the timing-only ablations
	if (!vgsdf_kernel_known(variant == 0 ? 50 : (variant == 13 ? 10 : variant)) || variant == 50) {
		ctx->err = "vgsdf_set_variant: unknown kernel variant " + std::to_string(variant);
		return VGSDF_E_ARG;
	}
	ctx->variant = variant;
	return VGSDF_OK;
}

int vgsdf_sync(vgsdf_ctx *ctx)
{
	if (!ctx)
		return VGSDF_E_ARG;
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return VGSDF_OK;
}

int vgsdf_batch_free(vgsdf_ctx *ctx, vgsdf_dbatch *b)
{
	if (!ctx)
		return VGSDF_E_ARG;
	if (!b)
		return VGSDF_OK;
	(void)hipSetDevice(ctx->device);
	(void)hipStreamSynchronize(ctx->stream);
	if (!b->borrowed) {
		if (b->d_arena)
			(void)hipFree(b->d_arena);
		if (b->h_stage)
			(void)hipHostFree(b->h_stage);
	}
	delete b;
	return VGSDF_OK;
}

static double fe_now()
{
	return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Fills the glyph descriptors and the tile list (routing + order) of a batch.  `ht` must hold
// stats.n_tiles entries.  Shared by the segment entry points and the outline front-end.
static void build_descs_and_tiles(const vgsdf_batch *in, vgsdf::GlyphDesc *hd, uint2 *ht, vgsdf_dbatch *b, bool span)
{
	const uint32_t n = in->n_glyphs;
	// Routing.  A glyph goes to the brute-force kernel when its winding histogram (rows touched
	// by 256 consecutive pixels, times w+1 columns) would not fit in LDS, or its segment index
	// needs more than 24 bits; everything else (class 0; every real font at 24 px/EM) takes the
	// main kernel.
	const uint64_t delta_cap = (uint64_t)vgsdf_filtered_delta_cap();
	// rows touched by T consecutive tiles, times w+1 columns, must fit the winding histogram
	// (32-bit division: this runs per glyph on the end-to-end path)
	// (the kernel pads a histogram row of w + 1 cells to an odd stride: at most w + 2)
	auto fits = [&](uint32_t w, uint32_t T) { return (uint64_t)((VGSDF_TILE_PIXELS * T - 2u) / w + 2u) * ((uint64_t)w + 2u) <= delta_cap; };
	// span list (default kernel): a workgroup takes up to 4 consecutive tiles of one glyph, the
	// largest count whose rows fit; the entry is (glyph, first pixel | count)
	const char *sm = std::getenv("VGSDF_SPAN_MAX");
	const uint32_t span_max = sm ? (uint32_t)std::min(4, std::max(1, std::atoi(sm))) : 4u;
	const char *sb = std::getenv("VGSDF_SPAN_BUDGET");
	const uint32_t span_budget = sb ? (uint32_t)std::max(1, std::atoi(sb)) : default_span_budget(n); // measured: 12-24 equally good
	b->span_list = span;
	const char *ord = std::getenv("VGSDF_TILE_ORDER");
	b->tile_order = ord ? std::atoi(ord) : 1;

	// One pass per glyph: class, span length T and the weight of its workgroups.
	//   class 0: main kernel; 2: brute force (winding histogram would not fit in LDS, or >= 2^24 segments).
	//   T: a workgroup sweeps T tiles per staged chunk, one after the other: the largest T <= span_max whose
	//   rows fit the histogram, with tiles x chunks bounded so that the glyphs with long segment lists stay
	//   spread over many workgroups (they set the makespan of a small batch) while short ones are staged once.
	//   weight: segments x tiles swept per staged chunk; heaviest first inside a class (the dispatcher hands
	//   workgroups out in list order, so the long ones start early and the tail is made of short ones).
	//   VGSDF_TILE_ORDER=0 keeps glyph order (+ per-XCD contiguous remap in-kernel).
	// (thread_local scratch, addressed through plain references below: in a PIC shared object every
	// use of a thread_local name is a __tls_get_addr call)
	static thread_local std::vector<uint8_t> tl_span_t;
	static thread_local std::vector<uint64_t> tl_keys[3]; // (~weight << 32) | glyph: ascending sort = heaviest first, stable
	static thread_local std::vector<uint2> tl_queue[8];
	std::vector<uint8_t> &span_t = tl_span_t;
	std::vector<uint64_t> *const keys = tl_keys;
	std::vector<uint2> *const queue = tl_queue;
	span_t.resize(n);
	for (int c = 0; c < 3; c++)
		keys[c].clear();
	for (uint32_t g = 0; g < n; g++) {
		const uint32_t w = in->w[g], h = in->h[g];
		const uint64_t px = (uint64_t)w * h;
		if (px == 0)
			continue;
		const uint32_t nseg = in->seg_off[g + 1] - in->seg_off[g];
		int cls = 0;
		uint32_t T = 1;
		if (!fits(w, 1) || nseg >= (1u << 24)) {
			cls = 2;
		} else if (span) {
			const uint32_t chunks = (nseg + 255u) / 256u;
			const uint32_t t_hi = std::min(span_max, std::max(1u, span_budget / std::max(chunks, 1u)));
			for (T = t_hi; T > 1; T--)
				if (fits(w, T))
					break;
		}
		span_t[g] = (uint8_t)T;
		const uint64_t tiles_g = (px + VGSDF_TILE_PIXELS - 1) >> 8;
		static_assert(VGSDF_TILE_PIXELS == 256, "shifts below");
		const uint64_t weight = std::min<uint64_t>((uint64_t)nseg * std::min<uint64_t>(T, tiles_g), 0xFFFFFFFFull);
		keys[cls].push_back(((0xFFFFFFFFull - weight) << 32) | g);
	}
	static const bool trace_l = std::getenv("VGSDF_TRACE") != nullptr;
	const double tl0 = trace_l ? fe_now() : 0;
	double tl_sort = 0, tl_deal = 0;
	uint64_t ti = 0;
	for (int cls = 0; cls < 3; cls++) {
		std::vector<uint64_t> &gl = keys[cls];
		const uint64_t first = ti;
		const double ts0 = trace_l ? fe_now() : 0;
		if (b->tile_order != 0 && cls < 2 && gl.size() > 1) {
			// heaviest first, to 1/16 of the weight (exact order does not matter): one counting pass over
			// 512 logarithmic buckets instead of a comparison sort (75 us for a 3000-glyph font)
			static thread_local std::vector<uint64_t> tl_tmp;
			std::vector<uint64_t> &tmp = tl_tmp;
			tmp.resize(gl.size());
			uint32_t hist[513] = {0};
			auto bucket = [](uint64_t key) -> uint32_t { // small key = heavy; bucket 0 = heaviest
				const uint32_t wgt = 0xFFFFFFFFu - (uint32_t)(key >> 32);
				if (wgt < 16u)
					return 511u - wgt;
				const uint32_t e = 31u - (uint32_t)__builtin_clz(wgt);           // 4..31
				return 511u - ((e - 3u) * 16u + ((wgt >> (e - 4u)) & 15u));      // 16..463 -> descending
			};
			for (uint64_t k : gl)
				hist[bucket(k) + 1]++;
			for (int i = 0; i < 512; i++)
				hist[i + 1] += hist[i];
			for (uint64_t k : gl) // stable: glyph order inside a bucket
				tmp[hist[bucket(k)]++] = k;
			gl.swap(tmp);
		}
		const double ts1 = trace_l ? fe_now() : 0;
		tl_sort += ts1 - ts0;
		// entries of glyph g: one per span of T tiles (T = 1 unless this is the span list's main class)
		auto emit = [&](uint32_t g, auto &&push) {
			const uint64_t px = (uint64_t)in->w[g] * in->h[g]; // <= 2^32 - 1 - 256 (validated by the callers)
			const uint32_t T = span_t[g];
			for (uint64_t p = 0; p < px; p += (uint64_t)VGSDF_TILE_PIXELS * T) { // 64-bit: p + 1024 may pass 2^32
				const uint32_t left = (uint32_t)((px - p + VGSDF_TILE_PIXELS - 1) >> 8);
				push(make_uint2(g, span && cls == 0 ? ((uint32_t)p | std::min(T, left)) : (uint32_t)p));
			}
		};
		uint64_t n_cls_tiles = 0;
		for (uint64_t k : gl) {
			const uint32_t g = (uint32_t)k;
			const uint32_t t256 = (uint32_t)(((uint64_t)in->w[g] * in->h[g] + 255u) >> 8), T = span_t[g];
			n_cls_tiles += T == 1 ? t256 : (T == 2 ? (t256 + 1u) >> 1 : (T == 4 ? (t256 + 3u) >> 2 : (t256 + 2u) / 3u));
		}
		if (b->tile_order != 0 && cls < 2 && n_cls_tiles >= 64) {
			// Workgroups are dealt round-robin over the 8 XCDs (position p runs on XCD p % 8, each
			// with its own L2).  Keep all tiles of a glyph on ONE XCD so its segment list is fetched
			// into one L2 only: glyphs are dealt to the currently shortest of 8 per-XCD queues, and
			// the queues are interleaved position by position.
			size_t qlen[8] = {0, 0, 0, 0, 0, 0, 0, 0};
			uint2 *qbuf[8];
			for (int q = 0; q < 8; q++) {
				if (queue[q].size() < n_cls_tiles)
					queue[q].resize(n_cls_tiles); // plain arrays below: no capacity checks per entry
				qbuf[q] = queue[q].data();
			}
			for (uint64_t k : gl) {
				size_t best = 0;
				for (size_t m = 1; m < 8; m++)
					if (qlen[m] < qlen[best])
						best = m;
				uint2 *dst = qbuf[best];
				size_t len = qlen[best];
				emit((uint32_t)k, [&](uint2 e) { dst[len++] = e; });
				qlen[best] = len;
			}
			size_t taken[8] = {0, 0, 0, 0, 0, 0, 0, 0};
			const uint64_t last = first + n_cls_tiles;
			while (ti < last)
				for (size_t k = 0; k < 8 && ti < last; k++) {
					size_t src = k; // position ti runs on XCD ti % 8 == k as long as no queue ran dry
					if (taken[src] >= qlen[src])
						for (size_t m = 0; m < 8; m++) // dry: borrow from the fullest queue
							if (qlen[m] - taken[m] > qlen[src] - taken[src])
								src = m;
					ht[ti++] = qbuf[src][taken[src]++];
				}
		} else {
			for (uint64_t k : gl)
				emit((uint32_t)k, [&](uint2 e) { ht[ti++] = e; });
		}
		if (cls == 1)
			b->n_main = (uint32_t)ti;
		tl_deal += (trace_l ? fe_now() : 0) - ts1;
	}
	const double tl1 = trace_l ? fe_now() : 0;
	b->stats.n_tiles = ti; // workgroups actually launched (<= the 256-pixel tile count the list was sized for)
	for (uint32_t g = 0; g < n; g++) {
		hd[g].seg_off = in->seg_off[g];
		hd[g].n_seg = in->seg_off[g + 1] - in->seg_off[g];
		hd[g].x0 = in->x0[g];
		hd[g].y0 = in->y0[g];
		hd[g].w = in->w[g];
		hd[g].h = in->h[g];
		hd[g].out_off = in->out_off[g];
	}
	if (trace_l)
		std::fprintf(stderr, "[vgsdf] list: sort %.3f ms, deal+emit %.3f ms, descs %.3f ms (total after pass 1: %.3f)\n", tl_sort * 1e3,
		             tl_deal * 1e3, (fe_now() - tl1) * 1e3, (fe_now() - tl0) * 1e3);
}

static int upload_impl(vgsdf_ctx *ctx, const vgsdf_batch *in, vgsdf_dbatch **out, bool use_ctx_scratch)
{
	if (!ctx)
		return VGSDF_E_ARG;
	if (!in || !out) {
		ctx->err = "vgsdf_batch_upload: NULL argument";
		return VGSDF_E_ARG;
	}
	*out = nullptr;
	const uint32_t n = in->n_glyphs;
	if (n && (!in->seg_off || !in->x0 || !in->y0 || !in->w || !in->h || !in->out_off)) {
		ctx->err = "vgsdf_batch_upload: NULL array in batch";
		return VGSDF_E_ARG;
	}
	const uint64_t n_seg = n ? in->seg_off[n] : 0;
	if (n_seg && (!in->seg_sx || !in->seg_sy || !in->seg_ex || !in->seg_ey)) {
		ctx->err = "vgsdf_batch_upload: NULL segment array";
		return VGSDF_E_ARG;
	}
	// validate shapes on the host BEFORE anything is launched: the kernel indexes with them
	uint64_t n_tiles = 0, n_pairs = 0, n_pixels = 0;
	if (n && in->seg_off[0] != 0) {
		ctx->err = "vgsdf_batch_upload: seg_off[0] must be 0";
		return VGSDF_E_ARG;
	}
	for (uint32_t g = 0; g < n; g++) {
		if (in->seg_off[g + 1] < in->seg_off[g]) {
			ctx->err = "vgsdf_batch_upload: seg_off not monotone";
			return VGSDF_E_ARG;
		}
		const uint64_t px = (uint64_t)in->w[g] * in->h[g];
		if (px > 0xFFFFFFFFull - VGSDF_TILE_PIXELS || in->out_off[g + 1] < in->out_off[g] || in->out_off[g + 1] - in->out_off[g] < px) {
			ctx->err = "vgsdf_batch_upload: out_off inconsistent with w*h (bitmap g needs out_off[g] + w*h <= out_off[g+1])";
			return VGSDF_E_ARG;
		}
		n_tiles += (px + VGSDF_TILE_PIXELS - 1) / VGSDF_TILE_PIXELS;
		n_pairs += px * (in->seg_off[g + 1] - in->seg_off[g]);
		n_pixels += px;
	}
	if (n_tiles > 0x7FFFFFFFull) {
		ctx->err = "vgsdf_batch_upload: batch too large (tile count exceeds 2^31-1); split it";
		return VGSDF_E_ARG;
	}
	const uint64_t n_pix = n ? in->out_off[n] : 0; // size of the output buffer (>= the pixels: gaps are allowed)

	vgsdf_dbatch *b = new (std::nothrow) vgsdf_dbatch();
	if (!b) {
		ctx->err = "vgsdf_batch_upload: out of host memory";
		return VGSDF_E_OOM;
	}
	b->stats.n_glyphs = n;
	b->stats.n_segments = n_seg;
	b->stats.n_pixels = n_pixels;
	b->stats.n_pairs = n_pairs;
	b->stats.n_tiles = n_tiles;
	b->stats.alg_bytes = 32 * n_seg + 32 * (uint64_t)n + n_pixels;
	b->out_bytes = n_pix;

	const size_t A = 256;
	size_t off = 0;
	const size_t off_desc = off;
	off = align_up(off + sizeof(vgsdf::GlyphDesc) * (size_t)n, A);
	const size_t off_tiles = off;
	off = align_up(off + sizeof(uint2) * (size_t)n_tiles, A);
	const size_t seg_bytes = align_up(sizeof(double) * (size_t)n_seg, A);
	const size_t off_sx = off, off_sy = off + seg_bytes, off_ex = off + 2 * seg_bytes,
	             off_ey = off + 3 * seg_bytes;
	off += 4 * seg_bytes;
	b->input_bytes = off;
	const size_t off_out = off;
	off = align_up(off + (size_t)n_pix, A);
	const size_t off_boxes = off;
	off = align_up(off + vgsdf_chunk_box_bytes(n_seg, n), A);
	b->arena_bytes = off ? off : A;

	(void)hipSetDevice(ctx->device);
	hipError_t e = hipSuccess;
	// segment arrays already page-locked: DMA them directly, stage only descriptors + tiles
	const size_t seg_nb = sizeof(double) * (size_t)n_seg;
	const bool direct = n_seg && is_pinned(in->seg_sx, seg_nb) && is_pinned(in->seg_sy, seg_nb) &&
	                    is_pinned(in->seg_ex, seg_nb) && is_pinned(in->seg_ey, seg_nb);
	const size_t stage_bytes = direct ? off_sx : b->input_bytes;
	if (use_ctx_scratch) {
		b->borrowed = true;
		if (ctx->d_scratch_bytes < b->arena_bytes) {
			(void)hipStreamSynchronize(ctx->stream);
			if (ctx->d_scratch)
				(void)hipFree(ctx->d_scratch);
			ctx->d_scratch = nullptr;
			ctx->d_scratch_bytes = 0;
			const size_t want = b->arena_bytes + b->arena_bytes / 4;
			if ((e = hipMalloc(&ctx->d_scratch, want)) != hipSuccess) {
				ctx->err = std::string("vgsdf_render_batch: hipMalloc: ") + hipGetErrorString(e);
				delete b;
				return VGSDF_E_OOM;
			}
			ctx->d_scratch_bytes = want;
		}
		if (ctx->h_scratch_bytes < stage_bytes) {
			(void)hipStreamSynchronize(ctx->stream);
			if (ctx->h_scratch)
				(void)hipHostFree(ctx->h_scratch);
			ctx->h_scratch = nullptr;
			ctx->h_scratch_bytes = 0;
			const size_t want = stage_bytes + stage_bytes / 4 + 4096;
			if ((e = hipHostMalloc(&ctx->h_scratch, want, hipHostMallocDefault)) != hipSuccess) {
				ctx->err = std::string("vgsdf_render_batch: hipHostMalloc: ") + hipGetErrorString(e);
				delete b;
				return VGSDF_E_OOM;
			}
			ctx->h_scratch_bytes = want;
		}
		b->d_arena = ctx->d_scratch;
		b->h_stage = ctx->h_scratch;
	} else {
		e = hipMalloc(&b->d_arena, b->arena_bytes);
		if (e != hipSuccess) {
			ctx->err = std::string("vgsdf_batch_upload: hipMalloc: ") + hipGetErrorString(e);
			delete b;
			return VGSDF_E_OOM;
		}
		if (stage_bytes) {
			e = hipHostMalloc(&b->h_stage, stage_bytes, hipHostMallocDefault);
			if (e != hipSuccess) {
				ctx->err = std::string("vgsdf_batch_upload: hipHostMalloc: ") + hipGetErrorString(e);
				vgsdf_batch_free(ctx, b);
				return VGSDF_E_OOM;
			}
		}
	}
	uint8_t *hs = (uint8_t *)b->h_stage, *da = (uint8_t *)b->d_arena;
	b->d_glyphs = (vgsdf::GlyphDesc *)(da + off_desc);
	b->d_tiles = (uint2 *)(da + off_tiles);
	b->d_sx = (double *)(da + off_sx);
	b->d_sy = (double *)(da + off_sy);
	b->d_ex = (double *)(da + off_ex);
	b->d_ey = (double *)(da + off_ey);
	b->d_out = da + off_out;
	b->d_boxes = da + off_boxes;

	if (n) {
		vgsdf::GlyphDesc *hd = (vgsdf::GlyphDesc *)(hs + off_desc);
		uint2 *ht = (uint2 *)(hs + off_tiles);
		build_descs_and_tiles(in, hd, ht, b, ctx->variant == 0 || (ctx->variant >= 50 && ctx->variant <= 69));
		if (n_seg && !direct) {
			std::memcpy(hs + off_sx, in->seg_sx, sizeof(double) * n_seg);
			std::memcpy(hs + off_sy, in->seg_sy, sizeof(double) * n_seg);
			std::memcpy(hs + off_ex, in->seg_ex, sizeof(double) * n_seg);
			std::memcpy(hs + off_ey, in->seg_ey, sizeof(double) * n_seg);
		}
		e = hipMemcpyAsync(b->d_arena, b->h_stage, stage_bytes, hipMemcpyHostToDevice, ctx->stream);
		if (e == hipSuccess && direct) {
			const size_t nb = sizeof(double) * n_seg;
			e = hipMemcpyAsync(b->d_sx, in->seg_sx, nb, hipMemcpyHostToDevice, ctx->stream);
			if (e == hipSuccess)
				e = hipMemcpyAsync(b->d_sy, in->seg_sy, nb, hipMemcpyHostToDevice, ctx->stream);
			if (e == hipSuccess)
				e = hipMemcpyAsync(b->d_ex, in->seg_ex, nb, hipMemcpyHostToDevice, ctx->stream);
			if (e == hipSuccess)
				e = hipMemcpyAsync(b->d_ey, in->seg_ey, nb, hipMemcpyHostToDevice, ctx->stream);
		}
		if (e == hipSuccess && b->span_list)
			e = (hipError_t)vgsdf_launch_chunk_boxes(b->d_glyphs, n, b->d_sx, b->d_sy, b->d_ex, b->d_ey, 1, b->d_boxes, nullptr, 0, ctx->stream);
		// page-locked caller arrays are DMA'd in place: the copies must be over before the caller may
		// touch them again (vgsdf.h: the batch is read-only "for the call")
		if (e == hipSuccess && direct)
			e = hipStreamSynchronize(ctx->stream);
		if (e != hipSuccess) {
			ctx->err = std::string("vgsdf_batch_upload: H2D: ") + hipGetErrorString(e);
			vgsdf_batch_free(ctx, b);
			return VGSDF_E_HIP;
		}
	}
	*out = b;
	return VGSDF_OK;
}

int vgsdf_batch_upload(vgsdf_ctx *ctx, const vgsdf_batch *in, vgsdf_dbatch **out)
{
	return upload_impl(ctx, in, out, false);
}

void *vgsdf_host_alloc(size_t bytes)
{
	void *p = nullptr;
	if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) { // every device of the process may read / write it
		(void)hipGetLastError();
		return nullptr;
	}
	return p;
}

void vgsdf_host_free(void *p)
{
	if (p)
		(void)hipHostFree(p);
}

int vgsdf_batch_launch(vgsdf_ctx *ctx, vgsdf_dbatch *b)
{
	if (!ctx)
		return VGSDF_E_ARG;
	if (!b) {
		ctx->err = "vgsdf_batch_launch: NULL batch";
		return VGSDF_E_ARG;
	}
	(void)hipSetDevice(ctx->device);
	// variant 0 (default) = kernel 50: bounded groups over spans of tiles; misfits: brute force.
	// 1: everything brute.  Other ids exist only in development builds (vgsdf_set_variant rejects them
	// otherwise): earlier generations and timing-only ablations, see vgsdf_launch_tiles.
	const uint32_t n_all = (uint32_t)b->stats.n_tiles;
	const int v = ctx->variant;
	const uint32_t n_main = v == 1 ? 0 : b->n_main;
	// kernel id understood by vgsdf_launch_tiles
	const int k_main = v == 0 ? 50 : (v == 13 ? 10 : v);
	if (b->stats.n_tiles != 0 && (k_main >= 50 && k_main <= 69) != b->span_list) {
		ctx->err = "vgsdf_batch_launch: the batch was uploaded for a different kernel variant (tile list layout)";
		return VGSDF_E_ARG;
	}
	const int list_order = b->tile_order == 1;
	int e = vgsdf_launch_tiles(k_main, list_order, b->d_glyphs, b->d_tiles, n_main, b->d_sx, b->d_sy, b->d_ex, b->d_ey,
	                           b->seg_stride, b->d_out, b->span_list ? b->d_boxes : nullptr, ctx->stream);
	if (e == 0)
		e = vgsdf_launch_tiles(1, list_order, b->d_glyphs, b->d_tiles + n_main, n_all - n_main, b->d_sx, b->d_sy, b->d_ex,
		                       b->d_ey, b->seg_stride, b->d_out, nullptr, ctx->stream);
	if (e != 0) {
		ctx->err = std::string("vgsdf_batch_launch: ") + hipGetErrorString((hipError_t)e);
		return VGSDF_E_HIP;
	}
	return VGSDF_OK;
}

int vgsdf_batch_download(vgsdf_ctx *ctx, vgsdf_dbatch *b, uint8_t *out_bitmaps)
{
	if (!ctx)
		return VGSDF_E_ARG;
	if (!b || (!out_bitmaps && b->out_bytes)) {
		ctx->err = "vgsdf_batch_download: NULL argument";
		return VGSDF_E_ARG;
	}
	(void)hipSetDevice(ctx->device);
	if (b->out_bytes)
		HIP_TRY(ctx, hipMemcpyAsync(out_bitmaps, b->d_out, b->out_bytes, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return VGSDF_OK;
}

int vgsdf_batch_stats(const vgsdf_dbatch *b, vgsdf_stats *out)
{
	if (!b || !out)
		return VGSDF_E_ARG;
	*out = b->stats;
	return VGSDF_OK;
}

void *vgsdf_batch_device_output(const vgsdf_dbatch *b) { return b ? b->d_out : nullptr; }

int vgsdf_batch_time(vgsdf_ctx *ctx, vgsdf_dbatch *b, int iters, float *total_ms)
{
	if (!ctx)
		return VGSDF_E_ARG;
	if (!b || !total_ms || iters < 1) {
		ctx->err = "vgsdf_batch_time: bad argument";
		return VGSDF_E_ARG;
	}
	(void)hipSetDevice(ctx->device);
	HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
	for (int i = 0; i < iters; i++) {
		int rc = vgsdf_batch_launch(ctx, b);
		if (rc != VGSDF_OK)
			return rc;
	}
	HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
	HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
	HIP_TRY(ctx, hipEventElapsedTime(total_ms, ctx->ev0, ctx->ev1));
	return VGSDF_OK;
}

int vgsdf_render_batch(vgsdf_ctx *ctx, const vgsdf_batch *in, uint8_t *out_bitmaps)
{
	if (!ctx)
		return VGSDF_E_ARG;
	vgsdf_dbatch *b = nullptr;
	int rc = upload_impl(ctx, in, &b, true);
	if (rc != VGSDF_OK)
		return rc;
	rc = vgsdf_batch_launch(ctx, b);
	if (rc == VGSDF_OK)
		rc = vgsdf_batch_download(ctx, b, out_bitmaps);
	vgsdf_batch_free(ctx, b);
	return rc;
}

// ---------------------------------------------------------------------------------------
// outline front-end: commands in, rects out (prepare); bitmaps out (render)
// ---------------------------------------------------------------------------------------
#define FE_TRY(expr)                                                                            \
	do {                                                                                        \
		hipError_t e__ = (expr);                                                                \
		if (e__ != hipSuccess) {                                                                \
			ctx->err = std::string("vgsdf_outlines: " #expr ": ") + hipGetErrorString(e__);     \
			return e__ == hipErrorOutOfMemory ? VGSDF_E_OOM : VGSDF_E_HIP;                      \
		}                                                                                       \
	} while (0)
#define FE_KERNEL(expr)                                                                         \
	do {                                                                                        \
		int e__ = (expr);                                                                       \
		if (e__ != 0) {                                                                         \
			ctx->err = std::string("vgsdf_outlines: " #expr ": ") + hipGetErrorString((hipError_t)e__); \
			return VGSDF_E_HIP;                                                                 \
		}                                                                                       \
	} while (0)

// ---- the front-end as two halves: submit (everything enqueued, nothing waited for) and wait (the one
// synchronisation, the read-back, second launches if a guess was too small).  With a destination (`spec_out`,
// `spec_cap` bytes) the raster is enqueued right behind the front-end kernels, before the host has seen the plan: its
// grid and every capacity are guesses the plan kernel checks on the device (PlanHeader::ok).  When they hold, the
// bitmaps are in `spec_out` after the wait (written there by the kernel itself if the buffer is page-locked).
namespace {
struct FeDev { // device views of a submitted batch
	const vgsdf::OutlineCmd *cmds;
	const double *scale, *shift;
	const uint32_t *cmd_off;
	vgsdf::OutlineRect *rects;
	vgsdf::PlanHeader *hdr;
	vgsdf::GlyphDesc *descs;
};
FeDev fe_dev(FrontEnd &fe)
{
	const FePending &p = fe.pend;
	const size_t n = p.n;
	FeDev d;
	d.cmds = (const vgsdf::OutlineCmd *)fe.cmds.p;
	d.scale = (const double *)fe.meta.p;
	d.shift = (const double *)((const uint8_t *)fe.meta.p + 8 * n);
	d.cmd_off = (const uint32_t *)((const uint8_t *)fe.meta.p + 16 * n);
	d.rects = (vgsdf::OutlineRect *)fe.rects_hdr.p;
	d.hdr = (vgsdf::PlanHeader *)((uint8_t *)fe.rects_hdr.p + p.hdr_off);
	d.descs = (vgsdf::GlyphDesc *)fe.descs.p;
	return d;
}
int fe_launch_plan(vgsdf_ctx *ctx, FrontEnd &fe, uint32_t spans_launched)
{
	const FePending &p = fe.pend;
	const FeDev d = fe_dev(fe);
	return vgsdf_outline_plan(d.rects, p.n, p.span ? 1 : 0, (uint32_t)vgsdf_filtered_delta_cap(), p.span_max, p.span_budget,
	                          (uint32_t)std::min<size_t>(fe.tile_cap, 0x7FFFFFFFu), d.descs, (uint2 *)fe.tiles.p, d.hdr,
	                          fe.flag_word(), (unsigned long long)fe.seg_cap, (unsigned long long)p.spec_cap,
	                          spans_launched, p.d_pbf_pre, p.d_pbf_fix,
	                          p.d_pbf_fix ? (unsigned long long *)((uint8_t *)fe.rects_hdr.p + p.at_off) : nullptr, fe.next_flag_word(), ctx->stream);
}
// The second flattening pass and the raster's chunk boxes.  Boxes: by default the first workgroups of the pass's own grid take
// them from the commands' boxes (outline_kernels.hip, chunk_boxes_of_glyph: supersets of the exact boxes, no launch of their
// own); VGSDF_CMD_BOXES=0 (measurement switch): from the segments, by sdf_chunk_boxes behind the pass
int fe_launch_emit(vgsdf_ctx *ctx, FrontEnd &fe)
{
	const FePending &p = fe.pend;
	const FeDev d = fe_dev(fe);
	static const char *cb_env = std::getenv("VGSDF_CMD_BOXES");
	const bool cmd_boxes = p.span && !(cb_env && cb_env[0] == '0');
	int e = vgsdf_outline_emit_segments(d.cmds, p.n_cmds, (const uint8_t *)fe.cmd_open.p, d.scale, d.shift, (const uint32_t *)fe.pt_local.p,
	                                    (const vgsdf::RingRec *)fe.rings.p, (const uint32_t *)fe.cmd_ring.p, d.descs, d.hdr,
	                                    (unsigned long long)fe.seg_cap, (double *)fe.seg.p, (const unsigned long long *)fe.cmd_mask.p,
	                                    cmd_boxes ? p.n : 0u, d.cmd_off, fe.cmd_box.p, fe.boxes.p, ctx->stream);
	if (e == 0 && p.span && !cmd_boxes)
		e = vgsdf_launch_chunk_boxes(d.descs, p.n, (const double *)fe.seg.p, (const double *)fe.seg.p + 1, (const double *)fe.seg.p + 2,
		                             (const double *)fe.seg.p + 3, 4, fe.boxes.p, d.hdr, (unsigned long long)fe.seg_cap, ctx->stream);
	return e;
}
hipError_t fe_ensure_tiles(FrontEnd &fe, size_t want)
{
	if (want <= fe.tile_cap)
		return hipSuccess;
	hipError_t e = fe.tiles.ensure(sizeof(uint2) * want);
	if (e == hipSuccess)
		fe.tile_cap = fe.tiles.cap / sizeof(uint2);
	return e;
}
hipError_t fe_ensure_segs(FrontEnd &fe, size_t want, uint32_t n_glyphs)
{
	if (want > fe.seg_cap) {
		if (hipError_t e = fe.seg.ensure(32 * want + 32); e != hipSuccess)
			return e;
		fe.seg_cap = fe.seg.cap / 32 - 1;
	}
	return fe.boxes.ensure(vgsdf_chunk_box_bytes(fe.seg_cap, n_glyphs) + 16);
}
} // namespace

// the two input forms of a submission: 28-byte command records, or kinds + coordinates (vgsdf_outlines_packed)
struct FeInput {
	uint32_t n_glyphs = 0;
	const uint32_t *cmd_off = nullptr;
	const double *scale = nullptr, *shift_x = nullptr;
	const vgsdf_outline_cmd *cmds = nullptr;
	const uint32_t *dat_off = nullptr;
	const uint8_t *kinds = nullptr;
	const float *coords = nullptr;
	const uint32_t *pbf_pre = nullptr; // in-place PBF assembly (vgsdf_outlines_packed): both or neither
	const uint8_t *pbf_fix = nullptr;
	bool packed = false;
	// vgsdf_outlines_glyf: the glyphs' `glyf` arrays instead of commands (cmd_off counts command SLOTS)
	bool glyf = false;
	const vgsdf_glyf_part *parts = nullptr;
	uint32_t n_parts = 0;
	const uint8_t *bytes = nullptr;
	uint32_t n_bytes = 0;
};

static int fe_submit(vgsdf_ctx *ctx, const FeInput *in, uint8_t *spec_out, size_t spec_cap)
{
	const double tr0 = fe_now();
	if (!ctx)
		return VGSDF_E_ARG;
	if (!in || (in->n_glyphs && (!in->cmd_off || !in->scale || !in->shift_x || (in->packed && !in->dat_off)))) {
		ctx->err = "vgsdf_outlines: NULL argument";
		return VGSDF_E_ARG;
	}
	if ((in->pbf_pre == nullptr) != (in->pbf_fix == nullptr) || (in->pbf_fix && !in->packed && !in->glyf)) {
		ctx->err = "vgsdf_outlines: pbf_pre and pbf_fix come together (packed and glyf forms only)";
		return VGSDF_E_ARG;
	}
	static_assert(sizeof(vgsdf_glyf_part) == 48, "ABI struct mirrors the kernel struct");
	static_assert(sizeof(vgsdf_outline_cmd) == sizeof(vgsdf::OutlineCmd), "ABI struct mirrors the kernel struct");
	static_assert(sizeof(vgsdf_rect) == sizeof(vgsdf::OutlineRect), "ABI struct mirrors the kernel struct");
	const uint32_t n = in->n_glyphs;
	if (n && in->cmd_off[0] != 0) {
		ctx->err = "vgsdf_outlines: cmd_off[0] must be 0";
		return VGSDF_E_ARG;
	}
	const uint32_t n_cmds = n ? in->cmd_off[n] : 0;
	if (n_cmds && !in->glyf && (in->packed ? !in->kinds : !in->cmds)) {
		ctx->err = "vgsdf_outlines: NULL command array";
		return VGSDF_E_ARG;
	}
	if (in->glyf && ((in->n_parts && (!in->parts || !in->bytes)) || (in->n_bytes & 3u))) {
		ctx->err = "vgsdf_outlines_glyf: NULL parts / bytes, or n_bytes not a multiple of 4";
		return VGSDF_E_ARG;
	}
	if (in->packed && n && (in->dat_off[0] != 0 || (in->dat_off[n] && !in->coords))) {
		ctx->err = in->dat_off[0] != 0 ? "vgsdf_outlines: dat_off[0] must be 0" : "vgsdf_outlines: NULL coordinate array";
		return VGSDF_E_ARG;
	}
	const uint32_t n_floats = in->packed && n ? in->dat_off[n] : 0u;
	// The walks over the input — offsets monotone, parts tiling the command slots inside their glyphs and inside `bytes`, scales —
	// are what every kernel's indexing rests on, so they come before the first kernel that reads the input; but not before the
	// UPLOAD, which reads nothing of it: a single-block submission starts its copy first and validates under it (25 k entries
	// of a 21-font group: ~45 us of this thread that the device used to wait for).
	uint32_t glyf_max_cap = 0, glyf_max_len = 0;
	bool parts_inside_glyphs = true; // every part's slots lie inside ONE glyph's range (what a sound caller sends)
	bool scales_plain = true;        // every scale positive and finite
	auto validate = [&]() -> int {
		uint32_t bad = 0;
		for (uint32_t g = 0; g < n; g++) {
			bad |= in->cmd_off[g + 1] < in->cmd_off[g];
			scales_plain = scales_plain && in->scale[g] > 0.0 && in->scale[g] < HUGE_VAL;
		}
		if (bad) {
			ctx->err = "vgsdf_outlines: cmd_off not monotone";
			return VGSDF_E_ARG;
		}
		if (in->glyf) {
			// the parts tile the command slots in order, and their bytes lie inside `bytes` (what the bytes SAY is checked on
			// the device, entry by entry)
			uint64_t slots = 0;
			uint32_t gi = 0;
			for (uint32_t i = 0; i < in->n_parts; i++) {
				const vgsdf_glyf_part &pt = in->parts[i];
				while (gi < n && in->cmd_off[gi + 1] <= pt.cmd_at)
					gi++;
				parts_inside_glyphs = parts_inside_glyphs && gi < n && pt.cmd_at >= in->cmd_off[gi] && (uint64_t)pt.cmd_at + pt.cmd_cap <= in->cmd_off[gi + 1];
				glyf_max_cap = std::max(glyf_max_cap, pt.cmd_cap);
				glyf_max_len = std::max(glyf_max_len, pt.byte_len);
				if (pt.cmd_at != slots || (pt.byte_off & 3u) || pt.byte_off > in->n_bytes || pt.byte_len > in->n_bytes - pt.byte_off ||
				    pt.n_contours == 0) {
					ctx->err = "vgsdf_outlines_glyf: parts must tile the command slots in order, with 4-aligned byte ranges inside `bytes`";
					return VGSDF_E_ARG;
				}
				slots += pt.cmd_cap;
			}
			if (slots != n_cmds) {
				ctx->err = "vgsdf_outlines_glyf: cmd_off[n_glyphs] differs from the parts' command slots";
				return VGSDF_E_ARG;
			}
		}
		if (in->packed) {
			for (uint32_t g = 0; g < n; g++)
				bad |= in->dat_off[g + 1] < in->dat_off[g];
			if (bad) {
				ctx->err = "vgsdf_outlines: dat_off not monotone";
				return VGSDF_E_ARG;
			}
		}
		return VGSDF_OK;
	};
	bool validated = false;
	// (the command kinds are checked on the device: the kernels treat an unknown kind as a no-op and the context
	// pass raises the batch's error flag, so nothing unsafe runs and the host need not walk the commands)
	(void)hipSetDevice(ctx->device);
	if (!ctx->fe)
		ctx->fe = new (std::nothrow) FrontEnd();
	if (!ctx->fe) {
		ctx->err = "vgsdf_outlines: out of host memory";
		return VGSDF_E_OOM;
	}
	FrontEnd &fe = *ctx->fe;
	if (fe.pend.active) {
		ctx->err = "vgsdf_outlines_submit: the previous submission of this context has not been waited for";
		return VGSDF_E_ARG;
	}
	fe.prepared = false;
	fe.peeked = false;
	fe.n_glyphs = n;
	fe.n_cmds = n_cmds;
	fe.n_segs = 0;
	fe.out_bytes = 0;
	FePending &p = fe.pend;
	p = FePending{};
	p.n = n;
	p.n_cmds = n_cmds;
	p.spec_out = spec_out;
	p.spec_cap = spec_out ? spec_cap : 0;
	p.t0 = tr0;
	if (n == 0) {
		fe.batch.stats = vgsdf_stats{};
		p.active = true;
		p.t1 = fe_now();
		return VGSDF_OK;
	}
	p.t1 = fe_now();
	hipStream_t st = ctx->stream;
	p.span = ctx->variant == 0 || (ctx->variant >= 50 && ctx->variant <= 69);
	// span policy of the work list (same switches as build_descs_and_tiles)
	const char *sm = std::getenv("VGSDF_SPAN_MAX");
	p.span_max = sm ? (uint32_t)std::min(4, std::max(1, std::atoi(sm))) : 4u;
	const char *sb = std::getenv("VGSDF_SPAN_BUDGET");
	p.span_budget = sb ? (uint32_t)std::max(1, std::atoi(sb)) : default_span_budget(n);
	FE_TRY(fe.cmds.ensure(sizeof(vgsdf::OutlineCmd) * (size_t)(n_cmds + 1)));
	// per-glyph inputs (scale, shift, command offsets) travel as ONE block through pinned staging
	const size_t meta_scale = 0, meta_shift = 8 * (size_t)n, meta_off = 16 * (size_t)n, meta_dat = meta_off + 4 * (size_t)(n + 1);
	const size_t meta_bytes = meta_dat + (in->packed ? 4 * (size_t)(n + 1) : 0);
	// packed input whose arrays sit back to back in one page-locked block, in the order of the device's own layout
	// (scale | shift_x | cmd_off | dat_off | pad to 8 | coords | kinds): ONE copy instead of three
	// (with in-place PBF assembly: ... | kinds | pad to 4 | pbf_pre u32[n] | pbf_fix u8[n])
	const size_t blob_coords = (meta_bytes + 7) & ~(size_t)7, blob_kinds = blob_coords + 4 * (size_t)n_floats;
	const bool pbf = in->pbf_fix != nullptr;
	const size_t blob_pre = (blob_kinds + n_cmds + 3) & ~(size_t)3, blob_fix = blob_pre + 4 * (size_t)n;
	const size_t blob_bytes = pbf ? blob_fix + n : blob_kinds + n_cmds;
	const uint8_t *hb = (const uint8_t *)in->scale;
	// glyf form in one block: scale | shift_x | cmd_off | (pad to 8) | parts | bytes | pbf_pre | pbf_fix
	const size_t gl_parts = (meta_off + 4 * (size_t)(n + 1) + 7) & ~(size_t)7, gl_bytes = gl_parts + sizeof(vgsdf_glyf_part) * (size_t)in->n_parts;
	const size_t gl_pre = gl_bytes + in->n_bytes, gl_fix = gl_pre + 4 * (size_t)n;
	const size_t gl_total = in->pbf_fix ? gl_fix + n : gl_pre;
	const bool gblob = in->glyf && (const uint8_t *)in->shift_x == hb + meta_shift && (const uint8_t *)in->cmd_off == hb + meta_off &&
	                   (const uint8_t *)in->parts == hb + gl_parts && in->bytes == hb + gl_bytes &&
	                   (!in->pbf_fix || ((const uint8_t *)in->pbf_pre == hb + gl_pre && in->pbf_fix == hb + gl_fix)) && is_pinned(hb, gl_total);
	const bool blob = in->packed && (const uint8_t *)in->shift_x == hb + meta_shift && (const uint8_t *)in->cmd_off == hb + meta_off &&
	                  (const uint8_t *)in->dat_off == hb + meta_dat && (const uint8_t *)in->coords == hb + blob_coords &&
	                  in->kinds == hb + blob_kinds &&
	                  (!pbf || ((const uint8_t *)in->pbf_pre == hb + blob_pre && in->pbf_fix == hb + blob_fix)) && is_pinned(hb, blob_bytes);
	FE_TRY(fe.meta.ensure((blob ? blob_bytes : (in->glyf ? gl_total : meta_bytes)) + 16));
	FE_TRY(fe.h_stage.ensure(meta_bytes + 16));
	FE_TRY(fe.cmd_open.ensure((size_t)n_cmds + 1));
	FE_TRY(fe.counts.ensure(4 * (size_t)(n_cmds + 1)));
	FE_TRY(fe.pt_local.ensure(4 * ((size_t)n_cmds + n + 2)));
	FE_TRY(fe.cmd_box.ensure(32 * (size_t)(n_cmds + 1)));
	FE_TRY(fe.cmd_mask.ensure(8 * (size_t)(n_cmds + 1)));
	FE_TRY(fe.rings.ensure(sizeof(vgsdf::RingRec) * (size_t)(n_cmds + 1)));
	FE_TRY(fe.cmd_ring.ensure(4 * (size_t)(n_cmds + 1)));
	p.hdr_off = align_up(sizeof(vgsdf::OutlineRect) * (size_t)n, 16); // rects and totals: one block, one read-back
	p.at_off = align_up(p.hdr_off + sizeof(vgsdf::PlanHeader), 16);
	p.rh_bytes = pbf ? p.at_off + 8 * (size_t)n : p.hdr_off + sizeof(vgsdf::PlanHeader);
	FE_TRY(fe.rects_hdr.ensure(p.rh_bytes));
	FE_TRY(fe.h_rects.ensure(p.rh_bytes));
	FE_TRY(fe.descs.ensure(sizeof(vgsdf::GlyphDesc) * (size_t)n + 16));
	FE_TRY(fe.flag.ensure(32));
	// capacities of what only the device knows the size of: the work list and the segment arrays.  Guessed from
	// the input (and kept from earlier batches); the plan / emit kernels write nothing past them and the totals
	// that come back with the rects say whether a second launch is needed.
	FE_TRY(fe_ensure_tiles(fe, 2 * (size_t)n + 1024));
	FE_TRY(fe_ensure_segs(fe, 12 * (size_t)n_cmds + 4096, n));

	if (std::getenv("VGSDF_TRACE") != nullptr)
		FE_TRY(hipEventRecord(ctx->ev0, st));
	// a single page-locked block that the device can address is uploaded by a kernel (outline_kernels.hip, copy_in) instead of
	// the copy engine; VGSDF_COPY_KERNEL=0: measurement switch
	static const char *ck_env = std::getenv("VGSDF_COPY_KERNEL");
	const void *hb_mapped = nullptr;
	if ((gblob || blob) && !(ck_env && ck_env[0] == '0') && ((uintptr_t)hb & 15u) == 0)
		hb_mapped = pinned_device_ptr(const_cast<uint8_t *>(hb), gblob ? gl_total : blob_bytes);
	if (!(hb_mapped != nullptr && (gblob || blob))) { // (not one block uploaded by a kernel: validate first, as ever)
		if (int rc = validate(); rc != VGSDF_OK)
			return rc;
		validated = true;
	}
	// error word of this submission (FrontEnd::flag_slot)
	if (!fe.flags_clean)
		FE_TRY(hipMemsetAsync(fe.flag.p, 0, 32, st));
	fe.flags_clean = false; // (until everything below is enqueued: its plan kernel zeroes the other slot)
	fe.flag_slot ^= 1u;
	uint32_t *const flagw = fe.flag_word();
	const uint8_t *d_kinds = nullptr;
	const float *d_coords = nullptr;
	const uint8_t *d_parts = nullptr, *d_bytes = nullptr;
	if (in->glyf) {
		// (device layout = the single-block layout, whether the arrays arrive as one block or one by one)
		uint8_t *dm = (uint8_t *)fe.meta.p;
		if (gblob && hb_mapped) {
			FE_KERNEL(vgsdf_copy_in(hb_mapped, dm, gl_total, st));
		} else if (gblob) {
			FE_TRY(hipMemcpyAsync(dm, hb, gl_total, hipMemcpyHostToDevice, st));
		} else {
			FE_TRY(hipMemcpyAsync(dm + meta_scale, in->scale, 8 * (size_t)n, hipMemcpyHostToDevice, st));
			FE_TRY(hipMemcpyAsync(dm + meta_shift, in->shift_x, 8 * (size_t)n, hipMemcpyHostToDevice, st));
			FE_TRY(hipMemcpyAsync(dm + meta_off, in->cmd_off, 4 * (size_t)(n + 1), hipMemcpyHostToDevice, st));
			if (in->n_parts)
				FE_TRY(hipMemcpyAsync(dm + gl_parts, in->parts, sizeof(vgsdf_glyf_part) * (size_t)in->n_parts, hipMemcpyHostToDevice, st));
			if (in->n_bytes)
				FE_TRY(hipMemcpyAsync(dm + gl_bytes, in->bytes, in->n_bytes, hipMemcpyHostToDevice, st));
			if (pbf) {
				FE_TRY(hipMemcpyAsync(dm + gl_pre, in->pbf_pre, 4 * (size_t)n, hipMemcpyHostToDevice, st));
				FE_TRY(hipMemcpyAsync(dm + gl_fix, in->pbf_fix, (size_t)n, hipMemcpyHostToDevice, st));
			}
		}
		d_parts = dm + gl_parts;
		d_bytes = dm + gl_bytes;
		if (pbf) {
			p.d_pbf_pre = (const uint32_t *)(dm + gl_pre);
			p.d_pbf_fix = dm + gl_fix;
		}
	} else if (blob) {
		if (hb_mapped)
			FE_KERNEL(vgsdf_copy_in(hb_mapped, fe.meta.p, blob_bytes, st));
		else
			FE_TRY(hipMemcpyAsync(fe.meta.p, hb, blob_bytes, hipMemcpyHostToDevice, st));
		d_coords = (const float *)((const uint8_t *)fe.meta.p + blob_coords);
		d_kinds = (const uint8_t *)fe.meta.p + blob_kinds;
		if (pbf) {
			p.d_pbf_pre = (const uint32_t *)((const uint8_t *)fe.meta.p + blob_pre);
			p.d_pbf_fix = (const uint8_t *)fe.meta.p + blob_fix;
		}
	} else if (in->packed) {
		FE_TRY(fe.kinds.ensure((size_t)n_cmds + 16));
		FE_TRY(fe.coords.ensure(4 * (size_t)n_floats + 16));
		if (n_cmds)
			FE_TRY(hipMemcpyAsync(fe.kinds.p, in->kinds, (size_t)n_cmds, hipMemcpyHostToDevice, st));
		if (n_floats)
			FE_TRY(hipMemcpyAsync(fe.coords.p, in->coords, 4 * (size_t)n_floats, hipMemcpyHostToDevice, st));
		d_kinds = (const uint8_t *)fe.kinds.p;
		d_coords = (const float *)fe.coords.p;
	} else if (n_cmds) {
		FE_TRY(hipMemcpyAsync(fe.cmds.p, in->cmds, sizeof(vgsdf::OutlineCmd) * (size_t)n_cmds, hipMemcpyHostToDevice, st));
	}
	if (!blob && !in->glyf) {
		uint8_t *hm = (uint8_t *)fe.h_stage.p;
		std::memcpy(hm + meta_scale, in->scale, 8 * (size_t)n);
		std::memcpy(hm + meta_shift, in->shift_x, 8 * (size_t)n);
		std::memcpy(hm + meta_off, in->cmd_off, 4 * (size_t)(n + 1));
		if (in->packed)
			std::memcpy(hm + meta_dat, in->dat_off, 4 * (size_t)(n + 1));
		FE_TRY(hipMemcpyAsync(fe.meta.p, hm, meta_bytes, hipMemcpyHostToDevice, st));
	}
	if (pbf && !blob && !in->glyf) { // arrays that do not sit in the single-copy block: their own copies
		FE_TRY(fe.pbf_in.ensure(5 * (size_t)n + 16));
		FE_TRY(hipMemcpyAsync(fe.pbf_in.p, in->pbf_pre, 4 * (size_t)n, hipMemcpyHostToDevice, st));
		FE_TRY(hipMemcpyAsync((uint8_t *)fe.pbf_in.p + 4 * (size_t)n, in->pbf_fix, (size_t)n, hipMemcpyHostToDevice, st));
		p.d_pbf_pre = (const uint32_t *)fe.pbf_in.p;
		p.d_pbf_fix = (const uint8_t *)fe.pbf_in.p + 4 * (size_t)n;
	}
	const FeDev d = fe_dev(fe);

	// the raster launch enqueued behind the front-end: default kernel only, destination the caller's page-locked
	// buffer itself or, for pageable memory, the context's device buffer
	p.spec = spec_out != nullptr && spec_cap != 0 && ctx->variant == 0;
	if (p.spec) {
		if (void *mapped = pinned_device_ptr(spec_out, spec_cap)) {
			p.d_spec = (uint8_t *)mapped;
			p.spec_direct = true;
		} else {
			FE_TRY(fe.out.ensure(spec_cap + 16));
			p.d_spec = (uint8_t *)fe.out.p;
		}
		const size_t guess = fe.last_spans ? (size_t)fe.last_spans + fe.last_spans / 2 + 256 : fe.tile_cap;
		p.launch_spans = (uint32_t)std::min<size_t>(std::min(guess, fe.tile_cap), 0x7FFFFFFFu);
	}
	if (!validated) { // the upload is under way: now the walks over the input, before the first kernel that reads it
		if (int rc = validate(); rc != VGSDF_OK)
			return rc;
	}
	// glyf form: the decoder writes the context bytes itself (the ring state follows from the contour rules) when no glyph
	// of the batch has an odd scale (not positive and finite: bit 1 of the context byte, which only the context pass forms)
	// and no part straddles two glyphs (the decoder's rule is per part; the ring pass trusts the context bytes to be those of
	// the glyph's own command sequence — a byte that says "open" in front of a glyph's first command would index a ring
	// that does not exist)
	bool decode_makes_context = in->glyf && parts_inside_glyphs && scales_plain;
	static const char *fuse_env = std::getenv("VGSDF_FUSE_CONTEXT"); // (measurement switch)
	if (fuse_env && fuse_env[0] == '0')
		decode_makes_context = false;
	if (in->glyf)
		FE_KERNEL(vgsdf_glyf_decode(d_parts, in->n_parts, d_bytes, (vgsdf::OutlineCmd *)fe.cmds.p, flagw, glyf_max_cap, glyf_max_len,
		                            decode_makes_context ? (uint8_t *)fe.cmd_open.p : nullptr, st));
	if (in->packed)
		FE_KERNEL(vgsdf_outline_context_packed(d_kinds, d_coords,
		                                       (const uint32_t *)((const uint8_t *)fe.meta.p + meta_dat), d.cmd_off, d.scale, n,
		                                       (vgsdf::OutlineCmd *)fe.cmds.p, (uint8_t *)fe.cmd_open.p, flagw, st));
	else if (!decode_makes_context)
		FE_KERNEL(vgsdf_outline_context(d.cmds, d.cmd_off, d.scale, n, (uint8_t *)fe.cmd_open.p, flagw, st));
	FE_KERNEL(vgsdf_outline_count(d.cmds, (const uint8_t *)fe.cmd_open.p, n_cmds, d.cmd_off, n, d.scale, d.shift,
	                              (uint32_t *)fe.counts.p, fe.cmd_box.p, (unsigned long long *)fe.cmd_mask.p, flagw, st));
	FE_KERNEL(vgsdf_outline_rings(d.cmds, d.cmd_off, (const uint8_t *)fe.cmd_open.p, d.scale, d.shift, n,
	                              (const uint32_t *)fe.counts.p, (uint32_t *)fe.pt_local.p,
	                              fe.cmd_box.p, (vgsdf::RingRec *)fe.rings.p, (uint32_t *)fe.cmd_ring.p, d.rects,
	                              flagw, st));
	FE_KERNEL(fe_launch_plan(ctx, fe, p.launch_spans));
	// The front-end's results (rects, totals, positions of the bitmaps) are final once the plan has run: they travel back
	// on a stream of their own, beside the flattening and the raster instead of behind them — the host can have them a
	// good 100 us before the bitmaps (vgsdf_outlines_peek), and the end of the submission loses a copy and its hand-over.
	static const char *early_env = std::getenv("VGSDF_EARLY_COPY"); // (measurement switch: 0 = read-back behind the raster, as in round 2)
	const bool early_copy = !(early_env && early_env[0] == '0');
	if (early_copy) {
		FE_TRY(hipEventRecord(ctx->ev_plan, st));
		FE_TRY(hipStreamWaitEvent(ctx->copy_stream, ctx->ev_plan, 0));
		FE_TRY(hipMemcpyAsync(fe.h_rects.p, fe.rects_hdr.p, p.rh_bytes, hipMemcpyDeviceToHost, ctx->copy_stream));
		FE_TRY(hipEventRecord(ctx->ev_rects, ctx->copy_stream));
	}
	FE_KERNEL(fe_launch_emit(ctx, fe));
	if (p.spec)
		FE_KERNEL(vgsdf_launch_span_planned(d.descs, (const uint2 *)fe.tiles.p, p.launch_spans, (const double *)fe.seg.p,
		                                    (const double *)fe.seg.p + 1, (const double *)fe.seg.p + 2, (const double *)fe.seg.p + 3, 4,
		                                    p.d_spec, fe.boxes.p, d.hdr, st));
	if (!early_copy) {
		FE_TRY(hipMemcpyAsync(fe.h_rects.p, fe.rects_hdr.p, p.rh_bytes, hipMemcpyDeviceToHost, st));
		FE_TRY(hipEventRecord(ctx->ev_rects, st));
	}
	static const bool trace_span = std::getenv("VGSDF_TRACE") != nullptr;
	if (trace_span)
		FE_TRY(hipEventRecord(ctx->ev1, st));
	fe.flags_clean = true; // the plan kernel enqueued above leaves the other slot zeroed for the next submission
	p.active = true;
	return VGSDF_OK;
}

static int fe_wait(vgsdf_ctx *ctx, vgsdf_rect *rects_out, uint64_t *out_bytes, uint64_t *n_segments, int *rendered)
{
	static const bool trace = std::getenv("VGSDF_TRACE") != nullptr;
	if (rendered)
		*rendered = 0;
	if (out_bytes)
		*out_bytes = 0;
	if (n_segments)
		*n_segments = 0;
	if (!ctx)
		return VGSDF_E_ARG;
	if (!ctx->fe || !ctx->fe->pend.active) {
		ctx->err = "vgsdf_outlines_wait: nothing was submitted";
		return VGSDF_E_ARG;
	}
	FrontEnd &fe = *ctx->fe;
	FePending &p = fe.pend;
	const uint32_t n = p.n;
	if (n && !rects_out) {
		ctx->err = "vgsdf_outlines: NULL argument";
		return VGSDF_E_ARG;
	}
	p.active = false;
	if (n == 0) {
		fe.prepared = true;
		if (rendered && p.spec_out)
			*rendered = 1;
		return VGSDF_OK;
	}
	(void)hipSetDevice(ctx->device);
	hipStream_t st = ctx->stream;
	FE_TRY(hipStreamSynchronize(st)); // the one synchronisation of the submission
	FE_TRY(hipEventSynchronize(ctx->ev_rects)); // (the read-back finished long ago: it left right behind the plan)
	const double tr2 = fe_now();
	std::memcpy(rects_out, fe.h_rects.p, sizeof(vgsdf_rect) * (size_t)n);
	vgsdf::PlanHeader hdr;
	std::memcpy(&hdr, (const uint8_t *)fe.h_rects.p + p.hdr_off, sizeof hdr);
	if (hdr.error & 16u) {
		ctx->err = "vgsdf_outlines_glyf: a `glyf` entry whose arrays do not fit its bytes (ttf-parser drops such a glyph): record this batch "
		           "with the host's reader";
		return VGSDF_E_GLYF;
	}
	if (hdr.error & 2u) {
		ctx->err = "vgsdf_outlines_prepare: unknown command kind";
		return VGSDF_E_ARG;
	}
	if (hdr.error & 8u) {
		ctx->err = "vgsdf_outlines: dat_off does not match the command kinds";
		return VGSDF_E_ARG;
	}
	if (hdr.error & 4u) {
		ctx->err = "vgsdf_outlines_prepare: internal error (a cubic exceeded its subdivision depth bound)";
		return VGSDF_E_HIP;
	}
	if (hdr.error) {
		ctx->err = "vgsdf_outlines_prepare: a glyph flattens to more than 2^28 points, the batch to more than 2^32 - 1 segments, or a "
		           "bitmap exceeds 2^32 pixels (non-finite or absurd control points?)";
		return VGSDF_E_ARG;
	}
	if (hdr.n_spans > 0x7FFFFFFFu) {
		ctx->err = "vgsdf_outlines_prepare: batch too large (tile count exceeds 2^31-1); split it";
		return VGSDF_E_ARG;
	}
	// second launches when a capacity guess was too small (first batches of a context, unusual fonts)
	const bool replan = hdr.n_spans > fe.tile_cap, reemit = hdr.n_segments > fe.seg_cap;
	if (replan) {
		FE_TRY(fe_ensure_tiles(fe, (size_t)hdr.n_spans + hdr.n_spans / 4 + 1024));
		FE_KERNEL(fe_launch_plan(ctx, fe, 0));
	}
	if (reemit) {
		FE_TRY(fe_ensure_segs(fe, (size_t)hdr.n_segments + hdr.n_segments / 4 + 4096, n));
		FE_KERNEL(fe_launch_emit(ctx, fe));
	}
	const double tr3 = fe_now();

	uint64_t n_pairs = 0, n_pixels = 0;
	for (uint32_t g = 0; g < n; g++) {
		const vgsdf_rect &r = rects_out[g];
		if (r.has_raster) {
			n_pairs += (uint64_t)r.w * r.h * r.n_segments;
			n_pixels += (uint64_t)r.w * r.h;
		}
	}
	const FeDev d = fe_dev(fe);
	fe.n_segs = (uint32_t)hdr.n_segments;
	fe.out_bytes = hdr.out_bytes;
	vgsdf_dbatch &b = fe.batch;
	b.stats.n_glyphs = n;
	b.stats.n_segments = fe.n_segs;
	b.stats.n_pixels = n_pixels; // (out_bytes is larger with in-place PBF assembly: headers and gaps)
	b.stats.n_pairs = n_pairs;
	b.stats.n_tiles = hdr.n_spans;
	b.stats.alg_bytes = 32 * (uint64_t)fe.n_segs + 32 * (uint64_t)n + n_pixels;
	b.out_bytes = (size_t)fe.out_bytes;
	b.n_main = hdr.n_main;
	b.span_list = p.span;
	b.tile_order = 1; // the device-built list is dispatched in list order
	fe.last_spans = hdr.n_spans;
	const bool done = p.spec && hdr.ok != 0; // the raster behind the plan ran over the whole list
	if (!(done && p.spec_direct))
		FE_TRY(fe.out.ensure(std::max((size_t)fe.out_bytes, done ? p.spec_cap : (size_t)0) + 16));
	b.d_glyphs = d.descs;
	b.d_tiles = (uint2 *)fe.tiles.p;
	b.d_sx = (double *)fe.seg.p;
	b.d_sy = (double *)fe.seg.p + 1;
	b.d_ex = (double *)fe.seg.p + 2;
	b.d_ey = (double *)fe.seg.p + 3;
	b.seg_stride = 4;
	b.d_out = (uint8_t *)fe.out.p;
	b.d_boxes = p.span ? fe.boxes.p : nullptr;
	fe.prepared = true;
	if (out_bytes)
		*out_bytes = fe.out_bytes;
	if (n_segments)
		*n_segments = fe.n_segs;
	if (p.spec_out && fe.out_bytes <= p.spec_cap) {
		int rc = VGSDF_OK;
		if (done) {
			if (!p.spec_direct && fe.out_bytes) { // pageable destination: the raster wrote the device buffer
				FE_TRY(hipMemcpyAsync(p.spec_out, p.d_spec, (size_t)fe.out_bytes, hipMemcpyDeviceToHost, st));
				FE_TRY(hipStreamSynchronize(st));
			}
		} else if (fe.out_bytes) { // a guess was too small (first batch of a context, a batch unlike the last one)
			rc = vgsdf_batch_launch(ctx, &fe.batch);
			if (rc == VGSDF_OK)
				rc = vgsdf_batch_download(ctx, &fe.batch, p.spec_out);
		}
		if (rc != VGSDF_OK)
			return rc;
		if (rendered)
			*rendered = 1;
	}
	if (trace) {
		float span_ms = 0;
		if (hipEventElapsedTime(&span_ms, ctx->ev0, ctx->ev1) == hipSuccess)
			std::fprintf(stderr, "[vgsdf] device span of the submission (upload ... last kernel, events on the kernel stream): %.1f us\n", span_ms * 1e3);
		else
			(void)hipGetLastError();
	}
	if (trace && p.spec_out)
		std::fprintf(stderr, "[vgsdf] one submission%s, %s destination\n", done ? "" : " (guess too small: second launches)",
		             p.spec_direct ? "page-locked" : "pageable");
	if (trace)
		std::fprintf(stderr, "[vgsdf] prepare: validate %.3f ms, submit ... read-back %.3f ms, second launches%s%s %.3f ms, host %.3f ms\n",
		             (p.t1 - p.t0) * 1e3, (tr2 - p.t1) * 1e3, replan ? " (plan)" : "", reemit ? " (emit)" : "", (tr3 - tr2) * 1e3,
		             (fe_now() - tr3) * 1e3);
	return VGSDF_OK;
}

static FeInput fe_input(const vgsdf_outlines *in)
{
	FeInput f;
	f.n_glyphs = in->n_glyphs;
	f.cmd_off = in->cmd_off;
	f.scale = in->scale;
	f.shift_x = in->shift_x;
	f.cmds = in->cmds;
	return f;
}

int vgsdf_outlines_prepare(vgsdf_ctx *ctx, const vgsdf_outlines *in, vgsdf_rect *rects_out, uint64_t *out_bytes,
                           uint64_t *n_segments)
{
	if (ctx && in && in->n_glyphs && !rects_out) {
		ctx->err = "vgsdf_outlines_prepare: NULL argument";
		return VGSDF_E_ARG;
	}
	if (ctx && !in) {
		ctx->err = "vgsdf_outlines: NULL argument";
		return VGSDF_E_ARG;
	}
	FeInput f;
	if (in)
		f = fe_input(in);
	const int rc = fe_submit(ctx, in ? &f : nullptr, nullptr, 0);
	return rc != VGSDF_OK ? rc : fe_wait(ctx, rects_out, out_bytes, n_segments, nullptr);
}

int vgsdf_outlines_submit(vgsdf_ctx *ctx, const vgsdf_outlines *in, uint8_t *out_bitmaps, size_t out_capacity)
{
	FeInput f;
	if (in)
		f = fe_input(in);
	return fe_submit(ctx, in ? &f : nullptr, out_bitmaps, out_bitmaps ? out_capacity : 0);
}

int vgsdf_outlines_submit_packed(vgsdf_ctx *ctx, const vgsdf_outlines_packed *in, uint8_t *out_bitmaps, size_t out_capacity)
{
	FeInput f;
	if (in) {
		f.n_glyphs = in->n_glyphs;
		f.cmd_off = in->cmd_off;
		f.scale = in->scale;
		f.shift_x = in->shift_x;
		f.dat_off = in->dat_off;
		f.kinds = in->kinds;
		f.coords = in->coords;
		f.pbf_pre = in->pbf_pre;
		f.pbf_fix = in->pbf_fix;
		f.packed = true;
	}
	return fe_submit(ctx, in ? &f : nullptr, out_bitmaps, out_bitmaps ? out_capacity : 0);
}

int vgsdf_outlines_submit_glyf(vgsdf_ctx *ctx, const vgsdf_outlines_glyf *in, uint8_t *out_bitmaps, size_t out_capacity)
{
	FeInput f;
	if (in) {
		f.n_glyphs = in->n_glyphs;
		f.cmd_off = in->cmd_off;
		f.scale = in->scale;
		f.shift_x = in->shift_x;
		f.pbf_pre = in->pbf_pre;
		f.pbf_fix = in->pbf_fix;
		f.glyf = true;
		f.parts = in->parts;
		f.n_parts = in->n_parts;
		f.bytes = in->bytes;
		f.n_bytes = in->n_bytes;
	}
	return fe_submit(ctx, in ? &f : nullptr, out_bitmaps, out_bitmaps ? out_capacity : 0);
}

int vgsdf_outlines_wait(vgsdf_ctx *ctx, vgsdf_rect *rects_out, uint64_t *out_bytes, uint64_t *n_segments, int *rendered)
{
	return fe_wait(ctx, rects_out, out_bytes, n_segments, rendered);
}

int vgsdf_outlines_peek(vgsdf_ctx *ctx, vgsdf_rect *rects_out, uint64_t *out_bytes, int *in_place)
{
	if (out_bytes)
		*out_bytes = 0;
	if (in_place)
		*in_place = 0;
	if (!ctx)
		return VGSDF_E_ARG;
	if (!ctx->fe || !ctx->fe->pend.active) {
		ctx->err = "vgsdf_outlines_peek: nothing was submitted";
		return VGSDF_E_ARG;
	}
	FrontEnd &fe = *ctx->fe;
	const FePending &p = fe.pend;
	if (p.n == 0)
		return VGSDF_OK;
	if (!rects_out) {
		ctx->err = "vgsdf_outlines: NULL argument";
		return VGSDF_E_ARG;
	}
	(void)hipSetDevice(ctx->device);
	FE_TRY(hipEventSynchronize(ctx->ev_rects));
	std::memcpy(rects_out, fe.h_rects.p, sizeof(vgsdf_rect) * (size_t)p.n);
	vgsdf::PlanHeader hdr;
	std::memcpy(&hdr, (const uint8_t *)fe.h_rects.p + p.hdr_off, sizeof hdr);
	if (out_bytes)
		*out_bytes = hdr.out_bytes;
	// the raster behind the plan runs over the whole list and stores through the caller's own (page-locked) buffer
	if (in_place)
		*in_place = p.spec && p.spec_direct && hdr.ok != 0 && hdr.error == 0;
	fe.peeked = true;
	return VGSDF_OK;
}

int vgsdf_outlines_pbf_positions(vgsdf_ctx *ctx, uint64_t *bitmap_at)
{
	if (!ctx || !bitmap_at)
		return VGSDF_E_ARG;
	if (!ctx->fe || !(ctx->fe->prepared || (ctx->fe->pend.active && ctx->fe->peeked)) || ctx->fe->pend.d_pbf_fix == nullptr) {
		ctx->err = "vgsdf_outlines_pbf_positions: the last batch was not submitted with pbf_pre / pbf_fix";
		return VGSDF_E_ARG;
	}
	const FePending &p = ctx->fe->pend;
	std::memcpy(bitmap_at, (const uint8_t *)ctx->fe->h_rects.p + p.at_off, 8 * (size_t)p.n);
	return VGSDF_OK;
}

int vgsdf_outlines_render_into(vgsdf_ctx *ctx, const vgsdf_outlines *in, vgsdf_rect *rects_out, uint8_t *out_bitmaps,
                               size_t out_capacity, uint64_t *out_bytes, uint64_t *n_segments, int *rendered)
{
	if (ctx && (!rendered || (in && in->n_glyphs && !rects_out))) {
		ctx->err = "vgsdf_outlines_render_into: NULL argument";
		return VGSDF_E_ARG;
	}
	FeInput f;
	if (in)
		f = fe_input(in);
	const int rc = fe_submit(ctx, in ? &f : nullptr, out_bitmaps, out_bitmaps ? out_capacity : 0);
	return rc != VGSDF_OK ? rc : fe_wait(ctx, rects_out, out_bytes, n_segments, rendered);
}

int vgsdf_outlines_render(vgsdf_ctx *ctx, uint8_t *out_bitmaps)
{
	if (!ctx)
		return VGSDF_E_ARG;
	if (!ctx->fe || !ctx->fe->prepared) {
		ctx->err = "vgsdf_outlines_render: call vgsdf_outlines_prepare first";
		return VGSDF_E_ARG;
	}
	FrontEnd &fe = *ctx->fe;
	if (fe.n_glyphs == 0 || fe.out_bytes == 0)
		return VGSDF_OK;
	if (!out_bitmaps) {
		ctx->err = "vgsdf_outlines_render: NULL output";
		return VGSDF_E_ARG;
	}
	static const bool trace = std::getenv("VGSDF_TRACE") != nullptr;
	const double t0 = fe_now();
	(void)hipSetDevice(ctx->device);
	FE_TRY(fe.out.ensure((size_t)fe.out_bytes + 16)); // (the one-submission form may have rendered elsewhere)
	fe.batch.d_out = (uint8_t *)fe.out.p;
	int rc = vgsdf_batch_launch(ctx, &fe.batch);
	if (rc != VGSDF_OK)
		return rc;
	rc = vgsdf_batch_download(ctx, &fe.batch, out_bitmaps);
	if (trace)
		std::fprintf(stderr, "[vgsdf] render: launch+D2H+sync %.3f ms\n", (fe_now() - t0) * 1e3);
	return rc;
}

int vgsdf_outlines_segments(vgsdf_ctx *ctx, uint32_t *seg_off, double *sx, double *sy, double *ex, double *ey)
{
	if (!ctx)
		return VGSDF_E_ARG;
	if (!ctx->fe || !ctx->fe->prepared) {
		ctx->err = "vgsdf_outlines_segments: call vgsdf_outlines_prepare first";
		return VGSDF_E_ARG;
	}
	FrontEnd &fe = *ctx->fe;
	(void)hipSetDevice(ctx->device);
	std::vector<vgsdf::GlyphDesc> hd;
	if (fe.n_glyphs && seg_off) {
		hd.resize(fe.n_glyphs);
		FE_TRY(hipMemcpyAsync(hd.data(), fe.descs.p, sizeof(vgsdf::GlyphDesc) * (size_t)fe.n_glyphs, hipMemcpyDeviceToHost, ctx->stream));
	}
	std::vector<double> rec;
	if (fe.n_segs && sx && sy && ex && ey) {
		rec.resize(4 * (size_t)fe.n_segs);
		FE_TRY(hipMemcpyAsync(rec.data(), fe.seg.p, 32 * (size_t)fe.n_segs, hipMemcpyDeviceToHost, ctx->stream));
	}
	FE_TRY(hipStreamSynchronize(ctx->stream));
	for (uint32_t g = 0; g < (uint32_t)hd.size(); g++) {
		seg_off[g] = hd[g].seg_off;
		seg_off[g + 1] = hd[g].seg_off + hd[g].n_seg;
	}
	for (size_t i = 0; i * 4 < rec.size(); i++) { // records -> the four arrays of the C ABI
		sx[i] = rec[4 * i];
		sy[i] = rec[4 * i + 1];
		ex[i] = rec[4 * i + 2];
		ey[i] = rec[4 * i + 3];
	}
	return VGSDF_OK;
}


// ---------------------------------------------------------------------------------------
// Run counters and their reduction over the contexts of a run (the one collective of the path —
// results need no exchange, every glyph is independent).  One process drives N devices, one context each; with
// distinct devices the sum is an RCCL all-reduce of 3 x u64 over a communicator of those devices (ncclCommInitAll;
// the library is loaded at first use with dlopen, so libvgsdf.so carries no link-time dependency on RCCL and shares
// the copy a host such as PyTorch has already mapped).  Contexts that share a device (a rehearsal of N lanes on one
// GPU) cannot form a communicator — RCCL refuses two ranks on one device — and are summed on the host.
// ---------------------------------------------------------------------------------------
namespace {
struct Rccl {
	void *lib = nullptr;
	int (*CommInitAll)(void **, int, const int *) = nullptr;
	int (*CommDestroy)(void *) = nullptr;
	int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
	int (*GroupStart)() = nullptr;
	int (*GroupEnd)() = nullptr;
	const char *(*GetErrorString)(int) = nullptr;
	std::string err;
	bool load()
	{
		if (lib)
			return true;
		for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
			lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
			if (lib)
				break;
		}
		if (!lib) {
			const char *why = dlerror(); // (one call: it clears the message)
			err = std::string("RCCL is not loadable (") + (why ? why : "librccl.so.1") + ")";
			return false;
		}
		auto sym = [&](const char *n) { return dlsym(lib, n); };
		CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
		CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
		AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
		GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
		GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
		GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
		if (!CommInitAll || !CommDestroy || !AllReduce || !GroupStart || !GroupEnd || !GetErrorString) {
			err = "RCCL: a collective entry point is missing from the library";
			dlclose(lib);
			lib = nullptr;
			return false;
		}
		return true;
	}
};
Rccl g_rccl;
// communicators by device list (creating one costs ~100 ms): kept for the life of the process
struct CommSet {
	std::vector<int> devices;
	std::vector<void *> comms;
};
std::vector<CommSet> g_comm_sets;
std::mutex g_comm_mu;
constexpr int kNcclUint64 = 5, kNcclSum = 0; // rccl.h: ncclDataType_t / ncclRedOp_t
} // namespace

void vgsdf_add_counters(vgsdf_ctx *ctx, uint64_t blocks, uint64_t glyphs, uint64_t pixels)
{
	if (!ctx)
		return;
	ctx->counters[0] += blocks;
	ctx->counters[1] += glyphs;
	ctx->counters[2] += pixels;
}

void vgsdf_reset_counters(vgsdf_ctx *ctx)
{
	if (ctx)
		ctx->counters[0] = ctx->counters[1] = ctx->counters[2] = 0;
}

namespace {
// the all-reduce proper: contexts on n DISTINCT devices (n >= 1).  VGSDF_OK: every rank holds `want` (checked)
int reduce_over_rccl(vgsdf_ctx **ctxs, int n, const std::vector<int> &devs, const uint64_t want[3])
{
	vgsdf_ctx *c0 = ctxs[0];
	std::lock_guard<std::mutex> lock(g_comm_mu);
	const char *no_rccl = std::getenv("VGSDF_NO_RCCL"); // (test switch: behave as if librccl were absent)
	if (no_rccl && no_rccl[0] == '1') {
		c0->err = "vgsdf_reduce_counters: RCCL switched off (VGSDF_NO_RCCL=1)";
		return VGSDF_E_HIP;
	}
	if (!g_rccl.load()) {
		c0->err = "vgsdf_reduce_counters: " + g_rccl.err;
		return VGSDF_E_HIP;
	}
	CommSet *set = nullptr;
	for (CommSet &cs : g_comm_sets)
		if (cs.devices == devs)
			set = &cs;
	if (!set) {
		CommSet cs;
		cs.devices = devs;
		cs.comms.assign((size_t)n, nullptr);
		const int rc = g_rccl.CommInitAll(cs.comms.data(), n, devs.data());
		if (rc != 0) {
			c0->err = std::string("vgsdf_reduce_counters: ncclCommInitAll: ") + g_rccl.GetErrorString(rc);
			return VGSDF_E_HIP;
		}
		g_comm_sets.push_back(std::move(cs));
		set = &g_comm_sets.back();
	}
	for (int i = 0; i < n; i++) {
		vgsdf_ctx *c = ctxs[i];
		HIP_TRY(c0, hipSetDevice(c->device));
		if (!c->d_counters)
			HIP_TRY(c0, hipMalloc((void **)&c->d_counters, 3 * sizeof(uint64_t)));
		HIP_TRY(c0, hipMemcpyAsync(c->d_counters, c->counters, 3 * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
		c->comm = set->comms[(size_t)i];
	}
	int rc = g_rccl.GroupStart();
	for (int i = 0; i < n && rc == 0; i++) {
		vgsdf_ctx *c = ctxs[i];
		(void)hipSetDevice(c->device);
		rc = g_rccl.AllReduce(c->d_counters, c->d_counters, 3, kNcclUint64, kNcclSum, c->comm, c->stream);
	}
	const int rc_end = g_rccl.GroupEnd();
	if (rc == 0)
		rc = rc_end;
	if (rc != 0) {
		c0->err = std::string("vgsdf_reduce_counters: RCCL all-reduce: ") + g_rccl.GetErrorString(rc);
		return VGSDF_E_HIP;
	}
	// every rank holds the sum; all of them are read back and must agree with each other and with the host's own sum
	for (int i = 0; i < n; i++) {
		vgsdf_ctx *c = ctxs[i];
		uint64_t got[3] = {0, 0, 0};
		HIP_TRY(c0, hipSetDevice(c->device));
		HIP_TRY(c0, hipMemcpyAsync(got, c->d_counters, sizeof got, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(c0, hipStreamSynchronize(c->stream));
		if (std::memcmp(got, want, sizeof got) != 0) {
			c0->err = "vgsdf_reduce_counters: the all-reduced counters of rank " + std::to_string(i) + " differ from the sum of the ranks' counters";
			return VGSDF_E_HIP;
		}
	}
	return VGSDF_OK;
}

int reduce_counters(vgsdf_ctx **ctxs, int n, uint64_t counters[3], bool strict)
{
	if (!ctxs || n <= 0 || !counters) {
		if (ctxs && n > 0 && ctxs[0])
			ctxs[0]->err = "vgsdf_reduce_counters: NULL argument";
		return VGSDF_E_ARG;
	}
	for (int i = 0; i < n; i++)
		if (!ctxs[i])
			return VGSDF_E_ARG;
	vgsdf_ctx *c0 = ctxs[0];
	std::vector<int> devs((size_t)n);
	bool distinct = true;
	for (int i = 0; i < n; i++) {
		devs[(size_t)i] = ctxs[i]->device;
		for (int j = 0; j < i; j++)
			distinct = distinct && ctxs[j]->device != ctxs[i]->device;
	}
	uint64_t host_sum[3] = {0, 0, 0};
	for (int i = 0; i < n; i++)
		for (int k = 0; k < 3; k++)
			host_sum[k] += ctxs[i]->counters[k];
	// (test switch: take the RCCL branch although contexts share a device — RCCL refuses the communicator, which is how a
	// one-GPU box exercises the fallback)
	if (const char *e = std::getenv("VGSDF_TEST_ASSUME_DISTINCT"))
		distinct = distinct || e[0] == '1';
	if (!distinct) { // lanes sharing a device: no communicator possible (see above)
		if (strict) {
			c0->err = "vgsdf_reduce_counters_rccl: two contexts share a device (RCCL refuses two ranks on one device)";
			return VGSDF_E_ARG;
		}
		c0->reduce_path = "host: contexts share a device";
	} else if (n == 1 && !strict) {
		c0->reduce_path = "host: one context";
	} else {
		const int rc = reduce_over_rccl(ctxs, n, devs, host_sum);
		if (rc == VGSDF_OK) {
			c0->reduce_path = "rccl";
		} else if (strict) {
			return rc;
		} else {
			// The collective carries 24 bytes the host already holds; losing a finished render to it would be absurd.  The
			// sum is taken on the host and the reason kept, loudly: vgsdf_reduce_path() / bench.py `collectives_fallback`.
			c0->reduce_path = "host: RCCL fallback: " + c0->err;
			std::fprintf(stderr, "[vgsdf] %s -- run counters summed on the host\n", c0->err.c_str());
		}
	}
	std::memcpy(counters, host_sum, sizeof host_sum);
	return VGSDF_OK;
}
} // namespace

int vgsdf_reduce_counters(vgsdf_ctx **ctxs, int n, uint64_t counters[3]) { return reduce_counters(ctxs, n, counters, false); }

int vgsdf_reduce_counters_rccl(vgsdf_ctx **ctxs, int n, uint64_t counters[3]) { return reduce_counters(ctxs, n, counters, true); }

const char *vgsdf_reduce_path(const vgsdf_ctx *ctx) { return ctx ? ctx->reduce_path.c_str() : ""; }

} // extern "C"
