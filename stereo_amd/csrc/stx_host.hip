// Host-pointer data path of libstereo_hip.so: how the caller's (pageable) buffers reach the GPU
// and come back.  This is what the C++ drop-in delivers, so it is built like the kernels are:
//   - the caller's rows are copied into pinned staging memory by a few worker threads (one
//     core moves ~10 GB/s; PCIe Gen5 moves ~55 GB/s each way),
//   - uploads, kernels and downloads run on three streams and are cut into chunks (row bands of
//     one pair, groups of pairs of a batch), so that chunk c+1 is being staged and uploaded while
//     chunk c is computed and chunk c-1 is on its way back,
//   - device images are dense (pitch == width): every DMA is one flat copy.
// ctx->host_pipeline == 0 (stx_set_host_pipeline, or STX_HOST_PIPELINE=0 in the environment)
// selects plain hipMemcpyAsync calls on the caller's buffers instead, for A/B measurements.
#include <atomic>
#if defined(__x86_64__)
#include <emmintrin.h>
#endif
#include <condition_variable>
#include <mutex>
#include <new>
#include <stdlib.h>
#include <string.h>
#include <thread>

#include "stx_internal.h"

// ---------------------------------------------------------------- workers --

struct stx_host_pool {
    static constexpr int MAX_WORKERS = 15;
    int n = 0;  // worker threads (the calling thread is one more)
    std::thread th[MAX_WORKERS];
    std::atomic<unsigned> gen{0};   // job generation: bumped to publish a job
    std::atomic<int> pending{0};    // workers that have not finished the current job
    std::atomic<bool> stop{false};
    std::atomic<int> sleepers{0};
    std::mutex mu;
    std::condition_variable cv;
    // the job: a 2-D byte copy cut into n + 1 row slices
    const uint8_t *src = nullptr;
    uint8_t *dst = nullptr;
    size_t spitch = 0, dpitch = 0, wbytes = 0, rows = 0;
};

// memcpy for the large staging copies: non-temporal 16-byte stores (SSE2, baseline x86-64), a
// whole cache line per iteration.  An ordinary store first reads the destination line into the
// cache (read for ownership), so a plain copy moves three bytes over the memory bus per byte
// copied; the staging buffers are written once and then read by the DMA engines (uploads) or
// the caller much later (downloads), never by this core again.  STX_HOST_NT_COPY=0 in the
// environment selects plain memcpy (A/B measurements).
#if defined(__x86_64__)
static bool nt_copy_on()
{
    static const int on = [] {
        const char *e = getenv("STX_HOST_NT_COPY");
        return (e && e[0] == '0') ? 0 : 1;
    }();
    return on != 0;
}
static void copy_bytes(uint8_t *dst, const uint8_t *src, size_t n)
{
    if (n < ((size_t)32 << 10) || !nt_copy_on()) {
        memcpy(dst, src, n);
        return;
    }
    const size_t head = (64 - ((uintptr_t)dst & 63)) & 63;  // up to the destination's next cache line
    memcpy(dst, src, head);
    dst += head; src += head; n -= head;
    const size_t lines = n / 64;
    for (size_t i = 0; i < lines; ++i) {
        const __m128i a = _mm_loadu_si128((const __m128i *)(src) + 0), b = _mm_loadu_si128((const __m128i *)(src) + 1),
                      c = _mm_loadu_si128((const __m128i *)(src) + 2), d = _mm_loadu_si128((const __m128i *)(src) + 3);
        _mm_stream_si128((__m128i *)(dst) + 0, a);
        _mm_stream_si128((__m128i *)(dst) + 1, b);
        _mm_stream_si128((__m128i *)(dst) + 2, c);
        _mm_stream_si128((__m128i *)(dst) + 3, d);
        src += 64; dst += 64;
    }
    _mm_sfence();  // the streamed lines are globally visible before the DMA / the pool's release
    memcpy(dst, src, n - lines * 64);
}
#else
static void copy_bytes(uint8_t *dst, const uint8_t *src, size_t n) { memcpy(dst, src, n); }
#endif

static void copy_rows(uint8_t *dst, size_t dpitch, const uint8_t *src, size_t spitch, size_t wbytes, size_t r0,
                      size_t r1)
{
    if (r1 <= r0) return;
    if (dpitch == wbytes && spitch == wbytes) {
        copy_bytes(dst + r0 * wbytes, src + r0 * wbytes, (r1 - r0) * wbytes);
        return;
    }
    for (size_t r = r0; r < r1; ++r) copy_bytes(dst + r * dpitch, src + r * spitch, wbytes);
}

static inline void cpu_relax()
{
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
}

static void pool_worker(stx_host_pool *p, int index)
{
    unsigned seen = 0;
    for (;;) {
        // spin for a while (calls in a loop arrive back to back), then sleep
        int spins = 0;
        while (p->gen.load(std::memory_order_acquire) == seen && !p->stop.load(std::memory_order_relaxed)) {
            if (++spins < 20000) {
                cpu_relax();
                continue;
            }
            std::unique_lock<std::mutex> lk(p->mu);
            p->sleepers.fetch_add(1);
            p->cv.wait(lk, [&] { return p->gen.load(std::memory_order_acquire) != seen || p->stop.load(); });
            p->sleepers.fetch_sub(1);
        }
        if (p->stop.load()) return;
        seen = p->gen.load(std::memory_order_acquire);
        const size_t parts = (size_t)p->n + 1, k = (size_t)index + 1;
        copy_rows(p->dst, p->dpitch, p->src, p->spitch, p->wbytes, p->rows * k / parts, p->rows * (k + 1) / parts);
        p->pending.fetch_sub(1, std::memory_order_release);
    }
}

// dst/src 2-D copy on the pool (blocking).  Small copies stay on the calling thread.
static void pool_copy2d(stx_ctx *ctx, void *dst, size_t dpitch, const void *src, size_t spitch, size_t wbytes,
                        size_t rows)
{
    stx_host_pool *p = ctx->pool;
    if (!p || p->n == 0 || wbytes * rows < (size_t)256 * 1024 || rows < (size_t)p->n + 1) {
        copy_rows((uint8_t *)dst, dpitch, (const uint8_t *)src, spitch, wbytes, 0, rows);
        return;
    }
    p->src = (const uint8_t *)src;
    p->dst = (uint8_t *)dst;
    p->spitch = spitch;
    p->dpitch = dpitch;
    p->wbytes = wbytes;
    p->rows = rows;
    p->pending.store(p->n, std::memory_order_relaxed);
    p->gen.fetch_add(1, std::memory_order_release);
    if (p->sleepers.load() > 0) {
        std::lock_guard<std::mutex> lk(p->mu);
        p->cv.notify_all();
    }
    copy_rows(p->dst, dpitch, p->src, spitch, wbytes, 0, rows / ((size_t)p->n + 1));
    while (p->pending.load(std::memory_order_acquire) != 0) cpu_relax();
}

static int pool_start(stx_ctx *ctx)
{
    if (ctx->pool) return STX_OK;
    int want = ctx->host_threads;
    if (want <= 0) {
        const char *e = getenv("STX_HOST_THREADS");
        want = e ? atoi(e) : 0;
    }
    if (want <= 0) {
        // hardware_concurrency() reports the whole machine, also inside a container with a
        // small CPU share: stay modest
        const unsigned hc = std::thread::hardware_concurrency();
        want = hc >= 16 ? 8 : hc >= 4 ? (int)hc / 2 : 1;
    }
    if (want > stx_host_pool::MAX_WORKERS + 1) want = stx_host_pool::MAX_WORKERS + 1;
    stx_host_pool *p = new (std::nothrow) stx_host_pool();
    if (!p) return STX_ERR_HOST_OOM;
    for (int i = 0; i < want - 1; ++i) {
        try {
            p->th[i] = std::thread(pool_worker, p, i);
        } catch (...) {  // no thread to be had: run with the ones we have
            break;
        }
        p->n = i + 1;
    }
    ctx->pool = p;
    return STX_OK;
}

static void pool_stop(stx_ctx *ctx)
{
    stx_host_pool *p = ctx->pool;
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->stop.store(true);
        p->cv.notify_all();
    }
    for (int i = 0; i < p->n; ++i)
        if (p->th[i].joinable()) p->th[i].join();
    delete p;
    ctx->pool = nullptr;
}

// -------------------------------------------------------------- resources --

void stx_host_release(stx_ctx *ctx)
{
    pool_stop(ctx);
    for (int i = 0; i < 2; ++i) {
        if (ctx->pin_ptr[i]) (void)hipHostFree(ctx->pin_ptr[i]);
        ctx->pin_ptr[i] = nullptr;
        ctx->pin_bytes[i] = 0;
    }
    for (int i = 0; i < STX_HOST_MAX_CHUNKS; ++i) {
        if (ctx->ev_in[i]) (void)hipEventDestroy(ctx->ev_in[i]);
        if (ctx->ev_k[i]) (void)hipEventDestroy(ctx->ev_k[i]);
        if (ctx->ev_out[i]) (void)hipEventDestroy(ctx->ev_out[i]);
        ctx->ev_in[i] = ctx->ev_k[i] = ctx->ev_out[i] = nullptr;
    }
    if (ctx->s_in) (void)hipStreamDestroy(ctx->s_in);
    if (ctx->s_out) (void)hipStreamDestroy(ctx->s_out);
    ctx->s_in = ctx->s_out = nullptr;
}

bool stx_host_pipeline_on(stx_ctx *ctx)
{
    if (ctx->host_pipeline < 0) {
        const char *e = getenv("STX_HOST_PIPELINE");
        ctx->host_pipeline = (e && e[0] == '0') ? 0 : 1;
    }
    return ctx->host_pipeline != 0;
}

// STX_HOST_ZERO_COPY=0 in the environment: kernels never write host memory directly (A/B measurements)
bool stx_host_zero_copy_on()
{
    static const int on = [] {
        const char *e = getenv("STX_HOST_ZERO_COPY");
        return (e && e[0] == '0') ? 0 : 1;
    }();
    return on != 0;
}

// Bytes of images and maps that one group of a batch holds on the device and in staging
// (STX_HOST_GROUP_MB: tests cut a small batch into several groups with it).
static size_t batch_group_bytes()
{
    const char *e = getenv("STX_HOST_GROUP_MB");
    const size_t mb = e && *e ? (size_t)strtoull(e, nullptr, 10) : 0;
    return (mb ? mb : 256) << 20;
}

// pinned staging of at least in_bytes / out_bytes, the two copy streams, events, workers
int stx_host_prepare(stx_ctx *ctx, size_t in_bytes, size_t out_bytes)
{
    const size_t want[2] = {in_bytes, out_bytes};
    for (int i = 0; i < 2; ++i) {
        if (ctx->pin_bytes[i] >= want[i]) continue;
        // the old buffer may still be the source / target of a copy in flight
        STX_HIP(hipDeviceSynchronize());
        if (ctx->pin_ptr[i]) (void)hipHostFree(ctx->pin_ptr[i]);
        ctx->pin_ptr[i] = nullptr;
        ctx->pin_bytes[i] = 0;
        const size_t bytes = (want[i] + (want[i] >> 2) + 65535) & ~(size_t)65535;  // 25 % headroom
        // STX_PIN_LIMIT_MB caps what one staging buffer may page-lock (hosts with a small memlock
        // allowance): a request above it is answered as an exhausted host would answer it, and the
        // callers go on with plain copies
        const char *cap = getenv("STX_PIN_LIMIT_MB");
        if (cap && *cap && bytes > ((size_t)strtoull(cap, nullptr, 10) << 20)) return (int)hipErrorOutOfMemory;
        void *p = nullptr;
        STX_HIP(hipHostMalloc(&p, bytes, hipHostMallocDefault));
        ctx->pin_ptr[i] = p;
        ctx->pin_bytes[i] = bytes;
    }
    if (!ctx->s_in) STX_HIP(hipStreamCreateWithFlags(&ctx->s_in, hipStreamNonBlocking));
    if (!ctx->s_out) STX_HIP(hipStreamCreateWithFlags(&ctx->s_out, hipStreamNonBlocking));
    for (int i = 0; i < STX_HOST_MAX_CHUNKS; ++i) {
        if (!ctx->ev_in[i]) STX_HIP(hipEventCreateWithFlags(&ctx->ev_in[i], hipEventDisableTiming));
        if (!ctx->ev_k[i]) STX_HIP(hipEventCreateWithFlags(&ctx->ev_k[i], hipEventDisableTiming));
        if (!ctx->ev_out[i]) STX_HIP(hipEventCreateWithFlags(&ctx->ev_out[i], hipEventDisableTiming));
    }
    return pool_start(ctx);
}

int stx_host_chunks(stx_ctx *ctx, size_t bytes, size_t units)
{
    int c = ctx->host_chunks;
    if (c <= 0) {
        const char *e = getenv("STX_HOST_CHUNKS");
        c = e ? atoi(e) : 0;
    }
    if (c <= 0) c = bytes < ((size_t)1 << 20) ? 1 : bytes < ((size_t)6 << 20) ? 2 : 4;
    if (c > STX_HOST_MAX_CHUNKS) c = STX_HOST_MAX_CHUNKS;
    if ((size_t)c > units) c = (int)(units > 0 ? units : 1);
    return c;
}

// A caller whose buffers are already page-locked (hipHostMalloc, hipHostRegister, a pinned
// cv::MatAllocator, torch pin_memory) needs no staging copy at all: the DMA engines read and write
// its memory directly.  [p, p + bytes) must lie inside ONE page-locked allocation the runtime knows:
// both ends must be host memory and the runtime's address range around the first byte must hold the
// last one too (two separate registrations that happen to be adjacent do not qualify; anything the
// runtime cannot vouch for goes through the staging buffers, which is always correct).
bool stx_host_is_pinned(const void *p, size_t bytes)
{
    if (!p || bytes == 0) return false;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess || at.type != hipMemoryTypeHost) {
        (void)hipGetLastError();
        return false;
    }
    hipPointerAttribute_t end;
    if (hipPointerGetAttributes(&end, (const char *)p + bytes - 1) != hipSuccess || end.type != hipMemoryTypeHost) {
        (void)hipGetLastError();
        return false;
    }
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) != hipSuccess || !base) {
        (void)hipGetLastError();
        return false;
    }
    const char *lo = (const char *)base, *q = (const char *)p;
    return q >= lo && q + bytes <= lo + size;
}

// The address through which kernels on the CURRENT device may store into the page-locked host range
// [p, p + bytes), or nullptr.  A registered range's device-visible address need not equal the host
// address and need not exist on this device at all (hipHostRegister without the mapped / portable
// flags, memory pinned for another GPU): the zero-copy paths use this address, never the raw host
// pointer, and fall back to the staging buffer when there is none.
void *stx_host_kernel_ptr(const void *p, size_t bytes)
{
    if (!stx_host_is_pinned(p, bytes)) return nullptr;
    void *dev = nullptr;
    if (hipHostGetDevicePointer(&dev, const_cast<void *>(p), 0) != hipSuccess || !dev) {
        (void)hipGetLastError();
        return nullptr;
    }
    // the last byte must map where the first one says it does
    void *dev_end = nullptr;
    if (hipHostGetDevicePointer(&dev_end, (char *)const_cast<void *>(p) + bytes - 1, 0) != hipSuccess ||
        (char *)dev_end - (char *)dev != (ptrdiff_t)(bytes - 1)) {
        (void)hipGetLastError();
        return nullptr;
    }
    return dev;
}

// user rows -> pinned (workers) -> device (s_in), then ev is recorded on s_in.
// pin == nullptr: the user's rows are dense and page-locked, the DMA reads them directly.
int stx_host_upload(stx_ctx *ctx, void *d_dst, void *pin, const void *src, size_t spitch, size_t wbytes,
                    size_t rows, hipEvent_t ev)
{
    if (rows > 0 && wbytes > 0) {
        if (pin) pool_copy2d(ctx, pin, wbytes, src, spitch, wbytes, rows);
        STX_HIP(hipMemcpyAsync(d_dst, pin ? pin : src, wbytes * rows, hipMemcpyHostToDevice, ctx->s_in));
    }
    if (ev) STX_HIP(hipEventRecord(ev, ctx->s_in));
    return STX_OK;
}

// device -> pinned on s_out once `after` (recorded on the compute stream) has passed; ev marks the end
int stx_host_download_begin(stx_ctx *ctx, void *pin, const void *d_src, size_t bytes, hipEvent_t after,
                            hipEvent_t ev)
{
    if (after) STX_HIP(hipStreamWaitEvent(ctx->s_out, after, 0));
    if (bytes > 0) STX_HIP(hipMemcpyAsync(pin, d_src, bytes, hipMemcpyDeviceToHost, ctx->s_out));
    STX_HIP(hipEventRecord(ev, ctx->s_out));
    return STX_OK;
}

// wait for ev, then pinned -> user rows (workers).  pin == dst: the download went straight into
// the user's (dense, page-locked) rows.
int stx_host_download_end(stx_ctx *ctx, void *dst, size_t dpitch, const void *pin, size_t wbytes, size_t rows,
                          hipEvent_t ev)
{
    STX_HIP(hipEventSynchronize(ev));
    if (rows > 0 && wbytes > 0 && pin != dst) pool_copy2d(ctx, dst, dpitch, pin, wbytes, wbytes, rows);
    return STX_OK;
}

// The copy streams start behind whatever the caller has already enqueued on the compute stream
// (the device buffers may still be in use by an earlier asynchronous call of this context).
static int fence_copy_streams(stx_ctx *ctx)
{
    STX_HIP(hipEventRecord(ctx->ev_k[0], ctx->stream));
    STX_HIP(hipStreamWaitEvent(ctx->s_in, ctx->ev_k[0], 0));
    STX_HIP(hipStreamWaitEvent(ctx->s_out, ctx->ev_k[0], 0));
    return STX_OK;
}

// ------------------------------------------------------------- D (host) ----

// Disparity of output rows [row0, row0 + rows) of n_pairs pairs held in host memory; `out`
// points at output row row0 of pair 0.  One pair: chunked over row bands (a band's scan starts
// as soon as its rows and win/2 halo rows are resident).  A batch: chunked over groups of pairs.
static int disparity_host_pipeline(stx_ctx *ctx, const uint8_t *img0, const uint8_t *img1, int n_pairs, int w, int h,
                                   size_t stride, size_t pair_stride, int row0, int rows, int win, int max_disp,
                                   int cost_mode, void *out, size_t out_stride, size_t out_pair_stride, int out_bytes)
{
    if (n_pairs <= 0 || w <= 0 || rows <= 0) return STX_OK;
    const int r = max_disp == 0 ? 0 : win / 2;
    const int b0 = row0 - r > 0 ? row0 - r : 0;                // first input row needed
    const int b1 = row0 + rows + r < h ? row0 + rows + r : h;  // one past the last
    const int bh = b1 - b0;
    const size_t p = (size_t)w, op = p * (size_t)out_bytes;
    const size_t img = p * (size_t)bh, oimg = op * (size_t)rows;
    uint8_t *o = static_cast<uint8_t *>(out);
    int err;

    if (!stx_host_pipeline_on(ctx)) {
        hipStream_t s = ctx->stream;
        // groups of pairs that fit a modest staging area
        const size_t per_pair = 2 * img + oimg;
        int group = (int)(batch_group_bytes() / (per_pair ? per_pair : 1));
        group = group < 1 ? 1 : group > n_pairs ? n_pairs : group;
        uint8_t *d0 = (uint8_t *)stx_workspace(ctx, STX_SLOT_IMG0, img * group, &err);
        if (!d0) return err;
        uint8_t *d1 = (uint8_t *)stx_workspace(ctx, STX_SLOT_IMG1, img * group, &err);
        if (!d1) return err;
        uint8_t *dout = (uint8_t *)stx_workspace(ctx, STX_SLOT_OUT, oimg * group, &err);
        if (!dout) return err;
        for (int g0 = 0; g0 < n_pairs; g0 += group) {
            const int g = n_pairs - g0 < group ? n_pairs - g0 : group;
            for (int k = 0; k < g; ++k) {
                const size_t off = (size_t)(g0 + k) * pair_stride + (size_t)b0 * stride;
                if (stride == p) {
                    STX_HIP(hipMemcpyAsync(d0 + k * img, img0 + off, img, hipMemcpyHostToDevice, s));
                    STX_HIP(hipMemcpyAsync(d1 + k * img, img1 + off, img, hipMemcpyHostToDevice, s));
                } else {
                    STX_HIP(hipMemcpy2DAsync(d0 + k * img, p, img0 + off, stride, p, bh, hipMemcpyHostToDevice, s));
                    STX_HIP(hipMemcpy2DAsync(d1 + k * img, p, img1 + off, stride, p, bh, hipMemcpyHostToDevice, s));
                }
            }
            int rc = stx_launch_disparity(ctx, d0, d1, g, w, bh, p, img, row0 - b0, rows, win, max_disp,
                                          cost_mode, dout, op, oimg, out_bytes);
            if (rc) return rc;
            for (int k = 0; k < g; ++k) {
                uint8_t *dst = o + (size_t)(g0 + k) * out_pair_stride;
                if (out_stride == op)
                    STX_HIP(hipMemcpyAsync(dst, dout + k * oimg, oimg, hipMemcpyDeviceToHost, s));
                else
                    STX_HIP(hipMemcpy2DAsync(dst, out_stride, dout + k * oimg, op, op, rows, hipMemcpyDeviceToHost, s));
            }
            STX_HIP(hipStreamSynchronize(s));
        }
        return STX_OK;
    }

    if (n_pairs == 1) {
        // ---- one pair: row-band chunks ------------------------------------------
        const int nc = stx_host_chunks(ctx, 2 * img + oimg, (size_t)(rows + 63) / 64);
        int rc = stx_host_prepare(ctx, 2 * img, oimg);
        if (rc) return rc;
        uint8_t *d0 = (uint8_t *)stx_workspace(ctx, STX_SLOT_IMG0, img, &err);
        if (!d0) return err;
        uint8_t *d1 = (uint8_t *)stx_workspace(ctx, STX_SLOT_IMG1, img, &err);
        if (!d1) return err;
        uint8_t *dout = (uint8_t *)stx_workspace(ctx, STX_SLOT_OUT, oimg, &err);
        if (!dout) return err;
        rc = fence_copy_streams(ctx);
        if (rc) return rc;
        uint8_t *pin_in = (uint8_t *)ctx->pin_ptr[0], *pin_out = (uint8_t *)ctx->pin_ptr[1];
        // page-locked, dense caller buffers are read / written by the DMA engines directly
        const bool in0_direct = stride == p && stx_host_is_pinned(img0 + (size_t)b0 * stride, img);
        const bool in1_direct = stride == p && stx_host_is_pinned(img1 + (size_t)b0 * stride, img);
        const bool out_direct = out_stride == op && stx_host_is_pinned(o, oimg);
        // chunk c uploads input rows [u(c), u(c+1)) of the band and computes the output rows
        // whose windows end inside what has been uploaded so far
        int up_prev = 0, out_prev = 0;  // band-relative input rows uploaded / output rows computed
        int out_lo[STX_HOST_MAX_CHUNKS], out_n[STX_HOST_MAX_CHUNKS];
        // A small map (up to 4 MB: 1080p, not config 2's 5.7 MB) is stored by the scan straight into
        // page-locked host memory, as the warp's destination is: no download to set up and wait for
        // (741x500: 0.165 -> 0.134 ms, 1080p: 0.29 -> 0.27 ms; at 5.7 MB the chunked download is as fast).
        const bool zc = oimg <= ((size_t)4 << 20) && stx_host_zero_copy_on();
        // (kernels store through the device-visible address of the caller's page-locked rows, if this
        // device has one; else through the staging buffer, copied out afterwards as for pageable rows)
        uint8_t *o_dev = zc && out_direct ? (uint8_t *)stx_host_kernel_ptr(o, oimg) : nullptr;
        const bool out_zc_direct = o_dev != nullptr;
        uint8_t *ktarget = zc ? (out_zc_direct ? o_dev : pin_out) : dout;
        for (int c = 0; c < nc; ++c) {
            const int up_next = c == nc - 1 ? bh : (int)((long long)bh * (c + 1) / nc);
            const size_t ro = (size_t)up_prev * p, rn = (size_t)(up_next - up_prev);
            const size_t uoff = (size_t)(b0 + up_prev) * stride;
            rc = stx_host_upload(ctx, d0 + ro, in0_direct ? nullptr : pin_in + ro, img0 + uoff, stride, p, rn, nullptr);
            if (rc) return rc;
            rc = stx_host_upload(ctx, d1 + ro, in1_direct ? nullptr : pin_in + img + ro, img1 + uoff, stride, p, rn,
                                 ctx->ev_in[c]);
            if (rc) return rc;
            up_prev = up_next;
            // output rows (relative to row0) computable now: image row y needs input rows up to
            // min(y + r, h - 1), i.e. band-relative y + r - b0 < up_prev (or the band is complete)
            int out_next = c == nc - 1 ? rows : b0 + up_prev - r - row0;
            if (out_next > rows) out_next = rows;
            if (out_next < out_prev) out_next = out_prev;
            out_lo[c] = out_prev;
            out_n[c] = out_next - out_prev;
            STX_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_in[c], 0));
            if (out_n[c] > 0) {
                rc = stx_launch_disparity(ctx, d0, d1, 1, w, bh, p, 0, row0 - b0 + out_prev, out_n[c], win,
                                          max_disp, cost_mode, ktarget + (size_t)out_prev * op, op, 0, out_bytes);
                if (rc) return rc;
            }
            STX_HIP(hipEventRecord(ctx->ev_k[c], ctx->stream));
            if (!zc) {
                rc = stx_host_download_begin(ctx, out_direct ? o + (size_t)out_prev * op : pin_out + (size_t)out_prev * op,
                                             dout + (size_t)out_prev * op, (size_t)out_n[c] * op, ctx->ev_k[c],
                                             ctx->ev_out[c]);
                if (rc) return rc;
            }
            out_prev = out_next;
        }
        if (zc) {
            for (int c = 0; c < nc; ++c) {
                STX_HIP(hipEventSynchronize(ctx->ev_k[c]));
                if (!out_zc_direct && out_n[c] > 0)
                    pool_copy2d(ctx, o + (size_t)out_lo[c] * out_stride, out_stride, pin_out + (size_t)out_lo[c] * op, op, op,
                                (size_t)out_n[c]);
            }
            return STX_OK;
        }
        for (int c = 0; c < nc; ++c) {
            rc = stx_host_download_end(ctx, o + (size_t)out_lo[c] * out_stride, out_stride,
                                       out_direct ? o + (size_t)out_lo[c] * op : pin_out + (size_t)out_lo[c] * op, op,
                                       (size_t)out_n[c], ctx->ev_out[c]);
            if (rc) return rc;
        }
        return STX_OK;
    }

    // ---- a batch: groups of pairs, each group cut into chunks of pairs -------------
    // (256 MB of images and maps per group: 41 pairs of 1080p, 3 ms of scanning between two fences;
    // the device workspace and the page-locked staging of every context of an stx_multi are this big)
    const size_t per_pair = 2 * img + oimg;
    int group = (int)(batch_group_bytes() / (per_pair ? per_pair : 1));
    group = group < 1 ? 1 : group > n_pairs ? n_pairs : group;
    // page-locked, dense caller arrays (checked over the whole batch) are read / written by the DMA
    // engines directly and need no staging: the staging copies are what bounds a pageable batch
    const size_t span = (size_t)(n_pairs - 1) * pair_stride + img;
    const bool in_direct = stride == p && stx_host_is_pinned(img0 + (size_t)b0 * stride, span) &&
                           stx_host_is_pinned(img1 + (size_t)b0 * stride, span);
    const bool out_direct = out_stride == op && out_pair_stride == oimg && stx_host_is_pinned(o, (size_t)n_pairs * oimg);
    int rc = stx_host_prepare(ctx, in_direct ? 0 : 2 * img * group, out_direct ? 0 : oimg * group);
    if (rc) return rc;
    uint8_t *d0 = (uint8_t *)stx_workspace(ctx, STX_SLOT_IMG0, img * group, &err);
    if (!d0) return err;
    uint8_t *d1 = (uint8_t *)stx_workspace(ctx, STX_SLOT_IMG1, img * group, &err);
    if (!d1) return err;
    uint8_t *dout = (uint8_t *)stx_workspace(ctx, STX_SLOT_OUT, oimg * group, &err);
    if (!dout) return err;
    uint8_t *pin_in = (uint8_t *)ctx->pin_ptr[0], *pin_out = (uint8_t *)ctx->pin_ptr[1];
    for (int g0 = 0; g0 < n_pairs; g0 += group) {
        const int g = n_pairs - g0 < group ? n_pairs - g0 : group;
        const int nc = stx_host_chunks(ctx, per_pair * g, (size_t)g);
        rc = fence_copy_streams(ctx);
        if (rc) return rc;
        int lo[STX_HOST_MAX_CHUNKS], cnt[STX_HOST_MAX_CHUNKS];
        for (int c = 0; c < nc; ++c) {
            lo[c] = (int)((long long)g * c / nc);
            cnt[c] = (int)((long long)g * (c + 1) / nc) - lo[c];
            // page-locked and dense: one DMA per image array and chunk, not per pair (a copy costs ~10 us
            // before its first byte)
            const size_t first = (size_t)(g0 + lo[c]) * pair_stride + (size_t)b0 * stride;
            const size_t cbytes = (size_t)cnt[c] * img;
            if (in_direct && (pair_stride == img || cnt[c] == 1)) {
                STX_HIP(hipMemcpyAsync(d0 + lo[c] * img, img0 + first, cbytes, hipMemcpyHostToDevice, ctx->s_in));
                STX_HIP(hipMemcpyAsync(d1 + lo[c] * img, img1 + first, cbytes, hipMemcpyHostToDevice, ctx->s_in));
            } else if (in_direct) {
                for (int k = lo[c]; k < lo[c] + cnt[c]; ++k) {
                    const size_t off = (size_t)(g0 + k) * pair_stride + (size_t)b0 * stride;
                    STX_HIP(hipMemcpyAsync(d0 + k * img, img0 + off, img, hipMemcpyHostToDevice, ctx->s_in));
                    STX_HIP(hipMemcpyAsync(d1 + k * img, img1 + off, img, hipMemcpyHostToDevice, ctx->s_in));
                }
            } else {  // pageable: pair by pair, so that the staging copy of one image runs behind the DMA of the last
                for (int k = lo[c]; k < lo[c] + cnt[c]; ++k) {
                    const size_t off = (size_t)(g0 + k) * pair_stride + (size_t)b0 * stride;
                    rc = stx_host_upload(ctx, d0 + k * img, pin_in + (size_t)k * img, img0 + off, stride, p, bh, nullptr);
                    if (rc) return rc;
                    rc = stx_host_upload(ctx, d1 + k * img, pin_in + ((size_t)group + k) * img, img1 + off, stride, p, bh,
                                         nullptr);
                    if (rc) return rc;
                }
            }
            STX_HIP(hipEventRecord(ctx->ev_in[c], ctx->s_in));
            STX_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_in[c], 0));
            rc = stx_launch_disparity(ctx, d0 + lo[c] * img, d1 + lo[c] * img, cnt[c], w, bh, p, img, row0 - b0, rows,
                                      win, max_disp, cost_mode, dout + lo[c] * oimg, op, oimg, out_bytes);
            if (rc) return rc;
            STX_HIP(hipEventRecord(ctx->ev_k[c], ctx->stream));
            rc = stx_host_download_begin(ctx, out_direct ? o + (size_t)(g0 + lo[c]) * oimg : pin_out + lo[c] * oimg,
                                         dout + lo[c] * oimg, cnt[c] * oimg, ctx->ev_k[c], ctx->ev_out[c]);
            if (rc) return rc;
        }
        for (int c = 0; c < nc; ++c) {
            STX_HIP(hipEventSynchronize(ctx->ev_out[c]));
            if (out_direct) continue;
            for (int k = lo[c]; k < lo[c] + cnt[c]; ++k)
                pool_copy2d(ctx, o + (size_t)(g0 + k) * out_pair_stride, out_stride, pin_out + (size_t)k * oimg, op, op,
                            (size_t)rows);
        }
    }
    return STX_OK;
}

// ------------------------------------------------------------- R (host) ----

// One image through the warp (RectifyImage's loop): the source goes up in chunks, the kernel
// runs once, destination row bands come back chunk by chunk.
static int rectify_host_pipeline(stx_ctx *ctx, const uint8_t *src, int src_w, int src_h, size_t src_stride, uint8_t *dst,
                                 int dst_w, int dst_h, size_t dst_stride, const float Hinv[9], int flags, uint8_t *d_keep)
{
    int err;
    const size_t sp = (size_t)(src_w > 0 ? src_w : 1), dp = (size_t)dst_w;
    const size_t sbytes = sp * (size_t)(src_h > 0 ? src_h : 1), dbytes = dp * (size_t)dst_h;
    uint8_t *dsrc = (uint8_t *)stx_workspace(ctx, STX_SLOT_IMG0, sbytes, &err);
    if (!dsrc) return err;
    uint8_t *ddst = d_keep ? d_keep : (uint8_t *)stx_workspace(ctx, STX_SLOT_IMG1, dbytes, &err);
    if (!ddst) return err;
    const bool have_src = src_w > 0 && src_h > 0;
    const bool preserve = flags == STX_RECT_PRESERVE_DST;

    if (!stx_host_pipeline_on(ctx)) {
        hipStream_t s = ctx->stream;
        if (have_src) {
            if (src_stride == sp)
                STX_HIP(hipMemcpyAsync(dsrc, src, sbytes, hipMemcpyHostToDevice, s));
            else
                STX_HIP(hipMemcpy2DAsync(dsrc, sp, src, src_stride, sp, src_h, hipMemcpyHostToDevice, s));
        }
        if (preserve) {
            if (dst_stride == dp)
                STX_HIP(hipMemcpyAsync(ddst, dst, dbytes, hipMemcpyHostToDevice, s));
            else
                STX_HIP(hipMemcpy2DAsync(ddst, dp, dst, dst_stride, dp, dst_h, hipMemcpyHostToDevice, s));
        }
        int rc = stx_rectify_u8_dev(ctx, dsrc, src_w, src_h, sp, 0, ddst, dst_w, dst_h, dp, 0, Hinv, 1, flags);
        if (rc) return rc;
        if (dst_stride == dp)
            STX_HIP(hipMemcpyAsync(dst, ddst, dbytes, hipMemcpyDeviceToHost, s));
        else
            STX_HIP(hipMemcpy2DAsync(dst, dst_stride, ddst, dp, dp, dst_h, hipMemcpyDeviceToHost, s));
        STX_HIP(hipStreamSynchronize(s));
        return STX_OK;
    }

    int rc = stx_host_prepare(ctx, sbytes + (preserve ? dbytes : 0), dbytes);
    if (rc) return rc;
    rc = fence_copy_streams(ctx);
    if (rc) return rc;
    uint8_t *pin_in = (uint8_t *)ctx->pin_ptr[0], *pin_out = (uint8_t *)ctx->pin_ptr[1];
    const bool src_direct = have_src && src_stride == sp && stx_host_is_pinned(src, sbytes);
    const bool dst_direct = dst_stride == dp && stx_host_is_pinned(dst, dbytes);
    // uploads: the source (and, to keep missed pixels, the destination) in row chunks so the
    // staging copy of chunk c+1 runs while chunk c is on the bus
    const int ncu = stx_host_chunks(ctx, sbytes, (size_t)(src_h > 0 ? src_h : 1));
    if (have_src)
        for (int c = 0; c < ncu; ++c) {
            const size_t r0 = (size_t)src_h * c / ncu, r1 = (size_t)src_h * (c + 1) / ncu;
            rc = stx_host_upload(ctx, dsrc + r0 * sp, src_direct ? nullptr : pin_in + r0 * sp, src + r0 * src_stride,
                                 src_stride, sp, r1 - r0, nullptr);
            if (rc) return rc;
        }
    if (preserve) {
        rc = stx_host_upload(ctx, ddst, dst_direct ? nullptr : pin_in + sbytes, dst, dst_stride, dp, (size_t)dst_h, nullptr);
        if (rc) return rc;
    }
    STX_HIP(hipEventRecord(ctx->ev_in[0], ctx->s_in));
    STX_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_in[0], 0));
    // A small destination that the kernel never reads (declared zero, not kept on the device) is
    // written by the kernel straight into page-locked host memory -- the staging buffer, or the
    // caller's own rows when those are dense and page-locked: the stores cross the bus while the
    // kernel runs and there is no download to set up and wait for (1080p: 0.19 -> 0.16 ms; letting the
    // kernel READ the source from host memory as well was slower than the upload it saves).
    if (!preserve && !d_keep && dbytes <= ((size_t)4 << 20) && stx_host_zero_copy_on()) {
        // the caller's rows through their device-visible address, if this device has one
        uint8_t *dst_dev = dst_direct ? (uint8_t *)stx_host_kernel_ptr(dst, dbytes) : nullptr;
        uint8_t *target = dst_dev ? dst_dev : pin_out;
        rc = stx_rectify_u8_dev(ctx, dsrc, src_w, src_h, sp, 0, target, dst_w, dst_h, dp, 0, Hinv, 1, flags);
        if (rc) return rc;
        STX_HIP(hipStreamSynchronize(ctx->stream));
        if (!dst_dev) pool_copy2d(ctx, dst, dst_stride, pin_out, dp, dp, (size_t)dst_h);
        return STX_OK;
    }
    // destination row bands: kernel per band, download behind it
    const int ncd = stx_host_chunks(ctx, dbytes, (size_t)(dst_h + 15) / 16);
    int lo[STX_HOST_MAX_CHUNKS], cnt[STX_HOST_MAX_CHUNKS];
    for (int c = 0; c < ncd; ++c) {
        lo[c] = (int)((long long)dst_h * c / ncd);
        cnt[c] = (int)((long long)dst_h * (c + 1) / ncd) - lo[c];
        rc = stx_rectify_rowband_u8_dev(ctx, dsrc, src_w, src_h, sp, ddst + (size_t)lo[c] * dp, dst_w, lo[c], cnt[c], dp,
                                        Hinv, flags);
        if (rc) return rc;
        STX_HIP(hipEventRecord(ctx->ev_k[c], ctx->stream));
        rc = stx_host_download_begin(ctx, dst_direct ? dst + (size_t)lo[c] * dp : pin_out + (size_t)lo[c] * dp,
                                     ddst + (size_t)lo[c] * dp, (size_t)cnt[c] * dp, ctx->ev_k[c], ctx->ev_out[c]);
        if (rc) return rc;
    }
    for (int c = 0; c < ncd; ++c) {
        rc = stx_host_download_end(ctx, dst + (size_t)lo[c] * dst_stride, dst_stride,
                                   dst_direct ? dst + (size_t)lo[c] * dp : pin_out + (size_t)lo[c] * dp, dp,
                                   (size_t)cnt[c], ctx->ev_out[c]);
        if (rc) return rc;
    }
    return STX_OK;
}

// Scan of two images that are already on the device (dense, pitch == w); the map comes back in
// row-band chunks, each band's download behind its kernel.
static int disparity_resident_host_pipeline(stx_ctx *ctx, const uint8_t *d_img0, const uint8_t *d_img1, int w, int h,
                                            int win, int max_disp, int cost_mode, void *out, size_t out_stride,
                                            int out_bytes)
{
    int err;
    const size_t p = (size_t)w, op = p * (size_t)out_bytes, oimg = op * (size_t)h;
    uint8_t *o = static_cast<uint8_t *>(out);
    uint8_t *dout = (uint8_t *)stx_workspace(ctx, STX_SLOT_OUT, oimg, &err);
    if (!dout) return err;
    if (!stx_host_pipeline_on(ctx)) {
        int rc = stx_launch_disparity(ctx, d_img0, d_img1, 1, w, h, p, 0, 0, h, win, max_disp, cost_mode, dout, op, 0,
                                      out_bytes);
        if (rc) return rc;
        if (out_stride == op)
            STX_HIP(hipMemcpyAsync(o, dout, oimg, hipMemcpyDeviceToHost, ctx->stream));
        else
            STX_HIP(hipMemcpy2DAsync(o, out_stride, dout, op, op, h, hipMemcpyDeviceToHost, ctx->stream));
        STX_HIP(hipStreamSynchronize(ctx->stream));
        return STX_OK;
    }
    int rc = stx_host_prepare(ctx, 0, oimg);
    if (rc) return rc;
    rc = fence_copy_streams(ctx);
    if (rc) return rc;
    uint8_t *pin_out = (uint8_t *)ctx->pin_ptr[1];
    const bool o_pinned = out_stride == op && stx_host_is_pinned(o, oimg);
    if (oimg <= ((size_t)4 << 20) && stx_host_zero_copy_on()) {  // a small map: stored straight into pinned memory
        // (the caller's own page-locked rows through their device-visible address, if this device has one)
        uint8_t *o_dev = o_pinned ? (uint8_t *)stx_host_kernel_ptr(o, oimg) : nullptr;
        rc = stx_launch_disparity(ctx, d_img0, d_img1, 1, w, h, p, 0, 0, h, win, max_disp, cost_mode,
                                  o_dev ? o_dev : pin_out, op, 0, out_bytes);
        if (rc) return rc;
        STX_HIP(hipEventRecord(ctx->ev_k[0], ctx->stream));
        return stx_host_download_end(ctx, o, out_stride, o_dev ? o : pin_out, op, (size_t)h, ctx->ev_k[0]);
    }
    if (o_pinned) pin_out = o;  // page-locked caller rows: the downloads land there directly
    const int nc = stx_host_chunks(ctx, oimg, (size_t)(h + 63) / 64);
    int lo[STX_HOST_MAX_CHUNKS], cnt[STX_HOST_MAX_CHUNKS];
    for (int c = 0; c < nc; ++c) {
        lo[c] = (int)((long long)h * c / nc);
        cnt[c] = (int)((long long)h * (c + 1) / nc) - lo[c];
        rc = stx_launch_disparity(ctx, d_img0, d_img1, 1, w, h, p, 0, lo[c], cnt[c], win, max_disp, cost_mode,
                                  dout + (size_t)lo[c] * op, op, 0, out_bytes);
        if (rc) return rc;
        STX_HIP(hipEventRecord(ctx->ev_k[c], ctx->stream));
        rc = stx_host_download_begin(ctx, pin_out + (size_t)lo[c] * op, dout + (size_t)lo[c] * op, (size_t)cnt[c] * op,
                                     ctx->ev_k[c], ctx->ev_out[c]);
        if (rc) return rc;
    }
    for (int c = 0; c < nc; ++c) {
        rc = stx_host_download_end(ctx, o + (size_t)lo[c] * out_stride, out_stride, pin_out + (size_t)lo[c] * op, op,
                                   (size_t)cnt[c], ctx->ev_out[c]);
        if (rc) return rc;
    }
    return STX_OK;
}

// The pipelines above return at the first failing step.  By then earlier chunks' DMAs and kernels may
// still be in flight on the three streams, aimed at the caller's buffers (direct and zero-copy paths)
// and at the staging the next call reuses: a failed call drains them before it hands control back,
// so the GPU never writes caller memory after the call has returned.
static int drain_on_error(stx_ctx *ctx, int rc)
{
    if (rc == STX_OK) return rc;
    if (ctx->s_in) (void)hipStreamSynchronize(ctx->s_in);
    if (ctx->s_out) (void)hipStreamSynchronize(ctx->s_out);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipGetLastError();
    return rc;
}

// Pinned staging is a convenience, not a requirement: if a pipeline ran out of (page-locked host)
// memory, the call is repeated once with plain copies of the caller's buffers, which need none.
template <class F>
static int with_plain_copy_fallback(stx_ctx *ctx, F call)
{
    int rc = call();
    if (rc == (int)hipErrorOutOfMemory && stx_host_pipeline_on(ctx)) {
        (void)hipGetLastError();
        const int saved = ctx->host_pipeline;
        ctx->host_pipeline = 0;
        rc = call();
        ctx->host_pipeline = saved;
    }
    return rc;
}

int stx_disparity_host(stx_ctx *ctx, const uint8_t *img0, const uint8_t *img1, int n_pairs, int w, int h,
                       size_t stride, size_t pair_stride, int row0, int rows, int win, int max_disp,
                       int cost_mode, void *out, size_t out_stride, size_t out_pair_stride, int out_bytes)
{
    return with_plain_copy_fallback(ctx, [&] {
        return drain_on_error(ctx, disparity_host_pipeline(ctx, img0, img1, n_pairs, w, h, stride, pair_stride, row0, rows,
                                                           win, max_disp, cost_mode, out, out_stride, out_pair_stride,
                                                           out_bytes));
    });
}

int stx_rectify_host(stx_ctx *ctx, const uint8_t *src, int src_w, int src_h, size_t src_stride, uint8_t *dst,
                     int dst_w, int dst_h, size_t dst_stride, const float Hinv[9], int flags, uint8_t *d_keep)
{
    return with_plain_copy_fallback(ctx, [&] {
        return drain_on_error(ctx, rectify_host_pipeline(ctx, src, src_w, src_h, src_stride, dst, dst_w, dst_h, dst_stride,
                                                         Hinv, flags, d_keep));
    });
}

int stx_disparity_resident_host(stx_ctx *ctx, const uint8_t *d_img0, const uint8_t *d_img1, int w, int h, int win,
                                int max_disp, int cost_mode, void *out, size_t out_stride, int out_bytes)
{
    return with_plain_copy_fallback(ctx, [&] {
        return drain_on_error(ctx, disparity_resident_host_pipeline(ctx, d_img0, d_img1, w, h, win, max_disp, cost_mode,
                                                                    out, out_stride, out_bytes));
    });
}

// ------------------------------------------------------ flat arrays (M, ...) --

// A flat host array to the device through the staging pipeline, asynchronous: the compute
// stream is made to wait for it.  `pin_off` is this array's place in the input staging buffer
// (the caller sized it with stx_host_prepare).
int stx_host_upload_flat(stx_ctx *ctx, void *d_dst, const void *src, size_t bytes, size_t pin_off)
{
    if (bytes == 0) return STX_OK;
    if (!stx_host_pipeline_on(ctx)) return (int)hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (stx_host_is_pinned(src, bytes)) {  // page-locked caller array: DMA it as it lies
        STX_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->s_in));
        STX_HIP(hipEventRecord(ctx->ev_in[0], ctx->s_in));
        STX_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_in[0], 0));
        return STX_OK;
    }
    uint8_t *pin = (uint8_t *)ctx->pin_ptr[0] + pin_off;
    const size_t chunk = (size_t)8 << 20;
    for (size_t off = 0; off < bytes; off += chunk) {
        const size_t n = bytes - off < chunk ? bytes - off : chunk;
        // rows of 64 KiB so that the workers share one chunk
        const size_t rowb = (size_t)64 << 10, rows = n / rowb, tail = n - rows * rowb;
        if (rows) pool_copy2d(ctx, pin + off, rowb, (const uint8_t *)src + off, rowb, rowb, rows);
        if (tail) copy_bytes(pin + off + rows * rowb, (const uint8_t *)src + off + rows * rowb, tail);
        STX_HIP(hipMemcpyAsync((uint8_t *)d_dst + off, pin + off, n, hipMemcpyHostToDevice, ctx->s_in));
    }
    STX_HIP(hipEventRecord(ctx->ev_in[0], ctx->s_in));
    STX_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_in[0], 0));
    return STX_OK;
}

int stx_host_fence(stx_ctx *ctx) { return fence_copy_streams(ctx); }
