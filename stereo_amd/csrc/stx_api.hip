// C-ABI entry points of libstereo_hip.so: context management and the host-pointer
// forms, which stage through context-owned device buffers and call the kernel
// launchers in disparity.hip / rectify.hip / match.hip.
#include <mutex>
#include <new>
#include <stdlib.h>
#include <string.h>

#include "stx_internal.h"

extern "C" int stx_version(void) { return STX_VERSION; }

extern "C" const char *stx_error_string(int code)
{
    switch (code) {
    case STX_OK: return "ok";
    case STX_ERR_INVALID_ARG: return "stereo_hip: invalid argument";
    case STX_ERR_UNSUPPORTED: return "stereo_hip: unsupported parameter combination";
    case STX_ERR_NO_DEVICE: return "stereo_hip: no such HIP device";
    case STX_ERR_HOST_OOM: return "stereo_hip: host allocation failed";
    case STX_ERR_TOO_LARGE: return "stereo_hip: problem size exceeds a kernel limit";
    default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "stereo_hip: unknown error";
}

extern "C" int stx_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return -(int)e;
    return n;
}

extern "C" int stx_create(stx_ctx **out, int device)
{
    if (!out) return STX_ERR_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return (int)e;
    if (device < 0 || device >= n) return STX_ERR_NO_DEVICE;
    stx_device_guard guard(device);
    if (guard.err) return guard.err;
    stx_ctx *ctx = new (std::nothrow) stx_ctx();
    if (!ctx) return STX_ERR_HOST_OOM;
    memset(ctx, 0, sizeof(*ctx));
    ctx->device = device;
    ctx->stream = nullptr;
    ctx->match_algo = STX_MATCH_AUTO;
    ctx->host_pipeline = -1;  // STX_HOST_PIPELINE, default on
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        delete ctx;
        return (int)e;
    }
    ctx->num_cus = prop.multiProcessorCount;
    // this device's copy of the static scratch array behind stx_disparity_dev_noalloc
    int rc = stx_disparity_scratch(&ctx->disp_scratch, &ctx->disp_scratch_bytes);
    if (rc != STX_OK) {
        delete ctx;
        return rc;
    }
    *out = ctx;
    return STX_OK;
}

extern "C" void stx_destroy(stx_ctx *ctx)
{
    if (!ctx) return;
    stx_device_guard guard(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    stx_host_release(ctx);
    for (int i = 0; i < STX_SLOT_COUNT; ++i)
        if (ctx->slot_ptr[i]) (void)hipFree(ctx->slot_ptr[i]);
    delete ctx;
}

extern "C" int stx_set_stream(stx_ctx *ctx, void *hip_stream)
{
    if (!ctx) return STX_ERR_INVALID_ARG;
    ctx->stream = (hipStream_t)hip_stream;
    return STX_OK;
}

extern "C" int stx_synchronize(stx_ctx *ctx)
{
    if (!ctx) return STX_ERR_INVALID_ARG;
    STX_ENTER(ctx);
    STX_HIP(hipStreamSynchronize(ctx->stream));
    return STX_OK;
}

extern "C" int stx_set_match_algo(stx_ctx *ctx, int algo)
{
    if (!ctx) return STX_ERR_INVALID_ARG;
    if (algo != STX_MATCH_AUTO && algo != STX_MATCH_EXACT && algo != STX_MATCH_MFMA &&
        algo != STX_MATCH_MFMA_BF16 && algo != STX_MATCH_MFMA_BF16X1 && algo != STX_MATCH_SMALL)
        return STX_ERR_INVALID_ARG;
    ctx->match_algo = algo;
    return STX_OK;
}

extern "C" size_t stx_workspace_bytes(const stx_ctx *ctx)
{
    size_t total = 0;
    if (ctx)
        for (int k = 0; k < STX_SLOT_COUNT; ++k) total += ctx->slot_bytes[k];
    return total;
}

extern "C" size_t stx_staging_bytes(const stx_ctx *ctx) { return ctx ? ctx->pin_bytes[0] + ctx->pin_bytes[1] : 0; }

extern "C" int stx_host_buffer_kind(stx_ctx *ctx, const void *p, size_t bytes)
{
    if (!ctx) return STX_ERR_INVALID_ARG;
    STX_ENTER(ctx);
    if (!stx_host_is_pinned(p, bytes)) return 0;
    return stx_host_kernel_ptr(p, bytes) ? 2 : 1;
}

extern "C" int stx_match_stats(stx_ctx *ctx, long long *total_candidates, int *max_candidates, int *overflow_queries,
                               int *n_queries)
{
    if (!ctx || !total_candidates || !max_candidates || !overflow_queries || !n_queries) return STX_ERR_INVALID_ARG;
    STX_ENTER(ctx);
    return stx_match_stats_impl(ctx, total_candidates, max_candidates, overflow_queries, n_queries);
}

extern "C" int stx_set_disparity_algo(stx_ctx *ctx, int algo)
{
    if (!ctx) return STX_ERR_INVALID_ARG;
    if (algo != STX_DISP_AUTO && algo != STX_DISP_BRUTEFORCE && algo != STX_DISP_COLSUM) return STX_ERR_INVALID_ARG;
    ctx->force_bruteforce = algo == STX_DISP_BRUTEFORCE;
    ctx->force_colsum = algo == STX_DISP_COLSUM;
    return STX_OK;
}

extern "C" int stx_set_host_pipeline(stx_ctx *ctx, int on, int threads, int chunks)
{
    if (!ctx || threads < 0 || chunks < 0 || chunks > STX_HOST_MAX_CHUNKS) return STX_ERR_INVALID_ARG;
    ctx->host_pipeline = on ? 1 : 0;
    if (threads != ctx->host_threads) {
        // the workers are started lazily with the new count
        stx_device_guard guard(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        stx_host_release(ctx);
    }
    ctx->host_threads = threads;
    ctx->host_chunks = chunks;
    return STX_OK;
}

void *stx_workspace(stx_ctx *ctx, int slot, size_t bytes, int *err)
{
    *err = STX_OK;
    if (bytes == 0) bytes = 256;
    if (ctx->slot_bytes[slot] >= bytes) return ctx->slot_ptr[slot];
    // not just ctx->stream: the caller may have switched streams (stx_set_stream) with work
    // that uses the old buffer still in flight on the previous one
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { *err = (int)e; return nullptr; }
    if (ctx->slot_ptr[slot]) {
        (void)hipFree(ctx->slot_ptr[slot]);
        ctx->slot_ptr[slot] = nullptr;
        ctx->slot_bytes[slot] = 0;
    }
    size_t want = (bytes + 4095) & ~(size_t)4095;
    void *p = nullptr;
    e = hipMalloc(&p, want);
    if (e != hipSuccess) { *err = (int)e; return nullptr; }
    ctx->slot_ptr[slot] = p;
    ctx->slot_bytes[slot] = want;
    return p;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel) and only ever
// upwards: the attribute belongs to the device's copy of the function, not to a context, so
// the record is process-wide.
int stx_set_lds_attr(stx_ctx *ctx, const void *fn, int lds)
{
    struct Entry { int device; const void *fn; int lds; };
    static Entry table[256];
    static int count = 0;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    int slot = -1;
    for (int i = 0; i < count; ++i)
        if (table[i].device == ctx->device && table[i].fn == fn) { slot = i; break; }
    if (slot >= 0 && table[slot].lds >= lds) return STX_OK;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    if (slot < 0 && count < 256) slot = count++;
    if (slot >= 0) table[slot] = Entry{ctx->device, fn, lds};
    return STX_OK;
}

// ---------------------------------------------------------------- M (host) --

extern "C" int stx_match_dev(stx_ctx *ctx, const float *d_desc1, const float *d_score1,
                             int n1, const float *d_desc2, const float *d_score2, int n2,
                             float dist_limit, int32_t *d_closest, int32_t *d_second,
                             float *d_min_dist, float *d_second_dist, uint8_t *d_status)
{
    if (!ctx || n1 < 0 || n2 < 0) return STX_ERR_INVALID_ARG;
    if (n1 > 0 && (!d_desc1 || !d_score1)) return STX_ERR_INVALID_ARG;
    if (n2 > 0 && (!d_desc2 || !d_score2)) return STX_ERR_INVALID_ARG;
    STX_ENTER(ctx);
    if (n1 == 0) return STX_OK;
    return stx_launch_match(ctx, d_desc1, d_score1, n1, d_desc2, d_score2, n2, dist_limit,
                            d_closest, d_second, d_min_dist, d_second_dist, d_status);
}

// Large host-pointer matching calls run list1 in TWO query chunks (queries are independent): the
// second chunk is staged and uploaded while the first is matched, so of the two uploads only list2's
// and the first chunk's stay in front of the compute.  The first chunk is a quarter of the queries:
// its matching (~2.3 ms at 200k x 200k) then just covers the upload of the other three quarters
// (~2.1 ms through the staging copies), and two launches cost 0.6 ms of kernel time over one where
// round 2's four equal chunks cost 1.3 ms (every chunk cuts the query blocks into more pieces and
// repeats list2's preparation).  STX_MATCH_CHUNKS=1 turns the chunking off, =4 gives round 2's.
static int match_first_chunk(int n1, int n2, int *rest_chunk)
{
    *rest_chunk = n1;
    if ((double)n1 * (double)n2 < 4.0e9 || n1 < 4 * 16384) return n1;
    const char *env = getenv("STX_MATCH_CHUNKS");
    const int mode = env ? atoi(env) : 2;
    if (mode == 1) return n1;
    int chunk = (n1 + 3) / 4;
    chunk = (chunk + 511) / 512 * 512;  // whole query blocks of the x1 kernel
    *rest_chunk = mode == 4 ? chunk : n1;
    return chunk;
}

// The five result arrays lie back to back in the workspace (n1a elements apart).  Up to 64k queries
// they come back in ONE copy into pinned memory and are split on the host: for a few hundred
// features (BASELINE config 1) five pageable copies of a few hundred bytes each cost more than
// the matching itself.  Larger results go straight to the caller's arrays.
// Page-locked staging for one call, or none: when the host has no page-locked memory left to give,
// the call goes on with plain copies of the caller's buffers (host_pipeline == 0 for its duration).
struct StagingScope {
    stx_ctx *ctx;
    int saved;
    explicit StagingScope(stx_ctx *c) : ctx(c), saved(c->host_pipeline) {}
    ~StagingScope() { ctx->host_pipeline = saved; }
    int prepare(size_t in_bytes, size_t out_bytes)
    {
        if (!stx_host_pipeline_on(ctx)) return STX_OK;
        const int rc = stx_host_prepare(ctx, in_bytes, out_bytes);
        if (rc == (int)hipErrorOutOfMemory) {
            (void)hipGetLastError();
            ctx->host_pipeline = 0;
            return STX_OK;
        }
        return rc ? rc : stx_host_fence(ctx);
    }
};

static int match_download(stx_ctx *ctx, int n1, size_t n1a, const int32_t *dcl, int32_t *closest, int32_t *second,
                          float *min_dist, float *second_dist, uint8_t *status)
{
    hipStream_t s = ctx->stream;
    const char *base = reinterpret_cast<const char *>(dcl);
    void *dst[5] = {closest, second, min_dist, second_dist, status};
    const size_t width[5] = {4, 4, 4, 4, 1};
    const size_t all = 17 * n1a;
    if (stx_host_pipeline_on(ctx) && n1 <= 65536 && ctx->pin_bytes[1] >= all) {
        char *pin = static_cast<char *>(ctx->pin_ptr[1]);
        STX_HIP(hipMemcpyAsync(pin, base, all, hipMemcpyDeviceToHost, s));
        STX_HIP(hipStreamSynchronize(s));
        for (int k = 0; k < 5; ++k)
            if (dst[k]) memcpy(dst[k], pin + 4 * n1a * k, width[k] * (size_t)n1);
        return STX_OK;
    }
    for (int k = 0; k < 5; ++k)
        if (dst[k]) STX_HIP(hipMemcpyAsync(dst[k], base + 4 * n1a * k, width[k] * (size_t)n1, hipMemcpyDeviceToHost, s));
    STX_HIP(hipStreamSynchronize(s));
    return STX_OK;
}

extern "C" int stx_match(stx_ctx *ctx, const float *desc1, const float *score1, int n1,
                         const float *desc2, const float *score2, int n2, float dist_limit,
                         int32_t *closest, int32_t *second, float *min_dist,
                         float *second_dist, uint8_t *status)
{
    if (!ctx || n1 < 0 || n2 < 0) return STX_ERR_INVALID_ARG;
    if (n1 > 0 && (!desc1 || !score1)) return STX_ERR_INVALID_ARG;
    if (n2 > 0 && (!desc2 || !score2)) return STX_ERR_INVALID_ARG;
    STX_ENTER(ctx);
    if (n1 == 0) return STX_OK;
    int err;
    const size_t d1b = (size_t)n1 * STX_DESC_LEN * sizeof(float);
    const size_t d2b = (size_t)n2 * STX_DESC_LEN * sizeof(float);
    float *dd1 = (float *)stx_workspace(ctx, STX_SLOT_DESC1, d1b, &err);
    if (!dd1) return err;
    float *dd2 = (float *)stx_workspace(ctx, STX_SLOT_DESC2, d2b, &err);
    if (!dd2) return err;
    // small arrays: score1[n1] score2[n2] closest[n1] second[n1] mind[n1] secd[n1] status[n1]
    const size_t n1a = ((size_t)n1 + 63) & ~(size_t)63, n2a = ((size_t)n2 + 63) & ~(size_t)63;
    const size_t small_bytes = 4 * n1a + 4 * n2a + 4 * 4 * n1a + n1a;
    char *sm = (char *)stx_workspace(ctx, STX_SLOT_SMALL, small_bytes, &err);
    if (!sm) return err;
    float *ds1 = (float *)sm;
    float *ds2 = (float *)(sm + 4 * n1a);
    int32_t *dcl = (int32_t *)(sm + 4 * n1a + 4 * n2a);
    int32_t *dse = dcl + n1a;
    float *dmd = (float *)(dse + n1a);
    float *dsd = dmd + n1a;
    uint8_t *dst = (uint8_t *)(dsd + n1a);
    // the two descriptor lists (config 5: 102 MB each) go through the staging pipeline
    const size_t s1b = 4 * (size_t)n1, s2b = 4 * (size_t)n2;
    // staging only for the arrays that need it: a page-locked list is DMA'd as it lies (the
    // call used to pin room for both lists -- 256 MB at config 5 -- whether they were staged or not)
    const size_t st1 = stx_host_is_pinned(desc1, d1b) ? 0 : d1b, st2 = stx_host_is_pinned(desc2, d2b) ? 0 : d2b;
    StagingScope staging(ctx);
    const int prc = staging.prepare(st1 + st2 + s1b + s2b, n1 <= 65536 ? 17 * n1a : 0);
    if (prc) return prc;
    // list2 first (its preparation depends on nothing in list1), then list1 in query chunks
    int rc = stx_host_upload_flat(ctx, dd2, desc2, d2b, st1);
    if (!rc) rc = stx_host_upload_flat(ctx, ds2, score2, s2b, st1 + st2 + s1b);
    if (!rc) rc = stx_host_upload_flat(ctx, ds1, score1, s1b, st1 + st2);
    int rest;
    int qc = match_first_chunk(n1, n2, &rest);
    for (int q0 = 0; q0 < n1 && !rc; q0 += qc, qc = rest) {
        const int q = n1 - q0 < qc ? n1 - q0 : qc;
        const size_t off = (size_t)q0 * STX_DESC_LEN;
        rc = stx_host_upload_flat(ctx, dd1 + off, desc1 + off, sizeof(float) * (size_t)q * STX_DESC_LEN,
                                  sizeof(float) * off);
        if (!rc)
            rc = stx_launch_match(ctx, dd1 + off, ds1 + q0, q, dd2, ds2, n2, dist_limit, dcl + q0, dse + q0, dmd + q0,
                                  dsd + q0, dst + q0);
    }
    if (rc != STX_OK) return rc;
    return match_download(ctx, n1, n1a, dcl, closest, second, min_dist, second_dist, status);
}

extern "C" int stx_match_features(stx_ctx *ctx, const void *feat1, int n1, const void *feat2, int n2,
                                  size_t feature_stride, size_t desc_offset, size_t score_offset, float dist_limit,
                                  int32_t *closest, int32_t *second, float *min_dist, float *second_dist,
                                  uint8_t *status)
{
    if (!ctx || n1 < 0 || n2 < 0) return STX_ERR_INVALID_ARG;
    if ((n1 > 0 && !feat1) || (n2 > 0 && !feat2)) return STX_ERR_INVALID_ARG;
    if ((feature_stride & 3) || (desc_offset & 3) || (score_offset & 3) ||
        desc_offset + sizeof(float) * STX_DESC_LEN > feature_stride || score_offset + sizeof(float) > feature_stride)
        return STX_ERR_INVALID_ARG;
    STX_ENTER(ctx);
    if (n1 == 0) return STX_OK;
    int err;
    const size_t r1b = (size_t)n1 * feature_stride, r2b = (size_t)n2 * feature_stride;
    const size_t d1b = (size_t)n1 * STX_DESC_LEN * sizeof(float), d2b = (size_t)n2 * STX_DESC_LEN * sizeof(float);
    unsigned char *drec = (unsigned char *)stx_workspace(ctx, STX_SLOT_AOS, r1b + r2b, &err);
    if (!drec) return err;
    float *dd1 = (float *)stx_workspace(ctx, STX_SLOT_DESC1, d1b, &err);
    if (!dd1) return err;
    float *dd2 = (float *)stx_workspace(ctx, STX_SLOT_DESC2, d2b, &err);
    if (!dd2) return err;
    const size_t n1a = ((size_t)n1 + 63) & ~(size_t)63, n2a = ((size_t)n2 + 63) & ~(size_t)63;
    char *sm = (char *)stx_workspace(ctx, STX_SLOT_SMALL, 4 * n1a + 4 * n2a + 4 * 4 * n1a + n1a, &err);
    if (!sm) return err;
    float *ds1 = (float *)sm;
    float *ds2 = (float *)(sm + 4 * n1a);
    int32_t *dcl = (int32_t *)(sm + 4 * n1a + 4 * n2a);
    int32_t *dse = dcl + n1a;
    float *dmd = (float *)(dse + n1a);
    float *dsd = dmd + n1a;
    uint8_t *dst = (uint8_t *)(dsd + n1a);
    // list2 first: its preparation (norms, fragments) depends on nothing in list1
    const size_t st1 = stx_host_is_pinned(feat1, r1b) ? 0 : r1b, st2 = stx_host_is_pinned(feat2, r2b) ? 0 : r2b;
    StagingScope staging(ctx);
    const int prc = staging.prepare(st1 + st2, n1 <= 65536 ? 17 * n1a : 0);
    if (prc) return prc;
    int rc = stx_host_upload_flat(ctx, drec + r1b, feat2, r2b, st1);
    if (!rc) rc = stx_launch_unpack_features(ctx, drec + r1b, n2, feature_stride, desc_offset, score_offset, dd2, ds2);
    int rest;
    int qc = match_first_chunk(n1, n2, &rest);
    for (int q0 = 0; q0 < n1 && !rc; q0 += qc, qc = rest) {  // list1 in query chunks: upload c + 1 behind match c
        const int q = n1 - q0 < qc ? n1 - q0 : qc;
        const size_t roff = (size_t)q0 * feature_stride;
        rc = stx_host_upload_flat(ctx, drec + roff, (const unsigned char *)feat1 + roff, (size_t)q * feature_stride, roff);
        if (!rc)
            rc = stx_launch_unpack_features(ctx, drec + roff, q, feature_stride, desc_offset, score_offset,
                                            dd1 + (size_t)q0 * STX_DESC_LEN, ds1 + q0);
        if (!rc)
            rc = stx_launch_match(ctx, dd1 + (size_t)q0 * STX_DESC_LEN, ds1 + q0, q, dd2, ds2, n2, dist_limit, dcl + q0,
                                  dse + q0, dmd + q0, dsd + q0, dst + q0);
    }
    if (rc) return rc;
    return match_download(ctx, n1, n1a, dcl, closest, second, min_dist, second_dist, status);
}

// ---------------------------------------------------------------- R (host) --

// Eigen 3.3 compute_inverse<Matrix3f,3> operation order (third-party arithmetic
// restated): cofactor<i,j> = m(i1,j1)*m(i2,j2) - m(i1,j2)*m(i2,j1),
// det = c00*m00 + (c10*m10 + c20*m20), every entry = cofactor * (1/det).
static inline float stx_cof3(const float *m, int i, int j)
{
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    volatile float p = m[i1 * 3 + j1] * m[i2 * 3 + j2];
    volatile float q = m[i1 * 3 + j2] * m[i2 * 3 + j1];
    return p - q;
}

extern "C" int stx_invert_homography(const float H[9], float Hinv[9])
{
    if (!H || !Hinv) return STX_ERR_INVALID_ARG;
    float c[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c[i][j] = stx_cof3(H, i, j);
    volatile float a0 = c[0][0] * H[0], a1 = c[1][0] * H[3], a2 = c[2][0] * H[6];
    volatile float t = a1 + a2;
    const float det = a0 + t;
    const float invdet = 1.0f / det;
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k) Hinv[r * 3 + k] = c[k][r] * invdet;
    return STX_OK;
}


// Host <-> device image copies.  hipMemcpy2DAsync on pageable memory runs at well under
// 1 GB/s on this stack (measured: 51 ms for three 5.7 MB images), a flat hipMemcpyAsync at
// ~16 GB/s, so densely packed host images (stride == width, the cv::Mat::zeros / imread
// case) are copied flat into an equally dense device image; only pitched views take the 2-D path.
static hipError_t copy_image_h2d(void *dst, size_t dpitch, const void *src, size_t spitch, size_t wbytes,
                                 size_t h, hipStream_t s)
{
    if (dpitch == wbytes && spitch == wbytes) return hipMemcpyAsync(dst, src, wbytes * h, hipMemcpyHostToDevice, s);
    return hipMemcpy2DAsync(dst, dpitch, src, spitch, wbytes, h, hipMemcpyHostToDevice, s);
}
static hipError_t copy_image_d2h(void *dst, size_t dpitch, const void *src, size_t spitch, size_t wbytes,
                                 size_t h, hipStream_t s)
{
    if (dpitch == wbytes && spitch == wbytes) return hipMemcpyAsync(dst, src, wbytes * h, hipMemcpyDeviceToHost, s);
    return hipMemcpy2DAsync(dst, dpitch, src, spitch, wbytes, h, hipMemcpyDeviceToHost, s);
}

static int check_image(const void *p, int w, int h, size_t stride)
{
    if (w < 0 || h < 0) return STX_ERR_INVALID_ARG;
    if (w > 0 && h > 0 && (!p || stride < (size_t)w)) return STX_ERR_INVALID_ARG;
    return STX_OK;
}

extern "C" int stx_rectify_u8_dev(stx_ctx *ctx, const uint8_t *d_src, int src_w, int src_h,
                                  size_t src_stride, size_t src_image_stride, uint8_t *d_dst,
                                  int dst_w, int dst_h, size_t dst_stride,
                                  size_t dst_image_stride, const float *Hinv, int n_images,
                                  int flags)
{
    if (!ctx || !Hinv || n_images < 0) return STX_ERR_INVALID_ARG;
    if (flags != STX_RECT_PRESERVE_DST && flags != STX_RECT_DST_IS_ZERO) return STX_ERR_INVALID_ARG;
    int rc = check_image(d_src, src_w, src_h, src_stride);
    if (rc) return rc;
    rc = check_image(d_dst, dst_w, dst_h, dst_stride);
    if (rc) return rc;
    STX_ENTER(ctx);
    if (n_images == 0 || dst_w == 0 || dst_h == 0) return STX_OK;
    if (n_images <= 2)  // the homographies ride in the kernel arguments: nothing to upload
        return stx_launch_rectify(ctx, d_src, src_w, src_h, src_stride, src_image_stride, d_dst, dst_w, dst_h,
                                  dst_stride, dst_image_stride, nullptr, n_images, flags, 0, Hinv);
    int err;
    float *d_hinv = (float *)stx_workspace(ctx, STX_SLOT_HINV, sizeof(float) * 9 * (size_t)n_images, &err);
    if (!d_hinv) return err;
    // pageable source: the copy is staged by the runtime before this returns
    STX_HIP(hipMemcpyAsync(d_hinv, Hinv, sizeof(float) * 9 * (size_t)n_images,
                           hipMemcpyHostToDevice, ctx->stream));
    return stx_launch_rectify(ctx, d_src, src_w, src_h, src_stride, src_image_stride, d_dst,
                              dst_w, dst_h, dst_stride, dst_image_stride, d_hinv, n_images, flags);
}

extern "C" int stx_rectify_rowband_u8_dev(stx_ctx *ctx, const uint8_t *d_src, int src_w, int src_h,
                                          size_t src_stride, uint8_t *d_dst, int dst_w, int dst_row0,
                                          int dst_rows, size_t dst_stride, const float Hinv[9],
                                          int flags)
{
    if (!ctx || !Hinv || dst_row0 < 0) return STX_ERR_INVALID_ARG;
    if (flags != STX_RECT_PRESERVE_DST && flags != STX_RECT_DST_IS_ZERO) return STX_ERR_INVALID_ARG;
    int rc = check_image(d_src, src_w, src_h, src_stride);
    if (rc) return rc;
    rc = check_image(d_dst, dst_w, dst_rows, dst_stride);
    if (rc) return rc;
    if ((long long)dst_row0 + dst_rows > (1ll << 24)) return STX_ERR_TOO_LARGE;  // (float)y exact
    STX_ENTER(ctx);
    if (dst_w == 0 || dst_rows == 0) return STX_OK;
    return stx_launch_rectify(ctx, d_src, src_w, src_h, src_stride, 0, d_dst, dst_w, dst_rows, dst_stride, 0,
                              nullptr, 1, flags, dst_row0, Hinv);
}

extern "C" int stx_rectify_u8(stx_ctx *ctx, const uint8_t *src, int src_w, int src_h,
                              size_t src_stride, uint8_t *dst, int dst_w, int dst_h,
                              size_t dst_stride, const float Hinv[9], int flags)
{
    if (!ctx || !Hinv) return STX_ERR_INVALID_ARG;
    if (flags != STX_RECT_PRESERVE_DST && flags != STX_RECT_DST_IS_ZERO) return STX_ERR_INVALID_ARG;
    int rc = check_image(src, src_w, src_h, src_stride);
    if (rc) return rc;
    rc = check_image(dst, dst_w, dst_h, dst_stride);
    if (rc) return rc;
    STX_ENTER(ctx);
    if (dst_w == 0 || dst_h == 0) return STX_OK;
    return stx_rectify_host(ctx, src, src_w, src_h, src_stride, dst, dst_w, dst_h, dst_stride, Hinv, flags);
}

static int check_disparity_args(int w, int h, size_t stride, int win, int max_disp, int cost_mode, int out_bytes,
                                size_t out_stride);

extern "C" int stx_rectify_u8_keep(stx_ctx *ctx, int slot, const uint8_t *src, int src_w, int src_h,
                                   size_t src_stride, uint8_t *dst, int dst_w, int dst_h, size_t dst_stride,
                                   const float Hinv[9], int flags)
{
    if (!ctx || !Hinv || (slot != 0 && slot != 1)) return STX_ERR_INVALID_ARG;
    if (flags != STX_RECT_PRESERVE_DST && flags != STX_RECT_DST_IS_ZERO) return STX_ERR_INVALID_ARG;
    int rc = check_image(src, src_w, src_h, src_stride);
    if (rc) return rc;
    rc = check_image(dst, dst_w, dst_h, dst_stride);
    if (rc) return rc;
    STX_ENTER(ctx);
    ctx->res_w[slot] = ctx->res_h[slot] = 0;
    if (dst_w == 0 || dst_h == 0) return STX_OK;
    int err;
    uint8_t *keep = (uint8_t *)stx_workspace(ctx, STX_SLOT_RES0 + slot, (size_t)dst_w * (size_t)dst_h, &err);
    if (!keep) return err;
    rc = stx_rectify_host(ctx, src, src_w, src_h, src_stride, dst, dst_w, dst_h, dst_stride, Hinv, flags, keep);
    if (rc) return rc;
    ctx->res_w[slot] = dst_w;
    ctx->res_h[slot] = dst_h;
    return STX_OK;
}

extern "C" int stx_disparity_resident_u8(stx_ctx *ctx, int slot_img0, int slot_img1, int w, int h, int win,
                                         int max_disp, int cost_mode, uint8_t *out, size_t out_stride)
{
    if (!ctx || (slot_img0 != 0 && slot_img0 != 1) || (slot_img1 != 0 && slot_img1 != 1)) return STX_ERR_INVALID_ARG;
    int rc = check_disparity_args(w, h, (size_t)(w > 0 ? w : 0), win, max_disp, cost_mode, 1, out_stride);
    if (rc) return rc;
    // both slots must hold an image of exactly this size (stx_rectify_u8_keep)
    if (ctx->res_w[slot_img0] != w || ctx->res_h[slot_img0] != h || ctx->res_w[slot_img1] != w ||
        ctx->res_h[slot_img1] != h)
        return STX_ERR_INVALID_ARG;
    if (w == 0 || h == 0) return STX_OK;
    if (!out) return STX_ERR_INVALID_ARG;
    STX_ENTER(ctx);
    return stx_disparity_resident_host(ctx, (const uint8_t *)ctx->slot_ptr[STX_SLOT_RES0 + slot_img0],
                                       (const uint8_t *)ctx->slot_ptr[STX_SLOT_RES0 + slot_img1], w, h, win, max_disp,
                                       cost_mode, out, out_stride, 1);
}

// ---------------------------------------------------------------- D (host) --

static int check_disparity_args(int w, int h, size_t stride, int win, int max_disp,
                                int cost_mode, int out_bytes, size_t out_stride)
{
    if (w < 0 || h < 0 || max_disp < 0) return STX_ERR_INVALID_ARG;
    if (win < 1 || (win & 1) == 0) return STX_ERR_INVALID_ARG;
    if (cost_mode != STX_COST_SAD && cost_mode != STX_COST_SSD) return STX_ERR_INVALID_ARG;
    if (out_bytes != 1 && out_bytes != 2) return STX_ERR_INVALID_ARG;
    if (stride < (size_t)w || out_stride < (size_t)w * (size_t)out_bytes) return STX_ERR_INVALID_ARG;
    if (out_bytes == 2 && (out_stride & 1)) return STX_ERR_INVALID_ARG;
    if (max_disp == 0 && win != 1) return STX_ERR_UNSUPPORTED;
    if (win > 31) return STX_ERR_UNSUPPORTED;
    return STX_OK;
}

extern "C" int stx_disparity_rowband_dev(stx_ctx *ctx, const uint8_t *d_img0,
                                         const uint8_t *d_img1, int w, int h, int band_row0,
                                         int full_h, size_t stride, int row0, int rows,
                                         int win, int max_disp, int cost_mode, void *d_out,
                                         size_t out_stride, int out_bytes)
{
    if (!ctx) return STX_ERR_INVALID_ARG;
    int rc = check_disparity_args(w, h, stride, win, max_disp, cost_mode, out_bytes, out_stride);
    if (rc) return rc;
    if (band_row0 < 0 || full_h < 0 || rows < 0 || row0 < band_row0 ||
        band_row0 + h > full_h || row0 + rows > band_row0 + h)
        return STX_ERR_INVALID_ARG;
    // the band must carry the halo rows the full image has
    const int r = win / 2;
    const int need_lo = row0 - r > 0 ? row0 - r : 0;
    const int need_hi = row0 + rows + r < full_h ? row0 + rows + r : full_h;
    if (rows > 0 && (band_row0 > need_lo || band_row0 + h < need_hi)) return STX_ERR_INVALID_ARG;
    if (w == 0 || rows == 0) return STX_OK;
    if (!d_img0 || !d_img1 || !d_out) return STX_ERR_INVALID_ARG;
    STX_ENTER(ctx);
    // A band whose first/last rows are interior rows of the full image has its
    // window clipped only where the halo is present, so clipping against the
    // band equals clipping against the full image.
    return stx_launch_disparity(ctx, d_img0, d_img1, 1, w, h, stride, 0, row0 - band_row0, rows,
                                win, max_disp, cost_mode, d_out, out_stride, 0, out_bytes);
}

extern "C" int stx_disparity_dev(stx_ctx *ctx, const uint8_t *d_img0, const uint8_t *d_img1,
                                 int n_pairs, int w, int h, size_t stride, size_t pair_stride,
                                 int win, int max_disp, int cost_mode, void *d_out,
                                 size_t out_stride, size_t out_pair_stride, int out_bytes)
{
    if (!ctx || n_pairs < 0) return STX_ERR_INVALID_ARG;
    int rc = check_disparity_args(w, h, stride, win, max_disp, cost_mode, out_bytes, out_stride);
    if (rc) return rc;
    if (n_pairs == 0 || w == 0 || h == 0) return STX_OK;
    if (!d_img0 || !d_img1 || !d_out) return STX_ERR_INVALID_ARG;
    if (n_pairs > 1 && (pair_stride < stride * (size_t)h || out_pair_stride < out_stride * (size_t)h))
        return STX_ERR_INVALID_ARG;
    STX_ENTER(ctx);
    return stx_launch_disparity(ctx, d_img0, d_img1, n_pairs, w, h, stride, pair_stride, 0, h, win,
                                max_disp, cost_mode, d_out, out_stride, out_pair_stride, out_bytes);
}

extern "C" int stx_disparity_dev_noalloc(stx_ctx *ctx, const uint8_t *d_img0, const uint8_t *d_img1,
                                         int n_pairs, int w, int h, size_t stride, size_t pair_stride,
                                         int win, int max_disp, int cost_mode, void *d_out,
                                         size_t out_stride, size_t out_pair_stride, int out_bytes)
{
    if (!ctx || n_pairs < 0) return STX_ERR_INVALID_ARG;
    int rc = check_disparity_args(w, h, stride, win, max_disp, cost_mode, out_bytes, out_stride);
    if (rc) return rc;
    if (n_pairs == 0 || w == 0 || h == 0) return STX_OK;
    if (!d_img0 || !d_img1 || !d_out) return STX_ERR_INVALID_ARG;
    if (n_pairs > 1 && (pair_stride < stride * (size_t)h || out_pair_stride < out_stride * (size_t)h))
        return STX_ERR_INVALID_ARG;
    if (!ctx->disp_scratch) return STX_ERR_UNSUPPORTED;
    STX_ENTER(ctx);
    // ctx->disp_scratch is this device's address of the static array (hipGetSymbolAddress at
    // stx_create); the launcher rejects scans whose keys do not fit it and never allocates
    return stx_launch_disparity(ctx, d_img0, d_img1, n_pairs, w, h, stride, pair_stride, 0, h, win,
                                max_disp, cost_mode, d_out, out_stride, out_pair_stride, out_bytes,
                                ctx->disp_scratch, ctx->disp_scratch_bytes);
}

static int disparity_host(stx_ctx *ctx, const uint8_t *img0, const uint8_t *img1, int w, int h,
                          size_t stride, int win, int max_disp, int cost_mode, void *out,
                          size_t out_stride, int out_bytes)
{
    if (!ctx) return STX_ERR_INVALID_ARG;
    int rc = check_disparity_args(w, h, stride, win, max_disp, cost_mode, out_bytes, out_stride);
    if (rc) return rc;
    if (w == 0 || h == 0) return STX_OK;
    if (!img0 || !img1 || !out) return STX_ERR_INVALID_ARG;
    STX_ENTER(ctx);
    return stx_disparity_host(ctx, img0, img1, 1, w, h, stride, 0, 0, h, win, max_disp, cost_mode, out, out_stride, 0,
                              out_bytes);
}

extern "C" int stx_disparity_u8(stx_ctx *ctx, const uint8_t *img0, const uint8_t *img1, int w,
                                int h, size_t stride, int win, int max_disp, int cost_mode,
                                uint8_t *out, size_t out_stride)
{
    return disparity_host(ctx, img0, img1, w, h, stride, win, max_disp, cost_mode, out, out_stride, 1);
}

extern "C" int stx_disparity_u16(stx_ctx *ctx, const uint8_t *img0, const uint8_t *img1, int w,
                                 int h, size_t stride, int win, int max_disp, int cost_mode,
                                 uint16_t *out, size_t out_stride)
{
    return disparity_host(ctx, img0, img1, w, h, stride, win, max_disp, cost_mode, out, out_stride, 2);
}

// ------------------------------------------------------------- R + D fused --

extern "C" int stx_rectify_disparity_u8(stx_ctx *ctx, const uint8_t *img0, const uint8_t *img1, int w,
                                        int h, size_t stride, const float Hinv0[9], const float Hinv1[9],
                                        int win, int max_disp, int cost_mode, uint8_t *out,
                                        size_t out_stride, uint8_t *rect0, uint8_t *rect1,
                                        size_t rect_stride)
{
    if (!ctx || !Hinv0 || !Hinv1) return STX_ERR_INVALID_ARG;
    int rc = check_disparity_args(w, h, stride, win, max_disp, cost_mode, 1, out_stride);
    if (rc) return rc;
    if ((rect0 || rect1) && rect_stride < (size_t)w) return STX_ERR_INVALID_ARG;
    if (w == 0 || h == 0) return STX_OK;
    if (!img0 || !img1 || !out) return STX_ERR_INVALID_ARG;
    STX_ENTER(ctx);
    int err;
    hipStream_t s = ctx->stream;
    const size_t p = (size_t)w;
    const size_t img = p * (size_t)h;
    // [src0 | src1] in IMG0, [rect0 | rect1] in IMG1, disparity in OUT
    uint8_t *dsrc = (uint8_t *)stx_workspace(ctx, STX_SLOT_IMG0, 2 * img, &err);
    if (!dsrc) return err;
    uint8_t *drect = (uint8_t *)stx_workspace(ctx, STX_SLOT_IMG1, 2 * img, &err);
    if (!drect) return err;
    uint8_t *dout = (uint8_t *)stx_workspace(ctx, STX_SLOT_OUT, img, &err);
    if (!dout) return err;
    const int nout = 1 + (rect0 ? 1 : 0) + (rect1 ? 1 : 0);
    StagingScope staging(ctx);
    rc = staging.prepare(2 * img, (size_t)nout * img);
    if (rc) return rc;
    const bool pipe = stx_host_pipeline_on(ctx);
    if (pipe) {
        uint8_t *pin_in = (uint8_t *)ctx->pin_ptr[0];
        // row chunks: the staging copy of chunk c+1 runs while chunk c is on the bus
        const int nc = stx_host_chunks(ctx, img, (size_t)h);
        for (int k = 0; k < 2; ++k)
            for (int c = 0; c < nc; ++c) {
                const size_t r0 = (size_t)h * c / nc, r1 = (size_t)h * (c + 1) / nc;
                rc = stx_host_upload(ctx, dsrc + k * img + r0 * p, pin_in + k * img + r0 * p,
                                     (k ? img1 : img0) + r0 * stride, stride, p, r1 - r0, nullptr);
                if (rc) return rc;
            }
        STX_HIP(hipEventRecord(ctx->ev_in[0], ctx->s_in));
        STX_HIP(hipStreamWaitEvent(s, ctx->ev_in[0], 0));
    } else {
        STX_HIP(copy_image_h2d(dsrc, p, img0, stride, (size_t)w, (size_t)h, s));
        STX_HIP(copy_image_h2d(dsrc + img, p, img1, stride, (size_t)w, (size_t)h, s));
    }
    float hinv[18];
    memcpy(hinv, Hinv0, 9 * sizeof(float));
    memcpy(hinv + 9, Hinv1, 9 * sizeof(float));
    // zero-initialised destinations (Mat::zeros, main.cpp:362,366): misses are written as 0
    rc = stx_rectify_u8_dev(ctx, dsrc, w, h, p, img, drect, w, h, p, img, hinv, 2, STX_RECT_DST_IS_ZERO);
    if (rc) return rc;
    if (pipe) STX_HIP(hipEventRecord(ctx->ev_k[1], s));  // the rectified images may leave while the scan runs
    // a small map is stored by the scan straight into the pinned buffer (as stx_disparity_u8 does)
    const bool zc = pipe && img <= ((size_t)4 << 20) && stx_host_zero_copy_on();
    rc = stx_launch_disparity(ctx, drect, drect + img, 1, w, h, p, 0, 0, h, win, max_disp, cost_mode,
                              zc ? (uint8_t *)ctx->pin_ptr[1] : dout, p, 0, 1);
    if (rc) return rc;
    if (pipe) {
        uint8_t *pin_out = (uint8_t *)ctx->pin_ptr[1];
        STX_HIP(hipEventRecord(ctx->ev_k[0], s));
        int slot = 1;
        const int s0 = rect0 ? slot++ : 0, s1 = rect1 ? slot++ : 0;
        if (rect0) {
            rc = stx_host_download_begin(ctx, pin_out + (size_t)s0 * img, drect, img, ctx->ev_k[1], ctx->ev_out[s0]);
            if (rc) return rc;
        }
        if (rect1) {
            rc = stx_host_download_begin(ctx, pin_out + (size_t)s1 * img, drect + img, img, ctx->ev_k[1], ctx->ev_out[s1]);
            if (rc) return rc;
        }
        if (!zc) {
            rc = stx_host_download_begin(ctx, pin_out, dout, img, ctx->ev_k[0], ctx->ev_out[0]);
            if (rc) return rc;
        }
        if (rect0) {
            rc = stx_host_download_end(ctx, rect0, rect_stride, pin_out + (size_t)s0 * img, p, (size_t)h, ctx->ev_out[s0]);
            if (rc) return rc;
        }
        if (rect1) {
            rc = stx_host_download_end(ctx, rect1, rect_stride, pin_out + (size_t)s1 * img, p, (size_t)h, ctx->ev_out[s1]);
            if (rc) return rc;
        }
        return stx_host_download_end(ctx, out, out_stride, pin_out, p, (size_t)h, zc ? ctx->ev_k[0] : ctx->ev_out[0]);
    }
    STX_HIP(copy_image_d2h(out, out_stride, dout, p, (size_t)w, (size_t)h, s));
    if (rect0) STX_HIP(copy_image_d2h(rect0, rect_stride, drect, p, (size_t)w, (size_t)h, s));
    if (rect1) STX_HIP(copy_image_d2h(rect1, rect_stride, drect + img, p, (size_t)w, (size_t)h, s));
    STX_HIP(hipStreamSynchronize(s));
    return STX_OK;
}

// ------------------------------------------------------------------- FAST --

extern "C" int stx_fast_detect_u8_dev(stx_ctx *ctx, const uint8_t *d_img, int w, int h, size_t stride,
                                      int threshold, float *d_xy, int max_out, int *d_count)
{
    if (!ctx || !d_count || max_out < 0 || (max_out > 0 && !d_xy)) return STX_ERR_INVALID_ARG;
    int rc = check_image(d_img, w, h, stride);
    if (rc) return rc;
    STX_ENTER(ctx);
    if (w == 0 || h == 0) return (int)hipMemsetAsync(d_count, 0, sizeof(int), ctx->stream);
    return stx_launch_fast(ctx, d_img, w, h, stride, threshold, d_xy, max_out, d_count);
}

extern "C" int stx_fast_detect_u8(stx_ctx *ctx, const uint8_t *img, int w, int h, size_t stride,
                                  int threshold, float *xy, int max_out, int *count)
{
    if (!ctx || !count || max_out < 0 || (max_out > 0 && !xy)) return STX_ERR_INVALID_ARG;
    int rc = check_image(img, w, h, stride);
    if (rc) return rc;
    *count = 0;
    if (w == 0 || h == 0) return STX_OK;
    STX_ENTER(ctx);
    int err;
    hipStream_t s = ctx->stream;
    const size_t p = (size_t)w;
    uint8_t *dimg = (uint8_t *)stx_workspace(ctx, STX_SLOT_IMG0, p * (size_t)h, &err);
    if (!dimg) return err;
    // [count (16 bytes)][xy max_out * 2]: the count and the first corners come back in ONE copy
    // through pinned memory and one wait; only an image with more corners than that needs a second
    // copy (a few hundred to a few thousand corners is the usual case, BASELINE config 1)
    char *dbase = (char *)stx_workspace(ctx, STX_SLOT_OUT, 16 + sizeof(float) * 2 * (size_t)max_out, &err);
    if (!dbase) return err;
    int *dcount = reinterpret_cast<int *>(dbase);
    float *dxy = reinterpret_cast<float *>(dbase + 16);
    STX_HIP(copy_image_h2d(dimg, p, img, stride, (size_t)w, (size_t)h, s));
    rc = stx_launch_fast(ctx, dimg, w, h, p, threshold, dxy, max_out, dcount);
    if (rc) return rc;
    const size_t first = (size_t)(max_out < 8192 ? max_out : 8192);
    size_t have = 0;  // corners already on the host
    if (stx_host_pipeline_on(ctx) && stx_host_prepare(ctx, 0, 16 + sizeof(float) * 2 * first) == STX_OK) {
        char *pin = static_cast<char *>(ctx->pin_ptr[1]);
        STX_HIP(hipMemcpyAsync(pin, dbase, 16 + sizeof(float) * 2 * first, hipMemcpyDeviceToHost, s));
        STX_HIP(hipStreamSynchronize(s));
        memcpy(count, pin, sizeof(int));
        const size_t n = (size_t)(*count < max_out ? *count : max_out);
        have = n < first ? n : first;
        if (have > 0) memcpy(xy, pin + 16, sizeof(float) * 2 * have);
    } else {
        STX_HIP(hipMemcpyAsync(count, dcount, sizeof(int), hipMemcpyDeviceToHost, s));
        STX_HIP(hipStreamSynchronize(s));
    }
    const size_t n = (size_t)(*count < max_out ? *count : max_out);
    if (n > have) {
        STX_HIP(hipMemcpyAsync(xy + 2 * have, dxy + 2 * have, sizeof(float) * 2 * (n - have), hipMemcpyDeviceToHost, s));
        STX_HIP(hipStreamSynchronize(s));
    }
    return STX_OK;
}

// ---------------------------------------------------------- scoring + NMS --

extern "C" int stx_gaussian_kernel(int size, float sigma, float *out)
{
    if (size <= 0 || !out) return STX_ERR_INVALID_ARG;
    // Features.cpp:22-31 (the reference writes img.at<float>(x, y); the kernel is symmetric)
    for (int y = 0; y < size; ++y)
        for (int x = 0; x < size; ++x) {
            const float x_u = x - size / 2.f;
            const float y_u = y - size / 2.f;
            const float g = (1.f / (2.f * sigma * sigma * (float)3.14159265358979323846)) *
                            expf(-1 * (x_u * x_u + y_u * y_u) / (2 * sigma * sigma));
            out[x * size + y] = g;
        }
    return STX_OK;
}

extern "C" int stx_score_cluster_dev(stx_ctx *ctx, const uint8_t *d_gx, const uint8_t *d_gy, int w, int h,
                                     size_t stride, const float *d_xy, int n, const float kernel3x3[9],
                                     float score_threshold, float cluster_dist, float *d_xy_out,
                                     float *d_score_out, int32_t *d_index_out, int *d_count)
{
    if (!ctx || !kernel3x3 || !d_count || n < 0) return STX_ERR_INVALID_ARG;
    if (n > 0 && (!d_xy || !d_xy_out || !d_score_out || !d_index_out)) return STX_ERR_INVALID_ARG;
    int rc = check_image(d_gx, w, h, stride);
    if (rc) return rc;
    rc = check_image(d_gy, w, h, stride);
    if (rc) return rc;
    STX_ENTER(ctx);
    return stx_launch_score_cluster(ctx, d_gx, d_gy, w, h, stride, d_xy, n, kernel3x3, score_threshold,
                                    cluster_dist, d_xy_out, d_score_out, d_index_out, d_count);
}

extern "C" int stx_score_cluster(stx_ctx *ctx, const uint8_t *gx, const uint8_t *gy, int w, int h,
                                 size_t stride, const float *xy, int n, const float kernel3x3[9],
                                 float score_threshold, float cluster_dist, float *xy_out, float *score_out,
                                 int32_t *index_out, int *count)
{
    if (!ctx || !kernel3x3 || !count || n < 0) return STX_ERR_INVALID_ARG;
    if (n > 0 && (!xy || !xy_out || !score_out || !index_out)) return STX_ERR_INVALID_ARG;
    int rc = check_image(gx, w, h, stride);
    if (rc) return rc;
    rc = check_image(gy, w, h, stride);
    if (rc) return rc;
    *count = 0;
    if (n == 0) return STX_OK;
    STX_ENTER(ctx);
    int err;
    hipStream_t s = ctx->stream;
    const size_t p = (size_t)(w > 0 ? w : 1), img = (p * (size_t)(h > 0 ? h : 1) + 15) & ~(size_t)15;
    // [gx][gy][xy n*2][xy_out n*2][score_out n][index_out n][count]
    const size_t bytes = 2 * img + sizeof(float) * (size_t)n * 6 + 16;
    uint8_t *base = (uint8_t *)stx_workspace(ctx, STX_SLOT_SCORE_IO, bytes, &err);
    if (!base) return err;
    uint8_t *dgx = base, *dgy = base + img;
    float *dxy = reinterpret_cast<float *>(base + 2 * img);
    float *dxy_out = dxy + 2 * (size_t)n;
    float *dscore = dxy_out + 2 * (size_t)n;
    int32_t *dindex = reinterpret_cast<int32_t *>(dscore + n);
    int *dcount = dindex + n;
    if (w > 0 && h > 0) {
        STX_HIP(copy_image_h2d(dgx, p, gx, stride, (size_t)w, (size_t)h, s));
        STX_HIP(copy_image_h2d(dgy, p, gy, stride, (size_t)w, (size_t)h, s));
    }
    STX_HIP(hipMemcpyAsync(dxy, xy, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice, s));
    rc = stx_launch_score_cluster(ctx, dgx, dgy, w, h, p, dxy, n, kernel3x3, score_threshold, cluster_dist,
                                  dxy_out, dscore, dindex, dcount);
    if (rc) return rc;
    // [xy_out][score_out][index_out][count] lie back to back: up to 64k corners they come back in one
    // copy through pinned memory and one wait (a few hundred corners per image in BASELINE config 1:
    // the count's round trip and three more copies cost as much as the kernels)
    const size_t out_all = sizeof(float) * 4 * (size_t)n + sizeof(int);
    if (stx_host_pipeline_on(ctx) && n <= 65536 && stx_host_prepare(ctx, 0, out_all) == STX_OK) {
        char *pin = static_cast<char *>(ctx->pin_ptr[1]);
        STX_HIP(hipMemcpyAsync(pin, dxy_out, out_all, hipMemcpyDeviceToHost, s));
        STX_HIP(hipStreamSynchronize(s));
        memcpy(count, pin + sizeof(float) * 4 * (size_t)n, sizeof(int));
        const size_t m = (size_t)*count;
        memcpy(xy_out, pin, sizeof(float) * 2 * m);
        memcpy(score_out, pin + sizeof(float) * 2 * (size_t)n, sizeof(float) * m);
        memcpy(index_out, pin + sizeof(float) * 3 * (size_t)n, sizeof(int32_t) * m);
        return STX_OK;
    }
    STX_HIP(hipMemcpyAsync(count, dcount, sizeof(int), hipMemcpyDeviceToHost, s));
    STX_HIP(hipStreamSynchronize(s));
    const size_t m = (size_t)*count;
    if (m > 0) {
        STX_HIP(hipMemcpyAsync(xy_out, dxy_out, sizeof(float) * 2 * m, hipMemcpyDeviceToHost, s));
        STX_HIP(hipMemcpyAsync(score_out, dscore, sizeof(float) * m, hipMemcpyDeviceToHost, s));
        STX_HIP(hipMemcpyAsync(index_out, dindex, sizeof(int32_t) * m, hipMemcpyDeviceToHost, s));
        STX_HIP(hipStreamSynchronize(s));
    }
    return STX_OK;
}

extern "C" int stx_cluster_features_dev(stx_ctx *ctx, const float *d_xy, const float *d_score, int n, int window,
                                        int32_t *d_index_out, int *d_count)
{
    if (!ctx || !d_count || n < 0) return STX_ERR_INVALID_ARG;
    if (n > 0 && (!d_xy || !d_score || !d_index_out)) return STX_ERR_INVALID_ARG;
    STX_ENTER(ctx);
    return stx_launch_cluster(ctx, d_xy, d_score, n, (float)window, d_index_out, d_count);
}

extern "C" int stx_cluster_features(stx_ctx *ctx, const float *xy, const float *score, int n, int window,
                                    int32_t *index_out, int *count)
{
    if (!ctx || !count || n < 0) return STX_ERR_INVALID_ARG;
    if (n > 0 && (!xy || !score || !index_out)) return STX_ERR_INVALID_ARG;
    *count = 0;
    if (n == 0) return STX_OK;
    STX_ENTER(ctx);
    int err;
    hipStream_t s = ctx->stream;
    // [xy n*2][score n][index n][count]
    float *dxy = (float *)stx_workspace(ctx, STX_SLOT_SCORE_IO, sizeof(float) * (size_t)n * 4 + 16, &err);
    if (!dxy) return err;
    float *dscore = dxy + 2 * (size_t)n;
    int32_t *dindex = reinterpret_cast<int32_t *>(dscore + n);
    int *dcount = dindex + n;
    STX_HIP(hipMemcpyAsync(dxy, xy, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice, s));
    STX_HIP(hipMemcpyAsync(dscore, score, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, s));
    int rc = stx_launch_cluster(ctx, dxy, dscore, n, (float)window, dindex, dcount);
    if (rc) return rc;
    STX_HIP(hipMemcpyAsync(count, dcount, sizeof(int), hipMemcpyDeviceToHost, s));
    STX_HIP(hipStreamSynchronize(s));
    if (*count > 0) {
        STX_HIP(hipMemcpyAsync(index_out, dindex, sizeof(int32_t) * (size_t)*count, hipMemcpyDeviceToHost, s));
        STX_HIP(hipStreamSynchronize(s));
    }
    return STX_OK;
}

// ------------------------------------------------------------ descriptors --

extern "C" int stx_sift_descriptors_dev(stx_ctx *ctx, const uint8_t *d_gx, const uint8_t *d_gy, int w, int h,
                                        size_t stride, const float *d_xy, int n, float *d_angle, float *d_desc)
{
    if (!ctx || n < 0) return STX_ERR_INVALID_ARG;
    if (n > 0 && (!d_xy || !d_angle || !d_desc)) return STX_ERR_INVALID_ARG;
    int rc = check_image(d_gx, w, h, stride);
    if (rc) return rc;
    rc = check_image(d_gy, w, h, stride);
    if (rc) return rc;
    STX_ENTER(ctx);
    return stx_launch_sift(ctx, d_gx, d_gy, w, h, stride, d_xy, n, d_angle, d_desc);
}

extern "C" int stx_sift_descriptors(stx_ctx *ctx, const uint8_t *gx, const uint8_t *gy, int w, int h,
                                    size_t stride, const float *xy, int n, float *angle, float *desc)
{
    if (!ctx || n < 0) return STX_ERR_INVALID_ARG;
    if (n > 0 && (!xy || !angle || !desc)) return STX_ERR_INVALID_ARG;
    int rc = check_image(gx, w, h, stride);
    if (rc) return rc;
    rc = check_image(gy, w, h, stride);
    if (rc) return rc;
    if (n == 0) return STX_OK;
    STX_ENTER(ctx);
    int err;
    hipStream_t s = ctx->stream;
    const size_t p = (size_t)(w > 0 ? w : 1), img = (p * (size_t)(h > 0 ? h : 1) + 15) & ~(size_t)15;
    // [gx][gy][xy n*2][angle n][desc n*128]
    uint8_t *base = (uint8_t *)stx_workspace(ctx, STX_SLOT_SCORE_IO, 2 * img + sizeof(float) * (size_t)n * 131, &err);
    if (!base) return err;
    uint8_t *dgx = base, *dgy = base + img;
    float *dxy = reinterpret_cast<float *>(base + 2 * img);
    float *dangle = dxy + 2 * (size_t)n;
    float *ddesc = dangle + n;
    if (w > 0 && h > 0) {
        STX_HIP(copy_image_h2d(dgx, p, gx, stride, (size_t)w, (size_t)h, s));
        STX_HIP(copy_image_h2d(dgy, p, gy, stride, (size_t)w, (size_t)h, s));
    }
    STX_HIP(hipMemcpyAsync(dxy, xy, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice, s));
    STX_HIP(hipMemcpyAsync(dangle, angle, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, s));
    rc = stx_launch_sift(ctx, dgx, dgy, w, h, p, dxy, n, dangle, ddesc);
    if (rc) return rc;
    STX_HIP(hipMemcpyAsync(angle, dangle, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, s));
    STX_HIP(hipMemcpyAsync(desc, ddesc, sizeof(float) * 128 * (size_t)n, hipMemcpyDeviceToHost, s));
    STX_HIP(hipStreamSynchronize(s));
    return STX_OK;
}

// ------------------------------------------------------ hypothesis scoring --

static int check_ransac_args(stx_ctx *ctx, const float *hyp, int n_hyp, const float *K1, const float *K2,
                             const float *p1, const float *p2, int n, const int32_t *inliers, const float *err_sum)
{
    if (!ctx || n_hyp < 0 || n < 0 || n_hyp > 65535) return STX_ERR_INVALID_ARG;
    if (n_hyp > 0 && (!hyp || !K1 || !K2 || !inliers || !err_sum)) return STX_ERR_INVALID_ARG;
    if (n_hyp > 0 && n > 0 && (!p1 || !p2)) return STX_ERR_INVALID_ARG;
    return STX_OK;
}

extern "C" int stx_ransac_score_dev(stx_ctx *ctx, const float *hyp, int n_hyp, const float K1[9],
                                    const float K2[9], const float *d_p1, const float *d_p2, int n,
                                    const int32_t *exclude, float threshold, int32_t *d_inliers,
                                    float *d_err_sum, float *d_errors, float *d_depths)
{
    int rc = check_ransac_args(ctx, hyp, n_hyp, K1, K2, d_p1, d_p2, n, d_inliers, d_err_sum);
    if (rc) return rc;
    STX_ENTER(ctx);
    return stx_launch_ransac_score(ctx, hyp, n_hyp, K1, K2, d_p1, d_p2, n, exclude, threshold, d_inliers,
                                   d_err_sum, d_errors, d_depths);
}

extern "C" int stx_inlier_sum_dev(stx_ctx *ctx, const float *d_errors, int n_hyp, int n, float threshold,
                                  int32_t *d_inliers, float *d_err_sum)
{
    if (!ctx || n_hyp < 0 || n < 0 || n_hyp > 65535) return STX_ERR_INVALID_ARG;
    if (n_hyp > 0 && (!d_inliers || !d_err_sum || (n > 0 && !d_errors))) return STX_ERR_INVALID_ARG;
    STX_ENTER(ctx);
    return stx_launch_inlier_sum(ctx, d_errors, n_hyp, n, threshold, d_inliers, d_err_sum);
}

extern "C" int stx_ransac_score(stx_ctx *ctx, const float *hyp, int n_hyp, const float K1[9], const float K2[9],
                                const float *p1, const float *p2, int n, const int32_t *exclude,
                                float threshold, int32_t *inliers, float *err_sum, float *errors, float *depths)
{
    int rc = check_ransac_args(ctx, hyp, n_hyp, K1, K2, p1, p2, n, inliers, err_sum);
    if (rc) return rc;
    if (n_hyp == 0) return STX_OK;
    STX_ENTER(ctx);
    int err;
    hipStream_t s = ctx->stream;
    const size_t np = (size_t)n, nh = (size_t)n_hyp, cells = nh * np;
    // [p1 n*2][p2 n*2][err_sum H][inliers H][errors H*n][depths H*n*2]
    const size_t floats = 4 * np + 2 * nh + (errors ? cells : 0) + (depths ? 2 * cells : 0);
    float *base = (float *)stx_workspace(ctx, STX_SLOT_SCORE_IO, sizeof(float) * floats, &err);
    if (!base) return err;
    float *dp1 = base, *dp2 = base + 2 * np, *dsum = base + 4 * np;
    int32_t *dinl = reinterpret_cast<int32_t *>(dsum + nh);
    float *derr = errors ? dsum + 2 * nh : nullptr;
    float *ddep = depths ? dsum + 2 * nh + (errors ? cells : 0) : nullptr;
    if (n > 0) {
        STX_HIP(hipMemcpyAsync(dp1, p1, sizeof(float) * 2 * np, hipMemcpyHostToDevice, s));
        STX_HIP(hipMemcpyAsync(dp2, p2, sizeof(float) * 2 * np, hipMemcpyHostToDevice, s));
    }
    rc = stx_launch_ransac_score(ctx, hyp, n_hyp, K1, K2, dp1, dp2, n, exclude, threshold, dinl, dsum, derr, ddep);
    if (rc) return rc;
    STX_HIP(hipMemcpyAsync(inliers, dinl, sizeof(int32_t) * nh, hipMemcpyDeviceToHost, s));
    STX_HIP(hipMemcpyAsync(err_sum, dsum, sizeof(float) * nh, hipMemcpyDeviceToHost, s));
    if (errors && cells) STX_HIP(hipMemcpyAsync(errors, derr, sizeof(float) * cells, hipMemcpyDeviceToHost, s));
    if (depths && cells) STX_HIP(hipMemcpyAsync(depths, ddep, sizeof(float) * 2 * cells, hipMemcpyDeviceToHost, s));
    STX_HIP(hipStreamSynchronize(s));
    return STX_OK;
}
