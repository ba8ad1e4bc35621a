// Several GPUs behind one handle, for a single-process host program such as the reference's
// main(): the drop-in cannot start one process per GPU, so the library shards
// the work itself -- one context, one stream and one host thread per device, no collective.
//   batch of pairs     -> contiguous shares of whole pairs            (stx_disparity_batch_u8)
//   one large pair     -> row bands + win/2 halo rows per device       (stx_disparity_rowband_*)
//   matching           -> list1's queries sharded, list2 replicated    (stx_match_multi)
//   one warp           -> destination row bands, source replicated     (stx_rectify_multi_u8)
// Inputs and outputs are host buffers: every device uploads exactly the rows it needs from the
// caller's memory and writes its share of the result straight back to the caller's output, so
// there is no device-to-device step to pay for (the one-process-per-GPU form with an RCCL
// gather lives in stereo_amd/dist.py).
#include <cstdlib>
#include <new>
#include <thread>

#include "stx_internal.h"

struct stx_multi {
    int n;
    stx_ctx *ctx[STX_MULTI_MAX_DEVICES];
    hipStream_t stream[STX_MULTI_MAX_DEVICES];
};

extern "C" void stx_multi_destroy(stx_multi *m)
{
    if (!m) return;
    for (int k = 0; k < m->n; ++k) {
        if (!m->ctx[k]) continue;
        {
            stx_device_guard guard(m->ctx[k]->device);
            if (m->stream[k]) {
                (void)hipStreamSynchronize(m->stream[k]);
                (void)stx_set_stream(m->ctx[k], nullptr);
                (void)hipStreamDestroy(m->stream[k]);
            }
        }
        stx_destroy(m->ctx[k]);
    }
    delete m;
}

extern "C" int stx_multi_create(stx_multi **out, const int *device_ids, int n_devices)
{
    if (!out || n_devices < 0 || n_devices > STX_MULTI_MAX_DEVICES) return STX_ERR_INVALID_ARG;
    *out = nullptr;
    int have = 0;
    hipError_t e = hipGetDeviceCount(&have);
    if (e != hipSuccess) return (int)e;
    if (n_devices == 0) n_devices = have < STX_MULTI_MAX_DEVICES ? have : STX_MULTI_MAX_DEVICES;
    if (n_devices == 0) return STX_ERR_NO_DEVICE;
    // Test-only switch: STX_MULTI_ALLOW_REPEAT=1 lets a device id appear more than once, so the
    // threaded n >= 2 path (shares, halo uploads, one staging pool per context) runs on a one-GPU
    // box.  Two contexts on one device are independent (own stream, own workspace slots).
    const char *rep = getenv("STX_MULTI_ALLOW_REPEAT");
    const bool allow_repeat = rep && rep[0] == '1' && rep[1] == 0;
    for (int k = 0; k < n_devices; ++k) {
        const int d = device_ids ? device_ids[k] : k;
        if (d < 0 || d >= have) return STX_ERR_NO_DEVICE;
        for (int j = 0; j < k && !allow_repeat; ++j)
            if ((device_ids ? device_ids[j] : j) == d) return STX_ERR_INVALID_ARG;  // a device twice
    }
    stx_multi *m = new (std::nothrow) stx_multi();
    if (!m) return STX_ERR_HOST_OOM;
    m->n = n_devices;
    for (int k = 0; k < STX_MULTI_MAX_DEVICES; ++k) {
        m->ctx[k] = nullptr;
        m->stream[k] = nullptr;
    }
    for (int k = 0; k < n_devices; ++k) {
        const int d = device_ids ? device_ids[k] : k;
        int rc = stx_create(&m->ctx[k], d);
        if (rc == STX_OK) {
            // a stream of its own per device: nothing serialises on a NULL stream
            stx_device_guard guard(d);
            rc = guard.err ? guard.err : (int)hipStreamCreateWithFlags(&m->stream[k], hipStreamNonBlocking);
            if (rc == STX_OK) rc = stx_set_stream(m->ctx[k], m->stream[k]);
            // the devices' staging threads share the host: keep each context's share small
            if (rc == STX_OK && n_devices > 1)
                rc = stx_set_host_pipeline(m->ctx[k], 1, n_devices >= 8 ? 2 : n_devices >= 4 ? 3 : 4, 0);
        }
        if (rc != STX_OK) {
            stx_multi_destroy(m);
            return rc;
        }
    }
    *out = m;
    return STX_OK;
}

extern "C" int stx_multi_size(const stx_multi *m) { return m ? m->n : STX_ERR_INVALID_ARG; }

extern "C" stx_ctx *stx_multi_context(stx_multi *m, int k)
{
    return (m && k >= 0 && k < m->n) ? m->ctx[k] : nullptr;
}

namespace {

// [lo, hi) share of n units for part k of parts (contiguous, balanced)
inline void share(int n, int k, int parts, int *lo, int *hi)
{
    const int base = n / parts, rem = n % parts;
    *lo = k * base + (k < rem ? k : rem);
    *hi = *lo + base + (k < rem ? 1 : 0);
}

// Runs fn(k) for every device: device 0's share on the calling thread, the others on a
// thread each.  Returns the first non-zero status.
template <class F>
int for_each_device(stx_multi *m, F fn)
{
    int rc[STX_MULTI_MAX_DEVICES] = {0};
    std::thread th[STX_MULTI_MAX_DEVICES];
    bool started[STX_MULTI_MAX_DEVICES] = {false};
    for (int k = 1; k < m->n; ++k) {
        try {
            th[k] = std::thread([&rc, &fn, k] { rc[k] = fn(k); });
            started[k] = true;
        } catch (...) {  // no thread to be had: this device's share runs here, after device 0's
        }
    }
    rc[0] = fn(0);
    for (int k = 1; k < m->n; ++k) {
        if (started[k])
            th[k].join();
        else
            rc[k] = fn(k);
    }
    for (int k = 0; k < m->n; ++k)
        if (rc[k]) return rc[k];
    return STX_OK;
}

int check_multi_disparity(const stx_multi *m, const void *img0, const void *img1, const void *out, int w, int h,
                          size_t stride, int win, int max_disp, int cost_mode, int out_bytes, size_t out_stride)
{
    if (!m || w < 0 || h < 0 || max_disp < 0) return STX_ERR_INVALID_ARG;
    if (win < 1 || (win & 1) == 0) return STX_ERR_INVALID_ARG;
    if (cost_mode != STX_COST_SAD && cost_mode != STX_COST_SSD) return STX_ERR_INVALID_ARG;
    if (out_bytes != 1 && out_bytes != 2) return STX_ERR_INVALID_ARG;
    if (stride < (size_t)w || out_stride < (size_t)w * (size_t)out_bytes) return STX_ERR_INVALID_ARG;
    if (out_bytes == 2 && (out_stride & 1)) return STX_ERR_INVALID_ARG;
    if (max_disp == 0 && win != 1) return STX_ERR_UNSUPPORTED;
    if (win > 31) return STX_ERR_UNSUPPORTED;
    if (w > 0 && h > 0 && (!img0 || !img1 || !out)) return STX_ERR_INVALID_ARG;
    return STX_OK;
}

}  // namespace

extern "C" int stx_disparity_batch_u8(stx_multi *m, const uint8_t *img0, const uint8_t *img1, int n_pairs, int w,
                                      int h, size_t stride, size_t pair_stride, int win, int max_disp,
                                      int cost_mode, uint8_t *out, size_t out_stride, size_t out_pair_stride)
{
    if (n_pairs < 0) return STX_ERR_INVALID_ARG;
    int rc = check_multi_disparity(m, img0, img1, out, w, h, stride, win, max_disp, cost_mode, 1, out_stride);
    if (rc) return rc;
    if (n_pairs == 0 || w == 0 || h == 0) return STX_OK;
    if (n_pairs > 1 && (pair_stride < stride * (size_t)h || out_pair_stride < out_stride * (size_t)h))
        return STX_ERR_INVALID_ARG;
    return for_each_device(m, [=](int k) -> int {
        int lo, hi;
        share(n_pairs, k, m->n, &lo, &hi);
        if (hi <= lo) return STX_OK;
        stx_ctx *ctx = m->ctx[k];
        STX_ENTER(ctx);
        int r = stx_disparity_host(ctx, img0 + (size_t)lo * pair_stride, img1 + (size_t)lo * pair_stride, hi - lo, w,
                                   h, stride, pair_stride, 0, h, win, max_disp, cost_mode,
                                   out + (size_t)lo * out_pair_stride, out_stride, out_pair_stride, 1);
        if (r) return r;
        return (int)hipStreamSynchronize(ctx->stream);
    });
}

static int rowband_multi(stx_multi *m, const uint8_t *img0, const uint8_t *img1, int w, int h, size_t stride, int win,
                         int max_disp, int cost_mode, void *out, size_t out_stride, int out_bytes)
{
    int rc = check_multi_disparity(m, img0, img1, out, w, h, stride, win, max_disp, cost_mode, out_bytes, out_stride);
    if (rc) return rc;
    if (w == 0 || h == 0) return STX_OK;
    return for_each_device(m, [=](int k) -> int {
        int lo, hi;
        share(h, k, m->n, &lo, &hi);
        if (hi <= lo) return STX_OK;
        stx_ctx *ctx = m->ctx[k];
        STX_ENTER(ctx);
        // rows [lo, hi) of the map; stx_disparity_host uploads them plus the win/2 halo rows the
        // image has on either side, straight from the caller's buffers
        int r = stx_disparity_host(ctx, img0, img1, 1, w, h, stride, 0, lo, hi - lo, win, max_disp, cost_mode,
                                   static_cast<uint8_t *>(out) + (size_t)lo * out_stride, out_stride, 0, out_bytes);
        if (r) return r;
        return (int)hipStreamSynchronize(ctx->stream);
    });
}

extern "C" int stx_disparity_rowband_u8(stx_multi *m, const uint8_t *img0, const uint8_t *img1, int w, int h,
                                        size_t stride, int win, int max_disp, int cost_mode, uint8_t *out,
                                        size_t out_stride)
{
    return rowband_multi(m, img0, img1, w, h, stride, win, max_disp, cost_mode, out, out_stride, 1);
}

extern "C" int stx_disparity_rowband_u16(stx_multi *m, const uint8_t *img0, const uint8_t *img1, int w, int h,
                                         size_t stride, int win, int max_disp, int cost_mode, uint16_t *out,
                                         size_t out_stride)
{
    return rowband_multi(m, img0, img1, w, h, stride, win, max_disp, cost_mode, out, out_stride, 2);
}

extern "C" int stx_match_multi(stx_multi *m, const float *desc1, const float *score1, int n1, const float *desc2,
                               const float *score2, int n2, float dist_limit, int32_t *closest, int32_t *second,
                               float *min_dist, float *second_dist, uint8_t *status)
{
    if (!m || n1 < 0 || n2 < 0) return STX_ERR_INVALID_ARG;
    if (n1 > 0 && (!desc1 || !score1)) return STX_ERR_INVALID_ARG;
    if (n2 > 0 && (!desc2 || !score2)) return STX_ERR_INVALID_ARG;
    if (n1 == 0) return STX_OK;
    return for_each_device(m, [=](int k) -> int {
        int lo, hi;
        share(n1, k, m->n, &lo, &hi);
        if (hi <= lo) return STX_OK;
        // queries are independent: a share's results are rows [lo, hi) of the whole result
        return stx_match(m->ctx[k], desc1 + (size_t)lo * STX_DESC_LEN, score1 + lo, hi - lo, desc2, score2, n2,
                         dist_limit, closest ? closest + lo : nullptr, second ? second + lo : nullptr,
                         min_dist ? min_dist + lo : nullptr, second_dist ? second_dist + lo : nullptr,
                         status ? status + lo : nullptr);
    });
}

extern "C" int stx_rectify_multi_u8(stx_multi *m, const uint8_t *src, int src_w, int src_h, size_t src_stride,
                                    uint8_t *dst, int dst_w, int dst_h, size_t dst_stride, const float Hinv[9],
                                    int flags)
{
    if (!m || !Hinv || src_w < 0 || src_h < 0 || dst_w < 0 || dst_h < 0) return STX_ERR_INVALID_ARG;
    if (flags != STX_RECT_PRESERVE_DST && flags != STX_RECT_DST_IS_ZERO) return STX_ERR_INVALID_ARG;
    if (src_w > 0 && src_h > 0 && (!src || src_stride < (size_t)src_w)) return STX_ERR_INVALID_ARG;
    if (dst_w > 0 && dst_h > 0 && (!dst || dst_stride < (size_t)dst_w)) return STX_ERR_INVALID_ARG;
    if (dst_w == 0 || dst_h == 0) return STX_OK;
    return for_each_device(m, [=](int k) -> int {
        int lo, hi;
        share(dst_h, k, m->n, &lo, &hi);
        if (hi <= lo) return STX_OK;
        stx_ctx *ctx = m->ctx[k];
        STX_ENTER(ctx);
        // the warp is output-pixel-local: this device fills destination rows [lo, hi) from its own
        // copy of the whole source; no halo, no exchange
        int err;
        hipStream_t s = ctx->stream;
        const size_t sp = (size_t)(src_w > 0 ? src_w : 1), dp = (size_t)dst_w;
        const size_t rows = (size_t)(hi - lo);
        uint8_t *dsrc = (uint8_t *)stx_workspace(ctx, STX_SLOT_IMG0, sp * (size_t)(src_h > 0 ? src_h : 1), &err);
        if (!dsrc) return err;
        uint8_t *ddst = (uint8_t *)stx_workspace(ctx, STX_SLOT_IMG1, dp * rows, &err);
        if (!ddst) return err;
        if (src_w > 0 && src_h > 0)
            STX_HIP(hipMemcpy2DAsync(dsrc, sp, src, src_stride, (size_t)src_w, (size_t)src_h, hipMemcpyHostToDevice, s));
        uint8_t *band = dst + (size_t)lo * dst_stride;
        if (flags == STX_RECT_PRESERVE_DST)
            STX_HIP(hipMemcpy2DAsync(ddst, dp, band, dst_stride, dp, rows, hipMemcpyHostToDevice, s));
        int r = stx_rectify_rowband_u8_dev(ctx, dsrc, src_w, src_h, sp, ddst, dst_w, lo, hi - lo, dp, Hinv, flags);
        if (r) return r;
        STX_HIP(hipMemcpy2DAsync(band, dst_stride, ddst, dp, dp, rows, hipMemcpyDeviceToHost, s));
        return (int)hipStreamSynchronize(s);
    });
}
