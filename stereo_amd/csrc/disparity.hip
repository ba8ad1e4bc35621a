// Dense disparity scan for gfx950.
//
//   disp_ref_kernel   reference mode (ComputeDepthImage, stereo/Stereography.cpp:649-691):
//                     1x1 window, every column of the row is a candidate.
//   disp_ext_kernel   D1-ext (this library's generalisation): win x win
//                     SAD/SSD, candidates x1 in [max(0,x-D+1), x].
//
// All arithmetic is integer; results are bit-exact against oracle/stereo_oracle.c.
#include "stx_internal.h"

// Scratch of stx_disparity_dev_noalloc (include/stereo_hip.h): the per-pixel keys a scan of
// more than 256 disparities carries between its passes.  Every GPU this library runs on holds
// its own copy of the array until the process ends.  In host code the NAME is only a stand-in
// for hipGetSymbolAddress; kernels and copies always get the per-device address it returns.
static __device__ __align__(256) unsigned disp_scratch[STX_NOALLOC_MAX_PIXELS];

int stx_disparity_scratch(void **ptr, size_t *bytes)
{
    // the caller (stx_create) has made the context's device current
    void *p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(disp_scratch));
    if (e != hipSuccess) return (int)e;
    if (!p) return STX_ERR_NO_DEVICE;
    *ptr = p;
    *bytes = sizeof(unsigned) * (size_t)STX_NOALLOC_MAX_PIXELS;
    return STX_OK;
}

namespace {

// ---------------------------------------------------------------------------
// Reference mode.
//
// One workgroup = one image row x REF_PX*256 consecutive pixels.  The row of img1
// is staged through LDS in chunks as a table of 32-bit "candidate words"
//     T[x1] = img1[y][x1] != 0 ? img1[y][x1] << 20 : 0xFFF00000
// and each lane keeps, per owned pixel, a packed key (|p1-p0| << 20) + x1 whose
// running unsigned minimum is exactly the reference's scan: smallest distance
// first, and for equal distances the smallest x1 (strict '<' at
// Stereography.cpp:677 keeps the first).  One candidate costs one v_sad_u32
// (|a-b|+c in one instruction, the candidate word is a broadcast LDS read and
// x1 a scalar) and half a v_min3_u32.  The zero-skip at :676 is the 0xFFF00000
// word, which can never beat the initial key 255<<20 (:671-672).
// ---------------------------------------------------------------------------

constexpr int REF_THREADS = 256;
constexpr int REF_PX = 4;        // pixels per lane
constexpr int REF_CHUNK = 2048;  // candidate words per LDS chunk (8 KiB)
constexpr unsigned REF_SHIFT = 20;
constexpr unsigned REF_SKIP = 0xFFF00000u;
constexpr unsigned REF_INIT = 255u << REF_SHIFT;

__device__ __forceinline__ unsigned absdiff_add(unsigned a, unsigned b, unsigned c)
{
    // pattern-matched by hipcc to v_sad_u32
    return (a > b ? a - b : b - a) + c;
}

__device__ __forceinline__ unsigned umin3(unsigned a, unsigned b, unsigned c)
{
    return min(a, min(b, c));
}

__global__ __launch_bounds__(REF_THREADS) void disp_ref_kernel(
    const uint8_t *__restrict__ img0, const uint8_t *__restrict__ img1, int w, size_t stride,
    size_t pair_stride, int row0, uint8_t *__restrict__ out, size_t out_stride,
    size_t out_pair_stride, int out_bytes)
{
    __shared__ uint4 tab4[REF_CHUNK / 4];
    unsigned *tab = reinterpret_cast<unsigned *>(tab4);

    const int y = row0 + blockIdx.y;
    const size_t pair = blockIdx.z;
    const uint8_t *r0 = img0 + pair * pair_stride + (size_t)y * stride;
    const uint8_t *r1 = img1 + pair * pair_stride + (size_t)y * stride;
    const int xbase = blockIdx.x * (REF_THREADS * REF_PX);

    unsigned a[REF_PX], best[REF_PX];
#pragma unroll
    for (int p = 0; p < REF_PX; ++p) {
        const int x = xbase + p * REF_THREADS + threadIdx.x;
        const unsigned p0 = x < w ? r0[x] : 0u;
        a[p] = p0 << REF_SHIFT;
        best[p] = REF_INIT;
    }

    for (int c0 = 0; c0 < w; c0 += REF_CHUNK) {
        const int n = min(REF_CHUNK, w - c0);
        const int n4 = (n + 3) & ~3;
        __syncthreads();
        for (int i = threadIdx.x; i < n4; i += REF_THREADS) {
            const unsigned p1 = i < n ? r1[c0 + i] : 0u;
            tab[i] = p1 ? (p1 << REF_SHIFT) : REF_SKIP;
        }
        __syncthreads();
        for (int i = 0; i < n4; i += 4) {
            const uint4 t = tab4[i >> 2];  // same address in every lane: LDS broadcast
            // the four candidate columns as scalars; the empty asm keeps hipcc from
            // folding the +1..+3 into a separate vector add after the |a-b|
            unsigned x1 = (unsigned)(c0 + i), x1b = x1 + 1, x1c = x1 + 2, x1d = x1 + 3;
            asm volatile("" : "+s"(x1), "+s"(x1b), "+s"(x1c), "+s"(x1d));
#pragma unroll
            for (int p = 0; p < REF_PX; ++p) {
                const unsigned k0 = absdiff_add(a[p], t.x, x1);
                const unsigned k1 = absdiff_add(a[p], t.y, x1b);
                const unsigned k2 = absdiff_add(a[p], t.z, x1c);
                const unsigned k3 = absdiff_add(a[p], t.w, x1d);
                best[p] = umin3(best[p], k0, k1);
                asm volatile("" : "+v"(best[p]));  // keep two v_min3_u32 (else min,min,min3)
                best[p] = umin3(best[p], k2, k3);
            }
        }
    }

    uint8_t *orow = out + pair * out_pair_stride + (size_t)blockIdx.y * out_stride;
#pragma unroll
    for (int p = 0; p < REF_PX; ++p) {
        const int x = xbase + p * REF_THREADS + threadIdx.x;
        if (x < w) {
            const int xbest = (int)(best[p] & ((1u << REF_SHIFT) - 1));
            int v = xbest - x;
            v = v < 0 ? -v : v;
            if (a[p] == 0) v = 0;  // pixel0 == 0: the reference leaves Mat::zeros (:670)
            if (out_bytes == 1)
                orow[x] = (uint8_t)(v & 0xFF);  // int -> uchar at :686
            else
                reinterpret_cast<uint16_t *>(orow)[x] = (uint16_t)(v & 0xFFFF);
        }
    }
}

// ---------------------------------------------------------------------------
// D1-ext, first version: every (pixel, candidate) cost is summed tap by tap
// from LDS tiles.  Correct for every supported parameter combination; used for
// the parameter ranges the sliding-sum kernel does not cover.
// ---------------------------------------------------------------------------

constexpr int EXT_TX = 64;  // tile columns
constexpr int EXT_TY = 16;  // tile rows
constexpr int EXT_THREADS = 256;

template <bool SSD>
__global__ __launch_bounds__(EXT_THREADS) void disp_ext_bruteforce_kernel(
    const uint8_t *__restrict__ img0, const uint8_t *__restrict__ img1, int w, int h,
    size_t stride, size_t pair_stride, int row0, int rows, int r, int max_disp,
    uint8_t *__restrict__ out, size_t out_stride, size_t out_pair_stride, int out_bytes)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lw = EXT_TX + 2 * r;                 // L tile width
    const int rw = EXT_TX + 2 * r + max_disp - 1;  // R tile width
    const int th = EXT_TY + 2 * r;
    uint8_t *Lt = smem;
    uint8_t *Rt = smem + (size_t)lw * th;

    const size_t pair = blockIdx.z;
    const uint8_t *i0 = img0 + pair * pair_stride;
    const uint8_t *i1 = img1 + pair * pair_stride;
    const int x0 = blockIdx.x * EXT_TX;
    const int y0 = row0 + blockIdx.y * EXT_TY;

    // stage tiles; both images are zero-extended outside their borders
    for (int i = threadIdx.x; i < lw * th; i += EXT_THREADS) {
        const int ty = i / lw, tx = i - ty * lw;
        const int gy = y0 - r + ty, gx = x0 - r + tx;
        Lt[i] = (gy >= 0 && gy < h && gx >= 0 && gx < w) ? i0[(size_t)gy * stride + gx] : 0;
    }
    for (int i = threadIdx.x; i < rw * th; i += EXT_THREADS) {
        const int ty = i / rw, tx = i - ty * rw;
        const int gy = y0 - r + ty, gx = x0 - r - (max_disp - 1) + tx;
        Rt[i] = (gy >= 0 && gy < h && gx >= 0 && gx < w) ? i1[(size_t)gy * stride + gx] : 0;
    }
    __syncthreads();

    const int lx = threadIdx.x & (EXT_TX - 1);
    const int x = x0 + lx;
    if (x >= w) return;
    const int win = 2 * r + 1;
    const unsigned init = (SSD ? 65025u : 255u) * (unsigned)win * (unsigned)win;

    for (int ly = threadIdx.x / EXT_TX; ly < EXT_TY; ly += EXT_THREADS / EXT_TX) {
        const int y = y0 + ly;
        if (y >= h || y >= row0 + rows) break;
        int v = 0;
        const unsigned p0 = Lt[(ly + r) * lw + lx + r];
        if (p0 != 0) {
            unsigned best = init;
            int xbest = 0;
            const int dmax = min(max_disp - 1, x);
            for (int d = dmax; d >= 0; --d) {  // ascending x1
                // R tile column of image column c: c - (x0 - r - (max_disp-1))
                const int rc = lx + r + (max_disp - 1) - d;  // column of x1 = x - d
                if (Rt[(ly + r) * rw + rc] == 0) continue;
                unsigned cost = 0;
                // taps outside either image read the zero fill of the tiles
                for (int dy = -r; dy <= r; ++dy) {
                    const uint8_t *lrow = Lt + (ly + r + dy) * lw + lx + r;
                    const uint8_t *rrow = Rt + (ly + r + dy) * rw + rc;
                    for (int dx = -r; dx <= r; ++dx) {
                        const int a = lrow[dx], b = rrow[dx];
                        const unsigned ad = (unsigned)(a > b ? a - b : b - a);
                        cost += SSD ? ad * ad : ad;
                    }
                }
                if (cost < best) {
                    best = cost;
                    xbest = x - d;
                }
            }
            v = x - xbest;  // xbest <= x
        }
        uint8_t *orow = out + pair * out_pair_stride + (size_t)(y - row0) * out_stride;
        if (out_bytes == 1)
            orow[x] = (uint8_t)(v & 0xFF);
        else
            reinterpret_cast<uint16_t *>(orow)[x] = (uint16_t)(v & 0xFFFF);
    }
}


// ---------------------------------------------------------------------------
// D1-ext, sliding-sum kernel (SAD, window radius 0..7; SSD radius 1..5; any max_disp).
//
// Work decomposition
//   workgroup = one tile of 256 image columns x one band of `bh` output rows;
//   wave w of the workgroup owns the 16 disparities d in [dbase+16w, dbase+16w+16);
//   lane l of a wave owns the 4 adjacent columns x0+4l .. x0+4l+3.
// Rows are streamed top to bottom through an LDS ring of band_ring(win) rows.  A ring
// row holds the img0 row (256 columns), the img1 row (256 + 16*nw columns to the
// left) and a candidate-validity mask row, all as 32-bit words with the pixel
// value pre-shifted by BAND_SH bits, so that every cost this kernel forms is
// already in the upper bits of its winner-take-all key.
//
// Per lane, per (d, column): V[d][c] is the running column sum of |img1 - img0|
// over the window's rows.  A new row is added with one v_sad_u32 (|a-b|+acc), the
// row leaving the window is subtracted (v_sad_u32 + v_sub); both operands come
// from registers filled by 128-bit LDS reads (one lane = 4 columns = one b128).
// The horizontal window sum slides across the lane's 4 columns and takes the
// columns it lacks from the neighbouring lanes with wave_shr:1 / wave_shl:1 DPP
// adds (window radius <= 4 => one lane to each side; radius 5..7 => two, with chained hops).  The key
//     (cost << BAND_SH) + (4095 - d) + mask(img1[y][x-d] == 0 or x-d < 0)
// is reduced with v_min3_u32 over the wave's 16 disparities, then across the
// workgroup's waves with one LDS atomic min per pixel: smallest cost wins, equal
// costs go to the larger d = the lower x1, which is the reference's strict-'<'
// first-wins order for an ascending-x1 scan.
// ---------------------------------------------------------------------------

// LDS ring rows for a window: win + 2 of them are live at once; 16 (a mask instead of a modulo)
// up to 13x13, 20 for 15x15 (32 rows would not fit the LDS next to 16 waves' img1 columns)
constexpr int band_ring(int win) { return win + 2 <= 16 ? 16 : win + 2 <= 20 ? 20 : win + 2; }
template <int RING> __device__ __forceinline__ int band_slot(int yy) { return (yy + 4 * RING) % RING; }
// The mask rows are read only for the centre row of the window, win / 2 rows behind the entering
// one: the wide windows (17x17 and up), whose image rows need up to 33 ring rows, keep just
// win / 2 + 2 mask rows, which is what lets 31x31 search 256 disparities in one pass.
constexpr int band_mask_ring(int win) { return win > 15 ? win / 2 + 2 : band_ring(win); }
constexpr int BAND_ND = 16;          // disparities per wave
constexpr int BAND_COLS = 256;       // columns per tile (64 lanes x 4)
constexpr unsigned BAND_SH = 12;     // key = cost << 12 | (4095 - d)
// Valid keys stay below 2^28 (cost <= 255*81 < 2^15, shifted by 12).  An invalid candidate
// (img1 centre 0 or left of the image) adds BAND_MASKED, a disparity past D adds
// BAND_PAST_D; the two can add up without wrapping, and "no candidate yet" (BAND_INIT)
// lies above every valid key and below every invalidated one.
constexpr unsigned BAND_MASKED = 0x80000000u;
constexpr unsigned BAND_PAST_D = 0x40000000u;
constexpr unsigned BAND_INIT = 0x3FFFFFFFu;
constexpr int BAND_MAX_WAVES = 16;

// |a - b| + c in one instruction.  Written as asm because hipcc only matches the
// pattern when the result is used as a sum; the "row leaving the window" form
// (V -= |a-b|) otherwise becomes max, min and two subtractions.
__device__ __forceinline__ unsigned sad_u32(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned sad_u32_0(unsigned a, unsigned b)
{
    unsigned r;
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// sum of the four byte-wise |a - b| (+ c), and the same sum shifted into the upper half-word:
// with one pixel per dword (bytes 1..3 zero) these add |a - b| to the low or the high 16 bits
__device__ __forceinline__ unsigned sad_u8(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_sad_u8 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned sad_hi_u8(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_sad_hi_u8 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned sad_u8_0(unsigned a, unsigned b)
{
    unsigned r;
    asm("v_sad_u8 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned min3_u32(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// t*t + c and t*t for t < 2^24 (one instruction each)
__device__ __forceinline__ unsigned sq_acc_u24(unsigned t, unsigned c)
{
    unsigned r;
    asm("v_mad_u32_u24 %0, %1, %1, %2" : "=v"(r) : "v"(t), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned sq_u24(unsigned t)
{
    unsigned r;
    asm("v_mul_u32_u24 %0, %1, %1" : "=v"(r) : "v"(t));
    return r;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Five or six ds_read_b128 issued back to back, then ONE wait.  The empty asm keeps the
// 16-byte loads whole (without it the element extracts are scalarised and re-merged as
// ds_read2_b32 pairs); it takes all the loaded vectors at once, so the compiler may issue
// every read before the single s_waitcnt it needs (one asm per load made it wait after each
// read: six serialised LDS round trips per phase, right behind the row barrier where all 16
// waves of the workgroup stall together).
__device__ __forceinline__ void lds_read_5(const u32x4 *p, unsigned (&v)[20])
{
    u32x4 t0 = p[0], t1 = p[1], t2 = p[2], t3 = p[3], t4 = p[4];
    asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), "+v"(t4));
    v[0] = t0.x; v[1] = t0.y; v[2] = t0.z; v[3] = t0.w;
    v[4] = t1.x; v[5] = t1.y; v[6] = t1.z; v[7] = t1.w;
    v[8] = t2.x; v[9] = t2.y; v[10] = t2.z; v[11] = t2.w;
    v[12] = t3.x; v[13] = t3.y; v[14] = t3.z; v[15] = t3.w;
    v[16] = t4.x; v[17] = t4.y; v[18] = t4.z; v[19] = t4.w;
}
__device__ __forceinline__ void lds_read_1_5(const u32x4 *q, const u32x4 *p, unsigned (&l)[4], unsigned (&v)[20])
{
    u32x4 tl = q[0], t0 = p[0], t1 = p[1], t2 = p[2], t3 = p[3], t4 = p[4];
    asm volatile("" : "+v"(tl), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), "+v"(t4));
    l[0] = tl.x; l[1] = tl.y; l[2] = tl.z; l[3] = tl.w;
    v[0] = t0.x; v[1] = t0.y; v[2] = t0.z; v[3] = t0.w;
    v[4] = t1.x; v[5] = t1.y; v[6] = t1.z; v[7] = t1.w;
    v[8] = t2.x; v[9] = t2.y; v[10] = t2.z; v[11] = t2.w;
    v[12] = t3.x; v[13] = t3.y; v[14] = t3.z; v[15] = t3.w;
    v[16] = t4.x; v[17] = t4.y; v[18] = t4.z; v[19] = t4.w;
}

__device__ __forceinline__ unsigned from_left(unsigned v)
{
    // lane l receives lane l-1's value (wave_shr:1); folded into the consuming add
    unsigned r = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);
    return r;
}
__device__ __forceinline__ unsigned from_right(unsigned v)
{
    // lane l receives lane l+1's value (wave_shl:1)
    unsigned r = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);
    return r;
}

// lane i of a 16-lane row receives lane i-N / i+N of the SAME row (row_shr:N / row_shl:N), zero
// from beyond the row: one instruction whatever the distance, and folded into the consuming add
template <int N> __device__ __forceinline__ unsigned row_from_left(unsigned v)
{
    static_assert(N >= 1 && N <= 15, "row_shr distance");
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + N, 0xf, 0xf, true);
}
template <int N> __device__ __forceinline__ unsigned row_from_right(unsigned v)
{
    static_assert(N >= 1 && N <= 15, "row_shl distance");
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + N, 0xf, 0xf, true);
}

struct BandArgs {
    const uint8_t *img0;
    const uint8_t *img1;
    int w, h;
    size_t stride, pair_stride;
    int row0, rows, bh;
    int D;      // total number of disparities
    int dbase;  // first disparity of this pass
    unsigned *keys;  // per-pixel best keys carried between passes (nullptr if one pass)
    int first_pass, last_pass;
    uint8_t *out;
    size_t out_stride, out_pair_stride;
    int out_bytes;
};

// SSD variant: a tap costs (a-b)^2 <= 65025, a 9x9 window < 2^23, so the key is
// (cost << 8) | (255 - (d - dbase)) with a pass-local disparity code; pixels are pre-shifted by
// 4 bits so that |a-b|^2 (v_sad_u32, v_mad_u32_u24) is born shifted by 8.  Invalid candidates
// and disparities past D set bit 31 (OR, so the two cannot wrap); passes are merged on the
// cost field alone (a later pass holds larger d and wins ties).
template <int R, bool SSD>
__global__ __launch_bounds__(BAND_MAX_WAVES * 64) void disp_band_kernel(BandArgs a)
{
    constexpr int WIN = 2 * R + 1;
    // PK (SAD windows): a window cost is below 2^15, so the column sums of two disparities, d
    // and d + 8, share one register as half-words (v_sad_u8 / v_sad_hi_u8 add a tap to either
    // half); the row-leaving subtraction and the whole horizontal slide then run once per PAIR
    // of disparities.  key = cost << 16 | x1 (the candidate's img1 column, carried by the mask row).  Disparities past D start from PK_PAST per
    // column, which lifts their costs above every valid one (PK_PAST > 255 * WIN) without
    // leaving the half-word (WIN * (PK_PAST + 255 * WIN) < 65536).
    // 13x13 and 15x15 (PKW): costs reach 255 * 225 < 2^16, so a key may pass 2^31 and neither the
    // bias nor bit 31 can mark an invalid candidate.  There the mask word saturates the cost field
    // (0xFFFF0000) and a disparity past D carries an all-ones code, both above every valid key.
    // WIDE (SAD windows 17x17 .. 31x31): the window reaches up to four lanes to each side, too far
    // for chained wave_shr:1 hops.  The wave is treated as four independent rows of 16 lanes, each
    // with its own halo lanes, because inside a row DPP row_shr:N / row_shl:N reach any distance in
    // one instruction: a row-wide prefix sum of the lanes' four-column totals gives the whole
    // lanes of a window as a difference of two fetched prefixes, the two partly covered lanes add
    // a fetched suffix / prefix, and the slide across the lane's four columns fetches one column
    // each way.  16 - 2 HL of a row's 16 lanes produce output (12, 10 or 8), the others only feed
    // their neighbours.  Costs reach 255 * 961 < 2^18: plain 32-bit column sums, key = cost << 12 |
    // code as for the 1x1 window.
    // SSDX (SSD, wide): a window cost needs 26 bits, so the 32-bit key holds cost << 4 | (15 - dd), the
    // order among the WAVE's 16 disparities only; the wave's winner is widened to cost << 32 |
    // (4095 - d) for the 64-bit LDS atomic min across the waves.
    constexpr bool WIDE = R > 7;
    constexpr bool SSDX = SSD && WIDE;
    constexpr bool PK = !SSD && R > 0 && !WIDE;
    constexpr bool PKW = PK && R > 5;
    constexpr unsigned PK_PAST = PKW ? 0u : WIN <= 9 ? 3584u : 3000u;
    constexpr unsigned PK_MASKED = PKW ? 0xFFFF0000u : BAND_MASKED;
    static_assert(!PK || PKW || (PK_PAST > 255u * WIN && WIN * (PK_PAST + 255u * WIN) < 65536u), "half-word costs");
    static_assert(!PKW || 255u * WIN * WIN < 65535u, "half-word costs");
    // SSD 13x13 and 15x15 (SSDW): a window cost needs 24 bits, so cost << 8 | code fills the whole
    // key and bit 31 cannot mark an invalid candidate.  As for PKW the mask word saturates the cost
    // field (0xFFFFFF00, OR-ed in: above every valid key, the code bits survive), and a disparity
    // past D gets the same word from a wave-uniform constant folded into the same OR.
    constexpr bool SSDW = SSD && R > 5 && !WIDE;
    constexpr unsigned SSDW_MASKED = 0xFFFFFF00u;
    static_assert(!SSD || SSDW || SSDX || (unsigned long long)WIN * WIN * 65025ull * 256ull < (1ull << 31), "SSD key width");
    static_assert(!SSDX || (unsigned long long)WIN * WIN * 65025ull * 16ull + 15ull < (1ull << 30), "wide SSD key width");
    static_assert(!SSDW || (unsigned long long)WIN * WIN * 65025ull * 256ull + 255ull < (unsigned long long)SSDW_MASKED,
                  "wide SSD key width");
    constexpr unsigned SH = PK ? 16u : SSDX ? 4u : SSD ? 8u : BAND_SH;  // key = cost << SH | code
    constexpr unsigned PSH = PK ? 0u : SSDX ? 2u : SSD ? 4u : BAND_SH;  // pixel pre-shift in the ring
    constexpr unsigned LOW = (1u << SH) - 1;
    constexpr unsigned PAST_D = SSD ? BAND_MASKED : BAND_PAST_D;
    constexpr unsigned INIT = PK ? (((unsigned)(WIN * WIN * 255 + 1) << 16) - 1u)
                                 : SSDW ? SSDW_MASKED : SSD ? 0x7FFFFFFFu : BAND_INIT;
    constexpr int HL = (R + 3) / 4;  // halo lanes on each side of the tile (WIDE: of each 16-lane row)
    constexpr int OL = 16 - 2 * HL;  // WIDE: output lanes per row
    constexpr int TILE_OUT = WIDE ? 16 * OL : BAND_COLS - 8 * HL;
    constexpr int RING = band_ring(WIN);
    static_assert(!WIDE || ((255ull * WIN * WIN) << BAND_SH) + 4095ull < BAND_INIT, "wide SAD key width");
    constexpr int RING_M = band_mask_ring(WIN);
    static_assert(WIN + 2 <= RING && RING >= 8 && R + 2 <= RING_M, "ring too small");

    extern __shared__ u32x4 band_lds4[];
    const int nw = blockDim.x >> 6;
    const int rw4 = BAND_COLS / 4 + (BAND_ND / 4) * nw;  // img1 row length in 4-column groups
    const int rw = 4 * rw4;                              // img1 columns per ring row
    u32x4 *Lrow4 = band_lds4;                            // [RING][64]
    u32x4 *Rrow4 = Lrow4 + RING * (BAND_COLS / 4);  // [RING][rw4]
    u32x4 *Mrow4 = Rrow4 + RING * rw4;              // [RING_M][rw4]
    unsigned *Lrow = reinterpret_cast<unsigned *>(Lrow4);
    unsigned *Rrow = reinterpret_cast<unsigned *>(Rrow4);
    unsigned *Mrow = reinterpret_cast<unsigned *>(Mrow4);
    unsigned *best = reinterpret_cast<unsigned *>(Mrow4 + RING_M * rw4);  // [2][256]

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    // this lane's group of four tile columns; WIDE: the rows' column ranges overlap by their halos
    const int lcg = WIDE ? (lane >> 4) * OL + (lane & 15) : lane;
    const bool out_lane = !WIDE || ((lane & 15) >= HL && (lane & 15) < 16 - HL);
    const size_t pair = blockIdx.z;
    const uint8_t *i0 = a.img0 + pair * a.pair_stride;
    const uint8_t *i1 = a.img1 + pair * a.pair_stride;
    const int w = a.w, h = a.h;
    const int x0 = (int)blockIdx.x * TILE_OUT - 4 * HL;  // image column of column group 0, column 0
    const int yb = a.row0 + (int)blockIdx.y * a.bh;      // first output row of the band
    const int yend = min(yb + a.bh, a.row0 + a.rows);    // one past the last output row
    const int dB = a.dbase + BAND_ND * wave;             // this wave's first disparity
    const bool active = dB < a.D;  // waves past the last disparity block only help fill the ring
    const int rbase = x0 - a.dbase - BAND_ND * nw;       // image column of Rrow[.][0]
    // this lane's first img1 word: column x0 + 4*lane - dB - 16
    const int roff4 = lcg + (BAND_ND / 4) * (nw - 1 - wave);

    unsigned long long *best64 = reinterpret_cast<unsigned long long *>(best);  // SSDX: [2][256] 64-bit keys
    if constexpr (SSDX) {
        for (int i = tid; i < 2 * BAND_COLS; i += blockDim.x) best64[i] = ~0ull;
    } else {
        for (int i = tid; i < 2 * BAND_COLS; i += blockDim.x) best[i] = INIT;
    }

    // ---- ring fill: one image row -> slot ---------------------------------
    // each thread carries at most FILL_MAX words between its global loads (issued
    // at the top of a step) and its LDS stores (issued just before the barrier)
    // Loads go through buffer descriptors over the pair's two images: the column is a 32-bit
    // offset fixed per thread (0x80000000, which the range check turns into a zero, for columns
    // outside the image or beyond this thread's share), the row a scalar offset -- no per-row
    // 64-bit address arithmetic.  (The launcher keeps one image below 2 GiB.)
    constexpr int FILL_MAX = 2;
    unsigned fill_l = 0, fill_r[FILL_MAX] = {0, 0};
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(i0), 0, (int)((unsigned)a.stride * (unsigned)h), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(i1), 0, (int)((unsigned)a.stride * (unsigned)h), 0x00020000);
    constexpr unsigned FILL_NONE = 0x80000000u;
    unsigned col_l = FILL_NONE, col_r[FILL_MAX];
    if (tid < BAND_COLS && x0 + tid >= 0 && x0 + tid < w) col_l = (unsigned)(x0 + tid);
#pragma unroll
    for (int k = 0; k < FILL_MAX; ++k) {
        const int i = tid + k * (int)blockDim.x;
        const int x = rbase + i;
        col_r[k] = (i < rw && x >= 0 && x < w) ? (unsigned)x : FILL_NONE;
    }
    auto fill_load = [&](int yy) {
        if (yy >= 0 && yy < h) {  // uniform
            const int rowoff = yy * (int)a.stride;
            fill_l = __builtin_amdgcn_raw_buffer_load_b8(rs0, col_l, rowoff, 0);
#pragma unroll
            for (int k = 0; k < FILL_MAX; ++k) fill_r[k] = __builtin_amdgcn_raw_buffer_load_b8(rs1, col_r[k], rowoff, 0);
        } else {
            fill_l = 0;
#pragma unroll
            for (int k = 0; k < FILL_MAX; ++k) fill_r[k] = 0;
        }
    };
    auto fill_store = [&](int yy) {
        const int slot = band_slot<RING>(yy);
        if (tid < BAND_COLS) Lrow[slot * BAND_COLS + tid] = fill_l << PSH;
#pragma unroll
        for (int k = 0; k < FILL_MAX; ++k) {
            const int i = tid + k * (int)blockDim.x;
            if (i < rw) {
                Rrow[slot * rw + i] = fill_r[k] << PSH;
                // PK: the low half of a mask word is the img1 COLUMN x1 of the candidate -- the key's
                // tie-break code (lower x1 wins, the reference's first-wins order) travels with the
                // mask, so a key is one shift-or / and-or of (cost, mask word)
                Mrow[band_slot<RING_M>(yy) * rw + i] = PK ? ((fill_r[k] ? 0u : PK_MASKED) | ((unsigned)(rbase + i) & 0xFFFFu))
                                         : fill_r[k] ? 0u : (SSDW ? SSDW_MASKED : BAND_MASKED);
            }
        }
    };

    // Column sums.  The disparity code of the key, (4095 - d), is folded into the sums: it is
    // the initial value of V in exactly the columns x with x mod WIN == 0 -- every WIN-wide
    // window holds exactly one such column, the sliding add/subtract of rows never touches
    // the constant, so each window cost comes out as (cost << 12) + (4095 - d) with no extra
    // instruction.  Disparities past D (partial last block) carry BAND_PAST_D the same way.
    unsigned V[PK ? BAND_ND / 2 : BAND_ND][4];
    // PKW only: all-ones for a disparity past D (the half-word bias cannot mark it there)
    unsigned past_lo[PKW ? BAND_ND / 2 : 1], past_hi[PKW ? BAND_ND / 2 : 1];
    if constexpr (PK) {
#pragma unroll
        for (int pp = 0; pp < BAND_ND / 2; ++pp) {
            const int dlo = dB + pp, dhi = dB + pp + BAND_ND / 2;
            if constexpr (PKW) {
                past_lo[pp] = dlo >= a.D ? ~0u : 0u;
                past_hi[pp] = dhi >= a.D ? ~0u : 0u;
            }
            const unsigned init = (dlo < a.D ? 0u : PK_PAST) | ((dhi < a.D ? 0u : PK_PAST) << 16);
#pragma unroll
            for (int c = 0; c < 4; ++c) V[pp][c] = init;
        }
    } else {
        bool carrier[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) carrier[c] = ((x0 + 4 * lcg + c + 1024 * WIN) % WIN) == 0;
#pragma unroll
        for (int dd = 0; dd < BAND_ND; ++dd) {
            const unsigned code = dB + dd < a.D ? LOW - (unsigned)(SSDX ? dd : SSD ? dB + dd - a.dbase : dB + dd)
                                                : SSDW ? 0u : PAST_D;
#pragma unroll
            for (int c = 0; c < 4; ++c) V[dd][c] = carrier[c] ? code : 0u;
        }
    }

    unsigned W1[R == 0 ? BAND_ND : 1][4];  // R == 0 only: this row's costs
    const int ys0 = yb - R, ys1 = yend - 1 + R;  // streamed rows
    fill_load(ys0);
    fill_store(ys0);
    __syncthreads();

    for (int ys = ys0; ys <= ys1; ++ys) {
        if (ys < ys1) fill_load(ys + 1);

        // ---- add the row entering the window --------------------------------
        if (active) {
            const int slot = band_slot<RING>(ys);
            unsigned rv[20], lv[4];
            lds_read_1_5(Lrow4 + slot * (BAND_COLS / 4) + lcg, Rrow4 + slot * rw4 + roff4, lv, rv);
            if constexpr (PK) {
#pragma unroll
                for (int pp = 0; pp < BAND_ND / 2; ++pp) {
                    // four independent instructions, then their four dependants: on gfx950 a
                    // VALU instruction that reads a v_sad_* result right behind it costs a wait
                    // state (the compiler pads with s_nop), 64 of them per row in the paired form
                    unsigned t[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) t[c] = sad_u8(rv[16 + c - pp], lv[c], V[pp][c]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) V[pp][c] = sad_hi_u8(rv[16 + c - pp - BAND_ND / 2], lv[c], t[c]);
                }
            } else
#pragma unroll
            for (int d2 = 0; d2 < BAND_ND; d2 += 2) {
                // eight independent instructions, then their eight dependants (see the PK branch)
                unsigned sum[2][4];
                if (SSD) {
                    unsigned t[2][4];
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int c = 0; c < 4; ++c) t[q][c] = sad_u32_0(rv[16 + c - d2 - q], lv[c]);
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int c = 0; c < 4; ++c) sum[q][c] = sq_acc_u24(t[q][c], V[d2 + q][c]);
                } else {
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int c = 0; c < 4; ++c) sum[q][c] = sad_u32(rv[16 + c - d2 - q], lv[c], V[d2 + q][c]);
                }
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (R == 0)  // 1x1 window: the sum is one term; V only holds the code
                            W1[d2 + q][c] = sum[q][c];
                        else
                            V[d2 + q][c] = sum[q][c];
                    }
            }
        }
        // ---- subtract the row leaving the window ----------------------------
        if (active && R > 0 && ys - WIN >= ys0) {
            const int slot = band_slot<RING>(ys - WIN);
            unsigned rv[20], lv[4];
            lds_read_1_5(Lrow4 + slot * (BAND_COLS / 4) + lcg, Rrow4 + slot * rw4 + roff4, lv, rv);
            if constexpr (PK) {
#pragma unroll
                for (int pp = 0; pp < BAND_ND / 2; ++pp) {
                    unsigned t[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) t[c] = sad_u8_0(rv[16 + c - pp], lv[c]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) t[c] = sad_hi_u8(rv[16 + c - pp - BAND_ND / 2], lv[c], t[c]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) V[pp][c] -= t[c];
                }
            } else
#pragma unroll
            for (int d2 = 0; d2 < BAND_ND; d2 += 2) {
                unsigned t[2][4];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int c = 0; c < 4; ++c) t[q][c] = sad_u32_0(rv[16 + c - d2 - q], lv[c]);
                if (SSD) {
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int c = 0; c < 4; ++c) t[q][c] = sq_u24(t[q][c]);
                }
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int c = 0; c < 4; ++c) V[d2 + q][c] -= t[q][c];
            }
        }

        // ---- output row y = ys - R: window sums, keys, winner-take-all --------
        const int y = ys - R;
        if (active && y >= yb) {
            unsigned mv[20];
            lds_read_5(Mrow4 + band_slot<RING_M>(y) * rw4 + roff4, mv);
            unsigned m[4] = {~0u, ~0u, ~0u, ~0u};
            unsigned pend[4];
#pragma unroll
            for (int dd = 0; dd < (PK ? BAND_ND / 2 : BAND_ND); ++dd) {
                unsigned cost[4];
                if (R == 0) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) cost[c] = W1[dd][c];
                } else if constexpr (WIDE) {
                    const unsigned t01 = V[dd][0] + V[dd][1], t23 = V[dd][2] + V[dd][3];
                    unsigned P = t01 + t23;  // inclusive prefix of the lanes' totals along the row
                    P += row_from_left<1>(P);
                    P += row_from_left<2>(P);
                    P += row_from_left<4>(P);
                    P += row_from_left<8>(P);
                    // cost[0] covers columns -R .. R of this lane: qL whole lanes and the last remL
                    // columns of one more to the left; this lane, qR whole lanes and the first remR
                    // columns of one more to the right
                    constexpr int qL = R / 4, remL = R % 4, qR = (R - 3) / 4, remR = (R - 3) % 4;
                    unsigned edge = 0;
                    if constexpr (remL > 0) {
                        const unsigned suf = remL == 1 ? V[dd][3] : remL == 2 ? t23 : t23 + V[dd][1];
                        edge = row_from_left<qL + 1>(suf);
                    }
                    if constexpr (remR > 0) {
                        const unsigned pre = remR == 1 ? V[dd][0] : remR == 2 ? t01 : t01 + V[dd][2];
                        edge += row_from_right<qR + 1>(pre);
                    }
                    unsigned hi = P;
                    if constexpr (qR > 0) hi = row_from_right<qR>(P);
                    // (a fetched value is always the MINUEND: see the DPP note in the narrow slide)
                    unsigned n0 = row_from_left<qL + 1>(P) - hi;  // a prefix from left of the row is 0
                    asm volatile("" : "+v"(n0));
                    cost[0] = edge - n0;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const int take = c + 1 + R, dropd = R - c;  // columns to the right / left of column 0
                        unsigned tk, dr;
                        switch (take / 4) {
                        case 2: tk = row_from_right<2>(V[dd][take % 4]); break;
                        case 3: tk = row_from_right<3>(V[dd][take % 4]); break;
                        default: tk = row_from_right<4>(V[dd][take % 4]); break;
                        }
                        const int dcol = (4 - dropd % 4) % 4;
                        switch ((dropd + 3) / 4) {
                        case 2: dr = row_from_left<2>(V[dd][dcol]); break;
                        case 3: dr = row_from_left<3>(V[dd][dcol]); break;
                        default: dr = row_from_left<4>(V[dd][dcol]); break;
                        }
                        unsigned n = dr - cost[c];
                        asm volatile("" : "+v"(n));
                        cost[c + 1] = tk - n;
                    }
                } else if constexpr (R > 4) {
                    // R = 5..7: the window spans two lanes to each side.  cost[0] = this lane's
                    // and the left lane's four columns, columns 8-R..3 of the lane two to the
                    // left and columns 0..R-4 of the right lane
                    const unsigned all4 = (V[dd][0] + V[dd][1]) + (V[dd][2] + V[dd][3]);
                    unsigned suffix2 = V[dd][3];
#pragma unroll
                    for (int c = 2; c >= 8 - R; --c) suffix2 += V[dd][c];
                    unsigned prefix = V[dd][0];
#pragma unroll
                    for (int c = 1; c <= R - 4; ++c) prefix += V[dd][c];
                    cost[0] = (all4 + from_left(all4)) + (from_left(from_left(suffix2)) + from_right(prefix));
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        // slide as below: the dropped column is the minuend (see the DPP note there)
                        const int drop = c - R, take = c + 1 + R;
                        const unsigned tk = take <= 7 ? from_right(V[dd][take - 4])
                                                      : from_right(from_right(V[dd][take - 8]));
                        unsigned n = (drop >= -4 ? from_left(V[dd][drop + 4])
                                                 : from_left(from_left(V[dd][drop + 8]))) - cost[c];
                        asm volatile("" : "+v"(n));
                        cost[c + 1] = tk - n;
                    }
                } else {
                    // cost[c] = sum of V over columns c-R .. c+R; columns < 0 live in the
                    // left neighbour lane (as its column +4), columns > 3 in the right one
                    unsigned own = V[dd][0];
#pragma unroll
                    for (int c = 1; c <= (R < 3 ? R : 3); ++c) own += V[dd][c];
                    unsigned suffix = V[dd][3];  // columns 4-R .. 3, wanted by the right neighbour
#pragma unroll
                    for (int c = 2; c >= 4 - R; --c) suffix += V[dd][c];
                    cost[0] = own + from_left(suffix);
                    if (R == 4) cost[0] += from_right(V[dd][0]);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        // slide: drop column c-R, take column c+1+R.  A column taken from the
                        // left lane is always the MINUEND of its subtraction (n = left - cost,
                        // then take - n): hipcc folds "x - dpp(y)" into v_subrev_u32_dpp, which
                        // this toolchain encodes with the DPP on the wrong operand (measured:
                        // tools/dpp_test.hip); "dpp(y) - x" (v_sub_u32_dpp) is correct.
                        const int drop = c - R, take = c + 1 + R;
                        const unsigned tk = take <= 3 ? V[dd][take] : from_right(V[dd][take - 4]);
                        if (drop >= 0) {
                            cost[c + 1] = tk + (cost[c] - V[dd][drop]);
                        } else {
                            unsigned n = from_left(V[dd][drop + 4]) - cost[c];
                            asm volatile("" : "+v"(n));  // keep hipcc from re-associating
                            cost[c + 1] = tk - n;
                        }
                    }
                }
                if constexpr (PK) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        // the two half-word costs become two keys; mask word: 0 or BAND_MASKED
                        // mask word: x1 in the low half, 0 or PK_MASKED above it
                        unsigned klo = (cost[c] << 16) | mv[16 + c - dd];
                        unsigned khi = (cost[c] & 0xFFFF0000u) | mv[16 + c - dd - BAND_ND / 2];
                        if constexpr (PKW) {
                            klo |= past_lo[dd];
                            khi |= past_hi[dd];
                        }
                        m[c] = min3_u32(m[c], klo, khi);
                    }
                } else
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    // mask word: 0 or BAND_MASKED (SSDW: 0 or the saturating word, and the same for d >= D)
                    const unsigned key = SSDW ? (cost[c] | mv[16 + c - dd] | (dB + dd < a.D ? 0u : SSDW_MASKED))
                                         : SSD ? (cost[c] | mv[16 + c - dd]) : (cost[c] + mv[16 + c - dd]);
                    if (dd & 1)
                        m[c] = min3_u32(m[c], pend[c], key);
                    else
                        pend[c] = key;
                }
            }
            unsigned *b = best + (y & 1) * BAND_COLS + 4 * lcg;
            if constexpr (SSDX) {
                if (out_lane) {
                    unsigned long long *b64 = best64 + (y & 1) * BAND_COLS + 4 * lcg;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (m[c] < INIT) {  // the wave's winner: cost and pass-local disparity, widened
                            const unsigned dloc = (unsigned)(dB - a.dbase) + (LOW - (m[c] & LOW));
                            atomicMin(b64 + c, ((unsigned long long)(m[c] >> SH) << 32) | (4095u - dloc));
                        }
                }
            } else if (out_lane) {  // (WIDE: a halo lane's columns are another row's output columns)
#pragma unroll
                for (int c = 0; c < 4; ++c) atomicMin(b + c, m[c]);
            }
        }

        if (ys < ys1) fill_store(ys + 1);
        __syncthreads();

        // ---- write out row y (all waves have contributed) ---------------------
        if (y >= yb && tid < BAND_COLS) {
            const int c = tid;
            unsigned key;
            unsigned long long key64 = ~0ull;
            if constexpr (SSDX) {
                key64 = best64[(y & 1) * BAND_COLS + c];
                best64[(y & 1) * BAND_COLS + c] = ~0ull;
                // the narrow view the code below shares: any valid key < INIT, the code in the low bits
                key = key64 == ~0ull ? INIT : 0u;
            } else {
                key = best[(y & 1) * BAND_COLS + c];
                best[(y & 1) * BAND_COLS + c] = INIT;
            }
            const int x = x0 + c;
            if ((c >> 2) >= HL && (c >> 2) < (WIDE ? HL + 4 * OL : 64 - HL) && x < w) {
                const size_t pix = (size_t)(y - a.row0) * w + x;
                unsigned *kp = a.keys ? a.keys + pair * (size_t)a.rows * w + pix : nullptr;
                bool emit;
                if constexpr (SSDX) {
                    // keys carries the best COST between passes (~0: none yet)
                    const unsigned cost = (unsigned)(key64 >> 32);
                    emit = true;
                    if (kp && !a.first_pass) {
                        const unsigned old = *kp;
                        emit = key < INIT && (old == ~0u || cost <= old);
                    }
                    if (kp && !a.last_pass && emit) *kp = key < INIT ? cost : ~0u;
                } else if (SSD) {
                    // the output array carries the best disparity between passes, keys the best key
                    emit = true;
                    if (kp && !a.first_pass) {
                        const unsigned old = *kp;
                        emit = key < INIT && (old >= INIT || (key >> SH) <= (old >> SH));
                    }
                    if (kp && !a.last_pass && emit) *kp = key;
                } else {
                    if (kp && !a.first_pass) key = min(key, *kp);
                    if (!a.last_pass) *kp = key;
                    emit = a.last_pass;
                }
                if (emit) {
                    const int slot = band_slot<RING>(y);
                    int v;
                    if (Lrow[slot * BAND_COLS + c] == 0)
                        v = 0;  // centre pixel of img0 is 0: output stays 0
                    else if (key >= INIT)
                        v = x;  // no valid candidate: xBest stays 0
                    else
                        v = PK ? (int)(((unsigned)x - key) & LOW)  // the key's code is xBest (mod 2^16)
                            : SSDX ? (int)(4095u - (unsigned)(key64 & 0xFFFFFFFFu)) + a.dbase
                               : (int)(LOW - (key & LOW)) + (SSD ? a.dbase : 0);  // d = x - xBest
                    uint8_t *orow = a.out + pair * a.out_pair_stride + (size_t)(y - a.row0) * a.out_stride;
                    if (a.out_bytes == 1)
                        orow[x] = (uint8_t)(v & 0xFF);
                    else
                        reinterpret_cast<uint16_t *>(orow)[x] = (uint16_t)(v & 0xFFFF);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// D1-ext, windows 17x17 .. 31x31 (SAD and SSD): column-sum kernel.
//
// The band kernel above keeps a lane's column sums in registers and reaches its neighbours'
// with DPP, which stops at two lanes to each side (window radius <= 7).  Larger windows keep
// the column sums in LDS instead: workgroup = a tile of 64 output columns x a band of rows x up
// to 256 disparities; colsum[d][i] is the running sum, over the window's rows, of the tap cost
// at tile column i (64 + 2R of them) for disparity d.  Per image row: the entering and the
// leaving row update every colsum entry (step 1), then every (column, d) window cost is a
// sliding sum of 2R+1 neighbouring colsum entries (step 2), one 64-bit LDS atomic min per
// (column, d) picks the winner -- cost in the upper word, ~d in the lower, so equal costs go to
// the larger d = the lower x1, the reference's first-wins order.  Work per (pixel, candidate)
// is independent of the window size, as in the band kernel; it is slower than that kernel (LDS
// instead of registers) and exists so that no window falls off a cliff.
// ---------------------------------------------------------------------------

constexpr int CS_TX = 64;       // output columns per tile
constexpr int CS_XS = 97;       // colsum row stride in words (64 + 2*15 = 94 used; odd: no bank conflicts down a column)
constexpr int CS_XT = 96;       // staged tile columns
constexpr int CS_D = 256;       // disparities per pass
constexpr int CS_THREADS = 1024;  // 16 waves: the 100 KB of column sums allow one workgroup per CU

struct ColsumArgs {
    const uint8_t *img0;
    const uint8_t *img1;
    int w, h;
    size_t stride, pair_stride;
    int row0, rows, bh;
    int r;
    int D, dbase, nd;   // total disparities, first of this pass, disparities of this pass
    unsigned *keys;     // best cost per pixel between passes (nullptr if one pass)
    int first_pass;
    uint8_t *out;
    size_t out_stride, out_pair_stride;
    int out_bytes;
};

template <bool SSD>
__global__ __launch_bounds__(CS_THREADS) void disp_colsum_kernel(ColsumArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned cs_lds[];
    const int nd = a.nd, R = a.r, WIN = 2 * R + 1;
    const int rseg = CS_XT + nd - 1;                 // img1 columns a tile row needs
    unsigned long long *best = reinterpret_cast<unsigned long long *>(cs_lds);  // [CS_TX]
    unsigned *colsum = cs_lds + 2 * CS_TX;           // [nd][CS_XS]
    unsigned *Le = colsum + nd * CS_XS;              // entering / leaving / centre rows of img0 [CS_XT]
    unsigned *Ll = Le + CS_XT;
    unsigned *Lc = Ll + CS_XT;
    unsigned *Re = Lc + CS_XT;                       // the same of img1 [rseg]
    unsigned *Rl = Re + rseg;
    unsigned *Rc = Rl + rseg;
    constexpr unsigned long long NONE = ~0ull;

    const int tid = threadIdx.x;
    const size_t pair = blockIdx.z;
    const uint8_t *i0 = a.img0 + pair * a.pair_stride;
    const uint8_t *i1 = a.img1 + pair * a.pair_stride;
    const int w = a.w, h = a.h;
    const int x0 = (int)blockIdx.x * CS_TX;            // first output column
    const int yb = a.row0 + (int)blockIdx.y * a.bh;
    const int yend = min(yb + a.bh, a.row0 + a.rows);
    const int lbase = x0 - R;                          // image column of tile column 0
    const int rbase = lbase - a.dbase - (nd - 1);      // image column of Re[0]

    for (int i = tid; i < nd * CS_XS; i += CS_THREADS) colsum[i] = 0;
    if (tid < CS_TX) best[tid] = NONE;

    auto stage_row = [&](unsigned *L, unsigned *Rr, int yy, bool live) {
        const bool in = live && yy >= 0 && yy < h;
        for (int i = tid; i < CS_XT; i += CS_THREADS) {
            const int x = lbase + i;
            L[i] = (in && x >= 0 && x < w) ? i0[(size_t)yy * a.stride + x] : 0u;
        }
        for (int i = tid; i < rseg; i += CS_THREADS) {
            const int x = rbase + i;
            Rr[i] = (in && x >= 0 && x < w) ? i1[(size_t)yy * a.stride + x] : 0u;
        }
    };

    const int ys0 = yb - R, ys1 = yend - 1 + R;
    for (int ys = ys0; ys <= ys1; ++ys) {
        const int y = ys - R;  // the output row whose window is complete after this step
        __syncthreads();       // the previous row's readers are done with the row segments
        stage_row(Le, Re, ys, true);
        stage_row(Ll, Rl, ys - WIN, ys - WIN >= ys0);  // a row that never entered leaves as zeros
        if (y >= yb) stage_row(Lc, Rc, y, true);
        __syncthreads();
        // ---- step 1: colsum[d][i] += cost(entering row) - cost(leaving row) ----------
        {
            const int xl = tid & 31, dg = tid >> 5;
            for (int i = xl; i < CS_XT; i += 32) {
                const unsigned le = Le[i], ll = Ll[i];
                for (int dd = dg; dd < nd; dd += CS_THREADS / 32) {
                    const int j = i + (nd - 1) - dd;  // img1 column lbase + i - (dbase + dd)
                    unsigned e = absdiff_add(le, Re[j], 0u), l = absdiff_add(ll, Rl[j], 0u);
                    if (SSD) {
                        e *= e;
                        l *= l;
                    }
                    colsum[dd * CS_XS + i] += e - l;
                }
            }
        }
        __syncthreads();
        if (y < yb) continue;
        // ---- step 2: window costs by sliding sums along x, winner per column ----------
        {
            const int dl = (tid & 31) + 32 * (tid >> 8), g = (tid >> 5) & 7;   // 8 groups of 8 output columns
            unsigned long long mine[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) mine[k] = NONE;
            for (int dd = dl; dd < nd; dd += CS_THREADS / 8) {
                const unsigned *row = colsum + dd * CS_XS + 8 * g;  // window of column 8g: row[0 .. 2R]
                unsigned sum = 0;
                for (int i = 0; i <= 2 * R; ++i) sum += row[i];
                const unsigned code = ~(unsigned)(a.dbase + dd);
                const unsigned *rc = Rc + 8 * g + R + (nd - 1) - dd;  // img1 centre pixel of column 8g at this d
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (k > 0) sum += row[2 * R + k] - row[k - 1];
                    // candidate x1 = x - d: inside the image, its centre pixel non-zero (:676), d < D
                    if (rc[k] != 0) {
                        const unsigned long long key = ((unsigned long long)sum << 32) | code;
                        mine[k] = key < mine[k] ? key : mine[k];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (mine[k] != NONE) atomicMin(best + 8 * g + k, mine[k]);
        }
        __syncthreads();
        // ---- output row y -----------------------------------------------------------------
        if (tid < CS_TX) {
            const unsigned long long key = best[tid];
            best[tid] = NONE;
            const int x = x0 + tid;
            if (x < w) {
                const bool centre = Lc[tid + R] != 0;       // img0 centre 0: the output stays 0 (:670)
                bool valid = centre && key != NONE;
                const unsigned cost = (unsigned)(key >> 32);
                const size_t pix = (size_t)(y - a.row0) * w + x;
                unsigned *kp = a.keys ? a.keys + pair * (size_t)a.rows * w + pix : nullptr;
                bool emit = true;
                if (kp) {
                    if (!a.first_pass) {
                        const unsigned old = *kp;   // best cost so far, ~0 = none
                        valid = valid && cost <= old;  // a later pass holds larger d: it wins ties
                        emit = valid;
                    }
                    if (a.first_pass || valid) *kp = valid ? cost : ~0u;
                }
                if (emit) {
                    const int v = !centre ? 0 : valid ? (int)~(unsigned)key : x;  // no candidate: xBest stays 0
                    uint8_t *orow = a.out + pair * a.out_pair_stride + (size_t)(y - a.row0) * a.out_stride;
                    if (a.out_bytes == 1)
                        orow[x] = (uint8_t)(v & 0xFF);
                    else
                        reinterpret_cast<uint16_t *>(orow)[x] = (uint16_t)(v & 0xFFFF);
                }
            }
        }
    }
}

template <bool SSD>
int launch_colsum(stx_ctx *ctx, ColsumArgs a, int n_pairs, void *keys, size_t keys_bytes)
{
    const int xt = (a.w + CS_TX - 1) / CS_TX;
    const int npass = (a.D + CS_D - 1) / CS_D;
    const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
    // band height: the 2R warm-up rows against an even fill of the CUs (as launch_band)
    int best_bh = a.rows;
    double best_t = 1e300;
    for (int bh = 8; bh <= 1024; bh += 4) {
        const int bands = (a.rows + bh - 1) / bh;
        const long wgs = (long)xt * bands * n_pairs;
        const long rounds = (wgs + cus - 1) / cus;
        const double t = (double)rounds * (bh + 2 * a.r + 1);
        if (t < best_t) { best_t = t; best_bh = bh; }
        if (bh >= a.rows) break;
    }
    a.bh = best_bh;
    const int bands = (a.rows + a.bh - 1) / a.bh;
    if (bands > 65535 || n_pairs > 65535) return STX_ERR_TOO_LARGE;
    a.keys = nullptr;
    if (npass > 1) {
        const size_t need = sizeof(unsigned) * (size_t)n_pairs * a.rows * a.w;
        if (keys) {
            if (need > keys_bytes) return STX_ERR_TOO_LARGE;
            a.keys = static_cast<unsigned *>(keys);
        } else {
            int err;
            a.keys = (unsigned *)stx_workspace(ctx, STX_SLOT_DISP_KEYS, need, &err);
            if (!a.keys) return err;
        }
    }
    for (int p = 0; p < npass; ++p) {
        a.dbase = p * CS_D;
        a.nd = a.D - a.dbase < CS_D ? a.D - a.dbase : CS_D;
        a.first_pass = p == 0;
        const int rseg = CS_XT + a.nd - 1;
        const size_t lds = sizeof(unsigned long long) * CS_TX +
                           sizeof(unsigned) * ((size_t)a.nd * CS_XS + 3 * CS_XT + 3 * (size_t)rseg);
        const int rc = stx_set_lds_attr(ctx, reinterpret_cast<const void *>(disp_colsum_kernel<SSD>), (int)lds);
        if (rc != STX_OK) return rc;
        hipLaunchKernelGGL((disp_colsum_kernel<SSD>), dim3(xt, bands, n_pairs), dim3(CS_THREADS), lds, ctx->stream, a);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    return STX_OK;
}

template <int R, bool SSD>
int launch_band(stx_ctx *ctx, BandArgs a, int n_pairs, void *keys, size_t keys_bytes)
{
    constexpr int HL = (R + 3) / 4;
    constexpr int TILE_OUT = R > 7 ? 16 * (16 - 2 * HL) : BAND_COLS - 8 * HL;
    const int xt = (a.w + TILE_OUT - 1) / TILE_OUT;
    // disparities per pass: 16 waves' worth (or 8, should 16 waves' img1 columns not fit the LDS
    // next to the ring)
    constexpr size_t LDS_MAX = 160 * 1024;
    constexpr int RING_ROWS = band_ring(2 * R + 1), RING_M = band_mask_ring(2 * R + 1);
    constexpr int BEST_WORDS = (SSD && R > 7 ? 4 : 2) * BAND_COLS;  // wide SSD: 64-bit keys
    constexpr int RW_MAX = BAND_COLS + BAND_ND * BAND_MAX_WAVES;
    constexpr int NW_MAX = sizeof(unsigned) * ((size_t)RING_ROWS * (BAND_COLS + RW_MAX) + (size_t)RING_M * RW_MAX + BEST_WORDS) <= LDS_MAX
                               ? BAND_MAX_WAVES : BAND_MAX_WAVES / 2;
    const int passD = BAND_ND * NW_MAX;
    const int npass = (a.D + passD - 1) / passD;
    // band height: trade the 2R warm-up rows against filling the CUs evenly
    const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
    int best_bh = a.rows;
    double best_t = 1e300;
    for (int bh = 8; bh <= 512; bh += 4) {
        const int bands = (a.rows + bh - 1) / bh;
        const long wgs = (long)xt * bands * n_pairs;
        const long rounds = (wgs + cus - 1) / cus;
        const int last = a.rows - (bands - 1) * bh;
        const double t = (double)rounds * (bh + 2 * R + 1) + 0.001 * (bh - last);
        if (t < best_t) { best_t = t; best_bh = bh; }
        if (bh >= a.rows) break;
    }
    a.bh = best_bh;
    const int bands = (a.rows + a.bh - 1) / a.bh;
    if (bands > 65535 || n_pairs > 65535) return STX_ERR_TOO_LARGE;
    if (npass > 1) {
        const size_t need = sizeof(unsigned) * (size_t)n_pairs * a.rows * a.w;
        if (keys) {  // caller-provided scratch (the no-allocation entry point)
            if (need > keys_bytes) return STX_ERR_TOO_LARGE;
            a.keys = static_cast<unsigned *>(keys);
        } else {
            int err;
            a.keys = (unsigned *)stx_workspace(ctx, STX_SLOT_DISP_KEYS, need, &err);
            if (!a.keys) return err;
        }
    } else {
        a.keys = nullptr;
    }
    for (int p = 0; p < npass; ++p) {
        a.dbase = p * passD;
        const int dleft = a.D - a.dbase;
        // at least 4 waves: the ring fill needs 256 threads for the img0 row
        const int nw = max(4, (min(dleft, passD) + BAND_ND - 1) / BAND_ND);
        a.first_pass = p == 0;
        a.last_pass = p == npass - 1;
        const int rw = BAND_COLS + BAND_ND * nw;
        const size_t lds = sizeof(unsigned) * ((size_t)RING_ROWS * (BAND_COLS + rw) + (size_t)RING_M * rw + BEST_WORDS);
        const int rc = stx_set_lds_attr(ctx, reinterpret_cast<const void *>(disp_band_kernel<R, SSD>), (int)lds);
        if (rc != STX_OK) return rc;
        hipLaunchKernelGGL((disp_band_kernel<R, SSD>), dim3(xt, bands, n_pairs), dim3(64 * nw), lds,
                           ctx->stream, a);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    return STX_OK;
}

}  // namespace

int stx_launch_disparity(stx_ctx *ctx, const uint8_t *d_img0, const uint8_t *d_img1,
                         int n_pairs, int w, int h, size_t stride, size_t pair_stride,
                         int row0, int rows, int win, int max_disp, int cost_mode,
                         void *d_out, size_t out_stride, size_t out_pair_stride,
                         int out_bytes, void *keys, size_t keys_bytes)
{
    if (w <= 0 || h <= 0 || rows <= 0 || n_pairs <= 0) return STX_OK;
    if (rows > 65535 * EXT_TY || n_pairs > 65535) return STX_ERR_TOO_LARGE;
    uint8_t *o = static_cast<uint8_t *>(d_out);

    if (max_disp == 0) {
        // reference mode (win == 1 checked by the caller); SSD orders candidates
        // exactly as SAD does for a single tap, so one kernel serves both.
        if (w > (1 << REF_SHIFT)) return STX_ERR_TOO_LARGE;
        if (rows > 65535) return STX_ERR_TOO_LARGE;
        dim3 grid((w + REF_THREADS * REF_PX - 1) / (REF_THREADS * REF_PX), rows, n_pairs);
        hipLaunchKernelGGL(disp_ref_kernel, grid, dim3(REF_THREADS), 0, ctx->stream, d_img0,
                           d_img1, w, stride, pair_stride, row0, o, out_stride, out_pair_stride,
                           out_bytes);
        return (int)hipGetLastError();
    }

    const int r = win / 2;
    const int D = max_disp < w ? max_disp : w;  // candidates beyond column 0 do not exist
    const bool ssd = cost_mode == STX_COST_SSD && r > 0;  // a single tap orders candidates as SAD does
    // (the band kernel addresses an image through 32-bit buffer offsets: one image below 2 GiB)
    const bool band_ok = D <= 4096 && w < 65536 && !ctx->force_bruteforce && (unsigned long long)stride * (unsigned)h < (1ull << 31);
    if (band_ok && r <= 15 && !ctx->force_colsum) {
        BandArgs a;
        a.img0 = d_img0; a.img1 = d_img1; a.w = w; a.h = h; a.stride = stride;
        a.pair_stride = pair_stride; a.row0 = row0; a.rows = rows; a.bh = rows; a.D = D;
        a.dbase = 0; a.keys = nullptr; a.first_pass = 1; a.last_pass = 1; a.out = o;
        a.out_stride = out_stride; a.out_pair_stride = out_pair_stride; a.out_bytes = out_bytes;
        if (ssd) {
            switch (r) {
            case 1: return launch_band<1, true>(ctx, a, n_pairs, keys, keys_bytes);
            case 2: return launch_band<2, true>(ctx, a, n_pairs, keys, keys_bytes);
            case 3: return launch_band<3, true>(ctx, a, n_pairs, keys, keys_bytes);
            case 4: return launch_band<4, true>(ctx, a, n_pairs, keys, keys_bytes);
            case 5: return launch_band<5, true>(ctx, a, n_pairs, keys, keys_bytes);
            case 6: return launch_band<6, true>(ctx, a, n_pairs, keys, keys_bytes);
            case 7: return launch_band<7, true>(ctx, a, n_pairs, keys, keys_bytes);
            case 8: return launch_band<8, true>(ctx, a, n_pairs, keys, keys_bytes);
            case 9: return launch_band<9, true>(ctx, a, n_pairs, keys, keys_bytes);
            case 10: return launch_band<10, true>(ctx, a, n_pairs, keys, keys_bytes);
            case 11: return launch_band<11, true>(ctx, a, n_pairs, keys, keys_bytes);
            case 12: return launch_band<12, true>(ctx, a, n_pairs, keys, keys_bytes);
            case 13: return launch_band<13, true>(ctx, a, n_pairs, keys, keys_bytes);
            case 14: return launch_band<14, true>(ctx, a, n_pairs, keys, keys_bytes);
            default: return launch_band<15, true>(ctx, a, n_pairs, keys, keys_bytes);
            }
        }
        switch (r) {
        case 0: return launch_band<0, false>(ctx, a, n_pairs, keys, keys_bytes);
        case 1: return launch_band<1, false>(ctx, a, n_pairs, keys, keys_bytes);
        case 2: return launch_band<2, false>(ctx, a, n_pairs, keys, keys_bytes);
        case 3: return launch_band<3, false>(ctx, a, n_pairs, keys, keys_bytes);
        case 4: return launch_band<4, false>(ctx, a, n_pairs, keys, keys_bytes);
        case 5: return launch_band<5, false>(ctx, a, n_pairs, keys, keys_bytes);
        case 6: return launch_band<6, false>(ctx, a, n_pairs, keys, keys_bytes);
        case 7: return launch_band<7, false>(ctx, a, n_pairs, keys, keys_bytes);
        case 8: return launch_band<8, false>(ctx, a, n_pairs, keys, keys_bytes);
        case 9: return launch_band<9, false>(ctx, a, n_pairs, keys, keys_bytes);
        case 10: return launch_band<10, false>(ctx, a, n_pairs, keys, keys_bytes);
        case 11: return launch_band<11, false>(ctx, a, n_pairs, keys, keys_bytes);
        case 12: return launch_band<12, false>(ctx, a, n_pairs, keys, keys_bytes);
        case 13: return launch_band<13, false>(ctx, a, n_pairs, keys, keys_bytes);
        case 14: return launch_band<14, false>(ctx, a, n_pairs, keys, keys_bytes);
        default: return launch_band<15, false>(ctx, a, n_pairs, keys, keys_bytes);
        }
    }
    // Beyond the register kernel's reach (more than 4096 disparities -- its 12-bit disparity code --,
    // 65536 columns or more, an image of 2 GiB): the column-sum kernel, whose radius is a run-time
    // value and whose passes of 256 disparities merge on the cost.  Every window 1x1 .. 31x31.
    if (r <= 15 && D <= 65536 && !ctx->force_bruteforce) {
        ColsumArgs c;
        c.img0 = d_img0; c.img1 = d_img1; c.w = w; c.h = h; c.stride = stride; c.pair_stride = pair_stride;
        c.row0 = row0; c.rows = rows; c.bh = rows; c.r = r; c.D = D; c.dbase = 0; c.nd = 0; c.keys = nullptr;
        c.first_pass = 1; c.out = o; c.out_stride = out_stride; c.out_pair_stride = out_pair_stride;
        c.out_bytes = out_bytes;
        return cost_mode == STX_COST_SSD ? launch_colsum<true>(ctx, c, n_pairs, keys, keys_bytes)
                                         : launch_colsum<false>(ctx, c, n_pairs, keys, keys_bytes);
    }
    const size_t lds = (size_t)(EXT_TX + 2 * r) * (EXT_TY + 2 * r) +
                       (size_t)(EXT_TX + 2 * r + D - 1) * (EXT_TY + 2 * r);
    if (lds > 64 * 1024) return STX_ERR_TOO_LARGE;
    dim3 grid((w + EXT_TX - 1) / EXT_TX, (rows + EXT_TY - 1) / EXT_TY, n_pairs);
    if (cost_mode == STX_COST_SSD)
        hipLaunchKernelGGL(disp_ext_bruteforce_kernel<true>, grid, dim3(EXT_THREADS), lds,
                           ctx->stream, d_img0, d_img1, w, h, stride, pair_stride, row0, rows, r, D,
                           o, out_stride, out_pair_stride, out_bytes);
    else
        hipLaunchKernelGGL(disp_ext_bruteforce_kernel<false>, grid, dim3(EXT_THREADS), lds,
                           ctx->stream, d_img0, d_img1, w, h, stride, pair_stride, row0, rows, r, D,
                           o, out_stride, out_pair_stride, out_bytes);
    return (int)hipGetLastError();
}
