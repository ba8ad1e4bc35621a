"""Seeded synthetic inputs.  Pure numpy; the generator is
splitmix64, implemented here so that this container and the GPU box produce
identical bytes without relying on numpy's Generator stream stability."""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, n):
    """n 64-bit outputs of splitmix64 started at `seed` (vectorised)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed, n):
    """n doubles in [0,1)."""
    return (splitmix64(seed, n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal(seed, n):
    """n standard normals (Box-Muller on splitmix64)."""
    m = (n + 1) // 2
    u1 = 1.0 - uniform01(seed, m)
    u2 = uniform01(seed ^ 0x5DEECE66D, m)
    r = np.sqrt(-2.0 * np.log(u1))
    out = np.concatenate([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)])
    return out[:n]


def _value_noise(seed, h, w, cell):
    gh, gw = h // cell + 2, w // cell + 2
    g = uniform01(seed, gh * gw).reshape(gh, gw)
    ys = np.arange(h) / cell
    xs = np.arange(w) / cell
    y0 = ys.astype(np.int64)
    x0 = xs.astype(np.int64)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    a = g[y0][:, x0]
    b = g[y0][:, x0 + 1]
    c = g[y0 + 1][:, x0]
    d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def texture(seed, h, w):
    """Band-limited noise texture in [1,255], uint8."""
    t = (0.5 * _value_noise(seed, h, w, 16) + 0.3 * _value_noise(seed + 1, h, w, 4)
         + 0.2 * uniform01(seed + 2, h * w).reshape(h, w))
    return np.ascontiguousarray(np.clip(1 + t * 254.0, 1, 255).astype(np.uint8))


def rectified_pair(seed, h, w, max_disp, border=4, zero_frac=0.02):
    """(left, right) uint8 images: right(y,x) = left(y, x + d(y,x)) with
    piecewise-constant d in [0, max_disp) on 64x64 blocks, +-2 grey levels of
    noise, a zero border frame and ~zero_frac scattered zero pixels (the
    rectification black regions the reference's zero-skip is there for)."""
    left = texture(seed, h, w)
    bh, bw = (h + 63) // 64, (w + 63) // 64
    dmax = max(1, min(max_disp, w))
    dblk = (uniform01(seed + 10, bh * bw) * dmax).astype(np.int64).reshape(bh, bw)
    d = np.repeat(np.repeat(dblk, 64, axis=0), 64, axis=1)[:h, :w]
    xs = np.clip(np.arange(w)[None, :] + d, 0, w - 1)
    right = np.take_along_axis(left, xs, axis=1).astype(np.int64)
    noise = (uniform01(seed + 11, h * w) * 5).astype(np.int64).reshape(h, w) - 2
    right = np.ascontiguousarray(np.clip(right + noise, 1, 255).astype(np.uint8))
    left = left.copy()
    for k, img in enumerate((left, right)):
        if border > 0 and h > 2 * border and w > 2 * border:
            img[:border] = 0
            img[-border:] = 0
            img[:, :border] = 0
            img[:, -border:] = 0
        z = uniform01(seed + 20 + k, h * w).reshape(h, w) < zero_frac * 0.25
        img[z] = 0
    return left, right


def sift_like(seed, n):
    """n x 128 fp32 SIFT-like descriptors (Features.cpp:908-932 value
    distribution): |N(0,1)| with ~50% zeros -> normalise -> clamp 0.2 -> renormalise."""
    g = np.abs(normal(seed, n * 128)).reshape(n, 128)
    keep = uniform01(seed + 1, n * 128).reshape(n, 128) < 0.5
    g = g * keep
    g[:, 0] += 1e-3
    g = g / np.linalg.norm(g, axis=1, keepdims=True)
    g = np.minimum(g, 0.2)
    g = g / np.linalg.norm(g, axis=1, keepdims=True)
    return g.astype(np.float32)


def descriptor_sets(seed, n1, n2):
    """(desc1, score1, desc2, score2): list2 = for half of its rows a noisy
    copy of a random list1 row (true matches), fresh rows otherwise."""
    d1 = sift_like(seed, n1)
    d2 = sift_like(seed + 100, n2)
    pick = (uniform01(seed + 200, n2) * n1).astype(np.int64)
    ismatch = uniform01(seed + 201, n2) < 0.5
    noisy = d1[pick].astype(np.float64) + 0.01 * normal(seed + 202, n2 * 128).reshape(n2, 128)
    noisy = np.maximum(noisy, 0)
    noisy = (noisy / np.linalg.norm(noisy, axis=1, keepdims=True)).astype(np.float32)
    d2 = np.where(ismatch[:, None], noisy, d2).astype(np.float32)
    s1 = (700 + 4300 * uniform01(seed + 300, n1)).astype(np.float32)
    s2 = (700 + 4300 * uniform01(seed + 301, n2)).astype(np.float32)
    return d1, s1, d2, s2


def descriptor_sets_clustered(seed, n1, n2, centres=2000, noise=1e-3, big_frac=0.005, big=2, with_clusters=False):
    """(desc1, score1, desc2, score2) with repeated structure: every row of both lists is one of
    `centres` SIFT-like descriptors plus N(0, noise^2) per component (clamped at 0, renormalised),
    so a query has n2 / centres near-duplicates in list2 whose distances differ in the last digits
    -- the case where near-ties pile up in the running-minimum scan (Features.cpp:1036-1047).  A
    fraction `big_frac` of the rows of BOTH lists is dealt to the first `big` centres: those
    queries see hundreds of near-ties (n2 * big_frac / big each), enough to overflow the
    prefilter's candidate lists.  Rows are in random order; scores as descriptor_sets.
    with_clusters: also return each row's centre (which1, which2)."""
    c = sift_like(seed, centres).astype(np.float64)

    def draw(sd, n):
        which = (uniform01(sd, n) * centres).astype(np.int64)
        if big > 0 and big_frac > 0:
            isbig = uniform01(sd + 1, n) < big_frac
            which = np.where(isbig, (uniform01(sd + 2, n) * big).astype(np.int64), which)
        d = c[which] + noise * normal(sd + 3, n * 128).reshape(n, 128)
        d = np.maximum(d, 0)
        return (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32), which

    d1, w1 = draw(seed + 1000, n1)
    d2, w2 = draw(seed + 2000, n2)
    s1 = (700 + 4300 * uniform01(seed + 300, n1)).astype(np.float32)
    s2 = (700 + 4300 * uniform01(seed + 301, n2)).astype(np.float32)
    if with_clusters:
        return d1, s1, d2, s2, w1, w2
    return d1, s1, d2, s2


def homography(seed, w, h, deg_y=2.0, deg_x=0.5):
    """H = K R K^-1 (main.cpp:365) for a Middlebury-like K scaled to w x h."""
    f = 3979.911 * w / 2964.0
    K = np.array([[f, 0, w * 0.42], [0, f, h * 0.51], [0, 0, 1]], np.float64)
    jit = uniform01(seed, 2) - 0.5
    ay = np.deg2rad(deg_y * (1 + 0.2 * jit[0]))
    ax = np.deg2rad(deg_x * (1 + 0.2 * jit[1]))
    Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    return (K @ (Ry @ Rx) @ np.linalg.inv(K)).astype(np.float32)


def two_view_matches(seed, n, outlier_frac=0.3, noise=0.5, w=2880, h=1988):
    """Matched positions of n scene points seen by two cameras a baseline apart (config 5's
    "RANSAC-fed rectification" input): (K1, K2, p1 [n,2], p2 [n,2], F) float32.  A fraction of
    the pairs are mismatches (uniform positions in the second image)."""
    rng = np.random.RandomState(seed)
    f = 0.9 * w
    k1 = np.array([[f, 0, w / 2.0], [0, f, h / 2.0], [0, 0, 1]], np.float64)
    k2 = np.array([[f * 1.01, 0, w / 2.0 + 7], [0, f * 1.01, h / 2.0 - 4], [0, 0, 1]], np.float64)
    ay, ax = np.deg2rad(2.0), np.deg2rad(-0.7)
    ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    r = ry @ rx
    t = np.array([-0.4, 0.02, 0.03])
    x = np.stack([rng.uniform(-2.5, 2.5, n), rng.uniform(-1.7, 1.7, n), rng.uniform(4, 12, n)], axis=1)
    a = (k1 @ x.T).T
    b = (k2 @ (r @ x.T + t[:, None])).T
    p1 = a[:, :2] / a[:, 2:] + rng.normal(0, noise, (n, 2))
    p2 = b[:, :2] / b[:, 2:] + rng.normal(0, noise, (n, 2))
    bad = rng.rand(n) < outlier_frac
    p2[bad] = np.stack([rng.uniform(0, w, bad.sum()), rng.uniform(0, h, bad.sum())], axis=1)
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    fm = np.linalg.inv(k2).T @ (tx @ r) @ np.linalg.inv(k1)
    fm = fm / fm[2, 2]
    # (the products above are transposed views: force C order, [n,2] rows of (x, y))
    c = np.ascontiguousarray
    return c(k1, np.float32), c(k2, np.float32), c(p1, np.float32), c(p2, np.float32), c(fm, np.float32)


def gradient_stand_ins(img):
    """Stand-ins for the CV_8U GaussianBlur + Sobel images the reference makes with OpenCV
    (Features.cpp:684-690; third-party arithmetic, out of scope): a 3x3 box blur and central
    differences, negative values clipped to 0 as ddepth = CV_8U does.  What matters is that they
    are functions of the image, so the two views' descriptors correspond."""
    f = img.astype(np.float32)
    p = np.pad(f, 1, mode="edge")
    blur = sum(p[dy:dy + f.shape[0], dx:dx + f.shape[1]] for dy in range(3) for dx in range(3)) / 9.0
    q = np.pad(blur, 1, mode="edge")
    gx = q[1:-1, 2:] - q[1:-1, :-2]
    gy = q[2:, 1:-1] - q[:-2, 1:-1]
    return (np.clip(gx * 4, 0, 255).astype(np.uint8), np.clip(gy * 4, 0, 255).astype(np.uint8))
