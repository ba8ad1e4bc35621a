// TEST-ONLY stand-in for the handful of cv:: names stereo_dropin.hpp touches.
// It exists so that OUR adapter can be compiled and run in an image without
// OpenCV; it is never used to build reference sources.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#define CV_8U 0
namespace cv {
struct Size { int width, height; Size(int w, int h) : width(w), height(h) {} };
struct Point2f { float x = 0, y = 0; };
class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;
    uint8_t *data = nullptr;
    Mat() {}
    Mat(Size s, int /*type*/) { create(s.height, s.width, 0); }
    Mat(int r, int c, size_t pitch) { create(r, c, pitch); }
    static Mat zeros(Size s, int type) { Mat m(s, type); if (m.data) std::memset(m.data, 0, m.step * m.rows); return m; }
    template <typename T> T &at(int y, int x) { return data[(size_t)y * step + x]; }
    template <typename T> const T &at(int y, int x) const { return data[(size_t)y * step + x]; }
private:
    void create(int r, int c, size_t pitch) {
        rows = r; cols = c; step = pitch ? pitch : (size_t)c;
        if (r > 0 && c > 0) { buf.reset(new uint8_t[step * r], std::default_delete<uint8_t[]>()); data = buf.get(); }
    }
    std::shared_ptr<uint8_t> buf;
};
}  // namespace cv
