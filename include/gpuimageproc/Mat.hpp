// Mat.hpp -- a minimal row-major image container standing in for cv::Mat at the C++ boundary (OpenCV is not a dependency
// of the HIP implementation).  Type codes follow OpenCV's numbering (GIP_8UC1 ...), so call sites port 1:1.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "gpuimageproc_hip.h"

namespace gpuimageproc
{

class Mat
{
  public:
    int rows = 0, cols = 0, type_ = GIP_8UC1;
    size_t step = 0;
    uint8_t *data = nullptr;

    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    // view onto foreign memory (no ownership), like cv::Mat(rows, cols, type, ptr, step)
    Mat(int r, int c, int type, void *ptr, size_t step_bytes = 0) : rows(r), cols(c), type_(type), step(step_bytes ? step_bytes : (size_t)c * elemSize(type)), data((uint8_t *)ptr) {}

    static int elemSize(int type)
    {
        const int depth = type & 7, cn = (type >> 3) + 1;
        const int ds = depth == GIP_8U ? 1 : (depth == GIP_16U || depth == GIP_16S) ? 2 : depth == GIP_32F ? 4 : 0;
        return ds * cn;
    }
    void create(int r, int c, int type)
    {
        rows = r; cols = c; type_ = type;
        step = (size_t)c * elemSize(type);
        store_ = std::make_shared<std::vector<uint8_t>>(step * (size_t)r);
        data = store_->data();
    }
    int type() const { return type_; }
    int channels() const { return (type_ >> 3) + 1; }
    size_t elemSize() const { return (size_t)elemSize(type_); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    template <typename T> T *ptr(int r = 0) { return (T *)(data + (size_t)r * step); }
    template <typename T> const T *ptr(int r = 0) const { return (const T *)(data + (size_t)r * step); }

  private:
    std::shared_ptr<std::vector<uint8_t>> store_;
};

inline bool mat_are_same(const Mat &a, const Mat &b)  // test/UTest.cpp:104-116
{
    if (a.rows != b.rows || a.cols != b.cols || a.type() != b.type()) return false;
    const size_t rb = (size_t)a.cols * a.elemSize();
    for (int y = 0; y < a.rows; y++)
        if (std::memcmp(a.data + (size_t)y * a.step, b.data + (size_t)y * b.step, rb) != 0) return false;
    return true;
}

} // namespace gpuimageproc
