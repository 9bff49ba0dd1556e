// C++ tests of the header-only mirror classes (include/gpuimageproc/*.hpp) on top of the C-ABI, after the reference's
// gtests (test/UTest.cpp).  Plain asserts, no gtest in the image.  usage: test_processor <dir with raw fixtures>
// The Python harness (tests/test_gpu_cpp.py) dumps the fixtures as raw files and checks the outputs against the oracle.
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>

#include "gpuimageproc/StereoProcessor.hpp"

using namespace gpuimageproc;

static Mat read_raw(const std::string &path, int rows, int cols, int type)
{
    Mat m(rows, cols, type);
    std::ifstream f(path, std::ios::binary);
    if (!f) { std::cerr << "cannot open " << path << "\n"; std::exit(2); }
    f.read((char *)m.data, (std::streamsize)(m.step * rows));
    return m;
}
static void write_raw(const std::string &path, const Mat &m)
{
    std::ofstream f(path, std::ios::binary);
    for (int y = 0; y < m.rows; y++) f.write((const char *)m.data + (size_t)y * m.step, (std::streamsize)((size_t)m.cols * m.elemSize()));
}
#define CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); std::exit(1); } } while (0)

int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : ".";
    // GpuTransfer (UTest.cpp:179-188)
    {
        GpuStereoProcessor sp;
        Mat img = read_raw(dir + "/left_raw.bin", 480, 752, GIP_8UC1);
        sp.uploadMat(GPU_MAT_SRC_L_RECT_MONO, img);
        sp.waitForStream(GPU_MAT_SRC_L_RECT_MONO);
        Mat back;
        sp.downloadMat(GPU_MAT_SRC_L_RECT_MONO, back);
        CHECK(mat_are_same(img, back));
    }
    // GpuColorConversionColorMono (UTest.cpp:190-207)
    {
        GpuStereoProcessor sp;
        Mat bgr(1, 1, GIP_8UC3);
        bgr.data[0] = 10; bgr.data[1] = 20; bgr.data[2] = 30;
        sp.uploadMat(GPU_MAT_SRC_L_RAW, bgr, "bgr8");
        sp.convertRawToColor(GPU_MAT_SIDE_L);
        sp.convertRawToMono(GPU_MAT_SIDE_L);
        sp.waitForStream(GPU_MAT_SIDE_L);
        Mat color, mono;
        sp.downloadMat(GPU_MAT_SRC_L_COLOR, color);
        sp.downloadMat(GPU_MAT_SRC_L_MONO, mono);
        CHECK(mono.data[0] == 22 && color.data[0] == 10 && color.data[1] == 20 && color.data[2] == 30);
    }
    // the reference asserts on an uninitialised model; here the call throws
    {
        GpuStereoProcessor sp;
        Mat img(8, 8, GIP_8UC1);
        sp.uploadMat(GPU_MAT_SRC_L_RAW, img, "mono8");
        bool threw = false;
        try { sp.rectifyImage(GPU_MAT_SRC_L_RAW, GPU_MAT_SRC_L_RECT_MONO); } catch (const std::runtime_error &) { threw = true; }
        CHECK(threw);
    }
    // The nodelet's per-frame path: configCb + imageCb with Disparity, DisparityVis and Pointcloud connected; outputs are
    // written next to the inputs for the Python side to compare with the oracle.
    {
        StereoProcessor node;
        GPUConfig cfg;
        cfg.xsobel = true;
        cfg.correlation_window_size = 14;   // becomes 15
        cfg.disparity_range = 70;           // becomes 64
        cfg.texture_threshold = 10;
        cfg.max_speckle_size = 100;
        cfg.max_speckle_diff = 2;
        node.configCb(cfg);
        CHECK(cfg.correlation_window_size == 15 && cfg.disparity_range == 64);
        node.processor().initStereoModel(dir + "/left.yaml", dir + "/right.yaml");
        CHECK(node.processor().isStereoModelInitialised());
        node.connected().Disparity(true);
        node.connected().DisparityVis(true);
        node.connected().Pointcloud(true);
        CHECK(node.connected().level() == 11);
        int n_disp = 0, n_pc = 0, n_vis = 0;
        Mat disp32(480, 752, GIP_32FC1), cloud(480, 752 * 8, GIP_32FC1);
        node.pub_disparity_ = [&](const gip_disparity_msg &m) {
            n_disp++;
            CHECK(m.image.height == 480 && m.image.width == 752 && m.image.step == 752 * 4 && std::string(m.image.encoding) == "32FC1");
            std::memcpy(disp32.data, m.image.data, m.image.data_size);
        };
        node.pub_pointcloud_ = [&](const gip_pointcloud2_msg &m) {
            n_pc++;
            CHECK(m.point_step == 32 && m.row_step == 32 * 752 && m.nfields == 4 && m.is_dense == 0);
            std::memcpy(cloud.data, m.data, m.data_size);
        };
        node.pub_disparity_vis_ = [&](const gip_image_msg &m) {
            n_vis++;
            CHECK(std::string(m.encoding) == "bgra8" && m.step == 752 * 4);
        };
        Mat l = read_raw(dir + "/left_raw.bin", 480, 752, GIP_8UC1), r = read_raw(dir + "/right_raw.bin", 480, 752, GIP_8UC1);
        gip_header hdr{};
        hdr.seq = 7;
        std::snprintf(hdr.frame_id, sizeof hdr.frame_id, "left_optical");
        node.imageCb(l, r, "mono8", hdr);
        CHECK(n_disp == 1 && n_pc == 1 && n_vis == 1);
        Mat d16;
        node.processor().downloadMat(GPU_MAT_SRC_L_DISPARITY, d16);
        write_raw(dir + "/out_disp16.bin", d16);
        write_raw(dir + "/out_disp32.bin", disp32);
        write_raw(dir + "/out_cloud.bin", cloud);
    }
    std::puts("cpp tests ok");
    return 0;
}
