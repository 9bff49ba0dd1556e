// C++ tests of the header-only mirror classes (include/gpuimageproc/*.hpp) on top of the C-ABI, after the reference's
// gtests (test/UTest.cpp).  Plain asserts, no gtest in the image.  usage: test_processor <dir with raw fixtures>
// The Python harness (tests/test_gpu_cpp.py) dumps the fixtures as raw files and checks the outputs against the oracle.
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>

#include <atomic>
#include <cstring>
#include <mutex>
#include <vector>

#include "gpuimageproc/StereoProcessor.hpp"
#include "gpuimageproc/StreamingStereo.hpp"

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
    // The pipelined host link from C++: the same pair through the streaming API (3 batches in flight) and through TWO handles dealt
    // round-robin (MultiDeviceStereo: one handle per device on an 8-GPU node, two handles on one device here) must deliver, in
    // submission order, exactly what the synchronous imageCb schedule produced above.
    {
        Mat l = read_raw(dir + "/left_raw.bin", 480, 752, GIP_8UC1), r = read_raw(dir + "/right_raw.bin", 480, 752, GIP_8UC1);
        std::ifstream f32(dir + "/out_disp32.bin", std::ios::binary), fc(dir + "/out_cloud.bin", std::ios::binary);
        std::vector<float> exp32((size_t)480 * 752);
        std::vector<unsigned char> expc((size_t)480 * 752 * 32);
        f32.read((char *)exp32.data(), (std::streamsize)(exp32.size() * 4));
        fc.read((char *)expc.data(), (std::streamsize)expc.size());
        gip_camera_info li, ri;
        char name[64];
        CHECK(gip_read_calibration((dir + "/left.yaml").c_str(), &li, name, sizeof name) == GIP_OK);
        CHECK(gip_read_calibration((dir + "/right.yaml").c_str(), &ri, name, sizeof name) == GIP_OK);
        gip_bm_params p{};
        p.prefilter_type = GIP_PREFILTER_XSOBEL; p.prefilter_size = 5; p.prefilter_cap = 31; p.block_size = 15; p.min_disparity = 0; p.num_disparities = 64;
        p.texture_threshold = 10; p.uniqueness_ratio = 0; p.disp12_max_diff = 0;
        std::mutex mu;
        std::vector<uint64_t> order[2];
        std::atomic<int> bad{0};
        const uint32_t outputs = GIP_TOPIC_DISPARITY | GIP_TOPIC_POINTCLOUD;
        MultiDeviceStereo multi(2, li, ri, p, 100, 2.0, 752, 480, "mono8", outputs, [&](int slot, const gip_streamer_result &res) {
            if (std::memcmp(res.disparity_f32, exp32.data(), exp32.size() * 4) != 0) bad++;
            if (std::memcmp(res.pointcloud, expc.data(), expc.size()) != 0) bad++;
            if (res.header.seq != 100 + 2 * res.seq + (uint32_t)slot) bad++;          // handle `slot` sees frames slot, slot + 2, ...
            std::lock_guard<std::mutex> g(mu);
            order[slot].push_back(res.seq);
        });
        CHECK(multi.size() == 2);
        const int nframes = 9;
        for (int i = 0; i < nframes; i++) {
            gip_header hdr{};
            hdr.seq = 100 + i;
            multi.imageCb((uint64_t)i, l, r, hdr);
        }
        multi.flush();
        CHECK(bad == 0);
        CHECK(order[0].size() == 5 && order[1].size() == 4);
        for (int s2 = 0; s2 < 2; s2++)
            for (size_t k = 0; k < order[s2].size(); k++) CHECK(order[s2][k] == k);      // each handle delivers in submission order
    }
    // One frame cut into THREE row tiles (RowSplitStereo: one handle per tile, halo rows by peer copy, border components merged on the device)
    // must give exactly what the synchronous single-handle schedule produced above.
    {
        Mat l = read_raw(dir + "/left_raw.bin", 480, 752, GIP_8UC1), r = read_raw(dir + "/right_raw.bin", 480, 752, GIP_8UC1);
        std::ifstream f16(dir + "/out_disp16.bin", std::ios::binary), f32(dir + "/out_disp32.bin", std::ios::binary), fc(dir + "/out_cloud.bin", std::ios::binary);
        std::vector<int16_t> exp16((size_t)480 * 752);
        std::vector<float> exp32((size_t)480 * 752);
        std::vector<unsigned char> expc((size_t)480 * 752 * 32);
        f16.read((char *)exp16.data(), (std::streamsize)(exp16.size() * 2));
        f32.read((char *)exp32.data(), (std::streamsize)(exp32.size() * 4));
        fc.read((char *)expc.data(), (std::streamsize)expc.size());
        gip_camera_info li, ri;
        char name[64];
        CHECK(gip_read_calibration((dir + "/left.yaml").c_str(), &li, name, sizeof name) == GIP_OK);
        CHECK(gip_read_calibration((dir + "/right.yaml").c_str(), &ri, name, sizeof name) == GIP_OK);
        gip_bm_params p{};
        p.prefilter_type = GIP_PREFILTER_XSOBEL; p.prefilter_size = 5; p.prefilter_cap = 31; p.block_size = 15; p.min_disparity = 0; p.num_disparities = 64;
        p.texture_threshold = 10; p.uniqueness_ratio = 0; p.disp12_max_diff = 0;
        RowSplitStereo split(3, li, ri, p, 100, 2.0, 752, 480, "mono8", GIP_TOPIC_DISPARITY | GIP_TOPIC_POINTCLOUD);
        CHECK(split.tiles() == 3);
        for (int rep = 0; rep < 2; rep++) {
            Mat d16, d32;
            std::vector<uint8_t> cloud;
            split.imageCb(l, r, &d16, &d32, &cloud);
            CHECK(std::memcmp(d16.data, exp16.data(), exp16.size() * 2) == 0);
            CHECK(std::memcmp(d32.data, exp32.data(), exp32.size() * 4) == 0);
            CHECK(std::memcmp(cloud.data(), expc.data(), expc.size()) == 0);
        }
        const std::vector<gip_rowsplit_timing> tm = split.profileSerial();
        CHECK(tm.size() == 3 && tm[1].match_ms > 0.f && tm[1].total_ms >= tm[1].match_ms);
    }
    std::puts("cpp tests ok");
    return 0;
}
