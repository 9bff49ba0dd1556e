// gip_model.h -- stereo camera model (host side): calibration ingest, rectification tables, Q matrix.
//
// Mirrors what the reference obtains from image_geometry::StereoCameraModel in
// GpuStereoProcessor::initStereoModel (reference src/GPUStereoProcessor.cpp:41-61):
//   PinholeCameraModel::initRectificationMaps -> cv::initUndistortRectifyMap(K, D, R, P, size, CV_16SC2)
//   StereoCameraModel::updateQ / baseline()
// The tables are computed once on the host in double precision, then live in HBM.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

#include "gpuimageproc_hip.h"

struct GipPinhole {
    gip_camera_info info{};            // EFFECTIVE camera: image size, K and P after binning / ROI (binning, roi fields cleared)
    gip_camera_info full{};            // as given
    bool distorted = false;            // image_geometry: D empty or D[0] == 0 -> NONE (rectifyImage copies)
    std::vector<int16_t> map1;         // w*h*2
    std::vector<uint16_t> map2;        // w*h
    double fx() const { return info.P[0]; }
    double fy() const { return info.P[5]; }
    double cx() const { return info.P[2]; }
    double cy() const { return info.P[6]; }
    double Tx() const { return info.P[3]; }
};

struct GipStereoModel {
    GipPinhole left, right;
    bool initialised = false;
    double Q[16] = {0};
    double baseline() const { return -right.Tx() / right.fx(); }
};

// builds maps + Q; returns GIP_OK or GIP_ERR_INVALID_ARG
int gip_model_init(GipStereoModel& m, const gip_camera_info& l, const gip_camera_info& r);
// camera_calibration_parsers YAML (the format of test/stereobm/test_data/left.yaml)
int gip_parse_calibration_yaml(const std::string& path, gip_camera_info& out, std::string& camera_name, std::string& err);
// tables for the images the camera delivers (ROI window of the binned full-frame tables, image_geometry initRectificationMaps)
void gip_build_rectify_map(const gip_camera_info& ci, std::vector<int16_t>& map1, std::vector<uint16_t>& map2);
// the camera after binning / ROI (fromCameraInfo): false when the ROI or binning is inconsistent with the full resolution
bool gip_effective_camera(const gip_camera_info& full, gip_camera_info& eff);
