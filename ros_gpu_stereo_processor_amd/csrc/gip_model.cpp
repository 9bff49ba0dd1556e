// gip_model.cpp -- see gip_model.h.  Host-only C++ (no HIP).
#include "gip_model.h"

#include <cctype>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace {

// Inverse of a 3x3 by Gaussian elimination with partial pivoting on the augmented matrix (double).
bool invert3x3(const double a[9], double out[9])
{
    double m[3][6];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            m[i][j] = a[i * 3 + j];
            m[i][j + 3] = (i == j) ? 1.0 : 0.0;
        }
    for (int i = 0; i < 3; i++) {
        int piv = i;
        for (int j = i + 1; j < 3; j++)
            if (std::fabs(m[j][i]) > std::fabs(m[piv][i])) piv = j;
        if (std::fabs(m[piv][i]) < 1e-300) return false;
        if (piv != i)
            for (int j = 0; j < 6; j++) std::swap(m[i][j], m[piv][j]);
        const double d = -1.0 / m[i][i];
        for (int j = i + 1; j < 3; j++) {
            const double alpha = m[j][i] * d;
            for (int c = i + 1; c < 6; c++) m[j][c] += alpha * m[i][c];
        }
    }
    for (int i = 2; i >= 0; i--)
        for (int c = 3; c < 6; c++) {
            double s = m[i][c];
            for (int k = i + 1; k < 3; k++) s -= m[i][k] * m[k][c];
            m[i][c] = s / m[i][i];
        }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) out[i * 3 + j] = m[i][j + 3];
    return true;
}

inline int round_sat(double v)
{
    if (v >= 2147483647.0) return INT_MAX;
    if (v <= -2147483648.0) return INT_MIN;
    return (int)std::lrint(v);  // round half to even, like cvRound
}

}  // namespace

// cv::initUndistortRectifyMap(K, D, R, P, size, CV_16SC2): for every rectified pixel, project back through
// inv(P[:,:3] * R), apply the plumb-bob / rational distortion, and store the source coordinate with 5 fractional bits.
// image_geometry::PinholeCameraModel::fromCameraInfo: K, P of the images actually delivered
bool gip_effective_camera(const gip_camera_info& full, gip_camera_info& eff)
{
    eff = full;
    const int bx = full.binning_x ? (int)full.binning_x : 1, by = full.binning_y ? (int)full.binning_y : 1;
    int rx = (int)full.roi_x_offset, ry = (int)full.roi_y_offset, rw = (int)full.roi_width, rh = (int)full.roi_height;
    if (rx == 0 && ry == 0 && rw == 0 && rh == 0) { rw = full.width; rh = full.height; }   // all zero = full resolution
    if (rw <= 0 || rh <= 0 || rx < 0 || ry < 0 || rx + rw > full.width || ry + rh > full.height) return false;
    if (rx != 0 || ry != 0) {
        eff.K[2] -= rx; eff.K[5] -= ry;
        eff.P[2] -= rx; eff.P[6] -= ry;
    }
    if (bx > 1) {
        const double sx = 1.0 / bx;
        eff.K[0] *= sx; eff.K[2] *= sx;
        eff.P[0] *= sx; eff.P[2] *= sx; eff.P[3] *= sx;
    }
    if (by > 1) {
        const double sy = 1.0 / by;
        eff.K[4] *= sy; eff.K[5] *= sy;
        eff.P[5] *= sy; eff.P[6] *= sy; eff.P[7] *= sy;
    }
    eff.width = rw / bx;
    eff.height = rh / by;
    eff.binning_x = eff.binning_y = 0;
    eff.roi_x_offset = eff.roi_y_offset = eff.roi_width = eff.roi_height = 0;
    return eff.width > 0 && eff.height > 0;
}

static void build_full_map(const gip_camera_info& ci, std::vector<int16_t>& map1, std::vector<uint16_t>& map2);

// image_geometry::PinholeCameraModel::initRectificationMaps: tables of the BINNED FULL frame (K, P scaled by 1 / binning, not
// shifted), then the ROI window of them with the ROI origin subtracted from the integer coordinates
void gip_build_rectify_map(const gip_camera_info& ci, std::vector<int16_t>& map1, std::vector<uint16_t>& map2)
{
    const int bx = ci.binning_x ? (int)ci.binning_x : 1, by = ci.binning_y ? (int)ci.binning_y : 1;
    int rx = (int)ci.roi_x_offset, ry = (int)ci.roi_y_offset, rw = (int)ci.roi_width, rh = (int)ci.roi_height;
    const bool full_roi = (rx == 0 && ry == 0 && rw == 0 && rh == 0) || (rx == 0 && ry == 0 && rw == ci.width && rh == ci.height);
    if (bx == 1 && by == 1 && full_roi) { build_full_map(ci, map1, map2); return; }
    gip_camera_info b = ci;
    b.binning_x = b.binning_y = 0;
    b.roi_x_offset = b.roi_y_offset = b.roi_width = b.roi_height = 0;
    b.width = ci.width / bx;
    b.height = ci.height / by;
    const double sx = 1.0 / bx, sy = 1.0 / by;
    if (bx > 1) { b.K[0] *= sx; b.K[2] *= sx; b.P[0] *= sx; b.P[2] *= sx; b.P[3] *= sx; }
    if (by > 1) { b.K[4] *= sy; b.K[5] *= sy; b.P[5] *= sy; b.P[6] *= sy; b.P[7] *= sy; }
    std::vector<int16_t> f1;
    std::vector<uint16_t> f2;
    build_full_map(b, f1, f2);
    if (full_roi) { map1.swap(f1); map2.swap(f2); return; }
    rx /= bx; ry /= by; rw /= bx; rh /= by;
    map1.resize((size_t)rw * rh * 2);
    map2.resize((size_t)rw * rh);
    for (int y = 0; y < rh; y++)
        for (int x = 0; x < rw; x++) {
            const size_t o = (size_t)(y + ry) * b.width + (x + rx), d = (size_t)y * rw + x;
            map1[2 * d] = (int16_t)(f1[2 * o] - rx);
            map1[2 * d + 1] = (int16_t)(f1[2 * o + 1] - ry);
            map2[d] = f2[o];
        }
}

static void build_full_map(const gip_camera_info& ci, std::vector<int16_t>& map1, std::vector<uint16_t>& map2)
{
    const int w = ci.width, h = ci.height, nD = ci.nD;
    const double* D = ci.D;
    const double k1 = nD > 0 ? D[0] : 0, k2 = nD > 1 ? D[1] : 0, p1 = nD > 2 ? D[2] : 0, p2 = nD > 3 ? D[3] : 0;
    const double k3 = nD > 4 ? D[4] : 0, k4 = nD > 5 ? D[5] : 0, k5 = nD > 6 ? D[6] : 0, k6 = nD > 7 ? D[7] : 0;
    const double fx = ci.K[0], fy = ci.K[4], u0 = ci.K[2], v0 = ci.K[5];
    double PR[9], ir[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += ci.P[i * 4 + k] * ci.R[k * 3 + j];
            PR[i * 3 + j] = s;
        }
    if (!invert3x3(PR, ir)) std::memset(ir, 0, sizeof ir);
    map1.resize((size_t)w * h * 2);
    map2.resize((size_t)w * h);
    for (int i = 0; i < h; i++) {
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        for (int j = 0; j < w; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
            const double iw = 1. / _w, x = _x * iw, y = _y * iw;
            const double x2 = x * x, y2 = y * y;
            const double r2 = x2 + y2, _2xy = 2 * x * y;
            const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
            const double xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2);
            const double yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy;
            const double u = fx * xd + u0, v = fy * yd + v0;
            const int iu = round_sat(u * 32.0), iv = round_sat(v * 32.0);
            const size_t o = (size_t)i * w + j;
            map1[2 * o] = (int16_t)(iu >> 5);
            map1[2 * o + 1] = (int16_t)(iv >> 5);
            map2[o] = (uint16_t)((iv & 31) * 32 + (iu & 31));
        }
    }
}

int gip_model_init(GipStereoModel& m, const gip_camera_info& l, const gip_camera_info& r)
{
    if (l.width <= 0 || l.height <= 0 || l.width > 32767 || l.height > 32767 || r.width <= 0 || r.height <= 0 || r.width > 32767 || r.height > 32767) return GIP_ERR_INVALID_ARG;
    if (l.nD < 0 || l.nD > 8 || r.nD < 0 || r.nD > 8) return GIP_ERR_INVALID_ARG;
    if (r.P[0] == 0.0 || l.P[0] == 0.0) return GIP_ERR_INVALID_ARG;
    gip_camera_info le, re;
    if (!gip_effective_camera(l, le) || !gip_effective_camera(r, re)) return GIP_ERR_INVALID_ARG;
    if (le.width != re.width || le.height != re.height) return GIP_ERR_INVALID_ARG;   // both eyes must deliver images of one size
    m.left.full = l;
    m.right.full = r;
    m.left.info = le;
    m.right.info = re;
    GipPinhole* cams[2] = {&m.left, &m.right};
    for (GipPinhole* c : cams) {
        c->distorted = !(c->info.nD == 0 || c->info.D[0] == 0.0);
        gip_build_rectify_map(c->full, c->map1, c->map2);
    }
    // image_geometry::StereoCameraModel::updateQ
    const double Tx = -m.baseline();
    std::memset(m.Q, 0, sizeof m.Q);
    m.Q[0 * 4 + 0] = m.left.fy() * Tx;
    m.Q[0 * 4 + 3] = -m.left.fy() * m.left.cx() * Tx;
    m.Q[1 * 4 + 1] = m.left.fx() * Tx;
    m.Q[1 * 4 + 3] = -m.left.fx() * m.left.cy() * Tx;
    m.Q[2 * 4 + 3] = m.left.fx() * m.left.fy() * Tx;
    m.Q[3 * 4 + 2] = -m.left.fy();
    m.Q[3 * 4 + 3] = m.left.fy() * (m.left.cx() - m.right.cx());
    m.initialised = true;
    return GIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Minimal reader for the camera_calibration_parsers YAML layout:
//   image_width: N / image_height: N / camera_name: S
//   camera_matrix | distortion_coefficients | rectification_matrix | projection_matrix: {rows, cols, data: [ ... ]}
namespace {

bool find_scalar(const std::string& text, const std::string& key, std::string& value)
{
    size_t pos = 0;
    while ((pos = text.find(key, pos)) != std::string::npos) {
        const bool at_line_start = (pos == 0 || text[pos - 1] == '\n');
        size_t colon = pos + key.size();
        if (at_line_start && colon < text.size() && text[colon] == ':') {
            size_t e = text.find('\n', colon);
            value = text.substr(colon + 1, e == std::string::npos ? std::string::npos : e - colon - 1);
            size_t a = value.find_first_not_of(" \t\r\"'"), b = value.find_last_not_of(" \t\r\"'");
            value = (a == std::string::npos) ? "" : value.substr(a, b - a + 1);
            return true;
        }
        pos += key.size();
    }
    return false;
}

bool find_matrix(const std::string& text, const std::string& key, std::vector<double>& data)
{
    size_t pos = text.find(key + ":");
    if (pos == std::string::npos) return false;
    size_t dpos = text.find("data:", pos);
    if (dpos == std::string::npos) return false;
    size_t lb = text.find('[', dpos), rb = text.find(']', dpos);
    if (lb == std::string::npos || rb == std::string::npos || rb < lb) return false;
    std::string body = text.substr(lb + 1, rb - lb - 1);
    for (char& ch : body)
        if (ch == ',' || ch == '\n' || ch == '\r') ch = ' ';
    std::istringstream ss(body);
    data.clear();
    std::string tok;
    while (ss >> tok) {
        char* end = nullptr;
        double v = std::strtod(tok.c_str(), &end);
        if (end == tok.c_str()) return false;
        data.push_back(v);
    }
    return true;
}

}  // namespace

// camera_calibration_parsers INI layout ("# oST version 5.0 parameters"):
//   [image] / width / <n> / height / <n> / [<camera name>] / camera matrix / 3x3 / distortion / 1xN / rectification / 3x3 / projection / 3x4
static bool ini_numbers_after(const std::string& text, const std::string& label, size_t max_count, std::vector<double>& out)
{
    size_t pos = text.find(label);
    if (pos == std::string::npos) return false;
    std::istringstream ss(text.substr(pos + label.size()));
    out.clear();
    std::string tok;
    while (out.size() < max_count && ss >> tok) {
        char* end = nullptr;
        double v = std::strtod(tok.c_str(), &end);
        if (end == tok.c_str() || *end != '\0') break;   // next label or section
        out.push_back(v);
    }
    return !out.empty();
}

static int parse_calibration_ini(const std::string& path, const std::string& text, gip_camera_info& out, std::string& camera_name, std::string& err)
{
    std::memset(&out, 0, sizeof out);
    std::vector<double> d;
    if (!ini_numbers_after(text, "width", 1, d)) { err = "width missing in " + path; return GIP_ERR_IO; }
    out.width = (int)d[0];
    if (!ini_numbers_after(text, "height", 1, d)) { err = "height missing in " + path; return GIP_ERR_IO; }
    out.height = (int)d[0];
    camera_name.clear();
    size_t pos = 0;
    while ((pos = text.find('[', pos)) != std::string::npos) {   // the section that is not [image] names the camera
        size_t e = text.find(']', pos);
        if (e == std::string::npos) break;
        std::string sec = text.substr(pos + 1, e - pos - 1);
        if (sec != "image") { camera_name = sec; break; }
        pos = e + 1;
    }
    if (!ini_numbers_after(text, "camera matrix", 9, d) || d.size() != 9) { err = "camera matrix malformed in " + path; return GIP_ERR_IO; }
    std::memcpy(out.K, d.data(), sizeof out.K);
    if (!ini_numbers_after(text, "distortion", 8, d)) { err = "distortion malformed in " + path; return GIP_ERR_IO; }
    out.nD = (int)d.size();
    for (size_t i = 0; i < d.size(); i++) out.D[i] = d[i];
    if (!ini_numbers_after(text, "rectification", 9, d) || d.size() != 9) { err = "rectification malformed in " + path; return GIP_ERR_IO; }
    std::memcpy(out.R, d.data(), sizeof out.R);
    if (!ini_numbers_after(text, "projection", 12, d) || d.size() != 12) { err = "projection malformed in " + path; return GIP_ERR_IO; }
    std::memcpy(out.P, d.data(), sizeof out.P);
    return GIP_OK;
}

// camera_calibration_parsers::readCalibration picks the format by extension (.ini, .yml/.yaml)
int gip_parse_calibration_yaml(const std::string& path, gip_camera_info& out, std::string& camera_name, std::string& err)
{
    out = gip_camera_info{};   // calibration files carry no binning / ROI: full frame
    std::ifstream f(path);
    if (!f) { err = "cannot open calibration file " + path; return GIP_ERR_IO; }
    std::stringstream buf;
    buf << f.rdbuf();
    const std::string text = buf.str();
    if (path.size() >= 4 && path.compare(path.size() - 4, 4, ".ini") == 0) return parse_calibration_ini(path, text, out, camera_name, err);
    std::memset(&out, 0, sizeof out);
    std::string v;
    if (!find_scalar(text, "image_width", v)) { err = "image_width missing in " + path; return GIP_ERR_IO; }
    out.width = std::atoi(v.c_str());
    if (!find_scalar(text, "image_height", v)) { err = "image_height missing in " + path; return GIP_ERR_IO; }
    out.height = std::atoi(v.c_str());
    camera_name.clear();
    if (find_scalar(text, "camera_name", v)) camera_name = v;
    std::vector<double> d;
    if (!find_matrix(text, "camera_matrix", d) || d.size() != 9) { err = "camera_matrix malformed in " + path; return GIP_ERR_IO; }
    std::memcpy(out.K, d.data(), sizeof out.K);
    if (!find_matrix(text, "distortion_coefficients", d) || d.size() > 8) { err = "distortion_coefficients malformed in " + path; return GIP_ERR_IO; }
    out.nD = (int)d.size();
    for (size_t i = 0; i < d.size(); i++) out.D[i] = d[i];
    if (!find_matrix(text, "rectification_matrix", d) || d.size() != 9) { err = "rectification_matrix malformed in " + path; return GIP_ERR_IO; }
    std::memcpy(out.R, d.data(), sizeof out.R);
    if (!find_matrix(text, "projection_matrix", d) || d.size() != 12) { err = "projection_matrix malformed in " + path; return GIP_ERR_IO; }
    std::memcpy(out.P, d.data(), sizeof out.P);
    return GIP_OK;
}
