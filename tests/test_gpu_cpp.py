"""GPU test (-m gpu): the C++ mirror classes (include/gpuimageproc/*.hpp) built through the CMake package (library configured, built and
installed by the top-level CMakeLists.txt; the test as a downstream find_package consumer) and driven like the reference's nodelet
(configCb + imageCb, the streaming classes, the row split); outputs are checked against the CPU oracle."""
import os
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_mirror_classes(gpu, oracle, kitchen, calib, calib_files, tmp_path):
    # the CMake package end to end: configure + build + install the library (top-level CMakeLists.txt), then build the test as a DOWNSTREAM
    # project that only knows find_package(gpuimageproc_hip) (tests/cpp/consumer) -- the way a catkin / CMake user consumes it
    cmake = shutil.which("cmake")
    assert cmake, "cmake not found"
    build, prefix, cbuild = tmp_path / "build", tmp_path / "prefix", tmp_path / "consumer_build"
    jobs = str(min(16, os.cpu_count() or 4))

    def run(*cmd):
        out = subprocess.run(list(cmd), capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, " ".join(cmd) + "\n" + out.stdout[-3000:] + out.stderr[-3000:]
    run(cmake, "-S", ROOT, "-B", str(build), "-DCMAKE_PREFIX_PATH=/opt/rocm", "-DGIP_BUILD_ROS=OFF", "-DGIP_BUILD_TESTS=OFF")
    run(cmake, "--build", str(build), "-j", jobs)
    run(cmake, "--install", str(build), "--prefix", str(prefix))
    assert (prefix / "lib" / "cmake" / "gpuimageproc_hip" / "gpuimageproc_hipConfig.cmake").exists() and (prefix / "include" / "gpuimageproc_hip.h").exists()
    # the C++ throughput harness (tools/gip_bench.cpp: C-ABI + HIP runtime only, no Python in the loop) builds with the package and runs
    import json
    gb = subprocess.run([str(build / "gip_bench"), "--batch", "4", "--steps", "6", "--warmup", "2"], capture_output=True, text=True, timeout=300)
    assert gb.returncode == 0, gb.stdout + gb.stderr
    line = json.loads(gb.stdout.strip().splitlines()[-1])
    assert line["value"] > 0 and line["pairs_per_step"] == 4 and line["independent_streams"] >= 2 and line["valid_fraction_first_image"] > 0.1
    run(cmake, "-S", os.path.join(ROOT, "tests", "cpp", "consumer"), "-B", str(cbuild), "-DCMAKE_PREFIX_PATH=%s;/opt/rocm" % prefix)
    run(cmake, "--build", str(cbuild), "-j", jobs)
    exe = cbuild / "test_processor"
    kitchen["left_raw"].tofile(tmp_path / "left_raw.bin")
    kitchen["right_raw"].tofile(tmp_path / "right_raw.bin")
    shutil.copy(calib_files[0], tmp_path / "left.yaml")
    shutil.copy(calib_files[1], tmp_path / "right.yaml")
    out = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "cpp tests ok" in out.stdout
    O = oracle
    ml = O.init_undistort_rectify_map(calib["left"]["K"], calib["left"]["D"], calib["left"]["R"], calib["left"]["P"], 752, 480)
    mr = O.init_undistort_rectify_map(calib["right"]["K"], calib["right"]["D"], calib["right"]["R"], calib["right"]["P"], 752, 480)
    rl, rr = O.remap(kitchen["left_raw"], *ml), O.remap(kitchen["right_raw"], *mr)
    # configCb: xsobel, window 15, range 64, texture 10, disparity_min 0; ctor defaults: prefilter size 5, uniqueness 0, disp12 0
    po = O.bm_params(prefilter_type=1, prefilter_size=5, prefilter_cap=31, block_size=15, min_disparity=0, num_disparities=64, texture_threshold=10,
                     uniqueness_ratio=0, disp12_max_diff=0)
    exp = O.filter_speckles(O.stereobm(rl, rr, po), -16, 100, 32)
    d16 = np.fromfile(tmp_path / "out_disp16.bin", np.int16).reshape(480, 752)
    assert np.array_equal(d16, exp)
    d32 = np.fromfile(tmp_path / "out_disp32.bin", np.float32).reshape(480, 752)
    assert np.array_equal(d32, O.disparity_to_f32(exp, calib["left"]["P"][2], calib["right"]["P"][2]))
    cloud = np.fromfile(tmp_path / "out_cloud.bin", np.uint8).reshape(480, 752, 32)
    xyz = cloud[..., :12].copy().view(np.float32)
    epc = O.pack_pointcloud2(O.reproject_to_3d(d32, O.stereo_q(calib["left"]["P"], calib["right"]["P"]), True), np.stack([rl] * 3, -1))
    exyz = epc[..., :12].copy().view(np.float32)
    assert np.array_equal(np.isnan(xyz), np.isnan(exyz))      # FILTERED pixels and zero disparities (Z = inf) are NaN
    ok = ~np.isnan(exyz)
    assert np.allclose(xyz[ok], exyz[ok], rtol=2.4e-7, atol=0)
    assert (np.isnan(xyz[..., 0]) >= (exp == -16)).all()
    assert np.array_equal(cloud[..., 16], rl) and np.array_equal(cloud[..., 17], rl) and np.array_equal(cloud[..., 18], rl)
