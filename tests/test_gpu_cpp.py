"""GPU test (-m gpu): the C++ mirror classes (include/gpuimageproc/*.hpp) compiled against libgpuimageproc_hip.so and driven
like the reference's nodelet (configCb + imageCb); outputs are checked against the CPU oracle."""
import os
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_mirror_classes(gpu, oracle, kitchen, calib, calib_files, tmp_path):
    from ros_gpu_stereo_processor_amd import capi
    libdir = os.path.dirname(capi.LIB_PATH)
    exe = tmp_path / "test_processor"
    cxx = shutil.which("hipcc") or shutil.which("g++")
    cmd = [cxx, "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_processor.cpp"), "-L", libdir,
           "-lgpuimageproc_hip", "-Wl,-rpath," + libdir, "-o", str(exe)]
    if "hipcc" not in cxx:
        cmd += ["-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
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
