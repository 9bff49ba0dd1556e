"""CPU tests: the C-ABI library builds for gfx950, loads without a GPU and exports every symbol include/*.h declares
(no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "gpuimageproc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(gipk?_[a-z0-9_]+)\s*\(", text)
    return sorted(set(n for n in names if not n.endswith("_cb")))


@pytest.fixture(scope="module")
def built_lib():
    from ros_gpu_stereo_processor_amd import build
    return build.build()


def test_library_exports_every_declared_symbol(built_lib):
    lib = ctypes.CDLL(built_lib)
    declared = _declared_symbols()
    assert len(declared) > 60
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing


def test_python_prototypes_cover_the_header(built_lib):
    from ros_gpu_stereo_processor_amd import capi
    assert sorted(capi.PROTOTYPES) == _declared_symbols()
    capi.lib()  # resolves every prototype against the .so


def test_no_gpu_paths_fail_loudly(built_lib):
    """Without a device the handle cannot be created; nothing falls back to the CPU."""
    import torch
    from ros_gpu_stereo_processor_amd import capi
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = ctypes.c_void_p()
    assert capi.lib().gip_create(-1, ctypes.byref(h)) == capi.ERR_NO_DEVICE
    assert capi.lib().gip_device_count() == 0
    assert capi.lib().gip_status_string(capi.ERR_UNSUPPORTED) == b"unsupported configuration"


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "ros_gpu_stereo_processor_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "gip_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f
    for f in os.listdir(os.path.join(ROOT, "include")):
        p = os.path.join(ROOT, "include", f)
        if os.path.isfile(p):
            assert "gip_oracle" not in open(p).read()


def test_calibration_yaml_reader(built_lib, calib_files, calib):
    from ros_gpu_stereo_processor_amd import processor as P
    import numpy as np
    for path, side in zip(calib_files, ("left", "right")):
        ci, name = P.read_calibration(path)
        assert (ci.width, ci.height, ci.nD) == (752, 480, 5)
        assert name == "narrow_stereo/" + side
        assert np.array_equal(np.array(ci.K[:]), calib[side]["K"])
        assert np.array_equal(np.array(ci.D[:5]), calib[side]["D"])
        assert np.array_equal(np.array(ci.R[:]), calib[side]["R"])
        assert np.array_equal(np.array(ci.P[:]), calib[side]["P"])


def test_calibration_ini_reader(built_lib, calib, tmp_path):
    """camera_calibration_parsers' INI layout (readCalibration picks the parser by extension)"""
    from ros_gpu_stereo_processor_amd import processor as P
    import numpy as np
    c = calib["left"]
    fmt = lambda a, n: "\n".join(" ".join("%.6f" % v for v in a[i:i + n]) for i in range(0, len(a), n))  # noqa: E731
    text = ("# oST version 5.0 parameters\n\n\n[image]\n\nwidth\n752\n\nheight\n480\n\n[narrow_stereo/left]\n\ncamera matrix\n%s\n\ndistortion\n%s\n\n"
            "rectification\n%s\n\nprojection\n%s\n\n" % (fmt(c["K"], 3), fmt(c["D"], 5), fmt(c["R"], 3), fmt(c["P"], 4)))
    path = tmp_path / "left.ini"
    path.write_text(text)
    ci, name = P.read_calibration(str(path))
    assert (ci.width, ci.height, ci.nD, name) == (752, 480, 5, "narrow_stereo/left")
    assert np.array_equal(np.array(ci.K[:]), c["K"]) and np.array_equal(np.array(ci.D[:5]), c["D"])
    assert np.array_equal(np.array(ci.R[:]), c["R"]) and np.array_equal(np.array(ci.P[:]), c["P"])
