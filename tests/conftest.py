import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kitchen():
    return dict(np.load(os.path.join(GOLDEN, "kitchen.npz")))


@pytest.fixture(scope="session")
def aloe():
    return dict(np.load(os.path.join(GOLDEN, "aloe.npz")))


@pytest.fixture(scope="session")
def calib_files():
    return os.path.join(GOLDEN, "left.yaml"), os.path.join(GOLDEN, "right.yaml")


@pytest.fixture(scope="session")
def calib(calib_files):
    import yaml

    def load(p):
        y = yaml.safe_load(open(p))
        return dict(K=np.array(y["camera_matrix"]["data"], np.float64), D=np.array(y["distortion_coefficients"]["data"], np.float64),
                    R=np.array(y["rectification_matrix"]["data"], np.float64), P=np.array(y["projection_matrix"]["data"], np.float64),
                    width=y["image_width"], height=y["image_height"])
    return dict(left=load(calib_files[0]), right=load(calib_files[1]))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def gpu():
    """torch CUDA device + the HIP library; the tests using it are marked gpu."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ros_gpu_stereo_processor_amd import capi
    capi.lib()  # fails loudly when the HIP library is missing -- no CPU fallback
    return torch.device("cuda:0")
