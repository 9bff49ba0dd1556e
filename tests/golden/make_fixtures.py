#!/usr/bin/env python3
"""Convert the reference's own test fixtures (data only: PNG images + calibration YAML,
/root/reference/test/stereobm/test_data/, BSD per package.xml:16) into the fixtures the
test-suite reads.  Run once in the build container (the reference tree does not exist on the
GPU box); outputs are committed.

  kitchen.npz : left/right-0022.png (raw 752x480 mono8) and *_rect.png (CPU-rectified goldens,
                asserted bit-exact by the reference's RectifyMonoCpu test, test/UTest.cpp:247-260)
  aloe.npz    : aloe-L/R.png (RGB), aloe-L-Mono.png (gray golden), aloe-disp.png (legacy CUDA SSD
                output; loaded at UTest.cpp:101 but never asserted)
  left.yaml / right.yaml : verbatim calibration data files.
"""
import os, shutil, sys
import numpy as np
from PIL import Image

SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/test/stereobm/test_data"
DST = os.path.dirname(os.path.abspath(__file__))

def load(name):
    im = Image.open(os.path.join(SRC, name))
    a = np.asarray(im)
    print(name, im.mode, a.shape, a.dtype)
    return a

np.savez_compressed(os.path.join(DST, "kitchen.npz"),
                    left_raw=load("left-0022.png"), right_raw=load("right-0022.png"),
                    left_rect=load("left-0022_rect.png"), right_rect=load("right-0022_rect.png"))
np.savez_compressed(os.path.join(DST, "aloe.npz"),
                    left_rgb=load("aloe-L.png"), right_rgb=load("aloe-R.png"),
                    left_mono=load("aloe-L-Mono.png"), disp=load("aloe-disp.png"))
for y in ("left.yaml", "right.yaml"):
    shutil.copyfile(os.path.join(SRC, y), os.path.join(DST, y))
