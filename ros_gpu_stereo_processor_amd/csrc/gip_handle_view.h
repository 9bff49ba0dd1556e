// gip_handle_view.h -- read-only view of a processor handle for the library's own pipelines (gip_rowsplit.hip): the device-resident
// rectification tables, the model constants and the parameter block, without exposing struct gip_processor.  Not part of the ABI.
#pragma once
#include "gip_internal.h"

struct GipHandleView {
    int device;
    int width, height;              // size of the rectification tables
    const int16_t* map1[2];         // device, [0] left, [1] right
    const uint16_t* map2[2];
    bool distorted[2];              // false: image_geometry's distortion state NONE (rectify = copy)
    double Q[16], cx_l, cx_r;
    gip_bm_params params;
    int max_speckle_size;
    double max_speckle_diff;
};

int gip_handle_view(gip_handle h, GipHandleView* out);

// A second handle on the same device with the same model, parameters and speckle settings that SHARES the source's device-resident
// rectification tables (the source must outlive the clone).  For pipelines that keep several batches in flight: gip_batch_run's scratch
// buffers belong to the handle, so every batch in flight needs a handle of its own (gip_streamer.cpp).
int gip_handle_clone(gip_handle src, gip_handle* out);
