// ConnectedTopics.hpp -- which of the 11 output topics currently have subscribers; the bit order is the contract
// between the ROS layer and gip_process_pair (reference include/gpuimageproc/ConnectedTopics.h:5-28).
#pragma once
#include <cstdint>

#include "gpuimageproc_hip.h"

namespace gpuimageproc
{

struct ConnectedTopics {
    uint32_t val = 0;

    bool get(uint32_t bit) const { return (val & bit) != 0; }
    void set(uint32_t bit, bool on) { val = on ? (val | bit) : (val & ~bit); }

#define GIP_TOPIC_ACCESSOR(Name, BIT)              \
    bool Name() const { return get(BIT); }         \
    void Name(bool on) { set(BIT, on); }
    GIP_TOPIC_ACCESSOR(DebayerMonoLeft, GIP_TOPIC_DEBAYER_MONO_LEFT)
    GIP_TOPIC_ACCESSOR(DebayerMonoRight, GIP_TOPIC_DEBAYER_MONO_RIGHT)
    GIP_TOPIC_ACCESSOR(DebayerColorLeft, GIP_TOPIC_DEBAYER_COLOR_LEFT)
    GIP_TOPIC_ACCESSOR(DebayerColorRight, GIP_TOPIC_DEBAYER_COLOR_RIGHT)
    GIP_TOPIC_ACCESSOR(RectifyMonoLeft, GIP_TOPIC_RECTIFY_MONO_LEFT)
    GIP_TOPIC_ACCESSOR(RectifyMonoRight, GIP_TOPIC_RECTIFY_MONO_RIGHT)
    GIP_TOPIC_ACCESSOR(RectifyColorLeft, GIP_TOPIC_RECTIFY_COLOR_LEFT)
    GIP_TOPIC_ACCESSOR(RectifyColorRight, GIP_TOPIC_RECTIFY_COLOR_RIGHT)
    GIP_TOPIC_ACCESSOR(Disparity, GIP_TOPIC_DISPARITY)
    GIP_TOPIC_ACCESSOR(DisparityVis, GIP_TOPIC_DISPARITY_VIS)
    GIP_TOPIC_ACCESSOR(Pointcloud, GIP_TOPIC_POINTCLOUD)
#undef GIP_TOPIC_ACCESSOR

    void operator|=(const ConnectedTopics &o) { val |= o.val; }
    // index of the highest connected stage, 0 when nothing is connected
    int level() const { return val == 0 ? 0 : 32 - __builtin_clz(val); }
};

} // namespace gpuimageproc
