// ros/StereoProcessorNode.cpp -- stand-alone executable of the stereo processor: loads the nodelet `gpuimageproc/Stereoproc` into a
// private nodelet loader, so node and nodelet run exactly the same code (the reference builds a second entry point around the same
// StereoProcessor class: src/StereoProcessorNode.cpp:4-34).  Built only with catkin (CMakeLists.txt).
#include <nodelet/loader.h>
#include <ros/ros.h>

int main(int argc, char **argv)
{
    ros::init(argc, argv, "gpuimageproc_node");
    nodelet::Loader loader(false);
    nodelet::M_string remappings(ros::names::getRemappings());
    nodelet::V_string args;
    for (int i = 1; i < argc; i++) args.push_back(argv[i]);
    if (!loader.load(ros::this_node::getName(), "gpuimageproc/Stereoproc", remappings, args)) {
        ROS_FATAL("cannot load nodelet gpuimageproc/Stereoproc (is plugins/nodelet_plugins.xml exported?)");
        return 1;
    }
    ros::spin();
    return 0;
}
