// Shared helpers of the node shells (source only, see README.md): packed xyzi <-> ROS / PCL containers, error handling.
#pragma once
#include <ros/ros.h>
#include <sensor_msgs/PointCloud2.h>
#include <sensor_msgs/point_cloud2_iterator.h>
#include <vector>
#include "scaloam_hip.h"

#define SCAL_CHECK(expr)                                              \
    do {                                                              \
        const int rc_ = (expr);                                       \
        if (rc_ != SCAL_OK) {                                         \
            ROS_ERROR("%s failed: %s", #expr, scal_last_error());     \
            ROS_BREAK();                                              \
        }                                                             \
    } while (0)

namespace scal_ros {

// x, y, z, intensity as four consecutive float32 (the layout pcl::toROSMsg gives pcl::PointXYZI without its padding)
inline sensor_msgs::PointCloud2 to_msg(const float* xyzi, int n, const ros::Time& stamp, const std::string& frame) {
    sensor_msgs::PointCloud2 m;
    m.header.stamp = stamp;
    m.header.frame_id = frame;
    m.height = 1;
    m.width = n;
    sensor_msgs::PointCloud2Modifier mod(m);
    mod.setPointCloud2Fields(4, "x", 1, sensor_msgs::PointField::FLOAT32, "y", 1, sensor_msgs::PointField::FLOAT32, "z", 1,
                             sensor_msgs::PointField::FLOAT32, "intensity", 1, sensor_msgs::PointField::FLOAT32);
    mod.resize(n);
    std::memcpy(m.data.data(), xyzi, sizeof(float) * 4 * n);
    m.is_dense = true;
    return m;
}

// any PointCloud2 with float32 x, y, z, intensity fields -> packed xyzi
inline std::vector<float> from_msg(const sensor_msgs::PointCloud2& m) {
    std::vector<float> v(4 * static_cast<size_t>(m.width) * m.height);
    sensor_msgs::PointCloud2ConstIterator<float> x(m, "x"), y(m, "y"), z(m, "z"), i(m, "intensity");
    for (size_t k = 0; k < v.size() / 4; ++k, ++x, ++y, ++z, ++i) v[4 * k] = *x, v[4 * k + 1] = *y, v[4 * k + 2] = *z, v[4 * k + 3] = *i;
    return v;
}

}  // namespace scal_ros
