// ascanRegistration with stage A on the GPU (source only, see README.md).  Topic surface of scanRegistration.cpp:492-504:
// sub /velodyne_points; pub /velodyne_cloud_2, /laser_cloud_sharp, /laser_cloud_less_sharp, /laser_cloud_flat, /laser_cloud_less_flat
// (frame /camera_init, stamp of the input); /laser_remove_points is advertised and never published, as in the reference.
#include "scal_common.hpp"

static scal_features_t* g_feat = nullptr;
static ros::Publisher pubFull, pubSharp, pubLessSharp, pubFlat, pubLessFlat, pubRemove;
static std::vector<float> cloud, less_flat;
static std::vector<int> sharp, less_sharp, flat;

static void gather(const std::vector<int>& idx, int n, std::vector<float>& out) {
    out.resize(4 * static_cast<size_t>(n));
    for (int k = 0; k < n; ++k) std::memcpy(&out[4 * k], &cloud[4 * idx[k]], 16);
}

static void laserCloudHandler(const sensor_msgs::PointCloud2ConstPtr& msg) {
    // x, y, z are float32 at offsets 0, 4, 8 of every point (velodyne / ouster drivers): the message buffer goes in as it is
    scal_features_out o{};
    o.cloud = cloud.data(), o.sharp = sharp.data(), o.less_sharp = less_sharp.data(), o.flat = flat.data(), o.less_flat = less_flat.data();
    const int rc = scal_features_run(g_feat, msg->data.data(), static_cast<int>(msg->width * msg->height), static_cast<int>(msg->point_step), &o);
    if (rc == SCAL_E_EMPTY) return;  // nothing survives the filters: the reference would index points[0] here
    SCAL_CHECK(rc);
    const ros::Time st = msg->header.stamp;
    std::vector<float> tmp;
    pubFull.publish(scal_ros::to_msg(cloud.data(), o.n_kept, st, "/camera_init"));  // :426-430
    gather(sharp, o.n_sharp, tmp);
    pubSharp.publish(scal_ros::to_msg(tmp.data(), o.n_sharp, st, "/camera_init"));
    gather(less_sharp, o.n_less_sharp, tmp);
    pubLessSharp.publish(scal_ros::to_msg(tmp.data(), o.n_less_sharp, st, "/camera_init"));
    gather(flat, o.n_flat, tmp);
    pubFlat.publish(scal_ros::to_msg(tmp.data(), o.n_flat, st, "/camera_init"));
    pubLessFlat.publish(scal_ros::to_msg(less_flat.data(), o.n_less_flat, st, "/camera_init"));
}

int main(int argc, char** argv) {
    ros::init(argc, argv, "scanRegistration");
    ros::NodeHandle nh;
    int n_scans;
    std::string lidar_type;
    double minimum_range;
    nh.param<int>("scan_line", n_scans, 16);                // :480
    nh.param<std::string>("lidar_type", lidar_type, "KITTI");  // :481 - matches no branch, as in the reference: create fails below
    nh.param<double>("minimum_range", minimum_range, 0.1);  // :482
    scal_features_config fc{};
    fc.lidar_type = lidar_type == "VLP16" ? SCAL_VLP16 : lidar_type == "HDL32" ? SCAL_HDL32 : lidar_type == "HDL64" ? SCAL_HDL64 : lidar_type == "OS1-64" ? SCAL_OS1_64 : -1;
    fc.n_scans = n_scans, fc.minimum_range = minimum_range, fc.max_points = 400000, fc.check_finite = 1;
    SCAL_CHECK(scal_features_create(&fc, &g_feat));         // unknown lidar_type -> error (ROS_BREAK at :214-218), scan_line not 16/32/64 (:486-490)
    cloud.resize(4 * 400000), less_flat.resize(4 * 400000);
    sharp.resize(2 * 6 * n_scans), less_sharp.resize(20 * 6 * n_scans), flat.resize(4 * 6 * n_scans);
    ros::Subscriber sub = nh.subscribe<sensor_msgs::PointCloud2>("/velodyne_points", 100, laserCloudHandler);
    pubFull = nh.advertise<sensor_msgs::PointCloud2>("/velodyne_cloud_2", 100);
    pubSharp = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_sharp", 100);
    pubLessSharp = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_less_sharp", 100);
    pubFlat = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_flat", 100);
    pubLessFlat = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_less_flat", 100);
    pubRemove = nh.advertise<sensor_msgs::PointCloud2>("/laser_remove_points", 100);
    ros::spin();
    scal_features_destroy(g_feat);
    return 0;
}
