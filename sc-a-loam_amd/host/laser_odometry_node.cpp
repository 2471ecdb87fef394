// alaserOdometry with stage B on the GPU (source only, see README.md).  Topic surface of laserOdometry.cpp:195-213: subs the five
// clouds of scanRegistration; pubs /laser_odom_to_init (frame /camera_init, child /laser_odom) and /laser_odom_path every scan,
// /laser_cloud_corner_last, /laser_cloud_surf_last, /velodyne_cloud_3 (frame /camera) every mapping_skip_frame scans (:570-591).
#include <mutex>
#include <queue>
#include <nav_msgs/Odometry.h>
#include <nav_msgs/Path.h>
#include <geometry_msgs/PoseStamped.h>
#include "scal_common.hpp"

static std::mutex mBuf;
static std::queue<sensor_msgs::PointCloud2ConstPtr> sharpBuf, lessSharpBuf, flatBuf, lessFlatBuf, fullBuf;
#define HANDLER(name, buf) static void name(const sensor_msgs::PointCloud2ConstPtr& m) { std::lock_guard<std::mutex> lk(mBuf); buf.push(m); }
HANDLER(sharpHandler, sharpBuf) HANDLER(lessSharpHandler, lessSharpBuf) HANDLER(flatHandler, flatBuf) HANDLER(lessFlatHandler, lessFlatBuf)
HANDLER(fullHandler, fullBuf)

int main(int argc, char** argv) {
    ros::init(argc, argv, "laserOdometry");
    ros::NodeHandle nh;
    int skipFrameNum;
    nh.param<int>("mapping_skip_frame", skipFrameNum, 2);  // :191
    scal_odom_config oc{400000, 0};
    scal_odom_t* od = nullptr;
    SCAL_CHECK(scal_odom_create(&oc, &od));
    ros::Subscriber s1 = nh.subscribe<sensor_msgs::PointCloud2>("/laser_cloud_sharp", 100, sharpHandler);
    ros::Subscriber s2 = nh.subscribe<sensor_msgs::PointCloud2>("/laser_cloud_less_sharp", 100, lessSharpHandler);
    ros::Subscriber s3 = nh.subscribe<sensor_msgs::PointCloud2>("/laser_cloud_flat", 100, flatHandler);
    ros::Subscriber s4 = nh.subscribe<sensor_msgs::PointCloud2>("/laser_cloud_less_flat", 100, lessFlatHandler);
    ros::Subscriber s5 = nh.subscribe<sensor_msgs::PointCloud2>("/velodyne_cloud_2", 100, fullHandler);
    ros::Publisher pubCornerLast = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_corner_last", 100);
    ros::Publisher pubSurfLast = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_surf_last", 100);
    ros::Publisher pubFull = nh.advertise<sensor_msgs::PointCloud2>("/velodyne_cloud_3", 100);
    ros::Publisher pubOdom = nh.advertise<nav_msgs::Odometry>("/laser_odom_to_init", 100);
    ros::Publisher pubPath = nh.advertise<nav_msgs::Path>("/laser_odom_path", 100);
    nav_msgs::Path path;
    int frameCount = 0;
    ros::Rate rate(100);  // :218
    while (ros::ok()) {
        ros::spinOnce();
        sensor_msgs::PointCloud2ConstPtr ms, mls, mf, mlf, mfull;
        {
            std::lock_guard<std::mutex> lk(mBuf);
            if (!sharpBuf.empty() && !lessSharpBuf.empty() && !flatBuf.empty() && !lessFlatBuf.empty() && !fullBuf.empty()) {  // :224
                ms = sharpBuf.front(), mls = lessSharpBuf.front(), mf = flatBuf.front(), mlf = lessFlatBuf.front(), mfull = fullBuf.front();
                sharpBuf.pop(), lessSharpBuf.pop(), flatBuf.pop(), lessFlatBuf.pop(), fullBuf.pop();
            }
        }
        if (ms) {
            const ros::Time st = mfull->header.stamp;
            if (ms->header.stamp != st || mls->header.stamp != st || mf->header.stamp != st || mlf->header.stamp != st) {  // :234-241
                ROS_ERROR("unsync messeage!");
                ROS_BREAK();
            }
            const auto s = scal_ros::from_msg(*ms), ls = scal_ros::from_msg(*mls), f = scal_ros::from_msg(*mf), lf = scal_ros::from_msg(*mlf);
            double q_lc[4], t_lc[3], q[4], t[3];  // (x, y, z, w) like para_q (:97-101); the first call only initialises (:267-271)
            SCAL_CHECK(scal_odom_step(od, s.data(), s.size() / 4, ls.data(), ls.size() / 4, f.data(), f.size() / 4, lf.data(), lf.size() / 4, q_lc, t_lc, q, t, nullptr));
            nav_msgs::Odometry o;  // :511-530
            o.header.frame_id = "/camera_init", o.child_frame_id = "/laser_odom", o.header.stamp = st;
            o.pose.pose.orientation.x = q[0], o.pose.pose.orientation.y = q[1], o.pose.pose.orientation.z = q[2], o.pose.pose.orientation.w = q[3];
            o.pose.pose.position.x = t[0], o.pose.pose.position.y = t[1], o.pose.pose.position.z = t[2];
            pubOdom.publish(o);
            geometry_msgs::PoseStamped ps;
            ps.header = o.header, ps.pose = o.pose.pose;
            path.header.stamp = st, path.header.frame_id = "/camera_init";
            path.poses.push_back(ps);
            pubPath.publish(path);
            if (frameCount % skipFrameNum == 0) {  // :570-591: this scan's lessSharp / lessFlat clouds are the "last" clouds
                frameCount = 0;
                sensor_msgs::PointCloud2 c = *mls, sf = *mlf, full = *mfull;
                c.header.frame_id = sf.header.frame_id = full.header.frame_id = "/camera";
                pubCornerLast.publish(c), pubSurfLast.publish(sf), pubFull.publish(full);
            }
            frameCount++;
        }
        rate.sleep();
    }
    scal_odom_destroy(od);
    return 0;
}
