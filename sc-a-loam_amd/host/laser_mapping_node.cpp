// alaserMapping with stage C on the GPU (source only, see README.md).  Topic surface of laserMapping.cpp:921-940: subs
// /laser_cloud_corner_last, /laser_cloud_surf_last, /laser_odom_to_init, /velodyne_cloud_3; pubs /aft_mapped_to_init (child
// /aft_mapped), /aft_mapped_to_init_high_frec, /aft_mapped_path, /velodyne_cloud_registered(_local), /laser_cloud_surround (every 5
// scans), /laser_cloud_map (every 20), tf /camera_init -> /aft_mapped.
#include <mutex>
#include <queue>
#include <thread>
#include <nav_msgs/Odometry.h>
#include <nav_msgs/Path.h>
#include <geometry_msgs/PoseStamped.h>
#include <tf/transform_broadcaster.h>
#include "scal_common.hpp"

static std::mutex mBuf, mPose;
static std::queue<sensor_msgs::PointCloud2ConstPtr> cornerBuf, surfBuf, fullBuf;
static std::queue<nav_msgs::Odometry::ConstPtr> odomBuf;
static scal_map_t* g_map = nullptr;
static double q_wmap_wodom[4] = {0, 0, 0, 1}, t_wmap_wodom[3] = {0, 0, 0};
static ros::Publisher pubSurround, pubMap, pubReg, pubRegLocal, pubAft, pubAftHigh, pubPath;
static nav_msgs::Path path;

static void qmul(const double* a, const double* b, double* o) {
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1], o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0], o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
static void qrot(const double* q, const double* v, double* o) {
    double ux = q[1] * v[2] - q[2] * v[1], uy = q[2] * v[0] - q[0] * v[2], uz = q[0] * v[1] - q[1] * v[0];
    ux += ux, uy += uy, uz += uz;
    o[0] = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy), o[1] = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz), o[2] = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
}

// :198-230: every odometry message is re-published in the map frame with the latest map <- odometry correction
static void laserOdometryHandler(const nav_msgs::Odometry::ConstPtr& m) {
    {
        std::lock_guard<std::mutex> lk(mBuf);
        odomBuf.push(m);
    }
    const double qo[4] = {m->pose.pose.orientation.x, m->pose.pose.orientation.y, m->pose.pose.orientation.z, m->pose.pose.orientation.w};
    const double to[3] = {m->pose.pose.position.x, m->pose.pose.position.y, m->pose.pose.position.z};
    double q[4], r[3], qm[4], tm[3];
    {
        std::lock_guard<std::mutex> lk(mPose);  // the reference reads these unlocked while process() writes them
        std::memcpy(qm, q_wmap_wodom, sizeof qm), std::memcpy(tm, t_wmap_wodom, sizeof tm);
    }
    qmul(qm, qo, q), qrot(qm, to, r);
    nav_msgs::Odometry o;
    o.header.frame_id = "/camera_init", o.child_frame_id = "/aft_mapped", o.header.stamp = m->header.stamp;
    o.pose.pose.orientation.x = q[0], o.pose.pose.orientation.y = q[1], o.pose.pose.orientation.z = q[2], o.pose.pose.orientation.w = q[3];
    o.pose.pose.position.x = r[0] + tm[0], o.pose.pose.position.y = r[1] + tm[1], o.pose.pose.position.z = r[2] + tm[2];
    pubAftHigh.publish(o);
}
#define HANDLER(name, buf) static void name(const sensor_msgs::PointCloud2ConstPtr& m) { std::lock_guard<std::mutex> lk(mBuf); buf.push(m); }
HANDLER(cornerHandler, cornerBuf) HANDLER(surfHandler, surfBuf) HANDLER(fullHandler, fullBuf)

static void process() {
    int frameCount = 0;
    std::vector<float> reg, exported(4 * static_cast<size_t>(2 * 4000000));  // both classes of the whole map (scal_map_config::max_map_points each)
    while (ros::ok()) {
        sensor_msgs::PointCloud2ConstPtr mc, ms, mf;
        nav_msgs::Odometry::ConstPtr mo;
        {
            std::lock_guard<std::mutex> lk(mBuf);
            if (!cornerBuf.empty() && !surfBuf.empty() && !fullBuf.empty() && !odomBuf.empty()) {  // :236-277: align the four queues on the corner stamp
                const ros::Time st = cornerBuf.front()->header.stamp;
                while (!odomBuf.empty() && odomBuf.front()->header.stamp < st) odomBuf.pop();
                while (!surfBuf.empty() && surfBuf.front()->header.stamp < st) surfBuf.pop();
                while (!fullBuf.empty() && fullBuf.front()->header.stamp < st) fullBuf.pop();
                if (!odomBuf.empty() && !surfBuf.empty() && !fullBuf.empty() && odomBuf.front()->header.stamp == st && surfBuf.front()->header.stamp == st &&
                    fullBuf.front()->header.stamp == st) {
                    mc = cornerBuf.front(), ms = surfBuf.front(), mf = fullBuf.front(), mo = odomBuf.front();
                    cornerBuf.pop(), surfBuf.pop(), fullBuf.pop(), odomBuf.pop();
                    while (!cornerBuf.empty()) cornerBuf.pop();  // :300-304: stay real-time, drop what queued up
                }
            }
        }
        if (mc) {
            const auto c = scal_ros::from_msg(*mc), s = scal_ros::from_msg(*ms), f = scal_ros::from_msg(*mf);
            const double qo[4] = {mo->pose.pose.orientation.x, mo->pose.pose.orientation.y, mo->pose.pose.orientation.z, mo->pose.pose.orientation.w};
            const double to[3] = {mo->pose.pose.position.x, mo->pose.pose.position.y, mo->pose.pose.position.z};
            double q[4], t[3];
            scal_map_stats st;
            reg.resize(f.size());
            SCAL_CHECK(scal_map_step(g_map, c.data(), c.size() / 4, s.data(), s.size() / 4, f.data(), f.size() / 4, qo, to, q, t, reg.data(), &st));
            if (!st.solved) ROS_WARN("time Map corner and surf num are not enough");  // :731-734
            {
                std::lock_guard<std::mutex> lk(mPose);
                SCAL_CHECK(scal_map_get_wmap_wodom(g_map, q_wmap_wodom, t_wmap_wodom));  // transformUpdate happened on the device (:735)
            }
            const ros::Time stamp = mo->header.stamp;
            // `exported` holds 2 x max_map_points records: both classes of the whole cube grid fit
            auto export_both = [&](int (*fn)(scal_map_t*, int, float*, int)) -> int {
                const int cap = static_cast<int>(exported.size() / 4);
                const int nc = fn(g_map, 0, exported.data(), cap);
                if (nc < 0) { ROS_ERROR("map export: %s", scal_last_error()); return -1; }   // <0 = error: never used as an offset
                const int ns = fn(g_map, 1, exported.data() + 4 * static_cast<size_t>(nc), cap - nc);
                if (ns < 0) { ROS_ERROR("map export: %s", scal_last_error()); return -1; }
                return nc + ns;
            };
            if (frameCount % 5 == 0) {  // :807-822: the 5x5x3 window (corner + surf)
                const int n = export_both(scal_map_export);
                if (n >= 0) pubSurround.publish(scal_ros::to_msg(exported.data(), n, stamp, "/camera_init"));
            }
            if (frameCount % 20 == 0) {  // :824-837: all 4851 cubes, corner + surf
                const int n = export_both(scal_map_export_all);
                if (n >= 0) pubMap.publish(scal_ros::to_msg(exported.data(), n, stamp, "/camera_init"));
            }
            pubRegLocal.publish(*mf);                                                                     // :839-843, sensor frame (PGO keyframes)
            pubReg.publish(scal_ros::to_msg(reg.data(), static_cast<int>(f.size() / 4), stamp, "/camera_init"));  // :845-855
            nav_msgs::Odometry o;  // :861-886
            o.header.frame_id = "/camera_init", o.child_frame_id = "/aft_mapped", o.header.stamp = stamp;
            o.pose.pose.orientation.x = q[0], o.pose.pose.orientation.y = q[1], o.pose.pose.orientation.z = q[2], o.pose.pose.orientation.w = q[3];
            o.pose.pose.position.x = t[0], o.pose.pose.position.y = t[1], o.pose.pose.position.z = t[2];
            pubAft.publish(o);
            geometry_msgs::PoseStamped ps;
            ps.header = o.header, ps.pose = o.pose.pose;
            path.header = o.header;
            path.poses.push_back(ps);
            pubPath.publish(path);
            static tf::TransformBroadcaster br;  // :888-899
            tf::Transform tr;
            tr.setOrigin(tf::Vector3(t[0], t[1], t[2]));
            tr.setRotation(tf::Quaternion(q[0], q[1], q[2], q[3]));
            br.sendTransform(tf::StampedTransform(tr, stamp, "/camera_init", "/aft_mapped"));
            frameCount++;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(2));  // :903-904
    }
}

int main(int argc, char** argv) {
    ros::init(argc, argv, "laserMapping");
    ros::NodeHandle nh;
    float lineRes, planeRes;
    nh.param<float>("mapping_line_resolution", lineRes, 0.4);    // :915
    nh.param<float>("mapping_plane_resolution", planeRes, 0.8);  // :916
    scal_map_config mc{lineRes, planeRes, 400000, 4000000, 0};
    SCAL_CHECK(scal_map_create(&mc, &g_map));
    ros::Subscriber s1 = nh.subscribe<sensor_msgs::PointCloud2>("/laser_cloud_corner_last", 100, cornerHandler);
    ros::Subscriber s2 = nh.subscribe<sensor_msgs::PointCloud2>("/laser_cloud_surf_last", 100, surfHandler);
    ros::Subscriber s3 = nh.subscribe<nav_msgs::Odometry>("/laser_odom_to_init", 100, laserOdometryHandler);
    ros::Subscriber s4 = nh.subscribe<sensor_msgs::PointCloud2>("/velodyne_cloud_3", 100, fullHandler);
    pubSurround = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_surround", 100);
    pubMap = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_map", 100);
    pubReg = nh.advertise<sensor_msgs::PointCloud2>("/velodyne_cloud_registered", 100);
    pubRegLocal = nh.advertise<sensor_msgs::PointCloud2>("/velodyne_cloud_registered_local", 100);
    pubAft = nh.advertise<nav_msgs::Odometry>("/aft_mapped_to_init", 100);
    pubAftHigh = nh.advertise<nav_msgs::Odometry>("/aft_mapped_to_init_high_frec", 100);
    pubPath = nh.advertise<nav_msgs::Path>("/aft_mapped_path", 100);
    std::thread worker(process);  // :948
    ros::spin();
    worker.join();
    scal_map_destroy(g_map);
    return 0;
}
