// auv_pf_node.cpp -- roscpp form of the `auv_pf` node (auv_particle_filter/scripts/auv_pf.py:25-119,264-285): the same
// node name, private parameters, topics, frames and message types; the per-particle numerics run in libmcl_hip.so
// through the C ABI (include/mcl.h).  All logic lives in pf_core.hpp (ROS-free, compiled and tested without ROS);
// this file is only the transport: parameters, the map <- odom lookup, subscribers, publishers, tf, the 10 Hz timer.
//
// CMakeLists.txt builds it when catkin finds roscpp & co.  In the build container (no ROS) it is compiled unchanged
// against the stand-in headers of tests/ros_stubs_cpp and run in one process (tests/test_roscpp_node_stub.py).
#include <mutex>

#include <geometry_msgs/PointStamped.h>
#include <geometry_msgs/PoseArray.h>
#include <geometry_msgs/TransformStamped.h>
#include <nav_msgs/Odometry.h>
#include <ros/ros.h>
#include <sensor_msgs/LaserScan.h>
#include <sensor_msgs/PointCloud2.h>
#include <sensor_msgs/point_cloud2_iterator.h>
#include <std_msgs/Bool.h>
#include <tf2_geometry_msgs/tf2_geometry_msgs.h>
#include <tf2_ros/transform_broadcaster.h>
#include <tf2_ros/transform_listener.h>

#include "auv_particle_filter_hip/pf_core.hpp"

namespace {

class AuvPfNode {
 public:
  explicit AuvPfNode(ros::NodeHandle& pnh) : pnh_(pnh), tf_listener_(tf_buffer_) {}

  bool init() {
    auv_pf_hip::Params p;
    // auv_pf.py:27-56, 64, 71, 101, 106, 110: same names, same defaults
    pnh_.param("particle_count", p.particle_count, p.particle_count);
    pnh_.param("map_frame", p.map_frame, p.map_frame);
    pnh_.param("base_frame", p.base_frame, p.base_frame);
    pnh_.param("utm_frame", p.utm_frame, p.utm_frame);
    pnh_.param("odom_frame", p.odom_frame, p.odom_frame);
    pnh_.param("measurement_std", p.measurement_std, p.measurement_std);
    pnh_.param("motion_covariance", p.motion_covariance, p.motion_covariance);
    pnh_.param("init_covariance", p.init_covariance, p.init_covariance);
    pnh_.param("resampling_noise_covariance", p.resampling_noise_covariance, p.resampling_noise_covariance);
    pnh_.param("particle_poses_topic", p.particle_poses_topic, p.particle_poses_topic);
    pnh_.param("odom_corrected_topic", p.odom_corrected_topic, p.odom_corrected_topic);
    pnh_.param("aux_dive", p.aux_dive, p.aux_dive);
    pnh_.param("gps_odom_topic", p.gps_odom_topic, p.gps_odom_topic);
    pnh_.param("odom_topic", p.odom_topic, p.odom_topic);
    // beyond the reference (all optional)
    pnh_.param("resample_scheme", p.resample_scheme, p.resample_scheme);
    pnh_.param("seed", p.seed, p.seed);
    pnh_.param("device", p.device, p.device);
    pnh_.param("mbes_topic", p.mbes_topic, p.mbes_topic);
    pnh_.param("mbes_pointcloud_topic", p.mbes_pointcloud_topic, p.mbes_pointcloud_topic);
    pnh_.param("mbes_points_frame", p.mbes_points_frame, p.mbes_points_frame);
    pnh_.param("mbes_std", p.mbes_std, p.mbes_std);
    pnh_.param("mbes_range_max", p.mbes_range_max, p.mbes_range_max);
    pnh_.param("mbes_sensor_offset", p.mbes_sensor_offset, p.mbes_sensor_offset);
    pnh_.param("map_grid_file", p.map_grid_file, p.map_grid_file);
    pnh_.param("map_mesh_file", p.map_mesh_file, p.map_mesh_file);
    pnh_.param("landmark_map_file", p.landmark_map_file, p.landmark_map_file);
    pnh_.param("rocks_depth", p.rocks_depth, p.rocks_depth);
    pnh_.param("lm_detect_topic", p.lm_detect_topic, p.lm_detect_topic);
    pnh_.param("landmark_std", p.landmark_std, p.landmark_std);
    pnh_.param("landmark_k", p.landmark_k, p.landmark_k);
    pnh_.param("landmark_gate", p.landmark_gate, p.landmark_gate);
    pnh_.param("landmark_sync_tol", p.landmark_sync_tol, p.landmark_sync_tol);
    pnh_.param("landmark_max_age", p.landmark_max_age, p.landmark_max_age);
    {
      std::string late = "update";
      pnh_.param("landmark_late", late, late);
      p.landmark_late_drop = late == "drop";
    }
    pnh_.param("max_published_poses", max_poses_, 5000);
    if (max_poses_ < 1) max_poses_ = 1;   // (the stride of the thinned PoseArray divides by it)

    // map <- odom once at start, 60 s timeout; exit quietly on failure (auv_pf.py:76-87)
    double m2o[16];
    try {
      ROS_INFO("Waiting for transforms");
      const geometry_msgs::TransformStamped t =
          tf_buffer_.lookupTransform(p.map_frame, p.odom_frame, ros::Time(0), ros::Duration(60.0));
      const double tr[3] = {t.transform.translation.x, t.transform.translation.y, t.transform.translation.z};
      const double q[4] = {t.transform.rotation.x, t.transform.rotation.y, t.transform.rotation.z, t.transform.rotation.w};
      mcl_matrix_from_tf(tr, q, m2o);
      ROS_INFO("PF: got transform %s to %s", p.map_frame.c_str(), p.odom_frame.c_str());
    } catch (const tf2::TransformException&) {
      ROS_ERROR("PF: Could not lookup transform %s to %s", p.map_frame.c_str(), p.odom_frame.c_str());
      return false;
    }
    if (!core_.init(p, m2o)) {
      ROS_ERROR("PF: %s", core_.error().c_str());
      return false;
    }
    if (!core_.has_map()) ROS_WARN("PF: no ~map_grid_file / ~map_mesh_file: MBES pings will be ignored (GPS updates only)");
    core_.start_timing(ros::Time::now().toSec());

    ros::NodeHandle nh;
    pf_pub_ = nh.advertise<geometry_msgs::PoseArray>(p.particle_poses_topic, 10);
    loc_pub_ = nh.advertise<nav_msgs::Odometry>(p.odom_corrected_topic, 100);
    dive_sub_ = nh.subscribe(p.aux_dive, 100, &AuvPfNode::diveCb, this);
    gps_sub_ = nh.subscribe(p.gps_odom_topic, 100, &AuvPfNode::gpsCb, this);
    scan_sub_ = nh.subscribe(p.mbes_topic, 10, &AuvPfNode::scanCb, this);
    if (!p.mbes_pointcloud_topic.empty()) cloud_sub_ = nh.subscribe(p.mbes_pointcloud_topic, 10, &AuvPfNode::cloudCb, this);
    if (core_.has_landmarks()) lm_sub_ = nh.subscribe(p.lm_detect_topic, 10, &AuvPfNode::lmCb, this);   // config 5
    odom_sub_ = nh.subscribe(p.odom_topic, 100, &AuvPfNode::odomCb, this);
    timer_ = nh.createTimer(ros::Duration(0.1), &AuvPfNode::locLoop, this);   // auv_pf.py:114
    ROS_INFO("Particle filter class successfully created");
    return true;
  }

 private:
  void diveCb(const std_msgs::Bool::ConstPtr& m) {
    std::lock_guard<std::mutex> lk(mu_);
    core_.dive(m->data);
  }
  void odomCb(const nav_msgs::Odometry::ConstPtr& m) {
    std::lock_guard<std::mutex> lk(mu_);
    const double v[3] = {m->twist.twist.linear.x, m->twist.twist.linear.y, m->twist.twist.linear.z};
    const double q[4] = {m->pose.pose.orientation.x, m->pose.pose.orientation.y, m->pose.pose.orientation.z,
                         m->pose.pose.orientation.w};
    if (!core_.odom(m->header.stamp.toSec(), v, m->twist.twist.angular.z, q, m->pose.pose.position.z))
      ROS_WARN_THROTTLE(1.0, "PF predict: %s", core_.error().c_str());
  }
  void gpsCb(const nav_msgs::Odometry::ConstPtr& m) {
    // the fix utm -> map, once (the reference repeats the same transformPoint per particle, auv_pf.py:143-151)
    geometry_msgs::PointStamped in, out;
    in.header.frame_id = core_.params().utm_frame;
    in.header.stamp = ros::Time(0);
    in.point.x = m->pose.pose.position.x;
    in.point.y = m->pose.pose.position.y;
    in.point.z = 0.0;
    try {
      tf_buffer_.transform(in, out, core_.params().map_frame);
    } catch (const tf2::TransformException&) {
      ROS_WARN("PF: Transform to utm-->map not available yet");   // auv_pf.py:158-160: skip the update
      return;
    }
    std::lock_guard<std::mutex> lk(mu_);
    if (!core_.gps(out.point.x, out.point.y)) ROS_WARN_THROTTLE(1.0, "PF update: %s", core_.error().c_str());
  }
  void scanCb(const sensor_msgs::LaserScan::ConstPtr& s) {
    std::lock_guard<std::mutex> lk(mu_);
    if (!core_.ping_scan(s->ranges.data(), (int)s->ranges.size(), s->angle_min, s->angle_increment, s->range_max,
                         s->header.stamp.toSec()))
      ROS_WARN_THROTTLE(1.0, "PF MBES update: %s", core_.error().c_str());
  }
  // detections of the MBES receptors: PoseArray in base_frame, positions only (toy_mbes_receptor.cpp:75-105)
  void lmCb(const geometry_msgs::PoseArray::ConstPtr& m) {
    std::vector<double> xyz;
    xyz.reserve(m->poses.size() * 3);
    for (const auto& ps : m->poses) {
      xyz.push_back(ps.position.x);
      xyz.push_back(ps.position.y);
      xyz.push_back(ps.position.z);
    }
    std::lock_guard<std::mutex> lk(mu_);
    if (!core_.detections(m->header.stamp.toSec(), xyz.data(), (int)(xyz.size() / 3)))
      ROS_WARN_THROTTLE(1.0, "PF landmark update: %s", core_.error().c_str());
  }
  void cloudCb(const sensor_msgs::PointCloud2::ConstPtr& c) {
    // x / y / z as FLOAT32 (7) or FLOAT64 (8), whichever the cloud declares; anything else: the cloud is dropped
    int dtype = 0;
    for (const auto& fld : c->fields)
      if (fld.name == "x") dtype = fld.datatype;
    for (const auto& fld : c->fields)
      if ((fld.name == "y" || fld.name == "z") && fld.datatype != dtype) dtype = 0;
    std::vector<float> xyz;
    xyz.reserve((size_t)c->width * c->height * 3);
    if (dtype == 7) {
      sensor_msgs::PointCloud2ConstIterator<float> ix(*c, "x"), iy(*c, "y"), iz(*c, "z");
      for (; ix != ix.end(); ++ix, ++iy, ++iz) {
        xyz.push_back(*ix);
        xyz.push_back(*iy);
        xyz.push_back(*iz);
      }
    } else if (dtype == 8) {
      sensor_msgs::PointCloud2ConstIterator<double> ix(*c, "x"), iy(*c, "y"), iz(*c, "z");
      for (; ix != ix.end(); ++ix, ++iy, ++iz) {
        xyz.push_back((float)*ix);
        xyz.push_back((float)*iy);
        xyz.push_back((float)*iz);
      }
    } else {
      ROS_WARN_THROTTLE(1.0, "PF: point cloud without FLOAT32 / FLOAT64 x, y, z fields: dropped");
      return;
    }
    std::lock_guard<std::mutex> lk(mu_);
    if (!core_.ping_points(xyz.data(), (int)(xyz.size() / 3), core_.params().mbes_points_frame == "sensor",
                           c->header.stamp.toSec()))
      ROS_WARN_THROTTLE(1.0, "PF MBES update: %s", core_.error().c_str());
  }
  void locLoop(const ros::TimerEvent&) {
    std::lock_guard<std::mutex> lk(mu_);
    const auv_pf_hip::Params& p = core_.params();
    double mean[6], yaw, cov36[36], q[4];
    if (!core_.loc_pose(mean, &yaw, cov36, q) || !core_.poses(pose7_)) {
      ROS_WARN_THROTTLE(1.0, "PF loc_loop: %s", core_.error().c_str());
      return;
    }
    const ros::Time now = ros::Time::now();
    nav_msgs::Odometry od;   // auv_pf.py:68-72,222-254
    od.header.frame_id = p.odom_frame;
    od.header.stamp = now;
    od.child_frame_id = p.base_frame;
    od.pose.pose.position.x = mean[0];
    od.pose.pose.position.y = mean[1];
    od.pose.pose.position.z = mean[2];
    od.pose.pose.orientation.x = q[0];
    od.pose.pose.orientation.y = q[1];
    od.pose.pose.orientation.z = q[2];
    od.pose.pose.orientation.w = q[3];
    for (int k = 0; k < 36; ++k) od.pose.covariance[k] = cov36[k];
    loc_pub_.publish(od);
    geometry_msgs::TransformStamped t;   // odom -> base, z = 0 (auv_pf.py:256-260)
    t.header.stamp = now;
    t.header.frame_id = p.odom_frame;
    t.child_frame_id = p.base_frame;
    t.transform.translation.x = mean[0];
    t.transform.translation.y = mean[1];
    t.transform.translation.z = 0.0;
    t.transform.rotation = od.pose.pose.orientation;
    tf_broadcaster_.sendTransform(t);
    geometry_msgs::PoseArray pa;   // auv_pf.py:264-285 (thinned above max_published_poses)
    pa.header.frame_id = p.odom_frame;
    pa.header.stamp = now;
    const size_t n = pose7_.size() / 7, stride = std::max<size_t>(1, (n + (size_t)max_poses_ - 1) / (size_t)max_poses_);
    for (size_t i = 0; i < n; i += stride) {
      geometry_msgs::Pose ps;
      ps.position.x = pose7_[7 * i];
      ps.position.y = pose7_[7 * i + 1];
      ps.position.z = pose7_[7 * i + 2];
      ps.orientation.x = pose7_[7 * i + 3];
      ps.orientation.y = pose7_[7 * i + 4];
      ps.orientation.z = pose7_[7 * i + 5];
      ps.orientation.w = pose7_[7 * i + 6];
      pa.poses.push_back(ps);
    }
    pf_pub_.publish(pa);
  }

  ros::NodeHandle pnh_;
  tf2_ros::Buffer tf_buffer_;
  tf2_ros::TransformListener tf_listener_;
  tf2_ros::TransformBroadcaster tf_broadcaster_;
  auv_pf_hip::Core core_;
  std::mutex mu_;
  ros::Publisher pf_pub_, loc_pub_;
  ros::Subscriber dive_sub_, gps_sub_, scan_sub_, cloud_sub_, lm_sub_, odom_sub_;
  ros::Timer timer_;
  std::vector<double> pose7_;
  int max_poses_ = 5000;
};

}  // namespace

int main(int argc, char** argv) {
  ros::init(argc, argv, "auv_pf");
  ros::NodeHandle pnh("~");
  AuvPfNode node(pnh);
  if (!node.init()) return 1;
  ros::AsyncSpinner spinner(3);   // the reference runs one thread per subscription plus the timer (rospy)
  spinner.start();
  ros::waitForShutdown();
  return 0;
}
