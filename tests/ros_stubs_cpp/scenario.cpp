// The other half of the stand-in ROS (README.md): ros::init sets the node's private parameters and the static
// transforms, ros::waitForShutdown plays the messages of examples/pf_core_example.cpp through the node's own
// subscribers -- three odometry messages, a LaserScan ping, the same ping as a PointCloud2 in base_frame, /dive false,
// a GPS fix, one timer tick -- and prints what the node published.
//   usage: <node> <map file> <ranges file: one range per line, angles linspace(-1, 1)>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>

#include <geometry_msgs/PoseArray.h>
#include <nav_msgs/Odometry.h>
#include <ros/ros.h>
#include <sensor_msgs/LaserScan.h>
#include <sensor_msgs/PointCloud2.h>
#include <std_msgs/Bool.h>
#include <tf2_ros/buffer.h>

namespace {
std::vector<float> g_ranges;
bool g_landmarks = false;
const double OFF[6] = {0.3, 0.0, -0.1, 0.0, 0.05, 0.0};
}  // namespace

namespace ros {

void init(int& argc, char** argv, const std::string& name) {
  stub::Master& m = stub::master();
  m.node_name = name;
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s <map file> <ranges file>\n", argv[0]);
    std::exit(2);
  }
  std::ifstream f(argv[2]);
  float r;
  while (f >> r) g_ranges.push_back(r);
  if (g_ranges.size() < 2) std::exit(2);
  m.params = {{"particle_count", "4096"},
              {"seed", "11"},
              {"init_covariance", "[0.5, 0.5, 0.0, 0.0, 0.0, 0.01]"},
              {"motion_covariance", "[0.001, 0.001, 0.0, 0.0, 0.0, 0.00001]"},
              {"resampling_noise_covariance", "[0.01, 0.01, 0.0, 0.0, 0.0, 0.0001]"},
              {"measurement_std", "1.0"},
              {"mbes_sensor_offset", "[0.3, 0.0, -0.1, 0.0, 0.05, 0.0]"},
              {"map_grid_file", argv[1]},
              {"odom_topic", "/sam/dr/odom"},
              {"gps_odom_topic", "/sam/dr/gps"},
              {"mbes_topic", "/sam/mbes_scan"},
              {"mbes_pointcloud_topic", "/sam/mbes_cloud"},
              {"base_frame", "sam/base_link"},
              {"odom_corrected_topic", "/sam/dr/odom_corrected"},
              {"particle_poses_topic", "/sam/dr/particle_poses"}};
  if (argc > 3 && std::strcmp(argv[3], "no-tf") == 0) return;   // (the start-up failure path, auv_pf.py:84-87)
  if (argc > 3) {   // BASELINE config 5: a landmark map, detections on the receptors' topic
    m.params["landmark_map_file"] = argv[3];
    m.params["lm_detect_topic"] = "/sam/mbes_detections";
    m.params["landmark_k"] = "2";
    g_landmarks = true;
  }
  geometry_msgs::Transform t;
  t.translation.x = 0.5;
  t.translation.y = -0.5;
  t.rotation.w = 1.0;
  tf2_ros::stub::transforms()[{"map", "sam/odom"}] = t;
  geometry_msgs::Transform id;
  id.rotation.w = 1.0;
  tf2_ros::stub::transforms()[{"map", "utm"}] = id;
  m.now = 100.0;
}

void waitForShutdown() {
  stub::Master& m = stub::master();
  const int B = (int)g_ranges.size();
  const double amin = -1.0, ainc = (double)(float)(2.0 / (B - 1));   // (what the LaserScan below carries)
  sensor_msgs::LaserScan scan;
  scan.angle_min = (float)amin;
  scan.angle_increment = (float)ainc;
  scan.range_max = 80.f;
  scan.ranges = g_ranges;
  // the ping as points in base_frame: sensor-frame hit (0, r sin a, -r cos a) through the sensor offset (pitch only)
  sensor_msgs::PointCloud2 pc;
  pc.header.frame_id = "sam/base_link";
  pc.height = 1;
  pc.width = (uint32_t)B;
  pc.point_step = 12;
  pc.row_step = 12 * (uint32_t)B;
  const char* names[3] = {"x", "y", "z"};
  for (int k = 0; k < 3; ++k) {
    sensor_msgs::PointField f;
    f.name = names[k];
    f.offset = 4u * (uint32_t)k;
    f.datatype = sensor_msgs::PointField::FLOAT32;
    f.count = 1;
    pc.fields.push_back(f);
  }
  pc.data.resize((size_t)12 * B);
  const double cp = std::cos(OFF[4]), sp = std::sin(OFF[4]);
  for (int k = B - 1, slot = 0; k >= 0; --k, ++slot) {   // reversed: the node sorts the beams
    const double a = amin + ainc * k, r = g_ranges[k];
    const double s[3] = {0.0, r * std::sin(a), -r * std::cos(a)};
    const float p[3] = {(float)(cp * s[0] + sp * s[2] + OFF[0]), (float)(s[1] + OFF[1]), (float)(-sp * s[0] + cp * s[2] + OFF[2])};
    std::memcpy(pc.data.data() + (size_t)12 * slot, p, 12);
  }
  for (int k = 0; k < 3; ++k) {
    nav_msgs::Odometry od;
    m.now = 100.02 + 0.02 * k;
    od.header.stamp = Time(m.now);
    od.twist.twist.linear.x = 1.0;
    od.twist.twist.angular.z = 0.05;
    od.pose.pose.orientation.w = 1.0;
    od.pose.pose.position.z = -2.0;
    stub::deliver("/sam/dr/odom", od);
    if (k == 1 && g_landmarks) {
      // the receptor's detections of this ping: PoseArray in base_frame, stamped like the ping
      // (toy_mbes_receptor.cpp:75-105); delivered BEFORE the ping, held by the node until it arrives
      geometry_msgs::PoseArray det;
      det.header.frame_id = "sam/base_link";
      det.header.stamp = Time(m.now);
      const double d[2][3] = {{3.0, 4.0, -16.0}, {-2.0, -6.0, -15.5}};   // (the vehicle is 2 m down: landmarks at -18 / -17.5)
      for (const auto& row : d) {
        geometry_msgs::Pose ps;
        ps.position.x = row[0];
        ps.position.y = row[1];
        ps.position.z = row[2];
        det.poses.push_back(ps);
      }
      stub::deliver("/sam/mbes_detections", det);
      scan.header.stamp = Time(m.now);
    }
    if (k == 1) stub::deliver("/sam/mbes_scan", scan);
    if (k == 2) stub::deliver("/sam/mbes_cloud", pc);
  }
  std_msgs::Bool dive;
  dive.data = false;
  stub::deliver("/dive", dive);
  nav_msgs::Odometry gps;
  gps.pose.pose.position.x = 0.6;
  gps.pose.pose.position.y = -0.4;
  stub::deliver("/sam/dr/gps", gps);
  for (auto& t : m.timers) t.second(TimerEvent());
  // ---- what the node published
  const auto& ods = m.sent.at("/sam/dr/odom_corrected");
  const auto& pas = m.sent.at("/sam/dr/particle_poses");
  if (ods.empty() || pas.empty() || tf2_ros::stub::broadcasts().empty()) {
    std::fprintf(stderr, "nothing published\n");
    std::exit(1);
  }
  const nav_msgs::Odometry& o = *std::static_pointer_cast<nav_msgs::Odometry>(ods.back());
  const geometry_msgs::PoseArray& pa = *std::static_pointer_cast<geometry_msgs::PoseArray>(pas.back());
  const geometry_msgs::TransformStamped& tb = tf2_ros::stub::broadcasts().back();
  std::printf("%.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %zu\n", o.pose.pose.position.x, o.pose.pose.position.y,
              o.pose.pose.position.z, o.pose.pose.orientation.w, o.pose.covariance[0], o.pose.covariance[1],
              o.pose.covariance[4], o.pose.pose.orientation.z, pa.poses.size() * 7);
  std::printf("frames %s %s %s | tf %s %s %.17g %.17g %.17g\n", o.header.frame_id.c_str(), o.child_frame_id.c_str(),
              pa.header.frame_id.c_str(), tb.header.frame_id.c_str(), tb.child_frame_id.c_str(), tb.transform.translation.x,
              tb.transform.translation.y, tb.transform.translation.z);
  std::printf("subs");
  for (const auto& s : m.subs) std::printf(" %s", s.first.c_str());
  std::printf(" | pubs");
  for (const auto& s : m.sent) std::printf(" %s:%u", s.first.c_str(), m.pub_queue[s.first]);
  std::printf(" | timers");
  for (const auto& t : m.timers) std::printf(" %.3f", t.first);
  std::printf(" | spinner %d | node %s\n", m.spinner_threads, m.node_name.c_str());
}

}  // namespace ros
