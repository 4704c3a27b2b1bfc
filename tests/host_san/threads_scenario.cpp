// threads_scenario.cpp -- TEST INFRASTRUCTURE: the stand-in ROS "master" of tests/ros_stubs_cpp driven the way the
// reference node is driven (auv_pf.py:102,107,111,114: one rospy thread per subscription plus the 10 Hz timer thread,
// all entering the filter UNLOCKED -- the race SURVEY 5 names): the roscpp node, compiled unchanged, receives
// odometry, GPS fixes, pings + landmark detections and timer ticks from FOUR threads at once, over the recording engine
// (recording_engine.cpp: unsynchronised state like the real handle).  Under ThreadSanitizer (`make host-tsan`) any
// path by which two callbacks reach the engine or the core's members together is reported; under ASan / UBSan
// (`make host-asan`) every buffer the node hands the ABI is read to its promised extent.
//   usage: <node> <map file>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>

#include <geometry_msgs/PoseArray.h>
#include <nav_msgs/Odometry.h>
#include <ros/ros.h>
#include <sensor_msgs/LaserScan.h>
#include <sensor_msgs/PointCloud2.h>
#include <std_msgs/Bool.h>
#include <tf2_ros/buffer.h>

namespace ros {

void init(int& argc, char** argv, const std::string& name) {
  stub::Master& m = stub::master();
  m.node_name = name;
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s <map file> <landmark file>\n", argv[0]);
    std::exit(2);
  }
  m.params = {{"particle_count", "512"},
              {"init_covariance", "[0.5, 0.5, 0.0, 0.0, 0.0, 0.01]"},
              {"motion_covariance", "[0.001, 0.001, 0.0, 0.0, 0.0, 0.00001]"},
              {"resampling_noise_covariance", "[0.01, 0.01, 0.0, 0.0, 0.0, 0.0001]"},
              {"measurement_std", "1.0"},
              {"map_grid_file", argv[1]},
              {"landmark_map_file", argv[2]},
              {"lm_detect_topic", "/sam/mbes_detections"},
              {"odom_topic", "/sam/dr/odom"},
              {"gps_odom_topic", "/sam/dr/gps"},
              {"mbes_topic", "/sam/mbes_scan"},
              {"mbes_pointcloud_topic", "/sam/mbes_cloud"},
              {"base_frame", "sam/base_link"},
              {"odom_corrected_topic", "/sam/dr/odom_corrected"},
              {"particle_poses_topic", "/sam/dr/particle_poses"}};
  geometry_msgs::Transform id;
  id.rotation.w = 1.0;
  tf2_ros::stub::transforms()[{"map", "sam/odom"}] = id;
  tf2_ros::stub::transforms()[{"map", "utm"}] = id;
  m.now = 100.0;
}

void waitForShutdown() {
  stub::Master& m = stub::master();
  const int B = 64, N = 400;
  std::atomic<int> go{0};
  auto wait = [&] {
    go.fetch_add(1);
    while (go.load() < 4) std::this_thread::yield();
  };
  std::thread odom([&] {   // auv_pf.py:110-111: odometry at the DR rate
    wait();
    for (int k = 0; k < 4 * N; ++k) {
      nav_msgs::Odometry od;
      od.header.stamp = Time(100.02 + 0.005 * k);
      od.twist.twist.linear.x = 1.0;
      od.twist.twist.angular.z = 0.05;
      od.pose.pose.orientation.w = 1.0;
      od.pose.pose.position.z = -2.0;
      stub::deliver("/sam/dr/odom", od);
    }
  });
  std::thread gps([&] {    // :102, :106-107: dive flag and GPS fixes
    wait();
    for (int k = 0; k < N; ++k) {
      std_msgs::Bool dive;
      dive.data = (k / 40) % 2 == 1;
      stub::deliver("/dive", dive);
      nav_msgs::Odometry fix;
      fix.pose.pose.position.x = 0.6 + 0.001 * k;
      fix.pose.pose.position.y = -0.4;
      stub::deliver("/sam/dr/gps", fix);
    }
  });
  std::thread pings([&] {  // the MBES receptor's topics: a ping, then the detections made from it (its real order), and
    wait();                // every third ping as a point cloud
    for (int k = 0; k < N; ++k) {
      const double stamp = 100.02 + 0.02 * k;
      if (k % 3 == 2) {
        sensor_msgs::PointCloud2 pc;
        pc.header.frame_id = "sam/base_link";
        pc.header.stamp = Time(stamp);
        pc.height = 1;
        pc.width = (uint32_t)B;
        pc.point_step = 12;
        pc.row_step = 12 * (uint32_t)B;
        const char* names[3] = {"x", "y", "z"};
        for (int c = 0; c < 3; ++c) {
          sensor_msgs::PointField f;
          f.name = names[c];
          f.offset = 4u * (uint32_t)c;
          f.datatype = sensor_msgs::PointField::FLOAT32;
          f.count = 1;
          pc.fields.push_back(f);
        }
        pc.data.resize((size_t)12 * B);
        for (int b = 0; b < B; ++b) {
          const double a = -1.0 + 2.0 * b / (B - 1), r = 20.0 / std::cos(a);
          const float p[3] = {0.f, (float)(r * std::sin(a)), (float)(-r * std::cos(a))};
          std::memcpy(pc.data.data() + (size_t)12 * b, p, 12);
        }
        stub::deliver("/sam/mbes_cloud", pc);
      } else {
        sensor_msgs::LaserScan scan;
        scan.header.stamp = Time(stamp);
        scan.angle_min = -1.f;
        scan.angle_increment = 2.f / (B - 1);
        scan.range_max = 80.f;
        for (int b = 0; b < B; ++b) scan.ranges.push_back((float)(20.0 / std::cos(-1.0 + 2.0 * b / (B - 1))));
        stub::deliver("/sam/mbes_scan", scan);
      }
      geometry_msgs::PoseArray det;
      det.header.frame_id = "sam/base_link";
      det.header.stamp = Time(stamp);
      for (int j = 0; j < 1 + k % 3; ++j) {
        geometry_msgs::Pose ps;
        ps.position.x = 3.0 + j;
        ps.position.y = -2.0;
        ps.position.z = -16.0;
        det.poses.push_back(ps);
      }
      stub::deliver("/sam/mbes_detections", det);
    }
  });
  std::thread timer([&] {  // :114: the publishing loop
    wait();
    for (int k = 0; k < N; ++k)
      for (auto& t : m.timers) t.second(TimerEvent());
  });
  odom.join();
  gps.join();
  pings.join();
  timer.join();
  const auto& ods = m.sent.at("/sam/dr/odom_corrected");
  const auto& pas = m.sent.at("/sam/dr/particle_poses");
  if ((int)ods.size() != N || (int)pas.size() != N) {
    std::fprintf(stderr, "expected %d publications, got %zu / %zu\n", N, ods.size(), pas.size());
    std::exit(1);
  }
  std::printf("threads_scenario: ok, %zu + %zu publications from four threads\n", ods.size(), pas.size());
}

}  // namespace ros
