#pragma once
#include <geometry_msgs/Pose.h>
namespace nav_msgs {
struct Odometry {
  typedef std::shared_ptr<const Odometry> ConstPtr;
  std_msgs::Header header;
  std::string child_frame_id;
  geometry_msgs::PoseWithCovariance pose;
  geometry_msgs::TwistWithCovariance twist;
};
}  // namespace nav_msgs
